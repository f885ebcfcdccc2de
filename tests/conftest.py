import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def _session_ctx():
    from string_grouper_amd import _native as N
    return N.default_context(0)


@pytest.fixture
def ctx(_session_ctx):
    """Device context of the HIP library; GPU tests fail (not skip) if it cannot be created.  One per session; options a
    test sets (``ctx.set_option``) end with the test."""
    yield _session_ctx
    _session_ctx.reset_options()


@pytest.fixture
def monkeypatch(monkeypatch):
    """The library reads its SG_* switches from the environment once, when a context is created; a test that sets one
    with ``monkeypatch.setenv`` means "for the calls that follow": the variable is also handed to the contexts that exist
    (``Context.set_option``), and when the test ends they re-read the restored environment."""
    from string_grouper_amd import _native as N

    def contexts():
        return [c for c in N._default_ctx.values() if c.h is not None] if N._lib is not None else []

    set_env, del_env = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, *a, **k):
        set_env(name, value, *a, **k)
        if name.startswith("SG_"):
            for c in contexts():
                c.set_option(name, value)

    def delenv(name, *a, **k):
        del_env(name, *a, **k)
        if name.startswith("SG_"):
            for c in contexts():
                c.set_option(name, None)

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield monkeypatch
    monkeypatch.undo()
    for c in contexts():
        c.reset_options()

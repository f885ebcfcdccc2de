"""CPU test: the C-ABI library loads and exports every symbol include/sg_hip.h declares, and the
ctypes binding covers exactly that set.  No compute call is made (there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sg_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+char\s*\*|int)\s*\**\s*(sg_[a-z0-9_]+)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    assert len(names) >= 30
    for must in ("sg_vec_fit", "sg_vec_transform", "sg_postings_build", "sg_spgemm_topn", "sg_topn_zip",
                 "sg_sp_matmul_topn_host", "sg_ctx_stats"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from string_grouper_amd import _native as N
    assert os.path.exists(N.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(N.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in sg_hip.h but not exported: {missing}"


def test_binding_covers_the_header_exactly():
    from string_grouper_amd import _native as N
    assert sorted(N.ABI) == declared_functions()
    N.lib()                                    # resolves every symbol with its prototype
    import re
    header = open(os.path.join(ROOT, "include", "sg_hip.h")).read()
    declared = int(re.search(r"#define\s+SG_ABI_VERSION\s+(\d+)", header).group(1))
    assert N.lib().sg_abi_version() == declared == N.ABI_VERSION     # header, library and binding in lock-step


def test_no_device_is_reported_loudly():
    from string_grouper_amd import _native as N
    if N.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(N.SgHipError) as ei:
        N.Context(0)
    assert "no CPU fallback" in str(ei.value)


def test_error_codes_map_to_reference_exceptions():
    from string_grouper_amd import _native as N
    N.lib()
    with pytest.raises(OverflowError):
        N.check(N.SG_ERR_OVERFLOW)            # what StringGrouper.fit() catches (string_grouper.py:400)
    with pytest.raises(MemoryError):
        N.check(N.SG_ERR_OOM)
    with pytest.raises(ValueError):
        N.check(N.SG_ERR_BADARG)
    with pytest.raises(NotImplementedError):
        N.check(N.SG_ERR_UNSUPPORTED)

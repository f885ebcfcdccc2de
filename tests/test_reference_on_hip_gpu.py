"""The UNMODIFIED reference on top of the HIP seams, on the GPU (INTEGRATION.md Option A): the
reference package -- shipped to the GPU box as the build output ``oracle/_ref/reference_pkg.zip`` (oracle/mount_reference.py;
git-ignored, never in the history) -- is imported with ``sparse_dot_topn`` = ``string_grouper_amd.sparse_dot_topn`` (seam b2,
tests/ref_shims with SG_SHIM_BACKEND=hip) and ``TfidfVectorizer`` = ``string_grouper_amd.vectorizer.TfidfVectorizer`` (seam
b1, by name), and its own 53 unit tests (string_grouper/test/test_string_grouper.py) run: every sp_matmul_topn /
zip_sp_matmul_topn / fit / transform they trigger goes through libsg_hip.so."""
import os
import shutil
import subprocess
import sys
import tempfile
import zipfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARCHIVE = os.path.join(ROOT, "oracle", "_ref", "reference_pkg.zip")

CONFTEST = '''
import sys
sys.path.append({root!r})        # (behind the mounted reference: the repository root holds a drop-in alias package of the same name)
import string_grouper.string_grouper as ref            # the unmodified reference (sparse_dot_topn = the HIP seam, by PYTHONPATH)
import string_grouper_amd.sparse_dot_topn as b2
import string_grouper_amd.vectorizer as b1
assert ref.__file__.startswith({mount!r}), ref.__file__
assert ref.sp_matmul_topn is b2.sp_matmul_topn and ref.zip_sp_matmul_topn is b2.zip_sp_matmul_topn
ref.TfidfVectorizer = b1.TfidfVectorizer               # seam b1, by name (string_grouper.py:6, :306)
CALLS = {{"multiply": 0, "zip": 0, "fit": 0, "transform": 0}}
def _counted(fn, key):
    def wrapper(*a, **k):
        CALLS[key] += 1
        return fn(*a, **k)
    return wrapper
ref.sp_matmul_topn = _counted(b2.sp_matmul_topn, "multiply")
ref.zip_sp_matmul_topn = _counted(b2.zip_sp_matmul_topn, "zip")
b1.TfidfVectorizer.fit = _counted(b1.TfidfVectorizer.fit, "fit")
b1.TfidfVectorizer.transform = _counted(b1.TfidfVectorizer.transform, "transform")
def pytest_sessionfinish(session, exitstatus):
    import json
    from string_grouper_amd import _native as N
    print("\\nHIP-SEAM-CALLS " + json.dumps(CALLS) + " lib=" + N.LIB_PATH)
'''


@pytest.mark.timeout(900)
def test_the_references_own_unit_tests_pass_on_the_hip_seams():
    if not os.path.exists(ARCHIVE):
        pytest.skip("oracle/_ref/reference_pkg.zip was not built (no reference tree where build() ran)")
    d = tempfile.mkdtemp(prefix="sg_ref_on_hip_")
    try:
        with zipfile.ZipFile(ARCHIVE) as z:
            z.extractall(d)
        with open(os.path.join(d, "conftest.py"), "w") as f:
            f.write(CONFTEST.format(root=ROOT, mount=d))
        env = dict(os.environ, SG_SHIM_BACKEND="hip",
                   PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests", "ref_shims"), d]))
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(d, "string_grouper", "test"), "-q", "-s", "-p",
                            "no:cacheprovider", "--rootdir=" + d, "-c", os.devnull], cwd=d, env=env, capture_output=True, text=True,
                           timeout=800)
        tail = r.stdout[-3000:] + r.stderr[-2000:]
        out_dir = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(out_dir):
            with open(os.path.join(out_dir, "reference_on_hip_seams.log"), "w") as f:
                f.write(r.stdout + "\n--- stderr\n" + r.stderr)
        assert r.returncode == 0 and "53 passed" in r.stdout, tail
        calls = [ln for ln in r.stdout.splitlines() if ln.startswith("HIP-SEAM-CALLS")]
        assert calls, tail
        import json
        n = json.loads(calls[0].split(" ", 1)[1].rsplit(" lib=", 1)[0])
        assert n["multiply"] > 20 and n["fit"] > 20 and n["transform"] > 20, n      # the seams really carried the suite
    finally:
        shutil.rmtree(d, ignore_errors=True)

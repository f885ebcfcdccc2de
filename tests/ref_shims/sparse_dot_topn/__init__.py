"""Stand-in for the third-party ``sparse_dot_topn`` wheel (absent in this image) so that the
UNMODIFIED reference package under /root/reference can be imported by tests.

Back-end is selected by the environment variable SG_SHIM_BACKEND:
  "oracle" (default) -> oracle/oracle.py   (scipy product + canonical top-n)
  "port"             -> oracle/sdtn_port.c (C/OpenMP restatement; the timed CPU baseline) + a vectorised zip
  "hip"              -> string_grouper_amd.sparse_dot_topn (the MI355X library; needs a GPU)
Test infrastructure only."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

_backend = os.environ.get("SG_SHIM_BACKEND", "oracle")
if _backend == "oracle":
    from oracle.oracle import sp_matmul_topn, zip_sp_matmul_topn  # noqa: F401
elif _backend == "port":
    from oracle.port import sp_matmul_topn_port as sp_matmul_topn  # noqa: F401
    from oracle.ref_pipeline import zip_port as _zip_port

    def zip_sp_matmul_topn(top_n, C_mats):      # vectorised: the upstream wheel's zip is native code, not a Python loop
        return _zip_port(top_n, C_mats)
elif _backend == "hip":
    from string_grouper_amd.sparse_dot_topn import sp_matmul_topn, zip_sp_matmul_topn  # noqa: F401
else:
    raise ImportError(f"unknown SG_SHIM_BACKEND={_backend!r}")

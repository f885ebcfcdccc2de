"""Pack the reference's Python package into ``oracle/_ref/reference_pkg.zip`` -- TEST INFRASTRUCTURE ONLY.

The reference is pure Python; the GPU box has no ``/root/reference``.  To run the UNMODIFIED reference and its own 53 unit
tests on top of the HIP seams ON THE GPU (INTEGRATION.md Option A; tests/test_reference_on_hip_gpu.py) its package has to
travel: ``__graft_entry__.build()`` calls this where the reference tree is mounted (the builder's container).  The
archive is a BUILD OUTPUT like a compiled ``oracle/_ref/*.so`` would be: ``oracle/_ref/`` is git-ignored (nothing of the
reference enters the history) but not gpurun-ignored (it ships with the snapshot).  Nothing under ``string_grouper_amd/``
reads it."""
from __future__ import annotations

import os
import zipfile

_HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"
ARCHIVE = os.path.join(_HERE, "_ref", "reference_pkg.zip")


def mount(force: bool = False) -> str:
    """Returns the archive's path, or '' when there is no reference tree to pack (the GPU box: the shipped archive, if any,
    is used as it is)."""
    pkg = os.path.join(REFERENCE, "string_grouper")
    if not os.path.isdir(pkg):
        return ARCHIVE if os.path.exists(ARCHIVE) else ""
    files = []
    for base, _, names in os.walk(pkg):
        if "__pycache__" in base:
            continue
        files += [os.path.join(base, n) for n in names if n.endswith(".py")]
    newest = max(os.path.getmtime(f) for f in files)
    if not force and os.path.exists(ARCHIVE) and os.path.getmtime(ARCHIVE) >= newest:
        return ARCHIVE
    os.makedirs(os.path.dirname(ARCHIVE), exist_ok=True)
    with zipfile.ZipFile(ARCHIVE, "w", zipfile.ZIP_DEFLATED) as z:
        for f in sorted(files):
            z.write(f, os.path.relpath(f, REFERENCE))
    return ARCHIVE


if __name__ == "__main__":
    print(mount(force=True) or "no reference tree")

#!/usr/bin/env python3
"""Compile the reference's own range coder (numpyAc/backend/numpyAc_backend.cpp) into oracle/_ref/.

The source is compiled from where it lies under /root/reference with the same recipe the reference
uses itself (torch.utils.cpp_extension.load, numpyAc/numpyAc.py:13-16); nothing is copied into the
repository and oracle/_ref/ is git-ignored.  The resulting extension module travels to the GPU box
with the snapshot and serves as the `"kind": "reference"` check of the range coder.

The reference's octree builder (Octree_python_lib.so) ships as a binary without source and is
therefore NOT rebuildable; it is pinned through golden vectors instead (tests/golden/oct_*.npz).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/numpyAc/backend/numpyAc_backend.cpp"
OUT = os.path.join(HERE, "_ref")


def build(verbose=False):
    if not os.path.exists(SRC):
        return None
    os.makedirs(OUT, exist_ok=True)
    from torch.utils.cpp_extension import load
    return load(name="numpyAc_backend_ref", sources=[SRC], build_directory=OUT, verbose=verbose)


def load_prebuilt():
    """Import the prebuilt module from oracle/_ref (GPU box: no reference tree, no rebuild)."""
    import glob
    import importlib.util
    import torch  # noqa: F401  (the extension links against libtorch)
    cands = sorted(glob.glob(os.path.join(OUT, "numpyAc_backend_ref*.so")))
    if not cands:
        return None
    spec = importlib.util.spec_from_file_location("numpyAc_backend_ref", cands[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    m = build(verbose="-v" in sys.argv)
    print("built" if m is not None else "reference source not present; skipped", file=sys.stderr)

"""Locate and import the reference (robotsorcerer/LevelSetPy) -- TEST INFRASTRUCTURE.

Only usable in the build container, where the read-only reference checkout is
mounted at /root/reference.  Nothing is copied: a symlink named `LevelSetPy`
(the package name the reference's absolute imports expect, e.g.
Grids/create_grid.py:7-8) is created in a temp dir and put on sys.path next to
the NumPy-backed `cupy` stand-in of this directory.
"""
import os
import sys
import tempfile

REF = os.environ.get("HJ_REFERENCE_PATH", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "ExplicitIntegration"))


def load():
    """Returns the imported `LevelSetPy` package (the unmodified reference)."""
    if not available():
        raise RuntimeError("reference checkout not present at %s" % REF)
    os.environ.setdefault("MPLBACKEND", "Agg")
    os.environ.setdefault("HJ_CUPY_WRAP_OOB", "1")
    here = os.path.dirname(os.path.abspath(__file__))
    link_dir = tempfile.mkdtemp(prefix="hj_ref_")
    os.symlink(REF, os.path.join(link_dir, "LevelSetPy"))
    for p in (REF, link_dir, here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import LevelSetPy  # noqa: F401
    return LevelSetPy

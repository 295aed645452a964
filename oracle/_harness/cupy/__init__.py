"""NumPy-backed stand-in for the `cupy` module -- TEST INFRASTRUCTURE ONLY.

The reference (robotsorcerer/LevelSetPy) does an unconditional `import cupy`
in every hot-path module, and CuPy is not installed in the build container.
This module lets the *unmodified* reference run on NumPy so that golden
vectors can be generated from it (tests/golden/make_golden.py).  It is our
own code; nothing here is taken from the reference or from CuPy.

Behaviours emulated (and why):
  * `cupy.ndarray` is a distinct type with `.get()` -> the reference calls
    `.get().item()` on reduced scalars (artificial_diss_glf.py:109) and tests
    `isinstance(alpha, cp.ndarray)` (artificial_diss_glf.py:101).
  * every function returns that type (0-d for scalars), as CuPy does.
  * `cupy.cuda.Device().synchronize()` no-op (add_ghost_extrapolate.py:112).
  * out-of-bounds integer-array indexing WRAPS when HJ_CUPY_WRAP_OOB=1 --
    CuPy's documented behaviour for advanced indexing; NumPy raises.  The
    reference's upwindFirstWENO5a relies on it (upwind_first_weno5a.py:143-145).

It is never imported by the product (levelsetpy_amd) and never ships to the
GPU box as part of any code path: only tests/golden/make_golden.py and the
reference-vs-oracle tests (skipped when /root/reference is absent) use it.
"""
import os
import types
import numpy as _np

_WRAP = os.environ.get("HJ_CUPY_WRAP_OOB", "1") == "1"


class ndarray(_np.ndarray):
    def get(self):
        return _np.asarray(self).view(_np.ndarray)

    def __array_finalize__(self, obj):
        pass

    def __getitem__(self, idx):
        try:
            out = _np.ndarray.__getitem__(self, idx)
        except IndexError:
            if not (_WRAP and isinstance(idx, tuple) and len(idx) == self.ndim):
                raise
            wrapped = []
            for ax, k in enumerate(idx):
                k = _np.asarray(k)
                if k.dtype.kind not in "iu":
                    raise
                wrapped.append(k % self.shape[ax])
            out = _np.ndarray.__getitem__(self, tuple(wrapped))
        return _wrap(out)


def _wrap(x):
    if isinstance(x, ndarray):
        return x
    if isinstance(x, _np.ndarray):
        return x.view(ndarray)
    if isinstance(x, _np.generic):
        return _np.asarray(x).view(ndarray)
    if isinstance(x, tuple):
        return tuple(_wrap(v) for v in x)
    return x


_NO_WRAP = {"ix_", "arange", "isscalar", "ndim", "shape", "size", "dtype",
            "iinfo", "finfo", "result_type", "can_cast"}


def _lift(fn):
    def wrapped(*a, **k):
        return _wrap(fn(*a, **k))
    wrapped.__name__ = getattr(fn, "__name__", "fn")
    return wrapped


def __getattr__(name):
    obj = getattr(_np, name)
    if name in _NO_WRAP or isinstance(obj, type) or not callable(obj):
        return obj
    return _lift(obj)


def asarray(a, dtype=None, order=None):
    return _wrap(_np.asarray(a, dtype=dtype, order=order))


def array(a, *args, **kw):
    return _wrap(_np.array(a, *args, **kw))


def asnumpy(a, *args, **kw):
    return _np.asarray(a).view(_np.ndarray)


class _Device:
    def __init__(self, *a, **k):
        pass

    def synchronize(self):
        pass

    def use(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


cuda = types.SimpleNamespace(Device=_Device)
linalg = types.SimpleNamespace(norm=_lift(_np.linalg.norm))
random = _np.random

"""artificialDissipationGLF (reference ExplicitIntegration/Dissipation/artificial_diss_glf.py:7):
    diss, stepBound = artificialDissipationGLF(t, data, derivL, derivR, schemeData)
This is the split-path form (arbitrary user partialFunc): array expressions on whatever array
type the derivatives come in (device tensors from hj_upwind, or NumPy).  With a native
Hamiltonian the whole term, dissipation included, runs fused in hj_lf_term instead."""
import numpy as np

from .context import is_tensor
from .utilities import isfield, cell

__all__ = ["artificialDissipationGLF", "artificialDissipationLLF", "artificialDissipationLLLF", "glf_device"]


# HJ_SPLIT_KERNELS=0: A/B knob -- the split path's epilogues as stock torch elementwise launches (round-1 behaviour)
import os
SPLIT_KERNELS = os.environ.get("HJ_SPLIT_KERNELS", "1") != "0"


def _amin(a):
    return float(a.min())


def _amax(a):
    return float(a.max())


def glf_device(t, data, derivL, derivR, schemeData, ham=None):
    """artificialDissipationGLF on device tensors in ONE kernel after the partialFunc callbacks
    (hj_lf_split_end): returns (diss, stepBound), or (ydot, stepBound) with ydot = -(ham - diss) when the
    term hands its Hamiltonian array in (term_lax_friedrich.py:122-128).  None if the arrays are not
    device tensors of the grid's ctx."""
    import ctypes as C
    from . import _ffi
    from .context import device_grid, array_dtype_name
    from .spatial import cached_minmax
    grid = schemeData.grid
    dim = grid.dim
    if not all(is_tensor(a) and a.is_cuda for a in list(derivL) + list(derivR)) or (ham is not None and not is_tensor(ham)):
        return None
    dg = device_grid(grid, array_dtype_name(derivL[0]))
    if any(tuple(a.shape) != dg.shape or a.dtype != dg.tdtype for a in list(derivL) + list(derivR)):
        return None
    derivMin, derivMax = cell(dim), cell(dim)
    for i in range(dim):
        mm = cached_minmax(derivL[i], derivR[i])
        if mm is None:
            mm = (min(_amin(derivL[i]), _amin(derivR[i])), max(_amax(derivL[i]), _amax(derivR[i])))
        derivMin[i], derivMax[i] = mm                             # artificial_diss_glf.py:80-88
    arrs, scal, keep = [None] * dim, [0.0] * dim, []
    for i in range(dim):
        alpha = schemeData.partialFunc(t, data, derivMin, derivMax, schemeData, i)   # :98
        if is_tensor(alpha) and alpha.dim() > 0:
            a = dg.to_device(alpha.expand(dg.shape) if tuple(alpha.shape) != dg.shape else alpha)
            keep.append(a)
            arrs[i] = a.data_ptr()
        elif isinstance(alpha, np.ndarray) and alpha.ndim > 0:
            a = dg.to_device(np.broadcast_to(alpha, dg.shape))
            keep.append(a)
            arrs[i] = a.data_ptr()
        else:
            scal[i] = float(alpha)
    dg.bind_stream()
    cL = [dg.to_device(a) for a in derivL]
    cR = [dg.to_device(a) for a in derivR]
    out = dg.empty()
    vp = C.c_void_p * dim
    sb = C.c_double()
    hamc = dg.to_device(ham.reshape(dg.shape)) if ham is not None else None
    _ffi.check(dg.lib.hj_lf_split_end(dg.ctx, vp(*[a.data_ptr() for a in cL]), vp(*[a.data_ptr() for a in cR]),
                                      vp(*arrs), _ffi.darr(scal), dg.ptr(hamc), dg.ptr(out), C.byref(sb), None))
    return out, float(sb.value)


def artificialDissipationGLF(t, data, derivL, derivR, schemeData):
    if not isfield(schemeData, 'grid'):
        raise ValueError('grid is not a structure')               # :65-66
    if not isfield(schemeData, 'partialFunc'):
        raise ValueError('partialFunc is not a structure')        # :67-68
    grid = schemeData.grid
    dim = grid.dim
    if is_tensor(derivL[0]) and derivL[0].is_cuda and SPLIT_KERNELS:
        res = glf_device(t, data, derivL, derivR, schemeData)
        if res is not None:
            return res
    derivMin, derivMax, derivDiff = cell(dim), cell(dim), cell(dim)
    from .spatial import cached_minmax
    for i in range(dim):
        mm = cached_minmax(derivL[i], derivR[i])
        if mm is None:
            mm = (min(_amin(derivL[i]), _amin(derivR[i])), max(_amax(derivL[i]), _amax(derivR[i])))
        derivMin[i], derivMax[i] = mm                             # :80-88
        derivDiff[i] = derivR[i] - derivL[i]                      # :91
    diss = 0
    stepBoundInv = 0
    for i in range(dim):
        alpha = schemeData.partialFunc(t, data, derivMin, derivMax, schemeData, i)   # :98
        diss = diss + (0.5 * derivDiff[i] * alpha)                # :100
        if is_tensor(alpha) or isinstance(alpha, np.ndarray):
            alpha = _amax(alpha)                                  # :101-104
        stepBoundInv += (float(alpha) / float(np.asarray(grid.dx).item(i)))         # :107
    stepBound = float(1 / stepBoundInv)                           # :109
    return diss, stepBound


def _minimum(a, b):
    return a.minimum(b) if is_tensor(a) else np.minimum(a, b)


def _maximum(a, b):
    return a.maximum(b) if is_tensor(a) else np.maximum(a, b)


def _local(t, data, derivL, derivR, schemeData, every_dim_local):
    if not isfield(schemeData, 'grid'):
        raise ValueError('grid is not a structure')
    if not isfield(schemeData, 'partialFunc'):
        raise ValueError('partialFunc is not a structure')
    grid = schemeData.grid
    dim = grid.dim
    nodeMin = [_minimum(derivL[i], derivR[i]) for i in range(dim)]
    nodeMax = [_maximum(derivL[i], derivR[i]) for i in range(dim)]
    if every_dim_local:
        gMin, gMax = nodeMin, nodeMax
    else:
        gMin = [min(_amin(derivL[i]), _amin(derivR[i])) for i in range(dim)]
        gMax = [max(_amax(derivL[i]), _amax(derivR[i])) for i in range(dim)]
    diss = 0
    stepBoundInv = 0
    for i in range(dim):
        derivMin, derivMax = list(gMin), list(gMax)
        derivMin[i], derivMax[i] = nodeMin[i], nodeMax[i]
        alpha = schemeData.partialFunc(t, data, derivMin, derivMax, schemeData, i)
        diss = diss + (0.5 * (derivR[i] - derivL[i]) * alpha)
        stepBoundInv = stepBoundInv + alpha / float(np.asarray(grid.dx).item(i))
    if is_tensor(stepBoundInv) or isinstance(stepBoundInv, np.ndarray):
        stepBoundInv = _amax(stepBoundInv)
    return diss, float(1 / float(stepBoundInv))


def artificialDissipationLLF(t, data, derivL, derivR, schemeData):
    """Local Lax-Friedrichs (reference ExplicitIntegration/Dissipation/diss_local_laxfried.py:8): alpha_i is
    evaluated with the costate of dimension i restricted, at every node, to [min(p_i^-, p_i^+),
    max(p_i^-, p_i^+)] (:108-111) and the global range in the other dimensions (:84-99);
    diss = sum_i (p_i^+ - p_i^-)/2 alpha_i (:117).  stepBound = 1 / max_x sum_i alpha_i(x)/dx_i: the
    shipped code calls .item() on that array (:121), which only works for a scalar alpha -- the
    reduction over the grid is what the level-set toolbox it ports does."""
    return _local(t, data, derivL, derivR, schemeData, False)


def artificialDissipationLLLF(t, data, derivL, derivR, schemeData):
    """Local local Lax-Friedrichs (reference .../diss_localsq_laxfried.py:7): per-node costate ranges in
    EVERY dimension (:87-90; the shipped code applies Python's scalar min/max to the arrays there, which
    raises), diss and stepBound as for LLF (:99-104, reduced over the grid)."""
    return _local(t, data, derivL, derivR, schemeData, True)

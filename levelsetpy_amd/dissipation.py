"""artificialDissipationGLF (reference ExplicitIntegration/Dissipation/artificial_diss_glf.py:7):
    diss, stepBound = artificialDissipationGLF(t, data, derivL, derivR, schemeData)
This is the split-path form (arbitrary user partialFunc): array expressions on whatever array
type the derivatives come in (device tensors from hj_upwind, or NumPy).  With a native
Hamiltonian the whole term, dissipation included, runs fused in hj_lf_term instead."""
import numpy as np

from .context import is_tensor
from .utilities import isfield, cell

__all__ = ["artificialDissipationGLF"]


def _amin(a):
    return float(a.min())


def _amax(a):
    return float(a.max())


def artificialDissipationGLF(t, data, derivL, derivR, schemeData):
    if not isfield(schemeData, 'grid'):
        raise ValueError('grid is not a structure')               # :65-66
    if not isfield(schemeData, 'partialFunc'):
        raise ValueError('partialFunc is not a structure')        # :67-68
    grid = schemeData.grid
    dim = grid.dim
    derivMin, derivMax, derivDiff = cell(dim), cell(dim), cell(dim)
    cache = None
    dgs = grid.__dict__.get("_hj_device") or {}
    for dg in dgs.values():
        cache = getattr(dg, "minmax", None) or cache
    for i in range(dim):
        mm = None
        if cache is not None and is_tensor(derivL[i]):
            mm = cache.get((derivL[i].data_ptr(), derivR[i].data_ptr()))
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

"""artificialDissipationGLF (reference ExplicitIntegration/Dissipation/artificial_diss_glf.py:7):
    diss, stepBound = artificialDissipationGLF(t, data, derivL, derivR, schemeData)
This is the split-path form (arbitrary user partialFunc): array expressions on whatever array
type the derivatives come in (device tensors from hj_upwind, or NumPy).  With a native
Hamiltonian the whole term, dissipation included, runs fused in hj_lf_term instead."""
import numpy as np

from .context import is_tensor
from .utilities import isfield, cell

__all__ = ["artificialDissipationGLF", "artificialDissipationLLF", "artificialDissipationLLLF"]


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

"""computeGradients (reference ValueFuncs/compute_gradients.py:11): costate estimate of a stored value
function by upwinded differences in every grid direction -- the step after the solve.

    derivC, derivL, derivR = computeGradients(g, data, dims=None, derivFunc=None)

The derivatives come from the same HIP kernels as the solver's (hj_upwind through spatial.py).  `data`
may be a NumPy array or a device tensor, with or without a leading time axis (this package stores
time FIRST, see hji_solver.py).

Deviations from the shipped reference: it restores NaN/inf through `nanInds`/`infInds`, which are never
defined (:75,78: NameError on every call), and returns only the LAST dimension's one-sided derivatives
(:81).  Here the masks are taken before the replacement, the replacement works on a copy (the
caller's array is not modified), and derivL/derivR are lists over dimensions like derivC.
"""
import numpy as np

from .context import is_tensor
from .spatial import upwindFirstWENO5, upwind_all_dims
from .utilities import cell, error

__all__ = ["computeGradients"]


def computeGradients(g, data, dims=None, derivFunc=None):
    if dims is None or (not is_tensor(dims) and not np.any(dims)):
        dims = np.ones(g.dim, dtype=bool)                        # :30-32
    dims = np.asarray(dims).astype(bool).ravel()
    if dims.size != g.dim:
        error('dims must have one entry per grid dimension')
    if derivFunc is None:
        derivFunc = upwindFirstWENO5                             # :34-36
    nd = data.dim() if is_tensor(data) else np.ndim(data)
    if nd == g.dim:
        tau_length = 1
    elif nd == g.dim + 1:
        tau_length = int(data.shape[0])
    else:
        error('Dimensions of input data and grid don\'t match!')  # :47
    # NaN / inf (usually from time-to-reach functions) would poison whole stencils: replace them by a
    # large number for the differences and put them back afterwards (:49-54,74-78)
    numInfty = 1e6
    if is_tensor(data):
        import torch
        nanInds, infInds = torch.isnan(data), torch.isinf(data)
        work = torch.where(nanInds | infInds, torch.full_like(data, numInfty), data)
    else:
        data = np.asarray(data, dtype=np.float64)
        nanInds, infInds = np.isnan(data), np.isinf(data)
        work = np.where(nanInds | infInds, numInfty, data)
    derivC, derivL, derivR = cell(g.dim), cell(g.dim), cell(g.dim)
    # a device tensor, every dimension, one of this package's derivative functions: ONE launch for all dimensions
    # (hj_lf_split_begin, round 4) instead of one per dimension
    both = upwind_all_dims(derivFunc, g, work) if (is_tensor(work) and nd == g.dim and dims.all()) else None
    # the masks are only applied where there is something to restore (one reduction instead of six masked assignments)
    masked = bool((nanInds | infInds).any())
    for i in range(g.dim):
        if not dims[i]:
            continue
        if both is not None:
            L, R = both[0][i], both[1][i]
        elif tau_length == 1 and nd == g.dim:
            L, R = derivFunc(g, work, i)                         # :63
        else:
            pairs = [derivFunc(g, work[t], i) for t in range(tau_length)]      # :69-71
            stack = (lambda xs: __import__("torch").stack(xs)) if is_tensor(pairs[0][0]) else np.stack
            L, R = stack([p[0] for p in pairs]), stack([p[1] for p in pairs])
        C = 0.5 * (L + R)                                        # :66
        if masked:
            for arr in (C, L, R):
                arr[nanInds] = float('nan')
                arr[infInds] = float('inf')
        derivC[i], derivL[i], derivR[i] = C, L, R
    return derivC, derivL, derivR

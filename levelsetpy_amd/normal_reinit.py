"""termNormal and termReinit (reference ExplicitIntegration/Term/term_normal.py:7, term_reinit.py:7): two more
schemeFuncs on the same upwind derivative kernels,
    ydot, stepBound, schemeData = termNormal(t, y, schemeData)     ydot = -a |grad phi|   (motion along the normal)
    ydot, stepBound, schemeData = termReinit(t, y, schemeData)     ydot = -S(phi_0)(|grad phi| - 1)
with Godunov's choice between the one-sided derivatives.  Same protocol as termLaxFriedrichs, so odeCFL1/2/3
integrate them; `y` and the arrays in schemeData may be NumPy arrays or device tensors (the derivatives of all
dimensions then come from ONE native call, hj_lf_split_begin; the remaining array expressions run on the
arrays it returns).

Deviations from the shipped reference, neither of which runs: both combine their upwinding masks with
Python's scalar `and` / `or` (term_normal.py:142-143, term_reinit.py:183-192: "truth value of an array is
ambiguous"), termNormal reads the speed from a field `forcing` it never asked for (:103-105), and termReinit
calls an `isNearInterface` that exists nowhere in the reference (:290) and zeroes the converging-flow
arrival time it is about to compare (`s[conv] *= ...` on zeros, :196).  Implemented here is what their
docstrings and the toolbox they port (Mitchell's ToolboxLS termNormal.m / termReinit.m) describe; the
formula lines are cited.  Parity is therefore UNPINNED for both (no reference output exists); they are
checked against oracle.term_normal / oracle.term_reinit (same formulas on the oracle's reference-pinned
derivatives) and by behaviour (unit-speed growth of a circle; |grad phi| -> 1 with the zero level set kept).
"""
import numpy as np

from .context import is_tensor
from .spatial import upwind_all_dims
from .utilities import isfield, iscell, error, eps

__all__ = ["termNormal", "termReinit"]


def _deriv_func(sd):
    if isfield(sd, 'derivFunc'):
        return sd.derivFunc
    if isfield(sd, 'CoStateCalc'):
        return sd.CoStateCalc
    return None


def _like(a, proto, shape):
    """`a` (scalar / NumPy array / tensor) as an array of `proto`'s kind and dtype with grid shape."""
    if is_tensor(proto):
        import torch
        if is_tensor(a):
            return a.to(device=proto.device, dtype=proto.dtype).reshape(shape)
        return torch.as_tensor(np.broadcast_to(np.asarray(a, dtype=np.float64), shape).copy(), device=proto.device).to(proto.dtype)
    if is_tensor(a):
        a = a.detach().cpu().numpy()
    return np.broadcast_to(np.asarray(a, dtype=np.float64), shape)


def _derivs(derivFunc, grid, data):
    both = upwind_all_dims(derivFunc, grid, data)
    if both is not None:
        return both
    pairs = [derivFunc(grid, data, i) for i in range(grid.dim)]
    return [p[0] for p in pairs], [p[1] for p in pairs]


def _amax(a):
    return float(a.max())


def _sqrt(a):
    return a.sqrt() if is_tensor(a) else np.sqrt(a)


def _sign(a):
    return a.sign() if is_tensor(a) else np.sign(a)


def termNormal(t, y, schemeData):
    thisSchemeData = schemeData[0] if iscell(schemeData) else schemeData
    assert isfield(thisSchemeData, 'grid'), "grid not in schemeData"
    assert _deriv_func(thisSchemeData) is not None, "derivFunc not in schemeData"
    assert isfield(thisSchemeData, 'speed'), "speed not in schemeData"
    grid = thisSchemeData.grid
    y0 = y[0] if iscell(y) else y
    data = y0.reshape(grid.shape)
    speed = thisSchemeData.speed
    if callable(speed):
        speed = speed(t, data, thisSchemeData)                              # term_normal.py:106-125
    elif not (np.isscalar(speed) or is_tensor(speed) or isinstance(speed, np.ndarray)):
        error('schemeData.speed must be a scalar, array or function handle')   # :126-127
    speed = _like(speed, data, grid.shape)
    derivL, derivR = _derivs(_deriv_func(thisSchemeData), grid, data)
    magnitude = 0
    stepBoundInv = 0
    for i in range(grid.dim):
        prodL, prodR = speed * derivL[i], speed * derivR[i]                  # :135-136
        magL, magR = abs(prodL), abs(prodR)
        # either both sides agree in sign, or the characteristics converge: the larger magnitude wins (:140-143)
        conv = (prodL >= 0) & (prodR <= 0)
        flowL = ((prodL >= 0) & (prodR >= 0)) | (conv & (magL >= magR))
        flowR = ((prodL <= 0) & (prodR <= 0)) | (conv & (magL < magR))
        # diverging characteristics contribute a zero gradient (:145-147)
        magnitude = magnitude + (derivL[i] ** 2 * flowL + derivR[i] ** 2 * flowR)     # :150
        effectiveVelocity = magL * flowL + magR * flowR                      # :153
        stepBoundInv = stepBoundInv + effectiveVelocity / float(np.asarray(grid.dx).item(i))   # :154-155
    magnitude = _sqrt(magnitude)                                             # :158
    delta = speed * magnitude                                                # :159
    nz = magnitude > 0                                                       # :162-164
    if bool(nz.any()):
        stepBound = float(1 / _amax(stepBoundInv[nz] / magnitude[nz]))
    else:
        stepBound = float('inf')
    return (-delta).reshape(-1, 1), stepBound, schemeData                    # :167


def _near_interface(initial):
    """Nodes with a neighbour (in any dimension) on the other side of the zero level set, or on it
    (ToolboxLS isNearInterface; the reference calls it at term_reinit.py:290 without defining it)."""
    sg = _sign(initial)
    near = sg == 0
    for d in range(initial.ndim if not is_tensor(initial) else initial.dim()):
        n = initial.shape[d]
        lo = [slice(None)] * len(initial.shape)
        hi = [slice(None)] * len(initial.shape)
        lo[d], hi[d] = slice(0, n - 1), slice(1, n)
        diff = sg[tuple(lo)] != sg[tuple(hi)]
        near = near.clone() if is_tensor(near) else near.copy()
        near[tuple(lo)] |= diff
        near[tuple(hi)] |= diff
    return near


def termReinit(t, y, schemeData):
    robust_small_epsilon = 1e6 * eps                                         # term_reinit.py:128
    thisSchemeData = schemeData[0] if iscell(schemeData) else schemeData
    assert isfield(thisSchemeData, 'grid'), "grid not in schemeData"
    assert _deriv_func(thisSchemeData) is not None, "derivFunc not in schemeData"
    assert isfield(thisSchemeData, 'initial'), "initial not in schemeData"
    grid = thisSchemeData.grid
    y0 = y[0] if iscell(y) else y
    data = y0.reshape(grid.shape)
    order = thisSchemeData.subcell_fix_order if isfield(thisSchemeData, 'subcell_fix_order') else 1   # :146-157
    if order not in (0, 1):
        error('Reinit subcell fix order of accuracy %s not supported' % order)
    initial = _like(thisSchemeData.initial, data, grid.shape)
    dxs = [float(v) for v in np.asarray(grid.dx).ravel()]
    if order:
        S = _sign(initial)                                                   # :161
    else:
        S = initial / _sqrt(initial ** 2 + max(dxs) ** 2)                    # smearedSign, O&F (7.5)  :163-168,318-325
    derivL, derivR = _derivs(_deriv_func(thisSchemeData), grid, data)
    deriv = [None] * grid.dim
    for i in range(grid.dim):
        sL, sR = S * derivL[i], S * derivR[i]
        flowL = (sR <= 0) & (sL <= 0)                                        # :183  information arrives from the right
        flowR = (sR >= 0) & (sL >= 0)                                        # :185  ... from the left
        flows = (sR < 0) & (sL > 0)                                          # :190  converging: which side arrives first?
        den = derivR[i] - derivL[i]
        den = den + (den == 0)                                               # only read where `flows` holds (den != 0 there)
        s = S * (abs(derivR[i]) - abs(derivL[i])) / den                      # :192-196 (O&F / Fedkiw et al. A.3)
        flowL = flowL | (flows & (s < 0))                                    # :199-200
        flowR = flowR | (flows & (s >= 0))
        deriv[i] = derivL[i] * flowR + derivR[i] * flowL                     # :201
    mag = 0
    for i in range(grid.dim):
        mag = mag + deriv[i] ** 2                                            # :203-205
    mag = _sqrt(mag)
    mag = mag.clamp_min(eps) if is_tensor(mag) else np.maximum(mag, eps)     # :206
    delta = -S                                                               # :208
    stepBoundInv = 0.0
    for i in range(grid.dim):
        v = S * deriv[i] / mag                                               # :213
        delta = delta + v * deriv[i]                                         # :215
        stepBoundInv += _amax(abs(v)) / dxs[i]                               # :217
    if order == 1:
        # Russo & Smereka's subcell fix, robust distance (17): long differences, short ones where they are larger
        denom = 0
        nd = grid.dim
        for d in range(nd):
            n = initial.shape[d]
            sl = lambda a, b: tuple([slice(None)] * d + [slice(a, b)] + [slice(None)] * (nd - d - 1))   # noqa: E731
            dx_inv = 1.0 / dxs[d]
            diff2 = (initial * 0)
            diff2[sl(1, n - 1)] = (0.5 * dx_inv * (initial[sl(2, n)] - initial[sl(0, n - 2)])) ** 2      # :262-264 interior
            diff2[sl(0, 1)] = (dx_inv * (initial[sl(1, 2)] - initial[sl(0, 1)])) ** 2                    # short at the edges
            diff2[sl(n - 1, n)] = (dx_inv * (initial[sl(n - 1, n)] - initial[sl(n - 2, n - 1)])) ** 2
            short2 = (dx_inv * (initial[sl(1, n)] - initial[sl(0, n - 1)])) ** 2                         # :266-270
            if is_tensor(diff2):
                import torch
                diff2[sl(0, n - 1)] = torch.maximum(diff2[sl(0, n - 1)], short2)                          # :272-273
                diff2[sl(1, n)] = torch.maximum(diff2[sl(1, n)], short2)
                diff2 = diff2.clamp_min(robust_small_epsilon ** 2)                                       # :274
            else:
                diff2[sl(0, n - 1)] = np.maximum(diff2[sl(0, n - 1)], short2)
                diff2[sl(1, n)] = np.maximum(diff2[sl(1, n)], short2)
                diff2 = np.maximum(diff2, robust_small_epsilon ** 2)
            denom = denom + diff2                                            # :276
        D = initial / _sqrt(denom)                                           # :283-286
        near = _near_interface(initial)                                      # :290
        delta = delta * (~near) + (S * abs(data) - D) / max(dxs) * near      # :299
    stepBound = float(1 / stepBoundInv) if stepBoundInv > 0 else float('inf')   # :305
    return (-delta).reshape(-1, 1), stepBound, schemeData                    # :308

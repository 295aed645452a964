"""termNormal and termReinit (reference ExplicitIntegration/Term/term_normal.py:7, term_reinit.py:7): two more
schemeFuncs on the same upwind derivative kernels,
    ydot, stepBound, schemeData = termNormal(t, y, schemeData)     ydot = -a |grad phi|   (motion along the normal)
    ydot, stepBound, schemeData = termReinit(t, y, schemeData)     ydot = -S(phi_0)(|grad phi| - 1)
with Godunov's choice between the one-sided derivatives.  Same protocol as termLaxFriedrichs, so odeCFL1/2/3
integrate them; `y` and the arrays in schemeData may be NumPy arrays or device tensors (the derivatives of all
dimensions then come from ONE native call, hj_lf_split_begin; the remaining array expressions run on the
arrays it returns).

Deviations from the shipped reference, neither of which runs: both combine their upwinding masks with
Python's scalar `and` / `or` (term_normal.py:156-157, term_reinit.py:190-200: "truth value of an array is
ambiguous"), termNormal reads the speed from a field `forcing` it never asked for (:107-110), and termReinit
calls an `isNearInterface` that exists nowhere in the reference (:292) and zeroes the converging-flow
arrival time it is about to compare (`s[conv] *= ...` on zeros, :203-204).  Implemented here is what their
docstrings and the toolbox they port (Mitchell's ToolboxLS termNormal.m / termReinit.m) describe; the
formula lines are cited.  Parity is therefore UNPINNED for both (no reference output exists); they are
checked against oracle.term_normal / oracle.term_reinit (same formulas on the oracle's reference-pinned
derivatives) and by behaviour (unit-speed growth of a circle; |grad phi| -> 1 with the zero level set kept).
"""
import ctypes as C

import numpy as np

from . import _ffi
from .context import is_tensor, device_grid, array_dtype_name
from .spatial import upwind_all_dims, scheme_id_of
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


def _fused(derivFunc, grid, data):
    """(DeviceGrid, scheme id, device phi) when `derivFunc` is one of this package's derivative functions: the
    whole term is then ONE kernel launch (hj_term_normal / hj_term_reinit, csrc/hj_terms.h; round 3).  NumPy data
    goes over PCIe and the result comes back as NumPy, like every NumPy-in call of this package."""
    sid = scheme_id_of(derivFunc)
    if sid is None:
        return None
    dg = device_grid(grid, array_dtype_name(data))
    if tuple(data.shape) != dg.shape:
        error('data parameter does not agree in array size with grid')
    dg.bind_stream()
    return dg, sid, dg.to_device(data)


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
        speed = speed(t, data, thisSchemeData)                              # term_normal.py:117-131
    elif not (np.isscalar(speed) or is_tensor(speed) or isinstance(speed, np.ndarray)):
        error('schemeData.speed must be a scalar, array or function handle')   # :132-133
    fused = _fused(_deriv_func(thisSchemeData), grid, data)
    if fused is not None:
        dg, sid, phi = fused
        scalar = np.isscalar(speed) or (isinstance(speed, np.ndarray) and speed.ndim == 0)
        arr = None if scalar else dg.to_device(_like(speed, data, grid.shape))
        out, sb = dg.empty(), C.c_double()
        _ffi.check(dg.lib.hj_term_normal(dg.ctx, sid, dg.ptr(phi), dg.ptr(arr), float(speed) if scalar else 0.0,
                                         dg.ptr(out), C.byref(sb)))
        return dg.like(out, y0, (-1, 1)), float(sb.value), schemeData
    # a foreign derivFunc: its derivatives, then array expressions on whatever kind of array it returns
    speed = _like(speed, data, grid.shape)
    derivL, derivR = _derivs(_deriv_func(thisSchemeData), grid, data)
    magnitude = 0
    stepBoundInv = 0
    for i in range(grid.dim):
        prodL, prodR = speed * derivL[i], speed * derivR[i]                  # term_normal.py:148-151
        magL, magR = abs(prodL), abs(prodR)
        # either both sides agree in sign, or the characteristics converge: the larger magnitude wins (:156-157)
        conv = (prodL >= 0) & (prodR <= 0)
        flowL = ((prodL >= 0) & (prodR >= 0)) | (conv & (magL >= magR))
        flowR = ((prodL <= 0) & (prodR <= 0)) | (conv & (magL < magR))
        # diverging characteristics contribute a zero gradient (:159-161)
        magnitude = magnitude + (derivL[i] ** 2 * flowL + derivR[i] ** 2 * flowR)     # :164
        effectiveVelocity = magL * flowL + magR * flowR                      # :181
        stepBoundInv = stepBoundInv + effectiveVelocity / float(np.asarray(grid.dx).item(i))   # :168-169
    magnitude = _sqrt(magnitude)                                             # :172
    delta = speed * magnitude                                                # :173
    nz = magnitude > 0                                                       # :176-178
    if bool(nz.any()):
        stepBound = float(1 / _amax(stepBoundInv[nz] / magnitude[nz]))
    else:
        stepBound = float('inf')
    return (-delta).reshape(-1, 1), stepBound, schemeData                    # :181


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
    robust_small_epsilon = 1e6 * eps                                         # term_reinit.py:129
    thisSchemeData = schemeData[0] if iscell(schemeData) else schemeData
    assert isfield(thisSchemeData, 'grid'), "grid not in schemeData"
    assert _deriv_func(thisSchemeData) is not None, "derivFunc not in schemeData"
    assert isfield(thisSchemeData, 'initial'), "initial not in schemeData"
    grid = thisSchemeData.grid
    y0 = y[0] if iscell(y) else y
    data = y0.reshape(grid.shape)
    order = thisSchemeData.subcell_fix_order if isfield(thisSchemeData, 'subcell_fix_order') else 1   # :149-161
    if order not in (0, 1):
        error('Reinit subcell fix order of accuracy %s not supported' % order)
    fused = _fused(_deriv_func(thisSchemeData), grid, data)
    if fused is not None:
        dg, sid, phi = fused
        init = dg.to_device(_like(thisSchemeData.initial, data, grid.shape))
        out, sb = dg.empty(), C.c_double()
        _ffi.check(dg.lib.hj_term_reinit(dg.ctx, sid, dg.ptr(phi), dg.ptr(init), int(order), dg.ptr(out), C.byref(sb)))
        return dg.like(out, y0, (-1, 1)), float(sb.value), schemeData
    # a foreign derivFunc: its derivatives, then array expressions on whatever kind of array it returns
    initial = _like(thisSchemeData.initial, data, grid.shape)
    dxs = [float(v) for v in np.asarray(grid.dx).ravel()]
    if order:
        S = _sign(initial)                                                   # :166
    else:
        S = initial / _sqrt(initial ** 2 + max(dxs) ** 2)                    # smearedSign, O&F (7.5)  :168-174
    derivL, derivR = _derivs(_deriv_func(thisSchemeData), grid, data)
    deriv = [None] * grid.dim
    for i in range(grid.dim):
        sL, sR = S * derivL[i], S * derivR[i]
        flowL = (sR <= 0) & (sL <= 0)                                        # :190  information arrives from the right
        flowR = (sR >= 0) & (sL >= 0)                                        # :193  ... from the left
        flows = (sR < 0) & (sL > 0)                                          # :200  converging: which side arrives first?
        den = derivR[i] - derivL[i]
        den = den + (den == 0)                                               # only read where `flows` holds (den != 0 there)
        s = S * (abs(derivR[i]) - abs(derivL[i])) / den                      # :202-204 (O&F / Fedkiw et al. A.3)
        flowL = flowL | (flows & (s < 0))                                    # :208-209
        flowR = flowR | (flows & (s >= 0))
        deriv[i] = derivL[i] * flowR + derivR[i] * flowL                     # :211
    mag = 0
    for i in range(grid.dim):
        mag = mag + deriv[i] ** 2                                            # :214-216
    mag = _sqrt(mag)
    mag = mag.clamp_min(eps) if is_tensor(mag) else np.maximum(mag, eps)     # :232
    delta = -S                                                               # :220
    stepBoundInv = 0.0
    for i in range(grid.dim):
        v = S * deriv[i] / mag                                               # :226
        delta = delta + v * deriv[i]                                         # :228
        stepBoundInv += _amax(abs(v)) / dxs[i]                               # :232
    if order == 1:
        # Russo & Smereka's subcell fix, robust distance (17): long differences, short ones where they are larger
        denom = 0
        nd = grid.dim
        for d in range(nd):
            n = initial.shape[d]
            sl = lambda a, b: tuple([slice(None)] * d + [slice(a, b)] + [slice(None)] * (nd - d - 1))   # noqa: E731
            dx_inv = 1.0 / dxs[d]
            diff2 = (initial * 0)
            diff2[sl(1, n - 1)] = (0.5 * dx_inv * (initial[sl(2, n)] - initial[sl(0, n - 2)])) ** 2      # :259-266 interior
            diff2[sl(0, 1)] = (dx_inv * (initial[sl(1, 2)] - initial[sl(0, 1)])) ** 2                    # short at the edges
            diff2[sl(n - 1, n)] = (dx_inv * (initial[sl(n - 1, n)] - initial[sl(n - 2, n - 1)])) ** 2
            short2 = (dx_inv * (initial[sl(1, n)] - initial[sl(0, n - 1)])) ** 2                         # :267-272
            if is_tensor(diff2):
                import torch
                diff2[sl(0, n - 1)] = torch.maximum(diff2[sl(0, n - 1)], short2)                          # :273-275
                diff2[sl(1, n)] = torch.maximum(diff2[sl(1, n)], short2)
                diff2 = diff2.clamp_min(robust_small_epsilon ** 2)                                       # :278
            else:
                diff2[sl(0, n - 1)] = np.maximum(diff2[sl(0, n - 1)], short2)
                diff2[sl(1, n)] = np.maximum(diff2[sl(1, n)], short2)
                diff2 = np.maximum(diff2, robust_small_epsilon ** 2)
            denom = denom + diff2                                            # :278
        D = initial / _sqrt(denom)                                           # :284-288
        near = _near_interface(initial)                                      # :292
        delta = delta * (~near) + (S * abs(data) - D) / max(dxs) * near      # :300
    stepBound = float(1 / stepBoundInv) if stepBoundInv > 0 else float('inf')   # :309
    return (-delta).reshape(-1, 1), stepBound, schemeData                    # :312

"""termConvection (reference ExplicitIntegration/Term/term_convection.py:7): the schemeFunc of pure motion
by an externally given velocity field,
    ydot, stepBound, schemeData = termConvection(t, y, schemeData)        ydot = -V . grad(phi)
with grad(phi) upwinded by the sign of every velocity component.  Same protocol as termLaxFriedrichs,
so odeCFL1/2/3 integrate it.  schemeData fields: grid, velocity (a list with one entry per dimension,
each a scalar or an array of grid.shape, or a callable (t, data, schemeData) -> such a list),
derivFunc (CoStateCalc is accepted too).

The one-sided derivatives come from the HIP upwind kernels (hj_upwind); the remaining array expressions
run on the device arrays those return.

Deviations from the shipped reference, which does not run: its accumulator is an integer array
(`zeros(size(data))`, :154), so `delta += deriv * v` (:170) raises a casting error under NumPy and CuPy
alike, and the upwinding block sits OUTSIDE the loop over dimensions (:157-172), so only the last
dimension's term would ever be added.  Here every dimension contributes, as the docstring
(`-V . grad phi`) and the toolbox it ports say.  Parity is therefore UNPINNED for this function: it is
checked against oracle.term_convection (same formulas on the oracle's reference-pinned derivatives).
"""
import ctypes as C

import numpy as np

from . import _ffi
from .context import is_tensor, device_grid, array_dtype_name
from .spatial import scheme_id_of
from .utilities import isfield, iscell, error

__all__ = ["termConvection"]


def termConvection(t, y, schemeData):
    thisSchemeData = schemeData[0] if iscell(schemeData) else schemeData
    assert isfield(thisSchemeData, 'grid'), "grid not in schemeData"
    assert isfield(thisSchemeData, 'velocity'), "velocity not in schemeData"
    derivFunc = (thisSchemeData.derivFunc if isfield(thisSchemeData, 'derivFunc')
                 else (thisSchemeData.CoStateCalc if isfield(thisSchemeData, 'CoStateCalc') else None))
    assert derivFunc is not None, "derivFunc not in schemeData"
    grid = thisSchemeData.grid
    y0 = y[0] if iscell(y) else y
    data = y0.reshape(grid.shape)
    velocity = thisSchemeData.velocity
    if callable(velocity):
        velocity = velocity(t, data, thisSchemeData)                       # :124-147
    if not isinstance(velocity, (list, tuple)) or len(velocity) != grid.dim:
        error('schemeData.velocity must be a cell vector or a function handle')   # :149-150
    sid = scheme_id_of(derivFunc)
    if sid is not None:
        # one of this package's derivative functions: the whole term is ONE kernel launch (hj_term_convection,
        # csrc/hj_terms.h; round 3)
        dg = device_grid(grid, array_dtype_name(data))
        if tuple(data.shape) != dg.shape:
            error('data parameter does not agree in array size with grid')
        dg.bind_stream()
        phi = dg.to_device(data)
        arrs, scal = [], []
        for v in velocity:
            if np.isscalar(v) or (isinstance(v, np.ndarray) and v.ndim == 0) or (is_tensor(v) and v.dim() == 0):
                arrs.append(None)
                scal.append(float(v))          # (a 0-dim tensor is a scalar speed as well: ADVICE r03)
            else:
                if is_tensor(v):
                    # broadcast-shaped components (e.g. (N0,1,1)) are accepted like NumPy ones; to_device materialises them
                    import torch
                    a = v.reshape(grid.shape) if v.numel() == int(np.prod(grid.shape)) else torch.broadcast_to(v, grid.shape)
                else:
                    a = np.broadcast_to(np.asarray(v, dtype=np.float64), grid.shape)
                arrs.append(dg.to_device(a))
                scal.append(0.0)
        out, sb = dg.empty(), C.c_double()
        vp = (C.c_void_p * dg.dim)(*[(a.data_ptr() if a is not None else None) for a in arrs])
        _ffi.check(dg.lib.hj_term_convection(dg.ctx, sid, dg.ptr(phi), vp, _ffi.darr(scal), dg.ptr(out), C.byref(sb)))
        return dg.like(out, y0, (-1, 1)), float(sb.value), schemeData
    delta = 0
    stepBoundInv = 0.0
    for i in range(grid.dim):
        derivL, derivR = derivFunc(grid, data, i)                          # :158
        v = velocity[i]
        if np.isscalar(v):
            v = float(v)
            deriv = derivL if v > 0 else (derivR if v < 0 else 0 * derivL)  # :161-167
            vmax = abs(v)
        else:
            if is_tensor(derivL) and not is_tensor(v):
                import torch
                v = torch.as_tensor(np.asarray(v, dtype=np.float64), device=derivL.device).to(derivL.dtype)
            elif not is_tensor(derivL) and is_tensor(v):
                v = v.detach().cpu().numpy()
            v = v.reshape(grid.shape)
            deriv = derivL * (v > 0) + derivR * (v < 0)                    # where v == 0 the derivative is irrelevant
            vmax = float(abs(v).max())
        delta = delta + deriv * v                                          # :170
        stepBoundInv += vmax / float(np.asarray(grid.dx).item(i))          # :175
    if stepBoundInv == 0.0:
        stepBound = float('inf')
    else:
        stepBound = float(1 / stepBoundInv)                                # :177
    ydot = (-delta).reshape(-1, 1)                                         # (N,1) like the reference (:180)
    return ydot, stepBound, schemeData

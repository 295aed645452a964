"""termLaxFriedrichs / termRestrictUpdate behind the reference's schemeFunc protocol
    ydot, stepBound, schemeData = schemeFunc(t, y, schemeData)
(reference ExplicitIntegration/Term/term_lax_friedrich.py:8, term_restrict_update.py:8).

Fused path: when hamFunc/partialFunc are the methods of one of dynamics.py's systems, dissFunc is
this package's artificialDissipationGLF and the derivative function is one of spatial.py's, the
whole term is ONE HIP kernel (hj_lf_term).  Anything else takes the split path: per-dimension
hj_upwind kernels, then the user's callbacks on arrays, exactly as the reference sequences them.
"""
import copy
import ctypes as C

import numpy as np

from . import _ffi
from .context import device_grid, array_dtype_name, is_tensor
from . import dissipation as _diss
from .dissipation import artificialDissipationGLF, artificialDissipationLLF, artificialDissipationLLLF, glf_device
from .dynamics import native_of
from .spatial import scheme_id_of, upwind_all_dims
from .utilities import isfield, iscell

__all__ = ["termLaxFriedrichs", "termRestrictUpdate"]


def _deriv_func(sd):
    # the reference's term reads CoStateCalc (term_lax_friedrich.py:86,107) while HJIPDE_solve and
    # the notebooks set derivFunc (hji_solver.py:434): accept either
    if isfield(sd, 'CoStateCalc'):
        return sd.CoStateCalc
    if isfield(sd, 'derivFunc'):
        return sd.derivFunc
    return None


class _Plan(tuple):
    """(grid, scheme_id, ham_id, params) plus .diss, the hj_ctx_set_dissipation kind."""

    def __new__(cls, items, diss):
        self = super(_Plan, cls).__new__(cls, items)
        self.diss = diss
        return self

    def bind(self, dg, post_op=0, post_a=None, post_b=None):
        # per-call state of the (cached, shared) ctx: which CFL bound, which fused post-step operators
        # (post_a / post_b: (op, device tensor) or None)
        _ffi.check(dg.lib.hj_ctx_set_dissipation(dg.ctx, self.diss))
        _ffi.check(dg.lib.hj_ctx_set_post_step(dg.ctx, post_op))
        _ffi.check(dg.lib.hj_ctx_set_post_arrays(dg.ctx, post_a[0] if post_a else 0, dg.ptr(post_a[1]) if post_a else None,
                                                 post_b[0] if post_b else 0, dg.ptr(post_b[1]) if post_b else None))


def native_plan(schemeData):
    """(grid, scheme_id, ham_id, params) if this LF schemeData can run fused, else None."""
    sd = schemeData[0] if iscell(schemeData) else schemeData
    for f in ('grid', 'dissFunc', 'hamFunc', 'partialFunc'):
        if not isfield(sd, f):
            return None
    fn = _deriv_func(sd)
    sid = scheme_id_of(fn) if fn is not None else None
    # all three Lax-Friedrichs variants run fused: a native Hamiltonian's alpha ignores the costate range, so
    # their dissipation terms coincide and only the CFL bound differs (hj_ctx_set_dissipation)
    if sid is None or sd.dissFunc not in (artificialDissipationGLF, artificialDissipationLLF, artificialDissipationLLLF):
        return None
    nat = native_of(sd.hamFunc, sd.partialFunc)
    if nat is None or nat[0].grid is not sd.grid:
        return None
    try:
        from .context import grid_bc
        grid_bc(sd.grid)
    except ValueError:
        return None
    return _Plan((sd.grid, sid, nat[1], nat[2]),
                 _ffi.DISS_GLF if sd.dissFunc is artificialDissipationGLF else _ffi.DISS_LOCAL)


def _fused_term(plan, t, y, restrict_sign):
    grid, sid, ham, par = plan
    dg = device_grid(grid, array_dtype_name(y))
    if int(np.prod(y.shape)) != dg.numel:
        raise ValueError('y does not agree in size with grid')
    dg.bind_stream()
    plan.bind(dg)
    yd = dg.to_device(y)
    out = dg.empty()
    sb = C.c_double()
    _ffi.check(dg.lib.hj_lf_term(dg.ctx, sid, ham, _ffi.darr(par), float(t), restrict_sign,
                                 dg.ptr(yd), dg.ptr(out), C.byref(sb)))
    return out, float(sb.value), dg


def termLaxFriedrichs(t, y, schemeData):
    """ydot = -(H(x, t, phi, (p^- + p^+)/2) - sum_i (p^+_i - p^-_i)/2 * alpha_i), returned as an
    (N,1) column like the reference (term_lax_friedrich.py:124-128); stepBound is a Python float."""
    if iscell(schemeData):
        thisSchemeData = copy.copy(schemeData[0])
    else:
        thisSchemeData = copy.copy(schemeData)
    assert isfield(thisSchemeData, 'grid'), 'grid not in bundle thisschemeData'
    assert _deriv_func(thisSchemeData) is not None, 'CoStateCalc not in bundle thisschemeData'
    assert isfield(thisSchemeData, 'dissFunc'), 'dissFunc not in bundle thisschemeData'
    assert isfield(thisSchemeData, 'hamFunc'), 'hamFunc not in bundle thisschemeData'
    assert isfield(thisSchemeData, 'partialFunc'), 'partialFunc not in bundle thisschemeData'
    y0 = y[0] if iscell(y) else y
    plan = native_plan(thisSchemeData)
    if plan is not None:
        out, stepBound, dg = _fused_term(plan, t, y0, 0)
        return dg.like(out, y0, (dg.numel, 1)), stepBound, schemeData
    # ---- split path (term_lax_friedrich.py:94-130)
    grid = thisSchemeData.grid
    data = y0.reshape(grid.shape)
    calc = _deriv_func(thisSchemeData)
    derivL, derivR, derivC = [None] * grid.dim, [None] * grid.dim, [None] * grid.dim
    both = upwind_all_dims(calc, grid, data) if _diss.SPLIT_KERNELS else None   # device tensors + one of our schemes: one call, one sync
    for i in range(grid.dim):
        derivL[i], derivR[i] = (both[0][i], both[1][i]) if both is not None else calc(grid, data, i)
        derivC[i] = 0.5 * (derivL[i] + derivR[i])
    result = thisSchemeData.hamFunc(t, data, derivC, thisSchemeData)
    if isinstance(result, tuple):
        ham, thisSchemeData = result
        if iscell(schemeData):
            schemeData[0] = copy.copy(thisSchemeData)
        else:
            schemeData = copy.copy(thisSchemeData)
    else:
        ham = result
    if thisSchemeData.dissFunc is artificialDissipationGLF and is_tensor(ham) and _diss.SPLIT_KERNELS:
        # dissipation, -(ham - diss) and the per-dimension max(alpha) in ONE kernel after the partialFunc callbacks
        res = glf_device(t, data, derivL, derivR, thisSchemeData, ham=ham)
        if res is not None:
            return res[0].reshape(-1, 1), res[1], schemeData
    diss, stepBound = thisSchemeData.dissFunc(t, data, derivL, derivR, thisSchemeData)
    delta = ham - diss
    ydot = (-delta).reshape(-1, 1)
    return ydot, stepBound, schemeData


def termRestrictUpdate(t, y, schemeData):
    """min(inner, 0) or max(inner, 0), squeezed to (N,) (term_restrict_update.py:83-102)."""
    thisSchemeData = schemeData[0] if iscell(schemeData) else schemeData
    assert isfield(thisSchemeData, 'innerFunc'), "innerFunc not in schemeData"
    assert isfield(thisSchemeData, 'innerData'), "innerData not in schemeData"
    positive = thisSchemeData.positive if isfield(thisSchemeData, 'positive') else True
    y0 = y[0] if iscell(y) else y
    if thisSchemeData.innerFunc is termLaxFriedrichs:
        plan = native_plan(thisSchemeData.innerData)
        if plan is not None:
            out, stepBound, dg = _fused_term(plan, t, y0, +1 if positive else -1)
            return dg.like(out, y0, (dg.numel,)), stepBound, schemeData
    if iscell(schemeData):
        innerData = schemeData
        innerData[0] = schemeData[0].innerData
    else:
        innerData = copy.copy(schemeData.innerData)
    unRestricted, stepBound, innerData = thisSchemeData.innerFunc(t, y, innerData)
    if iscell(schemeData):
        schemeData[0].innerData = innerData[0]
    else:
        schemeData.innerData = innerData
    if is_tensor(unRestricted):
        ydot = unRestricted.clamp_min(0) if positive else unRestricted.clamp_max(0)
        ydot = ydot.squeeze()
    else:
        ydot = (np.maximum(unRestricted, 0) if positive else np.minimum(unRestricted, 0)).squeeze()
    return ydot, stepBound, schemeData

"""termLaxFriedrichs / termRestrictUpdate behind the reference's schemeFunc protocol
    ydot, stepBound, schemeData = schemeFunc(t, y, schemeData)
(reference ExplicitIntegration/Term/term_lax_friedrich.py:8, term_restrict_update.py:8).

Fused path: when hamFunc/partialFunc are the methods of one of dynamics.py's systems, dissFunc is
this package's artificialDissipationGLF and the derivative function is one of spatial.py's, the
whole term is ONE HIP kernel (hj_lf_term).  Callbacks the package does not know are TRACED (trace_ham.py): called once
with symbolic arrays, written out as a device expression, compiled with hipRTC and checked against the callbacks on the
first data they meet -- then they run fused as well.  What cannot be traced takes the split path: per-dimension hj_upwind
kernels, then the user's callbacks on arrays, exactly as the reference sequences them.
"""
import copy
import ctypes as C
import weakref

import numpy as np

from . import _ffi
from .context import device_grid, array_dtype_name, is_tensor
from . import dissipation as _diss
from .dissipation import artificialDissipationGLF, artificialDissipationLLF, artificialDissipationLLLF, glf_device
from .dynamics import native_of, native_again
from .user_ham import ATTACH_EPOCH as _ATTACH_EPOCH
from . import trace_ham as _trace
from .spatial import scheme_id_of, upwind_all_dims
from .utilities import isfield, iscell

__all__ = ["termLaxFriedrichs", "termRestrictUpdate", "explain_plan"]


def _deriv_func(sd):
    # the reference's term reads CoStateCalc (term_lax_friedrich.py:86,107) while HJIPDE_solve and
    # the notebooks set derivFunc (hji_solver.py:434): accept either
    if isfield(sd, 'CoStateCalc'):
        return sd.CoStateCalc
    if isfield(sd, 'derivFunc'):
        return sd.derivFunc
    return None


class _Plan(tuple):
    """(grid, scheme_id, ham_id, params) plus .diss, the hj_ctx_set_dissipation kind; .parv = params as a C array;
    .system = the object whose bound methods hamFunc / partialFunc are; .traced: the kernel was generated from the callbacks."""

    def __new__(cls, items, diss, system=None, dynamic=False, traced=False):
        self = super(_Plan, cls).__new__(cls, items)
        self.diss = diss
        self.system = system
        self.traced = traced
        # alpha depends on the data (a run-time Hamiltonian that reads the costate range, user_ham.py): no static stepBound --
        # the integrators take deltaT from the first stage's reduced bound (hj_rk_step does, ode_cfl_3.py:142)
        self.dynamic = dynamic
        self.parv = _ffi.darr(items[3])
        return self

    def bind(self, dg, post_op=0, post_a=None, post_b=None):
        # per-call state of the (cached, shared) ctx: which CFL bound, which fused post-step operators
        # (post_a / post_b: (op, device tensor) or None).  Every Python path writes this state through here, so an
        # unchanged state (the common case: one schemeData stepped again and again) costs a tuple compare, not three C calls
        state = (self.diss, post_op,
                 (post_a[0], post_a[1].data_ptr()) if post_a else None, (post_b[0], post_b[1].data_ptr()) if post_b else None)
        # (another user of the ctx -- a test, a C caller -- may have written this state since: the library counts the writes)
        dg.forget_if_written_elsewhere()
        if dg.bound_state == state:
            return
        _ffi.check(dg.lib.hj_ctx_set_dissipation(dg.ctx, self.diss))
        _ffi.check(dg.lib.hj_ctx_set_post_step(dg.ctx, post_op))
        _ffi.check(dg.lib.hj_ctx_set_post_arrays(dg.ctx, post_a[0] if post_a else 0, dg.ptr(post_a[1]) if post_a else None,
                                                 post_b[0] if post_b else 0, dg.ptr(post_b[1]) if post_b else None))
        dg.bound_state = state
        dg.bound_gen = dg.lib.hj_ctx_state_generation(dg.ctx)

    def static_step_bound(self, dg):
        """stepBound of this (grid, system, dissipation kind): data independent for the native systems, one C call per ctx."""
        if self.dynamic:
            return None
        # kept ON the DeviceGrid (an id(dg) key could be reused by a later object once this one is collected)
        cache = dg.__dict__.setdefault("_sb_cache", {})
        key = (self[2], tuple(self[3]), self.diss)
        sb = cache.get(key)
        if sb is None:
            v = C.c_double()
            _ffi.check(dg.lib.hj_static_step_bound(dg.ctx, self[2], self.parv, C.byref(v), None))
            sb = cache[key] = float(v.value)
        return sb


# schemeData Bundle -> (the five callables / objects the plan was derived from, the plan).  A hit costs five identity
# compares and one system.native() call (the system's speeds may have been changed in place) instead of the full
# classification (35-40 us of host time per singleStep call went into re-deriving unchanged state).
# Weak keys: the cache never keeps a caller's Bundle alive.
_PLAN_CACHE = weakref.WeakKeyDictionary()


def native_plan(schemeData, y=None):
    """(grid, scheme_id, ham_id, params) if this LF schemeData can run fused, else None.  y: the data the caller is about to step --
    a plan TRACED from Python callbacks (trace_ham.py) is checked against them on the first data it meets, and is not used before."""
    sd = schemeData[0] if iscell(schemeData) else schemeData
    plan = _plan_of(sd)
    if plan is not None and plan.traced and not getattr(plan.system.reg, "verified", False):
        plan = _verify_traced(plan, sd, y)
    return plan


def explain_plan(schemeData):
    """Which path a Lax-Friedrichs schemeData takes and why -- for the question "why is this slow?".  Returns a dict: path = 'built-in' |
    'registered' (user_ham.attach) | 'traced' | 'split', reason (split: what stopped the classification or the tracer), source (traced: the
    generated device expression), verified (traced: has the kernel been checked against the callbacks yet -- it is on first use)."""
    sd = schemeData[0] if iscell(schemeData) else schemeData
    for f in ('grid', 'dissFunc', 'hamFunc', 'partialFunc'):
        if not isfield(sd, f):
            return dict(path='split', reason='schemeData has no %s' % f)
    fn = _deriv_func(sd)
    if fn is None or scheme_id_of(fn) is None:
        return dict(path='split', reason='the derivative function is not one of this package\'s upwindFirst* schemes')
    if sd.dissFunc not in (artificialDissipationGLF, artificialDissipationLLF, artificialDissipationLLLF):
        return dict(path='split', reason='dissFunc is not one of this package\'s artificialDissipation* functions')
    plan = _plan_of(sd)
    if plan is not None and not plan.traced:
        return dict(path='registered' if plan[2] >= _ffi.HAM_USER_BASE else 'built-in', ham_id=plan[2], params=list(plan[3]))
    if not _trace.enabled():
        return dict(path='split', reason='the callbacks are not known to the package and HJ_TRACE=0')
    try:
        tr = _trace.trace_callbacks(sd.grid, sd.hamFunc, sd.partialFunc, sd)
    except _trace.TraceError as e:
        return dict(path='split', reason='the callbacks cannot be traced: %s' % e)
    if plan is None:
        return dict(path='split', reason='the traced kernel disagreed with the callbacks on first use, the expression keeps changing, or the library '
                                         'refused the registration (HJ_TRACE_VERBOSE=1 says which)', source=tr.source)
    return dict(path='traced', ham_id=plan[2], params=list(plan[3]), source=tr.source, column_source=tr.column_source,
                uses_range=tr.uses_range, verified=bool(getattr(plan.system.reg, "verified", False)))


def _plan_of(sd):
    d = sd.__dict__
    fn = d['CoStateCalc'] if 'CoStateCalc' in d else d.get('derivFunc')
    try:
        hit = _PLAN_CACHE.get(sd)
    except TypeError:           # an unhashable / non-weakrefable stand-in for a Bundle: classify every time
        hit = None
    if hit is not None:
        src, plan = hit
        if src[0] is d.get('grid') and src[1] is d.get('dissFunc') and src[2] is d.get('hamFunc') \
                and src[3] is d.get('partialFunc') and src[4] is fn:
            if plan is None:
                if src[5] == _attach_epoch():
                    return None
                # (an object was attach()ed to a run-time Hamiltonian since: this schemeData may have become native)
            else:
                nat = native_again(plan.system)
                if nat is not None and nat[0] == plan[2] and nat[1] == plan[3] and scheme_id_of(fn) == plan[1]:
                    return plan
    plan = _classify(sd, fn)
    try:
        _PLAN_CACHE[sd] = ((d.get('grid'), d.get('dissFunc'), d.get('hamFunc'), d.get('partialFunc'), fn, _attach_epoch()), plan)
    except TypeError:
        pass
    return plan


def _attach_epoch():
    return _ATTACH_EPOCH[0]


def _classify(sd, fn):
    for f in ('grid', 'dissFunc', 'hamFunc', 'partialFunc'):
        if not isfield(sd, f):
            return None
    sid = scheme_id_of(fn) if fn is not None else None
    # all three Lax-Friedrichs variants run fused: a native Hamiltonian's alpha ignores the costate range, so
    # their dissipation terms coincide and only the CFL bound differs (hj_ctx_set_dissipation)
    if sid is None or sd.dissFunc not in (artificialDissipationGLF, artificialDissipationLLF, artificialDissipationLLLF):
        return None
    nat = native_of(sd.hamFunc, sd.partialFunc)
    traced = False
    try:
        from .context import grid_bc
        grid_bc(sd.grid)
    except ValueError:
        return None
    if nat is None and 2 <= int(sd.grid.dim) <= 4 and callable(sd.hamFunc) and callable(sd.partialFunc):
        # callbacks nobody has written a kernel for: record what they compute (trace_ham.py)
        nat = _trace.traced_native(sd)
        traced = nat is not None
    if nat is None or nat[0].grid is not sd.grid:
        return None
    att = getattr(nat[0], "_hj_native", None)
    dynamic = bool(att is not None and getattr(att.reg, "uses_range", False))
    # (for a Hamiltonian that ignores the costate range the two local variants coincide; one that reads it gets the per-node ranges of
    #  diss_local_laxfried.py / diss_localsq_laxfried.py inside the fused kernel: hj_mi355x.h, HJ_DISS_LLF / HJ_DISS_LLLF)
    kind = {artificialDissipationGLF: _ffi.DISS_GLF, artificialDissipationLLF: _ffi.DISS_LLF, artificialDissipationLLLF: _ffi.DISS_LLLF}[sd.dissFunc]
    return _Plan((sd.grid, sid, nat[1], nat[2]), kind, nat[0], dynamic, traced)


def _verify_traced(plan, sd, y):
    """First use of a kernel generated from Python callbacks: the term by BOTH paths on the caller's data.  Agreement to rounding marks the
    registration verified (once per process and expression); anything else drops the trace with a warning and the split path stays."""
    if y is None:
        return None
    reg = plan.system.reg
    import warnings
    try:
        y0 = y.reshape(-1, 1)
        fused, sb_f, dg = _fused_term(plan, 0.0, y0, 0)
        split, sb_s, _ = _split_term(0.0, y0, sd, sd)
        f = dg.like(fused, split, (dg.numel, 1))
        if is_tensor(split):
            diff = (f - split.reshape(-1, 1)).abs()
            scale = float(split.abs().max()) + 1e-300
            finite = bool(diff.isfinite().all())
            tol = (1e-9 if str(split.dtype).endswith("64") else 2e-4) * scale
            frac = float((diff > tol).double().mean())
            worst = float(diff.max())
        else:
            f, s_ = np.asarray(f).reshape(-1, 1), np.asarray(split).reshape(-1, 1)
            diff = np.abs(f - s_)
            scale = float(np.abs(s_).max()) + 1e-300
            finite = bool(np.isfinite(diff).all())
            tol = (1e-9 if s_.dtype == np.float64 else 2e-4) * scale
            frac = float((diff > tol).mean())
            worst = float(diff.max())
        # (an ENO stencil choice may flip where the two arithmetics differ in the last bit: a few isolated nodes, not an expression error)
        single = not str(getattr(split, "dtype", "float64")).endswith("64")
        ok = finite and frac <= 2e-3 and worst <= 5e-2 * scale and abs(sb_f - sb_s) <= (1e-4 if single else 1e-7) * abs(sb_s)
    except (_ffi.Unsupported, _trace.TraceError) as e:
        # (no kernel for this grid -- no tiling, too few nodes -- or a callback that met something symbolic it had kept: the split path)
        import os
        if os.environ.get("HJ_TRACE_VERBOSE"):
            warnings.warn("levelsetpy_amd: the traced kernel of %r was not used: %s: %s" % (getattr(sd.hamFunc, "__qualname__", sd.hamFunc), type(e).__name__, e))
        return None
    if not ok:
        _trace.mark_bad(reg)
        try:
            del _PLAN_CACHE[sd]
        except (KeyError, TypeError):
            pass
        warnings.warn("levelsetpy_amd: the kernel traced from %r disagrees with the callbacks on this data (max |diff| %.3g of %.3g, %.2g %% of the "
                      "nodes, stepBound %.17g / %.17g): the split path is used" % (getattr(sd.hamFunc, "__qualname__", sd.hamFunc), worst, scale,
                                                                                 100 * frac, sb_f, sb_s))
        return None
    reg.verified = True
    return plan


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
    _ffi.check(dg.lib.hj_lf_term(dg.ctx, sid, ham, plan.parv, float(t), restrict_sign,
                                 dg.ptr(yd), dg.ptr(out), C.byref(sb)))
    return out, float(sb.value), dg


def termLaxFriedrichs(t, y, schemeData):
    """ydot = -(H(x, t, phi, (p^- + p^+)/2) - sum_i (p^+_i - p^-_i)/2 * alpha_i), returned as an
    (N,1) column like the reference (term_lax_friedrich.py:124-128); stepBound is a Python float."""
    sd0 = schemeData[0] if iscell(schemeData) else schemeData
    assert isfield(sd0, 'grid'), 'grid not in bundle thisschemeData'
    assert _deriv_func(sd0) is not None, 'CoStateCalc not in bundle thisschemeData'
    assert isfield(sd0, 'dissFunc'), 'dissFunc not in bundle thisschemeData'
    assert isfield(sd0, 'hamFunc'), 'hamFunc not in bundle thisschemeData'
    assert isfield(sd0, 'partialFunc'), 'partialFunc not in bundle thisschemeData'
    y0 = y[0] if iscell(y) else y
    plan = native_plan(sd0, y0)      # looked up on the caller's own Bundle (the plan cache is keyed by it)
    if plan is not None:
        try:
            out, stepBound, dg = _fused_term(plan, t, y0, 0)
            return dg.like(out, y0, (dg.numel, 1), lazy=True), stepBound, schemeData
        except _ffi.Unsupported:
            # a run-time Hamiltonian the library has no kernel for on this grid (no tiling, ...): the callbacks still work --
            # the split path, as before the registration (the built-in systems fall back inside the library)
            if plan[2] < _ffi.HAM_USER_BASE:
                raise
    return _split_term(t, y0, sd0, schemeData)


def _split_term(t, y0, sd0, schemeData):
    """term_lax_friedrich.py:94-130: derivative kernels, the callbacks on arrays, the dissipation."""
    thisSchemeData = copy.copy(sd0)      # the reference's shallow copy (term_lax_friedrich.py:80-83)
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
        plan = native_plan(thisSchemeData.innerData, y0)
        if plan is not None:
            out, stepBound, dg = _fused_term(plan, t, y0, +1 if positive else -1)
            return dg.like(out, y0, (dg.numel,), lazy=True), stepBound, schemeData
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

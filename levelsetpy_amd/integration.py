"""CFL-constrained TVD Runge-Kutta integrators odeCFL1/2/3 and their option helpers (reference
ExplicitIntegration/Integration/ode_cfl_{1,2,3,set,get,mult,call}.py):
    t, y, schemeData = odeCFLn(schemeFunc, tspan, y0, options, schemeData)

Device path: when schemeFunc is this package's termLaxFriedrichs (or termRestrictUpdate around
it) with a native Hamiltonian, every time step is `order` fused HIP launches (hj_rk_step); y
stays in HBM for the whole tspan and no host synchronisation happens inside a step.
Generic path: any other schemeFunc is integrated with the reference's own sequence of array
expressions (the arrays may be NumPy or device tensors), so user terms keep working.
"""
import copy
import ctypes as C

import numpy as np

from . import _ffi
from .context import device_grid, array_dtype_name, is_tensor
from . import dissipation as _diss
from .term import termLaxFriedrichs, termRestrictUpdate, native_plan
from .utilities import (Bundle, isbundle, isfield, iscell, strcmp, warn, info, cputime, eps,
                        realmax, error)

__all__ = ["odeCFL1", "odeCFL2", "odeCFL3", "odeCFLset", "odeCFLget", "odeCFLmultipleSteps",
           "odeCFLcallPostTimestep"]


def odeCFLset(*args, **kw):
    """ode_cfl_set.py:5.  Takes a Bundle (as the reference) or keyword arguments.  factorCFL
    default 0.5; maxStep is read from key 'maxStep' and, as the reference does, from 'realmax'
    (ode_cfl_set.py:96); postTimeStep / postTimestep both accepted."""
    if args and args[0] is not None:
        kwargs = args[0]
        assert isbundle(kwargs), "kwargs must be a bundle type."
        d = dict(kwargs.__dict__)
    else:
        d = {}
    d.update(kw)
    if not d:
        raise ValueError('kwargs cannot be None')      # ode_cfl_set.py:91
    options = Bundle({})
    options.factorCFL = d.get('factorCFL', 0.5)
    options.maxStep = d.get('maxStep', d.get('realmax', realmax))
    post = d.get('postTimeStep', d.get('postTimestep', None))
    options.postTimeStep = post
    options.postTimestep = post
    options.singleStep = d.get('singleStep', 'off')
    options.stats = d.get('stats', 'off')
    options.terminalEvent = d.get('terminalEvent', None)
    if options.factorCFL < 0.0:
        raise ValueError('FactorCFL must be a positive scalar double value')
    if options.maxStep < 0.0:
        raise ValueError('MaxStep must be a positive scalar double value')
    if post is not None:
        items = post if isinstance(post, list) else [post]
        for f in items:
            if not callable(f):
                raise ValueError('Each element in a postTimeStep cell vector must be a function handle.')
    if options.singleStep not in ('on', 'off'):
        raise ValueError("SingleStep must be one of the strings 'on' or 'off'")
    if options.stats not in ('on', 'off'):
        raise ValueError("Stats must be one of the strings 'on' or 'off'")
    if options.terminalEvent is not None and not callable(options.terminalEvent):
        raise ValueError('terminalEvent parameter must be a function handle.')
    return options


def odeCFLget(options, name, default=None):
    """ode_cfl_get.py: value of one option field (case-insensitive prefix match)."""
    if options is None:
        return default
    names = ['factorCFL', 'maxStep', 'postTimeStep', 'singleStep', 'stats', 'terminalEvent']
    match = [n for n in names if n.lower().startswith(str(name).lower())]
    if len(match) != 1:
        error('Unrecognized or ambiguous property name %s' % name)
    val = getattr(options, match[0], None)
    return default if val is None else val


def odeCFLcallPostTimestep(t, yIn, schemeDataIn, options):
    """ode_cfl_call.py:6: run options.postTimeStep (callable or list) on (t, y, schemeData)."""
    yOut, schemeDataOut = copy.copy(yIn), copy.copy(schemeDataIn)
    if not options:
        return yOut, schemeDataOut
    post = getattr(options, 'postTimeStep', None) or getattr(options, 'postTimestep', None)
    if not post:
        return yOut, schemeDataOut
    for f in (post if isinstance(post, list) else [post]):
        yOut, schemeDataOut = f(t, yOut, schemeDataOut)
    return yOut, schemeDataOut


def odeCFLmultipleSteps(intFunc, schemeFunc, tspan, y0, options, schemeData):
    """ode_cfl_mult.py:7: solution at every entry of tspan, one row of y per time."""
    tspan = np.asarray(tspan, dtype=np.float64)
    numT = len(tspan)
    if numT <= 2:
        error('tspan must contain at least three entries')
    t = tspan.reshape(numT, 1).copy()
    flat0 = (y0.detach().cpu().numpy() if is_tensor(y0) else np.asarray(y0)).reshape(-1)
    y = np.zeros((numT, flat0.size), dtype=np.float64)
    y[0, :] = flat0
    yout = y0
    for n in range(1, numT):
        tn, yout, schemeData = intFunc(schemeFunc, [t[n - 1, 0], t[n, 0]], yout, options, schemeData)
        t[n, 0] = tn
        y[n, :] = (yout.detach().cpu().numpy() if is_tensor(yout) else np.asarray(yout)).reshape(-1)
    return t, y, schemeData


# ----------------------------------------------------------------------------------------------
def _options(options):
    if not options:
        return odeCFLset(factorCFL=0.5)
    o = options
    for k, v in (('factorCFL', 0.5), ('maxStep', realmax), ('singleStep', 'off'), ('stats', 'off')):
        if not isfield(o, k):
            setattr(o, k, v)
    return o


def _post_hook(options):
    return getattr(options, 'postTimeStep', None) or getattr(options, 'postTimestep', None)


def _device_plan(schemeFunc, schemeData, y0):
    """(plan, restrict_sign) if the integrator can run fused on the device."""
    if iscell(y0) or iscell(schemeData):
        return None
    if schemeFunc is termLaxFriedrichs:
        plan, rs = native_plan(schemeData, y0), 0
    elif schemeFunc is termRestrictUpdate and isfield(schemeData, 'innerFunc') \
            and schemeData.innerFunc is termLaxFriedrichs and isfield(schemeData, 'innerData'):
        plan = native_plan(schemeData.innerData, y0)
        positive = schemeData.positive if isfield(schemeData, 'positive') else True
        rs = +1 if positive else -1
    else:
        return None
    if plan is None:
        return None
    return plan, rs


def _check_shape(schemeFunc, y0):
    # termLaxFriedrichs returns an (N,1) column, termRestrictUpdate an (N,) vector; the reference
    # silently broadcasts the mixed case to (N,N) -- reject it instead.
    nd = y0.dim() if is_tensor(y0) else (len(y0.shape) if hasattr(y0, 'shape') else np.ndim(y0))
    if schemeFunc is termLaxFriedrichs and nd != 2:
        raise ValueError('termLaxFriedrichs needs y0 as an (N,1) column (got %d-D)' % nd)
    if schemeFunc is termRestrictUpdate and nd != 1:
        raise ValueError('termRestrictUpdate needs y0 as an (N,) vector (got %d-D)' % nd)


def _integrate(order, schemeFunc, tspan, y0, options, schemeData):
    small = 100 * eps                                    # ode_cfl_3.py:81
    options = _options(options)
    safetyFactorCFL = min(1.0, 1.2 * options.factorCFL)  # :95
    numT = len(tspan)
    if numT > 2:
        intFunc = {1: odeCFL1, 2: odeCFL2, 3: odeCFL3}[order]
        return odeCFLmultipleSteps(intFunc, schemeFunc, tspan, y0, options, schemeData)
    if numT < 2:
        raise ValueError('tspan must contain at least two entries')
    if iscell(y0):
        raise ValueError('vector level sets (list y0) are not supported (broken in the reference: '
                         'ode_cfl_3.py:147-149)')
    dev = _device_plan(schemeFunc, schemeData, y0)
    if dev is not None:
        _check_shape(schemeFunc, y0)
        return _integrate_device(order, dev, tspan, y0, options, schemeData)
    return _integrate_generic(order, schemeFunc, tspan, y0, options, schemeData, small, safetyFactorCFL)


def _integrate_device(order, dev, tspan, y0, options, schemeData):
    plan, rs = dev
    grid, sid, ham, par = plan
    small = 100 * eps
    dg = device_grid(grid, array_dtype_name(y0))
    if int(np.prod(y0.shape)) != dg.numel:
        raise ValueError('y0 does not agree in size with grid')
    dg.bind_stream()
    plan.bind(dg)
    shape0 = tuple(y0.shape)
    lib, ctx, ptr = dg.lib, dg.ctx, dg.ptr
    # the input is only ever read (hj_rk_step never writes y_in), so it is used in place; results go
    # to buffers allocated here (A/B ping-pong for multi-step spans), so nothing is cloned and the
    # caller's array is never mutated
    cur = dg.to_device(y0).reshape(dg.shape)
    nxt = dg.empty()
    spare = None
    # RK3: the first stage buffer doubles as the output (stage 3 reads w1 and y_in only); RK2's second
    # stage reads the first stage buffer as its stencil input, so it needs its own
    dynamic = getattr(plan, 'dynamic', False)       # alpha depends on the data: deltaT from the first stage's reduced bound, per step
    w0_own = dg.work('rk_w0') if order == 2 else None
    w1 = dg.work('rk_w1') if order == 3 else None
    parv = plan.parv
    t = float(tspan[0])
    tf = float(tspan[1])
    steps = 0
    startTime = cputime()
    post = _post_hook(options)
    tout, dtout = C.c_double(), C.c_double()
    eventValueOld = None
    factorCFL, maxStep = float(options.factorCFL), float(options.maxStep)
    single = strcmp(options.singleStep, 'on')
    terminal = getattr(options, 'terminalEvent', None)
    # the native Hamiltonians have a data-independent alpha, so stepBound is the same at every
    # substep: the reference's CFL warning (ode_cfl_3.py:173-175,215-217) can only fire when
    # factorCFL > min(1, 1.2 factorCFL), and it is decided on the host without a device read
    safetyFactorCFL = min(1.0, 1.2 * factorCFL)
    sb_static = plan.static_step_bound(dg)
    # no per-step callbacks and nothing to warn about: the whole span is one native call
    # (hj_rk_integrate runs the same loop in C; no Python and no host synchronisation per step)
    # (a data-dependent alpha can violate the bound at a later stage whatever factorCFL is: those plans step one call at a time below,
    # where the reference's warning is raised from the stages' bounds)
    if (not post and not terminal and not single and not dynamic and factorCFL <= safetyFactorCFL and tf - t >= small * abs(tf)):
        buf_b = dg.empty()
        work = dg.work('rk_w1')
        nsteps, where = C.c_int64(), C.c_int()
        _ffi.check(lib.hj_rk_integrate(ctx, order, sid, ham, parv, t, tf, factorCFL, maxStep, rs, ptr(cur), ptr(nxt), ptr(buf_b),
                                       ptr(work), 0, -1.0, C.byref(tout), C.byref(nsteps), C.byref(where)))
        cur = (cur, nxt, buf_b)[where.value]
        if where.value == 1:
            nxt = buf_b                # never step in place should the loop below still have something to do
        t = float(tout.value)
        steps = int(nsteps.value)
    while tf - t >= small * abs(tf):
        tOld = t
        _ffi.check(lib.hj_rk_step(ctx, order, sid, ham, parv, t, tf, factorCFL, maxStep, rs, ptr(cur), ptr(nxt),
                                  ptr(nxt if order == 3 else w0_own), ptr(w1), C.byref(tout), C.byref(dtout)))
        yOld = cur
        prev = cur
        cur = nxt
        t = float(tout.value)
        steps += 1
        if dynamic:
            if order > 1:
                # the later stages' bounds arrive asynchronously (the step returns while its last stage runs): warn about whatever has
                # arrived -- normally the previous step -- and drain after the loop (round 5: no host <-> device round trip here)
                _warn_late_bounds(lib, ctx, order, safetyFactorCFL, warn, wait=False)
        elif order > 1 and dtout.value > safetyFactorCFL * sb_static:
            for which in ('Second', 'Third')[:order - 1]:
                warn('%s substep violated CFL effective number %s' % (which, dtout.value / sb_static))
        if single and not post:
            break                      # the common drop-in call (hji_solver.py:542): one step, nothing else to prepare
        # next output buffer: recycle the one two steps back if it is ours, never the caller's input
        nxt = spare if spare is not None else dg.empty()
        spare = prev if steps >= 2 else None
        if post:
            yv = dg.like(cur.reshape(shape0), y0)
            yv, schemeData = odeCFLcallPostTimestep(t, yv, schemeData, options)
            cur = dg.to_device(yv).reshape(dg.shape)
            if cur.data_ptr() == nxt.data_ptr() or (spare is not None and cur.data_ptr() == spare.data_ptr()):
                cur = cur.clone()
        if single:
            break
        if terminal:
            # the old state lives in `nxt` until the next step overwrites it
            eventValue, schemeData = terminal(
                t, dg.like(cur.reshape(shape0), y0), tOld, dg.like(yOld.reshape(shape0), y0), schemeData)
            if steps > 1 and np.any(np.sign(eventValue) != np.sign(eventValueOld)):
                break
            eventValueOld = eventValue
    if dynamic and order > 1 and steps > 0 and not single:
        _warn_late_bounds(lib, ctx, order, safetyFactorCFL, warn, wait=True, dt_last=dtout.value)   # the last step's (a singleStep call leaves them to the next call)
    if strcmp(options.stats, 'on'):
        dg.sync()
        info('%d steps in %.2g seconds from  %.2f to %.2f.' % (steps, cputime() - startTime, tspan[0], t))
    out = cur.reshape(shape0)
    if is_tensor(y0) and steps == 0:
        out = out.clone()          # zero steps taken: do not hand the caller's own tensor back
    return np.float64(t), dg.like(out, y0, lazy=True), schemeData


def _warn_late_bounds(lib, ctx, order, safety, warn, wait, dt_last=None):
    """The reference's 'substep violated CFL' warnings (ode_cfl_3.py:173-175, 215-217) for a Hamiltonian whose alpha depends on the
    data: the later stages' stepBounds of a step reach the host asynchronously.  wait=False: whatever has arrived and has not been
    reported yet (hj_rk_prev_bounds: at most one step, with its own deltaT); wait=True: that, then the last step's (deltaT dt_last),
    waiting for its launches (hj_rk_last_bounds)."""
    sbs, nsb, dt = (C.c_double * 3)(), C.c_int(), C.c_double()
    _ffi.check(lib.hj_rk_prev_bounds(ctx, sbs, C.byref(nsb), C.byref(dt)))
    _check_bounds(sbs, nsb.value, dt.value, order, safety, warn)
    if wait:
        _ffi.check(lib.hj_rk_last_bounds(ctx, sbs, C.byref(nsb)))
        _check_bounds(sbs, nsb.value, float(dt_last), order, safety, warn)


def _check_bounds(sbs, n, dt, order, safety, warn):
    for k, which in enumerate(('Second', 'Third')[:order - 1]):
        if k + 1 < n and dt > safety * sbs[k + 1]:
            warn('%s substep violated CFL effective number %s' % (which, dt / sbs[k + 1]))


def integrate_span_device(schemeFunc, schemeData, y, t0, tf, options, stop_tol, post_op=0, order=3,
                          post_a=None, post_b=None):
    """HJIPDE_solve's inner loop for one tau interval in ONE native call (hji_solver.py:536-543 plus the
    min/max-over-time operator of :571-575 fused into the last RK stage): steps while t < tf - stop_tol.
    `y` is a device tensor; returns (t, new device tensor), or None if the problem cannot run fused."""
    dev = _device_plan(schemeFunc, schemeData, y)
    if dev is None:
        return None
    plan, rs = dev
    grid, sid, ham, par = plan
    dg = device_grid(grid, array_dtype_name(y))
    dg.bind_stream()
    plan.bind(dg, post_op, post_a, post_b)
    try:
        cur = dg.to_device(y).reshape(dg.shape)
        a, b, w = dg.empty(), dg.empty(), dg.work('rk_w1')
        tout, nsteps, where = C.c_double(), C.c_int64(), C.c_int()
        options = _options(options)
        _ffi.check(dg.lib.hj_rk_integrate(dg.ctx, order, sid, ham, plan.parv, float(t0), float(tf),
                                          float(options.factorCFL), float(options.maxStep), rs, dg.ptr(cur),
                                          dg.ptr(a), dg.ptr(b), dg.ptr(w), 0, float(stop_tol),
                                          C.byref(tout), C.byref(nsteps), C.byref(where)))
    finally:
        plan.bind(dg, 0)
    out = (cur, a, b)[where.value]
    if where.value == 0:
        out = out.clone()
    return float(tout.value), out.reshape(y.shape)


def _any_device_grid(schemeData, like):
    """The DeviceGrid of the grid a (possibly wrapped) schemeData carries, for the dtype of `like`; None if
    there is none (the elementwise stage kernels only need a ctx of the right dtype and device)."""
    sd = schemeData[0] if iscell(schemeData) else schemeData
    for _ in range(4):
        if sd is None:
            return None
        if isfield(sd, 'grid'):
            try:
                return device_grid(sd.grid, array_dtype_name(like))
            except (ValueError, RuntimeError):
                return None
        sd = getattr(sd, 'innerData', None)
    return None


def _integrate_generic(order, schemeFunc, tspan, y0, options, schemeData, small, safetyFactorCFL):
    t = tspan[0]
    tf = tspan[1]
    steps = 0
    startTime = cputime()
    y = copy.copy(y0)
    post = _post_hook(options)
    eventValueOld = None

    def combine(mode, deltaT, x0, ycur, ydot):
        """One odeCFLn stage expression (ode_cfl_3.py:151,184-193,226-241; ode_cfl_2.py:184-201): a single
        hj_rk_combine launch for device tensors of one of our grids, the reference's array expression otherwise."""
        e = None
        if _diss.SPLIT_KERNELS and is_tensor(ycur) and is_tensor(ydot) and ycur.is_cuda and ydot.is_cuda and ycur.dtype == ydot.dtype \
                and ycur.is_contiguous() and ydot.is_contiguous() and tuple(ycur.shape) == tuple(ydot.shape) \
                and (x0 is None or (is_tensor(x0) and x0.is_cuda and x0.is_contiguous() and x0.dtype == ycur.dtype)):
            dg = _any_device_grid(schemeData, ycur)
            if dg is not None:
                out = ycur.new_empty(ycur.shape)
                dg.bind_stream()
                _ffi.check(dg.lib.hj_rk_combine(dg.ctx, mode, float(deltaT), dg.ptr(x0), dg.ptr(ycur), dg.ptr(ydot),
                                                dg.ptr(out), ycur.numel()))
                return out
        e = ycur + deltaT * ydot
        if mode == 1:
            return e
        if mode == 2:
            return 0.25 * (3 * x0 + e)
        if mode == 3:
            return (1 / 3) * (x0 + 2 * e)
        return 0.5 * (x0 + e)

    def bound_check(deltaT, stepBound, which):
        if deltaT > safetyFactorCFL * stepBound:          # ode_cfl_3.py:173-175,215-217
            warn('%s substep violated CFL effective number %s' % (which, deltaT / stepBound))

    while tf - t >= small * np.abs(tf):
        ydot, stepBound, schemeData = schemeFunc(t, y, schemeData)
        if tuple(ydot.shape) != tuple(y.shape):
            raise ValueError('schemeFunc returned shape %s for a state of shape %s (use an (N,1) state '
                             'with termLaxFriedrichs and an (N,) state with termRestrictUpdate)'
                             % (tuple(ydot.shape), tuple(y.shape)))
        deltaT = min(options.factorCFL * stepBound, tf - t, options.maxStep)    # :142
        t1 = t + deltaT
        y1 = combine(1, deltaT, None, y, ydot)            # y + deltaT*ydot
        yOld, tOld = y, t
        if order == 1:
            y, t = y1, t1
        else:
            ydot, stepBound, schemeData = schemeFunc(t1, y1, schemeData)
            bound_check(deltaT, stepBound, 'Second')
            t2 = t1 + deltaT
            if order == 2:
                t = 0.5 * (t + t2)                        # ode_cfl_2.py:200-201
                y = combine(4, deltaT, y, y1, ydot)       # 0.5*(y + (y1 + deltaT*ydot))
            else:
                tHalf = 0.25 * (3 * t + t2)               # ode_cfl_3.py:188-193
                yHalf = combine(2, deltaT, y, y1, ydot)   # 0.25*(3*y + (y1 + deltaT*ydot))
                ydot, stepBound, schemeData = schemeFunc(tHalf, yHalf, schemeData)
                bound_check(deltaT, stepBound, 'Third')
                tThreeHalf = tHalf + deltaT
                t = (1 / 3) * (t + 2 * tThreeHalf)        # :236-241
                y = combine(3, deltaT, y, yHalf, ydot)    # (1/3)*(y + 2*(yHalf + deltaT*ydot))
        steps += 1
        if post:
            y, schemeData = odeCFLcallPostTimestep(t, y, schemeData, options)
        if strcmp(options.singleStep, 'on'):
            break
        if getattr(options, 'terminalEvent', None):
            eventValue, schemeData = options.terminalEvent(t, y, tOld, yOld, schemeData)
            if steps > 1 and np.any(np.sign(eventValue) != np.sign(eventValueOld)):
                break
            eventValueOld = eventValue
    if strcmp(options.stats, 'on'):
        info('%d steps in %.2g seconds from  %.2f to %.2f.' % (steps, cputime() - startTime, tspan[0], t))
    return np.float64(t), y, schemeData


def odeCFL1(schemeFunc, tspan, y0, options=None, schemeData=None):
    """Forward Euler (ode_cfl_1.py:9; the shipped version never stores the new state, :142 --
    this is the intended integrator)."""
    return _integrate(1, schemeFunc, tspan, y0, options, schemeData)


def odeCFL2(schemeFunc, tspan, y0, options=None, schemeData=None):
    """TVD RK2 (ode_cfl_2.py:13)."""
    return _integrate(2, schemeFunc, tspan, y0, options, schemeData)


def odeCFL3(schemeFunc, tspan, y0, options=None, schemeData=None):
    """TVD RK3 (ode_cfl_3.py:11)."""
    return _integrate(3, schemeFunc, tspan, y0, options, schemeData)

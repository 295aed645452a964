"""HJIPDE_solve: the solver front-end that drives the hot path (reference
ValueFuncs/hji_solver.py:24; core loop :509-656).
    data, tau, extraOuts = HJIPDE_solve(data0, tau, schemeData, compMethod, extraArgs)

The value function stays on the GPU for the whole tau loop when the problem can run fused
(native Hamiltonian): RK3 steps are hj_rk_step launches, the post-step operators (min/max with the
previous step, data0, targets; obstacle masking) are hj_minmax_with, the NaN guard is hj_any_nan.
Stopping conditions (stopInit via multilinear point evaluation, stopSetInclude / stopSetIntersect with
stopLevel, stopConverge with ignoreBoundary) and the discounting steps (default and 'Kene' mode,
discountAnneal) are applied to the device-resident state as well; flipOutput reverses the stored time
axis.  Visualisation, noise injection (addGaussianNoiseStandardDeviation) and trajectory extraction are
outside the path.

Deviations from the shipped reference, all listed in SURVEY Appendix D: the integrator honours
the schemeFunc that was built (minWithZero -> termRestrictUpdate; the reference ignores it,
hji_solver.py:542); store-all-times mode keeps time on axis 0 and works (the reference crashes,
:483-484); obstacle masking is elementwise (omax returns a scalar in the reference);
schemeData.CoStateCalc or .derivFunc is honoured if the caller set it, else upwindFirstWENO5;
the stop set is the region where stopSet < 0 (the reference extracts the VALUES there, :260, and its
truncation `tau[i+1:] = []` is not valid NumPy); discounting runs (the reference calls an undefined
`eisfield`, :603) with the toolbox's intended meaning: default mode unless discountMode == 'Kene'.
"""
import ctypes as C
import os

import numpy as np

from . import _ffi
from .context import device_grid, grid_bc, is_tensor
from .dissipation import artificialDissipationGLF
from .integration import odeCFL3, odeCFLset, integrate_span_device
from .spatial import upwindFirstWENO5
from .term import termLaxFriedrichs, termRestrictUpdate, native_plan
from .utilities import Bundle, isfield, error, info, cputime, expand

__all__ = ["HJIPDE_solve"]


def _get(b, name, default=None):
    return getattr(b, name) if (b is not None and isfield(b, name)) else default


class _Ops(object):
    """min/max/NaN on whatever holds y: device tensors through the C ABI, NumPy otherwise."""

    def __init__(self, dg):
        self.dg = dg

    def prep(self, a, like):
        if a is None:
            return None
        if self.dg is not None and is_tensor(like):
            return self.dg.to_device(a).reshape(like.shape)
        return np.asarray(a).reshape(like.shape)

    def minmax(self, op, y, other):
        if self.dg is not None and is_tensor(y):
            self.dg.bind_stream()
            _ffi.check(self.dg.lib.hj_minmax_with(self.dg.ctx, op, self.dg.ptr(y), self.dg.ptr(other),
                                                  y.numel()))
            return y
        if op == _ffi.OP_MIN:
            return np.minimum(y, other)
        if op == _ffi.OP_MAX:
            return np.maximum(y, other)
        return np.maximum(y, -other)

    def has_nan(self, y):
        if self.dg is not None and is_tensor(y):
            flag = C.c_int()
            _ffi.check(self.dg.lib.hj_any_nan(self.dg.ctx, self.dg.ptr(y), y.numel(), C.byref(flag)))
            return bool(flag.value)
        return bool(np.any(np.isnan(y)))


def _eval_point(g, data, x):
    """eval_u(g, data, x) for ONE state (ValueFuncs/evaluate_u.py:15,64: multilinear interpolation,
    periodic axes augmented by one wrapped node and the state wrapped into the period).  `data` may
    live on the GPU: only the 2^dim corner values are read.  Outside an extrapolated axis -> NaN."""
    x = np.asarray(x, dtype=np.float64).ravel()
    if x.size != g.dim:
        error('stopInit must be a vector of length g.dim!')
    bc, _ = grid_bc(g)
    N = [int(v) for v in np.asarray(g.N).ravel()]
    lo, w = [], []
    for d in range(g.dim):
        vs = np.asarray(g.vs[d], dtype=np.float64).ravel()
        dx = float(np.asarray(g.dx).ravel()[d])
        xd = float(x[d])
        if bc[d] == _ffi.BC_PERIODIC:
            period = N[d] * dx
            xd = vs[0] + ((xd - vs[0]) % period)
            i = min(int(np.floor((xd - vs[0]) / dx)), N[d] - 1)
        else:
            if xd < vs[0] or xd > vs[-1]:
                return float('nan')
            i = min(int(np.floor((xd - vs[0]) / dx)), N[d] - 2)
        lo.append(i)
        w.append((xd - (vs[0] + i * dx)) / dx)
    v = 0.0
    for corner in range(1 << g.dim):
        idx, wt = [], 1.0
        for d in range(g.dim):
            up = (corner >> d) & 1
            j = lo[d] + up
            if bc[d] == _ffi.BC_PERIODIC:
                j %= N[d]
            idx.append(j)
            wt *= w[d] if up else (1.0 - w[d])
        if wt != 0.0:
            v += wt * float(data[tuple(idx)])
    return v


def _trim(g, a):
    """truncateGrid(g, a, g.min + 4 dx, g.max - 4 dx) (Grids/truncate.py:8): nodes strictly inside."""
    sl = []
    for d in range(g.dim):
        vs = np.asarray(g.vs[d], dtype=np.float64).ravel()
        dx = float(np.asarray(g.dx).ravel()[d])
        keep = np.nonzero((vs > vs[0] + 4 * dx) & (vs < vs[-1] - 4 * dx))[0]
        sl.append(slice(int(keep[0]), int(keep[-1]) + 1) if keep.size else slice(0, 0))
    return a[tuple(sl)]


def HJIPDE_solve(data0, tau, schemeData, compMethod=None, extraArgs=None):
    extraArgs = extraArgs if extraArgs is not None else Bundle({})
    extraOuts = Bundle({})
    quiet = bool(_get(extraArgs, 'quiet', False))
    keepLast = bool(_get(extraArgs, 'keepLast', False))
    lowMemory = bool(_get(extraArgs, 'lowMemory', False))
    small = 1e-4                                              # hji_solver.py:185
    g = schemeData.grid
    gDim = g.dim
    tau = np.asarray(tau, dtype=np.float64).copy()
    if tau.ndim != 1 or len(tau) < 2:
        error('tau must be a vector of at least two times')
    if np.any(np.diff(tau) < 0):
        error('tau must be non-decreasing')

    # ---- targets / obstacles (hji_solver.py:215-262)
    targets = _get(extraArgs, 'targetFunction')
    obstacles = _get(extraArgs, 'obstacleFunction')
    targ_tv = targets is not None and np.ndim(targets) == gDim + 1
    obs_tv = obstacles is not None and np.ndim(obstacles) == gDim + 1
    if targets is not None and np.ndim(targets) not in (gDim, gDim + 1):
        error('Inconsistent target dimensions!')
    if obstacles is not None and np.ndim(obstacles) not in (gDim, gDim + 1):
        error('Inconsistent obstacle dimensions!')

    stopConverge = bool(_get(extraArgs, 'stopConverge', False))
    convergeThreshold = _get(extraArgs, 'convergeThreshold', 1e-5)
    ignoreBoundary = bool(_get(extraArgs, 'ignoreBoundary', False))
    flipOutput = bool(_get(extraArgs, 'flipOutput', False))
    # stopping sets (hji_solver.py:250-266,686-704)
    stopInit = _get(extraArgs, 'stopInit')
    if stopInit is not None and np.asarray(stopInit).size != gDim:
        error('stopInit must be a vector of length g.dim!')
    stopSet = _get(extraArgs, 'stopSetInclude')
    stopSetAll = stopSet is not None
    if stopSet is None:
        stopSet = _get(extraArgs, 'stopSetIntersect')
    stopLevel = float(_get(extraArgs, 'stopLevel', 0))
    if stopSet is not None:
        stopSet = np.asarray(stopSet.detach().cpu().numpy() if is_tensor(stopSet) else stopSet)
        if stopSet.ndim != gDim or tuple(stopSet.shape) != tuple(g.shape):
            error('Inconsistent stopSet dimensions!')
    # discounting (hji_solver.py:603-638,708-719)
    discountFactor = _get(extraArgs, 'discountFactor')
    discountFactor = float(discountFactor) if discountFactor else None
    discountKene = _get(extraArgs, 'discountMode') == 'Kene'
    discountAnneal = _get(extraArgs, 'discountAnneal')
    if discountFactor is not None and discountKene and targets is None:
        error('Need to define target function l(x)!')

    # ---- scheme (hji_solver.py:424-446)
    schemeFunc = termLaxFriedrichs
    schemeData.dissFunc = artificialDissipationGLF
    if not isfield(schemeData, 'CoStateCalc'):
        if not isfield(schemeData, 'derivFunc'):
            schemeData.derivFunc = upwindFirstWENO5
    sd_run = schemeData
    if compMethod in ('minWithZero', 'zero'):
        schemeFunc = termRestrictUpdate
        sd_run = Bundle(dict(innerFunc=termLaxFriedrichs, innerData=schemeData, positive=0))
    integratorOptions = odeCFLset(Bundle({'factorCFL': 0.8, 'singleStep': 'on'}))
    startTime = cputime()

    # ---- initial data (hji_solver.py:474-507; time axis FIRST in store-all mode)
    # A device tensor stays on the device: the host copy is only made when something needs it (store-all mode, an
    # SDModFunc, the NumPy path) -- a keepLast solve on a tensor never touches PCIe
    dev_in = is_tensor(data0) and data0.is_cuda
    nd0 = data0.dim() if is_tensor(data0) else np.ndim(data0)
    host = {}

    def data0_host():
        if "a" not in host:
            host["a"] = data0.detach().cpu().numpy() if is_tensor(data0) else np.asarray(data0)
        return host["a"]
    if nd0 == gDim:
        first = data0 if dev_in else data0_host()
        istart = 1
        hist = None
    elif nd0 == gDim + 1:
        first = data0[-1] if dev_in else data0_host()[-1]
        istart = int(_get(extraArgs, 'istart', data0.shape[0]))
        hist = data0_host()
    else:
        error('Inconsistent initial condition dimension!')
    if tuple(first.shape) != tuple(g.shape):
        error('data0 does not agree in array size with grid')
    store_all = not keepLast and not lowMemory
    if store_all:
        data = np.zeros((len(tau),) + tuple(g.shape), dtype=np.float64)
        if hist is not None:
            data[:hist.shape[0]] = hist
        else:
            data[0] = data0_host()
    cur_np = None if dev_in else first        # host copy of the current state: kept only where it is consumed (SDModFunc)

    plan = native_plan(schemeData, first)
    dg = device_grid(g, "float64") if plan is not None else None
    ops = _Ops(dg)
    col = (int(np.prod(g.shape)), 1) if schemeFunc is termLaxFriedrichs else (int(np.prod(g.shape)),)
    if dg is not None:
        y = dg.to_device(first).reshape(col).clone()
    else:
        y = np.array(first.detach().cpu().numpy() if is_tensor(first) else first, dtype=np.float64).reshape(col)
    first0 = first if nd0 == gDim else (data0[0] if dev_in else data0_host()[0])
    if dg is not None:
        y_init = dg.to_device(first0).reshape(col).clone()
    else:
        y_init = np.asarray(first0.detach().cpu().numpy() if is_tensor(first0) else first0).reshape(col)

    for i in range(istart, len(tau)):
        if not quiet:
            info('Computing value function at time tau[%d]: %.4f' % (i, tau[i]))
        if isfield(extraArgs, 'SDModFunc'):
            paramsIn = _get(extraArgs, 'SDModParams', [])
            if cur_np is None:
                cur_np = y.detach().cpu().numpy().reshape(g.shape) if is_tensor(y) else np.asarray(y).reshape(g.shape)
            schemeData = extraArgs.SDModFunc(schemeData, i, tau, cur_np, obstacles, paramsIn)
            # the reference hands the reassigned Bundle straight to the integrator (hji_solver.py:512-542):
            # re-apply the defaults and rebuild what was derived from the old Bundle
            schemeData.dissFunc = artificialDissipationGLF
            if not isfield(schemeData, 'CoStateCalc') and not isfield(schemeData, 'derivFunc'):
                schemeData.derivFunc = upwindFirstWENO5
            if schemeFunc is termRestrictUpdate:
                sd_run = Bundle(dict(innerFunc=termLaxFriedrichs, innerData=schemeData, positive=0))
            else:
                sd_run = schemeData
            if schemeData.grid is not g:
                error('SDModFunc must keep schemeData.grid (the stored arrays live on it)')
            new_plan = native_plan(schemeData, y)
            if (new_plan is None) != (plan is None):
                # the state changes sides (device tensor <-> NumPy) with the execution path
                if new_plan is None:
                    y = y.detach().cpu().numpy() if is_tensor(y) else y
                    y_init = y_init.detach().cpu().numpy() if is_tensor(y_init) else y_init
                    dg = None
                else:
                    dg = device_grid(g, "float64")
                    y = dg.to_device(y).reshape(col).clone()
                    y_init = dg.to_device(y_init).reshape(col).clone() if y_init is not None else None
                ops = _Ops(dg)
            plan = new_plan
        # the state the interval started from: only the convergence test reads it (the integrators never write their input)
        y_start = y if stopConverge else None
        tNow = tau[i - 1]
        target_i = ops.prep(targets[i] if targ_tv else targets, y)
        obstacle_i = ops.prep(obstacles[i] if obs_tv else obstacles, y)
        # Fast path: everything that happens between the steps of this interval -- the compMethod min/max
        # (with the previous step, data0 or the target) and the obstacle mask -- is applied by the last RK
        # stage itself, so the whole interval is ONE native call and the NaN guard runs once, at its end
        # (same error, raised a few steps later).  Discounting keeps the step-by-step loop below.
        post, post_a, known = _ffi.POST_NONE, None, True
        if compMethod in (None, 'zero', 'set', 'none', 'minWithZero'):
            pass
        elif compMethod == 'minVOverTime':
            post = _ffi.POST_MIN_PREV
        elif compMethod == 'maxVOverTime':
            post = _ffi.POST_MAX_PREV
        elif compMethod in ('minVWithV0', 'maxVWithV0'):
            post_a = (1 if compMethod == 'minVWithV0' else 2, y_init)
        elif compMethod in ('minVWithL', 'minVwithL', 'minVWithTarget', 'maxVWithL', 'maxVwithL', 'maxVWithTarget'):
            if target_i is None:
                error('Need to define target function l(x)!')
            post_a = (1 if compMethod.startswith('min') else 2, target_i)
        else:
            known = False
        post_b = (3, obstacle_i) if obstacle_i is not None else None
        stepwise = os.environ.get("HJ_HJIPDE_STEPWISE") == "1"     # test knob: force the loop below
        if dg is not None and known and discountFactor is None and not stepwise and tNow < tau[i] - small:
            res = integrate_span_device(schemeFunc, sd_run, y, tNow, tau[i], integratorOptions, small, post,
                                        post_a=post_a, post_b=post_b)
            if res is not None:
                tNow, y = res
                if ops.has_nan(y):
                    error('Nans encountered in the integrated result of HJI PDE data')   # :544-545
        while tNow < tau[i] - small:                           # hji_solver.py:536
            if compMethod in ('minVOverTime', 'maxVOverTime'):
                yLast = y          # the integrators never write their input: no copy needed
            if not quiet:
                info('Cur Time %s bound: %s' % (tNow, tau[i] - small))
            tNow, y, _ = odeCFL3(schemeFunc, [tNow, tau[i]], y, integratorOptions, sd_run)
            if ops.has_nan(y):
                error('Nans encountered in the integrated result of HJI PDE data')   # :544-545
            # ---- compMethod (hji_solver.py:566-599); in 'Kene' discount mode the min/max with the target
            # happens inside the discount step instead (:613-636)
            kene = discountFactor is not None and discountKene
            if kene or compMethod in (None, 'zero', 'set', 'none', 'minWithZero'):
                pass
            elif compMethod == 'minVOverTime':
                y = ops.minmax(_ffi.OP_MIN, y, yLast)
            elif compMethod == 'maxVOverTime':
                y = ops.minmax(_ffi.OP_MAX, y, yLast)
            elif compMethod == 'minVWithV0':
                y = ops.minmax(_ffi.OP_MIN, y, y_init)
            elif compMethod == 'maxVWithV0':
                y = ops.minmax(_ffi.OP_MAX, y, y_init)
            elif compMethod in ('maxVWithL', 'maxVwithL', 'maxVWithTarget'):
                if target_i is None:
                    error('Need to define target function l(x)!')
                y = ops.minmax(_ffi.OP_MAX, y, target_i)
            elif compMethod in ('minVWithL', 'minVwithL', 'minVWithTarget'):
                if target_i is None:
                    error('Need to define target function l(x)!')
                y = ops.minmax(_ffi.OP_MIN, y, target_i)
            else:
                error('Check which compMethod you are using')
            if discountFactor is not None:
                y = _discount(ops, y, discountFactor, discountKene, compMethod, target_i, y_init)
            if obstacle_i is not None:                         # "mask" using obstacles (:641-644)
                y = ops.minmax(_ffi.OP_MAX_NEG, y, obstacle_i)
        cur = y.reshape(g.shape)
        if store_all or isfield(extraArgs, 'SDModFunc'):
            cur_np = cur.detach().cpu().numpy() if is_tensor(cur) else np.asarray(cur)
        if store_all:
            data[i] = cur_np

        def stop_here():
            extraOuts.stoptau = tau[i]
            return tau[:i + 1], (data[:i + 1] if store_all else None)

        change = None
        if stopConverge:                                       # hji_solver.py:661-672
            a, b = cur, y_start.reshape(g.shape)
            if ignoreBoundary:
                a, b = _trim(g, a), _trim(g, b)
            count = a.numel() if is_tensor(a) else a.size
            change = float(abs(a - b).max()) if count else 0.0
            if not quiet:
                info('Max change since last iteration: %s' % change)
        if stopInit is not None:                               # :676-684
            initValue = _eval_point(g, cur, stopInit)
            if not np.isnan(initValue) and initValue <= 0:
                tau, d2 = stop_here()
                data = d2 if store_all else data
                break
        if stopSet is not None:                                # :686-704 (indices where stopSet < 0)
            inside = cur <= stopLevel
            inside = inside.detach().cpu().numpy() if is_tensor(inside) else np.asarray(inside)
            sel = inside[stopSet < 0]
            hit = bool(sel.all()) if stopSetAll else bool(sel.any())
            if sel.size and hit:
                tau, d2 = stop_here()
                data = d2 if store_all else data
                break
        if stopConverge and change < convergeThreshold:        # :706-722
            if discountFactor is not None and discountAnneal and discountFactor != 1:
                if discountAnneal == 'soft':
                    discountFactor = 1 - (1 - discountFactor) / 2
                    if abs(1 - discountFactor) < .00005:
                        discountFactor = 1.0
                else:                                          # 'hard' or 1
                    discountFactor = 1.0
                if not quiet:
                    info('Discount factor: %s' % discountFactor)
            else:
                tau, d2 = stop_here()
                data = d2 if store_all else data
                break
    if not store_all:
        data = cur = y.reshape(g.shape)
        data = data.detach().cpu().numpy() if (is_tensor(data) and not is_tensor(data0)) else data
    elif is_tensor(data0):
        import torch
        data = torch.as_tensor(data, device=data0.device)
    if flipOutput and store_all:
        data = data.flip(0) if is_tensor(data) else np.flip(data, 0).copy()
    endTime = cputime()
    if not quiet:
        info('Total execution time %s seconds' % (endTime - startTime))
    return data, tau, extraOuts


def _discount(ops, y, gamma, kene, compMethod, target_i, y_init):
    """Discounted value iteration steps of HJIPDE_solve (hji_solver.py:603-638).
    Default ('Jaime', ICRA 2019): y <- gamma*y + (1-gamma)*l with l the target (or data0).
    'Kene' (minimum discounted rewards): shift everything below zero by max|l|, discount, take the
    min/max with the shifted target, shift back."""
    if not kene:
        other = target_i if target_i is not None else y_init
        return y * gamma + other * (1.0 - gamma)
    if target_i is None:
        error('Need to define target function l(x)!')
    maxVal = float(abs(target_i).max())
    ytemp = (y - maxVal) * gamma
    ttemp = target_i - maxVal
    if compMethod in ('minVWithL', 'minVwithL', 'minVWithTarget'):
        ytemp = ops.minmax(_ffi.OP_MIN, ytemp, ttemp)
    elif compMethod in ('maxVWithL', 'maxVwithL', 'maxVWithTarget'):
        ytemp = ops.minmax(_ffi.OP_MAX, ytemp, ttemp)
    else:
        error('check your compMethod!')
    return ytemp + maxVal

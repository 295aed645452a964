"""HJIPDE_solve: the solver front-end that drives the hot path (reference
ValueFuncs/hji_solver.py:24; core loop :509-656).
    data, tau, extraOuts = HJIPDE_solve(data0, tau, schemeData, compMethod, extraArgs)

The value function stays on the GPU for the whole tau loop when the problem can run fused
(native Hamiltonian): RK3 steps are hj_rk_step launches, the post-step operators (min/max with the
previous step, data0, targets; obstacle masking) are hj_minmax_with, the NaN guard is hj_any_nan.
Visualisation, trajectory extraction and the other front-end extras are outside the path.

Deviations from the shipped reference, all listed in SURVEY Appendix D: the integrator honours
the schemeFunc that was built (minWithZero -> termRestrictUpdate; the reference ignores it,
hji_solver.py:542); store-all-times mode keeps time on axis 0 and works (the reference crashes,
:483-484); obstacle masking is elementwise (omax returns a scalar in the reference);
schemeData.CoStateCalc or .derivFunc is honoured if the caller set it, else upwindFirstWENO5.
"""
import ctypes as C

import numpy as np

from . import _ffi
from .context import device_grid, is_tensor
from .dissipation import artificialDissipationGLF
from .integration import odeCFL3, odeCFLset
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
    data0_np = data0.detach().cpu().numpy() if is_tensor(data0) else np.asarray(data0)
    if data0_np.ndim == gDim:
        first = data0_np
        istart = 1
        hist = None
    elif data0_np.ndim == gDim + 1:
        first = data0_np[-1]
        istart = int(_get(extraArgs, 'istart', data0_np.shape[0]))
        hist = data0_np
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
            data[0] = first
    cur_np = first

    plan = native_plan(schemeData)
    dg = device_grid(g, "float64") if plan is not None else None
    ops = _Ops(dg)
    col = (int(np.prod(g.shape)), 1) if schemeFunc is termLaxFriedrichs else (int(np.prod(g.shape)),)
    if dg is not None:
        y = dg.to_device(cur_np).reshape(col).clone()
    else:
        y = np.array(cur_np, dtype=np.float64).reshape(col)
    y_init = ops.prep(data0_np if data0_np.ndim == gDim else data0_np[0], y)
    if dg is not None and y_init is not None:
        y_init = y_init.clone()

    for i in range(istart, len(tau)):
        if not quiet:
            info('Computing value function at time tau[%d]: %.4f' % (i, tau[i]))
        if isfield(extraArgs, 'SDModFunc'):
            paramsIn = _get(extraArgs, 'SDModParams', [])
            schemeData = extraArgs.SDModFunc(schemeData, i, tau, cur_np, obstacles, paramsIn)
        y_start = y.clone() if is_tensor(y) else y.copy()
        tNow = tau[i - 1]
        target_i = ops.prep(targets[i] if targ_tv else targets, y)
        obstacle_i = ops.prep(obstacles[i] if obs_tv else obstacles, y)
        while tNow < tau[i] - small:                           # hji_solver.py:536
            if compMethod in ('minVOverTime', 'maxVOverTime'):
                yLast = y.clone() if is_tensor(y) else y
            if not quiet:
                info('Cur Time %s bound: %s' % (tNow, tau[i] - small))
            tNow, y, _ = odeCFL3(schemeFunc, [tNow, tau[i]], y, integratorOptions, sd_run)
            if ops.has_nan(y):
                error('Nans encountered in the integrated result of HJI PDE data')   # :544-545
            # ---- compMethod (hji_solver.py:566-599)
            if compMethod in (None, 'zero', 'set', 'none', 'minWithZero'):
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
            if obstacle_i is not None:                         # "mask" using obstacles (:641-644)
                y = ops.minmax(_ffi.OP_MAX_NEG, y, obstacle_i)
        cur = y.reshape(g.shape)
        if store_all or stopConverge or isfield(extraArgs, 'stopInit') or isfield(extraArgs, 'SDModFunc'):
            cur_np = cur.detach().cpu().numpy() if is_tensor(cur) else np.asarray(cur)
        if store_all:
            data[i] = cur_np
        if stopConverge:
            ys = y_start.detach().cpu().numpy() if is_tensor(y_start) else y_start
            change = float(np.max(np.abs(cur_np.reshape(-1) - ys.reshape(-1))))
            if not quiet:
                info('Max change since last iteration: %s' % change)
            if change < convergeThreshold:
                extraOuts.stoptau = tau[i]
                tau = tau[:i + 1]
                if store_all:
                    data = data[:i + 1]
                break
    if not store_all:
        data = cur = y.reshape(g.shape)
        data = data.detach().cpu().numpy() if (is_tensor(data) and not is_tensor(data0)) else data
    elif is_tensor(data0):
        import torch
        data = torch.as_tensor(data, device=data0.device)
    endTime = cputime()
    if not quiet:
        info('Total execution time %s seconds' % (endTime - startTime))
    return data, tau, extraOuts

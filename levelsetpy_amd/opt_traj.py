"""computeOptTraj (reference ValueFuncs/compute_opt_traj.py:16): the optimal trajectory of a dynamical system
through a stored (time-first) value function -- the step after HJIPDE_solve.

    traj, traj_tau = computeOptTraj(g, data, tau, dynSys, extraArgs)

`data` is time-first and ordered like the toolbox's `dataTraj = flip(data)`: index 0 is the reachable set of
the full horizon, the last index the target (HJIPDE_solve's flipOutput produces it).  At every step the
latest index whose set still contains the current state is found by bisection (find_earliest_BRS_ind), the costate there comes from computeGradients (the HIP upwind
kernels) evaluated at the state by multilinear interpolation (eval_u semantics; only 2^dim corner values are
read from the device), and the system integrates its own dynamics over `subSamples` sub-steps.  dynSys
protocol as the reference calls it (:124-131): attribute `x`; get_opt_u(t, deriv, uMode, x);
get_opt_v(t, deriv, dMode, x) (optional); update_state(u, dt, x, d) returning / storing the new state.

Deviations from the shipped reference, which cannot run: `find_earliest_BRS_ind` is commented out (:88, so
the time index never advances), the disturbance branch reads an undefined `var` (:126), the loop bound
`iter <= tauLength` (:80) writes past `traj`, and `np.any(np.diff(tau)) < 0` (:67) never fires.  Implemented
is the algorithm of the toolbox it ports (helperOC computeOptTraj.m), with the reference's time-first
data layout.  Parity UNPINNED (no reference output exists); checked on a problem with a closed-form
optimal trajectory (tests).  Visualisation (extraArgs.visualize) is outside the path.
"""
import numpy as np

from .gradients import computeGradients
from .hji_solver import _eval_point
from .utilities import isfield, error

__all__ = ["computeOptTraj", "find_earliest_BRS_ind"]


def find_earliest_BRS_ind(g, data, x, upper=None, lower=0):
    """Bisection of the toolbox's find_earliest_BRS_ind (commented out in the reference, :88): `data` is
    time-first with the reachable sets SHRINKING along the time axis (the flipped output of HJIPDE_solve:
    index 0 = the full horizon, last index = the target itself); returns the largest index in [lower, upper]
    whose set still contains x (value < 1e-4), i.e. the earliest moment of the remaining horizon at which x is
    inside.  `lower` if none does."""
    upper = (data.shape[0] - 1) if upper is None else upper
    small = 1e-4
    while upper > lower:
        mid = (upper + lower + 1) // 2
        if _eval_point(g, data[mid], x) < small:
            lower = mid            # inside: everything before mid is settled
        else:
            upper = mid - 1        # too late
    return upper


def computeOptTraj(g, data, tau, dynSys, extraArgs=None):
    uMode = extraArgs.uMode if isfield(extraArgs, 'uMode') else 'min'        # :40-46
    dMode = extraArgs.dMode if isfield(extraArgs, 'dMode') else None         # :48-49
    subSamples = int(extraArgs.subSamples) if isfield(extraArgs, 'subSamples') else 4   # :64-65
    tau = np.asarray(tau, dtype=np.float64).ravel()
    if np.any(np.diff(tau) < 0):
        error('Time stamps must be in ascending order!')                     # :67-68
    if data.shape[0] != len(tau) or tuple(data.shape[1:]) != tuple(g.shape):
        error('data must hold one value function per time stamp (time first)')
    tauLength = len(tau)
    dtSmall = (tau[1] - tau[0]) / subSamples                                 # :73
    traj = np.full((g.dim, tauLength), np.nan)                               # :77-79
    traj[:, 0] = np.asarray(dynSys.x, dtype=np.float64).ravel()
    tEarliest = 0
    it = 0
    while it < tauLength - 1:
        tEarliest = find_earliest_BRS_ind(g, data, np.asarray(dynSys.x).ravel(), tauLength - 1, tEarliest)   # :84-88
        if tEarliest == tauLength - 1:
            break                                                            # the trajectory has entered the target (:114-116)
        BRS_at_t = data[tEarliest]                                           # :91
        Deriv, _, _ = computeGradients(g, BRS_at_t)                          # :119
        for _ in range(subSamples):                                          # :121-131
            x = np.asarray(dynSys.x, dtype=np.float64).ravel()
            deriv = [_eval_point(g, Deriv[d], x) for d in range(g.dim)]
            u = dynSys.get_opt_u(tau[tEarliest], deriv, uMode, x)
            d = dynSys.get_opt_v(tau[tEarliest], deriv, dMode, x) if hasattr(dynSys, 'get_opt_v') else None
            xt = dynSys.update_state(u, dtSmall, x, d)
            if xt is not None:
                dynSys.x = xt
        it += 1                                                              # :134-135
        traj[:, it] = np.asarray(dynSys.x, dtype=np.float64).ravel()
    return traj[:, :it + 1], tau[:it + 1]                                    # :138-139

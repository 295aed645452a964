"""Problem callbacks with a native (fused-kernel) implementation:
DubinsVehicleRel (reference DynamicalSystems/dubins_relative.py:12), DoubleIntegrator
(double_integrator.py:9) and the build-defined DoublePendulum4D (BASELINE config C5).

Each class keeps the reference's callback protocol -- `.hamiltonian(t, data, derivs, sd)` and
`.dissipation(t, data, derivMin, derivMax, sd, dim)` work on NumPy arrays or torch tensors, for
use with foreign terms / dissipation functions -- and additionally advertises `native()` =
(ham_id, params): termLaxFriedrichs and odeCFLn then run the fused HIP kernels instead of
calling back into Python.
"""
import numpy as np

from . import _ffi
from .context import is_tensor

__all__ = ["DubinsVehicleRel", "DoubleIntegrator", "DoublePendulum4D", "native_of", "native_again"]


def native_again(system):
    """(ham_id, params) of a system object as of NOW (its speeds / parameters may have been changed in place), None if it
    is not native any more: what a cached plan is re-validated with (term.native_plan)."""
    att = getattr(system, "_hj_native", None)
    if att is not None:
        par = att.params(system)            # (a traced pair re-traces here and may change its registration: the id is read after)
        return att.reg.ham_id, par
    return system.native()


def _xs(grid, i, like):
    """grid.xs[i] in the array type of `like`."""
    x = grid.xs[i]
    if is_tensor(like):
        import torch
        cache = grid.__dict__.setdefault("_hj_xs_t", {})
        key = (i, like.device, like.dtype)
        if key not in cache:
            cache[key] = torch.as_tensor(np.asarray(x), dtype=like.dtype, device=like.device)
        return cache[key]
    return np.asarray(x)


def _abs(a):
    return a.abs() if is_tensor(a) else np.abs(a)


def _cos(a):
    return a.cos() if is_tensor(a) else np.cos(a)


def _sin(a):
    return a.sin() if is_tensor(a) else np.sin(a)


class DubinsVehicleRel(object):
    """Two Dubins vehicles in relative coordinates (dubins_relative.py:13-61 for the speed
    conventions: v(u)=u*u_bound, w(u)=u*w_bound; scalar bounds give v_e=v_p, w_e=w_p)."""

    def __init__(self, grid, u_bound=5, w_bound=5, x=None):
        self.grid = grid
        self.cur_state = x if x is not None else grid.xs
        self.v = lambda u: u * u_bound
        self.w = lambda w: w * w_bound
        if not np.isscalar(u_bound) and len(u_bound) > 1:
            self.v_e, self.v_p = self.v(1), self.v(-1)
        else:
            self.v_e = self.v_p = self.v(1)
        if not np.isscalar(w_bound) and len(w_bound) > 1:
            self.w_e, self.w_p = self.w(1), self.w(-1)
        else:
            self.w_e = self.w_p = self.w(1)
        self._scalar = np.isscalar(u_bound) and np.isscalar(w_bound)

    def native(self):
        if not self._scalar or self.grid.dim != 3:
            return None
        return _ffi.HAM_DUBINS_REL, [float(self.v_e), float(self.v_p), float(self.w(1)),
                                     float(self.w_e + self.w_p)]

    def hamiltonian(self, t, data, value_derivs, finite_diff_bundle=None):
        """dubins_relative.py:83-90."""
        p1, p2, p3 = value_derivs[0], value_derivs[1], value_derivs[2]
        x1, x2, x3 = (_xs(self.grid, i, p1) for i in range(3))
        p1_coeff = self.v_e - self.v_p * _cos(x3)
        p2_coeff = self.v_p * _sin(x3)
        return (p1 * p1_coeff - p2 * p2_coeff - self.w(1) * _abs(p1 * x2 - p2 * x1 - p3)
                + self.w(1) * _abs(p3))

    def dissipation(self, t, data, derivMin, derivMax, schemeData, dim):
        """dubins_relative.py:104-111."""
        assert dim >= 0 and dim < 3, "Dubins vehicle dimension has to between 0 and 2 inclusive."
        if dim == 0:
            return _abs(self.v_e - self.v_p * _cos(_xs(self.grid, 2, data))) + _abs(self.w(1) * _xs(self.grid, 1, data))
        if dim == 1:
            return _abs(self.v_p * _sin(_xs(self.grid, 2, data))) + _abs(self.w(1) * _xs(self.grid, 0, data))
        return self.w_e + self.w_p


class DoubleIntegrator(object):
    """double_integrator.py:9: xddot = u, |u| <= u_bound."""

    def __init__(self, grid, u_bound=1):
        self.grid = grid
        self.control_law = u_bound

    def native(self):
        if self.grid.dim != 2 or not np.isscalar(self.control_law):
            return None
        return _ffi.HAM_DOUBLE_INTEGRATOR, [float(self.control_law), 0.0, 0.0, 0.0]

    @property
    def switching_curve(self):
        x2 = np.asarray(self.grid.xs[1])
        return -.5 * x2 * np.abs(x2)                       # :44-47

    def hamiltonian(self, t, data, value_derivs, finite_diff_bundle=None):
        x2 = _xs(self.grid, 1, value_derivs[0])
        return -(value_derivs[0] * x2 - _abs(value_derivs[1]) * self.control_law)   # :71-74

    def dissipation(self, t, data, derivMin, derivMax, schemeData, dim):
        return [_abs(_xs(self.grid, 1, data)), abs(self.control_law)][dim]           # :84-89

    def mttr(self):
        """Closed-form minimum time to reach the origin (double_integrator.py:91-119)."""
        x1, x2 = np.asarray(self.grid.xs[0]), np.asarray(self.grid.xs[1])
        gamma = self.switching_curve
        above, below, on = x1 > gamma, x1 < gamma, x1 == gamma
        t1 = (x2 + np.emath.sqrt(4 * x1 + 2 * x2 ** 2)) * above
        t2 = (-x2 + np.emath.sqrt(-4 * x1 + 2 * x2 ** 2)) * below
        t3 = np.abs(x2) * on
        return (t1 + t2 + t3).real


class DoublePendulum4D(object):
    """Build-defined 4-D Hamiltonian (the reference ships none; BASELINE config C5).  State
    (th1, w1, th2, w2); drift of the frictionless double pendulum with unit masses and lengths,
    g = 9.8 (the dynamics written out in the reference's Tests/double_pendulum.py:29-51), torque
    control |u| <= u_max on both angular accelerations:
        H = sum_i p_i f_i(x) + u_max(|p_2| + |p_4|),   alpha_i = |f_i| + u_max [i in {1,3}]."""
    G, L1, L2, M1, M2 = 9.8, 1.0, 1.0, 1.0, 1.0

    def __init__(self, grid, u_max=1.0):
        self.grid = grid
        self.u_max = u_max

    def native(self):
        if self.grid.dim != 4:
            return None
        return _ffi.HAM_DOUBLE_PENDULUM, [float(self.u_max), 0.0, 0.0, 0.0]

    def _drift(self, like):
        th1, w1, th2, w2 = (_xs(self.grid, i, like) for i in range(4))
        G, L1, L2, M1, M2 = self.G, self.L1, self.L2, self.M1, self.M2
        s1, c1, s2, c2 = _sin(th1), _cos(th1), _sin(th2), _cos(th2)
        sd, cd = s2 * c1 - c2 * s1, c2 * c1 + s2 * s1
        den1 = (M1 + M2) * L1 - M2 * L1 * cd * cd
        f1 = (M2 * L1 * w1 * w1 * sd * cd + M2 * G * s2 * cd + M2 * L2 * w2 * w2 * sd
              - (M1 + M2) * G * s1) / den1
        den2 = (L2 / L1) * den1
        f3 = (-M2 * L2 * w2 * w2 * sd * cd + (M1 + M2) * G * s1 * cd
              - (M1 + M2) * L1 * w1 * w1 * sd - (M1 + M2) * G * s2) / den2
        return [w1 + 0 * th1, f1, w2 + 0 * th1, f3]

    def hamiltonian(self, t, data, p, sd=None):
        f = self._drift(p[0])
        return (p[0] * f[0] + p[1] * f[1] + p[2] * f[2] + p[3] * f[3]
                + self.u_max * (_abs(p[1]) + _abs(p[3])))

    def dissipation(self, t, data, derivMin, derivMax, sd, dim):
        return _abs(self._drift(data)[dim]) + (self.u_max if dim in (1, 3) else 0.0)


def native_of(hamFunc, partialFunc):
    """(system, ham_id, params) when hamFunc/partialFunc are the bound methods of ONE of the
    systems above (so the fused kernel computes exactly what the callbacks would), else None.
    A subclass that overrides hamiltonian() / dissipation() / native() is NOT native: the methods are
    compared with the ones of the class that owns the kernel, not with type(self)'s."""
    sys_h = getattr(hamFunc, "__self__", None)
    sys_p = getattr(partialFunc, "__self__", None)
    if sys_h is None or sys_h is not sys_p:
        return None
    att = getattr(sys_h, "_hj_native", None)
    if att is not None:
        # a Hamiltonian registered at run time (user_ham.py): the methods must be the ones the registration saw
        if getattr(hamFunc, "__func__", None) is not att.ham_func or getattr(partialFunc, "__func__", None) is not att.diss_func:
            return None
        return sys_h, att.reg.ham_id, att.params(sys_h)
    owner = next((k for k in (DubinsVehicleRel, DoubleIntegrator, DoublePendulum4D) if isinstance(sys_h, k)), None)
    if owner is None:
        return None
    if getattr(hamFunc, "__func__", None) is not owner.hamiltonian:
        return None
    if getattr(partialFunc, "__func__", None) is not owner.dissipation:
        return None
    if type(sys_h).native is not owner.native:
        return None
    nat = sys_h.native()
    if nat is None:
        return None
    return sys_h, nat[0], nat[1]

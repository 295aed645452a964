"""A user's hamFunc / partialFunc pair as a fused HIP kernel, compiled at run time (C ABI: hj_ham_register).

The reference's termLaxFriedrichs calls arbitrary Python callables on whole arrays
(ExplicitIntegration/Term/term_lax_friedrich.py:111, Dissipation/artificial_diss_glf.py:98); this package fuses the
systems of dynamics.py and sends everything else down the split path (derivative kernels -> the callbacks -> a
dissipation kernel: 25x the fused step at 201^3).  `register_native_hamiltonian` closes that gap: write H and alpha ONCE
more, as a device expression, and the schemeData that carries the system's own bound methods runs fused -- selected by
callable identity, exactly as the built-in systems are (dynamics.native_of).  alpha may depend on x and the parameters
(every system the reference ships) and -- round 5 -- on the costate RANGE the reference's artificialDissipationGLF hands
to partialFunc: `dmin[d]`, `dmax[d]` = derivMin[d], derivMax[d] (artificial_diss_glf.py:80-99).  Such an expression is
detected (or declared: uses_range=True); each substep is then two launches (range pass + fused substep with the in-kernel
max(alpha)), and the integrators take deltaT from the first stage's reduced stepBound as ode_cfl_3.py:142 does.

    DubinsAbs = register_native_hamiltonian("dubins_abs", 3, '''
        H = p[0] * par[0] * cos(x[2]) + p[1] * par[0] * sin(x[2]) + par[1] * fabs(p[2]);
        alpha[0] = fabs(par[0] * cos(x[2]));  alpha[1] = fabs(par[0] * sin(x[2]));  alpha[2] = par[1];
    ''', nparams=2)
    DubinsAbs.attach(vehicle, params=lambda v: [v.v, v.w])     # an EXISTING object with .hamiltonian / .dissipation / .grid
    # or: sys_ = DubinsAbs(grid, [1.0, 1.0], hamiltonian=py_ham, dissipation=py_diss)   # a new object around Python callbacks

In the expression: x[d] node coordinates, p[d] costates (the reference's derivC), par[k] parameters; assign H and every
alpha[d].  Values that depend on the in-plane coordinates only (cos / sin of x[2] above) can be hoisted out of the march:
`column_src="col[0] = cos(x[2]); col[1] = sin(x[2]);", ncol=2` evaluates them once per grid column, `col[k]` is then
readable in the expression (8 % faster for the Dubins systems).  fp64 and fp32, 2-D / 3-D / 4-D grids; the kernels are built with hipRTC on first
use (1-2 s per scheme) and kept on disk (kernel_cache_stats): a later process loads them in milliseconds.
"""
import ctypes as C
import os
import re

from . import _ffi

__all__ = ["register_native_hamiltonian", "NativeRegistration", "RegisteredSystem", "kernel_cache_stats"]

HERE = os.path.dirname(os.path.abspath(__file__))
# bumped by every attach(): a schemeData that was classified "not native" BEFORE its system object was attached must be looked at again
# (term.native_plan caches the classification per schemeData)
ATTACH_EPOCH = [0]


def _hiprtc_path():
    """The libhiprtc.so next to the HIP runtime this process uses (torch's wheel bundles both), else the loader's default."""
    if os.environ.get("HJ_HIPRTC"):
        return os.environ["HJ_HIPRTC"]
    try:
        import torch
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libhiprtc.so")
        if os.path.exists(cand):
            return cand
    except ImportError:
        pass
    return None


class _Attached(object):
    """What native_of needs to recognise an object's bound methods: the registration, the functions the methods must be,
    and where the parameters come from (a list, or a callable of the object: re-read at every call, like system.native())."""

    def __init__(self, reg, params, ham_func, diss_func):
        self.reg, self._params, self.ham_func, self.diss_func = reg, params, ham_func, diss_func

    def params(self, obj):
        p = self._params(obj) if callable(self._params) else self._params
        p = [float(v) for v in p]
        if len(p) != self.reg.nparams:
            raise ValueError("Hamiltonian '%s' takes %d parameters, got %d" % (self.reg.name, self.reg.nparams, len(p)))
        return p


class RegisteredSystem(object):
    """A system object around a registration: the reference's callback protocol (.hamiltonian / .dissipation, from the
    Python callables given -- they serve the split path and parity checks) plus the native kernel."""

    def __init__(self, reg, grid, params, hamiltonian=None, dissipation=None):
        if int(grid.dim) != reg.dim:
            raise ValueError("Hamiltonian '%s' is %d-dimensional, the grid has dim %d" % (reg.name, reg.dim, grid.dim))
        self.grid = grid
        self.params = [float(v) for v in params]
        self._py_ham, self._py_diss = hamiltonian, dissipation
        self._hj_native = _Attached(reg, lambda s: s.params, RegisteredSystem.hamiltonian, RegisteredSystem.dissipation)

    def hamiltonian(self, t, data, value_derivs, finite_diff_bundle=None):
        if self._py_ham is None:
            raise NotImplementedError("no Python hamFunc was given for '%s': it only runs fused" % self._hj_native.reg.name)
        return self._py_ham(self, t, data, value_derivs, finite_diff_bundle)

    def dissipation(self, t, data, derivMin, derivMax, schemeData, dim):
        if self._py_diss is None:
            raise NotImplementedError("no Python partialFunc was given for '%s': it only runs fused" % self._hj_native.reg.name)
        return self._py_diss(self, t, data, derivMin, derivMax, schemeData, dim)


class NativeRegistration(object):
    def __init__(self, name, dim, device_src, nparams=0, column_src=None, ncol=0, uses_range=None):
        self.name, self.dim, self.nparams, self.device_src = str(name), int(dim), int(nparams), str(device_src)
        self.column_src, self.ncol = (str(column_src) if column_src else None), int(ncol)
        # alpha reads the costate range (partialFunc's derivMin / derivMax, artificial_diss_glf.py:80-99): detected from the text
        if uses_range is None:
            uses_range = bool(re.search(r"\bd(min|max)\s*\[", self.device_src))
        self.uses_range = bool(uses_range)
        ham = C.c_int()
        rtc = _hiprtc_path()
        _ffi.check(_ffi.lib().hj_ham_register2(self.name.encode(), self.dim, self.nparams, self.device_src.encode(),
                                               self.column_src.encode() if self.column_src else None, self.ncol,
                                               _ffi.HAM_RANGE if self.uses_range else 0,
                                               os.path.join(HERE, "csrc").encode(), rtc.encode() if rtc else None, C.byref(ham)))
        self.ham_id = int(ham.value)

    def check(self, scheme="WENO5_ASSHIPPED"):
        """Compile now (no GPU needed): raises ValueError with the compiler's message if the expression is wrong."""
        _ffi.check(_ffi.lib().hj_ham_compile_check(self.ham_id, _ffi.SCHEME_IDS[scheme]))
        return self

    def __call__(self, grid, params=(), hamiltonian=None, dissipation=None):
        return RegisteredSystem(self, grid, params, hamiltonian, dissipation)

    def attach(self, obj, params=()):
        """Mark an existing system object (it must have .grid and bound .hamiltonian / .dissipation methods): a schemeData
        whose hamFunc / partialFunc are THOSE methods then runs fused.  params: a list, or a callable of the object."""
        cls = type(obj)
        if not hasattr(obj, "grid"):
            raise ValueError("the system object needs a .grid (the fused path checks that schemeData.grid is the same grid)")
        if int(obj.grid.dim) != self.dim:
            raise ValueError("Hamiltonian '%s' is %d-dimensional, the object's grid has dim %d" % (self.name, self.dim, obj.grid.dim))
        att = _Attached(self, params, getattr(cls, "hamiltonian", None), getattr(cls, "dissipation", None))
        if att.ham_func is None or att.diss_func is None:
            raise ValueError("the system object's class needs hamiltonian() and dissipation() methods")
        att.params(obj)                      # validates the count now
        obj._hj_native = att
        ATTACH_EPOCH[0] += 1
        return obj


def kernel_cache_stats():
    """(kernels compiled with hipRTC, kernels loaded from the on-disk cache) in this process.  The cache lives in
    $HJ_RTC_CACHE, else $XDG_CACHE_HOME/levelsetpy_amd, else ~/.cache/levelsetpy_amd; HJ_RTC_CACHE=0 turns it off."""
    a, b = C.c_int(), C.c_int()
    _ffi.check(_ffi.lib().hj_ham_cache_stats(C.byref(a), C.byref(b)))
    return int(a.value), int(b.value)


def register_native_hamiltonian(name, dim, device_src, nparams=0, column_src=None, ncol=0, uses_range=None):
    """Register H / alpha as a device expression (module docstring); returns a NativeRegistration: call it to make a
    system object, or .attach() it to an existing one.
    column_src / ncol: statements assigning col[0..ncol-1] from x[1..] and par, evaluated once per grid column outside the
    march along axis 0 and readable in device_src -- the place for cos / sin of in-plane coordinates.
    uses_range: the expression reads dmin[d] / dmax[d] (default: detected from the text) -- the general Lax-Friedrichs
    protocol, alpha a function of the costate range; every substep then runs a range pass first (two launches)."""
    return NativeRegistration(name, dim, device_src, nparams, column_src, ncol, uses_range)

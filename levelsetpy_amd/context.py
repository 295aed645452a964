"""Device-side twin of a grid Bundle: owns the hj_ctx (C ABI) for one grid, dtype and
GPU, and marshals NumPy / torch arrays across the boundary.

torch is plumbing here: device memory, the current HIP stream, (in dist.py)
torch.distributed.  All arithmetic on the path runs in libhj_mi355x.so.
"""
import ctypes as C
import os
import weakref

import numpy as np

from . import _ffi
from . import lazy as _lazy
from .lazy import HostView, DeviceArray


def _torch():
    import torch
    return torch


_GPU_OK = [None]       # torch, once torch.cuda.is_available() has answered True (it does not change within a process)


def require_gpu():
    torch = _GPU_OK[0]
    if torch is not None:
        return torch
    torch = _torch()
    if not torch.cuda.is_available():
        raise RuntimeError("levelsetpy_amd needs an AMD GPU (gfx950): torch.cuda.is_available() is "
                           "False and there is no CPU fallback")
    _GPU_OK[0] = torch
    return torch


def _raw_stream_getter(torch):
    """torch's current HIP stream of a device as an integer handle: the private fast accessor when this torch has it
    (0.3 us), else through torch.cuda.current_stream (6 us: it builds a Stream object)."""
    fast = getattr(getattr(torch, "_C", None), "_cuda_getCurrentRawStream", None)
    if fast is not None:
        return fast
    return lambda index: torch.cuda.current_stream(index).cuda_stream


def is_tensor(x):
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


def grid_bc(grid):
    """(bc[], toward_zero[]) from grid.bdry / grid.bdryData (Grids/create_grid.py:61-65,
    add_ghost_extrapolate.py:60-64).  Only the two boundary functions of the path are known."""
    from .boundary import addGhostExtrapolate, addGhostPeriodic
    bc, tz = [], []
    bdata = getattr(grid, "bdryData", None) or [None] * grid.dim
    for i in range(grid.dim):
        f = grid.bdry[i]
        if f is addGhostPeriodic or getattr(f, "__name__", "") == "addGhostPeriodic":
            bc.append(_ffi.BC_PERIODIC)
        elif f is addGhostExtrapolate or getattr(f, "__name__", "") == "addGhostExtrapolate":
            bc.append(_ffi.BC_EXTRAPOLATE)
        else:
            raise ValueError("grid.bdry[%d]=%r: only addGhostExtrapolate / addGhostPeriodic have a "
                             "device implementation" % (i, f))
        gd = bdata[i]
        tz.append(1 if (gd is not None and hasattr(gd, "towardZero") and gd.towardZero) else 0)
    return bc, tz


class DeviceGrid(object):
    """hj_ctx + array marshalling for one (grid, dtype, device[, slab])."""

    def __init__(self, grid, dtype="float64", device=None, slab=None, pad=0):
        torch = require_gpu()
        self.torch = torch
        self.lib = _ffi.lib()
        self.dim = int(grid.dim)
        self.shape = tuple(int(n) for n in np.asarray(grid.N).ravel())
        self.dtype_name = dtype
        self.tdtype = torch.float64 if dtype == "float64" else torch.float32
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.dx = [float(v) for v in np.asarray(grid.dx).ravel()]
        xmin = [float(v) for v in np.asarray(grid.min).ravel()]
        bc, tz = grid_bc(grid)
        self.bc = bc
        vs = [np.ascontiguousarray(np.asarray(v, dtype=np.float64).ravel()) for v in grid.vs]
        # slab = (plane_begin, plane_end, halo_lo, halo_hi): this rank owns planes [b, e) of axis 0
        self.slab = slab
        vs0_ext = None
        if slab is not None:
            b, e, hlo, hhi = slab
            self.shape = (e - b,) + self.shape[1:]
            if pad:
                # coordinates of the pad planes (deep-halo stepper computes on them): the neighbour's
                # nodes, wrapped on a periodic axis; clamped where there is no neighbour (never read)
                n0 = vs[0].size
                ids = np.arange(b - pad, e + pad)
                ids = ids % n0 if bc[0] == _ffi.BC_PERIODIC else np.clip(ids, 0, n0 - 1)
                vs0_ext = np.ascontiguousarray(vs[0][ids])
            vs[0] = np.ascontiguousarray(vs[0][b:e])
            xmin[0] = float(vs[0][0])
        self.numel = int(np.prod(self.shape))
        self.plane = self.numel // self.shape[0]
        n = (C.c_int64 * self.dim)(*self.shape)
        ctx = C.c_void_p()
        _ffi.check(self.lib.hj_ctx_create(C.byref(ctx), self.dim, n, _ffi.darr(xmin), _ffi.darr(self.dx),
                                          (C.c_int * self.dim)(*bc), (C.c_int * self.dim)(*tz),
                                          _ffi.F64 if dtype == "float64" else _ffi.F32,
                                          self.device.index))
        self.ctx = ctx
        self._finalizer = weakref.finalize(self, self.lib.hj_ctx_destroy, ctx)
        for d in range(self.dim):
            _ffi.check(self.lib.hj_ctx_set_coords(ctx, d, vs[d].ctypes.data_as(_ffi._pd)))
        # trig tables computed by NumPy so they are bit-identical to cp.cos(grid.xs[2]) etc.
        if self.dim == 3:
            self._aux(0, np.cos(vs[2]))
            self._aux(1, np.sin(vs[2]))
        elif self.dim == 4:
            self._aux(0, np.sin(vs[0]))
            self._aux(1, np.cos(vs[0]))
            self._aux(2, np.sin(vs[2]))
            self._aux(3, np.cos(vs[2]))
        if slab is not None:
            _ffi.check(self.lib.hj_ctx_set_slab(ctx, int(slab[2]), int(slab[3])))
            if vs0_ext is not None:
                a0 = a1 = None
                if self.dim == 4:       # aux slots 0/1 are sin/cos of the axis-0 node
                    t0, t1 = np.ascontiguousarray(np.sin(vs0_ext)), np.ascontiguousarray(np.cos(vs0_ext))
                    a0, a1 = t0.ctypes.data_as(_ffi._pd), t1.ctypes.data_as(_ffi._pd)
                _ffi.check(self.lib.hj_ctx_set_axis0_pad(ctx, int(pad), vs0_ext.ctypes.data_as(_ffi._pd), a0, a1))
        self._work = {}
        self._raw_stream = _raw_stream_getter(torch)
        self._stream_bound = None       # stream handle the ctx was last bound to (bind_stream skips the C call when unchanged)
        self.bound_gen = -1             # hj_ctx_state_generation right after OUR last write to the ctx's per-call state (forget_if_written_elsewhere)
        self.bound_state = None         # (dissipation kind, post-step state) last written by term._Plan.bind

    def _aux(self, slot, tab):
        tab = np.ascontiguousarray(tab, dtype=np.float64)
        _ffi.check(self.lib.hj_ctx_set_aux(self.ctx, slot, tab.ctypes.data_as(_ffi._pd), tab.size))

    # ------------------------------------------------------------------ marshalling
    def bind_stream(self):
        """Launch on torch's CURRENT stream of this device (asked every call: the caller may have switched streams)."""
        s = self._raw_stream(self.device.index)
        self.forget_if_written_elsewhere()
        if s != self._stream_bound:
            _ffi.check(self.lib.hj_ctx_set_stream(self.ctx, C.c_void_p(s)))
            self._stream_bound = s
            self.bound_gen = self.lib.hj_ctx_state_generation(self.ctx)

    def forget_if_written_elsewhere(self):
        """The ctx is cached and shared: somebody else (a test, a C caller) may have set its stream / dissipation kind / post-step
        operators since this object last did.  The library counts those writes; a count we did not produce voids what we remember."""
        if self.lib.hj_ctx_state_generation(self.ctx) != self.bound_gen:
            self._stream_bound = None
            self.bound_state = None

    def to_device(self, a):
        """NumPy array or torch tensor -> contiguous device tensor of the ctx dtype (flat view ok)."""
        torch = self.torch
        if isinstance(a, HostView):
            # a result of this package handed straight back (the NumPy caller's loop): consumed where it lives
            a = a.device_tensor() if a.device_tensor() is not None else a.__array__()
        elif isinstance(a, DeviceArray) and a.device_tensor() is not None:
            a = a.device_tensor()
        if is_tensor(a):
            t = a
            if t.device != self.device or t.dtype != self.tdtype:
                t = t.to(device=self.device, dtype=self.tdtype)
            return t.contiguous()
        arr = np.ascontiguousarray(a, dtype=np.float64)
        if arr.flags.writeable:
            h = torch.from_numpy(arr)
        else:       # a read-only source (e.g. np.asarray of one of our results): only read here, torch's warning does not apply
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", UserWarning)
                h = torch.from_numpy(arr)
        # a page-locked source (e.g. an array this package returned) goes over at the DMA rate
        return h.to(device=self.device, dtype=self.tdtype, non_blocking=False)

    def like(self, t, proto, shape=None, lazy=False):
        """Return `t` in the array type of `proto` (NumPy in -> NumPy out, tensor in -> tensor out).
        lazy=True (the state / ydot results of the fused path): a NumPy caller gets a HostView -- an ndarray-compatible
        handle that is copied to the host when somebody looks at it and is consumed on the device when it is passed
        back in (lazy.py; HJ_LAZY_NUMPY=0 restores the eager copy).
        The NumPy result lives in page-locked host memory from torch's caching host allocator (the D2H copy
        runs at the DMA rate instead of through a pageable bounce buffer, and an array handed back to the next
        call is recognised as pinned by to_device)."""
        if shape is not None:
            t = t.reshape(shape)
        if is_tensor(proto):
            return t
        if lazy and _lazy.LAZY and t.is_cuda:
            return _lazy.device_array(t) if _lazy.LAZY == "ndarray" else HostView(t.detach())
        t = t.detach()
        nbytes = t.numel() * t.element_size()
        # HJ_PIN_RESULTS=0 turns the page-locked results off; HJ_PIN_MAX_MB caps a single pinned result (default 2048:
        # arrays held by the caller keep their pages locked for their lifetime, and torch's host cache keeps the blocks)
        if t.is_cuda and nbytes >= (1 << 20) and _PIN_RESULTS and nbytes <= _PIN_MAX_BYTES:
            try:
                host = self.torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                host.copy_(t)
                return host.numpy()
            except RuntimeError:        # no page-locked memory to be had (ulimit -l, host pressure): pageable copy
                pass
        return t.cpu().numpy()

    def empty(self, shape=None):
        return self.torch.empty(self.shape if shape is None else shape, dtype=self.tdtype,
                                device=self.device)

    def work(self, key):
        w = self._work.get(key)
        if w is None:
            w = self._work[key] = self.empty()
        return w

    @staticmethod
    def ptr(t):
        return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)

    def sync(self):
        _ffi.check(self.lib.hj_sync(self.ctx))


_PIN_RESULTS = os.environ.get("HJ_PIN_RESULTS", "1") != "0"
_PIN_MAX_BYTES = int(float(os.environ.get("HJ_PIN_MAX_MB", "2048")) * (1 << 20))


def _current_device(torch):
    f = getattr(getattr(torch, "_C", None), "_cuda_getDevice", None)
    return f() if f is not None else torch.cuda.current_device()


def device_grid(grid, dtype="float64"):
    """Cached DeviceGrid of a grid Bundle (kept on the Bundle itself)."""
    cache = grid.__dict__.get("_hj_device")
    if cache is None:
        cache = {}
        object.__setattr__(grid, "_hj_device", cache)
    torch = require_gpu()
    key = (dtype, _current_device(torch))
    dg = cache.get(key)
    if dg is None:
        dg = cache[key] = DeviceGrid(grid, dtype)
    return dg


def array_dtype_name(a):
    """'float32' only when the caller hands float32 data explicitly; the reference path is fp64
    (ghost functions force float64: add_ghost_extrapolate.py:77)."""
    if isinstance(a, (HostView, DeviceArray)):
        a = a.device_tensor() if a.device_tensor() is not None else (None if isinstance(a, HostView) else a)
    if a is not None and is_tensor(a):
        return "float32" if str(a.dtype) == "torch.float32" else "float64"
    return "float64"

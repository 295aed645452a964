"""Multi-GPU time stepping: 1-D slab decomposition of axis 0 with ghost-plane halo exchange.

No reference counterpart (the reference is single-process).  One process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests):

  * axis 0 is the slowest-varying axis (C order), so a slab is one contiguous block and a ghost
    plane is one contiguous run of N1*...*N_{D-1} values;
  * every rank stores its slab with HJ_STENCIL (3) pad planes below and above; the pads on an
    internal slab face are filled by the neighbour's edge planes (`HaloExchanger`), the pads on a
    physical boundary are unused (the kernel synthesises extrapolated ghosts there).  With a
    periodic axis 0 the ring closes rank P-1 <-> 0;
  * per RK substep: (1) the EDGE plane ranges [0,3) and [n-3,n) of the output are computed first,
    (2) their exchange is started on a second stream while (3) the INTERIOR planes are computed
    on the compute stream, (4) the next substep waits for both.  Only nearest neighbours talk
    (<= 2 of the 7 xGMI links), 3*plane bytes per direction;
  * scalars: all native Hamiltonians have data-independent alpha, so stepBound is one
    all-reduce(MAX) of D numbers at set-up; true WENO5 adds one all-reduce(MAX) of D numbers per
    substep for its 'maxOverGrid' epsilon.

`SlabIntegrator` holds the decomposition / exchange / RK sequencing and is device-agnostic: the
arithmetic comes from a backend object.  `HipSlabBackend` (the product) calls libhj_mi355x.so; the
CPU tests plug in an oracle-backed backend to exercise the same sequencing under gloo.
"""
import ctypes as C
import os
import sys

import numpy as np

from . import _ffi

HALO = _ffi.STENCIL


class SlabDecomposition(object):
    """Planes [begin, end) of axis 0 owned by `rank`; the remainder is spread over the low ranks
    (513 = 8*64+1 -> rank 0 gets 65 planes)."""

    def __init__(self, n0, world, rank, periodic0=False, self_exchange=False):
        """self_exchange: with ONE rank and a periodic axis 0, close the ring through the transport
        (rank 0 <-> rank 0) instead of wrapping inside the kernel -- exercises the RCCL path on a
        single GPU (tests / tools); results are identical."""
        if world < 1 or not (0 <= rank < world):
            raise ValueError("bad rank/world")
        base, rem = divmod(int(n0), world)
        if base < HALO:
            raise ValueError("axis 0 has %d planes: fewer than %d per rank for %d ranks" % (n0, HALO, world))
        self.n0, self.world, self.rank, self.periodic0 = int(n0), world, rank, bool(periodic0)
        self.counts = [base + (1 if r < rem else 0) for r in range(world)]
        self.begin = sum(self.counts[:rank])
        self.end = self.begin + self.counts[rank]
        self.n_local = self.counts[rank]
        # neighbours (None on a physical boundary)
        ring = periodic0 and (world > 1 or self_exchange)
        self.lo = rank - 1 if rank > 0 else (world - 1 if ring else None)
        self.hi = rank + 1 if rank < world - 1 else (0 if ring else None)

    @property
    def halo_lo(self):
        return self.lo is not None

    @property
    def halo_hi(self):
        return self.hi is not None


class HaloExchanger(object):
    """Fills the pad planes of a padded slab buffer (n_local + 2*HALO planes) from the neighbours.

    Message order per pair is fixed so that the case lo == hi (two ranks, periodic axis) matches:
    sends  [my low planes -> lo, my high planes -> hi]; receives [hi pad <- hi, lo pad <- lo]."""

    def __init__(self, slab, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.slab = slab
        self.group = group
        # gloo moves device tensors through host staging buffers with copies of its own that are NOT ordered after the
        # caller's current stream: a send could read edge planes whose kernel had not finished (seen in the two-rank rehearsal
        # on one card: one stale halo plane now and then).  With gloo the stream is drained before the sends are posted and
        # after the receives have completed; RCCL ("nccl") is stream-ordered and needs neither.
        try:
            self.host_staged = dist.is_initialized() and dist.get_backend(group) != "nccl"
        except Exception:  # noqa: BLE001
            self.host_staged = True

    def start(self, buf):
        """Post the sends/receives for `buf`; returns the request list (wait with finish())."""
        dist, s = self.dist, self.slab
        n = s.n_local
        ops = []
        if s.lo is not None:
            ops.append(dist.P2POp(dist.isend, buf[HALO:2 * HALO], s.lo, group=self.group, tag=0))
        if s.hi is not None:
            ops.append(dist.P2POp(dist.isend, buf[n:n + HALO], s.hi, group=self.group, tag=1))
        if s.hi is not None:
            ops.append(dist.P2POp(dist.irecv, buf[n + HALO:n + 2 * HALO], s.hi, group=self.group, tag=0))
        if s.lo is not None:
            ops.append(dist.P2POp(dist.irecv, buf[0:HALO], s.lo, group=self.group, tag=1))
        if not ops:
            return []
        if self.host_staged and buf.is_cuda:
            import torch
            torch.cuda.current_stream(buf.device).synchronize()
        return dist.batch_isend_irecv(ops)

    def finish(self, reqs):
        for r in reqs:
            r.wait()
        if self.host_staged and reqs:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()

    def exchange(self, buf):
        self.finish(self.start(buf))


class SlabIntegrator(object):
    """odeCFL{1,2,3} single steps on a slab-decomposed grid (native Hamiltonians)."""

    # (stage, y source, y0 source, destination) per substep; buffers named cur/w0/w1/nxt
    PLANS = {
        1: [(_ffi.STAGE_EULER, "cur", None, "nxt")],
        2: [(_ffi.STAGE_EULER, "cur", None, "w0"), (_ffi.STAGE_RK2_FULL, "w0", "cur", "nxt")],
        3: [(_ffi.STAGE_EULER, "cur", None, "w0"), (_ffi.STAGE_RK3_HALF, "w0", "cur", "w1"),
            (_ffi.STAGE_RK3_FULL, "w1", "cur", "nxt")],
    }

    def __init__(self, slab, backend, dx, order=3, factor_cfl=0.8, group=None, needs_eps=False,
                 exchanger=None, allreduce_max=None, dynamic=False, diss="glf"):
        """exchanger / allreduce_max: transport overrides (tests run several ranks inside one
        process); default is torch.distributed (RCCL on GPUs, gloo on CPU).
        dynamic: the Hamiltonian's alpha depends on the costate range (user_ham.register_native_hamiltonian with dmin / dmax:
        artificial_diss_glf.py:80-99).  The range is a property of the WHOLE grid: before every substep each rank reduces
        derivL / derivR of its slab (backend.range_pass, pads in place), the ranks all-reduce (MAX) and every launch of the substep
        reads the reduced range (backend.set_range); deltaT comes from the all-reduced max(alpha) of the first stage's range.
        diss (with dynamic): "glf" as above; "llf" / "lllf" = artificialDissipationLLF / LLLF of such a Hamiltonian
        (diss_local_laxfried.py:108-128, diss_localsq_laxfried.py:87-107): alpha is evaluated with every node's OWN costate range in
        dimension i (LLF: the all-reduced grid range in the others; LLLF: the node's own everywhere -- no range pass, no range
        all-reduce), so stepBound is 1 / max over ALL ranks of max_x sum_i alpha_i(x) / dx_i: a bound pass of the slab before the first
        stage (backend.local_bound) and one scalar all-reduce."""
        import torch
        self.torch = torch
        self.slab, self.be, self.order, self.factor_cfl = slab, backend, order, factor_cfl
        self.group = group
        self.needs_eps = needs_eps
        self.ex = exchanger if exchanger is not None else HaloExchanger(slab, group)
        if allreduce_max is None:
            def allreduce_max(t):
                import torch.distributed as dist
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        self._allreduce_max = allreduce_max
        self.buf = {k: backend.alloc() for k in ("cur", "w0", "w1", "nxt")}
        self.dynamic, self.dx = bool(dynamic), [float(v) for v in dx]
        self.diss = str(diss).lower()
        if self.diss not in ("glf", "llf", "lllf"):
            raise ValueError("diss must be 'glf', 'llf' or 'lllf'")
        if self.diss != "glf" and not self.dynamic:
            raise ValueError("diss='%s' belongs to dynamic=True (native systems: the backend's static local bound)" % self.diss)
        if self.dynamic:
            backend.set_dissipation(self.diss)
            self.alpha_max, self.step_bound = None, None       # per step (step())
            return
        # stepBound = 1 / sum_d max_grid(alpha_d)/dx_d with the max over ALL ranks
        # (artificial_diss_glf.py:101-109); alpha is data-independent for the native systems
        amax = torch.tensor(backend.local_alpha_max(), dtype=torch.float64, device=backend.device)
        if slab.world > 1:
            self._allreduce_max(amax)
        self.alpha_max = [float(v) for v in amax.cpu()]
        self.step_bound = 1.0 / sum(a / d for a, d in zip(self.alpha_max, dx))

    # -------------------------------------------------------------------------------------
    def set_state(self, local_planes):
        """local_planes: array of this rank's n_local planes; fills `cur` and its halos."""
        n = self.slab.n_local
        self.buf["cur"][HALO:HALO + n].copy_(local_planes)
        self.ex.exchange(self.buf["cur"])
        self.be.sync()

    def state(self):
        n = self.slab.n_local
        return self.buf["cur"][HALO:HALO + n]

    def _eps(self, src):
        """global max(D1^2) per dim for true WENO5 (needs src's halos in place)."""
        v = self.be.max_d1sq(src)
        if self.slab.world > 1:
            self._allreduce_max(v)
        self.be.set_weno_eps(v)

    def step(self, t, tf=float("inf"), max_step=float("inf")):
        """One odeCFLn step: dt = min(factorCFL*stepBound, tf - t, maxStep) (ode_cfl_3.py:142)."""
        be, s = self.be, self.slab
        n = s.n_local
        dt = None if self.dynamic else min(self.factor_cfl * self.step_bound, tf - t, max_step)
        for k, (stage, ysrc, y0src, dst) in enumerate(self.PLANS[self.order]):
            y, y0, out = self.buf[ysrc], (self.buf[y0src] if y0src else None), self.buf[dst]
            if self.needs_eps:
                self._eps(y)
            if self.dynamic:
                # the costate range of the stage's input over the whole grid, then (first stage) deltaT from max(alpha) of that range
                if self.diss != "lllf":
                    rng = be.range_pass(y)
                    if s.world > 1:
                        self._allreduce_max(rng)
                    be.set_range(rng)
                if k == 0 and self.diss != "glf":
                    inv = self.torch.tensor([1.0 / be.local_bound(y)], dtype=self.torch.float64, device=be.device)
                    if s.world > 1:
                        self._allreduce_max(inv)
                    self.step_bound = 1.0 / float(inv.cpu()[0])
                    dt = min(self.factor_cfl * self.step_bound, tf - t, max_step)      # ode_cfl_3.py:142
                elif k == 0:
                    amax = self.torch.tensor(be.alpha_max_now(), dtype=self.torch.float64, device=be.device)
                    if s.world > 1:
                        self._allreduce_max(amax)
                    self.alpha_max = [float(v) for v in amax.cpu()]
                    self.step_bound = 1.0 / sum(a / d for a, d in zip(self.alpha_max, self.dx))
                    dt = min(self.factor_cfl * self.step_bound, tf - t, max_step)      # ode_cfl_3.py:142
            # (1) edge planes first: they are what the neighbours wait for
            lo_e = min(HALO, n) if s.halo_lo else 0
            hi_b = max(n - HALO, lo_e) if s.halo_hi else n
            if lo_e > 0:
                be.substep(stage, dt, y, y0, out, 0, lo_e)
            if hi_b < n:
                be.substep(stage, dt, y, y0, out, hi_b, n)
            # (2) start the exchange of `out`'s edge planes on the side stream
            reqs = be.on_comm_stream(lambda: self.ex.start(out))
            # (3) interior planes overlap with the exchange
            if hi_b > lo_e:
                be.substep(stage, dt, y, y0, out, lo_e, hi_b)
            # (4) the next substep reads out's pads
            be.join_comm(reqs, self.ex.finish)
        self.buf["cur"], self.buf["nxt"] = self.buf["nxt"], self.buf["cur"]
        if self.order == 1:
            return t + dt, dt
        t1 = t + dt
        t2 = t1 + dt
        if self.order == 2:
            return 0.5 * (t + t2), dt                       # ode_cfl_2.py:200
        t_half = 0.25 * (3 * t + t2)                       # ode_cfl_3.py:188
        return (1.0 / 3.0) * (t + 2 * (t_half + dt)), dt   # :221,236


class HipSlabBackend(object):
    """The product backend: padded device buffers + hj_rk_substep over plane ranges."""

    def __init__(self, grid, slab, scheme_id, ham_id, ham_params, dtype="float64", overlap=True):
        import torch
        from .context import DeviceGrid
        self.torch = torch
        self.slab = slab
        self.dg = DeviceGrid(grid, dtype, None, (slab.begin, slab.end, slab.halo_lo, slab.halo_hi))
        self.device = self.dg.device
        self.sid, self.ham, self.par = scheme_id, ham_id, _ffi.darr(ham_params)
        self.plane_shape = self.dg.shape[1:]
        self.n = slab.n_local
        self.comm_stream = torch.cuda.Stream(device=self.device) if overlap else None
        self._slot = 0
        self._eps = None
        self.dg.bind_stream()

    def alloc(self):
        return self.torch.zeros((self.n + 2 * HALO,) + tuple(self.plane_shape), dtype=self.dg.tdtype,
                                device=self.device)

    def _interior_ptr(self, buf):
        return C.c_void_p(buf[HALO:].data_ptr()) if buf is not None else C.c_void_p(0)

    def substep(self, stage, dt, y, y0, out, p0, p1):
        dg = self.dg
        self._slot = (self._slot + 1) % (_ffi.BOUND_SLOTS - 1)
        _ffi.check(dg.lib.hj_rk_substep(dg.ctx, self.sid, self.ham, self.par, 0.0, stage, float(dt), 0,
                                        self._interior_ptr(y), self._interior_ptr(y0),
                                        self._interior_ptr(out), self._slot, int(p0), int(p1)))

    def local_alpha_max(self):
        dg = self.dg
        sb = C.c_double()
        am = (C.c_double * 4)()
        _ffi.check(dg.lib.hj_static_step_bound(dg.ctx, self.ham, self.par, C.byref(sb), am))
        return [am[d] for d in range(dg.dim)]

    # ---- Hamiltonians whose alpha reads the costate range (SlabIntegrator(dynamic=True))
    def range_pass(self, y):
        """derivL / derivR of this slab (pads read where it has neighbours) reduced to 2*ndim order-preserving keys, returned as
        an int64 tensor whose element-wise MAX over ranks is the reduction (the keys are unsigned: the sign bit is flipped so that
        signed MAX orders them; set_range flips it back)."""
        dg, torch = self.dg, self.torch
        keys = torch.zeros(8, dtype=torch.int64, device=self.device)
        _ffi.check(dg.lib.hj_range_pass(dg.ctx, self.sid, self.ham, self.par, self._interior_ptr(y), C.c_void_p(keys.data_ptr())))
        return keys.bitwise_xor_(torch.tensor(-2 ** 63, dtype=torch.int64, device=self.device))

    def set_range(self, keys_signed):
        torch = self.torch
        self._range = keys_signed.bitwise_xor(torch.tensor(-2 ** 63, dtype=torch.int64, device=self.device))   # keep alive
        _ffi.check(self.dg.lib.hj_ctx_set_range_source(self.dg.ctx, C.c_void_p(self._range.data_ptr())))

    def alpha_max_now(self):
        dg = self.dg
        am = (C.c_double * 4)()
        _ffi.check(dg.lib.hj_range_alpha_max(dg.ctx, self.ham, self.par, am))
        return [am[d] for d in range(dg.dim)]

    def set_dissipation(self, kind):
        _ffi.check(self.dg.lib.hj_ctx_set_dissipation(self.dg.ctx, {"glf": _ffi.DISS_GLF, "llf": _ffi.DISS_LLF, "lllf": _ffi.DISS_LLLF}[kind]))

    def local_bound(self, y):
        """1 / max_x sum_d alpha_d(x) / dx_d over this slab under the local rule (hj_bound_pass; LLF: after set_range)."""
        dg = self.dg
        sb = C.c_double()
        _ffi.check(dg.lib.hj_bound_pass(dg.ctx, self.sid, self.ham, self.par, self._interior_ptr(y), C.byref(sb)))
        return sb.value

    def max_d1sq(self, y):
        dg = self.dg
        v = self.torch.empty(dg.dim, dtype=dg.tdtype, device=self.device)
        _ffi.check(dg.lib.hj_max_d1sq(dg.ctx, self._interior_ptr(y), C.c_void_p(v.data_ptr())))
        return v

    def set_weno_eps(self, v):
        self._eps = v   # keep alive
        _ffi.check(self.dg.lib.hj_ctx_set_weno_eps_source(self.dg.ctx, C.c_void_p(v.data_ptr())))

    def on_comm_stream(self, fn):
        torch = self.torch
        if self.comm_stream is None:
            return fn()
        cur = torch.cuda.current_stream(self.device)
        self.comm_stream.wait_stream(cur)           # edge planes are complete before they are sent
        with torch.cuda.stream(self.comm_stream):
            return fn()

    def join_comm(self, reqs, finish):
        torch = self.torch
        if self.comm_stream is None:
            finish(reqs)
            return
        with torch.cuda.stream(self.comm_stream):
            finish(reqs)                            # NCCL: orders the comm stream after the transfers
        torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)

    def sync(self):
        self.torch.cuda.synchronize(self.device)


def rccl_library_path():
    """The librccl.so this process already has loaded (torch's), so the native stepper shares it."""
    import os
    import torch
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return cand if os.path.exists(cand) else None


class NativeSlabStepper(object):
    """Slab time stepping with the halo exchange in native code (RCCL ncclSend/ncclRecv over xGMI posted
    by the C library on its own high-priority stream; one C call per RK step, no Python on the critical
    path; torch.distributed only broadcasts the ncclUniqueId).

    deep=True (default for slabs below 20 M cells): hj_slab_rk_step_deep -- ONE exchange of 3*order planes per step, the stages
    recompute the few planes they need beyond the slab, interior launches never wait for the exchange.
    deep=False: hj_slab_rk_step -- one 3-plane exchange per substep (edge-first overlap).
    external=True: no communicator; `exchange` (a callable taking this stepper) moves the pad planes
    (tests: several virtual ranks in one process)."""

    def __init__(self, grid, slab, scheme_id, ham_id, ham_params, dx, dtype="float64", order=3,
                 factor_cfl=0.8, group=None, deep=None, external=None):
        import torch
        import torch.distributed as dist
        from .context import DeviceGrid
        self.torch = torch
        self.slab, self.order, self.factor_cfl = slab, order, factor_cfl
        if deep is None:
            # measured on a self ring (tools/slab_self.py): one deep exchange per step wins on small slabs
            # (201^3: 0.163 vs 0.198 ms/step), the per-substep exchange on large ones (401^3: 1.01 vs 1.08),
            # where the redundant planes cost more than the dependency bubbles they remove
            env = os.environ.get("HJ_SLAB_DEEP")
            cells = slab.n_local * int(np.prod([int(v) for v in np.asarray(grid.N).ravel()[1:]]))
            # (and not on thin slabs: the 6*order redundant planes are a fixed cost per slab)
            deep = (env != "0") if env is not None else (cells < 20e6 and slab.n_local >= 128)
        self.deep = bool(deep)
        if external is not None and not self.deep:
            raise ValueError("an external transport can only serve the deep-halo schedule (one exchange per step)")
        self.pad = HALO * order if self.deep else HALO
        self.external = external
        if self.deep and slab.n_local < 2 * self.pad and (slab.halo_lo or slab.halo_hi):
            raise ValueError("slab of %d planes is too thin for the deep-halo stepper (needs %d)" % (slab.n_local, 2 * self.pad))
        self.dg = dg = DeviceGrid(grid, dtype, None, (slab.begin, slab.end, slab.halo_lo, slab.halo_hi),
                                  pad=self.pad if self.deep else 0)
        self.device = dg.device
        self.sid, self.ham, self.par = scheme_id, ham_id, _ffi.darr(ham_params)
        self.n = slab.n_local
        lib = dg.lib
        dg.bind_stream()
        lo = -1 if slab.lo is None else slab.lo
        hi = -1 if slab.hi is None else slab.hi
        if external is not None:
            _ffi.check(lib.hj_comm_init_external(dg.ctx, slab.rank, slab.world, lo, hi))
        else:
            path = rccl_library_path()
            cpath = path.encode() if path else None
            uid = torch.zeros(128, dtype=torch.uint8)
            if slab.rank == 0:
                buf = (C.c_char * 128)()
                _ffi.check(lib.hj_comm_unique_id(cpath, buf))
                uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
            if slab.world > 1:
                if dist.get_backend(group) == "nccl":
                    uid = uid.to(self.device)
                dist.broadcast(uid, src=0, group=group)
            raw = bytes(uid.cpu().numpy().tobytes())
            _ffi.check(lib.hj_comm_init(dg.ctx, cpath, slab.rank, slab.world, raw, lo, hi))
        shape = (self.n + 2 * self.pad,) + tuple(dg.shape[1:])
        self.buf = {k: torch.zeros(shape, dtype=dg.tdtype, device=self.device) for k in ("cur", "w1", "nxt")}
        # stepBound from the all-reduced per-dimension alpha maxima (artificial_diss_glf.py:101-109)
        sb, am = C.c_double(), (C.c_double * 4)()
        _ffi.check(lib.hj_static_step_bound(dg.ctx, self.ham, self.par, C.byref(sb), am))
        self.alpha_local = [am[d] for d in range(dg.dim)]
        amax = torch.tensor(self.alpha_local, dtype=torch.float64, device=self.device)
        if slab.world > 1 and external is None:
            dist.all_reduce(amax, op=dist.ReduceOp.MAX, group=group)
        self.dx = list(dx)
        self.set_alpha_max([float(v) for v in amax.cpu()])

    def set_alpha_max(self, alpha_max):
        self.alpha_max = list(alpha_max)
        self.step_bound = 1.0 / sum(a / d for a, d in zip(self.alpha_max, self.dx))

    def _ip(self, buf):
        return C.c_void_p(buf[self.pad:].data_ptr())

    def _exchange(self, buf):
        self.dg.bind_stream()
        if self.external is not None:
            _ffi.check(self.dg.lib.hj_slab_join(self.dg.ctx))
            self.torch.cuda.synchronize(self.device)
            self.external(self)
        elif self.deep:
            _ffi.check(self.dg.lib.hj_halo_exchange_depth(self.dg.ctx, self._ip(buf), self.pad))
        else:
            _ffi.check(self.dg.lib.hj_halo_exchange(self.dg.ctx, self._ip(buf)))

    def set_state(self, local_planes):
        self.buf["cur"][self.pad:self.pad + self.n].copy_(local_planes)
        self._exchange(self.buf["cur"])
        self.torch.cuda.synchronize(self.device)

    def state(self):
        _ffi.check(self.dg.lib.hj_slab_join(self.dg.ctx))     # the last exchange may still be in flight
        return self.buf["cur"][self.pad:self.pad + self.n]

    def step(self, t, tf=float("inf"), max_step=float("inf")):
        dt = min(self.factor_cfl * self.step_bound, tf - t, max_step)
        b = self.buf
        fn = self.dg.lib.hj_slab_rk_step_deep if self.deep else self.dg.lib.hj_slab_rk_step
        # three arrays: for RK3 the first stage buffer doubles as the output (see bench.py); RK2's second
        # stage reads the first stage buffer as its stencil input, so it uses the spare array
        w0 = b["nxt"] if self.order == 3 else b["w1"]
        _ffi.check(fn(self.dg.ctx, self.order, self.sid, self.ham, self.par, float(dt), 0,
                      self._ip(b["cur"]), self._ip(b["nxt"]), self._ip(w0), self._ip(b["w1"])))
        b["cur"], b["nxt"] = b["nxt"], b["cur"]
        if self.external is not None:      # the caller's transport fills the pads of the new state
            self._exchange(b["cur"])
        if self.order == 1:
            return t + dt, dt
        t2 = (t + dt) + dt
        if self.order == 2:
            return 0.5 * (t + t2), dt
        return (1.0 / 3.0) * (t + 2 * (0.25 * (3 * t + t2) + dt)), dt

    @property
    def nranks(self):
        """size of the communicator as RCCL reports it (ncclCommCount)."""
        n = C.c_int()
        _ffi.check(self.dg.lib.hj_comm_info(self.dg.ctx, None, C.byref(n), None, None))
        return int(n.value)

    def close(self):
        _ffi.check(self.dg.lib.hj_comm_destroy(self.dg.ctx))


def _slab_self_check(g, slab, integ, wl, tdtype, device, steps=2):
    """max |slab result - single-domain result| over this rank's planes after `steps` RK3 steps."""
    import torch
    from .context import DeviceGrid
    t = 0.0
    for _ in range(steps):
        t, _dt = integ.step(t)
    mine = integ.state().clone()
    torch.cuda.synchronize(device)
    n0 = slab.n0
    dg = DeviceGrid(g, wl["dtype"])
    dg.bind_stream()
    cur = wl["planes"](torch, g, 0, n0, tdtype, device)
    assert cur.shape[0] == n0
    nxt, w1 = torch.empty_like(cur), torch.empty_like(cur)
    tout, dtout = C.c_double(), C.c_double()
    par = _ffi.darr(wl["par"])
    ts = 0.0
    for _ in range(steps):
        _ffi.check(dg.lib.hj_rk_step(dg.ctx, 3, _ffi.SCHEME_IDS[wl["scheme"]], wl["ham"], par, ts, 1e9, 0.8,
                                     1e300, 0, dg.ptr(cur), dg.ptr(nxt), dg.ptr(nxt), dg.ptr(w1),
                                     C.byref(tout), C.byref(dtout)))
        cur, nxt = nxt, cur
        ts = float(tout.value)
    torch.cuda.synchronize(device)
    diff = float((cur[slab.begin:slab.end] - mine).abs().max())
    if os.environ.get("HJ_DEBUG_SLABCHECK") and diff > 0:
        d = (cur[slab.begin:slab.end] - mine).abs()
        per_plane = d.reshape(d.shape[0], -1).max(dim=1).values
        bad = [int(i) for i in torch.nonzero(per_plane > 0).ravel()[:12]]
        sys.stderr.write("[slabcheck] rank %d planes [%d,%d) n_local %d: diff %g, first bad local planes %s, t slab %r ref %r, dt %r, "
                         "step_bound %r\n" % (slab.rank, slab.begin, slab.end, slab.n_local, diff, bad, t, ts, _dt, getattr(integ, "step_bound", None)))
    if abs(ts - t) > 1e-14 * max(1.0, abs(t)):
        diff = max(diff, abs(ts - t))
    del cur, nxt, w1
    return diff


def _cylinder_planes(torch, g, b, e, tdtype, device):
    """shapeCylinder(g, 2, 0, .5) restricted to axis-0 planes [b, e), built on the device (cylinder.py:55-59)."""
    x0 = torch.as_tensor(np.asarray(g.vs[0]).ravel()[b:e], device=device).reshape(-1, 1, 1)
    x1 = torch.as_tensor(np.asarray(g.vs[1]).ravel(), device=device).reshape(1, -1, 1)
    n2 = int(np.asarray(g.N).ravel()[2])
    d = (x0 * x0 + x1 * x1).sqrt() - 0.5
    return d.expand(e - b, x1.shape[1], n2).to(tdtype).contiguous()


def _sphere_planes(torch, g, b, e, tdtype, device, radius=0.5):
    """shapeSphere(g, 0, radius) restricted to axis-0 planes [b, e), built on the device (sphere.py:50-57)."""
    dim = int(g.dim)
    acc = None
    for i, v in enumerate(g.vs):
        v = np.asarray(v).ravel()
        if i == 0:
            v = v[b:e]
        view = [1] * dim
        view[i] = -1
        sq = torch.as_tensor(v, device=device).reshape(view) ** 2
        acc = sq if acc is None else acc + sq
    return (acc.sqrt() - radius).to(tdtype).contiguous()


def slab_grid(L, world, n, global_n):
    """The bench grid: strong scaling (global_n > 0) integrates the global_n^3 Dubins grid of BASELINE C4
    whatever the rank count; weak scaling gives every rank an n^3 slab of an (world*n) x n x n grid."""
    if global_n > 0:
        n = n0 = int(global_n)
        gmax0 = 3.25
    else:
        n0 = world * n
        gmax0 = -.75 + 4.0 / (n - 1) * (n0 - 1)
    gmin = np.array([[-.75, -1.25, -np.pi]]).T
    gmax = np.array([[gmax0, 1.25, np.pi * (1 - 2 / n)]]).T
    return L.createGrid(gmin, gmax, np.array([[n0], [n], [n]], dtype=np.int64), 2, low_mem=True), n0, n


def slab_workload(L, name, world, args, global_n):
    """What bench.py's slab leg integrates.  "C4": BASELINE config 4, the Dubins-relative 3-D grid (fp64, axis 0
    extrapolated: the end ranks synthesise their ghosts).  "C5": BASELINE config 5, the double-pendulum 4-D grid, fp32,
    ALL axes periodic -- the slab ring closes rank P-1 <-> 0 and the drift's sin/cos tables of the axis-0 node are
    per-slab (context.py); `global_n` is the points per axis (129)."""
    if name == "C4":
        g, n0, n = slab_grid(L, world, args.n, global_n)
        return {"name": "C4", "grid": g, "n0": n0, "plane": n * n, "periodic0": False, "ham": _ffi.HAM_DUBINS_REL,
                "par": [1.0, 1.0, 1.0, 2.0], "scheme": args.scheme, "dtype": args.dtype, "planes": _cylinder_planes,
                "grid_txt": ("%d x %d x %d" % (n0, n, n)),
                "desc": "Dubins-relative (air3D) 3-D HJI, %s + GLF, odeCFL3 (factorCFL 0.8), cylinder r=0.5 initial data" % args.scheme}
    if name == "C5":
        n = int(global_n) if global_n and global_n > 0 else 129
        gmin = np.array([[-np.pi, -8, -np.pi, -8]]).T
        gmax = np.array([[np.pi * (1 - 2 / n), 8 * (1 - 2 / n), np.pi * (1 - 2 / n), 8 * (1 - 2 / n)]]).T
        g = L.createGrid(gmin, gmax, n * np.ones((4, 1), dtype=np.int64), [0, 1, 2, 3], low_mem=True)
        return {"name": "C5", "grid": g, "n0": n, "plane": n ** 3, "periodic0": True, "ham": _ffi.HAM_DOUBLE_PENDULUM,
                "par": [1.0, 0.0, 0.0, 0.0], "scheme": "WENO5_ASSHIPPED", "dtype": "float32", "planes": _sphere_planes,
                "grid_txt": "%d^4" % n,
                "desc": "double pendulum 4-D HJI, fp32, all axes periodic (slab ring closed), WENO5_ASSHIPPED + GLF, odeCFL3 "
                        "(factorCFL 0.8), sphere r=0.5 initial data"}
    raise ValueError("unknown slab workload %r (C4, C5)" % (name,))


# measured on ONE MI355X with the ring closed through a real RCCL self send/recv (tools/thin_slab_ring.py, slabs of the 513^3 Dubins
# grid, WENO5_ASSHIPPED, fp64; profiles/r04_thin_slab_gated.txt, profiles/r05_thin_slab.txt): planes -> ms per RK3 step of the schedule
# bench_slab picks for that thickness.  Everything but the link is in these numbers.
SELF_RING_MS_C4 = [(64, 0.357), (65, 0.362), (129, 0.616), (257, 1.099), (513, 1.745)]
# ... and the slabs of C5 (129^4 fp32, planes of 129^3 cells, all axes periodic; round 6, profiles/r06_thin_slab.txt): per-substep schedule
# (the deep-halo stepper is slower on every one of them: 9 of a 17-plane slab's planes would be recomputed)
SELF_RING_MS_C5 = [(17, 0.787), (33, 1.148), (65, 2.051), (129, 3.40)]
XGMI_LINK_GBS = 153.0        # per link and neighbour, peak (the pool's figure for MI355X: 7 links x ~153 GB/s per GPU)


def plan_substep(N, bc, dtype, scheme, ham, stage, p0, p1, halo_lo=False, halo_hi=False, num_cus=256):
    """The launch plan of one substep over planes [p0, p1) -- kernel, workgroups, tiles, chunks -- from the C library's own launch
    code run WITHOUT a device (hj_plan_substep).  Returns a dict."""
    import ctypes as C
    out = (C.c_int64 * 12)()
    name = C.create_string_buffer(64)
    nd = len(N)
    _ffi.check(_ffi.lib().hj_plan_substep(nd, (C.c_int64 * nd)(*[int(v) for v in N]), (C.c_int * nd)(*[int(v) for v in bc]),
                                          _ffi.F64 if str(dtype) in ("float64", "f64") else _ffi.F32, int(scheme), int(ham), int(stage),
                                          int(p0), int(p1), int(bool(halo_lo)), int(bool(halo_hi)), int(num_cus), out, name, 64))
    v = [int(x) for x in out]
    return {"kernel": name.value.decode(), "threads": v[0], "workgroups": v[1], "tiles": v[2], "chunks": v[3], "chunk_planes": v[4],
            "tile": [e for e in v[5:8] if e > 0], "lds_bytes": v[8], "workgroups_per_cu": v[9],
            "rounds": (v[1] + num_cus * max(1, v[9]) - 1) // (num_cus * max(1, v[9]))}


def plan_slab_run(args, world, global_n=513, workload="C4", num_cus=256):
    """bench.py --gpus N --plan-only: what every rank of the N-rank slab leg will do -- its slab, the stepper and schedule bench_slab
    picks, the launches of one substep (from the library's launch code, no device), the halo bytes it exchanges per step, and a
    predicted ms/step from the single-GPU self-ring measurements (C4) with the link time at the xGMI peak beside it.  No GPU, no
    process group: the first real N-GPU run can be checked against this line by line."""
    import levelsetpy_amd as L
    wl = slab_workload(L, workload, world, args, global_n)
    g, n0, plane = wl["grid"], wl["n0"], wl["plane"]
    esz = 8 if wl["dtype"] == "float64" else 4
    sid = _ffi.SCHEME_IDS[wl["scheme"]]
    nd = int(g.dim)
    Ng = [int(v) for v in np.asarray(g.N).ravel()]
    bc = [0, 0, 1] if wl["name"] == "C4" else [1] * nd       # (slab_grid / slab_workload: heading periodic; the 4-D grid all periodic)
    counts = SlabDecomposition(n0, world, 0, wl["periodic0"], self_exchange=wl["periodic0"]).counts
    thick = min(counts) >= 128 and wl["name"] == "C4"
    ranks = []
    for r in range(world):
        sl = SlabDecomposition(n0, world, r, wl["periodic0"], self_exchange=wl["periodic0"])
        n = sl.n_local
        lo, hi = sl.halo_lo, sl.halo_hi
        Nl = [n] + Ng[1:]
        deep = thick and n >= 18 and (lo or hi)
        ent = {"rank": r, "planes": [sl.begin, sl.end], "n_local": n, "lo": sl.lo, "hi": sl.hi,
               "cells": n * plane, "slab_bytes_per_array": n * plane * esz}
        nb = int(lo) + int(hi)
        if deep:
            ent["stepper"] = "native deep-halo (hj_slab_rk_step_deep): ONE 9-plane exchange per RK3 step"
            ent["halo_bytes_sent_per_step"] = nb * 9 * plane * esz
            ent["launches_per_step"] = []
            for st in (1, 2, 3):
                ext = 3 * (3 - st)
                ent["launches_per_step"].append({
                    "stage": st,
                    "interior": plan_substep(Nl, bc, wl["dtype"], sid, wl["ham"], _ffi.STAGE_EULER if st == 1 else _ffi.STAGE_RK3_FULL,
                                             3 * st, n - 3 * st, lo, hi, num_cus),
                    "edges": plan_substep(Nl, bc, wl["dtype"], sid, wl["ham"], _ffi.STAGE_EULER if st == 1 else _ffi.STAGE_RK3_FULL,
                                          -ext if lo else 0, 3 * st, lo, hi, num_cus) if nb else None,
                    "edge_planes_each_side": 3 * st + ext})
        else:
            sched = "serial (edges, then interior, on one stream; exchange beside the interior)" if n >= 192 else \
                    "overlap (edges on the high-priority stream beside the interior)"
            ent["stepper"] = "native per-substep (hj_slab_rk_step): a 3-plane exchange per substep, %s" % sched if nb else \
                             "single slab, no exchange"
            ent["halo_bytes_sent_per_step"] = nb * 3 * 3 * plane * esz
            lo_e, hi_b = (3 if lo else 0), (n - 3 if hi else n)
            edges = plan_substep(Nl, bc, wl["dtype"], sid, wl["ham"], _ffi.STAGE_EULER, 0 if lo else hi_b, 3 if lo else n, lo, hi, num_cus) if nb else None
            if edges and nb == 2:      # both edge ranges ride in ONE launch (hj_api.hip, slab_substep): twice the chunks
                edges["chunks"] *= 2
                edges["workgroups"] *= 2
                edges["rounds"] = (edges["workgroups"] + num_cus * max(1, edges["workgroups_per_cu"]) - 1) // (num_cus * max(1, edges["workgroups_per_cu"]))
            ent["launches_per_substep"] = {
                "interior": plan_substep(Nl, bc, wl["dtype"], sid, wl["ham"], _ffi.STAGE_EULER, lo_e, hi_b, lo, hi, num_cus),
                "edges": edges, "edge_ranges": [[0, lo_e]] * int(lo) + [[hi_b, n]] * int(hi)}
        # prediction: the self-ring table (C4's 513^2-cell planes), scaled by plane size for other grids
        table, plane0 = None, 1.0
        if wl["name"] == "C4" and wl["scheme"] == "WENO5_ASSHIPPED" and wl["dtype"] == "float64":
            table, plane0 = SELF_RING_MS_C4, float(513 * 513)
        elif wl["name"] == "C5":
            table, plane0 = SELF_RING_MS_C5, float(129 ** 3)
        if table is not None:
            xs = [a for a, _ in table]
            ys = [b for _, b in table]
            # (below the thinnest measured slab: that slab's time scaled by the plane count -- an optimistic guess, the edges do not shrink)
            ms = float(np.interp(n, xs, ys)) * (plane / plane0) * (min(1.0, n / float(xs[0])))
            ent["self_ring_ms_per_step"] = round(ms, 4)
        else:
            ent["self_ring_ms_per_step"] = None
        per_dir = ent["halo_bytes_sent_per_step"] / max(1, nb)
        ent["link_ms_per_step_at_peak"] = round(per_dir / (XGMI_LINK_GBS * 1e9) * 1e3, 4) if nb else 0.0
        ranks.append(ent)
    known = [e["self_ring_ms_per_step"] for e in ranks if e["self_ring_ms_per_step"] is not None]
    slow = max(known) if len(known) == len(ranks) else None
    link = max(e["link_ms_per_step_at_peak"] for e in ranks)
    total = n0 * plane
    return {"plan_only": True, "workload": wl["name"], "grid": wl["grid_txt"], "scheme": wl["scheme"], "dtype": wl["dtype"], "n_gpus": world,
            "planes_per_rank": counts, "ranks": ranks,
            "predicted": {
                "ms_per_step_compute_self_ring": slow,
                "ms_per_step_link_at_peak": link,
                "link_hidden_if": "the exchange (per substep: 3 planes each way) finishes under the interior launch; at the xGMI peak of "
                                  "%.0f GB/s per link it needs %.3f ms per step against %s ms of compute" % (XGMI_LINK_GBS, link, slow),
                "value_cell_substeps_per_s": (total * 3 / (max(slow, link) * 1e-3)) if slow else None,
                "basis": "tools/thin_slab_ring.py on one MI355X (RCCL self send/recv, everything but the link): "
                         + ", ".join("%d planes %.3f ms" % ab for ab in (SELF_RING_MS_C5 if wl["name"] == "C5" else SELF_RING_MS_C4))}}


def plane_cells(wl):
    return int(wl["plane"])


def _agree(dist, ok, device):
    """Collective decision: True only if EVERY rank reports ok (a rank that failed locally must not leave
    the others inside a different collective: ADVICE r01)."""
    return _agree3(dist, ok, device) == "all"


def _agree3(dist, ok, device):
    """'all' if every rank reports ok, 'none' if every rank failed, 'some' otherwise."""
    import torch
    flag = torch.tensor([0.0 if ok else 1.0, 1.0 if ok else 0.0], dtype=torch.float64, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    any_failed, any_ok = float(flag[0].item()) != 0.0, float(flag[1].item()) != 0.0
    return "all" if not any_failed else ("some" if any_ok else "none")


def bench_slab(args, rank, world, global_n=513, workload="C4", transport=None, diagnostics=True):
    """bench.py's slab leg.  Workload C4 (default) -- global_n > 0: STRONG scaling of the global_n^3 Dubins grid (BASELINE
    C4: 513^3, slabs of 65/64 planes at 8 ranks); global_n == 0: weak scaling, every rank owns an n^3 slab.  Workload C5:
    the 4-D double-pendulum grid (129^4 fp32, all axes periodic; 17/16-plane slabs at 8 ranks), strong scaling."""
    import time
    import torch
    import torch.distributed as dist
    import levelsetpy_amd as L
    wl = slab_workload(L, workload, world, args, global_n)
    g, n0 = wl["grid"], wl["n0"]
    # one rank with a periodic axis 0: the ring closes rank 0 <-> rank 0 through the transport, so that N = 1 of the
    # slab leg runs the same exchange code as N > 1
    slab = SlabDecomposition(n0, world, rank, wl["periodic0"], self_exchange=wl["periodic0"])
    dxs = [float(v) for v in np.asarray(g.dx).ravel()]
    sid, par, ham, dtype = _ffi.SCHEME_IDS[wl["scheme"]], wl["par"], wl["ham"], wl["dtype"]
    device = torch.device("cuda", torch.cuda.current_device())

    def make(kind):
        if kind == "torch":
            be = HipSlabBackend(g, slab, sid, ham, par, dtype)
            return SlabIntegrator(slab, be, dxs, 3, 0.8, needs_eps=(wl["scheme"] == "WENO5")), be, \
                "3-plane halo exchange per substep, torch.distributed P2P over %s, edge-first overlap" % (
                    "RCCL" if dist.get_backend() == "nccl" else dist.get_backend())
        deep = {"native-deep": True, "native": False}[kind]
        it = NativeSlabStepper(g, slab, sid, ham, par, dxs, dtype, 3, 0.8, deep=deep)
        how = ("ONE 9-plane exchange per RK3 step, stages recompute the planes beyond the slab" if deep
               else "3-plane exchange per substep, edge-first overlap")
        return it, it, how + ", ncclSend/ncclRecv over RCCL inside the C library"

    # Transports in order of preference; each one has to reproduce the single-domain result ON THIS HARDWARE
    # before it is timed (two RK3 steps of the whole grid on every rank, untimed).  Every stage ends in a
    # collective agreement: a rank whose set-up raised reports it there instead of skipping ahead, so the
    # ranks never sit in different collectives.  A set-up that failed on EVERY rank moves on to the next transport; one
    # that failed on some ranks only (communicator creation half done) cannot be recovered from and aborts the run with
    # a non-zero exit.
    want = transport or os.environ.get("HJ_SLAB_TRANSPORT")
    # the deep-halo schedule pays 18 redundant planes per slab and step: only worth it on thick slabs
    thick = min(slab.counts) >= 128 and wl["name"] == "C4"
    order = [want] if want else ((["native-deep"] if thick else []) + ["native", "torch"])
    integ = be = how = None
    check = float("inf")
    y_init = None
    last_err = None

    def drop(it):
        # a stepper that is given up owns an RCCL communicator and side streams: release them before the next one
        if it is not None and hasattr(it, "close"):
            try:
                it.close()
            except Exception:  # noqa: BLE001
                pass

    for kind in order:
        ok, err = True, None
        integ = None
        try:
            integ, be, how = make(kind)
            y_init = wl["planes"](torch, g, slab.begin, slab.end, be.dg.tdtype, be.device)
            integ.set_state(y_init)
        except Exception as e:  # noqa: BLE001 -- decided collectively below
            ok, err = False, e
            last_err = e
            sys.stderr.write("[bench_slab] rank %d: transport %s failed to set up: %r\n" % (rank, kind, e))
        verdict = _agree3(dist, ok, device)
        if verdict == "some":
            # a rank that got past its communicator while another did not cannot be re-synchronised safely
            raise RuntimeError("slab transport %s could not be set up on every rank (%r)" % (kind, err))
        if verdict == "none":
            # every rank failed the same set-up (e.g. the RCCL library could not be opened): nobody sits in a communicator,
            # the next transport can be tried
            if rank == 0:
                sys.stderr.write("[bench_slab] transport %s could not be set up on any rank (%r)\n" % (kind, err))
            drop(integ)        # make() may have succeeded (communicator, streams) before a later set-up step raised
            integ = None
            continue
        bad = 0.0
        if os.environ.get("HJ_BENCH_SLAB_CHECK", "1") != "0":
            try:
                bad = _slab_self_check(g, slab, integ, wl, be.dg.tdtype, be.device)
            except Exception as e:  # noqa: BLE001
                sys.stderr.write("[bench_slab] rank %d: self check of %s raised: %r\n" % (rank, kind, e))
                bad = float("inf")
        worst = torch.tensor([bad], dtype=torch.float64, device=device)
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        check = float(worst.item())
        if check <= (1e-12 if dtype == "float64" else 0.0):    # (fp32: the decomposition is bitwise or it is wrong)
            integ.set_state(y_init)
            dist.barrier()
            break
        if rank == 0:
            sys.stderr.write("[bench_slab] transport %s rejected: max |slab - single domain| = %g\n" % (kind, check))
        drop(integ)
        integ = None
    if integ is None:
        if check == float("inf") and last_err is not None:
            raise RuntimeError("no slab transport could be set up (last error: %r)" % (last_err,))
        raise RuntimeError("no slab transport reproduced the single-domain result (last diff %g)" % check)
    t = 0.0
    # untimed device spin-up (clock ramp), as in bench.py's single-GPU leg
    for _ in range(int(os.environ.get("HJ_BENCH_SPINUP", "300")) + args.warmup):
        t, _dt = integ.step(t)
    walls, devs = [], []
    for _ in range(max(1, args.repeats)):
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(args.steps):
            t, _dt = integ.step(t)
        e1.record()
        torch.cuda.synchronize()
        dist.barrier()
        walls.append(time.perf_counter() - t0)
        devs.append(e0.elapsed_time(e1))
    # ---- where the time of a step goes, for the FIRST real N > 1 run: the exchange alone and the launches alone, 20
    # iterations each, per rank (max over ranks beside it) -- a bad scaling curve can then be read as link or kernel without a second lease.
    # Every rank runs the same sequence; a failure is recorded, never raised (the timed figure above is already taken).
    diag = None
    if diagnostics and isinstance(integ, NativeSlabStepper) and integ.external is None:
        diag = {}
        try:
            lib, ctx, nloc = integ.dg.lib, integ.dg.ctx, integ.n
            _ffi.check(lib.hj_slab_join(ctx))
            torch.cuda.synchronize()

            def timed(fn, iters=20):
                torch.cuda.synchronize()
                dist.barrier()
                t0d = time.perf_counter()
                for _ in range(iters):
                    fn()
                torch.cuda.synchronize()
                ms = 1e3 * (time.perf_counter() - t0d) / iters
                both = torch.tensor([ms, -ms], dtype=torch.float64, device=device)
                dist.all_reduce(both, op=dist.ReduceOp.MAX)
                return {"this_rank_ms": round(ms, 4), "max_over_ranks_ms": round(float(both[0]), 4), "min_over_ranks_ms": round(-float(both[1]), 4)}

            def exch():
                integ._exchange(integ.buf["cur"])
                _ffi.check(lib.hj_slab_join(ctx))
            if slab.halo_lo or slab.halo_hi:
                planes = integ.pad
                diag["exchange_only"] = dict(timed(exch), planes_each_way=planes, bytes_each_way=planes * plane_cells(wl) * (8 if dtype == "float64" else 4),
                                             what="the halo exchange of one %s, alone (send + receive to each neighbour, then the join)"
                                                  % ("RK3 step (deep halo)" if integ.deep else "substep"))
            lo_e = 3 if slab.halo_lo else 0
            hi_b = nloc - 3 if slab.halo_hi else nloc
            b = integ.buf

            def sub(p0, p1):
                def f():
                    _ffi.check(lib.hj_rk_substep(ctx, sid, ham, integ.par, 0.0, _ffi.STAGE_RK3_FULL, 1e-4, 0, integ._ip(b["cur"]),
                                                 integ._ip(b["cur"]), integ._ip(b["w1"]), 5, int(p0), int(p1)))
                return f
            if hi_b > lo_e:
                diag["interior_launch_only"] = dict(timed(sub(lo_e, hi_b)), planes=[lo_e, hi_b], kernel=(lib.hj_last_kernel(ctx) or b"?").decode(),
                                                    what="one substep over the planes that need no pad, alone")
            if lo_e > 0:
                diag["low_edge_launch_only"] = dict(timed(sub(0, lo_e)), planes=[0, lo_e])
            if hi_b < nloc:
                diag["high_edge_launch_only"] = dict(timed(sub(hi_b, nloc)), planes=[hi_b, nloc])
            diag["whole_slab_launch_only"] = dict(timed(sub(0, nloc)), planes=[0, nloc], kernel=(lib.hj_last_kernel(ctx) or b"?").decode())
        except Exception as e:  # noqa: BLE001
            diag["error"] = repr(e)
    ok = bool(torch.isfinite(integ.state()).all())
    # ranks of the transport that moved the halos: RCCL's own count (ncclCommCount) for the native steppers, the
    # process group's for the torch.distributed one
    nranks = integ.nranks if hasattr(integ, "nranks") else dist.get_world_size()
    kern = be.dg.lib.hj_last_kernel(be.dg.ctx)
    if hasattr(integ, "close"):
        integ.close()
    assert ok, "non-finite state after the timed steps"
    plane = wl["plane"]
    deep = bool(getattr(integ, "deep", False))
    return {"walls": walls, "devs": devs, "total_cells": n0 * plane, "local_cells": slab.n_local * plane, "transport": kind, "diagnostics": diag,
            "alternate_transport": (None if kind == "torch" else ("native" if kind == "native-deep" else
                                                                 ("native-deep" if min(slab.counts) >= 18 else None))),
            "planes": "/".join(str(c) for c in sorted(set(slab.counts), reverse=True)),
            "slab_check_max_abs_diff": check, "nranks": nranks, "workload": wl,
            "kernel": kern.decode() if kern else "?",
            "launches": "3 whole-slab launches incl. the redundant pad planes" if deep else
                        "per substep one launch for the edge planes and one for the interior",
            "parallelism": "slab%d (axis-0 slabs; %s)" % (world, how)}

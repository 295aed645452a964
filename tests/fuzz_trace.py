"""Fuzzer for the callback tracer (levelsetpy_amd/trace_ham.py): random hamFunc / partialFunc pairs built from the operations array code
uses, as closures over NumPy (and, on the GPU, torch) arrays.

    python tests/fuzz_trace.py [seconds] [seed] [cpu|gpu]

cpu: the traced graph evaluated in NumPy (Traced.evaluate) against the callbacks on real arrays -- thousands of cases per minute.
gpu: termLaxFriedrichs with the callbacks traced / compiled / fused against the same schemeData on the split path (HJ_TRACE=0), random
     scheme, dissipation variant, dimension 2..4 and dtype -- a hipRTC compile per case.
Exit code 1 and the seed of the failing case on a mismatch."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import levelsetpy_amd as L  # noqa: E402
from levelsetpy_amd import trace_ham as TH  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
mode = sys.argv[3] if len(sys.argv) > 3 else "cpu"
torch = None
if mode == "gpu":
    import torch  # noqa: E402


def _is_t(a):
    return type(a).__module__.startswith("torch")


def lib_of(a):
    return torch if (torch is not None and _is_t(a)) else np


class Expr(object):
    """A random expression tree; __call__(env) evaluates it with the array library of the operands (NumPy, torch or symbolic)."""

    def __init__(self, rng, depth, leaves, smooth):
        self.rng, self.leaves, self.smooth = rng, leaves, smooth
        self.tree = self.gen(depth)

    def gen(self, depth):
        r = self.rng
        if depth <= 0 or r.random() < 0.18:
            k = r.random()
            if k < 0.7:
                return ("leaf", self.leaves[int(r.integers(len(self.leaves)))])
            if k < 0.9:
                return ("float", float(np.round(r.uniform(-2, 2), 3)))
            return ("int", int(r.integers(-2, 4)))
        ops = ["add", "sub", "mul", "abs", "neg", "cos", "sin", "tanh", "square", "sqrtabs", "expneg", "div", "max", "min", "where", "gate", "clip", "pow3", "sign"]
        if self.smooth:          # (H: no jumps -- a jump in H moves an ENO / LF result by O(1) where two arithmetics differ in the last bit of p)
            ops = [o for o in ops if o not in ("gate", "sign", "where")]
        op = ops[int(r.integers(len(ops)))]
        n = {"add": 2, "sub": 2, "mul": 2, "div": 2, "max": 2, "min": 2, "where": 4, "gate": 2, "clip": 1}.get(op, 1)
        return (op,) + tuple(self.gen(depth - 1) for _ in range(n))

    def __call__(self, env):
        return self.ev(self.tree, env)

    def ev(self, t, env):
        op = t[0]
        if op == "leaf":
            return env[t[1]]
        if op in ("float", "int"):
            return t[1]
        a = [self.ev(q, env) for q in t[1:]]
        like = next((v for v in a if not isinstance(v, (int, float))), None)
        if like is None:                       # numbers only: plain Python arithmetic would not be array code -- anchor it on an array
            like = env[self.leaves[0]]
            a[0] = a[0] + 0 * like
        xp = lib_of(like)

        def arr(v):                            # torch functions want tensors -- of the working dtype (full_like of an integer tensor would truncate)
            if xp is np or not isinstance(v, (int, float)):
                return v
            return torch.full(like.shape, float(v), dtype=env["x0"].dtype, device=like.device)
        if op == "add":
            return a[0] + a[1]
        if op == "sub":
            return a[0] - a[1]
        if op == "mul":
            return a[0] * a[1]
        if op == "neg":
            return -a[0]
        if op == "abs":
            return abs(a[0])
        if op == "cos":
            return xp.cos(arr(a[0]))
        if op == "sin":
            return xp.sin(arr(a[0]))
        if op == "tanh":
            return xp.tanh(arr(a[0]))
        if op == "square":
            return a[0] ** 2
        if op == "pow3":
            return a[0] ** 3
        if op == "sqrtabs":
            return xp.sqrt(abs(arr(a[0])) + 0.25)
        if op == "expneg":
            return xp.exp(-abs(arr(a[0])))
        if op == "div":
            return a[0] / (2.5 + xp.cos(arr(a[1])))
        if op == "max":
            return xp.maximum(arr(a[0]), arr(a[1]))
        if op == "min":
            return xp.minimum(arr(a[0]), arr(a[1]))
        if op == "where":
            return xp.where(arr(a[0]) > arr(a[1]), arr(a[2]), arr(a[3]))
        if op == "gate":
            c = arr(a[0]) > 0.1
            # (a torch bool tensor times a Python float is float32 whatever the data is; NumPy gives float64: say which one is meant)
            return (c.to(env["x0"].dtype) if xp is not np else c) * a[1]
        if op == "clip":
            return xp.clip(arr(a[0]), -0.75, 1.25) if xp is np else arr(a[0]).clamp(-0.75, 1.25)
        if op == "sign":
            return xp.sign(arr(a[0]))
        raise AssertionError(op)


class RandomSystem(object):
    def __init__(self, grid, rng, dim, use_range):
        self.grid, self.dim = grid, dim
        self.scale = float(np.round(rng.uniform(0.5, 1.5), 3))
        # per-axis arrays computed before the call (as a system's __init__ would): cos of the coordinate
        self.stored = [np.cos(1.7 * np.asarray(grid.vs[d]).ravel()).reshape([-1 if k == d else 1 for k in range(dim)]) for d in range(dim)]
        pl = ["p%d" % d for d in range(dim)]
        xl = ["x%d" % d for d in range(dim)] + ["c%d" % int(rng.integers(dim))]
        self.H = Expr(rng, int(rng.integers(2, 5)), pl + pl + xl, smooth=True)
        rl = (["lo%d" % d for d in range(dim)] + ["hi%d" % d for d in range(dim)]) if use_range else []
        # (an alpha that JUMPS as a function of the costate range would move the LF term by O(1) at nodes where the two paths' ranges differ in the last bit)
        self.A = [Expr(rng, int(rng.integers(1, 4)), xl + rl, smooth=use_range) for _ in range(dim)]

    def _coords(self, like):
        out = {}
        for d in range(self.dim):
            x = self.grid.xs[d]
            c = self.stored[d]
            if torch is not None and _is_t(like):
                x = torch.as_tensor(np.asarray(x), device=like.device, dtype=like.dtype)
                c = torch.as_tensor(c, device=like.device, dtype=like.dtype)
            out["x%d" % d] = x
            out["c%d" % d] = c            # a REAL array in every mode: the tracer writes it into the source as a table over axis d
        return out

    def hamiltonian(self, t, data, p, sd=None):
        env = self._coords(p[0])
        for d in range(self.dim):
            env["p%d" % d] = p[d]
        return self.scale * self.H(env) + 0 * p[0]          # (the grid's shape even where the random tree happens to ignore the costates)

    def dissipation(self, t, data, dmin, dmax, sd, dim):
        env = self._coords(data)
        for d in range(self.dim):
            env["lo%d" % d], env["hi%d" % d] = dmin[d], dmax[d]
        return abs(self.A[dim](env)) + 0.2 + 0 * data


def make_grid(rng, dim, n):
    gmin = -np.ones((dim, 1))
    pd = [d for d in range(dim) if rng.random() < 0.4]
    gmax = np.array([[1.0 - (2.0 / n[d] if d in pd else 0.0)] for d in range(dim)])
    return L.createGrid(gmin, gmax, np.array(n, dtype=np.int64).reshape(-1, 1), pd if pd else None)


def cpu_case(seed):
    rng = np.random.default_rng(seed)
    dim = int(rng.integers(2, 5))
    n = [int(rng.integers(5, 9)) for _ in range(dim)]
    g = make_grid(rng, dim, n)
    use_range = rng.random() < 0.5
    s = RandomSystem(g, rng, dim, use_range)
    try:
        tr = TH.trace_callbacks(g, s.hamiltonian, s.dissipation, None)
    except TH.TraceError as e:
        # alpha may not read data: `0 * data` folds away; nothing else here is untraceable
        return "TraceError: %s" % e
    p = [rng.standard_normal(g.shape) for _ in range(dim)]
    lo, hi = [float(-abs(rng.standard_normal()) - 0.1) for _ in range(dim)], [float(abs(rng.standard_normal()) + 0.1) for _ in range(dim)]
    data = np.zeros(g.shape)
    H, al = tr.evaluate(np.meshgrid(*[np.asarray(v).ravel() for v in g.vs], indexing="ij"), p, lo, hi)
    with np.errstate(all="ignore"):
        Hr = s.hamiltonian(0., data, p, None)
        alr = [s.dissipation(0., data, lo, hi, None, d) for d in range(dim)]
    # (x ** 3 is x * x * x on the device and pow() in NumPy: last-bit differences, amplified where the expression cancels)
    scale = float(np.nanmax(np.abs(Hr))) + 1.0
    with np.errstate(all="ignore"):            # how far a relative 1e-13 in the costates moves H: cos(p ** 18) is not a test of the tracer
        cond = float(np.nanmax(np.abs(s.hamiltonian(0., data, [q * (1 + 1e-13) for q in p], None) - Hr)))
    if not np.allclose(np.broadcast_to(H, g.shape), Hr, rtol=1e-10, atol=1e-11 * scale + 100 * cond, equal_nan=True):
        return "H differs by %g" % float(np.nanmax(np.abs(H - Hr)))
    for d in range(dim):
        if not np.allclose(np.broadcast_to(al[d], g.shape), alr[d], rtol=1e-10, atol=1e-10, equal_nan=True):
            return "alpha[%d] differs" % d
    return None


def gpu_case(seed):
    from levelsetpy_amd.context import device_grid
    rng = np.random.default_rng(seed)
    dim = int(rng.choice([2, 3, 3, 4]))
    n = [int(rng.integers(14, 30)) for _ in range(dim)] if dim < 4 else [int(rng.integers(9, 15)) for _ in range(dim)]
    g = make_grid(rng, dim, n)
    use_range = rng.random() < 0.5
    s = RandomSystem(g, rng, dim, use_range)
    scheme = str(rng.choice(["ENO2", "ENO3", "WENO5_ASSHIPPED", "WENO5"]))
    dfn = [L.artificialDissipationGLF, L.artificialDissipationLLF, L.artificialDissipationLLLF][int(rng.integers(3))]
    calc = {"ENO2": L.upwindFirstENO2, "ENO3": L.upwindFirstENO3, "WENO5_ASSHIPPED": L.upwindFirstWENO5, "WENO5": L.upwindFirstWENO5Intended}[scheme]
    dtype = torch.float64 if rng.random() < 0.75 else torch.float32
    xs = np.meshgrid(*[np.asarray(v).ravel() for v in g.vs], indexing="ij")
    y0 = np.sqrt(sum(x * x for x in xs)) - 0.6 + 0.02 * rng.standard_normal(g.shape)
    y = torch.as_tensor(y0.reshape(-1, 1), device="cuda", dtype=dtype)

    def bundle():
        return L.Bundle(dict(grid=g, hamFunc=s.hamiltonian, partialFunc=s.dissipation, dissFunc=dfn, CoStateCalc=calc))
    os.environ["HJ_TRACE"] = "0"
    try:
        split, sb_s, _ = L.termLaxFriedrichs(0., y, bundle())
    finally:
        del os.environ["HJ_TRACE"]
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        fused, sb_f, _ = L.termLaxFriedrichs(0., y, bundle())
    dg = device_grid(g, "float64" if dtype == torch.float64 else "float32")
    kern = dg.lib.hj_last_kernel(dg.ctx).decode()
    tag = "dim %d n %s %s %s %s range %d" % (dim, n, scheme, dfn.__name__[-3:], str(dtype)[-7:], use_range)
    if any("disagrees" in str(m.message) for m in w):
        return "the check against the callbacks FAILED (%s): %s" % (tag, [str(m.message)[:300] for m in w]), kern
    if "hipRTC" not in kern:
        try:
            TH.trace_callbacks(g, s.hamiltonian, s.dissipation, None)
            why = "traceable, but the plan was not taken"
        except TH.TraceError as e:
            why = "TraceError: %s" % e
        return "not fused (%s): %s" % (tag, why), kern
    scale = float(split.abs().max()) + 1e-300
    diff = (fused - split).abs()
    tol = (1e-9 if dtype == torch.float64 else 5e-4) * scale
    frac = float((diff > tol).double().mean())
    if not (frac <= 2e-3 and abs(sb_f - sb_s) <= (1e-7 if dtype == torch.float64 else 1e-3) * abs(sb_s)):
        return "fused vs split (%s): %.3g of the nodes beyond %.1e, max %.3g of %.3g, stepBound %r / %r" % (tag, frac, tol, float(diff.max()), scale, sb_f, sb_s), kern
    return None, kern


t_end = time.time() + budget
k, bad, kernels = 0, 0, {}
while time.time() < t_end:
    seed = seed0 + k
    if mode == "cpu":
        msg = cpu_case(seed)
    else:
        msg, kern = gpu_case(seed)
        kernels[kern] = kernels.get(kern, 0) + 1
    if msg:
        print("case %d (seed %d): %s" % (k, seed, msg), flush=True)
        print("FAILED: replay with  python tests/fuzz_trace.py 1 %d %s" % (seed, mode))
        sys.exit(1)
    k += 1
    if mode == "gpu" and k % 10 == 0:
        print("%d cases ok" % k, flush=True)
print("trace fuzz (%s): %d cases ok in %.0f s %s" % (mode, k, budget, kernels if kernels else ""))

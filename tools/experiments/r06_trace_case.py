"""round 6: one case of tests/fuzz_trace.py (gpu mode) under every dissipation variant and dtype: traced-fused and split against the CPU oracle."""
import os, sys, warnings
import numpy as np, torch
root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
seed = int(sys.argv[1])
argv = sys.argv
sys.argv = ["x", "0", str(seed), "gpu"]
src = open(os.path.join(root, "tests", "fuzz_trace.py")).read().split("t_end = time.time()")[0]
ns = {"__file__": os.path.join(root, "tests", "fuzz_trace.py")}
exec(compile(src, "fuzz", "exec"), ns)
import levelsetpy_amd as L
from oracle import hj_oracle as O
rng = np.random.default_rng(seed)
dim = int(rng.choice([2, 3, 3, 4]))
n = [int(rng.integers(14, 30)) for _ in range(dim)] if dim < 4 else [int(rng.integers(9, 15)) for _ in range(dim)]
g = ns["make_grid"](rng, dim, n)
use_range = rng.random() < 0.5
s = ns["RandomSystem"](g, rng, dim, use_range)
scheme0 = str(rng.choice(["ENO2", "ENO3", "WENO5_ASSHIPPED", "WENO5"]))
xs = np.meshgrid(*[np.asarray(v).ravel() for v in g.vs], indexing="ij")
rng2 = np.random.default_rng(seed + 1)
y0 = np.sqrt(sum(x * x for x in xs)) - 0.6 + 0.02 * rng2.standard_normal(g.shape)
pd = [d for d in range(dim) if g.bdry[d] is L.addGhostPeriodic] if hasattr(g, "bdry") else []
og = O.Grid([-1.0] * dim, [float(np.asarray(g.max).ravel()[d]) for d in range(dim)], n, pd)
so = ns["RandomSystem"].__new__(ns["RandomSystem"]); so.__dict__.update(s.__dict__); so.grid = og
for scheme in (scheme0,):
    calc = {"ENO2": L.upwindFirstENO2, "ENO3": L.upwindFirstENO3, "WENO5_ASSHIPPED": L.upwindFirstWENO5, "WENO5": L.upwindFirstWENO5Intended}[scheme]
    for dname, dfn in (("glf", L.artificialDissipationGLF), ("llf", L.artificialDissipationLLF), ("lllf", L.artificialDissipationLLLF)):
        yo, sbo = O.term_lax_friedrichs(og, so, scheme, 0., y0.reshape(-1, 1), diss=dname)
        for dtype in (torch.float64, torch.float32):
            y = torch.as_tensor(y0.reshape(-1, 1), device="cuda", dtype=dtype)
            def bundle():
                return L.Bundle(dict(grid=g, hamFunc=s.hamiltonian, partialFunc=s.dissipation, dissFunc=dfn, CoStateCalc=calc))
            os.environ["HJ_TRACE"] = "0"
            split, sb_s, _ = L.termLaxFriedrichs(0., y, bundle())
            del os.environ["HJ_TRACE"]
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                fused, sb_f, _ = L.termLaxFriedrichs(0., y, bundle())
            from levelsetpy_amd import term
            plan = term._plan_of(bundle())
            os.environ["HJ_TRACE_FORCE"] = "1"
            raw = None
            if plan is not None:
                raw, sb_r, _dg = term._fused_term(plan, 0., y, 0)
            yo_t = torch.as_tensor(yo, device="cuda")
            def fr(a):
                d = (a.double() - yo_t).abs()
                return "%.3g / max %.3g" % (float((d > 1e-4 * float(yo_t.abs().max())).double().mean()), float(d.max()))
            print("%s %-4s %-7s oracle sb %.6g | split sb %.6g diff %s | fused(raw) sb %s diff %s | warned %d" % (
                scheme, dname, str(dtype)[-7:], sbo, sb_s, fr(split), ("%.6g" % sb_r) if raw is not None else "-", fr(raw) if raw is not None else "-", len(w)), flush=True)

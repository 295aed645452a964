"""one gpu case of tests/fuzz_trace.py, with the reason a traceable pair was not fused"""
import os, sys, traceback
import numpy as np, torch
root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
seed = int(sys.argv[1])
sys.argv = ["x", "0", str(seed), "gpu"]
src = open(os.path.join(root, "tests", "fuzz_trace.py")).read().split("t_end = time.time()")[0]
ns = {"__file__": os.path.join(root, "tests", "fuzz_trace.py")}
exec(compile(src, "fuzz", "exec"), ns)
import levelsetpy_amd as L
from levelsetpy_amd import term, _ffi
rng = np.random.default_rng(seed)
dim = int(rng.choice([2, 3, 3, 4]))
n = [int(rng.integers(14, 30)) for _ in range(dim)] if dim < 4 else [int(rng.integers(9, 15)) for _ in range(dim)]
g = ns["make_grid"](rng, dim, n)
use_range = rng.random() < 0.5
s = ns["RandomSystem"](g, rng, dim, use_range)
scheme = str(rng.choice(["ENO2", "ENO3", "WENO5_ASSHIPPED", "WENO5"]))
dfn = [L.artificialDissipationGLF, L.artificialDissipationLLF, L.artificialDissipationLLLF][int(rng.integers(3))]
calc = {"ENO2": L.upwindFirstENO2, "ENO3": L.upwindFirstENO3, "WENO5_ASSHIPPED": L.upwindFirstWENO5, "WENO5": L.upwindFirstWENO5Intended}[scheme]
dtype = torch.float64 if rng.random() < 0.75 else torch.float32
xs = np.meshgrid(*[np.asarray(v).ravel() for v in g.vs], indexing="ij")
y0 = np.sqrt(sum(x * x for x in xs)) - 0.6 + 0.02 * rng.standard_normal(g.shape)
y = torch.as_tensor(y0.reshape(-1, 1), device="cuda", dtype=dtype)
sd = L.Bundle(dict(grid=g, hamFunc=s.hamiltonian, partialFunc=s.dissipation, dissFunc=dfn, CoStateCalc=calc))
plan = term._plan_of(sd)
print("dim", dim, "n", n, scheme, dfn.__name__, dtype, "plan", plan, "traced", getattr(plan, "traced", None))
tr = ns["TH"].trace_callbacks(g, s.hamiltonian, s.dissipation, sd)
print(tr.source[:3000]); print("params", tr.params, "ncol", tr.ncol, "tables", len(tr.tables))
try:
    print(term._fused_term(plan, 0., y, 0)[1])
except Exception:
    traceback.print_exc()

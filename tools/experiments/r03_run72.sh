#!/bin/bash
# round 3, run 72: per-step cost through the drop-in API on the final build, 51^3 / 101^3 / 201^3: native span loop, Python loop with a hook, singleStep calls
out=gpurun_out/r03bt; mkdir -p $out; rm -rf $out/*
for n in 51 101 201; do echo "== N=$n" | tee -a $out/api.txt; N=$n timeout -k 10 300 python tools/api_overhead.py 2>/dev/null | tee -a $out/api.txt; done
cat > /tmp/single.py <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
import levelsetpy_amd as L
for n in (51, 101, 201):
    g = L.createGrid(np.array([[-.75, -1.25, -np.pi]]).T, np.array([[3.25, 1.25, np.pi * (1 - 2 / n)]]).T, n * np.ones((3, 1), dtype=np.int64), 2, low_mem=True)
    sys_ = L.DubinsVehicleRel(g, 1, 1)
    sd = L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation, dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstWENO5))
    y = torch.as_tensor(np.asarray(L.shapeCylinder(g, 2, np.zeros((3, 1)), .5)), device="cuda").reshape(-1, 1)
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    t = 0.
    for _ in range(20): t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, sd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 300
    for _ in range(K): t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, sd)
    torch.cuda.synchronize(); sec = time.perf_counter() - t0
    print("singleStep odeCFL3 calls, tensor in / tensor out, n=%d: %.1f us per call  (%.3e cell-substeps/s)" % (n, 1e6 * sec / K, n ** 3 * 3 * K / sec), flush=True)
PY
timeout -k 10 300 python /tmp/single.py 2>/dev/null | tee -a $out/api.txt

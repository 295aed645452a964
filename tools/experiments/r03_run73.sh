#!/bin/bash
# round 3, run 73: where do the ~35 us of host time per singleStep odeCFL3 call go?  cProfile, 51^3, 3000 calls
out=gpurun_out/r03bu; mkdir -p $out; rm -rf $out/*
cat > /tmp/prof.py <<'PY'
import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
import levelsetpy_amd as L
n = 51
g = L.createGrid(np.array([[-.75, -1.25, -np.pi]]).T, np.array([[3.25, 1.25, np.pi * (1 - 2 / n)]]).T, n * np.ones((3, 1), dtype=np.int64), 2, low_mem=True)
sys_ = L.DubinsVehicleRel(g, 1, 1)
sd = L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation, dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstWENO5))
y = torch.as_tensor(np.asarray(L.shapeCylinder(g, 2, np.zeros((3, 1)), .5)), device="cuda").reshape(-1, 1)
op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
t = 0.
for _ in range(50): t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 100.], y, op, sd)
def run():
    global t, y
    for _ in range(3000): t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 100.], y, op, sd)
    torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
PY
timeout -k 10 300 python /tmp/prof.py > $out/prof.txt 2>&1; tail -45 $out/prof.txt

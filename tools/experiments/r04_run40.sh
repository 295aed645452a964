#!/bin/bash
# round 4, call 40: derivatives of all dimensions in ONE launch (upwind_all_kernel) against one launch per dimension: tests, split-path step, computeGradients
out=gpurun_out/r04_run40; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -2 $out/pytest.log; [ $rc -eq 0 ] || exit $rc
for v in 1 0; do
  echo "== HJ_UPWIND_ALL=$v" | tee -a $out/ab.txt
  HJ_UPWIND_ALL=$v timeout -k 10 300 python3 tools/split_bench.py 201 2>&1 | grep "ms per" | tee -a $out/ab.txt
  HJ_UPWIND_ALL=$v timeout -k 10 300 python3 - 2>&1 <<'PY' | grep "computeGradients\|upwind_all" | tee -a $out/ab.txt
import time, numpy as np, torch
import levelsetpy_amd as L
from levelsetpy_amd.spatial import upwind_all_dims
n = 201
g = L.createGrid(np.array([[-.75, -1.25, -np.pi]]).T, np.array([[3.25, 1.25, np.pi * (1 - 2 / n)]]).T, n * np.ones((3, 1), dtype=np.int64), 2, low_mem=True)
y = torch.as_tensor(L.shapeCylinder(g, 2, np.zeros((3, 1)), .5), device="cuda")
for name, fn in (("upwind_all_dims (as-shipped WENO5)", lambda: upwind_all_dims(L.upwindFirstWENO5, g, y)), ("upwind_all_dims (intended WENO5)", lambda: upwind_all_dims(L.upwindFirstWENO5Intended, g, y)),
                 ("computeGradients", lambda: L.computeGradients(g, y))):
    for _ in range(3): fn()
    ts = []
    for _ in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("%-40s %.3f ms per call at 201^3" % (name, 1e3 * sorted(ts)[4]))
PY
done

#!/bin/bash
# round 4, call 23: on-disk cache of the hipRTC code objects (run-time Hamiltonians): tests, then the first-use latency with a cold and a warm cache
out=gpurun_out/r04_run23; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_user_ham.py tests/test_cabi.py -x -q -rs > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log; [ $rc -eq 0 ] || exit $rc
export HJ_RTC_CACHE=/tmp/hj_rtc_cache_probe
for pass in cold warm; do
python3 - <<'PY' 2>&1 | tail -1
import time, numpy as np, torch
import levelsetpy_amd as L
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
reg = L.register_native_hamiltonian("latency_probe", 3, "H = par[0] * (p[0] * cos(x[2]) + p[1] * sin(x[2])) + par[1] * fabs(p[2]); alpha[0] = fabs(par[0] * cos(x[2])); alpha[1] = fabs(par[0] * sin(x[2])); alpha[2] = par[1];", nparams=2)
g = L.createGrid(np.array([[-1., -1., -np.pi]]).T, np.array([[1., 1., np.pi * (1 - 2 / 101)]]).T, 101 * np.ones((3, 1), dtype=np.int64), 2)
s = reg(g, [1., 1.])
sd = L.Bundle(dict(grid=g, hamFunc=s.hamiltonian, partialFunc=s.dissipation, dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstWENO5))
y0 = torch.as_tensor(L.shapeCylinder(g, 2, np.zeros((3, 1)), .5), device="cuda").reshape(-1, 1)
t0 = time.perf_counter()
t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [0., 1.], y0, L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on'))), sd)
torch.cuda.synchronize()
print("first odeCFL3 step of a run-time Hamiltonian at 101^3: %.3f s  (compiled, loaded from cache) = %s" % (time.perf_counter() - t0, L.kernel_cache_stats()))
PY
done

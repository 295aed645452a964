import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import levelsetpy_amd as L
from oracle import hj_oracle as O
from test_gpu_parity import dubins, sdata, DERIV, mk
G = dict(np.load("/root/repo/tests/golden/ode.npz"))
for scheme in ("ENO2", "ENO3"):
    g, og = dubins(G["dubn_data"].shape)
    sd = sdata(g, L.DubinsVehicleRel(g, 1, 1), DERIV[scheme])
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    L.set_eno_mode('fast')
    y, t = torch.as_tensor(G["dubn_data"].reshape(-1, 1), device="cuda"), 0.
    for k in range(5):
        t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, sd)
        key = "rk3n_%s_y%d" % (scheme, k + 1)
        if key in G:
            d = np.abs(y.cpu().numpy() - G[key])
            print(scheme, "step", k + 1, "max diff %.3e" % d.max(), "frac>1e-11 %.3e" % np.mean(d > 1e-11), "frac>1e-13 %.3e" % np.mean(d > 1e-13))
    L.set_eno_mode('exact')
    data = G["dubn_data"]
    for d in range(3):
        m = O.eno_selector_margin(og, data, d, scheme)
        print("  dim", d, "margin<1e-12 frac %.3e" % np.mean(m < 1e-12), "where:", np.argwhere(m < 1e-12)[:5].tolist())

import os, sys, traceback
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import levelsetpy_amd as L
import bench
class A: pass
a = A(); a.n = 201; a.scheme = "WENO5_ASSHIPPED"; a.steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200; a.repeats = 15
try:
    print(bench.time_range_ham(L, torch, a))
except Exception:
    traceback.print_exc()

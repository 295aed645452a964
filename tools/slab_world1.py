import os, sys, types
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
from levelsetpy_amd import dist as hjdist
a = types.SimpleNamespace(n=201, scheme="WENO5_ASSHIPPED", dtype="float64", warmup=3, steps=20)
r = hjdist.bench_slab(a, 0, 1)
print("slab world=1:", r, "value=%.3e" % (r["cells"]*3*a.steps/r["wall"]))
a.scheme="WENO5"
r = hjdist.bench_slab(a, 0, 1)
print("slab world=1 weno5:", "value=%.3e" % (r["cells"]*3*a.steps/r["wall"]))
dist.destroy_process_group()

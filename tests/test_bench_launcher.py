"""CPU: `bench.py --gpus N` starts its own N ranks -- the launcher's contract, exercised with a
stub rank program over gloo (no GPU is touched here): rank environment, ONE relayed JSON line, exit codes, loud
refusals when the rank count and --gpus disagree or the devices are not there."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

STUB = textwrap.dedent('''
    import json, os, sys
    import torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["LOCAL_RANK"]) == rank
    mode = sys.argv[1]
    if mode == "die" and rank == 1:
        sys.exit(7)
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    print("banner from rank %d" % rank)           # noise on stdout must not reach the relayed line
    if rank == 0:
        print(json.dumps({"n_gpus": world, "value": float(t), "argv": sys.argv[1:]}))
    dist.destroy_process_group()
''')


class Args(object):
    def __init__(self, gpus):
        self.gpus, self.steps, self.warmup = gpus, 3, 1


@pytest.fixture
def stub(tmp_path):
    p = tmp_path / "stub_rank.py"
    p.write_text(STUB)
    return str(p)


def last_json(text):
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, text
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 3])
def test_launcher_starts_n_ranks_and_relays_one_line(stub, capfd, n):
    rc = bench.launch_ranks(Args(n), ["ok", "--gpus", str(n)], script=stub, visible_devices=n)
    out = last_json(capfd.readouterr().out)
    assert rc == 0
    assert out["n_gpus"] == n and out["value"] == n * (n + 1) / 2      # every rank joined the all-reduce
    assert out["argv"] == ["ok", "--gpus", str(n)]                     # the command line reaches the ranks unchanged


def test_launcher_propagates_a_dead_rank(stub, capfd, monkeypatch):
    monkeypatch.setenv("HJ_BENCH_WATCHDOG_S", "5")
    rc = bench.launch_ranks(Args(2), ["die"], script=stub, visible_devices=2)
    out = last_json(capfd.readouterr().out)
    assert rc != 0
    assert out["value"] is None and "rank 1 exited with code 7" in out["error"] and out["n_gpus"] == 2


def test_launcher_refuses_more_ranks_than_devices(stub, capfd, monkeypatch):
    monkeypatch.delenv("HJ_BENCH_ONE_DEVICE", raising=False)
    rc = bench.launch_ranks(Args(4), ["ok"], script=stub, visible_devices=1)
    out = last_json(capfd.readouterr().out)
    assert rc == 2 and out["value"] is None and "1 GPU(s) visible" in out["error"]


def run_bench(args, env_extra):
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        if k not in env_extra:
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT,
                          capture_output=True, text=True, timeout=300)


def test_bench_gpus_2_without_devices_fails_loudly():
    """`python bench.py --gpus 2` where two GPUs are not visible: non-zero exit and an `error` line, never a 1-GPU figure."""
    r = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1"], {})
    out = last_json(r.stdout)
    assert r.returncode != 0 and out["value"] is None and out["n_gpus"] == 2 and "error" in out


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    r = run_bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    out = last_json(r.stdout)
    assert r.returncode == 2 and "WORLD_SIZE=2" in out["error"]

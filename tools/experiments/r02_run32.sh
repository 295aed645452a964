#!/bin/bash
# halo ring parked in LDS three planes ahead (HJ_PAIR_RING=1): correctness, traffic, throughput
out=gpurun_out/r02af; mkdir -p $out; rm -f $out/*
HJ_PAIR_RING=1 timeout -k 10 300 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "pair or virtual or plain" > $out/t1.log 2>&1; tail -2 $out/t1.log
HJ_PAIR=2 HJ_PAIR_RING=1 timeout -k 10 400 python -m pytest tests -m gpu -q --deselect tests/test_gpu_configs.py::test_pair_kernel_bitwise_equals_scalar_kernel > $out/t2.log 2>&1; tail -2 $out/t2.log
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --steps 30 --repeats 5 $EXTRA >> $out/ab.txt 2> $out/last.err; grep "pair tiling" $out/last.err | sort | uniq -c | sort -rn | head -1 >> $out/ab.txt; }
for n in 201 301 401 513; do
  EXTRA="--n $n" run HJ_PAIR_RING=0
  EXTRA="--n $n" run HJ_PAIR_RING=1
done
python - <<'PY'
import json
for ln in open("gpurun_out/r02af/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:230]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
cd /tmp; export TMPDIR=/tmp
for ring in 0 1; do for ctr in FETCH_SIZE; do
HJ_PAIR_RING=$ring HJ_BENCH_SPINUP=20 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_${ring}_$ctr -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-also --steps 4 --warmup 1 --repeats 1 --n 513 > /dev/null 2> $GRAFT_REPO_ROOT/$out/pmc.err
echo "== ring=$ring $ctr"; python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $GRAFT_REPO_ROOT/$out/pmc_${ring}_$ctr
done; done

#!/bin/bash
# round 6: vertical pairs (HJ_VPAIR, hj_fusedv.h): parity subset, then the headline at 201^3 / 513^3 with per-stage kernel times
mkdir -p gpurun_out
root=$PWD
out=$root/gpurun_out/r06_vpair.log
: > $out
timeout -k 10 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round6.py tests/test_gpu_configs.py -x -q -m gpu -k "c2_201 or paired_chunk or transposed or c4_513 or full_size or deep_halo or per_substep or self_ring" > gpurun_out/r06_vpair_t.log 2>&1 || { tail -30 gpurun_out/r06_vpair_t.log; exit 1; }
tail -2 gpurun_out/r06_vpair_t.log >> $out
export TMPDIR=/tmp
cd /tmp
for n in 201 513; do
  d=/tmp/vp_$n; rm -rf $d
  HJ_BENCH_SPINUP=100 HJ_BENCH_SETTLE_BLOCKS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $root/bench.py --n $n --no-cpu-baseline --no-also --steps 20 --warmup 5 --repeats 9 > $d.json 2> $d.err || { tail -3 $d.err; exit 1; }
  echo "== n=$n" >> $out
  python3 - $d $d.json >> $out <<'PY'
import csv, glob, sys, json
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("   value %.4g  %.4f ms  frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac_from_value"]))
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fused_pair_kernel" in r["Name"]:
            print("   calls %s  avg %.2f us  min %.2f" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
cat $out

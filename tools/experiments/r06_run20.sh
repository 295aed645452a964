#!/bin/bash
# round 6: vertical pairs A/B in ONE run (tune builds with HJ_VPAIR 0 / 1, alternating), per-stage kernel times
mkdir -p gpurun_out
root=$PWD
out=$root/gpurun_out/r06_vpair_ab.log
: > $out
export TMPDIR=/tmp
cd /tmp
for rep in 1 2; do
for n in 201 513; do
for lib in novp vp; do
  d=/tmp/vp_${n}_$lib; rm -rf $d
  HJ_LIB=$root/levelsetpy_amd/csrc/libhj_v$lib.so HJ_BENCH_SPINUP=100 HJ_BENCH_SETTLE_BLOCKS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $root/bench.py --n $n --no-cpu-baseline --no-also --steps 20 --warmup 5 --repeats 9 > $d.json 2> $d.err || { tail -3 $d.err; exit 1; }
  python3 - $d $d.json $rep $n $lib >> $out <<'PY'
import csv, glob, sys, json
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
t = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fused_pair_kernel" in r["Name"]:
            t[r["Calls"]] = float(r["AverageNs"]) / 1e3
ks = sorted(t, key=int)
print("rep %s n=%s %-5s value %.4g  %.4f ms  frac %.3f   stage 1 %.2f us  stages 2,3 %.2f us" % (sys.argv[3], sys.argv[4], sys.argv[5], d["value"], d["ms_per_step"], d["roofline"]["frac_from_value"], t[ks[0]], t[ks[-1]]))
PY
done
done
done
cat $out

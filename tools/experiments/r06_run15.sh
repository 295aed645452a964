#!/bin/bash
# round 6: two independent 256-thread workgroups per CU (2 pairs per thread, 1024-cell tiles) against the one 512-thread workgroup -- PER STAGE
mkdir -p gpurun_out
root=$PWD
out=$root/gpurun_out/r06_two_wg.log
: > $out
export TMPDIR=/tmp
cd /tmp
for n in 201 513; do
for cfg in "" "HJ_PAIR_NT=256 HJ_PAIR_R=2 HJ_PAIR_KH=2 HJ_PAIR_RING=1" "HJ_PAIR_NT=256 HJ_PAIR_R=2 HJ_PAIR_KH=2 HJ_PAIR_RING=0" "HJ_PAIR_NT=256 HJ_PAIR_R=2 HJ_PAIR_KH=3 HJ_PAIR_RING=1"; do
  d=/tmp/tw_${n}; rm -rf $d
  env $cfg HJ_AUTOTUNE=0 HJ_DEBUG=1 HJ_LIB=$root/levelsetpy_amd/csrc/libhj_vp256r2.so HJ_BENCH_SPINUP=100 HJ_BENCH_SETTLE_BLOCKS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $root/bench.py --n $n --no-cpu-baseline --no-also --steps 20 --warmup 5 --repeats 5 > $d.json 2> $d.err || { tail -3 $d.err >> $out; continue; }
  echo "== n=$n $cfg" >> $out
  grep "^\[hj\] pair tiling" $d.err | head -1 >> $out
  python3 - $d >> $out <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fused_pair_kernel" in r["Name"]:
            print("   calls %s (410: stages 2, 3; 205: stage 1)  avg %.2f us  min %.2f" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
done
cat $out

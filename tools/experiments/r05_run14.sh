#!/bin/bash
# round 5, call 14: the intended WENO5 at 201^3 through kernel shapes with more waves per SIMD (tune build with every configuration compiled:
# libhj_vALLC.so).  fp64 instructions issue every 4.45 cycles per SIMD at two waves per SIMD and every 3.24 at four (profiles/r05_valu_rate.txt),
# and this kernel is bound by fp64 issue (335 operations per cell): do fewer cells per thread at a higher occupancy pay?
root=$PWD; export TMPDIR=/tmp
out=$root/gpurun_out/r05_run14; rm -rf $out; mkdir -p $out
export HJ_LIB=$root/levelsetpy_amd/csrc/libhj_vALLC.so
run() {
  echo "== $*" >> $out/summary.txt
  env "$@" HJ_DEBUG=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-also --no-live-traffic --scheme WENO5 --steps 20 --warmup 5 --repeats 9 --n 201 > $out/b.json 2> $out/b.err
  grep "\[hj\]" $out/b.err | head -2 >> $out/summary.txt
  python3 - $out/b.json >> $out/summary.txt <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("   %.4e  frac %.4f  ms/step %.4f  kernel %s" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["roofline"].get("kernel")))
except Exception as e:
    print("   failed %r" % (e,))
PY
}
for rep in 1 2; do
run HJ_X=0
run HJ_PAIR=0
run HJ_PAIR=0 HJ_NT=512 HJ_R=1 HJ_KH=1 HJ_OCC=4 HJ_PD=2
run HJ_PAIR=0 HJ_NT=1024 HJ_R=1 HJ_KH=1 HJ_OCC=4 HJ_PD=2
run HJ_PAIR=0 HJ_NT=256 HJ_R=1 HJ_KH=2 HJ_OCC=6 HJ_PD=2
run HJ_PAIR=0 HJ_NT=512 HJ_R=2 HJ_KH=1 HJ_OCC=3 HJ_PD=2
run HJ_PAIR=0 HJ_NT=256 HJ_R=2 HJ_KH=2 HJ_OCC=3 HJ_PD=2
run HJ_PAIR=0 HJ_NT=256 HJ_R=2 HJ_KH=2 HJ_OCC=4 HJ_PD=2
done
cat $out/summary.txt

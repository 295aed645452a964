#!/bin/bash
# round 4, call 47: what the down-marching support costs the plane loop.  A = previous build (queue in march order, reversed copy per cell), B = this build
# (queue in grid order, direction-dependent shift), C = tune build without any down support (-DHJ_MAYDOWN=0: the loop of rounds 1-3); 201^3 with pairing on
# (A, B) / off (B0, C), 513^3 (pairing off everywhere), three alternations
out=gpurun_out/r04_run47; mkdir -p $out; : > $out/ab.txt
D=$PWD/levelsetpy_amd/csrc
run() { # tag lib n env...
  tag=$1; lib=$2; n=$3; shift 3
  env "$@" HJ_LIB=$D/$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-live-traffic --no-also --repeats 21 --steps 20 --n $n > $out/b.json 2> $out/b.err || tail -3 $out/b.err
  python3 - $out/b.json "$tag" $n >> $out/ab.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-34s n=%s  %.4e  frac %.4f  us/launch %.2f  iqr %.4f" % (sys.argv[2], sys.argv[3], d["value"], d["roofline"]["frac"], d["ms_per_step"] * 1e3 / 3, d["repeats"]["iqr_over_median"]))
PY
}
for rep in 1 2 3; do
  run "A previous, pairing on" libhj_vPREV.so 201 HJ_X=0
  run "B grid-order queue, pairing on" libhj_mi355x.so 201 HJ_X=0
  run "B0 grid-order queue, pairing off" libhj_mi355x.so 201 HJ_PAIR_DIRS=0
  run "C no down support" libhj_vNODOWN.so 201 HJ_X=0
done
for rep in 1 2; do
  run "A previous" libhj_vPREV.so 513 HJ_X=0
  run "B grid-order queue" libhj_mi355x.so 513 HJ_X=0
  run "C no down support" libhj_vNODOWN.so 513 HJ_X=0
done
cat $out/ab.txt

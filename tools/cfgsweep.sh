#!/bin/bash
# usage: tools/cfgsweep.sh "NT,R NT,R ..." [n] [schemes]   (GPU box) -- kernel-config sweep
mkdir -p gpurun_out
out=gpurun_out/cfgsweep.txt
: > $out
n=${2:-201}
for sch in ${3:-WENO5_ASSHIPPED WENO5}; do
for cfg in $1; do
  IFS=, read nt r pd occ kh <<< "$cfg"
  if [ -n "$kh" ]; then export HJ_KH=$kh; else unset HJ_KH; fi
  res=$(HJ_NT=$nt HJ_R=$r HJ_PD=${pd:-2} HJ_OCC=${occ:--1} HJ_DEBUG=1 timeout -k 5 120 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --scheme $sch --extra-schemes "" --n $n 2>gpurun_out/cfg.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3e frac=%.4f ms/substep=%.4f' % (d['value'], d['roofline']['frac'], d['roofline']['kernel_ms']))")
  til=$(grep -m1 "\[hj\] tiling" gpurun_out/cfg.err | sed 's/\[hj\] tiling//')
  echo "n=$n $sch $cfg -> $res |$til" | tee -a $out
done
done

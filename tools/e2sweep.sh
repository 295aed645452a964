#!/bin/bash
# usage (GPU box): tools/e2sweep.sh "34 45 67 101" n  -- force the last-axis tile extent (HJ_FULL_ROWS=E2)
mkdir -p gpurun_out
: > gpurun_out/e2sweep.txt
for e in 0 $1; do
  res=$(HJ_FULL_ROWS=$e HJ_DEBUG=1 timeout -k 5 120 python bench.py --no-cpu-baseline --steps 30 --warmup 3 --extra-schemes "" --n $2 2>gpurun_out/cfg.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3e frac=%.4f ms=%.4f' % (d['value'], d['roofline']['frac'], d['roofline']['kernel_ms']))")
  til=$(grep -m1 "\[hj\] tiling" gpurun_out/cfg.err | sed 's/\[hj\] tiling//')
  echo "n=$2 E2=$e -> $res |$til" | tee -a gpurun_out/e2sweep.txt
done

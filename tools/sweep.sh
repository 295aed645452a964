#!/bin/bash
# usage: tools/sweep.sh  (runs on the GPU box) -- tiling / chunking sweep of the fused kernel
mkdir -p gpurun_out
out=gpurun_out/sweep.txt
: > $out
run() {
  desc="$1"; shift
  res=$(env "$@" timeout -k 5 120 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --scheme ${SCH:-WENO5_ASSHIPPED} --extra-schemes "" --n ${N:-201} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3e %.4f ms/substep=%.4f' % (d['value'], d['roofline']['frac'], d['roofline']['kernel_ms']))")
  echo "$desc $* -> $res" | tee -a $out
}
for tb in 256 512 1024 2048; do
  for mc in 4 8 16 32; do
    run "n=${N:-201}" HJ_TARGET_BLOCKS=$tb HJ_MIN_CHUNK=$mc
  done
done

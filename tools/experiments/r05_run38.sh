#!/bin/bash
# C5: chunks per tile column with round 5's kernel (HJ_MIN_CHUNK caps the count: 129 -> 1 chunk, 60 -> 2 (the chooser's pick), 40 -> 3, 30 -> 4)
mkdir -p gpurun_out
out=gpurun_out/r38_c5_chunks.txt; : > $out
for rep in 1 2; do
for mc in 129 60 40 30; do
  v=$(HJ_MIN_CHUNK=$mc HJ_DEBUG=1 C5_STEPS=12 C5_WARMUP=4 timeout -k 10 200 python tools/bench_configs.py c5 2>&1 | grep -E "^C5|pair4 tiling" | sed 's/C5 double pendulum 129^4 (one GPU)           float32 WENO5_ASSHIPPED //' | tr '\n' ' ')
  echo "rep $rep HJ_MIN_CHUNK=$mc: $v" >> $out
done
done
cut -c1-330 $out

#!/bin/bash
# C5: how much do the 516-byte rows of the 129^4 grid cost the CURRENT kernel (8-byte loads)?  The same kernel on 128^4 (512-B rows, line aligned),
# 132^4 (528-B rows, 16-B aligned) and 130^4 (520-B rows, 8-B aligned) -- cell-substeps/s are comparable across these sizes
mkdir -p gpurun_out
o=gpurun_out/r39_c5_alignment.txt; : > $o
for rep in 1 2; do
for n in 129 128 132 130; do
  echo "== n=$n rep $rep" >> $o
  C5_N=$n C5_STEPS=20 C5_WARMUP=40 timeout -k 10 200 python tools/bench_configs.py c5 2>&1 | grep -v "^$" | tail -2 >> $o
done
done
cat $o

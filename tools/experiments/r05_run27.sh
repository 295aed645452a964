#!/bin/bash
# small grids: the stage-fused kernel (stages 1+2 in one launch: two launches per RK3 step instead of three) against the default
mkdir -p gpurun_out
out=gpurun_out/r27_small_fuse12.txt; : > $out
for n in 41 51 65 81 101 129; do
  for f in 0 1; do
    v=$(HJ_FUSE12=$f timeout -k 10 120 python bench.py --n $n --steps 200 --warmup 20 --repeats 11 --no-also --no-cpu-baseline --no-live-traffic 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e %.4f ms %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel'][:60]))")
    echo "n=$n HJ_FUSE12=$f  $v" >> $out
  done
done
cat $out

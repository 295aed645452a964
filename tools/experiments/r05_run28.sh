#!/bin/bash
# small grids: the direct kernel (HJ_DIRECT_BELOW) against the tiled default
mkdir -p gpurun_out
out=gpurun_out/r28_small_direct.txt; : > $out
for n in 31 41 51 65 81 101; do
  for f in 0 3000000; do
    v=$(HJ_DIRECT_BELOW=$f timeout -k 10 120 python bench.py --n $n --steps 200 --warmup 20 --repeats 11 --no-also --no-cpu-baseline --no-live-traffic 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e %.4f ms %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel'][:60]))")
    echo "n=$n HJ_DIRECT_BELOW=$f  $v" >> $out
  done
done
cat $out

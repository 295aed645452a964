#!/bin/bash
# non-uniform chunk counts on the big grids (HJ_NONUNIFORM=2: whenever the model says so): C5 (1144 tiles x 1 chunk = 4.47 rounds) and 513^3
mkdir -p gpurun_out
out=gpurun_out/r33_nonuniform_big.txt; : > $out
for rep in 1 2; do
  for nu in 0 3; do
    v=$(HJ_NONUNIFORM=$nu timeout -k 10 300 python bench.py --steps 20 --warmup 5 --repeats 9 --no-cpu-baseline --no-live-traffic --also "513,C5" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); a=d['also']
print('201^3 %.4e | ' % d['value'] + ' | '.join('%s %.4e (%.4f, %s)' % (k[:8], v['value'], v['roofline_frac'], v.get('kernel')) for k,v in a.items() if isinstance(v,dict) and 'value' in v))")
    echo "rep $rep HJ_NONUNIFORM=$nu  $v" >> $out
  done
done
cat $out

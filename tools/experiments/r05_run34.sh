#!/bin/bash
# look-before-atomic on the bound / range keys: the CFL-kept 201^3 launch and the range-dependent step, plus the tests that read bounds
mkdir -p gpurun_out
out=gpurun_out/r34_key_max.txt; : > $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py tests/test_gpu_user_ham.py -q -m gpu -x 2>&1 | tail -2 >> $out
for rep in 1 2 3; do
  v=$(timeout -k 10 300 python bench.py --steps 20 --warmup 5 --repeats 15 --no-cpu-baseline --no-live-traffic --also "CFL,RANGE" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); a=d['also']
k=[x for x in a if 'CFL' in x][0]; r=[x for x in a if 'data-dependent' in x][0]
print('headline %.4e  kept %.4e (%.3f of headline)  range %.4f ms/step (%.2fx split)' % (d['value'], a[k]['value'], a[k]['vs_headline'], a[r]['ms_per_step'], a[r]['vs_split_path']))")
  echo "rep $rep  $v" >> $out
done
cat $out

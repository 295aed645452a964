#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "flat4" > gpurun_out/r06_t11.log 2>&1
rc=$?
tail -5 gpurun_out/r06_t11.log
[ $rc -ne 0 ] && exit $rc
out=gpurun_out/r06_c5_flat_d.log
: > $out
for f in 1 0 1; do
  echo "== HJ_FLAT4=$f" >> $out
  HJ_FLAT4=$f timeout -k 10 300 python bench.py --single C5 --no-cpu-baseline --no-also --steps 10 --warmup 3 --repeats 9 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['value']*32/3/8e12)" >> $out || exit 1
done
cat $out

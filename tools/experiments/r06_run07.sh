#!/bin/bash
# round 6: the 4-D full-row kernel -- parity first, then C5 with and without it
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "flat4" > gpurun_out/r06_t7.log 2>&1
rc=$?
tail -15 gpurun_out/r06_t7.log
[ $rc -ne 0 ] && exit $rc
for f in 0 1; do
  echo "== HJ_FLAT4=$f" >> gpurun_out/r06_c5_flat.log
  HJ_FLAT4=$f HJ_DEBUG=1 timeout -k 10 300 python bench.py --single C5 --no-cpu-baseline --no-also --steps 10 --warmup 3 --repeats 9 2>gpurun_out/r06_c5_flat_$f.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'])" >> gpurun_out/r06_c5_flat.log || exit 1
  grep "^\[hj\]" gpurun_out/r06_c5_flat_$f.err | head -3 >> gpurun_out/r06_c5_flat.log
done
cat gpurun_out/r06_c5_flat.log

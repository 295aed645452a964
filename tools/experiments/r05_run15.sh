#!/bin/bash
# round 5, call 15: intended WENO5 with the smoothness values shared along the march and inside the lane's pair + one Newton step:
# the WENO5 tests of the suite, then 201^3 against the round-4 numbers (6.9-7.3e10; this round's box before the change: 7.45e10)
root=$PWD; export TMPDIR=/tmp
out=$root/gpurun_out/r05_run15; rm -rf $out; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "weno5 or WENO5 or eps or convergence or full_size_201" > $out/tests.log 2>&1; tail -3 $out/tests.log
for rep in 1 2 3; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-also --no-live-traffic --scheme WENO5 --steps 20 --warmup 5 --repeats 9 --n 201 > $out/b.json 2> $out/b.err
  python3 -c "
import json
d=json.loads([l for l in open('$out/b.json') if l.startswith('{')][-1])
print('201^3 WENO5 %.4e frac %.4f ms/step %.4f' % (d['value'], d['roofline']['frac'], d['ms_per_step']))"
done

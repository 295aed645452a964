#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/r06_c5_flat_e.log
: > $out
run() { echo "== $*" >> $out; env "$@" timeout -k 10 300 python bench.py --single C5 --no-cpu-baseline --no-also --steps 10 --warmup 3 --repeats 9 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['value']*32/3/8e12)" >> $out || exit 1; }
run HJ_FLAT4_SEL=0
run HJ_FLAT4_SEL=1
run HJ_FLAT4_SEL=0 HJ_TB1=0
run HJ_FLAT4_SEL=0 HJ_TB1=8 HJ_TB2=4
run HJ_FLAT4_SEL=0 HJ_TB1=4 HJ_TB2=8
run HJ_FLAT4_SEL=0 HJ_TB1=2 HJ_TB2=2
run HJ_FLAT4_SEL=0 HJ_MIN_CHUNK=30
run HJ_FLAT4_SEL=0 HJ_MIN_CHUNK=129
cat $out

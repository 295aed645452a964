#!/bin/bash
# round 6: full-row kernel after the predicates went -- parity, C5 timing, instruction counters
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "flat4" > gpurun_out/r06_t9.log 2>&1
rc=$?
tail -5 gpurun_out/r06_t9.log
[ $rc -ne 0 ] && exit $rc
out=gpurun_out/r06_c5_flat_b.log
: > $out
for f in 1 0 1; do
  echo "== HJ_FLAT4=$f $HJX" >> $out
  HJ_FLAT4=$f timeout -k 10 300 python bench.py --single C5 --no-cpu-baseline --no-also --steps 10 --warmup 3 --repeats 9 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['value']*32/3/8e12)" >> $out || exit 1
done
root=$PWD; export TMPDIR=/tmp; cd /tmp
for ctr in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
  d=/tmp/pmcb_$(echo $ctr | tr ' ' '_'); rm -rf $d
  HJ_FLAT4=1 HJ_BENCH_SPINUP=6 HJ_BENCH_SETTLE_BLOCKS=0 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -- python3 $root/bench.py --single C5 --no-cpu-baseline --no-also --steps 2 --warmup 1 --repeats 1 > /dev/null 2> $d.err || { tail -3 $d.err; continue; }
  python3 $root/tools/pmc_summary.py $d >> $root/$out 2>&1
done
cat $root/$out

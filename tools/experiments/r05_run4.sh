#!/bin/bash
# round 5, call 4: pair4 kernel with 16-lane groups inside one tile row (LDS bank conflicts): tests, C5 A/B, LDS counters
root=$PWD; export TMPDIR=/tmp
out=$root/gpurun_out/r05_run4; rm -rf $out; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "pair4 or 4d or c5 or pendulum or four_d" > $out/tests.log 2>&1; tail -3 $out/tests.log
export C5_STEPS=20 C5_WARMUP=40
for rep in 1 2; do
  for v in 1 0; do
    echo "== HJ_PAIR4=$v" >> $out/c5.txt
    HJ_PAIR4=$v python3 tools/bench_configs.py c5 >> $out/c5.txt 2>&1
  done
done
grep -v "amdgpu.ids" $out/c5.txt
export C5_STEPS=3 C5_WARMUP=2
cd /tmp
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/p$i -- python3 $root/tools/bench_configs.py c5 > /dev/null 2> $out/p$i.err
  echo "== $ctr" >> $out/summary.txt
  python3 $root/tools/pmc_summary.py $out/p$i >> $out/summary.txt 2>&1
done
cd $root
cat $out/summary.txt

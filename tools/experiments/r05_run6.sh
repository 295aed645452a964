#!/bin/bash
# round 5, call 6: counters of the pair4 kernel on the 5x6x66 tile (512 threads) and the 3x5x66 tile (256 threads)
root=$PWD; export TMPDIR=/tmp
out=$root/gpurun_out/r05_run6; rm -rf $out; mkdir -p $out
export C5_STEPS=3 C5_WARMUP=2
cd /tmp
for sel in 1 0; do
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  HJ_TILE4_SEL=$sel timeout -k 10 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/s${sel}_p$i -- python3 $root/tools/bench_configs.py c5 > /dev/null 2> $out/s${sel}_p$i.err
  echo "== tile $sel: $ctr" >> $out/summary.txt
  python3 $root/tools/pmc_summary.py $out/s${sel}_p$i >> $out/summary.txt 2>&1
done
done
cd $root
cat $out/summary.txt

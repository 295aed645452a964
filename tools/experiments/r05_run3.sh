#!/bin/bash
# round 5, call 3: counters of the new C5 kernel (fused_pair4_kernel): where do the cycles go once the instruction count is down 40 %?
root=$PWD; export TMPDIR=/tmp
out=$root/gpurun_out/r05_run3; rm -rf $out; mkdir -p $out
export C5_STEPS=3 C5_WARMUP=2
cd /tmp
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_IFETCH SQ_INSTS_BRANCH" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_GATE_EN1_sum" \
           "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TD_TD_BUSY_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/p$i -- python3 $root/tools/bench_configs.py c5 > /dev/null 2> $out/p$i.err
  echo "== $ctr" >> $out/summary.txt
  python3 $root/tools/pmc_summary.py $out/p$i >> $out/summary.txt 2>&1
  tail -2 $out/p$i.err >> $out/summary.txt
done
cd $root
cat $out/summary.txt

#!/bin/bash
# round 5, call 1: issue costs of the vector instructions (valu_rate), C5 baseline + its SQ counters on this round's first box
root=$PWD; export TMPDIR=/tmp
out=$root/gpurun_out/r05_run1; rm -rf $out; mkdir -p $out
tools/ubench/valu_rate > $out/valu_rate.txt 2>&1
python3 tools/bench_configs.py c5 > $out/c5_base.txt 2>&1
cd /tmp
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/p$i -- python3 $root/tools/bench_configs.py c5 > /dev/null 2> $out/p$i.err
  echo "== $ctr" >> $out/summary.txt
  python3 $root/tools/pmc_summary.py $out/p$i >> $out/summary.txt 2>&1
done
cd $root
cat $out/valu_rate.txt $out/c5_base.txt $out/summary.txt

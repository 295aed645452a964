#!/bin/bash
# round 3, run 32: counters of the C5 launch (129^4 fp32): the default one-cell-per-lane kernel (1024,1,3) against the pair
# kernel in 256-thread workgroups (256,2,10), two workgroups per CU (tune build libhj_vC5Q.so)
root=$PWD; out=$root/gpurun_out/r03af; mkdir -p $out; rm -rf $out/*
export HJ_LIB=$root/levelsetpy_amd/csrc/libhj_vC5Q.so TMPDIR=/tmp HJ_AUTOTUNE=0 HJ_BENCH_SPINUP=5
cd /tmp
for v in scalar pair; do
  if [ $v = pair ]; then export HJ_PAIR_NT=256 HJ_PAIR_R=2 HJ_PAIR_KH=10; fi
  for ctr in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
    name=$(echo $ctr | cut -d' ' -f1)
    timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_${v}_$name -- python3 $root/bench.py --no-cpu-baseline --no-live-traffic --n 65 --steps 2 --warmup 1 --repeats 1 --also C5 > /dev/null 2> $out/pmc_${v}_$name.err || { echo "FAILED $v $ctr" >> $out/pmc_summary.txt; tail -3 $out/pmc_${v}_$name.err >> $out/pmc_summary.txt; continue; }
    echo "== $v: $ctr" >> $out/pmc_summary.txt
    python3 $root/tools/pmc_summary.py $out/pmc_${v}_$name 2>&1 | awk '/^[^ ]/{show = ($0 ~ /Pendulum/)} show' >> $out/pmc_summary.txt
  done
done
cat $out/pmc_summary.txt

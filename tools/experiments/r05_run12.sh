#!/bin/bash
# round 5, call 12: texture-address / L1 counters of the headline kernel at 201^3 and 513^3 (is the TA path what binds it, as it binds C5?)
root=$PWD; export TMPDIR=/tmp
out=$root/gpurun_out/r05_run12; rm -rf $out; mkdir -p $out
cd /tmp
for n in 201 513; do
  for ctr in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" "SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
    name=$(echo $ctr | cut -d' ' -f1)
    HJ_BENCH_SPINUP=60 HJ_BENCH_SETTLE_BLOCKS=0 timeout -k 10 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_${n}_$name -- python3 $root/bench.py --no-cpu-baseline --no-also --no-live-traffic --steps 4 --warmup 1 --repeats 1 --n $n > /dev/null 2> $out/pmc_${n}_$name.err
    echo "== n=$n $ctr" >> $out/summary.txt
    python3 $root/tools/pmc_summary.py $out/pmc_${n}_$name >> $out/summary.txt 2>&1
  done
done
cd $root
cat $out/summary.txt

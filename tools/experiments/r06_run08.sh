#!/bin/bash
# round 6: C5 through the full-row kernel -- fabric traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and the address / LDS / issue counters
mkdir -p gpurun_out
root=$PWD
export TMPDIR=/tmp
cd /tmp
out=$root/gpurun_out/r06_c5_pmc.txt
: > $out
for f in 1 0; do
for ctr in FETCH_SIZE WRITE_SIZE "TA_BUSY_avr GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
  d=/tmp/pmc_${f}_$(echo $ctr | tr ' ' '_')
  rm -rf $d
  HJ_FLAT4=$f HJ_BENCH_SPINUP=6 HJ_BENCH_SETTLE_BLOCKS=0 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -- python3 $root/bench.py --single C5 --no-cpu-baseline --no-also --steps 2 --warmup 1 --repeats 1 > /dev/null 2> $d.err || { tail -3 $d.err; continue; }
  echo "== HJ_FLAT4=$f  $ctr" >> $out
  python3 $root/tools/pmc_summary.py $d >> $out 2>&1
done
done
cat $out

#!/bin/bash
# usage: tools/r02_pmc.sh <tag> <bench args...>   env passes through.  Three PMC passes + kernel-trace stats.
tag=$1; shift
root=$PWD; export TMPDIR=/tmp
out=$root/gpurun_out/pmc_$tag; rm -rf $out; mkdir -p $out
cd /tmp
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/p$i -- python3 $root/bench.py --no-cpu-baseline --no-also --steps 4 --warmup 1 --repeats 1 "$@" > /dev/null 2> $out/p$i.err
  python3 $root/tools/pmc_summary.py $out/p$i >> $out/summary.txt 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --no-cpu-baseline --no-also --steps 20 --warmup 2 --repeats 2 "$@" > /dev/null 2> $out/stats.err
cd $root
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
cut -c1-160 $out/kernel_stats.csv | head -6
cat $out/summary.txt

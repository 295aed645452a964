#!/bin/bash
# usage: tools/profile_round.sh <round-tag>      (on the GPU box; writes under gpurun_out/<tag>/)
# 1. rocprofv3 --kernel-trace --stats of the default bench command
# 2. separate --pmc passes: FETCH_SIZE, WRITE_SIZE (HBM traffic), SQ issue/wait counters
tag=${1:-r01}
root=$PWD
export TMPDIR=/tmp
out=$root/gpurun_out/$tag
rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --no-cpu-baseline > $out/stats_bench.json 2> $out/stats.err
for n in 201 401; do
  for ctr in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
    name=$(echo $ctr | cut -d' ' -f1)
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_${n}_$name -- python3 $root/bench.py --no-cpu-baseline --steps 4 --warmup 1 --extra-schemes "" --n $n > /dev/null 2> $out/pmc_${n}_$name.err
    echo "== n=$n $ctr" >> $out/pmc_summary.txt
    python3 $root/tools/pmc_summary.py $out/pmc_${n}_$name >> $out/pmc_summary.txt 2>&1
  done
done
cd $root
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
cat $out/kernel_stats.csv | cut -c1-220 | head -8
cat $out/pmc_summary.txt

#!/bin/bash
# round 4, call 31: one kernel-trace pass over every kernel family of the library (tools/all_kernels.py) -> per-kernel roofline table (tools/kernel_table.py)
out=gpurun_out/r04_run31; mkdir -p $out
export TMPDIR=/tmp; root=$PWD; cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/trace -- python3 $root/tools/all_kernels.py 10 > $root/$out/all_kernels.out 2> $root/$out/all_kernels.err; echo "rocprofv3 rc=$?"
cd $root; tail -2 $out/all_kernels.out
f=$(find $out/trace -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats_all.csv
python3 tools/kernel_table.py $out/kernel_stats_all.csv > $out/kernel_table.txt; cat $out/kernel_table.txt | cut -c1-230
rm -rf $out/trace

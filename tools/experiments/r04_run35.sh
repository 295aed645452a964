#!/bin/bash
# round 4, call 35: per-kernel table with the terms on the tiled kernel
out=gpurun_out/r04_run35; mkdir -p $out
export TMPDIR=/tmp; root=$PWD; cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/trace -- python3 $root/tools/all_kernels.py 10 > $root/$out/all_kernels.out 2> $root/$out/all_kernels.err; echo "rocprofv3 rc=$?"
cd $root
f=$(find $out/trace -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats_all.csv
python3 tools/kernel_table.py $out/kernel_stats_all.csv > $out/kernel_table.txt; grep -i "term\|upwind_kernel" $out/kernel_table.txt | cut -c1-215
rm -rf $out/trace

#!/bin/bash
# round 4, call 33: upwind_kernel (split path / hj_upwind) with unconditional loads and 32-bit index arithmetic: whole GPU suite, the split path's
# step time, the per-kernel table
out=gpurun_out/r04_run33; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -2 $out/pytest.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/split_bench.py 201 2>&1 | grep -v amdgpu.ids | tee $out/split_bench.txt
export TMPDIR=/tmp; root=$PWD; cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/trace -- python3 $root/tools/all_kernels.py 10 > $root/$out/all_kernels.out 2> $root/$out/all_kernels.err; echo "rocprofv3 rc=$?"
cd $root
f=$(find $out/trace -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats_all.csv
python3 tools/kernel_table.py $out/kernel_stats_all.csv > $out/kernel_table.txt; cat $out/kernel_table.txt | cut -c1-215
rm -rf $out/trace

#!/bin/bash
# round 4, call 32: term_kernel with every stencil load issued unconditionally (gather_stencils): the term tests (bitwise against the array
# path, oracle), tools/term_timing.py, then the per-kernel table again
out=gpurun_out/r04_run32; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "term or normal or reinit or convection or direct or small" > $out/pytest.log 2>&1; rc=$?; tail -2 $out/pytest.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/term_timing.py 201 2>&1 | grep -v amdgpu.ids | tee $out/term_timing.txt
export TMPDIR=/tmp; root=$PWD; cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/trace -- python3 $root/tools/all_kernels.py 10 > $root/$out/all_kernels.out 2> $root/$out/all_kernels.err; echo "rocprofv3 rc=$?"
cd $root
f=$(find $out/trace -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats_all.csv
python3 tools/kernel_table.py $out/kernel_stats_all.csv > $out/kernel_table.txt; grep "term_kernel\|51^3" $out/kernel_table.txt | cut -c1-200
rm -rf $out/trace

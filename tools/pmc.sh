#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" [bench args...]   (on the GPU box)
# rocprofv3 PMC pass over a short bench run; summary (mean per dispatch of the fused kernel) to gpurun_out/<tag>.pmc.txt
tag=$1; ctrs=$2; shift 2
root=$PWD
export TMPDIR=/tmp
d=$root/gpurun_out/prof_$tag
rm -rf $d; mkdir -p $d
cd /tmp
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -- python3 $root/bench.py --no-cpu-baseline --steps 4 --warmup 1 --extra-schemes "" "$@" > $d/stdout.txt 2> $d/stderr.txt
cd $root
python3 tools/pmc_summary.py $d > gpurun_out/$tag.pmc.txt 2>&1
cat gpurun_out/$tag.pmc.txt

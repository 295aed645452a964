#!/bin/bash
# C3 (4096^2 ENO3): fabric traffic and VALU utilisation
out=$GRAFT_REPO_ROOT/gpurun_out/r02al; mkdir -p $out; rm -rf $out/*
cd /tmp; export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
name=$(echo $ctr | cut -d' ' -f1)
HJ_BENCH_SPINUP=5 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_$name -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py c3 > $out/c3_$name.txt 2> $out/pmc_$name.err
echo "== $ctr"; python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/pmc_$name
done
cat $out/c3_FETCH_SIZE.txt

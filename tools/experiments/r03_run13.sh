#!/bin/bash
# round 3, run 13: the driver's bench command + the profile round (kernel-trace stats, PMC passes at 201^3 / 513^3)
out=gpurun_out/r03m; mkdir -p $out; rm -f $out/*
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench.err; echo "bench rc=$?"
bash tools/profile_round.sh r03prof > $out/profile_round.txt 2>&1; tail -50 $out/profile_round.txt
# VALU operations per cell of the intended WENO5 substep kernel (roofline_valu): SQ_INSTS_VALU x 64 / cells
cd /tmp; export TMPDIR=/tmp
HJ_BENCH_SPINUP=20 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OLDPWD/$out/pmc_weno5 -- python3 $OLDPWD/bench.py --no-cpu-baseline --no-also --no-live-traffic --steps 4 --warmup 1 --repeats 1 --n 201 --scheme WENO5 > /dev/null 2> $OLDPWD/$out/pmc_weno5.err
cd $OLDPWD; python3 tools/pmc_summary.py $out/pmc_weno5 > $out/weno5_valu.txt 2>&1; cat $out/weno5_valu.txt

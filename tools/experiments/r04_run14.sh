#!/bin/bash
# round 4, call 14: the committed evidence of the final build -- default bench line, then tools/profile_round.sh (kernel-trace stats of
# the driver's command and of the 201^3 workload alone, PMC passes at 201^3 / 513^3)
out=gpurun_out/r04_run14; mkdir -p gpurun_out/r04_run14
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; echo "bench default rc=$?"
bash tools/profile_round.sh r04_profile > $out/profile_round.out 2>&1; echo "profile_round rc=$?"
tail -5 $out/profile_round.out | cut -c1-200

#!/bin/bash
# late round 5: the fuzz scripts once more on the final library (the slab script from the seed whose first range-alpha case met the compile race)
mkdir -p gpurun_out
o=gpurun_out/r38_fuzz.txt; : > $o
timeout -k 10 300 python tests/fuzz_slabs.py 150 32000 > gpurun_out/r38_slabs.log 2>&1; echo "fuzz_slabs rc=$? $(tail -1 gpurun_out/r38_slabs.log) range-alpha cases: $(grep -c range-alpha gpurun_out/r38_slabs.log)" >> $o
timeout -k 10 200 python tests/fuzz_parity.py 90 91000 0.3 > gpurun_out/r38_parity.log 2>&1; echo "fuzz_parity rc=$? $(tail -1 gpurun_out/r38_parity.log)" >> $o
timeout -k 10 200 python tests/fuzz_terms.py 60 93000 > gpurun_out/r38_terms.log 2>&1; echo "fuzz_terms rc=$? $(tail -1 gpurun_out/r38_terms.log)" >> $o
timeout -k 10 200 python tests/fuzz_solver.py 60 94000 > gpurun_out/r38_solver.log 2>&1; echo "fuzz_solver rc=$? $(tail -1 gpurun_out/r38_solver.log)" >> $o
cat $o

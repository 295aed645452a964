#!/bin/bash
# late round 5: a Hamiltonian whose alpha_i reads the range of ANOTHER dimension (the case in which GLF / LLF / LLLF really differ): tests, then the two
# fuzz scripts that now draw it in half of their range-alpha cases
mkdir -p gpurun_out
o=gpurun_out/r43_coupled.txt; : > $o
timeout -k 10 600 python -m pytest tests/test_gpu_round5.py -x -q -m gpu -k "cross_dimension" > gpurun_out/r43_tests.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/r43_tests.log)" >> $o
timeout -k 10 300 python tests/fuzz_parity.py 150 97000 1.0 > gpurun_out/r43_parity.log 2>&1; echo "fuzz_parity (range cases only) rc=$? $(tail -1 gpurun_out/r43_parity.log)" >> $o
timeout -k 10 300 python tests/fuzz_slabs.py 120 98000 > gpurun_out/r43_slabs.log 2>&1; echo "fuzz_slabs rc=$? $(tail -1 gpurun_out/r43_slabs.log); range-alpha cases $(grep -c range-alpha gpurun_out/r43_slabs.log), coupled $(grep -c 'lf+ ' gpurun_out/r43_slabs.log)" >> $o
cat $o

#!/bin/bash
# round 4, call 34: termNormal / termReinit / termConvection through the tiled kernel (TermOp): bitwise against the direct term_kernel, the
# existing term tests (array path, oracle), then timing at 201^3
out=gpurun_out/r04_run34; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "tiled_kernel_equal" > $out/pytest_new.log 2>&1; rc=$?; tail -12 $out/pytest_new.log | cut -c1-250; [ $rc -eq 0 ] || exit $rc
HJ_TERM_TILED_FROM=0 timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "term or normal or reinit or convection" > $out/pytest_terms_tiled.log 2>&1; rc=$?; tail -3 $out/pytest_terms_tiled.log | cut -c1-250; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/term_timing.py 201 2>&1 | grep -v amdgpu.ids | tee $out/term_timing_tiled.txt
HJ_TERM_TILED_FROM=-1 timeout -k 10 300 python3 tools/term_timing.py 201 2>&1 | grep -v amdgpu.ids | tee $out/term_timing_direct.txt

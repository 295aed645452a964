#!/bin/bash
# round 4, call 46: whole GPU suite under further global knobs: one-cell-per-lane kernel everywhere with 1-plane chunks; eager NumPy results; terms tiled at every size
out=gpurun_out/r04_run46; mkdir -p $out
HJ_PAIR=0 HJ_MIN_CHUNK=1 timeout -k 10 1000 python3 -m pytest tests -m gpu -q > $out/pytest_scalar.log 2>&1; echo "HJ_PAIR=0 HJ_MIN_CHUNK=1 rc=$?"; tail -6 $out/pytest_scalar.log | cut -c1-220
HJ_LAZY_NUMPY=0 timeout -k 10 1000 python3 -m pytest tests -m gpu -q > $out/pytest_eager.log 2>&1; echo "HJ_LAZY_NUMPY=0 rc=$?"; tail -6 $out/pytest_eager.log | cut -c1-220
HJ_TERM_TILED_FROM=0 timeout -k 10 1000 python3 -m pytest tests -m gpu -q > $out/pytest_terms.log 2>&1; echo "HJ_TERM_TILED_FROM=0 rc=$?"; tail -6 $out/pytest_terms.log | cut -c1-220

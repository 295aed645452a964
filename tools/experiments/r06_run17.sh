#!/bin/bash
# round 6: two planes per barrier (tune build -DHJ_TWO_PLANES=1) against the product build: parity first, then 201^3 / 513^3 per stage
mkdir -p gpurun_out
root=$PWD
out=$root/gpurun_out/r06_twob.log
: > $out
HJ_LIB=$root/levelsetpy_amd/csrc/libhj_vtwob.so HJ_PAIR_RING=0 timeout -k 10 300 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "c2_201 or (paired_chunk and WENO5_ASSHIPPED and (n0 or n1))" > gpurun_out/r06_twob_t.log 2>&1 || { tail -20 gpurun_out/r06_twob_t.log; exit 1; }
tail -2 gpurun_out/r06_twob_t.log >> $out
for n in 201 513; do
for cfg in "HJ_LIB= " "HJ_LIB=$root/levelsetpy_amd/csrc/libhj_vtwob.so HJ_PAIR_RING=0" "HJ_LIB=$root/levelsetpy_amd/csrc/libhj_vtwob.so HJ_PAIR_AH=1" "HJ_LIB= HJ_PAIR_RING=0"; do
  echo "== n=$n $cfg" >> $out
  env $cfg HJ_AUTOTUNE=0 HJ_DEBUG=1 timeout -k 10 200 python bench.py --n $n --no-cpu-baseline --no-also --steps 20 --warmup 5 --repeats 15 2>gpurun_out/r06_twob.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.4g  %.4f ms  frac %.3f  parity-free' % (d['value'], d['ms_per_step'], d['roofline']['frac_from_value']))" >> $out || { tail -5 gpurun_out/r06_twob.err; exit 1; }
  grep "^\[hj\] pair tiling" gpurun_out/r06_twob.err | head -1 >> $out
done
done
cat $out

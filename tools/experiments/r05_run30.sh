#!/bin/bash
# intended WENO5: the middle axis' smoothness values shared between lanes through LDS (default build) against a build without (libhj_vW0.so: tune build)
mkdir -p gpurun_out
out=gpurun_out/r30_weno5_lds.txt; : > $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x -k "WENO5 or weno" 2>&1 | tail -2 >> $out
for rep in 1 2; do
  for lib in default W0; do
    if [ $lib = default ]; then unset HJ_LIB; else export HJ_LIB=$PWD/levelsetpy_amd/csrc/libhj_v$lib.so; fi
    for n in 201 301; do
      v=$(timeout -k 10 200 python bench.py --n $n --scheme WENO5 --steps 20 --warmup 5 --repeats 15 --no-also --no-cpu-baseline --no-live-traffic 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e %.4f ms' % (d['value'], d['ms_per_step']))")
      echo "rep $rep lib=$lib n=$n  $v" >> $out
    done
  done
done
cat $out

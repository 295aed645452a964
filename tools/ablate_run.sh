#!/bin/bash
# usage (GPU box): tools/ablate_run.sh "1 2 3 4 6 7" "201 401"  -- time the HJ_ABLATE variants under tools/ablate/
# build the variants first:  for ab in 1 2 3 4 6 7; do hipcc -DHJ_TUNE_BUILD -DHJ_ABLATE=$ab -ffp-contract=on -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -o tools/ablate/libhj_ab$ab.so levelsetpy_amd/csrc/hj_api.hip -ldl; done
# (HJ_ABLATE bits: 1 no stencil arithmetic, 2 no LDS stencil reads, 4 no stores)
mkdir -p gpurun_out
: > gpurun_out/ablate.txt
for n in $2; do
  for ab in base $1; do
    if [ "$ab" = base ]; then unset HJ_LIB; else export HJ_LIB=$PWD/tools/ablate/libhj_ab$ab.so; fi
    res=$(timeout -k 5 120 python bench.py --no-cpu-baseline --steps 30 --warmup 3 --extra-schemes "" --n $n 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.4f' % d['roofline']['kernel_ms'])")
    echo "n=$n ablate=$ab kernel_ms=$res" | tee -a gpurun_out/ablate.txt
  done
done

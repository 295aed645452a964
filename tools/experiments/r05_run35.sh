#!/bin/bash
# same-box A/B: bound keys updated blindly (round 4: libhj_vKB.so) or after a look (libhj_vKL.so), 201^3 with the CFL reduction kept in every launch
mkdir -p gpurun_out
out=gpurun_out/r35_key_max_ab.txt; : > $out
for rep in 1 2 3; do
  for lib in KB KL; do
    v=$(HJ_LIB=$PWD/levelsetpy_amd/csrc/libhj_v$lib.so HJ_KEEP_BOUNDS=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --repeats 15 --no-also --no-cpu-baseline --no-live-traffic 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e %.4f ms' % (d['value'], d['ms_per_step']))")
    w=$(HJ_LIB=$PWD/levelsetpy_amd/csrc/libhj_v$lib.so timeout -k 10 200 python bench.py --steps 20 --warmup 5 --repeats 15 --no-also --no-cpu-baseline --no-live-traffic 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e' % d['value'])")
    echo "rep $rep lib=$lib  kept: $v   skipped: $w" >> $out
  done
done
cat $out

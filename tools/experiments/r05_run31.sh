#!/bin/bash
# headline kernel: non-temporal cache policy on the y0 loads / the output stores (tune builds libhj_vA*.so), 201^3 and 513^3, alternating
mkdir -p gpurun_out
out=gpurun_out/r31_headline_nt.txt; : > $out
for rep in 1 2; do
  for lib in A0 AY AO AYO; do
    export HJ_LIB=$PWD/levelsetpy_amd/csrc/libhj_v$lib.so
    for n in 201 513; do
      st=20; [ $n = 513 ] && st=10
      v=$(timeout -k 10 200 python bench.py --n $n --steps $st --warmup 5 --repeats 15 --no-also --no-cpu-baseline --no-live-traffic 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e %.4f ms frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))")
      echo "rep $rep lib=$lib n=$n  $v" >> $out
    done
  done
done
cat $out

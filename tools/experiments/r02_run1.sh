#!/bin/bash
# round-2 GPU call 1: full GPU suite (incl. the new C4/C5 tests), default bench, warm-up-cost A/B
out=gpurun_out/r02a; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/test.log 2>&1; echo "pytest rc=$?" | tee -a $out/test.log
tail -5 $out/test.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err; echo "bench rc=$?"
cat $out/bench_driver.json
for wc in 4 6; do
  for n in 201 401 513; do
    echo "== HJ_WARMUP_COST=$wc n=$n" >> $out/warmup_ab.txt
    HJ_WARMUP_COST=$wc HJ_DEBUG=1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-also --n $n --steps 40 --repeats 3 >> $out/warmup_ab.txt 2>> $out/warmup_ab.err
  done
done
grep -E "==|value" $out/warmup_ab.txt | sed -e 's/"unit".*"repeats"/ "repeats"/' | cut -c1-260
grep "tiling" $out/warmup_ab.err | sort | uniq -c

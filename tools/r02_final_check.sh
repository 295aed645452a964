#!/bin/bash
# what the driver runs at round end (smoke, GPU suite, default bench) + the slab rehearsal + pair-kernel timing profile
out=gpurun_out/r02fin; mkdir -p $out; rm -f $out/*
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
timeout -k 10 500 python -m pytest tests -m gpu -q > $out/gpu_tests.log 2>&1; tail -2 $out/gpu_tests.log
python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02fin/bench_default.json'))
print("default: %.4e frac %.3f kernel=%s traffic=%s cpu=%.3e" % (d['value'], d['roofline']['frac'], d['roofline']['kernel'], d['roofline']['traffic'], d['cpu_baseline']['value']))
for k,v in d['also'].items(): print("   also", k, "%.4e" % v['value'], "%.3f" % v['roofline_frac'])
PY
HJ_BENCH_FORCE_SLAB=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_slab.json 2> $out/bench_slab.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02fin/bench_slab.json'))
print("slab rehearsal: %.4e frac %.3f scaling=%s check=%s" % (d['value'], d['roofline']['frac'], d['scaling'], d['also']))
PY
for n in 201 401; do
HJ_TIMING_DUMP=$out/t$n.txt timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --n $n --steps 3 --warmup 3 --repeats 1 > /dev/null 2> $out/b$n.err
python tools/pair_timing.py $out/t$n.txt > $out/s$n.txt
done
(echo "# tools/pair_timing.py on HJ_TIMING_DUMP files of bench.py --n 201 / --n 401 --steps 3 --warmup 3 (pair kernel, final round-2 build; every launch synchronised for the dump; last three launches = one RK3 step)"; echo "# ---- 201^3"; cut -c1-420 $out/s201.txt; echo "# ---- 401^3"; cut -c1-420 $out/s401.txt) > $out/pair_timing.txt
rm -f $out/t201.txt $out/t401.txt

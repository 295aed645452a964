#!/bin/bash
# round 4, call 42: SQ_INSTS_VALU of the intended WENO5's substep kernel at 201^3 after the smoothness-measure change (VALU operations per cell-substep)
out=gpurun_out/r04_run42; mkdir -p $out
export TMPDIR=/tmp; root=$PWD; cd /tmp
HJ_BENCH_SETTLE_BLOCKS=0 HJ_BENCH_SPINUP=10 timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $root/$out/pmc -- python3 $root/bench.py --no-cpu-baseline --no-also --no-live-traffic --scheme WENO5 --steps 4 --warmup 1 --repeats 1 > $root/$out/pmc.out 2> $root/$out/pmc.err; echo "rc=$?"
cd $root
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r04_run42/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "SQ_INSTS_VALU": tot[r["Kernel_Name"][:90]].append(float(r["Counter_Value"]))
for k, v in tot.items():
    print("%-92s n=%4d mean=%.4e  -> x64 / 8120601 cells = %.1f lane-operations per cell" % (k, len(v), sum(v) / len(v), 64 * sum(v) / len(v) / 8120601))
PY
rm -rf $out/pmc

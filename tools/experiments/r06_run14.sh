#!/bin/bash
# round 6: PER-STAGE bound model of the headline kernel -- ablation builds of the plane loop (HJ_ABLATE bits: 1 no Hamiltonian
# arithmetic, 2 no LDS stencil reads, 8 no halo loads, 16 no halo LDS stores; 27 = the streaming skeleton), kernel durations per instantiation
# (MODE 1 = stage 1: 1R + 1W; MODE 2 = stages 2, 3: 2R + 1W) from rocprofv3 --kernel-trace --stats, next to this box's copy / triad rates
mkdir -p gpurun_out
root=$PWD
out=$root/gpurun_out/r06_stage1_bound.log
: > $out
export TMPDIR=/tmp
cd /tmp
echo "== streaming rates of this box (tools/ubench/bw2 --quick: TB/s)" >> $out
$root/tools/ubench/bw2 --quick >> $out 2>&1
for n in 201 513; do
for ab in 0 27 3 24 2; do
  d=/tmp/st_${n}_$ab; rm -rf $d
  HJ_LIB=$root/levelsetpy_amd/csrc/libhj_vs1ab$ab.so HJ_BENCH_SPINUP=100 HJ_BENCH_SETTLE_BLOCKS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $root/bench.py --n $n --no-cpu-baseline --no-also --steps 20 --warmup 5 --repeats 5 > $d.json 2> $d.err || { tail -3 $d.err; continue; }
  echo "== n=$n HJ_ABLATE=$ab" >> $out
  python3 - $d >> $out <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fused_pair_kernel" in r["Name"]:
            mode = r["Name"].rstrip(">)").split(",")[-1].split(">")[0].strip()
            print("   MODE %s  calls %s  avg %.2f us  min %.2f  max %.2f" % (mode, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
done
cat $out

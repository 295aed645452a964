#!/bin/bash
# usage: tools/profile_round.sh <round-tag>      (on the GPU box; writes under gpurun_out/<tag>/)
# 1. rocprofv3 --kernel-trace --stats of the default bench command (the driver's: --steps 20 --warmup 5)
# 2. separate --pmc passes at 201^3 and 513^3: FETCH_SIZE, WRITE_SIZE (HBM traffic), SQ issue/wait and LDS counters
# 1c/1d. a 513^3-only kernel-trace stats file; the static instruction mix of the built library (tools/isa_mix.py)
# 3. profiles/traffic.json rows (with the hash of the kernel sources they were measured on) printed at the end
tag=${1:-r05}
root=$PWD
export TMPDIR=/tmp
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic > $out/stats_bench.json 2> $out/stats.err
# 1b. the headline workload alone (201^3, the default K/W): this stats file's average for the dominant kernel is the
#     number bench.py's roofline.kernel_ms has to agree with (the default command above mixes 201^3 and 513^3 launches
#     of the same instantiation in one row)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats201 -- python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also > $out/stats201_bench.json 2> $out/stats201.err
find $out/stats201 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_201.csv
# 1c. the 513^3 workload alone (the HBM-resident point of the headline kernel): its own kernel-trace stats
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats513 -- python3 $root/bench.py --n 513 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-live-traffic > $out/stats513_bench.json 2> $out/stats513.err
find $out/stats513 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_513.csv
# 1d. static instruction mix of the library these profiles were taken with (no GPU involved; cannot go stale)
python3 $root/tools/isa_mix.py $tag > /dev/null && cp $root/profiles/${tag}_isa_mix.txt $out/isa_mix.txt
for n in 201 513; do
  for ctr in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
    name=$(echo $ctr | cut -d' ' -f1)
    HJ_BENCH_SPINUP=60 HJ_BENCH_SETTLE_BLOCKS=0 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_${n}_$name -- python3 $root/bench.py --no-cpu-baseline --no-also --steps 4 --warmup 1 --repeats 1 --n $n > /dev/null 2> $out/pmc_${n}_$name.err
    echo "== n=$n $ctr" >> $out/pmc_summary.txt
    python3 $root/tools/pmc_summary.py $out/pmc_${n}_$name >> $out/pmc_summary.txt 2>&1
  done
done
cd $root
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
python3 - "$out" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1] + "/kernel_stats.csv")))[:12]:
    print("%-110s calls %5s avg_ns %10.0f  %5.1f %%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]), float(r["Percentage"])))
PY
cat $out/pmc_summary.txt
python3 tools/make_traffic.py $out

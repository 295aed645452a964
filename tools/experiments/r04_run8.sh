#!/bin/bash
# round 4, call 8: C5 (129^4 fp32) with the tiles of a chunk ordered in blocks of TB1 x TB2 tiles of axes 1 and 2 (HJ_TB1 / HJ_TB2; 0 = the
# plain order of rounds 1-3), timing + FETCH_SIZE / WRITE_SIZE of the substep kernel
out=gpurun_out/r04_run8; mkdir -p $out; rm -rf $out/*
timeout -k 10 200 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "c5 or 4d or pendulum" > $out/pytest_4d.log 2>&1; echo "pytest 4-D rc=$?"; tail -2 $out/pytest_4d.log
run() { echo "== HJ_TB1=$1 HJ_TB2=$2" >> $out/ab.txt; HJ_DEBUG=1 HJ_TB1=$1 HJ_TB2=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --no-live-traffic --also C5 --steps 8 --warmup 2 --repeats 3 >> $out/ab.txt 2> $out/last.err || tail -3 $out/last.err >> $out/ab.txt; grep -h "tiling" $out/last.err | tail -1 >> $out/ab.txt; }
for tb in "0 0" "4 4" "8 8" "4 8" "8 4" "2 8" "8 2" "3 6" "6 3" "2 2" "16 1" "1 16"; do run $tb; done
python - <<'PY'
import json
n = None
for ln in open("gpurun_out/r04_run8/ab.txt"):
    if ln.startswith("=="): n = ln.strip(); continue
    if ln.startswith("{"):
        d = json.loads(ln)["also"]["C5"]; print("%-24s %.4e  frac %.4f  ms/launch %.4f" % (n, d["value"], d["roofline_frac"], d["ms_per_step"] / 3))
    elif "tiling" in ln: print("     ", ln.strip()[:170])
PY
export TMPDIR=/tmp; root=$PWD; cd /tmp
for tb in "0 0" "4 4"; do set -- $tb
for ctr in FETCH_SIZE WRITE_SIZE; do
  HJ_TB1=$1 HJ_TB2=$2 HJ_BENCH_SPINUP=3 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $root/$out/pmc_$1_$2_$ctr -- python3 $root/bench.py --no-cpu-baseline --no-live-traffic --also C5 --steps 2 --warmup 1 --repeats 1 > /dev/null 2> $root/$out/pmc.err
done; done
cd $root
python - <<'PY'
import csv, glob
for tb in ("0_0", "4_4"):
    tot = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob("gpurun_out/r04_run8/pmc_%s_%s/**/*counter_collection.csv" % (tb, ctr), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == ctr and "fused_pair_kernel" in r["Kernel_Name"] and "float" in r["Kernel_Name"] and "Pendulum" in r["Kernel_Name"]:
                    vals.append(float(r["Counter_Value"]))
        vals = vals[-6:]
        tot[ctr] = sum(vals) / max(1, len(vals))
    rd, wr = 2 * tot["FETCH_SIZE"] * 1024, tot["WRITE_SIZE"] * 1024
    alg = 129 ** 4 * 4
    print("TB %s: per launch read %.3f GB (2 x FETCH_SIZE), write %.3f GB; algorithmic read %.3f GB (5/3 arrays avg), write %.3f GB;  read ratio %.2f  total ratio %.2f"
          % (tb, rd / 1e9, wr / 1e9, alg * 5 / 3 / 1e9, alg / 1e9, rd / (alg * 5 / 3), (rd + wr) / (alg * 8 / 3)))
PY
rm -rf $out/pmc_*

#!/bin/bash
# round 3, run 40: kernel durations of an intended-WENO5 step at 201^3 and 51^3, epsilon pre-pass against in-launch reduction + seam kernel
root=$PWD; out=$root/gpurun_out/r03an; mkdir -p $out; rm -rf $out/*
export TMPDIR=/tmp; cd /tmp
for n in 201 51; do for f in 0 1; do
  HJ_EPS_FUSE=$f timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t${n}_$f -o t -- python3 $root/bench.py --no-cpu-baseline --no-live-traffic --no-also --scheme WENO5 --n $n --steps 30 --repeats 2 > $out/b${n}_$f.txt 2>&1 || exit 1
done; done
cd $root
python - <<'PY'
import csv, glob
for n in (201, 51):
    for f in (0, 1):
        fn = glob.glob("gpurun_out/r03an/t%d_%d/**/*kernel_trace.csv" % (n, f), recursive=True)[0]
        rows = list(csv.DictReader(open(fn)))
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("fused_", "max_d1sq", "partials_to", "eps_seam"))][-270:]
        agg = {}
        for r in rows:
            k = r["Kernel_Name"].split("<")[0].replace("void hj::", "")
            agg.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        print("n=%d HJ_EPS_FUSE=%d: " % (n, f) + "; ".join("%s x%d mean %.2f us" % (k, len(v), sum(v) / len(v) / 1e3) for k, v in agg.items()))
        span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
        print("     span of these %d launches: %.1f us, sum of durations %.1f us" % (len(rows), span / 1e3, sum(sum(v) for v in agg.values()) / 1e3))
PY

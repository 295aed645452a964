#!/bin/bash
# round 3, run 29: grid-barrier latency probe + where a 51^3 / 101^3 step spends its 26 / 40 us (kernel durations against the step period)
out=gpurun_out/r03ac; mkdir -p $out; rm -rf $out/*
cp profiles/r03_grid_barrier_probe.txt $out/barrier.txt
cat $out/barrier.txt
cd /tmp && export TMPDIR=/tmp
for n in 51 101; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/trace$n -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-also --no-live-traffic --n $n --steps 100 --repeats 3 > $GRAFT_REPO_ROOT/$out/bench$n.txt 2>&1 || exit 1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for n in (51, 101):
    f = glob.glob("gpurun_out/r03ac/trace%d/**/*kernel_trace.csv" % n, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "substep" in r["Kernel_Name"] or "pair" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-240:]
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    p = [int(b["Start_Timestamp"]) - int(a["Start_Timestamp"]) for a, b in zip(rows, rows[1:])]
    p = [x for x in p if x < 50000]
    print("n=%d: kernel duration mean %.2f us (min %.2f), start-to-start period mean %.2f us, grid %s wg %s" % (n, sum(d) / len(d) / 1e3, min(d) / 1e3, sum(p) / len(p) / 1e3, rows[-1].get("Grid_Size"), rows[-1].get("Workgroup_Size")))
PY

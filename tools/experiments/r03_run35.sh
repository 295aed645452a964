#!/bin/bash
# round 3, run 35: LDS row pitch of the fp32 4-D pair kernel (ds_read_b64 pairs): C5 with 0 ... 32 extra cells of row padding
out=gpurun_out/r03ai; mkdir -p $out; rm -rf $out/*
for add in 0 2 4 6 8 10 12 14 16 18 20 22 24 26 28 30 32; do
  echo "== HJ_LDS_PITCH_ADD=$add" >> $out/ab.txt
  HJ_LDS_PITCH_ADD=$add HJ_DEBUG=1 HJ_AUTOTUNE=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-traffic --n 101 --steps 10 --repeats 3 --also C5 >> $out/ab.txt 2> $out/last.err || exit 1
  grep -E "pair tiling" $out/last.err | sort | uniq -c | sort -rn | head -1 >> $out/ab.txt
done
python - <<'PY'
import json
for ln in open("gpurun_out/r03ai/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:220]); continue
    d = json.loads(ln)
    for k, v in (d.get("also") or {}).items(): print("      also", k, {x: v.get(x) for x in ("value", "ms_per_step", "roofline_frac", "kernel", "error")})
PY

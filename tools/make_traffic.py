#!/usr/bin/env python3
"""profiles/traffic.json rows from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh.
bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB per launch (FETCH_SIZE doubled: MI355X_MICROARCH.md, gfx950 counts 128-B
read requests as 64 B; confirmed for this kernel's 8-B-per-lane accesses in profiles/r01_pmc_calibration.txt),
averaged over the launches of the fused kernel and multiplied by the launches of one RK3 step.  Each row carries
the hash of the kernel sources it was measured on; bench.py only reports a row whose hash matches."""
import csv, glob, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash():
    """Hash of every kernel / host source of the library (csrc/*.h, *.hip)."""
    import glob
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "levelsetpy_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.hip"))):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def mean_counter(d, name):
    vals = []
    # only the newest pass (gpurun merges every call's output into the local copy of the directory)
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and ("fused_substep_kernel" in r["Kernel_Name"] or "fused_pair_kernel" in r["Kernel_Name"]):
                vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals) if vals else None


out = sys.argv[1]
rows = {}
for n in (201, 513):
    f = mean_counter(os.path.join(out, "pmc_%d_FETCH_SIZE" % n), "FETCH_SIZE")
    w = mean_counter(os.path.join(out, "pmc_%d_WRITE_SIZE" % n), "WRITE_SIZE")
    if f is None or w is None:
        continue
    per_launch = (2 * f + w) * 1024
    rows["%d/WENO5_ASSHIPPED/float64" % n] = {
        "fetch_kib": f, "write_kib": w, "bytes_per_launch": per_launch, "bytes_per_step": 3 * per_launch,
        "algorithmic_bytes_per_step": n ** 3 * 64.0, "source_hash": source_hash(), "round": os.environ.get("HJ_ROUND", "r05")}
doc = {"_doc": __doc__.strip().replace("\n", " ")}
doc.update(rows)
print(json.dumps(doc, indent=1))
if rows and len(sys.argv) > 2 and sys.argv[2] == "--write":
    with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as fh:
        json.dump(doc, fh, indent=1)

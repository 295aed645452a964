#!/usr/bin/env python3
"""Per-kernel roofline table from a `rocprofv3 --kernel-trace --stats` CSV of tools/all_kernels.py: for every kernel of the library the
algorithmic bytes of one launch (units x bytes per unit, both stated), its average duration and the fraction of 8 TB/s.
usage: kernel_table.py <kernel_stats.csv>"""
import csv, re, sys

N3, N2, N4, N51 = 201 ** 3, 4096 ** 2, 129 ** 4, 51 ** 3
rows = list(csv.DictReader(open(sys.argv[1])))


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("hj::", "")
    return name


def classify(name):
    """-> (label, units, bytes per unit, what the bytes are, bound) or None for kernels that are not ours"""
    s = short(name)
    m = re.match(r"fused_pair_kernel<(\w+), Ham(\w+)<\w+>, (\d), (\d+), (\d+), (\d+), (\d+), (\d)>", s)
    if m:
        T, ham, sch, nt, r, kh, occ, mode = m.groups()
        es = 8 if T == "double" else 4
        units = {"DubinsRel": N3, "DoubleIntegrator": N2, "DoublePendulum": N4}[ham]
        words = {"1": 2, "2": 3, "0": 3}[mode]
        scheme = ["ENO2", "ENO3", "WENO5", "WENO5 as shipped"][int(sch)]
        bound = "fp64 VALU" if scheme in ("WENO5", "ENO3") and T == "double" else ("HBM / fp32 VALU" if ham == "DoublePendulum" else "HBM")
        return ("fused_pair_kernel %s %s (%s,%s,%s) %s" % (ham, scheme, nt, r, kh, {"1": "stage 1", "2": "stages 2,3", "0": "general"}[mode]), units, words * es,
                "read y%s, write out" % (" + y0" if words == 3 else ""), bound)
    m = re.match(r"fused_substep_kernel<(\w+), TermOp<\w+, (\d), (\d)>, (\d), (\d+), (\d+)", s)
    if m:
        T, nd, kind, sch, nt, r = m.groups()
        kindn = ["termNormal", "termReinit", "termConvection"][int(kind)]
        scheme = ["ENO2", "ENO3", "WENO5", "WENO5 as shipped"][int(sch)]
        return ("fused_substep_kernel %s %s (%s,%s) tiled, TermOp" % (kindn, scheme, nt, r), N3, 24, "read y + one coefficient array, write ydot", "fp64 VALU (uncontracted, sqrt / divisions)")
    m = re.match(r"fused_substep_kernel<(\w+), Ham(\w+)<\w+>, (\d)", s)
    if m:
        T, ham, sch = m.groups()
        es = 8 if T == "double" else 4
        units = {"DubinsRel": N3, "DoubleIntegrator": N2, "DoublePendulum": N4}[ham]
        if ham == "DubinsRel":      # tools/all_kernels.py runs the one-cell-per-lane kernel on the 51^3 grid only
            return ("fused_substep_kernel DubinsRel (51^3, tiled, one cell per lane)", N51, 8 / 3 * es, "8/3 words (stage average)", "launch latency (7-9 us floor)")
        return ("fused_substep_kernel %s scheme %s" % (ham, sch), units, 8 / 3 * es, "8/3 words (stage average)", "HBM")
    if s.startswith("direct_substep_kernel"):
        return ("direct_substep_kernel (51^3)", N51, 64 / 3, "8/3 words (stage average)", "launch latency (8 us floor)")
    if s.startswith("upwind_kernel"):
        return ("upwind_kernel (one dimension)", N3, 24, "read phi, write derivL + derivR", "HBM")
    if s.startswith("lf_split_end_kernel"):
        return ("lf_split_end_kernel", N3, 80, "read 6 derivatives + 2 alpha arrays + ham, write ydot", "HBM")
    if s.startswith("rk_combine_kernel"):
        return ("rk_combine_kernel", N3, 28, "read 2-3 arrays, write 1 (average 3.5 words)", "HBM")
    if s.startswith("max_d1sq_kernel"):
        return ("max_d1sq_kernel (epsilon pre-pass)", N3, 8, "read y", "HBM")
    if s.startswith("eps_seam_kernel"):
        return ("eps_seam_kernel", N3, 0.9, "read the seam planes / rows of the output (~11 % of the cells)", "launch latency")
    m = re.match(r"term_kernel<\w+, \d, (\d), (\d)>", s)
    if m:
        kind = ["termNormal", "termReinit", "termConvection"][int(m.group(2))]
        return ("term_kernel %s (direct form, one cell per thread)" % kind, N3, 24, "read y + one coefficient array, write ydot", "L2 / fp64 VALU")
    if s.startswith("minmax_kernel"):
        return ("minmax_kernel (post-step min / max)", N3, 24, "read 2, write 1", "HBM")
    if s.startswith("any_nan_kernel"):
        return ("any_nan_kernel", N3, 8, "read y", "HBM")
    if s.startswith("alpha_bound_kernel"):
        ham = "DoublePendulum 129^4" if "Pendulum" in s else ("DoubleIntegrator 4096^2" if "Integrator" in s else "DubinsRel 201^3 / 51^3")
        return ("alpha_bound_kernel %s (static step bound, once per grid)" % ham, 1, 0, "no array traffic (coordinates from tables)", "VALU")
    if s.startswith("partials_to_values") or s.startswith("keys_to_values"):
        return (s.split("<")[0], 1, 0, "a few hundred bytes", "launch latency")
    if s.startswith("ghost_kernel"):
        return ("ghost_kernel", N3, 16, "read n, write n + 2w", "HBM")
    return None


out = []
for r in rows:
    c = classify(r["Name"])
    if c is None:
        continue
    label, units, bpu, what, bound = c
    avg_us = float(r["AverageNs"]) / 1e3
    gbs = units * bpu / (avg_us * 1e-6) / 1e9 if bpu else 0.0
    out.append((label, int(r["Calls"]), avg_us, units, bpu, what, gbs, gbs / 8000.0, bound))
out.sort(key=lambda x: -x[2] * x[1])
print("%-74s %6s %10s %12s %7s %9s %6s  %s" % ("kernel", "calls", "avg us", "units", "B/unit", "GB/s", "frac", "bound; what the bytes are"))
for label, calls, us, units, bpu, what, gbs, frac, bound in out:
    print("%-74s %6d %10.1f %12d %7.2f %9.0f %6.3f  %s; %s" % (label[:74], calls, us, units, bpu, gbs, frac, bound, what))

#!/usr/bin/env python3
"""Static instruction mix of the plane loops of the dominant kernels, from the BUILT libhj_mi355x.so (no GPU needed):
   tools/isa_mix.py [tag]  ->  profiles/<tag>_isa_mix.txt
Extracts the gfx950 code objects of the library, disassembles them and runs tools/kernel_isa_stats.py on the kernels bench.py's
legs launch.  tools/profile_round.sh calls it, so the file cannot go stale against the library the profiles were taken with
(VERDICT r04 item 7c)."""
import hashlib, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from check_store_hazard import code_objects, LLVM      # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
lib = os.path.join(ROOT, "levelsetpy_amd", "csrc", "libhj_mi355x.so")
# (what to look for in the demangled name, label, cells per thread and plane)
KERNELS = [
    (r"fused_pair_kernel<double, hj::HamDubinsRel<double>, 3, 512, 2, 2, 2, 2>", "headline 201^3 / 513^3, stages 2-3: pair kernel, WENO5_ASSHIPPED, fp64 Dubins", 4),
    (r"fused_pair_kernel<double, hj::HamDubinsRel<double>, 3, 512, 2, 2, 2, 1>", "headline, stage 1 (no y0)", 4),
    (r"fused_pair_kernel<double, hj::HamDubinsRel<double>, 2, 256, 1, 2, 2, 2>", "intended WENO5 (201^3 WENO5), stages 2-3", 2),
    (r"fused_pair_kernel<double, hj::HamDoubleIntegrator<double>, 1, 256, 1, 2, 2, 2>", "C3 (4096^2 ENO3, bit-exact arithmetic), stages 2-3", 2),
    (r"fused_pair_kernel<double, hj::HamDoubleIntegrator<double>, 5, 512, 2, 2, 2, 2>", "C3 fast (ENO3 in the lean arithmetic, two pairs per thread), stages 2-3", 4),
    (r"fused_pair4_kernel<float, hj::HamDoublePendulum<float>, 3, 512, 2, 5, 6, 66, 2, false, 2>", "C5 (129^4 fp32 pendulum), stages 2-3: compile-time tile 5x6x66", 4),
    (r"fused_flat4_kernel<float, hj::HamDoublePendulum<float>, 3, 512, 2, 3, 5, 140, 2, false, 2>", "C5, stages 2-3: full-row kernel 3x5 rows (round 6)", 4),
]
out = os.path.join(ROOT, "profiles", "%s_isa_mix.txt" % tag)
with tempfile.TemporaryDirectory(dir="/tmp") as tmp, open(out, "w") as fh:
    sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
    fh.write("%s: static instruction mix of the plane loops of the dominant kernels (tools/isa_mix.py -> tools/kernel_isa_stats.py on the code objects of the built\n"
             "libhj_mi355x.so, sha256 %s...; a loop iteration = TWO planes, the prefetch depth the loop is unrolled by)\n" % (tag, sha))
    names = {}
    for k, co in enumerate(code_objects(lib, tmp)):
        dis = os.path.join(tmp, "co%d.s" % k)
        with open(dis, "w") as f:
            subprocess.run([LLVM + "/llvm-objdump", "-d", co], stdout=f, check=True)
        syms = re.findall(r"^[0-9a-f]+ <(_Z\S+)>:$", open(dis).read(), re.M)
        if syms:
            dem = subprocess.run(["c++filt"], input="\n".join(syms), capture_output=True, text=True).stdout.splitlines()
            for m, d in zip(syms, dem):
                names[d] = (dis, m)
    for pat, label, cpt in KERNELS:
        hit = [(d, v) for d, v in names.items() if pat in d]
        fh.write("\n== %s\n   %s\n" % (label, pat))
        if not hit:
            fh.write("   (not in this build)\n")
            continue
        d, (dis, mangled) = hit[0]
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_isa_stats.py"), dis, mangled], capture_output=True, text=True)
        txt = r.stdout
        loop = txt[txt.find("loop of"):] if "loop of" in txt else txt
        fh.write(loop)
        m = re.search(r"loop of (\d+) instructions", loop)
        valu = sum(int(x) for x in re.findall(r"VALU \S+\s+(\d+)", loop))
        salu = sum(int(x) for x in re.findall(r"SALU\s+(\d+)", loop))
        if m:
            fh.write("   -> per cell-plane (%d cells per thread and plane, 2 planes per iteration): %.1f VALU, %.1f SALU, %.1f instructions in all\n"
                     % (cpt, valu / (2.0 * cpt), salu / (2.0 * cpt), int(m.group(1)) / (2.0 * cpt)))
print(out)

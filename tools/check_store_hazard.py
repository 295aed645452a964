#!/usr/bin/env python3
"""Machine check of the 16-byte-store hazard rule on the BUILT library (no GPU needed).

gfx950 / ROCm 7.2 finding of round 2: `buffer_store_dwordx4 vD[0:3], vOff, sRsrc, sOff offen` with an
SGPR soffset reads its data registers late, and hipcc's hazard recogniser adds no wait states for that form.  A VALU
(or VMEM/LDS-return) write to vD within WAIT wait states of the store corrupts the stored value.  The kernels keep the
data registers alive up to an `s_nop` placed after the store; this script disassembles every gfx950 code object in
libhj_mi355x.so and verifies, for every >8-byte buffer/global store, that no instruction writes one of its data
registers before at least WAIT wait states have passed (s_nop N counts N+1, every other instruction counts 1;
a branch/label ends the scan conservatively as a violation unless the wait states are already satisfied).

usage: check_store_hazard.py [lib.so]      exit status 0 = clean
"""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
WAIT = 2
WIDE = re.compile(r"^\s*(buffer_store_dwordx[34]|global_store_dwordx[34]|flat_store_dwordx[34])\s+(.*)$")
VREG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def code_objects(lib, tmp):
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    blob = open(fat, "rb").read()
    offs, p = [], blob.find(MAGIC)
    while p >= 0:
        offs.append(p)
        p = blob.find(MAGIC, p + 1)
    outs = []
    for i, o in enumerate(offs):
        part = os.path.join(tmp, f"b{i}.bin")
        open(part, "wb").write(blob[o:offs[i + 1] if i + 1 < len(offs) else len(blob)])
        co = os.path.join(tmp, f"co{i}.o")
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        outs.append(co)
    return outs


def regs(tok):
    m = VREG.fullmatch(tok.strip())
    if not m:
        return set()
    if m.group(3) is not None:
        return {int(m.group(3))}
    return set(range(int(m.group(1)), int(m.group(2)) + 1))


def dest_regs(ins, ops):
    """VGPRs an instruction writes.  VALU / DPP / SDWA / MFMA / loads write their first operand (a scalar or condition
    destination there yields no VGPR); the swap family writes BOTH of its operands; loads into LDS (`... lds`),
    stores, SALU, waits and compares write none."""
    if ins.startswith(("buffer_store", "global_store", "flat_store", "scratch_store", "ds_write", "ds_store", "s_",
                       "buffer_wbl2", "buffer_inv", "buffer_atomic", "global_atomic", "flat_atomic")) and "_rtn" not in ins:
        if not (ins.startswith(("buffer_atomic", "global_atomic", "flat_atomic")) and " sc0" in (" " + ops)):
            return set()
    if ins.startswith("v_cmp"):                         # v_cmp / v_cmpx write VCC / EXEC / an SGPR pair
        return set()
    if ops.rstrip().endswith(" lds") or " lds " in ops:  # LDS-DMA: no register destination
        return set()
    toks = [t.strip() for t in ops.split(",")] if ops else []
    if not toks:
        return set()
    d = regs(toks[0].split()[0]) if toks[0] else set()
    if ins.startswith(("v_swap_b", "v_permlane16_swap", "v_permlane32_swap")) and len(toks) > 1:
        d |= regs(toks[1].split()[0])
    return d


def check(co):
    txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True,
                         text=True).stdout.splitlines()
    lines = []
    for ln in txt:
        s = ln.split("//")[0].strip()
        if not s:
            continue
        lines.append(s)
    bad, nstores = [], 0
    for i, s in enumerate(lines):
        m = WIDE.match(s)
        if not m:
            continue
        nstores += 1
        data = regs(m.group(2).split(",")[0])
        waited, j = 0, i + 1
        while waited < WAIT and j < len(lines):
            t = lines[j]
            j += 1
            if t.endswith(":"):                       # label: another path may join here; only fine if already waited
                bad.append((co, i, s, "label before wait states: " + t))
                break
            parts = t.split(None, 1)
            ins, ops = parts[0], (parts[1] if len(parts) > 1 else "")
            if ins == "s_nop":
                waited += int(ops, 0) + 1
                continue
            if dest_regs(ins, ops) & data:
                bad.append((co, i, s, "data register written after %d wait state(s) by: %s" % (waited, t)))
                break
            if ins.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
                if ins == "s_endpgm":
                    break
                bad.append((co, i, s, "branch before wait states: " + t))
                break
            waited += 1
    return nstores, bad


def main(lib):
    with tempfile.TemporaryDirectory() as tmp:
        total, bad = 0, []
        for co in code_objects(lib, tmp):
            n, b = check(co)
            total += n
            bad += b
    return total, bad


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "levelsetpy_amd", "csrc", "libhj_mi355x.so")
    total, bad = main(lib)
    print("%d wide stores checked, %d violations" % (total, len(bad)))
    for co, i, s, why in bad[:40]:
        print("  %s line %d: %s\n      %s" % (os.path.basename(co), i, s, why))
    sys.exit(1 if bad else 0)

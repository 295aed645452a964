#!/usr/bin/env python3
"""Instruction mix of one kernel in a disassembly (`llvm-objdump -d` of a gfx950 code object, e.g. the .s files
tools/check_store_hazard.py's code_objects() + objdump produce):
   kernel_isa_stats.py file.s <mangled-name-substring>
Counts per instruction class for the whole kernel and for its longest backward-branch loop (the plane loop)."""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.endswith(">:") and key in l and re.match(r"^[0-9a-f]+ <_Z", l))
base = int(lines[start].split()[0], 16)
ins = []          # (addr, mnemonic, operands, branch target or None)
for l in lines[start + 1:]:
    if re.match(r"^[0-9a-f]+ <_Z", l):
        break
    m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$", l)
    if not m:
        continue
    t = re.search(r"\+0x([0-9a-fA-F]+)>", m.group(4))
    ins.append((int(m.group(3), 16), m.group(1), m.group(2), base + int(t.group(1), 16) if t else None))

def cls(i):
    if i.startswith("v_accvgpr"): return "accvgpr move"
    if i.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane move (SGPR spill)"
    if i.startswith("v_") and "f64" in i: return "VALU f64"
    if i.startswith("v_"): return "VALU other"
    if i.startswith("ds_"): return "LDS"
    if i.startswith(("buffer_", "global_", "flat_", "scratch_")): return "VMEM"
    if i.startswith("s_waitcnt"): return "s_waitcnt"
    if i.startswith("s_barrier"): return "s_barrier"
    if i.startswith("s_nop"): return "s_nop"
    if i.startswith("s_"): return "SALU"
    return "other"

def summarize(sel, title):
    c = collections.Counter(cls(i[1]) for i in sel)
    print("%s: %d instructions" % (title, sum(c.values())))
    for k, v in c.most_common():
        print("   %-24s %6d" % (k, v))
    f = collections.Counter(i[1] for i in sel if cls(i[1]) in ("LDS", "VMEM", "VALU other"))
    print("   detail:", dict(f.most_common(24)))

summarize(ins, "whole kernel")
index = {a: k for k, (a, _, _, _) in enumerate(ins)}
loops = []
for k, (a, i, o, t) in enumerate(ins):
    if t is not None and t < a and t in index:
        loops.append((k - index[t], index[t], k))
loops.sort(reverse=True)
for span, lo, hi in loops[:int(sys.argv[3]) if len(sys.argv) > 3 else 1]:
    summarize(ins[lo:hi + 1], "loop of %d instructions (index %d..%d)" % (span + 1, lo, hi))

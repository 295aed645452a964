#!/usr/bin/env python3
"""Which kernel instantiations of libhj_mi355x.so does a run launch?  Compares the (calls, name) table that tools/experiments/r05_run52.sh brings back from
a rocprofv3 --kernel-trace --stats run of the GPU suite with the host stubs of the library (nm -C: one per instantiated __global__ template).
    python tools/kernel_coverage.py gpurun_out/r52_kernels_launched.txt [levelsetpy_amd/csrc/libhj_mi355x.so]"""
import collections
import re
import subprocess
import sys


def norm(s):
    s = re.sub(r'^void ', '', s.strip()).replace('hj::__device_stub__', 'hj::')
    i = s.find('<')
    if i < 0:
        return s.split('(')[0]
    depth = 0
    for k in range(i, len(s)):
        depth += s[k] == '<'
        if s[k] == '>':
            depth -= 1
            if depth == 0:
                return s[:k + 1]
    return s


args = [a for a in sys.argv[1:] if not a.startswith("-")]
table = args[0]
lib = args[1] if len(args) > 1 else "levelsetpy_amd/csrc/libhj_mi355x.so"
launched = collections.Counter()
for ln in open(table):
    c, n = ln.rstrip('\n').split('\t', 1)
    launched[norm(n)] += int(c)
stubs = subprocess.run("nm -C %s | grep __device_stub__" % lib, shell=True, capture_output=True, text=True).stdout
inst = {norm(ln.split(' ', 2)[2]) for ln in stubs.splitlines()}
hit = {k for k in launched if k in inst}
rtc = {k for k in launched if k.startswith('hj::') and k not in inst}
print("%d instantiations in the library, %d launched (%d launches); %d run-time (hipRTC) instantiations launched besides" % (
    len(inst), len(hit), sum(launched[k] for k in hit), len(rtc)))
by = collections.Counter(m.split('<')[0].replace('hj::', '') for m in inst - hit)
tot = collections.Counter(m.split('<')[0].replace('hj::', '') for m in inst)
for k in sorted(tot):
    print("  %-24s %3d of %3d launched" % (k, tot[k] - by[k], tot[k]))
if "-v" in sys.argv:
    for m in sorted(inst - hit):
        print("not launched:", m)

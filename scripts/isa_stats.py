#!/usr/bin/env python3
"""Static instruction statistics of one kernel of csrc/mcq_engine.hip (device ISA via hipcc -S): totals by class and the
SGPR-spill traffic (v_readlane / v_writelane).  python3 scripts/isa_stats.py KERNEL_REGEX [--src FILE] [hipcc flags]
e.g.  scripts/isa_stats.py 'k_query_waveIjLi512ELb0ELb0ELb0E'      (mangled-name regex)"""
import collections
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
pat = re.compile(args.pop(0))
src = os.path.join(ROOT, "metacache-mpi_amd", "csrc", "mcq_engine.hip")
if "--src" in args:
    i = args.index("--src"); src = args[i + 1]; del args[i:i + 2]
asm = src if src.endswith(".s") else "/tmp/isa_stats_%d.s" % os.getpid()
if not src.endswith(".s"):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", asm] + args,
                   check=True, stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
i = 0
while i < len(lines):
    m = re.match(r"^(_Z\w+):", lines[i])
    if m and pat.search(m.group(1)):
        j = i
        while not lines[j].startswith(".Lfunc_end"):
            j += 1
        c = collections.Counter()
        for l in lines[i + 1:j]:
            l = l.strip()
            if not l or l[0] in ";." or l.endswith(":"):
                continue
            c[l.split()[0]] += 1
        tot = sum(c.values())
        cls = lambda f: sum(v for k, v in c.items() if f(k))
        print(m.group(1)[:90])
        print("  total %d  valu %d  salu %d  ds %d  vmem %d  s_load %d  waitcnt %d  v_readlane %d  v_writelane %d  s_nop %d" % (
            tot, cls(lambda k: k.startswith("v_")), cls(lambda k: k.startswith("s_") and not k.startswith(("s_load", "s_waitcnt"))),
            cls(lambda k: k.startswith("ds_")), cls(lambda k: k.startswith(("global_", "buffer_", "flat_", "scratch_"))),
            cls(lambda k: k.startswith("s_load")), c["s_waitcnt"], c["v_readlane_b32"], c["v_writelane_b32"], c["s_nop"]))
        i = j
    i += 1

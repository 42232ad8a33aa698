#!/usr/bin/env python3
"""Kernel resource usage of csrc/mcq_engine.hip as hipcc reports it (-Rpass-analysis=kernel-resource-usage):
one line per kernel with VGPRs, spills, scratch, LDS, occupancy.  Runs in the build container (no GPU needed):
  python3 scripts/resource_usage.py [--filter REGEX] [extra hipcc flags]
  python3 scripts/resource_usage.py --from saved_stderr.txt          (a compile's remarks kept in a file)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
flt = None
if "--filter" in args:
    i = args.index("--filter"); flt = re.compile(args[i + 1]); del args[i:i + 2]
if args[:1] == ["--from"]:
    err = open(args[1]).read()
else:
    src = os.path.join(ROOT, "metacache-mpi_amd", "csrc", "mcq_engine.hip")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
           "-c", src, "-o", "/dev/null"] + args
    err = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
names = re.findall(r"remark: Function Name: (\S+)", err)
dem = dict(zip(names, subprocess.run(["c++filt"] + names, stdout=subprocess.PIPE, text=True).stdout.splitlines())) if names else {}
cur, rows = None, {}
for line in err.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = re.sub(r"\(.*", "", dem[m.group(1)]).replace("void ", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[A-Za-z/ ]*\])?): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print("%-78s %5s %6s %6s %7s %6s %4s" % ("kernel", "VGPR", "SGPRsp", "VGPRsp", "scratch", "LDS", "occ"))
for k, r in rows.items():
    if flt and not flt.search(k):
        continue
    print("%-78s %5d %6d %6d %7d %6d %4d" % (k[:78], r.get("VGPRs", -1), r.get("SGPRs Spill", -1), r.get("VGPRs Spill", -1),
                                            r.get("ScratchSize [bytes/lane]", -1), r.get("LDS Size [bytes/block]", -1),
                                            r.get("Occupancy [waves/SIMD]", -1)))

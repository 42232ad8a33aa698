#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (gpurun_out/<dir>) into small files under profiles/.
usage: summarize_prof.py <tag> <kernel_stats_dir> [<pmc_dir> ...]"""
import csv, glob, os, sys, collections

def short(n):
    n = n.replace("void ", "")
    return n[:n.index("(")] if "(" in n else n[:80]

tag = sys.argv[1]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(out, exist_ok=True)
d = sys.argv[2]
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(out, tag + "_kernel_stats.csv"), "w") as w:
        w.write("# rocprofv3 --kernel-trace --stats (condensed: our kernels + top 8 others; names shortened)\n")
        w.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        ours = [r for r in rows if short(r["Name"]).startswith("k_")]
        others = [r for r in rows if not short(r["Name"]).startswith("k_")][:8]
        for r in ours + others:
            w.write("%s,%s,%s,%s,%s,%s,%s\n" % (short(r["Name"])[:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]))
    print(open(os.path.join(out, tag + "_kernel_stats.csv")).read())
pm = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[3:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.startswith("k_query") or k.startswith("k_reduce") or k.startswith("k_sketch") or k.startswith("k_lookup") or k.startswith("k_shard"):
                pm[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
if pm:
    with open(os.path.join(out, tag + "_pmc.csv"), "w") as w:
        w.write("# rocprofv3 --pmc passes (separate runs), per-dispatch mean over the dispatches of each kernel\n")
        w.write("Kernel,Counter,Dispatches,MeanPerDispatch\n")
        for k in sorted(pm):
            for c in sorted(pm[k]):
                v = pm[k][c]
                w.write("%s,%s,%d,%.6g\n" % (k, c, len(v), sum(v) / len(v)))
    print(open(os.path.join(out, tag + "_pmc.csv")).read())

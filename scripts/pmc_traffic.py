#!/usr/bin/env python3
"""profiles/<tag>_pmc.csv (scripts/profile.sh + summarize_prof.py) -> profiles/pmc_traffic_<workload>.json: the HBM traffic and
instruction counts per launch of the workload's dominant kernel, stamped with the digest of the kernel sources they were
collected from (bench.py quotes them only for that state).  usage: pmc_traffic.py <tag> <workload> <kernel prefix> <reads per launch>"""
import csv
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag, workload, kprefix, reads = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
rows = []                  # (kernel names hold commas and are not quoted: the three value columns are split off from the right)
for l in open(os.path.join(ROOT, "profiles", tag + "_pmc.csv")):
    if l.startswith("#") or l.startswith("Kernel,"):
        continue
    k_, c_, d_, m_ = l.rstrip("\n").rsplit(",", 3)
    rows.append({"Kernel": k_.strip('"'), "Counter": c_, "Dispatches": d_, "MeanPerDispatch": m_})
sum_all = kprefix == "ALL"            # a workload carried by several kernels (RefSeq scale: three wave stages + the workgroup kernels): their sum
kernels = sorted({r["Kernel"] for r in rows if r["Kernel"].startswith("k_query" if sum_all else kprefix)})
if not kernels:
    sys.exit("no kernel starting with %r in profiles/%s_pmc.csv" % (kprefix, tag))
# the instantiation with the most FETCH_SIZE is the one that ran the workload
def val1(k, c):
    v = [float(r["MeanPerDispatch"]) for r in rows if r["Kernel"] == k and r["Counter"] == c]
    return v[0] if v else None
k = max(kernels, key=lambda x: val1(x, "FETCH_SIZE") or 0)
def val(k_, c):
    if not sum_all:
        return val1(k_, c)
    v = [val1(x, c) for x in kernels]          # (every kernel of the path runs once per batch: per-dispatch means add up)
    return sum(x for x in v if x is not None) if any(x is not None for x in v) else None
fetch, write = val(k, "FETCH_SIZE"), val(k, "WRITE_SIZE")
out = {"kernel": ("sum over " + ", ".join(kernels)) if sum_all else k, "reads_per_launch": reads, "fetch_size_kb": fetch, "write_size_kb": write,
       "hbm_bytes_per_launch": (fetch + write) * 1024.0,
       "valu_insts_per_launch": val(k, "SQ_INSTS_VALU"), "salu_insts_per_launch": val(k, "SQ_INSTS_SALU"),
       "lds_insts_per_launch": val(k, "SQ_INSTS_LDS"), "tcc_ea0_rdreq_per_launch": val(k, "TCC_EA0_RDREQ_sum"),
       "sq_wait_any_over_wave_cycles": (val(k, "SQ_WAIT_ANY") / val(k, "SQ_WAVE_CYCLES")) if val(k, "SQ_WAIT_ANY") and val(k, "SQ_WAVE_CYCLES") else None,
       "lds_bank_conflict_over_idx_active": (val(k, "SQ_LDS_BANK_CONFLICT") / val(k, "SQ_LDS_IDX_ACTIVE")) if val(k, "SQ_LDS_BANK_CONFLICT") and val(k, "SQ_LDS_IDX_ACTIVE") else None,
       "csrc_digest": importlib.import_module("metacache-mpi_amd").source_digest(),
       "source": "profiles/%s_pmc.csv (rocprofv3 --pmc, separate passes; %s)" % (tag, "all k_query* kernels of a batch, summed" if sum_all else "the dominant kernel of the workload"),
       "correction": "none: FETCH_SIZE x 1 KB = 64 B x TCC_EA0_RDREQ for this random-sector pattern (profiles/r01_fetch_calibration.txt); the x2 of wide coalesced streams does not apply"}
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % workload), "w"), indent=1)
print(json.dumps(out, indent=1))

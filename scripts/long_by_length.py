#!/usr/bin/env python3
"""What a long read costs the workgroup kernel by its length (run on the GPU box): batches of fixed-length reads (8 % substitutions) on
the configs[1] table, time of mcq_query per batch -> microseconds of one workgroup per read (512 workgroups)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
eng = importlib.import_module("metacache-mpi_amd.engine")
synth = importlib.import_module("metacache-mpi_amd.synth")
dev = torch.device("cuda", 0)
gb, goff, species = synth.make_genomes(50, 10, 2_000_000, 6_000_000, 0.02, seed=3, device=dev)
sp32 = species.to(torch.int32).contiguous()
db = eng.Database.build(gb.data_ptr(), goff.data_ptr(), sp32.data_ptr(), goff.numel() - 1, emulate_ranks=2)
st = torch.cuda.current_stream(dev).cuda_stream
out = []
for L, n in ((2000, 32768), (4000, 16384), (8000, 16384), (12000, 8192), (16000, 8192), (24000, 8192), (30000, 4096), (40000, 4096), (56000, 4096)):
    r, ro, _ = synth.sample_reads(gb, goff, n, L, 0.08, 0.0, seed=L)
    ws = eng.Workspace(db, n, r.numel(), max_locs_per_query=1 << 16)
    c = torch.zeros((n, 2, 4), dtype=torch.int32, device=dev); nc = torch.zeros(n, dtype=torch.int32, device=dev)
    for _ in range(2):
        ws.query_device(r.data_ptr(), ro.data_ptr(), n, False, c.data_ptr(), nc.data_ptr(), max_cand=2, emulate_ranks=2, stream=st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 4
    for _ in range(K):
        ws.query_device(r.data_ptr(), ro.data_ptr(), n, False, c.data_ptr(), nc.data_ptr(), max_cand=2, emulate_ranks=2, stream=st)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / K
    s = ws.sync()
    out.append(dict(L=L, n=n, ms=round(1e3 * el, 3), us_wg_per_read=round(1e6 * el * 512 / n, 1), T=round(s["n_locations"] / n), us_per_kb=round(1e6 * el * 512 / n / (L / 1000), 2)))
    print(out[-1], flush=True)
    ws.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "long_by_length.json"), "w"), indent=1)

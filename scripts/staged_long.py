#!/usr/bin/env python3
"""Long reads: the fused path (one 1024-thread workgroup per read for everything) against the same work as separate
kernels through the staged entry points -- mcq_sketch (one wave per window), mcq_lookup_count, mcq_lookup_gather (one
wave per 64 features), mcq_reduce (sort + sweep + top lists per read) -- with the match lists going through HBM.
Prints ms per batch of both and whether the results are the same.  (Run on the GPU box from the repo root.)"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    pkg = importlib.import_module("metacache-mpi_amd")
    pkg.build_hip()
    eng = importlib.import_module("metacache-mpi_amd.engine")
    synth = importlib.import_module("metacache-mpi_amd.synth")
    dev = torch.device("cuda", 0)
    B, mean = 16384, 8000
    gb, goff, species = synth.make_genomes(50, 10, 3_000_000, 5_000_000, 0.02, seed=3, device=dev)
    table = eng.Table(gb.data_ptr(), goff.data_ptr(), goff.numel() - 1, emulate_ranks=2, device=0)
    sp32 = species.to(torch.int32).contiguous()
    db = eng.Database(None, None, None, None, device=0,
                      device_ptrs=dict(keys=table.keys_ptr, list_off=table.list_off_ptr, locs=table.locs_ptr, tgt2tax=sp32.data_ptr(),
                                       n_keys=table.n_keys, n_locs=table.n_locs, n_targets=sp32.numel()))
    table.close()
    batches = [synth.sample_long_reads(gb, goff, B, mean, 0.08, seed=1000 + i)[:2] for i in range(4)]
    max_bases = max(int(o[-1].item()) for _, o in batches)
    ws = eng.Workspace(db, B, max_bases)
    st = torch.cuda.current_stream(dev).cuda_stream
    s = db.sketch_size
    cands = torch.zeros((B, 2, 4), dtype=torch.int32, device=dev); ncand = torch.zeros(B, dtype=torch.int32, device=dev)
    cands2 = torch.zeros_like(cands); ncand2 = torch.zeros_like(ncand)
    loc_dtype = torch.int32 if db.loc_bytes() == 4 else torch.int64
    bound = max_bases // db.winstride + 2 * B
    win_off = torch.empty(B + 1, dtype=torch.int64, device=dev)
    feats = torch.empty((bound, s), dtype=torch.int32, device=dev)
    nfeat = torch.empty(bound, dtype=torch.int32, device=dev)
    lens = torch.zeros(bound * s, dtype=torch.int32, device=dev)
    src = torch.empty(bound * s, dtype=torch.int64, device=dev)
    off = torch.zeros(bound * s + 1, dtype=torch.int64, device=dev)
    locs = torch.empty(bound * s * 8, dtype=loc_dtype, device=dev)

    def fused(i):
        r, ro = batches[i % 4]
        ws.query_device(r.data_ptr(), ro.data_ptr(), B, False, cands.data_ptr(), ncand.data_ptr(), max_cand=2, emulate_ranks=2, stream=st)

    def staged(i):
        r, ro = batches[i % 4]
        db.count_windows(r.data_ptr(), ro.data_ptr(), B, win_off.data_ptr(), st)
        db.sketch(r.data_ptr(), ro.data_ptr(), B, win_off.data_ptr(), feats.data_ptr(), nfeat.data_ptr(), st)
        n = bound * s                    # (slots behind the batch's last window hold stale features: harmless for timing; masked below)
        db.lookup_count(feats.data_ptr(), n, lens.data_ptr(), src.data_ptr(), st)
        torch.cumsum(lens, 0, dtype=torch.int64, out=off[1:])
        db.lookup_gather(feats.data_ptr(), n, off.data_ptr(), locs.data_ptr(), lens.data_ptr(), src.data_ptr(), st)
        loc_off = off[win_off * s]
        qlen = (ro[1:] - ro[:-1]).to(torch.int32)
        ws.reduce_device(B, loc_off.data_ptr(), locs.data_ptr(), qlen.data_ptr(), cands2.data_ptr(), ncand2.data_ptr(), max_cand=2,
                         emulate_ranks=2, stream=st)

    feats.fill_(-1)                          # MCQ_EMPTY: unused slots have no list
    for name, fn in (("fused", fused), ("staged", staged)):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(12):
            fn(3 + i)
        torch.cuda.synchronize(dev)
        print("%s: %.3f ms per batch of %d long reads" % (name, (time.perf_counter() - t0) / 12 * 1e3, B))
    fused(1); feats.fill_(-1); staged(1)
    torch.cuda.synchronize(dev)
    okn = bool(torch.equal(ncand, ncand2))
    m = torch.arange(2, device=dev)[None, :] < ncand[:, None]
    print("same results:", okn and bool(torch.equal(cands[m], cands2[m])))


if __name__ == "__main__":
    main()

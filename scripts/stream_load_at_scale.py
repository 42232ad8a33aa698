#!/usr/bin/env python3
"""Row f1 at size (run on the GPU box from the repo root): a shard set of the reference's file format, >= 20 GB, written from a
GPU-built table (mcq_refdb_write_shard, one file per reference rank, targets split by tgt % P as the reference does), then opened
by `mcq_query_cli` through the STREAMING route (include/mcq_open.hpp: heads on the host, key records to the GPU in chunks of
4 M locations, the P ranks merged there per feature-hash range) -- wall time of the load, peak host memory of the process
(ru_maxrss), and the mapping lines of 65 536 read pairs against the lines the directly built handle gives for the same reads.

Writes gpurun_out/stream_load_at_scale.json (and prints it)."""
import argparse
import ctypes as C
import importlib
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--species", type=int, default=400)
    ap.add_argument("--strains", type=int, default=10)
    ap.add_argument("--ranks", type=int, default=4)
    ap.add_argument("--pairs", type=int, default=65536)
    ap.add_argument("--workdir", default="/tmp/mcq_streamload")
    ap.add_argument("--keep", action="store_true")
    a = ap.parse_args()
    import torch
    pkg = importlib.import_module("metacache-mpi_amd")
    pkg.build_hip(); pkg.build_host()
    eng = importlib.import_module("metacache-mpi_amd.engine")
    host = importlib.import_module("metacache-mpi_amd.host")
    synth = importlib.import_module("metacache-mpi_amd.synth")
    dev = torch.device("cuda", 0)
    P = a.ranks
    shutil.rmtree(a.workdir, ignore_errors=True)
    os.makedirs(a.workdir)
    res = {"table": "%d species x %d strains" % (a.species, a.strains), "reference_ranks": P}
    free_disk = shutil.disk_usage(a.workdir).free
    res["free_disk_gb"] = round(free_disk / 1e9, 1)

    gb, goff, species = synth.make_genomes(a.species, a.strains, 2_000_000, 6_000_000, 0.02, seed=3, device=dev)
    n_targets = goff.numel() - 1
    r, ro, _ = synth.sample_pairs(gb, goff, a.pairs, 150, 300, 500, 0.005, 0.001, seed=1000)
    table = eng.Table(gb.data_ptr(), goff.data_ptr(), n_targets, emulate_ranks=P, device=0)
    glen = np.diff(goff.cpu().numpy().astype(np.int64))
    sp = species.cpu().numpy().astype(np.int64)
    del gb
    torch.cuda.empty_cache()
    nk, nl = table.n_keys, table.n_locs
    est_bytes = nk * 21 * 1.3 + nl * 8                      # (a key is in 1.3 rank files on average here)
    res.update(db_bp=int(glen.sum()), db_keys=int(nk), db_locations=int(nl), targets=int(n_targets), estimated_shard_bytes=int(est_bytes))
    if free_disk < 1.5 * est_bytes:
        res["skipped"] = "not enough disk space under %s" % a.workdir
        print(json.dumps(res, indent=1)); return

    # the table's arrays as tensors (device copies of the builder's buffers)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def dev_tensor(ptr, n, dt):
        t = torch.empty(n, dtype=dt, device=dev)
        assert hip.hipMemcpy(t.data_ptr(), ptr, t.numel() * t.element_size(), 3) == 0          # device to device
        return t
    keys = dev_tensor(table.keys_ptr, nk, torch.int32)
    off = dev_tensor(table.list_off_ptr, nk + 1, torch.int64)
    locs = dev_tensor(table.locs_ptr, nl, torch.int64)
    table_handle = table                                    # (its buffers stay until the direct handle is made, after the files are written)

    # ---- the shard files, rank by rank (host memory: one rank's arrays)
    W, S = 128, 113
    nwin = np.where(glen <= W, 1, (glen - W) // S + 1 + (((glen - W) // S + 1) * S < glen))
    lens = off[1:] - off[:-1]
    key_of = torch.repeat_interleave(torch.arange(nk, device=dev, dtype=torch.int32), lens)
    del lens
    t0 = time.time()
    total_bytes = 0
    for rk in range(P):
        lp, kp = [], []
        CH = 1 << 29                                        # (boolean-mask indexing of more than 2^31 elements is not supported)
        for c0 in range(0, nl, CH):
            lc = locs[c0:c0 + CH]
            sel = ((lc >> 32) % P) == rk
            lp.append(lc[sel]); kp.append(key_of[c0:c0 + CH][sel])
            del sel, lc
        l_r = torch.cat(lp); k_r = torch.cat(kp)
        del lp, kp
        kk, cnt = torch.unique_consecutive(k_r, return_counts=True)
        del k_r
        keys_r = keys[kk.long()].cpu().numpy().view(np.uint32)
        o = np.zeros(kk.numel() + 1, np.uint64); o[1:] = np.cumsum(cnt.cpu().numpy())
        locs_r = l_r.cpu().numpy().view(np.uint64)
        del l_r, kk, cnt
        torch.cuda.empty_cache()
        taxa = [dict(id=-(t + 1), parent=1000 + int(sp[t]), rank=0, name="genome_%d strain" % t, file="genomes/all.fna", index=t + 1,
                     windows=int(nwin[t]) if t % P == rk else 0) for t in range(n_targets - 1, -1, -1)]
        taxa.append(dict(id=1, parent=1, rank=20, name="root", file="", index=0, windows=0))
        taxa.append(dict(id=2, parent=1, rank=19, name="Bacteria", file="", index=0, windows=0))
        for s_ in sorted(set(int(x) for x in sp)):
            taxa.append(dict(id=1000 + s_, parent=2, rank=4, name="Synthetica species%d" % s_, file="", index=0, windows=0))
        path = os.path.join(a.workdir, "db.db_%d" % rk)
        host.write_shard(path, dict(k=16, sketch_size=16, winlen=W, winstride=S, q_k=16, q_sketch_size=16, q_winlen=W, q_winstride=S,
                                    max_locs_per_feature=254), taxa, n_targets, keys_r, o, locs_r)
        total_bytes += os.path.getsize(path)
        del keys_r, o, locs_r
    del key_of, locs, keys, off
    torch.cuda.empty_cache()
    res["shard_bytes"] = int(total_bytes)
    res["shard_files_written_s"] = round(time.time() - t0, 1)
    # the directly built handle, with the taxon keys of the files just written
    rdb = host.RefDb(os.path.join(a.workdir, "db"), P, meta_only=True)
    t2t = torch.from_numpy(rdb.tgt2tax(4).view(np.int32).copy()).to(dev)
    db_direct = eng.Database(None, None, None, None, device_ptrs=dict(keys=table_handle.keys_ptr, list_off=table_handle.list_off_ptr, locs=table_handle.locs_ptr,
                                                                      tgt2tax=t2t.data_ptr(), n_keys=nk, n_locs=nl, n_targets=n_targets))
    table_handle.close()

    # ---- the reads as two FASTA files; the direct handle's answer for them
    npairs = a.pairs
    rb = r.cpu().numpy().reshape(npairs, 2, 150)
    hdr = np.array([(">r%08d\n" % i).encode() for i in range(npairs)], dtype="S11")
    for mate in (0, 1):
        rec = np.empty((npairs, 11 + 150 + 1), dtype=np.uint8)
        rec[:, :11] = hdr.view(np.uint8).reshape(npairs, 11)
        rec[:, 11:161] = rb[:, mate, :]
        rec[:, -1] = ord("\n")
        rec.tofile(os.path.join(a.workdir, "reads_%d.fa" % (mate + 1)))
    ws = eng.Workspace(db_direct, npairs, r.numel())
    cands = torch.zeros((npairs, 2, 4), dtype=torch.int32, device=dev); ncand = torch.zeros(npairs, dtype=torch.int32, device=dev)
    ws.query_device(r.data_ptr(), ro.data_ptr(), 2 * npairs, True, cands.data_ptr(), ncand.data_ptr(), max_cand=2, emulate_ranks=P,
                    flags=eng.MCQ_QUIRK_SEQ_DROP)
    ws.sync()
    gc = cands.cpu().numpy().view(np.uint32); gn = ncand.cpu().numpy()
    ws.close(); db_direct.close()
    del cands, ncand
    torch.cuda.empty_cache()

    # ---- mcq_query_cli through the streaming route; its peak RSS and wall time (load + 65 536 pairs: the load dominates)
    opts = ["-lowest", "species", "-maxcand", "2", "-hitmin", "4", "-hitdiff", "80", "-tophits", "-taxids-only", "-omit-ranks"]
    env = dict(os.environ, MCQ_STREAM_LOAD_MIN_MB="0", MCQ_BUILD_TRACE="1")
    t0 = time.time()
    p = subprocess.Popen([pkg.cli_path(), "db", str(P), "reads_1.fa", "reads_2.fa", "-threads", "8", "-out", "ours.out"] + opts,
                         cwd=a.workdir, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    out, err = p.communicate()
    wall = time.time() - t0
    import resource
    ruc = resource.getrusage(resource.RUSAGE_CHILDREN)
    res["cli_streamed_wall_s"] = round(wall, 1)
    res["cli_streamed_max_rss_mb"] = round(ruc.ru_maxrss / 1024.0, 1)          # (KB on Linux; the largest child so far = this one)
    res["cli_streamed_gb_per_s"] = round(total_bytes / 1e9 / wall, 2)
    res["cli_rss_trace"] = [ln for ln in err.decode(errors="replace").split("\n") if ln.startswith("[mcq_open]")]
    if p.returncode != 0:
        res["error"] = "mcq_query_cli failed"
        print(json.dumps(res, indent=1)); sys.exit(1)

    # ---- the same lines from the direct handle's candidates (taxon keys -> ids, classify by the host library)
    hitdiff = float(np.float32(np.float32(80) * np.float32(0.01)))
    want = []
    for q in range(npairs):
        c = gc[q, :gn[q]]
        best = rdb.classify(c, 4, hitdiff, 19)
        want.append("r%08d\t|\t%s\t|\t%d\n" % (q, ",".join("%d:%d" % (rdb.taxon_id(int(t)), int(h)) for t, h in zip(c[:, 0], c[:, 1])),
                                             rdb.taxon_id(best) if best != 0xFFFFFFFF else 0))
    got = [ln for ln in open(os.path.join(a.workdir, "ours.out")) if not ln.startswith("#")]
    res["mapping_lines"] = [len(want), len(got)]
    res["identical_mapping_lines"] = want == got or sorted(want) == sorted(got)
    if not res["identical_mapping_lines"]:
        bad = [i for i, (x, y) in enumerate(zip(sorted(want), sorted(got))) if x != y][:3]
        res["first_differences"] = [[sorted(want)[i], sorted(got)[i]] for i in bad]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "stream_load_at_scale.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))
    if not a.keep:
        shutil.rmtree(a.workdir, ignore_errors=True)
    sys.exit(0 if res["identical_mapping_lines"] else 1)


if __name__ == "__main__":
    main()

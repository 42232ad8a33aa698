"""worker of tests/test_gpu_shard_native.py: one rank of the native sharded path (mcq_shard_*), several ranks on ONE
GPU, blocks exchanged through gloo (host-staged callback transport).  Checks its own batch against the oracle."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import mc_oracle as orc        # noqa: E402
from shard_exchange_gloo import make_gloo_exchange      # noqa: E402


def main():
    outp, paired, locs64 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    c2 = len(sys.argv) > 4 and sys.argv[4] == "c2"      # the bench table (BASELINE configs[1] / [3] / [4] shapes) instead of the small one
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rccl = os.environ.get("MCQ_TEST_TRANSPORT") == "rccl"        # one GPU per rank over real RCCL; else every rank on GPU 0 over gloo
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if rccl else 0)
    torch.cuda.set_device(dev)
    eng = importlib.import_module("metacache-mpi_amd.engine")
    synth = importlib.import_module("metacache-mpi_amd.synth")
    P, M = 4, 4
    if c2:
        gb, goff, species = synth.make_genomes(50, 10, 2_000_000, 6_000_000, 0.02, seed=3, device=dev)
    else:
        gb, goff, species = synth.make_genomes(5, 8, 150_000, 250_000, 0.02, seed=21, device=dev)
    table = eng.Table(gb.data_ptr(), goff.data_ptr(), goff.numel() - 1, emulate_ranks=P, device=dev.index)
    keys, off, locs, _ = table.to_host()
    sp32 = species.to(torch.int32).contiguous()
    db = eng.Database(None, None, None, None, n_shards=world, shard_id=rank, device=dev.index,
                      flags=(0, eng.MCQ_DB_LOCS_64, eng.MCQ_DB_LOCS_GW | eng.MCQ_DB_SLOTS_16)[locs64],   # bit fields 32 / 64, global window
                      device_ptrs=dict(keys=table.keys_ptr, list_off=table.list_off_ptr, locs=table.locs_ptr, tgt2tax=sp32.data_ptr(),
                                       n_keys=table.n_keys, n_locs=table.n_locs, n_targets=sp32.numel()))
    table.close()
    odb = orc.OracleDb(keys, off, locs, species.cpu().numpy().astype(np.uint32))

    def batch(seed):
        n, L = 20000, 150
        reads, roff, _ = synth.sample_reads(gb, goff, n, L, 0.01, 0.002, seed=seed)
        # ragged tail: wide reads (second wave stage) and long ones (workgroup kernel), an empty and a tiny one
        wr, wroff, _ = synth.sample_reads(gb, goff, 64, 500, 0.01, 0.0, seed=seed + 1)
        lr, lroff, _ = synth.sample_reads(gb, goff, 8, 5000, 0.03, 0.0, seed=seed + 2)
        tiny = torch.tensor(list(b"ACGTACGTAC"), dtype=torch.uint8, device=dev)
        reads = torch.cat([reads, wr, lr, tiny])
        roff = torch.cat([roff, wroff[1:] + roff[-1], lroff[1:] + roff[-1] + wroff[-1],
                          (roff[-1] + wroff[-1] + lroff[-1]).reshape(1) + torch.tensor([0, 10], device=dev)])
        return reads.contiguous(), roff.contiguous(), n + 64 + 8 + 2

    if c2:
        c2_run(eng, synth, orc, db, odb, gb, goff, dev, rank, world, P, M, outp)
        dist.barrier()
        dist.destroy_process_group()
        return
    if len(sys.argv) > 4 and sys.argv[4] == "mismatch":
        # shards whose location words mean different things (rank 0: global-window words, the others: bit fields): the ranks compare
        # a signature of the format before the first word travels and every one of them fails with MCQ_E_ARG -- nobody answers wrongly
        db.close()
        table = eng.Table(gb.data_ptr(), goff.data_ptr(), goff.numel() - 1, emulate_ranks=P, device=dev.index)
        db = eng.Database(None, None, None, None, n_shards=world, shard_id=rank, device=dev.index,
                          flags=(eng.MCQ_DB_LOCS_GW | eng.MCQ_DB_SLOTS_16) if rank == 0 else 0,
                          device_ptrs=dict(keys=table.keys_ptr, list_off=table.list_off_ptr, locs=table.locs_ptr, tgt2tax=sp32.data_ptr(),
                                           n_keys=table.n_keys, n_locs=table.n_locs, n_targets=sp32.numel()))
        table.close()
        r, ro, _ = synth.sample_reads(gb, goff, 2000, 150, 0.01, 0.002, seed=5)
        sh = eng.Shard(db, world, rank, max_queries=2000, max_bases=r.numel(), max_seqs=2000)
        sh.set_exchange(make_gloo_exchange())
        cands = torch.zeros((2000, M, 4), dtype=torch.int32, device=dev); ncand = torch.zeros(2000, dtype=torch.int32, device=dev)
        code, text = 0, ""
        try:
            sh.query(r.data_ptr(), ro.data_ptr(), 2000, False, cands.data_ptr(), ncand.data_ptr(), max_cand=M, emulate_ranks=P,
                     stream=torch.cuda.current_stream(dev).cuda_stream)
        except eng.McqError as e:
            code, text = e.code, str(e)
        np.savez(outp + ".%d.npz" % rank, code=np.array([code]), told=np.array(["location words" in text]))
        dist.barrier()
        sh.close()
        dist.destroy_process_group()
        return
    b0, b1 = batch(500 + 10 * rank), batch(900 + 10 * rank)
    n_seqs = b0[2]
    nq = n_seqs // 2 if paired else n_seqs
    sh = eng.Shard(db, world, rank, max_queries=nq, max_bases=max(b0[0].numel(), b1[0].numel()), max_seqs=n_seqs)
    if rccl:
        box = [eng.Shard.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        sh.comm_rccl(box[0])
    else:
        sh.set_exchange(make_gloo_exchange())
    st = torch.cuda.current_stream(dev).cuda_stream
    res = []
    ok = True
    # batch 0 in the exact mode (first batch of a context, learns the block sizes), announcing batch 1; batch 1 and then
    # batch 0 again in the padded mode, the last one with the raw-sort hook
    plan = [(b0, 0, b1), (b1, 0, b0), (b0, eng.MCQ_FORCE_RAW_SORT, None)]
    for (bb, qf, nxt) in plan:
        cands = torch.zeros((nq, M, 4), dtype=torch.int32, device=dev)
        ncand = torch.zeros(nq, dtype=torch.int32, device=dev)
        sh.query(bb[0].data_ptr(), bb[1].data_ptr(), n_seqs, bool(paired), cands.data_ptr(), ncand.data_ptr(), max_cand=M,
                 emulate_ranks=P, flags=qf, stream=st,
                 next_batch=None if nxt is None else (nxt[0].data_ptr(), nxt[1].data_ptr(), n_seqs))
        stats = sh.sync(st)
        res.append(stats)
        oc, on = odb.query(bb[0].cpu().numpy().tobytes(), bb[1].cpu().numpy().astype(np.uint64), bool(paired), max_cand=M,
                           emulate_ranks=P, threads=4)
        gc = cands.cpu().numpy().view(np.uint32); gn = ncand.cpu().numpy().view(np.uint32)
        good = bool(np.array_equal(gn, on))
        if good:
            mask = np.arange(M)[None, :] < on[:, None]
            good = bool(np.array_equal(gc[mask], oc[mask]))
        ok = ok and good
    caps = sh.caps()
    np.savez(outp + ".%d.npz" % rank, ok=np.array([ok]), overflow=np.array([r["n_overflow"] for r in res]),
             feats=np.array([r["n_features"] for r in res]), locs=np.array([r["n_locations"] for r in res]), caps=np.array(caps), nq=np.array([nq]))
    dist.barrier()
    sh.close()
    dist.destroy_process_group()


def c2_run(eng, synth, orc, db, odb, gb, goff, dev, rank, world, P, M, outp):
    """the bench table over `world` ranks: a batch of 2 x 150 bp pairs (configs[3] shape) and a batch of ONT-like reads, mean
    8 kb (configs[4] shape), each rank its own reads, against the oracle on the whole table"""
    pr, pro, _ = synth.sample_pairs(gb, goff, 60_000, 150, 300, 500, 0.005, 0.001, seed=4000 + rank)
    lr, lro, _ = synth.sample_long_reads(gb, goff, 1500, 8000, 0.08, seed=5000 + rank)
    sh = eng.Shard(db, world, rank, max_queries=60_000, max_bases=max(pr.numel(), lr.numel()), max_seqs=120_000, max_locs_per_query=1 << 16)
    sh.set_exchange(make_gloo_exchange())
    st = torch.cuda.current_stream(dev).cuda_stream
    ok, res = True, []
    for (bb, bo, n_seqs, paired) in ((pr, pro, 120_000, True), (lr, lro, 1500, False), (pr, pro, 120_000, True)):
        nq = n_seqs // 2 if paired else n_seqs
        cands = torch.zeros((nq, M, 4), dtype=torch.int32, device=dev)
        ncand = torch.zeros(nq, dtype=torch.int32, device=dev)
        sh.query(bb.data_ptr(), bo.data_ptr(), n_seqs, paired, cands.data_ptr(), ncand.data_ptr(), max_cand=M, emulate_ranks=P, stream=st,
                 exact=not paired)       # (the block sizes learned from the pairs do not fit the long reads: exact sizes for them)
        res.append(sh.sync(st))
        oc, on = odb.query(bb.cpu().numpy().tobytes(), bo.cpu().numpy().astype(np.uint64), paired, max_cand=M, emulate_ranks=P, threads=4)
        gc = cands.cpu().numpy().view(np.uint32); gn = ncand.cpu().numpy().view(np.uint32)
        good = bool(np.array_equal(gn, on))
        if good:
            mask = np.arange(M)[None, :] < on[:, None]
            good = bool(np.array_equal(gc[mask], oc[mask]))
        ok = ok and good
    np.savez(outp + ".%d.npz" % rank, ok=np.array([ok]), overflow=np.array([r["n_overflow"] for r in res]),
             feats=np.array([r["n_features"] for r in res]), locs=np.array([r["n_locations"] for r in res]), caps=np.array(sh.caps()), nq=np.array([60_000]))
    sh.close()


if __name__ == "__main__":
    main()

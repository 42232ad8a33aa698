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
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    eng = importlib.import_module("metacache-mpi_amd.engine")
    synth = importlib.import_module("metacache-mpi_amd.synth")
    P, M = 4, 4
    gb, goff, species = synth.make_genomes(5, 8, 150_000, 250_000, 0.02, seed=21, device=dev)
    table = eng.Table(gb.data_ptr(), goff.data_ptr(), goff.numel() - 1, emulate_ranks=P)
    keys, off, locs, _ = table.to_host()
    sp32 = species.to(torch.int32).contiguous()
    db = eng.Database(None, None, None, None, n_shards=world, shard_id=rank,
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

    b0, b1 = batch(500 + 10 * rank), batch(900 + 10 * rank)
    n_seqs = b0[2]
    nq = n_seqs // 2 if paired else n_seqs
    sh = eng.Shard(db, world, rank, max_queries=nq, max_bases=max(b0[0].numel(), b1[0].numel()), max_seqs=n_seqs)
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


if __name__ == "__main__":
    main()

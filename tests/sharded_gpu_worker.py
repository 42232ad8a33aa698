"""Worker of tests/test_gpu_sharded.py: 2 ranks sharing ONE GPU, gloo for the exchange
(host-staged), the real HIP stage kernels on each rank's hash-range shard."""
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


def main():
    outp, paired = sys.argv[1], int(sys.argv[2])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    eng = importlib.import_module("metacache-mpi_amd.engine")
    dbbuild = importlib.import_module("dbbuild_torch")
    synth = importlib.import_module("metacache-mpi_amd.synth")
    sharded = importlib.import_module("sharded_staged")
    P, M = 4, 4
    gb, goff, species = synth.make_genomes(5, 8, 150_000, 250_000, 0.02, seed=21, device=dev)
    keys, off, locs, _ = dbbuild.build_table(gb, goff, emulate_ranks=P)
    db = dbbuild.make_database(keys, off, locs, species, n_shards=world, shard_id=rank)
    n, L = 20000, 150
    reads, roff, _ = synth.sample_reads(gb, goff, n, L, 0.01, 0.002, seed=500 + rank)
    # a few long reads to exercise the reduce overflow path
    lr, lroff, _ = synth.sample_reads(gb, goff, 8, 5000, 0.03, 0.0, seed=700 + rank)
    reads = torch.cat([reads, lr]); roff = torch.cat([roff, lroff[1:] + roff[-1]])
    n_seqs = n + 8
    nq = n_seqs // 2 if paired else n_seqs
    sq = sharded.ShardedQuery(db, world, rank, dev, max_queries=nq)
    cands = torch.zeros((nq, M, 4), dtype=torch.int32, device=dev)
    ncand = torch.zeros(nq, dtype=torch.int32, device=dev)
    # window count of the batch, so that the next batch's sketch can be announced (second-stream prefetch path)
    win_off = torch.empty(n_seqs + 1, dtype=torch.int64, device=dev)
    db.count_windows(reads.data_ptr(), roff.data_ptr(), n_seqs, win_off.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    hint = int(win_off[-1].item())
    sq.query(reads, roff, n_seqs, bool(paired), cands, ncand, max_cand=M, emulate_ranks=P, n_win_hint=hint,
             next_batch=(reads, roff, n_seqs))
    assert sq._prepared is not None                      # the same batch again: its sketch is already under way
    st = sq.last_stats()
    torch.cuda.synchronize()
    odb = orc.OracleDb(keys.cpu().numpy().astype(np.uint32), off.cpu().numpy().astype(np.uint64),
                       locs.cpu().numpy().astype(np.uint64), species.cpu().numpy().astype(np.uint32))
    oc, on = odb.query(reads.cpu().numpy().tobytes(), roff.cpu().numpy().astype(np.uint64), bool(paired), max_cand=M,
                       emulate_ranks=P, threads=4)
    gc = cands.cpu().numpy().view(np.uint32); gn = ncand.cpu().numpy().view(np.uint32)
    ok = bool(np.array_equal(gn, on))
    if ok:
        mask = np.arange(M)[None, :] < on[:, None]
        ok = bool(np.array_equal(gc[mask], oc[mask]))
    # the reduce kernel without its de-duplicating pass must agree as well
    cands2 = torch.zeros_like(cands); ncand2 = torch.zeros_like(ncand)
    sq.query(reads, roff, n_seqs, bool(paired), cands2, ncand2, max_cand=M, emulate_ranks=P, flags=eng.MCQ_FORCE_RAW_SORT,
             n_win_hint=hint)                            # consumes the prepared sketch + buckets
    assert sq._prepared is None
    torch.cuda.synchronize()
    if ok:
        g2 = cands2.cpu().numpy().view(np.uint32); n2 = ncand2.cpu().numpy().view(np.uint32)
        ok = bool(np.array_equal(n2, on)) and bool(np.array_equal(g2[mask], oc[mask]))
    np.savez(outp + ".%d.npz" % rank, ok=np.array([ok]), served=np.array([st["n_features_served"]]),
             overflow=np.array([st["n_overflow"]]), nq=np.array([nq]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

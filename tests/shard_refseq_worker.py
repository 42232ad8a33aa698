"""worker of tests/test_gpu_refseq_scale.py::test_config2_through_the_sharded_path_two_ranks: BASELINE configs[2]'s table
(RefSeq scale: >= 100 Gbp, 52 001 targets, global-window words) hash-range-sharded over TWO ranks that share the one GPU of the
box, the blocks of mcq_shard_* exchanged through gloo (RCCL refuses two ranks on one device).  Every rank queries its own
reads -- a batch of 150 bp reads, a batch of 2 x 150 bp pairs -- and checks them against the CPU oracle on the part of the
table its reads can touch, read back from BOTH shards (oracle/subtable.py)."""
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
from oracle import subtable                # noqa: E402
from shard_exchange_gloo import make_gloo_exchange      # noqa: E402


def main():
    outp = sys.argv[1]
    n_species = int(sys.argv[2]) if len(sys.argv) > 2 else 2600
    n = int(sys.argv[3]) if len(sys.argv) > 3 else (1 << 16)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    eng = importlib.import_module("metacache-mpi_amd.engine")
    synth = importlib.import_module("metacache-mpi_amd.synth")
    P, M = 2, 2
    db = None
    sets = {}
    # the ranks take turns: the 100 Gbp of sequences exist once at a time on the GPU they share
    for turn in range(world):
        if turn == rank:
            gb, goff, species = synth.make_genomes_big(n_species, 10, 2_000_000, 6_000_000, 0.02, seed=3, device=dev, extra_genome=16_000_000)
            goff, species = synth.split_targets(goff, species, 2, keep_last_whole=True)
            tw = synth.window_counts(goff)
            for x in range(world):          # the read sets of every rank (this rank reads the sub-tables of all of them from its shard)
                r, ro, _ = synth.sample_reads(gb, goff, n, 150, 0.005, 0.001, seed=3000 + x)
                p, po, _ = synth.sample_pairs(gb, goff, n // 2, 150, 300, 500, 0.005, 0.001, seed=4000 + x)
                sets[x] = ((r, ro, n, False), (p, po, n, True))
            parts = eng.Parts(gb.data_ptr(), goff.data_ptr(), species.numel(), emulate_ranks=P, n_shards=world, shard_id=rank,
                              flags=eng.MCQ_BUILD_REMOVE_OVERPOPULATED)
            del gb
            torch.cuda.empty_cache()
            sp32 = species.to(torch.int32).contiguous()
            db = parts.database(sp32.data_ptr(), n_shards=world, shard_id=rank)
            parts.close()
            torch.cuda.synchronize(dev)
        dist.barrier()
    lay = db.layout()
    assert lay["loc_format"] == eng.MCQ_LOC_GLOBAL_WINDOW and lay["loc_bytes"] == 4, lay
    sp = species.cpu().numpy().astype(np.uint32)
    for x in range(world):
        for kind, (rd, ro_t, n_seqs, paired) in enumerate(sets[x]):
            k_, o_, l_ = subtable.batch_subtable(eng, db, rd.data_ptr(), ro_t.data_ptr(), n_seqs, dev, tw)
            np.savez(outp + ".sub.%d.%d.%d.npz" % (x, kind, rank), k=k_, o=o_, l=l_)
    dist.barrier()
    sh = eng.Shard(db, world, rank, max_queries=n, max_bases=max(sets[rank][0][0].numel(), sets[rank][1][0].numel()), max_seqs=n)
    sh.set_exchange(make_gloo_exchange())
    st = torch.cuda.current_stream(dev).cuda_stream
    ok, res = True, []
    for kind, (rd, ro_t, n_seqs, paired) in enumerate(sets[rank]):
        nq = n_seqs // 2 if paired else n_seqs
        tabs = []
        for r2 in range(world):
            z = np.load(outp + ".sub.%d.%d.%d.npz" % (rank, kind, r2))
            tabs.append((z["k"], z["o"], z["l"]))
        odb = orc.OracleDb(*subtable.merge_subtables(tabs), sp)
        oc, on = odb.query(rd.cpu().numpy().tobytes(), ro_t.cpu().numpy().astype(np.uint64), paired, max_cand=M, emulate_ranks=P, threads=8)
        for rep in range(2):        # exact sizes (the blocks are sized by what the table serves), then the padded mode
            cands = torch.zeros((nq, M, 4), dtype=torch.int32, device=dev); ncand = torch.zeros(nq, dtype=torch.int32, device=dev)
            sh.query(rd.data_ptr(), ro_t.data_ptr(), n_seqs, paired, cands.data_ptr(), ncand.data_ptr(), max_cand=M, emulate_ranks=P, stream=st,
                     exact=(rep == 0))
            stats = sh.sync(st)
            gc = cands.cpu().numpy().view(np.uint32); gn = ncand.cpu().numpy().view(np.uint32)
            good = bool(np.array_equal(gn, on))
            if good:
                mask = np.arange(M)[None, :] < on[:, None]
                good = bool(np.array_equal(gc[mask], oc[mask]))
            ok = ok and good
            res.append((nq, stats["n_locations"], stats["n_two_class"], stats["n_overflow"]))
    xb = sh.exchange_bytes()
    np.savez(outp + ".%d.npz" % rank, ok=np.array([ok]), res=np.array(res, dtype=np.int64), caps=np.array(sh.caps()),
             xb=np.array([xb[k] for k in ("batches", "x1", "x2_ends", "x2_locations", "own_blocks")], dtype=np.int64))
    dist.barrier()
    sh.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

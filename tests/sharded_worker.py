"""Worker of tests/test_sharded_gloo.py: runs tests/sharded_staged.py's routing on CPU
tensors over gloo, with the stage functions supplied by the oracle (test infrastructure).
Launched by torch.distributed.run; writes its ranks' results to <out>.<rank>.npz."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import Fixture            # noqa: E402
from oracle import dbfile                  # noqa: E402
from oracle import mc_oracle as orc        # noqa: E402


class OracleBackend:
    """Same interface as sharded.HipBackend, on CPU tensors, computed by the oracle."""

    def __init__(self, keys, off, locs, t2t, p, n_shards, shard_id, eng):
        self.s = p["qs"]; self.p = p; self.eng = eng
        self.odb = orc.OracleDb(keys, off, locs, t2t, k=p["qk"], s=p["qs"], winlen=p["qwinlen"],
                                winstride=p["qwinstride"], tgt_winstride=p["winstride"])
        self.shard = {int(k): locs[int(off[i]):int(off[i + 1])] for i, k in enumerate(keys)
                      if eng.owner(int(k), n_shards) == shard_id}

    def sketch(self, bases, seq_off, n_seqs, n_win_hint=None):
        b = bases.numpy().tobytes(); so = seq_off.numpy()
        rows, win_off = [], [0]
        for i in range(n_seqs):
            seq = b[so[i]:so[i + 1]]
            ws = orc.windows(len(seq), self.p["qwinlen"], self.p["qwinstride"])
            for (x, y) in ws:
                sk = orc.sketch(seq[x:y], self.p["qk"], self.s)
                rows.append(np.concatenate([sk.astype(np.int64), np.full(self.s - len(sk), 0xFFFFFFFF, np.int64)]))
            win_off.append(win_off[-1] + len(ws))
        f = np.array(rows, dtype=np.int64).reshape(-1, self.s)
        f = np.where(f >= (1 << 31), f - (1 << 32), f).astype(np.int32)
        return torch.tensor(win_off, dtype=torch.int64), torch.from_numpy(f)

    def bucket(self, feats_flat, n_shards):
        f = feats_flat.numpy().astype(np.int64) & 0xFFFFFFFF
        idx = np.nonzero(f != 0xFFFFFFFF)[0]
        own = np.array([self.eng.owner(int(x), n_shards) for x in f[idx]], dtype=np.int64)
        order = np.argsort(own, kind="stable")
        counts = np.bincount(own, minlength=n_shards).tolist()
        return counts, feats_flat[idx[order]].contiguous(), torch.from_numpy(idx[order].astype(np.int32))

    def lookup(self, feats):
        f = feats.numpy().astype(np.int64) & 0xFFFFFFFF
        lens = np.array([len(self.shard.get(int(x), ())) for x in f], dtype=np.int32)
        off = np.zeros(len(f) + 1, np.int64); off[1:] = np.cumsum(lens)
        return torch.from_numpy(lens), torch.from_numpy(off)

    def gather(self, feats, off, total):
        f = feats.numpy().astype(np.int64) & 0xFFFFFFFF
        parts = [self.shard[int(x)] for x in f if int(x) in self.shard]
        out = np.concatenate(parts) if parts else np.zeros(0, np.uint64)
        assert len(out) == total
        return torch.from_numpy(out.astype(np.int64))

    def assemble(self, m, lens_back, src_idx, n_slots, locs_back, total, bases, seq_off, n_seqs, paired, win_off):
        lens = lens_back.numpy().astype(np.int64); slot = src_idx.numpy().astype(np.int64)
        slot_len = np.zeros(n_slots + 1, np.int64); slot_len[slot] = lens
        dst_off = np.zeros(n_slots + 1, np.int64); dst_off[1:] = np.cumsum(slot_len[:n_slots])
        src_off = np.zeros(m + 1, np.int64); src_off[1:] = np.cumsum(lens)
        dst = torch.zeros(max(total, 1), dtype=torch.int64)
        for i in range(m):
            dst[dst_off[slot[i]]:dst_off[slot[i]] + lens[i]] = locs_back[src_off[i]:src_off[i + 1]]
        qstep = 2 if paired else 1
        nq = n_seqs // qstep
        wo = win_off.numpy(); so = seq_off.numpy()
        loc_off = torch.from_numpy(dst_off[wo[0:n_seqs + 1:qstep][:nq + 1] * self.s].copy())
        ql = so[0:n_seqs + 1:qstep][:nq + 1]
        return loc_off, torch.from_numpy((ql[1:] - ql[:-1]).astype(np.int32)), dst

    def reduce(self, nq, loc_off, locs, query_len, cands, ncand, max_cand, emulate_ranks, insert_size_max, flags):
        lo = loc_off.numpy(); l = locs.numpy().astype(np.uint64); ql = query_len.numpy()
        for q in range(nq):
            out, n = self.odb.reduce_query(l[lo[q]:lo[q + 1]], int(ql[q]), max_cand, emulate_ranks, insert_size_max,
                                           1 if flags & self.eng.MCQ_QUIRK_SEQ_DROP else 0)
            cands[q] = torch.from_numpy(out.astype(np.int64)).to(cands.dtype)
            ncand[q] = n


def main():
    tag, P, outp = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    eng = importlib.import_module("metacache-mpi_amd.engine")
    sharded = importlib.import_module("sharded_staged")
    fx = Fixture(tag, P)
    keys, off, locs = dbfile.union_shards(fx.shards)
    be = OracleBackend(keys, off, locs, fx.tgt2tax(), fx.params, world, rank, eng)
    # this rank's share of the queries (pairs)
    nq = len(fx.names)
    mine = list(range(rank * nq // world, (rank + 1) * nq // world))
    seqs = []
    for q in mine:
        seqs += [fx.r1[q], fx.r2[q]]
    bases, so = orc.pack_reads(seqs)
    tb = torch.from_numpy(np.frombuffer(bases, dtype=np.uint8).copy()) if bases else torch.zeros(0, dtype=torch.uint8)
    sq = sharded.ShardedQuery(None, world, rank, torch.device("cpu"), max_queries=len(mine), backend=be)
    cands = torch.zeros((max(len(mine), 1), fx.maxcand, 4), dtype=torch.int64)
    ncand = torch.zeros(max(len(mine), 1), dtype=torch.int64)
    sq.query(tb, torch.from_numpy(so.astype(np.int64)), len(seqs), True, cands, ncand, max_cand=fx.maxcand,
             emulate_ranks=P, flags=eng.MCQ_QUIRK_SEQ_DROP)
    np.savez(outp + ".%d.npz" % rank, q=np.array(mine), cands=cands.numpy()[:len(mine)], ncand=ncand.numpy()[:len(mine)],
             served=np.array([sq.last_stats()["n_features_served"]]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

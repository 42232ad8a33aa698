"""TEST CODE (moved out of the product package in round 3): the round-1 form of the feature-sharded query path, a
torch.distributed loop over the engine's STAGED entry points.  The product path at N > 1 is mcq_shard_* behind the C ABI
(csrc/mcq_shard.hpp); this loop is kept as an independent second implementation of the routing for the tests
(tests/test_sharded_gloo.py on CPU with an oracle backend, tests/test_gpu_sharded.py with the staged kernels).

Feature-sharded multi-GPU query path over torch.distributed.

One process per GPU.  The feature -> locations table is partitioned by hash range of
h2(feature) (engine.owner); reads are split across ranks.  Per batch, every rank

  1. sketches its own reads                      (mcq_count_windows + mcq_sketch)
  2. buckets the features by owning shard        (mcq_bucket_features)
  3. all-to-all: counts, then features           -> owners
  4. owner looks the features up in its shard    (mcq_lookup_count + mcq_lookup_gather)
  5. all-to-all: list lengths, then locations    -> home ranks
  6. puts the lists into per-query order         (mcq_assemble)
  7. sorts / sweeps / folds per query            (mcq_reduce)

All locations of a read reach its home rank, so per-target hit counts equal the
reference's (which computes them on the single MPI rank owning the target,
src/sketch_database.h:540); the only cross-shard semantic left, the reference's tree-merge
order (src/querying.h:867-1073), is emulated inside mcq_reduce with virtual rank = tgt % P.

The collective layer is torch.distributed (backend "nccl" = RCCL over xGMI on the GPU
node).  The stage functions come from a backend object: HipBackend below is the product;
the CPU gloo test of the routing logic injects its own (tests/test_sharded_gloo.py).
"""
import torch
import torch.distributed as dist

import importlib

engine = importlib.import_module("metacache-mpi_amd.engine")

EMPTY = -1          # 0xFFFFFFFF as int32


class HipBackend:
    """Stage functions on CUDA tensors through the C ABI (no CPU fallback)."""

    def __init__(self, db, dev, max_queries, max_locs_per_query=0):
        self.db, self.dev = db, dev
        self.s = db.sketch_size
        self.loc_dtype = torch.int32 if db.loc_bytes() == 4 else torch.int64
        self.ws = engine.Workspace(db, max_queries, 1, max_locs_per_query)

    def _st(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    def sketch(self, bases, seq_off, n_seqs, n_win_hint=None):
        win_off = torch.empty(n_seqs + 1, dtype=torch.int64, device=self.dev)
        self.db.count_windows(bases.data_ptr(), seq_off.data_ptr(), n_seqs, win_off.data_ptr(), self._st())
        # n_win_hint (no host wait) is only right for fixed-length reads: the buffers are sized by a bound that holds
        # for any batch (a sequence of n bases has at most n // stride + 2 windows), the hint only trims the view
        bound = bases.numel() // self.db.winstride + 2 * n_seqs
        n_win = int(win_off[-1].item()) if n_win_hint is None else n_win_hint
        if n_win > bound:
            raise ValueError("n_win_hint %d exceeds what %d bases in %d sequences can have" % (n_win, bases.numel(), n_seqs))
        feats = torch.empty((max(bound, 1), self.s), dtype=torch.int32, device=self.dev)
        nfeat = torch.empty(max(bound, 1), dtype=torch.int32, device=self.dev)
        self.db.sketch(bases.data_ptr(), seq_off.data_ptr(), n_seqs, win_off.data_ptr(), feats.data_ptr(),
                       nfeat.data_ptr(), self._st())
        return win_off, feats[:n_win]

    def bucket(self, feats_flat, n_shards):
        n = feats_flat.numel()
        counts = torch.empty(2 * n_shards, dtype=torch.int64, device=self.dev)
        bucketed = torch.empty(max(n, 1), dtype=torch.int32, device=self.dev)
        src = torch.empty(max(n, 1), dtype=torch.int32, device=self.dev)
        engine.bucket_features(feats_flat.data_ptr(), n, n_shards, counts.data_ptr(), bucketed.data_ptr(), src.data_ptr(), self._st())
        c = counts[:n_shards].cpu().tolist()
        m = sum(c)
        return c, bucketed[:m], src[:m]

    def lookup(self, feats):
        """-> (list lengths int32 [n], exclusive offsets int64 [n+1]); keeps the list starts for gather()"""
        n = feats.numel()
        lens = torch.zeros(max(n, 1), dtype=torch.int32, device=self.dev)
        self._src = torch.empty(max(n, 1), dtype=torch.int64, device=self.dev)
        self.db.lookup_count(feats.data_ptr(), n, lens.data_ptr(), self._src.data_ptr(), self._st())
        off = torch.zeros(n + 1, dtype=torch.int64, device=self.dev)
        torch.cumsum(lens[:n], 0, dtype=torch.int64, out=off[1:])
        self._lens = lens
        return lens[:n], off

    def gather(self, feats, off, total):
        """locations in the shard's native width: int32 words when compact, else int64"""
        locs = torch.empty(max(total, 1), dtype=self.loc_dtype, device=self.dev)
        self.db.lookup_gather(feats.data_ptr(), feats.numel(), off.data_ptr(), locs.data_ptr(), self._lens.data_ptr(),
                              self._src.data_ptr(), self._st())
        return locs[:total]

    def assemble(self, m, lens_back, src_idx, n_slots, locs_back, total, bases, seq_off, n_seqs, paired, win_off):
        nq = n_seqs // 2 if paired else n_seqs
        loc_off = torch.empty(nq + 1, dtype=torch.int64, device=self.dev)
        query_len = torch.empty(max(nq, 1), dtype=torch.int32, device=self.dev)
        dst = torch.empty(max(total, 1), dtype=self.loc_dtype, device=self.dev)
        self.db.assemble(m, lens_back.data_ptr(), src_idx.data_ptr(), n_slots, locs_back.data_ptr(), bases.data_ptr(),
                         seq_off.data_ptr(), n_seqs, paired, win_off.data_ptr(), loc_off.data_ptr(), query_len.data_ptr(),
                         dst.data_ptr(), self._st())
        return loc_off, query_len, dst

    def reduce(self, nq, loc_off, locs, query_len, cands, ncand, max_cand, emulate_ranks, insert_size_max, flags):
        self.ws.reduce_device(nq, loc_off.data_ptr(), locs.data_ptr(), query_len.data_ptr(), cands.data_ptr(),
                              ncand.data_ptr(), max_cand=max_cand, emulate_ranks=emulate_ranks,
                              insert_size_max=insert_size_max, flags=flags, stream=self._st())

    def stats(self):
        return self.ws.sync(self._st())

    def prepare(self, bases, seq_off, n_seqs, n_shards, n_win_hint, stream):
        """Steps 1-2 (sketch + bucket) enqueued on `stream` with no host wait: the per-shard counts go to
        pinned memory behind an event.  Lets the next batch's sketching run under this batch's exchange."""
        with torch.cuda.stream(stream):
            win_off, feats = self.sketch(bases, seq_off, n_seqs, n_win_hint)
            flat = feats.reshape(-1)
            n = flat.numel()
            counts = torch.empty(2 * n_shards, dtype=torch.int64, device=self.dev)
            bucketed = torch.empty(max(n, 1), dtype=torch.int32, device=self.dev)
            src = torch.empty(max(n, 1), dtype=torch.int32, device=self.dev)
            engine.bucket_features(flat.data_ptr(), n, n_shards, counts.data_ptr(), bucketed.data_ptr(), src.data_ptr(), self._st())
            host_counts = torch.empty(n_shards, dtype=torch.int64, pin_memory=True)
            host_counts.copy_(counts[:n_shards], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
        return {"win_off": win_off, "F": n, "bucketed": bucketed, "src": src, "host_counts": host_counts, "event": ev,
                "keep": (feats, counts), "key": (bases.data_ptr(), seq_off.data_ptr(), n_seqs)}


def _a2a(out_numel, inp, send_counts, recv_counts, group):
    """all_to_all_single with explicit split sizes.  On the GPU node the backend is nccl (RCCL)
    and the tensors stay in HBM; under gloo (tests on a box without a second GPU) device
    tensors are staged through the host."""
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        out = torch.empty(out_numel, dtype=inp.dtype)
        dist.all_to_all_single(out, inp.contiguous().cpu(), list(recv_counts), list(send_counts), group=group)
        return out.to(inp.device)
    out = torch.empty(out_numel, dtype=inp.dtype, device=inp.device)
    dist.all_to_all_single(out, inp.contiguous(), list(recv_counts), list(send_counts), group=group)
    return out


class ShardedQuery:
    def __init__(self, db, world, rank, dev, max_queries, max_bases=0, read_len_hint=None, backend=None, group=None):
        self.world, self.rank, self.dev, self.group = world, rank, dev, group
        self.be = backend if backend is not None else HipBackend(db, dev, max_queries)
        self.s = self.be.s
        self._last = {}
        self._prepared = None
        self._side = torch.cuda.Stream(device=dev) if (backend is None and torch.cuda.is_available()) else None

    def query(self, bases, seq_off, n_seqs, paired, cands, ncand, max_cand=2, emulate_ranks=1, insert_size_max=0,
              flags=0, n_win_hint=None, next_batch=None):
        """next_batch = (bases, seq_off, n_seqs) of the following call (needs n_win_hint): its sketching and
        bucketing are enqueued on a second stream now and run under this batch's exchange."""
        N, be, s = self.world, self.be, self.s
        nq = n_seqs // 2 if paired else n_seqs
        # 1-2. sketch, bucket by owner (already under way if the previous call was told about this batch)
        prep, self._prepared = self._prepared, None
        if prep is not None and prep["key"] != (bases.data_ptr(), seq_off.data_ptr(), n_seqs):
            prep["event"].synchronize()
            prep = None
        if prep is not None:
            main = torch.cuda.current_stream(self.dev)
            prep["event"].synchronize()                    # host: the counts are in pinned memory
            main.wait_event(prep["event"])
            send_counts = prep["host_counts"].tolist()
            m = sum(send_counts)
            win_off, F = prep["win_off"], prep["F"]
            bucketed, src_idx = prep["bucketed"][:m], prep["src"][:m]
            for t in (win_off, prep["bucketed"], prep["src"]) + tuple(prep["keep"]):
                t.record_stream(main)
        else:
            win_off, feats = be.sketch(bases, seq_off, n_seqs, n_win_hint)
            F = feats.numel()
            send_counts, bucketed, src_idx = be.bucket(feats.reshape(-1), N)
            m = bucketed.numel()
        if next_batch is not None and self._side is not None and n_win_hint is not None and hasattr(be, "prepare"):
            self._side.wait_stream(torch.cuda.current_stream(self.dev))     # the next batch's inputs are ready
            self._prepared = be.prepare(next_batch[0], next_batch[1], next_batch[2], N, n_win_hint, self._side)
        # 3. counts, then features to their owners
        sc = torch.tensor(send_counts, dtype=torch.int64, device=bucketed.device)
        rc = _a2a(N, sc, [1] * N, [1] * N, self.group)
        recv_counts = rc.cpu().tolist()
        R = sum(recv_counts)
        recv_feats = _a2a(R, bucketed, send_counts, recv_counts, self.group)
        # 4. owner lookup
        lens_r, off_r = be.lookup(recv_feats)
        # 5. lengths back, then locations back (split sizes = per-peer sums of lengths)
        lens_back = _a2a(m, lens_r, recv_counts, send_counts, self.group)
        src_off = torch.zeros(m + 1, dtype=torch.int64, device=bucketed.device)
        torch.cumsum(lens_back, 0, dtype=torch.int64, out=src_off[1:])
        rb = torch.tensor([0] + recv_counts, dtype=torch.int64).cumsum(0).to(off_r.device)
        sb = torch.tensor([0] + send_counts, dtype=torch.int64).cumsum(0).to(off_r.device)
        sizes = torch.cat([off_r[rb], src_off[sb]]).cpu()
        send_loc = (sizes[1:N + 1] - sizes[:N]).tolist()
        recv_loc = (sizes[N + 2:] - sizes[N + 1:-1]).tolist()
        total_r, total_b = int(sizes[N]), int(sizes[-1])
        locs_r = be.gather(recv_feats, off_r, total_r)
        locs_back = _a2a(total_b, locs_r, send_loc, recv_loc, self.group)
        # 6. per-query order (feature slot f's list goes to the exclusive prefix sum of the slot lengths)
        loc_off, query_len, locs_q = be.assemble(m, lens_back, src_idx, F, locs_back, total_b, bases, seq_off, n_seqs,
                                                 paired, win_off)
        # 7. sort / sweep / top lists / fold on the home rank
        be.reduce(nq, loc_off, locs_q, query_len, cands, ncand, max_cand, emulate_ranks, insert_size_max, flags)
        self._last = {"n_features": m, "n_locations": total_b, "n_queries": nq,
                      "n_features_served": R, "n_locations_served": total_r}

    def last_stats(self):
        st = dict(self._last)
        if hasattr(self.be, "stats"):
            st.update({k: v for k, v in self.be.stats().items() if k in ("n_cands", "n_overflow")})
        st.setdefault("n_hit_features", 0)
        st.setdefault("n_cands", 0)
        st.setdefault("n_overflow", 0)
        return st

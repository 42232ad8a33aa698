"""TEST INFRASTRUCTURE (not product): torch-op second implementation of the table build, kept as an
independent check of csrc/mcq_build.hip.

GPU construction of the feature -> locations table from reference sequences.

Restates the reference's build-side insertion (add_all_window_sketches,
src/sketch_database.h:1079-1097, with the target -> rank assignment tgt % P of
:540-542): every window of every target is sketched with the same kernel the query
path uses; per (feature, virtual rank) only the first maxLocs=254 locations in
(target, window) order survive (src/sketch_database.h:1090-1092, max bucket
254: :375-378).  The result is the union table mcq_db_create expects.

Sorting / compaction here is torch plumbing around the HIP sketch kernel; it runs
once per database, outside any timed region.
"""
import importlib

import torch

engine = importlib.import_module("metacache-mpi_amd.engine")      # TEST INFRASTRUCTURE: a second, torch-plumbing table build

MAXLOCS = 254


def _empty_db(k, s, winlen, winstride, device):
    import numpy as np
    return engine.Database(np.zeros(0, np.uint32), np.zeros(1, np.uint64), np.zeros(0, np.uint64),
                           np.zeros(0, np.uint32), k=k, sketch_size=s, winlen=winlen, winstride=winstride,
                           device=device)


def sketch_windows(bases, seq_off, k=16, s=16, winlen=128, winstride=113, sketcher=None):
    """bases: uint8 cuda tensor; seq_off: int64 cuda tensor [n+1].
    Returns (win_off int64 [n+1], features int64 [n_win, s] with -1 = none, n_feat int32 [n_win])."""
    dev = bases.device
    n = seq_off.numel() - 1
    sk = sketcher or _empty_db(k, s, winlen, winstride, dev.index or 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    win_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    sk.count_windows(bases.data_ptr(), seq_off.data_ptr(), n, win_off.data_ptr(), st)
    n_win = int(win_off[-1].item())
    feats = torch.empty((max(n_win, 1), s), dtype=torch.int32, device=dev)
    nfeat = torch.empty(max(n_win, 1), dtype=torch.int32, device=dev)
    sk.sketch(bases.data_ptr(), seq_off.data_ptr(), n, win_off.data_ptr(), feats.data_ptr(), nfeat.data_ptr(), st)
    torch.cuda.synchronize(dev)
    f64 = feats[:n_win].to(torch.int64)
    f64 = torch.where(f64 == -1, f64, f64 & 0xFFFFFFFF)
    return win_off, f64, nfeat[:n_win]


def build_table(bases, seq_off, emulate_ranks=1, k=16, s=16, winlen=128, winstride=113, maxlocs=MAXLOCS,
                sketcher=None, remove_overpopulated=False):
    """Returns (keys int64 [nk] ascending, list_off int64 [nk+1], locs int64 [(tgt<<32)|win], win_off)."""
    dev = bases.device
    win_off, feats, _ = sketch_windows(bases, seq_off, k, s, winlen, winstride, sketcher)
    n_win = feats.shape[0]
    valid = (feats >= 0).reshape(-1)
    f = feats.reshape(-1)[valid]
    gwin = torch.arange(n_win, device=dev, dtype=torch.int64).repeat_interleave(s)[valid]
    del feats, valid
    tgt = torch.searchsorted(win_off, gwin, right=True) - 1
    P = max(1, int(emulate_ranks))
    # first maxlocs per (feature, virtual rank) in (tgt, win) == gwin order
    key = f * P + (tgt % P)
    key, order = torch.sort(key, stable=True)
    f, gwin, tgt = f[order], gwin[order], tgt[order]
    del order
    n = key.numel()
    idx = torch.arange(n, device=dev, dtype=torch.int64)
    is_start = torch.ones(n, dtype=torch.bool, device=dev)
    if n > 1:
        is_start[1:] = key[1:] != key[:-1]
    start = torch.where(is_start, idx, torch.zeros_like(idx))
    start = torch.cummax(start, 0).values
    keep = (idx - start) < maxlocs
    del key, idx, is_start, start
    f, gwin, tgt = f[keep], gwin[keep], tgt[keep]
    # merge the virtual ranks' lists of a feature into (tgt, win) order
    if P > 1:
        assert n_win < (1 << 31)
        k2, order = torch.sort((f << 31) | gwin)
        f, gwin, tgt = f[order], gwin[order], tgt[order]
        del k2, order
    if remove_overpopulated:
        # -remove-overpopulated-features (src/mode_build.cpp:847-1074): per-rank counts summed over the ranks
        _, inv, cnt = torch.unique_consecutive(f, return_inverse=True, return_counts=True)
        ok = cnt[inv] <= maxlocs - 1
        f, gwin, tgt = f[ok], gwin[ok], tgt[ok]
    win = gwin - win_off[tgt]
    locs = (tgt << 32) | win
    keys, counts = torch.unique_consecutive(f, return_counts=True)
    list_off = torch.zeros(keys.numel() + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=list_off[1:])
    return keys, list_off, locs, win_off


def make_database(keys, list_off, locs, tgt2tax, n_shards=1, shard_id=0, k=16, s=16, winlen=128, winstride=113,
                  tgt_winstride=0, flags=0):
    """cuda tensors -> engine.Database (device-pointer create, no host round trip)."""
    dev = keys.device

    def as_u32_bits(t):                      # values in [0, 2^32) -> int32 tensor with the same bits
        t = t.to(torch.int64) & 0xFFFFFFFF
        return torch.where(t >= (1 << 31), t - (1 << 32), t).to(torch.int32).contiguous()

    k32, t2t = as_u32_bits(keys), as_u32_bits(tgt2tax)
    list_off = list_off.contiguous(); locs = locs.contiguous()
    torch.cuda.synchronize(dev)
    db = engine.Database(None, None, None, None, k=k, sketch_size=s, winlen=winlen, winstride=winstride,
                         tgt_winstride=tgt_winstride, n_shards=n_shards, shard_id=shard_id, device=dev.index or 0, flags=flags,
                         device_ptrs=dict(keys=k32.data_ptr(), list_off=list_off.data_ptr(), locs=locs.data_ptr(),
                                          tgt2tax=t2t.data_ptr(), n_keys=k32.numel(), n_locs=locs.numel(),
                                          n_targets=t2t.numel()))
    return db

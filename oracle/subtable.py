"""TEST INFRASTRUCTURE (only tests/ and bench.py's cpu_baseline leg import anything under oracle/): the part of a GPU-resident table that
a batch of reads can touch, read back through the engine's staged entry points, so that the CPU oracle can be run against
tables that are too large to copy to the host (RefSeq scale: 1.5e10 locations).

Every feature of every window of the batch is looked up (mcq_count_windows + mcq_sketch give the features, mcq_lookup_count /
mcq_lookup_gather their lists in the handle's native words, decoded here to the public (tgt << 32) | win form): an oracle built
from the result sees, for each of the batch's features, exactly the list the full table holds, and no list for a feature the
full table does not hold -- so its answers for this batch are its answers on the full table.  (That the table's lists are the
reference's is pinned separately: tests/test_gpu_dbbuild.py against the reference's shard files.)"""
import numpy as np
import torch


def batch_subtable(eng, db, bases_ptr, seq_off_ptr, n_seqs, device, tgt_windows=None):
    """-> (keys u32 [n], list_off u64 [n+1], locs u64 [list_off[-1]]) as numpy arrays.  tgt_windows: int64 device tensor
    [n_targets], needed when the handle stores global-window words."""
    st = torch.cuda.current_stream(device).cuda_stream
    win_off = torch.empty(n_seqs + 1, dtype=torch.int64, device=device)
    db.count_windows(bases_ptr, seq_off_ptr, n_seqs, win_off.data_ptr(), st)
    torch.cuda.synchronize(device)
    nw = int(win_off[-1].item())
    s = db.sketch_size
    feats = torch.full((max(nw, 1) * s,), -1, dtype=torch.int32, device=device)
    nfeat = torch.zeros(max(nw, 1), dtype=torch.int32, device=device)
    db.sketch(bases_ptr, seq_off_ptr, n_seqs, win_off.data_ptr(), feats.data_ptr(), nfeat.data_ptr(), st)
    torch.cuda.synchronize(device)
    uniq = torch.unique(feats[feats != -1])
    del feats, nfeat
    n = uniq.numel()
    lens = torch.zeros(max(n, 1), dtype=torch.int32, device=device)
    if n:
        db.lookup_count(uniq.data_ptr(), n, lens.data_ptr(), None, st)
    torch.cuda.synchronize(device)
    hit = lens[:n] > 0
    uniq, lens = uniq[hit].contiguous(), lens[:n][hit].contiguous()
    n = uniq.numel()
    ooff = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(lens.to(torch.int64), 0, out=ooff[1:])
    lay = db.layout()
    total = int(ooff[-1].item())
    native = torch.zeros(total + 1, dtype=torch.int32 if lay["loc_bytes"] == 4 else torch.int64, device=device)
    if n:
        db.lookup_gather(uniq.data_ptr(), n, ooff.data_ptr(), native.data_ptr(), stream=st)
    torch.cuda.synchronize(device)
    w = native[:total].to(torch.int64)
    del native
    if lay["loc_bytes"] == 4:
        w &= 0xFFFFFFFF
    if lay["loc_format"] == eng.MCQ_LOC_GLOBAL_WINDOW:
        assert tgt_windows is not None, "global-window words need the windows per target"
        go = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), torch.cumsum(tgt_windows.to(torch.int64), 0)])
        t = torch.searchsorted(go, w, right=True) - 1
        w = (t << 32) | (w - go[t])
    elif lay["loc_bytes"] == 4:
        wb = lay["win_bits"]
        w = ((w >> wb) << 32) | (w & ((1 << wb) - 1))
    keys = uniq.cpu().numpy().view(np.uint32)
    order = np.argsort(keys, kind="stable")          # (torch sorted them as signed values)
    lens_h = lens.cpu().numpy().astype(np.int64)
    starts = ooff[:-1].cpu().numpy()
    locs_h = w.cpu().numpy().view(np.uint64)
    keys_s, lens_s = keys[order], lens_h[order]
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum(lens_s)
    idx = np.repeat(starts[order] - off[:-1].astype(np.int64), lens_s) + np.arange(int(off[-1]), dtype=np.int64)
    return np.ascontiguousarray(keys_s), off, np.ascontiguousarray(locs_h[idx])


def merge_subtables(tables):
    """Sub-tables of the same read set read back from the hash-range shards of one table (keys disjoint by construction) ->
    one (keys, list_off, locs) for the oracle."""
    keys = np.concatenate([t[0] for t in tables])
    lens = np.concatenate([np.diff(t[1].astype(np.int64)) for t in tables])
    base, starts = 0, []
    for t in tables:
        starts.append(t[1][:-1].astype(np.int64) + base)
        base += int(t[1][-1])
    starts = np.concatenate(starts)
    locs = np.concatenate([t[2] for t in tables])
    assert len(np.unique(keys)) == len(keys), "a feature came back from two shards"
    order = np.argsort(keys, kind="stable")
    keys_s, lens_s = keys[order], lens[order]
    off = np.zeros(len(keys) + 1, np.uint64)
    off[1:] = np.cumsum(lens_s)
    idx = np.repeat(starts[order] - off[:-1].astype(np.int64), lens_s) + np.arange(int(off[-1]), dtype=np.int64)
    return np.ascontiguousarray(keys_s.astype(np.uint32)), off, np.ascontiguousarray(locs[idx].astype(np.uint64))

"""Row f1 at scale: the reference's shard files through the STREAMING route (host: mcq_refdb_open_meta + mcq_shard_stream_*; GPU:
mcq_parts_builder_* -> mcq_db_from_parts) must give the table the host-side union route gives: every key's list, read back
from the handle, equals the union of the reference-written shard files; and a handle made this way answers the fixture reads
like the reference's own run."""
import importlib

import numpy as np
import pytest
import torch

from golden_util import Fixture
from oracle import dbfile
from oracle import mc_oracle as orc

pytestmark = pytest.mark.gpu


def _lists_of(eng, db, keys, tgt_windows):
    """(list lengths [n_keys], lists back to back as (tgt << 32) | win) of `keys` out of a handle"""
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    k32 = torch.from_numpy(np.ascontiguousarray(keys).view(np.int32).copy()).to(dev)
    n = k32.numel()
    lens = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    db.lookup_count(k32.data_ptr(), n, lens.data_ptr(), None, st)
    ooff = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens[:n].to(torch.int64), 0, out=ooff[1:])
    lay = db.layout()
    native = torch.zeros(int(ooff[-1].item()) + 1, dtype=torch.int32 if lay["loc_bytes"] == 4 else torch.int64, device=dev)
    db.lookup_gather(k32.data_ptr(), n, ooff.data_ptr(), native.data_ptr(), stream=st)
    torch.cuda.synchronize()
    w = native[:-1].to(torch.int64)
    if lay["loc_bytes"] == 4:
        w &= 0xFFFFFFFF
    if lay["loc_format"] == eng.MCQ_LOC_GLOBAL_WINDOW:
        go = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(torch.from_numpy(tgt_windows.astype(np.int64)).to(dev), 0)])
        t = torch.searchsorted(go, w, right=True) - 1
        w = (t << 32) | (w - go[t])
    elif lay["loc_bytes"] == 4:
        wb = lay["win_bits"]
        w = ((w >> wb) << 32) | (w & ((1 << wb) - 1))
    return lens[:n].cpu().numpy().astype(np.int64), w.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("tag,P", [("mini", 4), ("mini", 8), ("overpop", 2), ("wide", 16), ("tie", 4)])
@pytest.mark.parametrize("n_ranges,n_shards,chunk", [(0, 1, 1 << 20), (3, 1, 1000), (2, 2, 255)])
def test_streamed_table_equals_the_union_of_the_shard_files(tag, P, n_ranges, n_shards, chunk):
    eng = importlib.import_module("metacache-mpi_amd.engine")
    host = importlib.import_module("metacache-mpi_amd.host")
    importlib.import_module("metacache-mpi_amd").build_host()
    fx = Fixture(tag, P)
    rdb = host.RefDb(fx.shard_paths[0][: -len(".db_0")], P, meta_only=True)
    i = rdb.info
    tw = rdb.tgt_windows()
    rk, ro, rl = dbfile.union_shards(fx.shards)
    want_len = np.diff(ro.astype(np.int64))
    t2t = rdb.tgt2tax(fx.lowest)
    got_len = np.zeros(len(rk), np.int64)
    got_lists = {}
    for sid in range(n_shards):
        pb = eng.PartsBuilder(tw, k=i.k, sketch_size=i.q_sketch_size, winlen=i.q_winlen, winstride=i.q_winstride, tgt_winstride=i.winstride,
                              expected_locations=i.n_locs, n_ranges=n_ranges, n_shards=n_shards, shard_id=sid)
        for r in reversed(range(P)):                    # (any order of files and chunks)
            for f, t, w in rdb.stream(r, chunk):
                pb.add(f, t, w)
        parts = pb.finish()
        assert parts.n_parts == (n_ranges or 1)
        db = parts.database(t2t.ctypes.data, n_shards=n_shards, shard_id=sid, device_ptrs=False)
        parts.close()
        assert db.layout()["loc_format"] == eng.MCQ_LOC_GLOBAL_WINDOW
        lens, lists = _lists_of(eng, db, rk, tw)
        own = np.array([eng.owner(int(k), n_shards) == sid for k in rk])
        assert np.array_equal(lens[own], want_len[own]) and not lens[~own].any()
        got_len += lens
        off = np.zeros(len(rk) + 1, np.int64); off[1:] = np.cumsum(lens)
        for j in np.nonzero(own)[0][:: max(1, len(rk) // 4000)]:
            assert np.array_equal(lists[off[j]:off[j + 1]], rl[int(ro[j]):int(ro[j + 1])]), (sid, j)
        if n_shards == 1:
            assert np.array_equal(lists, rl)
            # ... and the handle answers the fixture's reads like the reference run (oracle on the union table)
            p = fx.params
            odb = orc.OracleDb(rk, ro, rl, t2t, k=p["qk"], s=p["qs"], winlen=p["qwinlen"], winstride=p["qwinstride"], tgt_winstride=p["winstride"])
            bases, seq_off = orc.pack_reads(fx.interleaved())
            ws = eng.Workspace(db, len(fx.names), len(bases))
            cands, ncand = ws.query_host(bases, seq_off, True, max_cand=fx.maxcand, emulate_ranks=P, flags=eng.MCQ_QUIRK_SEQ_DROP)
            oc, on = odb.query(bases, seq_off, True, max_cand=fx.maxcand, emulate_ranks=P, quirk_seq_drop=1)
            assert np.array_equal(ncand, on)
            for q in range(len(on)):
                assert np.array_equal(cands[q, :on[q]], oc[q, :on[q]]), q
            ws.close()
        db.close()
    assert np.array_equal(got_len, want_len)


def test_builder_rejects_triples_outside_the_database():
    eng = importlib.import_module("metacache-mpi_amd.engine")
    pb = eng.PartsBuilder(np.array([10, 20], np.uint32))
    pb.add(np.array([5], np.uint32), np.array([1], np.uint32), np.array([19], np.uint32))
    with pytest.raises(eng.McqError):
        pb.add(np.array([5], np.uint32), np.array([1], np.uint32), np.array([20], np.uint32))      # window 20 of a target with 20 windows
    with pytest.raises(eng.McqError):
        pb.add(np.array([5], np.uint32), np.array([2], np.uint32), np.array([0], np.uint32))       # target 2 of 2
    pb.close()

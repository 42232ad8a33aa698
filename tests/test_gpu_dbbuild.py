"""Row f2 (GPU DB build): dbbuild.build_table on the fixture genomes must reproduce the
union of the shard tables the reference wrote at P ranks -- same keys, same lists in the
same order, including the per-rank truncation to 254 locations."""
import gzip
import importlib
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN, Fixture
from oracle import dbfile

pytestmark = pytest.mark.gpu


def _load_genomes(tag, dev):
    seqs = []
    with gzip.open(os.path.join(GOLDEN, tag, "genomes.fa.gz"), "rt") as f:
        for line in f:
            if not line.startswith(">"):
                seqs.append(line.strip().encode())
    off = np.zeros(len(seqs) + 1, np.int64); off[1:] = np.cumsum([len(s) for s in seqs])
    bases = torch.from_numpy(np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()).to(dev)
    return bases, torch.from_numpy(off).to(dev)


@pytest.mark.parametrize("tag,P", [("mini", 2), ("mini", 4), ("mini", 8), ("tie", 4), ("noanc", 2)])
def test_table_equals_reference_shards(tag, P):
    dbbuild = importlib.import_module("dbbuild_torch")
    dev = torch.device("cuda", 0)
    fx = Fixture(tag, P)
    bases, off = _load_genomes(tag, dev)
    keys, loff, locs, win_off = dbbuild.build_table(bases, off, emulate_ranks=P)
    rk, ro, rl = dbfile.union_shards(fx.shards)
    assert np.array_equal(keys.cpu().numpy().astype(np.uint32), rk)
    assert np.array_equal(loff.cpu().numpy().astype(np.uint64), ro)
    assert np.array_equal(locs.cpu().numpy().astype(np.uint64), rl)
    # window counts per target as recorded by the owning rank (taxon source.windows)
    nwin = (win_off[1:] - win_off[:-1]).cpu().numpy()
    for t in range(fx.n_targets):
        rec = [s["taxa"][fx.tax.by_id[-(t + 1)]]["windows"] for s in fx.shards]
        assert max(rec) == nwin[t]


def test_truncation_to_254_per_virtual_rank():
    # one 128-base window repeated 700 times: every feature has 700 locations in one target set
    dbbuild = importlib.import_module("dbbuild_torch")
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    unit = rng.choice(list(b"ACGT"), 113).astype(np.uint8)
    g = np.tile(unit, 400)                       # windows at stride 113 all see the same k-mers
    seqs = [g.tobytes(), g.tobytes(), g.tobytes()]
    off = np.zeros(4, np.int64); off[1:] = np.cumsum([len(s) for s in seqs])
    bases = torch.from_numpy(np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()).to(dev)
    for P in (1, 2):
        keys, loff, locs, _ = dbbuild.build_table(bases, torch.from_numpy(off).to(dev), emulate_ranks=P)
        n = (loff[1:] - loff[:-1]).cpu().numpy()
        l = locs.cpu().numpy()
        tg = l >> 32
        for i in range(len(n)):
            lst = tg[int(loff[i]):int(loff[i + 1])]
            for r in range(P):
                assert (lst % P == r).sum() <= 254
        assert n.max() == (254 if P == 1 else 254 + min(254, 399 - 0))   # ranks: targets {0,2} and {1}


# ---- the same through the C ABI (mcq_build_table / mcq_db_build, csrc/mcq_build.hip) -------------

@pytest.mark.parametrize("tag,P", [("mini", 2), ("mini", 4), ("mini", 8), ("tie", 4), ("noanc", 2)])
def test_abi_table_equals_reference_shards(tag, P):
    engine = importlib.import_module("metacache-mpi_amd.engine")
    dev = torch.device("cuda", 0)
    fx = Fixture(tag, P)
    bases, off = _load_genomes(tag, dev)
    tb = engine.Table(bases.data_ptr(), off.data_ptr(), off.numel() - 1, emulate_ranks=P)
    keys, loff, locs, _ = tb.to_host()
    rk, ro, rl = dbfile.union_shards(fx.shards)
    assert np.array_equal(keys, rk)
    assert np.array_equal(loff, ro)
    assert np.array_equal(locs, rl)
    tb.close()


def test_abi_table_host_pointers_and_truncation():
    engine = importlib.import_module("metacache-mpi_amd.engine")
    dbbuild = importlib.import_module("dbbuild_torch")
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    unit = rng.choice(list(b"ACGT"), 113).astype(np.uint8)
    g = np.tile(unit, 400)
    extra = rng.choice(list(b"ACGTN"), 5000, p=[.24, .24, .24, .24, .04]).astype(np.uint8)
    seqs = [g.tobytes(), g.tobytes(), extra.tobytes(), g.tobytes(), b"ACGT"]      # a target shorter than k as well
    off = np.zeros(len(seqs) + 1, np.uint64); off[1:] = np.cumsum([len(s) for s in seqs])
    host = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    bases = torch.from_numpy(host).to(dev)
    for P in (1, 2, 3):
        tb = engine.Table(host.ctypes.data, off.ctypes.data, len(seqs), emulate_ranks=P, device_ptrs=False)
        keys, loff, locs, win_off = tb.to_host()
        k2, o2, l2, w2 = dbbuild.build_table(bases, torch.from_numpy(off.astype(np.int64)).to(dev), emulate_ranks=P)
        assert np.array_equal(keys, k2.cpu().numpy().astype(np.uint32))
        assert np.array_equal(loff, o2.cpu().numpy().astype(np.uint64))
        assert np.array_equal(locs, l2.cpu().numpy().astype(np.uint64))
        assert np.array_equal(win_off, w2.cpu().numpy().astype(np.uint64))
        tb.close()


@pytest.mark.parametrize("flags", [0, 0x200, 0x2000 | 0x4000], ids=["loc32", "loc64", "gw-slots16"])
@pytest.mark.parametrize("tag,P", [("mini", 4), ("tie", 2)])
def test_abi_db_build_queries_like_the_reference(tag, P, flags):
    """mcq_db_build from the fixture genomes, then the fixture reads: the top hits the reference's
    own `mpiexec -n P` run printed (final.json)"""
    engine = importlib.import_module("metacache-mpi_amd.engine")
    from oracle import mc_oracle as orc
    dev = torch.device("cuda", 0)
    fx = Fixture(tag, P)
    bases, off = _load_genomes(tag, dev)
    t2t = torch.from_numpy(np.asarray(fx.tgt2tax(), np.uint32).view(np.int32).copy()).to(dev)
    db = engine.Database.build(bases.data_ptr(), off.data_ptr(), t2t.data_ptr(), off.numel() - 1, emulate_ranks=P, flags=flags)
    assert db.loc_bytes() == (8 if flags == 0x200 else 4)
    assert db.layout()["loc_format"] == (2 if flags & 0x2000 else 0 if flags == 0x200 else 1)
    rb, ro = orc.pack_reads(fx.interleaved())
    ws = engine.Workspace(db, len(fx.names), len(rb))
    cands, ncand = ws.query_host(rb, ro, True, max_cand=fx.maxcand, emulate_ranks=P, flags=engine.MCQ_QUIRK_SEQ_DROP)
    for q, name in enumerate(fx.names):
        mine = [[fx.tax.id_of_key(c[0]), int(c[1])] for c in cands[q, :ncand[q]]]
        assert mine == fx.final[name]["tophits"], (name, mine, fx.final[name])


def test_abi_build_rejects_bad_arguments():
    engine = importlib.import_module("metacache-mpi_amd.engine")
    host = np.frombuffer(b"ACGTACGTACGTACGTACGTACGT", dtype=np.uint8).copy()
    off = np.array([0, 24], np.uint64)
    with pytest.raises(engine.McqError):
        engine.Table(host.ctypes.data, off.ctypes.data, 1, k=40, device_ptrs=False)       # k > 32
    with pytest.raises(engine.McqError):
        engine.Table(host.ctypes.data, off.ctypes.data, 0, device_ptrs=False)             # no targets


@pytest.mark.parametrize("P", [2, 4])
def test_remove_overpopulated_features_like_the_reference(P):
    """fixture built by the reference with -remove-overpopulated-features: counts are summed over the ranks"""
    engine = importlib.import_module("metacache-mpi_amd.engine")
    dbbuild = importlib.import_module("dbbuild_torch")
    dev = torch.device("cuda", 0)
    fx = Fixture("overpop", P)
    bases, off = _load_genomes("overpop", dev)
    rk, ro, rl = dbfile.union_shards(fx.shards)
    tb = engine.Table(bases.data_ptr(), off.data_ptr(), off.numel() - 1, emulate_ranks=P,
                      flags=engine.MCQ_BUILD_REMOVE_OVERPOPULATED)
    keys, loff, locs, _ = tb.to_host()
    assert np.array_equal(keys, rk) and np.array_equal(loff, ro) and np.array_equal(locs, rl)
    k2, o2, l2, _ = dbbuild.build_table(bases, off, emulate_ranks=P, remove_overpopulated=True)
    assert np.array_equal(k2.cpu().numpy().astype(np.uint32), rk)
    assert np.array_equal(o2.cpu().numpy().astype(np.uint64), ro)
    assert np.array_equal(l2.cpu().numpy().astype(np.uint64), rl)
    # without the option the repeats stay (and the table differs)
    tb2 = engine.Table(bases.data_ptr(), off.data_ptr(), off.numel() - 1, emulate_ranks=P)
    assert tb2.n_keys > tb.n_keys and tb2.n_locs > tb.n_locs


# ---- f2 -> f1: shard files written from the GPU-built table ARE the files the reference wrote --------------------------
@pytest.mark.parametrize("tag,P", [("mini", 2), ("mini", 4), ("mini", 8), ("overpop", 2)])
def test_shards_written_from_the_gpu_build_are_the_references_files(tag, P, tmp_path):
    """The table built on the GPU, split by tgt % P and written through the host library (mcq_refdb_write_shard), byte for byte
    the shard files the reference's own `build` wrote for these genomes (tests/golden/*/P*/*.db_<r>: what its `query`, and
    ref_query for the committed M / T / C dumps, read).  The keys of a file are written in the order the reference's hash map
    happened to iterate in; the GPU-built keys (sorted) are put into that order -- the same SET of keys is asserted first.
    (Until r04 this test handed the written files to the compiled reference on the GPU box; nothing of the reference travels
    there any more.)"""
    engine = importlib.import_module("metacache-mpi_amd.engine")
    host = importlib.import_module("metacache-mpi_amd.host")
    importlib.import_module("metacache-mpi_amd").build_host()
    dev = torch.device("cuda", 0)
    fx = Fixture(tag, P)
    bases, off = _load_genomes(tag, dev)
    flags = engine.MCQ_BUILD_REMOVE_OVERPOPULATED if tag == "overpop" else 0
    tb = engine.Table(bases.data_ptr(), off.data_ptr(), off.numel() - 1, emulate_ranks=P, flags=flags)
    keys, loff, locs, _ = tb.to_host()
    key_of = np.repeat(np.arange(len(keys)), np.diff(loff.astype(np.int64)))
    rank_of = (locs >> np.uint64(32)).astype(np.int64) % P
    for r in range(P):
        sel = rank_of == r
        kk, cnt = np.unique(key_of[sel], return_counts=True)
        s = fx.shards[r]
        mine = keys[kk]
        assert np.array_equal(np.sort(s["keys"]), mine), "rank %d: other keys than the reference's file" % r
        # into the file's key order
        pos = np.searchsorted(mine, s["keys"])
        starts = np.zeros(len(kk) + 1, np.int64); starts[1:] = np.cumsum(cnt)
        rl = locs[sel]
        o = np.zeros(len(kk) + 1, np.uint64); o[1:] = np.cumsum(cnt[pos])
        idx = np.repeat(starts[pos] - o[:-1].astype(np.int64), cnt[pos]) + np.arange(int(o[-1]), dtype=np.int64)
        p = s["params"]
        out = str(tmp_path / ("%s.db_%d" % (tag, r)))
        host.write_shard(out, dict(k=p["k"], sketch_size=p["s"], winlen=p["winlen"], winstride=p["winstride"], q_k=p["qk"],
                                   q_sketch_size=p["qs"], q_winlen=p["qwinlen"], q_winstride=p["qwinstride"],
                                   max_locs_per_feature=p["maxlocs"]),
                         s["taxa"], s["target_count"], mine[pos], o, rl[idx])
        assert open(out, "rb").read() == open(fx.shard_paths[r], "rb").read(), "rank %d" % r


# ---- the build in parts (mcq_build_parts / mcq_db_create_parts: tables whose one-piece build does not fit) ----------------
def _all_lists_of(engine, db, keys, win_off):
    """every key's list out of a handle (mcq_lookup_count / _gather), as (tgt << 32) | win: list lengths [n_keys] and
    the lists back to back"""
    dev = torch.device("cuda", 0)
    k32 = torch.from_numpy(np.ascontiguousarray(keys).view(np.int32).copy()).to(dev)
    n = k32.numel()
    lens = torch.zeros(n, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    db.lookup_count(k32.data_ptr(), n, lens.data_ptr(), None, st)
    ooff = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens.to(torch.int64), 0, out=ooff[1:])
    lay = db.layout()
    native = torch.zeros(int(ooff[-1].item()) + 1, dtype=torch.int32 if lay["loc_bytes"] == 4 else torch.int64, device=dev)
    db.lookup_gather(k32.data_ptr(), n, ooff.data_ptr(), native.data_ptr(), stream=st)
    torch.cuda.synchronize()
    w = native[:-1].to(torch.int64) & (0xFFFFFFFF if lay["loc_bytes"] == 4 else -1)
    if lay["loc_format"] == engine.MCQ_LOC_GLOBAL_WINDOW:
        go = torch.from_numpy(win_off.astype(np.int64)).to(dev)
        t = torch.searchsorted(go, w, right=True) - 1
        out = (t << 32) | (w - go[t])
    elif lay["loc_bytes"] == 4:
        wb = lay["win_bits"]
        out = ((w >> wb) << 32) | (w & ((1 << wb) - 1))
    else:
        out = w
    return lens.cpu().numpy().astype(np.int64), out.cpu().numpy().astype(np.uint64)


@pytest.mark.parametrize("P,flags,cap", [(1, 0, 0), (2, 0, 0), (3, 0x1000, 0), (2, 0x1000, 5000)])
def test_build_in_parts_equals_the_one_piece_build(P, flags, cap, monkeypatch):
    """the same sequences through mcq_build_table (the build that is pinned to the reference's shard files above) and through
    mcq_build_parts with 3 feature ranges and sketch chunks of a few targets: every key, every list, and any shard of it.
    cap > 0: the pair arrays of a part start that small and must GROW (a range that holds more than its even share of the features:
    repeats; until r04 that was MCQ_E_CAPACITY)"""
    engine = importlib.import_module("metacache-mpi_amd.engine")
    synth = importlib.import_module("metacache-mpi_amd.synth")
    dev = torch.device("cuda", 0)
    gb, goff, species = synth.make_genomes(5, 9, 80_000, 160_000, 0.02, seed=12, device=dev)
    if flags:        # a repeat family, so that -remove-overpopulated-features and the 254 limit have something to do
        unit = gb[:113].clone()
        rep = unit.repeat(300)
        gb = torch.cat([gb, rep, rep]); goff = torch.cat([goff, goff[-1:] + rep.numel(), goff[-1:] + 2 * rep.numel()])
        species = torch.cat([species, species[-1:] + 1, species[-1:] + 1])
    nt = goff.numel() - 1
    tb = engine.Table(gb.data_ptr(), goff.data_ptr(), nt, emulate_ranks=P, flags=flags)
    keys, loff, locs, win_off = tb.to_host()
    tb.close()
    monkeypatch.setenv("MCQ_BUILD_PARTS", "3")
    monkeypatch.setenv("MCQ_BUILD_CHUNK_WINDOWS", "9000")
    if cap:
        monkeypatch.setenv("MCQ_BUILD_PART_CAP", str(cap))
    parts = engine.Parts(gb.data_ptr(), goff.data_ptr(), nt, emulate_ranks=P, flags=flags)
    assert parts.n_parts == 3 and parts.n_keys == len(keys) and parts.n_locs == len(locs) and parts.n_windows == int(win_off[-1])
    sp32 = species.to(torch.int32).contiguous()
    want_len = np.diff(loff.astype(np.int64))
    for layout in (0, engine.MCQ_DB_SLOTS_16, engine.MCQ_DB_BUCKETS_64):
        db = parts.database(sp32.data_ptr(), flags=layout)
        lay = db.layout()
        assert lay["loc_format"] == engine.MCQ_LOC_GLOBAL_WINDOW and lay["n_keys"] == len(keys) and lay["n_locs"] == len(locs)
        lens, lists = _all_lists_of(engine, db, keys, win_off)
        assert np.array_equal(lens, want_len) and np.array_equal(lists, locs), layout
        db.close()
    # shards of it, from the same parts and from a build that only made the shard
    seen = 0
    for sid in range(2):
        own = np.array([engine.owner(int(k), 2) == sid for k in keys])
        for src in (parts, engine.Parts(gb.data_ptr(), goff.data_ptr(), nt, emulate_ranks=P, flags=flags, n_shards=2, shard_id=sid)):
            db = src.database(sp32.data_ptr(), n_shards=2, shard_id=sid)
            lens, lists = _all_lists_of(engine, db, keys, win_off)
            assert np.array_equal(lens, np.where(own, want_len, 0))
            assert np.array_equal(lists, locs[np.repeat(own, want_len)])
            assert db.layout()["n_keys"] == int(own.sum())
            db.close()
            if src is not parts:
                assert src.n_keys == int(own.sum())
                src.close()
        seen += int(own.sum())
    assert seen == len(keys)
    parts.close()


@pytest.mark.parametrize("tag,P", [("mini", 4), ("tie", 2), ("overpop", 2)])
def test_db_build_in_parts_queries_like_the_reference(tag, P, monkeypatch):
    """mcq_db_build taking the in-parts way (forced): the fixture reads get the top hits the reference's own run printed"""
    engine = importlib.import_module("metacache-mpi_amd.engine")
    from oracle import mc_oracle as orc
    dev = torch.device("cuda", 0)
    fx = Fixture(tag, P)
    bases, off = _load_genomes(tag, dev)
    t2t = torch.from_numpy(np.asarray(fx.tgt2tax(), np.uint32).view(np.int32).copy()).to(dev)
    monkeypatch.setenv("MCQ_BUILD_PARTS", "2")
    monkeypatch.setenv("MCQ_BUILD_CHUNK_WINDOWS", "700")
    db = engine.Database.build(bases.data_ptr(), off.data_ptr(), t2t.data_ptr(), off.numel() - 1, emulate_ranks=P,
                               flags=engine.MCQ_BUILD_REMOVE_OVERPOPULATED if tag == "overpop" else 0)
    assert db.layout()["loc_format"] == engine.MCQ_LOC_GLOBAL_WINDOW
    rb, ro = orc.pack_reads(fx.interleaved())
    ws = engine.Workspace(db, len(fx.names), len(rb))
    cands, ncand = ws.query_host(rb, ro, True, max_cand=fx.maxcand, emulate_ranks=P, flags=engine.MCQ_QUIRK_SEQ_DROP)
    for q, name in enumerate(fx.names):
        mine = [[fx.tax.id_of_key(c[0]), int(c[1])] for c in cands[q, :ncand[q]]]
        assert mine == fx.final[name]["tophits"], (name, mine, fx.final[name])

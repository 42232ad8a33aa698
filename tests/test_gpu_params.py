"""Non-default sketching parameters (k, sketch size, window length / stride, query-side
overrides): engine vs oracle on a synthetic database built with the same parameters."""
import importlib

import numpy as np
import pytest
import torch

from oracle import mc_oracle as orc

pytestmark = pytest.mark.gpu

# (k, s, winlen, winstride, query winstride or None)
PARAMS = [
    (16, 16, 128, 113, None),     # the reference's defaults
    (12, 8, 64, 53, None),        # small windows: more windows per read (block path for 150 bp x 2 mates)
    (16, 32, 128, 113, None),     # large sketches: 64 features per single read
    (8, 4, 100, 93, None),
    (16, 16, 128, 64, None),      # overlapping windows (stride < winlen - k + 1)
    (16, 16, 128, 113, 57),       # query-side stride differs from the target stride (src/mode_query.cpp:380-387)
    (1, 3, 20, 20, None),         # degenerate but legal
]


@pytest.mark.parametrize("k,s,W,S,qS", PARAMS)
def test_parameters(k, s, W, S, qS):
    eng = importlib.import_module("metacache-mpi_amd.engine")
    dbbuild = importlib.import_module("dbbuild_torch")
    synth = importlib.import_module("metacache-mpi_amd.synth")
    dev = torch.device("cuda", 0)
    gb, goff, species = synth.make_genomes(4, 5, 30_000, 60_000, 0.03, seed=17, device=dev)
    keys, off, locs, _ = dbbuild.build_table(gb, goff, emulate_ranks=2, k=k, s=s, winlen=W, winstride=S)
    qs = qS or S
    db = dbbuild.make_database(keys, off, locs, species, k=k, s=s, winlen=W, winstride=qs, tgt_winstride=S)
    odb = orc.OracleDb(keys.cpu().numpy().astype(np.uint32), off.cpu().numpy().astype(np.uint64),
                       locs.cpu().numpy().astype(np.uint64), species.cpu().numpy().astype(np.uint32),
                       k=k, s=s, winlen=W, winstride=qs, tgt_winstride=S)
    n, L = 6000, 150
    reads, roff, _ = synth.sample_reads(gb, goff, n, L, 0.01, 0.003, seed=5)
    rb = reads.cpu().numpy().tobytes(); ro = roff.cpu().numpy().astype(np.uint64)
    ws = eng.Workspace(db, n, n * L)
    for paired in (False, True):
        for P, M in ((1, 3), (2, 2)):
            cands, ncand = ws.query_host(rb, ro, paired, max_cand=M, emulate_ranks=P, insert_size_max=400 if paired else 0)
            oc, on = odb.query(rb, ro, paired, max_cand=M, emulate_ranks=P, insert_size_max=400 if paired else 0, threads=8)
            assert np.array_equal(ncand, on), (paired, P, np.nonzero(ncand != on)[0][:5])
            mask = np.arange(M)[None, :] < on[:, None]
            assert np.array_equal(cands[mask], oc[mask]), (paired, P)
    st = ws.sync()
    assert st["n_features"] > 0


def test_rejected_parameters():
    eng = importlib.import_module("metacache-mpi_amd.engine")
    z = (np.zeros(0, np.uint32), np.zeros(1, np.uint64), np.zeros(0, np.uint64), np.zeros(0, np.uint32))
    for kw in (dict(k=17), dict(k=0), dict(sketch_size=33), dict(winlen=129), dict(winlen=8, k=16), dict(winstride=0)):
        with pytest.raises(eng.McqError) as e:
            eng.Database(*z, **kw)
        assert e.value.code in (eng.MCQ_E_UNSUPPORTED, eng.MCQ_E_ARG)
    db = eng.Database(*z)
    ws = eng.Workspace(db, 4, 64)
    for kw in (dict(max_cand=0), dict(max_cand=17), dict(emulate_ranks=65), dict(flags=0x1000)):
        with pytest.raises(eng.McqError):
            ws.query_host(b"ACGT", np.array([0, 4], np.uint64), False, **kw)
    # an empty database answers every query with no candidates
    c, n = ws.query_host(b"ACGTACGTACGTACGTACGTACGT", np.array([0, 24], np.uint64), False)
    assert n.tolist() == [0]


def test_capacity_error_is_reported_not_hidden():
    """a query whose match list exceeds the workspace's per-query capacity gets no result and
    mcq_ws_sync returns MCQ_E_CAPACITY (the reference's analogue: a FAIL log line, src/querying.h:833-847)"""
    eng = importlib.import_module("metacache-mpi_amd.engine")
    dbbuild = importlib.import_module("dbbuild_torch")
    synth = importlib.import_module("metacache-mpi_amd.synth")
    dev = torch.device("cuda", 0)
    gb, goff, species = synth.make_genomes(2, 12, 60_000, 80_000, 0.01, seed=23, device=dev)
    keys, off, locs, _ = dbbuild.build_table(gb, goff, emulate_ranks=1)
    db = dbbuild.make_database(keys, off, locs, species)
    reads, roff, _ = synth.sample_long_reads(gb, goff, 4, 20000, 0.0, seed=1, min_len=15000, max_len=25000)
    rb = reads.cpu().numpy().tobytes(); ro = roff.cpu().numpy().astype(np.uint64)
    small = eng.Workspace(db, 4, len(rb), max_locs_per_query=1024)
    with pytest.raises(eng.McqError) as e:
        small.query_host(rb, ro, False, max_cand=2)
    assert e.value.code == eng.MCQ_E_CAPACITY
    big = eng.Workspace(db, 4, len(rb))
    c, n = big.query_host(rb, ro, False, max_cand=2)
    assert (n > 0).all() and big.sync()["n_overflow"] == 4


def _reduce_against_oracle(make_lists):
    import torch
    from golden_util import Fixture
    from oracle import dbfile
    eng = importlib.import_module("metacache-mpi_amd.engine")
    dev = torch.device("cuda", 0)
    fx = Fixture("mini", 2)
    keys, off, locs = dbfile.union_shards(fx.shards)
    p = fx.params
    t2t = fx.tgt2tax()
    odb = orc.OracleDb(keys, off, locs, t2t, k=p["qk"], s=p["qs"], winlen=p["qwinlen"], winstride=p["qwinstride"],
                       tgt_winstride=p["winstride"])
    lists = make_lists(locs)
    loc_off = np.zeros(len(lists) + 1, np.int64); loc_off[1:] = np.cumsum([len(x) for x in lists])
    allv = np.concatenate(lists)
    # global-window form: windows per target = 1 + its largest window id in the table (what the handle derives)
    ext = np.zeros(len(t2t), np.int64)
    np.maximum.at(ext, (locs >> np.uint64(32)).astype(np.int64), (locs & np.uint64(0xFFFFFFFF)).astype(np.int64) + 1)
    gw_off = np.concatenate([[0], np.cumsum(ext)]).astype(np.uint64)
    qlen = torch.full((len(lists),), 150, dtype=torch.int32, device=dev)
    doff = torch.from_numpy(loc_off).to(dev)
    for dbflags in (0, eng.MCQ_DB_LOCS_GW):           # the staged reduce kernels in both 32-bit location forms
        db = eng.Database(keys, off, locs, t2t, k=p["qk"], sketch_size=p["qs"], winlen=p["qwinlen"], winstride=p["qwinstride"],
                          tgt_winstride=p["winstride"], flags=dbflags)
        wb = db.win_bits()
        if db.layout()["loc_format"] == eng.MCQ_LOC_GLOBAL_WINDOW:
            dl = torch.from_numpy((gw_off[(allv >> np.uint64(32)).astype(np.int64)] + (allv & np.uint64(0xFFFFFFFF))).astype(np.uint32).view(np.int32)).to(dev)
        elif db.loc_bytes() == 4:
            dl = torch.from_numpy((((allv >> np.uint64(32)) << np.uint64(wb)) | (allv & np.uint64(0xFFFFFFFF))).astype(np.uint32).view(np.int32)).to(dev)
        else:
            dl = torch.from_numpy(allv.view(np.int64)).to(dev)
        ws = eng.Workspace(db, len(lists), 1)
        for P, M in ((1, 4), (2, 2), (32, 4)):          # (32, 4): lists in the workgroup kernel's LDS
            for flags in (0, eng.MCQ_FORCE_RAW_SORT):
                cands = torch.zeros((len(lists), M, 4), dtype=torch.int32, device=dev)
                ncand = torch.zeros(len(lists), dtype=torch.int32, device=dev)
                ws.reduce_device(len(lists), doff.data_ptr(), dl.data_ptr(), qlen.data_ptr(), cands.data_ptr(), ncand.data_ptr(),
                                 max_cand=M, emulate_ranks=P, flags=flags, stream=torch.cuda.current_stream(dev).cuda_stream)
                ws.sync()
                gc = cands.cpu().numpy().view(np.uint32); gn = ncand.cpu().numpy().view(np.uint32)
                for q, lst in enumerate(lists):
                    oc, on = odb.reduce_query(lst, 150, max_cand=M, emulate_ranks=P)
                    assert gn[q] == on, (q, P, M, flags, len(lst), len(np.unique(lst)))
                    assert np.array_equal(gc[q, :on], oc[:on]), (q, P, M, flags, len(lst), len(np.unique(lst)), gc[q], oc)


def test_reduce_counts_any_multiset_exactly():
    """mcq_reduce on caller-made location lists with hundreds of copies of one (target, window): the de-duplicating
    tail must count them exactly (16-bit table counters), like the oracle's reduce on the same multiset"""
    def make(locs):
        rng = np.random.default_rng(9)
        lists = []
        for q in range(300):
            base = rng.choice(locs, size=int(rng.integers(1, 12)))
            reps = rng.integers(1, 500, size=len(base))
            reps = np.minimum(reps, max(1, 510 // len(base)))
            lists.append(np.sort(np.repeat(base, reps)))                 # up to ~510 locations, a few distinct keys
        return lists
    _reduce_against_oracle(make)


def test_reduce_distinct_key_boundaries():
    """location lists with exactly D distinct (target, window) keys for D on both sides of every limit of the
    de-duplicating tail (64 / 128 / 256 keys in 1 / 2 / 4 registers per lane, more than 256: raw sort) and list
    lengths up to the wave kernel's 512, plus a few longer ones for the workgroup kernel"""
    def make(locs):
        rng = np.random.default_rng(10)
        uniq = np.unique(locs)
        lists = []
        for D in (1, 2, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 258, 300, 320, 321, 383, 384, 385, 511, 512):
            for T in sorted({D, min(512, D + 1), min(512, D + 63), min(512, 2 * D), 384 if D <= 384 else 512, 512, 600}):
                base = rng.choice(uniq, size=D, replace=False)
                extra = rng.choice(base, size=T - D) if T > D else base[:0]
                lists.append(np.sort(np.concatenate([base, extra])))
        return lists
    _reduce_against_oracle(make)

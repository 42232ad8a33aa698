"""Rows of SURVEY.md 8a in isolation on the GPU, against the reference's own vectors.

* rows 1-5: every window-split and sketch vector of tests/golden/kat.json (made by the reference's headers,
  oracle/ref_kat.cpp) goes through mcq_count_windows + mcq_sketch and must come back exactly -- a feature
  that is absent from every table is invisible to the end-to-end tests, not to this one.
* rows 7-8: the sorted match list of every fixture query, tapped on EVERY path of the engine (first wave stage
  with and without the distinct-key pass, second wave stage, workgroup kernel), against the reference's
  per-rank M dumps and against the oracle on lists long enough to reach the later stages.
"""
import importlib

import numpy as np
import pytest
import torch

from golden_util import Fixture, load_kat
from oracle import dbfile
from oracle import mc_oracle as orc

pytestmark = pytest.mark.gpu
KAT = load_kat()


@pytest.fixture(scope="module")
def eng():
    return importlib.import_module("metacache-mpi_amd.engine")


def _empty_db(eng, k, s, winlen, winstride):
    return eng.Database(np.zeros(0, np.uint32), np.zeros(1, np.uint64), np.zeros(0, np.uint64), np.zeros(0, np.uint32),
                        k=k, sketch_size=s, winlen=winlen, winstride=winstride)


def _gpu_sketches(eng, db, seqs, pad_front=0):
    """windows per sequence and the features of every window through the staged entry points"""
    dev = torch.device("cuda", 0)
    bases, seq_off = orc.pack_reads(seqs)
    raw = np.frombuffer(bases, np.uint8)
    if pad_front:                       # unaligned base pointer: the 16-bit loads of the sketch must not care
        raw = np.concatenate([np.full(pad_front, ord("G"), np.uint8), raw])
    tb = torch.from_numpy(raw.copy()).to(dev) if len(raw) else torch.zeros(1, dtype=torch.uint8, device=dev)
    to = torch.from_numpy((seq_off.astype(np.int64) + pad_front)).to(dev)
    n = len(seqs)
    win_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    db.count_windows(tb.data_ptr(), to.data_ptr(), n, win_off.data_ptr())
    torch.cuda.synchronize()
    wo = win_off.cpu().numpy()
    nw = int(wo[-1])
    feats = torch.full((max(nw, 1), db.sketch_size), -2, dtype=torch.int32, device=dev)
    nfeat = torch.full((max(nw, 1),), -2, dtype=torch.int32, device=dev)
    db.sketch(tb.data_ptr(), to.data_ptr(), n, win_off.data_ptr(), feats.data_ptr(), nfeat.data_ptr())
    torch.cuda.synchronize()
    return wo, feats.cpu().numpy().view(np.uint32), nfeat.cpu().numpy()


def test_kat_window_counts(eng):
    """row 1: for_each_window (src/dna_encoding.h:259-276) -- window counts of the reference's split vectors"""
    by_geom = {}
    for w in KAT["windows"]:
        by_geom.setdefault((w["len"], w["stride"]), []).append(w)
    for (wl, st), ws in by_geom.items():
        if wl > 128:
            continue
        db = _empty_db(eng, 1, 1, wl, st)
        seqs = [b"A" * w["n"] for w in ws]
        wo, feats, nfeat = _gpu_sketches(eng, db, seqs)
        for i, w in enumerate(ws):
            # the reference calls the consumer once even for an empty sequence; an empty window has no k-mers
            assert wo[i + 1] - wo[i] == len(w["win"]), w
            for j, (b, e) in enumerate(w["win"]):
                # k = 1, s = 1: a window of n >= 1 bases yields exactly one feature, an empty one none
                assert nfeat[wo[i] + j] == (1 if e > b else 0), (w, j)


@pytest.mark.parametrize("pad", [0, 1, 3])
def test_kat_sketches(eng, pad):
    """rows 2-5: every sketch vector of the reference (N, IUPAC, lowercase, len < k, len = k, low complexity;
    (k,s) = (16,16) (16,8) (12,16) (16,32) (8,4)) through mcq_sketch, compared feature by feature"""
    by_ks = {}
    for v in KAT["sketch"]:
        by_ks.setdefault((v["k"], v["s"]), []).append(v)
    assert len(by_ks) >= 5
    total = 0
    for (k, s), vs in by_ks.items():
        db = _empty_db(eng, k, s, 128, 113)
        seqs = [v["seq"].encode() for v in vs]
        assert max(len(x) for x in seqs) <= 128            # one window each
        wo, feats, nfeat = _gpu_sketches(eng, db, seqs, pad_front=pad)
        assert wo[-1] == len(vs)
        for i, v in enumerate(vs):
            want = v["sketch"]
            assert nfeat[i] == len(want), (k, s, v["seq"], nfeat[i], want)
            got = feats[i, :len(want)].tolist()
            assert got == want, (k, s, v["seq"])
            assert (feats[i, len(want):] == 0xFFFFFFFF).all()
            total += 1
    assert total == len(KAT["sketch"]) > 300


def test_sketch_multiwindow_vs_oracle(eng):
    """rows 1-5 on reads of many lengths (several windows, tails shorter than k, N runs): every window's sketch
    equals the oracle's (which test_oracle_kat.py pins to the reference)"""
    rng = np.random.default_rng(11)
    seqs = []
    for L in list(range(0, 40)) + [113, 127, 128, 129, 143, 144, 145, 226, 241, 242, 256, 257, 500, 1000, 1017, 4000]:
        for rep in range(3):
            a = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L)
            if rep == 1 and L > 4:
                a[rng.integers(0, L, size=max(1, L // 50))] = ord("N")
            if rep == 2 and L > 4:
                a[:L // 2] |= 0x20                          # lowercase
            seqs.append(a.tobytes())
    for (k, s) in ((16, 16), (11, 7), (16, 32), (5, 3)):
        db = _empty_db(eng, k, s, 128, 113)
        wo, feats, nfeat = _gpu_sketches(eng, db, seqs)
        for i, sq in enumerate(seqs):
            wins = orc.windows(len(sq), 128, 113)
            assert wo[i + 1] - wo[i] == len(wins), (len(sq), wins)
            for j, (b, e) in enumerate(wins):
                want = orc.sketch(sq[b:e], k, s).tolist()
                w = wo[i] + j
                assert nfeat[w] == len(want) and feats[w, :len(want)].tolist() == want, (k, s, len(sq), j)


# --------------------------------------------------------------------------------------- rows 7-8: match lists
def _dbs(eng, fx, shards=None, flags=0):
    keys, off, locs = dbfile.union_shards(shards if shards is not None else fx.shards)
    p = fx.params
    kw = dict(k=p["qk"], winlen=p["qwinlen"], winstride=p["qwinstride"], tgt_winstride=p["winstride"])
    t2t = fx.tgt2tax()
    return (eng.Database(keys, off, locs, t2t, sketch_size=p["qs"], flags=flags, **kw),
            orc.OracleDb(keys, off, locs, t2t, s=p["qs"], **kw))


def _pairs(x):
    return [[int(v >> np.uint64(32)), int(v & np.uint64(0xFFFFFFFF))] for v in x]


@pytest.mark.parametrize("tag,P", [("mini", 2), ("mini", 4), ("tie", 2)])
def test_match_lists_on_every_path_vs_reference(eng, tag, P):
    """the reference's per-rank sorted match lists (ref_query dump 'M': accumulate_matches x 2 + merge_sort) from
    the first wave stage (distinct-key pass and raw sort), the second wave stage and the workgroup kernel"""
    fx = Fixture(tag, P)
    bases, seq_off = orc.pack_reads(fx.interleaved())
    for r in range(P):
        for dbflags in (0, eng.MCQ_DB_LOCS_64, eng.MCQ_DB_LOCS_GW | eng.MCQ_DB_SLOTS_16):
            db, odb = _dbs(eng, fx, [fx.shards[r]], flags=dbflags)
            ws = eng.Workspace(db, len(fx.names), len(bases))
            for pf in (0, eng.MCQ_FORCE_RAW_SORT, eng.MCQ_NO_WAVE16, eng.MCQ_FORCE_BLOCK_PATH):
                moff, m = ws.debug_matches(bases, seq_off, True, path_flags=pf)
                for q in range(len(fx.names)):
                    M = fx.ranks["M"][str(q)][str(r)]
                    assert _pairs(m[int(moff[q]):int(moff[q + 1])]) == M, (q, r, dbflags, pf)


def test_match_lists_of_long_lists_vs_oracle(eng):
    """a hand-made table whose lists make reads of 60..2000 locations: every tap (one..eight dedup registers,
    more than 256 distinct keys, 513..1024 locations in the second wave stage, longer ones in the workgroup
    kernel) returns the oracle's sorted multiset"""
    rng = np.random.default_rng(9)
    n, L, n_tgt = 1500, 150, 700
    seqs = ["".join(rng.choice(list("ACGT"), size=L)) for _ in range(n)]
    feat_len = {}
    for i, sq in enumerate(seqs):
        per = int(rng.integers(2, 19)) if i % 10 else int(rng.integers(20, 70))
        for w0, w1 in orc.windows(L):
            for f in orc.sketch(sq[w0:w1].encode()):
                feat_len.setdefault(int(f), per)
    keys = np.array(sorted(feat_len), np.uint32)
    lens = np.array([feat_len[int(k)] for k in keys], np.int64)
    off = np.zeros(len(keys) + 1, np.uint64); off[1:] = np.cumsum(lens)
    locs = np.empty(int(off[-1]), np.uint64)
    for j in range(len(keys)):
        t = rng.integers(0, n_tgt, size=lens[j]).astype(np.uint64)
        w = rng.integers(0, 12, size=lens[j]).astype(np.uint64)       # few windows: many repeated (tgt,win)
        locs[int(off[j]):int(off[j + 1])] = np.sort((t << np.uint64(32)) | w)
    t2t = (np.arange(n_tgt) // 7).astype(np.uint32)
    rb, ro = orc.pack_reads([s.encode() for s in seqs])
    odb = orc.OracleDb(keys, off, locs, t2t)
    want = [odb.matches(s.encode()) for s in seqs]
    T = np.array([len(w) for w in want])
    assert (T <= 64).any() and ((T > 64) & (T <= 512)).any() and ((T > 512) & (T <= 1024)).any() and (T > 1024).any(), np.percentile(T, [0, 50, 100])
    for dbflags in (0, eng.MCQ_DB_LOCS_64, eng.MCQ_DB_LOCS_GW, eng.MCQ_DB_LOCS_GW | eng.MCQ_DB_BUCKETS_64):
        db = eng.Database(keys, off, locs, t2t, flags=dbflags)
        ws = eng.Workspace(db, n, n * L)
        for pf in (0, eng.MCQ_FORCE_RAW_SORT, eng.MCQ_NO_WAVE16, eng.MCQ_FORCE_BLOCK_PATH):
            moff, m = ws.debug_matches(rb, ro, False, path_flags=pf)
            assert np.array_equal(np.diff(moff.astype(np.int64)), T), (dbflags, pf)
            for q in range(n):
                got = m[int(moff[q]):int(moff[q + 1])]
                assert np.array_equal(got, want[q]), (q, dbflags, pf, T[q])


def test_unknown_query_flag_bits_are_rejected(eng):
    """a stray flag bit (e.g. a build flag reused in mcq_query_opts.flags) must fail loudly, not change results"""
    fx = Fixture("mini", 2)
    db, _ = _dbs(eng, fx)
    bases, seq_off = orc.pack_reads(fx.interleaved())
    ws = eng.Workspace(db, len(fx.names), len(bases))
    for bad in (0x1000, 0x2000, 0x10000, 0x10, 0x80000000):
        with pytest.raises(eng.McqError) as e:
            ws.query_host(bases, seq_off, True, max_cand=2, emulate_ranks=2, flags=bad)
        assert e.value.code == eng.MCQ_E_ARG

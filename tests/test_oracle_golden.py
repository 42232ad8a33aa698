"""Oracle vs the reference run in the build container (rows 6-12).

ranks.json.gz  : per-rank dumps from the reference's own accumulate_matches /
                 merge_sort / candidates code (oracle/ref_query.cpp driver)
final.json     : the reference CLI's -tophits list and classification at P ranks
"""
import numpy as np
import pytest

from golden_util import Fixture
from oracle import dbfile
from oracle import mc_oracle as orc

CASES = [("mini", 2), ("mini", 4), ("mini", 8), ("tie", 2), ("tie", 4), ("noanc", 2), ("noanc", 4), ("overpop", 2), ("overpop", 4),
         # the reference's scripted rank counts (mpiexec -n 32 / -n 64 with -maxcand 4): shards + CLI output only
         ("wide", 16), ("wide", 32), ("wide", 64)]


@pytest.fixture(scope="module", params=CASES, ids=lambda c: "%s-P%d" % c)
def fx(request):
    return Fixture(*request.param)


def _shard_db(fx, r, tgt2tax):
    s = fx.shards[r]
    locs = (s["tgt"].astype(np.uint64) << np.uint64(32)) | s["win"].astype(np.uint64)
    p = s["params"]
    return orc.OracleDb(s["keys"], s["off"], locs, tgt2tax, k=p["qk"], s=p["qs"], winlen=p["qwinlen"],
                        winstride=p["qwinstride"], tgt_winstride=p["winstride"])


def test_dbfile_params_and_lineage(fx):
    P = fx.ranks["params"]      # k s winlen winstride qwinlen qwinstride ntargets
    p = fx.params
    assert [p["k"], p["s"], p["winlen"], p["winstride"], p["qwinlen"], p["qwinstride"], fx.n_targets] == P
    for t, lin in fx.ranks["lineage"].items():
        i = fx.tax.by_id[-(int(t) + 1)]
        got = [fx.tax.taxa[j]["id"] if j != dbfile.NONE else 0 for j in fx.tax.lineage[i]]
        assert got == lin
    assert sorted((t["id"], t["rank"]) for t in fx.tax.taxa) == sorted((a, b) for a, b, _ in fx.ranks["taxa"])
    # every target is stored on exactly one rank: tgt % P (src/sketch_database.h:540)
    for r, s in enumerate(fx.shards):
        assert np.all(s["tgt"] % fx.P == r)
        assert np.all(np.diff(s["off"]).astype(np.int64) <= 254)


def test_per_rank_matches_and_candidates(fx):
    if not fx.ranks["M"]:
        pytest.skip("fixture without per-rank dumps")
    t2t = fx.tgt2tax()
    for r in range(fx.P):
        db = _shard_db(fx, r, t2t)
        bases, off = orc.pack_reads(fx.interleaved())
        cand, ncand = db.query(bases, off, paired=True, max_cand=fx.maxcand, emulate_ranks=1)
        for q in range(len(fx.r1)):
            M = fx.ranks["M"][str(q)][str(r)]
            got = db.matches(fx.r1[q], fx.r2[q])
            assert [[int(x >> np.uint64(32)), int(x & np.uint64(0xFFFFFFFF))] for x in got] == M, (q, r)
            T = fx.ranks["T"][str(q)][str(r)]
            assert db.target_cands(fx.r1[q], fx.r2[q]).tolist() == T, (q, r)
            Cx = fx.ranks["C"][str(q)][str(r)]
            mine = [[fx.tax.id_of_key(c[0]), int(c[1]), int(c[2]), int(c[3])] for c in cand[q, :ncand[q]]]
            assert mine == Cx, (q, r)


def test_final_tophits_and_classification(fx):
    keys, off, locs = dbfile.union_shards(fx.shards)
    p = fx.params
    db = orc.OracleDb(keys, off, locs, fx.tgt2tax(), k=p["qk"], s=p["qs"], winlen=p["qwinlen"],
                      winstride=p["qwinstride"], tgt_winstride=p["winstride"])
    bases, off_r = orc.pack_reads(fx.interleaved())
    cand, ncand = db.query(bases, off_r, paired=True, max_cand=fx.maxcand, emulate_ranks=fx.P, quirk_seq_drop=1)
    assert len(fx.final) == len(fx.names)
    for q, name in enumerate(fx.names):
        ref = fx.final[name]
        mine = [[fx.tax.id_of_key(c[0]), int(c[1])] for c in cand[q, :ncand[q]]]
        assert mine == ref["tophits"], (name, mine, ref)
        assert all(c[2] == 0 and c[3] == 0 for c in cand[q, :ncand[q]])
        best = orc.classify([(c[0], c[1]) for c in cand[q, :ncand[q]]], fx.tax.lineage, fx.tax.rank_of,
                            fx.hitmin, fx.hitdiff, fx.highest)
        best_id = 0 if best == dbfile.NONE else fx.tax.taxa[best]["id"]
        assert best_id == ref["best"], (name, mine, best_id, ref)


def test_fold_is_order_sensitive():
    # same data, different P => different top list (tie fixture)
    f2, f4 = Fixture("tie", 2), Fixture("tie", 4)
    n = f2.names[0]
    assert f2.final[n]["tophits"] != f4.final[n]["tophits"]
    assert f2.final[n]["best"] != f4.final[n]["best"]

"""HIP engine (through the C ABI) vs the oracle and the reference's own outputs.
Everything here needs a real MI355X."""
import importlib

import numpy as np
import pytest

from golden_util import Fixture
from oracle import dbfile
from oracle import mc_oracle as orc

pytestmark = pytest.mark.gpu

CASES = [("mini", 2), ("mini", 4), ("mini", 8), ("tie", 2), ("tie", 4), ("noanc", 2), ("noanc", 4), ("overpop", 2), ("overpop", 4),
         # the reference's scripted rank counts, -maxcand 4: P = 16 in the lanes of a wave, P = 32 / 64 in the workgroup
         # kernel's LDS lists
         ("wide", 16), ("wide", 32), ("wide", 64)]


@pytest.fixture(scope="module")
def eng():
    return importlib.import_module("metacache-mpi_amd.engine")


def _dbs(eng, fx, shards=None, flags=0):
    keys, off, locs = dbfile.union_shards(fx.shards if shards is None else shards)
    p = fx.params
    kw = dict(k=p["qk"], winlen=p["qwinlen"], winstride=p["qwinstride"], tgt_winstride=p["winstride"])
    t2t = fx.tgt2tax()
    return (eng.Database(keys, off, locs, t2t, sketch_size=p["qs"], flags=flags, **kw),
            orc.OracleDb(keys, off, locs, t2t, s=p["qs"], **kw))


def _same(cands, ncand, oc, on):
    assert np.array_equal(ncand, on), np.nonzero(ncand != on)[0][:10]
    for q in range(len(on)):
        assert np.array_equal(cands[q, :on[q]], oc[q, :on[q]]), (q, cands[q, :on[q]], oc[q, :on[q]])


@pytest.mark.parametrize("tag,P", CASES)
@pytest.mark.parametrize("block", [False, True, "raw"], ids=["wave", "block", "wave-rawsort"])
@pytest.mark.parametrize("fmt", ["loc32", "loc64", "gw", "loc32-slots16", "gw-buckets64"])
def test_final_vs_reference_cli(eng, tag, P, block, fmt):
    # location formats (bit fields in 32 / 64 bits, global window index) x table layouts (64-B buckets, 16-B slots)
    fx = Fixture(tag, P)
    dbflags = {"loc32": 0, "loc64": eng.MCQ_DB_LOCS_64, "gw": eng.MCQ_DB_LOCS_GW | eng.MCQ_DB_SLOTS_16,
               "loc32-slots16": eng.MCQ_DB_SLOTS_16, "gw-buckets64": eng.MCQ_DB_LOCS_GW | eng.MCQ_DB_BUCKETS_64}[fmt]
    db, odb = _dbs(eng, fx, flags=dbflags)
    lay = db.layout()
    assert lay["loc_format"] == (eng.MCQ_LOC_GLOBAL_WINDOW if fmt.startswith("gw") else eng.MCQ_LOC_FIELDS64 if fmt == "loc64" else eng.MCQ_LOC_FIELDS32)
    assert lay["bucket_bytes"] == (16 if "slots16" in fmt or fmt == "gw" else 64)
    bases, seq_off = orc.pack_reads(fx.interleaved())
    ws = eng.Workspace(db, len(fx.names), len(bases))
    flags = eng.MCQ_QUIRK_SEQ_DROP | {False: 0, True: eng.MCQ_FORCE_BLOCK_PATH, "raw": eng.MCQ_FORCE_RAW_SORT}[block]
    cands, ncand = ws.query_host(bases, seq_off, True, max_cand=fx.maxcand, emulate_ranks=P, flags=flags)
    for q, name in enumerate(fx.names):
        mine = [[fx.tax.id_of_key(c[0]), int(c[1])] for c in cands[q, :ncand[q]]]
        assert mine == fx.final[name]["tophits"], (name, mine, fx.final[name])
    oc, on = odb.query(bases, seq_off, True, max_cand=fx.maxcand, emulate_ranks=P, quirk_seq_drop=1)
    _same(cands, ncand, oc, on)
    st = ws.sync()
    # A table without sequence-level taxa: the P lists and the tree are one selection (any P x M in the wave stages).  With them,
    # under the quirk, lists and tree are carried out: P x M > 64 takes four list registers per lane in the first wave stage
    # (32-bit words, up to 256 list slots), else -- 64-bit words -- every query goes to the workgroup kernel (lists in its LDS)
    t2t = np.asarray(fx.tgt2tax(), np.uint32)
    by_lists = bool(np.any((t2t != 0xFFFFFFFF) & (t2t >= 0x80000000)))
    p2 = 1 << (P - 1).bit_length(); m2 = 1 << (fx.maxcand - 1).bit_length()
    all_block = block is True or (by_lists and P * fx.maxcand > 64 and (fmt == "loc64" or p2 * m2 > 256))
    assert st["n_overflow"] == len(fx.names) if all_block else st["n_overflow"] < len(fx.names)
    if not by_lists or block is True:
        return
    # ... and the same lists without the quirk (one selection) and with MCQ_FOLD_BY_LISTS (lists, no drop) agree with the oracle
    oc, on = odb.query(bases, seq_off, True, max_cand=fx.maxcand, emulate_ranks=P, quirk_seq_drop=0)
    for f2 in (0, eng.MCQ_FOLD_BY_LISTS):
        cands, ncand = ws.query_host(bases, seq_off, True, max_cand=fx.maxcand, emulate_ranks=P, flags=(flags & ~eng.MCQ_QUIRK_SEQ_DROP) | f2)
        _same(cands, ncand, oc, on)


@pytest.mark.parametrize("tag,P", [("mini", 2), ("mini", 8), ("tie", 4)])
def test_per_rank_candidates_with_positions(eng, tag, P):
    # emulate_ranks = 1 on a single reference shard: the reference's own per-rank
    # top list including window ranges (ref_query dump 'C')
    fx = Fixture(tag, P)
    bases, seq_off = orc.pack_reads(fx.interleaved())
    for r in range(P):
        db, odb = _dbs(eng, fx, [fx.shards[r]], flags=(0, eng.MCQ_DB_LOCS_64, eng.MCQ_DB_LOCS_GW | eng.MCQ_DB_SLOTS_16)[r % 3])
        ws = eng.Workspace(db, len(fx.names), len(bases))
        for flags in (0, eng.MCQ_FORCE_BLOCK_PATH, eng.MCQ_FORCE_RAW_SORT):
            cands, ncand = ws.query_host(bases, seq_off, True, max_cand=fx.maxcand, emulate_ranks=1, flags=flags)
            for q in range(len(fx.names)):
                Cx = fx.ranks["C"][str(q)][str(r)]
                mine = [[fx.tax.id_of_key(c[0]), int(c[1]), int(c[2]), int(c[3])] for c in cands[q, :ncand[q]]]
                assert mine == Cx, (q, r, flags)


@pytest.mark.parametrize("tag,P", [("mini", 4)])
def test_sorted_match_lists(eng, tag, P):
    fx = Fixture(tag, P)
    bases, seq_off = orc.pack_reads(fx.interleaved())
    for r in range(P):
        db, odb = _dbs(eng, fx, [fx.shards[r]], flags=(0, eng.MCQ_DB_LOCS_64, eng.MCQ_DB_LOCS_GW, eng.MCQ_DB_SLOTS_16)[r % 4])
        ws = eng.Workspace(db, len(fx.names), len(bases))
        moff, m = ws.debug_matches(bases, seq_off, True)
        for q in range(len(fx.names)):
            M = fx.ranks["M"][str(q)][str(r)]
            got = m[int(moff[q]):int(moff[q + 1])]
            assert [[int(x >> np.uint64(32)), int(x & np.uint64(0xFFFFFFFF))] for x in got] == M, (q, r)


def test_single_end_and_empty(eng):
    fx = Fixture("mini", 2)
    db, odb = _dbs(eng, fx)
    seqs = fx.r1 + ["", "ACGT", "N" * 200]
    bases, seq_off = orc.pack_reads(seqs)
    ws = eng.Workspace(db, len(seqs), len(bases) + 1)
    for P in (1, 2):
        cands, ncand = ws.query_host(bases, seq_off, False, max_cand=4, emulate_ranks=P)
        oc, on = odb.query(bases, seq_off, False, max_cand=4, emulate_ranks=P)
        _same(cands, ncand, oc, on)
    # empty batch
    c, n = ws.query_host(b"", np.zeros(1, np.uint64), False)
    assert len(n) == 0

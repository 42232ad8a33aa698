"""Host C++ library (include/mcq_host.h): the reference's shard files, taxon keys and
classify(), checked against the numpy restatement (oracle/dbfile.py), the reference's own
lineage dump and the reference CLI's classification."""
import importlib

import numpy as np
import pytest

from golden_util import Fixture
from oracle import dbfile
from oracle import mc_oracle as orc

CASES = [("mini", 2), ("mini", 8), ("tie", 2), ("tie", 4), ("noanc", 4), ("overpop", 2)]


@pytest.fixture(scope="module")
def host():
    pkg = importlib.import_module("metacache-mpi_amd")
    pkg.build_host()
    return importlib.import_module("metacache-mpi_amd.host")


@pytest.mark.parametrize("tag,P", CASES)
def test_shard_reader_and_keys(host, tag, P):
    fx = Fixture(tag, P)
    prefix = fx.shard_paths[0][: -len(".db_0")]
    db = host.RefDb(prefix, P)
    i, p = db.info, fx.params
    assert (i.k, i.sketch_size, i.winlen, i.winstride, i.q_sketch_size, i.q_winlen, i.q_winstride) == \
           (p["k"], p["s"], p["winlen"], p["winstride"], p["qs"], p["qwinlen"], p["qwinstride"])
    assert i.max_locs_per_feature == 254 and i.n_targets == fx.n_targets and i.n_taxa == len(fx.tax.taxa)
    keys, off, locs = db.table()
    rk, ro, rl = dbfile.union_shards(fx.shards)
    assert np.array_equal(keys, rk) and np.array_equal(off, ro) and np.array_equal(locs, rl)
    for rank in (0, 4, 6, 10):
        assert np.array_equal(db.tgt2tax(rank), fx.tax.target_keys(fx.n_targets, rank))
    # lineage of every target as the reference reports it
    for t, lin in fx.ranks["lineage"].items():
        key = int(db.tgt2tax(0)[int(t)])
        assert key & 0x80000000
        got = [db.taxon_id(host.lib().mcq_refdb_ancestor(db.h, key, r)) for r in range(21)]
        assert got == lin


def test_missing_and_bad_files(host, tmp_path):
    with pytest.raises(RuntimeError, match="can't open"):
        host.RefDb(str(tmp_path / "nope"), 2)
    bad = tmp_path / "bad.db_0"
    bad.write_bytes(b"\x00" * 64)
    with pytest.raises(RuntimeError, match="incompatible"):
        host.RefDb(str(tmp_path / "bad"), 1)


def test_rank_names(host):
    assert host.rank_from_name("species") == 4 and host.rank_from_name("Superkingdom") == 19
    assert host.rank_from_name("genome") == 0 and host.rank_from_name("bogus") == 21
    assert host.lib().mcq_default_hits_min(16) == 5 and host.lib().mcq_default_hits_min(5) == 2


@pytest.mark.parametrize("tag,P", CASES)
def test_classify_matches_reference_cli(host, tag, P):
    fx = Fixture(tag, P)
    db = host.RefDb(fx.shard_paths[0][: -len(".db_0")], P)
    keys, off, locs = db.table()
    p = fx.params
    odb = orc.OracleDb(keys, off, locs, db.tgt2tax(fx.lowest), k=p["qk"], s=p["qs"], winlen=p["qwinlen"],
                       winstride=p["qwinstride"], tgt_winstride=p["winstride"])
    bases, seq_off = orc.pack_reads(fx.interleaved())
    cand, ncand = odb.query(bases, seq_off, True, max_cand=fx.maxcand, emulate_ranks=P, quirk_seq_drop=1)
    for q, name in enumerate(fx.names):
        best = db.classify(cand[q, :ncand[q]], fx.hitmin, fx.hitdiff, fx.highest)
        assert db.taxon_id(best) == fx.final[name]["best"], (name, cand[q, :ncand[q]])


# ---- shard writer (row f1, the other direction) ----------------------------------------------------
def _shard_params(s):
    p = s["params"]
    return dict(k=p["k"], sketch_size=p["s"], winlen=p["winlen"], winstride=p["winstride"], q_k=p["qk"],
                q_sketch_size=p["qs"], q_winlen=p["qwinlen"], q_winstride=p["qwinstride"], max_locs_per_feature=p["maxlocs"])


@pytest.mark.parametrize("tag,P", [("mini", 4), ("tie", 2), ("overpop", 2)])
def test_written_shards_are_byte_identical_to_the_reference_files(host, tag, P, tmp_path):
    """parse a reference shard (numpy restatement), write it back through the C library: the same bytes"""
    fx = Fixture(tag, P)
    for r, s in enumerate(fx.shards):
        locs = (s["tgt"].astype(np.uint64) << np.uint64(32)) | s["win"].astype(np.uint64)
        out = str(tmp_path / ("w.db_%d" % r))
        host.write_shard(out, _shard_params(s), s["taxa"], s["target_count"], s["keys"], s["off"], locs)
        assert open(out, "rb").read() == open(fx.shard_paths[r], "rb").read()


def test_writer_rejects_oversized_lists(host, tmp_path):
    fx = Fixture("mini", 2)
    s = fx.shards[0]
    with pytest.raises(RuntimeError):
        host.write_shard(str(tmp_path / "x"), _shard_params(s), s["taxa"], s["target_count"], np.array([5], np.uint32),
                         np.array([0, 300], np.uint64), np.zeros(300, np.uint64))

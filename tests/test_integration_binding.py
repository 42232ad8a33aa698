"""The reference-side binding (integration/mcq_reference_binding.{h,cpp}) against the reference's OWN headers and
code -- build container only (skipped where /root/reference is absent, e.g. on the GPU box).

`make -C integration check` compiles the binding against /root/reference/src/{config,sketch_database,candidates,
query_options,sequence_io,hash_multimap}.h and links it with the reference's translation units and libmcq_hip.so /
libmcq_host.so: the type-level proof of the drop-in boundary.  The self-test driver then runs the CPU half of it:

  table  the flattened table the binding hands to mcq_db_create == the reference's in-memory hash_multimap content
         (its print_feature_map dump, and a live mc::hash_multimap flattened through its bucket interface)
  map    engine candidate lists (here: the oracle's, computed with the binding's taxon keys) -> the reference's
         classification_candidates through its real insert() -> the reference's own map_candidates_to_targets
         (classify + show_query_mapping): the lines equal what its CLI printed under mpiexec -n P (final.json)
"""
import os
import subprocess

import numpy as np
import pytest

from golden_util import Fixture
from oracle import dbfile
from oracle import mc_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "integration", "_build", "binding_selftest")
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/src") or not os.path.isdir("/opt/conda/include"),
                                reason="needs the reference (build container only)")


@pytest.fixture(scope="module")
def selftest():
    pkg = __import__("importlib").import_module("metacache-mpi_amd")
    pkg.build_hip(); pkg.build_host()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "integration"), "check"])
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "oracle", "_ref", "mpilib") + ":" + env.get("LD_LIBRARY_PATH", "")

    def run(*args):
        r = subprocess.run([BIN] + [str(a) for a in args], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, (args, r.stderr[-2000:])
        return r.stdout
    return run


def _prefix(fx):
    return fx.shard_paths[0][:-len(".db_0")]


@pytest.mark.parametrize("tag,P", [("mini", 2), ("mini", 8), ("tie", 4), ("overpop", 4), ("wide", 32)])
def test_flattened_table_is_the_reference_table(selftest, tag, P):
    fx = Fixture(tag, P)
    out = selftest("table", _prefix(fx), P).split()
    keys, off, locs = dbfile.union_shards(fx.shards)
    assert out[:2] == ["table", "ok"] and int(out[3]) == len(keys) and int(out[5]) == len(locs)


def _parse_tophits(col):
    out = []
    col = col.strip()
    if not col or col == "--":
        return out
    for tok in col.split(","):
        a, h = tok.strip().rsplit(":", 1)
        out.append([int(a), int(h)])
    return out


@pytest.mark.parametrize("tag,P", [("mini", 2), ("mini", 4), ("mini", 8), ("tie", 2), ("tie", 4), ("noanc", 4), ("overpop", 2),
                                   ("wide", 16), ("wide", 64)])
def test_candidates_through_the_references_insert_and_classify(selftest, tag, P, tmp_path):
    fx = Fixture(tag, P)
    # 1. the binding's taxon keys (make_taxon_keys on the reference's database object)
    rows = [l.split() for l in selftest("keys", _prefix(fx), fx.q["lowest"]).strip().split("\n")]
    assert [int(r[0]) for r in rows] == list(range(fx.n_targets))
    t2t = np.array([int(r[1]) for r in rows], np.uint32)
    # same partition of the targets as this repo's own host library computes (keys are opaque, ids are not)
    mine = fx.tgt2tax()
    assert [int(r[2]) for r in rows] == [fx.tax.id_of_key(int(k)) for k in mine]
    # 2. the engine's answer with those keys (the oracle stands in for the GPU here: same lists, tests/test_gpu_parity.py)
    keys, off, locs = dbfile.union_shards(fx.shards)
    p = fx.params
    odb = orc.OracleDb(keys, off, locs, t2t, k=p["qk"], s=p["qs"], winlen=p["qwinlen"], winstride=p["qwinstride"],
                       tgt_winstride=p["winstride"])
    bases, seq_off = orc.pack_reads(fx.interleaved())
    cand, ncand = odb.query(bases, seq_off, True, max_cand=fx.maxcand, emulate_ranks=P, quirk_seq_drop=1)
    f = tmp_path / "cands.txt"
    with open(f, "w") as fh:
        for q, name in enumerate(fx.names):
            fh.write("%s\t%d\t%s\n" % (name, ncand[q], " ".join("%d:%d:%d:%d" % tuple(int(x) for x in c) for c in cand[q, :ncand[q]])))
    # 3. through to_candidates (insert()) and the reference's map_candidates_to_targets
    lines = [l for l in selftest("map", _prefix(fx), fx.q["lowest"], fx.maxcand, f).split("\n") if l and not l.startswith("#")]
    got = {}
    for line in lines:
        cols = line.split("\t|\t")
        got[cols[0]] = {"tophits": _parse_tophits(cols[1]), "best": int(cols[2]) if cols[2].strip() not in ("", "--") else 0}
    assert got == fx.final


def test_query_block_is_instantiated_with_the_references_result_map():
    """gpu_engine::query_block<ResultMap> -- the one function that replaces the reference's seam -- was never
    instantiated.  binding_selftest.cpp now instantiates it explicitly with tsl::hopscotch_map<uint_least64_t,
    classification_candidates>, the container query_batched_parallel2 keeps its results in (src/querying.h:733): the symbol
    must be in the linked binary."""
    exe = os.path.join(ROOT, "integration", "_build", "binding_selftest")
    if not os.path.exists(exe):
        import pytest
        pytest.skip("integration/_build/binding_selftest is built in the build container only (make -C integration check)")
    out = subprocess.run(["nm", "-C", exe], stdout=subprocess.PIPE, text=True).stdout
    hits = [l for l in out.splitlines() if "gpu_engine::query_block<tsl::hopscotch_map<unsigned long" in l and "best_distinct_matches_in_contiguous_window_ranges" in l]
    assert hits, "no instantiation of gpu_engine::query_block with the reference's map type"

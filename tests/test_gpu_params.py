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
    dbbuild = importlib.import_module("metacache-mpi_amd.dbbuild")
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
    for kw in (dict(max_cand=0), dict(max_cand=17), dict(max_cand=16, emulate_ranks=8), dict(emulate_ranks=65)):
        with pytest.raises(eng.McqError):
            ws.query_host(b"ACGT", np.array([0, 4], np.uint64), False, **kw)
    # an empty database answers every query with no candidates
    c, n = ws.query_host(b"ACGTACGTACGTACGTACGTACGT", np.array([0, 24], np.uint64), False)
    assert n.tolist() == [0]


def test_capacity_error_is_reported_not_hidden():
    """a query whose match list exceeds the workspace's per-query capacity gets no result and
    mcq_ws_sync returns MCQ_E_CAPACITY (the reference's analogue: a FAIL log line, src/querying.h:833-847)"""
    eng = importlib.import_module("metacache-mpi_amd.engine")
    dbbuild = importlib.import_module("metacache-mpi_amd.dbbuild")
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

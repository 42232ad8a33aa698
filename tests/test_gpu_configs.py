"""BASELINE.json's configs at their stated sizes, in the test suite (not only inside bench.py).

configs[0]  3 E. coli-sized genomes (4.6 Mbp; genomes 2, 3 = genome 1 with 1 % / 5 % substitutions),
            10 k synthetic 100 bp reads: built on the GPU, queried, compared with the oracle read by read, and the
            classification is the species of origin for (almost) every read.
configs[1]  500 synthetic genomes (1.96 Gbp) in HBM, one full batch of 1 048 576 x 150 bp reads: every candidate list
            equals the oracle's (16 threads, ~1 s).
configs[3]  the same table, one full batch of 524 288 pairs of 2 x 150 bp.
"""
import importlib

import numpy as np
import pytest
import torch

from oracle import mc_oracle as orc

pytestmark = pytest.mark.gpu


def _compare(cands, ncand, oc, on, what):
    bad = np.nonzero(ncand != on)[0]
    assert len(bad) == 0, (what, "ncand differs at", bad[:5], ncand[bad[:5]], on[bad[:5]])
    mask = np.arange(cands.shape[1])[None, :] < on[:, None]
    neq = np.any((cands != oc) & mask[:, :, None], axis=(1, 2))
    bad = np.nonzero(neq)[0]
    assert len(bad) == 0, (what, "cands differ at", bad[:5], cands[bad[0]], oc[bad[0]])


def test_config0_three_ecoli_sized_genomes():
    eng = importlib.import_module("metacache-mpi_amd.engine")
    synth = importlib.import_module("metacache-mpi_amd.synth")
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(1)
    L = 4_600_000
    acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    anc = torch.randint(0, 4, (L,), generator=g, device=dev)
    parts = [anc]
    for div in (0.01, 0.05):
        mut = torch.rand(L, generator=g, device=dev) < div
        parts.append(torch.where(mut, (anc + torch.randint(1, 4, (L,), generator=g, device=dev)) & 3, anc))
    gb = torch.cat([acgt[p] for p in parts])
    goff = torch.tensor([0, L, 2 * L, 3 * L], dtype=torch.int64, device=dev)
    species = np.array([0, 1, 2], np.uint32)                                   # one species each under one genus
    reads, roff, origin = synth.sample_reads(gb, goff, 10_000, 100, 0.01, 0.0, seed=2)
    rb = reads.cpu().numpy().tobytes(); ro = roff.cpu().numpy().astype(np.uint64)
    for P in (1, 2):
        table = eng.Table(gb.data_ptr(), goff.data_ptr(), 3, emulate_ranks=P)
        keys, off, locs, _ = table.to_host()
        table.close()
        odb = orc.OracleDb(keys, off, locs, species)
        db = eng.Database(keys, off, locs, species)
        ws = eng.Workspace(db, 10_000, len(rb))
        for M in (2, 4):
            oc, on = odb.query(rb, ro, False, max_cand=M, emulate_ranks=P, threads=8)
            for qf in (0, eng.MCQ_FORCE_RAW_SORT, eng.MCQ_FORCE_BLOCK_PATH):
                cands, ncand = ws.query_host(rb, ro, False, max_cand=M, emulate_ranks=P, flags=qf)
                _compare(cands, ncand, oc, on, "configs[0] P=%d M=%d qf=%x" % (P, M, qf))
        # a 100 bp read is one window of 16 features: at 1 % read error nearly every read has hits, and the best
        # candidate (most hits, first in target order on ties) is its species of origin for most reads
        assert (on > 0).mean() > 0.99
        top = oc[:, 0, 0]
        assert (top[on > 0] == origin.cpu().numpy().astype(np.uint32)[on > 0]).mean() > 0.5


@pytest.fixture(scope="module")
def c2():
    """the bench database: 50 species x 10 strains, U[2,6] Mbp, 2 % divergence (seed 3), table built on the GPU"""
    eng = importlib.import_module("metacache-mpi_amd.engine")
    synth = importlib.import_module("metacache-mpi_amd.synth")
    dev = torch.device("cuda", 0)
    gb, goff, species = synth.make_genomes(50, 10, 2_000_000, 6_000_000, 0.02, seed=3, device=dev)
    torch.cuda.empty_cache()
    table = eng.Table(gb.data_ptr(), goff.data_ptr(), goff.numel() - 1, emulate_ranks=2)
    keys, off, locs, _ = table.to_host()
    sp = species.cpu().numpy().astype(np.uint32)
    sp32 = species.to(torch.int32).contiguous()
    db = eng.Database(None, None, None, None, device_ptrs=dict(keys=table.keys_ptr, list_off=table.list_off_ptr, locs=table.locs_ptr,
                                                               tgt2tax=sp32.data_ptr(), n_keys=table.n_keys, n_locs=table.n_locs,
                                                               n_targets=sp32.numel()))
    table.close()
    odb = orc.OracleDb(keys, off, locs, sp)
    return eng, synth, gb, goff, db, odb


def test_config1_full_batch(c2):
    eng, synth, gb, goff, db, odb = c2
    n, L = 1 << 20, 150
    assert int(goff[-1].item()) > 1_900_000_000 and goff.numel() - 1 == 500
    reads, roff, _ = synth.sample_reads(gb, goff, n, L, 0.005, 0.001, seed=1000)
    rb = reads.cpu().numpy().tobytes(); ro = roff.cpu().numpy().astype(np.uint64)
    ws = eng.Workspace(db, n, len(rb))
    oc, on = odb.query(rb, ro, False, max_cand=2, emulate_ranks=2, threads=16)
    cands, ncand = ws.query_host(rb, ro, False, max_cand=2, emulate_ranks=2)
    _compare(cands, ncand, oc, on, "configs[1] full batch")
    cands, ncand = ws.query_host(rb, ro, False, max_cand=2, emulate_ranks=2, flags=eng.MCQ_FOLD_BY_LISTS)
    _compare(cands, ncand, oc, on, "configs[1] full batch, the two lists and the fold one by one")
    st = ws.sync()
    assert st["n_queries"] == n and st["n_locations"] > 50 * n and st["n_features"] > 30 * n
    assert (on > 0).mean() > 0.99


def test_config3_full_batch_of_pairs(c2):
    eng, synth, gb, goff, db, odb = c2
    n, L = 1 << 19, 150
    reads, roff, _ = synth.sample_pairs(gb, goff, n, L, 300, 500, 0.005, 0.001, seed=1001)
    rb = reads.cpu().numpy().tobytes(); ro = roff.cpu().numpy().astype(np.uint64)
    ws = eng.Workspace(db, n, len(rb))
    oc, on = odb.query(rb, ro, True, max_cand=4, emulate_ranks=4, threads=16)
    cands, ncand = ws.query_host(rb, ro, True, max_cand=4, emulate_ranks=4)
    _compare(cands, ncand, oc, on, "configs[3] shape, full batch of pairs")


def test_config4_long_reads(c2):
    """configs[4] shape on one GPU: 16 384 ONT-like reads, lengths log-normal with mean 8 kb, 8 % substitutions, on the 2 Gbp
    table (~2 400 locations per read: every query in the workgroup kernel) -- every candidate list equals the oracle's"""
    eng, synth, gb, goff, db, odb = c2
    n = 1 << 14
    reads, roff, _ = synth.sample_long_reads(gb, goff, n, 8000, 0.08, seed=1002)
    rb = reads.cpu().numpy().tobytes(); ro = roff.cpu().numpy().astype(np.uint64)
    assert 7000 < len(rb) / n < 9000
    ws = eng.Workspace(db, n, len(rb))
    for P, M in ((2, 2), (8, 4)):
        oc, on = odb.query(rb, ro, False, max_cand=M, emulate_ranks=P, threads=16)
        cands, ncand = ws.query_host(rb, ro, False, max_cand=M, emulate_ranks=P)
        _compare(cands, ncand, oc, on, "configs[4] shape P=%d M=%d" % (P, M))
    st = ws.sync()
    assert st["n_overflow"] == n and st["n_locations"] > 1500 * n

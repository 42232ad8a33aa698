"""BASELINE.json configs[2] at its stated scale, on ONE GPU: the RefSeq-scale table (a file of its own: the other modules'
fixtures -- tens of GB of tables -- are gone when it runs)."""
import glob
import importlib
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

from oracle import mc_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compare(cands, ncand, oc, on, what):
    bad = np.nonzero(ncand != on)[0]
    assert len(bad) == 0, (what, "ncand differs at", bad[:5], ncand[bad[:5]], on[bad[:5]])
    mask = np.arange(cands.shape[1])[None, :] < on[:, None]
    neq = np.any((cands != oc) & mask[:, :, None], axis=(1, 2))
    bad = np.nonzero(neq)[0]
    assert len(bad) == 0, (what, "cands differ at", bad[:5], cands[bad[0]], oc[bad[0]])


def _spot_check_against_a_one_piece_build(eng, db, spot, n_targets, tw):
    """Every feature of the 40 picked targets' own one-piece table: the locations the big handle returns for it, filtered to those
    targets, must be exactly the one-piece build's list (target ids mapped back) -- part boundaries, the long-list area beyond 2^32
    B, the global-window offsets of targets all over the table.  A feature may be missing from the big table only because
    -remove-overpopulated-features took it out (more than 254 locations in the whole database): rare, and then missing whole."""
    from oracle import subtable                                     # (only for its decoding of the handle's words)
    pick, keys, off, locs = spot
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    step = max(1, len(keys) // 200_000)
    sel = np.arange(0, len(keys), step)
    k32 = torch.from_numpy(keys[sel].view(np.int32).copy()).to(dev)
    n = k32.numel()
    lens = torch.zeros(n, dtype=torch.int32, device=dev)
    db.lookup_count(k32.data_ptr(), n, lens.data_ptr(), None, st)
    ooff = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens.to(torch.int64), 0, out=ooff[1:])
    native = torch.zeros(int(ooff[-1].item()) + 1, dtype=torch.int32, device=dev)
    db.lookup_gather(k32.data_ptr(), n, ooff.data_ptr(), native.data_ptr(), stream=st)
    torch.cuda.synchronize()
    go = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(tw.to(torch.int64), 0)])
    w = native[:-1].to(torch.int64) & 0xFFFFFFFF
    t = torch.searchsorted(go, w, right=True) - 1
    big = ((t << 32) | (w - go[t])).cpu().numpy().view(np.uint64)
    lens_h = lens.cpu().numpy().astype(np.int64)
    o = np.zeros(n + 1, np.int64); o[1:] = np.cumsum(lens_h)
    back = {p: i for i, p in enumerate(pick)}
    is_pick = np.zeros(n_targets, bool); is_pick[pick] = True
    remap = np.zeros(n_targets, np.uint64); remap[pick] = np.arange(len(pick), dtype=np.uint64)
    missing = 0
    for j, ki in enumerate(sel):
        want = locs[int(off[ki]):int(off[ki + 1])]
        got = big[o[j]:o[j + 1]]
        if len(got) == 0:
            missing += 1
            continue
        gt = (got >> np.uint64(32)).astype(np.int64)
        m = is_pick[gt]
        mine = (remap[gt[m]] << np.uint64(32)) | (got[m] & np.uint64(0xFFFFFFFF))
        assert np.array_equal(np.sort(mine), want), (int(keys[ki]), mine[:8], want[:8])
    assert missing < 0.01 * len(sel), (missing, len(sel))


def test_config2_refseq_scale_table_on_one_gpu():
    """configs[2] shape on ONE GPU: 2 600 species x 10 strains of 2-6 Mbp (>= 100 Gbp), every genome two
    sequences (>= 2^15 targets), one 16 Mbp chromosome (> 2^17 windows), built with -remove-overpopulated-features in
    feature-hash parts; the handle must come out in the global-window form (the (target, window) bit fields need 34 bits).
    A batch of 150 bp reads and a batch of 2 x 150 bp pairs against the CPU oracle, which is given the part of the table
    each batch can touch (oracle/subtable.py): bit-exact.  Needs ~250 GB of HBM: skipped on a smaller part."""
    eng = importlib.import_module("metacache-mpi_amd.engine")
    synth = importlib.import_module("metacache-mpi_amd.synth")
    from oracle import subtable
    dev = torch.device("cuda", 0)
    torch.cuda.empty_cache()
    if torch.cuda.mem_get_info(dev)[0] < 250e9:
        pytest.skip("needs 250 GB of free HBM")
    gb, goff, species = synth.make_genomes_big(2600, 10, 2_000_000, 6_000_000, 0.02, seed=3, device=dev, extra_genome=16_000_000)
    goff, species = synth.split_targets(goff, species, 2, keep_last_whole=True)
    n_targets = species.numel()
    tw = synth.window_counts(goff)
    assert int(goff[-1].item()) >= 100e9 and n_targets >= (1 << 15) and int(tw.max().item()) > (1 << 17)
    n = 1 << 18
    reads, roff, _ = synth.sample_reads(gb, goff, n, 150, 0.005, 0.001, seed=2000)
    pairs, poff, _ = synth.sample_pairs(gb, goff, n // 2, 150, 300, 500, 0.005, 0.001, seed=2001)
    # an INDEPENDENT derivation of some of the table's lists (the oracle's sub-tables are read back through the handle itself): 40
    # targets from all over the database -- among them the two halves of a part-boundary genome and the 16 Mbp chromosome --
    # sketched and merged by the one-piece build (mcq_build_table: pinned to the reference's shard files, tests/test_gpu_dbbuild.py)
    pick = sorted(set([0, 1, 2, 3, n_targets - 1, n_targets - 2] + [int(x) for x in torch.randint(0, n_targets, (34,), generator=torch.Generator().manual_seed(5)).tolist()]))
    sub_off = [0]
    pieces = []
    for t in pick:
        a, b_ = int(goff[t].item()), int(goff[t + 1].item())
        pieces.append(gb[a:b_]); sub_off.append(sub_off[-1] + (b_ - a))
    sub_bases = torch.cat(pieces).contiguous()
    sub_goff = torch.tensor(sub_off, dtype=torch.int64, device=dev)
    tb = eng.Table(sub_bases.data_ptr(), sub_goff.data_ptr(), len(pick), emulate_ranks=1)
    spot = (pick,) + tb.to_host()[:3]
    tb.close()
    del sub_bases, pieces
    parts = eng.Parts(gb.data_ptr(), goff.data_ptr(), n_targets, emulate_ranks=2, flags=eng.MCQ_BUILD_REMOVE_OVERPOPULATED)
    assert parts.n_parts > 1 and parts.n_locs > 1e10
    del gb
    torch.cuda.empty_cache()
    sp32 = species.to(torch.int32).contiguous()
    db = parts.database(sp32.data_ptr())
    parts.close()
    lay = db.layout()
    assert lay["loc_format"] == eng.MCQ_LOC_GLOBAL_WINDOW and lay["loc_bytes"] == 4 and lay["bucket_bytes"] == 16, lay
    _spot_check_against_a_one_piece_build(eng, db, spot, n_targets, tw)
    sp = species.cpu().numpy().astype(np.uint32)
    st = torch.cuda.current_stream(dev).cuda_stream
    for (rd, ro_t, n_seqs, paired) in ((reads, roff, n, False), (pairs, poff, n, True)):
        nq = n_seqs // 2 if paired else n_seqs
        ws = eng.Workspace(db, nq, rd.numel())
        cands = torch.zeros((nq, 2, 4), dtype=torch.int32, device=dev); ncand = torch.zeros(nq, dtype=torch.int32, device=dev)
        ws.query_device(rd.data_ptr(), ro_t.data_ptr(), n_seqs, paired, cands.data_ptr(), ncand.data_ptr(), max_cand=2, emulate_ranks=2, stream=st)
        stats = ws.sync(st)
        k_, o_, l_ = subtable.batch_subtable(eng, db, rd.data_ptr(), ro_t.data_ptr(), n_seqs, dev, tw)
        odb = orc.OracleDb(k_, o_, l_, sp)
        oc, on = odb.query(rd.cpu().numpy().tobytes(), ro_t.cpu().numpy().astype(np.uint64), paired, max_cand=2, emulate_ranks=2, threads=16)
        _compare(cands.cpu().numpy().view(np.uint32), ncand.cpu().numpy().view(np.uint32), oc, on, "configs[2] shape on one GPU, paired=%d" % paired)
        assert stats["n_locations"] > 500 * nq and stats["n_two_class"] > nq // 2, stats        # ~900 locations per read: the two-class tails
        ws.close()
        # ... and through the path that runs this configuration on more than one GPU: mcq_shard_* (one rank: a rank's own blocks
        # never travel), whose home side runs the SH instantiations of the two-class kernels on the lists the owner side served.
        # No capacity is given: the exact mode sizes the location blocks by what this table delivers (~900 per read).
        sh = eng.Shard(db, 1, 0, max_queries=nq, max_bases=rd.numel(), max_seqs=n_seqs)
        for rep in range(2):            # exact (first batch of the context), then padded at the learned block sizes
            cands.zero_(); ncand.zero_()
            sh.query(rd.data_ptr(), ro_t.data_ptr(), n_seqs, paired, cands.data_ptr(), ncand.data_ptr(), max_cand=2, emulate_ranks=2, stream=st)
            sst = sh.sync(st)
            _compare(cands.cpu().numpy().view(np.uint32), ncand.cpu().numpy().view(np.uint32), oc, on, "configs[2] through mcq_shard_*, paired=%d rep=%d" % (paired, rep))
            assert sst["n_locations"] == stats["n_locations"] and sst["n_two_class"] > nq // 2, (sst, stats)
        assert sh.caps()[1] > 500 * nq
        sh.close()
    db.close()


def test_config2_through_the_sharded_path_two_ranks():
    """configs[2] as north_star runs it, rehearsed at world 2 on the one GPU of the box: the same RefSeq-scale table, one
    feature-hash-range shard per rank (~38 GB each), mcq_shard_* with its blocks exchanged through gloo, every rank its own
    150 bp reads and 2 x 150 bp pairs against the oracle on the sub-table read back from both shards (tests/shard_refseq_worker.py)."""
    dev = torch.device("cuda", 0)
    torch.cuda.empty_cache()
    if torch.cuda.mem_get_info(dev)[0] < 250e9:
        pytest.skip("needs 250 GB of free HBM")
    with tempfile.TemporaryDirectory() as d:
        outp = os.path.join(d, "res")
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="8", HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", "29981", os.path.join(ROOT, "tests", "shard_refseq_worker.py"), outp]
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
        assert r.returncode == 0, r.stdout[-3000:]
        files = sorted(glob.glob(outp + ".[0-9].npz"))
        assert len(files) == 2
        for f in files:
            z = np.load(f)
            assert bool(z["ok"][0]), f
            for nq, n_locs, n_two, n_ovf in z["res"]:
                assert n_locs > 500 * nq and n_two > nq // 2, (f, z["res"])
            assert z["xb"][3] > 0            # locations travelled to the other rank

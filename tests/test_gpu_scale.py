"""At-scale parity on seeded synthetic data: HIP engine vs the oracle on a database big
enough to have long location lists, many targets per read and overflowing queries."""
import importlib

import numpy as np
import pytest
import torch

from oracle import mc_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world():
    eng = importlib.import_module("metacache-mpi_amd.engine")
    dbbuild = importlib.import_module("dbbuild_torch")
    synth = importlib.import_module("metacache-mpi_amd.synth")
    dev = torch.device("cuda", 0)
    gb, goff, species = synth.make_genomes(6, 12, 200_000, 400_000, 0.02, seed=5, device=dev)
    out = {}
    for P in (1, 2, 4):
        keys, off, locs, _ = dbbuild.build_table(gb, goff, emulate_ranks=P)
        # P = 1: global-window locations in 16-B slots; P = 2: what the handle picks (bit fields, 64-B buckets); P = 4: 64-bit
        db = dbbuild.make_database(keys, off, locs, species,
                                   flags={1: eng.MCQ_DB_LOCS_GW | eng.MCQ_DB_SLOTS_16, 2: 0, 4: eng.MCQ_DB_LOCS_64}[P])
        odb = orc.OracleDb(keys.cpu().numpy().astype(np.uint32), off.cpu().numpy().astype(np.uint64),
                           locs.cpu().numpy().astype(np.uint64), species.cpu().numpy().astype(np.uint32))
        out[P] = (db, odb)
        if P == 2:
            # the same table with every seventh target at sequence level (no taxon of its own: key bit 31, src/candidates.h:240):
            # under MCQ_QUIRK_SEQ_DROP the P lists and the tree are carried out one by one, without it the one selection
            sp = species.cpu().numpy().astype(np.uint32)
            t = np.arange(len(sp), dtype=np.uint32)
            sq = np.where(t % 7 == 3, np.uint32(0x80000000) | t, sp).astype(np.uint32)
            out["seq"] = (dbbuild.make_database(keys, off, locs, torch.from_numpy(sq.astype(np.int64)).to(dev)),
                          orc.OracleDb(keys.cpu().numpy().astype(np.uint32), off.cpu().numpy().astype(np.uint64),
                                       locs.cpu().numpy().astype(np.uint64), sq))
    return eng, synth, gb, goff, out


def _compare(cands, ncand, oc, on, what):
    bad = np.nonzero(ncand != on)[0]
    assert len(bad) == 0, (what, "ncand differs at", bad[:5], ncand[bad[:5]], on[bad[:5]])
    mask = np.arange(cands.shape[1])[None, :] < on[:, None]
    neq = np.any((cands != oc) & mask[:, :, None], axis=(1, 2))
    bad = np.nonzero(neq)[0]
    assert len(bad) == 0, (what, "cands differ at", bad[:5], cands[bad[0]], oc[bad[0]])


@pytest.mark.parametrize("P,M", [(1, 4), (2, 2), (2, 8), (4, 4), (4, 16)])
def test_short_reads(world, P, M):
    eng, synth, gb, goff, dbs = world
    db, odb = dbs[P]
    n, L = 60000, 150
    reads, off, _ = synth.sample_reads(gb, goff, n, L, 0.01, 0.002, seed=11 + P)
    ws = eng.Workspace(db, n, n * L)
    rb = reads.cpu().numpy().tobytes(); ro = off.cpu().numpy().astype(np.uint64)
    cands, ncand = ws.query_host(rb, ro, False, max_cand=M, emulate_ranks=P)
    oc, on = odb.query(rb, ro, False, max_cand=M, emulate_ranks=P, threads=8)
    _compare(cands, ncand, oc, on, "single-end P=%d M=%d" % (P, M))
    st = ws.sync()
    assert st["n_locations"] > 0
    # the wave path without the de-duplicating pass (what 64-bit keys and T > 192 take)
    cands, ncand = ws.query_host(rb, ro, False, max_cand=M, emulate_ranks=P, flags=eng.MCQ_FORCE_RAW_SORT)
    _compare(cands, ncand, oc, on, "single-end raw sort P=%d M=%d" % (P, M))
    # wide window ranges (insertSizeMax far above the read length): long lower-bound searches
    for ins in (700, 5000):
        cands, ncand = ws.query_host(rb, ro, False, max_cand=M, emulate_ranks=P, insert_size_max=ins)
        oc2, on2 = odb.query(rb, ro, False, max_cand=M, emulate_ranks=P, insert_size_max=ins, threads=8)
        _compare(cands, ncand, oc2, on2, "single-end insert_size_max=%d P=%d M=%d" % (ins, P, M))
    # paired: the same reads taken as mates
    cands, ncand = ws.query_host(rb, ro, True, max_cand=M, emulate_ranks=P)
    oc, on = odb.query(rb, ro, True, max_cand=M, emulate_ranks=P, threads=8)
    _compare(cands, ncand, oc, on, "paired P=%d M=%d" % (P, M))


def test_long_reads_take_the_block_path(world):
    eng, synth, gb, goff, dbs = world
    db, odb = dbs[2]
    n, L = 300, 6000
    reads, off, _ = synth.sample_reads(gb, goff, n, L, 0.05, 0.001, seed=99)
    ws = eng.Workspace(db, n, n * L)
    rb = reads.cpu().numpy().tobytes(); ro = off.cpu().numpy().astype(np.uint64)
    cands, ncand = ws.query_host(rb, ro, False, max_cand=4, emulate_ranks=2)
    oc, on = odb.query(rb, ro, False, max_cand=4, emulate_ranks=2, threads=8)
    _compare(cands, ncand, oc, on, "long")
    st = ws.sync()
    assert st["n_overflow"] == n


@pytest.mark.parametrize("L,n,sub", [(1000, 1500, 0.05), (2500, 800, 0.06), (9000, 400, 0.10), (20000, 200, 0.14), (45000, 80, 0.18)])
def test_long_reads_of_every_length(world, L, n, sub):
    """The workgroup kernel over read lengths from 1 kb (numWindows 10) to 45 kb (400), lists in LDS and in global scratch, P x M
    from one entry to 16 and a non-power-of-two P, against the oracle.  (Written in r04 for a counting tail -- window histograms
    instead of sort + sweep -- that was parity-green on all of it and 8 % slower: DESIGN.md section 11.)"""
    eng, synth, gb, goff, dbs = world
    db, odb = dbs[2]
    reads, off, _ = synth.sample_reads(gb, goff, n, L, sub, 0.001, seed=4242 + L)
    ws = eng.Workspace(db, n, n * L)
    rb = reads.cpu().numpy().tobytes(); ro = off.cpu().numpy().astype(np.uint64)
    for P, M in ((2, 1), (2, 2), (4, 4), (8, 2), (16, 16), (3, 4)):
        oc, on = odb.query(rb, ro, False, max_cand=M, emulate_ranks=P, threads=8)
        cands, ncand = ws.query_host(rb, ro, False, max_cand=M, emulate_ranks=P)
        _compare(cands, ncand, oc, on, "L=%d P=%d M=%d" % (L, P, M))
        assert ws.sync()["n_overflow"] == n


def test_db_roundtrip_lookup(world):
    """every key's list comes back from the GPU table exactly (mcq_lookup_count/_gather)"""
    eng, synth, gb, goff, dbs = world
    dbbuild = importlib.import_module("dbbuild_torch")
    dev = torch.device("cuda", 0)
    keys, off, locs, _ = dbbuild.build_table(gb, goff, emulate_ranks=2)
    species = torch.zeros(goff.numel() - 1, dtype=torch.int64, device=dev)
    # global-window form: windows per target = 1 + its largest window id (what the handle derives), offsets = their prefix sums
    l_t, l_w = (locs >> 32), (locs & 0xFFFFFFFF)
    ext = torch.zeros(goff.numel() - 1, dtype=torch.int64, device=dev).scatter_reduce(0, l_t, l_w + 1, "amax")
    gw_off = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(ext, 0)])
    for n_shards, dbflags in ((1, 0), (3, 0), (1, eng.MCQ_DB_LOCS_GW), (3, eng.MCQ_DB_LOCS_GW | eng.MCQ_DB_SLOTS_16), (2, eng.MCQ_DB_SLOTS_16)):
        total = 0
        for sid in range(n_shards):
            db = dbbuild.make_database(keys, off, locs, species, n_shards=n_shards, shard_id=sid, flags=dbflags)
            k32 = torch.where(keys >= (1 << 31), keys - (1 << 32), keys).to(torch.int32).contiguous()
            # plus some absent keys and the reserved value
            probe = torch.cat([k32, torch.tensor([-1, 5, 7], dtype=torch.int32, device=dev)])
            n = probe.numel()
            lens = torch.zeros(n, dtype=torch.int32, device=dev)
            st = torch.cuda.current_stream(dev).cuda_stream
            db.lookup_count(probe.data_ptr(), n, lens.data_ptr(), None, st)
            ooff = torch.zeros(n + 1, dtype=torch.int64, device=dev)
            torch.cumsum(lens.to(torch.int64), 0, out=ooff[1:])
            native = torch.zeros(int(ooff[-1].item()) + 1, dtype=torch.int32 if db.loc_bytes() == 4 else torch.int64, device=dev)
            db.lookup_gather(probe.data_ptr(), n, ooff.data_ptr(), native.data_ptr(), stream=st)
            torch.cuda.synchronize()
            if db.layout()["loc_format"] == eng.MCQ_LOC_GLOBAL_WINDOW:        # gw_off[tgt] + win  ->  (tgt << 32) | win
                assert dbflags & eng.MCQ_DB_LOCS_GW and db.loc_bytes() == 4
                w = native.to(torch.int64) & 0xFFFFFFFF
                t = torch.searchsorted(gw_off, w, right=True) - 1
                out = (t << 32) | (w - gw_off[t])
            elif db.loc_bytes() == 4:        # (tgt << win_bits) | win  ->  (tgt << 32) | win
                wb = db.win_bits(); w = native.to(torch.int64) & 0xFFFFFFFF
                out = ((w >> wb) << 32) | (w & ((1 << wb) - 1))
            else:
                out = native
            own = torch.tensor([eng.owner(int(k) & 0xFFFFFFFF, n_shards) == sid for k in keys[:2000].tolist()])
            exp_len = (off[1:] - off[:-1])
            got = lens[:keys.numel()].to(torch.int64)
            assert torch.equal(got[:2000][own.to(dev)], exp_len[:2000][own.to(dev)])
            assert torch.all((got == exp_len) | (got == 0))
            mine = got > 0
            total += int(mine.sum().item())
            # gathered lists equal the source lists of the owned keys
            src_idx = torch.repeat_interleave(torch.arange(keys.numel(), device=dev)[mine], exp_len[mine])
            assert out[:-1].numel() == src_idx.numel()
            starts = off[:-1][mine]
            within = torch.arange(src_idx.numel(), device=dev) - torch.repeat_interleave(ooff[:-1][:keys.numel()][mine], exp_len[mine])
            src = locs[torch.repeat_interleave(starts, exp_len[mine]) + within]
            assert torch.equal(out[:-1], src)
        assert total == keys.numel()


@pytest.mark.parametrize("P,M", [(3, 2), (5, 4), (6, 4), (8, 4), (8, 8), (16, 4), (32, 2), (64, 1),
                                 (32, 4), (64, 4), (16, 16), (8, 16), (33, 3), (64, 16), (24, 7)])
def test_fold_orders_of_other_rank_counts(world, P, M):
    """tree-fold schedules with several edges per round and the reference's non-power-of-two behaviour; from
    pow2ceil(P) x M > 64 on (the reference's -n 32 / -n 64 with -maxcand 4) the lists live in the workgroup kernel's LDS"""
    eng, synth, gb, goff, dbs = world
    db, odb = dbs[2]
    n, L = 20000, 150
    reads, off, _ = synth.sample_reads(gb, goff, n, L, 0.01, 0.002, seed=300 + P)
    ws = eng.Workspace(db, n, n * L)
    rb = reads.cpu().numpy().tobytes(); ro = off.cpu().numpy().astype(np.uint64)
    for quirk in (0, 1):
        oc, on = odb.query(rb, ro, False, max_cand=M, emulate_ranks=P, quirk_seq_drop=quirk, threads=8)
        for flags in (0, eng.MCQ_FORCE_RAW_SORT, eng.MCQ_FOLD_BY_LISTS, eng.MCQ_FOLD_BY_LISTS | eng.MCQ_FORCE_RAW_SORT, eng.MCQ_FORCE_BLOCK_PATH):
            cands, ncand = ws.query_host(rb, ro, False, max_cand=M, emulate_ranks=P,
                                         flags=flags | (eng.MCQ_QUIRK_SEQ_DROP if quirk else 0))
            _compare(cands, ncand, oc, on, "P=%d M=%d quirk=%d flags=%x" % (P, M, quirk, flags))


@pytest.mark.parametrize("P,M", [(2, 2), (4, 4), (5, 4), (8, 4), (16, 4), (32, 4), (64, 16)])
def test_sequence_level_taxa_and_the_wire_quirk(world, P, M):
    """every seventh target at sequence level: with MCQ_QUIRK_SEQ_DROP a non-root rank's sequence-level entry is dropped when it
    is sent, after it has held a slot of its rank's list (the lists and the tree level by level); without the flag the whole
    tree is one selection.  Wave paths, raw sort and the workgroup kernel; 150-base reads and 700-base ones."""
    eng, synth, gb, goff, dbs = world
    db, odb = dbs["seq"]
    for L, n in ((150, 12000), (700, 3000)):
        reads, off, _ = synth.sample_reads(gb, goff, n, L, 0.01, 0.002, seed=700 + P + L)
        ws = eng.Workspace(db, n, n * L)
        rb = reads.cpu().numpy().tobytes(); ro = off.cpu().numpy().astype(np.uint64)
        differs = 0
        base = None
        for quirk in (0, 1):
            oc, on = odb.query(rb, ro, False, max_cand=M, emulate_ranks=P, quirk_seq_drop=quirk, threads=8)
            if quirk == 0: base = (oc.copy(), on.copy())
            else: differs = int(np.sum(on != base[1])) + int(np.sum(np.any(oc[:, :, :2] != base[0][:, :, :2], axis=(1, 2))))
            for flags in (0, eng.MCQ_FORCE_RAW_SORT, eng.MCQ_FORCE_BLOCK_PATH, eng.MCQ_FOLD_BY_LISTS):
                cands, ncand = ws.query_host(rb, ro, False, max_cand=M, emulate_ranks=P,
                                             flags=flags | (eng.MCQ_QUIRK_SEQ_DROP if quirk else 0))
                _compare(cands, ncand, oc, on, "seq-level L=%d P=%d M=%d quirk=%d flags=%x" % (L, P, M, quirk, flags))
        assert differs > 0, "the quirk changes nothing on this table: the test does not test it"


def test_many_strains_cross_every_list_size_boundary():
    """36 strains per species: 300-700 locations and 80-160 distinct (target, window) keys per read, so single-end
    reads cross the dedup limits (384 locations, 128 distinct keys), the wave kernel's raw-sort sizes and its
    512-location limit into the workgroup kernel -- all in one batch, against the oracle"""
    eng = importlib.import_module("metacache-mpi_amd.engine")
    dbbuild = importlib.import_module("dbbuild_torch")
    synth = importlib.import_module("metacache-mpi_amd.synth")
    dev = torch.device("cuda", 0)
    gb, goff, species = synth.make_genomes(3, 36, 100_000, 140_000, 0.012, seed=77, device=dev)
    keys, off, locs, _ = dbbuild.build_table(gb, goff, emulate_ranks=2)
    odb = orc.OracleDb(keys.cpu().numpy().astype(np.uint32), off.cpu().numpy().astype(np.uint64),
                       locs.cpu().numpy().astype(np.uint64), species.cpu().numpy().astype(np.uint32))
    n, L = 30000, 150
    reads, roff, _ = synth.sample_reads(gb, goff, n, L, 0.004, 0.001, seed=5)
    rb = reads.cpu().numpy().tobytes(); ro = roff.cpu().numpy().astype(np.uint64)
    oc, on, st = odb.query(rb, ro, False, max_cand=4, emulate_ranks=2, threads=8, want_stats=True)
    for flags in (0, eng.MCQ_DB_LOCS_64, eng.MCQ_DB_LOCS_GW | eng.MCQ_DB_BUCKETS_64):
        db = dbbuild.make_database(keys, off, locs, species, flags=flags)
        ws = eng.Workspace(db, n, n * L)
        for qf in (0, eng.MCQ_FORCE_RAW_SORT, eng.MCQ_NO_WAVE16, eng.MCQ_NO_TWO_CLASS):
            cands, ncand = ws.query_host(rb, ro, False, max_cand=4, emulate_ranks=2, flags=qf)
            _compare(cands, ncand, oc, on, "many strains flags=%x qf=%x" % (flags, qf))
        s = ws.sync()
        moff, m = ws.debug_matches(rb[:2000 * L], ro[:2001], False)
        T = np.diff(moff.astype(np.int64))
        assert T.max() > 512 and (T <= 384).any() and ((T > 384) & (T <= 512)).any(), np.percentile(T, [1, 25, 50, 75, 99])
        assert 0 < s["n_overflow"] < n, s         # some reads went to the workgroup kernel, most did not


def test_crafted_lists_cross_the_distinct_key_limits():
    """A table made by hand: the features of random reads get lists of mostly distinct random (target, window)
    locations whose length depends on the read, so the fused kernel sees 60..512 locations with about as many
    distinct keys -- one, two and four sort registers, more than 256 distinct keys (the attempt fails, the raw list is
    sorted), lists the wave then sends to the raw sort without trying, 513..1024 locations for the second wave stage
    (16 keys per lane; MCQ_NO_WAVE16 sends them to the workgroups instead) and longer ones for the workgroup kernel."""
    eng = importlib.import_module("metacache-mpi_amd.engine")
    rng = np.random.default_rng(21)
    n, L, n_tgt = 6000, 150, 3000
    seqs = ["".join(rng.choice(list("ACGT"), size=L)) for _ in range(n)]
    feat_len = {}
    for i, sq in enumerate(seqs):
        per = int(rng.integers(2, 19)) if i % 40 else int(rng.integers(20, 80))    # 32 features x 2..18 locations; a few x 20..79
        for w0, w1 in orc.windows(L):
            for f in orc.sketch(sq[w0:w1].encode()):
                feat_len.setdefault(int(f), per)
    keys = np.array(sorted(feat_len), np.uint32)
    lens = np.array([feat_len[int(k)] for k in keys], np.int64)
    off = np.zeros(len(keys) + 1, np.uint64); off[1:] = np.cumsum(lens)
    locs = np.empty(int(off[-1]), np.uint64)
    for j in range(len(keys)):
        t = rng.integers(0, n_tgt, size=lens[j]).astype(np.uint64)
        w = rng.integers(0, 40, size=lens[j]).astype(np.uint64)
        locs[int(off[j]):int(off[j + 1])] = np.sort((t << np.uint64(32)) | w)
    t2t = (np.arange(n_tgt) // 7).astype(np.uint32)
    rb, ro = orc.pack_reads([s.encode() for s in seqs])
    odb = orc.OracleDb(keys, off, locs, t2t)
    for flags in (0, eng.MCQ_DB_LOCS_64, eng.MCQ_DB_LOCS_GW):
        db = eng.Database(keys, off, locs, t2t, flags=flags)
        ws = eng.Workspace(db, n, n * L)
        for P, M in ((2, 2), (4, 4), (1, 3)):
            oc, on = odb.query(rb, ro, False, max_cand=M, emulate_ranks=P, threads=8)
            for qf in (0, eng.MCQ_FORCE_RAW_SORT, eng.MCQ_NO_WAVE16, eng.MCQ_NO_TWO_CLASS):
                cands, ncand = ws.query_host(rb, ro, False, max_cand=M, emulate_ranks=P, flags=qf)
                _compare(cands, ncand, oc, on, "crafted lists flags=%x P=%d M=%d qf=%x" % (flags, P, M, qf))
                tc = ws.sync()["n_two_class"]
                assert (tc == 0) if (qf & eng.MCQ_NO_TWO_CLASS or flags == eng.MCQ_DB_LOCS_64) else True, (qf, tc)    # (the workgroup kernel has the tail too)
        s = ws.sync()
        moff, m = ws.debug_matches(rb, ro, False)
        T = np.diff(moff.astype(np.int64))
        starts = moff[:-1].astype(np.int64)
        D = np.array([len(np.unique(m[a:a + t])) for a, t in zip(starts, T)])
        for lo, hi in ((1, 64), (65, 128), (129, 256)):
            assert ((D >= lo) & (D <= hi) & (T <= 512)).any(), (lo, hi)
        assert ((D > 256) & (T <= 512)).sum() > 500, (np.percentile(T, [1, 50, 99]), np.percentile(D, [1, 50, 99]))
        assert ((T > 512) & (T <= 1024)).sum() > 50 and (T > 1024).sum() > 50, np.percentile(T, [50, 90, 99, 100])
        assert 0 < s["n_overflow"] < n


@pytest.mark.parametrize("P,M", [(1, 4), (2, 2), (4, 4)])
def test_wide_reads_take_the_second_wave_stage(world, P, M):
    """500 bp single-end reads (5 windows, 80 features) and 2 x 250 bp pairs (6 windows, 96 features): more than one
    feature per lane, so the first wave stage hands them to the second one (two features per lane) -- or, with 64-bit
    locations (P = 4 here) or MCQ_NO_WAVE16, to the workgroup kernel.  Same results as the oracle either way."""
    eng, synth, gb, goff, dbs = world
    db, odb = dbs[P]
    n = 20000
    for paired in (False, True):
        if paired:
            reads, off, _ = synth.sample_pairs(gb, goff, n // 2, 250, 500, 700, 0.01, 0.002, seed=31 + P)
            nq = n // 2
        else:
            reads, off, _ = synth.sample_reads(gb, goff, n, 500, 0.01, 0.002, seed=41 + P)
            nq = n
        rb = reads.cpu().numpy().tobytes(); ro = off.cpu().numpy().astype(np.uint64)
        oc, on = odb.query(rb, ro, paired, max_cand=M, emulate_ranks=P, threads=8)
        ws = eng.Workspace(db, nq, len(rb))
        for qf in (0, eng.MCQ_NO_WAVE16, eng.MCQ_FORCE_RAW_SORT):
            cands, ncand = ws.query_host(rb, ro, paired, max_cand=M, emulate_ranks=P, flags=qf)
            _compare(cands, ncand, oc, on, "wide reads paired=%d P=%d M=%d qf=%x" % (paired, P, M, qf))
            assert ws.sync()["n_overflow"] == nq          # every query left the first stage


def test_ragged_lengths_cross_every_window_count(world):
    """One batch with reads of 15..1100 bases, every window-count boundary included (128/129, 241/242, ... 919/920):
    1..4 windows stay in the first wave stage, 5..8 take the second, more the workgroup kernel; single-end and as
    pairs of unequal mates."""
    eng, synth, gb, goff, dbs = world
    rng = np.random.default_rng(5)
    host = gb.cpu().numpy()
    offs = goff.cpu().numpy()
    lens = [15, 16, 17, 100, 127, 128, 129, 150, 240, 241, 242, 353, 354, 355, 466, 467, 468, 500, 579, 580, 581,
            692, 693, 694, 805, 806, 807, 918, 919, 920, 921, 1000, 1100]
    seqs = []
    for rep in range(40):
        for L in lens:
            t = int(rng.integers(0, len(offs) - 1))
            a = int(rng.integers(offs[t], offs[t + 1] - L))
            sq = host[a:a + L].copy()
            if rep % 7 == 3 and L > 20:
                sq[int(rng.integers(0, L))] = ord("N")
            seqs.append(sq.tobytes())
    order = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in order]
    if len(seqs) % 2:
        seqs.append(seqs[0])
    rb, ro = orc.pack_reads(seqs)
    for P, M in ((2, 2), (1, 4), (4, 4)):
        db, odb = dbs[P]
        for paired in (False, True):
            nq = len(seqs) // 2 if paired else len(seqs)
            oc, on = odb.query(rb, ro, paired, max_cand=M, emulate_ranks=P, threads=8)
            ws = eng.Workspace(db, nq, len(rb))
            for qf in (0, eng.MCQ_NO_WAVE16):
                cands, ncand = ws.query_host(rb, ro, paired, max_cand=M, emulate_ranks=P, flags=qf)
                _compare(cands, ncand, oc, on, "ragged lengths paired=%d P=%d M=%d qf=%x" % (paired, P, M, qf))


def test_more_wide_queries_than_the_old_queue_held():
    """540 000 pairs of 2 x 250 bp (96 features: back queue of the first stage) whose lists exceed 1024 locations
    (so the second stage passes every one of them on to the front queue while other waves still drain the back
    queue).  The two queues of the overflow array must never meet (with max_queries + 2^19 entries the
    front writes landed on unread back-queue slots and queries were lost or ran twice without any error)."""
    eng = importlib.import_module("metacache-mpi_amd.engine")
    rng = np.random.default_rng(33)
    n_tmpl, L, n_tgt, per = 8, 250, 900, 16
    tmpl = [("".join(rng.choice(list("ACGT"), size=L)), "".join(rng.choice(list("ACGT"), size=L))) for _ in range(n_tmpl)]
    feats = set()
    for a, b in tmpl:
        for sq in (a, b):
            for w0, w1 in orc.windows(L):
                feats.update(int(f) for f in orc.sketch(sq[w0:w1].encode()))
    keys = np.array(sorted(feats), np.uint32)
    off = (np.arange(len(keys) + 1) * per).astype(np.uint64)
    t = rng.integers(0, n_tgt, size=len(keys) * per).astype(np.uint64)
    w = rng.integers(0, 30, size=len(keys) * per).astype(np.uint64)
    locs = np.sort(((t << np.uint64(32)) | w).reshape(len(keys), per), axis=1).reshape(-1)
    t2t = (np.arange(n_tgt) // 5).astype(np.uint32)
    odb = orc.OracleDb(keys, off, locs, t2t)
    tb, to = orc.pack_reads([s.encode() for pr in tmpl for s in pr])
    oc, on = odb.query(tb, to, True, max_cand=4, emulate_ranks=2)
    assert min(len(odb.matches(a, b)) for a, b in tmpl) > 1024
    nq = 540_000
    reps = nq // n_tmpl
    rb = np.tile(np.frombuffer(tb, np.uint8), reps)
    ro = (np.arange(2 * nq + 1) * L).astype(np.uint64)
    db = eng.Database(keys, off, locs, t2t)
    ws = eng.Workspace(db, nq, len(rb))
    cands, ncand = ws.query_host(rb, ro, True, max_cand=4, emulate_ranks=2)
    st = ws.sync()
    assert st["n_overflow"] == nq and st["n_queries"] == nq
    want_n = np.tile(on, reps)
    assert np.array_equal(ncand, want_n)
    want_c = np.tile(oc, (reps, 1, 1))
    mask = np.arange(4)[None, :] < want_n[:, None]
    assert np.array_equal(cands[mask], want_c[mask])


def test_two_class_tail_on_chance_hits():
    """What a RefSeq-scale table does to a read, in small: every feature of a read has a long list of locations that are
    (almost all) alone on their target -- chance hits -- plus the read's true target.  600-1000 locations per read: the second
    wave stage, whose two-class tail sorts only the heavy words and takes the light ones as far as they can enter a list.
    Same results as the oracle with and without it, in all three location forms, single-end and paired, P x M up to 16 (and
    32, where the tail is not attempted); most reads must have taken it."""
    eng = importlib.import_module("metacache-mpi_amd.engine")
    rng = np.random.default_rng(77)
    n, L, n_tgt, n_win = 3000, 150, 20000, 3000
    seqs = ["".join(rng.choice(list("ACGT"), size=L)) for _ in range(n)]
    feats = {}
    for i, sq in enumerate(seqs):
        for w0, w1 in orc.windows(L):
            for f in orc.sketch(sq[w0:w1].encode()):
                feats.setdefault(int(f), []).append(i)
    keys = np.array(sorted(feats), np.uint32)
    lists = []
    for f in keys:
        per = int(rng.integers(18, 34))
        t = rng.integers(0, n_tgt, size=per).astype(np.uint64)
        w = rng.integers(0, n_win, size=per).astype(np.uint64)
        own = []
        for i in feats[int(f)]:                   # the read's own target: a few windows around one locus, so that ranges add up
            own.append((np.uint64(i % n_tgt) << np.uint64(32)) | np.uint64(100 + int(rng.integers(0, 3))))
        lists.append(np.sort(np.concatenate([(t << np.uint64(32)) | w, np.array(own, np.uint64)])))
    off = np.zeros(len(keys) + 1, np.uint64); off[1:] = np.cumsum([len(x) for x in lists])
    locs = np.concatenate(lists)
    t2t = (np.arange(n_tgt) // 3).astype(np.uint32)
    rb, ro = orc.pack_reads([s.encode() for s in seqs])
    odb = orc.OracleDb(keys, off, locs, t2t)
    retries = 0
    for dbflags in (0, eng.MCQ_DB_LOCS_GW, eng.MCQ_DB_LOCS_64):
        db = eng.Database(keys, off, locs, t2t, flags=dbflags)
        for paired in (False, True):
            nq = n // 2 if paired else n
            ws = eng.Workspace(db, nq, n * L)
            for P, M in ((2, 2), (1, 4), (4, 4), (8, 2), (8, 4)):
                oc, on = odb.query(rb, ro, paired, max_cand=M, emulate_ranks=P, threads=8)
                for qf in (0, eng.MCQ_NO_TWO_CLASS):
                    cands, ncand = ws.query_host(rb, ro, paired, max_cand=M, emulate_ranks=P, flags=qf)
                    _compare(cands, ncand, oc, on, "two-class dbflags=%x paired=%d P=%d M=%d qf=%x" % (dbflags, paired, P, M, qf))
                    st = ws.sync()
                    retries += st["n_two_class_retry"]
                    if dbflags != eng.MCQ_DB_LOCS_64 and not qf and not paired:      # (one list for all ranks: tried up to P = 8)
                        assert st["n_two_class"] > nq // 2, st
                    if qf or dbflags == eng.MCQ_DB_LOCS_64:
                        assert st["n_two_class"] == 0, st
                # the P lists and the tree carried out one by one (MCQ_FOLD_BY_LISTS): tried up to P x M = 16
                cands, ncand = ws.query_host(rb, ro, paired, max_cand=M, emulate_ranks=P, flags=eng.MCQ_FOLD_BY_LISTS)
                _compare(cands, ncand, oc, on, "two-class by lists dbflags=%x paired=%d P=%d M=%d" % (dbflags, paired, P, M))
                st = ws.sync()
                if dbflags == eng.MCQ_DB_LOCS_64 or P * M > 16:
                    assert st["n_two_class"] == 0, st
    # lists the second wave stage could not prove go on to the front queue with their probe results (the third stage reads
    # them from the slot's row): that hand-over must have happened here
    assert retries > 0
    # The same reads through the sharded path's home side (mcq_shard_* at one rank: the SH instantiations of the second and third
    # wave stage and of the two-class workgroup kernel, which get their probe results from the exchange instead of the
    # hand-over rows): same answers, and most reads through the tail there too.
    import torch
    dev = torch.device("cuda", 0)
    st_ = torch.cuda.current_stream(dev).cuda_stream
    d_rb = torch.frombuffer(bytearray(rb), dtype=torch.uint8).to(dev)
    d_ro = torch.from_numpy(ro.astype(np.int64)).to(dev)
    for dbflags in (0, eng.MCQ_DB_LOCS_GW):
        db = eng.Database(keys, off, locs, t2t, flags=dbflags)
        for paired in (False, True):
            nq = n // 2 if paired else n
            sh = eng.Shard(db, 1, 0, max_queries=nq, max_bases=n * L, max_seqs=n)
            for P, M in ((2, 2), (4, 4)):
                oc, on = odb.query(rb, ro, paired, max_cand=M, emulate_ranks=P, threads=8)
                for rep in range(2):            # exact mode (first batch of the context), then padded
                    cands = torch.zeros((nq, M, 4), dtype=torch.int32, device=dev); ncand = torch.zeros(nq, dtype=torch.int32, device=dev)
                    sh.query(d_rb.data_ptr(), d_ro.data_ptr(), n, paired, cands.data_ptr(), ncand.data_ptr(), max_cand=M, emulate_ranks=P, stream=st_)
                    st = sh.sync(st_)
                    _compare(cands.cpu().numpy().view(np.uint32), ncand.cpu().numpy().view(np.uint32), oc, on,
                             "two-class under Shard dbflags=%x paired=%d P=%d M=%d rep=%d" % (dbflags, paired, P, M, rep))
                    assert st["n_two_class"] > nq // 2, st
            sh.close()
        db.close()

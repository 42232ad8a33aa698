"""The native sharded path (mcq_shard_*: csrc/mcq_shard.hpp) against the oracle.

* n_ranks = 1 in-process (device-copy transport): exact and padded mode, prepared next batch, every query class, both
  location widths, capacity errors reported.
* 2 and 4 ranks on one GPU, one hash-range shard each, blocks exchanged through gloo by a caller-supplied exchange
  function (RCCL refuses two ranks on one device): same results as the oracle on every rank.
"""
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


def _same(cands, ncand, oc, on, what):
    gc = cands.cpu().numpy().view(np.uint32); gn = ncand.cpu().numpy().view(np.uint32)
    bad = np.nonzero(gn != on)[0]
    assert len(bad) == 0, (what, "ncand differs at", bad[:5], gn[bad[:5]], on[bad[:5]])
    mask = np.arange(gc.shape[1])[None, :] < on[:, None]
    neq = np.any((gc != oc) & mask[:, :, None], axis=(1, 2))
    bad = np.nonzero(neq)[0]
    assert len(bad) == 0, (what, "cands differ at", bad[:5], gc[bad[0]], oc[bad[0]])


@pytest.fixture(scope="module")
def world1():
    eng = importlib.import_module("metacache-mpi_amd.engine")
    synth = importlib.import_module("metacache-mpi_amd.synth")
    dev = torch.device("cuda", 0)
    gb, goff, species = synth.make_genomes(6, 12, 200_000, 400_000, 0.02, seed=5, device=dev)
    table = eng.Table(gb.data_ptr(), goff.data_ptr(), goff.numel() - 1, emulate_ranks=2)
    keys, off, locs, _ = table.to_host()
    sp32 = species.to(torch.int32).contiguous()
    dbs = {}
    for f in (0, eng.MCQ_DB_LOCS_64, eng.MCQ_DB_LOCS_GW | eng.MCQ_DB_SLOTS_16):
        dbs[f] = eng.Database(None, None, None, None, flags=f,
                              device_ptrs=dict(keys=table.keys_ptr, list_off=table.list_off_ptr, locs=table.locs_ptr, tgt2tax=sp32.data_ptr(),
                                               n_keys=table.n_keys, n_locs=table.n_locs, n_targets=sp32.numel()))
    table.close()
    odb = orc.OracleDb(keys, off, locs, species.cpu().numpy().astype(np.uint32))
    return eng, synth, dev, gb, goff, dbs, odb


@pytest.mark.parametrize("locs64", [0, 1, 2], ids=["loc32", "loc64", "gw"])
def test_one_rank_every_query_class(world1, locs64):
    eng, synth, dev, gb, goff, dbs, odb = world1
    db = dbs[(0, eng.MCQ_DB_LOCS_64, eng.MCQ_DB_LOCS_GW | eng.MCQ_DB_SLOTS_16)[locs64]]
    st = torch.cuda.current_stream(dev).cuda_stream
    for paired, L, n in ((False, 150, 40000), (True, 150, 40000), (False, 500, 6000), (True, 250, 6000), (False, 6000, 300)):
        batches = []
        for sd in (1, 2):
            if paired:
                r, ro, _ = synth.sample_pairs(gb, goff, n // 2, L, 300, 700, 0.01, 0.002, seed=100 * L + sd)
            else:
                r, ro, _ = synth.sample_reads(gb, goff, n, L, 0.01, 0.002, seed=100 * L + sd)
            batches.append((r, ro))
        nq = n // 2 if paired else n
        sh = eng.Shard(db, 1, 0, max_queries=nq, max_bases=n * L, max_seqs=n)
        for P, M in ((2, 2), (4, 4), (32, 4)):
            for i, (r, ro) in enumerate(batches + batches[:1]):
                nxt = batches[(i + 1) % 2] if i < 2 else None
                cands = torch.zeros((nq, M, 4), dtype=torch.int32, device=dev); ncand = torch.zeros(nq, dtype=torch.int32, device=dev)
                sh.query(r.data_ptr(), ro.data_ptr(), n, paired, cands.data_ptr(), ncand.data_ptr(), max_cand=M, emulate_ranks=P, stream=st,
                         next_batch=None if nxt is None else (nxt[0].data_ptr(), nxt[1].data_ptr(), n))
                stats = sh.sync(st)
                oc, on, ost = odb.query(r.cpu().numpy().tobytes(), ro.cpu().numpy().astype(np.uint64), paired, max_cand=M, emulate_ranks=P,
                                        threads=8, want_stats=True)
                _same(cands, ncand, oc, on, "paired=%d L=%d P=%d M=%d batch %d" % (paired, L, P, M, i))
                assert stats["n_queries"] == nq and stats["n_features"] == int(ost[0]) and stats["n_locations"] == int(ost[2]), (stats, ost)
        f, l = sh.caps()
        assert f > 0 and l > 0            # learned from the first (exact) batch
        sh.close()


def test_capacity_errors_are_reported(world1):
    eng, synth, dev, gb, goff, dbs, odb = world1
    db = dbs[0]
    st = torch.cuda.current_stream(dev).cuda_stream
    n, L = 20000, 150
    r, ro, _ = synth.sample_reads(gb, goff, n, L, 0.01, 0.002, seed=77)
    cands = torch.zeros((n, 2, 4), dtype=torch.int32, device=dev); ncand = torch.zeros(n, dtype=torch.int32, device=dev)
    oc, on = odb.query(r.cpu().numpy().tobytes(), ro.cpu().numpy().astype(np.uint64), False, max_cand=2, emulate_ranks=2, threads=8)
    # location blocks given far too small: the first batch of a context is exact, and the exact mode sizes the blocks by what the
    # table serves (r04; until then: MCQ_E_CAPACITY)
    sh = eng.Shard(db, 1, 0, max_queries=n, max_bases=n * L, max_locations_per_peer=4096)
    sh.query(r.data_ptr(), ro.data_ptr(), n, False, cands.data_ptr(), ncand.data_ptr(), max_cand=2, emulate_ranks=2, stream=st)
    sh.sync(st)
    _same(cands, ncand, oc, on, "blocks grown by the exact mode")
    assert sh.caps()[1] > 4096
    sh.close()
    # blocks learned from a small batch, then a batch ten times as large in the padded mode: reported; its exact repeat grows them
    sh = eng.Shard(db, 1, 0, max_queries=n, max_bases=n * L)
    small = n // 10
    c2 = torch.zeros((small, 2, 4), dtype=torch.int32, device=dev); n2 = torch.zeros(small, dtype=torch.int32, device=dev)
    sh.query(r.data_ptr(), ro.data_ptr(), small, False, c2.data_ptr(), n2.data_ptr(), max_cand=2, emulate_ranks=2, stream=st)
    sh.sync(st)
    _same(c2, n2, oc[:small], on[:small], "small first batch")
    learned = sh.caps()
    sh.query(r.data_ptr(), ro.data_ptr(), n, False, cands.data_ptr(), ncand.data_ptr(), max_cand=2, emulate_ranks=2, stream=st)
    with pytest.raises(eng.McqError) as e:
        sh.sync(st)
    assert e.value.code == eng.MCQ_E_CAPACITY
    sh.query(r.data_ptr(), ro.data_ptr(), n, False, cands.data_ptr(), ncand.data_ptr(), max_cand=2, emulate_ranks=2, stream=st, exact=True)
    sh.sync(st)
    _same(cands, ncand, oc, on, "exact repeat after a padded overflow")
    assert sh.caps()[0] > learned[0] and sh.caps()[1] > learned[1]
    sh.query(r.data_ptr(), ro.data_ptr(), n, False, cands.data_ptr(), ncand.data_ptr(), max_cand=2, emulate_ranks=2, stream=st)     # padded again, at the new sizes
    sh.sync(st)
    _same(cands, ncand, oc, on, "padded mode at the grown sizes")
    sh.close()
    # padded blocks smaller than the batch needs: reported, and the exact mode then answers correctly
    sh = eng.Shard(db, 1, 0, max_queries=n, max_bases=n * L)
    sh.set_caps(1024, 4096)
    sh.query(r.data_ptr(), ro.data_ptr(), n, False, cands.data_ptr(), ncand.data_ptr(), max_cand=2, emulate_ranks=2, stream=st)
    with pytest.raises(eng.McqError) as e:
        sh.sync(st)
    assert e.value.code == eng.MCQ_E_CAPACITY
    sh.query(r.data_ptr(), ro.data_ptr(), n, False, cands.data_ptr(), ncand.data_ptr(), max_cand=2, emulate_ranks=2, stream=st, exact=True)
    sh.sync(st)
    _same(cands, ncand, oc, on, "exact mode after a capacity error")
    sh.close()
    # a batch with more windows than max_bases was given for: reported, nothing written or read out of bounds
    nl, Ll = 300, 6000
    rl, rol, _ = synth.sample_reads(gb, goff, nl, Ll, 0.01, 0.002, seed=78)
    cl = torch.zeros((nl, 2, 4), dtype=torch.int32, device=dev); ncl = torch.zeros(nl, dtype=torch.int32, device=dev)
    sh2 = eng.Shard(db, 1, 0, max_queries=nl, max_bases=nl * Ll // 10, max_seqs=nl)
    sh2.query(rl.data_ptr(), rol.data_ptr(), nl, False, cl.data_ptr(), ncl.data_ptr(), max_cand=2, emulate_ranks=2, stream=st)
    with pytest.raises(eng.McqError) as e:
        sh2.sync(st)
    assert e.value.code == eng.MCQ_E_CAPACITY and "max_bases" in str(e.value)
    sh2.close()
    # the S1 of a `next` batch that overflows its feature blocks: the error belongs to THAT batch (its flags are not cleared
    # by the sync of the batch it was prepared under)
    sh = eng.Shard(db, 1, 0, max_queries=n, max_bases=n * L, max_features_per_peer=8192)
    small = 200
    c2 = torch.zeros((small, 2, 4), dtype=torch.int32, device=dev); n2 = torch.zeros(small, dtype=torch.int32, device=dev)
    sh.query(r.data_ptr(), ro.data_ptr(), small, False, c2.data_ptr(), n2.data_ptr(), max_cand=2, emulate_ranks=2, stream=st,
             next_batch=(r.data_ptr(), ro.data_ptr(), n))
    sh.sync(st)                                      # the small batch fits
    _same(c2, n2, oc[:small], on[:small], "small batch before an overflowing prepared one")
    sh.query(r.data_ptr(), ro.data_ptr(), n, False, cands.data_ptr(), ncand.data_ptr(), max_cand=2, emulate_ranks=2, stream=st)
    with pytest.raises(eng.McqError) as e:
        sh.sync(st)
    assert e.value.code == eng.MCQ_E_CAPACITY
    sh.close()


@pytest.mark.parametrize("paired,world,locs64", [(0, 2, 0), (1, 2, 1), (0, 4, 0), (1, 2, 2)])
def test_ranks_on_one_gpu_over_gloo(paired, world, locs64):
    with tempfile.TemporaryDirectory() as d:
        outp = os.path.join(d, "res")
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(29800 + paired + 10 * world + 100 * locs64),
               os.path.join(ROOT, "tests", "shard_native_worker.py"), outp, str(paired), str(locs64)]
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:]
        files = sorted(glob.glob(outp + ".*.npz"))
        assert len(files) == world
        for f in files:
            z = np.load(f)
            assert bool(z["ok"][0]), f
            assert (z["overflow"] >= (36 if paired else 72)).all(), z["overflow"]     # wide and long reads left the first stage
            assert (z["caps"] > 0).all()


def test_two_ranks_on_the_bench_table_pairs_and_long_reads():
    """world 2 on the configs[1] table (500 genomes, 1.96 Gbp, one hash-range shard per rank): a batch of 2 x 150 bp pairs
    (configs[3] shape) and a batch of ONT-like reads, mean 8 kb (configs[4] shape: every query in the workgroup kernel),
    blocks through gloo; every rank's results equal the oracle's on the whole table"""
    with tempfile.TemporaryDirectory() as d:
        outp = os.path.join(d, "res")
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", "29977",
               os.path.join(ROOT, "tests", "shard_native_worker.py"), outp, "1", "0", "c2"]
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:]
        files = sorted(glob.glob(outp + ".*.npz"))
        assert len(files) == 2
        for f in files:
            z = np.load(f)
            assert bool(z["ok"][0]), f
            assert z["overflow"][1] == 1500, z["overflow"]          # every long read left the wave stages
            assert (z["locs"] > 0).all()


def test_shards_with_different_location_words_are_refused():
    """two ranks whose handles store their locations in different forms (global-window words on rank 0, bit
    fields on rank 1) -- the home side would decode what it receives with its own tables.  The first (exact) batch compares a
    signature of the format across the ranks: MCQ_E_ARG on every rank, before any location word has travelled."""
    with tempfile.TemporaryDirectory() as d:
        outp = os.path.join(d, "res")
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", "29979",
               os.path.join(ROOT, "tests", "shard_native_worker.py"), outp, "0", "0", "mismatch"]
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:]
        eng = importlib.import_module("metacache-mpi_amd.engine")
        files = sorted(glob.glob(outp + ".*.npz"))
        assert len(files) == 2
        for f in files:
            z = np.load(f)
            assert int(z["code"][0]) == eng.MCQ_E_ARG and bool(z["told"][0]), (f, z["code"])


def test_rccl_over_two_gpus_when_the_box_has_them():
    """mcq_shard_* over REAL RCCL between two GPUs (ncclSend / ncclRecv groups over xGMI): skipped on a one-GPU box, so that
    the driver's multi-GPU node runs it.  One rank per GPU, each its own reads, padded and exact mode, per-rank oracle check."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    with tempfile.TemporaryDirectory() as d:
        outp = os.path.join(d, "res")
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0", MCQ_TEST_TRANSPORT="rccl")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", "29978",
               os.path.join(ROOT, "tests", "shard_native_worker.py"), outp, "0", "0"]
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:]
        files = sorted(glob.glob(outp + ".*.npz"))
        assert len(files) == 2
        for f in files:
            assert bool(np.load(f)["ok"][0]), f


def test_rccl_transport_with_one_rank(tmp_path):
    """the RCCL code path itself (dlopen of librccl, ncclGetUniqueId, ncclCommInitRank, grouped ncclSend / ncclRecv of
    every block) on a box with one GPU: a communicator of ONE rank whose blocks travel to itself through RCCL
    (MCQ_SHARD_FORCE_RCCL) -- exact mode, then padded mode -- against the fused kernel.  In a child process with a
    timeout: a transport that hangs must fail the test, not the suite."""
    code = r"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, %r)
eng = importlib.import_module("metacache-mpi_amd.engine")
synth = importlib.import_module("metacache-mpi_amd.synth")
dev = torch.device("cuda", 0)
gb, goff, species = synth.make_genomes(5, 6, 150_000, 250_000, 0.02, seed=3, device=dev)
sp32 = species.to(torch.int32).contiguous()
db = eng.Database.build(gb.data_ptr(), goff.data_ptr(), sp32.data_ptr(), goff.numel() - 1, emulate_ranks=2)
n, L = 30000, 150
st = torch.cuda.current_stream(dev).cuda_stream
sh = eng.Shard(db, 1, 0, max_queries=n, max_bases=n * L)
sh.comm_rccl(eng.Shard.unique_id())
ws = eng.Workspace(db, n, n * L)
ok = True
for i in range(3):
    r, ro, _ = synth.sample_reads(gb, goff, n, L, 0.01, 0.002, seed=10 + i)
    c0 = torch.zeros((n, 2, 4), dtype=torch.int32, device=dev); n0 = torch.zeros(n, dtype=torch.int32, device=dev)
    c1 = torch.zeros_like(c0); n1 = torch.zeros_like(n0)
    ws.query_device(r.data_ptr(), ro.data_ptr(), n, False, c0.data_ptr(), n0.data_ptr(), max_cand=2, emulate_ranks=2, stream=st)
    sh.query(r.data_ptr(), ro.data_ptr(), n, False, c1.data_ptr(), n1.data_ptr(), max_cand=2, emulate_ranks=2, stream=st)
    sh.sync(st)
    m = torch.arange(2, device=dev)[None, :] < n0[:, None]
    ok = ok and bool(torch.equal(n0, n1)) and bool(torch.equal(c0[m], c1[m])) and int(n0.sum()) > n
print("RCCL_SELF_OK" if ok else "RCCL_SELF_MISMATCH", sh.caps())
""" % ROOT
    env = dict(os.environ, MCQ_SHARD_FORCE_RCCL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_SELF_OK" in r.stdout, r.stdout[-3000:]

"""debug helper (test infrastructure, run by hand): full-size synthetic DB, one batch, GPU vs oracle, print mismatches."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import mc_oracle as orc
eng = importlib.import_module("metacache-mpi_amd.engine")
dbbuild = importlib.import_module("dbbuild_torch")
synth = importlib.import_module("metacache-mpi_amd.synth")
dev = torch.device("cuda", 0)
nsp = int(sys.argv[1]) if len(sys.argv) > 1 else 50
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2
gb, goff, species = synth.make_genomes(nsp, 10, 2_000_000, 6_000_000, 0.02, seed=3, device=dev)
keys, off, locs, _ = dbbuild.build_table(gb, goff, emulate_ranks=P)
print("keys", keys.numel(), "locs", locs.numel(), "max list", int((off[1:] - off[:-1]).max()))
db = dbbuild.make_database(keys, off, locs, species)
n, L = 1 << 18, 150
reads, roff, _ = synth.sample_reads(gb, goff, n, L, 0.005, 0.001, seed=1001)
ws = eng.Workspace(db, n, n * L)
cands = torch.zeros((n, 2, 4), dtype=torch.int32, device=dev); ncand = torch.zeros(n, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
ws.query_device(reads.data_ptr(), roff.data_ptr(), n, False, cands.data_ptr(), ncand.data_ptr(), max_cand=2, emulate_ranks=P, stream=st)
print(ws.sync())
t0 = time.time()
odb = orc.OracleDb(keys.cpu().numpy().astype(np.uint32), off.cpu().numpy().astype(np.uint64),
                   locs.cpu().numpy().astype(np.uint64), species.cpu().numpy().astype(np.uint32))
print("oracle db", time.time() - t0)
rb = reads.cpu().numpy().tobytes(); ro = roff.cpu().numpy().astype(np.uint64)
for thr in (16,):
    t0 = time.time()
    oc, on, ost = odb.query(rb, ro, False, max_cand=2, emulate_ranks=P, threads=thr, want_stats=True)
    print("oracle threads", thr, n / (time.time() - t0), "reads/s", ost)
gc = cands.cpu().numpy().view(np.uint32); gn = ncand.cpu().numpy().view(np.uint32)
badn = np.nonzero(gn != on)[0]
print("ncand mismatches", len(badn), badn[:10])
mask = np.arange(2)[None, :] < np.minimum(on, gn)[:, None]
neq = np.any((gc != oc) & mask[:, :, None], axis=(1, 2))
bad = np.nonzero(neq)[0]
print("cand mismatches", len(bad), bad[:10])
for q in list(badn[:5]) + list(bad[:5]):
    s = rb[q * L:(q + 1) * L]
    m = odb.matches(s)
    tc = odb.target_cands(s)
    print("q", q, "gpu", gn[q], gc[q].tolist(), "cpu", on[q], oc[q].tolist(), "T", len(m), "ntgt", len(tc))
    print("   tcands", tc[np.argsort(-tc[:, 1].astype(np.int64))][:6].tolist())
    # GPU again through host path + block path
    c2, n2 = ws.query_host(s, np.array([0, L], np.uint64), False, max_cand=2, emulate_ranks=P)
    c3, n3 = ws.query_host(s, np.array([0, L], np.uint64), False, max_cand=2, emulate_ranks=P, flags=eng.MCQ_FORCE_BLOCK_PATH)
    print("   gpu single wave", n2, c2.tolist(), "block", n3, c3.tolist())
q = int(badn[0]) if len(badn) else int(bad[0])
s = rb[q * L:(q + 1) * L]
moff, m = ws.debug_matches(s, np.array([0, L], np.uint64), False)
om = odb.matches(s)
print("gpu matches", len(m), "oracle", len(om), "equal", np.array_equal(m, om))
print("gpu   ", [(int(x >> 32), int(x & 0xFFFFFFFF)) for x in m[:24]])
print("oracle", [(int(x >> 32), int(x & 0xFFFFFFFF)) for x in om[:24]])
for P2 in (1, 2):
    c2, n2 = ws.query_host(s, np.array([0, L], np.uint64), False, max_cand=4, emulate_ranks=P2)
    oc2, on2 = odb.query(s, np.array([0, L], np.uint64), False, max_cand=4, emulate_ranks=P2)
    print("P", P2, "gpu", c2[0, :n2[0]].tolist(), "cpu", oc2[0, :on2[0]].tolist())

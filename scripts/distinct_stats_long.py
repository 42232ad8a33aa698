import importlib, sys
import numpy as np, torch
sys.path.insert(0, ".")
pkg = importlib.import_module("metacache-mpi_amd"); pkg.build_hip()
eng = importlib.import_module("metacache-mpi_amd.engine")
synth = importlib.import_module("metacache-mpi_amd.synth")
dev = torch.device("cuda", 0)
g, off, sp = synth.make_genomes(50, 10, 2_000_000, 6_000_000, 0.02, seed=3, device=dev)
db = eng.Database.build(g.data_ptr(), off.data_ptr(), sp.to(torch.int32).data_ptr(), off.numel() - 1, emulate_ranks=2)
n = 2000
r, ro, _ = synth.sample_long_reads(g, off, n, 8000, 0.08, seed=1000)
ws = eng.Workspace(db, n, int(ro[-1]))
moff, m = ws.debug_matches(r.cpu().numpy().tobytes(), ro.cpu().numpy().astype(np.uint64), False)
T = np.diff(moff.astype(np.int64))
D = np.array([len(np.unique(m[int(moff[i]):int(moff[i+1])])) for i in range(n)])
NT = np.array([len(np.unique(m[int(moff[i]):int(moff[i+1])] >> np.uint64(32))) for i in range(n)])
pc = lambda x: [int(np.percentile(x, p)) for p in (5, 25, 50, 75, 95, 99)]
print("long reads: T", T.mean(), pc(T), " D", D.mean(), pc(D), " D/T %.2f" % (D.sum() / T.sum()), " targets", NT.mean(), pc(NT))
print("T<=4096 %.1f%%  T<=6144 %.1f%%  T<=8192 %.1f%%  D<=2048 %.1f%%  D<=4096 %.1f%%" % tuple(100 * x for x in ((T <= 4096).mean(), (T <= 6144).mean(), (T <= 8192).mean(), (D <= 2048).mean(), (D <= 4096).mean())))

"""How many DISTINCT (target, window) pairs does a read's match list hold?  (decides whether a
de-duplicating pass before the sort pays).  Bench DB, 32k reads, via mcq_debug_matches."""
import importlib
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
pkg = importlib.import_module("metacache-mpi_amd"); pkg.build_hip()
eng = importlib.import_module("metacache-mpi_amd.engine")
synth = importlib.import_module("metacache-mpi_amd.synth")
dev = torch.device("cuda", 0)
NSPECIES = int(sys.argv[1]) if len(sys.argv) > 1 else 50
g, off, sp = synth.make_genomes(NSPECIES, 10, 2_000_000, 6_000_000, 0.02, seed=3, device=dev)
db = eng.Database.build(g.data_ptr(), off.data_ptr(), sp.to(torch.int32).data_ptr(), off.numel() - 1, emulate_ranks=2)
for name, paired in (("c2", False), ("paired", True)):
    n = 1 << 15
    if paired:
        r, ro, _ = synth.sample_pairs(g, off, n // 2, 150, 300, 500, 0.005, 0.001, seed=1000)
    else:
        r, ro, _ = synth.sample_reads(g, off, n, 150, 0.005, 0.001, seed=1000)
    ws = eng.Workspace(db, n, int(ro[-1]))
    moff, m = ws.debug_matches(r.cpu().numpy().tobytes(), ro.cpu().numpy().astype(np.uint64), paired)
    T = np.diff(moff.astype(np.int64))
    newq = np.zeros(len(m), bool); newq[moff[:-1][T > 0].astype(np.int64)] = True
    distinct = np.ones(len(m), bool); distinct[1:] = m[1:] != m[:-1]; distinct |= newq
    D = np.add.reduceat(distinct.astype(np.int64), moff[:-1][T > 0].astype(np.int64))
    tg = m >> np.uint64(32)
    dt = np.ones(len(m), bool); dt[1:] = tg[1:] != tg[:-1]; dt |= newq
    NT = np.add.reduceat(dt.astype(np.int64), moff[:-1][T > 0].astype(np.int64))
    Tn = T[T > 0]
    def pct(x): return [int(np.percentile(x, p)) for p in (5, 25, 50, 75, 95, 99)]
    print(name, "locations/query mean %.1f pct(5,25,50,75,95,99) %s" % (Tn.mean(), pct(Tn)))
    print(name, "distinct (tgt,win) mean %.1f %s   D<=64: %.2f%%  D<=32: %.2f%%" % (D.mean(), pct(D), 100 * (D <= 64).mean(), 100 * (D <= 32).mean()))
    print(name, "distinct targets mean %.1f %s" % (NT.mean(), pct(NT)))
    print(name, "T<=64 %.1f%%  <=128 %.1f%%  <=256 %.1f%%  <=384 %.1f%%  <=512 %.1f%%" % tuple(100 * (Tn <= c).mean() for c in (64, 128, 256, 384, 512)))
    print(name, "D<=128 %.1f%%  D<=192 %.1f%%  D<=256 %.1f%%;  T<=384 and D>128: %.1f%%;  T<=512 and D<=256: %.1f%%" % (100 * (D <= 128).mean(), 100 * (D <= 192).mean(), 100 * (D <= 256).mean(), 100 * ((Tn <= 384) & (D > 128)).mean(), 100 * ((Tn <= 512) & (D <= 256)).mean()), flush=True)

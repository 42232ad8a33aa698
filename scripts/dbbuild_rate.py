"""Times mcq_build_table (csrc/mcq_build.hip) against the torch-plumbing build of dbbuild.py on the
bench's C2 genomes and checks the two tables are identical."""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import sys
import time

import torch

sys.path.insert(0, ".")
pkg = importlib.import_module("metacache-mpi_amd"); pkg.build_hip()
eng = importlib.import_module("metacache-mpi_amd.engine")
dbbuild = importlib.import_module("dbbuild_torch")
synth = importlib.import_module("metacache-mpi_amd.synth")
dev = torch.device("cuda", 0)
species, strains = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 4)
g, off, sp = synth.make_genomes(species, strains, 400000, 600000, 0.02, seed=3, device=dev)
nt = off.numel() - 1
for P in (1, 8):
    torch.cuda.synchronize(); t0 = time.time()
    tb = eng.Table(g.data_ptr(), off.data_ptr(), nt, emulate_ranks=P)
    torch.cuda.synchronize(); t1 = time.time()
    k2, o2, l2, _ = dbbuild.build_table(g, off, emulate_ranks=P)
    torch.cuda.synchronize(); t2 = time.time()
    k, o, l, _ = tb.to_host()
    same = (k.view("int32") == k2.to(torch.int32).cpu().numpy()).all() and (o.view("int64") == o2.cpu().numpy()).all() \
        and (l.view("int64") == l2.cpu().numpy()).all()
    print("P=%d bp=%.3g keys=%d locs=%d  mcq_build_table %.2fs (%.2f Gbp/s)  dbbuild.py %.2fs  identical=%s"
          % (P, int(off[-1]), tb.n_keys, tb.n_locs, t1 - t0, int(off[-1]) / (t1 - t0) / 1e9, t2 - t1, bool(same)), flush=True)
    tb.close(); del k2, o2, l2

"""Is a batch of the fused path capturable in a HIP graph, and does replaying it beat the plain launches?
(memset + k_query_wave + k_query_wave16 + k_query_block per batch.)"""
import importlib
import sys
import time

import torch

sys.path.insert(0, ".")
pkg = importlib.import_module("metacache-mpi_amd"); pkg.build_hip()
eng = importlib.import_module("metacache-mpi_amd.engine")
synth = importlib.import_module("metacache-mpi_amd.synth")
dev = torch.device("cuda", 0)
g, off, sp = synth.make_genomes(50, 10, 2_000_000, 6_000_000, 0.02, seed=3, device=dev)
db = eng.Database.build(g.data_ptr(), off.data_ptr(), sp.to(torch.int32).data_ptr(), off.numel() - 1, emulate_ranks=2)
B, L, nb = 1 << 20, 150, 12
batches = [synth.sample_reads(g, off, B, L, 0.005, 0.001, seed=1000 + i)[:2] for i in range(nb)]
ws = eng.Workspace(db, B, B * L)
cands = torch.zeros((B, 2, 4), dtype=torch.int32, device=dev)
ncand = torch.zeros(B, dtype=torch.int32, device=dev)


def step(i, stream):
    r, ro = batches[i % nb]
    ws.query_device(r.data_ptr(), ro.data_ptr(), B, False, cands.data_ptr(), ncand.data_ptr(), max_cand=2, emulate_ranks=2, stream=stream)


def run(fn, n=48):
    for i in range(4):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


main = torch.cuda.current_stream(dev)
print("plain launches: %.4f ms/step" % run(lambda i: step(i, main.cuda_stream)))
ref = (cands.clone(), ncand.clone())
graphs = []
for i in range(nb):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        step(i, torch.cuda.current_stream(dev).cuda_stream)
    graphs.append(gr)
print("graph replay:   %.4f ms/step" % run(lambda i: graphs[i % nb].replay()))
print("plain again:    %.4f ms/step" % run(lambda i: step(i, main.cuda_stream)))
print("same result as the plain launches:", bool(torch.equal(ref[0], cands) and torch.equal(ref[1], ncand)))

"""throughput of mcq_fastq_index and of the query reading bases in place from raw FASTQ text"""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
eng = importlib.import_module("metacache-mpi_amd.engine")
dbbuild = importlib.import_module("dbbuild_torch")
synth = importlib.import_module("metacache-mpi_amd.synth")
dev = torch.device("cuda", 0)
gb, goff, species = synth.make_genomes(50, 10, 2_000_000, 6_000_000, 0.02, seed=3, device=dev)
keys, off, locs, _ = dbbuild.build_table(gb, goff, emulate_ranks=2)
db = dbbuild.make_database(keys, off, locs, species)
n, L = 1 << 20, 150
reads, roff, _ = synth.sample_reads(gb, goff, n, L, 0.005, 0.001, seed=1)
# FASTQ text on the GPU: "@r0000000\n" + seq + "\n+\n" + qual + "\n"  (fixed-width records: 10 + 151 + 2 + 151 = 314 B)
rec = torch.zeros((n, 314), dtype=torch.uint8, device=dev)
rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
idx = torch.arange(n, device=dev)
for d in range(7):
    rec[:, 8 - d] = (48 + (idx // 10 ** d) % 10).to(torch.uint8)
rec[:, 9] = 10
rec[:, 10:160] = reads.reshape(n, L); rec[:, 160] = 10
rec[:, 161] = ord("+"); rec[:, 162] = 10
rec[:, 163:313] = ord("I"); rec[:, 313] = 10
text = rec.reshape(-1).contiguous()
ranges = torch.zeros(2 * n, dtype=torch.int64, device=dev); cnt = torch.zeros(1, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
ws = eng.Workspace(db, n, text.numel())
cands = torch.zeros((n, 2, 4), dtype=torch.int32, device=dev); ncand = torch.zeros(n, dtype=torch.int32, device=dev)
c2 = torch.zeros_like(cands); n2 = torch.zeros_like(ncand)
def run(k):
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    for _ in range(k):
        eng.fastq_index(text.data_ptr(), text.numel(), ranges.data_ptr(), n, cnt.data_ptr(), st)
    e1.record()
    for _ in range(k):
        ws.query_device(text.data_ptr(), ranges.data_ptr(), n, False, cands.data_ptr(), ncand.data_ptr(), max_cand=2, emulate_ranks=2, stream=st, ranges=True)
    e2.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k, e1.elapsed_time(e2) / k
run(2)
ti, tq = run(10)
ws.query_device(reads.data_ptr(), roff.data_ptr(), n, False, c2.data_ptr(), n2.data_ptr(), max_cand=2, emulate_ranks=2, stream=st)
torch.cuda.synchronize()
print("records", int(cnt.item()), "index ms", round(ti, 3), "= %.1f GB/s of text" % (text.numel() / ti / 1e6), "| query from raw text ms", round(tq, 3),
      "| same results as packed batch:", bool(torch.equal(cands, c2) and torch.equal(ncand, n2)))

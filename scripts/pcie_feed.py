"""experiment: how to feed mcq_query from HOST buffers at rate.
  seq       copy in -> kernel -> copy out on one stream (bench.py's pcie_inclusive leg)
  zerocopy  the kernel reads the bases straight out of pinned host memory (no copy engine, no staging buffer)
  overlap   double-buffered copies on their own streams under the kernel of the previous batch
"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
eng = importlib.import_module("metacache-mpi_amd.engine")
synth = importlib.import_module("metacache-mpi_amd.synth")
dev = torch.device("cuda", 0)
gb, goff, species = synth.make_genomes(50, 10, 2_000_000, 6_000_000, 0.02, seed=3, device=dev)
torch.cuda.empty_cache()
table = eng.Table(gb.data_ptr(), goff.data_ptr(), goff.numel() - 1, emulate_ranks=2)
sp32 = species.to(torch.int32).contiguous()
db = eng.Database(None, None, None, None, device_ptrs=dict(keys=table.keys_ptr, list_off=table.list_off_ptr, locs=table.locs_ptr, tgt2tax=sp32.data_ptr(),
                                                           n_keys=table.n_keys, n_locs=table.n_locs, n_targets=sp32.numel()))
table.close()
B, L, NB, steps = 1 << 20, 150, 4, 24
batches = [synth.sample_reads(gb, goff, B, L, 0.005, 0.001, seed=1000 + i) for i in range(NB)]
off = batches[0][1]
hb = [b[0].cpu().pin_memory() for b in batches]
ws = eng.Workspace(db, B, B * L)
cands = [torch.zeros((B, 2, 4), dtype=torch.int32, device=dev) for _ in range(2)]
ncand = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(2)]
hc = [torch.zeros((B, 2, 4), dtype=torch.int32).pin_memory() for _ in range(2)]
hn = [torch.zeros(B, dtype=torch.int32).pin_memory() for _ in range(2)]
dbuf = [torch.empty_like(batches[0][0]) for _ in range(2)]
main = torch.cuda.current_stream(dev)


def run(name, fn):
    fn(2); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(steps); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-10s %.3f ms/step  %.3e reads/s" % (name, 1e3 * dt / steps, steps * B / dt), flush=True)


def q(bases_ptr, k, stream):
    ws.query_device(bases_ptr, off.data_ptr(), B, False, cands[k].data_ptr(), ncand[k].data_ptr(), max_cand=2, emulate_ranks=2, stream=stream.cuda_stream)


def seq(n):
    for i in range(n):
        dbuf[0].copy_(hb[i % NB], non_blocking=True)
        q(dbuf[0].data_ptr(), 0, main)
        hc[0].copy_(cands[0], non_blocking=True); hn[0].copy_(ncand[0], non_blocking=True)


def resident(n):
    for i in range(n):
        q(batches[i % NB][0].data_ptr(), 0, main)


def zerocopy(n):
    for i in range(n):
        q(hb[i % NB].data_ptr(), 0, main)
        hc[0].copy_(cands[0], non_blocking=True); hn[0].copy_(ncand[0], non_blocking=True)


cin, cout = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def overlap(n):
    ev_in = [torch.cuda.Event() for _ in range(2)]; ev_k = [torch.cuda.Event() for _ in range(2)]; ev_out = [torch.cuda.Event() for _ in range(2)]
    for i in range(n):
        k = i & 1
        with torch.cuda.stream(cin):
            if i >= 2: cin.wait_event(ev_k[k])                 # the kernel that read this buffer two steps ago is done
            dbuf[k].copy_(hb[i % NB], non_blocking=True); ev_in[k].record(cin)
        main.wait_event(ev_in[k])
        if i >= 2: main.wait_event(ev_out[k])                  # its result buffers have been copied out
        q(dbuf[k].data_ptr(), k, main); ev_k[k].record(main)
        with torch.cuda.stream(cout):
            cout.wait_event(ev_k[k])
            hc[k].copy_(cands[k], non_blocking=True); hn[k].copy_(ncand[k], non_blocking=True); ev_out[k].record(cout)


for name, fn in (("resident", resident), ("seq", seq), ("zerocopy", zerocopy), ("overlap", overlap)):
    if len(sys.argv) > 1 and name not in sys.argv[1:]: continue
    run(name, fn)
ok = True
print("done")
ws.timing(True)
resident(10); torch.cuda.synchronize()
print("kernel times", ws.kernel_times())

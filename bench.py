#!/usr/bin/env python3
"""bench.py -- query reads/sec of the MI355X query-path engine on BASELINE.json's configs.

One step = one pass of the hot path (sketch -> probe -> gather -> sort -> candidates ->
fold) over one batch of synthetic reads already resident in HBM.  Prints ONE JSON line.

  N = 1   BASELINE configs[1]: 500 synthetic genomes in HBM, 150 bp reads, fused kernel
          (the default --steps x --batch covers the config's 50 M reads).
  N > 1   BASELINE configs[2] shape: the feature table is hash-range-sharded over the N
          GPUs (one process per GPU), features go to their owners and hits come back by
          RCCL all-to-all -- that is `value`; the same reads on a replicated table
          ("replicas only", no collective) are reported beside it as the second curve.
          The database grows with N (--species x N species) unless --db-fixed.

  python bench.py --gpus N --steps K --warmup W      (starts the N ranks itself)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (or is started as one of them)
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak
# the three timed slots of a batch: first wave stage; second + third wave stage (k_query_wave32 runs only where the two-class tail
# does); the workgroup kernels (plain, and the one with the two-class tail behind it)
KERNELS = {"fused": ("k_query_wave", "k_query_wave16+k_query_wave32", "k_query_block"),
           "sharded": ("k_query_wave<sharded>", "k_query_wave16+k_query_wave32<sharded>", "k_query_block<sharded>")}
EXIT_SHARDED_FAILED = 3        # the line is printed (replicas leg), the exit status says the sharded leg failed
EXIT_PARITY_FAILED = 4         # GPU result differs from the CPU oracle / the fused kernel


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1 << 20, help="reads per step per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--species", type=int, default=None, help="species of the database (x strains genomes; default 50, with --refseq-scale 2600); at N > 1 per GPU unless --db-fixed")
    ap.add_argument("--strains", type=int, default=10)
    ap.add_argument("--genome-min", type=int, default=2_000_000)
    ap.add_argument("--genome-max", type=int, default=6_000_000)
    ap.add_argument("--divergence", type=float, default=0.02)
    ap.add_argument("--db-fixed", action="store_true", help="N > 1: keep the N = 1 database instead of --species x N")
    ap.add_argument("--emulate-ranks", type=int, default=2, help="reference rank count whose results are reproduced")
    ap.add_argument("--max-cand", type=int, default=2)
    ap.add_argument("--mode", default="auto", choices=["auto", "replicas", "sharded"],
                    help="auto = fused kernel at N = 1, sharded (+ replicas beside it) at N > 1")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo only to rehearse N>1 on a box with one GPU")
    ap.add_argument("--no-replicas-leg", action="store_true", help="N>1: skip the replicated-table measurement")
    ap.add_argument("--sharded-timeout", type=int, default=240, help="seconds the sharded leg may take before the run ends without it")
    ap.add_argument("--small", action="store_true", help="tiny DB / few reads (plumbing check)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-threads", type=int, default=16, help="threads of the CPU baseline (a 1-GPU box has a 16-core share)")
    ap.add_argument("--long-batch", type=int, default=1 << 14, help="reads per step for --workload long")
    ap.add_argument("--long-mean", type=int, default=8000)
    ap.add_argument("--workload", default="c2", choices=["c2", "paired", "long"],
                    help="c2 = BASELINE configs[1] (150 bp single-end); paired = 2x150 bp pairs (configs[3] shape); "
                         "long = ONT-like reads, mean 8 kb (configs[4] shape)")
    ap.add_argument("--stop-stage", type=int, default=0, help="profiling builds (-DMCQ_PROFILE_HOOKS): stop the fused kernel after stage 1..5 (results invalid)")
    ap.add_argument("--query-flags", type=lambda x: int(x, 0), default=0,
                    help="experiments: extra mcq_query_opts.flags (0x400 raw sort); results stay exact")
    ap.add_argument("--distinct-batches", type=int, default=0, help="0 = one per step (capped by memory)")
    ap.add_argument("--no-pcie-leg", action="store_true")
    ap.add_argument("--packed-input", action="store_true", help="experiment: the resident batches in the MCQ_BATCH_PACKED form (c2 / paired, fused leg)")
    ap.add_argument("--loc-format", default="auto", choices=["auto", "fields32", "fields64", "gw"],
                    help="location words of the table: auto = what mcq_db_create picks (32-bit bit fields if they fit, else the 32-bit "
                         "global window index, else 64 bit); gw / fields64 force a form (fields32 = auto, fails the run if it does not fit)")
    ap.add_argument("--bucket-bytes", type=int, default=0, choices=[0, 16, 64], help="table layout: 0 = per table (mean list length), 16 / 64 force it")
    ap.add_argument("--contigs", type=int, default=None, help="split every genome into this many targets (RefSeq assemblies: many sequences per genome; default 1, with --refseq-scale 2)")
    ap.add_argument("--refseq-scale", action="store_true",
                    help="BASELINE configs[2] shape on ONE GPU: --species 2600 x 10 strains (104 Gbp), --contigs 2 (52 001 targets), one 16 Mbp "
                         "chromosome, -remove-overpopulated-features, table built in parts; any of these can still be given explicitly")
    ap.add_argument("--remove-overpopulated", action="store_true", help="build option -remove-overpopulated-features (src/mode_build.cpp:847-1074)")
    ap.add_argument("--build-parts", action="store_true", help="build the table in feature-hash parts (mcq_build_parts) whatever its size")
    ap.add_argument("--long-genome-mbp", type=float, default=None,
                    help="add one genome of this many Mbp (>= 14.9 Mbp = 2^17 windows: with >= 2^15 targets the (target, window) "
                         "fields no longer fit 32 bits, as on RefSeq)")
    ap.add_argument("--shard-serial", action="store_true",
                    help="diagnosis: the sharded leg without overlap -- no `next` batch, a sync after every step -- so that stage_ms_per_step are the stages' times ALONE (results invalid as a throughput)")
    ap.add_argument("--no-refseq-block", action="store_true",
                    help="default N = 1 line: skip the `refseq_scale` side block (the RefSeq-scale table built and timed in the same run, ~80 s)")
    a = ap.parse_args()
    # defaults that depend on --refseq-scale (explicit values win, however they were abbreviated)
    if a.species is None: a.species = 2600 if a.refseq_scale else 50
    if a.contigs is None: a.contigs = 2 if a.refseq_scale else 1
    if a.long_genome_mbp is None: a.long_genome_mbp = 16.0 if a.refseq_scale else 0.0
    if a.refseq_scale: a.remove_overpopulated = True
    return a


def arm_watchdog(seconds, last_words, exit_code):
    """Ends the process `seconds` from now with `exit_code`, whatever happens: last_words() (prints the line of what has
    been measured so far) may fail or hang in a library -- os._exit still runs.  Returns the timer (cancel() disarms it).
    Never restarts or re-executes anything: a process that touched the GPU just ends."""
    def give_up():
        try:
            last_words()
        except BaseException as e:          # noqa: BLE001 -- nothing may keep the exit from happening
            try:
                sys.stderr.write("[bench watchdog] could not print the fallback line: %r\n" % (e,))
            except Exception:
                pass
        finally:
            try:
                sys.stderr.flush()
            finally:
                os._exit(exit_code)
    t = threading.Timer(seconds, give_up)
    t.daemon = True
    t.start()
    return t


def spawn_ranks(a):
    """--gpus N without a launcher: start the N ranks as fresh child processes (torch.distributed.run) BEFORE this
    process touches the GPU -- nothing that initialised HIP is ever replaced or re-executed.  Rank 0 prints the line
    on the inherited stdout; the exit status is the launcher's."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def algorithmic_bytes(n_bases, st):
    # B = L_bases + 16 B x features (one slot per probe) + 8 B x locations + 16 B x candidates
    return n_bases + 16 * st["n_features"] + 8 * st["n_locations"] + 16 * st["n_cands"]


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a))
    import torch
    if a.small:
        a.species, a.strains, a.genome_min, a.genome_max = 6, 4, 150_000, 300_000
        a.batch = min(a.batch, 1 << 16)
        a.steps = min(a.steps, 4)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.stderr.write("[bench] --gpus %d but WORLD_SIZE=%d: the launcher decides, n_gpus = %d\n" % (a.gpus, world, world))
    # stdout carries exactly one JSON line: libraries that print to fd 1 (RCCL's version banner at communicator
    # creation does) are sent to stderr; emit() below writes the line to the real stdout
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    dist = None
    if world > 1:
        # control plane (barriers, the max over ranks, carrying the RCCL id): torch.distributed over gloo.  The data
        # plane of the sharded path is the engine's own RCCL communicator (ncclSend / ncclRecv groups, csrc/mcq_shard.hpp).
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU")
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)

    pkg = importlib.import_module("metacache-mpi_amd")
    if rank == 0:
        pkg.build_hip()                 # a no-op when the shipped library is current; never N concurrent builds
    if dist is not None:
        dist.barrier()
    eng = importlib.import_module("metacache-mpi_amd.engine")
    synth = importlib.import_module("metacache-mpi_amd.synth")

    # N = 1: the fused kernel on the whole table.  N > 1: the hash-range-sharded table with the all-to-all exchange is
    # the primary number (north_star / configs[2]); "replicas only" (table replicated, reads split, no collective: what
    # one would deploy while the table fits one 288 GB GPU) is measured in the same run as the second curve.
    mode = a.mode
    if mode == "auto":
        mode = "single" if world == 1 else "sharded"
    if world == 1 and mode == "replicas":
        mode = "single"
    with_sharded = mode == "sharded"
    with_fused = mode != "sharded" or not a.no_replicas_leg
    n_species = a.species * (world if (world > 1 and not a.db_fixed) else 1)

    def setup_data():
        """genomes -> read batches -> table (this rank's shard and / or the whole table); returns everything the legs need"""
        # ---- database: same seeded genomes on every rank; each rank keeps its hash-range shard
        t_setup = time.time()
        if a.refseq_scale:
            # BASELINE configs[2] shape on one GPU: >= 100 Gbp, >= 2^15 sequences, one chromosome of > 2^17 windows
            gen_bases, gen_off, species = synth.make_genomes_big(n_species, a.strains, a.genome_min, a.genome_max, a.divergence, seed=3,
                                                                 device=dev, extra_genome=int(a.long_genome_mbp * 1e6))
        else:
            gen_bases, gen_off, species = synth.make_genomes(n_species, a.strains, a.genome_min, a.genome_max,
                                                             a.divergence, seed=3, device=dev)
            if a.long_genome_mbp > 0:
                gen_bases, gen_off, species = synth.add_genome(gen_bases, gen_off, species, int(a.long_genome_mbp * 1e6), seed=4)
        if a.contigs > 1:
            gen_off, species = synth.split_targets(gen_off, species, a.contigs, keep_last_whole=a.long_genome_mbp > 0)
        n_targets = species.numel()
        db_bp = int(gen_off[-1].item())
        tgt_windows = synth.window_counts(gen_off)
        t_genomes = time.time() - t_setup

        # ---- reads (distinct batches, resident in HBM before the clock starts; sampled before the table is built, because the
        # sequences of a table built in parts are released before the table itself is allocated)
        L, B = a.read_len, a.batch
        paired = a.workload == "paired"
        if a.workload == "long":
            B = min(B, a.long_batch)
        nb = a.distinct_batches or a.steps
        free = torch.cuda.mem_get_info(dev)[0]
        per_batch = (B * a.long_mean * 12) if a.workload == "long" else (B * L * 3)
        nb = max(1, min(nb, int(free * (0.1 if a.refseq_scale else 0.5)) // per_batch))
        batches, offsets = [], []
        for i in range(nb):
            sd = 1000 + 7919 * rank + i
            if a.workload == "c2":
                r, off, _ = synth.sample_reads(gen_bases, gen_off, B, L, 0.005, 0.001, seed=sd)
            elif paired:
                r, off, _ = synth.sample_pairs(gen_bases, gen_off, B // 2, L, 300, 500, 0.005, 0.001, seed=sd)
            else:
                r, off, _ = synth.sample_long_reads(gen_bases, gen_off, B, a.long_mean, 0.08, seed=sd)
            batches.append(r); offsets.append(off)

        # ---- table built on the GPU through the C ABI (csrc/mcq_build.hip): in one piece (mcq_build_table), or in feature-hash
        # parts (mcq_build_parts) when the one-piece temporaries would not fit -- the RefSeq-scale table
        torch.cuda.empty_cache()            # the builder allocates with hipMalloc, outside torch's cache
        sp32 = species.to(torch.int32).contiguous()
        bflags = eng.MCQ_BUILD_REMOVE_OVERPOPULATED if a.remove_overpopulated else 0
        dbflags = {"auto": 0, "fields32": 0, "fields64": eng.MCQ_DB_LOCS_64, "gw": eng.MCQ_DB_LOCS_GW}[a.loc_format] | \
                  {0: 0, 16: eng.MCQ_DB_SLOTS_16, 64: eng.MCQ_DB_BUCKETS_64}[a.bucket_bytes]
        in_parts = a.refseq_scale or a.build_parts
        keys, list_off, locs = (None, None, None)
        t_build = time.time()
        if in_parts:
            # (only this rank's hash range of the features when nothing else is wanted: half the passes over the sequences at N = 2)
            own_only = with_sharded and not with_fused
            parts = eng.Parts(gen_bases.data_ptr(), gen_off.data_ptr(), n_targets, emulate_ranks=a.emulate_ranks, flags=bflags, device=dev.index or 0,
                              n_shards=world if own_only else 1, shard_id=rank if own_only else 0)
            torch.cuda.synchronize(dev)
            t_build = time.time() - t_build
            n_keys, n_locs, n_parts = parts.n_keys, parts.n_locs, parts.n_parts
            del gen_bases                   # the sequences go before the table comes
            torch.cuda.empty_cache()
            lflags = dbflags & (eng.MCQ_DB_SLOTS_16 | eng.MCQ_DB_BUCKETS_64)
            db = parts.database(sp32.data_ptr(), flags=lflags) if with_fused else None
            db_shard = parts.database(sp32.data_ptr(), n_shards=world, shard_id=rank, flags=lflags) if with_sharded else None
            parts.close()
            # the CPU leg (the checker) works on the part of the table a batch can touch, read back through the staged entry points
            want_cpu = rank == 0 and not a.no_cpu_baseline and not a.stop_stage and with_fused
            host_table = None
        else:
            table = eng.Table(gen_bases.data_ptr(), gen_off.data_ptr(), n_targets, emulate_ranks=a.emulate_ranks, flags=bflags,
                              device=dev.index or 0)
            torch.cuda.synchronize(dev)
            t_build = time.time() - t_build
            n_parts = 1

            def make_db(n_shards=1, shard_id=0):
                return eng.Database(None, None, None, None, n_shards=n_shards, shard_id=shard_id, device=dev.index or 0, flags=dbflags,
                                    device_ptrs=dict(keys=table.keys_ptr, list_off=table.list_off_ptr, locs=table.locs_ptr,
                                                     tgt2tax=sp32.data_ptr(), n_keys=table.n_keys, n_locs=table.n_locs,
                                                     n_targets=sp32.numel()))
            # full table (fused single-GPU path / replicas) and/or this rank's hash-range shard
            db = make_db() if with_fused else None
            db_shard = make_db(world, rank) if with_sharded else None
            n_keys, n_locs = table.n_keys, table.n_locs
            want_cpu = rank == 0 and not a.no_cpu_baseline and not a.stop_stage and (world == 1 or n_locs <= 600_000_000)
            host_table = table.to_host()[:3] if want_cpu else None      # only the CPU baseline (the checker) reads these
            table.close()
            del gen_bases
        return dict(species=species, n_targets=n_targets, db_bp=db_bp, tgt_windows=tgt_windows, t_genomes=t_genomes, L=L, B=B, paired=paired,
                    nb=nb, batches=batches, offsets=offsets, sp32=sp32, t_build=t_build, n_parts=n_parts, db=db, db_shard=db_shard,
                    n_keys=n_keys, n_locs=n_locs, want_cpu=want_cpu, host_table=host_table, t_setup=t_setup)

    # ranks that share a device (the gloo rehearsal of N > 1 on a one-GPU box) take turns: a RefSeq-scale table needs its 100 Gbp of
    # sequences on the GPU while it is built, once at a time
    share = world > 1 and torch.cuda.device_count() < world
    D = None
    for turn in range(world if share else 1):
        if not share or turn == rank:
            D = setup_data()
            torch.cuda.synchronize(dev)
        if share:
            dist.barrier()
    species, n_targets, db_bp, tgt_windows, t_genomes, L, B, paired = (D[k] for k in ("species", "n_targets", "db_bp", "tgt_windows", "t_genomes", "L", "B", "paired"))
    nb, batches, offsets, sp32, t_build, n_parts, db, db_shard = (D[k] for k in ("nb", "batches", "offsets", "sp32", "t_build", "n_parts", "db", "db_shard"))
    n_keys, n_locs, want_cpu, host_table, t_setup = (D[k] for k in ("n_keys", "n_locs", "want_cpu", "host_table", "t_setup"))
    db_layout = (db or db_shard).layout()
    db_layout.pop("gw_offsets", None)
    db_layout["loc_format"] = {eng.MCQ_LOC_FIELDS64: "fields64", eng.MCQ_LOC_FIELDS32: "fields32", eng.MCQ_LOC_GLOBAL_WINDOW: "global_window"}[db_layout["loc_format"]]
    if a.loc_format == "fields32" and db_layout["loc_format"] != "fields32":
        sys.exit("--loc-format fields32: the (target, window) fields of this table do not fit 32 bits (the handle chose %s)" % db_layout["loc_format"])
    max_bases = max(int(o[-1].item()) for o in offsets)
    nq = B // 2 if paired else B
    sharded = None
    ws = eng.Workspace(db, nq, max_bases) if with_fused else None
    cands = torch.zeros((nq, a.max_cand, 4), dtype=torch.int32, device=dev)
    ncand = torch.zeros(nq, dtype=torch.int32, device=dev)
    cands_s, ncand_s = torch.zeros_like(cands), torch.zeros_like(ncand)       # results of the sharded leg
    stream = torch.cuda.current_stream(dev).cuda_stream
    torch.cuda.synchronize(dev)
    t_setup = time.time() - t_setup
    qflags = (((a.stop_stage & 15) << 12) if a.stop_stage else 0) | a.query_flags

    def step_sharded(i):
        # the next step's batch is announced so that its sketching runs on the second stream under this step's exchange
        nxt = (batches[(i + 1) % nb].data_ptr(), offsets[(i + 1) % nb].data_ptr(), B) if (i + 1 < a.warmup + a.steps and not a.shard_serial) else None
        sharded.query(batches[i % nb].data_ptr(), offsets[i % nb].data_ptr(), B, paired, cands_s.data_ptr(), ncand_s.data_ptr(),
                      max_cand=a.max_cand, emulate_ranks=a.emulate_ranks, flags=a.query_flags, stream=stream, next_batch=nxt)
        if a.shard_serial:
            sharded.sync(stream)

    packed_batches = None
    if a.packed_input and with_fused:
        packed_batches = []
        for r, ro in zip(batches, offsets):
            nbs = int(ro[-1].item())
            t = torch.empty(eng.packed_bytes(nbs), dtype=torch.uint8, device=dev)
            eng.pack_bases_device(r.data_ptr(), nbs, t.data_ptr(), stream)
            packed_batches.append((t, nbs))
        torch.cuda.synchronize(dev)

    def step_fused(i):
        r, ro = batches[i % nb], offsets[i % nb]
        if packed_batches is not None:
            t, nbs = packed_batches[i % nb]
            ws.query_device(t.data_ptr(), ro.data_ptr(), B, paired, cands.data_ptr(), ncand.data_ptr(),
                            max_cand=a.max_cand, emulate_ranks=a.emulate_ranks, flags=qflags, stream=stream, packed_bases=nbs)
            return
        ws.query_device(r.data_ptr(), ro.data_ptr(), B, paired, cands.data_ptr(), ncand.data_ptr(),
                        max_cand=a.max_cand, emulate_ranks=a.emulate_ranks, flags=qflags, stream=stream)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(step, on_start=None):
        """W untimed steps, then exactly K steps between barrier+synchronize; max over ranks.  on_start runs after the
        warm-up (the per-kernel event timing is switched on there: its averages cover the timed steps only)."""
        for i in range(a.warmup):
            step(i)
        barrier()
        if on_start is not None:
            on_start()
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(a.warmup + i)
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    # ---- leg 1: fused kernel on the full table: THE path at N=1; "replicas only" (reads split, no collective) at N>1
    fused_elapsed, st, kms, kn = None, None, None, 0
    phase_clocks = None
    if with_fused:
        fused_elapsed = timed(step_fused, lambda: ws.timing(True))
        st = ws.sync()
        kms, kn = ws.kernel_times()
        ws.timing(False)
        phase_clocks = ws.phase_clocks()

    # (state of the sharded leg; roofline() / make_line() are defined BEFORE that leg so that its watchdog can print a line)
    sharded_elapsed, sharded_error, sh_stats, sh_kms, sh_kn, sharded_ok = None, None, None, None, 0, None

    def roofline(kind, ms3, n_batches, stats, elapsed):
        names = KERNELS[kind]
        algo = algorithmic_bytes(max_bases, stats)
        per = [m / n_batches for m in ms3]
        tot = sum(per)
        dom = int(np.argmax(per))
        ach = algo / (tot * 1e-3) / 1e9
        rf = {"bound": "hbm", "kernel": names[dom], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
              "traffic": None,
              "kernel_ms": {names[i]: per[i] for i in range(3)}, "kernel_ms_sum": tot,
              "kernel_note": "HIP events between the path's kernels on their stream (mcq_ws_timing), averaged over %d batches; "
                             "achieved = algorithmic bytes per batch / SUM of the three kernels' times; `kernel` = the one with the largest share" % n_batches,
              "algorithmic_bytes_per_launch": algo, "bytes_per_read": algo / B, "launches_timed": n_batches,
              "per_launch": {k: stats[k] for k in ("n_features", "n_hit_features", "n_locations", "n_cands", "n_overflow", "n_two_class", "n_two_class_retry", "n_narrow_queued") if k in stats}}
        if kind == "sharded":
            rf["kernel_note"] += "; k_shard_sketch / k_shard_lookup and the exchange are not in this sum -- see whole_step"
            rf["whole_step"] = {"achieved": algo / (elapsed / a.steps) / 1e9, "frac": algo / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBS,
                                "note": "algorithmic bytes / ms_per_step (exchange included): the conservative figure for the sharded path"}
        # HBM traffic per launch from the committed PMC passes of this workload (bench.py cannot run rocprofv3 itself)
        tname = a.workload if n_species == 50 and not a.refseq_scale else ({"c2": "refseq", "paired": "refseqp"}.get(a.workload) if a.refseq_scale and n_species == 2600 else None)
        if kind == "fused" and not a.small and tname:
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % tname)))
                # counters belong to the kernels they were collected from: another state of csrc/ -> no figure, said so
                fresh = tj.get("csrc_digest") == pkg.source_digest()
                rf["traffic_fresh"] = fresh
                if not fresh:
                    rf["traffic_source"] = "STALE, not quoted: profiles/pmc_traffic_%s.json was collected from csrc digest %s, this run is %s (scripts/profile.sh + scripts/pmc_traffic.py renew it)" % (
                        tname, tj.get("csrc_digest"), pkg.source_digest())
                    raise LookupError("stale")
                rf["traffic"] = tj["hbm_bytes_per_launch"] * B / tj["reads_per_launch"]
                rf["traffic_source"] = "profiles/pmc_traffic_%s.json: %s" % (tname, tj.get("source", ""))
                if tj.get("valu_insts_per_launch"):
                    # second roofline: wave64 VALU instructions (SQ_INSTS_VALU of the committed PMC pass) over the live kernel
                    # time, against 1 instruction / 4 cycles / SIMD x 1024 SIMDs x 2.4 GHz
                    valu = tj["valu_insts_per_launch"] * B / tj["reads_per_launch"]
                    peak = 1024 * 2.4e9 / 4
                    rf["valu_issue"] = {"achieved": valu / (tot * 1e-3), "peak": peak, "unit": "wave64 VALU inst/s",
                                        "frac": valu / (tot * 1e-3) / peak, "insts_per_read": valu / B}
            except Exception:
                pass
        return rf

    def make_line(mode, sharded_elapsed, sharded_error):
        if mode == "sharded" and sharded_elapsed is None:
            mode = "replicas" if world > 1 else "single"
        elapsed = sharded_elapsed if mode == "sharded" else fused_elapsed
        total_reads = a.steps * B * world          # paired-end: each mate counts (src/printing.cpp:626-627)
        value = total_reads / elapsed
        out = {
            "metric": "query reads/sec (whole node)", "value": value, "unit": "reads/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic" + (" (resident batches in the packed 3-bit form)" if a.packed_input else ""),
            "config": {
                "workload": "%s: %d synthetic genomes (%d species x %d strains, %.1f%% divergence, %.2f Gbp) in HBM, %s per step per GPU, "
                            "k=16 s=16 w=128/113" %
                            ({"c2": ("BASELINE configs[2] shape (RefSeq scale) on one GPU" if a.refseq_scale else "BASELINE configs[1]") if world == 1 else "BASELINE configs[2] shape",
                              "paired": "BASELINE configs[3] shape", "long": "BASELINE configs[4] shape"}[a.workload],
                             n_targets, n_species, a.strains, 100 * a.divergence, db_bp / 1e9,
                             {"c2": "%d x %d bp single-end reads" % (B, L), "paired": "%d reads = %d pairs of 2x%d bp" % (B, B // 2, L),
                              "long": "%d ONT-like reads, mean %d bp, 8%% substitutions" % (B, a.long_mean)}[a.workload]),
                "reads_total": total_reads, "db_keys": n_keys, "db_locations": n_locs,
                "db_hbm_bytes": (db_shard if mode == "sharded" else db).bytes(),
                "db_scaling": "fixed" if (world == 1 or a.db_fixed) else "%d species per GPU: the table grows with N, and with it the locations per read" % a.species,
                "emulate_ranks": a.emulate_ranks, "max_cand": a.max_cand, "P_x_M": a.emulate_ranks * a.max_cand, "distinct_batches": nb,
                "db_targets": n_targets, "db_layout": db_layout,
                "parallelism": {"single": "1 GPU", "replicas": "replicas only (DB replicated, reads split)",
                                "sharded": "feature table hash-range-sharded over %d GPU(s), features to their owners and hits back (%s)" %
                                           (world, "device copy" if world == 1 else ("RCCL send/recv groups" if a.backend == "nccl" else "host-staged gloo rehearsal"))}[mode],
                "setup_s": round(t_setup, 1), "db_build_s": round(t_build, 3), "db_build_parts": n_parts, "genomes_s": round(t_genomes, 1),
                "remove_overpopulated_features": bool(a.remove_overpopulated),
            },
        }
        if sharded_elapsed is not None:
            out["sharded_all_to_all"] = {"value": total_reads / sharded_elapsed, "unit": "reads/s", "ms_per_step": 1e3 * sharded_elapsed / a.steps,
                                         "rccl_ranks": (sh_stats or {}).get("exchange_bytes_per_step", {}).get("rccl_ranks"),
                                         "per_step_per_gpu": sh_stats, "matches_fused_kernel_on_every_rank": sharded_ok,
                                         "note": "feature table hash-range-sharded over %d GPU(s), features and hits exchanged by all-to-all" % world}
        if fused_elapsed is not None and world > 1:
            out["replicas_only"] = {"value": total_reads / fused_elapsed, "unit": "reads/s", "ms_per_step": 1e3 * fused_elapsed / a.steps,
                                    "note": "same reads, table replicated on every GPU, fused kernel, no collective"}
        if sharded_error:
            out["sharded_error"] = sharded_error
        if mode == "sharded" and sh_kn:
            out["roofline"] = roofline("sharded", sh_kms, sh_kn, sh_stats, sharded_elapsed)
            if kn:
                out["roofline"]["fused_kernel_on_replica"] = roofline("fused", kms, kn, st, fused_elapsed)
        elif kn:
            out["roofline"] = roofline("fused", kms, kn, st, fused_elapsed)
        return out

    def setup_sharded():
        """creates the context and connects it; a failure on ANY rank skips the leg on every rank (returns the error text)"""
        nonlocal sharded
        err = None
        try:
            sharded = eng.Shard(db_shard, world, rank, max_queries=nq, max_bases=max_bases, max_seqs=B)
        except Exception as e:
            err = "%s: %s" % (type(e).__name__, str(e)[:300])
        if world > 1:               # nobody enters ncclCommInitRank unless everybody can
            flag = torch.tensor([1 if err else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()):
                sharded = None
                return err or "another rank failed to create its shard context"
            try:
                if a.backend == "nccl":
                    box = [eng.Shard.unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(box, src=0)
                    sharded.comm_rccl(box[0])
                else:       # rehearsal on a box with one GPU: blocks through the host and gloo
                    sys.path.insert(0, os.path.join(ROOT, "tests"))
                    from shard_exchange_gloo import make_gloo_exchange
                    sharded.set_exchange(make_gloo_exchange())
            except Exception as e:
                err = "%s: %s" % (type(e).__name__, str(e)[:300])
            flag = torch.tensor([1 if err else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()):
                sharded = None
                return err or "another rank failed to connect its shard context"
        elif err:
            sharded = None
        elif os.environ.get("MCQ_SHARD_FORCE_RCCL"):
            # one rank, but its blocks through ncclSend / ncclRecv to itself (the engine's test hook): how the RCCL kernels
            # share the GPU with the path's own kernels can be looked at on a box with one GPU
            try:
                sharded.comm_rccl(eng.Shard.unique_id())
            except Exception as e:
                err = "%s: %s" % (type(e).__name__, str(e)[:300]); sharded = None
        return err

    # ---- leg 2: sharded table + all-to-all exchange
    if with_sharded:
        # The exchange runs on real xGMI only in the driver's multi-GPU runs.  Should a rank fail inside it and leave the
        # others waiting in a collective, the watchdog prints the line of the leg already measured and ends every rank
        # with a non-zero status.
        def last_words():
            if rank == 0 and fused_elapsed is not None:
                emit(make_line("replicas" if world > 1 else "single", None,
                               "watchdog: sharded leg did not finish within %d s" % a.sharded_timeout))
            sys.stderr.write("[bench rank %d] sharded leg timed out; exiting\n" % rank)
        dog = arm_watchdog(a.sharded_timeout, last_words, EXIT_SHARDED_FAILED)
        sharded_error = setup_sharded()
        try:
            if sharded is None:
                raise RuntimeError(sharded_error or "no shard context")
            # the first batch of a context runs in the exact mode and learns the block sizes of the padded mode (host
            # round trips); it is an extra untimed step in front of the warmup
            step_sharded(0)
            sharded.sync(stream)
            xb0 = {}

            def start_sharded_clock():
                sharded.timing(True)
                xb0.update(sharded.exchange_bytes())
            sharded_elapsed = timed(step_sharded, start_sharded_clock)
            sh_stats = sharded.sync(stream)
            sh_kms, sh_kn = sharded.kernel_times()
            stage_ms, stage_n = sharded.stage_times()
            sharded.timing(False)
            xb1 = sharded.exchange_bytes()
            nbt = max(1, xb1["batches"] - xb0["batches"])
            per = {k: (xb1[k] - xb0[k]) / nbt for k in ("x1", "x2_ends", "x2_locations", "own_blocks")}
            loc_bytes = (db_shard.layout()["loc_bytes"])
            payload = 4 * sh_stats["n_features"] + 4 * sh_stats["n_features"] + loc_bytes * sh_stats["n_locations"]
            sent = per["x1"] + per["x2_ends"] + per["x2_locations"] + per["own_blocks"]
            sh_stats["exchange_bytes_per_step"] = dict(
                per, rccl_ranks=xb1["rccl_ranks"], block_features=xb1["block_features"], block_locations=xb1["block_locations"],
                to_other_ranks=per["x1"] + per["x2_ends"] + per["x2_locations"],
                padding_share=(1.0 - payload / sent) if sent else None,
                note="bytes this rank hands to the transport per batch: X1 feature blocks, X2 list ends + tile starts, X2 locations (%d-B words); "
                     "own_blocks never leave the device; padding_share = 1 - (features x 4 B out + list ends x 4 B + locations back) / all of it" % loc_bytes)
            sh_stats["stage_ms_per_step"] = dict({k: v / max(1, stage_n) for k, v in stage_ms.items()},
                                                 S3=sum(sh_kms) / max(1, sh_kn),
                                                 note="events around the stages on their streams (they overlap across batches: the sum exceeds ms_per_step); "
                                                      "S1 sketch + route, X1 features out, S2 owner-side lookup, X2 lists back, S3 home-side reduce kernels")
            sh_stats["exchange_block_features_locations"] = list(sharded.caps())
            if with_fused:
                # same batch through the fused kernel on the replicated table: bit-identical results expected on every rank
                last = a.warmup + a.steps - 1
                step_fused(last)
                torch.cuda.synchronize(dev)
                okn = bool(torch.equal(ncand, ncand_s))
                m = torch.arange(a.max_cand, device=dev)[None, :] < ncand[:, None]
                okc = bool(torch.equal(cands[m], cands_s[m]))
                flag = torch.tensor([1 if (okn and okc) else 0], dtype=torch.int32)
                if world > 1:
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                sharded_ok = bool(flag.item())
        except Exception as e:          # keep the line; the replicas leg stands
            sharded_error = sharded_error or "%s: %s" % (type(e).__name__, str(e)[:300])
        dog.cancel()

    out = make_line(mode, sharded_elapsed, sharded_error)
    if phase_clocks and any(phase_clocks):
        out["DIAGNOSTIC_phase_clocks_of_the_workgroup_kernel"] = phase_clocks
    if a.stop_stage:
        out["INVALID_profiling_stop_stage"] = a.stop_stage
    if a.shard_serial:
        out["INVALID_diagnostic_run_without_overlap"] = True
    if world == 1 and mode == "single" and a.workload == "c2" and not a.stop_stage and not a.small and not a.no_pcie_leg:
        # the boundary may hand over HOST buffers (what the reference's readers fill): mcq_query_pipelined keeps two batches
        # in flight -- copy in, compute, copy out on three streams -- from pinned memory; once with ASCII bases (PCIe Gen5 x16
        # carries 157 MB per step) and once with MCQ_BATCH_PACKED (59 MB).  Reported for DESIGN.md, never as `value`.
        nbases = int(offsets[0][-1].item())
        ho = offsets[0].cpu().pin_memory()
        hb = [b.cpu().pin_memory() for b in batches[:4]]
        hp = []
        for b in batches[:4]:
            t = torch.empty(eng.packed_bytes(nbases), dtype=torch.uint8, device=dev)
            eng.pack_bases_device(b.data_ptr(), nbases, t.data_ptr(), stream)
            hp.append(t.cpu().pin_memory())
        hc = [torch.zeros_like(cands, device="cpu").pin_memory() for _ in range(2)]
        hn = [torch.zeros_like(ncand, device="cpu").pin_memory() for _ in range(2)]
        nsteps = max(4, min(a.steps, 16))

        def pipe(src, packed_bases):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            tickets = []
            for i in range(nsteps):
                if i >= 2:
                    ws.wait(tickets[i - 2])                 # its host result buffers are free again
                tickets.append(ws.query_pipelined(src[i % len(src)].data_ptr(), ho.data_ptr(), B, paired, hc[i & 1].data_ptr(), hn[i & 1].data_ptr(),
                                                  max_cand=a.max_cand, emulate_ranks=a.emulate_ranks, packed_bases=packed_bases))
            ws.wait(tickets[-2]); ws.wait(tickets[-1])
            return nsteps * B / (time.perf_counter() - t0)
        pipe(hb, 0)                                          # warm-up: staging buffers, first touch
        r_ascii = pipe(hb, 0)
        r_packed = pipe(hp, nbases)
        # the last pipelined batch against the resident result of the same batch
        j = (nsteps - 1) % len(hp)
        ws.query_device(batches[j].data_ptr(), offsets[0].data_ptr(), B, paired, cands.data_ptr(), ncand.data_ptr(),
                        max_cand=a.max_cand, emulate_ranks=a.emulate_ranks, stream=stream)
        torch.cuda.synchronize(dev)
        same = bool(torch.equal(hn[(nsteps - 1) & 1], ncand.cpu()))
        if same:
            m = torch.arange(a.max_cand)[None, :] < hn[(nsteps - 1) & 1][:, None]
            same = bool(torch.equal(hc[(nsteps - 1) & 1][m], cands.cpu()[m]))
        step_fused(a.warmup + a.steps - 1); torch.cuda.synchronize(dev)      # `cands` holds the last timed batch again (CPU check below)
        out["pcie_inclusive"] = {"value": r_ascii, "unit": "reads/s",
                                 "note": "mcq_query_pipelined from pinned host memory, ASCII bases in + candidates out, copies on their own streams under the kernels"}
        out["pcie_inclusive_packed"] = {"value": r_packed, "unit": "reads/s", "matches_resident_result": same,
                                        "note": "the same with MCQ_BATCH_PACKED bases (3 bits per base; packing not timed)"}
    parity_ok = True
    if want_cpu:
        # the GPU buffers still hold the last timed batch's result (fused leg if it ran, else the sharded leg)
        gc, gn = (cands, ncand) if with_fused else (cands_s, ncand_s)
        if host_table is not None:
            from oracle import mc_oracle as orc
            whole = orc.OracleDb(host_table[0], host_table[1], host_table[2], species.cpu().numpy().astype(np.uint32))
            odb_of = lambda i: whole
            note = ""
        else:
            # table too large for the host: the oracle runs on the part of it this batch can touch (oracle/subtable.py)
            from oracle import mc_oracle as orc
            from oracle import subtable

            def odb_of(i):
                k_, o_, l_ = subtable.batch_subtable(eng, db, batches[i % nb].data_ptr(), offsets[i % nb].data_ptr(), B, dev, tgt_windows)
                return orc.OracleDb(k_, o_, l_, species.cpu().numpy().astype(np.uint32))
            note = "; oracle on the sub-table of each batch's features, read back from the GPU table (oracle/subtable.py)"
        out["cpu_baseline"] = cpu_baseline(a, odb_of, batches, offsets, (a.warmup + a.steps - 1) % nb,
                                           gc, gn, B, paired, bounded=world > 1 or host_table is None, note=note)
        parity_ok = out["cpu_baseline"]["gpu_matches_cpu_on_first_batch"] is not False
    out["timed_region_ms"] = 1e3 * (sharded_elapsed if (mode == "sharded" and sharded_elapsed is not None) else fused_elapsed)
    if (world == 1 and mode == "single" and a.workload == "c2" and not a.small and not a.refseq_scale and a.species == 50 and not a.stop_stage
            and not a.no_cpu_baseline and not a.no_pcie_leg and not a.no_refseq_block and not a.packed_input and a.contigs == 1):
        # the regime north_star targets, in the same driver-timed run (a side block like pcie_inclusive: never `value`): everything of
        # the configs[1] measurement is released first
        try:
            if ws is not None:
                ws.close()
            db.close()
            del batches, offsets, cands, ncand, cands_s, ncand_s
            host_table = None
            torch.cuda.empty_cache()
            out["refseq_scale"] = refseq_side_block(a, eng, synth, torch, dev, stream)
            for k in ("single_end", "paired"):
                if isinstance(out["refseq_scale"].get(k), dict) and out["refseq_scale"][k].get("gpu_matches_cpu") is False:
                    parity_ok = False
        except Exception as e:
            out["refseq_scale"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:400])}
    if rank == 0:
        emit(out)
    rc = 0
    if sharded_error or (with_sharded and sharded_elapsed is None):
        rc = EXIT_SHARDED_FAILED
    elif not parity_ok or sharded_ok is False:
        rc = EXIT_PARITY_FAILED
    if dist is not None:
        if rc == EXIT_SHARDED_FAILED:    # a rank that failed inside the exchange may have left its peers in a collective
            sys.stderr.flush()
            os._exit(rc)
        dist.destroy_process_group()
    sys.exit(rc)


def refseq_side_block(a, eng, synth, torch, dev, stream):
    """The regime north_star targets, timed in the driver's own default run (never `value`): BASELINE configs[2]'s table --
    2 600 species x 10 strains (>= 100 Gbp), every genome two sequences (52 001 targets), one 16 Mbp chromosome, built with
    -remove-overpopulated-features in feature-hash parts -- on this ONE GPU; 8 timed steps of 1 M x 150 bp reads and of 524 288
    pairs of 2 x 150 bp through the fused path, the same through mcq_shard_* at one rank (the N > 1 product path; a rank's own
    blocks never travel), and the first 2^18 reads of a batch of each against the CPU oracle on the sub-table they can touch."""
    from oracle import mc_oracle as orc
    from oracle import subtable
    t0 = time.time()
    torch.cuda.empty_cache()
    free = torch.cuda.mem_get_info(dev)[0]
    if free < 250e9:
        return {"skipped": "needs 250 GB of free HBM, %.0f GB are free" % (free / 1e9)}
    P, M, B, K, W, NB, NC = a.emulate_ranks, a.max_cand, 1 << 20, 8, 2, 4, 1 << 18
    gb, goff, species = synth.make_genomes_big(2600, 10, 2_000_000, 6_000_000, 0.02, seed=3, device=dev, extra_genome=16_000_000)
    goff, species = synth.split_targets(goff, species, 2, keep_last_whole=True)
    n_targets, db_bp = species.numel(), int(goff[-1].item())
    tw = synth.window_counts(goff)
    sets = {"single_end": [synth.sample_reads(gb, goff, B, 150, 0.005, 0.001, seed=1000 + i)[:2] for i in range(NB)],
            "paired": [synth.sample_pairs(gb, goff, B // 2, 150, 300, 500, 0.005, 0.001, seed=2000 + i)[:2] for i in range(NB)]}
    torch.cuda.empty_cache()
    t_b = time.time()
    parts = eng.Parts(gb.data_ptr(), goff.data_ptr(), n_targets, emulate_ranks=P, flags=eng.MCQ_BUILD_REMOVE_OVERPOPULATED, device=dev.index or 0)
    torch.cuda.synchronize(dev)
    t_b = time.time() - t_b
    n_keys, n_locs, n_parts = parts.n_keys, parts.n_locs, parts.n_parts
    del gb
    torch.cuda.empty_cache()
    sp32 = species.to(torch.int32).contiguous()
    db = parts.database(sp32.data_ptr())
    parts.close()
    lay = db.layout(); lay.pop("gw_offsets", None)
    sp = species.cpu().numpy().astype(np.uint32)
    out = {"workload": "BASELINE configs[2] shape (RefSeq scale) on ONE GPU: %d synthetic targets (2600 species x 10 strains x 2 sequences + one 16 Mbp chromosome, "
                       "%.1f Gbp), -remove-overpopulated-features, %d x 150 bp reads / %d pairs of 2x150 bp per step" % (n_targets, db_bp / 1e9, B, B // 2),
           "db_keys": n_keys, "db_locations": n_locs, "db_hbm_bytes": db.bytes(), "db_build_s": round(t_b, 1), "db_build_parts": n_parts,
           "db_layout": {k: lay[k] for k in ("loc_bytes", "loc_format", "bucket_bytes", "slots_per_key", "n_windows")},
           "steps": K, "warmup": W, "emulate_ranks": P, "max_cand": M}
    for name, paired in (("single_end", False), ("paired", True)):
        bt = sets[name]
        nq = B // 2 if paired else B
        max_bases = max(int(o[-1].item()) for _, o in bt)
        cands = torch.zeros((nq, M, 4), dtype=torch.int32, device=dev); ncand = torch.zeros(nq, dtype=torch.int32, device=dev)
        ws = eng.Workspace(db, nq, max_bases)

        def step(i):
            r, o = bt[i % NB]
            ws.query_device(r.data_ptr(), o.data_ptr(), B, paired, cands.data_ptr(), ncand.data_ptr(), max_cand=M, emulate_ranks=P, stream=stream)
        for i in range(W):
            step(i)
        torch.cuda.synchronize(dev)
        ws.timing(True)
        t1 = time.perf_counter()
        for i in range(K):
            step(W + i)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t1
        st = ws.sync()
        kms, kn = ws.kernel_times()
        ws.timing(False)
        algo = algorithmic_bytes(max_bases, st)
        ksum = sum(kms) / max(1, kn)
        blk = {"ms_per_step": 1e3 * el / K, "reads_per_s": K * B / el,
               "roofline": {"bound": "hbm", "achieved": algo / (ksum * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": algo / (ksum * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                            "kernel_ms": {KERNELS["fused"][i]: kms[i] / max(1, kn) for i in range(3)}, "kernel_ms_sum": ksum,
                            "algorithmic_bytes_per_launch": algo, "bytes_per_read": algo / B,
                            "per_launch": {k: st[k] for k in ("n_features", "n_hit_features", "n_locations", "n_cands", "n_overflow", "n_two_class", "n_two_class_retry")}}}
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_refseq%s.json" % ("p" if paired else ""))))
            pkg = importlib.import_module("metacache-mpi_amd")
            if tj.get("csrc_digest") == pkg.source_digest():
                blk["roofline"]["traffic"] = tj["hbm_bytes_per_launch"]
                blk["roofline"]["traffic_source"] = tj.get("source", "")
            else:
                blk["roofline"]["traffic_source"] = "STALE, not quoted: collected from csrc digest %s, this run is %s" % (tj.get("csrc_digest"), pkg.source_digest())
        except Exception:
            pass
        # the oracle on the first NC sequences of the last timed batch (the result buffers still hold it)
        r, o = bt[(W + K - 1) % NB]
        nqc = NC // 2 if paired else NC
        k_, o_, l_ = subtable.batch_subtable(eng, db, r.data_ptr(), o.data_ptr(), NC, dev, tw)
        odb = orc.OracleDb(k_, o_, l_, sp)
        nbytes = int(o[NC].item())
        aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = max(1, min(aff, a.cpu_threads))
        tc = time.perf_counter()
        oc, on = odb.query(r[:nbytes].cpu().numpy().tobytes(), o[:NC + 1].cpu().numpy().astype(np.uint64), paired, max_cand=M, emulate_ranks=P, threads=cores)
        tc = time.perf_counter() - tc
        gc = cands[:nqc].cpu().numpy().view(np.uint32); gn = ncand[:nqc].cpu().numpy().view(np.uint32)
        ok = bool(np.array_equal(gn, on))
        if ok:
            mask = np.arange(M)[None, :] < on[:, None]
            ok = bool(np.array_equal(gc[mask], oc[mask]))
        blk["gpu_matches_cpu"] = ok
        blk["cpu_baseline"] = {"value": NC / tc, "unit": "reads/s", "cores": cores, "kind": "port",
                               "sample": "%d reads of the last timed batch, oracle on the sub-table of their features read back from the GPU table (oracle/subtable.py)" % NC}
        del odb, k_, o_, l_
        ws.close()
        # the same batches through mcq_shard_* at one rank: S1 sketch + route, S2 owner-side lookup into the location block, S3 the SH
        # instantiations of the reduce kernels; block sizes learned from the first (exact) batch
        try:
            sh = eng.Shard(db, 1, 0, max_queries=nq, max_bases=max_bases, max_seqs=B)
            c2 = torch.zeros_like(cands); n2 = torch.zeros_like(ncand)

            def sstep(i, last=False):
                r_, o_2 = bt[i % NB]
                nx = None if last else (bt[(i + 1) % NB][0].data_ptr(), bt[(i + 1) % NB][1].data_ptr(), B)
                sh.query(r_.data_ptr(), o_2.data_ptr(), B, paired, c2.data_ptr(), n2.data_ptr(), max_cand=M, emulate_ranks=P, stream=stream, next_batch=nx)
            for i in range(NB):          # every distinct batch once in the exact mode: the padded sizes cover all of them
                r_, o_2 = bt[i]
                sh.query(r_.data_ptr(), o_2.data_ptr(), B, paired, c2.data_ptr(), n2.data_ptr(), max_cand=M, emulate_ranks=P, stream=stream, exact=True)
                sh.sync(stream)
            for i in range(W):
                sstep(i)
            torch.cuda.synchronize(dev)
            xb0 = sh.exchange_bytes()
            sh.timing(True)
            t1 = time.perf_counter()
            for i in range(K):
                sstep(W + i, last=(i == K - 1))
            torch.cuda.synchronize(dev)
            el_s = time.perf_counter() - t1
            sst = sh.sync(stream)
            skms, skn = sh.kernel_times()
            stage_ms, stage_n = sh.stage_times()
            sh.timing(False)
            xb1 = sh.exchange_bytes()
            same = bool(torch.equal(ncand, n2))
            if same:
                m = torch.arange(M, device=dev)[None, :] < ncand[:, None]
                same = bool(torch.equal(cands[m], c2[m]))
            nbt = max(1, xb1["batches"] - xb0["batches"])
            blk["sharded_one_rank"] = {"ms_per_step": 1e3 * el_s / K, "matches_fused_kernel": same,
                                       "stage_ms_per_step": dict({k: v / max(1, stage_n) for k, v in stage_ms.items()}, S3=sum(skms) / max(1, skn)),
                                       "block_bytes_per_step": {"all_blocks": (xb1["own_blocks"] - xb0["own_blocks"]) / nbt,
                                                                "note": "X1 features + X2 list ends + X2 locations of this rank's own blocks (padded sizes; never travel at one rank); "
                                                                        "at N ranks (N - 1) / N of the same total do"},
                                       "block_features_locations": list(sh.caps()), "n_two_class": sst["n_two_class"], "n_locations": sst["n_locations"]}
            sh.close()
        except Exception as e:          # the fused figures stand
            blk["sharded_one_rank"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        out[name] = blk
        del cands, ncand
    db.close()
    out["seconds"] = round(time.time() - t0, 1)
    return out


def cpu_baseline(a, odb_of, batches, offsets, first, cands, ncand, B, paired, bounded=False, note=""):
    """The oracle (bit-exact CPU restatement of the reference path) timed on this box's
    host cores on a bounded sample of the same workload (whole batches, starting with the
    last timed one, until ~cpu_seconds), and checked against the GPU result of that batch."""
    from oracle import mc_oracle as orc
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(aff, a.cpu_threads))
    nq = B // 2 if paired else B
    total_t, total_n, ok, nb = 0.0, 0, None, len(batches)
    i = first
    while total_t < a.cpu_seconds and total_n < 16 * B:
        rb = batches[i % nb].cpu().numpy().tobytes()
        ro = offsets[i % nb].cpu().numpy().astype(np.uint64)
        odb = odb_of(i)                 # (not timed: the table is given, as the reference's database load is)
        t0 = time.perf_counter()
        oc, on = odb.query(rb, ro, paired, max_cand=a.max_cand, emulate_ranks=a.emulate_ranks, threads=cores)
        total_t += time.perf_counter() - t0
        total_n += B
        if ok is None:          # the GPU buffers still hold this batch's result
            gc = cands[:nq].cpu().numpy().view(np.uint32); gn = ncand[:nq].cpu().numpy().view(np.uint32)
            ok = bool(np.array_equal(gn, on))
            if ok:
                mask = np.arange(a.max_cand)[None, :] < on[:, None]
                ok = bool(np.array_equal(gc[mask], oc[mask]))
        i += 1
        if bounded:
            break
    return {"value": total_n / total_t, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": "%d reads (%d whole batches of the timed workload), %d threads, DB build excluded%s" % (total_n, total_n // B, cores, note),
            "seconds": total_t, "gpu_matches_cpu_on_first_batch": ok}


if __name__ == "__main__":
    main()

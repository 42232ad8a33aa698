#!/usr/bin/env python3
"""The reference ITSELF beside the engine, on bench.py's table (run on the GPU box from the repo root).

The reference binary (oracle/_ref/metacache_mpi: the reference's own sources compiled in the build container, see
oracle/Makefile; it travels with gpurun, /root/reference does not) cannot build a 2 Gbp database in minutes, but it can
READ one: the table bench.py queries is built on the GPU (mcq_build_table), written out as the reference's own shard
files (mcq_refdb_write_shard, one per reference rank, targets split by tgt % P as the reference does), and then

  1. `mpiexec -n P metacache_mpi query <db> r1.fa r2.fa -pairfiles ...`  -- the reference, timed by its own summary line, and
  2. `mcq_query_cli <db> P r1.fa r2.fa ...`                               -- the engine's drop-in CLI on the GPU,

classify the same read pairs from the same files (pairs, BASELINE configs[3]'s shape: the reference's MPI `query`
segfaults on a single read file -- measured here with its own build of a small database -- and on a trailing block in
which a thread gets no reads, so the pair count is a multiple of threads x query-limit).  The two -out files are compared line by line (both are in input order):
the check that the engine answers the reference's headline workload exactly as the reference does, at the size the
bench runs, and the reference's own CPU number on this box's host cores.

Writes gpurun_out/reference_at_scale.json (and prints it).  oracle/ is the checker here; nothing in the product path
uses it.
"""
import argparse
import importlib
import json
import os
import re
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--species", type=int, default=50)
    ap.add_argument("--strains", type=int, default=10)
    ap.add_argument("--genome-min", type=int, default=3_000_000)
    ap.add_argument("--genome-max", type=int, default=5_000_000)
    ap.add_argument("--divergence", type=float, default=0.02)
    ap.add_argument("--reads", type=int, default=1 << 20, help="reads = 2 x pairs")
    ap.add_argument("--query-limit", type=int, default=4096)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--ranks", type=int, default=2, help="reference ranks (mpiexec -n), a power of two")
    ap.add_argument("--threads", type=int, default=8, help="-threads per reference rank")
    ap.add_argument("--maxcand", type=int, default=2)
    ap.add_argument("--mpi-ranks", type=int, default=1, help="ranks of mcq_query_mpi (they share this box's GPU)")
    ap.add_argument("--workdir", default="/tmp/mcq_refscale")
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--checker", default="auto", choices=["auto", "reference", "oracle"],
                    help="what the engine's CLI output is compared with: the reference binary itself (oracle/_ref/metacache_mpi: only where the "
                         "reference has been compiled AND a GPU is present) or the CPU oracle's candidates + the host library's classify "
                         "(auto = the reference if its binary is there)")
    a = ap.parse_args()

    import torch
    ref = os.path.join(ROOT, "oracle", "_ref", "metacache_mpi")
    if a.checker == "auto":
        a.checker = "reference" if os.path.exists(ref) else "oracle"
    if a.checker == "reference" and not os.path.exists(ref):
        sys.exit("oracle/_ref/metacache_mpi is not there: build it in the build container (make -C oracle ref)")
    pkg = importlib.import_module("metacache-mpi_amd")
    pkg.build_hip(); pkg.build_host()
    eng = importlib.import_module("metacache-mpi_amd.engine")
    host = importlib.import_module("metacache-mpi_amd.host")
    synth = importlib.import_module("metacache-mpi_amd.synth")
    dev = torch.device("cuda", 0)
    P = a.ranks
    shutil.rmtree(a.workdir, ignore_errors=True)
    os.makedirs(a.workdir)
    res = {"table": "%d species x %d strains" % (a.species, a.strains), "reference_ranks": P, "threads_per_rank": a.threads, "maxcand": a.maxcand}

    # ---- the table of bench.py (same generator, same seed), built on the GPU
    gb, goff, species = synth.make_genomes(a.species, a.strains, a.genome_min, a.genome_max, a.divergence, seed=3, device=dev)
    table = eng.Table(gb.data_ptr(), goff.data_ptr(), goff.numel() - 1, emulate_ranks=P, device=0)
    keys, loff, locs, _ = table.to_host()
    table.close()
    res["checker"] = a.checker
    n_targets = goff.numel() - 1
    glen = np.diff(goff.cpu().numpy().astype(np.int64))
    sp = species.cpu().numpy().astype(np.int64)
    res.update(db_bp=int(glen.sum()), db_keys=int(len(keys)), db_locations=int(len(locs)), targets=int(n_targets))

    # ---- the pairs of bench.py's first `--workload paired` batch, as two FASTA files
    npairs = a.reads // 2
    blk = a.threads * a.query_limit
    npairs -= npairs % blk
    if npairs == 0:
        sys.exit("--reads must be at least 2 x threads x query-limit")
    r, ro, _ = synth.sample_pairs(gb, goff, npairs, a.read_len, 300, 500, 0.005, 0.001, seed=1000)
    rb = r.cpu().numpy().reshape(npairs, 2, a.read_len)
    reads_bytes, reads_off = r.cpu().numpy().tobytes(), ro.cpu().numpy().astype(np.uint64)       # (for the oracle checker)
    del gb
    torch.cuda.empty_cache()
    hdr = np.array([(">r%08d\n" % i).encode() for i in range(npairs)], dtype="S11")
    for mate in (0, 1):
        rec = np.empty((npairs, 11 + a.read_len + 1), dtype=np.uint8)
        rec[:, :11] = hdr.view(np.uint8).reshape(npairs, 11)
        rec[:, 11:11 + a.read_len] = rb[:, mate, :]
        rec[:, -1] = ord("\n")
        # (not "r1.fa": the reference takes any 5-character file name for FASTQ -- n - 6 wraps to npos, src/sequence_io.cpp:540-542)
        rec.tofile(os.path.join(a.workdir, "reads_%d.fa" % (mate + 1)))
    del rec, hdr
    res["pairs"] = npairs

    # ---- the reference's shard files: taxonomy root > Bacteria > species > one sequence-level taxon per target
    W, S = 128, 113
    nwin = np.where(glen <= W, 1, (glen - W) // S + 1 + (((glen - W) // S + 1) * S < glen))     # src/dna_encoding.h:259-276
    key_of = np.repeat(np.arange(len(keys), dtype=np.int64), np.diff(loff.astype(np.int64)))
    rank_of = (locs >> np.uint64(32)).astype(np.int64) % P
    t0 = time.time()
    for rk in range(P):
        taxa = [dict(id=-(t + 1), parent=1000 + int(sp[t]), rank=0, name="genome_%d strain" % t, file="genomes/all.fna", index=t + 1,
                     windows=int(nwin[t]) if t % P == rk else 0) for t in range(n_targets - 1, -1, -1)]
        taxa.append(dict(id=1, parent=1, rank=20, name="root", file="", index=0, windows=0))
        taxa.append(dict(id=2, parent=1, rank=19, name="Bacteria", file="", index=0, windows=0))
        for s_ in sorted(set(int(x) for x in sp)):
            taxa.append(dict(id=1000 + s_, parent=2, rank=4, name="Synthetica species%d" % s_, file="", index=0, windows=0))
        sel = rank_of == rk
        kk, cnt = np.unique(key_of[sel], return_counts=True)
        o = np.zeros(len(kk) + 1, np.uint64); o[1:] = np.cumsum(cnt)
        host.write_shard(os.path.join(a.workdir, "db.db_%d" % rk),
                         dict(k=16, sketch_size=16, winlen=W, winstride=S, q_k=16, q_sketch_size=16, q_winlen=W, q_winstride=S,
                              max_locs_per_feature=254),
                         taxa, n_targets, keys[kk], o, locs[sel])
    res["shard_files_written_s"] = round(time.time() - t0, 1)
    del key_of, rank_of

    opts = ["-lowest", "species", "-maxcand", str(a.maxcand), "-hitmin", "4", "-hitdiff", "80", "-tophits", "-taxids-only", "-omit-ranks"]
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "mpilib"))

    # ---- 1. the reference (or, where its binary does not travel: the oracle's candidates through the host library's classify)
    if a.checker == "reference":
        t0 = time.time()
        p = subprocess.run(["/opt/conda/bin/mpiexec", "-n", str(P), ref, "query", "db", "reads_1.fa", "reads_2.fa", "-pairfiles", "-threads", str(a.threads),
                            "-query-limit", str(a.query_limit), "-out", "ref.out"] + opts,
                           cwd=a.workdir, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        res["reference_wall_s"] = round(time.time() - t0, 1)
        if p.returncode != 0 or "ABORT" in p.stdout or "FAIL" in p.stdout:         # (the reference exits 0 after an exception, src/main.cpp:91-104)
            print(p.stdout[-3000:]); sys.exit("the reference failed")
        res["reference_stdout_tail"] = p.stdout[-600:]
    else:
        from oracle import mc_oracle as orc
        rdb = host.RefDb(os.path.join(a.workdir, "db"), P)
        odb = orc.OracleDb(keys, loff, locs, rdb.tgt2tax(4))             # -lowest species
        t0 = time.time()
        oc, on = odb.query(reads_bytes, reads_off, True, max_cand=a.maxcand, emulate_ranks=P, quirk_seq_drop=1, threads=a.threads)
        res["oracle_wall_s"] = round(time.time() - t0, 1)
        hitdiff = float(np.float32(np.float32(80) * np.float32(0.01)))
        with open(os.path.join(a.workdir, "ref.out"), "w") as f:          # the layout of -tophits -taxids-only -omit-ranks
            for q in range(npairs):
                c = oc[q, :on[q]]
                best = rdb.classify(c, 4, hitdiff, 19)
                f.write("r%08d\t|\t%s\t|\t%d\n" % (q, ",".join("%d:%d" % (rdb.taxon_id(int(t)), int(h)) for t, h in zip(c[:, 0], c[:, 1])),
                                                   rdb.taxon_id(best) if best != 0xFFFFFFFF else 0))

    # ---- 2. the engine's CLI
    t0 = time.time()
    p2 = subprocess.run([pkg.cli_path(), "db", str(P), "reads_1.fa", "reads_2.fa", "-threads", str(a.threads), "-out", "ours.out"] + opts,
                        cwd=a.workdir, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    res["engine_cli_wall_s"] = round(time.time() - t0, 1)
    if p2.returncode != 0:
        print(p2.stdout[-3000:]); sys.exit("mcq_query_cli failed")

    # ---- 3. the multi-GPU host (mcq_query_mpi: the feature-sharded path behind the C ABI) with the one GPU of this box:
    # one rank, its blocks through RCCL (a communicator of one, MCQ_SHARD_FORCE_RCCL); with --mpi-ranks N > 1 the ranks
    # share the GPU and exchange through MPI_Alltoallv
    have_mpi_cli = os.path.exists(pkg.mpi_cli_path())
    if have_mpi_cli:
        env3 = dict(os.environ, LD_LIBRARY_PATH=pkg.mpi_lib_dir() + ":" + os.environ.get("LD_LIBRARY_PATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if a.mpi_ranks == 1:
            env3["MCQ_SHARD_FORCE_RCCL"] = "1"
        t0 = time.time()
        p3 = subprocess.run(["/opt/conda/bin/mpiexec", "-n", str(a.mpi_ranks), pkg.mpi_cli_path(), "db", str(P), "reads_1.fa", "reads_2.fa",
                             "-threads", str(a.threads), "-transport", "rccl" if a.mpi_ranks == 1 else "mpi", "-out", "ours_mpi.out"] + opts,
                            cwd=a.workdir, env=env3, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        res["engine_mpi_wall_s"] = round(time.time() - t0, 1)
        if p3.returncode != 0:
            print(p3.stdout[-3000:]); sys.exit("mcq_query_mpi failed")

    def parse(path):
        lines, summary = [], {}
        with open(os.path.join(a.workdir, path)) as f:
            for ln in f:
                if ln.startswith("#"):
                    m = re.match(r"#\s*(queries|time|speed):\s*(.*)", ln)
                    if m:
                        summary[m.group(1)] = m.group(2).strip()
                else:
                    lines.append(ln)
        return lines, summary
    rl, rs = parse("ref.out")
    ol, os_ = parse("ours.out")
    res["reference_summary"] = rs
    res["engine_cli_summary"] = os_
    m = re.match(r"([0-9.e+]+)\s*queries/min", rs.get("speed", ""))
    if m:
        res["reference_reads_per_s"] = float(m.group(1)) / 60.0        # a pair counts as two (src/printing.cpp:626-627)
    if have_mpi_cli:
        ml, ms = parse("ours_mpi.out")
        res["engine_mpi_summary"] = ms
        res["engine_mpi_ranks"] = a.mpi_ranks
        res["engine_mpi_identical_after_sorting"] = sorted(ml) == sorted(rl)
    res["mapping_lines"] = [len(rl), len(ol)]
    res["identical_mapping_lines"] = rl == ol
    if rl != ol:
        rs_, os2 = sorted(rl), sorted(ol)
        res["identical_after_sorting"] = rs_ == os2
        bad = [i for i, (x, y) in enumerate(zip(rs_, os2)) if x != y][:3]
        res["first_differences"] = [[rs_[i], os2[i]] for i in bad]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "reference_at_scale.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))
    if not a.keep:
        shutil.rmtree(a.workdir, ignore_errors=True)
    ok = (res["identical_mapping_lines"] or res.get("identical_after_sorting")) and res.get("engine_mpi_identical_after_sorting", True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

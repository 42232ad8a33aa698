#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the reference.

Runs only in the build container (needs /root/reference and oracle/_ref, built by
`make -C oracle ref`).  Nothing of the reference is copied: the fixtures are inputs
(seeded synthetic genomes/reads/taxonomy) and the reference's OUTPUTS on them:

  kat.json            rows 1-5: hash / revcomp / canonical / window / sketch KATs
                      (oracle/_ref/ref_kat = reference headers, hash_int.h,
                      dna_encoding.h, hash_dna.h)
  mini/               10-target DB built by the reference CLI at P = 2, 4, 8
    genomes.fa.gz, nodes.dmp, names.dmp            inputs
    P<p>/mini.db_<r>                                reference-written shard files
    queries.json                                    read pairs (incl. edge cases)
    P<p>/ranks.json.gz                              per-rank M/T/C dumps (ref_query)
    P<p>/final.json                                 CLI -tophits + classification
  tie/                row-11 fold-order case (4 identical genomes, P = 2 vs 4)
  noanc/              target without an ancestor at -lowest species (wire quirk)
  overpop/            -remove-overpopulated-features build
  */P*/cli_*.out.gz   the reference CLI's whole -out file in several output layouts (make_cliout)
  wide/               64 targets at P = 16, 32, 64 with -maxcand 4 (the reference's scripted rank counts);
                      shards + CLI output only

usage: python tests/golden/make_golden.py [--only kat|mini|tie|noanc|overpop|wide|cliout]
"""
import argparse
import gzip
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")
MPILIB = os.path.join(REF, "mpilib")
RANKS = ["sequence", "form", "variety", "subspecies", "species", "subgenus", "genus",
         "subtribe", "tribe", "subfamily", "family", "suborder", "order", "subclass",
         "class", "subphylum", "phylum", "subkingdom", "kingdom", "domain", "root"]


def sh(cmd, **kw):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = MPILIB
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, **kw)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-2000:] + "\n" + r.stderr[-2000:] + "\n")
        raise RuntimeError("command failed: %s" % " ".join(cmd))
    return r.stdout


def ensure_mpilib():
    os.makedirs(MPILIB, exist_ok=True)
    for lib in ["libmpi.so.12", "libgfortran.so.4", "libquadmath.so.0", "libgomp.so.1"]:
        dst = os.path.join(MPILIB, lib)
        if not os.path.lexists(dst):
            os.symlink(os.path.join("/opt/conda/lib", lib), dst)


def rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def mutate(rng, s, rate):
    out = list(s)
    for i in range(len(out)):
        if rng.random() < rate:
            out[i] = rng.choice([c for c in "ACGT" if c != out[i]])
    return "".join(out)


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGTacgtN", "TGCAtgcaN"))


# ------------------------------------------------------------------------- KATs
def make_kat():
    rng = random.Random(20261003)
    cmds, meta = [], []
    xs = [0, 1, 2, 0x12345678, 0xFFFFFFFF, 0x80000000, 0x45d9f3b, 0xdeadbeef] + \
         [rng.getrandbits(32) for _ in range(248)]
    for x in xs:
        cmds.append("H %d" % x); meta.append(("hash", x))
    for x in xs:
        for k in (16, 12, 8, 4, 1):
            xm = x & ((1 << (2 * k)) - 1)
            cmds.append("R %d %d" % (xm, k)); meta.append(("revcomp", xm, k))
            cmds.append("C %d %d" % (xm, k)); meta.append(("canonical", xm, k))
    for n in [0, 1, 15, 16, 17, 100, 127, 128, 129, 150, 226, 240, 241, 242, 250, 300, 354, 355,
              1000, 8000, 8113]:
        cmds.append("W %d 128 113" % n); meta.append(("windows", n, 128, 113))
    for (n, l, s) in [(300, 100, 100), (300, 100, 150), (300, 64, 16), (50, 128, 128), (257, 128, 1)]:
        cmds.append("W %d %d %d" % (n, l, s)); meta.append(("windows", n, l, s))

    seqs = []
    base = ("ACGTTGCATGCCGATAGCTAGCTAGGATCCGATCGATTAGCTAGCTAGCTAGGGCTCTAGAGATCGATCGGC"
            "TAGCTAGCTAGCATCGATCGATTCGAGGCT")
    seqs += [base, base[:50] + "N" + base[51:], base[:20], base.lower(), base[:16], base[:15], "",
             "N" * 128, "A" * 128, "ACGT" * 32, "AC" * 64, "ACGTTGCA" * 16,
             "A" * 60 + "N" + "C" * 67, base[:40] + "RYKM" + base[44:]]
    for n in [16, 17, 31, 37, 64, 100, 113, 127, 128]:
        for _ in range(6):
            seqs.append(rand_seq(rng, n))
    for _ in range(20):   # windows with N / lowercase / IUPAC noise
        s = list(rand_seq(rng, 128))
        for _ in range(rng.randint(1, 6)):
            s[rng.randrange(128)] = rng.choice("NnRYxX-")
        for _ in range(rng.randint(0, 30)):
            j = rng.randrange(128); s[j] = s[j].lower()
        seqs.append("".join(s))
    for _ in range(10):   # low-complexity: duplicate k-mers inside a window
        unit = rand_seq(rng, rng.randint(3, 40))
        seqs.append((unit * 50)[:128])
    for s in seqs:
        for (k, sk) in [(16, 16)] + ([(16, 8), (12, 16), (16, 32), (8, 4)] if len(s) in (128, 102, 37) else []):
            cmds.append("S %d %d %s" % (k, sk, s if s else "-")); meta.append(("sketch", k, sk, s))

    out = sh([os.path.join(REF, "ref_kat")], input="\n".join(cmds) + "\n").split("\n")
    kat = {"hash": [], "revcomp": [], "canonical": [], "windows": [], "sketch": []}
    for m, line in zip(meta, out):
        if m[0] == "hash":
            kat["hash"].append([m[1], int(line)])
        elif m[0] in ("revcomp", "canonical"):
            kat[m[0]].append([m[1], m[2], int(line)])
        elif m[0] == "windows":
            w = [[int(a) for a in t.split(":")] for t in line.split()]
            kat["windows"].append({"n": m[1], "len": m[2], "stride": m[3], "win": w})
        else:
            kat["sketch"].append({"k": m[1], "s": m[2], "seq": m[3], "sketch": [int(t) for t in line.split()]})
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f, separators=(",", ":"))
    print("kat.json:", {k: len(v) for k, v in kat.items()})


# ----------------------------------------------------------------- DB fixtures
def write_taxonomy(d, nodes):
    """nodes: list of (taxid, parent, rank, name)."""
    with open(os.path.join(d, "nodes.dmp"), "w") as f:
        for (t, p, r, _) in nodes:
            f.write("%d\t|\t%d\t|\t%s\t|\t\t|\n" % (t, p, r))
    with open(os.path.join(d, "names.dmp"), "w") as f:
        for (t, _, _, n) in nodes:
            f.write("%d\t|\t%s\t|\t\t|\tscientific name\t|\n" % (t, n))


def write_fasta(path, genomes):
    """genomes: list of (accession, taxid, seq); one file, order = target ids."""
    with open(path, "w") as f:
        for (acc, taxid, seq) in genomes:
            f.write(">%s taxid|%d synthetic\n" % (acc, taxid))
            for i in range(0, len(seq), 80):
                f.write(seq[i:i + 80] + "\n")


def write_fastq(path, names, seqs):
    with open(path, "w") as f:
        for n, s in zip(names, seqs):
            f.write("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)))


def ref_build(work, name, P, extra=()):
    sh([os.path.join("/opt/conda/bin/mpiexec"), "-n", str(P), os.path.join(REF, "metacache_mpi"),
        "build", name, "genomes", "-taxonomy", "tax"] + list(extra), cwd=work)


def parse_tophits(col):
    """'taxname:hits,taxname:hits' or with ids -> list of [taxid, hits]."""
    out = []
    col = col.strip()
    if not col or col == "--":
        return out
    for tok in col.split(","):
        tok = tok.strip()
        if not tok:
            continue
        a, h = tok.rsplit(":", 1)
        out.append([int(a), int(h)])
    return out


def ref_query_cli(work, name, P, maxcand, lowest, extra=(), query_limit=64):
    out = os.path.join(work, "out_P%d.txt" % P)
    if os.path.exists(out):
        os.remove(out)
    sh([os.path.join("/opt/conda/bin/mpiexec"), "-n", str(P), os.path.join(REF, "metacache_mpi"),
        "query", name, "r1.fq", "r2.fq", "-pairfiles", "-lowest", lowest, "-threads", "2",
        "-maxcand", str(maxcand), "-hitmin", "4", "-hitdiff", "80", "-query-limit", str(query_limit),
        "-tophits", "-taxids-only", "-omit-ranks", "-out", out] + list(extra), cwd=work)
    res = {}
    with open(out) as f:
        for line in f:
            if line.startswith("#") or not line.strip():
                continue
            cols = line.rstrip("\n").split("\t|\t")
            hdr = cols[0]
            res[hdr] = {"tophits": parse_tophits(cols[1]), "best": int(cols[2]) if cols[2].strip() not in ("", "--") else 0}
    return res


def ref_ranks_dump(work, name, P, maxcand, lowest, insmax=0):
    txt = sh([os.path.join(REF, "ref_query"), name, str(P), "queries.txt", str(maxcand), lowest, str(insmax)], cwd=work)
    dump = {"params": None, "lineage": {}, "taxa": [], "M": {}, "T": {}, "C": {}}
    for line in txt.split("\n"):
        if not line:
            continue
        t = line.split(" ")
        if t[0] == "P":
            dump["params"] = [int(x) for x in t[1:]]
        elif t[0] == "L":
            dump["lineage"][t[1]] = [int(x) for x in t[2:]]
        elif t[0] == "N":
            dump["taxa"].append([int(t[1]), int(t[2]), " ".join(t[3:])])
        else:
            q, r = t[1], t[2]
            items = [[int(x) for x in it.split(":")] for it in t[4:] if it]
            dump[t[0]].setdefault(q, {})[r] = items
    return dump


def sample_reads(rng, genomes, n_pairs, rlen_lo=100, rlen_hi=150, err=0.01):
    names, r1, r2, truth = [], [], [], []
    for i in range(n_pairs):
        g = rng.randrange(len(genomes))
        seq = genomes[g][2]
        l1, l2 = rng.randint(rlen_lo, rlen_hi), rng.randint(rlen_lo, rlen_hi)
        ins = rng.randint(max(l1, l2), 500)
        p = rng.randrange(0, len(seq) - ins)
        frag = seq[p:p + ins]
        a = mutate(rng, frag[:l1], err)
        b = mutate(rng, revcomp(frag)[:l2], err)
        if rng.random() < 0.5:
            a, b = b, a
        names.append("q%04d_g%d" % (i, g)); r1.append(a); r2.append(b); truth.append(g)
    return names, r1, r2, truth


def run_db_fixture(tag, nodes, genomes, names, r1, r2, Ps, maxcand, lowest, build_extra=(), dumps=True, query_limit=64):
    outdir = os.path.join(HERE, tag)
    shutil.rmtree(outdir, ignore_errors=True)
    os.makedirs(outdir)
    work = tempfile.mkdtemp(prefix="golden_" + tag + "_")
    os.makedirs(os.path.join(work, "genomes")); os.makedirs(os.path.join(work, "tax"))
    write_taxonomy(os.path.join(work, "tax"), nodes)
    write_fasta(os.path.join(work, "genomes", "all.fna"), genomes)
    write_fastq(os.path.join(work, "r1.fq"), names, r1)
    write_fastq(os.path.join(work, "r2.fq"), names, r2)
    with open(os.path.join(work, "queries.txt"), "w") as f:
        for a, b in zip(r1, r2):
            f.write("%s %s\n" % (a if a else "-", b if b else "-"))
    shutil.copy(os.path.join(work, "tax", "nodes.dmp"), outdir)
    shutil.copy(os.path.join(work, "tax", "names.dmp"), outdir)
    with gzip.open(os.path.join(outdir, "genomes.fa.gz"), "wt") as f:
        for (acc, taxid, seq) in genomes:
            f.write(">%s taxid|%d synthetic\n%s\n" % (acc, taxid, seq))
    with open(os.path.join(outdir, "queries.json"), "w") as f:
        json.dump({"names": names, "r1": r1, "r2": r2, "maxcand": maxcand, "lowest": lowest,
                   "hitmin": 4, "hitdiff": 80, "highest": "domain"}, f, separators=(",", ":"))
    for P in Ps:
        pd = os.path.join(outdir, "P%d" % P)
        os.makedirs(pd)
        ref_build(work, tag, P, build_extra)
        for r in range(P):
            if dumps:
                shutil.copy(os.path.join(work, "%s.db_%d" % (tag, r)), pd)
            else:           # many small shard files: stored gzipped (golden_util.Fixture unpacks them)
                with open(os.path.join(work, "%s.db_%d" % (tag, r)), "rb") as fi, \
                        gzip.GzipFile(os.path.join(pd, "%s.db_%d.gz" % (tag, r)), "wb", mtime=0) as fo:
                    fo.write(fi.read())
        dump = ref_ranks_dump(work, tag, P, maxcand, lowest)
        if not dumps:       # many ranks: keep the fixture small, the per-rank dumps of the other fixtures cover rows 7-10
            dump["M"], dump["T"], dump["C"] = {}, {}, {}
        with gzip.open(os.path.join(pd, "ranks.json.gz"), "wt") as f:
            json.dump(dump, f, separators=(",", ":"))
        final = ref_query_cli(work, tag, P, maxcand, lowest, query_limit=query_limit)
        with open(os.path.join(pd, "final.json"), "w") as f:
            json.dump(final, f, separators=(",", ":"))
        ncls = sum(1 for v in final.values() if v["best"])
        print("%s P=%d: %d queries in CLI output, %d classified" % (tag, P, len(final), ncls))
        for r in range(P):
            os.remove(os.path.join(work, "%s.db_%d" % (tag, r)))
    shutil.rmtree(work, ignore_errors=True)


def make_mini():
    rng = random.Random(7)
    # taxonomy: root 1 > domain 2 > phylum 10 > class 20 > order 30 > family 40 >
    #   genus 100 > species 101 (g0,g1), 102 (g2);  genus 200 > species 201 (g3,g4), 202 (g5,g6)
    #   family 41 > genus 300 > species 301 (g7,g8,g9)
    nodes = [(1, 1, "no rank", "root"), (2, 1, "superkingdom", "Bacteria"), (10, 2, "phylum", "Phy"),
             (20, 10, "class", "Cls"), (30, 20, "order", "Ord"), (40, 30, "family", "FamA"),
             (41, 30, "family", "FamB"), (100, 40, "genus", "GenA"), (200, 40, "genus", "GenB"),
             (300, 41, "genus", "GenC"), (101, 100, "species", "GenA one"), (102, 100, "species", "GenA two"),
             (201, 200, "species", "GenB one"), (202, 200, "species", "GenB two"),
             (301, 300, "species", "GenC one")]
    L = 12000
    a0 = rand_seq(rng, L)
    b0 = rand_seq(rng, L)
    c0 = rand_seq(rng, L)
    shared = rand_seq(rng, 1500)           # block present in several genomes (cross-species ties)
    def with_shared(s, pos):
        return s[:pos] + shared + s[pos + len(shared):]
    seqs = [
        (101, a0),                                  # g0
        (101, mutate(rng, a0, 0.03)),               # g1 near-duplicate strain
        (102, with_shared(mutate(rng, a0, 0.12), 3000)),   # g2 sister species + shared block
        (201, with_shared(b0, 5000)),               # g3
        (201, b0[:6000] + rand_seq(rng, 6000)),     # g4 half identical to g3 (exact ties)
        (202, mutate(rng, b0, 0.08)),               # g5
        (202, with_shared(rand_seq(rng, L), 100)),  # g6
        (301, c0),                                  # g7
        (301, c0),                                  # g8 exact duplicate of g7
        (301, mutate(rng, c0, 0.01) + "ACGT" * 400 + "A" * 700),  # g9 with low-complexity tail
    ]
    genomes = [("NC_%06d.1" % (i + 1), t, s) for i, (t, s) in enumerate(seqs)]
    names, r1, r2, _ = sample_reads(rng, genomes, 180)
    # edge cases
    def add(n, a, b):
        names.append(n); r1.append(a); r2.append(b)
    add("e_n_in_read", c0[100:150] + "N" + c0[151:250], revcomp(c0[300:420]))
    add("e_many_n", "N" * 60 + a0[500:560] + "NNNN" + a0[564:600], revcomp(a0[700:800]))
    add("e_short_mate", a0[1000:1130], "ACGTACGTAC")
    add("e_len15", a0[2000:2015], a0[2100:2115])
    add("e_len16", a0[2000:2016], revcomp(a0[2100:2116]))
    add("e_lower", a0[4000:4140].lower(), revcomp(a0[4200:4330]))
    add("e_shared_block", shared[100:250], revcomp(shared[300:450]))
    add("e_shared_block2", shared[700:828], revcomp(shared[900:1050]))
    add("e_dup_g7g8", c0[6000:6150], revcomp(c0[6200:6350]))
    add("e_half_g3g4", b0[1000:1150], revcomp(b0[1200:1350]))
    add("e_lowcomplex", "ACGT" * 37, "A" * 150)
    add("e_random_nohit", rand_seq(rng, 150), rand_seq(rng, 150))
    add("e_len128", a0[7000:7128], revcomp(a0[7200:7328]))
    add("e_len129", a0[7000:7129], revcomp(a0[7200:7329]))
    add("e_len241", a0[8000:8241], revcomp(a0[8300:8541]))
    add("e_long_2k", b0[2000:4000], revcomp(b0[4100:6100]))
    add("e_g9_tail", seqs[9][1][-600:-450], revcomp(seqs[9][1][-400:-250]))
    run_db_fixture("mini", nodes, genomes, names, r1, r2, Ps=(2, 4, 8), maxcand=4, lowest="species")


def make_tie():
    rng = random.Random(11)
    # four identical genomes, species 562,564 (genus 561), 1280,1282 (genus 1279)
    nodes = [(1, 1, "no rank", "root"), (2, 1, "superkingdom", "Bacteria"),
             (561, 2, "genus", "Escherichia"), (1279, 2, "genus", "Staphylococcus"),
             (562, 561, "species", "E one"), (564, 561, "species", "E two"),
             (1280, 1279, "species", "S one"), (1282, 1279, "species", "S two")]
    g = rand_seq(rng, 9000)
    genomes = [("NC_%06d.1" % (i + 1), t, g) for i, t in enumerate([562, 564, 1280, 1282])]
    names, r1, r2, _ = sample_reads(rng, genomes, 40, err=0.0)
    run_db_fixture("tie", nodes, genomes, names, r1, r2, Ps=(2, 4), maxcand=2, lowest="species")


def make_noanc():
    rng = random.Random(13)
    # target 1 hangs directly under a genus: no ancestor at -lowest species, so its
    # candidate keeps the sequence-level taxon (negative id) and is dropped on the wire.
    nodes = [(1, 1, "no rank", "root"), (2, 1, "superkingdom", "Bacteria"),
             (100, 2, "genus", "GenA"), (101, 100, "species", "GenA one"),
             (200, 2, "genus", "GenB"), (201, 200, "species", "GenB one")]
    a, b = rand_seq(rng, 8000), rand_seq(rng, 8000)
    genomes = [("NC_000001.1", 101, a), ("NC_000002.1", 200, b), ("NC_000003.1", 201, mutate(rng, b, 0.05)),
               ("NC_000004.1", 100, mutate(rng, a, 0.05))]
    names, r1, r2, _ = sample_reads(rng, genomes, 60)
    run_db_fixture("noanc", nodes, genomes, names, r1, r2, Ps=(2, 4), maxcand=4, lowest="species")


def make_overpop():
    rng = random.Random(17)
    # build with -remove-overpopulated-features (src/mode_build.cpp:847-1074): a feature whose location counts,
    # summed over the ranks, exceed 253 is removed everywhere.  Tandem repeats of a 113-base unit (= the window
    # stride) put the same 16 features into every window of the repeat:
    #   U: 130 copies in targets 0 and 1  -> 130 + 130 = 260 over two ranks: removed, no rank truncated
    #   V: 100 copies in targets 2 and 3  -> 200: kept
    #   W: 300 copies in target 0         -> truncated to 254 on its rank, 254 > 253: removed
    nodes = [(1, 1, "no rank", "root"), (2, 1, "superkingdom", "Bacteria"),
             (10, 2, "genus", "GenR"), (11, 10, "species", "GenR one"), (12, 10, "species", "GenR two"),
             (20, 2, "genus", "GenS"), (21, 20, "species", "GenS one"), (22, 20, "species", "GenS two")]
    U, V, W = rand_seq(rng, 113), rand_seq(rng, 113), rand_seq(rng, 113)
    genomes = [("NC_000001.1", 11, rand_seq(rng, 3000) + U * 130 + rand_seq(rng, 2000) + W * 300 + rand_seq(rng, 1500)),
               ("NC_000002.1", 12, rand_seq(rng, 2500) + U * 130 + rand_seq(rng, 3000)),
               ("NC_000003.1", 21, rand_seq(rng, 3000) + V * 100 + rand_seq(rng, 3000)),
               ("NC_000004.1", 22, rand_seq(rng, 2000) + V * 100 + rand_seq(rng, 2500))]
    names, r1, r2, _ = sample_reads(rng, genomes, 80)
    run_db_fixture("overpop", nodes, genomes, names, r1, r2, Ps=(2, 4), maxcand=4, lowest="species",
                   build_extra=("-remove-overpopulated-features",))


CLI_VARIANTS = {"default": [], "tophits": ["-tophits"], "lineage": ["-tophits", "-taxids", "-lineage"],
                "idsonly": ["-tophits", "-taxids-only", "-omit-ranks", "-mapped-only"]}


def make_cliout():
    """the reference CLI's whole -out file (parameter lines, table layout, mapping lines, summary) for a committed fixture,
    in several output layouts: tests/golden/<tag>/P<p>/cli_<variant>.out.gz"""
    for tag, P, maxcand in (("mini", 4, 4), ("tie", 2, 2)):
        d = os.path.join(HERE, tag)
        work = tempfile.mkdtemp(prefix="golden_cli_" + tag + "_")
        os.makedirs(os.path.join(work, "genomes")); os.makedirs(os.path.join(work, "tax"))
        shutil.copy(os.path.join(d, "nodes.dmp"), os.path.join(work, "tax"))
        shutil.copy(os.path.join(d, "names.dmp"), os.path.join(work, "tax"))
        with gzip.open(os.path.join(d, "genomes.fa.gz"), "rt") as f, open(os.path.join(work, "genomes", "all.fna"), "w") as o:
            o.write(f.read())
        with open(os.path.join(d, "queries.json")) as f:
            q = json.load(f)
        write_fastq(os.path.join(work, "r1.fq"), q["names"], q["r1"])
        write_fastq(os.path.join(work, "r2.fq"), q["names"], q["r2"])
        ref_build(work, tag, P)
        for name, extra in CLI_VARIANTS.items():
            out = os.path.join(work, "out_%s.txt" % name)
            sh([os.path.join("/opt/conda/bin/mpiexec"), "-n", str(P), os.path.join(REF, "metacache_mpi"),
                "query", tag, "r1.fq", "r2.fq", "-pairfiles", "-lowest", q["lowest"], "-threads", "2",
                "-maxcand", str(maxcand), "-hitmin", "4", "-hitdiff", "80", "-query-limit", "128", "-out", out] + extra, cwd=work)
            with open(out, "rb") as fi, gzip.GzipFile(os.path.join(d, "P%d" % P, "cli_%s.out.gz" % name), "wb", mtime=0) as fo:
                fo.write(fi.read())
            print("%s P=%d cli_%s: %d lines" % (tag, P, name, sum(1 for _ in open(out))))
        shutil.rmtree(work, ignore_errors=True)


def make_wide():
    rng = random.Random(19)
    # the reference's scripted rank counts: -n 32 and -n 64 with -maxcand 4 (script/ft/QueryGeneric_FT.sh:115,
    # script/ft/queries_s4/Run_Query_AFS31_64_8T_S4.sh:2) and -n 16: 64 small targets (every one of 64 ranks
    # owns one: the reference segfaults in `query` when a rank's shard is empty), 16 species x 4 strains in 4 genera,
    # so that a read's candidates sit on many ranks and ties between strains decide the fold
    nodes = [(1, 1, "no rank", "root"), (2, 1, "superkingdom", "Bacteria")]
    for g in range(4):
        nodes.append((100 + g, 2, "genus", "Gen%d" % g))
    genomes = []
    for sp in range(16):
        taxid = 1000 + sp
        nodes.append((taxid, 100 + sp % 4, "species", "Gen%d sp%d" % (sp % 4, sp)))
        anc = rand_seq(rng, 1600)
        if sp % 3 == 1:                      # sister species share half of their sequence with the previous one
            anc = genomes[-1][2][:800] + anc[800:]
        for st in range(4):
            seq = anc if st == 3 and sp % 2 == 0 else mutate(rng, anc, 0.03 if st else 0.0)   # some exact duplicates
            genomes.append(("NC_%06d.1" % (len(genomes) + 1), taxid, seq))
    names, r1, r2, _ = sample_reads(rng, genomes, 150)
    run_db_fixture("wide", nodes, genomes, names, r1, r2, Ps=(16, 32, 64), maxcand=4, lowest="species", dumps=False,
                   query_limit=128)   # one block of 2 threads x 128 reads: at -n 64 the reference segfaults in a trailing block whose second thread gets no reads


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    if not os.path.isdir("/root/reference"):
        sys.exit("needs /root/reference (build container only)")
    ensure_mpilib()
    todo = [a.only] if a.only else ["kat", "mini", "tie", "noanc", "overpop", "wide", "cliout"]
    for t in todo:
        {"kat": make_kat, "mini": make_mini, "tie": make_tie, "noanc": make_noanc, "overpop": make_overpop, "wide": make_wide, "cliout": make_cliout}[t]()

"""mcq_query_cli end to end on the GPU: reference shard files + FASTQ pair in, mapping
lines out; must equal what the reference CLI printed for the same inputs."""
import importlib
import json
import os
import subprocess
import sys

import pytest

from golden_util import Fixture

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tag,P", [("mini", 4), ("tie", 2), ("tie", 4), ("noanc", 2)])
def test_cli_output_equals_reference(tag, P, tmp_path):
    pkg = importlib.import_module("metacache-mpi_amd")
    pkg.build_host()
    fx = Fixture(tag, P)
    for fn, seqs in (("r1.fq", fx.r1), ("r2.fq", fx.r2)):
        with open(tmp_path / fn, "w") as f:
            for n, s in zip(fx.names, seqs):
                f.write("@%s extra words\n%s\n+\n%s\n" % (n, s, "I" * len(s)))
    out = tmp_path / "out.txt"
    prefix = fx.shard_paths[0][: -len(".db_0")]
    r = subprocess.run([pkg.cli_path(), prefix, str(P), str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq"),
                        "-lowest", fx.q["lowest"], "-maxcand", str(fx.maxcand), "-hitmin", str(fx.hitmin),
                        "-hitdiff", str(fx.q["hitdiff"]), "-tophits", "-taxids-only", "-omit-ranks", "-out", str(out)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    lines = [l for l in open(out).read().strip().split("\n") if not l.startswith("#")]
    assert len(lines) == len(fx.names)
    for line in lines:
        name, top, best = line.split("\t|\t")
        ref = fx.final[name]
        mine = [[int(a), int(b)] for a, b in (t.rsplit(":", 1) for t in top.split(",") if t)]
        assert mine == ref["tophits"], (name, line)
        assert int(best) == ref["best"], (name, line)


VARIANTS = {"default": [], "tophits": ["-tophits"], "lineage": ["-tophits", "-taxids", "-lineage"],
            "idsonly": ["-tophits", "-taxids-only", "-omit-ranks", "-mapped-only"]}
BATCHING = {"": [], "batch16": ["-batch", "16"], "bases5000": ["-batch-bases", "5000"],      # how the CLI cuts the reads (mcq_query_pipelined)
            "streamed": []}              # ... and the database opened by the streaming route (MCQ_STREAM_LOAD_MIN_MB=0: include/mcq_open.hpp)


@pytest.mark.parametrize("tag,P", [("mini", 4), ("tie", 2)])
@pytest.mark.parametrize("variant,batching", [(v, "") for v in sorted(VARIANTS)] + [("tophits", "batch16"), ("tophits", "bases5000"), ("tophits", "streamed")])
def test_cli_out_file_equals_the_references_byte_for_byte(tag, P, variant, batching, tmp_path):
    """the whole -out file -- parameter lines, TABLE_LAYOUT, mapping lines in the default rank:name layout and the
    other layouts, the summary with its statistics (src/printing.cpp:622-641, :522-555; src/classification.cpp:583-632)
    -- against the file the reference wrote under mpiexec -n P (tests/golden/*/P*/cli_*.out.gz); compared sorted (the
    reference's line order depends on its threads), the measured values of "# time:" / "# speed:" masked"""
    import gzip
    import re
    pkg = importlib.import_module("metacache-mpi_amd")
    pkg.build_host()
    fx = Fixture(tag, P)
    cwd = tmp_path
    for fn, seqs in (("r1.fq", fx.r1), ("r2.fq", fx.r2)):
        with open(cwd / fn, "w") as f:
            for n, s in zip(fx.names, seqs):
                f.write("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)))
    prefix = fx.shard_paths[0][: -len(".db_0")]
    r = subprocess.run([pkg.cli_path(), prefix, str(P), "r1.fq", "r2.fq", "-lowest", fx.q["lowest"], "-maxcand", str(fx.maxcand),
                        "-hitmin", str(fx.hitmin), "-hitdiff", str(fx.q["hitdiff"]), "-threads", "2", "-out", "out.txt"] + VARIANTS[variant] + BATCHING[batching],
                       cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600,
                       env=dict(os.environ, MCQ_STREAM_LOAD_MIN_MB="0") if batching == "streamed" else None)
    assert r.returncode == 0, r.stderr

    def norm(text):
        text = re.sub(r"^# time:    .*$", "# time:    T ms", text, flags=re.M)
        text = re.sub(r"^# speed:   .*$", "# speed:   S queries/min", text, flags=re.M)
        return sorted(text.split("\n"))
    with gzip.open(os.path.join(os.path.dirname(fx.shard_paths[0]) if tag != "wide" else "", "cli_%s.out.gz" % variant), "rt") as f:
        ref = f.read()
    mine = open(cwd / "out.txt").read()
    assert norm(mine) == norm(ref)


@pytest.mark.parametrize("n_ranks,transport,extra", [(1, "rccl", []), (2, "mpi", []), (3, "mpi", []),
                                                     (1, "rccl", ["-batch", "50"]),            # 4 batches, RCCL, padded after the first
                                                     (2, "mpi", ["-batch", "16"]),             # 7 batches per rank
                                                     (3, "mpi", ["-batch-bases", "6000"]),     # cut by bases: ranks differ in their batch counts
                                                     (2, "mpi", ["STREAMED"])])                # every rank's shard made by the streaming route
def test_mpi_program_writes_the_references_out_file(n_ranks, transport, extra, tmp_path):
    """mcq_query_mpi -- the multi-GPU host in C++ under mpiexec, one hash-range shard of the table per rank, the sharded
    path behind the C ABI -- writes the file the reference wrote under mpiexec -n 4 (emulate_ranks = 4, whatever the
    number of GPU ranks and however the reads are cut into batches).  2 and 3 ranks share the one GPU of the test box, their blocks travel through MPI_Alltoallv
    (-transport mpi); 1 rank takes the RCCL path with a communicator of one (MCQ_SHARD_FORCE_RCCL)."""
    import gzip
    import re
    import shutil
    pkg = importlib.import_module("metacache-mpi_amd")
    pkg.build_host()
    mpiexec = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
    if not os.path.exists(pkg.mpi_cli_path()) or not os.path.exists(mpiexec):
        pytest.skip("no MPI on this box")
    fx = Fixture("mini", 4)
    for fn, seqs in (("r1.fq", fx.r1), ("r2.fq", fx.r2)):
        with open(tmp_path / fn, "w") as f:
            for n, s in zip(fx.names, seqs):
                f.write("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)))
    prefix = fx.shard_paths[0][: -len(".db_0")]
    env = dict(os.environ, LD_LIBRARY_PATH=pkg.mpi_lib_dir() + ":" + os.environ.get("LD_LIBRARY_PATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if n_ranks == 1:
        env["MCQ_SHARD_FORCE_RCCL"] = "1"
    if extra == ["STREAMED"]:
        env["MCQ_STREAM_LOAD_MIN_MB"] = "0"; extra = []
    r = subprocess.run([mpiexec, "-n", str(n_ranks), pkg.mpi_cli_path(), prefix, "4", "r1.fq", "r2.fq", "-lowest", fx.q["lowest"],
                        "-maxcand", str(fx.maxcand), "-hitmin", str(fx.hitmin), "-hitdiff", str(fx.q["hitdiff"]), "-threads", "2",
                        "-tophits", "-transport", transport, "-out", "out.txt"] + extra,
                       cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])

    def norm(text):
        text = re.sub(r"^# time:    .*$", "# time:    T ms", text, flags=re.M)
        text = re.sub(r"^# speed:   .*$", "# speed:   S queries/min", text, flags=re.M)
        return sorted(text.split("\n"))
    with gzip.open(os.path.join(os.path.dirname(fx.shard_paths[0]), "cli_tophits.out.gz"), "rt") as f:
        ref = f.read()
    assert norm(open(tmp_path / "out.txt").read()) == norm(ref)


def test_engine_cli_on_a_gpu_built_table_at_mid_size(tmp_path):
    """scripts/reference_at_scale.py at a size of seconds, with the oracle as the checker (the compiled reference stays in the
    build container: nothing of /root/reference travels to the GPU box): a table built on the GPU, written as the reference's
    shard files, 65 536 read pairs classified by mcq_query_cli (and by mcq_query_mpi where an MPI is installed) from those
    files -- the same mapping lines as the oracle's candidates through the host library's classify.  (The same script with the
    reference's own binary as the checker, at bench size: profiles/r02_reference_at_scale*.json.)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "reference_at_scale.py"), "--checker", "oracle", "--species", "6", "--strains", "4",
                        "--genome-min", "300000", "--genome-max", "500000", "--reads", "131072", "--query-limit", "1024",
                        "--workdir", str(tmp_path / "w")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    res = json.loads(r.stdout[r.stdout.index("{"):])
    assert res["checker"] == "oracle" and res["mapping_lines"] == [65536, 65536]
    assert res["identical_mapping_lines"] or res["identical_after_sorting"]

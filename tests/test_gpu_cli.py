"""mcq_query_cli end to end on the GPU: reference shard files + FASTQ pair in, mapping
lines out; must equal what the reference CLI printed for the same inputs."""
import importlib
import os
import subprocess

import pytest

from golden_util import Fixture

pytestmark = pytest.mark.gpu


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
                        "-hitdiff", str(fx.q["hitdiff"]), "-out", str(out)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    lines = open(out).read().strip().split("\n")
    assert len(lines) == len(fx.names)
    for line in lines:
        name, top, best = line.split("\t|\t")
        ref = fx.final[name]
        mine = [[int(a), int(b)] for a, b in (t.rsplit(":", 1) for t in top.split(",") if t)]
        assert mine == ref["tophits"], (name, line)
        assert int(best) == ref["best"], (name, line)

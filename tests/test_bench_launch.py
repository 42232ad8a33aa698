"""bench.py --gpus N without a launcher must start N ranks itself (the flag was parsed and ignored).  On a box
without a GPU every rank refuses to run; what is checked here is that N fresh rank processes were started (before any
GPU call) and that the failure reaches the exit status."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_spawns_ranks():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-only check")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--small", "--backend", "gloo"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ""                                   # no JSON line without a measurement
    assert r.stderr.count("bench.py needs a GPU") == 2, r.stderr[-2000:]

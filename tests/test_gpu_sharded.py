"""Sharded path with the real HIP stage kernels: 2 or 4 ranks on one GPU, each holding one
hash-range shard, exchange over gloo (host-staged).  Results must equal the oracle's."""
import glob
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("paired,world", [(0, 2), (1, 2), (0, 4)])
def test_shards_on_one_gpu(paired, world):
    with tempfile.TemporaryDirectory() as d:
        outp = os.path.join(d, "res")
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(29700 + paired + 10 * world),
               os.path.join(ROOT, "tests", "sharded_gpu_worker.py"), outp, str(paired)]
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:]
        files = sorted(glob.glob(outp + ".*.npz"))
        assert len(files) == world
        for f in files:
            z = np.load(f)
            assert bool(z["ok"][0]), f
            assert int(z["served"][0]) > 0
            assert int(z["overflow"][0]) >= (4 if paired else 8)     # the long reads took the block path

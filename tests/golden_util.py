"""Loaders for the committed golden fixtures (tests/golden/, made by make_golden.py)."""
import atexit
import gzip
import json
import os
import shutil
import tempfile

import numpy as np

from oracle import dbfile

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_unpacked = {}


def load_kat():
    with open(os.path.join(GOLDEN, "kat.json")) as f:
        return json.load(f)


class Fixture:
    """One DB fixture (mini / tie / noanc) at a given reference rank count P."""

    def __init__(self, tag, P):
        self.tag, self.P = tag, P
        d = os.path.join(GOLDEN, tag)
        with open(os.path.join(d, "queries.json")) as f:
            self.q = json.load(f)
        with gzip.open(os.path.join(d, "P%d" % P, "ranks.json.gz"), "rt") as f:
            self.ranks = json.load(f)
        with open(os.path.join(d, "P%d" % P, "final.json")) as f:
            self.final = json.load(f)
        self.shard_paths = [os.path.join(d, "P%d" % P, "%s.db_%d" % (tag, r)) for r in range(P)]
        if not os.path.exists(self.shard_paths[0]):         # stored gzipped (many-rank fixtures): unpack once per process
            tmp = _unpacked.get((tag, P))
            if tmp is None:
                tmp = _unpacked[(tag, P)] = tempfile.mkdtemp(prefix="golden_%s_P%d_" % (tag, P))
                atexit.register(shutil.rmtree, tmp, True)
                for sp in self.shard_paths:
                    with gzip.open(sp + ".gz", "rb") as fi, open(os.path.join(tmp, os.path.basename(sp)), "wb") as fo:
                        fo.write(fi.read())
            self.shard_paths = [os.path.join(tmp, os.path.basename(sp)) for sp in self.shard_paths]
        self.shards = [dbfile.parse_shard(p) for p in self.shard_paths]
        self.tax = dbfile.Taxonomy(self.shards[0]["taxa"])
        self.n_targets = self.shards[0]["target_count"]
        self.lowest = dbfile.RANK_NAMES.index(self.q["lowest"])
        self.highest = dbfile.RANK_NAMES.index(self.q["highest"])
        self.maxcand = self.q["maxcand"]
        self.hitmin = self.q["hitmin"]
        self.hitdiff = float(np.float32(np.float32(self.q["hitdiff"]) * 0.01)) if self.q["hitdiff"] > 1 else float(self.q["hitdiff"])
        self.names, self.r1, self.r2 = self.q["names"], self.q["r1"], self.q["r2"]
        self.params = self.shards[0]["params"]

    def tgt2tax(self):
        return self.tax.target_keys(self.n_targets, self.lowest)

    def interleaved(self):
        seqs = []
        for a, b in zip(self.r1, self.r2):
            seqs += [a, b]
        return seqs

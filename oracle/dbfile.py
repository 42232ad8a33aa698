"""oracle/dbfile.py -- TEST INFRASTRUCTURE ONLY (see oracle/mc_oracle.cpp header).

numpy restatement of the reference's database shard file layout, used to check the
product's C++ shard reader and to feed the oracle.  Layout (little-endian, no
padding) as written by sketch_database::write (src/sketch_database.h:959-998),
taxonomy/taxon write_binary (src/taxonomy.h:326-335, :680-686), strings/vectors as
u64 length + payload (src/io_serialize.h:48-66) and hash_multimap::serialize
(src/hash_multimap.h:972-1029).
"""
import struct

import numpy as np

DB_VERSION = 20181001
NUM_RANKS = 21            # taxonomy::num_ranks (src/taxonomy.h:97), rank::root == 20
RANK_NAMES = ["sequence", "form", "variety", "subspecies", "species", "subgenus", "genus",
              "subtribe", "tribe", "subfamily", "family", "suborder", "order", "subclass",
              "class", "subphylum", "phylum", "subkingdom", "kingdom", "domain", "root", "none"]
NONE = 0xFFFFFFFF


def parse_shard(path):
    """Returns dict(params, taxa, target_count, keys, off, tgt, win) for one <db>.db_<r>."""
    buf = open(path, "rb").read()
    pos = 0

    def rd(fmt):
        nonlocal pos
        v = struct.unpack_from("<" + fmt, buf, pos)
        pos += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def rstr():
        nonlocal pos
        n = rd("Q")
        s = buf[pos:pos + n].decode("latin-1")
        pos += n
        return s

    ver = rd("Q")
    if ver != DB_VERSION:
        raise ValueError("db version %d" % ver)
    sizes = rd("6B")
    if tuple(sizes[:4]) != (4, 4, 4, 1) or sizes[4] != 8 or sizes[5] != NUM_RANKS:
        raise ValueError("unsupported type widths %r" % (sizes,))
    tk, ts, twin, tstride, qk, qs, qwin, qstride, maxlocs = rd("9Q")
    ntaxa = rd("Q")
    taxa = []
    for _ in range(ntaxa):
        tid, parent, rank = rd("qqB")
        name = rstr()
        fname = rstr()
        index, windows = rd("QQ")
        taxa.append(dict(id=tid, parent=parent, rank=rank, name=name, file=fname, index=index, windows=windows))
    target_count = rd("I")
    keys, lens, tgts, wins = [], [], [], []
    if target_count >= 1:
        nkeys, nvalues = rd("QQ")
        for _ in range(nkeys):
            key, n = rd("IB")
            if n > 0:
                n1 = rd("Q")
                t = np.frombuffer(buf, dtype="<u4", count=n1, offset=pos); pos += 4 * n1
                n2 = rd("Q")
                w = np.frombuffer(buf, dtype="<u4", count=n2, offset=pos); pos += 4 * n2
                assert n1 == n and n2 == n
                keys.append(key); lens.append(n); tgts.append(t); wins.append(w)
        assert sum(lens) == nvalues
    off = np.zeros(len(keys) + 1, dtype=np.uint64)
    if keys:
        off[1:] = np.cumsum(np.asarray(lens, dtype=np.uint64))
    return dict(
        params=dict(k=tk, s=ts, winlen=twin, winstride=tstride, qk=qk, qs=qs, qwinlen=qwin,
                    qwinstride=qstride, maxlocs=maxlocs),
        taxa=taxa, target_count=target_count,
        keys=np.asarray(keys, dtype=np.uint32), off=off,
        tgt=np.concatenate(tgts).astype(np.uint32) if tgts else np.zeros(0, np.uint32),
        win=np.concatenate(wins).astype(np.uint32) if wins else np.zeros(0, np.uint32))


def union_shards(shards):
    """Union of P per-rank tables: per key the merged (tgt,win)-sorted list.

    Each target lives on exactly one rank (tgt % P), so the union list of a key is
    the multiset union of the per-rank lists."""
    ks, ls = [], []
    for s in shards:
        n = np.diff(s["off"]).astype(np.int64)
        ks.append(np.repeat(s["keys"], n))
        ls.append((s["tgt"].astype(np.uint64) << np.uint64(32)) | s["win"].astype(np.uint64))
    k = np.concatenate(ks) if ks else np.zeros(0, np.uint32)
    l = np.concatenate(ls) if ls else np.zeros(0, np.uint64)
    order = np.lexsort((l, k))
    k, l = k[order], l[order]
    keys, start = np.unique(k, return_index=True)
    off = np.append(start, len(k)).astype(np.uint64)
    return keys.astype(np.uint32), off, l


class Taxonomy:
    """Ranked lineages as the reference computes them (taxonomy::ranks,
    src/taxonomy.h:576-597): walk parents from the taxon, record every ancestor
    that has a rank (including the taxon itself) at lineage[rank]."""

    def __init__(self, taxa):
        self.taxa = taxa
        self.by_id = {t["id"]: i for i, t in enumerate(taxa)}
        n = len(taxa)
        self.lineage = np.full((n, NUM_RANKS), NONE, dtype=np.uint32)
        self.rank_of = np.array([t["rank"] for t in taxa], dtype=np.uint8)
        for i, t in enumerate(taxa):
            cur = t["id"]
            while cur != 0:
                j = self.by_id.get(cur)
                if j is None:
                    break
                r = taxa[j]["rank"]
                if r != NUM_RANKS:           # rank::none
                    self.lineage[i, r] = j
                p = taxa[j]["parent"]
                cur = p if p != cur else 0

    def key_of_id(self, tid):
        return self.by_id[tid]

    def target_keys(self, target_count, merge_below_rank):
        """tgt -> taxon key used by candidate insert (src/candidates.h:242-245):
        ancestor at merge_below_rank if it exists, else the sequence-level taxon
        (ids -(tgt+1), src/sketch_database.h:149-150).  A key is the taxon's index
        in the DB's taxon list; bit 31 marks a sequence-level taxon."""
        out = np.zeros(target_count, dtype=np.uint32)
        for t in range(target_count):
            i = self.by_id[-(t + 1)]
            a = self.lineage[i, merge_below_rank] if merge_below_rank > 0 else NONE
            out[t] = a if a != NONE else (0x80000000 | i)
        return out

    def id_of_key(self, key):
        return self.taxa[int(key) & 0x7FFFFFFF]["id"]

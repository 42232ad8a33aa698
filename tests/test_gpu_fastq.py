"""Row f4: raw FASTQ text in HBM -> sequence ranges (mcq_fastq_index) -> mcq_query reading the
bases in place (MCQ_BATCH_RANGES).  Must give exactly what the packed batch gives."""
import importlib

import numpy as np
import pytest
import torch

from golden_util import Fixture
from oracle import dbfile
from oracle import mc_oracle as orc

pytestmark = pytest.mark.gpu


def _fastq(names, seqs, eol="\n", final_newline=True):
    t = "".join("@%s some description%s%s%s+%s%s%s" % (n, eol, s, eol, eol, "@" * len(s), eol) for n, s in zip(names, seqs))
    return t if final_newline else t[: -len(eol)]


def _py_ranges(text):
    lines, pos, out = text.split("\n"), 0, []
    for i, ln in enumerate(lines):
        if i % 4 == 1 and (i < len(lines) - 1):
            out.append((pos, pos + len(ln)))
        pos += len(ln) + 1
    return out


def _index(eng, dev, text, cap):
    tb = torch.from_numpy(np.frombuffer(text.encode(), dtype=np.uint8).copy()).to(dev)
    ranges = torch.zeros(2 * cap, dtype=torch.int64, device=dev)
    n = torch.zeros(1, dtype=torch.int64, device=dev)
    eng.fastq_index(tb.data_ptr(), tb.numel(), ranges.data_ptr(), cap, n.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    return tb, ranges, int(n.item())


@pytest.mark.parametrize("eol,final", [("\n", True), ("\n", False), ("\r\n", True)])
def test_index_matches_line_parser(eol, final):
    eng = importlib.import_module("metacache-mpi_amd.engine")
    dev = torch.device("cuda", 0)
    fx = Fixture("mini", 2)
    seqs = [s if s else "N" for s in fx.r1]                  # FASTQ cannot hold an empty sequence line reliably
    text = _fastq(fx.names, seqs, eol, final)
    tb, ranges, n = _index(eng, dev, text, len(seqs) + 5)
    exp = _py_ranges(text)
    assert n == len(seqs) == len(exp)
    got = ranges.cpu().numpy()[: 2 * n].reshape(-1, 2).tolist()
    assert got == [list(x) for x in exp]
    if eol == "\r\n":                                          # like getline, the '\r' stays in the line
        assert all(text[e - 1] == "\r" for _, e in exp)


@pytest.mark.parametrize("tag,P", [("mini", 4), ("tie", 2)])
def test_query_from_raw_fastq_pair(tag, P):
    eng = importlib.import_module("metacache-mpi_amd.engine")
    dev = torch.device("cuda", 0)
    fx = Fixture(tag, P)
    keys, off, locs = dbfile.union_shards(fx.shards)
    p = fx.params
    db = eng.Database(keys, off, locs, fx.tgt2tax(), k=p["qk"], sketch_size=p["qs"], winlen=p["qwinlen"],
                      winstride=p["qwinstride"], tgt_winstride=p["winstride"])
    r1 = [s if s else "N" for s in fx.r1]; r2 = [s if s else "N" for s in fx.r2]
    t1, t2 = _fastq(fx.names, r1), _fastq(fx.names, r2)
    cap = len(r1) + 1
    tb, ra, n1 = _index(eng, dev, t1 + t2, 2 * cap)            # one buffer, two files back to back
    assert n1 == 2 * len(r1)
    rr = ra[: 4 * len(r1)].reshape(2, len(r1), 2)              # [file][record][begin,end]
    pairs = torch.stack([rr[0], rr[1]], dim=1).reshape(-1).contiguous()   # record q of file 1, then of file 2
    nq = len(r1)
    cands = torch.zeros((nq, fx.maxcand, 4), dtype=torch.int32, device=dev)
    ncand = torch.zeros(nq, dtype=torch.int32, device=dev)
    ws = eng.Workspace(db, nq, tb.numel())
    for flags in (0, eng.MCQ_FORCE_BLOCK_PATH):
        ws.query_device(tb.data_ptr(), pairs.data_ptr(), 2 * nq, True, cands.data_ptr(), ncand.data_ptr(), max_cand=fx.maxcand,
                        emulate_ranks=P, flags=eng.MCQ_QUIRK_SEQ_DROP | flags, stream=torch.cuda.current_stream(dev).cuda_stream, ranges=True)
        ws.sync()
        bases, so = orc.pack_reads([x for ab in zip(r1, r2) for x in ab])
        hc, hn = ws.query_host(bases, so, True, max_cand=fx.maxcand, emulate_ranks=P, flags=eng.MCQ_QUIRK_SEQ_DROP | flags)
        gn = ncand.cpu().numpy().view(np.uint32); gc = cands.cpu().numpy().view(np.uint32)
        assert np.array_equal(gn, hn)
        mask = np.arange(fx.maxcand)[None, :] < hn[:, None]
        assert np.array_equal(gc[mask], hc[mask])
    # and against the reference CLI where the reads were not altered
    for q, name in enumerate(fx.names):
        if fx.r1[q] and fx.r2[q]:
            mine = [[fx.tax.id_of_key(c[0]), int(c[1])] for c in gc[q, :gn[q]]]
            assert mine == fx.final[name]["tophits"]


@pytest.mark.parametrize("shift", [0, 1, 7, 13])
def test_index_of_unaligned_text_and_odd_bytes(shift):
    """the 16-byte fast path needs alignment: shifted buffers take the byte path; bytes 0x0B right after a newline
    (one off '\\n' in the xor domain), runs of newlines and a tail shorter than 16 bytes must all come out exact"""
    eng = importlib.import_module("metacache-mpi_amd.engine")
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    recs = []
    for i in range(3000):
        n = int(rng.integers(1, 200))
        recs.append("@r%d\n%s\n+\x0b\n%s\n" % (i, "".join(rng.choice(list("ACGTN"), n)), "\x0b" * n))
    text = "".join(recs) + "@tail\nACGT"                        # last record incomplete: its sequence line has no newline
    raw = np.frombuffer(text.encode("latin-1"), dtype=np.uint8)
    big = torch.zeros(len(raw) + 64, dtype=torch.uint8, device=dev)
    big[shift:shift + len(raw)] = torch.from_numpy(raw.copy()).to(dev)
    cap = len(recs) + 4
    ranges = torch.zeros(2 * cap, dtype=torch.int64, device=dev)
    n = torch.zeros(1, dtype=torch.int64, device=dev)
    eng.fastq_index(big.data_ptr() + shift, len(raw), ranges.data_ptr(), cap, n.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    exp = _py_ranges(text)
    assert int(n.item()) == len(exp) == len(recs)
    assert ranges.cpu().numpy()[: 2 * len(exp)].reshape(-1, 2).tolist() == [list(x) for x in exp]


@pytest.mark.parametrize("final", [True, False])
def test_two_line_fasta_index_matches_line_parser(final):
    """mcq_fasta_index: '>' header line + one sequence line per record"""
    eng = importlib.import_module("metacache-mpi_amd.engine")
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(9)
    seqs = ["".join(rng.choice(list("ACGTNacgt"), int(rng.integers(1, 300)))) for _ in range(2500)]
    text = "".join(">read%d description\n%s\n" % (i, s) for i, s in enumerate(seqs))
    if not final:
        text = text[:-1] + ""                                   # last sequence line without its newline: not a complete record
    tb = torch.from_numpy(np.frombuffer(text.encode(), dtype=np.uint8).copy()).to(dev)
    cap = len(seqs) + 3
    ranges = torch.zeros(2 * cap, dtype=torch.int64, device=dev)
    n = torch.zeros(1, dtype=torch.int64, device=dev)
    eng.fasta_index(tb.data_ptr(), tb.numel(), ranges.data_ptr(), cap, n.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    lines, pos, exp = text.split("\n"), 0, []
    for i, ln in enumerate(lines):
        if i % 2 == 1 and i < len(lines) - 1:
            exp.append([pos, pos + len(ln)])
        pos += len(ln) + 1
    assert int(n.item()) == len(exp) == (len(seqs) if final else len(seqs) - 1)
    assert ranges.cpu().numpy()[: 2 * len(exp)].reshape(-1, 2).tolist() == exp
    got = [text[a:b] for a, b in exp]
    assert got == seqs[: len(exp)]

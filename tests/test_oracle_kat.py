"""Oracle vs the reference's own headers (tests/golden/kat.json, rows 1-5)."""
from golden_util import load_kat
from oracle import mc_oracle as orc

KAT = load_kat()


def test_survey_kats():
    # SURVEY.md 8a rows 3-4 (probe values from the compiled reference)
    assert orc.tmh(0) == 0 and orc.tmh(1) == 824515495
    assert orc.tmh(0x12345678) == 89967310 and orc.tmh(0xFFFFFFFF) == 539527247
    assert orc.revcomp(0x12345678, 16) == 3530220411 and orc.revcomp(0x1B, 4) == 0x1B


def test_hash():
    for x, h in KAT["hash"]:
        assert orc.tmh(x) == h


def test_revcomp_canonical():
    for x, k, r in KAT["revcomp"]:
        assert orc.revcomp(x, k) == r
    for x, k, r in KAT["canonical"]:
        assert orc.canonical(x, k) == r


def test_windows():
    for w in KAT["windows"]:
        assert orc.windows(w["n"], w["len"], w["stride"]) == [tuple(x) for x in w["win"]], w


def test_sketch():
    assert len(KAT["sketch"]) > 250
    for s in KAT["sketch"]:
        got = orc.sketch(s["seq"], s["k"], s["s"]).tolist()
        assert got == s["sketch"], s

"""The engine does not replay the reference's order-dependent insert() (src/candidates.h:
236-285) candidate by candidate.  It uses this closed form, checked here against the
restated insert() on many random sequences with heavy ties and evictions:

    bounded list after inserting c_1..c_n into an empty list
      ==  the first M of { (taxon, max hits of the taxon, first index reaching that max) }
          ordered by (hits descending, that index ascending)

Why it holds: the minimum of a full list never decreases, so a taxon that was evicted
can only come back with more hits than everything it ever had; and entries with equal
hits are kept in the order in which they reached that value (upper_bound insert, stable
re-sort on update)."""
import numpy as np

from oracle import mc_oracle as orc


def closed_form(tax, hits, M):
    best = {}
    for i, (t, h) in enumerate(zip(tax, hits)):
        if t not in best or h > best[t][0]:
            best[t] = (h, i)
    items = sorted(((h, i, t) for t, (h, i) in best.items()), key=lambda x: (-x[0], x[1]))
    return [(t, h, i) for (h, i, t) in items[:M]]


def test_closed_form_equals_reference_insert():
    rng = np.random.default_rng(12345)
    n_checked = 0
    for trial in range(4000):
        n = int(rng.integers(0, 40))
        ntax = int(rng.integers(1, 9))
        hmax = int(rng.integers(1, 7))
        M = int(rng.integers(1, 9))
        tax = rng.integers(0, ntax, n).astype(np.uint32)
        hits = rng.integers(1, hmax + 1, n).astype(np.uint32)
        ref = orc.insert_sequence(tax, hits, M)
        assert ref == closed_form(tax.tolist(), hits.tolist(), M), (tax, hits, M, ref)
        n_checked += 1
    assert n_checked == 4000


def test_fold_is_the_same_closed_form_on_the_concatenation():
    # a tree-fold step inserts the receiver's list, then the sender's (src/querying.h:910-971)
    rng = np.random.default_rng(7)
    for trial in range(2000):
        M = int(rng.integers(1, 6))
        def rand_list():
            n = int(rng.integers(0, M + 1))
            t = rng.choice(8, size=n, replace=False)
            h = np.sort(rng.integers(1, 6, n))[::-1]
            return [(int(a), int(b)) for a, b in zip(t, h)]
        a, b = rand_list(), rand_list()
        ref = orc.tree_fold([a, b], M)
        seq = a + b
        cf = closed_form([x[0] for x in seq], [x[1] for x in seq], M)
        assert ref == [(t, h) for (t, h, _) in cf], (a, b, M)

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


def _tree_order(P):
    """the order in which the tree fold concatenates the ranks into list 0: one 1-hit entry of its own taxon per rank, room
    for all of them -- ties keep the receiver's entries before the sender's, so the folded list IS that order"""
    out = orc.tree_fold([[(r, 1)] for r in range(P)], max(P, 1))
    return [t for (t, _) in out]


def test_the_tree_keeps_a_prefix_of_the_ranks_in_rank_order():
    # odd ranks send to even ones, every second receiver becomes a sender (src/querying.h:1035-1066): the ranks arrive in
    # list 0 in ascending order, and the ranks a non-power-of-two P never routes to rank 0 are the ones from 2^floor(log2 P) up
    for P in range(1, 65):
        keep = 1 << (P.bit_length() - 1)
        assert _tree_order(P) == list(range(keep)), P


def whole_tree_closed_form(cands, P, M):
    """cands: [(tax, hits, rank)] in candidate order.  ONE selection instead of P lists and a tree: the first M distinct taxa
    in the order (hits descending, rank ascending, candidate index ascending) over the candidates of the ranks the tree keeps."""
    keep = 1 << (P.bit_length() - 1)
    best = {}
    for i, (t, h, r) in enumerate(cands):
        if r >= keep:
            continue
        k = (-h, r, i)
        if t not in best or k < best[t]:
            best[t] = k
    items = sorted((k, t) for t, k in best.items())
    return [(t, -k[0]) for (k, t) in items[:M]]


def test_whole_tree_fold_is_one_selection_in_rank_major_tie_order():
    """Truncating a rank's list to M and folding pairwise loses nothing the final M could contain: an entry that is not in its
    rank's list has M better distinct taxa (or a better entry of its own taxon) in that rank alone, and ties between ranks go
    to the list that was the receiver -- the rank order above.  (Without MCQ_QUIRK_SEQ_DROP, or with it on a table that has no
    sequence-level taxa: a dropped entry has held a slot of an intermediate list, which no single selection reproduces.)"""
    rng = np.random.default_rng(99)
    for trial in range(3000):
        P = int(rng.integers(1, 65)) if trial % 3 else int(2 ** rng.integers(0, 7))
        M = int(rng.integers(1, 17))
        n = int(rng.integers(0, 120))
        ntax = int(rng.integers(1, 30))
        hmax = int(rng.integers(1, 6))
        tax = rng.integers(0, ntax, n).astype(np.uint32)
        hits = rng.integers(1, hmax + 1, n).astype(np.uint32)
        rank = rng.integers(0, P, n)
        lists = []
        for r in range(P):
            sel = np.nonzero(rank == r)[0]
            lists.append([(t, h) for (t, h, _) in orc.insert_sequence(tax[sel], hits[sel], M)])
        ref = orc.tree_fold(lists, M) if P > 1 else lists[0]
        got = whole_tree_closed_form(list(zip(tax.tolist(), hits.tolist(), rank.tolist())), P, M)
        assert ref == got, (P, M, tax, hits, rank, ref, got)

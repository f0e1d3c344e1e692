"""Property tests (hypothesis) of the two oracles against each other on random inputs, and of the algebraic
identities the scan kernel relies on (closed-form IVOM, one-pass KLD)."""
import math

import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import frisk_oracle as O
from oracle import frisk_oracle_np as N

ALPHABET = "ACGTacgtNnRY"


@settings(max_examples=60, deadline=None)
@given(seq=st.text(alphabet=ALPHABET, min_size=30, max_size=260), kmin=st.integers(1, 4), span=st.integers(0, 3),
       w=st.integers(12, 90), inc=st.integers(3, 60), rescue=st.booleans())
def test_numpy_oracle_equals_python_oracle(seq, kmin, span, w, inc, rescue):
    kmax = kmin + span
    recs = [("s", seq), ("t", seq[::-1][: len(seq) // 2])]
    gmaps, gmeta = O.count_kmers(recs, kmin, kmax, genome_mode=True)
    sym, meta = N.genome_profile([s for _, s in recs], kmin, kmax)
    assert np.array_equal(np.asarray(O.flatten(gmaps, kmin, kmax)), sym)
    assert (gmeta["totalLen"], gmeta["exMax"], gmeta["nnTotal"]) == tuple(meta)
    rows_np = N.scan(recs, (sym, meta), kmin, kmax, w, inc, scaffolds_all=rescue)
    rows_py = []
    for win, name, start, stop in O.iter_windows(recs, w, inc, rescue):
        r = O.score_window(win, gmaps, gmeta, kmin, kmax)
        r.update(name=name, start=start, stop=stop)
        rows_py.append(r)
    assert len(rows_np) == len(rows_py)
    for a, b in zip(rows_np, rows_py):
        assert (a["name"], a["start"], a["stop"], a["meta"]) == (b["name"], b["start"], b["stop"], b["meta"])
        assert np.array_equal(a["counts"], np.asarray(O.flatten(b["maps"], kmin, kmax)))
        assert ("error" in a) == ("error" in b)
        if "error" not in a:
            assert abs(a["KLD"] - b["KLD"]) <= 1e-12
        assert a["GC"] == b.get("GC")


@settings(max_examples=200, deadline=None)
@given(counts=st.lists(st.integers(1, 5000), min_size=1, max_size=8), space=st.integers(5100, 9000), kmin=st.integers(1, 3))
def test_ivom_recursion_telescopes_to_closed_form(counts, space, kmin):
    """I_K = a_K p_K + (1-a_K) I_{K-1} with a_x = w_x / W_x  equals  (sum_x w_x p_x) / W_K  (the kernel's form)."""
    counts = sorted(counts, reverse=True)                  # prefix counts never increase with the order
    run, interp = 0, 0.0
    num = 0.0
    for j, c in enumerate(counts):
        x = kmin + j
        weight = c * 4 ** x
        prob = float(c) / ((space - (x - 1)) * 2)
        run += weight
        a = float(weight) / run
        interp = a * prob + ((1 - a) * interp)
        num += float(c) * float(c) * (float(4 ** x) / float((space - (x - 1)) * 2))
    closed = num / float(run)
    assert abs(closed - interp) <= 4e-16 * max(interp, 1e-300) * len(counts) + 1e-300


@settings(max_examples=100, deadline=None)
@given(pairs=st.lists(st.tuples(st.floats(1e-6, 0.5), st.floats(1e-7, 0.5)), min_size=1, max_size=200))
def test_one_pass_kld_identity(pairs):
    """sum Pw log2(Pw/Pg) = (T/Sw - ln Sw + ln Sg) / ln 2 with T = sum Iw ln(Iw/Ig)."""
    iw = np.array([p[0] for p in pairs])
    ig = np.array([p[1] for p in pairs])
    sw, sg = math.fsum(iw), math.fsum(ig)
    two_pass = math.fsum((iw / sw) * (np.log((iw / sw) / (ig / sg)) / math.log(2)))
    t = math.fsum(iw * np.log(iw / ig))
    one_pass = ((t / sw - math.log(sw)) + math.log(sg)) / math.log(2)
    assert abs(one_pass - two_pass) <= 1e-13 * max(1.0, abs(math.log(sw / sg)))


@settings(max_examples=200, deadline=None)
@given(size=st.integers(0, 3000), w=st.integers(1, 500), inc=st.integers(1, 600))
def test_window_enumeration_matches_reference_shape(size, w, inc):
    wins = list(N.iter_windows(size, w, inc))
    if size <= w + ((w * 0.75) - inc):
        assert wins == []
        assert list(N.iter_windows(size, w, inc, True)) == ([(0, size, 1, size)])
    else:
        assert len(wins) == len(range(0, size - inc + 1, inc))
        for j, (a, b, start, stop) in zip(range(0, size - inc + 1, inc), wins):
            if j + w <= size:
                assert (a, b, start, stop) == (j, j + w, j + 1, j + w)
            else:
                assert (b, start, stop) == (size, size - w, size) and a == max(0, 2 * size - w if size < w else size - w)


def test_side_table_index_arithmetic_of_the_scan_kernel():
    """The identities scan8_kernel.h's SIDE form rests on (csrc/scan8_kernel.h, "SIDE"), by brute force over all 4^8 max-mers:
    with t = (c ^ c >> 8) & 0xFF the period-4 max-mers (y)(y) below the (K-2)-mer / (K-1)-mer of c, and c itself, are exactly
    {(c >> 8)(c >> 8)} where t < 16 / t < 4 / t == 0 and none otherwise; the orphan placement's exclusions name exactly the
    period-4 child of a (K-1)-mer and the (K-1)-mers with a period-4 child below a (K-2)-mer; the side count fits the bits
    above the prefix weight."""
    import numpy as np
    c = np.arange(1 << 16, dtype=np.int64)
    period4 = (c >> 8) == (c & 0xFF)                                  # x0..x3 == x4..x7
    t = (c ^ (c >> 8)) & 0xFF
    assert np.array_equal(period4, t == 0) and period4.sum() == 256
    side_code = ((c >> 8) << 8) | (c >> 8)                           # (y)(y), y = the code's first four bases
    # below the (K-2)-mer c >> 4: its sixteen children; below the (K-1)-mer c >> 2: its four
    for shift, bound in ((4, 16), (2, 4)):
        kids = ((c >> shift) << shift)[:, None] + np.arange(1 << shift)[None, :]
        is_p4 = (kids >> 8) == (kids & 0xFF)
        n_p4 = is_p4.sum(axis=1)
        assert np.array_equal(n_p4, (t < bound).astype(np.int64))     # one where t < bound, none elsewhere
        which = np.where(n_p4 == 1, (kids * is_p4).sum(axis=1), -1)
        assert np.array_equal(which[t < bound], side_code[t < bound])
    # orphan (K-1)-mer e (14 bits): its period-4 child, if any, is e << 2 | x3 and exists iff x4 x5 x6 == x0 x1 x2
    e = np.arange(1 << 14, dtype=np.int64)
    kids = (e << 2)[:, None] + np.arange(4)[None, :]
    is_p4 = (kids >> 8) == (kids & 0xFF)
    has = ((e ^ (e >> 8)) & 0x3F) == 0
    assert np.array_equal(is_p4.any(axis=1), has)
    assert np.array_equal(np.argmax(is_p4, axis=1)[has], ((e >> 6) & 3)[has])
    # orphan (K-2)-mer q (12 bits): the (K-1)-mers q << 2 | b below it that hold a period-4 max-mer: b == x2, iff x4 x5 == x0 x1
    q = np.arange(1 << 12, dtype=np.int64)
    grand = (q << 4)[:, None] + np.arange(16)[None, :]
    is_p4 = ((grand >> 8) == (grand & 0xFF)).reshape(len(q), 4, 4).any(axis=2)      # [q, b]
    has = ((q ^ (q >> 8)) & 0xF) == 0
    assert np.array_equal(is_p4.any(axis=1), has)
    assert np.array_equal(np.argmax(is_p4, axis=1)[has], ((q >> 6) & 3)[has])
    # the prefix weight of the orders <= K-3 (at most 5 120 positions per window) leaves nine bits: an 8-bit side count fits
    assert 5120 * sum(4 ** x for x in range(1, 6)) < 1 << 23
    # three workgroups per CU: LDS is handed out in pieces of 1 280 bytes, 128 of them per CU
    assert 3 * -(-53376 // 1280) <= 128 and 3 * -(-49280 // 1280) <= 128 and 2 * -(-81920 // 1280) <= 128

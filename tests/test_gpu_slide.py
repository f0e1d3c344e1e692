"""The sliding order-K table (scan8_kernel.h: inside a chunk of 8 consecutive windows the table of window j + 1 is the
table of window j minus the max-mers that leave plus those that enter) against the compiled CPU oracle and against the
same scan with every window counted afresh - the SAME BITS in every column, KLD included: what a window reads from the
table, and the order in which its terms are summed, do not depend on how the table came about.

FRISK_SCAN_CHUNKS (Engine.scan(chunks=True)) gives small inputs the schedule of a long scan; without it a scan of fewer
than ~12 000 windows is dealt window by window and nothing slides."""
import numpy as np
import pytest

from frisk_amd import _ffi
from frisk_amd.engine import Engine
from oracle import frisk_oracle_c as OC

pytestmark = pytest.mark.gpu

COLS = ("seq_index", "start", "stop", "status", "kld", "gc")
HANDED = []
SIDED = []


def _same_bits(a, b, rip, tag):
    assert len(a) == len(b), tag
    for col in COLS + (("pi", "si", "cri") if rip else ()):
        x, y = getattr(a, col), getattr(b, col)
        keep = a.kept if col not in ("seq_index", "start", "stop", "status") else slice(None)
        assert np.array_equal(x[keep].view(np.uint64 if x.dtype.itemsize == 8 else x.dtype),
                              y[keep].view(np.uint64 if y.dtype.itemsize == 8 else y.dtype)), (tag, col)


def _against_oracle(res, c, tag):
    osym, ometa = OC.genome_profile(c["seqs"], c["kmin"], c["kmax"], False)
    ig = OC.genome_ivom(osym, ometa, c["kmin"], c["kmax"])
    exp = OC.scan(c["seqs"], ig, c["kmin"], c["kmax"], c["w"], c["inc"], scaffolds_all=c["scaffolds_all"], rip=c["rip"])
    k = np.nonzero(res.kept)[0]
    assert len(k) == len(exp["kld"]), tag
    if not len(k):
        return 0
    assert np.array_equal(res.seq_index[k], exp["seq"]) and np.array_equal(res.start[k], exp["start"]), tag
    assert np.array_equal(res.stop[k], exp["stop"]) and np.array_equal(res.gc[k], exp["gc"], equal_nan=True), tag
    zero = (exp["status"] & OC.ROW_ZERO_DIV) != 0
    assert np.array_equal((res.status[k] & _ffi.ROW_ZERO_WEIGHT) != 0, zero), tag
    if c["rip"]:
        for col in ("pi", "si", "cri"):
            assert np.array_equal(getattr(res, col)[k], exp[col], equal_nan=True), tag
    ok = ~zero
    if ok.any():
        assert np.max(np.abs(res.kld[k][ok] - exp["kld"][ok])) <= 1e-11, tag
    return len(k)


def _case(rng):
    kmax = int(rng.choice([6, 7, 8, 8, 8]))
    kmin = int(rng.integers(1, kmax - 2))              # the narrow-counter kernels: kmin <= K - 3
    w = int(rng.choice([600, 1000, 2000, 2048, 3001, 5000, 5120]))
    inc = max(1, int(w * rng.choice([0.013, 0.05, 0.1, 0.2, 0.25, 0.4, 0.49, 0.5])))
    seqs = []
    for _ in range(int(rng.integers(1, 5))):
        n = int(rng.choice([w // 2, w + inc, 3 * w, 9 * inc + w + int(rng.integers(0, inc + 1)), 30 * inc + w, 70 * inc + 17]))
        p = rng.dirichlet([2, 2, 2, 2])
        s = rng.choice(np.frombuffer(b"ATGC", dtype=np.uint8), size=n, p=p)
        for _ in range(int(rng.integers(0, 8))):
            a = int(rng.integers(0, n))
            ln = int(rng.choice([1, 2, 7, 8, 9, 40, inc, w // 3 + 1, w]))
            kind = rng.integers(0, 5)
            if kind == 0:
                s[a:a + ln] = ord("N")
            elif kind == 1:
                s[a:a + ln] |= 0x20
            elif kind == 2:
                s[a:a + ln] = rng.choice(np.frombuffer(b"RYKMnrx-*", dtype=np.uint8), size=len(s[a:a + ln]))
            elif kind == 3:
                s[a:a + ln] = s[a]                                             # poly-X: wraps 4- and 8-bit counters
            else:
                unit = rng.choice(np.frombuffer(b"ATGC", dtype=np.uint8), size=int(rng.integers(2, 5)))
                s[a:a + ln] = np.resize(unit, len(s[a:a + ln]))                # a microsatellite
        seqs.append(s.tobytes())
    return dict(kmin=kmin, kmax=kmax, w=w, inc=inc, seqs=seqs, scaffolds_all=bool(rng.integers(0, 2)),
                rip=bool(rng.integers(0, 2)) and kmin <= 2)


@pytest.mark.parametrize("block", range(8))
def test_sliding_tables_equal_fresh_counts_and_the_oracle(block):
    rng = np.random.default_rng(7300 + block)
    checked = slid = handed = 0
    for case_no in range(16):
        c = _case(rng)
        tag = "block %d case %d: k=%d..%d w=%d i=%d all=%s rip=%s lens=%s" % (
            block, case_no, c["kmin"], c["kmax"], c["w"], c["inc"], c["scaffolds_all"], c["rip"], [len(s) for s in c["seqs"]])
        with Engine(c["kmin"], c["kmax"]) as e:
            e.load(c["seqs"])
            e.profile_reset(); e.profile_add(); e.profile_finalize()
            fresh = e.scan(c["w"], c["inc"], rip=c["rip"], scaffolds_all=c["scaffolds_all"])
            chunked = e.scan(c["w"], c["inc"], rip=c["rip"], scaffolds_all=c["scaffolds_all"], chunks=True)
            n = len(fresh)
            _same_bits(fresh, chunked, c["rip"], tag)
            if c["kmax"] == 8:      # 4-bit counters first (what a long scan of sequence without long repeats does): hand-overs
                narrow = e.scan(c["w"], c["inc"], rip=c["rip"], scaffolds_all=c["scaffolds_all"], chunks=True, bits4=True)
                _same_bits(fresh, narrow, c["rip"], tag + " (4-bit bulk)")
                handed_plain = e.scan_stat()[1]
                handed += handed_plain
                # ... and with the side table for the max-mers of period <= 4 beside them (what a long scan of repeat-rich sequence does)
                sided = e.scan(c["w"], c["inc"], rip=c["rip"], scaffolds_all=c["scaffolds_all"], chunks=True, side4=True)
                _same_bits(fresh, sided, c["rip"], tag + " (4-bit bulk + side table)")
                if n > 0:
                    assert e.scan_side() and e.scan_stat()[0] == 4, tag
                    SIDED.append((e.scan_stat()[1], handed_plain))
            if n > 12:       # another range: the chunks start elsewhere, other windows are slid into
                c0 = int(rng.integers(1, 8))
                part = e.scan(c["w"], c["inc"], rip=c["rip"], scaffolds_all=c["scaffolds_all"], c0=c0, c1=n - 1, chunks=True)
                for col in COLS:
                    x, y = getattr(part, col), getattr(fresh, col)[c0:n - 1]
                    assert np.array_equal(x.view(np.uint64 if x.dtype.itemsize == 8 else x.dtype),
                                          y.view(np.uint64 if y.dtype.itemsize == 8 else y.dtype), ), (tag, col, "range")
            if case_no % 4 == 0 and n <= 400:      # the count tables themselves, window by window
                dbg_f = e.scan(c["w"], c["inc"], scaffolds_all=c["scaffolds_all"], debug=True)
                dbg_c = e.scan(c["w"], c["inc"], scaffolds_all=c["scaffolds_all"], debug=True, chunks=True)
                k = dbg_f.kept
                assert np.array_equal(dbg_f.counts[k], dbg_c.counts[k]) and np.array_equal(dbg_f.meta[k], dbg_c.meta[k]), tag
        checked += _against_oracle(chunked, c, tag)
        slid += 2 * c["inc"] <= c["w"] - (c["kmax"] - 1)
    assert checked > 150 and slid >= 8
    HANDED.append(handed)


def test_some_windows_overflowed_four_bits():
    """(runs after the blocks above) the poly-X / microsatellite inserts did wrap 4-bit counters somewhere: the hand-over chain
    4-bit -> 8-bit -> 16-bit was part of what was compared."""
    assert not HANDED or sum(HANDED) > 0
    # ... and the side table kept most of them on the 4-bit form (poly-X and microsatellites of period <= 4 are what the cases insert)
    if SIDED:
        with_side, plain = sum(a for a, _ in SIDED), sum(b for _, b in SIDED)
        assert with_side < plain, (with_side, plain)


def test_long_handover_lists_of_satellite_arrays():
    """Satellite arrays hand EVERY window they cover from the 4-bit to the 8-bit form: thousands of list entries in runs of
    consecutive windows.  Same rows as the C oracle for every window, and as scans of sub-ranges (short lists), bit for bit."""
    from oracle import frisk_oracle_c as OC
    from frisk_amd import Engine
    lens = [9_000_000, 2_500_000]
    kw = dict(island_frac=0.02, n_frac=0.03, lower_frac=0.0, repeats_per_kb=0.35, period_mix=0.25, sat_frac=0.25)
    with Engine(1, 8) as e:
        e.synth(lens, seed=77, **kw)
        e.profile_reset(); e.profile_add(); e.profile_finalize()
        res = e.scan(5000, 1000, rip=True, bits4=True)
        handed8, handed16 = e.scan_stat()[1:3]
        n = len(res)
        assert handed8 > max(1024, n // 48), (handed8, n)
        S = OC.Seqs([e.read_seq(q) for q in range(len(lens))])
        osym, ometa = OC.genome_profile(S, 1, 8)
        exp = OC.scan(S, OC.genome_ivom(osym, ometa, 1, 8), 1, 8, 5000, 1000, rip=True)
        k = np.nonzero(res.kept)[0]
        assert len(k) == len(exp["kld"]) and np.array_equal(res.start[k], exp["start"]) and np.array_equal(res.gc[k], exp["gc"])
        assert np.max(np.abs(res.kld[k] - exp["kld"])) <= 1e-11
        for col in ("pi", "si", "cri"):
            assert np.array_equal(getattr(res, col)[k], exp[col], equal_nan=True)
        step = 700
        for c0 in range(0, min(n, 5600), step):
            part = e.scan(5000, 1000, rip=True, bits4=True, c0=c0, c1=min(n, c0 + step))
            for f in ("start", "stop", "status", "kld", "gc", "pi", "si", "cri"):
                assert np.array_equal(getattr(part, f), getattr(res, f)[c0:c0 + step], equal_nan=True), (f, c0)


@pytest.mark.parametrize("factor", [0.4994, 0.5, 0.55, 0.6])
def test_increments_around_half_a_window(factor):
    """Increments from just below half a window (the last geometry whose tables slide) to 0.6 w (the CLI's default -i 2500 at w = 5000:
    every window counted afresh, in chunks): same bits as the window-by-window scan, in every bulk form, on a sub-range, and the
    oracle's rows."""
    rng = np.random.default_rng(int(factor * 10000))
    checked = 0
    for case_no in range(6):
        c = _case(rng)
        c["kmax"] = 8
        c["kmin"] = int(rng.integers(1, 6))
        c["rip"] = c["rip"] and c["kmin"] <= 2
        c["w"] = int(rng.choice([2000, 5000, 5120, 3001]))
        c["inc"] = int(c["w"] * factor)
        seqs = []
        for q in range(int(rng.integers(1, 4))):
            n = int(rng.choice([c["w"] + c["inc"], 9 * c["inc"] + c["w"] + 11, 40 * c["inc"] + 17]))
            s = rng.choice(np.frombuffer(b"ATGC", dtype=np.uint8), size=n, p=rng.dirichlet([2, 2, 2, 2]))
            for _ in range(int(rng.integers(0, 6))):
                a = int(rng.integers(0, n))
                ln = int(rng.choice([1, 8, 40, c["inc"] // 2, c["w"] // 3]))
                if rng.integers(0, 2):
                    s[a:a + ln] = ord("N")
                else:
                    s[a:a + ln] = np.resize(np.frombuffer(b"CA", dtype=np.uint8), len(s[a:a + ln]))
            seqs.append(s.tobytes())
        c["seqs"] = seqs
        tag = "factor %s case %d: k=%d..8 w=%d i=%d all=%s lens=%s" % (factor, case_no, c["kmin"], c["w"], c["inc"], c["scaffolds_all"],
                                                                       [len(s) for s in seqs])
        with Engine(c["kmin"], 8) as e:
            e.load(seqs)
            e.profile_reset(); e.profile_add(); e.profile_finalize()
            fresh = e.scan(c["w"], c["inc"], rip=c["rip"], scaffolds_all=c["scaffolds_all"])
            for kw in (dict(), dict(bits4=True), dict(side4=True)):
                got = e.scan(c["w"], c["inc"], rip=c["rip"], scaffolds_all=c["scaffolds_all"], chunks=True, **kw)
                _same_bits(fresh, got, c["rip"], tag + " " + str(kw))
            n = len(fresh)
            if n > 12:
                c0 = int(rng.integers(1, 8))
                part = e.scan(c["w"], c["inc"], rip=c["rip"], scaffolds_all=c["scaffolds_all"], c0=c0, c1=n - 1, chunks=True, bits4=True)
                for col in COLS:
                    x, y = getattr(part, col), getattr(fresh, col)[c0:n - 1]
                    assert np.array_equal(x.view(np.uint64 if x.dtype.itemsize == 8 else x.dtype),
                                          y.view(np.uint64 if y.dtype.itemsize == 8 else y.dtype)), (tag, col, "range")
            checked += _against_oracle(fresh, c, tag)
    assert checked > 60

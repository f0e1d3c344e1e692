"""Vectorised (numpy) CPU oracle for the frisk hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Same algorithm as oracle/frisk_oracle.py (the reference-shaped restatement, pinned to the golden
vectors of the reference's own functions), restated on integer arrays so that inputs of a few Mb
finish in seconds.  It is pinned twice: against the golden vectors directly and against
frisk_oracle.py on random inputs (tests/test_oracle_np.py).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.

Citations: /root/reference/frisk/__init__.py (L<n>).
"""
import math

import numpy as np

_LUT_CODE = np.zeros(256, dtype=np.uint8)
_LUT_VALID = np.zeros(256, dtype=bool)       # A/C/G/T in either case  (word.upper() in map, L334-341)
_LUT_UPPER = np.zeros(256, dtype=bool)       # uppercase A/C/G/T only  (countN / calcGC, L106-137)
for _i, _ch in enumerate("ATGC"):            # digit order of the canonical index, L70
    for _c in (_ch, _ch.lower()):
        _LUT_CODE[ord(_c)] = _i
        _LUT_VALID[ord(_c)] = True
    _LUT_UPPER[ord(_ch)] = True


def table_offset(kmin, x):
    return (4 ** x - 4 ** kmin) // 3


def profile_len(kmin, kmax):
    return table_offset(kmin, kmax + 1)


class Encoded:
    """A sequence as digit / validity / case arrays."""

    def __init__(self, seq):
        if isinstance(seq, str):
            seq = seq.encode("ascii")
        raw = np.frombuffer(bytes(seq), dtype=np.uint8)
        self.n = raw.size
        self.code = _LUT_CODE[raw].astype(np.int64)
        self.valid = _LUT_VALID[raw]
        self.upper = _LUT_UPPER[raw]

    def slice(self, a, b):
        e = Encoded.__new__(Encoded)
        e.code, e.valid, e.upper = self.code[a:b], self.valid[a:b], self.upper[a:b]
        e.n = e.code.size
        return e


def _word_codes(code, ok, x):
    """(codes, good): code of the x-mer starting at every position j <= n-x, and whether all of its
    bases satisfy `ok`."""
    n = code.size
    if n < x:
        return np.zeros(0, np.int64), np.zeros(0, bool)
    m = n - x + 1
    c = np.zeros(m, dtype=np.int64)
    bad = np.zeros(m, dtype=np.int64)
    for p in range(x):
        c = (c << 2) | code[p:p + m]
        bad += ~ok[p:p + m]
    return c, bad == 0


def forward_counts(enc, kmin, kmax, ok=None):
    """int64[profile_len]: forward counts of every order (one-base step, invalid words skipped, L327-348),
    plus the number of skipped max-mer positions (exMax, L344-345)."""
    ok = enc.valid if ok is None else ok
    out = np.zeros(profile_len(kmin, kmax), dtype=np.int64)
    ex_max = 0
    for x in range(kmin, kmax + 1):
        c, good = _word_codes(enc.code, ok, x)
        out[table_offset(kmin, x):table_offset(kmin, x + 1)] = np.bincount(c[good], minlength=4 ** x)
        if x == kmax:
            ex_max = int(good.size - good.sum())
    return out, ex_max


def revcomp_index(x):
    """index permutation c -> reverse complement of c, for x-mers (A<->T, G<->C is XOR 1 per digit)."""
    c = np.arange(4 ** x, dtype=np.int64)
    r = np.zeros_like(c)
    t = c.copy()
    for _ in range(x):
        r = (r << 2) | ((t & 3) ^ 1)
        t >>= 2
    return r


def genome_profile(seqs, kmin, kmax, mask_host=False):
    """computeKmers(genomeMode=True) (L280-367): symmetric counts + (totalLen, exMax, nnTotal)."""
    fwd = np.zeros(profile_len(kmin, kmax), dtype=np.int64)
    total = ex = nn = 0
    for s in seqs:
        e = s if isinstance(s, Encoded) else Encoded(s)
        ok = (e.valid & e.upper) if mask_host else e.valid       # --maskHost: no .upper(), L336-337
        f, x = forward_counts(e, kmin, kmax, ok)
        fwd += f
        ex += x
        total += e.n
        nn += int(e.n - e.upper.sum())
    sym = fwd.copy()
    for x in range(kmin, kmax + 1):
        o = table_offset(kmin, x)
        sym[o:o + 4 ** x] += fwd[o:o + 4 ** x][revcomp_index(x)]  # L350-351
    return sym, (total, ex, nn)


def raw_profile(seqs, kmin, kmax, mask_host=False, ranges=None):
    """The library's linear ("raw") profile state, restated: position p contributes one count to the
    table of order r = min(run, K) at its longest valid word; followed by {totalLen, #K-mer start
    positions, nnTotal, 0}.  ranges = optional per-sequence (a, b): only start positions in [a, b).
    Used to test the multi-GPU all-reduce path on the CPU."""
    raw = np.zeros(profile_len(kmin, kmax) + 4, dtype=np.int64)
    for si, s in enumerate(seqs):
        e = s if isinstance(s, Encoded) else Encoded(s)
        ok = (e.valid & e.upper) if mask_host else e.valid
        n = e.n
        a, b = (0, n) if ranges is None else (max(0, ranges[si][0]), min(n, ranges[si][1]))
        if b <= a:
            continue
        run = np.zeros(n + 1, dtype=np.int64)
        for p in range(n - 1, -1, -1):                           # plain loop: small inputs only
            run[p] = run[p + 1] + 1 if ok[p] else 0
        run = np.minimum(run[:n], kmax)
        inside = np.zeros(n, dtype=bool)
        inside[a:b] = True
        for x in range(kmin, kmax + 1):
            sel = np.nonzero((run == x) & inside)[0]
            if sel.size:
                c = np.zeros(sel.size, dtype=np.int64)
                for p in range(x):
                    c = (c << 2) | e.code[sel + p]
                raw[table_offset(kmin, x):table_offset(kmin, x + 1)] += np.bincount(c, minlength=4 ** x)
        raw[-4] += b - a
        raw[-3] += max(0, min(b, n - kmax + 1) - a)
        raw[-2] += int((b - a) - e.upper[a:b].sum())
    return raw


def finalize_raw(raw, kmin, kmax):
    """raw -> (sym, (totalLen, exMax, nnTotal)): marginalise, then add reverse complements."""
    n = profile_len(kmin, kmax)
    cnt = raw[:n].copy()
    for x in range(kmax - 1, kmin - 1, -1):
        o, o1 = table_offset(kmin, x), table_offset(kmin, x + 1)
        cnt[o:o + 4 ** x] += cnt[o1:o1 + 4 ** (x + 1)].reshape(-1, 4).sum(axis=1)
    sym = cnt.copy()
    for x in range(kmin, kmax + 1):
        o = table_offset(kmin, x)
        sym[o:o + 4 ** x] += cnt[o:o + 4 ** x][revcomp_index(x)]
    top = cnt[table_offset(kmin, kmax):n].sum()
    return sym, (int(raw[-4]), int(raw[-3] - top), int(raw[-2]))


def genome_ivom_table(sym, meta, kmin, kmax):
    """Un-normalised genome-side IVOM of every max-mer (L411-450); NaN where the reference divides by 0."""
    space = meta[0] - meta[2]
    k = np.arange(4 ** kmax, dtype=np.int64)
    W = np.zeros(k.size, dtype=np.int64)
    I = np.zeros(k.size, dtype=np.float64)
    bad = np.zeros(k.size, dtype=bool)
    with np.errstate(divide="ignore", invalid="ignore"):
        for x in range(kmin, kmax + 1):
            c = sym[table_offset(kmin, x) + (k >> (2 * (kmax - x)))]
            wt = c << (2 * x)
            W = W + wt
            D = (space - (x - 1)) * 2
            bad |= (W == 0) | (D == 0)
            p = c.astype(np.float64) / np.float64(D) if D != 0 else np.zeros(k.size)
            a = wt.astype(np.float64) / W.astype(np.float64)
            I = a * p if x == kmin else a * p + ((1.0 - a) * I)
    I[bad] = np.nan
    return I


def score_window(win, ig, kmin, kmax, rip=False):
    """One iteration of the scan loop L1478-1494 on an Encoded window.
    Returns dict(counts, meta, KLD | error, GC, RIP)."""
    counts, ex_max = forward_counts(win, kmin, kmax)
    n = win.n
    up = win.upper
    S = int(up.sum())
    row = {"counts": counts, "meta": [n, ex_max, n - S]}
    top = counts[table_offset(kmin, kmax):]
    present = np.nonzero(top)[0]
    if present.size == 0:
        row["KLD"] = 0
    else:
        W = np.zeros(present.size, dtype=np.int64)
        I = np.zeros(present.size, dtype=np.float64)
        zero_div = bool(np.isnan(ig[present]).any())
        for x in range(kmin, kmax + 1):
            c = counts[table_offset(kmin, x) + (present >> (2 * (kmax - x)))]
            wt = c << (2 * x)
            W = W + wt
            D = (S - (x - 1)) * 2
            if D == 0:
                zero_div = True
                break
            p = c.astype(np.float64) / np.float64(D)
            a = wt.astype(np.float64) / W.astype(np.float64)
            I = a * p if x == kmin else a * p + ((1.0 - a) * I)
        if zero_div:
            row["error"] = "ZeroDivisionError"
        else:
            g = ig[present]
            pw = I / I.sum()
            pg = g / g.sum()
            nz = pg != 0
            row["KLD"] = float(np.sum(pw[nz] * (np.log(pw[nz] / pg[nz]) / math.log(2.0))))
    gc = int((up & (win.code >= 2)).sum())
    row["GC"] = float(gc) / S if S else None
    if rip and kmin <= 2 <= kmax:
        di = counts[table_offset(kmin, 2):table_offset(kmin, 3)]
        nan = float("nan")
        AT, TA, TG, GT, CA, AC = (int(di[i]) for i in (1, 4, 6, 9, 12, 3))
        pi = TA / float(AT) if AT > 0 else nan
        si = (CA + TG) / float(AC + GT) if (AC + GT) > 0 else nan
        row["RIP"] = [pi, si, (pi - si) if (pi and si) else nan]
    return row


def iter_windows(size, w, i, scaffolds_all=False):
    """(a, b, start, stop) for every candidate window of a scaffold of `size` bases, BEFORE the N filter:
    [a, b) = the slice of the scaffold, (start, stop) = the coordinates the reference reports (L194-251)."""
    if size <= w + ((w * 0.75) - i):
        if scaffolds_all:
            yield 0, size, 1, size
        return
    for j in range(0, size - i + 1, i):
        if j + w > size:
            a = size - w
            yield (max(0, a + size) if a < 0 else a), size, size - w, size     # negative start: Python slice
        else:
            yield j, j + w, j + 1, j + w


def scan(records, profile, kmin, kmax, w, i, scaffolds_all=False, rip=False):
    """Phase B over (name, sequence) records with a finished profile (sym, meta).  Rows as dicts."""
    sym, meta = profile
    ig = genome_ivom_table(sym, meta, kmin, kmax)
    rows = []
    for name, seq in records:
        enc = Encoded(seq)
        for a, b, start, stop in iter_windows(enc.n, w, i, scaffolds_all):
            win = enc.slice(a, b)
            nn = win.n - int(win.upper.sum())
            if nn >= 0.3 * win.n:                                   # L213 / L238
                continue
            row = score_window(win, ig, kmin, kmax, rip)
            row.update(name=name, start=start, stop=stop)
            rows.append(row)
    return rows

"""Host-side post-processing of the window table: thresholds on log10(KLD),
interval merging, GFF3 writers.  numpy/pandas on <= millions of floats - no kernels here.

Reference counterparts (frisk/__init__.py): FDBins L508-513, otsu L515-543, setKLDThresh L664-690,
thresholdKLD L647-662, anomaly2GFF L553-567, thresholdRIP L692-720, RIP2GFF L577-587, natural_sort L85-89.
The reference delegates interval merging to the external `bedtools` binary through pybedtools
(`merge -d D -c 4,4,4 -o max,min,mean`, `window -w 0 -u`), which is not available here: `merge_intervals`
restates the documented bedtools semantics (sorted input; features whose gap is <= D are merged, so
book-ended features merge at D = 0; numeric summaries printed with bedtools' default precision `-prec 5`).
Parity for that part is pinned by construction and by tests only, not by the reference's own output.
"""
import math
import os
import re

import numpy as np

FRISK_VERSION = "0+unknown"      # what the reference's versioneer reports outside a git checkout (_version.py)


# ------------------------------------------------------------------------------------ number -> text
def py2_str(x):
    """str() of a value as Python 2 prints it (the reference is Python 2): floats carry 12 significant
    digits ('%.12g', with '.0' appended to integral values); ints and strings are unchanged."""
    if isinstance(x, (int, np.integer)):
        return str(int(x))
    if isinstance(x, (float, np.floating)):
        x = float(x)
        if x != x:
            return "nan"
        if x in (float("inf"), float("-inf")):
            return "inf" if x > 0 else "-inf"
        s = "%.12g" % x
        if "." not in s and "e" not in s:
            s += ".0"
        return s
    return str(x)


def py3_str(x):
    if isinstance(x, (float, np.floating)):
        return repr(float(x))
    if isinstance(x, (int, np.integer)):
        return str(int(x))
    return str(x)


def natural_sort(items, key=str):
    """Scaffold-name order for the RIP GFF: digit runs compare as numbers, text case-insensitively (L85-89)."""
    def parts(item):
        return [int(t) if t.isdigit() else t.lower() for t in re.split("([0-9]+)", key(item))]
    return sorted(items, key=parts)


# ------------------------------------------------------------------------------------ thresholds
def FDBins(data):
    """Freedman-Diaconis-style bin count as the reference computes it: round(2 * IQR * n^(1/3)) (L508-513)."""
    q75, q25 = np.percentile(data, [75, 25])
    return int(round(2 * (q75 - q25) * math.pow(len(data), 1.0 / 3.0)))


def _py_max(a):
    """Python's max() over the rows of an array, without the 3 M Python steps: max() keeps its first element until a LARGER one
    comes (comparisons with NaN are false), so the result is NaN iff the first element is NaN, else the largest non-NaN one."""
    a = np.asarray(a, dtype=float)
    flat = a.reshape(a.shape[0], -1)[:, 0] if a.ndim > 1 else a
    if flat.size == 0:
        raise ValueError("max() arg is an empty sequence")
    m = flat[0] if np.isnan(flat[0]) else np.nanmax(flat)
    return a[0] * 0 + m if a.ndim > 1 else m


def otsu(data, optBins):
    """Otsu split of the log10(KLD) histogram, restated from L515-543: the values are scaled by
    -1/max|x| (so they become positive), binned into optBins bins normalised to the tallest bin, and the
    split i in 1..optBins-1 minimising v1*q1 + v2*q2 (variances and masses of the bin heights on either
    side) is mapped back to log10(KLD) through the histogram's bin edge."""
    raw = data
    x = np.atleast_1d(data)
    x = x[~np.isnan(x)]
    scale = _py_max(np.abs(x)) * -1.0
    x = x / scale
    hist, edges = np.histogram(x, bins=optBins)
    h = hist * 1.0
    h = h.ravel() / h.max()
    cum = h.cumsum()
    best, thresh = np.inf, -1
    for i in range(1, optBins):
        left, right = h[:i], h[i:]
        q1, q2 = cum[i - 1], cum[optBins - 1] - cum[i - 1]
        m1, m2 = q1 / len(left), q2 / len(right)
        v1 = np.sum(np.square(left - m1)) / len(left)
        v2 = np.sum(np.square(right - m2)) / len(right)
        score = (v1 * q1) + (v2 * q2)
        if score < best:
            best, thresh = score, i
    return edges[thresh] * _py_max(np.abs(raw)) * -1.0         # (max(abs(raw)) of the reference: NaN rows included, as there)


def setKLDThresh(args, logKLD):
    """(threshold on log10(KLD), number of histogram bins) - L664-690.  Like the reference this needs one of
    --forceThresholdKLD / --threshTypeKLD; without either the reference dies with UnboundLocalError."""
    optBins = max(FDBins(logKLD), 30)
    if args.forceThresholdKLD:
        return np.log10(float(args.forceThresholdKLD)), optBins
    if args.threshTypeKLD == "otsu":
        return otsu(logKLD, optBins), optBins
    if args.threshTypeKLD == "percentile":
        return np.percentile(logKLD, args.percentileKLD), optBins
    raise UnboundLocalError("local variable 'KLDthreshold' referenced before assignment "
                            "(give --forceThresholdKLD or --threshTypeKLD, as with the reference)")


# ------------------------------------------------------------------------------------ interval merging
def _prec5(v):
    return "%.5g" % v           # bedtools -prec default


def merge_intervals(records, dist=0, ops=("max",), cols=(3,)):
    """bedtools-merge semantics on records (chrom, start, end, v3, v4, ...) ALREADY sorted by chrom, start:
    consecutive features on one chrom whose start - running_end <= dist are merged.  Returns tuples
    (chrom, start, end, summary...) with one summary per (op, col) pair, formatted like bedtools (-prec 5)."""
    out = []
    cur = None
    for rec in records:
        chrom, start, end = rec[0], int(rec[1]), int(rec[2])
        if cur is not None and chrom == cur[0] and start - cur[2] <= dist:
            cur[2] = max(cur[2], end)
            cur[3].append(rec)
        else:
            if cur is not None:
                out.append(cur)
            cur = [chrom, start, end, [rec]]
    if cur is not None:
        out.append(cur)
    merged = []
    for chrom, start, end, members in out:
        summary = []
        for op, col in zip(ops, cols):
            vals = np.array([float(m[col]) for m in members], dtype=float)
            # (mean: a sequential sum over the members / their number, as bedtools accumulates it - and as merge_columns does)
            v = {"max": vals.max, "min": vals.min, "mean": lambda: np.add.reduceat(vals, [0])[0] / len(vals)}[op]()
            summary.append(_prec5(v))
        merged.append((chrom, start, end) + tuple(summary))
    return merged


def _name_ranks(names):
    """rank of every scaffold NAME in Python's string order (the sort key of L655 / L710), equal names sharing a rank."""
    order = {nm: r for r, nm in enumerate(sorted(set(names)))}
    return np.asarray([order[nm] for nm in names], dtype=np.int64)


def _sort_rows(rank, start, stop, idx):
    """Rows idx ordered by (name rank, start, stop).  A table comes out of the scan scaffold by scaffold with ascending windows,
    so a stable sort on the rank alone usually does it (checked; jumpback rows can step back by one base: then the full sort)."""
    o = np.argsort(rank, kind="stable")
    i2, r2 = idx[o], rank[o]
    st, sp = start[i2], stop[i2]
    same = r2[1:] == r2[:-1]
    if np.any(same & ((st[1:] < st[:-1]) | ((st[1:] == st[:-1]) & (sp[1:] < sp[:-1])))):
        o = np.lexsort((stop[idx], start[idx], rank))
        i2, r2 = idx[o], rank[o]
    return i2, r2


def merge_columns(names, name_rank, start, stop, values, dist=0, ops=("max",), cols=(0,)):
    """merge_intervals on COLUMNS: rows already sorted by (name, start, stop); name_rank[i] = rank of row i's scaffold name,
    values = list of float arrays the (op, col) pairs index.  Same features as merge_intervals (tested against it), without a
    Python step per row: the running end of bedtools' sweep is a running maximum of `stop` inside a scaffold, a feature ends
    where the next start lies more than `dist` behind it."""
    n = len(start)
    if n == 0:
        return []
    big = int(max(int(stop.max()), int(start.max()), 0)) + abs(int(dist)) + 2
    run_end = np.maximum.accumulate(stop.astype(np.int64) + name_rank * big) - name_rank * big
    brk = np.ones(n, dtype=bool)
    brk[1:] = (name_rank[1:] != name_rank[:-1]) | (start[1:] - run_end[:-1] > dist)
    a = np.nonzero(brk)[0]
    cnt = np.diff(np.concatenate((a, [n])))
    ends = np.maximum.reduceat(stop, a)
    summ = []
    for op, col in zip(ops, cols):
        v = np.ascontiguousarray(values[col], dtype=np.float64)
        if op == "max":
            r = np.maximum.reduceat(v, a)
        elif op == "min":
            r = np.minimum.reduceat(v, a)
        else:
            r = np.add.reduceat(v, a) / cnt
        summ.append(["%.5g" % x for x in r.tolist()])
    first = names[a].tolist() if isinstance(names, np.ndarray) else [names[i] for i in a.tolist()]
    return [(nm, s0, e0) + tuple(c[k] for c in summ)
            for k, (nm, s0, e0) in enumerate(zip(first, start[a].tolist(), ends.tolist()))]


def thresholdKLD(table, threshold, args, merge=True):
    """Windows whose log10(KLD) is >= threshold (<= with --findSelf), sorted by (name, start, stop), merged with
    `-d mergeDist -c 4,4,4 -o max,min,mean` (L647-662).  table: list of rows (name, start, stop, KLD, ...).
    Returns (features, selected_rows)."""
    if hasattr(table, "kld"):           # a ScoreTable: select on the columns, make tuples of the chosen rows only
        kld = np.where(table.kld_is_int0 != 0, 0.0, table.kld)
        with np.errstate(divide="ignore", invalid="ignore"):
            logs = np.log10(kld)
        pick = (logs <= threshold) if getattr(args, "findSelf", False) else (logs >= threshold)
        idx = np.nonzero(pick & ~np.isnan(kld))[0]
        if merge and idx.size > 20000 and getattr(args, "mergeDist", 0) >= 0:      # many windows selected (3 M rows, an Otsu cut): sort and merge on the columns
            rank = _name_ranks(table.names)[table.seq_index[idx]]
            idx, rank = _sort_rows(rank, table.start, table.stop, idx)
            nm = np.asarray(table.names, dtype=object)[table.seq_index[idx]]
            feats = merge_columns(nm, rank, table.start[idx], table.stop[idx], [kld[idx]], dist=getattr(args, "mergeDist", 0),
                                  ops=("max", "min", "mean"), cols=(0, 0, 0))
            return feats, _LazyRows(table, idx)
        chosen = table.rows(idx)
        chosen.sort(key=lambda r: (r[0], r[1], r[2]))
    else:
        rows = [r for r in table if not (isinstance(r[3], float) and r[3] != r[3])]
        rows.sort(key=lambda r: (r[0], r[1], r[2]))
        with np.errstate(divide="ignore", invalid="ignore"):
            logs = np.log10(np.array([float(r[3]) for r in rows], dtype=float)) if rows else np.zeros(0)
        pick = (logs <= threshold) if getattr(args, "findSelf", False) else (logs >= threshold)
        chosen = [r for r, p in zip(rows, pick) if p]
    recs = [(r[0], int(r[1]), int(r[2]), float(r[3])) for r in chosen]
    if merge:
        feats = merge_intervals(recs, dist=getattr(args, "mergeDist", 0), ops=("max", "min", "mean"), cols=(3, 3, 3))
    else:
        fmt = py3_str if os.environ.get("FRISK_FLOAT_REPR", "py2") == "py3" else py2_str    # the table's float text
        feats = [(c, s, e, fmt(v)) for c, s, e, v in recs]
    return feats, chosen


class _LazyRows:
    """The selected rows of a big table, in sorted order, as tuples on demand (the callers of thresholdKLD use them rarely)."""

    def __init__(self, table, idx):
        self.table, self.idx = table, idx

    def __len__(self):
        return int(self.idx.size)

    def __iter__(self):
        return iter(self.table.rows(self.idx))

    def __getitem__(self, k):
        return self.table.rows(self.idx[k] if isinstance(k, slice) else [self.idx[k]])[0 if not isinstance(k, slice) else slice(None)]


def anomaly2GFF(features, args, category="Kmer-anomaly", version=FRISK_VERSION):
    """GFF3 lines for merged anomalies (L553-567): source frisk_<version>, ID zero-padded to the width of the
    feature count, KLD=<max> (or max/min/mean with --dimReduce features)."""
    width = len(str(len(features)))
    for n, f in enumerate(features, 1):
        if getattr(args, "dimReduce", "windows") == "windows":
            attrs = ["ID=Anomaly_" + str(n).zfill(width), "KLD=" + str(f[3])]
        else:
            attrs = ["ID=Anomaly_" + str(n).zfill(width), "maxKLD=" + str(f[3]), "minKLD=" + str(f[4]), "meanKLD=" + str(f[5])]
        if n == 1:
            yield "##gff-version 3\n"
        yield "\t".join([str(f[0]), "frisk_" + version, category, str(f[1]), str(f[2]), ".", "+", ".", ";".join(attrs)]) + "\n"


def thresholdRIP(table, args):
    """RIP features (L692-720): windows with PI >= minPI, SI <= maxSI, CRI >= minCRI are merged
    (`-d 0 -c 4,5,6,7,7 -o max,min,max,min,max`) and kept if they overlap at least one window with
    CRI >= peakCRI (`window -w 0 -u`).  table rows: (name, start, stop, KLD, GC, PI, SI, CRI)."""
    if hasattr(table, "kld"):           # a ScoreTable: everything on the columns (3 M rows: no tuple per window)
        with np.errstate(invalid="ignore"):
            ok_m = ~(np.isnan(table.kld) | np.isnan(table.pi) | np.isnan(table.si) | np.isnan(table.cri))
            basic_i = np.nonzero(ok_m & (table.pi >= args.minPI) & (table.si <= args.maxSI) & (table.cri >= args.minCRI))[0]
            peak_i = np.nonzero(ok_m & (table.cri >= args.peakCRI))[0]
        if basic_i.size == 0 or peak_i.size == 0:
            return None
        ranks = _name_ranks(table.names)
        names_o = np.asarray(table.names, dtype=object)

        def ordered(idx):
            return _sort_rows(ranks[table.seq_index[idx]], table.start, table.stop, idx)
        basic_i, brank = ordered(basic_i)
        peak_i, prank = ordered(peak_i)
        kld = np.where(table.kld_is_int0[basic_i] != 0, 0.0, table.kld[basic_i])
        merged = merge_columns(names_o[table.seq_index[basic_i]], brank, table.start[basic_i], table.stop[basic_i],
                               [kld, table.pi[basic_i], table.si[basic_i], table.cri[basic_i]], dist=0,
                               ops=("max", "min", "max", "min", "max"), cols=(0, 1, 2, 3, 3))
        # bedtools window -w 0 -u: keep a feature that overlaps a peak window (half-open BED arithmetic).  Peaks sorted by
        # (scaffold, start) with the running maximum of their ends inside the scaffold: a feature overlaps one iff, among the
        # scaffold's peaks that start before its end, some end lies behind its start
        big = int(max(int(table.stop.max()), int(table.start.max()), 0)) + 2
        pkey = prank * big + table.start[peak_i]
        pend = np.maximum.accumulate(prank * big + table.stop[peak_i])
        rank_of = dict(zip(table.names, ranks.tolist()))
        frank = np.asarray([rank_of[f[0]] for f in merged], dtype=np.int64)
        fstart = np.asarray([f[1] for f in merged], dtype=np.int64)
        fend = np.asarray([f[2] for f in merged], dtype=np.int64)
        k = np.searchsorted(pkey, frank * big + fend, side="left")
        hit = (k > 0) & (pend[np.maximum(k, 1) - 1] > frank * big + fstart)
        keep = [f for f, h in zip(merged, hit.tolist()) if h]
        return keep or None
    else:
        ok = [r for r in table if not any(isinstance(v, float) and v != v for v in (r[3], r[5], r[6], r[7]))]
        ok.sort(key=lambda r: (r[0], r[1], r[2]))
        basic = [r for r in ok if r[5] >= args.minPI and r[6] <= args.maxSI and r[7] >= args.minCRI]
        peaks = [r for r in ok if r[7] >= args.peakCRI]
    if not basic or not peaks:
        return None
    recs = [(r[0], int(r[1]), int(r[2]), r[3], r[5], r[6], r[7]) for r in basic]
    merged = merge_intervals(recs, dist=0, ops=("max", "min", "max", "min", "max"), cols=(3, 4, 5, 6, 6))
    # bedtools window -w 0: A and B overlap (half-open BED arithmetic on the given coordinates).  Peaks per scaffold sorted by
    # start, with the running maximum of their ends: a feature overlaps one iff, among the peaks that start before its end,
    # some end lies behind its start
    by = {}
    for p in peaks:
        by.setdefault(p[0], []).append((int(p[1]), int(p[2])))
    idx = {}
    for nm, lst in by.items():
        lst.sort()
        st = np.asarray([x[0] for x in lst], dtype=np.int64)
        idx[nm] = (st, np.maximum.accumulate(np.asarray([x[1] for x in lst], dtype=np.int64)))
    keep = []
    for f in merged:
        got = idx.get(f[0])
        if got is None:
            continue
        k = int(np.searchsorted(got[0], f[2], side="left"))        # peaks with start < feature end
        if k > 0 and got[1][k - 1] > f[1]:
            keep.append(f)
    return keep or None


def RIP2GFF(features, version=FRISK_VERSION):
    """GFF3 lines for RIP features in natural scaffold order (L577-587)."""
    feats = natural_sort(features, key=lambda f: f[0])
    width = len(str(len(feats)))
    for n, f in enumerate(feats, 1):
        attrs = ["ID=Anomaly_" + str(n).zfill(width), "maxKLD=" + str(f[3]), "minPI=" + str(f[4]), "maxSI=" + str(f[5]),
                 "minCRI=" + str(f[6]), "maxCRI=" + str(f[7])]
        if n == 1:
            yield "##gff-version 3\n"
        yield "\t".join([str(f[0]), "frisk_" + version, "RIP", str(f[1]), str(f[2]), ".", "+", ".", ";".join(attrs)]) + "\n"

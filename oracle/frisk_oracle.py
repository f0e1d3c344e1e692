"""CPU oracle for the frisk hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch restatement, in plain Python, of the algorithm of the reference's
window scan (k-mer counting -> IVOM interpolation -> Kullback-Leibler score, plus
the GC and RIP columns).  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module, and only as the checker;
the product path (`frisk_amd`) never does.

Parity pin: every function here is checked against golden vectors produced from
the reference's own functions (tools/make_golden.py -> tests/golden/*.json|npz;
tests/test_oracle_golden.py).  It is deliberately "reference-shaped": one window
at a time, string-keyed dictionaries, Python loops - so that timing it is a fair
stand-in for "frisk's own Python path on the host cores" (BASELINE.md).

Citations are to /root/reference/frisk/__init__.py (written L<n>).
"""
import gzip
import math
from collections import Counter

BASES = ("A", "T", "G", "C")          # L70: digit order of the canonical k-mer index
_COMP = {"A": "T", "T": "A", "G": "C", "C": "G"}


# --------------------------------------------------------------------------- I/O
def iter_fasta(path):
    """(name, sequence) per record; name = first token after '>'; blank lines skipped;
    case preserved; '.gz' read through gzip (L139-164)."""
    opener = gzip.open if path.endswith(".gz") else open
    name, chunks = None, []
    with opener(path, "rt") as fh:
        for raw in fh:
            line = raw.strip()
            if not line:
                continue
            if line[0] == ">":
                if name:
                    yield name, "".join(chunks)
                name, chunks = line.strip(">").split()[0], []
            else:
                chunks.append(line)
    if name:
        yield name, "".join(chunks)


# ------------------------------------------------------------------ base tallies
def count_acgt(seq):
    """(#uppercase A/T/G/C, #everything else) - case-sensitive (L106-118)."""
    tally = Counter(seq)
    good = sum(tally[b] for b in BASES)
    return good, len(seq) - good


def gc_fraction(seq):
    """(G+C)/(A+T+G+C) over uppercase bases only; ZeroDivisionError if none (L120-137)."""
    tally = Counter(seq)
    gc = tally["G"] + tally["C"]
    at = tally["A"] + tally["T"]
    return float(gc) / (gc + at)


# ------------------------------------------------------------- window enumeration
def small_scaffold_limit(w, i):
    return w + ((w * 0.75) - i)       # float, as at L211 / L222


def iter_windows(records, w, i, scaffolds_all=False):
    """Yield (window_seq, name, start, stop) exactly as the reference's crawler does
    (L194-251): floor(size/i) candidates per scaffold, the sticky 'jumpback' window
    with its 0-based start, the >= 30 % non-ACGT filter, small-scaffold skip/rescue."""
    for name, seq in records:
        size = len(seq)
        if size <= small_scaffold_limit(w, i):
            if scaffolds_all:
                if count_acgt(seq)[1] >= 0.3 * size:
                    continue
                yield seq, name, 1, size
            continue
        jumped = False
        for j in range(0, size - i + 1, i):
            if j + w > size:
                win = seq[size - w:size]
                jumped = True
            else:
                win = seq[j:j + w]
            if count_acgt(win)[1] >= 0.3 * len(win):
                continue
            if jumped:
                yield win, name, size - w, size
            else:
                yield win, name, j + 1, j + w


# ------------------------------------------------------------------ k-mer counting
def kmer_of(code, x):
    return "".join(BASES[(code >> (2 * (x - 1 - p))) & 3] for p in range(x))


def code_of(kmer):
    """Canonical index of a k-mer: digits A=0,T=1,G=2,C=3, first base most significant (L70, L253-274)."""
    c = 0
    for ch in kmer:
        c = c * 4 + BASES.index(ch)
    return c


def blank_maps(kmin, kmax):
    """One dict per order, every k-mer -> 0, keys in canonical index order (L253-274)."""
    return [{kmer_of(c, x): 0 for c in range(4 ** x)} for x in range(kmin, kmax + 1)]


def revcomp(word):
    return "".join(_COMP[b] for b in reversed(word))   # L276-278


def count_kmers(records, kmin, kmax, genome_mode=False, mask_host=False):
    """All orders kmin..kmax, one-base step, words with a non-ACGT letter skipped (L280-367).

    window mode : words are upper-cased first, so soft-masked bases ARE counted.
    genome mode : adds the reverse complement of every counted word; with mask_host the
                  words are NOT upper-cased, so soft-masked words are skipped.
    Returns (maps, meta) with meta = dict(totalLen, exMax, nnTotal)."""
    maps = blank_maps(kmin, kmax)
    meta = {"totalLen": 0, "exMax": 0, "nnTotal": 0}
    for _name, seq in records:
        n = len(seq)
        meta["totalLen"] += n
        meta["nnTotal"] += count_acgt(seq)[1]
        for x in range(kmin, kmax + 1):
            table = maps[x - kmin]
            for j in range(n - x + 1):
                word = seq[j:j + x]
                if not (genome_mode and mask_host):
                    word = word.upper()
                if word not in table:
                    if x == kmax:
                        meta["exMax"] += 1
                    continue
                table[word] += 1
                if genome_mode:
                    table[revcomp(word)] += 1
    return maps, meta


# ----------------------------------------------------------------------- IVOM / KLD
def ivom(window_maps, window_meta, source_maps, source_meta, kmin, kmax):
    """Interpolated variable-order probability of every max-mer PRESENT IN THE WINDOW,
    estimated from `source` counts (the window itself, or the genome), normalised over
    that present set (L369-457).

    For max-mer k and x = kmin..kmax, with c_x the source count of k's length-x prefix
    and S = source totalLen - nnTotal:
        w_x = c_x * 4**x                      (exact integer)
        p_x = c_x / ((S - (x-1)) * 2)
        a_x = w_x / (w_kmin + ... + w_x)      (ZeroDivisionError if the sum is 0)
        I_kmin = a*p ;  I_x = a_x*p_x + (1-a_x)*I_{x-1}
    """
    space = source_meta["totalLen"] - source_meta["nnTotal"]
    top = window_maps[kmax - kmin]
    raw = {}
    total = 0
    for kmer, present in top.items():
        if present == 0:
            continue
        run = 0
        interp = None
        for x in range(kmin, kmax + 1):
            c = source_maps[x - kmin][kmer[:x]]
            weight = c * 4 ** x
            prob = float(c) / ((space - (x - 1)) * 2)
            run += weight
            a = float(weight) / run
            if x == kmin:
                interp = a * prob
            else:
                interp = a * prob + ((1 - a) * interp)
        raw[kmer] = interp
        total += interp
    for kmer in raw:
        raw[kmer] = float(raw[kmer]) / total
    return raw


def kld(genome_ivom, window_ivom):
    """sum_k Pw*log(Pw/Pg, 2) over present max-mers, terms with Pg == 0 skipped (L459-472).
    math.log(x, 2) is ln(x)/ln(2), not log2(x)."""
    acc = 0
    for kmer, pw in window_ivom.items():
        pg = float(genome_ivom[kmer])
        if pg != 0:
            acc += pw * math.log(pw / pg, 2)
    return acc


def rip_indices(window_maps, kmin, kmax):
    """(PI, SI, CRI) from the window's dinucleotide counts (L474-495).  CRI is NaN unless
    both PI and SI are truthy - so PI == 0.0 or SI == 0.0 gives NaN."""
    di = window_maps[list(range(kmin, kmax + 1)).index(2)]
    nan = float("nan")
    pi = di["TA"] / float(di["AT"]) if di["AT"] > 0 else nan
    den = di["AC"] + di["GT"]
    si = (di["CA"] + di["TG"]) / float(den) if den > 0 else nan
    cri = pi - si if (pi and si) else nan
    return pi, si, cri


# -------------------------------------------------------------------------- drivers
def flatten(maps, kmin, kmax):
    out = []
    for x in range(kmin, kmax + 1):
        table = maps[x - kmin]
        out.extend(table[kmer_of(c, x)] for c in range(4 ** x))
    return out


def genome_profile(host_path, kmin, kmax, mask_host=False):
    """Phase A (L1442): symmetric counts over every record of the host FASTA."""
    return count_kmers(iter_fasta(host_path), kmin, kmax, genome_mode=True, mask_host=mask_host)


def score_window(seq, genome_maps, genome_meta, kmin, kmax, rip=False):
    """One iteration of the scan loop L1478-1494.  Returns a dict with counts/meta/KLD/GC/RIP;
    'error' is set where the reference would raise ZeroDivisionError."""
    wmaps, wmeta = count_kmers([("w", seq)], kmin, kmax)
    row = {"maps": wmaps, "meta": [wmeta["totalLen"], wmeta["exMax"], wmeta["nnTotal"]]}
    try:
        g = ivom(wmaps, wmeta, genome_maps, genome_meta, kmin, kmax)
        w = ivom(wmaps, wmeta, wmaps, wmeta, kmin, kmax)
        row["KLD"] = kld(g, w)
        row["ivom"] = (g, w)            # IvomBuild's two return values (L1481-1482), for the golden IVOM vectors
    except ZeroDivisionError:
        row["error"] = "ZeroDivisionError"
    try:
        row["GC"] = gc_fraction(seq)
    except ZeroDivisionError:
        row["GC_error"] = "ZeroDivisionError"
    if rip and kmin <= 2:
        row["RIP"] = list(rip_indices(wmaps, kmin, kmax))
    return row


def scan(host_path, query_path, kmin, kmax, w, i, mask_host=False, scaffolds_all=False, rip=False,
         max_rows=None, profile=None):
    """Phase A + phase B.  Returns (genome_maps, genome_meta, rows)."""
    if profile is None:
        profile = genome_profile(host_path, kmin, kmax, mask_host)
    gmaps, gmeta = profile
    rows = []
    for seq, name, start, stop in iter_windows(iter_fasta(query_path or host_path), w, i, scaffolds_all):
        row = score_window(seq, gmaps, gmeta, kmin, kmax, rip)
        row.update(name=name, start=start, stop=stop)
        rows.append(row)
        if max_rows is not None and len(rows) >= max_rows:
            break
    return gmaps, gmeta, rows

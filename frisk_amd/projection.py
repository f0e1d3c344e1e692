"""Symmetric k-mer proportions of anomalous windows - the input of the reference's projection/clustering step
(SURVEY.md section 8, row f4).  The projection itself (sklearn PCA/t-SNE/..., DBSCAN/k-means) is out of scope.

Reference (frisk/__init__.py): computeKmers(sym=True, pcaMode=True) L280-367 counts every valid word AND its
reverse complement for orders pcaMin..pcaMax; scrubMirrors L797-811 keeps one key of each reverse-complement pair
(the first in canonical key order, i.e. codes with c <= revcomp(c)); flattenKmerMap(prop=True) L813-831 turns each
order's kept counts into proportions of their sum and concatenates the orders (loop L1573-1591).
The counting runs on the GPU (per-window forward counts from frisk_scan's count dump); folding and proportions are
host numpy.
"""
import numpy as np

from .engine import Engine, table_offset
from .hotpath import kmerString


def revcomp_index(x):
    c = np.arange(4 ** x, dtype=np.int64)
    r = np.zeros_like(c)
    t = c.copy()
    for _ in range(x):
        r = (r << 2) | ((t & 3) ^ 1)        # A<->T, G<->C is XOR 1 on the digit (A,T,G,C = 0,1,2,3)
        t >>= 2
    return r


def mirror_keep(x):
    """codes kept by scrubMirrors at order x, in canonical order."""
    c = np.arange(4 ** x, dtype=np.int64)
    return c[c <= revcomp_index(x)]


def feature_keys(kmin, kmax):
    return [kmerString(int(c), x) for x in range(kmin, kmax + 1) for c in mirror_keep(x)]


def proportions_from_forward(fwd, kmin, kmax):
    """fwd: forward counts in profile layout (orders kmin..kmax) -> the flattenKmerMap(prop=True) vector."""
    out = []
    for x in range(kmin, kmax + 1):
        o = table_offset(kmin, x)
        f = np.asarray(fwd[o:o + 4 ** x], dtype=np.int64)
        sym = f + f[revcomp_index(x)]                      # sym=True: word and reverse complement (L350-351)
        kept = sym[mirror_keep(x)]
        out.append(kept.astype(np.float64) / float(int(kept.sum())))      # float(v) / sum(d.values()) (L822)
    return np.concatenate(out)


def symmetricCounts(labelled_seqs, pcaMin, pcaMax, device=0):
    """(anomLabels, anomCounts) as the reference builds them at L1573-1591: one row of proportions per sequence.
    labelled_seqs: list of (label, sequence)."""
    seqs = [s for _, s in labelled_seqs]
    if not seqs:
        return np.zeros((0, 1), dtype=object), np.zeros((0, len(feature_keys(pcaMin, pcaMax))))
    longest = max(len(s) for s in seqs)
    with Engine(pcaMin, pcaMax, device) as e:
        e.load(seqs)
        e.profile_reset(); e.profile_add(); e.profile_finalize()
        # every sequence as ONE window: with inc = 1 the small-scaffold limit is 1.75 w - 1 >= every length
        res = e.scan(max(longest, 2), 1, scaffolds_all=True, debug=True)
    rows = []
    for i in range(len(seqs)):
        hit = np.nonzero(res.seq_index == i)[0]
        if len(hit) != 1 or not res.kept[hit[0]]:
            raise ValueError("sequence %r has >= 30 %% unresolved bases: no k-mer vector" % (labelled_seqs[i][0],))
        rows.append(proportions_from_forward(res.counts[hit[0]], pcaMin, pcaMax))
    labels = np.array([[lab] for lab, _ in labelled_seqs])
    return labels, np.vstack(rows)

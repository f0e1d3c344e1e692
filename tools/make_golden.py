#!/usr/bin/env python3
"""Generate golden vectors for the hot path from the reference's OWN functions.

Runs only in the build container (needs /root/reference).  Nothing of the
reference is copied into this repository: the reference module is read at run
time, converted py2->py3 with the stdlib's lib2to3 in a scratch directory
outside the repo, and only the hot-path FunctionDefs are exec'd (the module as
a whole cannot be imported: py2 syntax at frisk/__init__.py:95, and top-level
imports of hmmlearn / pybedtools / seaborn which are absent - SURVEY.md §8c).

Outputs (committed, data only):
  tests/golden/<case>.json   rows, scalars (floats round-trip through repr)
  tests/golden/<case>.npz    integer count tables (genome profile, per-window)

Driver = the reference's own call pattern: phase A is the call at
frisk/__init__.py:1442, phase B the loop at frisk/__init__.py:1478-1494.
"""
import argparse
import ast
import copy
import gzip
import itertools
import json
import logging
import math
import os
import pickle
import shutil
import subprocess
import sys
import tempfile
from collections import Counter

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, ".."))
GOLD = os.path.join(REPO, "tests", "golden")
INP = os.path.join(GOLD, "inputs")
REF_SRC = "/root/reference/frisk/__init__.py"

HOT_FUNCS = ["countN", "calcGC", "iterFasta", "crawlGenome", "prepareMaps", "rangeMaps",
             "revComplement", "computeKmers", "IvomBuild", "KLD", "calcRIP", "makePicklePath",
             "FDBins", "otsu", "setKLDThresh", "natural_sort",      # row f1 (thresholds), host-side numpy
             "mainArgs",                                            # the argparse surface (row f2)
             "scrubMirrors", "flattenKmerMap",                      # symmetric counts for the projection (row f4)
             # rows f1 / f3: feature selection and the GFF3 writers, the HMM's run extraction (pure Python around
             # the third-party calls, which run against the stand-ins of class BedToolStub / the model handed in)
             "thresholdKLD", "thresholdRIP", "anomaly2GFF", "RIP2GFF", "hmmBED2GFF", "findBaseRanges",
             "range2interval", "hmm2BED"]


class BedToolStub:
    """Stand-in for pybedtools.BedTool (pybedtools and the bedtools binary are absent here - SURVEY.md section 8c): holds
    records the way pybedtools Intervals index (fields are strings) and answers the two bedtools calls the reference makes,
    `merge -d D -c cols -o ops` and `window -w W -u`, from bedtools' documented semantics.  Written for this generator
    only, independently of frisk_amd.postprocess, so that the product's merge is at least checked differentially.  Every
    record handed to the constructor is logged in `BedToolStub.made` (the reference's pre-merge selections)."""
    made = []

    def __init__(self, records):
        self.recs = [[str(f) for f in r] for r in records]
        BedToolStub.made.append([list(r) for r in self.recs])

    def __iter__(self):
        return iter(self.recs)

    def __len__(self):
        return len(self.recs)

    def merge(self, d=0, c="", o=""):
        cols = [int(x) - 1 for x in str(c).split(",")]
        ops = str(o).split(",")
        out, i, n = [], 0, len(self.recs)
        while i < n:
            chrom, lo, hi = self.recs[i][0], int(self.recs[i][1]), int(self.recs[i][2])
            j = i + 1
            while j < n and self.recs[j][0] == chrom and int(self.recs[j][1]) <= hi + d:
                hi = max(hi, int(self.recs[j][2]))
                j += 1
            fields = [chrom, str(lo), str(hi)]
            for col, op in zip(cols, ops):
                v = [float(r[col]) for r in self.recs[i:j]]
                x = max(v) if op == "max" else min(v) if op == "min" else math.fsum(v) / len(v)
                fields.append("%.5g" % x)           # bedtools prints numeric summaries with -prec 5
            out.append(fields)
            i = j
        res = BedToolStub.__new__(BedToolStub)
        res.recs = out
        return res

    def window(self, b=None, w=0, u=True):
        keep = [a for a in self.recs
                if any(p[0] == a[0] and min(int(a[2]) + w, int(p[2])) - max(int(a[1]) - w, int(p[1])) > 0 for p in b.recs)]
        res = BedToolStub.__new__(BedToolStub)
        res.recs = keep
        return res


class PyBedToolsStub:
    BedTool = BedToolStub


def load_reference_functions():
    scratch = tempfile.mkdtemp(prefix="frisk_oracle_")
    dst = os.path.join(scratch, "frisk_py2.py")
    shutil.copy(REF_SRC, dst)
    subprocess.run([sys.executable, "-m", "lib2to3", "-w", "-n", dst], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    tree = ast.parse(open(dst).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in HOT_FUNCS]
    assert sorted(n.name for n in keep) == sorted(HOT_FUNCS), [n.name for n in keep]
    mod = ast.Module(body=keep, type_ignores=[])

    class _NP:  # calcRIP uses np.NaN (removed in numpy 2)
        NaN = float("nan")

    class _NPX:  # numpy, plus the alias numpy 2 removed
        def __getattr__(self, k):
            return float("nan") if k == "NaN" else getattr(np, k)

    import re
    from operator import itemgetter
    import pandas as pd
    ns = dict(Counter=Counter, math=math, copy=copy, gzip=gzip, logging=logging, pickle=pickle, re=re,
              sys=sys, os=os, itertools=itertools, np=_NPX(), LETTERS=("A", "T", "G", "C"), argparse=argparse,
              FRISK_VERSION="0+unknown", itemgetter=itemgetter, pd=pd, pybedtools=PyBedToolsStub())
    exec(compile(mod, "<reference hot path via lib2to3>", "exec"), ns)
    shutil.rmtree(scratch)
    return ns


class Args:
    def __init__(self, hostSeq, querySeq=None, m=1, k=8, w=5000, i=2500, maskHost=False,
                 scaffoldsAll=False, RIP=False, tempDir="."):
        self.hostSeq = hostSeq
        self.querySeq = querySeq
        self.minWordSize = m
        self.maxWordSize = k
        self.windowlen = w
        self.increment = i
        self.maskHost = maskHost
        self.scaffoldsAll = scaffoldsAll
        self.RIP = RIP
        self.tempDir = tempDir


def code_to_kmer(code, x):
    return "".join("ATGC"[(code >> (2 * (x - 1 - p))) & 3] for p in range(x))


def flatten(maps, m, k):
    """list-of-dicts -> int64[sum 4^x], canonical order A,T,G,C = 0..3, first base most significant."""
    out = []
    for x in range(m, k + 1):
        d = maps[x - m]
        assert len(d) == 4 ** x
        out.extend(d[code_to_kmer(c, x)] for c in range(4 ** x))
    return np.asarray(out, dtype=np.int64)


def meta(maps, m, k):
    r = k - m
    return [maps[r + 1]["totalLen"], maps[r + 2]["exMax"], maps[r + 3]["nnTotal"]]


def dense(d, k):
    v = np.zeros(4 ** k, dtype=np.float64)
    for kmer, val in d.items():
        c = 0
        for ch in kmer:
            c = (c << 2) | "ATGC".index(ch)
        v[c] = val
    return v


CASES = [
    # name, host, query, kwargs, store_ivom
    ("kat", "kat.fa", None, dict(m=1, k=3, w=20, i=8, RIP=True), True),
    ("uniform_k4", "uniform3k.fa", None, dict(m=1, k=4, w=500, i=100, RIP=True), True),
    ("markov_k6", "markov_islands.fa", None, dict(m=1, k=6, w=400, i=150, RIP=True), True),      # (k = 6, m <= 3: the narrow-counter kernels)
    ("markov_k5_i100", "markov_islands.fa", None, dict(m=1, k=5, w=400, i=100), False),
    ("markov_m2k4", "markov_islands.fa", None, dict(m=2, k=4, w=400, i=150, RIP=True), True),
    ("markov_m3k3", "markov_islands.fa", None, dict(m=3, k=3, w=400, i=150), True),
    ("markov_m3k5", "markov_islands.fa", None, dict(m=3, k=5, w=400, i=150), False),
    ("smalls_skip", "smalls.fa", None, dict(m=1, k=4, w=400, i=150, RIP=True), False),
    ("smalls_all", "smalls.fa", None, dict(m=1, k=4, w=400, i=150, RIP=True, scaffoldsAll=True), False),
    ("smalls_all_k8", "smalls.fa", None, dict(m=1, k=8, w=400, i=150, scaffoldsAll=True), False),
    ("nheavy_k4", "nheavy.fa", None, dict(m=1, k=4, w=400, i=100, RIP=True), False),
    ("maskhost", "host.fa", None, dict(m=1, k=5, w=500, i=250, maskHost=True), False),
    ("hq_k6", "host.fa", "query.fa", dict(m=1, k=6, w=500, i=100, RIP=True), False),
    ("hq_m5k6_zero", "host.fa", "query.fa", dict(m=5, k=6, w=500, i=100), False),
    ("k7", "k8.fa", None, dict(m=1, k=7, w=3000, i=700, RIP=True), False),
    ("k8", "k8.fa", None, dict(m=1, k=8, w=5000, i=1000, RIP=True), False),
    ("k8_w2000", "k8.fa", None, dict(m=1, k=8, w=2000, i=500), False),
    ("k8_m2", "k8.fa", None, dict(m=2, k=8, w=5000, i=2500, RIP=True), False),
    ("overshoot", "overshoot.fa", None, dict(m=1, k=3, w=100, i=90, RIP=True), False),
    ("hq_k8", "host.fa", "query.fa", dict(m=1, k=8, w=1000, i=250, RIP=True), False),
    ("maskhost_k8", "host.fa", None, dict(m=1, k=8, w=2000, i=1000, maskHost=True, scaffoldsAll=True), False),
    ("hq_m7k8_zero", "host.fa", "query.fa", dict(m=7, k=8, w=1000, i=500), False),
]


def run_case(ns, name, host, query, kw, store_ivom):
    tmp = tempfile.mkdtemp(prefix="frisk_gold_")
    a = Args(os.path.join(INP, host), os.path.join(INP, query) if query else None, tempDir=tmp, **kw)
    m, k = a.minWordSize, a.maxWordSize
    querySeq = a.querySeq if a.querySeq else a.hostSeq
    blankMap = ns["rangeMaps"](m, k)
    # phase A: frisk/__init__.py:1442
    genomepickle = ns["makePicklePath"](a, space="genome")
    genomeKmers = ns["computeKmers"](a, genomepickle=genomepickle, window=None, genomeMode=True,
                                     kmerMap=blankMap, getMeta=True)
    arrays = {"genome_counts": flatten(genomeKmers, m, k)}
    out = {
        "case": name, "host": host, "query": query,
        "args": dict(minWordSize=m, maxWordSize=k, windowlen=a.windowlen, increment=a.increment,
                     maskHost=a.maskHost, scaffoldsAll=a.scaffoldsAll, RIP=a.RIP),
        "genome_meta": meta(genomeKmers, m, k),
        "genome_pickle_basename": os.path.basename(genomepickle),
        "window_pickle_basename": os.path.basename(ns["makePicklePath"](a, space="window")),
        "rows": [],
    }
    win_counts, w_ivom, g_ivom = [], [], []
    rip_on = a.RIP and m <= 2
    # phase B: frisk/__init__.py:1478-1494
    for seq, sname, start, stop in ns["crawlGenome"](a, querySeq):
        row = {"name": sname, "start": start, "stop": stop}
        windowKmers = ns["computeKmers"](a, genomepickle=None, window=[(sname, seq)], genomeMode=False,
                                         kmerMap=blankMap, getMeta=True)
        win_counts.append(flatten(windowKmers, m, k))
        row["meta"] = meta(windowKmers, m, k)
        try:
            GenomeIVOM = ns["IvomBuild"](windowKmers, a, genomeKmers, True)
            windowIVOM = ns["IvomBuild"](windowKmers, a, genomeKmers, False)
            row["KLD"] = ns["KLD"](GenomeIVOM, windowIVOM, a)
            if store_ivom:
                w_ivom.append(dense(windowIVOM, k))
                g_ivom.append(dense(GenomeIVOM, k))
        except ZeroDivisionError:
            row["error"] = "ZeroDivisionError"
            if store_ivom:
                w_ivom.append(np.zeros(4 ** k))
                g_ivom.append(np.zeros(4 ** k))
        try:
            row["GC"] = ns["calcGC"](seq)
        except ZeroDivisionError:
            row["GC_error"] = "ZeroDivisionError"
        if rip_on:
            PI, SI, CRI = ns["calcRIP"](windowKmers, a)
            row["RIP"] = [PI, SI, CRI]
            # the table text the reference writes (py3 str(); py2 differs only in float digits)
        out["rows"].append(row)
    arrays["window_counts"] = (np.stack(win_counts).astype(np.int32) if win_counts
                               else np.zeros((0, arrays["genome_counts"].size), np.int32))
    if store_ivom:
        arrays["window_ivom"] = np.stack(w_ivom) if w_ivom else np.zeros((0, 4 ** k))
        arrays["genome_ivom"] = np.stack(g_ivom) if g_ivom else np.zeros((0, 4 ** k))
    shutil.rmtree(tmp)
    with open(os.path.join(GOLD, name + ".json"), "w") as fh:
        json.dump(out, fh, indent=0)
        fh.write("\n")
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **arrays)
    nerr = sum(1 for r in out["rows"] if "error" in r)
    print("%-16s rows=%-4d errors=%-3d genome_meta=%s" % (name, len(out["rows"]), nerr, out["genome_meta"]))


def run_thresholds(ns):
    """Row f1: FDBins / otsu / setKLDThresh (L508-543, L664-690) on the KLD columns of three scan cases."""
    class A:
        pass
    out = {}
    for case in ("uniform_k4", "markov_k6", "k8_w2000", "hq_k6"):
        doc = json.load(open(os.path.join(GOLD, case + ".json")))
        kld = np.array([[r["KLD"]] for r in doc["rows"] if "KLD" in r], dtype=float)     # as_matrix(columns=[..]) shape (n,1)
        logk = np.log10(kld)
        rec = {"KLD": kld.ravel().tolist(), "FDBins": ns["FDBins"](logk)}
        for mode, kw in (("otsu", dict(threshTypeKLD="otsu", forceThresholdKLD=None, percentileKLD=99.0)),
                         ("pct99", dict(threshTypeKLD="percentile", forceThresholdKLD=None, percentileKLD=99.0)),
                         ("pct80", dict(threshTypeKLD="percentile", forceThresholdKLD=None, percentileKLD=80.0)),
                         ("force", dict(threshTypeKLD=None, forceThresholdKLD=0.05, percentileKLD=99.0))):
            a = A()
            a.__dict__.update(kw)
            thr, bins = ns["setKLDThresh"](a, logk)
            rec[mode] = [float(thr), int(bins)]
        out[case] = rec
    out["natural_sort"] = ns["natural_sort"](["chr10", "chr2", "Chr1", "scaffold_12b", "scaffold_3", "x"], key=str)
    with open(os.path.join(GOLD, "thresholds.json"), "w") as fh:
        json.dump(out, fh, indent=0)
        fh.write("\n")
    print("thresholds      ", {k: (v["FDBins"], v["otsu"]) for k, v in out.items() if isinstance(v, dict)})


def run_projection_counts(ns):
    """Row f4: computeKmers(sym=True, pcaMode=True) -> scrubMirrors -> flattenKmerMap(prop=True) (L1571-1591) on a few
    windows of a fixture."""
    class A:
        pcaMin, pcaMax, minWordSize, maxWordSize, maskHost, hostSeq = 1, 4, 1, 8, False, None
    recs = list(ns["iterFasta"](os.path.join(INP, "markov_islands.fa")))
    seq = recs[0][1]
    wins = [("chrA:%d:%d" % (a + 1, b), seq[a:b]) for a, b in ((0, 400), (380, 800), (2200, 2600), (2690, 3090))]
    blank = ns["rangeMaps"](A.pcaMin, A.pcaMax)
    out = {"pcaMin": A.pcaMin, "pcaMax": A.pcaMax, "windows": [], "keys": None}
    for name, target in wins:
        cm = ns["computeKmers"](A, genomepickle=None, window=[(name, target)], genomeMode=False, pcaMode=True,
                                kmerMap=blank, getMeta=False, sym=True)
        uniq = ns["scrubMirrors"](cm)
        vec = ns["flattenKmerMap"](uniq, window=400, seqLen=len(target), kmin=A.pcaMin, kmax=A.pcaMax, prop=True)
        if out["keys"] is None:
            out["keys"] = [k for d in uniq for k in d.keys()]
        out["windows"].append({"label": name, "seq": target, "vector": [float(v) for v in vec]})
    with open(os.path.join(GOLD, "projection_counts.json"), "w") as fh:
        json.dump(out, fh, indent=0)
        fh.write("\n")
    print("projection_counts %d windows x %d features" % (len(out["windows"]), len(out["keys"])))


def _frame(rows, rip):
    import pandas as pd
    cols = ["name", "start", "stop", "windowKLD", "GC"] + (["PI", "SI", "CRI"] if rip else [])
    df = pd.DataFrame([tuple(r[:len(cols)]) for r in rows], columns=cols)
    return df


class _ThresholdModel:
    """Deterministic stand-in for the fitted hmmlearn model: state 1 above a cut."""
    def __init__(self, cut):
        self.cut = cut

    def predict(self, data):
        return (np.asarray(data, dtype=float)[:, 0] > self.cut).astype(int)


def run_writers(ns):
    """Rows f1 / f3: the reference's own selection logic and GFF3 writers (thresholdKLD L647-662, thresholdRIP L692-720,
    anomaly2GFF L553-567, RIP2GFF L577-587, hmmBED2GFF L589-596, findBaseRanges L91-104, range2interval L787-795,
    hmm2BED L757-785) on hand-made inputs and on the rows of a golden scan case.  Third-party calls go to stand-ins:
    pybedtools -> BedToolStub, the fitted hmmlearn model -> a cut-off model or frisk_amd.hmm.GaussianHMM2 (hmmlearn's
    numbers are unpinnable; everything the reference does AROUND the model is its own code, run here).
    pandas: DataFrame.as_matrix (removed in pandas 1.0) and the old `df[['col']] = Series` assignment are supplied for
    the duration of the run."""
    import warnings
    import pandas as pd
    warnings.simplefilter("ignore")
    pd.options.mode.chained_assignment = None
    had = hasattr(pd.DataFrame, "as_matrix")
    if not had:
        pd.DataFrame.as_matrix = lambda self, columns=None: (self[columns] if columns is not None else self).values
    orig_setitem = pd.DataFrame.__setitem__

    def old_setitem(self, key, value):      # pandas of the reference's day took df[['col']] = Series (L658)
        if isinstance(key, list) and len(key) == 1 and isinstance(value, pd.Series):
            key = key[0]
        return orig_setitem(self, key, value)
    pd.DataFrame.__setitem__ = old_setitem
    nan = float("nan")

    class A:
        pass

    def args(**kw):
        a = A()
        a.__dict__.update(dict(findSelf=False, mergeDist=0, dimReduce="windows", minPI=1.0, maxSI=1.0, minCRI=0.0, peakCRI=1.0))
        a.__dict__.update(kw)
        return a
    out = {}
    try:
        # ---- findBaseRanges
        fb = []
        for s, ch, name, minlen in (([0, 0, 1, 0, 1, 1], 0, None, 0), ([0, 0, 1, 0, 1, 1], 1, None, 0), ([], 0, None, 0),
                                    ([1, 1, 1], 0, None, 0), ([1, 1, 1], 1, "scaf", 0), ([0, 1, 1, 1, 0, 0, 1, 0], 1, None, 2),
                                    ([0, 1, 1, 1, 0, 0, 1, 0], 0, "x", 1), ("aaNNNaNa", "N", None, 0)):
            got = ns["findBaseRanges"](np.array(s) if not isinstance(s, str) else s, ch, name=name, minlen=minlen)
            fb.append({"s": s if isinstance(s, str) else list(s), "ch": ch, "name": name, "minlen": minlen,
                       "ranges": [list(map(lambda v: v if isinstance(v, str) else int(v), r)) for r in got]})
        out["findBaseRanges"] = fb
        # ---- hmmBED2GFF on hand-made interval lists (1, 3 and 12 records: ID padding)
        hb = []
        base = [("chr10", "1", "34000", "State1"), ("chr10", "31001", "45000", "State2"), ("chr2", "1", "64000", "State1")]
        for recs in (base[:1], base, base * 4):
            hb.append({"intervals": [list(r) for r in recs], "text": "".join(ns["hmmBED2GFF"](list(recs)))})
        hb.append({"intervals": [], "text": "".join(ns["hmmBED2GFF"]([]))})
        out["hmmBED2GFF"] = hb
        # ---- anomaly2GFF: merged features as pybedtools hands them over (strings)
        feats = [["chrA", "1", "550", "0.3", "0.2", "0.25"], ["chrA", "901", "1300", "0.25", "0.25", "0.25"],
                 ["chrB", "1", "400", "0.9", "0.9", "0.9"]]
        an = []
        for f, kw, cat in ((feats, dict(dimReduce="windows"), None), (feats * 4, dict(dimReduce="features"), None),
                           (feats[:1], dict(dimReduce="windows"), "Self"), ([], dict(dimReduce="windows"), None)):
            bed = BedToolStub(f)
            gen = ns["anomaly2GFF"](bed, args(**kw), **({"category": cat} if cat else {}))
            an.append({"features": f, "dimReduce": kw["dimReduce"], "category": cat, "text": "".join(gen)})
        out["anomaly2GFF"] = an
        # ---- RIP2GFF (natural scaffold order, ID padding)
        rf = [["s1", "1", "150", "0.2", "1.2", "0.5", "0.7", "1.1"], ["s10", "1", "100", "0.1", "2", "0.1", "1.9", "1.9"],
              ["s2", "1", "100", "0.1", "2", "0.1", "1.9", "1.9"], ["S2", "500", "900", "0.4", "1.5", "0.2", "0.1", "1.3"]]
        out["RIP2GFF"] = [{"features": f, "text": "".join(ns["RIP2GFF"](BedToolStub(f)))} for f in (rf, rf[:1], rf * 3)]
        # ---- thresholdKLD: selection (what reaches BedTool) and merged features, hand-made table
        rows = [("chrB", 1, 400, 0.9, 0.5), ("chrA", 151, 550, 0.2, 0.5), ("chrA", 1, 400, 0.3, 0.5), ("chrA", 901, 1300, 0.25, 0.5),
                ("chrA", 301, 700, 0.01, 0.5), ("chrA", 2000, 2400, nan, 0.5), ("chrA", 1290, 1700, 0.125, 0.5),
                ("chr10", 5, 80, 0.1, 0.5), ("chr10", 81, 90, 0.1000000001, 0.5)]
        tk = []
        for thr, kw, merge in ((np.log10(0.1), {}, True), (np.log10(0.1), dict(findSelf=True), True),
                               (np.log10(0.1), dict(mergeDist=600), True), (np.log10(0.2), dict(mergeDist=1), True),
                               (np.log10(0.1), {}, False), (np.log10(5.0), {}, True)):
            BedToolStub.made = []
            bed, picked = ns["thresholdKLD"](_frame(rows, False), float(thr), args(**kw), threshCol="windowKLD", merge=merge)
            tk.append({"threshold": float(thr), "findSelf": bool(kw.get("findSelf", False)), "mergeDist": kw.get("mergeDist", 0),
                       "merge": merge, "selected": BedToolStub.made[0], "features": [list(r) for r in bed],
                       "picked_index": [int(i) for i in picked.index]})
        out["thresholdKLD"] = {"rows": [list(r) for r in rows], "runs": tk}
        # ---- thresholdRIP: hand-made table
        rrows = [("s1", 1, 100, 0.1, 0.5, 1.2, 0.5, 0.7), ("s1", 51, 150, 0.2, 0.5, 1.5, 0.4, 1.1), ("s1", 500, 600, 0.3, 0.5, 1.1, 0.9, 0.2),
                 ("s1", 700, 800, 0.3, 0.5, nan, 0.9, nan), ("s10", 1, 100, 0.1, 0.5, 2.0, 0.1, 1.9), ("s2", 1, 100, 0.1, 0.5, 2.0, 0.1, 1.9),
                 ("s2", 100, 220, 0.15, 0.5, 1.0, 1.0, 0.0), ("s2", 400, 500, 0.15, 0.5, 1.3, 0.2, 0.4)]
        tr = []
        for kw in ({}, dict(peakCRI=1.5), dict(minPI=1.4), dict(peakCRI=5.0)):
            bed = ns["thresholdRIP"](_frame(rrows, True), args(**kw))
            tr.append({"args": kw, "features": None if bed is None else [list(r) for r in bed]})
        out["thresholdRIP"] = {"rows": [list(r) for r in rrows], "runs": tr}
        # ---- hmm2BED with a cut-off model, hand-made track (two scaffolds, a NaN row, single-window runs)
        k1 = [0.04, 0.05, 0.2, 0.21, 0.04, 0.3, 0.04, 0.04]
        k2 = [0.3, 0.04, nan, 0.04, 0.35]
        hrows = [("chr2", 1 + 1000 * i, 5000 + 1000 * i, v, 0.5) for i, v in enumerate(k1)]
        hrows += [("chr10", 1 + 1000 * i, 5000 + 1000 * i, v, 0.5) for i, v in enumerate(k2)]
        hrows += [("solo", 1, 5000, 0.01, 0.5)]
        BedToolStub.made = []
        bed, frame = ns["hmm2BED"](_frame(hrows, False), _ThresholdModel(0.1), dataCol="windowKLD")
        out["hmm2BED"] = {"rows": [list(r) for r in hrows], "cut": 0.1, "intervals": [list(r) for r in bed],
                          "gff": "".join(ns["hmmBED2GFF"](bed)),
                          "hmmState": [None if v != v else int(v) for v in frame["hmmState"].tolist()]}
        # ---- end to end on the rows of a golden scan case, as the CLI tests run it (reference KLD values)
        sys.path.insert(0, REPO)
        from frisk_amd.hmm import GaussianHMM2
        e2e = {}
        for case, force, md, ripkw in (("markov_k6", 0.12, 10, dict(minCRI=-0.5, peakCRI=0.1, minPI=0.9, maxSI=1.1)),
                                       ("markov_m2k4", 0.02, 0, dict(minCRI=-1.0, peakCRI=0.0, minPI=0.8, maxSI=1.2)),
                                       ("k8_w2000", 0.2, 600, None)):
            doc = json.load(open(os.path.join(GOLD, case + ".json")))
            rip = ripkw is not None
            trows = [(r["name"], r["start"], r["stop"], r["KLD"], r["GC"]) + (tuple(r["RIP"]) if rip else ()) for r in doc["rows"]]
            df = _frame(trows, rip)
            a = args(mergeDist=md, **(ripkw or {}))
            bed, _ = ns["thresholdKLD"](df.copy(), float(np.log10(force)), a, threshCol="windowKLD", merge=True)
            rec = {"forceThresholdKLD": force, "mergeDist": md, "rip_args": ripkw,
                   "anomaly_gff": "".join(ns["anomaly2GFF"](bed, a))}
            if rip:
                rb = ns["thresholdRIP"](df.copy(), a)
                rec["rip_gff"] = None if rb is None else "".join(ns["RIP2GFF"](rb))
            kld = np.array([r[3] for r in trows], dtype=float)
            model = GaussianHMM2().fit(kld[~np.isnan(kld)])
            hb2, _ = ns["hmm2BED"](df.copy(), model, dataCol="windowKLD")
            rec["hmm_gff"] = "".join(ns["hmmBED2GFF"](hb2))
            rec["hmm_model"] = "frisk_amd.hmm.GaussianHMM2 fitted on the case's KLD column (hmmlearn absent: numbers unpinned)"
            e2e[case] = rec
        out["e2e"] = e2e
        # ---- the progress lines crawlGenome logs (L212-250), per golden scan case with small / N-heavy scaffolds
        class Grab(logging.Handler):
            def __init__(self):
                logging.Handler.__init__(self, level=logging.INFO)
                self.lines = []

            def emit(self, record):
                self.lines.append(record.getMessage())
        crawl = {}
        root = logging.getLogger()
        level = root.level
        root.setLevel(logging.INFO)
        for name, host, query, kw, _ in CASES:
            if name not in ("smalls_skip", "smalls_all", "nheavy_k4", "markov_k6", "overshoot", "hq_k6"):
                continue
            grab = Grab()
            root.addHandler(grab)
            try:
                a = Args(os.path.join(INP, host), os.path.join(INP, query) if query else None, **kw)
                for _ in ns["crawlGenome"](a, a.querySeq if a.querySeq else a.hostSeq):
                    pass
            finally:
                root.removeHandler(grab)
            crawl[name] = [ln for ln in grab.lines if not ln.startswith("Loading fasta")]      # (carries an absolute path)
        root.setLevel(level)
        out["crawl_log"] = crawl
    finally:
        pd.DataFrame.__setitem__ = orig_setitem
        if not had:
            del pd.DataFrame.as_matrix
    with open(os.path.join(GOLD, "writers.json"), "w") as fh:
        json.dump(out, fh, indent=0)
        fh.write("\n")
    print("writers          ", {k: (len(v) if isinstance(v, list) else "ok") for k, v in out.items()})


def run_cli_surface(ns):
    """Row f2: every option of the reference's parser (L1130-1393): flags, dest, default, type, choices, nargs,
    action, required.  mainArgs() parses sys.argv at its end; parse_args is intercepted to get the parser."""
    class Grab(Exception):
        pass
    orig = argparse.ArgumentParser.parse_args

    def grab(self, *a, **k):
        e = Grab()
        e.parser = self
        raise e
    argparse.ArgumentParser.parse_args = grab
    try:
        ns["mainArgs"]()
    except Grab as e:
        parser = e.parser
    finally:
        argparse.ArgumentParser.parse_args = orig
    opts = []
    for a in parser._actions:
        if a.dest == "help":
            continue
        opts.append({"flags": list(a.option_strings), "dest": a.dest, "default": a.default,
                     "type": getattr(a.type, "__name__", None), "choices": list(a.choices) if a.choices else None,
                     "nargs": a.nargs, "action": type(a).__name__, "required": bool(a.required)})
    with open(os.path.join(GOLD, "cli_surface.json"), "w") as fh:
        json.dump({"prog": parser.prog, "options": opts}, fh, indent=0)
        fh.write("\n")
    print("cli_surface      %d options" % len(opts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*")
    opts = ap.parse_args()
    logging.basicConfig(level=logging.WARNING)
    ns = load_reference_functions()
    for name, host, query, kw, store_ivom in CASES:
        if opts.cases and name not in opts.cases:
            continue
        run_case(ns, name, host, query, kw, store_ivom)
    if not opts.cases or "thresholds" in opts.cases:
        run_thresholds(ns)
    if not opts.cases or "cli_surface" in opts.cases:
        run_cli_surface(ns)
    if not opts.cases or "projection_counts" in opts.cases:
        run_projection_counts(ns)
    if not opts.cases or "writers" in opts.cases:
        run_writers(ns)


if __name__ == "__main__":
    main()

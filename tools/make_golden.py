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
             "scrubMirrors", "flattenKmerMap"]                      # symmetric counts for the projection (row f4)


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
    ns = dict(Counter=Counter, math=math, copy=copy, gzip=gzip, logging=logging, pickle=pickle, re=re,
              sys=sys, os=os, itertools=itertools, np=_NPX(), LETTERS=("A", "T", "G", "C"), argparse=argparse,
              FRISK_VERSION="0+unknown")
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
    ("markov_k6", "markov_islands.fa", None, dict(m=1, k=6, w=400, i=150, RIP=True), False),
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


if __name__ == "__main__":
    main()

"""Rows f1 / f3 against the REFERENCE'S OWN functions (tests/golden/writers.json, made by tools/make_golden.py from
thresholdKLD L647-662, thresholdRIP L692-720, anomaly2GFF L553-567, RIP2GFF L577-587, hmmBED2GFF L589-596,
findBaseRanges L91-104, range2interval L787-795, hmm2BED L757-785): selection, ordering, run extraction and every
byte of the GFF3 text.  The two third-party pieces stay unpinned and are named in the golden: interval merging
(bedtools absent; the generator's stand-in is an independent restatement of `merge` / `window`) and the HMM's
numbers (hmmlearn absent; a cut-off model and this package's own model are handed to the reference's hmm2BED)."""
import json
import os

import numpy as np
import pytest

from golden_util import GOLD
from frisk_amd import hmm
from frisk_amd import postprocess as pp

W = json.load(open(os.path.join(GOLD, "writers.json")))


def _args(**kw):
    base = dict(forceThresholdKLD=None, threshTypeKLD=None, percentileKLD=99.0, findSelf=False, mergeDist=0,
                dimReduce="windows", minPI=1.0, maxSI=1.0, minCRI=0.0, peakCRI=1.0)
    base.update(kw)
    return type("A", (), base)()


def _strs(recs):
    return [[str(f) for f in r] for r in recs]


def _rows(rows):
    return [tuple(float("nan") if v is None else v for v in r) for r in rows]


@pytest.mark.parametrize("g", W["findBaseRanges"], ids=lambda g: "%s-%s-%s" % (g["ch"], g["name"], g["minlen"]))
def test_findBaseRanges(g):
    got = hmm.findBaseRanges(g["s"], g["ch"], name=g["name"], minlen=g["minlen"])
    assert [list(r) for r in got] == g["ranges"]
    if g["minlen"] == 0 and g["name"] is None:
        assert [list(r) for r in hmm.state_runs(list(g["s"]), g["ch"])] == g["ranges"]


@pytest.mark.parametrize("g", W["hmmBED2GFF"], ids=lambda g: str(len(g["intervals"])))
def test_hmmBED2GFF_text(g):
    assert "".join(hmm.hmmBED2GFF([tuple(r) for r in g["intervals"]])) == g["text"]


@pytest.mark.parametrize("g", W["anomaly2GFF"], ids=lambda g: "%d-%s-%s" % (len(g["features"]), g["dimReduce"], g["category"]))
def test_anomaly2GFF_text(g):
    kw = {"category": g["category"]} if g["category"] else {}
    assert "".join(pp.anomaly2GFF([tuple(f) for f in g["features"]], _args(dimReduce=g["dimReduce"]), **kw)) == g["text"]


@pytest.mark.parametrize("g", W["RIP2GFF"], ids=lambda g: str(len(g["features"])))
def test_RIP2GFF_text(g):
    assert "".join(pp.RIP2GFF([tuple(f) for f in g["features"]])) == g["text"]


@pytest.mark.parametrize("run", W["thresholdKLD"]["runs"],
                         ids=lambda r: "thr%.2f-self%d-d%d-m%d" % (r["threshold"], r["findSelf"], r["mergeDist"], r["merge"]))
def test_thresholdKLD_selection_and_features(run, monkeypatch):
    monkeypatch.setenv("FRISK_FLOAT_REPR", "py3")       # the golden was made under Python 3: str(float) is repr there
    rows = _rows(W["thresholdKLD"]["rows"])
    a = _args(findSelf=run["findSelf"], mergeDist=run["mergeDist"])
    feats, chosen = pp.thresholdKLD(list(rows), run["threshold"], a, merge=run["merge"])
    assert [[r[0], str(r[1]), str(r[2]), repr(float(r[3]))] for r in chosen] == run["selected"]
    assert [rows.index(r) for r in chosen] == run["picked_index"]
    assert _strs(feats) == run["features"]
    # the same through the column form the CLI uses
    from frisk_amd.table import ScoreTable
    feats2, chosen2 = pp.thresholdKLD(ScoreTable.from_rows(list(rows), rip=False), run["threshold"], a, merge=run["merge"])
    assert _strs(feats2) == run["features"] and [tuple(r[:3]) for r in chosen2] == [tuple(r[:3]) for r in chosen]


@pytest.mark.parametrize("run", W["thresholdRIP"]["runs"], ids=lambda r: json.dumps(r["args"]))
def test_thresholdRIP_features(run):
    rows = _rows(W["thresholdRIP"]["rows"])
    feats = pp.thresholdRIP(list(rows), _args(**run["args"]))
    assert (None if feats is None else _strs(feats)) == run["features"]
    from frisk_amd.table import ScoreTable
    feats2 = pp.thresholdRIP(ScoreTable.from_rows(list(rows), rip=True), _args(**run["args"]))
    assert (None if feats2 is None else _strs(feats2)) == run["features"]


class _Cut:
    def __init__(self, cut):
        self.cut = cut

    def predict(self, x):
        return (np.asarray(x, dtype=float).ravel() > self.cut).astype(int)


def test_hmm2BED_around_a_given_model():
    g = W["hmm2BED"]
    rows = _rows(g["rows"])
    intervals, _ = hmm.hmm2BED(list(rows), model=_Cut(g["cut"]))
    assert [list(r) for r in intervals] == g["intervals"]
    assert "".join(hmm.hmmBED2GFF(intervals)) == g["gff"]
    from frisk_amd.table import ScoreTable
    intervals2, _ = hmm.hmm2BED(ScoreTable.from_rows(list(rows), rip=False), model=_Cut(g["cut"]))
    assert [list(r) for r in intervals2] == g["intervals"]


@pytest.mark.parametrize("case", sorted(W["e2e"]))
def test_feature_files_from_reference_scores(case):
    """The three GFF3 files of a run, from the reference's KLD / RIP values of a golden scan case: the text equals what
    the reference's own post-processing writes for them."""
    g = W["e2e"][case]
    doc = json.load(open(os.path.join(GOLD, case + ".json")))
    rip = g["rip_args"] is not None
    rows = [(r["name"], r["start"], r["stop"], r["KLD"], r["GC"]) + (tuple(r["RIP"]) if rip else ()) for r in doc["rows"]]
    a = _args(mergeDist=g["mergeDist"], **(g["rip_args"] or {}))
    feats, _ = pp.thresholdKLD(list(rows), float(np.log10(g["forceThresholdKLD"])), a, merge=True)
    assert "".join(pp.anomaly2GFF(feats, a)) == g["anomaly_gff"]
    if rip:
        rf = pp.thresholdRIP(list(rows), a)
        assert (None if rf is None else "".join(pp.RIP2GFF(rf))) == g["rip_gff"]
    intervals, _ = hmm.hmm2BED(list(rows))
    assert "".join(hmm.hmmBED2GFF(intervals)) == g["hmm_gff"]


@pytest.mark.parametrize("case", sorted(W["crawl_log"]))
def test_crawl_progress_lines(case):
    """crawlGenome's per-scaffold log lines (L212-250) rebuilt from per-candidate arrays: candidates laid out the way the
    scan numbers them (floor-stepped starts per scaffold, one candidate for a rescued scaffold), kept = the golden's rows."""
    from frisk_amd.fasta import readFasta
    from frisk_amd.hotpath import crawlLog
    doc = json.load(open(os.path.join(GOLD, case + ".json")))
    a = doc["args"]
    w, inc, all_ = a["windowlen"], a["increment"], a["scaffoldsAll"]
    names, seqs = readFasta(os.path.join(GOLD, "inputs", doc["query"] or doc["host"]))
    seq_index, kept = [], []
    rows = {(r["name"], r["start"], r["stop"]) for r in doc["rows"]}
    for s, (name, seq) in enumerate(zip(names, seqs)):
        size = len(seq)
        if size <= w + (w * 0.75 - inc):
            if all_:
                seq_index.append(s)
                kept.append((name, 1, size) in rows)
            continue
        for j in range(0, size - inc + 1, inc):
            seq_index.append(s)
            kept.append(((name, size - w, size) if j + w > size else (name, j + 1, j + w)) in rows)
    lines = []
    crawlLog(names, [len(x) for x in seqs], np.array(seq_index, dtype=np.int64), np.array(kept, dtype=bool), w, inc, all_,
             emit=lines.append)
    want = [ln for ln in W["crawl_log"][case] if not ln.startswith("Window from ")]
    assert lines == want

"""Row f1/f2 host logic without a GPU: thresholds against goldens from the reference's own functions, the
bedtools-merge restatement, GFF3 text, Python-2 float text, and the argparse surface."""
import json
import os

import numpy as np
import pytest

from golden_util import GOLD
from frisk_amd import postprocess as pp


def _args(**kw):
    base = dict(forceThresholdKLD=None, threshTypeKLD=None, percentileKLD=99.0, findSelf=False, mergeDist=0,
                dimReduce="windows", minPI=1.0, maxSI=1.0, minCRI=0.0, peakCRI=1.0)
    base.update(kw)
    return type("A", (), base)()


THRESH = json.load(open(os.path.join(GOLD, "thresholds.json")))


@pytest.mark.parametrize("case", [k for k in THRESH if k != "natural_sort"])
def test_thresholds_match_reference(case):
    g = THRESH[case]
    logk = np.log10(np.array(g["KLD"], dtype=float).reshape(-1, 1))        # (n,1), as as_matrix(columns=[...]) gives
    assert pp.FDBins(logk) == g["FDBins"]
    for mode, kw in (("otsu", dict(threshTypeKLD="otsu")), ("pct99", dict(threshTypeKLD="percentile")),
                     ("pct80", dict(threshTypeKLD="percentile", percentileKLD=80.0)), ("force", dict(forceThresholdKLD=0.05))):
        thr, bins = pp.setKLDThresh(_args(**kw), logk)
        assert bins == g[mode][1]
        assert float(np.ravel(thr)[0]) == g[mode][0], mode
    with pytest.raises(UnboundLocalError):
        pp.setKLDThresh(_args(), logk)


def test_natural_sort_matches_reference():
    items = ["chr10", "chr2", "Chr1", "scaffold_12b", "scaffold_3", "x"]
    assert pp.natural_sort(items) == THRESH["natural_sort"]


def test_python2_float_text():
    assert pp.py2_str(0.02102412267481919) == "0.0210241226748"
    assert pp.py2_str(0.55) == "0.55" and pp.py2_str(1.0) == "1.0" and pp.py2_str(2.0 / 3.0) == "0.666666666667"
    assert pp.py2_str(float("nan")) == "nan" and pp.py2_str(1e-5) == "1e-05" and pp.py2_str(123456789012345.0) == "1.23456789012e+14"
    assert pp.py2_str(0) == "0" and pp.py2_str(17) == "17" and pp.py2_str("kat") == "kat"
    assert pp.py3_str(0.02102412267481919) == "0.02102412267481919"


def test_merge_follows_bedtools_semantics():
    recs = [("a", 1, 100, 0.5), ("a", 100, 200, 0.7), ("a", 201, 300, 0.1), ("a", 500, 600, 0.9), ("b", 1, 50, 0.2)]
    m0 = pp.merge_intervals(recs, dist=0, ops=("max", "min", "mean"), cols=(3, 3, 3))
    # book-ended (100 == 100) merges at d=0; a gap of 1 does not
    assert m0 == [("a", 1, 200, "0.7", "0.5", "0.6"), ("a", 201, 300, "0.1", "0.1", "0.1"), ("a", 500, 600, "0.9", "0.9", "0.9"),
                  ("b", 1, 50, "0.2", "0.2", "0.2")]
    m1 = pp.merge_intervals(recs, dist=1, ops=("max",), cols=(3,))
    assert m1[0] == ("a", 1, 300, "0.7") and len(m1) == 3
    # nested interval keeps the running end
    assert pp.merge_intervals([("a", 1, 500, 1.0), ("a", 10, 20, 2.0), ("a", 400, 450, 3.0)])[0][:3] == ("a", 1, 500)
    assert pp.merge_intervals([("a", 1, 2, 0.123456789)])[0][3] == "0.12346"            # bedtools -prec 5


def test_threshold_merge_and_gff_text():
    rows = [("chrB", 1, 400, 0.9, 0.5), ("chrA", 151, 550, 0.2, 0.5), ("chrA", 1, 400, 0.3, 0.5), ("chrA", 901, 1300, 0.25, 0.5),
            ("chrA", 301, 700, 0.01, 0.5), ("chrA", 2000, 2400, float("nan"), 0.5)]
    feats, chosen = pp.thresholdKLD(rows, np.log10(0.1), _args(), merge=True)
    assert feats == [("chrA", 1, 550, "0.3", "0.2", "0.25"), ("chrA", 901, 1300, "0.25", "0.25", "0.25"), ("chrB", 1, 400, "0.9", "0.9", "0.9")]
    assert [r[:3] for r in chosen] == [("chrA", 1, 400), ("chrA", 151, 550), ("chrA", 901, 1300), ("chrB", 1, 400)]
    gff = list(pp.anomaly2GFF(feats, _args()))
    assert gff[0] == "##gff-version 3\n"
    assert gff[1] == "chrA\tfrisk_0+unknown\tKmer-anomaly\t1\t550\t.\t+\t.\tID=Anomaly_1;KLD=0.3\n"
    assert len(gff) == 4
    low, _ = pp.thresholdKLD(rows, np.log10(0.1), _args(findSelf=True), merge=True)
    assert low == [("chrA", 301, 700, "0.01", "0.01", "0.01")]
    gff2 = list(pp.anomaly2GFF(feats * 4, _args(dimReduce="features")))
    assert "ID=Anomaly_01;maxKLD=0.3;minKLD=0.2;meanKLD=0.25" in gff2[1]


def test_rip_features():
    nan = float("nan")
    rows = [("s1", 1, 100, 0.1, 0.5, 1.2, 0.5, 0.7), ("s1", 51, 150, 0.2, 0.5, 1.5, 0.4, 1.1), ("s1", 500, 600, 0.3, 0.5, 1.1, 0.9, 0.2),
            ("s1", 700, 800, 0.3, 0.5, nan, 0.9, nan), ("s10", 1, 100, 0.1, 0.5, 2.0, 0.1, 1.9), ("s2", 1, 100, 0.1, 0.5, 2.0, 0.1, 1.9)]
    feats = pp.thresholdRIP(rows, _args())
    assert feats == [("s1", 1, 150, "0.2", "1.2", "0.5", "0.7", "1.1"), ("s10", 1, 100, "0.1", "2", "0.1", "1.9", "1.9"),
                     ("s2", 1, 100, "0.1", "2", "0.1", "1.9", "1.9")]
    gff = list(pp.RIP2GFF(feats))
    assert [line.split("\t")[0] for line in gff[1:]] == ["s1", "s2", "s10"]           # natural order
    assert gff[1].rstrip().endswith("ID=Anomaly_1;maxKLD=0.2;minPI=1.2;maxSI=0.5;minCRI=0.7;maxCRI=1.1")
    assert pp.thresholdRIP(rows[:1], _args()) is None


def test_cli_surface_equals_reference():
    from frisk_amd.cli import build_parser, makePicklePath
    ref = json.load(open(os.path.join(GOLD, "cli_surface.json")))
    parser = build_parser()
    assert parser.prog == ref["prog"]
    mine = []
    for a in parser._actions:
        if a.dest == "help":
            continue
        mine.append({"flags": list(a.option_strings), "dest": a.dest, "default": a.default,
                     "type": getattr(a.type, "__name__", None), "choices": list(a.choices) if a.choices else None,
                     "nargs": a.nargs, "action": type(a).__name__, "required": bool(a.required)})
    assert mine == ref["options"]
    # cache file names (L497-506) against what the reference produced for a golden case
    doc = json.load(open(os.path.join(GOLD, "hq_k6.json")))
    a = parser.parse_args(["-H", "x/host.fa", "-Q", "y/query.fa", "-k", "6", "-w", "500", "-i", "100", "-t", "T"])
    assert os.path.basename(makePicklePath(a, "genome")) == doc["genome_pickle_basename"]
    assert os.path.basename(makePicklePath(a, "window")) == doc["window_pickle_basename"]
    # --recalc / --recalcWin are store_false: PASSING them forces recomputation
    assert a.recalc is True and parser.parse_args(["-H", "h", "--recalc"]).recalc is False


def test_column_merge_equals_the_row_merge():
    """thresholdKLD's path for big tables (sort + bedtools-style merge on numpy columns) against the row-by-row restatement:
    same features, same text, for random tables with duplicate names, overlapping / book-ended / nested windows and merge
    distances 0 and > 0; and the reload of a pickled frame (ScoreTable.from_frame on columns) against from_rows."""
    from types import SimpleNamespace
    from frisk_amd import postprocess as pp
    from frisk_amd.table import ScoreTable
    rng = np.random.default_rng(8)
    for trial in range(6):
        names = ["chr%d" % i for i in rng.permutation(12)] + ["chr3"]            # a name twice: grouped by NAME
        rows = []
        for s, nm in enumerate(names):
            pos = 1
            for _ in range(int(rng.integers(2500, 3500))):
                pos += int(rng.integers(0, 4)) * 500
                w = int(rng.choice([1000, 5000, 200]))
                rows.append((nm, pos, pos + w - 1, float(abs(rng.normal(0.05, 0.05)) + 1e-4), 0.5))
        table = ScoreTable.from_rows(rows, rip=False)
        thr = float(np.log10(np.percentile(table.kld, 30)))
        for dist in (0, 700):
            a = SimpleNamespace(findSelf=False, mergeDist=dist)
            fast, sel = pp.thresholdKLD(table, thr, a, merge=True)
            slow, sel2 = pp.thresholdKLD(list(rows), thr, a, merge=True)
            assert len(sel) == len(sel2) > 20000
            assert fast == slow
            assert list(sel)[:50] == sel2[:50]
        back = ScoreTable.from_frame(table.to_frame(), rip=False)
        assert back.names == ScoreTable.from_rows(rows, rip=False).names
        for f in ("seq_index", "start", "stop", "kld", "gc"):
            assert np.array_equal(getattr(back, f), getattr(table, f)), f


def test_python_max_semantics_of_the_otsu_scale():
    from frisk_amd import postprocess as pp
    nan = float("nan")
    for vals in ([1.0, 3.0, 2.0], [nan, 3.0, 2.0], [1.0, nan, 5.0, nan], [2.0], [0.5, nan]):
        a = np.asarray(vals).reshape(-1, 1)
        want = max(a)           # what the reference evaluates (row by row)
        got = pp._py_max(a)
        assert np.array_equal(np.ravel(want), np.ravel(got), equal_nan=True), vals
        assert np.array_equal(np.ravel(max(np.ravel(a))), np.ravel(pp._py_max(np.ravel(a))), equal_nan=True)


def test_merge_against_documented_bedtools_semantics():
    """tests/golden/bedtools_merge_semantics.json: cases transcribed from the documented behaviour of `bedtools merge`
    (the manual's own examples first) - data that neither restatement of this package produced.  Both the row merge and the
    column merge must give them."""
    import json
    import os
    from frisk_amd import postprocess as pp
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bedtools_merge_semantics.json")
    doc = json.load(open(path))
    assert len(doc["cases"]) >= 8
    for c in doc["cases"]:
        recs = [tuple(r) for r in c["records"]]
        got = pp.merge_intervals(recs, dist=c["d"], ops=tuple(c["ops"]), cols=tuple(c["cols"]))
        assert [list(g) for g in got] == c["expected"], c["name"]
        if c["d"] >= 0:                  # (the column merge is used for non-negative distances only)
            names = [r[0] for r in recs]
            rank = pp._name_ranks(names)
            start = np.array([r[1] for r in recs], dtype=np.int64)
            stop = np.array([r[2] for r in recs], dtype=np.int64)
            width = max(len(r) for r in recs)
            vals = [np.array([float(r[k]) if k < len(r) else 0.0 for r in recs]) for k in range(3, width)]
            fast = pp.merge_columns(names, rank, start, stop, vals, dist=c["d"], ops=tuple(c["ops"]), cols=tuple(k - 3 for k in c["cols"]))
            assert [list(g) for g in fast] == c["expected"], c["name"]


def test_rip_features_on_columns_equal_the_row_path():
    """thresholdRIP on a ScoreTable (merge and the bedtools-window overlap test on numpy columns) == the row-by-row path."""
    from types import SimpleNamespace
    from frisk_amd import postprocess as pp
    from frisk_amd.table import ScoreTable
    rng = np.random.default_rng(12)
    for trial in range(8):
        rows = []
        for nm in ["c%d" % i for i in rng.permutation(9)]:
            for j in range(int(rng.integers(50, 900))):
                pi, si = float(rng.normal(1.05, 0.15)), float(rng.normal(0.95, 0.15))
                cri = float("nan") if rng.random() < 0.03 else pi - si
                rows.append((nm, 1 + 1000 * j, 5000 + 1000 * j, float(abs(rng.normal(0.05, 0.02))), 0.5, pi, si, cri))
        a = SimpleNamespace(minPI=1.1, maxSI=0.9, minCRI=0.1, peakCRI=float(rng.choice([0.3, 0.5, 5.0])))
        slow = pp.thresholdRIP(list(rows), a)
        fast = pp.thresholdRIP(ScoreTable.from_rows(rows, rip=True), a)
        assert fast == slow
        if slow:
            assert "".join(pp.RIP2GFF(fast)) == "".join(pp.RIP2GFF(slow))

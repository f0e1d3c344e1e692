"""ScoreTable + the native row formatter (frisk_format_rows) against the per-value Python 2 str() restatement."""
import numpy as np

from frisk_amd import postprocess as pp
from frisk_amd.table import ScoreTable


def _random_table(n, rip, seed=5):
    rng = np.random.default_rng(seed)
    names = ["chr1", "scaf_2 weird", "x"]
    kld = rng.random(n) * 10.0 ** rng.integers(-14, 3, n)
    kld[::7] = np.round(kld[::7])                      # integral values get '.0'
    kld[3] = 1e-5; kld[4] = 123456789012.0; kld[5] = 1e16; kld[6] = 0.1 + 0.2
    gc = rng.random(n)
    gc[1] = 0.5; gc[2] = 1.0; gc[8] = 0.0
    f = lambda: np.where(rng.random(n) < 0.2, np.nan, rng.normal(size=n) * 3)    # noqa: E731
    t = ScoreTable(names, rng.integers(0, 3, n), rng.integers(1, 10 ** 9, n), rng.integers(1, 10 ** 10, n), kld, gc,
                   f() if rip else None, f() if rip else None, f() if rip else None, (rng.random(n) < 0.05).astype(np.uint8))
    if rip:
        t.pi[0] = np.inf; t.si[0] = -np.inf
    return t


def test_native_text_equals_python2_str_per_value():
    for rip in (False, True):
        for n in (0, 12, 50, 30000):                   # 30000: the multi-threaded path
            t = _random_table(n, rip) if n else ScoreTable(["a"], [], [], [], [], [], *([[]] * 3 if rip else []))
            native = t.text()
            slow = t.text(fmt=pp.py2_str)
            assert native == slow
            assert native.count("\n") == n


def test_round_trips():
    t = _random_table(200, True)
    back = ScoreTable.from_rows(t.rows(), rip=True)
    assert back.text() == t.text()
    assert back.rows()[:5] == t.rows()[:5] or all(str(a) == str(b) for a, b in zip(back.rows()[:5], t.rows()[:5]))
    fr = t.to_frame()
    assert list(fr.columns) == ["name", "start", "stop", "windowKLD", "GC", "PI", "SI", "CRI"] and len(fr) == 200
    again = ScoreTable.from_frame(fr, rip=True)
    assert np.array_equal(again.start, t.start) and np.array_equal(again.gc, t.gc)
    legacy = fr.rename(columns={"windowKLD": "windowKLI"})
    assert np.array_equal(ScoreTable.from_frame(legacy, rip=True).kld, again.kld)

"""Row f3 on the CPU: the 2-state Gaussian HMM and the reference's logic around it (stacking, per-scaffold
decoding, run extraction, string-sorted intervals, GFF3 text)."""
import numpy as np

from frisk_amd.hmm import GaussianHMM2, hmm2BED, hmmBED2GFF, state_runs


def test_state_runs_keep_single_windows():
    assert state_runs([0, 0, 1, 0, 1, 1], 0) == [(0, 1), (3, 3)]
    assert state_runs([0, 0, 1, 0, 1, 1], 1) == [(2, 2), (4, 5)]
    assert state_runs([], 0) == []


def test_hmm_recovers_two_regimes_and_writes_gff():
    rng = np.random.default_rng(5)
    low = lambda n: (0.04 + 0.004 * rng.standard_normal(n)).tolist()       # noqa: E731
    high = lambda n: (0.12 + 0.01 * rng.standard_normal(n)).tolist()       # noqa: E731
    k1 = low(60) + high(25) + low(40)
    k2 = low(30) + [float("nan")] + high(10)
    rows = [("chr2", 1 + 1000 * i, 5000 + 1000 * i, v, 0.5) for i, v in enumerate(k1)]
    rows += [("chr10", 1 + 1000 * i, 5000 + 1000 * i, v, 0.5) for i, v in enumerate(k2)]
    intervals, model = hmm2BED(rows)
    assert model.means_[0] < 0.06 < model.means_[1]
    got = {(n, s, e, st) for n, s, e, st in intervals}
    assert ("chr2", "1", "64000", "State1") in got and ("chr2", "60001", "89000", "State2") in got
    assert ("chr2", "85001", "129000", "State1") in got
    assert ("chr10", "1", "34000", "State1") in got and ("chr10", "31001", "45000", "State2") in got   # NaN row skipped
    # the reference sorts the STRING tuples (L783): 'chr10' < 'chr2', and starts compare as text
    assert intervals == sorted(intervals, key=lambda t: (t[0], t[1], t[2]))
    assert [t[0] for t in intervals][:2] == ["chr10", "chr10"]
    gff = list(hmmBED2GFF(intervals))
    assert gff[0] == "##gff-version 3\n" and len(gff) == len(intervals) + 1
    f = gff[1].rstrip("\n").split("\t")
    assert f[1] == "frisk_0+unknown" and f[2] in ("State1", "State2") and f[8] == "ID=%s_1" % f[2]
    # deterministic: same input, same model
    again, model2 = hmm2BED(rows)
    assert again == intervals and np.array_equal(model.means_, model2.means_)


def test_viterbi_prefers_staying():
    m = GaussianHMM2().fit(np.array([0.0] * 50 + [1.0] * 50 + [0.0] * 50) + 1e-3 * np.arange(150) % 0.01)
    p = m.predict(np.array([0.0, 0.0, 0.55, 0.0, 0.0]))
    assert p.tolist() == [0, 0, 0, 0, 0] or p.tolist() == [0, 0, 1, 0, 0]
    assert m.predict(np.array([])).size == 0

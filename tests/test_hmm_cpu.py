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


# ---------------------------------------------------------------------------------------------------------------------------
# The model's arithmetic against an exhaustive enumeration of state paths (oracle/hmm_exhaustive.py): nothing below compares
# the model with itself.  hmmlearn itself is absent (parity with the third party unpinned); what is pinned here
# is that forward-backward, the M step with hmmlearn's default priors, and Viterbi compute what their definitions say - for
# the numpy specification (native=False) and for the library's host-native form (csrc/hmm_host.h).
# ---------------------------------------------------------------------------------------------------------------------------
import pytest

from oracle import hmm_exhaustive as X


def _cases(n_cases=40, seed=17):
    rng = np.random.default_rng(seed)
    for c in range(n_cases):
        n = int(rng.integers(1, 13))
        regime = rng.integers(0, 2, n)
        x = np.where(regime == 0, rng.normal(0.04, 0.01, n), rng.normal(0.15, 0.04, n))
        if c % 7 == 0 and n > 2:
            x[int(rng.integers(0, n))] = 2.5            # an outlier far from both means (underflows a naive scaled recursion)
        means = sorted(rng.uniform(0.0, 0.3, 2).tolist())
        covars = rng.uniform(1e-4, 2e-2, 2).tolist()
        p0 = float(rng.uniform(0.05, 0.95))
        a, b = float(rng.uniform(0.02, 0.98)), float(rng.uniform(0.02, 0.98))
        yield x, means, covars, [p0, 1 - p0], [[a, 1 - a], [1 - b, b]]


def _model(means, covars, start, trans, native):
    m = GaussianHMM2(native=native)
    m.means_, m.covars_ = np.array(means, float), np.array(covars, float)
    m.startprob_, m.transmat_ = np.array(start, float), np.array(trans, float)
    return m


def test_forward_backward_equals_the_sum_over_all_paths():
    """Likelihood and posteriors of the numpy specification (log-space recursion) == brute force over 2^n paths, to 1e-12."""
    for x, means, covars, start, trans in _cases():
        e = X.enumerate_paths(x.tolist(), means, covars, start, trans)
        m = _model(means, covars, start, trans, native=False)
        b = m._loglik(x)
        fwd, bwd, ll, _lt = m._forward_backward(b)
        assert abs(ll - e["loglik"]) <= 1e-12 * max(1.0, abs(e["loglik"]))
        post = np.exp(fwd + bwd - ll)
        assert np.max(np.abs(post - np.array(e["gamma"]))) <= 1e-12


@pytest.mark.parametrize("native", [False, True])
def test_viterbi_is_the_most_probable_path(native):
    checked = 0
    for x, means, covars, start, trans in _cases(60, seed=23):
        e = X.enumerate_paths(x.tolist(), means, covars, start, trans)
        if e["best_logp"] - e["runner_up_logp"] < 1e-9:         # (a tie: either path is a correct answer)
            continue
        got = _model(means, covars, start, trans, native).predict(x)
        assert got.tolist() == e["best_path"], (x, means, covars)
        checked += 1
    assert checked > 50


@pytest.mark.parametrize("native", [False, True])
def test_one_em_round_equals_the_closed_form_on_exact_posteriors(native):
    """fit(n_iter=1) from the deterministic start == the M step on the enumerated posteriors (hmmlearn's default priors), and
    the reported log-likelihood is that of the START model; three rounds == the closed form applied three times."""
    rng = np.random.default_rng(31)
    for c in range(25):
        n = int(rng.integers(3, 13))
        regime = (np.arange(n) * 3 // n) % 2
        x = np.where(regime == 0, rng.normal(0.04, 0.01, n), rng.normal(0.15, 0.04, n))
        init = GaussianHMM2(native=False)
        init._init(x)
        p = (init.means_.tolist(), init.covars_.tolist(), init.startprob_.tolist(), init.transmat_.tolist())
        for rounds in (1, 3):
            q, ll = p, None
            for _ in range(rounds):
                nm, nc, ns, nt, ll = X.em_step(x.tolist(), *q)
                q = (nm, nc, ns, nt)
            m = GaussianHMM2(n_iter=rounds, tol=-1e300, native=native).fit(x)
            assert m.n_iter_ == rounds
            assert abs(m.loglik_ - ll) <= 1e-10 * max(1.0, abs(ll))
            for got, want in zip((m.means_, m.covars_, m.startprob_, m.transmat_), q):
                assert np.max(np.abs(np.ravel(got) - np.ravel(np.array(want)))) <= 1e-10 * max(1.0, float(np.max(np.abs(want)))), (c, rounds)


def test_native_fit_equals_the_numpy_specification():
    """csrc/hmm_host.h (scaled recursion over fixed pieces, in parallel) against the numpy model on sequences long enough for
    several pieces: same number of EM rounds, parameters to 1e-8 relative (the log-space recursion itself carries ~1e-11: its
    terms are of the size of the log-likelihood, 1e4..1e5), same Viterbi states; and on the golden writer cases' scale."""
    rng = np.random.default_rng(41)
    for n, flip in ((5000, 0.03), (30000, 0.01), (9, 0.3), (1, 0.5), (2, 0.5)):
        st = np.cumsum(rng.random(n) < flip) % 2
        x = np.where(st == 0, rng.normal(0.03, 0.01, n), rng.normal(0.12, 0.05, n)).clip(1e-4, None)
        a, b = GaussianHMM2(native=False).fit(x), GaussianHMM2(native=True).fit(x)
        assert a.n_iter_ == b.n_iter_
        for f in ("means_", "covars_", "startprob_", "transmat_"):
            u, v = np.ravel(getattr(a, f)), np.ravel(getattr(b, f))
            assert np.max(np.abs(u - v)) <= 1e-8 * max(1.0, float(np.max(np.abs(u)))), (n, f)
        assert abs(a.loglik_ - b.loglik_) <= 1e-9 * max(1.0, abs(a.loglik_))
        assert np.array_equal(a._predict_py(x), b.predict(x))
        off = np.array([0, n // 3, n // 3, n], dtype=np.int64)       # segments (one empty) decoded independently
        seg = b.predict_segments(x, off)
        assert np.array_equal(seg[:n // 3], b.predict(x[:n // 3])) and np.array_equal(seg[n // 3:], b.predict(x[n // 3:]))
    with pytest.raises(ValueError):
        GaussianHMM2().fit(np.array([0.1, float("nan")]))


def test_table_path_of_hmm2BED_equals_the_row_path():
    """hmm2BED on a ScoreTable (columns, one native Viterbi call over all scaffolds) == the row-by-row path, including a name
    that occurs on two scaffolds, NaN rows, and scaffolds whose rows interleave."""
    from frisk_amd.table import ScoreTable
    rng = np.random.default_rng(43)
    rows = []
    for name, n in (("chr2", 300), ("chr10", 120), ("chr2", 80), ("chrX", 1), ("chr7", 40)):
        st = np.cumsum(rng.random(n) < 0.05) % 2
        k = np.where(st == 0, rng.normal(0.03, 0.008, n), rng.normal(0.12, 0.03, n)).clip(1e-4, None)
        if n > 50:
            k[int(rng.integers(0, n))] = float("nan")
        base = len(rows) * 1000
        rows += [(name, base + 1 + 1000 * i, base + 5000 + 1000 * i, float(v), 0.5) for i, v in enumerate(k)]
    want, m1 = hmm2BED(list(rows))
    got, m2 = hmm2BED(ScoreTable.from_rows(rows, rip=False))
    assert got == want and len(got) > 10
    assert np.array_equal(m1.means_, m2.means_)

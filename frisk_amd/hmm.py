"""2-state Gaussian HMM segmentation of the KLD track - host numpy.

The reference fits `hmmlearn.hmm.GaussianHMM(n_components=2, covariance_type="full")` on all non-NaN window scores
stacked as ONE sequence (frisk/__init__.py L1537-1541) and Viterbi-decodes each scaffold (hmm2BED L757-785),
then writes runs of equal state as GFF3 features (range2interval L787-795, hmmBED2GFF L589-596).
hmmlearn is a third-party dependency that is absent here and seeds its k-means initialisation randomly, so
parity for the fitted numbers is UNPINNED: this module restates the documented model (Baum-Welch with hmmlearn's
default priors: covars_prior 1e-2, covars_weight 1, n_iter 10, tol 1e-2; Viterbi decoding) with a deterministic
initialisation (1-D 2-means started at the extremes; state 0 = the lower mean).  What is pinned by tests is the
reference's own logic around the model: stacking, per-scaffold decoding, run extraction, the string sort of
the intervals, and the GFF3 text; and the model's arithmetic itself against an exhaustive enumeration of state paths
(oracle/hmm_exhaustive.py, tests/test_hmm_cpu.py).

Two implementations of the same model: the numpy one below (`native=False`: one Python step per window, the specification,
fine up to ~10^4 windows) and the library's host-native one (csrc/hmm_host.h through frisk_hmm_fit / frisk_hmm_viterbi: scaled
forward-backward over fixed pieces in parallel, Viterbi per scaffold in parallel - 3 M windows in a fraction of a second),
which `fit` / `predict` use by default.  They agree to rounding (parameters ~1e-12, tests).
"""
import ctypes as C

import numpy as np

from .postprocess import FRISK_VERSION

_LOG2PI = np.log(2.0 * np.pi)


def _logsumexp(a, axis):
    m = np.max(a, axis=axis, keepdims=True)
    m = np.where(np.isfinite(m), m, 0.0)
    return np.squeeze(m, axis=axis) + np.log(np.sum(np.exp(a - m), axis=axis))


class GaussianHMM2:
    """1-D, 2 states, one variance per state ("full" covariance of one feature)."""

    def __init__(self, n_iter=10, tol=1e-2, min_covar=1e-3, covars_prior=1e-2, native=True):
        self.n_iter, self.tol, self.min_covar, self.covars_prior, self.native = n_iter, tol, min_covar, covars_prior, native
        self.loglik_, self.n_iter_ = None, 0

    # -- initialisation: 2-means from the extremes, hmmlearn-style shared variance, flat start / transitions
    def _init(self, x):
        c = np.array([x.min(), x.max()], dtype=float)
        for _ in range(100):
            lab = np.abs(x[:, None] - c[None, :]).argmin(axis=1)
            new = np.array([x[lab == s].mean() if np.any(lab == s) else c[s] for s in (0, 1)])
            if np.allclose(new, c):
                break
            c = new
        self.means_ = np.sort(c)
        self.covars_ = np.full(2, np.var(x) + self.min_covar)
        self.startprob_ = np.array([0.5, 0.5])
        self.transmat_ = np.full((2, 2), 0.5)

    def _loglik(self, x):
        return -0.5 * (_LOG2PI + np.log(self.covars_)[None, :] + (x[:, None] - self.means_[None, :]) ** 2 / self.covars_[None, :])

    def _forward_backward(self, b):
        n = b.shape[0]
        with np.errstate(divide="ignore"):
            lt, ls = np.log(self.transmat_), np.log(self.startprob_)
        fwd = np.empty((n, 2))
        fwd[0] = ls + b[0]
        for t in range(1, n):
            fwd[t] = _logsumexp(fwd[t - 1][:, None] + lt, axis=0) + b[t]
        bwd = np.zeros((n, 2))
        for t in range(n - 2, -1, -1):
            bwd[t] = _logsumexp(lt + (b[t + 1] + bwd[t + 1])[None, :], axis=1)
        return fwd, bwd, _logsumexp(fwd[-1], axis=0), lt

    def _fit_native(self, x):
        from . import _ffi
        lib = _ffi.lib()
        x = np.ascontiguousarray(x, dtype=np.float64)
        means, covars, start, trans = np.zeros(2), np.zeros(2), np.zeros(2), np.zeros(4)
        ll, iters = C.c_double(), C.c_int32()
        p = lambda a: a.ctypes.data_as(C.c_void_p)      # noqa: E731
        rc = lib.frisk_hmm_fit(p(x), x.size, int(self.n_iter), float(self.tol), float(self.min_covar), float(self.covars_prior),
                               p(means), p(covars), p(start), p(trans), C.byref(ll), C.byref(iters))
        if rc != _ffi.OK:
            raise ValueError("frisk_hmm_fit: the scores must be finite and at least one (code %d)" % rc)
        self.means_, self.covars_, self.startprob_, self.transmat_ = means, covars, start, trans.reshape(2, 2)
        self.loglik_, self.n_iter_ = float(ll.value), int(iters.value)
        return self

    def predict_segments(self, x, seg_off):
        """Viterbi paths of the sequences x[seg_off[s]:seg_off[s+1]], decoded independently (one scaffold each): int8 states."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        seg_off = np.ascontiguousarray(seg_off, dtype=np.int64)
        out = np.zeros(x.size, dtype=np.int8)
        if not self.native:
            for a, b in zip(seg_off[:-1].tolist(), seg_off[1:].tolist()):
                out[a:b] = self._predict_py(x[a:b])
            return out
        from . import _ffi
        p = lambda a: a.ctypes.data_as(C.c_void_p)      # noqa: E731
        pars = [np.ascontiguousarray(a, dtype=np.float64) for a in (self.means_, self.covars_, self.startprob_, np.ravel(self.transmat_))]
        rc = _ffi.lib().frisk_hmm_viterbi(p(x), p(seg_off), int(seg_off.size - 1), p(pars[0]), p(pars[1]), p(pars[2]), p(pars[3]), p(out))
        if rc != _ffi.OK:
            raise ValueError("frisk_hmm_viterbi failed (code %d)" % rc)
        return out

    def fit(self, x):
        x = np.asarray(x, dtype=float).ravel()
        if self.native:
            return self._fit_native(x)
        self._init(x)
        prev = -np.inf
        for _ in range(self.n_iter):
            b = self._loglik(x)
            fwd, bwd, ll, lt = self._forward_backward(b)
            post = np.exp(fwd + bwd - ll)
            post /= post.sum(axis=1, keepdims=True)
            if len(x) > 1:
                xi = fwd[:-1, :, None] + lt[None] + (b[1:] + bwd[1:])[:, None, :] - ll
                trans = np.exp(_logsumexp(xi, axis=0))
            else:
                trans = np.zeros((2, 2))
            # M step (hmmlearn's defaults: flat Dirichlet priors, means_weight 0, covars_prior/weight 1e-2 / 1)
            self.startprob_ = post[0] / post[0].sum()
            rows = trans.sum(axis=1, keepdims=True)
            self.transmat_ = np.where(rows > 0, trans / np.where(rows > 0, rows, 1.0), 0.5)
            w = post.sum(axis=0)
            self.means_ = (post * x[:, None]).sum(axis=0) / w
            self.covars_ = (self.covars_prior + (post * (x[:, None] - self.means_[None, :]) ** 2).sum(axis=0)) / w
            self.covars_ = np.maximum(self.covars_, 1e-300)      # (hmmlearn applies min_covar at initialisation only)
            self.loglik_, self.n_iter_ = float(ll), self.n_iter_ + 1
            if ll - prev < self.tol:
                break
            prev = ll
        return self

    def predict(self, x):
        """Viterbi path."""
        x = np.asarray(x, dtype=float).ravel()
        if self.native:
            return self.predict_segments(x, [0, x.size]).astype(int)
        return self._predict_py(x)

    def _predict_py(self, x):
        if x.size == 0:
            return np.zeros(0, dtype=int)
        b = self._loglik(x)
        with np.errstate(divide="ignore"):
            lt, ls = np.log(self.transmat_), np.log(self.startprob_)
        n = x.size
        score = ls + b[0]
        back = np.zeros((n, 2), dtype=int)
        for t in range(1, n):
            cand = score[:, None] + lt
            back[t] = cand.argmax(axis=0)
            score = cand.max(axis=0) + b[t]
        path = np.empty(n, dtype=int)
        path[-1] = int(score.argmax())
        for t in range(n - 1, 0, -1):
            path[t - 1] = back[t, path[t]]
        return path


def findBaseRanges(s, ch, name=None, minlen=0):
    """(first, last) index - or (name, first, last) - of every maximal run of `ch` in `s` whose span last - first is not
    below minlen (L91-104: the test is `<`, so minlen 0 keeps single elements)."""
    runs, start = [], None
    n = len(s)
    for i in range(n + 1):
        hit = i < n and s[i] == ch
        if hit and start is None:
            start = i
        elif not hit and start is not None:
            if (i - 1) - start >= minlen:
                runs.append((name, start, i - 1) if name else (start, i - 1))
            start = None
    return runs


def state_runs(states, value):
    """Runs of one HMM state, single windows included (how hmm2BED L778-779 calls findBaseRanges)."""
    return findBaseRanges(states, value)


def range2interval(rangeList, windows, state):
    """Runs of window indices -> (name, start of the first window, stop of the last, state), all strings (L787-795).
    windows: the scaffold's scored rows (name, start, stop, ...) in table order."""
    for first, last in rangeList:
        yield (str(windows[0][0]), str(int(windows[first][1])), str(int(windows[last][2])), str(state))


def hmm2BED(rows, model=None):
    """rows: (name, start, stop, KLD, ...) in table order.  Fits the model on all non-NaN scores stacked
    (L1541), decodes per scaffold, and returns intervals (name, start, stop, 'State1'|'State2') as STRINGS
    sorted the way the reference sorts them - lexicographically on the string fields (L783)."""
    if hasattr(rows, "kld"):            # a ScoreTable: everything on the columns (3 M rows at GRCh38 scale)
        t = rows
        kld = np.where(t.kld_is_int0 != 0, 0.0, t.kld)
        ok = np.nonzero(~np.isnan(kld))[0]
        if model is None:
            model = GaussianHMM2().fit(kld[ok])
        if ok.size == 0:
            return [], model
        # scaffolds BY NAME in order of first appearance (L762: Counter over the name column; L765: all rows of that name)
        ids, first = {}, []
        for nm in t.names:
            if nm not in ids:
                ids[nm] = len(ids)
                first.append(nm)
        name_id = np.asarray([ids[nm] for nm in t.names], dtype=np.int64)[t.seq_index[ok]]
        seen_order = name_id[np.sort(np.unique(name_id, return_index=True)[1])]      # ids in order of first appearance among the rows
        rank_of = np.empty(len(ids), dtype=np.int64)
        rank_of[seen_order] = np.arange(seen_order.size)
        order = np.argsort(rank_of[name_id], kind="stable")                            # rows grouped by scaffold, table order inside
        rr = ok[order]
        grp = rank_of[name_id][order]
        seg_off = np.concatenate(([0], np.nonzero(np.diff(grp))[0] + 1, [rr.size])).astype(np.int64)
        if hasattr(model, "predict_segments"):
            states = np.asarray(model.predict_segments(kld[rr], seg_off))
        else:                               # any object with hmmlearn's predict(): one call per scaffold, as L769
            xs = kld[rr]
            states = np.concatenate([np.asarray(model.predict(xs[a:b].reshape(-1, 1))).ravel()
                                     for a, b in zip(seg_off[:-1].tolist(), seg_off[1:].tolist())])
        # maximal runs of one state inside one scaffold (findBaseRanges with minlen 0 keeps single windows, L778-779)
        brk = np.ones(rr.size, dtype=bool)
        brk[1:] = (states[1:] != states[:-1]) | (grp[1:] != grp[:-1])
        a = np.nonzero(brk)[0]
        b = np.concatenate((a[1:], [rr.size])) - 1
        run_name = [str(first[int(seen_order[g])]) for g in grp[a].tolist()]
        label = ("State1", "State2")
        out = [(nm, str(s0), str(s1), label[st]) for nm, s0, s1, st in
               zip(run_name, t.start[rr[a]].tolist(), t.stop[rr[b]].tolist(), states[a].tolist())]
        return sorted(out, key=lambda x: (x[0], x[1], x[2])), model
    good = [r for r in rows if not (isinstance(r[3], float) and r[3] != r[3])]
    if model is None:
        model = GaussianHMM2().fit(np.array([float(r[3]) for r in good]))
    names = []
    for r in good:
        if r[0] not in names:
            names.append(r[0])
    out = []
    for name in names:
        win = [r for r in good if r[0] == name]
        states = model.predict(np.array([float(r[3]) for r in win]))
        for value, label in ((0, "State1"), (1, "State2")):
            out.extend(range2interval(state_runs(states.tolist(), value), win, label))
    return sorted(out, key=lambda t: (t[0], t[1], t[2])), model


def hmmBED2GFF(intervals, version=FRISK_VERSION):
    width = len(str(len(intervals)))
    for n, rec in enumerate(intervals, 1):
        if n == 1:
            yield "##gff-version 3\n"
        yield "\t".join([rec[0], "frisk_" + version, str(rec[3]), str(rec[1]), str(rec[2]), ".", "+", ".",
                         "ID=" + rec[3] + "_" + str(n).zfill(width)]) + "\n"

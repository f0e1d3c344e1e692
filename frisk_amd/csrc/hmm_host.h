// hmm_host.h - host-native 2-state Gaussian HMM for the segmentation of the KLD track (BASELINE config 5:
// "hmmlearn segmentation on host").  The reference fits hmmlearn's GaussianHMM(n_components=2, covariance_type="full") on all
// window scores stacked as ONE sequence (frisk/__init__.py L1537-1541) and Viterbi-decodes every scaffold (hmm2BED, L757-785).
// hmmlearn is absent here; frisk_amd/hmm.py restates the documented model (Baum-Welch, hmmlearn's default priors, deterministic
// 2-means start) in numpy with one Python step per window - 20 s per 100 000 windows.  This file is the same model for 3 M
// windows in a fraction of a second:
//   * E step in scaled (not log) space: per window the two emission densities are divided by the larger one (so one of them is
//     exactly 1 and an outlier far from both means cannot underflow both), forward and backward vectors are renormalised at
//     every step, and posteriors / transition posteriors are normalised per window - the same quantities as the log-space
//     recursion, to rounding;
//   * the recursions are products of 2 x 2 matrices, which are associative: the sequence is cut into FIXED pieces (their number
//     does not depend on the machine, so neither do the sums' rounding), every piece's product is formed in parallel, a short
//     serial pass gives the vector at every cut, and the pieces are walked again in parallel from there;
//   * Viterbi stays in log space, one scaffold per task (its arithmetic is the Python loop's, operation for operation).
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <limits>
#include <thread>
#include <vector>

namespace frisk_hmm {

constexpr double LOG2PI = 1.8378770664093454835606594728112;      // log(2 pi)
constexpr int PIECES = 256;                                         // fixed: results do not depend on the host's thread count

struct Model {
    double means[2], covars[2], startprob[2], transmat[4];          // transmat row-major: [i * 2 + j] = P(j | i)
};

template <class F>
inline void parallel_for(int n, F&& fn) {
    unsigned hw = std::thread::hardware_concurrency();
    int T = int(std::min<unsigned>(hw ? hw : 1u, 32u));
    if (T > n) T = n;
    if (T <= 1) { for (int i = 0; i < n; ++i) fn(i); return; }
    std::atomic<int> next{0};
    auto work = [&] { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i); };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
}

// log N(x; mean_j, covar_j) - the expression of GaussianHMM2._loglik, operation for operation
inline double loglik(double x, double mean, double covar, double logcov) {
    const double d = x - mean;
    return -0.5 * ((LOG2PI + logcov) + d * d / covar);
}

// deterministic start (GaussianHMM2._init): 1-D 2-means from the extremes, shared variance, flat start and transitions
inline void init(const double* x, int64_t n, double min_covar, Model& m) {
    double lo = x[0], hi = x[0], s1 = 0.0;
    for (int64_t t = 0; t < n; ++t) { lo = std::min(lo, x[t]); hi = std::max(hi, x[t]); s1 += x[t]; }
    double c[2] = {lo, hi};
    const int P = int(std::min<int64_t>(PIECES, std::max<int64_t>(1, n / 4096)));
    for (int it = 0; it < 100; ++it) {
        std::vector<double> ps(size_t(P) * 4, 0.0);
        parallel_for(P, [&](int p) {
            const int64_t a = n * p / P, b = n * (p + 1) / P;
            double s0 = 0, s1_ = 0, n0 = 0, n1 = 0;
            for (int64_t t = a; t < b; ++t) {
                if (std::fabs(x[t] - c[0]) <= std::fabs(x[t] - c[1])) { s0 += x[t]; n0 += 1; }       // (argmin: ties to state 0)
                else { s1_ += x[t]; n1 += 1; }
            }
            ps[size_t(p) * 4] = s0; ps[size_t(p) * 4 + 1] = n0; ps[size_t(p) * 4 + 2] = s1_; ps[size_t(p) * 4 + 3] = n1;
        });
        double s0 = 0, n0 = 0, sB = 0, n1 = 0;
        for (int p = 0; p < P; ++p) { s0 += ps[size_t(p) * 4]; n0 += ps[size_t(p) * 4 + 1]; sB += ps[size_t(p) * 4 + 2]; n1 += ps[size_t(p) * 4 + 3]; }
        const double nw[2] = {n0 > 0 ? s0 / n0 : c[0], n1 > 0 ? sB / n1 : c[1]};
        // numpy.allclose(new, c): |new - c| <= 1e-8 + 1e-5 |c|; on convergence the centres of the PREVIOUS round are kept
        if (std::fabs(nw[0] - c[0]) <= 1e-8 + 1e-5 * std::fabs(c[0]) && std::fabs(nw[1] - c[1]) <= 1e-8 + 1e-5 * std::fabs(c[1])) break;
        c[0] = nw[0]; c[1] = nw[1];
    }
    m.means[0] = std::min(c[0], c[1]); m.means[1] = std::max(c[0], c[1]);
    const double mu = s1 / double(n);
    double v = 0.0;
    for (int64_t t = 0; t < n; ++t) v += (x[t] - mu) * (x[t] - mu);
    m.covars[0] = m.covars[1] = v / double(n) + min_covar;
    m.startprob[0] = m.startprob[1] = 0.5;
    for (double& a : m.transmat) a = 0.5;
}

struct Fit { Model m; double loglik; int iters; };

// Baum-Welch, GaussianHMM2.fit: n_iter rounds at most, stop when the log-likelihood gains less than tol (the parameters of the
// round that met the test are kept, as there)
inline Fit fit(const double* x, int64_t n, int n_iter, double tol, double min_covar, double covars_prior) {
    Fit F;
    init(x, n, min_covar, F.m);
    Model& m = F.m;
    F.loglik = -std::numeric_limits<double>::infinity();
    F.iters = 0;
    const int P = int(std::min<int64_t>(PIECES, std::max<int64_t>(1, n / 2048)));
    std::vector<double> B(size_t(n) * 2), A(size_t(n) * 2);           // scaled emissions; forward vectors, then posteriors
    const size_t np_ = size_t(P);
    std::vector<double> pm(np_ * 4), edge((np_ + 1) * 2), edgeB((np_ + 1) * 2), pll(np_), acc(np_ * 8);
    double prev = -std::numeric_limits<double>::infinity();
    for (int it = 0; it < n_iter; ++it) {
        const double lc0 = std::log(m.covars[0]), lc1 = std::log(m.covars[1]), ic0 = 1.0 / m.covars[0], ic1 = 1.0 / m.covars[1];
        const double a00 = m.transmat[0], a01 = m.transmat[1], a10 = m.transmat[2], a11 = m.transmat[3];
        // 1. scaled emissions + the product of every piece's forward steps (row convention: alpha_t = (alpha_{t-1} A) o b_t)
        parallel_for(P, [&](int p) {
            const int64_t a = n * p / P, b = n * (p + 1) / P;
            double m00 = 1, m01 = 0, m10 = 0, m11 = 1, ll = 0;
            for (int64_t t = a; t < b; ++t) {
                const double d0 = x[t] - m.means[0], d1 = x[t] - m.means[1];
                const double l0 = -0.5 * ((LOG2PI + lc0) + d0 * d0 * ic0), l1 = -0.5 * ((LOG2PI + lc1) + d1 * d1 * ic1);
                const double mx = std::max(l0, l1);
                const double b0 = std::exp(l0 - mx), b1 = std::exp(l1 - mx);
                B[size_t(t) * 2] = b0; B[size_t(t) * 2 + 1] = b1;
                ll += mx;
                if (t == 0) continue;                               // (the first window's step is the start vector, not a transition)
                const double s00 = a00 * b0, s01 = a01 * b1, s10 = a10 * b0, s11 = a11 * b1;
                const double n00 = m00 * s00 + m01 * s10, n01 = m00 * s01 + m01 * s11;
                const double n10 = m10 * s00 + m11 * s10, n11 = m10 * s01 + m11 * s11;
                const double r = 1.0 / (n00 + n01 + n10 + n11);
                m00 = n00 * r; m01 = n01 * r; m10 = n10 * r; m11 = n11 * r;
            }
            pm[size_t(p) * 4] = m00; pm[size_t(p) * 4 + 1] = m01; pm[size_t(p) * 4 + 2] = m10; pm[size_t(p) * 4 + 3] = m11;
            pll[size_t(p)] = ll;
        });
        // 2. forward vector in front of every piece (normalised), serial over the pieces
        {
            double v0 = m.startprob[0] * B[0], v1 = m.startprob[1] * B[1];
            edge[0] = v0; edge[1] = v1;                             // piece 0 starts from the UNnormalised first vector
            double s = v0 + v1;
            v0 /= s; v1 /= s;
            for (int p = 0; p < P; ++p) {
                const double* M = &pm[size_t(p) * 4];
                const double w0 = v0 * M[0] + v1 * M[2], w1 = v0 * M[1] + v1 * M[3];
                s = w0 + w1;
                v0 = w0 / s; v1 = w1 / s;
                if (p + 1 < P) { edge[size_t(p + 1) * 2] = v0; edge[size_t(p + 1) * 2 + 1] = v1; }
            }
        }
        // 3. forward vectors of every window, and the log-likelihood
        parallel_for(P, [&](int p) {
            const int64_t a = n * p / P, b = n * (p + 1) / P;
            double v0, v1, ll = 0;
            int64_t t = a;
            if (p == 0) {
                const double s = edge[0] + edge[1];
                ll += std::log(s);
                v0 = edge[0] / s; v1 = edge[1] / s;
                A[0] = v0; A[1] = v1;
                t = 1;
            } else { v0 = edge[size_t(p) * 2]; v1 = edge[size_t(p) * 2 + 1]; }
            double prod = 1.0;                                      // the scales, four to a logarithm (each is >= the smallest transition
            int held = 0;                                           // probability: one of the two scaled emissions is exactly 1)
            for (; t < b; ++t) {
                const double w0 = (v0 * a00 + v1 * a10) * B[size_t(t) * 2], w1 = (v0 * a01 + v1 * a11) * B[size_t(t) * 2 + 1];
                const double s = w0 + w1, r = 1.0 / s;
                prod *= s;
                if (++held == 4 || prod < 1e-200) { ll += std::log(prod); prod = 1.0; held = 0; }
                v0 = w0 * r; v1 = w1 * r;
                A[size_t(t) * 2] = v0; A[size_t(t) * 2 + 1] = v1;
            }
            ll += std::log(prod);
            pll[size_t(p)] += ll;
        });
        double ll = 0;
        for (int p = 0; p < P; ++p) ll += pll[size_t(p)];
        // 4. the product of every piece's backward steps (column convention: beta_t = A (b_{t+1} o beta_{t+1}))
        parallel_for(P, [&](int p) {
            const int64_t a = n * p / P, b = n * (p + 1) / P;
            double m00 = 1, m01 = 0, m10 = 0, m11 = 1;
            for (int64_t t = b - 1; t >= a; --t) {                  // beta_{t-1} from beta_t: the step uses window t's emissions
                if (t == 0) break;
                const double b0 = B[size_t(t) * 2], b1 = B[size_t(t) * 2 + 1];
                const double s00 = a00 * b0, s01 = a01 * b1, s10 = a10 * b0, s11 = a11 * b1;      // S = A diag(b_t)
                const double n00 = s00 * m00 + s01 * m10, n01 = s00 * m01 + s01 * m11;            // S M
                const double n10 = s10 * m00 + s11 * m10, n11 = s10 * m01 + s11 * m11;
                const double r = 1.0 / (n00 + n01 + n10 + n11);
                m00 = n00 * r; m01 = n01 * r; m10 = n10 * r; m11 = n11 * r;
            }
            pm[size_t(p) * 4] = m00; pm[size_t(p) * 4 + 1] = m01; pm[size_t(p) * 4 + 2] = m10; pm[size_t(p) * 4 + 3] = m11;
        });
        // 5. backward vector of every piece's LAST window, serial from the end: E[P] = beta of the sequence's last window (flat),
        //    E[p] = M_p E[p + 1] (M_p = the steps of all of piece p's windows: it carries beta from p's last window to p-1's last)
        {
            double v0 = 0.5, v1 = 0.5;
            edgeB[size_t(P) * 2] = v0; edgeB[size_t(P) * 2 + 1] = v1;
            for (int p = P - 1; p >= 1; --p) {
                const double* M = &pm[size_t(p) * 4];
                const double w0 = M[0] * v0 + M[1] * v1, w1 = M[2] * v0 + M[3] * v1;
                const double s = w0 + w1;
                v0 = w0 / s; v1 = w1 / s;
                edgeB[size_t(p) * 2] = v0; edgeB[size_t(p) * 2 + 1] = v1;
            }
        }
        // 6. walk every piece backwards: posteriors (into A), transition posteriors, and the M step's sums
        parallel_for(P, [&](int p) {
            const int64_t a = n * p / P, b = n * (p + 1) / P;
            double be0 = edgeB[size_t(p + 1) * 2], be1 = edgeB[size_t(p + 1) * 2 + 1];        // beta of the piece's last window
            double g0s = 0, g1s = 0, gx0 = 0, gx1 = 0, x00 = 0, x01 = 0, x10 = 0, x11 = 0;
            for (int64_t t = b - 1; t >= a; --t) {
                const double al0 = A[size_t(t) * 2], al1 = A[size_t(t) * 2 + 1];
                // posterior of window t
                double g0 = al0 * be0, g1 = al1 * be1;
                const double gr = 1.0 / (g0 + g1);
                g0 *= gr; g1 *= gr;
                A[size_t(t) * 2] = g0; A[size_t(t) * 2 + 1] = g1;
                g0s += g0; g1s += g1; gx0 += g0 * x[t]; gx1 += g1 * x[t];
                if (t == 0) break;
                // transition posterior between windows t-1 and t, and beta of window t-1
                const double b0 = B[size_t(t) * 2] * be0, b1 = B[size_t(t) * 2 + 1] * be1;
                // forward vector of window t-1: still in A inside the piece; the last window of the piece before belongs to another
                // task's walk (which turns it into a posterior) - its forward vector is the edge this piece started from
                const double p0 = t > a ? A[size_t(t - 1) * 2] : edge[size_t(p) * 2], p1 = t > a ? A[size_t(t - 1) * 2 + 1] : edge[size_t(p) * 2 + 1];
                const double e00 = p0 * a00 * b0, e01 = p0 * a01 * b1, e10 = p1 * a10 * b0, e11 = p1 * a11 * b1;
                const double er = 1.0 / (e00 + e01 + e10 + e11);
                x00 += e00 * er; x01 += e01 * er; x10 += e10 * er; x11 += e11 * er;
                const double nb0 = a00 * b0 + a01 * b1, nb1 = a10 * b0 + a11 * b1;
                const double br = 1.0 / (nb0 + nb1);
                be0 = nb0 * br; be1 = nb1 * br;
            }
            double* q = &acc[size_t(p) * 8];
            q[0] = g0s; q[1] = g1s; q[2] = gx0; q[3] = gx1; q[4] = x00; q[5] = x01; q[6] = x10; q[7] = x11;
        });
        double S[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int p = 0; p < P; ++p) for (int k = 0; k < 8; ++k) S[k] += acc[size_t(p) * 8 + k];
        // M step (hmmlearn's defaults: flat Dirichlet priors, means_weight 0, covars_prior / weight 1e-2 / 1)
        m.startprob[0] = A[0] / (A[0] + A[1]); m.startprob[1] = A[1] / (A[0] + A[1]);
        if (n > 1) {
            const double r0 = S[4] + S[5], r1 = S[6] + S[7];
            m.transmat[0] = r0 > 0 ? S[4] / r0 : 0.5; m.transmat[1] = r0 > 0 ? S[5] / r0 : 0.5;
            m.transmat[2] = r1 > 0 ? S[6] / r1 : 0.5; m.transmat[3] = r1 > 0 ? S[7] / r1 : 0.5;
        } else { for (double& v : m.transmat) v = 0.5; }
        m.means[0] = S[2] / S[0]; m.means[1] = S[3] / S[1];
        parallel_for(P, [&](int p) {
            const int64_t a = n * p / P, b = n * (p + 1) / P;
            double c0 = 0, c1 = 0;
            for (int64_t t = a; t < b; ++t) {
                const double d0 = x[t] - m.means[0], d1 = x[t] - m.means[1];
                c0 += A[size_t(t) * 2] * (d0 * d0); c1 += A[size_t(t) * 2 + 1] * (d1 * d1);
            }
            acc[size_t(p) * 8] = c0; acc[size_t(p) * 8 + 1] = c1;
        });
        double c0 = 0, c1 = 0;
        for (int p = 0; p < P; ++p) { c0 += acc[size_t(p) * 8]; c1 += acc[size_t(p) * 8 + 1]; }
        m.covars[0] = std::max((covars_prior + c0) / S[0], 1e-300);
        m.covars[1] = std::max((covars_prior + c1) / S[1], 1e-300);
        F.loglik = ll;
        F.iters = it + 1;
        if (ll - prev < tol) break;
        prev = ll;
    }
    return F;
}

// Viterbi path of one sequence (GaussianHMM2.predict, operation for operation; ties to the lower state, as numpy's argmax)
inline void viterbi(const double* x, int64_t n, const Model& m, int8_t* path, std::vector<uint8_t>& back) {
    if (n <= 0) return;
    const double lc0 = std::log(m.covars[0]), lc1 = std::log(m.covars[1]);
    const double t00 = std::log(m.transmat[0]), t01 = std::log(m.transmat[1]), t10 = std::log(m.transmat[2]), t11 = std::log(m.transmat[3]);
    back.assign(size_t(n), 0);
    double s0 = std::log(m.startprob[0]) + loglik(x[0], m.means[0], m.covars[0], lc0);
    double s1 = std::log(m.startprob[1]) + loglik(x[0], m.means[1], m.covars[1], lc1);
    for (int64_t t = 1; t < n; ++t) {
        const double c00 = s0 + t00, c10 = s1 + t10, c01 = s0 + t01, c11 = s1 + t11;
        const uint8_t k0 = c10 > c00 ? 1 : 0, k1 = c11 > c01 ? 1 : 0;
        back[size_t(t)] = uint8_t(k0 | (k1 << 1));
        s0 = (k0 ? c10 : c00) + loglik(x[t], m.means[0], m.covars[0], lc0);
        s1 = (k1 ? c11 : c01) + loglik(x[t], m.means[1], m.covars[1], lc1);
    }
    int8_t cur = s1 > s0 ? 1 : 0;
    path[n - 1] = cur;
    for (int64_t t = n - 1; t > 0; --t) {
        cur = int8_t((back[size_t(t)] >> cur) & 1);
        path[t - 1] = cur;
    }
}

// sequences [off[s], off[s + 1]) of x, one task each
inline void viterbi_segments(const double* x, const int64_t* off, int32_t n_seg, const Model& m, int8_t* path) {
    parallel_for(n_seg, [&](int s) {
        std::vector<uint8_t> back;
        viterbi(x + off[s], off[s + 1] - off[s], m, path + off[s], back);
    });
}

}  // namespace frisk_hmm

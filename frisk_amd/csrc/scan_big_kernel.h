// frisk_amd - windows longer than 65 535 bases (any length up to 2^31-1), and every window when kmax > 8 (the reference's -k
// is unbounded, L1197-1206; this path serves orders up to FRISK_MAX_K = 12): the histogram of a window does not fit
// 16-bit LDS counters any more, so a workgroup keeps 32-bit tables of ALL orders in its own slice of a global
// scratch buffer (L2-resident: 350 KB per workgroup at k = 1..8).  Same algorithm as scan_kernel.h -
//   one update per position at the order of its longest valid word, marginalisation C_x = D_x + sum children,
//   closed-form IVOM, one-pass KLD, exact order-independent sums -
// but scoring walks the order-K table instead of the positions (every non-empty bin is a present max-mer), so there
// is no election and no orphan list.  A slow path by design: such windows are 13+ times the default length and a
// scaffold has few of them.  Reference: crawlGenome L194-251, computeKmers L280-367, IvomBuild L369-457, KLD L459-472,
// calcGC L120-137, calcRIP L474-495 of /root/reference/frisk/__init__.py.
#pragma once
#include "scan_kernel.h"

#define FRISK_BIG_NT 1024

template <bool DEBUG>
__global__ __launch_bounds__(FRISK_BIG_NT) void scan_big_kernel(const ScanParams P, uint32_t* __restrict__ big,
                                                                 int64_t big_stride) {
    constexpr int NT = FRISK_BIG_NT, NW = NT / 64;
    __shared__ uint32_t tl[8];
    __shared__ double red[NW * 6];
    const int tid = threadIdx.x, lane = tid & 63;
    const int kmin = P.kmin, kmax = P.kmax;
    uint32_t* cnt = big + int64_t(blockIdx.x) * big_stride;                 // all-zero between windows
    const int64_t offK = table_offset(kmin, kmax);
    const uint32_t nK = 1u << (2 * kmax);
    ScafDesc d;
    d.cand0 = 0; d.ncand = 0; d.off = 0; d.size = 0; d.kind = 0; d.base0 = 0; d.j0 = 0;
    int dsi = -1;
    for (int64_t cand = P.c0 + blockIdx.x; cand < P.c1; cand += gridDim.x) {
        // ---- which scaffold / window is this candidate? (as in scan_kernel)
        if (cand < d.cand0 || cand >= d.cand0 + d.ncand) {
            int lo = 0, hi = P.n_desc - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (P.descs[mid].cand0 <= cand) lo = mid; else hi = mid - 1;
            }
            d = P.descs[lo];
            dsi = lo;
        }
        const int64_t j = cand - d.cand0 + d.j0;           // window index inside the scaffold
        int64_t st, rep_start, rep_stop, n;
        bool jump = false;
        if (d.kind == 1) { st = 0; n = d.size; rep_start = 1; rep_stop = d.size; }              // L219
        else {
            st = j * P.inc;
            n = P.w;
            rep_start = st + 1; rep_stop = st + P.w;                                            // L245
            if (st + P.w > d.size) {                                                            // L230-232
                jump = true;
                st = d.size - P.w;
                rep_start = st; rep_stop = d.size;                                              // L243: 0-based start
                if (st < 0) { st += d.size; if (st < 0) st = 0; }                               // Python slice semantics
                n = d.size - st;
            }
        }
        const int64_t g0 = d.off + (st - d.base0);            // resident position of the window's first base
        const int64_t row = cand - P.c0;
        if (tid < 8) tl[tid] = 0;
        __syncthreads();

        // ---- stage 1: one global-atomic update per position; uppercase composition by wave ballots
        uint32_t cA = 0, cT = 0, cG = 0, cC = 0, ntop = 0;
        for (int64_t base = 0; base < n; base += NT) {
            const int64_t jj = base + tid;
            const bool act = jj < n;
            const int64_t g = g0 + (act ? jj : 0);
            const uint32_t c24 = fetch_codes24(P.codes, g);                 // 12 bases: orders up to FRISK_MAX_K
            const uint32_t inv16 = fetch_mask16(P.inv, g);
            const uint32_t low1 = fetch_mask1(P.low, g);
            int run = lead_clear16(inv16);                                  // window words are upper-cased: L334-335
            const int64_t rem = n - jj;
            run = run < rem ? run : int(rem);
            run = run < kmax ? run : kmax;
            if (act && run >= kmin) atomicAdd(&cnt[table_offset(kmin, run) + (c24 >> (24 - 2 * run))], 1u);
            const bool up = act && !((inv16 >> 15) | low1);
            const uint32_t c2 = c24 >> 22;
            cA += __popcll(__ballot(up && c2 == 0));
            cT += __popcll(__ballot(up && c2 == 1));
            cG += __popcll(__ballot(up && c2 == 2));
            cC += __popcll(__ballot(up && c2 == 3));
            ntop += __popcll(__ballot(act && run == kmax));
        }
        if (lane == 0) {
            if (cA) atomicAdd(&tl[0], cA);
            if (cT) atomicAdd(&tl[1], cT);
            if (cG) atomicAdd(&tl[2], cG);
            if (cC) atomicAdd(&tl[3], cC);
            if (ntop) atomicAdd(&tl[4], ntop);
        }
        // the updates were made by atomics at L2; the tables are read with ordinary loads from here on: drop whatever this
        // CU's vector L1 still holds of them (agent-scope acquire = L1 invalidate), on both sides of the barrier
        __threadfence();
        __syncthreads();
        const int64_t upA = tl[0], upT = tl[1], upG = tl[2], upC = tl[3];
        const uint32_t nvalid_top = tl[4];
        const int64_t S = upA + upT + upG + upC;                            // windowSpace (L380)
        const int64_t nn = n - S;
        const bool keep = !(double(nn) >= 0.3 * double(n));                 // L237-241 / L213
        uint32_t status = (jump ? ROW_JUMPBACK : 0u);
        if (tid == 0) {
            P.seq_index[row] = dsi;
            P.start[row] = rep_start;
            P.stop[row] = rep_stop;
        }
        auto zero_tables = [&]() {          // ordinary stores, made visible at L2 before the next window's atomics
            for (int64_t i = tid; i < big_stride / 4; i += NT) reinterpret_cast<uint4*>(cnt)[i] = make_uint4(0, 0, 0, 0);
            __threadfence();
        };
        if (!keep) {
            zero_tables();
            if (tid == 0) {
                P.status[row] = status;
                const double qnan = __longlong_as_double(0x7FF8000000000000LL);
                P.kld[row] = qnan; P.gc[row] = qnan;
                if (P.flags & 1u) { P.pi[row] = qnan; P.si[row] = qnan; P.cri[row] = qnan; }
            }
            __syncthreads();
            continue;
        }

        // ---- stage 2: C_x[q] = D_x[q] + sum_b C_{x+1}[4q+b], level by level (the barrier orders the global accesses
        // of one workgroup)
        for (int x = kmax - 1; x >= kmin; --x) {
            const int64_t ox = table_offset(kmin, x), ou = table_offset(kmin, x + 1);
            for (uint32_t c = tid; c < (1u << (2 * x)); c += NT) {
                const uint4 ch = *reinterpret_cast<const uint4*>(cnt + ou + 4 * c);
                cnt[ox + c] += ch.x + ch.y + ch.z + ch.w;
            }
            __syncthreads();
        }
        if (DEBUG && P.dbg_counts) {
            uint32_t* out = P.dbg_counts + row * int64_t(P.nprof);
            for (int64_t i = tid; i < P.nprof; i += NT) out[i] = cnt[i];
        }
        if (DEBUG && P.dbg_meta && tid == 0) {
            P.dbg_meta[row * 3 + 0] = n;
            P.dbg_meta[row * 3 + 1] = (n >= kmax ? n - kmax + 1 : 0) - int64_t(nvalid_top);
            P.dbg_meta[row * 3 + 2] = nn;
        }
        if (nvalid_top == 0) status |= ROW_NO_MAXMER;
        if (nvalid_top > 0 && S >= kmin - 1 && S <= kmax - 1) status |= ROW_ZERO_WEIGHT;       // L401-409
        status |= ROW_KEPT;
        if (tid == 0) {
            P.gc[row] = double(upG + upC) / double(S);                      // L136
            if (P.flags & 1u) {                                             // calcRIP L474-495
                const double qnan = __longlong_as_double(0x7FF8000000000000LL);
                const uint32_t* di = cnt + table_offset(kmin, 2);
                const uint32_t AT = di[1], TA = di[4], TG = di[6], GT = di[9], CA = di[12], AC = di[3];
                const double pi = AT > 0 ? double(TA) / double(AT) : qnan;
                const double si = (AC + GT) > 0 ? double(CA + TG) / double(AC + GT) : qnan;
                P.pi[row] = pi;
                P.si[row] = si;
                P.cri[row] = (pi == 0.0 || si == 0.0) ? qnan : pi - si;
            }
        }

        // ---- stages 3 + 4: window constants, then every non-empty bin of the order-K table is a present max-mer
        double r[FRISK_MAX_K + 1];
        for (int x = 0; x <= FRISK_MAX_K; ++x) r[x] = double(1u << (2 * x)) / double((S - (x - 1)) * 2);
        ExactSum accw = exact_begin(), accg = exact_begin(), acct = exact_begin();
        for (uint32_t code = tid; code < nK; code += NT) {
            if (cnt[offK + code] == 0) continue;
            unsigned long long W = 0;
            double A = 0.0;
#pragma unroll
            for (int x = 1; x <= FRISK_MAX_K; ++x) {
                if (x < kmin || x > kmax) continue;
                const uint32_t cx = cnt[table_offset(kmin, x) + (code >> (2 * (kmax - x)))];
                const double cd = double(cx);
                W += (unsigned long long)cx << (2 * x);                     // count * 4**x (L399-408)
                A = __builtin_fma(cd * cd, r[x], A);                        // w_x * p_x = c^2 4^x / D_x
            }
            const double Ig = P.ig[code];
            const double ratio = div_exact(A, double(W) * Ig);
            const double Iw = ratio * Ig;
            exact_add(accw, Iw);
            exact_add(accg, Ig);
            exact_add(acct, Iw * log_pos(ratio));
        }
        exact_end(accw); exact_end(accg); exact_end(acct);
        block_sum3<NW>(accw, accg, acct, red, tid);
        const double Sw = exact_value(accw), Sg = exact_value(accg), Tt = exact_value(acct);
        const double LN2 = 0.69314718055994530942;
        const double acc = (nvalid_top == 0) ? 0.0 : ((Tt / Sw - log(Sw)) + log(Sg)) / LN2;    // L453-454, L465-470
        zero_tables();
        if (tid == 0) {
            if (nvalid_top > 0 && Sg != Sg) status |= ROW_ZERO_WEIGHT;      // a max-mer without genome weight (L437)
            P.status[row] = status;
            P.kld[row] = acc;
        }
        __syncthreads();
    }
}

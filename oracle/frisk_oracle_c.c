/* CPU oracle for the frisk hot path, plain C + OpenMP  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Third statement of the same algorithm as oracle/frisk_oracle.py (reference-shaped Python) and
 * oracle/frisk_oracle_np.py (numpy).  It exists so that parity can be checked row by row at
 * BASELINE.json's full single-GPU sizes (10^5..10^6 windows), which the Python oracles cannot reach,
 * and so that bench.py can quote a compiled multi-core CPU line next to the interpreted one.
 * Pinned by tests/test_oracle_c.py: against the golden vectors of the reference's own functions
 * (tests/golden/, tools/make_golden.py) and against frisk_oracle_np.py on random inputs.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library built
 * from this file; the product (frisk_amd/, libfrisk_hip.so) never does.
 *
 * Citations: /root/reference/frisk/__init__.py (L<n>).
 * Build: make -C oracle   ->  oracle/_build/libfrisk_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define FO_ROW_ZERO_DIV 2u   /* the reference raises ZeroDivisionError for this window (L437 / L132) */
#define FO_ROW_NO_MAXMER 8u  /* no max-mer in the window: KLD is the integer 0 of an empty sum (L478) */

static unsigned char lut_code[256];  /* A=0 T=1 G=2 C=3: digit order of the canonical index, L70 */
static unsigned char lut_valid[256]; /* ACGT in either case: word.upper() in map, L334-341 */
static unsigned char lut_upper[256]; /* uppercase ACGT only: countN / calcGC, L106-137 */
static int lut_ready;

static void lut_init(void) {
    if (lut_ready) return;
    const char* up = "ATGC";
    const char* lo = "atgc";
    for (int d = 0; d < 4; ++d) {
        lut_code[(unsigned char)up[d]] = lut_code[(unsigned char)lo[d]] = (unsigned char)d;
        lut_valid[(unsigned char)up[d]] = lut_valid[(unsigned char)lo[d]] = 1;
        lut_upper[(unsigned char)up[d]] = 1;
    }
    lut_ready = 1;
}

static int64_t tab_off(int kmin, int x) { return ((1ll << (2 * x)) - (1ll << (2 * kmin))) / 3; }

int64_t fo_profile_len(int kmin, int kmax) { return tab_off(kmin, kmax + 1); }

int fo_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static int64_t revcomp_idx(int64_t c, int x) {
    int64_t r = 0;
    for (int p = 0; p < x; ++p) {
        r = (r << 2) | ((c & 3) ^ 1); /* A<->T, G<->C */
        c >>= 2;
    }
    return r;
}

/* forward counts of one sequence into cnt[profile_len] (every order, one-base step, words with an
 * unacceptable letter skipped, L327-348); returns the number of skipped max-mer positions (exMax). */
static int64_t count_forward(const unsigned char* s, int64_t n, int kmin, int kmax, int upper_only, int64_t* cnt) {
    int64_t run = 0, ex = 0;
    uint64_t code = 0;
    const uint64_t mask = (kmax == 32) ? ~0ull : ((1ull << (2 * kmax)) - 1);
    for (int64_t p = 0; p < n; ++p) {
        unsigned char ch = s[p];
        int ok = upper_only ? lut_upper[ch] : lut_valid[ch];
        if (ok) {
            code = ((code << 2) | lut_code[ch]) & mask;
            ++run;
        } else {
            run = 0;
            code = 0;
        }
        /* words ENDING at p: order x is countable iff run >= x */
        for (int x = kmin; x <= kmax; ++x) {
            if (p + 1 < x) break;
            if (run >= x)
                ++cnt[tab_off(kmin, x) + (int64_t)(code & ((1ull << (2 * x)) - 1))];
            else if (x == kmax)
                ++ex;
        }
    }
    return ex;
}

/* computeKmers(genomeMode=True), L280-367: symmetric counts + {totalLen, exMax, nnTotal}. */
int fo_genome_profile(const char* const* seqs, const int64_t* lens, int64_t nseq, int kmin, int kmax, int mask_host,
                      int64_t* sym, int64_t* meta) {
    lut_init();
    if (kmin < 1 || kmax < kmin || kmax > 12) return 1;
    const int64_t L = fo_profile_len(kmin, kmax);
    int64_t* fwd = (int64_t*)calloc((size_t)L, sizeof(int64_t));
    if (!fwd) return 2;
    int64_t total = 0, ex = 0, nn = 0;
#pragma omp parallel reduction(+ : total, ex, nn)
    {
        int64_t* mine = (int64_t*)calloc((size_t)L, sizeof(int64_t)); /* counts are integers: any order of addition */
#pragma omp for schedule(dynamic, 1)
        for (int64_t i = 0; i < nseq; ++i) {
            const unsigned char* s = (const unsigned char*)seqs[i];
            ex += count_forward(s, lens[i], kmin, kmax, mask_host, mine); /* --maskHost: no .upper(), L336-337 */
            total += lens[i];
            for (int64_t p = 0; p < lens[i]; ++p) nn += !lut_upper[s[p]];
        }
#pragma omp critical
        for (int64_t c = 0; c < L; ++c) fwd[c] += mine[c];
        free(mine);
    }
    for (int x = kmin; x <= kmax; ++x) {
        const int64_t o = tab_off(kmin, x), m = 1ll << (2 * x);
        for (int64_t c = 0; c < m; ++c) sym[o + c] = fwd[o + c] + fwd[o + revcomp_idx(c, x)]; /* L350-351 */
    }
    meta[0] = total;
    meta[1] = ex;
    meta[2] = nn;
    free(fwd);
    return 0;
}

/* un-normalised genome-side IVOM of every max-mer (L411-450); NaN where the reference divides by 0 */
int fo_genome_ivom(const int64_t* sym, const int64_t* meta, int kmin, int kmax, double* ig) {
    const int64_t space = meta[0] - meta[2];
    const int64_t m = 1ll << (2 * kmax);
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < m; ++k) {
        int64_t W = 0;
        double I = 0.0;
        int bad = 0;
        for (int x = kmin; x <= kmax; ++x) {
            const int64_t c = sym[tab_off(kmin, x) + (k >> (2 * (kmax - x)))];
            const int64_t wt = c << (2 * x);
            W += wt;
            const int64_t D = (space - (x - 1)) * 2;
            if (W == 0 || D == 0) {
                bad = 1;
                break;
            }
            const double p = (double)c / (double)D;
            const double a = (double)wt / (double)W;
            I = (x == kmin) ? a * p : a * p + ((1.0 - a) * I);
        }
        ig[k] = bad ? NAN : I;
    }
    return 0;
}

typedef struct {
    int32_t* cnt;     /* dense tables of one window, all orders */
    int32_t* present; /* distinct max-mers of the window */
    double* iw;
} Scratch;

static int cmp_i32(const void* a, const void* b) {
    int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
    return (x > y) - (x < y);
}

/* one iteration of the loop L1478-1494 on s[0..n) */
static void score_window(const unsigned char* s, int64_t n, const double* ig, int kmin, int kmax, int rip, Scratch* sc,
                         uint32_t* status, double* kld, double* gc, double* pi, double* si, double* cri,
                         int32_t* dbg_counts, int64_t* dbg_meta) {
    int32_t* cnt = sc->cnt;
    int64_t off[16];
    uint64_t msk[16];
    for (int x = kmin; x <= kmax; ++x) off[x] = tab_off(kmin, x), msk[x] = (1ull << (2 * x)) - 1;
    int64_t run = 0, S = 0, GC = 0, np_ = 0;
    uint64_t code = 0;
    const uint64_t mask = (1ull << (2 * kmax)) - 1;
    for (int64_t p = 0; p < n; ++p) {
        unsigned char ch = s[p];
        if (lut_upper[ch]) {
            ++S;
            GC += lut_code[ch] >= 2;
        }
        if (lut_valid[ch]) { /* window mode upper-cases: soft-masked bases count, L334-335 */
            code = ((code << 2) | lut_code[ch]) & mask;
            ++run;
        } else {
            run = 0;
            code = 0;
        }
        for (int x = kmin; x <= kmax && x <= run; ++x) {
            int32_t* slot = &cnt[off[x] + (int64_t)(code & msk[x])];
            if (x == kmax && *slot == 0) sc->present[np_++] = (int32_t)code;
            ++*slot;
        }
    }
    uint32_t st = 0;
    double K = 0.0;
    if (np_ == 0) {
        st |= FO_ROW_NO_MAXMER;
    } else {
        qsort(sc->present, (size_t)np_, sizeof(int32_t), cmp_i32);
        int zero_div = 0;
        double sw = 0.0, sg = 0.0;
        for (int64_t j = 0; j < np_ && !zero_div; ++j) {
            const int64_t k = sc->present[j];
            if (isnan(ig[k])) {
                zero_div = 1;
                break;
            }
            int64_t W = 0;
            double I = 0.0;
            for (int x = kmin; x <= kmax; ++x) {
                const int64_t c = cnt[off[x] + (k >> (2 * (kmax - x)))];
                const int64_t wt = c << (2 * x);
                W += wt;
                const int64_t D = (S - (x - 1)) * 2;
                if (D == 0) {
                    zero_div = 1;
                    break;
                }
                const double p = (double)c / (double)D;
                const double a = (double)wt / (double)W;
                I = (x == kmin) ? a * p : a * p + ((1.0 - a) * I);
            }
            sc->iw[j] = I;
            sw += I;
            sg += ig[k];
        }
        if (zero_div) {
            st |= FO_ROW_ZERO_DIV;
        } else {
            const double ln2 = log(2.0);
            for (int64_t j = 0; j < np_; ++j) { /* KLD, L452-495 */
                const double pw = sc->iw[j] / sw, pg = ig[sc->present[j]] / sg;
                if (pg != 0.0) K += pw * (log(pw / pg) / ln2);
            }
        }
    }
    *kld = K;
    if (S)
        *gc = (double)GC / (double)S;
    else {
        *gc = NAN;
        st |= FO_ROW_ZERO_DIV; /* calcGC divides by zero, L132 */
    }
    if (rip) {
        double PI = NAN, SI = NAN, CRI = NAN;
        if (kmin <= 2 && kmax >= 2) { /* calcRIP, L369-409; digits A0 T1 G2 C3 */
            const int32_t* di = cnt + tab_off(kmin, 2);
            const int64_t AT = di[1], TA = di[4], TG = di[6], GT = di[9], CA = di[12], AC = di[3];
            if (AT > 0) PI = (double)TA / (double)AT;
            if (AC + GT > 0) SI = (double)(CA + TG) / (double)(AC + GT);
            if (!isnan(PI) && !isnan(SI) && PI != 0.0 && SI != 0.0) CRI = PI - SI;
        }
        *pi = PI;
        *si = SI;
        *cri = CRI;
    }
    *status = st;
    if (dbg_counts) memcpy(dbg_counts, cnt, (size_t)fo_profile_len(kmin, kmax) * sizeof(int32_t));
    if (dbg_meta) { /* the window's metadata dict: totalLen, exMax, nnTotal (L356-359) */
        int64_t top = 0;
        for (int64_t j = 0; j < np_; ++j) top += cnt[tab_off(kmin, kmax) + sc->present[j]];
        dbg_meta[0] = n;
        dbg_meta[1] = (n - kmax + 1 > 0 ? n - kmax + 1 : 0) - top;
        dbg_meta[2] = n - S;
    }
    /* clear the touched counters by replaying the window */
    run = 0;
    code = 0;
    for (int64_t p = 0; p < n; ++p) {
        unsigned char ch = s[p];
        if (lut_valid[ch]) {
            code = ((code << 2) | lut_code[ch]) & mask;
            ++run;
        } else {
            run = 0;
            code = 0;
        }
        for (int x = kmin; x <= kmax && x <= run; ++x) cnt[off[x] + (int64_t)(code & msk[x])] = 0;
    }
}

/* candidate windows of one scaffold before the N filter (crawlGenome, L194-251) */
static int64_t n_candidates(int64_t size, int64_t w, int64_t i, int scaffolds_all) {
    if ((double)size <= (double)w + (((double)w * 0.75) - (double)i)) return scaffolds_all ? 1 : 0;
    return size >= i ? size / i : 0; /* len(range(0, size - i + 1, i)) */
}

/* Phase B.  Fills one row per KEPT window, in the reference's order; returns the number of rows, or
 * -(needed) if cap is too small, or INT64_MIN on a bad argument.  cand_begin/cand_end restrict the scan
 * to a slice of the global candidate list (end < 0: all), for bounded timing samples. */
int64_t fo_scan(const char* const* seqs, const int64_t* lens, int64_t nseq, const double* ig, int kmin, int kmax,
                int64_t w, int64_t inc, int scaffolds_all, int rip, int64_t cand_begin, int64_t cand_end, int64_t cap,
                int32_t* seq_index, int64_t* start, int64_t* stop, uint32_t* status, double* kld, double* gc,
                double* pi, double* si, double* cri, int32_t* dbg_counts, int64_t* dbg_meta) {
    lut_init();
    if (kmin < 1 || kmax < kmin || kmax > 12 || w < 1 || inc < 1) return INT64_MIN;
    int64_t* first = (int64_t*)malloc((size_t)(nseq + 1) * sizeof(int64_t));
    first[0] = 0;
    for (int64_t q = 0; q < nseq; ++q) first[q + 1] = first[q] + n_candidates(lens[q], w, inc, scaffolds_all);
    const int64_t ncand_all = first[nseq];
    if (cand_end < 0 || cand_end > ncand_all) cand_end = ncand_all;
    if (cand_begin < 0) cand_begin = 0;
    const int64_t ncand = cand_end > cand_begin ? cand_end - cand_begin : 0;
    unsigned char* keep = (unsigned char*)calloc((size_t)ncand + 1, 1);
    int32_t* c_seq = (int32_t*)malloc((size_t)(ncand + 1) * sizeof(int32_t));
    int64_t* c_a = (int64_t*)malloc((size_t)(ncand + 1) * sizeof(int64_t));
    int64_t* c_b = (int64_t*)malloc((size_t)(ncand + 1) * sizeof(int64_t));
    int64_t* c_s = (int64_t*)malloc((size_t)(ncand + 1) * sizeof(int64_t));
    int64_t* c_e = (int64_t*)malloc((size_t)(ncand + 1) * sizeof(int64_t));
    /* enumerate + N filter (>= 30 % not uppercase ACGT drops the window, L213 / L238) */
    int64_t q = 0;
    for (int64_t g = cand_begin; g < cand_end; ++g) {
        while (first[q + 1] <= g) ++q;
        const int64_t size = lens[q], r = g - first[q], j = r * inc, t = g - cand_begin;
        int64_t a, b, s1, e1;
        if ((double)size <= (double)w + (((double)w * 0.75) - (double)inc)) {
            a = 0, b = size, s1 = 1, e1 = size;
        } else if (j + w > size) {
            a = size - w;
            if (a < 0) a = (a + size) > 0 ? a + size : 0; /* negative start: Python slice semantics */
            b = size, s1 = size - w, e1 = size;
        } else {
            a = j, b = j + w, s1 = j + 1, e1 = j + w;
        }
        c_seq[t] = (int32_t)q, c_a[t] = a, c_b[t] = b, c_s[t] = s1, c_e[t] = e1;
    }
    const int64_t L = fo_profile_len(kmin, kmax);
    int64_t maxw = 1;
    for (int64_t t = 0; t < ncand; ++t)
        if (c_b[t] - c_a[t] > maxw) maxw = c_b[t] - c_a[t];
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < ncand; ++t) {
        const unsigned char* s = (const unsigned char*)seqs[c_seq[t]] + c_a[t];
        const int64_t n = c_b[t] - c_a[t];
        int64_t nn = 0;
        for (int64_t p = 0; p < n; ++p) nn += !lut_upper[s[p]];
        keep[t] = !((double)nn >= 0.3 * (double)n);
    }
    int64_t* slot = (int64_t*)malloc((size_t)(ncand + 1) * sizeof(int64_t));
    int64_t rows = 0;
    for (int64_t t = 0; t < ncand; ++t) {
        slot[t] = rows;
        rows += keep[t];
    }
    int64_t ret = rows;
    if (rows > cap) {
        ret = -rows;
    } else {
#pragma omp parallel
        {
            Scratch sc;
            sc.cnt = (int32_t*)calloc((size_t)L, sizeof(int32_t));
            sc.present = (int32_t*)malloc((size_t)maxw * sizeof(int32_t));
            sc.iw = (double*)malloc((size_t)maxw * sizeof(double));
#pragma omp for schedule(dynamic, 16)
            for (int64_t t = 0; t < ncand; ++t) {
                if (!keep[t]) continue;
                const int64_t r = slot[t];
                double dpi = NAN, dsi = NAN, dcri = NAN;
                score_window((const unsigned char*)seqs[c_seq[t]] + c_a[t], c_b[t] - c_a[t], ig, kmin, kmax, rip, &sc,
                             &status[r], &kld[r], &gc[r], &dpi, &dsi, &dcri, dbg_counts ? dbg_counts + r * L : NULL,
                             dbg_meta ? dbg_meta + r * 3 : NULL);
                seq_index[r] = c_seq[t];
                start[r] = c_s[t];
                stop[r] = c_e[t];
                if (pi) pi[r] = dpi;
                if (si) si[r] = dsi;
                if (cri) cri[r] = dcri;
            }
            free(sc.cnt);
            free(sc.present);
            free(sc.iw);
        }
    }
    free(slot);
    free(first);
    free(keep);
    free(c_seq);
    free(c_a);
    free(c_b);
    free(c_s);
    free(c_e);
    return ret;
}

// scan_kernel.h - phase B: one launch scores every candidate window of the resident batch.
//
// Replaces the reference's per-window loop frisk/__init__.py L1478-1494:
//   crawlGenome (L194-251) -> computeKmers(window) (L280-367) -> IvomBuild(genome side) + IvomBuild(window
//   side) (L369-457) -> KLD (L459-472) -> calcGC (L120-137) [-> calcRIP (L474-495)].
//
// Mapping onto CDNA4: one workgroup (512 threads = 8 wavefronts on the unrolled fast paths) owns one window at a
// time and keeps the
// window's k-mer histograms in LDS as dense 16-bit counters (two per dword, updated with 32-bit ds_add):
//   * every position does ONE update of the "small" tables (orders kmin..6, or kmin..K when K <= 7): to the
//     table of order r = min(run, top) at the code of its longest valid word; lower orders then follow by an
//     in-LDS marginalisation  C_x[q] = D_x[q] + sum_b C_{x+1}[4q+b]  (exact, incl. window tail and N-adjacent
//     positions) - this replaces K updates per position, most of them on the few, heavily contended
//     low-order bins;
//   * order 8 (K = 8): dense 16-bit counters, 128 KiB - the reason a workgroup owns a whole CU; order 7 is
//     then NOT stored: c7(q) = sum of the four order-8 children of q (one 8-byte LDS read) + the few "orphan"
//     7-mers whose 8th base is missing (window tail, or an invalid base), kept in a short LDS list.
// The update of the max-mer table is a returning atomic; the lane that sees the old value 0 becomes the
// *representative* of that max-mer and later evaluates its IVOM recursion, so no scan over the 4^K bins is
// ever needed, and bins are re-zeroed by their representatives instead of a 128 KiB clear.
// Scoring uses two algebraic identities of the reference's arithmetic (IvomBuild L426-446, KLD L465-470):
//   * the interpolation recursion  I_x = a_x p_x + (1-a_x) I_{x-1},  a_x = w_x / (w_m+..+w_x),  telescopes
//     (1 - a_x = W_{x-1}/W_x) to the closed form  I_K = (sum_x w_x p_x) / W_K = (sum_x c_x^2 4^x / D_x) / (sum_x c_x 4^x):
//     one division per max-mer instead of two per order.  The partial sums over orders kmin..5 depend only on
//     the 5-mer prefix and are computed once per present 5-mer (<= 1024 per window) into an LDS table;
//   * with Pw = Iw/Sw and Pg = Ig/Sg,  sum Pw log(Pw/Pg) = T/Sw - log Sw + log Sg,  T = sum Iw log(Iw/Ig):
//     one pass over the max-mers, no normalisation pass.
// Same mathematics, different rounding: |KLD - reference| stays below ~1e-13 (north-star bound 1e-6; the parity
// tests assert 1e-11 / 1e-10).  The oracles keep the reference's operation order.
// The 128 KiB table allows one workgroup per CU, and the scoring code wants ~200 VGPRs per lane to keep several
// positions' division/log chains in flight, so the fast paths run 512 threads (2 wavefronts per SIMD, 256 VGPRs
// available) with the per-position loops fully unrolled (ITS iterations, a template parameter): loads are
// unconditional and issued ahead of their use, every position of a lane is evaluated in straight-line code
// (non-representatives are masked out when accumulating).  Measured: same speed as 1024 threads at 128 VGPRs
// for identical code - and room for the cheaper exact summation below, which spills at 128 VGPRs.
// FP64 throughout (the terms cancel from O(1) to O(1e-2)).  No MFMA: this is histogramming + scalar arithmetic.
#pragma once
#include <type_traits>

#include "frisk_device.h"

#define FRISK_T8_BYTES 131072

struct ScanParams {
    const uint32_t* codes;
    const uint32_t* inv;
    const uint32_t* low;
    const ScafDesc* descs;
    const double* ig;         // genome-side IVOM, 4^kmax entries (NaN = zero weight)
    const double* log_tab;    // FRISK_LOGTAB_N pairs {1/c_i, ln c_i}: range reduction of log_tab_pos()
    const double* log_tab64;  // the same with 64 bins (scan8_kernel.h, where LDS is short)
    const double* log_tab32;  // ... and with 32
    int32_t n_desc;
    int32_t kmin, kmax;
    int32_t w, inc;
    uint32_t flags;           // FRISK_SCAN_*
    int64_t c0, c1;           // candidate range
    int32_t chunk;            // consecutive candidates handed to a workgroup at a time
    int32_t orphan_cap;       // capacity of the orphan list (entries)
    int32_t lv;               // shared_level(), or 0 when the prefix tables do not fit beside a long window's orphan list
    int32_t nprof;            // profile length (debug dump stride)
    // outputs, indexed by (candidate - c0)
    int32_t* seq_index;
    int64_t* start;
    int64_t* stop;
    uint32_t* status;
    double* kld;              // scan_kernel leaves T = sum Iw ln(Iw/Ig) here, finish_rows_kernel turns it into the KLD
    double* gc;               // scan_kernel leaves {S << 32 | G+C} here, finish_rows_kernel the fraction
    double* sw;               // raw sums of the row: sum Iw ...
    double* sg;               // ... and sum Ig (NaN: a max-mer without genome weight)
    double* pi;
    double* si;
    double* cri;
    uint32_t* dbg_counts;
    int64_t* dbg_meta;
    double* dbg_ivom;             // debug, kmax <= 6: row x 2 x 4^kmax - un-normalised window-side and genome-side IVOM per max-mer
    unsigned long long* stamps;   // -DFRISK_STAMPS builds only: s_memtime at the stage boundaries of the first workgroups
    // scan8_kernel.h (narrow order-8 counters): windows whose counters wrap are handed to the next wider form through
    // device-side lists of candidate indices
    const double* rc_tab;         // 1/c for c = 0..255 (entry 0 = 0): the weight of one of the c positions that share a max-mer
    const int64_t* in_list;       // != nullptr: take the candidates from in_list[0 .. *in_count) instead of [c0, c1)
    const unsigned int* in_count;
    int64_t* out_list;            // scan8_kernel appends the windows it could not hold ...
    unsigned int* out_count;      // ... and counts them
    int32_t sel_mode, sel_mod;    // scan8_kernel, range mode: 0 all chunks, 1 every sel_mod-th chunk, 2 all the others
    unsigned int* queue;          // scan8_kernel: != nullptr: chunks are dealt by these counters (zero at launch) instead of by block index:
    int32_t queue_n;              // ... queue_n (1 or 8) of them, one per XCD, each over a contiguous share of the chunks
    double* ig_ring;              // scan8_kernel: per-workgroup ring of genome-side values by window position (see scan8_kernel.h), or nullptr
    const unsigned int* verdict;  // scan8_kernel: != nullptr: this launch runs only if *verdict == my_form (the adaptive width's sample decides on the
    unsigned int my_form;         // ... device which of the bulk forms - 1 plain 4-bit, 2 4-bit + side table, 3 8-bit - scores the rest; the others return at once)
    int32_t slide_pp;             // scan8_kernel: > 0: inside a chunk the order-K table slides from window to window, this many positions of
                                  // the leaving and of the entering range per thread (= ceil(inc / threads)); 0: every window counted afresh
};

#define ROW_KEPT 1u
#define ROW_ZERO_WEIGHT 2u
#define ROW_JUMPBACK 4u
#define ROW_NO_MAXMER 8u

#pragma clang fp contract(off)

// Diagnostic builds only (-DFRISK_STAMPS): wave FRISK_STAMP_WAVE (0) of the first 4 workgroups records s_memtime at up to 12 points of its first
// 16 windows; the launcher prints the per-stage cycle differences.  Never in the product library.
#ifdef FRISK_STAMPS
#ifndef FRISK_STAMP_WAVE
#define FRISK_STAMP_WAVE 0
#endif
#define STAMP(i)                                                                                          \
    if (tid == 64 * FRISK_STAMP_WAVE && blockIdx.x < 4 && stamp_win < 16)                                 \
        P.stamps[(blockIdx.x * 16 + stamp_win) * 12 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define STAMP(i)
#endif

#ifndef FRISK_ABL
#define FRISK_ABL 0       // diagnostic builds (tools/ablate.py): drop one ingredient at a time; results wrong by design
#endif

// Diagnostic builds only (tools/ablate.py, -DFRISK_STOP=<n>): finish every window right after stage <n> with a
// value that depends on the stage's results, to time the stages cumulatively.  Never in the product library.
#ifdef FRISK_STOP
#define STOP_AFTER(stage, value)                                                              \
    if (FRISK_STOP == (stage)) {                                                              \
        const double v_ = double(value);                                                      \
        __syncthreads();                                                                      \
        cleanup();                                                                            \
        if (tid == 0) { P.status[row] = ROW_KEPT; P.kld[row] = v_; P.gc[row] = 0.5; }         \
        __syncthreads();                                                                      \
        continue;                                                                             \
    }
#else
#define STOP_AFTER(stage, value)
#endif

// LDS carve-up (dynamic, all offsets multiples of 16 bytes)
struct LdsLayout {
    uint32_t t8;        // byte offset of the order-8 table (K8 only)
    uint32_t small;     // byte offset of the small tables (orders kmin..ks), u16 bins
    uint32_t small_bytes;
    uint32_t orphans;   // u16 list
    uint32_t pre_i;     // f64[4^lv]: sum_{x<=lv} c_x^2 4^x/D_x of the lv-mer prefix
    uint32_t pre_w;     // u32[4^lv]: running weight sum after order lv
    uint32_t rtab;      // f64[16]: 4^x / ((S-(x-1))*2) per order x (window constants)
    uint32_t logtab;    // f64[2*FRISK_LOGTAB_N]: {1/c_i, ln c_i}, written once per workgroup
    uint32_t misc;      // 2 x 16 u32 counters (double-buffered by window parity) + reduction scratch
    uint32_t t8_bytes;  // 128 KiB at K = 8
    uint32_t total;
};

// order at which the recursion is shared between max-mers (0 = not shared)
__host__ __device__ inline int shared_level(int kmin, int kmax) { return (kmin <= 5 && kmax >= 6) ? 5 : 0; }

#define FRISK_MISC_SLOTS 16
#define FRISK_LOGTAB_N 128
#define FRISK_MISC_BYTES (2 * FRISK_MISC_SLOTS * 4 + 16 * 6 * 8)       // counters x2, scratch (16 waves x 3 x 128 bit)

__host__ __device__ inline LdsLayout make_layout(int kmin, int kmax, int orphan_cap, int lv) {
    LdsLayout L;
    const bool k8 = (kmax == 8);
    const int ks = k8 ? 6 : kmax;
    uint32_t o = 0;
    L.t8 = o;
    L.t8_bytes = k8 ? FRISK_T8_BYTES : 0;
    o += L.t8_bytes;
    L.small = o;
    int64_t bins = (ks >= kmin) ? table_offset(kmin, ks + 1) : 0;
    L.small_bytes = uint32_t((bins * 2 + 15) / 16 * 16);
    o += L.small_bytes;
    L.orphans = o;
    o += uint32_t((k8 ? orphan_cap : 0) * 2 + 15) / 16 * 16;
    L.pre_i = o;
    if (lv) o += (1u << (2 * lv)) * 8;
    L.pre_w = o;
    if (lv) o += (1u << (2 * lv)) * 4;
    L.rtab = o;
    o += 16 * 8;
    L.logtab = o;
    o += FRISK_LOGTAB_N * 16;
    L.misc = o;
    o += FRISK_MISC_BYTES;
    L.total = (o + 15) / 16 * 16;
    return L;
}

// misc counter slots
enum { M_UPA = 0, M_UPT, M_UPG, M_UPC, M_NORPH, M_NVALID, M_FLAGS };

template <bool K8>
struct WinTables {
    static constexpr uint32_t M7 = 0x3FFFu;
    const uint16_t* t8_16;
    const uint16_t* small16;
    const uint16_t* orph;
    int n_orph;
    uint32_t o0, o1, o2, o3;     // the first four orphan 7-mers (0xFFFFFFFF = none), wave-uniform
    int kmin;

    // occurrences of the 7-mer `c` in the orphan list: a window has one orphan (its tail) plus one per invalid run -
    // almost always <= 4
    template <int ORPH = 0>
    __device__ inline uint32_t orphans(uint32_t c) const {
        uint32_t s = (c == o0 ? 1u : 0u);
        if (ORPH != 1) s += (c == o1 ? 1u : 0u) + (c == o2 ? 1u : 0u) + (c == o3 ? 1u : 0u);
        if (ORPH == 0) for (int o = 4; o < n_orph; ++o) s += (orph[o] == c) ? 1u : 0u;
        return s;
    }

    // the max-mer's own count and the sum over the four children of its 7-mer prefix, from ONE aligned read
    __device__ inline void top(uint32_t code, uint32_t& c8, uint32_t& children) const {
        const uint32_t q7 = (code >> 2) & M7;
        const uint2 ch = *reinterpret_cast<const uint2*>(t8_16 + 4 * q7);
        const uint64_t both = (uint64_t(ch.y) << 32) | ch.x;
        c8 = uint32_t(both >> ((code & 3u) * 16)) & 0xFFFFu;
        children = (ch.x & 0xFFFFu) + (ch.x >> 16) + (ch.y & 0xFFFFu) + (ch.y >> 16);
    }

    // count of the x-mer `c` in the current window.  ORPH = what the caller knows about the window's orphan list:
    // 1: at most one entry (the usual case: the window's tail), 4: at most four, 0: anything
    template <int ORPH = 0>
    __device__ inline uint32_t count(int x, uint32_t c) const {
        if (K8) {
            if (x == 8) { uint32_t c8, ch; top(c, c8, ch); return c8; }
            if (x == 7) { uint32_t c8, ch; top(c << 2, c8, ch); return ch + orphans<ORPH>(c); }
        }
        return small16[table_offset(kmin, x) + c];
    }
};

// natural logarithm of a positive normal double, < 1 ulp: the classic reduction x = 2^k * (1+f),
// sqrt(2)/2 < 1+f < sqrt(2), log(1+f) = f - f^2/2 + s*(f^2/2 + R(s^2)), s = f/(2+f), with the degree-14 minimax
// polynomial R of fdlibm's e_log.c (Sun Microsystems, freely distributable constants).  A third of the
// instructions of the device library's log; the argument here is a ratio of probabilities.
__device__ inline double div_exact(double n, double d);
__device__ inline double log_pos(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    int k = __builtin_amdgcn_frexp_exp(x);              // x = m * 2^k, m in [0.5, 1)
    double m = __builtin_amdgcn_frexp_mant(x);
    const bool lowhalf = m < 0.70710678118654752440;
    m = lowhalf ? m * 2.0 : m;
    k = lowhalf ? k - 1 : k;
    const double f = m - 1.0;
    const double s = div_exact(f, 2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = double(k);
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// ln(x) by table range reduction: x = m 2^k with m in [0.5, 1); the top 7 mantissa bits pick c_i = the midpoint of
// m's 1/256-wide bin; tab[i] = {u_i = 1/c_i rounded to double, -ln(u_i)}; r = m u_i - 1 (one fma, |r| < 2^-8) and
// ln x = k ln2 - ln u_i + log1p(r), log1p by its degree-7 Taylor polynomial (truncation < 2^-67).  ABSOLUTE error
// ~1e-16 for the ratios scored here (|k| small) - what the sum T = sum Iw ln(Iw/Ig) needs - at 17 instructions
// instead of log_pos()'s 42 (no division).
__device__ inline double log_tab_pos(double x, const double2* tab) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const int k = __builtin_amdgcn_frexp_exp(x);
    const double m = __builtin_amdgcn_frexp_mant(x);
    const uint32_t i = (uint32_t(__double2hiint(m)) >> 13) & (FRISK_LOGTAB_N - 1);
    const double2 e = tab[i];
    const double r = __builtin_fma(m, e.x, -1.0);
    double p = __builtin_fma(r, 1.0 / 7.0, -1.0 / 6.0);
    p = __builtin_fma(r, p, 1.0 / 5.0);
    p = __builtin_fma(r, p, -1.0 / 4.0);
    p = __builtin_fma(r, p, 1.0 / 3.0);
    p = __builtin_fma(r, p, -0.5);
    const double dk = double(k);
    const double small = __builtin_fma(dk, ln2_lo, __builtin_fma(r * r, p, r));
    return __builtin_fma(dk, ln2_hi, e.y) + small;
}

// ---- order-independent summation ---------------------------------------------------------------------
// Which lane becomes the representative of a max-mer depends on the arrival order of LDS atomics, so an ordinary
// floating-point sum over representatives would depend on timing.  Every per-window sum is therefore accumulated
// as a PAIR of doubles whose additions are exact, hence associative:
//   hi = the term rounded to a multiple of 2^-26   ((x + 1.5*2^26) - 1.5*2^26, round-to-nearest-even)
//   lo = the remainder (exact, |lo| <= 2^-27) rounded to a multiple of 2^-64
// |term| < 2^11 and a window has < 2^16 max-mers, so every partial sum of hi's is a multiple of 2^-26 below 2^27 and
// every partial sum of lo's a multiple of 2^-64 below 2^-11: both fit the 53-bit mantissa, no addition ever rounds.
// The per-term rounding at 2^-64 is a fixed function of the term.  Results are therefore bit-identical across
// runs, builds, grids and GPUs (a 128-bit integer accumulator cost 18 instructions per term).
struct ExactSum {
    double hi, lo;
};

#define FRISK_EXACT_BIAS 0x1.8p26                            // 1.5 * 2^(52-26): an accumulator in [2^26, 2^27) has ulp 2^-26
// The lane-private accumulators carry the bias in `hi` (exact_begin / exact_end), which makes the running sum itself do
// the rounding of the term: hi' = hi + x is hi + (x rounded to 2^-26), exactly (|lane sum| < 2^17 keeps hi in range),
// and h = hi' - hi, x - h are exact (Fast2Sum).  On a tie h depends on the parity of hi, i.e. on what the lane added
// before; h + l does not (the 2^-64 grid of l is shifted by a multiple of itself), so the totals still do not depend
// on which lane scored which max-mer.  6 instructions per term.
__device__ inline ExactSum exact_begin() { return ExactSum{FRISK_EXACT_BIAS, 0.0}; }
__device__ inline void exact_end(ExactSum& a) { a.hi -= FRISK_EXACT_BIAS; }
__device__ inline void exact_add(ExactSum& a, double x) {
    const double C2 = 0x1.8p-12;                            // 1.5 * 2^(52-64)
    const double s = a.hi + x;
    const double h = s - a.hi;
    const double l = ((x - h) + C2) - C2;
    a.hi = s;
    a.lo += l;
}

__device__ inline double exact_value(const ExactSum& a) { return a.hi + a.lo; }

// n / d for operands whose quotient needs no exponent scaling (here: positive integers < 2^53 as doubles, and
// ratios of normal probabilities): reciprocal refinement + residual correction as the compiler emits for an IEEE
// fdiv, without the v_div_scale / v_div_fmas / v_div_fixup range handling and with ONE Newton step instead of two:
// v_rcp_f64 is good to 24.4 bits on gfx950 (measured), one step gives 2^-48.8, and the residual correction then
// lands within 2^-97 of the true quotient - identical to the IEEE quotient in 4 194 304 of 4 194 304 random
// divisions (tools/exp/rcp_accuracy.hip; a difference needs the quotient within 2^-44 ulp of a rounding boundary).
__device__ inline double div_exact(double n, double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
#ifdef FRISK_DIV_2NR
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
#endif
    const double q = n * r;
    e = __builtin_fma(-d, q, n);
    return __builtin_fma(e, r, q);
}

// sum three accumulators over the workgroup; every thread returns the same totals.  One barrier: the scratch is
// rewritten only after the window's last barrier.
// WAVE0_ONLY: only the first wave adds up the per-wave partials (the others return garbage): the caller's scalar tail -
// two logarithms and a division - then costs one wave's instructions per window instead of every wave's.
// x + (x of another lane of the same row of 16, chosen by a DPP control): one register move per 32-bit half
template <int CTRL>
__device__ inline double dpp_add(double x) {
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, false);
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, false);
    return x + __hiloint2double(hi, lo);
}

// sum of x over the wave, valid in every lane: a butterfly inside each row of 16 lanes by DPP (quad_perm [1,0,3,2] and
// [2,3,0,1], row_half_mirror, row_mirror), then the four row sums by v_readlane.  The additions here are exact (see
// ExactSum), so their order is free.  (A __shfl_down tree is six dependent LDS-crossbar round trips per value.)
__device__ inline double wave_sum_exact(double x) {
    x = dpp_add<0xB1>(x);
    x = dpp_add<0x4E>(x);
    x = dpp_add<0x141>(x);
    x = dpp_add<0x140>(x);
    const int hi = __double2hiint(x), lo = __double2loint(x);
    auto row = [&](int l) { return __hiloint2double(__builtin_amdgcn_readlane(hi, l), __builtin_amdgcn_readlane(lo, l)); };
    return (row(0) + row(16)) + (row(32) + row(48));
}

template <int NW, bool WAVE0_ONLY = false>
__device__ inline void block_sum3(ExactSum& a, ExactSum& b, ExactSum& c, double* scratch, int tid) {
    a.hi = wave_sum_exact(a.hi); a.lo = wave_sum_exact(a.lo);
    b.hi = wave_sum_exact(b.hi); b.lo = wave_sum_exact(b.lo);
    c.hi = wave_sum_exact(c.hi); c.lo = wave_sum_exact(c.lo);
    if ((tid & 63) == 0) {
        double* p = scratch + (tid >> 6) * 6;
        p[0] = a.hi; p[1] = a.lo; p[2] = b.hi; p[3] = b.lo; p[4] = c.hi; p[5] = c.lo;
    }
    __syncthreads();
    if (WAVE0_ONLY && tid >= 64) return;
    ExactSum sa = {0.0, 0.0}, sb = {0.0, 0.0}, sc = {0.0, 0.0};
    for (int w = 0; w < NW; ++w) {
        const double* p = scratch + w * 6;
        sa.hi += p[0]; sa.lo += p[1]; sb.hi += p[2]; sb.lo += p[3]; sc.hi += p[4]; sc.lo += p[5];
    }
    a = sa;
    b = sb;
    c = sc;
}

// NT: threads per workgroup.
// ITS > 0: the window has at most ITS*NT positions; per-position loops are fully unrolled and the per-position
//          codes and IVOM values stay in registers between the passes.
// ITS == 0: any length up to 65535; runtime loops, values recomputed in the last pass.
template <int NT, bool K8, int ITS, bool DEBUG>
__global__ __launch_bounds__(NT, (NT == 256 ? 2 : 1)) void scan_kernel(const ScanParams P) {
    constexpr int NW = NT / 64;
    constexpr int NREG = ITS > 0 ? ITS : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int kmin = P.kmin;
    const int kmax = K8 ? 8 : P.kmax;                   // compile-time at K = 8: the order loops unroll
    const LdsLayout L = make_layout(kmin, kmax, P.orphan_cap, P.lv);
    uint32_t* t8 = reinterpret_cast<uint32_t*>(lds + L.t8);
    uint32_t* small32 = reinterpret_cast<uint32_t*>(lds + L.small);
    uint16_t* small16 = reinterpret_cast<uint16_t*>(lds + L.small);
    uint16_t* orph = reinterpret_cast<uint16_t*>(lds + L.orphans);
    double* pre_i = reinterpret_cast<double*>(lds + L.pre_i);
    uint32_t* pre_w = reinterpret_cast<uint32_t*>(lds + L.pre_w);
    double* rtab = reinterpret_cast<double*>(lds + L.rtab);
    double2* logtab = reinterpret_cast<double2*>(lds + L.logtab);
    using wsum_t = typename std::conditional<(ITS > 0), uint32_t, unsigned long long>::type;   // W < 2^32 for n <= 8192
    uint32_t* misc_base = reinterpret_cast<uint32_t*>(lds + L.misc);
    double* scratch_base = reinterpret_cast<double*>(lds + L.misc + 2 * FRISK_MISC_SLOTS * 4);
    const int ks = K8 ? 6 : kmax;                       // highest order kept in the small tables
    const int lv = P.lv;                                // recursion shared up to this order (0: not shared)
    const int kshift = 16 - 2 * kmax;

    // (a launch over a hand-over list finds it empty nearly always: nothing to set up)
    if (P.in_list != nullptr && *P.in_count == 0u) return;
    // one-time clear of the histograms and counters
    if (K8) for (int i = tid; i < int(L.t8_bytes / 16); i += NT) reinterpret_cast<uint4*>(t8)[i] = make_uint4(0, 0, 0, 0);
    for (uint32_t i = tid; i < L.small_bytes / 16; i += NT) reinterpret_cast<uint4*>(small32)[i] = make_uint4(0, 0, 0, 0);
    if (tid < 2 * FRISK_MISC_SLOTS) misc_base[tid] = 0;
    if (tid < FRISK_LOGTAB_N) logtab[tid] = reinterpret_cast<const double2*>(P.log_tab)[tid];
    __syncthreads();

    // XCD-aware work split: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a
    // contiguous run of chunk ids - neighbouring windows overlap by w-inc bases and then share one L2.
    const int G = gridDim.x;
    int v = blockIdx.x;
    if ((G & 7) == 0) v = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    // candidates: the range [c0, c1) in chunks, or (in_list) the windows that scan8_kernel handed over, one at a time
    const bool listed = P.in_list != nullptr;
    const int64_t ncand = listed ? int64_t(*P.in_count) : P.c1 - P.c0;
    const int64_t chunk = listed ? 1 : P.chunk;
    const int64_t nchunks = (ncand + chunk - 1) / chunk;

    ScafDesc d;
    d.cand0 = 0; d.ncand = 0; d.off = 0; d.size = 0; d.kind = 0; d.base0 = 0; d.j0 = 0;
    int dsi = -1;
    uint32_t parity = 0;
#ifdef FRISK_STAMPS
    int stamp_win = -1;
#endif

    for (int64_t q = v; q < nchunks; q += G) {
        const int64_t cb = listed ? q : P.c0 + q * chunk;
        const int64_t ce = listed ? q + 1 : ((cb + chunk < P.c1) ? cb + chunk : P.c1);
        for (int64_t ci = cb; ci < ce; ++ci) {
            const int64_t cand = listed ? P.in_list[ci] : ci;
            // ---- which scaffold / window is this candidate? (uniform across the workgroup)
            if (cand < d.cand0 || cand >= d.cand0 + d.ncand) {
                int lo = 0, hi = P.n_desc - 1;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (P.descs[mid].cand0 <= cand) lo = mid; else hi = mid - 1;
                }
                // (descriptors without candidates share cand0 with their successor, so the LAST
                //  descriptor with cand0 <= cand is always the one that owns the candidate)
                d = P.descs[lo];
                dsi = lo;
            }
#ifdef FRISK_STAMPS
            ++stamp_win;
#endif
            const int64_t j = cand - d.cand0 + d.j0;           // window index inside the scaffold
            int64_t st;                // 0-based first base of the window inside the scaffold
            int64_t rep_start, rep_stop;   // coordinates as the reference reports them
            int n;
            bool jump = false;
            if (d.kind == 1) { st = 0; n = int(d.size); rep_start = 1; rep_stop = d.size; }      // L219
            else {
                st = j * P.inc;
                n = P.w;
                rep_start = st + 1; rep_stop = st + P.w;                                        // L245
                if (st + P.w > d.size) {                                                        // L230-232
                    jump = true;
                    st = d.size - P.w;
                    rep_start = st; rep_stop = d.size;                                          // L243: 0-based start
                    // a scaffold shorter than w: seq[size-w:size] with a negative start is a Python slice
                    // counted from the end, clamped at 0
                    if (st < 0) { st += d.size; if (st < 0) st = 0; }
                    n = int(d.size - st);
                }
            }
            const int64_t g0 = d.off + (st - d.base0);            // resident position of the window's first base
            const int64_t row = cand - P.c0;
            STAMP(0)
            uint32_t* misc = misc_base + parity * FRISK_MISC_SLOTS;         // this window's counters
            uint32_t* misc_other = misc_base + (parity ^ 1u) * FRISK_MISC_SLOTS;
            parity ^= 1u;

            // ---- stage 1: one pass over the window's positions ------------------------------------------
            //   * uppercase base composition (calcGC L120-137, countN L106-118): order-1 counts minus the soft-masked
            //     bases (wave ballots only where a wave meets lowercase; full ballots when kmin > 1)
            //   * ONE small-table update per position + the max-mer update that elects representatives
            // (done before the N filter is known: 93 % of windows pass it, the others are cleaned up after stage 2)
            unsigned long long repmask = 0;
            uint32_t c16v[NREG];
            const bool tally_by_ballot = (kmin != 1);
            {
                uint32_t cA = 0, cT = 0, cG = 0, cC = 0, nvalid = 0;
                // any position: ONE small-table update at the order of its longest valid word (<= 6 at K = 8), the
                // max-mer update with the election, or the orphan list.  Returns whether the position holds a max-mer.
                auto position_generic = [&](int it, int jj, uint32_t c16, uint32_t inv8) -> bool {
                    int run = lead_clear8(inv8);                             // window words are upper-cased: L334-335
                    const int rem = n - jj;
                    run = run < rem ? run : rem;
                    run = run < kmax ? run : kmax;
                    bool is_top = false;
                    if (K8) {
                        const int rs = run < 6 ? run : 6;
                        if (rs >= kmin) {
                            const uint32_t b = uint32_t(table_offset(kmin, rs)) + (c16 >> (16 - 2 * rs));
                            atomicAdd(&small32[b >> 1], 1u << ((b & 1u) * 16));
                        }
                        if (run == 8) {
                            const uint32_t old = atomicAdd(&t8[c16 >> 1], 1u << ((c16 & 1u) * 16));
                            is_top = true;
                            if (((old >> ((c16 & 1u) * 16)) & 0xFFFFu) == 0) repmask |= 1ull << it;
                        } else if (run == 7 && kmin <= 7) {
                            const uint32_t slot = atomicAdd(&misc[M_NORPH], 1u);
                            orph[slot] = uint16_t(c16 >> 2);
                        }
                    } else if (run >= kmin) {
                        const uint32_t b = uint32_t(table_offset(kmin, run)) + (c16 >> (16 - 2 * run));
                        if (run == kmax) {
                            const uint32_t old = atomicAdd(&small32[b >> 1], 1u << ((b & 1u) * 16));
                            is_top = true;
                            if (((old >> ((b & 1u) * 16)) & 0xFFFFu) == 0) repmask |= 1ull << it;
                        } else {
                            atomicAdd(&small32[b >> 1], 1u << ((b & 1u) * 16));
                        }
                    }
                    return is_top;
                };
                // composition tallies of one position by wave ballots (`sel`: the lanes whose base is counted)
                auto tally = [&](bool sel, uint32_t c2) {
                    cA += __popcll(__ballot(sel && c2 == 0));
                    cT += __popcll(__ballot(sel && c2 == 1));
                    cG += __popcll(__ballot(sel && c2 == 2));
                    cC += __popcll(__ballot(sel && c2 == 3));
                };
                if constexpr (ITS > 0) {
                    // A lane owns ITS CONSECUTIVE positions j0 .. j0+ITS-1: one set of loads (3 + 2 + 2 words; ITS + 7 <= 23
                    // bases) serves all of them.  Bit 31-it of the lane masks below <-> position j0+it.
                    const int j0 = tid * ITS;
                    const int64_t gl = g0 + (j0 < n ? j0 : 0);               // clamped: loads are unconditional
                    const int64_t wi = gl >> 4, mi = gl >> 5;
                    const int shc = 32 - int(gl & 15) * 2, shm = 32 - int(gl & 31);
#if defined(FRISK_ABL) && (FRISK_ABL & 32)      // diagnostic: no window loads at all (fake bases, all valid)
                    const uint32_t w0 = uint32_t(wi) * 2654435761u, w1 = w0 ^ 0x9E3779B9u, w2 = w1 * 40503u;
#else
                    const uint32_t w0 = P.codes[wi], w1 = P.codes[wi + 1], w2 = P.codes[wi + 2];
#endif
                    const uint32_t hi = uint32_t(((uint64_t(w0) << 32) | w1) >> shc);
                    const uint32_t lo = uint32_t(((uint64_t(w1) << 32) | w2) >> shc);
                    const uint64_t acode = (uint64_t(hi) << 32) | lo;        // bases j0 .. j0+31, first base in the top bits
#if defined(FRISK_ABL) && (FRISK_ABL & 32)
                    const uint32_t ainv = 0u, alow = uint32_t(mi >> 40) + uint32_t(shm >> 8);
#else
                    const uint32_t ainv = uint32_t(((uint64_t(P.inv[mi]) << 32) | P.inv[mi + 1]) >> shm);
                    const uint32_t alow = uint32_t(((uint64_t(P.low[mi]) << 32) | P.low[mi + 1]) >> shm);
#endif
                    auto topbits = [](int k) -> uint32_t {                  // the k most significant bits (k clamped to 0..32)
                        k = k < 0 ? 0 : (k > 32 ? 32 : k);
                        return uint32_t(0xFFFFFFFF00000000ull >> k);
                    };
                    constexpr uint32_t MINE = uint32_t(0xFFFFFFFF00000000ull >> ITS);
                    const int nleft = n - j0;                                // this lane's positions inside the window
                    const uint32_t actm = topbits(nleft) & MINE;
                    const uint32_t vld = ~ainv;
                    uint32_t fullm = vld;                                    // kmax valid bases from here on ...
                    if (K8) {
                        fullm &= fullm << 1; fullm &= fullm << 2; fullm &= fullm << 4;
                    } else {
                        for (int i = 1; i < kmax; ++i) fullm &= vld << i;
                    }
                    fullm &= topbits(nleft - kmax + 1) & MINE;               // ... all of them inside the window
                    const bool small_on = kmin <= 6;                         // K8: positions feed the order-6 table
                    const uint32_t off_full = uint32_t(table_offset(kmin, K8 ? 6 : kmax));
                    const int sh_full = K8 ? 4 : 16 - 2 * kmax;
#pragma unroll
                    for (int it = 0; it < ITS; ++it) {
                        constexpr uint32_t TOP = 0x80000000u;
                        const uint32_t bit = TOP >> it;
                        const uint32_t c16 = uint32_t(acode >> (48 - 2 * it)) & 0xFFFFu;
                        c16v[it] = c16;
                        if (fullm & bit) {                                   // the usual case: a max-mer starts here
                            if (K8) {
                                if (small_on) {
                                    const uint32_t b = off_full + (c16 >> 4);
                                    atomicAdd(&small32[b >> 1], 1u << ((b & 1u) * 16));
                                }
                                const uint32_t old = atomicAdd(&t8[c16 >> 1], 1u << ((c16 & 1u) * 16));
                                if (((old >> ((c16 & 1u) * 16)) & 0xFFFFu) == 0) repmask |= 1ull << it;
                            } else {
                                const uint32_t b = off_full + (c16 >> sh_full);
                                const uint32_t old = atomicAdd(&small32[b >> 1], 1u << ((b & 1u) * 16));
                                if (((old >> ((b & 1u) * 16)) & 0xFFFFu) == 0) repmask |= 1ull << it;
                            }
                        } else if (actm & bit) {                             // near an invalid base or the window's end
                            position_generic(it, j0 + it, c16, (ainv >> (24 - it)) & 0xFFu);
                        }
                    }
                    // wave total of the lanes' max-mer counts (<= ITS <= 16 each) from one ballot per bit - the compiler's
                    // own reduction of a divergent atomic operand is a 64-step scalar loop
                    const uint32_t ntop = __popc(fullm);
#pragma unroll
                    for (int b = 0; (1 << b) <= ITS; ++b) nvalid += uint32_t(__popcll(__ballot((ntop >> b) & 1u))) << b;
                    if (tally_by_ballot) {          // kmin > 1: no order-1 table to read the composition from
                        const uint32_t upm = actm & vld & ~alow;
#pragma unroll
                        for (int it = 0; it < ITS; ++it)
                            tally((upm >> (31 - it)) & 1u, uint32_t(acode >> (62 - 2 * it)) & 3u);
                    } else {                        // count only the soft-masked valid bases (rare): upper = C_1 - these
                        const uint32_t lowm = actm & vld & alow;
                        if (__ballot(lowm != 0)) {
#pragma unroll
                            for (int it = 0; it < ITS; ++it)
                                tally((lowm >> (31 - it)) & 1u, uint32_t(acode >> (62 - 2 * it)) & 3u);
                        }
                    }
                } else {
                    for (int it = 0; it * NT < n; ++it) {
                        const int jj = tid + it * NT;
                        const bool act = jj < n;
                        const int64_t g = g0 + (act ? jj : 0);               // clamped: loads are unconditional
                        const uint32_t c16 = fetch_codes16(P.codes, g);
                        const uint32_t inv8 = fetch_mask8(P.inv, g);
                        const uint32_t low1 = fetch_mask1(P.low, g);
                        const uint32_t c2 = c16 >> 14;
                        const bool is_top = act && position_generic(it, jj, c16, inv8);
                        if (tally_by_ballot) {
                            tally(act && !((inv8 >> 7) | low1), c2);
                        } else {
                            const bool lowv = act && low1 && !(inv8 >> 7);
                            if (__ballot(lowv)) tally(lowv, c2);
                        }
                        nvalid += __popcll(__ballot(is_top));
                    }
                }
                if (lane == 0) {
                    if (cA) atomicAdd(&misc[M_UPA], cA);
                    if (cT) atomicAdd(&misc[M_UPT], cT);
                    if (cG) atomicAdd(&misc[M_UPG], cG);
                    if (cC) atomicAdd(&misc[M_UPC], cC);
                    if (nvalid) atomicAdd(&misc[M_NVALID], nvalid);
                }
            }
            STAMP(1)
            __syncthreads();
            STAMP(2)
            if (tid < FRISK_MISC_SLOTS) misc_other[tid] = 0;        // the previous window's counters: nobody reads them now
            auto code16_at = [&](int it) -> uint32_t {
                if (ITS > 0) return c16v[it];
                return fetch_codes16(P.codes, g0 + tid + int64_t(it) * NT);
            };
            auto zero_own_bins = [&]() {        // representatives zero their max-mer bin (K8; the small tables are
                if (K8) {                       // cleared wholesale)
#pragma unroll
                    for (int it = 0; ITS > 0 ? it < ITS : it * NT < n; ++it)
                        if ((repmask >> it) & 1ull) reinterpret_cast<uint16_t*>(t8)[code16_at(it)] = 0;
                }
            };
            auto clear_small = [&]() {
                for (uint32_t i = tid; i < L.small_bytes / 16; i += NT)
                    reinterpret_cast<uint4*>(small32)[i] = make_uint4(0, 0, 0, 0);
            };
            auto cleanup = [&]() { zero_own_bins(); clear_small(); };
            STOP_AFTER(0, misc[M_NVALID])

            // ---- stage 2: marginalise the small tables: C_x[q] = D_x[q] + sum_b C_{x+1}[4q+b] --------------
            auto marg_level = [&](int x, int first, int step) {
                const uint32_t ox = uint32_t(table_offset(kmin, x)), ou = uint32_t(table_offset(kmin, x + 1));
                for (uint32_t c = first; c < (1u << (2 * x)); c += step) {
                    const uint2 ch = *reinterpret_cast<const uint2*>(small16 + ou + 4 * c);     // 8-byte aligned
                    small16[ox + c] = uint16_t(small16[ox + c] + (ch.x & 0xFFFFu) + (ch.x >> 16) + (ch.y & 0xFFFFu) +
                                               (ch.y >> 16));
                }
            };
            if (ks >= 6 && kmin <= 4) {
                for (int x = ks - 1; x >= 6; --x) {             // K = 7: its order-6 level first, all waves
                    marg_level(x, tid, NT);
                    __syncthreads();
                }
                // from order 6 down (K = 8: the small tables are orders kmin..6), two phases instead of five dependent levels:
                //   A  thread q < 256 owns the 4-mer q: its sixteen 6-mer counts give the four C_5 and C_4[q] in one go;
                //   B  wave 0: lane l owns the 3-mer l, orders 2 and 1 follow inside the wave by DPP sums over quads and
                //      rows of 16 lanes - no LDS round trip between the levels.
                const uint32_t o6 = uint32_t(table_offset(kmin, 6)), o5 = uint32_t(table_offset(kmin, 5)),
                               o4 = uint32_t(table_offset(kmin, 4));
                if (tid < 256) {
                    uint32_t c5[4], c4 = small16[o4 + tid];
                    const uint2 d5 = *reinterpret_cast<const uint2*>(small16 + o5 + 4 * tid);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint2 ch = *reinterpret_cast<const uint2*>(small16 + o6 + 16 * tid + 4 * j);
                        const uint32_t d = (j == 0) ? (d5.x & 0xFFFFu) : (j == 1) ? (d5.x >> 16) : (j == 2) ? (d5.y & 0xFFFFu) : (d5.y >> 16);
                        c5[j] = d + (ch.x & 0xFFFFu) + (ch.x >> 16) + (ch.y & 0xFFFFu) + (ch.y >> 16);
                        c4 += c5[j];
                    }
                    *reinterpret_cast<uint2*>(small16 + o5 + 4 * tid) = make_uint2(c5[0] | (c5[1] << 16), c5[2] | (c5[3] << 16));
                    small16[o4 + tid] = uint16_t(c4);
                }
                __syncthreads();
                if (kmin <= 3) {
                    if (tid < 64) {
                        const uint32_t o3 = uint32_t(table_offset(kmin, 3));
                        const uint2 ch = *reinterpret_cast<const uint2*>(small16 + o4 + 4 * tid);
                        const uint32_t c3 = small16[o3 + tid] + (ch.x & 0xFFFFu) + (ch.x >> 16) + (ch.y & 0xFFFFu) + (ch.y >> 16);
                        small16[o3 + tid] = uint16_t(c3);
                        if (kmin <= 2) {
                            auto dppi = [](uint32_t x, auto ctrl) { return uint32_t(__builtin_amdgcn_update_dpp(0, int(x), decltype(ctrl)::value, 0xF, 0xF, false)); };
                            uint32_t q = c3 + dppi(c3, std::integral_constant<int, 0xB1>{});         // quad_perm [1,0,3,2]
                            q += dppi(q, std::integral_constant<int, 0x4E>{});                       // quad_perm [2,3,0,1]: the quad's sum
                            const uint32_t o2 = uint32_t(table_offset(kmin, 2));
                            uint32_t c2 = 0;
                            if ((tid & 3) == 0) { c2 = small16[o2 + (tid >> 2)] + q; small16[o2 + (tid >> 2)] = uint16_t(c2); }
                            if (kmin <= 1) {
                                // c2 sits in one lane of every quad: spread it over the quad, then fold the row's four quads
                                uint32_t row = c2 + dppi(c2, std::integral_constant<int, 0xB1>{});
                                row += dppi(row, std::integral_constant<int, 0x4E>{});
                                row += dppi(row, std::integral_constant<int, 0x141>{});              // row_half_mirror
                                row += dppi(row, std::integral_constant<int, 0x140>{});              // row_mirror: the four C_2 of the row
                                if ((tid & 15) == 0) small16[tid >> 4] = uint16_t(small16[tid >> 4] + row);
                            }
                        }
                    }
                    __syncthreads();
                }
            } else {
                int x = ks - 1;
#ifndef FRISK_MARG_WIDE
#define FRISK_MARG_WIDE 3
#endif
                for (; x >= kmin && x > FRISK_MARG_WIDE; --x) { // wide levels: all waves, one barrier each
                    marg_level(x, tid, NT);
                    __syncthreads();
                }
                if (x >= kmin) {                                // levels of <= 64 bins: wave 0 alone, in order
                    if (tid < 64) {
                        for (; x >= kmin; --x) {
                            marg_level(x, tid, 64);
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                    __syncthreads();
                }
            }
            STAMP(3)
            // window-uniform values read back from LDS are moved to scalar registers: they stay live to the end of the
            // window and must not cost a vector register (nor depend on which lanes a later loop leaves active)
            auto uni = [](uint32_t v) -> uint32_t { return __builtin_amdgcn_readfirstlane(v); };
            uint32_t upA = uni(misc[M_UPA]), upT = uni(misc[M_UPT]), upG = uni(misc[M_UPG]), upC = uni(misc[M_UPC]);
            if (!tally_by_ballot) {             // kmin == 1: the order-1 table starts the small tables
                upA = uni(small16[0]) - upA; upT = uni(small16[1]) - upT; upG = uni(small16[2]) - upG; upC = uni(small16[3]) - upC;
            }
            const int64_t S = int64_t(upA) + upT + upG + upC;       // windowSpace (L380)
            const int64_t nn = n - S;                               // nnTotal of the window
            // N filter (L237-241 / L213): dropped when nn >= 0.3 * len, evaluated in double like CPython
            const bool keep = !(double(nn) >= 0.3 * double(n));
            uint32_t status = (jump ? ROW_JUMPBACK : 0u);
            const uint32_t nvalid_top = uni(misc[M_NVALID]);
            const int n_orph = K8 ? int(uni(misc[M_NORPH])) : 0;
            if (tid == 0) {
                P.seq_index[row] = dsi;
                P.start[row] = rep_start;
                P.stop[row] = rep_stop;
            }
            if (!keep) {
                cleanup();
                if (tid == 0) {
                    P.status[row] = status;
                    const double qnan = __longlong_as_double(0x7FF8000000000000LL);
                    P.kld[row] = qnan; P.gc[row] = qnan;
                    if (P.flags & 1u) { P.pi[row] = qnan; P.si[row] = qnan; P.cri[row] = qnan; }
                }
                __syncthreads();
                continue;
            }
            STOP_AFTER(1, small16[tid & 3] + nvalid_top)

            STOP_AFTER(2, small16[tid & 3] + nvalid_top)

            WinTables<K8> T;
            T.t8_16 = reinterpret_cast<const uint16_t*>(t8);
            T.small16 = small16;
            T.orph = orph; T.n_orph = n_orph;
            T.o0 = T.o1 = T.o2 = T.o3 = 0xFFFFFFFFu;
            if (K8) {           // uniform loads (same address in every lane) made scalar
                if (n_orph > 0) T.o0 = __builtin_amdgcn_readfirstlane(uint32_t(orph[0]));
                if (n_orph > 1) T.o1 = __builtin_amdgcn_readfirstlane(uint32_t(orph[1]));
                if (n_orph > 2) T.o2 = __builtin_amdgcn_readfirstlane(uint32_t(orph[2]));
                if (n_orph > 3) T.o3 = __builtin_amdgcn_readfirstlane(uint32_t(orph[3]));
            }
            T.kmin = kmin;

            // ---- stage 3: window constants r_x = 4^x / D_x, D_x = (S-(x-1))*2 (L401-409), and the shared prefix ---
            // Lanes 0..8 of EVERY wave do the nine divisions; the other lanes read them by lane index (no barrier, no
            // LDS round trip).  Wave 0 also stores them for the paths that index r_x at run time.
            double r_lane = 0.0;
            if (lane <= 8) r_lane = div_exact(double(1u << (2 * lane)), double(int32_t((S - (lane - 1)) * 2)));    // (the IEEE quotient
            // for these operands; a zero divisor gives inf/NaN like the IEEE division, and the row is flagged anyway)
            if (tid <= 8) rtab[tid] = r_lane;
            auto r_of = [&](int x) -> double {                               // x: wave-uniform
                return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(r_lane), x),
                                        __builtin_amdgcn_readlane(__double2loint(r_lane), x));
            };
            // numerator / denominator of the closed form, orders kmin..lv, for every lv-mer (all orders of one entry are
            // fetched before any is used; orders below kmin get weight 0 and a clamped address)
            if (lv) {
                constexpr int LV = 5;                                       // == shared_level() whenever it is not 0
                double rx[LV + 1];
                uint32_t ox[LV + 1], wm[LV + 1];
#pragma unroll
                for (int x = 1; x <= LV; ++x) {
                    const bool on = x >= kmin;
                    rx[x] = on ? r_of(x) : 0.0;
                    ox[x] = on ? uint32_t(table_offset(kmin, x)) : 0u;
                    wm[x] = on ? 0xFFFFFFFFu : 0u;
                }
#pragma unroll
                for (int e = 0; e < (1 << (2 * LV)) / NT; ++e) {
                    const uint32_t c = tid + e * NT;
                    uint32_t cx[LV + 1];
#pragma unroll
                    for (int x = 1; x <= LV; ++x) cx[x] = small16[ox[x] + (c >> (2 * (LV - x)))];
                    uint32_t W = 0;
                    double A = 0.0;
#pragma unroll
                    for (int x = 1; x <= LV; ++x) {
                        const double cd = double(cx[x]);
                        W += (cx[x] & wm[x]) << (2 * x);                    // count * 4**x (L399-408)
                        A = __builtin_fma(cd * cd, rx[x], A);               // w_x * p_x = c^2 4^x / D_x
                    }
                    pre_i[c] = A;
                    pre_w[c] = W;                   // < 65536 * 4^6
                }
            }
            STAMP(4)
            __syncthreads();
            STAMP(5)
            STOP_AFTER(3, lv ? pre_i[tid & 1023] : rtab[1])

            if (DEBUG && P.dbg_counts) {
                uint32_t* out = P.dbg_counts + row * int64_t(P.nprof);
                for (int x = kmin; x <= kmax; ++x) {
                    const int64_t off = table_offset(kmin, x);
                    for (uint32_t c = tid; c < (1u << (2 * x)); c += NT) out[off + c] = T.count(x, c);
                }
            }
            if (DEBUG && P.dbg_meta && tid == 0) {
                P.dbg_meta[row * 3 + 0] = n;                                                   // totalLen
                P.dbg_meta[row * 3 + 1] = (n >= kmax ? n - kmax + 1 : 0) - int64_t(nvalid_top); // exMax (L344-345)
                P.dbg_meta[row * 3 + 2] = nn;                                                  // nnTotal
            }
            // Everything about the row that is already known is written NOW (composition, RIP indices, status bits):
            // nothing but the three sums has to stay live across stage 4.
            if (nvalid_top == 0) status |= ROW_NO_MAXMER;
            // a zero divisor on the window side (L401-409) needs windowSpace in [kmin-1, kmax-1]
            if (nvalid_top > 0 && S >= kmin - 1 && S <= kmax - 1) status |= ROW_ZERO_WEIGHT;
            status |= ROW_KEPT;
            if (tid == 0) {
                // composition: numerator and denominator only - the division (L136) is finish_rows_kernel's, off this
                // workgroup's critical path
                P.gc[row] = __longlong_as_double((long long)((uint64_t(uint32_t(S)) << 32) | uint32_t(upG + upC)));
                // RIP indices from the window's dinucleotide counts (L474-495); codes: AT=1 TA=4 TG=6 GT=9 CA=12 AC=3
                if (P.flags & 1u) {
                    const double qnan = __longlong_as_double(0x7FF8000000000000LL);
                    const uint32_t AT = T.count(2, 1), TA = T.count(2, 4), TG = T.count(2, 6), GT = T.count(2, 9),
                                   CA = T.count(2, 12), AC = T.count(2, 3);
                    const double pi = AT > 0 ? double(TA) / double(AT) : qnan;
                    const double si = (AC + GT) > 0 ? double(CA + TG) / double(AC + GT) : qnan;
                    P.pi[row] = pi;
                    P.si[row] = si;
                    P.cri[row] = (pi == 0.0 || si == 0.0) ? qnan : pi - si; // "if PI and SI" (L491): 0.0 is falsy
                }
            }

            // ---- stage 4: every max-mer position of the lane: window-side IVOM (closed form), genome side from the
            // table, and the three sums  Sw = sum Iw,  Sg = sum Ig,  T = sum Iw ln(Iw/Ig)  over representatives
            // plain_c: bound on the orphan list known to count(7) (1, 4, or 0 = none);  lv_c: the shared prefix table is in
            // use.  Both are window-uniform and resolved OUTSIDE the per-position loop so that its unrolled body is one
            // basic block.
            double r_hi[3];                                                 // r_6, r_7, r_8: the orders above the shared prefix
#pragma unroll
            for (int x = 6; x <= 8; ++x) r_hi[x - 6] = r_of(x);
            auto window_ivom = [&](const auto& Tm, uint32_t code, auto plain_c, auto lv_c, double& A_out) __attribute__((always_inline)) -> double {
                constexpr int PLAIN = decltype(plain_c)::value;
                wsum_t W = 0;
                double A = 0.0;
                if constexpr (decltype(lv_c)::value) {
                    constexpr int LV = 5;                                   // == shared_level() whenever it is not 0
                    const uint32_t pc = code >> (2 * (kmax - LV));
                    W = pre_w[pc];
                    A = pre_i[pc];
                    if constexpr (K8) {
                        // orders 6, 7, 8 with ONE read of the max-mer table: the max-mer's own count is one of the four
                        // children summed for its 7-mer prefix
                        const uint32_t c6 = Tm.template count<PLAIN>(6, code >> 4);
                        uint32_t c8, c7;
                        Tm.top(code, c8, c7);
                        c7 += Tm.template orphans<PLAIN>(code >> 2);
                        const double d6 = double(c6), d7 = double(c7), d8 = double(c8);
                        W += (wsum_t(c6) << 12) + (wsum_t(c7) << 14) + (wsum_t(c8) << 16);
                        A = __builtin_fma(d6 * d6, r_hi[0], A);
                        A = __builtin_fma(d7 * d7, r_hi[1], A);
                        A = __builtin_fma(d8 * d8, r_hi[2], A);
                    } else
#pragma unroll
                    for (int x = LV + 1; x <= kmax; ++x) {
                        const uint32_t cx = Tm.template count<PLAIN>(x, code >> (2 * (kmax - x)));
                        const double cd = double(cx);
                        W += wsum_t(cx) << (2 * x);
                        A = __builtin_fma(cd * cd, K8 ? r_hi[x - LV - 1] : rtab[x], A);    // K8: x is compile-time
                    }
                } else {
                    for (int x = kmin; x <= kmax; ++x) {
                        const uint32_t cx = Tm.template count<PLAIN>(x, code >> (2 * (kmax - x)));
                        const double cd = double(cx);
                        W += wsum_t(cx) << (2 * x);
                        A = __builtin_fma(cd * cd, rtab[x], A);
                    }
                }
                A_out = A;
                return double(W);
            };
            ExactSum accw = exact_begin(), accg = exact_begin(), acct = exact_begin();
            // A lane that is not a representative must add exactly nothing.  Clearing only the HIGH word of its term
            // leaves a subnormal (< 2^-1022), which both roundings of exact_add() turn into 0 - one select per term.
            // Representatives are never screened: a max-mer without genome weight (Ig = NaN, L437) makes Sg NaN, and
            // that is how the row's ZeroDivisionError is recognised below.
            auto only_rep = [](bool rep, double x) -> double {
                return __hiloint2double(rep ? __double2hiint(x) : 0, __double2loint(x));
            };
            // one max-mer: genome side gathered, window side from the tables, three exact additions
            auto score_one = [&](const auto& Tm, uint32_t code, bool rep, auto plain_c, auto lv_c) __attribute__((always_inline)) {
#ifndef FRISK_ABL
#define FRISK_ABL 0
#endif
                // (FRISK_ABL: diagnostic builds that drop one ingredient at a time - tools/ablate.py; results wrong by design)
                const double Ig = (FRISK_ABL & 8) ? 1e-3 + 1e-9 * double(code) : P.ig[code];   // unconditional gather (code < 4^K always)
                // Iw = A/W and Iw/Ig with ONE division: ratio = A / (W * Ig), Iw = ratio * Ig
                double A;
                double Wd;
                if (FRISK_ABL & 16) { A = 1e-4 * double(code & 1023u); Wd = double(code | 1u); }
                else Wd = window_ivom(Tm, code, plain_c, lv_c, A);
                const double ratio = (FRISK_ABL & 2) ? A * (Wd * Ig) : div_exact(A, Wd * Ig);
                const double Iw = ratio * Ig;
                if (DEBUG && P.dbg_ivom && rep) {       // IvomBuild's per-max-mer values before normalisation (L442-450)
                    double* o = P.dbg_ivom + (row * 2) * (int64_t(1) << (2 * kmax));
                    o[code] = Iw;
                    o[(int64_t(1) << (2 * kmax)) + code] = Ig;
                }
                // Iw ln(Iw/Ig): the log of the RATIO (|ln| ~ 1) keeps the absolute error of T at the 1e-16 level
#ifdef FRISK_LOG_FDLIBM
                const double t = Iw * log_pos(ratio);
#else
                const double t = (FRISK_ABL & 1) ? Iw * ratio : Iw * log_tab_pos(ratio, logtab);
#endif
                if (FRISK_ABL & 4) {
                    accw.hi += only_rep(rep, Iw); accg.hi += only_rep(rep, Ig); acct.hi += only_rep(rep, t);
                } else {
                    exact_add(accw, only_rep(rep, Iw));
                    exact_add(accg, only_rep(rep, Ig));
                    exact_add(acct, only_rep(rep, t));
                }
            };
            // window-uniform decisions resolved OUTSIDE the scoring loops: bound on the orphan list, shared prefix in use
            using orph1 = std::integral_constant<int, 1>;
            using orph4 = std::integral_constant<int, 4>;
            using orphN = std::integral_constant<int, 0>;
            auto with_variant = [&](auto&& body) __attribute__((always_inline)) {
                if (lv) {
                    if (n_orph <= 1) body(orph1{}, std::true_type{});
                    else if (n_orph <= 4) body(orph4{}, std::true_type{});
                    else body(orphN{}, std::true_type{});
                } else {
                    if (n_orph <= 1) body(orph1{}, std::false_type{});
                    else if (n_orph <= 4) body(orph4{}, std::false_type{});
                    else body(orphN{}, std::false_type{});
                }
            };
                with_variant([&](auto plain_c, auto lv_c) {
#pragma unroll
                    for (int it = 0; ITS > 0 ? it < ITS : it * NT < n; ++it) {
                        // (iterations past the window are not skipped: their lanes are clamped and masked, and a
                        //  branch here would split the unrolled body into blocks the scheduler cannot interleave)
                        const bool rep = (repmask >> it) & 1ull;
                        if (ITS == 0 && !rep) continue;
                        score_one(T, code16_at(it) >> kshift, rep, plain_c, lv_c);
#ifndef FRISK_S4_GROUP
#define FRISK_S4_GROUP 2
#endif
                        // interleave at most FRISK_S4_GROUP positions: more overlap needs more live registers than the
                        // 128 a 1024-thread workgroup leaves per lane, and the scheduler would spill
                        if ((it % FRISK_S4_GROUP) == FRISK_S4_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
                    }
                });
            STAMP(6)
            exact_end(accw); exact_end(accg); exact_end(acct);
#ifdef FRISK_STOP
            block_sum3<NW>(accw, accg, acct, scratch_base, tid);
#else
            block_sum3<NW, true>(accw, accg, acct, scratch_base, tid);
#endif
            STAMP(7)
            zero_own_bins();                                // behind the barrier: nobody reads the max-mer table any more
            clear_small();                                  // all reads of the small tables are behind the barrier
#ifdef FRISK_STOP
            { const double Sw = exact_value(accw), Sg = exact_value(accg), Tt = exact_value(acct); STOP_AFTER(4, Sw + Sg + Tt) }
#endif
            // the scalar tail of the row - two logarithms and a division, ~2 000 cycles of dependent instructions that every
            // other wave would wait for at the window's last barrier - is left to finish_rows_kernel (all rows in parallel)
            if (tid == 0) {
                P.status[row] = status;
                P.kld[row] = exact_value(acct);
                P.sw[row] = exact_value(accw);
                P.sg[row] = exact_value(accg);
            }
            STAMP(8)
            __syncthreads();
            STAMP(9)
        }
    }
}

// Per-row scalar tail of scan_kernel: KLD = sum Pw log2(Pw/Pg) = (T/Sw - ln Sw + ln Sg) / ln 2 (L453-454, L465-470), the GC
// fraction (L136) and the ZeroDivisionError flag of a max-mer without genome weight (L437).  Rows that were dropped by the
// N filter keep their NaNs.
__global__ __launch_bounds__(256) void finish_rows_kernel(int64_t n, uint32_t* __restrict__ status, double* __restrict__ kld,
                                                           double* __restrict__ gc, const double* __restrict__ sw,
                                                           const double* __restrict__ sg) {
    for (int64_t row = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; row < n; row += int64_t(gridDim.x) * blockDim.x) {
        const uint32_t st = status[row];
        if (!(st & ROW_KEPT)) continue;
        const uint64_t packed = uint64_t(__double_as_longlong(gc[row]));
        gc[row] = double(uint32_t(packed)) / double(int64_t(packed >> 32));
        if (st & ROW_NO_MAXMER) { kld[row] = 0.0; continue; }
        const double Tt = kld[row], Sw = sw[row], Sg = sg[row];
        const double LN2 = 0.69314718055994530942;
        kld[row] = ((Tt / Sw - log(Sw)) + log(Sg)) / LN2;
        if (Sg != Sg) status[row] = st | ROW_ZERO_WEIGHT;
    }
}

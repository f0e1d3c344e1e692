// scan8_kernel.h - phase B at K = 6, 7, 8 with SEVERAL independent workgroups per CU (the default path for these orders).
//
// Same per-window computation as scan_kernel.h (reference frisk/__init__.py L1478-1494: crawlGenome L194-251 ->
// computeKmers(window) L280-367 -> IvomBuild x2 L369-457 -> KLD L459-472 -> calcGC L120-137 [-> calcRIP L474-495]),
// different data structure.  scan_kernel.h keeps the order-8 histogram as 4^8 16-bit counters = 128 KiB, so ONE
// 512-thread workgroup owns a CU and its two waves per SIMD run the same stage between the same barriers: they want
// the LDS pipeline at the same time and the VALU at the same time (r1 profile: VALU < 50 % busy, 52 % of wave cycles
// waiting).  Here the order-K table is NARROW - BITS = 8 (64 KiB at K = 8) or 4 (32 KiB) per counter - so that two /
// three (K = 8) or four (K = 6, 7) 256-thread workgroups share a CU, each on its own window and in its own stage:
//   * stage 1   ONE non-returning ds_add per max-mer position (field BITS * (code & 3|7) of dword code >> 2|3).  Nothing
//               else is counted there: no lower-order update, no election of representatives.
//   * stage 2   the (K-3)-mer counts are the sums of 64 neighbouring counters: every thread sums whole 16-byte reads
//               (v_sad_u8 / v_dot8_u32_u4), the read order rotated per lane (conflict-free); the orders below follow
//               inside a wave by DPP sums.  The grand total of the table must equal the number of max-mer positions;
//               a counter that wrapped (a max-mer occurring >= 2^BITS times: poly-A, microsatellites) breaks that
//               equality, and the window is handed to the next wider form through a device-side list: 4-bit ->
//               8-bit -> scan_kernel.h's 16-bit form (later launches on the same stream).
//   * stage 4   c_K, c_{K-1} = sum of 4 children, c_{K-2} = sum of 16 children come from the counter's own aligned
//               neighbourhood (8-bit: u8 + b32 + b128 reads at the code with low bits cleared; 4-bit: u16 + b64);
//               the few (K-1)- and (K-2)-mers that are not prefixes of a max-mer (window tail, next to invalid
//               bases) sit in a short "orphan" list held in scalar registers.
//   * sums      every max-mer POSITION adds its max-mer's terms with weight 1/c_K (c_K positions share a max-mer), so
//               a lane's set of terms is fixed by the window alone: plain FP64 sums in a fixed order are
//               bit-reproducible across runs, grids, candidate ranges and ranks, without the exact (double-pair)
//               summation that the election of representatives by atomic arrival order forced on scan_kernel.h (6 FP64
//               instructions per term there, 1 multiply + 1 add here).  Positions that start no max-mer score a
//               stand-in (a real max-mer of the window) with weight 0: no masks.
// Windows up to NT*ITS bases, kmin <= K-3 (the shared prefix level); everything else stays on scan_kernel.h.
// Measurements, the adaptive choice between 4 and 8 bits, and what was tried and dropped: DESIGN.md section 3.3.
#pragma once
#include "scan_kernel.h"

#define FRISK8_ORPH_CAP 24         // orphan entries kept in LDS (two per invalid run); a window with more goes to the 16-bit form
#ifndef FRISK8_UNROLL1
#define FRISK8_UNROLL1 2           // unroll factor of the stage-1 position loop
#endif
#define FRISK8_SLOTS 8             // misc counters per window (double-buffered by window parity)

enum { M8_TSUM = 6,                // misc slots: grand total of the order-8 table (overflow check) ...
       M8_SAFE = 7 };              // ... and the code of SOME max-mer of the window (what lanes without one score instead)

// LDS carve-up, all compile-time: the kernels declare it as ONE static array, so every table address is a constant that
// folds into the 16-bit offset field of the ds_ instructions (a dynamic `extern __shared__` base costs one VALU add per
// address).  The 32 / 64 KiB order-8 table comes LAST: its own offset is then the only large one, and it is an immediate.
template <int KMAX, int BITS, int LOGN, int NT>
struct Lds8 {
    static constexpr uint32_t small = 0;
    static constexpr uint32_t small_bytes = 2736;                                  // orders kmin..KMAX-3 as u16 bins (sized for 1..5)
    static constexpr uint32_t orphans = small + small_bytes;                       // u16[FRISK8_ORPH_CAP]
    static constexpr uint32_t NL = 1u << (2 * (KMAX - 3));                         // entries of the shared prefix tables (level KMAX-3)
    static constexpr uint32_t pre_i = (orphans + FRISK8_ORPH_CAP * 2 + 15) / 16 * 16;   // f64[NL]: shared prefix sums ...
    static constexpr uint32_t pre_w = pre_i + NL * 8;                              // ... and u32[NL], as in scan_kernel.h
    static constexpr uint32_t logtab = pre_w + NL * 4;                             // {1/c_i, -ln(1/c_i)} x LOGN
    static constexpr uint32_t rctab = logtab + uint32_t(LOGN) * 16;                // 1/c for c < 16
    static constexpr uint32_t misc = rctab + 16 * 8;                               // counters x2, then one {Sw, Sg, T} per wave
    static constexpr uint32_t t8 = (misc + 2 * FRISK8_SLOTS * 4 + uint32_t(NT / 64) * 3 * 8 + 15) / 16 * 16;
    static constexpr uint32_t t8_bytes = (1u << (2 * KMAX)) * BITS / 8;
    static constexpr uint32_t total = t8 + t8_bytes;
};

// ln(x), x positive and normal, by table range reduction as scan_kernel.h's log_tab_pos: x = m 2^k, m in [0.5, 1); the top
// log2(N) mantissa bits pick the bin, tab[i] = {u_i = 1/c_i rounded, -ln u_i}, r = m u_i - 1 exactly (one fma),
// ln x = k ln2 - ln u_i + log1p(r) with log1p by its Taylor polynomial of degree DEG.  Absolute error < 1e-15 for the ratios
// of probabilities scored here (|k| small): what the sum T = sum Iw ln(Iw/Ig) needs.  DEG + 5 instructions.
template <int N, int DEG>
__device__ inline double log_tab_n(double x, const double2* tab) {
    const int k = __builtin_amdgcn_frexp_exp(x);
    const double m = __builtin_amdgcn_frexp_mant(x);
    const uint32_t i = (uint32_t(__double2hiint(m)) >> (N == 128 ? 13 : (N == 64 ? 14 : 15))) & uint32_t(N - 1);
    const double2 e = tab[i];
    const double r = __builtin_fma(m, e.x, -1.0);
    double p = (DEG & 1) ? 1.0 / DEG : -1.0 / DEG;
#pragma unroll
    for (int d = DEG - 1; d >= 2; --d) p = __builtin_fma(r, p, (d & 1) ? 1.0 / d : -1.0 / d);
    return __builtin_fma(double(k), 0.69314718055994530942, e.y) + __builtin_fma(r * r, p, r);
}

template <int CTRL>
__device__ inline uint32_t dpp_addu(uint32_t x) {
    return x + uint32_t(__builtin_amdgcn_update_dpp(0, int(x), CTRL, 0xF, 0xF, false));
}
// sum over the wave, valid in every lane (same butterfly as wave_sum_exact)
__device__ inline uint32_t wave_sum_u32(uint32_t x) {
    x = dpp_addu<0xB1>(x); x = dpp_addu<0x4E>(x); x = dpp_addu<0x141>(x); x = dpp_addu<0x140>(x);
    return __builtin_amdgcn_readlane(int(x), 0) + __builtin_amdgcn_readlane(int(x), 16) +
           __builtin_amdgcn_readlane(int(x), 32) + __builtin_amdgcn_readlane(int(x), 48);
}

// ROLE only names the launch (0 bulk / list, 1 the sample of the adaptive width): the code is the same, but a profiler's
// per-kernel statistics then keep the 1/16 sample launches apart from the bulk launches.
// NT threads, windows of at most NT*ITS bases, BITS per order-8 counter, LOGN: bins of the logarithm table, WPS: waves per SIMD the register allocation must allow (= workgroups per
// CU * NT / 256).
template <int KMAX, int NT, int ITS, int BITS, int LOGN, int WPS, bool DEBUG, int ROLE = 0>
__global__ __launch_bounds__(NT, WPS) void scan8_kernel(const ScanParams P) {
    static_assert(KMAX >= 6 && KMAX <= 8, "highest order 6, 7 or 8");
    constexpr int K = KMAX, LVL = KMAX - 3;              // highest order; level of the shared prefix tables (and of the small tables' top)
    constexpr uint32_t NK = 1u << (2 * K), NL = 1u << (2 * LVL);
    static_assert(BITS == 4 || BITS == 8, "order-8 counters are 4 or 8 bits wide");
    static_assert(LOGN == 32 || LOGN == 64 || LOGN == 128, "logarithm table of 32, 64 or 128 bins");
    constexpr int LOGDEG = LOGN == 128 ? 5 : (LOGN == 64 ? 6 : 7);   // |r| < 2^-8 / 2^-7 / 2^-6: truncation r^(DEG+1)/(DEG+1) < 6e-16
    static_assert(ITS + 7 <= 32, "a lane's positions and their max-mers must fit the 32 bases it loads");
    constexpr int NW = NT / 64;
    constexpr int SHW = BITS == 8 ? 2 : 3;               // code >> SHW = dword of the table
    constexpr uint32_t PERM = (32 / BITS) - 1;           // code & PERM = field inside the dword
    using L = Lds8<KMAX, BITS, LOGN, NT>;
    __shared__ __attribute__((aligned(16))) unsigned char lds[L::total];
    const int tid0 = threadIdx.x;
    const int kmin0 = P.kmin;
    uint32_t* t8 = reinterpret_cast<uint32_t*>(lds + L::t8);
    const unsigned char* t8b = lds + L::t8;
    uint32_t* small32 = reinterpret_cast<uint32_t*>(lds + L::small);
    uint16_t* small16 = reinterpret_cast<uint16_t*>(lds + L::small);
    uint16_t* orph = reinterpret_cast<uint16_t*>(lds + L::orphans);
    double* pre_i = reinterpret_cast<double*>(lds + L::pre_i);
    uint32_t* pre_w = reinterpret_cast<uint32_t*>(lds + L::pre_w);
    uint32_t* misc_base = reinterpret_cast<uint32_t*>(lds + L::misc);
    double* scratch = reinterpret_cast<double*>(lds + L::misc + 2 * FRISK8_SLOTS * 4);
    const double2* logtab = reinterpret_cast<const double2*>(lds + L::logtab);
    const double* rctab = reinterpret_cast<const double*>(lds + L::rctab);

    auto clear_t8 = [&]() {
        for (int i = tid0; i < int(L::t8_bytes / 16); i += NT) reinterpret_cast<uint4*>(t8)[i] = make_uint4(0, 0, 0, 0);
    };
    auto clear_small = [&]() {
        for (uint32_t i = tid0; i < L::small_bytes / 16; i += NT) reinterpret_cast<uint4*>(small32)[i] = make_uint4(0, 0, 0, 0);
    };
    clear_t8();
    clear_small();
    if (tid0 < 2 * FRISK8_SLOTS) misc_base[tid0] = 0;
    {
        double2* lt = reinterpret_cast<double2*>(lds + L::logtab);
        for (int i = tid0; i < LOGN; i += NT) lt[i] = reinterpret_cast<const double2*>(LOGN == 128 ? P.log_tab : (LOGN == 64 ? P.log_tab64 : P.log_tab32))[i];
        if (tid0 < 16) reinterpret_cast<double*>(lds + L::rctab)[tid0] = P.rc_tab[tid0];
    }
    __syncthreads();

    // XCD-aware work split (as scan_kernel.h): blocks b and b+8 share an XCD, neighbouring chunks share an L2
    const int G = gridDim.x;
    int v = blockIdx.x;
    if ((G & 7) == 0) v = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    // Which candidates: the chunks of [c0, c1) - all of them (sel_mode 0), every sel_mod-th (1: the sample that decides the
    // counter width for the rest), all but those (2) - or, one at a time, the windows a narrower form handed over (in_list).
    const bool listed = P.in_list != nullptr;
    const int64_t chunk = listed ? 1 : P.chunk;
    const int64_t nall = listed ? int64_t(*P.in_count) : (P.c1 - P.c0 + chunk - 1) / chunk;
    const int64_t M = P.sel_mod;
    const int64_t nsample = (nall + M - 1) / M;
    const int64_t nchunks = listed || P.sel_mode == 0 ? nall : (P.sel_mode == 1 ? nsample : nall - nsample);

    ScafDesc d;
    d.cand0 = 0; d.ncand = 0; d.off = 0; d.size = 0; d.kind = 0; d.base0 = 0; d.j0 = 0;
    int dsi = -1;
    uint32_t parity = 0;

    for (int64_t q = v; q < nchunks; q += G) {
        int64_t qq = q;                                                  // chunk index inside [c0, c1)
        if (!listed && P.sel_mode == 1) qq = q * M;
        if (!listed && P.sel_mode == 2) qq = (q / (M - 1)) * M + 1 + q % (M - 1);
        const int64_t cb = listed ? q : P.c0 + qq * chunk;
        const int64_t ce = listed ? q + 1 : ((cb + chunk < P.c1) ? cb + chunk : P.c1);
        for (int64_t ci = cb; ci < ce; ++ci) {
            const int64_t cand = listed ? P.in_list[ci] : ci;
            // ---- which scaffold / window is this candidate? (uniform; crawlGenome L194-251)
            if (cand < d.cand0 || cand >= d.cand0 + d.ncand) {
                int lo = 0, hi = P.n_desc - 1;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (P.descs[mid].cand0 <= cand) lo = mid; else hi = mid - 1;
                }
                d = P.descs[lo];
                dsi = lo;
            }
            // The thread index and the lowest order, opaque to the optimiser from here on: otherwise it hoists every
            // per-position constant that depends on them (20 x {tid*20+it, masks, table offsets}: > 100 registers and
            // dozens of spilled scalars) out of the window loop and keeps them alive across all stages.
            int tid = tid0, kmin = kmin0;
            asm volatile("" : "+v"(tid), "+s"(kmin));
            const int lane = tid & 63;
            const int64_t j = cand - d.cand0 + d.j0;           // window index inside the scaffold
            int64_t st, rep_start, rep_stop;
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
                    if (st < 0) { st += d.size; if (st < 0) st = 0; }                           // negative slice start
                    n = int(d.size - st);
                }
            }
            const int64_t g0 = d.off + (st - d.base0);            // resident position of the window's first base
            const int64_t row = cand - P.c0;
            if (n > NT * ITS) {
                // a rescued small scaffold (--scaffoldsAll, L211-221: up to 1.75 w bases) longer than this kernel's lanes cover:
                // straight to the wider forms - per WINDOW, so that which kernel scores a window never depends on what else
                // is resident (a rank of a multi-GPU job sees other scaffolds than the one-GPU run)
                if (tid == 0) { const unsigned int slot = atomicAdd(P.out_count, 1u); P.out_list[slot] = cand; }
                continue;
            }
            uint32_t* misc = misc_base + parity * FRISK8_SLOTS;
            uint32_t* misc_other = misc_base + (parity ^ 1u) * FRISK8_SLOTS;
            parity ^= 1u;

            // ---- stage 1: one pass over the window's positions (a lane owns ITS consecutive ones) -----------
            const bool tally_by_ballot = (kmin != 1);
            const int j0 = tid * ITS;
            const int64_t gl = g0 + (j0 < n ? j0 : 0);                       // clamped: loads are unconditional
            const int64_t wi = gl >> 4, mi = gl >> 5;
            const int shc = 32 - int(gl & 15) * 2, shm = 32 - int(gl & 31);
            const uint32_t w0 = P.codes[wi], w1 = P.codes[wi + 1], w2 = P.codes[wi + 2];
            const uint32_t chi = uint32_t(((uint64_t(w0) << 32) | w1) >> shc);
            const uint32_t clo = uint32_t(((uint64_t(w1) << 32) | w2) >> shc);
            const uint64_t acode = (uint64_t(chi) << 32) | clo;             // bases j0 .. j0+31, first base in the top bits
            const uint32_t ainv = uint32_t(((uint64_t(P.inv[mi]) << 32) | P.inv[mi + 1]) >> shm);
            const uint32_t alow = uint32_t(((uint64_t(P.low[mi]) << 32) | P.low[mi + 1]) >> shm);
            auto topbits = [](int k) -> uint32_t {
                k = k < 0 ? 0 : (k > 32 ? 32 : k);
                return uint32_t(0xFFFFFFFF00000000ull >> k);
            };
            constexpr uint32_t MINE = uint32_t(0xFFFFFFFF00000000ull >> ITS);
            const int nleft = n - j0;
            const uint32_t actm = topbits(nleft) & MINE;
            const uint32_t vld = ~ainv;
            uint32_t fullm = vld;                                            // K valid bases from here on ...
            fullm &= fullm << 1; fullm &= fullm << 2;                        // (4 in a row)
            fullm &= fullm << (K - 4);                                       // (K = 6, 7, 8 in a row)
            fullm &= topbits(nleft - (K - 1)) & MINE;                        // ... all inside the window: a max-mer starts here
            auto code_at = [&](int it) -> uint32_t { return uint32_t(acode >> (64 - 2 * K - 2 * it)) & (NK - 1u); };     // the K-mer at position it
            {
                uint32_t cAll = 0, cGC = 0, nvalid = 0;
                // The composition the row needs (calcGC L120-137, countN L106-118) is two numbers: how many bases are uppercase
                // A/T/G/C, and how many of those are G or C - the HIGH bit of the 2-bit code (A=0,T=1,G=2,C=3).  Both are
                // popcounts over the lane's positions once the codes' high bits are gathered into a plane (bit 31-it <->
                // position it, like the validity masks): no per-position loop, no ballots.  (Real assemblies are soft-masked
                // over half their length: the per-position ballots this replaces ran for almost every wave there.)
                auto gc_plane = [&]() -> uint32_t {
                    auto squeeze = [](uint32_t w) -> uint32_t {      // the odd bits 31, 29, ..., 1 of w -> bits 15..0
                        uint32_t x = (w >> 1) & 0x55555555u;
                        x = (x | (x >> 1)) & 0x33333333u;
                        x = (x | (x >> 2)) & 0x0F0F0F0Fu;
                        x = (x | (x >> 4)) & 0x00FF00FFu;
                        x = (x | (x >> 8)) & 0x0000FFFFu;
                        return x;
                    };
                    return (squeeze(uint32_t(acode >> 32)) << 16) | squeeze(uint32_t(acode));
                };
                auto tally = [&](uint32_t sel) {                     // sel: the lane's positions to count
                    cAll = wave_sum_u32(uint32_t(__popc(sel)));
                    cGC = wave_sum_u32(uint32_t(__popc(sel & gc_plane())));
                };
#pragma unroll FRISK8_UNROLL1
                for (int it = 0; it < ITS; ++it) {
                    const uint32_t bit = 0x80000000u >> it;
                    const uint32_t c16 = code_at(it);
                    if (fullm & bit) {
                        atomicAdd(&t8[c16 >> SHW], 1u << ((c16 & PERM) * BITS));
                    } else if (actm & bit) {
                        // next to an invalid base or the window's end: the longest valid word here has 0..7 bases.
                        // Orders <= 5 count it in the small tables (at order min(run, 5): lower orders follow by
                        // marginalisation); a 6- or 7-base word is not a prefix of any counted max-mer: orphan list.
                        int run = lead_clear8((ainv >> (24 - it)) & 0xFFu);
                        const int rem = n - (j0 + it);
                        run = run < rem ? run : rem;
                        run = run < K ? run : K - 1;                         // (a K-mer inside the window would have been a max-mer)
                        const int rs = run < LVL ? run : LVL;
                        if (rs >= kmin) {
                            const uint32_t b = uint32_t(table_offset(kmin, rs)) + (c16 >> (2 * K - 2 * rs));
                            atomicAdd(&small32[b >> 1], 1u << ((b & 1u) * 16));
                        }
                        if (run >= K - 2) {
                            const uint32_t slot = atomicAdd(&misc[M_NORPH], 1u);
                            if (slot < FRISK8_ORPH_CAP)
                                orph[slot] = uint16_t(run == K - 1 ? (c16 >> 2) : (0x8000u | ((c16 >> 4) << 2)));
                        }
                    }
                }
                const uint32_t ntop = __popc(fullm);
#pragma unroll
                for (int b = 0; (1 << b) <= ITS; ++b) nvalid += uint32_t(__popcll(__ballot((ntop >> b) & 1u))) << b;
                if (tally_by_ballot) {              // kmin > 1: no order-1 table: count the uppercase bases directly
                    tally(actm & vld & ~alow);
                } else {                            // kmin = 1: the order-1 table counts ALL valid bases; subtract the soft-masked
                    const uint32_t lowm = actm & vld & alow;
                    if (__ballot(lowm != 0)) tally(lowm);
                }
                {   // the code of one max-mer of this window, any: positions that start none score it with weight 0, so that
                    // every lane computes finite values and no term needs masking (which wave's wins does not matter)
                    const unsigned long long have = __ballot(fullm != 0u);
                    if (have) {
                        const uint32_t mine = uint32_t(acode >> (64 - 2 * K - 2 * int(__clz(int(fullm | 1u))))) & (NK - 1u);
                        const uint32_t pick = uint32_t(__builtin_amdgcn_readlane(int(mine), int(__ffsll((long long)have)) - 1));
                        if (lane == 0) misc[M8_SAFE] = pick;
                    }
                }
                if (lane == 0) {
                    if (cAll) atomicAdd(&misc[M_UPA], cAll);        // (M_UPA: all four bases, M_UPG: G + C)
                    if (cGC) atomicAdd(&misc[M_UPG], cGC);
                    if (nvalid) atomicAdd(&misc[M_NVALID], nvalid);
                }
            }
            __syncthreads();
            if (tid < FRISK8_SLOTS) misc_other[tid] = 0;        // the previous window's counters: nobody reads them now

            // ---- stage 2: C_5[q] = D_5[q] + (sum of the 64 order-8 counters below q); grand total for the overflow check
            const uint32_t o5 = uint32_t(table_offset(kmin, LVL));         // the small tables' top order: K-3
            {
                uint32_t tot = 0;
                for (uint32_t q5 = tid; q5 < NL; q5 += NT) {
                    uint32_t s = 0;
                    if (BITS == 8) {
#pragma unroll
                        for (int m = 0; m < 4; ++m) {               // 64 bytes; the read order is rotated per lane: conflict-free
                            const uint32_t mm = (uint32_t(m) + (uint32_t(tid) >> 2)) & 3u;
                            const uint4 x = *reinterpret_cast<const uint4*>(t8b + q5 * 64u + mm * 16u);
                            s = __builtin_amdgcn_sad_u8(x.x, 0u, s); s = __builtin_amdgcn_sad_u8(x.y, 0u, s);
                            s = __builtin_amdgcn_sad_u8(x.z, 0u, s); s = __builtin_amdgcn_sad_u8(x.w, 0u, s);
                        }
                    } else {
#pragma unroll
                        for (int m = 0; m < 2; ++m) {               // 32 bytes
                            const uint32_t mm = (uint32_t(m) + (uint32_t(tid) >> 3)) & 1u;
                            const uint4 x = *reinterpret_cast<const uint4*>(t8b + q5 * 32u + mm * 16u);
                            s = __builtin_amdgcn_udot8(x.x, 0x11111111u, s, false); s = __builtin_amdgcn_udot8(x.y, 0x11111111u, s, false);
                            s = __builtin_amdgcn_udot8(x.z, 0x11111111u, s, false); s = __builtin_amdgcn_udot8(x.w, 0x11111111u, s, false);
                        }
                    }
                    tot += s;
                    small16[o5 + q5] = uint16_t(small16[o5 + q5] + s);
                }
                tot = wave_sum_u32(tot);
                if (lane == 0 && tot) atomicAdd(&misc[M8_TSUM], tot);
            }
            __syncthreads();
            // the orders below: inside a wave, no LDS round trip between the levels.  LVL >= 4: lane l of wave i takes the 4-mer
            // l + 64 i (its count comes from its four 5-mers, or is final already when LVL = 4); 3-mers are sums over quads,
            // 2-mers over rows of 16 lanes, the 1-mer i over the wave; the four quarters of the 4-mer space are independent
            // (one wave each; a workgroup of fewer waves loops).  LVL = 3: one wave, lane l = the 3-mer l.
            if constexpr (LVL >= 4) {
                if (tid < 256 && kmin <= 4 && (LVL == 5 || kmin <= 3)) {
                    const uint32_t o4 = uint32_t(table_offset(kmin, 4));
                    for (int i = tid >> 6; i < 4; i += (NT >= 256 ? 4 : NT / 64)) {
                        const uint32_t q4 = uint32_t(tid & 63) + 64u * i;
                        uint32_t c4 = small16[o4 + q4];
                        if constexpr (LVL == 5) {
                            const uint2 ch = *reinterpret_cast<const uint2*>(small16 + o5 + 4 * q4);
                            c4 += (ch.x & 0xFFFFu) + (ch.x >> 16) + (ch.y & 0xFFFFu) + (ch.y >> 16);
                            small16[o4 + q4] = uint16_t(c4);
                        }
                        if (kmin <= 3) {
                            uint32_t qs = dpp_addu<0xB1>(c4);
                            qs = dpp_addu<0x4E>(qs);                                             // the quad's sum, in all four lanes
                            const uint32_t o3 = uint32_t(table_offset(kmin, 3));
                            uint32_t c3 = 0;
                            if ((lane & 3) == 0) { c3 = small16[o3 + (lane >> 2) + 16 * i] + qs; small16[o3 + (lane >> 2) + 16 * i] = uint16_t(c3); }
                            if (kmin <= 2) {
                                uint32_t rs = dpp_addu<0xB1>(c3);
                                rs = dpp_addu<0x4E>(rs); rs = dpp_addu<0x141>(rs); rs = dpp_addu<0x140>(rs);   // the row's four C_3
                                const uint32_t o2 = uint32_t(table_offset(kmin, 2));
                                uint32_t c2 = 0;
                                if ((lane & 15) == 0) { c2 = small16[o2 + (lane >> 4) + 4 * i] + rs; small16[o2 + (lane >> 4) + 4 * i] = uint16_t(c2); }
                                if (kmin <= 1) {
                                    const uint32_t ws = __builtin_amdgcn_readlane(int(c2), 0) + __builtin_amdgcn_readlane(int(c2), 16) +
                                                        __builtin_amdgcn_readlane(int(c2), 32) + __builtin_amdgcn_readlane(int(c2), 48);
                                    if (lane == 0) small16[i] = uint16_t(small16[i] + ws);
                                }
                            }
                        }
                    }
                }
            } else {
                if (tid < 64 && kmin <= 2) {
                    const uint32_t c3 = small16[uint32_t(table_offset(kmin, 3)) + uint32_t(tid)];     // final already
                    uint32_t qs = dpp_addu<0xB1>(c3);
                    qs = dpp_addu<0x4E>(qs);                                                         // the four 3-mers of a 2-mer
                    const uint32_t o2 = uint32_t(table_offset(kmin, 2));
                    uint32_t c2 = 0;
                    if ((tid & 3) == 0) { c2 = small16[o2 + (tid >> 2)] + qs; small16[o2 + (tid >> 2)] = uint16_t(c2); }
                    if (kmin <= 1) {
                        uint32_t rs = dpp_addu<0xB1>(c2);
                        rs = dpp_addu<0x4E>(rs); rs = dpp_addu<0x141>(rs); rs = dpp_addu<0x140>(rs);   // the row's four C_2
                        if ((tid & 15) == 0) small16[tid >> 4] = uint16_t(small16[tid >> 4] + rs);
                    }
                }
            }
            __syncthreads();

            auto uni = [](uint32_t x) -> uint32_t { return __builtin_amdgcn_readfirstlane(x); };
            uint32_t upAll = uni(misc[M_UPA]), upGC = uni(misc[M_UPG]);
            if (!tally_by_ballot) {             // kmin = 1: order-1 counts (all valid bases) minus the soft-masked ones
                const uint2 c1 = *reinterpret_cast<const uint2*>(small16);      // A, T | G, C
                upAll = uni((c1.x & 0xFFFFu) + (c1.x >> 16) + (c1.y & 0xFFFFu) + (c1.y >> 16)) - upAll;
                upGC = uni((c1.y & 0xFFFFu) + (c1.y >> 16)) - upGC;
            }
            const int64_t S = int64_t(upAll);                       // windowSpace (L380): uppercase A + T + G + C
            const int64_t nn = n - S;                               // nnTotal of the window
            const bool keep = !(double(nn) >= 0.3 * double(n));     // N filter (L237-241 / L213)
            uint32_t status = (jump ? ROW_JUMPBACK : 0u);
            const uint32_t nvalid_top = uni(misc[M_NVALID]);
            const int n_orph = int(uni(misc[M_NORPH]));
            const bool wrapped = uni(misc[M8_TSUM]) != nvalid_top || n_orph > FRISK8_ORPH_CAP;
            const uint32_t safe_code = uni(misc[M8_SAFE]);

            auto zero_own = [&]() {             // every max-mer position clears its dword (all reads are behind a barrier)
#pragma unroll 4
                for (int it = 0; it < ITS; ++it)
                    if (fullm & (0x80000000u >> it)) t8[code_at(it) >> SHW] = 0u;
            };
            if (wrapped || !keep) {
                if (wrapped) clear_t8(); else zero_own();
                clear_small();
                if (tid == 0) {
                    if (wrapped) {
                        // a counter wrapped (every sum above is then unreliable, the N filter's included), or too many
                        // orphans: the next wider form (8-bit, then scan_kernel.h's 16-bit) redoes this window from scratch
                        const unsigned int slot = atomicAdd(P.out_count, 1u);
                        P.out_list[slot] = cand;
                    } else {
                        P.seq_index[row] = dsi; P.start[row] = rep_start; P.stop[row] = rep_stop;
                        P.status[row] = status;
                        const double qnan = __longlong_as_double(0x7FF8000000000000LL);
                        P.kld[row] = qnan; P.gc[row] = qnan;
                        if (P.flags & 1u) { P.pi[row] = qnan; P.si[row] = qnan; P.cri[row] = qnan; }
                        if (DEBUG && P.dbg_meta) {      // (dropped rows are not compared; keep the dump well defined)
                            P.dbg_meta[row * 3 + 0] = n; P.dbg_meta[row * 3 + 1] = 0; P.dbg_meta[row * 3 + 2] = nn;
                        }
                    }
                }
                __syncthreads();
                continue;
            }
            if (tid == 0) { P.seq_index[row] = dsi; P.start[row] = rep_start; P.stop[row] = rep_stop; }

            // The orphan list in scalar registers.  A run-7 entry is its 7-mer; a run-6 entry has bit 15 set.  o6[k] = the
            // 6-mer of entry k (both kinds count towards c6), o7[0..n7) = the 7-mers of the run-7 entries.  A window without
            // invalid bases has exactly one of each kind (its tail).
            uint32_t o6[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, o7[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            int n7 = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < n_orph) {
                    const uint32_t e = uni(uint32_t(orph[k]));
                    o6[k] = (e >> 2) & (NK / 16u - 1u);
                    if (!(e & 0x8000u)) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (q == n7) o7[q] = e;
                        ++n7;
                    }
                }
            }
            // what a max-mer position reads, all of it addressed by the code alone (so it can be fetched ahead of use):
            // genome-side value, the order-8 counters of its 6-mer / 7-mer / itself, the shared-prefix sums
            struct Fetched { double Ig, A5; uint32_t W5, c8, w7; uint4 w6; };
            auto fetch = [&](uint32_t c16) __attribute__((always_inline)) -> Fetched {
                Fetched f;
                f.Ig = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(P.ig) + (c16 << 3));   // c16 < 4^K always
                if (BITS == 8) {        // the counter of code c is byte c of the table
                    f.c8 = t8b[c16];
                    f.w7 = *reinterpret_cast<const uint32_t*>(t8b + (c16 & ~3u));
                    f.w6 = *reinterpret_cast<const uint4*>(t8b + (c16 & ~15u));
                } else {                // the four nibbles of the 7-mer c >> 2 are the 16 bits at byte 2 (c >> 2)
                    f.c8 = 0;
                    f.w7 = *reinterpret_cast<const uint16_t*>(t8b + ((c16 >> 2) << 1));
                    const uint2 x = *reinterpret_cast<const uint2*>(t8b + ((c16 >> 4) << 3));
                    f.w6 = make_uint4(x.x, x.y, 0u, 0u);
                }
                const uint32_t pc = c16 >> 6;
                f.W5 = pre_w[pc];
                f.A5 = pre_i[pc];
                return f;
            };
            // counts of the three top orders of the max-mer c16.  ORPH: what the caller knows about the orphan list -
            // 2: at most two entries, at most one of them a 7-mer (the usual window); 4: at most four entries; 0: any length
            auto top_counts = [&](const Fetched& f, uint32_t c16, auto orph_c, uint32_t& c8, uint32_t& c7, uint32_t& c6) __attribute__((always_inline)) {
                constexpr int ORPH = decltype(orph_c)::value;
                const uint32_t q6 = c16 >> 4, q7 = c16 >> 2;
                if (BITS == 8) {
                    c8 = f.c8;
                    c7 = __builtin_amdgcn_sad_u8(f.w7, 0u, 0u);
                    c6 = __builtin_amdgcn_sad_u8(f.w6.x, 0u, __builtin_amdgcn_sad_u8(f.w6.y, 0u, __builtin_amdgcn_sad_u8(f.w6.z, 0u, __builtin_amdgcn_sad_u8(f.w6.w, 0u, 0u))));
                } else {
                    c8 = __builtin_amdgcn_ubfe(f.w7, (c16 & 3u) * 4u, 4u);
                    c7 = __builtin_amdgcn_udot8(f.w7, 0x1111u, 0u, false);
                    c6 = __builtin_amdgcn_udot8(f.w6.x, 0x11111111u, __builtin_amdgcn_udot8(f.w6.y, 0x11111111u, 0u, false), false);
                }
                constexpr int N6 = ORPH == 2 ? 2 : 4, N7 = ORPH == 2 ? 1 : 4;
#pragma unroll
                for (int k = 0; k < N7; ++k) c7 += (q7 == o7[k]) ? 1u : 0u;
#pragma unroll
                for (int k = 0; k < N6; ++k) c6 += (q6 == o6[k]) ? 1u : 0u;
                if (ORPH == 0)
                    for (int k = 4; k < n_orph; ++k) {
                        const uint32_t e = orph[k];
                        c7 += (q7 == e) ? 1u : 0u;
                        c6 += (q6 == ((e >> 2) & (NK / 16u - 1u))) ? 1u : 0u;
                    }
            };
            using orph2 = std::integral_constant<int, 2>;
            using orph4 = std::integral_constant<int, 4>;
            using orphN = std::integral_constant<int, 0>;
            // count of the x-mer c in this window (row metadata, RIP, debug dump)
            auto count = [&](int x, uint32_t c) -> uint32_t {
                if (x <= LVL) return small16[table_offset(kmin, x) + c];
                uint32_t c8, c7, c6;
                const uint32_t c16 = c << (2 * (K - x));
                top_counts(fetch(c16), c16, orphN{}, c8, c7, c6);
                return x == K ? c8 : (x == K - 1 ? c7 : c6);
            };

            // ---- stage 3: window constants r_x = 4^x / D_x, D_x = (S-(x-1))*2 (L401-409), and the shared prefix tables
            double r_lane = 0.0;
            if (lane <= 8) r_lane = div_exact(double(1u << (2 * lane)), double(int32_t((S - (lane - 1)) * 2)));
            auto r_of = [&](int x) -> double {
                return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(r_lane), x),
                                        __builtin_amdgcn_readlane(__double2loint(r_lane), x));
            };
            {
                constexpr int LV = LVL;
                double rx[LV + 1];
                uint32_t ox[LV + 1], wm[LV + 1];
#pragma unroll
                for (int x = 1; x <= LV; ++x) {
                    const bool on = x >= kmin;
                    rx[x] = on ? r_of(x) : 0.0;
                    ox[x] = on ? uint32_t(table_offset(kmin, x)) : 0u;
                    wm[x] = on ? 0xFFFFFFFFu : 0u;
                }
#pragma unroll 2
                for (uint32_t c = tid; c < NL; c += NT) {
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
                    pre_w[c] = W;
                }
            }
            __syncthreads();

            if (DEBUG && P.dbg_counts) {
                uint32_t* out = P.dbg_counts + row * int64_t(P.nprof);
                for (int x = kmin; x <= K; ++x) {
                    const int64_t off = table_offset(kmin, x);
                    for (uint32_t c = tid; c < (1u << (2 * x)); c += NT) out[off + c] = count(x, c);
                }
            }
            if (DEBUG && P.dbg_meta && tid == 0) {
                P.dbg_meta[row * 3 + 0] = n;                                                   // totalLen
                P.dbg_meta[row * 3 + 1] = (n >= K ? n - K + 1 : 0) - int64_t(nvalid_top);      // exMax (L344-345)
                P.dbg_meta[row * 3 + 2] = nn;                                                  // nnTotal
            }
            if (nvalid_top == 0) status |= ROW_NO_MAXMER;
            if (nvalid_top > 0 && S >= kmin - 1 && S <= K - 1) status |= ROW_ZERO_WEIGHT;          // zero divisor on the window side
            status |= ROW_KEPT;
            if (tid == 0) {
                P.gc[row] = __longlong_as_double((long long)((uint64_t(uint32_t(S)) << 32) | upGC));
                if (P.flags & 1u) {             // RIP indices (L474-495); codes: AT=1 TA=4 TG=6 GT=9 CA=12 AC=3
                    const double qnan = __longlong_as_double(0x7FF8000000000000LL);
                    const uint32_t AT = count(2, 1), TA = count(2, 4), TG = count(2, 6), GT = count(2, 9), CA = count(2, 12), AC = count(2, 3);
                    const double pi = AT > 0 ? double(TA) / double(AT) : qnan;
                    const double si = (AC + GT) > 0 ? double(CA + TG) / double(AC + GT) : qnan;
                    P.pi[row] = pi;
                    P.si[row] = si;
                    P.cri[row] = (pi == 0.0 || si == 0.0) ? qnan : pi - si;                     // "if PI and SI" (L491)
                }
            }

            // ---- stage 4: every max-mer position: window-side IVOM in closed form, genome side gathered, and the sums
            //      Sw = sum Iw/c8,  Sg = sum Ig/c8,  T = sum Iw ln(Iw/Ig)/c8  over POSITIONS (= sums over distinct max-mers)
            const double r6 = r_of(K - 2), r7 = r_of(K - 1), r8 = r_of(K);      // (named for K = 8: the three orders above the prefix)
            double sw = 0.0, sg = 0.0, stt = 0.0;
            // A position that starts no max-mer scores `safe_code` (a real max-mer of this window: finite values) with weight
            // 1/c8 replaced by 0: it adds exactly +0.0 to every sum, and no term needs a mask.
            auto score_one = [&](const Fetched& f, uint32_t c16, bool on, auto orph_c) __attribute__((always_inline)) {
                uint32_t c8, c7, c6;
                top_counts(f, c16, orph_c, c8, c7, c6);
                double rc = rctab[(on ? c8 : 0u) & 15u];                     // 1/c8 (1.0 for the 19 in 20 max-mers seen once); [0] = 0
                if (BITS == 8 && __builtin_expect(__any(c8 >= 16u), 0)) {     // (wave-uniform, rare: low-complexity sequence)
                    if (on && c8 >= 16u) {                                    // beyond the table: reciprocal + two Newton steps
                        const double dc = double(c8);
                        double r = __builtin_amdgcn_rcp(dc);
                        r = __builtin_fma(r, __builtin_fma(-dc, r, 1.0), r);
                        r = __builtin_fma(r, __builtin_fma(-dc, r, 1.0), r);
                        rc = r;
                    }
                }
                const uint32_t W = f.W5 + (c6 << (2 * K - 4)) + (c7 << (2 * K - 2)) + (c8 << (2 * K));
                // c^2 exactly, as integers (< 2^32), then 4^x / D_x times it
                double A = __builtin_fma(double(__umul24(c6, c6)), r6, f.A5);
                A = __builtin_fma(double(__umul24(c7, c7)), r7, A);
                A = __builtin_fma(double(__umul24(c8, c8)), r8, A);
                // Iw = A/W and Iw/Ig with ONE reciprocal: ratio = A / (W * Ig), Iw = ratio * Ig.  v_rcp_f64 (24.4 bits) + one
                // Newton step = 2^-48.8: the ratio carries a relative error of ~2e-15 - the level of the log table's - instead
                // of being the correctly rounded quotient (two more instructions)
                const double den = double(W) * f.Ig;
                double rr = __builtin_amdgcn_rcp(den);
                rr = __builtin_fma(rr, __builtin_fma(-den, rr, 1.0), rr);
                const double ratio = A * rr;
                const double Iwr = (ratio * f.Ig) * rc;                      // this position's share of Iw
                const double ln = log_tab_n<LOGN, LOGDEG>(ratio, logtab);
                sw += Iwr;
                sg += f.Ig * rc;
                stt += Iwr * ln;
            };
            // the lane's codes and flags again, opaque to the optimiser: without this it keeps every position's pre-shifted
            // code variants of stage 1 alive across the whole window (60 registers) instead of re-deriving them here
            uint32_t ah = uint32_t(acode >> 32), al = uint32_t(acode), fm4 = fullm;
            asm volatile("" : "+v"(ah), "+v"(al), "+v"(fm4));
            const uint64_t acode4 = (uint64_t(ah) << 32) | al;
            auto code4_at = [&](int it) -> uint32_t {          // the position's max-mer, or the stand-in where it starts none
                return ((fm4 >> (31 - it)) & 1u) ? (uint32_t(acode4 >> (64 - 2 * K - 2 * it)) & (NK - 1u)) : safe_code;
            };
            // Shape of the scoring loop, measured per K (bench shard / C2 shape, M windows/s):
            //   K = 8 (LDS allows 3 / 2 workgroups per CU): unrolled, groups of 2: 44.8 / 36.2; rolled, groups of 1: 43.2 / 35.6
            //   K = 6, 7 (tables of 4 / 16 KiB: registers bound the occupancy): unrolled at 3 per CU spills (26 / 24); rolled,
            //   groups of 1, needs 89..92 registers, so FOUR workgroups share a CU: 50.2 / 45.3 (16-bit form: 31.0 / 26.0)
#ifdef FRISK8_S4_GROUP
            constexpr int GR = FRISK8_S4_GROUP;
#else
            constexpr int GR = K == 8 ? 2 : 1;
#endif
#ifdef FRISK8_ROLLED
            constexpr bool ROLLED = FRISK8_ROLLED != 0;
#else
            constexpr bool ROLLED = K < 8;
#endif
            auto score_all = [&](auto orph_c) __attribute__((always_inline)) {
                // software pipeline, fully unrolled: the reads of group g+1 are issued before the arithmetic of group g.
                // (A rolled loop - two groups per trip, ping-pong buffers - needs 86..129 registers and no scratch, but
                // measured 4..6 % slower at three workgroups per CU; thread counts 320 / 384 / 512 per workgroup 18..60 %.)
                // the rolled form: two groups per trip, ping-pong buffers (ITS is a multiple of 2 GR for GR = 1, 2)
                if constexpr (ROLLED && ITS % (2 * GR) == 0) {
                    Fetched bufA[GR], bufB[GR];
#pragma unroll
                    for (int k = 0; k < GR; ++k) bufA[k] = fetch(code4_at(k));
#pragma unroll 1
                    for (int g = 0; g < ITS; g += 2 * GR) {
#pragma unroll
                        for (int k = 0; k < GR; ++k) bufB[k] = fetch(code4_at(g + GR + k));
#pragma unroll
                        for (int k = 0; k < GR; ++k) score_one(bufA[k], code4_at(g + k), (fm4 >> (31 - (g + k))) & 1u, orph_c);
                        __builtin_amdgcn_sched_barrier(0);
                        const int gn = g + 2 * GR < ITS ? g + 2 * GR : 0;       // (the last trip fetches group 0 again, unused)
#pragma unroll
                        for (int k = 0; k < GR; ++k) bufA[k] = fetch(code4_at(gn + k));
#pragma unroll
                        for (int k = 0; k < GR; ++k) score_one(bufB[k], code4_at(g + GR + k), (fm4 >> (31 - (g + GR + k))) & 1u, orph_c);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    return;
                }
                Fetched buf[2][GR];
#pragma unroll
                for (int k = 0; k < GR; ++k) if (k < ITS) buf[0][k] = fetch(code4_at(k));
#pragma unroll
                for (int g = 0; g < ITS; g += GR) {
                    const int cur = (g / GR) & 1;
#pragma unroll
                    for (int k = 0; k < GR; ++k) if (g + GR + k < ITS) buf[cur ^ 1][k] = fetch(code4_at(g + GR + k));
#pragma unroll
                    for (int k = 0; k < GR; ++k) if (g + k < ITS) score_one(buf[cur][k], code4_at(g + k), (fm4 >> (31 - (g + k))) & 1u, orph_c);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (n_orph <= 2 && n7 <= 1) score_all(orph2{});
            else if (n_orph <= 4) score_all(orph4{});
            else score_all(orphN{});

            // workgroup totals in a fixed order: DPP butterfly per wave, then the NW partials in wave order
            sw = wave_sum_exact(sw); sg = wave_sum_exact(sg); stt = wave_sum_exact(stt);
            if (lane == 0) { double* p = scratch + (tid >> 6) * 3; p[0] = sw; p[1] = sg; p[2] = stt; }
            __syncthreads();
#pragma unroll 4
            for (int it = 0; it < ITS; ++it)                // behind the barrier: nobody reads the tables any more
                if (fm4 & (0x80000000u >> it)) t8[code4_at(it) >> SHW] = 0u;
            clear_small();
            if (tid == 0) {
                double a = 0.0, b = 0.0, c = 0.0;
                for (int w = 0; w < NW; ++w) { a += scratch[3 * w]; b += scratch[3 * w + 1]; c += scratch[3 * w + 2]; }
                P.status[row] = status;
                P.sw[row] = a;
                P.sg[row] = b;
                P.kld[row] = c;                             // T; finish_rows_kernel turns (T, Sw, Sg) into the KLD
            }
            __syncthreads();
        }
    }
}

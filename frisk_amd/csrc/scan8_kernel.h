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
//               inside a wave by DPP sums (K = 7, 8: as the first lines of stage 3, where thread t holds the 4-mer t -
//               no barrier and no LDS round trip of their own).  The grand total of the table must equal the number of max-mer positions;
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
//   * work      chunks of 8 consecutive windows, dealt by one counter per XCD (a workgroup takes the next chunk when it is
//               done with its last; neighbouring chunks stay on one L2; an XCD that has run dry takes from the next one).
//   * round 3    inside a chunk of 16 consecutive windows the order-K table SLIDES (2 inc updates per window instead of w - K + 1 and a
//               cleared table); orphans are FOLDED INTO THE TABLE as max-mers that do not occur (no orphan compares in stage 4); and the
//               genome-side value of a position travels from window to window through a per-workgroup RING in global memory instead of
//               being gathered - a whole L2 line per 8 bytes - by every window that covers it: that gather's L2 -> L1 traffic was what
//               the kernel was bound by.  Each is described where it is implemented; measurements: DESIGN.md section 3.3a.
// Windows up to NT*ITS bases, kmin <= K-3 (the shared prefix level); everything else stays on scan_kernel.h.
// Measurements, the adaptive choice between 4 and 8 bits, and what was tried and dropped: DESIGN.md section 3.3.
#pragma once
#include "scan_kernel.h"

#define FRISK8_ORPH_CAP 24         // orphan entries kept in LDS (two per invalid run); a window with more goes to the 16-bit form
#ifndef FRISK8_UNROLL1
#define FRISK8_UNROLL1 2           // unroll factor of the stage-1 position loop
#endif
#ifndef FRISK8_SHORT_LANES
#define FRISK8_SHORT_LANES 6        // stage 1: up to this many lanes of a wave with short words get a pass each (more: per position)
#endif
#ifndef FRISK8_PRIO
#define FRISK8_PRIO 3               // wave priority (s_setprio) of every stage but the scoring loop; 0 = no priorities
#endif
#ifndef FRISK8_W7_READ
#define FRISK8_W7_READ 0            // 4-bit form: 1 = the (K-1)-mer's four nibbles come from an LDS read of their own (round 2's form)
#endif
#ifndef FRISK8_PRE_SPLIT
#define FRISK8_PRE_SPLIT 1          // shared prefix sums as two arrays - A[] read by one ds_read_b64, W[] by one ds_read_b32 - instead of
#endif                              // interleaved 12-byte entries (a ds_read2_b32 and a ds_read_b32: half again as many LDS passes)
#ifndef FRISK8_PLACE
#define FRISK8_PLACE 1              // orphans are folded into the order-K table where it has room (stage 3): no orphan compares in the scoring loop
#endif
#ifndef FRISK8_RING
#define FRISK8_RING 1               // genome-side values travel from window to window through a per-workgroup ring in global memory (below)
#endif
#ifndef FRISK8_PARK_LATE
#define FRISK8_PARK_LATE 0          // 1: a parking wave stores behind its scoring loop instead of inside it (measured: +-0)
#endif
#ifndef FRISK8_RING_COLS
#define FRISK8_RING_COLS 256        // ring geometry: ITS rows x 256 columns of doubles per workgroup (position p <-> row p % ITS, column p / ITS % 256):
#endif                              // ITS x 256 = the most positions a window of this instantiation has - 40 KB per workgroup at 20 positions per lane
                                    // (round 3 had 512 columns, 80 KB: the same time, twice the footprint beside 4 MB of L2 per XCD)
#define FRISK8_RING_PAD 16          // doubles behind every workgroup's slice of the ring (see dummy_off)
#define FRISK8_SLOTS 8             // misc counters per window (double-buffered by window parity)

enum { M8_NPLACED = 1,             // misc slots (0, 2, 4, 5: M_UPA, M_UPG, M_NORPH, M_NVALID): orphans folded into the order-K table ...
       M8_PMASK = 3,               // ... and which entries of the orphan list those are (bit k <-> entry k, which then holds the fake code)
       M8_TSUM = 6,                // grand total of the order-8 table (overflow check) ...
       M8_SAFE = 7 };              // ... and the code of SOME max-mer of the window (what lanes without one score instead)

// LDS carve-up, all compile-time: the kernels declare it as ONE static array, so every table address is a constant that
// folds into the 16-bit offset field of the ds_ instructions (a dynamic `extern __shared__` base costs one VALU add per
// address).  The 32 / 64 KiB order-8 table comes LAST: its own offset is then the only large one, and it is an immediate.
template <int KMAX, int BITS, int LOGN, int NT, bool SIDE = false>
struct Lds8 {
    static constexpr uint32_t small = 0;
    static constexpr uint32_t small_bytes = 2736;                                  // orders kmin..KMAX-3 as u16 bins (sized for 1..5)
    static constexpr uint32_t orphans = small + small_bytes;                       // u16[FRISK8_ORPH_CAP]
    static constexpr uint32_t NL = 1u << (2 * (KMAX - 3));                         // entries of the shared prefix tables (level KMAX-3)
    static constexpr uint32_t pre = (orphans + FRISK8_ORPH_CAP * 2 + 15) / 16 * 16;     // Pre8[NL]: the shared prefix sums, 12 bytes each
    // SIDE (below): the side table follows the prefix sums (whose weights carry the side count of the (K-3)-mer's 4-mer in their top bits)
    // (LDS is handed out in pieces of 1280 bytes on gfx950: three workgroups per CU get 42 of them each = 53 760 bytes)
    static constexpr uint32_t side = pre + NL * 12;                                // u8[256]: counts of the period-4 max-mers (SIDE)
    static constexpr uint32_t logtab = side + (SIDE ? 256 : 0);                    // {1/c_i, -ln(1/c_i)} x LOGN
    // {1/c, c^2 r_K} for c < 16 (second half per window); the 8-bit form has 128 bytes to spare, not 256: 1/c only, the other computed
    static constexpr uint32_t rctab = logtab + uint32_t(LOGN) * 16;
    // (SIDE: {1/c, c^2 r_K} for c < 256 - a side count goes up to 255)
    static constexpr uint32_t misc = rctab + (SIDE ? 256 * 16 : 16 * (BITS == 4 ? 16 : 8));     // counters x2, then one {Sw, Sg, T} per wave
    static constexpr uint32_t t8 = (misc + 2 * FRISK8_SLOTS * 4 + uint32_t(NT / 64) * 3 * 8 + 15) / 16 * 16;
    static constexpr uint32_t t8_bytes = (1u << (2 * KMAX)) * BITS / 8;
    static constexpr uint32_t total = t8 + t8_bytes;
    static constexpr uint32_t granules = (total + 1279) / 1280;                    // what the hardware allocates
};

// ln(x), x positive and normal, by table range reduction as scan_kernel.h's log_tab_pos: x = m 2^k, m in [0.5, 1); the top
// log2(N) mantissa bits pick the bin, tab[i] = {u_i = 1/c_i rounded, -ln u_i}, r = m u_i - 1 exactly (one fma),
// ln x = k ln2 - ln u_i + log1p(r) with log1p by its Taylor polynomial of degree DEG.  Absolute error < 1e-15 for the ratios
// of probabilities scored here (|k| small): what the sum T = sum Iw ln(Iw/Ig) needs.  DEG + 5 instructions.
template <int N, int DEG>
__device__ inline double log_tab_n(double x, const double2* tab) {
    const int k = __builtin_amdgcn_frexp_exp(x);
    const double m = __builtin_amdgcn_frexp_mant(x);
    // (the bin's BYTE offset straight from the mantissa's top bits: shift + mask, no index scaling)
    const uint32_t off = (uint32_t(__double2hiint(m)) >> ((N == 128 ? 13 : (N == 64 ? 14 : 15)) - 4)) & (uint32_t(N - 1) << 4);
    const double2 e = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(tab) + off);
    const double r = __builtin_fma(m, e.x, -1.0);
    double p = (DEG & 1) ? 1.0 / DEG : -1.0 / DEG;
#pragma unroll
    for (int d = DEG - 1; d >= 2; --d) p = __builtin_fma(r, p, (d & 1) ? 1.0 / d : -1.0 / d);
    return __builtin_fma(double(k), 0.69314718055994530942, e.y) + __builtin_fma(r * r, p, r);
}

// shared prefix sums of one (K-3)-mer, the orders kmin..K-3 of a max-mer's two sums (as scan_kernel.h's pre_i / pre_w): numerator
// term and integer weight, interleaved so that ONE address serves both reads of a position
struct __attribute__((packed, aligned(4))) Pre8 {
    double A;
    uint32_t W;
};

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

// The adaptive width's verdict, on the device (one thread, behind the sample launch): which form scores the rest of the scan.
// counts[0] = sampled windows handed on anyway, [2] = scored, but a plain 4-bit counter would have wrapped, [3] = scored;
// n_sampled = windows in the sample.  The rule is frisk_abi.hip's (measured break-evens there): 8-bit bulk when more than three
// sampled windows in ten overflow 4 bits anyway; else the side table when the plain form would hand on more than side_share of
// the windows that are scored.  verdict[0] = 1 plain 4-bit, 2 4-bit + side table, 3 8-bit; verdict[1..3] = the three counts (for
// the host's statistics, read at the end of the scan).
__global__ void scan8_decide_kernel(const unsigned int* __restrict__ counts, unsigned int n_sampled, double side_share, int side_ok,
                                    unsigned int* __restrict__ verdict) {
    const unsigned int handed = counts[0], would = counts[2], scored = counts[3];
    unsigned int form = (double(handed) <= 0.3 * double(n_sampled)) ? 1u : 3u;
    if (form == 1u && side_ok && double(handed + would) > side_share * double(handed + scored)) form = 2u;
    verdict[0] = form; verdict[1] = handed; verdict[2] = would; verdict[3] = scored;
}

// ROLE names the launch (bit 0: the sample of the adaptive width, bit 1: no sliding, hence no ring): the code is otherwise the same, but a profiler's
// per-kernel statistics then keep the 1/16 sample launches apart from the bulk launches.
// NT threads, windows of at most NT*ITS bases, BITS per order-8 counter, LOGN: bins of the logarithm table, WPS: waves per SIMD the register allocation must allow (= workgroups per
// CU * NT / 256).
// SIDE (K = 8, 4-bit counters): the max-mers of period <= 4 - (x0 x1 x2 x3)(x0 x1 x2 x3): poly-A, (CA)n, (AAAT)n ..., the words that
// wrap a 4-bit counter in real assemblies - are counted in a side table of 256 16-bit counters instead of the order-K table;
// see "SIDE" in the kernel body.
template <int KMAX, int NT, int ITS, int BITS, int LOGN, int WPS, bool DEBUG, int ROLE = 0, bool SIDE = false>
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
    static_assert(!SIDE || (KMAX == 8 && BITS == 4 && !DEBUG && NT == 256 && FRISK8_PRE_SPLIT), "the side table exists for the 4-bit form at K = 8");
    using L = Lds8<KMAX, BITS, LOGN, NT, SIDE>;
    // the order-K table is cleared whole when that takes no more stores per thread than a lane has positions (measured: 64 KiB
    // for windows of 2000 bases is the one case where every position clearing its own dword is cheaper)
    constexpr bool CLEAR_ALL = L::t8_bytes / 16 / NT <= uint32_t(ITS);
    // K = 7, 8: thread t sums the table below ITS OWN (K-3)-mers (the 4-mer t / its four 5-mers), so that stage 2 runs inside stage 3: no
    // barrier and no LDS round trip between the table sums and the prefix tables made from them
    constexpr bool FUSED = (KMAX >= 7 && NT >= 256 && NT % 256 == 0);
    // Orphans - the (K-1)- and (K-2)-mers that are no prefix of a counted max-mer - need not be compared against every position's
    // code in the scoring loop: an orphan (K-1)-mer can be ADDED TO THE TABLE as a max-mer that does not occur in the window (a
    // zero counter among its four children), an orphan (K-2)-mer as one under a (K-1)-mer that does not occur (four zero counters
    // in a row).  Sums of 4 and of 16 neighbours - c_{K-1}, c_{K-2} - and the table sums of stage 3 then count the orphan like any
    // max-mer, and the fake counter itself is never read: no position has its code (c_K), and no position sits under a (K-1)-mer
    // that does not occur.  Stage 3 does this, the thread that owns the orphan's 4-mer (whose sums over that part of the table
    // follow in program order); where the table has no room the orphan stays on the list and the window takes the scoring loop
    // with compares.  The same counts either way, so the same bits.  (Debug builds dump every counter: they keep the list.)
    constexpr bool PLACE = FRISK8_PLACE && FUSED && !DEBUG;
#ifdef FRISK8_ROLLED
    constexpr bool ROLLED_K = FRISK8_ROLLED != 0;
#else
    constexpr bool ROLLED_K = KMAX < 8 || NT > 256;      // the scoring loop is rolled (two groups per trip) - measured per K, see stage 4
#endif                 // (K = 7: the 64 counters below the 4-mer t itself)
    __shared__ __attribute__((aligned(16))) unsigned char lds[L::total];
    const int tid0 = threadIdx.x;
    const int kmin0 = P.kmin;
    uint32_t* t8 = reinterpret_cast<uint32_t*>(lds + L::t8);
    const unsigned char* t8b = lds + L::t8;
    uint32_t* small32 = reinterpret_cast<uint32_t*>(lds + L::small);
    uint16_t* small16 = reinterpret_cast<uint16_t*>(lds + L::small);
    uint16_t* orph = reinterpret_cast<uint16_t*>(lds + L::orphans);
    Pre8* pre = reinterpret_cast<Pre8*>(lds + L::pre);
    double* preA = reinterpret_cast<double*>(lds + L::pre);                    // (FRISK8_PRE_SPLIT: NL doubles, then NL words)
    uint32_t* preW = reinterpret_cast<uint32_t*>(lds + L::pre + NL * 8);
    // SIDE.  A window of a real assembly that holds a max-mer 16+ times nearly always holds a SIMPLE one: a poly-A tail, a
    // (CA)n or (AAAT)n run.  Such a window wraps a 4-bit counter and had to be redone with 8-bit counters - two workgroups per
    // CU instead of three - and on repeat-rich sequence that was every other window.  The SIDE form keeps the 4-bit table and
    // counts the 256 max-mers of period <= 4 - code (y << 8) | y, y = x0 x1 x2 x3 - beside it, in 8-bit counters indexed by y:
    //   * stage 1 sends a position with such a max-mer to side[y] instead of the table (same for the sliding updates);
    //   * stage 3: thread t holds the 4-mer t, so side[t] is ITS count - one of the 64 max-mers below its (K-3)-mer 4 t + x0: it
    //     adds side[t] there (and to the table's grand total), which makes every order <= K-3 right, and stores side[t]
    //     in the top bits of the weight of each of its four (K-3)-mers' prefix sums: the scoring loop's read of W brings it along;
    //   * stage 4: the only period-4 max-mer below the (K-2)-mer x0..x5 of a position is (x0 x1 x2 x3)^2 - if x4 x5 = x0 x1 -, and
    //     likewise for its (K-1)-mer (x4 x5 x6 = x0 x1 x2) and the max-mer itself: with t = (code ^ code >> 8) & 0xFF the side
    //     count joins c_{K-2} if t < 16, c_{K-1} if t < 4, c_K if t = 0 - as the start values of the sums that are there anyway;
    //   * {1 / c_K, c_K^2 r_K} come from a table of 256 entries instead of 16 - the reciprocals made once per workgroup with the
    //     8-bit form's instruction sequence for counts beyond 15, the products per window as before: the same bits.
    // Max-mers that are NOT of period <= 4 and occur 16+ times still wrap their counter: grand-total test, hand-over, as before -
    // and so does a side counter at 256 (the side counts are part of the grand total): a window with a max-mer that occurs 256+ times
    // is the 16-bit form's whichever form the bulk runs in - a property of the window, so rows do not depend on the launch's shape.
    uint32_t* side32 = reinterpret_cast<uint32_t*>(lds + L::side);
    const unsigned char* side8 = lds + L::side;
    auto put_pre = [&](uint32_t idx, double A, uint32_t W, uint32_t sd = 0u) __attribute__((always_inline)) {
        if constexpr (SIDE) { preA[idx] = A; preW[idx] = W | (sd << 23); }     // (W < 5120 (4 + 16 + ... + 4^5) < 2^23; the side count < 2^8)
        else if (FRISK8_PRE_SPLIT) { preA[idx] = A; preW[idx] = W; } else { pre[idx].A = A; pre[idx].W = W; }
    };
    // (stage 1) one max-mer position more (SIGN = +1) or less (-1): the table's field, or the side counter of a period-4 max-mer
    auto bump = [&](uint32_t c16, auto sign_c) __attribute__((always_inline)) {
        constexpr int SIGN = decltype(sign_c)::value;
        if (SIDE && ((c16 ^ (c16 >> 8)) & 0xFFu) == 0u) {
            const uint32_t one = 1u << ((c16 & 3u) * 8u);
            if (SIGN > 0) atomicAdd(&side32[(c16 & 0xFFu) >> 2], one); else atomicSub(&side32[(c16 & 0xFFu) >> 2], one);
        } else {
            const uint32_t one = 1u << ((c16 & PERM) * BITS);
            if (SIGN > 0) atomicAdd(&t8[c16 >> SHW], one); else atomicSub(&t8[c16 >> SHW], one);
        }
    };
    uint32_t* misc_base = reinterpret_cast<uint32_t*>(lds + L::misc);
    double* scratch = reinterpret_cast<double*>(lds + L::misc + 2 * FRISK8_SLOTS * 4);
    const double2* logtab = reinterpret_cast<const double2*>(lds + L::logtab);
    double2* rstab = reinterpret_cast<double2*>(lds + L::rctab);        // (BITS == 4)
    double* rctab = reinterpret_cast<double*>(lds + L::rctab);          // (BITS == 8)

    auto clear_t8 = [&]() {
#pragma unroll
        for (int i = tid0; i < int(L::t8_bytes / 16); i += NT) reinterpret_cast<uint4*>(t8)[i] = make_uint4(0, 0, 0, 0);
        if (SIDE && tid0 < 64) side32[tid0] = 0u;         // (the side table lives and dies with the order-K table)
    };
    auto clear_small = [&]() {
        for (uint32_t i = tid0; i < L::small_bytes / 16; i += NT) reinterpret_cast<uint4*>(small32)[i] = make_uint4(0, 0, 0, 0);
    };
    // (a launch over a hand-over list finds it empty nearly always: nothing to set up)
    if (P.in_list != nullptr && *P.in_count == 0u) return;
    // (the first scan of a batch queues all three bulk forms behind its sample; the sample's verdict, on the device, picks one)
    if (P.verdict != nullptr && *P.verdict != P.my_form) return;
#if FRISK8_PRIO
    __builtin_amdgcn_s_setprio(FRISK8_PRIO);
#endif
    clear_t8();
    clear_small();
    if (tid0 < 2 * FRISK8_SLOTS) misc_base[tid0] = 0;
    {
        double2* lt = reinterpret_cast<double2*>(lds + L::logtab);
        for (int i = tid0; i < LOGN; i += NT) lt[i] = reinterpret_cast<const double2*>(LOGN == 128 ? P.log_tab : (LOGN == 64 ? P.log_tab64 : P.log_tab32))[i];
        if constexpr (SIDE) {
            static_assert(!SIDE || NT >= 256, "one thread per entry of the reciprocal table");
            if (tid0 < 256) {
                double r = P.rc_tab[tid0 & 15];
                if (tid0 >= 16) {                       // reciprocal + two Newton steps: the 8-bit form's sequence for a count beyond its table
                    const double dc = double(tid0);
                    r = __builtin_amdgcn_rcp(dc);
                    r = __builtin_fma(r, __builtin_fma(-dc, r, 1.0), r);
                    r = __builtin_fma(r, __builtin_fma(-dc, r, 1.0), r);
                }
                rstab[tid0] = make_double2(r, 0.0);
            }
        } else if (tid0 < 16) {
            if (BITS == 4) rstab[tid0] = make_double2(P.rc_tab[tid0], 0.0); else rctab[tid0] = P.rc_tab[tid0];
        }
    }
    __syncthreads();

    // XCD-aware work split (as scan_kernel.h): blocks b and b+8 share an XCD, neighbouring chunks share an L2
    const int G = gridDim.x;
    int v = blockIdx.x;
    if ((G & 7) == 0) v = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    // Which candidates: the chunks of [c0, c1) - all of them (sel_mode 0), every sel_mod-th (1: the sample that decides the
    // counter width for the rest), all but those (2) - or, one at a time, the windows a narrower form handed over (in_list).
    // (a launch over a hand-over list takes its windows one by one, in the order they were appended.  Walking a LONG list in
    //  candidate order instead - the launch going through the range in chunks and scoring the windows flagged as handed on, so that
    //  the runs a satellite array hands over slide - was built in round 4 and measured slower: 8.05 against 7.43 ms per scan of
    //  the shard with 3 % satellite arrays, 40 804 windows handed on; NOTES.md)
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
    // SLIDING (P.slide_pp > 0): consecutive windows of a chunk share w - inc of their w bases, so the order-K table of window
    // j + 1 is the table of window j minus the max-mers that start in its first inc positions plus those that start in the
    // inc positions before the new window's last K - 1 - 2 inc updates spread over all threads (slide_pp positions of each
    // range per thread) instead of w - K + 1 and a cleared table.  A counter word is a plain 32-bit sum of its fields' terms,
    // so additions and subtractions commute mod 2^32: the word holds sum(count_f << BITS f) mod 2^32 whatever the order of the
    // updates, which IS the packed counts whenever every count of the window fits its field - a window with a wrapped field is
    // caught by the grand-total test as before (and handed on), and the table is right again as soon as the counts fit again.
    // Everything else of a window - composition, short words, orphans, stages 3 and 4 - is computed as before, by the same
    // lanes in the same order: rows do not depend on whether a window was slid into or counted afresh.
    const int slide_pp = P.in_list == nullptr ? P.slide_pp : 0;
    // THE GENOME-SIDE GATHER.  Every scored position needs Ig[its max-mer] - 8 bytes at a random place of a 512 KB table, which
    // costs a whole 128-byte line from L2 each: 5 000 lines per window, 247 GB per scan of the bench shard, which at the L2's
    // ~34 TB/s IS the scan's 7.4 ms (round 3's ablations: with the gather confined to L1 the same kernel took 15 % less, and no
    // diet of instructions or LDS reads showed while the gather stood).  But a position's max-mer - hence its Ig - is the same
    // in every window that covers it, and consecutive windows share w - inc of w positions.  So the value is gathered ONCE, by
    // the scoring loop of the first window of the chunk that scores the position - a window counted afresh: every lane; a window
    // slid into: the lanes that hold its entering range - and parked in a ring in global memory that belongs to this workgroup,
    // indexed by the position in the scaffold: row = p % ITS, column = p / ITS % 512.  The other lanes read ring[p] instead of
    // Ig[code]: for a fixed step `it` of the scoring loop the lanes' positions are ITS apart, i.e. the same row and consecutive
    // columns - 512 contiguous bytes per wave instruction instead of 64 lines.  A wave with lanes that gather parks what its lanes
    // used (a lane that read the ring rewrites what it read; a position that starts no max-mer parks a 1.0 - its stand-in has
    // weight 0): its copy of the loop has no branch on who is who - the load's offset is a select between two, the store is
    // unconditional; a wave whose lanes all read the ring runs a copy with neither (stage 4).  L2 -> L1 traffic per window: about
    // inc gathers + w coalesced doubles (170 KB) instead of w gathers (640 KB).  Same values, same lanes, same order of
    // summation: same bits.  (The ring lives in the Infinity Cache: 80 KB per workgroup, more in all than L2 holds.)
    // (K = 8 with 4-bit counters only: the 8-bit form - two workgroups per CU - is bound by instruction issue at its occupancy,
    //  not by the gather: measured 8.80 ms without the ring, 9.06 with it, on the repeat-rich shape; at K = 6, 7 the table is 32 / 128 KB)
    //  ROLE & 2: a launch whose windows do not slide (increment above half a window, or too few windows for chunks): every window
    //  would gather everything and park it for nobody - such launches take the instantiation without the ring.)
    constexpr bool RING = FRISK8_RING != 0 && KMAX == 8 && BITS == 4 && !(ROLE & 2);
    // (one buffer: a copy of the genome table first, the workgroups' slices behind it - so that "from the table" and "from the
    //  ring" are two 32-bit offsets from one base, and the scoring loop's load is one instruction either way)
    char* const ring = RING ? reinterpret_cast<char*>(P.ig_ring) : nullptr;
    // (a workgroup's slice: ITS x COLS doubles, then FRISK8_RING_PAD doubles that nobody reads - where the lanes of a parking wave that
    //  gathered nothing send their store: an unconditional store with a selected address keeps the scoring loop one scheduling region,
    //  a store under a per-lane condition put a branch behind every position and made the scan 2.6 x slower)
    const uint32_t slice_off = RING ? uint32_t((size_t(NK) + size_t(blockIdx.x) * (size_t(ITS) * FRISK8_RING_COLS + FRISK8_RING_PAD)) * 8) : 0u;
    const uint32_t dummy_off = slice_off + uint32_t(ITS) * FRISK8_RING_COLS * 8u;
    bool slide_next = false;            // the table is left standing for the next window (which slides); false: it is cleared
    bool ring_next = false;             // ... and so is the ring: the window before this one went through the scoring loop (a window
                                        // that the N filter drops, or that is handed on, parks nothing - its successor gathers afresh)
#ifdef FRISK_STAMPS
    int stamp_win = -1;         // (diagnostic builds: STAMP of scan_kernel.h - wave 0 of the first workgroups, s_memtime per stage)
#endif

    // candidate `cand` of the descriptor in `d`: first base, reported coordinates, length (crawlGenome L211-245)
    auto window_of = [&](int64_t cand, int64_t& st, int64_t& rep_start, int64_t& rep_stop, int& n, bool& jump) {
        const int64_t j = cand - d.cand0 + d.j0;               // window index inside the scaffold
        jump = false;
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
    };
    // Chunks are dealt by counters when the launcher provides them: a workgroup takes the next chunk when it is done with
    // its last.  Chunks differ a lot in cost (windows that the N filter drops stop after stage 2), and with the static deal -
    // chunk v, v + G, v + 2 G ... - a launch ended with 8 % of its time spent waiting for the unluckiest workgroups.  One
    // counter per XCD (blocks b and b + 8 share one), each over a contiguous eighth of the chunks: consecutive chunks - whose
    // windows overlap - go through the same L2, as with the static deal; a workgroup whose XCD has run dry takes from the
    // next one's.  Thread 0 asks for the chunk after this one before it starts on this one, so the round trip of the atomic
    // hides behind a chunk's work.
    // The pointers that are used once per window or per chunk, by one thread - the row's output columns, the hand-over list, the chunk
    // queues, the descriptors - come from the kernel-argument segment where they are used, through a pointer the optimiser cannot see
    // through (laundered per chunk and per window): held in scalar registers across the window loop, these sixteen pointers were
    // sixteen pairs parked in VGPR lanes and read back every window (120 -> 89 spilled scalars, 334 -> 186 v_readlane in the loop).
    typedef const __attribute__((address_space(4))) ScanParams* KernargView;
    KernargView Pk = (KernargView)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t* next_q = reinterpret_cast<uint32_t*>(scratch);         // (the waves' partial sums live here at the END of a window)
    const bool dealt = Pk->queue != nullptr;
    const uint32_t NQ = dealt ? uint32_t(P.queue_n) : 1u;
    const uint32_t myq = blockIdx.x % NQ;
    uint32_t dry = 0, ahead = 0;                                     // (thread 0) queues found empty so far; the index asked for ahead
    if (dealt && tid0 == 0) ahead = atomicAdd(&Pk->queue[myq], 1u);
    for (int64_t qs = v;; qs += G) {
        int64_t q = qs;
        asm volatile("" : "+s"(Pk));
        if (dealt) {
            if (tid0 == 0) {
                uint32_t got = 0xFFFFFFFFu;
                while (dry < NQ) {
                    const uint32_t j = (myq + dry) % NQ;
                    const int64_t b = nchunks * j / NQ, len = nchunks * (j + 1) / NQ - b;
                    if (int64_t(ahead) < len) { got = uint32_t(b + ahead); break; }
                    if (++dry < NQ) ahead = atomicAdd(&Pk->queue[(myq + dry) % NQ], 1u);
                }
                *next_q = got;
                if (got != 0xFFFFFFFFu) ahead = atomicAdd(&Pk->queue[(myq + dry) % NQ], 1u);
            }
            __syncthreads();
            const uint32_t got = uint32_t(__builtin_amdgcn_readfirstlane(int(*next_q)));
            __syncthreads();
            if (got == 0xFFFFFFFFu) break;
            q = int64_t(got);
        }
        if (q >= nchunks) break;
        int64_t qq = q;                                                  // chunk index inside [c0, c1)
        if (!listed && P.sel_mode == 1) qq = q * M;
        if (!listed && P.sel_mode == 2) qq = (q / (M - 1)) * M + 1 + q % (M - 1);
        const int64_t cb = listed ? q : P.c0 + qq * chunk;
        const int64_t ce = listed ? q + 1 : ((cb + chunk < P.c1) ? cb + chunk : P.c1);
        for (int64_t ci = cb; ci < ce; ++ci) {
            const int64_t cand = listed ? Pk->in_list[ci] : ci;
            // ---- which scaffold / window is this candidate? (uniform; crawlGenome L194-251)
            if (cand < d.cand0 || cand >= d.cand0 + d.ncand) {
                int lo = 0, hi = P.n_desc - 1;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (Pk->descs[mid].cand0 <= cand) lo = mid; else hi = mid - 1;
                }
                d = Pk->descs[lo];
                dsi = lo;
                // (into scalar registers: the descriptor is the same for every lane, but a plain global load leaves it in vector
                //  registers, and the window geometry below - 64-bit multiplies and compares per window - then runs on the VALU)
                auto s64 = [](int64_t x) -> int64_t {
                    return int64_t((uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(uint64_t(x) >> 32)))) << 32) |
                                   uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(uint64_t(x))))));
                };
                d.off = s64(d.off); d.size = s64(d.size); d.cand0 = s64(d.cand0); d.ncand = s64(d.ncand);
                d.base0 = s64(d.base0); d.j0 = s64(d.j0); d.kind = __builtin_amdgcn_readfirstlane(d.kind);
            }
            // The thread index and the lowest order, opaque to the optimiser from here on: otherwise it hoists every
            // per-position constant that depends on them (20 x {tid*20+it, masks, table offsets}: > 100 registers and
            // dozens of spilled scalars) out of the window loop and keeps them alive across all stages.
            int tid = tid0, kmin = kmin0;
            asm volatile("" : "+v"(tid), "+s"(kmin));
            asm volatile("" : "+s"(Pk));            // (kernel arguments used once per window: re-read from their segment, see Pk)
#ifdef FRISK_STAMPS
            ++stamp_win;
#endif
            STAMP(0)
            const int lane = tid & 63;
            int64_t st, rep_start, rep_stop;
            int n;
            bool jump = false;
            window_of(cand, st, rep_start, rep_stop, n, jump);
            const int64_t g0 = d.off + (st - d.base0);            // resident position of the window's first base
            const int64_t row = cand - P.c0;
            // this window slides into the table its predecessor left; its successor - the next candidate of this chunk, in the
            // same scaffold, a full window like this one (no jumpback, L230-232) - will slide into this one's
            // where the window's first base sits in the ring: row rb_r, column rb_q (uniform)
            const uint32_t ring_base = RING ? uint32_t(uint64_t(st) % uint64_t(ITS * FRISK8_RING_COLS)) : 0u;
            const uint32_t rb_q = ring_base / uint32_t(ITS), rb_r = ring_base - rb_q * uint32_t(ITS);
            // WHICH positions a lane holds.  Block b = positions [b ITS, (b + 1) ITS) of the window; lane t holds block t.  FRISK8_DEAL = 1
            // (round 4, measured and NOT taken) deals the blocks to the four waves in groups of sixteen - wave w holds the groups w,
            // w + 4, w + 8, w + 12 - so that the entering range of a slid window (the lanes that gather from the genome table, a cache
            // line per lane, and park) is shared by all four waves (9 / 16 / 16 / 10 lanes at w = 5000, inc = 1000) instead of sitting
            // in wave 3 (57 lanes) with the other three waiting at the loop's barrier.  But then every wave runs the parking copy of
            // the scoring loop: 6.30-6.35 ms per scan of the bench shard against 6.13-6.17 (stores behind the loop: 6.21-6.31), and
            // with the store under a per-lane condition - a branch behind every position - 16.1 ms.
#ifndef FRISK8_DEAL
#define FRISK8_DEAL 0
#endif
            const int blk = (FRISK8_DEAL && NT == 256) ? ((((tid & 63) >> 4) * 4 + (tid >> 6)) * 16 + (tid & 15)) : tid;
            // byte offset (from the buffer's base) of the ring's place for this lane's it-th position
            // ... split into what is the same for every lane (the row, and the slice: scalar arithmetic, and it becomes part of the load's
            // scalar base) and the lane's column - one of two values per window, by whether the row index wrapped (`FRISK8_RING_SPLIT`;
            // round 3 computed row, carry, column, mask, shift and sum on the vector unit for every position: four VALU instructions
            // of the scoring loop's ~52 per position)
#ifndef FRISK8_RING_SPLIT
#define FRISK8_RING_SPLIT 1
#endif
            const uint32_t lane_col0 = RING ? (((rb_q + uint32_t(blk)) & (FRISK8_RING_COLS - 1u)) << 3) : 0u;
            const uint32_t lane_col1 = RING ? (((rb_q + 1u + uint32_t(blk)) & (FRISK8_RING_COLS - 1u)) << 3) : 0u;
            auto ring_uni = [&](int it) __attribute__((always_inline)) -> uint32_t {        // (uniform) slice + row
                const uint32_t rr = rb_r + uint32_t(it);
                const uint32_t cy = rr >= uint32_t(ITS) ? 1u : 0u;
                return slice_off + (((rr - cy * uint32_t(ITS)) * FRISK8_RING_COLS) << 3);
            };
            auto ring_lane = [&](int it) __attribute__((always_inline)) -> uint32_t {       // the lane's column, in bytes
                return (rb_r + uint32_t(it) >= uint32_t(ITS)) ? lane_col1 : lane_col0;
            };
            auto ring_mine = [&](int it) __attribute__((always_inline)) -> uint32_t {
                if (FRISK8_RING_SPLIT) return ring_uni(it) + ring_lane(it);
                const uint32_t rr = rb_r + uint32_t(it);                         // (uniform: the row, and whether it wraps into the next column)
                const uint32_t cy = rr >= uint32_t(ITS) ? 1u : 0u;
                return slice_off + ((((rr - cy * uint32_t(ITS)) * FRISK8_RING_COLS) + ((rb_q + cy + uint32_t(blk)) & (FRISK8_RING_COLS - 1u))) << 3);
            };
            const bool sliding = slide_next;
            // this lane gathers its positions' genome-side values from the table (and parks them in the ring): every lane of a
            // window counted afresh, the lanes that hold a position of the entering range in a window slid into
            const bool lane_new = !sliding || !ring_next || blk * ITS + (ITS - 1) >= P.w - (K - 1) - P.inc;
            ring_next = false;                                  // (true again where this window's scoring loop has run)
            slide_next = slide_pp > 0 && ci + 1 < ce && d.kind == 0 && !jump && cand + 1 < d.cand0 + d.ncand &&
                         st + int64_t(P.inc) + P.w <= d.size;
            if (n > NT * ITS) {
                // a rescued small scaffold (--scaffoldsAll, L211-221: up to 1.75 w bases) longer than this kernel's lanes cover:
                // straight to the wider forms - per WINDOW, so that which kernel scores a window never depends on what else
                // is resident (a rank of a multi-GPU job sees other scaffolds than the one-GPU run)
                if (tid == 0) { const unsigned int slot = atomicAdd(Pk->out_count, 1u); Pk->out_list[slot] = cand; }
                continue;
            }
            uint32_t* misc = misc_base + parity * FRISK8_SLOTS;
            uint32_t* misc_other = misc_base + (parity ^ 1u) * FRISK8_SLOTS;
            parity ^= 1u;

            // ---- stage 1: one pass over the window's positions (a lane owns ITS consecutive ones) -----------
            const int j0 = blk * ITS;
            const int64_t gl = g0 + (j0 < n ? j0 : 0);                       // clamped: loads are unconditional
            const int64_t wi = gl >> 4, mi = gl >> 5;
            const int shc = 32 - int(gl & 15) * 2, shm = 32 - int(gl & 31);
            // (requesting the NEXT window's words after stage 4, to have them in registers here: measured +-0 at K = 8, -2 % at
            //  K = 6, 7 - with three or four workgroups per CU the other windows' waves already cover these two round trips)
            // (all seven words requested before the first is used: one round trip to L2 / HBM, not two)
            const uint32_t w0 = P.codes[wi], w1 = P.codes[wi + 1], w2 = P.codes[wi + 2];
            const uint32_t i0 = P.inv[mi], i1 = P.inv[mi + 1], l0 = P.low[mi], l1 = P.low[mi + 1];
            // sliding: the thread's slide_pp positions of the range that leaves (the predecessor's first inc positions) and of
            // the range that enters (the inc positions before this window's last K - 1), requested with the words above
            uint64_t scode = 0, ecode = 0;              // 32 bases from the thread's first position of either range
            uint32_t sfull = 0, efull = 0;              // a max-mer starts at the thread's position it <-> bit 31 - it
            if (sliding) {
                const int q0 = tid * slide_pp;
                const int cnt = P.inc - q0;             // the thread's positions: min(cnt, slide_pp), none if <= 0
                const int64_t gs = g0 - P.inc + (cnt > 0 ? q0 : 0);
                const int64_t ge = g0 + (P.w - (K - 1) - P.inc) + (cnt > 0 ? q0 : 0);
                const int64_t swi = gs >> 4, smi = gs >> 5, ewi = ge >> 4, emi = ge >> 5;
                const uint32_t s0 = P.codes[swi], s1 = P.codes[swi + 1], s2 = P.codes[swi + 2], si0 = P.inv[smi], si1 = P.inv[smi + 1];
                const uint32_t e0 = P.codes[ewi], e1 = P.codes[ewi + 1], e2 = P.codes[ewi + 2], ei0 = P.inv[emi], ei1 = P.inv[emi + 1];
                auto funnel = [](uint32_t a, uint32_t b, uint32_t c, int sh) -> uint64_t {
                    return (uint64_t(uint32_t(((uint64_t(a) << 32) | b) >> sh)) << 32) | uint32_t(((uint64_t(b) << 32) | c) >> sh);
                };
                auto starts = [](uint32_t inv32, int cnt_, int pp) -> uint32_t {      // K valid bases from the position on
                    uint32_t f = ~inv32;
                    f &= f << 1; f &= f << 2;
                    f &= f << (K - 4);
                    const int m = cnt_ < pp ? cnt_ : pp;
                    return m > 0 ? f & uint32_t(0xFFFFFFFF00000000ull >> m) : 0u;
                };
                scode = funnel(s0, s1, s2, 32 - int(gs & 15) * 2);
                ecode = funnel(e0, e1, e2, 32 - int(ge & 15) * 2);
                sfull = starts(uint32_t(((uint64_t(si0) << 32) | si1) >> (32 - int(gs & 31))), cnt, slide_pp);
                efull = starts(uint32_t(((uint64_t(ei0) << 32) | ei1) >> (32 - int(ge & 31))), cnt, slide_pp);
            }
            const uint32_t chi = uint32_t(((uint64_t(w0) << 32) | w1) >> shc);
            const uint32_t clo = uint32_t(((uint64_t(w1) << 32) | w2) >> shc);
            const uint64_t acode = (uint64_t(chi) << 32) | clo;             // bases j0 .. j0+31, first base in the top bits
            const uint32_t ainv = uint32_t(((uint64_t(i0) << 32) | i1) >> shm);
            const uint32_t alow = uint32_t(((uint64_t(l0) << 32) | l1) >> shm);
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
                // (`ntop_`: the lane's max-mer starts.  The three counts of a lane are at most ITS each, so the sums over a row of 16
                //  lanes fit ten bits and travel through ONE butterfly; the squeeze behind the G + C plane is needed only where some
                //  lane of the wave holds an invalid or soft-masked base: elsewhere the selected positions are the lane's first
                //  popc(actm), and their G + C count is one masked popcount of the codes' high bits.)
                auto tally = [&](uint32_t sel, uint32_t ntop_) {     // sel: the lane's positions to count
                    static_assert(ITS <= 31, "three ten-bit fields: a lane's counts below 32, a pair of rows' sums below 1024");
                    uint32_t gcl;
                    if (__any(sel != actm)) gcl = uint32_t(__popc(sel & gc_plane()));
                    else gcl = uint32_t(__popcll(acode & 0xAAAAAAAAAAAAAAAAull & ~(0xFFFFFFFFFFFFFFFFull >> (2 * __popc(actm)))));
                    uint32_t x = uint32_t(__popc(sel)) | (gcl << 10) | (ntop_ << 20);
                    x = dpp_addu<0xB1>(x); x = dpp_addu<0x4E>(x); x = dpp_addu<0x141>(x); x = dpp_addu<0x140>(x);     // (as wave_sum_u32)
                    const uint32_t a = uint32_t(__builtin_amdgcn_readlane(int(x), 0) + __builtin_amdgcn_readlane(int(x), 16));
                    const uint32_t b = uint32_t(__builtin_amdgcn_readlane(int(x), 32) + __builtin_amdgcn_readlane(int(x), 48));
                    cAll = (a & 1023u) + (b & 1023u);
                    cGC = ((a >> 10) & 1023u) + ((b >> 10) & 1023u);
                    nvalid = (a >> 20) + (b >> 20);
                };
                // a position next to an invalid base or the window's end: the longest valid word there has 0..K-1 bases.
                // Orders <= K-3 count it in the small tables (at order min(run, K-3): lower orders follow by
                // marginalisation); a (K-2)- or (K-1)-base word is not a prefix of any counted max-mer: orphan list.
                auto short_word = [&](uint32_t c16, uint32_t inv8, int rem) {
                    int run = lead_clear8(inv8);
                    run = run < rem ? run : rem;
                    run = run < K ? run : K - 1;                         // (a K-mer inside the window would have been a max-mer)
                    const int rs = run < LVL ? run : LVL;
                    // (PLACE: an orphan reaches the orders <= K-3 through the table sums of stage 3, or - no room - through its owner there)
                    if (rs >= kmin && !(PLACE && run >= K - 2)) {
                        const uint32_t b = uint32_t(table_offset(kmin, rs)) + (c16 >> (2 * K - 2 * rs));
                        atomicAdd(&small32[b >> 1], 1u << ((b & 1u) * 16));
                    }
                    if (run >= K - 2) {
                        const uint32_t slot = atomicAdd(&misc[M_NORPH], 1u);
                        if (slot < FRISK8_ORPH_CAP)
                            orph[slot] = uint16_t(run == K - 1 ? (c16 >> 2) : (0x8000u | ((c16 >> 4) << 2)));
                    }
                };
                const uint32_t shortm = actm & ~fullm;                   // the lane's positions of that kind
                const unsigned long long short_lanes = __ballot(shortm != 0u);
                const bool few = __popcll(short_lanes) <= FRISK8_SHORT_LANES;
                if (!sliding) {
#pragma unroll FRISK8_UNROLL1
                    for (int it = 0; it < ITS; ++it) {
                        const uint32_t bit = 0x80000000u >> it;
                        const uint32_t c16 = code_at(it);
                        if (fullm & bit) bump(c16, std::integral_constant<int, 1>{});
                        else if (!few && (shortm & bit)) short_word(c16, (ainv >> (24 - it)) & 0xFFu, n - (j0 + it));
                    }
                } else {
                    // the table slides: slide_pp positions of the leaving and of the entering range per thread
#pragma unroll 2
                    for (int it = 0; it < slide_pp; ++it) {
                        const uint32_t bit = 0x80000000u >> it;
                        const uint32_t cs = uint32_t(scode >> (64 - 2 * K - 2 * it)) & (NK - 1u);
                        const uint32_t cn = uint32_t(ecode >> (64 - 2 * K - 2 * it)) & (NK - 1u);
                        if (sfull & bit) bump(cs, std::integral_constant<int, -1>{});
                        if (efull & bit) bump(cn, std::integral_constant<int, 1>{});
                    }
                    if (!few) {
                        for (int it = 0; it < ITS; ++it)
                            if (shortm & (0x80000000u >> it)) short_word(code_at(it), (ainv >> (24 - it)) & 0xFFu, n - (j0 + it));
                    }
                }
                // Few lanes with such positions (a window's last lane: K-1 of them in a row; the edges of an invalid run): one
                // pass per LANE with its positions spread over the wave's lanes, instead of one exec-masked pass per position
                // inside the loop above - the other three waves of the workgroup wait at the barrier for this lane.
                if (few) {
                    for (unsigned long long rest = short_lanes; rest; rest &= rest - 1) {
                        const int src = int(__ffsll((long long)rest)) - 1;
                        const uint32_t sm = uint32_t(__builtin_amdgcn_readlane(int(shortm), src));
                        const uint32_t si = uint32_t(__builtin_amdgcn_readlane(int(ainv), src));
                        const uint32_t sh = uint32_t(__builtin_amdgcn_readlane(int(uint32_t(acode >> 32)), src));
                        const uint32_t sl = uint32_t(__builtin_amdgcn_readlane(int(uint32_t(acode)), src));
                        const int sleft = __builtin_amdgcn_readlane(nleft, src);
                        const int it = lane < ITS ? lane : 0;
                        if (lane < ITS && ((sm >> (31 - it)) & 1u)) {
                            const uint64_t sc = (uint64_t(sh) << 32) | sl;
                            short_word(uint32_t(sc >> (64 - 2 * K - 2 * it)) & (NK - 1u), (si >> (24 - it)) & 0xFFu, sleft - it);
                        }
                    }
                }
                // (the order-1 table could give these two when kmin = 1, but only after the marginalisation - which now runs
                //  inside stage 3, where the window constants made from them are already needed)
                tally(actm & vld & ~alow, uint32_t(__popc(fullm)));
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
            STAMP(1)
            __syncthreads();
            STAMP(2)
            if (tid < FRISK8_SLOTS) misc_other[tid] = 0;        // the previous window's counters: nobody reads them now

            // ---- stage 2: C_5[q] = D_5[q] + (sum of the 64 order-8 counters below q); grand total for the overflow check
            const uint32_t o5 = uint32_t(table_offset(kmin, LVL));         // the small tables' top order: K-3
            if constexpr (!FUSED) {
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
                __syncthreads();
            }
            STAMP(3)
            // The orders below follow inside a wave, no LDS round trip between the levels.  LVL >= 4 (K = 7, 8): as part of
            // stage 3, where thread t holds the 4-mer t anyway.  LVL = 3 (K = 6): here - one wave, lane l = the 3-mer l.
            if constexpr (LVL == 3) {
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
                __syncthreads();
            }
            STAMP(4)

            auto uni = [](uint32_t x) -> uint32_t { return __builtin_amdgcn_readfirstlane(x); };
            const uint32_t upAll = uni(misc[M_UPA]), upGC = uni(misc[M_UPG]);
            const int64_t S = int64_t(upAll);                       // windowSpace (L380): uppercase A + T + G + C
            const int64_t nn = n - S;                               // nnTotal of the window
            const bool keep = !(double(nn) >= 0.3 * double(n));     // N filter (L237-241 / L213)
            uint32_t status = (jump ? ROW_JUMPBACK : 0u);
            const uint32_t nvalid_top = uni(misc[M_NVALID]);
            const int n_orph = int(uni(misc[M_NORPH]));
            // (fused form: the table's total is not known yet - that half of the test follows stage 3)
            const bool wrapped = (!FUSED && uni(misc[M8_TSUM]) != nvalid_top) || n_orph > FRISK8_ORPH_CAP;
            const uint32_t safe_code = uni(misc[M8_SAFE]);

            auto zero_own = [&]() {             // every max-mer position clears its dword (all reads are behind a barrier)
#pragma unroll 4
                for (int it = 0; it < ITS; ++it)
                    if (fullm & (0x80000000u >> it)) t8[code_at(it) >> SHW] = 0u;
            };
            // a counter wrapped, or too many orphans: the next wider form (8-bit, then scan_kernel.h's 16-bit) redoes this window
            // from scratch
            auto hand_over = [&]() {
                if (!slide_next) clear_t8();
                clear_small();
                if (tid == 0) { const unsigned int slot = atomicAdd(Pk->out_count, 1u); Pk->out_list[slot] = cand; }
                __syncthreads();
            };
            if (wrapped) { hand_over(); continue; }
            if (!keep) {
                // dropped by the N filter (the composition comes from popcounts of stage 1: reliable whatever the counters did)
                if (!slide_next) { if (CLEAR_ALL) clear_t8(); else zero_own(); }
                clear_small();
                if (tid == 0) {
                    Pk->seq_index[row] = dsi; Pk->start[row] = rep_start; Pk->stop[row] = rep_stop;
                    Pk->status[row] = status;
                    const double qnan = __longlong_as_double(0x7FF8000000000000LL);
                    Pk->kld[row] = qnan; Pk->gc[row] = qnan;
                    if (P.flags & 1u) { Pk->pi[row] = qnan; Pk->si[row] = qnan; Pk->cri[row] = qnan; }
                    if (DEBUG && P.dbg_meta) {      // (dropped rows are not compared; keep the dump well defined)
                        P.dbg_meta[row * 3 + 0] = n; P.dbg_meta[row * 3 + 1] = 0; P.dbg_meta[row * 3 + 2] = nn;
                    }
                }
                __syncthreads();
                continue;
            }
            if (tid == 0) { Pk->seq_index[row] = dsi; Pk->start[row] = rep_start; Pk->stop[row] = rep_stop; }

            // The orphan list: its first four entries in scalar registers.  A run-(K-1) entry is its (K-1)-mer; a run-(K-2) entry
            // has bit 15 set and holds its (K-2)-mer << 2.  o6[k] = the (K-2)-mer of entry k (both kinds count towards c6), o7[k] =
            // the (K-1)-mer of a run-(K-1) entry.  A window without invalid bases has exactly one of each kind (its tail).
            uint32_t o6[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, o7[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            uint32_t o7c[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};     // the (K-1)-mers again, compacted: n7 of them
            int n7 = 0, n_list = 0;             // n_list: orphans that stayed on the list (all of them without PLACE)
            uint32_t rest_mask = 0;             // ... those beyond the first four, as bits over the list's entries
            static_assert(L::orphans % 8 == 0 && FRISK8_ORPH_CAP <= 31, "the orphan list: eight-byte aligned, one mask bit per entry (bit 31 of the mask word: SIDE)");
            // (called behind stage 3's barrier: `placed` = the entries that were folded into the table there)
            auto load_orphans = [&](uint32_t placed) {
                uint32_t todo = (n_orph >= 32 ? 0xFFFFFFFFu : ((1u << n_orph) - 1u)) & ~placed;
                n_list = __popc(todo);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (todo) {
                        const int j = __ffs(int(todo)) - 1;
                        todo &= todo - 1u;
                        const uint32_t e = uni(orph[j]);
                        o6[k] = (e >> 2) & (NK / 16u - 1u);
                        if (!(e & 0x8000u)) o7[k] = e;
                    }
                }
                rest_mask = todo;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (o7[k] != 0xFFFFFFFFu) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (q == n7) o7c[q] = o7[k];
                        ++n7;
                    }
                }
            };
            // what a max-mer position reads, all of it addressed by the code alone (so it can be fetched ahead of use):
            // genome-side value, the order-8 counters of its 6-mer / 7-mer / itself, the shared-prefix sums
            struct Fetched { double Ig, A5; uint32_t W5, c8, w7, roff; uint4 w6; };       // (SIDE: W5 carries the side count of the code's 4-mer in its top bits)
            // (`it`: the lane's position the code belongs to, whose genome-side value waits in the ring - a 1.0 where the position
            //  starts no max-mer: the stand-in it scores has weight 0, any finite number will do;
            //  mode: 0 counts only; 1 the scoring loop of a wave whose lanes all read the ring; 2 ... of a wave with lanes that gather)
            auto fetch = [&](uint32_t c16, int it = 0, int mode = 0) __attribute__((always_inline)) -> Fetched {
                Fetched f;
                f.roff = 0;
                if constexpr (RING) {
                    f.Ig = 1.0;
                    if (mode) {
                        f.roff = ring_mine(it);
                        if (FRISK8_RING_SPLIT && mode == 1)      // every lane reads the ring: scalar base (slice + row) + the lane's column
                            f.Ig = *reinterpret_cast<const double*>((ring + ring_uni(it)) + ring_lane(it));
                        else
                            f.Ig = *reinterpret_cast<const double*>(ring + ((mode == 2 && lane_new) ? (c16 << 3) : f.roff));
                    }
                } else {
                f.Ig = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(P.ig) + (c16 << 3));   // c16 < 4^K always
                }
                if (BITS == 8) {        // the counter of code c is byte c of the table
                    f.c8 = t8b[c16];
                    f.w7 = *reinterpret_cast<const uint32_t*>(t8b + (c16 & ~3u));
                    f.w6 = *reinterpret_cast<const uint4*>(t8b + (c16 & ~15u));
                } else {                // the sixteen nibbles of the 6-mer c >> 4 are the 8 bytes at 8 (c >> 4)
                    f.c8 = 0;
#if FRISK8_W7_READ
                    f.w7 = *reinterpret_cast<const uint16_t*>(t8b + ((c16 >> 2) << 1));        // the 7-mer's four nibbles, read on their own
#else
                    f.w7 = 0;
#endif
                    const uint2 x = *reinterpret_cast<const uint2*>(t8b + ((c16 >> 4) << 3));
                    f.w6 = make_uint4(x.x, x.y, 0u, 0u);
                }
                if (FRISK8_PRE_SPLIT) {
                    f.W5 = *reinterpret_cast<const uint32_t*>(lds + L::pre + NL * 8 + ((c16 >> 6) << 2));
                    f.A5 = *reinterpret_cast<const double*>(lds + L::pre + ((c16 >> 6) << 3));
                } else {
                    const Pre8* e = reinterpret_cast<const Pre8*>(lds + L::pre + __umul24(c16 >> 6, 12u));     // (one v_mul_u32_u24)
                    f.W5 = e->W;
                    f.A5 = e->A;
                }
                return f;
            };
            // counts of the three top orders of the max-mer c16 as the order-K table holds them: without the orphans
            // (tper, SIDE: (code ^ code >> 8) & 0xFF of the POSITION's own max-mer - 0xFFFFFFFF: computed here from c16)
            auto table_counts = [&](const Fetched& f, uint32_t c16, uint32_t& c8, uint32_t& c7, uint32_t& c6, uint32_t tper = 0xFFFFFFFFu) __attribute__((always_inline)) {
                if (BITS == 8) {
                    c8 = f.c8;
                    c7 = __builtin_amdgcn_sad_u8(f.w7, 0u, 0u);
                    c6 = __builtin_amdgcn_sad_u8(f.w6.x, 0u, __builtin_amdgcn_sad_u8(f.w6.y, 0u, __builtin_amdgcn_sad_u8(f.w6.z, 0u, __builtin_amdgcn_sad_u8(f.w6.w, 0u, 0u))));
                } else {
#if FRISK8_W7_READ
                    const uint32_t w7 = f.w7;
#else
                    // the 7-mer's four nibbles are 16 of the 64 bits already here: one 64-bit shift instead of an LDS read of its own
                    // (the dot product's 0x1111 ignores what the shift leaves above them)
                    const uint32_t w7 = uint32_t(((uint64_t(f.w6.y) << 32) | f.w6.x) >> ((c16 & 12u) << 2));
#endif
                    // (SIDE: the period-4 max-mer below the code's (K-2)-mer / (K-1)-mer / the code itself, where there is one)
                    uint32_t s6 = 0u, s7 = 0u, s8 = 0u;
                    if constexpr (SIDE) {
                        const uint32_t t = tper != 0xFFFFFFFFu ? tper : ((c16 ^ (c16 >> 8)) & 0xFFu);
                        const uint32_t sd = f.W5 >> 23;
                        s6 = t < 16u ? sd : 0u; s7 = t < 4u ? sd : 0u; s8 = t == 0u ? sd : 0u;
                    }
                    c8 = __builtin_amdgcn_ubfe(w7, (c16 & 3u) * 4u, 4u) + s8;
                    c7 = __builtin_amdgcn_udot8(w7, 0x1111u, s7, false);
                    c6 = __builtin_amdgcn_udot8(f.w6.x, 0x11111111u, __builtin_amdgcn_udot8(f.w6.y, 0x11111111u, s6, false), false);
                }
            };
            // ... and the orphans on top (any number of them: row metadata and the debug dump; stage 4 has its own, unrolled)
            auto add_orphans = [&](uint32_t c16, uint32_t& c7, uint32_t& c6) __attribute__((always_inline)) {
                const uint32_t q6 = c16 >> 4, q7 = c16 >> 2;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    c7 += (q7 == o7[k]) ? 1u : 0u;
                    c6 += (q6 == o6[k]) ? 1u : 0u;
                }
                for (uint32_t m = rest_mask; m; m &= m - 1u) {
                    const uint32_t e = orph[__ffs(int(m)) - 1];
                    c7 += (q7 == e) ? 1u : 0u;
                    c6 += (q6 == ((e >> 2) & (NK / 16u - 1u))) ? 1u : 0u;
                }
            };
            // count of the x-mer c in this window (row metadata, RIP, debug dump)
            auto count = [&](int x, uint32_t c) -> uint32_t {
                if (x <= LVL) return small16[table_offset(kmin, x) + c];
                uint32_t c8, c7, c6;
                const uint32_t c16 = c << (2 * (K - x));
                table_counts(fetch(c16), c16, c8, c7, c6);
                add_orphans(c16, c7, c6);
                return x == K ? c8 : (x == K - 1 ? c7 : c6);
            };
            // ---- stage 3: window constants r_x = 4^x / D_x, D_x = (S-(x-1))*2 (L401-409), and the shared prefix tables
            uint32_t pend0 = 0xFFFFFFFFu, pend1 = 0xFFFFFFFFu;      // PLACE: (list entry << 16 | fake code) of the orphans this thread folded in
            double r_lane = 0.0;
            if (lane <= 8) r_lane = div_exact(double(1u << (2 * lane)), double(int32_t((S - (lane - 1)) * 2)));
            auto r_of = [&](int x) -> double {
                return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(r_lane), x),
                                        __builtin_amdgcn_readlane(__double2loint(r_lane), x));
            };
            // the top order's terms by count: {1 / c, c^2 r_K} for c < 16 (c = 0: a position that starts no max-mer, weight 0)
            if (BITS == 4 && (SIDE || tid < 16)) rstab[tid & 255].y = double(uint32_t((tid & 255) * (tid & 255))) * r_of(K);     // (SIDE: 256 entries)
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
                if constexpr (LV >= 4) {
                    // K = 7, 8: thread t takes the 4-mer t.  Its count is final (K = 7) or follows from its four 5-mers (K = 8);
                    // the 3-mers are sums over quads of lanes, the 2-mers over rows of 16, the 1-mer i over wave i - DPP, no
                    // LDS round trip and no barrier between the levels (this was a stage of its own).  The finished counts go
                    // back to the small tables (RIP indices, debug dump) and, from registers, into the prefix sums.
                    static_assert(NT >= 256 && NT % 64 == 0, "thread t <-> 4-mer t: the first four waves (the others sit this stage out)");
                    if (NT == 256 || tid < 256) {
                    const uint32_t q4 = uint32_t(tid);
                    // PLACE: the orphans under this thread's 4-mer go into the table before the thread sums that part of it (its
                    // own LDS operations stay in order).  `extra`: orphans that found no room, per (K-3)-mer of the thread (four
                    // 16-bit fields at K = 8, one count at K = 7) - what stage 1 would have added to the small table for them.
                    uint64_t extra = 0;
                    if constexpr (PLACE) {
                        pend0 = 0xFFFFFFFFu; pend1 = 0xFFFFFFFFu;
                        for (int k = 0; k < n_orph; ++k) {              // (uniform; normally two entries: the window's tail)
                            const uint32_t e = orph[k];
                            const uint32_t q6 = (e >> 2) & (NK / 16u - 1u);                     // the orphan's (K-2)-mer
                            if ((q6 >> (2 * (K - 2) - 8)) != q4) continue;
                            uint32_t slot = 0xFFFFFFFFu;                 // the fake max-mer's code
                            if constexpr (BITS == 4) {
                                const uint2 g = *reinterpret_cast<const uint2*>(t8b + q6 * 8u);   // sixteen counters: four (K-1)-mers of four
                                if (!(e & 0x8000u)) {                    // a (K-1)-mer: a zero counter among its four children
                                    const uint32_t f = ((e & 2u) ? g.y : g.x) >> ((e & 1u) * 16u) & 0xFFFFu;
                                    uint32_t z = ~(f | (f >> 1) | (f >> 2) | (f >> 3)) & 0x1111u;
                                    // (SIDE: the period-4 child of x0..x6 - the one that ends in x3, if x4 x5 x6 = x0 x1 x2 - has a zero
                                    //  counter in the table whatever its count: never the fake's place)
                                    if (SIDE && ((e ^ (e >> 8)) & 0x3Fu) == 0u) z &= ~(1u << (((e >> 6) & 3u) * 4u));
                                    if (z) slot = ((e & 0x3FFFu) << 2) | uint32_t((__ffs(int(z)) - 1) >> 2);
                                } else {                                 // a (K-2)-mer: a (K-1)-mer below it whose four children are all zero
                                    uint32_t zz = ((g.x & 0xFFFFu) == 0u ? 1u : 0u) | ((g.x >> 16) == 0u ? 2u : 0u) |
                                                  ((g.y & 0xFFFFu) == 0u ? 4u : 0u) | ((g.y >> 16) == 0u ? 8u : 0u);
                                    // (SIDE: the (K-1)-mer x0..x5 x2 may occur through its period-4 child although its four counters are zero)
                                    if (SIDE && ((q6 ^ (q6 >> 8)) & 0xFu) == 0u) zz &= ~(1u << ((q6 >> 6) & 3u));
                                    if (zz) slot = ((q6 << 2) | uint32_t(__ffs(int(zz)) - 1)) << 2;
                                }
                            } else {
                                const uint4 g = *reinterpret_cast<const uint4*>(t8b + q6 * 16u);
                                if (!(e & 0x8000u)) {
                                    const uint32_t sub = e & 3u;
                                    const uint32_t f = sub == 0u ? g.x : (sub == 1u ? g.y : (sub == 2u ? g.z : g.w));
                                    const uint32_t z = (f - 0x01010101u) & ~f & 0x80808080u;          // (the lowest set bit marks a zero byte)
                                    if (z) slot = ((e & 0x3FFFu) << 2) | uint32_t((__ffs(int(z)) - 1) >> 3);
                                } else {
                                    const uint32_t zz = (g.x == 0u ? 1u : 0u) | (g.y == 0u ? 2u : 0u) | (g.z == 0u ? 4u : 0u) | (g.w == 0u ? 8u : 0u);
                                    if (zz) slot = ((q6 << 2) | uint32_t(__ffs(int(zz)) - 1)) << 2;
                                }
                            }
                            if (slot != 0xFFFFFFFFu && pend1 == 0xFFFFFFFFu) {
                                atomicAdd(&t8[slot >> SHW], 1u << ((slot & PERM) * BITS));
                                // (the entry will name the fake counter, which comes out again after the window - written behind
                                //  this stage's barrier: the other threads are still reading the list)
                                if (pend0 == 0xFFFFFFFFu) pend0 = (uint32_t(k) << 16) | slot; else pend1 = (uint32_t(k) << 16) | slot;
                                atomicOr(&misc[M8_PMASK], 1u << k);
                                atomicAdd(&misc[M8_NPLACED], 1u);
                            } else {
                                extra += K == 8 ? (1ull << (16u * ((q6 >> 2) & 3u))) : 1ull;
                            }
                        }
                    }
                    // c5[m] = the count of the 5-mer 4 q4 + j5[m] (K = 8).  Fused form: the sums of the 64 order-8 counters below each,
                    // read here - thread t owns 128 (256) contiguous table bytes; which 5-mer comes first and which 16 bytes of it
                    // rotate with the lane, so that the eight lanes of a bank group never meet (a b128 read takes eight passes anyway)
                    uint32_t c5[4] = {0u, 0u, 0u, 0u}, j5[4] = {0u, 1u, 2u, 3u};
                    uint32_t side_mine = 0;             // SIDE: the count of the period-4 max-mer (q4)(q4)
                    if constexpr (LV == 5) {
                        const uint2 ch = *reinterpret_cast<const uint2*>(small16 + ox[5] + 4 * q4);     // D_5 (fused) or C_5, four u16
                        if constexpr (FUSED) {
                            // (SIDE: the period-4 max-mer (q4)(q4) sits below the thread's (K-3)-mer 4 q4 + x0)
                            if constexpr (SIDE) { side_mine = side8[q4]; extra += uint64_t(side_mine) << (16u * (q4 >> 6)); }
                            const uint64_t d64 = ((uint64_t(ch.y) << 32) | ch.x) + extra;
                            const unsigned char* mine = t8b + q4 * (BITS == 8 ? 256u : 128u);
                            // (all of the thread's table bytes requested before the first is summed: one LDS round trip, not eight)
                            constexpr int NH = BITS == 8 ? 4 : 2;
                            uint4 x[4][NH];
#pragma unroll
                            for (int m = 0; m < 4; ++m) {
                                const uint32_t j = (uint32_t(m) + uint32_t(tid)) & 3u;
#pragma unroll
                                for (int h = 0; h < NH; ++h) {
                                    const uint32_t hh = BITS == 8 ? ((uint32_t(h) + (uint32_t(tid) >> 1)) & 3u) : ((uint32_t(h) + (uint32_t(tid) >> 2)) & 1u);
                                    x[m][h] = *reinterpret_cast<const uint4*>(mine + j * (BITS == 8 ? 64u : 32u) + hh * 16u);
                                }
                            }
                            uint32_t tot = side_mine;
#pragma unroll
                            for (int m = 0; m < 4; ++m) {
                                const uint32_t j = (uint32_t(m) + uint32_t(tid)) & 3u;
                                uint32_t sm = 0;
#pragma unroll
                                for (int h = 0; h < NH; ++h) {
                                    if (BITS == 8) {
                                        sm = __builtin_amdgcn_sad_u8(x[m][h].x, 0u, sm); sm = __builtin_amdgcn_sad_u8(x[m][h].y, 0u, sm);
                                        sm = __builtin_amdgcn_sad_u8(x[m][h].z, 0u, sm); sm = __builtin_amdgcn_sad_u8(x[m][h].w, 0u, sm);
                                    } else {
                                        sm = __builtin_amdgcn_udot8(x[m][h].x, 0x11111111u, sm, false); sm = __builtin_amdgcn_udot8(x[m][h].y, 0x11111111u, sm, false);
                                        sm = __builtin_amdgcn_udot8(x[m][h].z, 0x11111111u, sm, false); sm = __builtin_amdgcn_udot8(x[m][h].w, 0x11111111u, sm, false);
                                    }
                                }
                                tot += sm;
                                j5[m] = j;
                                c5[m] = sm + (uint32_t(d64 >> (16u * j)) & 0xFFFFu);
                                if (DEBUG) small16[ox[5] + 4 * q4 + j] = uint16_t(c5[m]);       // (only the dump reads C_5 again)
                            }
                            // the table's grand total, for the overflow test behind this stage's barrier
                            tot = wave_sum_u32(tot);
                            if (lane == 0 && tot) atomicAdd(&misc[M8_TSUM], tot);
                        } else {
                            c5[0] = ch.x & 0xFFFFu; c5[1] = ch.x >> 16; c5[2] = ch.y & 0xFFFFu; c5[3] = ch.y >> 16;
                        }
                    }
                    uint32_t cx[5] = {0u, 0u, 0u, 0u, 0u};
                    uint32_t below4 = 0;                // K = 7, fused form: the 64 order-7 counters (bytes) below the 4-mer
                    if constexpr (LV == 4 && FUSED) {
                        const unsigned char* mine = t8b + q4 * 64u;
                        uint4 x[4];
#pragma unroll
                        for (int h = 0; h < 4; ++h) x[h] = *reinterpret_cast<const uint4*>(mine + ((uint32_t(h) + (uint32_t(tid) >> 1)) & 3u) * 16u);
#pragma unroll
                        for (int h = 0; h < 4; ++h) {
                            below4 = __builtin_amdgcn_sad_u8(x[h].x, 0u, below4); below4 = __builtin_amdgcn_sad_u8(x[h].y, 0u, below4);
                            below4 = __builtin_amdgcn_sad_u8(x[h].z, 0u, below4); below4 = __builtin_amdgcn_sad_u8(x[h].w, 0u, below4);
                        }
                        const uint32_t tot = wave_sum_u32(below4);
                        if (lane == 0 && tot) atomicAdd(&misc[M8_TSUM], tot);
                    }
                    // (the D counts of all four lower orders requested up front, before the DPP sums that they are added to - an order
                    //  that is off reads a harmless bin of the first table)
                    const uint32_t d4 = small16[ox[4] + q4], d3 = small16[ox[3] + (q4 >> 2)], d2 = small16[ox[2] + (q4 >> 4)], d1 = small16[q4 >> 6];
                    if (kmin <= 4) {
                        cx[4] = d4 + below4 + (LV == 4 ? uint32_t(extra) : 0u);
                        if constexpr (LV == 4 && FUSED) small16[ox[4] + q4] = uint16_t(cx[4]);
                        if constexpr (LV == 5) {
                            cx[4] += c5[0] + c5[1] + c5[2] + c5[3];
                            small16[ox[4] + q4] = uint16_t(cx[4]);
                        }
                        if (kmin <= 3) {
                            uint32_t qs = dpp_addu<0xB1>(cx[4]);
                            qs = dpp_addu<0x4E>(qs);                                             // the quad's sum, in all four lanes
                            cx[3] = d3 + qs;                                                     // (every lane of the quad has read D_3; one writes C_3)
                            if ((lane & 3) == 0) small16[ox[3] + (q4 >> 2)] = uint16_t(cx[3]);
                            if (kmin <= 2) {
                                uint32_t rs = dpp_addu<0xB1>((lane & 3) == 0 ? cx[3] : 0u);
                                rs = dpp_addu<0x4E>(rs); rs = dpp_addu<0x141>(rs); rs = dpp_addu<0x140>(rs);   // the row's four C_3, in all 16 lanes
                                cx[2] = d2 + rs;
                                if ((lane & 15) == 0) small16[ox[2] + (q4 >> 4)] = uint16_t(cx[2]);
                                if (kmin <= 1) {
                                    const uint32_t ws = __builtin_amdgcn_readlane(int(cx[2]), 0) + __builtin_amdgcn_readlane(int(cx[2]), 16) +
                                                        __builtin_amdgcn_readlane(int(cx[2]), 32) + __builtin_amdgcn_readlane(int(cx[2]), 48);
                                    cx[1] = d1 + ws;
                                    if (lane == 0) small16[q4 >> 6] = uint16_t(cx[1]);
                                }
                            }
                        }
                    }
                    uint32_t W4 = 0;
                    double A4 = 0.0;
#pragma unroll
                    for (int x = 1; x <= 4; ++x) {                  // (same operations in the same order as the generic loop below: same bits)
                        const double cd = double(cx[x]);
                        W4 += (cx[x] & wm[x]) << (2 * x);
                        A4 = __builtin_fma(cd * cd, rx[x], A4);
                    }
                    if constexpr (LV == 4) {
                        put_pre(q4, A4, W4);
                    } else {
                        double A5[4];
                        uint32_t W5[4];
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            const double cd = double(c5[m]);
                            W5[m] = W4 + ((c5[m] & wm[5]) << 10);
                            A5[m] = __builtin_fma(cd * cd, rx[5], A4);
                        }
                        if constexpr (FUSED) {          // (the four entries in the lane's rotated order)
#pragma unroll
                            for (int m = 0; m < 4; ++m) put_pre(4 * q4 + j5[m], A5[m], W5[m], side_mine);
                            // (the sample of the adaptive width counts the windows that would have wrapped a 4-bit counter without the side table)
                            if (SIDE && (ROLE & 1) && side_mine >= 16u) atomicOr(&misc[M8_PMASK], 0x80000000u);
                        } else {
                            auto lo = [](double x) -> uint32_t { return uint32_t(__double2loint(x)); };
                            auto hi = [](double x) -> uint32_t { return uint32_t(__double2hiint(x)); };
                            if (FRISK8_PRE_SPLIT) {
                                uint4* oa = reinterpret_cast<uint4*>(preA + 4 * q4);
                                oa[0] = make_uint4(lo(A5[0]), hi(A5[0]), lo(A5[1]), hi(A5[1]));
                                oa[1] = make_uint4(lo(A5[2]), hi(A5[2]), lo(A5[3]), hi(A5[3]));
                                *reinterpret_cast<uint4*>(preW + 4 * q4) = make_uint4(W5[0], W5[1], W5[2], W5[3]);
                            } else {
                            uint4* out = reinterpret_cast<uint4*>(pre + 4 * q4);        // four entries = 48 bytes, 16-byte aligned
                            out[0] = make_uint4(lo(A5[0]), hi(A5[0]), W5[0], lo(A5[1]));
                            out[1] = make_uint4(hi(A5[1]), W5[1], lo(A5[2]), hi(A5[2]));
                            out[2] = make_uint4(W5[2], lo(A5[3]), hi(A5[3]), W5[3]);
                            }
                        }
                    }
                    }
                } else {
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
                        put_pre(c, A, W);
                    }
                }
            }
            __syncthreads();
            STAMP(5)
            const uint32_t pmask_raw = PLACE ? uni(misc[M8_PMASK]) : 0u;
            const uint32_t placed = pmask_raw & 0x7FFFFFFFu;                        // entries of the orphan list that went into the table
            if constexpr (PLACE) {
                if (pend0 != 0xFFFFFFFFu) orph[pend0 >> 16] = uint16_t(pend0);
                if (pend1 != 0xFFFFFFFFu) orph[pend1 >> 16] = uint16_t(pend1);
            }
            // the fakes out again, where the table is not about to be cleared as a whole: subtracted where the table lives on
            // (a successor slides into it), their words zeroed where every position clears its own word
            auto remove_fakes = [&]() {
                if (PLACE && placed && tid < FRISK8_ORPH_CAP && ((placed >> tid) & 1u) && (slide_next || !CLEAR_ALL)) {
                    const uint32_t slot = orph[tid];
                    if (slide_next) atomicSub(&t8[slot >> SHW], 1u << ((slot & PERM) * BITS));
                    else t8[slot >> SHW] = 0u;
                }
            };
            if constexpr (FUSED) {          // the table's total against the number of max-mer positions (and fakes): a counter wrapped
                if (uni(misc[M8_TSUM]) != nvalid_top + (PLACE ? uni(misc[M8_NPLACED]) : 0u)) {
                    if (PLACE && placed) __syncthreads();          // (the owners' entries above, before other threads read them)
                    remove_fakes(); hand_over(); continue;
                }
            }
            // (the sample of the adaptive width: windows that reach the scoring loop, and those among them that a plain 4-bit table would
            //  have handed on - a side count of 16+)
            if (SIDE && (ROLE & 1) && tid == 0) { atomicAdd(Pk->out_count + 3, 1u); if (pmask_raw >> 31) atomicAdd(Pk->out_count + 2, 1u); }
            load_orphans(placed);

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
                Pk->gc[row] = __longlong_as_double((long long)((uint64_t(uint32_t(S)) << 32) | upGC));
                if (P.flags & 1u) {             // RIP indices (L474-495); codes: AT=1 TA=4 TG=6 GT=9 CA=12 AC=3
                    const double qnan = __longlong_as_double(0x7FF8000000000000LL);
                    const uint32_t AT = count(2, 1), TA = count(2, 4), TG = count(2, 6), GT = count(2, 9), CA = count(2, 12), AC = count(2, 3);
                    const double pi = AT > 0 ? double(TA) / double(AT) : qnan;
                    const double si = (AC + GT) > 0 ? double(CA + TG) / double(AC + GT) : qnan;
                    Pk->pi[row] = pi;
                    Pk->si[row] = si;
                    Pk->cri[row] = (pi == 0.0 || si == 0.0) ? qnan : pi - si;                     // "if PI and SI" (L491)
                }
            }

            // ---- stage 4: every max-mer position: window-side IVOM in closed form, genome side gathered, and the sums
            //      Sw = sum Iw/c8,  Sg = sum Ig/c8,  T = sum Iw ln(Iw/Ig)/c8  over POSITIONS (= sums over distinct max-mers)
            const double r6 = r_of(K - 2), r7 = r_of(K - 1), r8 = r_of(K);      // (named for K = 8: the three orders above the prefix)
            double sw = 0.0, sg = 0.0, stt = 0.0;
#if FRISK8_PRIO
            // The long scoring loop yields issue slots to the short stages of the other workgroups' windows: those are chains of
            // dependent steps between barriers, where a lost slot delays four waves, while a scoring wave has work for every slot
            // it gets.  Measured +3..4 % (all short stages high, scoring low; raising only some of them: less).
            __builtin_amdgcn_s_setprio(0);
#endif
            // (a << SH) + b in one instruction (the compiler's own choice for the weight below is two shifts, a shift-add and an add3)
            auto shl_add = [](uint32_t a, auto sh, uint32_t b) __attribute__((always_inline)) -> uint32_t {
                uint32_t r;
                asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(decltype(sh)::value), "v"(b));
                return r;
            };
            // One position, its counts known.  `sel`: the position's index into the {1/c, c^2 r_K} table - its top count, or 0
            // where it starts no max-mer: such a position scores the stand-in code (a real max-mer of this window: finite values)
            // with weight 0 and adds exactly +0.0 to every sum, so no term needs a mask.
            auto score_one = [&](const Fetched& f, uint32_t c8, uint32_t c7, uint32_t c6, uint32_t sel, bool on)
                                 __attribute__((always_inline)) {
                double2 rs;                                                   // {1/c8 (1.0 for the 19 in 20 max-mers seen once), c8^2 r8}
                if constexpr (BITS == 4) rs = rstab[sel & (SIDE ? 255u : 15u)];            // (SIDE, beyond 15: the 8-bit form's values)
                else {
                    rs = make_double2(rctab[sel & 15u], double(__umul24(sel, sel)) * r8);     // (the same product, rounded alike)
                    if (__builtin_expect(__any(c8 >= 16u), 0)) {              // (wave-uniform, rare: low-complexity sequence)
                        if (on && c8 >= 16u) {                                // beyond the table: reciprocal + two Newton steps
                            const double dc = double(c8);
                            double r = __builtin_amdgcn_rcp(dc);
                            r = __builtin_fma(r, __builtin_fma(-dc, r, 1.0), r);
                            r = __builtin_fma(r, __builtin_fma(-dc, r, 1.0), r);
                            rs.x = r;
                        }
                    }
                }
                const uint32_t W = shl_add(c8, std::integral_constant<int, 2 * K>{},
                                           shl_add(c7, std::integral_constant<int, 2 * K - 2>{},
                                                   shl_add(c6, std::integral_constant<int, 2 * K - 4>{}, SIDE ? (f.W5 & 0x7FFFFFu) : f.W5)));
                // c^2 exactly, as integers (< 2^32), then 4^x / D_x times it; the top order's term comes rounded from the table
                double A = __builtin_fma(double(__umul24(c6, c6)), r6, f.A5);
                A = __builtin_fma(double(__umul24(c7, c7)), r7, A);
                A = A + rs.y;
                // Iw = A/W and Iw/Ig with ONE reciprocal: ratio = A / (W * Ig), Iw = ratio * Ig.  v_rcp_f64 (24.4 bits) + one
                // Newton step = 2^-48.8: the ratio carries a relative error of ~2e-15 - the level of the log table's - instead
                // of being the correctly rounded quotient (two more instructions)
                const double den = double(W) * f.Ig;
                double rr = __builtin_amdgcn_rcp(den);
                rr = __builtin_fma(rr, __builtin_fma(-den, rr, 1.0), rr);
                const double ratio = A * rr;
                const double Igr = f.Ig * rs.x;                               // this position's share of Ig ...
                const double Iwr = ratio * Igr;                               // ... and of Iw
                const double ln = log_tab_n<LOGN, LOGDEG>(ratio, logtab);
                sw += Iwr;
                sg += Igr;
                stt = __builtin_fma(Iwr, ln, stt);
            };
            // the lane's codes and flags again, opaque to the optimiser: without this it keeps every position's pre-shifted
            // code variants of stage 1 alive across the whole window (60 registers) instead of re-deriving them here
            uint32_t ah = uint32_t(acode >> 32), al = uint32_t(acode), fm4 = fullm;
            asm volatile("" : "+v"(ah), "+v"(al), "+v"(fm4));
            const uint64_t acode4 = (uint64_t(ah) << 32) | al;
            auto raw4_at = [&](int it) -> uint32_t { return uint32_t(acode4 >> (64 - 2 * K - 2 * it)) & (NK - 1u); };
            // SIDE: the period test of all the lane's positions in one go - the low byte of the max-mer at position `it`, in
            // acode4 ^ (acode4 >> 8), is its bases 4..7 xor its bases 0..3.  (A position that starts no max-mer scores a stand-in code
            // with weight 0 and adds +0.0 to every sum whatever counts it reads - finite is all they need to be - so the scoring
            // loop takes the position's own period byte there too: no select.)
            const uint64_t aper = SIDE ? (acode4 ^ (acode4 >> 8)) : 0ull;
            auto per4_at = [&](int it) -> uint32_t { return SIDE ? (uint32_t(aper >> (64 - 2 * K - 2 * it)) & 0xFFu) : 0xFFFFFFFFu; };
            // Shape of the scoring loop, measured per K (bench shard / C2 shape, M windows/s):
            //   K = 8 (LDS allows 3 / 2 workgroups per CU): unrolled, groups of 2: 44.8 / 36.2; rolled, groups of 1: 43.2 / 35.6
            //   K = 6, 7 (tables of 4 / 16 KiB: registers bound the occupancy): unrolled at 3 per CU spills (26 / 24); rolled,
            //   groups of 1, needs 89..92 registers, so FOUR workgroups share a CU: 50.2 / 45.3 (16-bit form: 31.0 / 26.0)
#ifdef FRISK8_S4_GROUP
            constexpr int GR = FRISK8_S4_GROUP;
#else
            constexpr int GR = (K == 8 && BITS == 4 && !SIDE) ? 2 : 1;       // (8-bit form at K = 8: groups of one are 2..3 % ahead; SIDE: registers)
#endif
#ifdef FRISK8_ROLLED
            constexpr bool ROLLED = FRISK8_ROLLED != 0;
#else
            constexpr bool ROLLED = K < 8 || NT > 256;
#endif
            // ALLON: every lane of this wave starts a max-mer at every one of its positions (three waves in four of a window
            // without invalid bases) - no stand-in code to select, no weight to mask
            // (PARK: this wave has lanes that gather from the table - it parks what it used in the ring for the windows to come; a
            //  wave whose lanes all read the ring has nothing new to park, and its loop carries neither the select nor the store.
            //  Stores cost more than they look in this loop: loads and stores return in order on one counter, so every load behind a
            //  store waits for the store's acknowledgement - measured 1.1 ms per scan with every wave parking)
            auto score_all = [&](auto allon_c, auto orph_c, auto park_c) __attribute__((always_inline)) {
                constexpr bool ALLON = decltype(allon_c)::value;
                constexpr bool PARK = RING && decltype(park_c)::value;
                constexpr int FMODE = PARK ? 2 : 1;
                // the unrolled loop parks behind itself: a store in front of a load holds that load's data back until the store is
                // acknowledged (one in-order counter), and this wave - the one that gathers - is the one the others wait for
                constexpr bool PARK_LATE = PARK && FRISK8_PARK_LATE && !(ROLLED_K && ITS % (2 * GR) == 0);
                double parked[PARK_LATE ? ITS : 1];
                constexpr int ORPH = decltype(orph_c)::value;           // the orphan list holds <= 2 entries (one (K-1)-mer at most) / <= 4 / any number: 2 / 4 / 0
                auto on_at = [&](int it) -> bool { return ALLON || ((fm4 >> (31 - it)) & 1u); };
                auto code4_at = [&](int it) -> uint32_t {      // the position's max-mer, or the stand-in where it starts none
                    return on_at(it) ? raw4_at(it) : safe_code;
                };
                // a group of GR positions: counts from the table, plus the orphans
                auto score_group = [&](Fetched (&f)[GR], int g, auto check_c) __attribute__((always_inline)) {
                    constexpr bool CHECK = decltype(check_c)::value;         // (the unrolled form's last group may be short)
                    uint32_t c8[GR], c7[GR], c6[GR];
#pragma unroll
                    for (int k = 0; k < GR; ++k) {
                        if (!CHECK || g + k < ITS) table_counts(f[k], code4_at(g + k), c8[k], c7[k], c6[k], per4_at(g + k));
                    }
#pragma unroll
                    for (int k = 0; k < GR; ++k) {
                        if (!CHECK || g + k < ITS) {
                            const uint32_t c16 = code4_at(g + k), q6 = c16 >> 4, q7 = c16 >> 2;
                            constexpr int N6 = ORPH == -1 ? 0 : (ORPH == 2 ? 2 : 4), N7 = ORPH == -1 ? 0 : (ORPH == 2 ? 1 : 4);
#pragma unroll
                            for (int j = 0; j < N7; ++j) c7[k] += (q7 == o7c[j]) ? 1u : 0u;
#pragma unroll
                            for (int j = 0; j < N6; ++j) c6[k] += (q6 == o6[j]) ? 1u : 0u;
                            if (ORPH == 0)
                                for (uint32_t m = rest_mask; m; m &= m - 1u) {
                                    const uint32_t e = orph[__ffs(int(m)) - 1];
                                    c7[k] += (q7 == e) ? 1u : 0u;
                                    c6[k] += (q6 == ((e >> 2) & (NK / 16u - 1u))) ? 1u : 0u;
                                }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < GR; ++k) {
                        if (!CHECK || g + k < ITS) {
                            // (the value this position used, into the ring for the windows to come: 1.0 where it starts no max-mer)
                            if constexpr (PARK) {
                                const double v = on_at(g + k) ? f[k].Ig : 1.0;
                                if constexpr (PARK_LATE) parked[g + k] = v;      // (unrolled form: stored behind the loop)
                                else *reinterpret_cast<double*>(ring + ((!FRISK8_DEAL || lane_new) ? f[k].roff : dummy_off)) = v;   // (DEAL: the lanes that gathered park; the others' values are there)
                            }
                            score_one(f[k], c8[k], c7[k], c6[k], on_at(g + k) ? c8[k] : 0u, on_at(g + k));
                        }
                    }
                };
                // software pipeline: the reads of group g+1 are issued before the arithmetic of group g.
                // The rolled form: two groups per trip, ping-pong buffers (ITS is a multiple of 2 GR for GR = 1, 2); it needs
                // 86..129 registers and no scratch, but measured 4..6 % slower at K = 8 and three workgroups per CU; thread counts
                // of 320 / 384 / 512 per workgroup 18..60 % slower.
                if constexpr (ROLLED && ITS % (2 * GR) == 0) {
                    Fetched bufA[GR], bufB[GR];
#pragma unroll
                    for (int k = 0; k < GR; ++k) bufA[k] = fetch(code4_at(k), k, FMODE);
#pragma unroll 1
                    for (int g = 0; g < ITS; g += 2 * GR) {
#pragma unroll
                        for (int k = 0; k < GR; ++k) bufB[k] = fetch(code4_at(g + GR + k), g + GR + k, FMODE);
                        score_group(bufA, g, std::false_type{});
                        __builtin_amdgcn_sched_barrier(0);
                        const int gn = g + 2 * GR < ITS ? g + 2 * GR : 0;       // (the last trip fetches group 0 again, unused)
#pragma unroll
                        for (int k = 0; k < GR; ++k) bufA[k] = fetch(code4_at(gn + k), gn + k, FMODE);
                        score_group(bufB, g + GR, std::false_type{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    return;
                }
                Fetched buf[2][GR];
#pragma unroll
                for (int k = 0; k < GR; ++k) if (k < ITS) buf[0][k] = fetch(code4_at(k), k, FMODE);
#pragma unroll
                for (int g = 0; g < ITS; g += GR) {
                    const int cur = (g / GR) & 1;
#pragma unroll
                    for (int k = 0; k < GR; ++k) if (g + GR + k < ITS) buf[cur ^ 1][k] = fetch(code4_at(g + GR + k), g + GR + k, FMODE);
                    score_group(buf[cur], g, std::true_type{});
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (PARK_LATE) {
#pragma unroll
                    for (int it = 0; it < ITS; ++it) *reinterpret_cast<double*>(ring + ((!FRISK8_DEAL || lane_new) ? ring_mine(it) : dummy_off)) = parked[it];
                }
            };
            constexpr uint32_t ALL_MINE = uint32_t(0xFFFFFFFF00000000ull >> ITS);
            using orphX = std::integral_constant<int, -1>;                          // no orphan on the list: no compares at all
            using orph2 = std::integral_constant<int, 2>;
            using orph4 = std::integral_constant<int, 4>;
            using orphN = std::integral_constant<int, 0>;
            const bool wave_parks = RING && __any(lane_new);                // (wave-uniform)
            using yes = std::true_type;
            using no = std::false_type;
            if constexpr (PLACE) {
                if (n_list == 0) {              // (every orphan found room in the table: nearly every window)
                    if (__all(fm4 == ALL_MINE)) { if (wave_parks) score_all(yes{}, orphX{}, yes{}); else score_all(yes{}, orphX{}, no{}); }
                    else { if (wave_parks) score_all(no{}, orphX{}, yes{}); else score_all(no{}, orphX{}, no{}); }
                } else if (n_list <= 4) score_all(no{}, orph4{}, yes{});
                else score_all(no{}, orphN{}, yes{});
            } else {
                if (n_list <= 2 && n7 <= 1) {       // (every window without invalid bases)
                    if (__all(fm4 == ALL_MINE)) score_all(yes{}, orph2{}, yes{});
                    else score_all(no{}, orph2{}, yes{});
                } else if (n_list <= 4) score_all(no{}, orph4{}, yes{});
                else score_all(no{}, orphN{}, yes{});
            }

            // workgroup totals in a fixed order: DPP butterfly per wave, then the NW partials in wave order
#if FRISK8_PRIO
            __builtin_amdgcn_s_setprio(FRISK8_PRIO);
#endif
            STAMP(6)
            sw = wave_sum_exact(sw); sg = wave_sum_exact(sg); stt = wave_sum_exact(stt);
            if (lane == 0) { double* p = scratch + (tid >> 6) * 3; p[0] = sw; p[1] = sg; p[2] = stt; }
            __syncthreads();
            STAMP(7)
            // behind the barrier: nobody reads the tables any more.  The whole order-K table in 16-byte stores (8 / 16 per thread at
            // K = 8) is cheaper than every position clearing its own dword (20 tests, extracts and masked 4-byte stores per lane)
            ring_next = true;
            remove_fakes();
            if (!slide_next) {          // (a successor that slides takes the table as it stands)
                if constexpr (CLEAR_ALL) clear_t8();
                else {
#pragma unroll 4
                    for (int it = 0; it < ITS; ++it)
                        if (fm4 & (0x80000000u >> it)) t8[raw4_at(it) >> SHW] = 0u;
                }
            }
            clear_small();
            if (tid == 0) {
                double a = 0.0, b = 0.0, c = 0.0;
                for (int w = 0; w < NW; ++w) { a += scratch[3 * w]; b += scratch[3 * w + 1]; c += scratch[3 * w + 2]; }
                Pk->status[row] = status;
                Pk->sw[row] = a;
                Pk->sg[row] = b;
                Pk->kld[row] = c;                             // T; finish_rows_kernel turns (T, Sw, Sg) into the KLD
            }
            STAMP(8)
            __syncthreads();
            STAMP(9)
        }
    }
}

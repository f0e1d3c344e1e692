// scan8_kernel.h - phase B at K = 8 with SEVERAL independent workgroups per CU (the default K = 8 fast path).
//
// Same per-window computation as scan_kernel.h (reference frisk/__init__.py L1478-1494: crawlGenome L194-251 ->
// computeKmers(window) L280-367 -> IvomBuild x2 L369-457 -> KLD L459-472 -> calcGC L120-137 [-> calcRIP L474-495]),
// different data structure.  scan_kernel.h keeps the order-8 histogram as 4^8 16-bit counters = 128 KiB, so ONE
// 512-thread workgroup owns a CU and its two waves per SIMD run the same stage between the same barriers: they want
// the LDS pipeline at the same time and the VALU at the same time (r1 profile: VALU < 50 % busy, 52 % of wave cycles
// waiting).  Here the order-8 table is NARROW - BITS = 8 (64 KiB) or 4 (32 KiB) per counter - so that two (three)
// 256-thread workgroups share a CU, each on its own window and in its own stage:
//   * stage 1   ONE non-returning ds_add per max-mer position (field of the dword `code >> 2|3`).  Nothing else is
//               counted there: no order-6 update, no election of representatives.
//   * stage 2   the 5-mer counts are the sums of 64 neighbouring counters: every thread sums whole 16-byte reads
//               (v_sad_u8 / v_dot8_u32_u4), bank-swizzled; orders 4..1 follow inside wave 0 by DPP sums.  The grand
//               total of the table must equal the number of max-mer positions; a counter that wrapped (a max-mer
//               occurring >= 2^BITS times: poly-A, microsatellites) breaks that equality, and the window is then
//               handed to scan_kernel.h's 16-bit form through a device-side list (second launch, same stream).
//   * stage 4   c8, c7 = sum of 4 children, c6 = sum of 16 children come from ONE aligned LDS read (16 / 8 bytes);
//               the few 6- and 7-mers that are not prefixes of a max-mer (window tail, next to invalid bases) sit in
//               a short "orphan" list held in scalar registers.
//   * sums      every max-mer POSITION adds its max-mer's terms with weight 1/c8 (c8 positions share a max-mer), so
//               a lane's set of terms is fixed by the window alone: plain FP64 sums in a fixed order are
//               bit-reproducible across runs, grids and candidate ranges, without the exact (double-pair) summation
//               that the election of representatives by atomic arrival order forced on scan_kernel.h (6 FP64
//               instructions per term there, 1 multiply + 1 add here).
// Windows up to NT*ITS bases, kmin <= 5 (the shared prefix level); everything else stays on scan_kernel.h.
#pragma once
#include "scan_kernel.h"

#define FRISK8_ORPH_CAP 192        // orphan entries kept in LDS; a window with more goes to the 16-bit form
#define FRISK8_MISC_BYTES (2 * FRISK_MISC_SLOTS * 4 + 16 * 3 * 8)

enum { M8_TSUM = 6 };              // misc slot: grand total of the order-8 table (overflow check)

struct Lds8 {
    uint32_t t8, t8_bytes;      // order-8 table, BITS per counter
    uint32_t small, small_bytes;   // orders kmin..5, u16 bins
    uint32_t orphans;           // u16[FRISK8_ORPH_CAP]: run-7 positions store their 7-mer, run-6 positions 0x8000 | 6-mer << 2
    uint32_t pre_i, pre_w;      // shared prefix tables (level 5), as in scan_kernel.h
    uint32_t logtab, rctab;     // {1/c_i, ln c_i} x FRISK_LOGTAB_N and 1/c for c < 2^BITS - only when held in LDS
    uint32_t misc;
    uint32_t total;
};

template <int BITS>
__host__ __device__ inline Lds8 make_layout8(int kmin, bool tabs_lds) {
    Lds8 L;
    uint32_t o = 0;
    L.t8 = o; L.t8_bytes = 65536u * BITS / 8; o += L.t8_bytes;
    L.small = o; L.small_bytes = uint32_t((table_offset(kmin, 6) * 2 + 15) / 16 * 16); o += L.small_bytes;
    L.orphans = o; o += FRISK8_ORPH_CAP * 2;
    L.pre_i = o; o += 1024 * 8;
    L.pre_w = o; o += 1024 * 4;
    L.logtab = o; if (tabs_lds) o += FRISK_LOGTAB_N * 16;
    L.rctab = o; if (tabs_lds) o += (1u << BITS) * 8;
    L.misc = o; o += FRISK8_MISC_BYTES;
    L.total = (o + 15) / 16 * 16;
    return L;
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

// NT threads, windows of at most NT*ITS bases, BITS per order-8 counter, TABS_LDS: logarithm / reciprocal tables in LDS
// (otherwise read through the vector cache), WPS: waves per SIMD the register allocation must allow (= workgroups per
// CU * NT / 256).
template <int NT, int ITS, int BITS, bool TABS_LDS, int WPS, bool DEBUG>
__global__ __launch_bounds__(NT, WPS) void scan8_kernel(const ScanParams P) {
    static_assert(BITS == 4 || BITS == 8, "order-8 counters are 4 or 8 bits wide");
    static_assert(ITS + 7 <= 32, "a lane's positions and their max-mers must fit the 32 bases it loads");
    constexpr int NW = NT / 64;
    constexpr int SHW = BITS == 8 ? 2 : 3;               // code >> SHW = dword of the table
    constexpr uint32_t PERM = (32 / BITS) - 1;           // code & PERM = field inside the dword
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int kmin = P.kmin;
    const Lds8 L = make_layout8<BITS>(kmin, TABS_LDS);
    uint32_t* t8 = reinterpret_cast<uint32_t*>(lds + L.t8);
    const unsigned char* t8b = lds + L.t8;
    uint32_t* small32 = reinterpret_cast<uint32_t*>(lds + L.small);
    uint16_t* small16 = reinterpret_cast<uint16_t*>(lds + L.small);
    uint16_t* orph = reinterpret_cast<uint16_t*>(lds + L.orphans);
    double* pre_i = reinterpret_cast<double*>(lds + L.pre_i);
    uint32_t* pre_w = reinterpret_cast<uint32_t*>(lds + L.pre_w);
    uint32_t* misc_base = reinterpret_cast<uint32_t*>(lds + L.misc);
    double* scratch = reinterpret_cast<double*>(lds + L.misc + 2 * FRISK_MISC_SLOTS * 4);
    const double2* logtab = TABS_LDS ? reinterpret_cast<const double2*>(lds + L.logtab)
                                     : reinterpret_cast<const double2*>(P.log_tab);
    const double* rctab = TABS_LDS ? reinterpret_cast<const double*>(lds + L.rctab) : P.rc_tab;

    auto clear_t8 = [&]() {
        for (int i = tid; i < int(L.t8_bytes / 16); i += NT) reinterpret_cast<uint4*>(t8)[i] = make_uint4(0, 0, 0, 0);
    };
    auto clear_small = [&]() {
        for (uint32_t i = tid; i < L.small_bytes / 16; i += NT) reinterpret_cast<uint4*>(small32)[i] = make_uint4(0, 0, 0, 0);
    };
    clear_t8();
    clear_small();
    if (tid < 2 * FRISK_MISC_SLOTS) misc_base[tid] = 0;
    if (TABS_LDS) {
        double2* lt = reinterpret_cast<double2*>(lds + L.logtab);
        double* rt = reinterpret_cast<double*>(lds + L.rctab);
        for (int i = tid; i < FRISK_LOGTAB_N; i += NT) lt[i] = reinterpret_cast<const double2*>(P.log_tab)[i];
        for (int i = tid; i < (1 << BITS); i += NT) rt[i] = P.rc_tab[i];
    }
    __syncthreads();

    // XCD-aware work split (as scan_kernel.h): blocks b and b+8 share an XCD, neighbouring chunks share an L2
    const int G = gridDim.x;
    int v = blockIdx.x;
    if ((G & 7) == 0) v = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int64_t ncand = P.c1 - P.c0;
    const int64_t nchunks = (ncand + P.chunk - 1) / P.chunk;

    ScafDesc d;
    d.cand0 = 0; d.ncand = 0; d.off = 0; d.size = 0; d.kind = 0;
    int dsi = -1;
    uint32_t parity = 0;

    for (int64_t q = v; q < nchunks; q += G) {
        const int64_t cb = P.c0 + q * P.chunk;
        const int64_t ce = (cb + P.chunk < P.c1) ? cb + P.chunk : P.c1;
        for (int64_t cand = cb; cand < ce; ++cand) {
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
            const int64_t j = cand - d.cand0;
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
            const int64_t g0 = d.off + st;
            const int64_t row = cand - P.c0;
            uint32_t* misc = misc_base + parity * FRISK_MISC_SLOTS;
            uint32_t* misc_other = misc_base + (parity ^ 1u) * FRISK_MISC_SLOTS;
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
            uint32_t fullm = vld;                                            // 8 valid bases from here on ...
            fullm &= fullm << 1; fullm &= fullm << 2; fullm &= fullm << 4;
            fullm &= topbits(nleft - 7) & MINE;                              // ... all inside the window: a max-mer starts here
            auto code_at = [&](int it) -> uint32_t { return uint32_t(acode >> (48 - 2 * it)) & 0xFFFFu; };
            {
                uint32_t cA = 0, cT = 0, cG = 0, cC = 0, nvalid = 0;
                auto tally = [&](bool sel, uint32_t c2) {
                    cA += __popcll(__ballot(sel && c2 == 0));
                    cT += __popcll(__ballot(sel && c2 == 1));
                    cG += __popcll(__ballot(sel && c2 == 2));
                    cC += __popcll(__ballot(sel && c2 == 3));
                };
#pragma unroll
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
                        const int rs = run < 5 ? run : 5;
                        if (rs >= kmin) {
                            const uint32_t b = uint32_t(table_offset(kmin, rs)) + (c16 >> (16 - 2 * rs));
                            atomicAdd(&small32[b >> 1], 1u << ((b & 1u) * 16));
                        }
                        if (run >= 6) {
                            const uint32_t slot = atomicAdd(&misc[M_NORPH], 1u);
                            if (slot < FRISK8_ORPH_CAP)
                                orph[slot] = uint16_t(run == 7 ? (c16 >> 2) : (0x8000u | ((c16 >> 4) << 2)));
                        }
                    }
                }
                const uint32_t ntop = __popc(fullm);
#pragma unroll
                for (int b = 0; (1 << b) <= ITS; ++b) nvalid += uint32_t(__popcll(__ballot((ntop >> b) & 1u))) << b;
                if (tally_by_ballot) {
                    const uint32_t upm = actm & vld & ~alow;
#pragma unroll
                    for (int it = 0; it < ITS; ++it) tally((upm >> (31 - it)) & 1u, uint32_t(acode >> (62 - 2 * it)) & 3u);
                } else {
                    const uint32_t lowm = actm & vld & alow;
                    if (__ballot(lowm != 0)) {
#pragma unroll
                        for (int it = 0; it < ITS; ++it) tally((lowm >> (31 - it)) & 1u, uint32_t(acode >> (62 - 2 * it)) & 3u);
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
            __syncthreads();
            if (tid < FRISK_MISC_SLOTS) misc_other[tid] = 0;        // the previous window's counters: nobody reads them now

            // ---- stage 2: C_5[q] = D_5[q] + (sum of the 64 order-8 counters below q); grand total for the overflow check
            const uint32_t o5 = uint32_t(table_offset(kmin, 5));
            {
                uint32_t tot = 0;
                for (uint32_t q5 = tid; q5 < 1024u; q5 += NT) {
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
            // orders 4..1 inside wave 0: lane l takes the 4-mers l + 64 i; 3-mers are sums over quads, 2-mers over rows of
            // 16 lanes, 1-mers over the wave - no LDS round trip between the levels
            if (tid < 64 && kmin <= 4) {
                const uint32_t o4 = uint32_t(table_offset(kmin, 4));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t q4 = uint32_t(tid) + 64u * i;
                    const uint2 ch = *reinterpret_cast<const uint2*>(small16 + o5 + 4 * q4);
                    const uint32_t c4 = small16[o4 + q4] + (ch.x & 0xFFFFu) + (ch.x >> 16) + (ch.y & 0xFFFFu) + (ch.y >> 16);
                    small16[o4 + q4] = uint16_t(c4);
                    if (kmin <= 3) {
                        uint32_t qs = dpp_addu<0xB1>(c4);
                        qs = dpp_addu<0x4E>(qs);                                             // the quad's sum, in all four lanes
                        const uint32_t o3 = uint32_t(table_offset(kmin, 3));
                        uint32_t c3 = 0;
                        if ((tid & 3) == 0) { c3 = small16[o3 + (tid >> 2) + 16 * i] + qs; small16[o3 + (tid >> 2) + 16 * i] = uint16_t(c3); }
                        if (kmin <= 2) {
                            uint32_t rs = dpp_addu<0xB1>(c3);
                            rs = dpp_addu<0x4E>(rs); rs = dpp_addu<0x141>(rs); rs = dpp_addu<0x140>(rs);   // the row's four C_3
                            const uint32_t o2 = uint32_t(table_offset(kmin, 2));
                            uint32_t c2 = 0;
                            if ((tid & 15) == 0) { c2 = small16[o2 + (tid >> 4) + 4 * i] + rs; small16[o2 + (tid >> 4) + 4 * i] = uint16_t(c2); }
                            if (kmin <= 1) {
                                const uint32_t ws = __builtin_amdgcn_readlane(int(c2), 0) + __builtin_amdgcn_readlane(int(c2), 16) +
                                                    __builtin_amdgcn_readlane(int(c2), 32) + __builtin_amdgcn_readlane(int(c2), 48);
                                if (tid == 0) small16[i] = uint16_t(small16[i] + ws);
                            }
                        }
                    }
                }
            }
            __syncthreads();

            auto uni = [](uint32_t x) -> uint32_t { return __builtin_amdgcn_readfirstlane(x); };
            uint32_t upA = uni(misc[M_UPA]), upT = uni(misc[M_UPT]), upG = uni(misc[M_UPG]), upC = uni(misc[M_UPC]);
            if (!tally_by_ballot) {
                upA = uni(small16[0]) - upA; upT = uni(small16[1]) - upT; upG = uni(small16[2]) - upG; upC = uni(small16[3]) - upC;
            }
            const int64_t S = int64_t(upA) + upT + upG + upC;       // windowSpace (L380)
            const int64_t nn = n - S;                               // nnTotal of the window
            const bool keep = !(double(nn) >= 0.3 * double(n));     // N filter (L237-241 / L213)
            uint32_t status = (jump ? ROW_JUMPBACK : 0u);
            const uint32_t nvalid_top = uni(misc[M_NVALID]);
            const int n_orph = int(uni(misc[M_NORPH]));
            const bool wrapped = uni(misc[M8_TSUM]) != nvalid_top || n_orph > FRISK8_ORPH_CAP;

            auto zero_own = [&]() {             // every max-mer position clears its dword (all reads are behind a barrier)
#pragma unroll
                for (int it = 0; it < ITS; ++it)
                    if (fullm & (0x80000000u >> it)) t8[code_at(it) >> SHW] = 0u;
            };
            if (wrapped || !keep) {
                if (wrapped) clear_t8(); else zero_own();
                clear_small();
                if (tid == 0) {
                    if (wrapped) {
                        // a counter wrapped (every sum above is then unreliable, the N filter's included), or too many
                        // orphans: scan_kernel.h's 16-bit form redoes this window from scratch
                        const unsigned int slot = atomicAdd(P.ovf_count, 1u);
                        P.ovf_list[slot] = cand;
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

            // the orphan list in scalar registers: oe7 = the entry as stored (a run-7 entry equals its 7-mer, a run-6 entry
            // has bit 15 set and equals none), oe6 = the entry's 6-mer.  A window without invalid bases has exactly two.
            uint32_t oe7[4], oe6[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                oe7[k] = 0xFFFFFFFFu; oe6[k] = 0xFFFFFFFFu;
                if (k < n_orph) {
                    const uint32_t e = uni(uint32_t(orph[k]));
                    oe7[k] = e; oe6[k] = (e >> 2) & 0xFFFu;
                }
            }
            // counts of the three top orders for the max-mer `c16`, from ONE aligned read.  ORPH: bound on the orphan list
            // known to the caller (2, 4, or 0 = any length)
            auto top_counts = [&](uint32_t c16, auto orph_c, uint32_t& c8, uint32_t& c7, uint32_t& c6) __attribute__((always_inline)) {
                constexpr int ORPH = decltype(orph_c)::value;
                const uint32_t q6 = c16 >> 4, q7 = c16 >> 2;
                if (BITS == 8) {
                    const uint4 tw = *reinterpret_cast<const uint4*>(t8b + q6 * 16u);
                    const uint32_t lo = (c16 & 4u) ? tw.y : tw.x, hi = (c16 & 4u) ? tw.w : tw.z;
                    const uint32_t w = (c16 & 8u) ? hi : lo;
                    c8 = __builtin_amdgcn_ubfe(w, (c16 & 3u) * 8u, 8u);
                    c7 = __builtin_amdgcn_sad_u8(w, 0u, 0u);
                    c6 = __builtin_amdgcn_sad_u8(tw.x, 0u, __builtin_amdgcn_sad_u8(tw.y, 0u, __builtin_amdgcn_sad_u8(tw.z, 0u, __builtin_amdgcn_sad_u8(tw.w, 0u, 0u))));
                } else {
                    const uint2 tw = *reinterpret_cast<const uint2*>(t8b + q6 * 8u);
                    const uint32_t w = (c16 & 8u) ? tw.y : tw.x;
                    const uint32_t f = __builtin_amdgcn_ubfe(w, (c16 & 4u) * 4u, 16u);
                    c8 = __builtin_amdgcn_ubfe(f, (c16 & 3u) * 4u, 4u);
                    c7 = __builtin_amdgcn_udot8(f, 0x1111u, 0u, false);
                    c6 = __builtin_amdgcn_udot8(tw.x, 0x11111111u, __builtin_amdgcn_udot8(tw.y, 0x11111111u, 0u, false), false);
                }
                constexpr int NS = ORPH == 2 ? 2 : 4;
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    c7 += (q7 == oe7[k]) ? 1u : 0u;
                    c6 += (q6 == oe6[k]) ? 1u : 0u;
                }
                if (ORPH == 0)
                    for (int k = 4; k < n_orph; ++k) {
                        const uint32_t e = orph[k];
                        c7 += (q7 == e) ? 1u : 0u;
                        c6 += (q6 == ((e >> 2) & 0xFFFu)) ? 1u : 0u;
                    }
            };
            using orph2 = std::integral_constant<int, 2>;
            using orph4 = std::integral_constant<int, 4>;
            using orphN = std::integral_constant<int, 0>;
            // count of the x-mer c in this window (row metadata, RIP, debug dump)
            auto count = [&](int x, uint32_t c) -> uint32_t {
                if (x <= 5) return small16[table_offset(kmin, x) + c];
                uint32_t c8, c7, c6;
                top_counts(c << (2 * (8 - x)), orphN{}, c8, c7, c6);
                return x == 8 ? c8 : (x == 7 ? c7 : c6);
            };

            // ---- stage 3: window constants r_x = 4^x / D_x, D_x = (S-(x-1))*2 (L401-409), and the shared prefix tables
            double r_lane = 0.0;
            if (lane <= 8) r_lane = div_exact(double(1u << (2 * lane)), double(int32_t((S - (lane - 1)) * 2)));
            auto r_of = [&](int x) -> double {
                return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(r_lane), x),
                                        __builtin_amdgcn_readlane(__double2loint(r_lane), x));
            };
            {
                constexpr int LV = 5;
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
                for (int e = 0; e < 1024 / NT; ++e) {
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
                    pre_w[c] = W;
                }
            }
            __syncthreads();

            if (DEBUG && P.dbg_counts) {
                uint32_t* out = P.dbg_counts + row * int64_t(P.nprof);
                for (int x = kmin; x <= 8; ++x) {
                    const int64_t off = table_offset(kmin, x);
                    for (uint32_t c = tid; c < (1u << (2 * x)); c += NT) out[off + c] = count(x, c);
                }
            }
            if (DEBUG && P.dbg_meta && tid == 0) {
                P.dbg_meta[row * 3 + 0] = n;                                                   // totalLen
                P.dbg_meta[row * 3 + 1] = (n >= 8 ? n - 8 + 1 : 0) - int64_t(nvalid_top);      // exMax (L344-345)
                P.dbg_meta[row * 3 + 2] = nn;                                                  // nnTotal
            }
            if (nvalid_top == 0) status |= ROW_NO_MAXMER;
            if (nvalid_top > 0 && S >= kmin - 1 && S <= 7) status |= ROW_ZERO_WEIGHT;          // zero divisor on the window side
            status |= ROW_KEPT;
            if (tid == 0) {
                P.gc[row] = __longlong_as_double((long long)((uint64_t(uint32_t(S)) << 32) | uint32_t(upG + upC)));
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
            const double r6 = r_of(6), r7 = r_of(7), r8 = r_of(8);
            double sw = 0.0, sg = 0.0, stt = 0.0;
            // a position that starts no max-mer must add exactly nothing: clearing the HIGH word of its term leaves a
            // subnormal or zero, whatever garbage (NaN included) its lanes computed
            auto only_on = [](bool on, double x) -> double { return __hiloint2double(on ? __double2hiint(x) : 0, __double2loint(x)); };
            auto score_one = [&](uint32_t c16, bool on, auto orph_c) __attribute__((always_inline)) {
                const double Ig = P.ig[c16];                                 // unconditional gather (c16 < 4^8 always)
                uint32_t c8, c7, c6;
                top_counts(c16, orph_c, c8, c7, c6);
                const uint32_t pc = c16 >> 6;
                const uint32_t W = pre_w[pc] + (c6 << 12) + (c7 << 14) + (c8 << 16);
                double A = pre_i[pc];
                const double d6 = double(c6), d7 = double(c7), d8 = double(c8);
                A = __builtin_fma(d6 * d6, r6, A);
                A = __builtin_fma(d7 * d7, r7, A);
                A = __builtin_fma(d8 * d8, r8, A);
                // Iw = A/W and Iw/Ig with ONE division: ratio = A / (W * Ig), Iw = ratio * Ig
                const double ratio = div_exact(A, double(W) * Ig);
                const double Iw = ratio * Ig;
                const double t = Iw * log_tab_pos(ratio, logtab);
                const double rc = rctab[c8];                                 // 1/c8 (exactly 1.0 for the 9 in 10 max-mers seen once)
                sw += only_on(on, Iw * rc);
                sg += only_on(on, Ig * rc);
                stt += only_on(on, t * rc);
            };
            auto score_all = [&](auto orph_c) __attribute__((always_inline)) {
#pragma unroll
                for (int it = 0; it < ITS; ++it) {
                    score_one(code_at(it), (fullm >> (31 - it)) & 1u, orph_c);
#ifndef FRISK8_S4_GROUP
#define FRISK8_S4_GROUP 2
#endif
                    if ((it % FRISK8_S4_GROUP) == FRISK8_S4_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (n_orph <= 2) score_all(orph2{});
            else if (n_orph <= 4) score_all(orph4{});
            else score_all(orphN{});

            // workgroup totals in a fixed order: DPP butterfly per wave, then the NW partials in wave order
            sw = wave_sum_exact(sw); sg = wave_sum_exact(sg); stt = wave_sum_exact(stt);
            if (lane == 0) { double* p = scratch + (tid >> 6) * 3; p[0] = sw; p[1] = sg; p[2] = stt; }
            __syncthreads();
            zero_own();                                     // behind the barrier: nobody reads the tables any more
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

// profile_kernels.h - pack / unpack of scaffolds and phase A (genome k-mer profile) kernels.
// Reference semantics: computeKmers(genomeMode=True), frisk/__init__.py L280-367, called at L1442.
#pragma once
#include "frisk_device.h"

// ------------------------------------------------------------------------------------------------
// ASCII -> (codes, inv, low).  One thread packs 32 consecutive bases (two 16-byte loads, coalesced
// across the wave: 2 KiB of ASCII per wave-instruction pair) and writes two code words + one word of
// each mask.  HBM-bound: 1 B/base read, 0.5 B/base written.
// ------------------------------------------------------------------------------------------------
__device__ inline void classify_byte(uint32_t c, uint32_t& code, uint32_t& inv, uint32_t& low) {
    const uint32_t u = c & 0xDFu;                       // fold ASCII case
    const uint32_t isA = (u == 'A'), isT = (u == 'T'), isG = (u == 'G'), isC = (u == 'C');
    const uint32_t valid = isA | isT | isG | isC;
    code = isT * 1u + isG * 2u + isC * 3u;              // A=0,T=1,G=2,C=3 (reference L70)
    const uint32_t is_pad = (c == FRISK_PAD_BYTE);
    inv = (valid ^ 1u);
    low = (valid & ((c >> 5) & 1u)) | is_pad;           // lowercase acgt, or PAD (inv & low)
}

__global__ __launch_bounds__(256) void pack_kernel(const uint8_t* __restrict__ ascii, int64_t nwords32,
                                                    uint32_t* __restrict__ codes, uint32_t* __restrict__ inv,
                                                    uint32_t* __restrict__ low) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < nwords32; t += stride) {
        const uint4* src = reinterpret_cast<const uint4*>(ascii + t * 32);
        const uint4 q0 = src[0], q1 = src[1];
        const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        uint32_t c0 = 0, c1 = 0, mi = 0, ml = 0;
#pragma unroll
        for (int b = 0; b < 32; ++b) {
            const uint32_t ch = (w[b >> 2] >> (8 * (b & 3))) & 0xFFu;
            uint32_t code, i1, l1;
            classify_byte(ch, code, i1, l1);
            if (b < 16) c0 |= code << (30 - 2 * b); else c1 |= code << (30 - 2 * (b - 16));
            mi |= i1 << (31 - b);
            ml |= l1 << (31 - b);
        }
        codes[2 * t] = c0;
        codes[2 * t + 1] = c1;
        inv[t] = mi;
        low[t] = ml;
    }
}

// packed -> canonical ASCII (A/T/G/C, a/t/g/c, N; PAD -> '\0'): test / read-back utility
__global__ __launch_bounds__(256) void unpack_kernel(const uint32_t* __restrict__ codes,
                                                      const uint32_t* __restrict__ inv,
                                                      const uint32_t* __restrict__ low, int64_t p0, int64_t n,
                                                      uint8_t* __restrict__ out) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < n; t += stride) {
        const int64_t g = p0 + t;
        const uint32_t i1 = fetch_mask1(inv, g), l1 = fetch_mask1(low, g), c = fetch_code2(codes, g);
        uint8_t ch;
        if (i1 && l1) ch = 0;
        else if (i1) ch = 'N';
        else ch = uint8_t("ATGC"[c] | (l1 ? 0x20 : 0));
        out[t] = ch;
    }
}

// Run lists -> bitmap (the 0.25 B/base upload form, seq_pack2.h): `runs` holds n_runs pairs [begin, end) of padded positions,
// disjoint within one list; every position of a run gets its bit set (position p = bit 31 - (p & 31) of word p >> 5).  One
// wavefront per run: whole words inside a run are plain stores (a word covered entirely by one run belongs to no other run
// of the list), the ragged first / last word an atomic OR (another list - the PADs - may share it).  The bitmap was zeroed
// on the same stream.  HBM-bound at 1 bit per covered position; a list of millions of short runs costs a wave each.
__global__ __launch_bounds__(256) void expand_runs_kernel(const int64_t* __restrict__ runs, int64_t n_runs,
                                                           uint32_t* __restrict__ bits) {
    const int lane = int(threadIdx.x & 63u);
    const int64_t wave = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    for (int64_t r = wave; r < n_runs; r += nwaves) {
        const int64_t b = runs[2 * r], e = runs[2 * r + 1];
        if (e <= b) continue;
        const int64_t w0 = b >> 5, w1 = (e - 1) >> 5;
        for (int64_t w = w0 + lane; w <= w1; w += 64) {
            const int64_t lo = w << 5;
            const int from = b > lo ? int(b - lo) : 0, to = e < lo + 32 ? int(e - lo) : 32;      // bits [from, to) of the word
            const uint32_t m = uint32_t((0xFFFFFFFFull >> from) & ~(0xFFFFFFFFull >> to));
            if (m == 0xFFFFFFFFu) bits[w] = m;
            else atomicOr(bits + w, m);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Phase A counting.  Every padded position p contributes exactly ONE histogram update: to the table
// of order r = min(run, K) at the code of its longest valid word, where run = number of consecutive
// valid bases starting at p (a PAD or an invalid letter stops it; with --maskHost a soft-masked base
// stops it too, L336-337).  Lower orders follow by marginalisation (marginalize_kernel): a valid
// (r)-mer contains a valid x-mer prefix for every x <= r, so
//      C_x[q] = D_x[q] + sum_b C_{x+1}[4q+b]          (C_K = D_K)
// which is exact, including scaffold ends and N-adjacent positions.  `raw` = the D tables (profile
// layout) followed by {totalLen, #K-mer start positions, nnTotal, 0}; everything in it is a plain sum
// over positions, hence summable across batches and across GPUs.
// The order-K table (all but a vanishing share of the updates) is privatised per workgroup in LDS as 32-bit
// counters and flushed once with one global atomic per non-empty bin.  4^8 x 4 B = 256 KiB does not fit a
// CU's 160 KiB, so at K = 8 the k-mer space is split in two halves by the leading bit of the code and a
// workgroup makes its pass over its chunk for ONE half (grid = chunks x halves; the sequence is 0.5 B/base,
// reading it twice is free).  Orders below K (positions next to an invalid base or a scaffold end) go straight
// to global atomics, as do the three scalars (one atomic per wave).  Half 0 alone counts those.
// A lane owns one 32-position WORD of the bitmaps (7 loads: two words of each bitmap, three of the codes); the
// run flags of its 32 positions come from whole-word shifts, the codes from constant shifts, so a position costs
// about ten instructions instead of its own three unaligned fetches.
// ------------------------------------------------------------------------------------------------
#define FRISK_PROF_NT 1024

__global__ __launch_bounds__(FRISK_PROF_NT) void profile_add_kernel(const uint32_t* __restrict__ codes,
                                                                     const uint32_t* __restrict__ inv,
                                                                     const uint32_t* __restrict__ low, int64_t p0, int64_t p1,
                                                                     int kmin, int kmax, int mask_host, int nprof,
                                                                     int halves, int64_t chunk_words,
                                                                     unsigned long long* __restrict__ raw) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    const int half = int(blockIdx.x) % halves;
    const int64_t chunk = int64_t(blockIdx.x) / halves;
    const uint32_t nbins = (1u << (2 * kmax)) / uint32_t(halves);
    for (uint32_t b = threadIdx.x; b < nbins / 4; b += blockDim.x) reinterpret_cast<uint4*>(hist)[b] = make_uint4(0, 0, 0, 0);
    if (nbins < 4) for (uint32_t b = threadIdx.x; b < nbins; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    unsigned long long tot = 0, kpos = 0, nn = 0;
    const int64_t w_first = p0 >> 5, w_end = (p1 + 31) >> 5;               // bitmap words that hold [p0, p1)
    const int64_t wb = w_first + chunk * chunk_words;
    int64_t we = wb + chunk_words;
    if (we > w_end) we = w_end;
    const int64_t offK = table_offset(kmin, kmax);
    const int shK = 16 - 2 * kmax;
    const uint32_t halfbit = (halves == 2) ? nbins : 0u;                    // the leading bit of the code picks the half
    auto topbits = [](int64_t k) -> uint32_t {                             // the k most significant bits (k clamped to 0..32)
        k = k < 0 ? 0 : (k > 32 ? 32 : k);
        return uint32_t(0xFFFFFFFF00000000ull >> k);
    };
    for (int64_t wq = wb + threadIdx.x; wq < we; wq += blockDim.x) {
        const int64_t base = wq << 5;                                       // position of bit 31 of this word
        const uint32_t i0 = inv[wq], i1 = inv[wq + 1], l0 = low[wq], l1 = low[wq + 1];
        const uint32_t c0 = codes[2 * wq], c1 = codes[2 * wq + 1], c2 = codes[2 * wq + 2];
        const uint32_t inr = topbits(p1 - base) & ~topbits(p0 - base);      // positions of the word inside [p0, p1)
        const uint32_t e0 = i0 | (mask_host ? l0 : 0u), e1 = i1 | (mask_host ? l1 : 0u);
        const uint64_t V = ~((uint64_t(e0) << 32) | e1);                    // countable bases; position base+j at bit 63-j
        uint64_t F = V;                                                     // ... that start a run of kmax of them
        for (int s = 1; s < kmax; ++s) F &= V << s;
        const uint32_t fullm = uint32_t(F >> 32) & inr;
        const uint64_t lo64 = (uint64_t(c0) << 32) | c1, hi64 = (uint64_t(c1) << 32) | c2;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            if (fullm & (0x80000000u >> j)) {
                const uint32_t c16 = uint32_t((j < 16 ? lo64 : hi64) >> (48 - 2 * (j & 15))) & 0xFFFFu;
                const uint32_t code = c16 >> shK;
                if ((code & halfbit) == (half ? halfbit : 0u)) atomicAdd(&hist[code & (nbins - 1)], 1u);
            }
        }
        if (half == 0) {
            // the few countable positions whose run is shorter than kmax: straight to the global tables
            uint32_t shortm = uint32_t(V >> 32) & ~uint32_t(F >> 32) & inr;
            while (shortm) {
                const int j = __clz(int(shortm));
                shortm &= ~(0x80000000u >> j);
                const int run = __clzll((long long)(~(V << j)));            // < kmax here
                if (run >= kmin) {
                    const uint32_t c16 = uint32_t((j < 16 ? lo64 : hi64) >> (48 - 2 * (j & 15))) & 0xFFFFu;
                    atomicAdd(&raw[table_offset(kmin, run) + (c16 >> (16 - 2 * run))], 1ull);
                }
            }
            const uint64_t NP = ~((uint64_t(i0 & l0) << 32) | (i1 & l1));   // not a PAD
            uint64_t G = NP;                                                // a K-mer can start here (L329): no PAD in reach
            for (int s = 1; s < kmax; ++s) G &= NP << s;
            const uint32_t real = uint32_t(NP >> 32) & inr;
            tot += __popc(real);
            nn += __popc(real & (i0 | l0));                                 // not an uppercase A/T/G/C (countN, L106-118)
            kpos += __popc(uint32_t(G >> 32) & inr);
        }
    }
    if (half == 0) {                     // wave-level reduction of the three scalars, one atomic per wave
        for (int o = 32; o > 0; o >>= 1) {
            tot += __shfl_down(tot, o);
            kpos += __shfl_down(kpos, o);
            nn += __shfl_down(nn, o);
        }
        if ((threadIdx.x & 63) == 0) {
            if (tot) atomicAdd(&raw[nprof + 0], tot);
            if (kpos) atomicAdd(&raw[nprof + 1], kpos);
            if (nn) atomicAdd(&raw[nprof + 2], nn);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nbins; b += blockDim.x) {
        const uint32_t v = hist[b];
        if (v) atomicAdd(&raw[offK + uint32_t(half) * nbins + b], (unsigned long long)v);
    }
}

// K = 8 in ONE pass (round 4): the order-8 table as 4^8 SIXTEEN-bit counters, two to a word = 128 KiB, so that a workgroup
// counts every max-mer of its chunk in one walk instead of making the walk once per half of the k-mer space (the walk - masks,
// run flags, code extraction: ~10 instructions per position - is what the kernel's time is; the two-half form did it twice).
// A 16-bit field can wrap (65 536 copies of one 8-mer inside one workgroup's chunk: a megabase of poly-A), and a wrap is
// found the way the scan kernels find theirs: the table's grand total must equal the number of max-mer positions the
// workgroup counted.  It never does in assemblies; where it does, the workgroup throws its table away and counts its chunk
// again in the two-half form, one half after the other (32-bit counters in the same LDS).  Everything that does not go through
// the table - short runs, the three scalars - is added to the global tables during the first walk only.
__global__ __launch_bounds__(FRISK_PROF_NT) void profile_add16_kernel(const uint32_t* __restrict__ codes,
                                                                       const uint32_t* __restrict__ inv,
                                                                       const uint32_t* __restrict__ low, int64_t p0, int64_t p1,
                                                                       int kmin, int mask_host, int nprof, int64_t chunk_words,
                                                                       unsigned long long* __restrict__ raw) {
    constexpr int K = 8;
    constexpr uint32_t NW = (1u << (2 * K)) / 2;                            // 32 768 words: two 16-bit counters each, or one half's 32-bit counters
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    __shared__ unsigned long long red[2];                                   // max-mer positions counted / the table's grand total
    const int64_t chunk = int64_t(blockIdx.x);
    auto clear = [&]() {
        for (uint32_t b = threadIdx.x; b < NW / 4; b += blockDim.x) reinterpret_cast<uint4*>(hist)[b] = make_uint4(0, 0, 0, 0);
    };
    clear();
    if (threadIdx.x < 2) red[threadIdx.x] = 0ull;
    __syncthreads();
    const int64_t w_first = p0 >> 5, w_end = (p1 + 31) >> 5;               // bitmap words that hold [p0, p1)
    const int64_t wb = w_first + chunk * chunk_words;
    int64_t we = wb + chunk_words;
    if (we > w_end) we = w_end;
    const int64_t offK = table_offset(kmin, K);
    auto topbits = [](int64_t k) -> uint32_t {
        k = k < 0 ? 0 : (k > 32 ? 32 : k);
        return uint32_t(0xFFFFFFFF00000000ull >> k);
    };
    // one walk over the chunk.  MODE 0: every max-mer into its 16-bit field, and everything that bypasses the table; MODE 1 / 2: the
    // max-mers of half 0 / 1 into 32-bit counters, nothing else
    auto walk = [&](auto mode_c) {
        constexpr int MODE = decltype(mode_c)::value;
        unsigned long long tot = 0, kpos = 0, nn = 0, mine = 0;
        for (int64_t wq = wb + threadIdx.x; wq < we; wq += blockDim.x) {
            const int64_t base = wq << 5;
            const uint32_t i0 = inv[wq], i1 = inv[wq + 1], l0 = low[wq], l1 = low[wq + 1];
            const uint32_t c0 = codes[2 * wq], c1 = codes[2 * wq + 1], c2 = codes[2 * wq + 2];
            const uint32_t inr = topbits(p1 - base) & ~topbits(p0 - base);
            const uint32_t e0 = i0 | (mask_host ? l0 : 0u), e1 = i1 | (mask_host ? l1 : 0u);
            const uint64_t V = ~((uint64_t(e0) << 32) | e1);
            uint64_t F = V;
            F &= F << 1; F &= F << 2; F &= F << 4;                          // eight countable bases in a row
            const uint32_t fullm = uint32_t(F >> 32) & inr;
            const uint64_t lo64 = (uint64_t(c0) << 32) | c1, hi64 = (uint64_t(c1) << 32) | c2;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                if (fullm & (0x80000000u >> j)) {
                    const uint32_t code = uint32_t((j < 16 ? lo64 : hi64) >> (48 - 2 * (j & 15))) & 0xFFFFu;
                    if (MODE == 0) atomicAdd(&hist[code >> 1], 1u << ((code & 1u) * 16u));
                    else if ((code >> 15) == uint32_t(MODE - 1)) atomicAdd(&hist[code & (NW - 1u)], 1u);
                }
            }
            if (MODE == 0) {
                mine += uint32_t(__popc(fullm));
                uint32_t shortm = uint32_t(V >> 32) & ~uint32_t(F >> 32) & inr;
                while (shortm) {
                    const int j = __clz(int(shortm));
                    shortm &= ~(0x80000000u >> j);
                    const int run = __clzll((long long)(~(V << j)));        // < 8 here
                    if (run >= kmin) {
                        const uint32_t c16 = uint32_t((j < 16 ? lo64 : hi64) >> (48 - 2 * (j & 15))) & 0xFFFFu;
                        atomicAdd(&raw[table_offset(kmin, run) + (c16 >> (16 - 2 * run))], 1ull);
                    }
                }
                const uint64_t NP = ~((uint64_t(i0 & l0) << 32) | (i1 & l1));   // not a PAD
                uint64_t G = NP;
                G &= G << 1; G &= G << 2; G &= G << 4;                      // a K-mer can start here (L329): no PAD in reach
                const uint32_t real = uint32_t(NP >> 32) & inr;
                tot += __popc(real);
                nn += __popc(real & (i0 | l0));                             // not an uppercase A/T/G/C (countN, L106-118)
                kpos += __popc(uint32_t(G >> 32) & inr);
            }
        }
        if (MODE == 0) {
            for (int o = 32; o > 0; o >>= 1) {
                tot += __shfl_down(tot, o); kpos += __shfl_down(kpos, o); nn += __shfl_down(nn, o); mine += __shfl_down(mine, o);
            }
            if ((threadIdx.x & 63) == 0) {
                if (tot) atomicAdd(&raw[nprof + 0], tot);
                if (kpos) atomicAdd(&raw[nprof + 1], kpos);
                if (nn) atomicAdd(&raw[nprof + 2], nn);
                if (mine) atomicAdd(&red[0], mine);
            }
        }
    };
    walk(std::integral_constant<int, 0>{});
    __syncthreads();
    {   // the table's grand total
        unsigned long long sum = 0;
        for (uint32_t b = threadIdx.x; b < NW; b += blockDim.x) { const uint32_t v = hist[b]; sum += (v & 0xFFFFu) + (v >> 16); }
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
        if ((threadIdx.x & 63) == 0 && sum) atomicAdd(&red[1], sum);
    }
    __syncthreads();
    if (red[0] == red[1]) {                 // (uniform) no field wrapped: every counter is what it says
        for (uint32_t b = threadIdx.x; b < NW; b += blockDim.x) {
            const uint32_t v = hist[b];
            if (v & 0xFFFFu) atomicAdd(&raw[offK + 2u * b], (unsigned long long)(v & 0xFFFFu));
            if (v >> 16) atomicAdd(&raw[offK + 2u * b + 1u], (unsigned long long)(v >> 16));
        }
        return;
    }
    for (int half = 0; half < 2; ++half) {  // a wrapped field: the chunk again, half by half, in 32-bit counters
        __syncthreads();
        clear();
        __syncthreads();
        if (half == 0) walk(std::integral_constant<int, 1>{}); else walk(std::integral_constant<int, 2>{});
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < NW; b += blockDim.x) {
            const uint32_t v = hist[b];
            if (v) atomicAdd(&raw[offK + uint32_t(half) * NW + b], (unsigned long long)v);
        }
    }
}

// The same counting for kmax > 8 (4^K 32-bit counters no longer fit a CU's LDS): every position's ONE update goes straight to
// the global table of its order.  A lane owns one 32-position word of the bitmaps as above; 24-bit codes (12 bases).
// ~20 ms per 400 Mb - the reference's -k is unbounded (L1197-1206) but its own cost grows with 4^K; this path is for
// completeness, not speed.
__global__ __launch_bounds__(256) void profile_add_big_kernel(const uint32_t* __restrict__ codes, const uint32_t* __restrict__ inv,
                                                               const uint32_t* __restrict__ low, int64_t p0, int64_t p1, int kmin,
                                                               int kmax, int mask_host, int nprof,
                                                               unsigned long long* __restrict__ raw) {
    unsigned long long tot = 0, kpos = 0, nn = 0;
    const int64_t w_first = p0 >> 5, w_end = (p1 + 31) >> 5;
    auto topbits = [](int64_t k) -> uint32_t {
        k = k < 0 ? 0 : (k > 32 ? 32 : k);
        return uint32_t(0xFFFFFFFF00000000ull >> k);
    };
    for (int64_t wq = w_first + int64_t(blockIdx.x) * blockDim.x + threadIdx.x; wq < w_end; wq += int64_t(gridDim.x) * blockDim.x) {
        const int64_t base = wq << 5;
        const uint32_t i0 = inv[wq], i1 = inv[wq + 1], l0 = low[wq], l1 = low[wq + 1];
        const uint32_t c0 = codes[2 * wq], c1 = codes[2 * wq + 1], c2 = codes[2 * wq + 2];
        const uint32_t inr = topbits(p1 - base) & ~topbits(p0 - base);
        const uint32_t e0 = i0 | (mask_host ? l0 : 0u), e1 = i1 | (mask_host ? l1 : 0u);
        const uint64_t V = ~((uint64_t(e0) << 32) | e1);                    // countable bases; position base+j at bit 63-j
        const uint64_t lo64 = (uint64_t(c0) << 32) | c1, hi64 = (uint64_t(c1) << 32) | c2;
        uint32_t todo = uint32_t(V >> 32) & inr;
        while (todo) {
            const int j = __clz(int(todo));
            todo &= ~(0x80000000u >> j);
            int run = __clzll((long long)(~(V << j)));                      // countable bases from here on (<= 32 visible)
            run = run < kmax ? run : kmax;
            if (run >= kmin) {
                const uint32_t c24 = uint32_t((j < 16 ? lo64 : hi64) >> (40 - 2 * (j & 15))) & 0xFFFFFFu;
                atomicAdd(&raw[table_offset(kmin, run) + (c24 >> (24 - 2 * run))], 1ull);
            }
        }
        const uint64_t NP = ~((uint64_t(i0 & l0) << 32) | (i1 & l1));       // not a PAD
        uint64_t G = NP;                                                    // a K-mer can start here (L329): no PAD in reach
        for (int s = 1; s < kmax; ++s) G &= NP << s;
        const uint32_t real = uint32_t(NP >> 32) & inr;
        tot += __popc(real);
        nn += __popc(real & (i0 | l0));                                     // not an uppercase A/T/G/C (countN, L106-118)
        kpos += __popc(uint32_t(G >> 32) & inr);
    }
    for (int o = 32; o > 0; o >>= 1) {
        tot += __shfl_down(tot, o);
        kpos += __shfl_down(kpos, o);
        nn += __shfl_down(nn, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (tot) atomicAdd(&raw[nprof + 0], tot);
        if (kpos) atomicAdd(&raw[nprof + 1], kpos);
        if (nn) atomicAdd(&raw[nprof + 2], nn);
    }
}

// C_x[q] = D_x[q] + sum_b C_{x+1}[4q+b], in place on a copy of the D tables; one launch per order,
// from K-1 down to kmin (4^x threads each).
__global__ __launch_bounds__(256) void marginalize_kernel(int64_t* __restrict__ cnt, int kmin, int x) {
    const int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (q >= (int64_t(1) << (2 * x))) return;
    const int64_t* up = cnt + table_offset(kmin, x + 1) + 4 * q;
    cnt[table_offset(kmin, x) + q] += up[0] + up[1] + up[2] + up[3];
}

// The same for all orders x_hi, x_hi-1, ..., kmin in ONE launch of one workgroup (x_hi <= 7: at most 16 K bins on the widest
// level): below order 8 a launch per order is all launch latency (seven launches of ~2 us of work, ~7 us apart).
__global__ __launch_bounds__(1024) void marginalize_low_kernel(int64_t* __restrict__ cnt, int kmin, int x_hi) {
    for (int x = x_hi; x >= kmin; --x) {
        const int64_t n = int64_t(1) << (2 * x);
        int64_t* dst = cnt + table_offset(kmin, x);
        const int64_t* up = cnt + table_offset(kmin, x + 1);
        for (int64_t q = threadIdx.x; q < n; q += blockDim.x) dst[q] += up[4 * q] + up[4 * q + 1] + up[4 * q + 2] + up[4 * q + 3];
        __threadfence_block();
        __syncthreads();
    }
}

// genome mode adds the reverse complement of every counted word (L350-351):
// sym[c] = fwd[c] + fwd[rc(c)] - a palindromic word therefore counts twice per occurrence.
__global__ __launch_bounds__(256) void symmetrize_kernel(const int64_t* __restrict__ fwd, int64_t* __restrict__ sym,
                                                          int kmin, int kmax) {
    const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    for (int x = kmin; x <= kmax; ++x) {
        const int64_t off = table_offset(kmin, x), n = int64_t(1) << (2 * x);
        if (t >= off && t < off + n) {
            const uint32_t c = uint32_t(t - off);
            sym[t] = fwd[t] + fwd[off + revcomp_code(c, x)];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Genome-side IVOM value of EVERY max-mer, before normalisation (IvomBuild with isGenomeIVOM=True,
// L411-450).  It depends only on the genome profile, so it is computed once per profile; the scan
// kernel gathers from it and renormalises over the window's present max-mers (L453-454).
// A zero running weight or a zero divisor is a ZeroDivisionError in the reference (L437, L416-424):
// encoded as NaN and reported per window if such a max-mer is present there.  The genome side keeps the
// reference's operation order (it is computed once; the scan kernel uses the closed form on the window side).
// ------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
// metadata of L356-359 on the device: meta[0] = totalLen, meta[1] = exMax = the K-mer start positions that were NOT counted,
// meta[2] = nnTotal.  `cnt` = marginalised forward counts (profile layout), `raw_tail` = {totalLen, K-mer start positions,
// nnTotal} summed over positions.  One workgroup; the order-K table has at most 4^8 entries.
__global__ __launch_bounds__(1024) void profile_meta_kernel(const int64_t* __restrict__ cnt, const int64_t* __restrict__ raw_tail,
                                                             int kmin, int kmax, int64_t* __restrict__ meta) {
    __shared__ long long part[16];
    const int64_t n = int64_t(1) << (2 * kmax);
    const int64_t* top = cnt + table_offset(kmin, kmax);
    long long s = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += top[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long counted = 0;
        for (int w = 0; w < int(blockDim.x >> 6); ++w) counted += part[w];
        meta[0] = raw_tail[0];
        meta[1] = raw_tail[1] - counted;
        meta[2] = raw_tail[2];
    }
}

// `meta` (device): {totalLen, exMax, nnTotal}; genomeSpace = totalLen - nnTotal (L379)
__global__ __launch_bounds__(256) void genome_ivom_kernel(const int64_t* __restrict__ sym, int kmin, int kmax,
                                                           const int64_t* __restrict__ meta, double* __restrict__ ig) {
    const int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= (int64_t(1) << (2 * kmax))) return;
    const int64_t genome_space = meta[0] - meta[2];
    unsigned long long W = 0;
    double I = 0.0;
    bool bad = false;
    for (int x = kmin; x <= kmax; ++x) {
        const int64_t c = sym[table_offset(kmin, x) + (k >> (2 * (kmax - x)))];
        const unsigned long long wt = (unsigned long long)c << (2 * x);
        W += wt;
        const int64_t D = (genome_space - (x - 1)) * 2;
        if (W == 0 || D == 0) { bad = true; break; }
        const double p = double(c) / double(D);
        const double a = double(wt) / double(W);
        I = (x == kmin) ? a * p : a * p + ((1.0 - a) * I);
    }
    ig[k] = bad ? __longlong_as_double(0x7FF8000000000000LL) : I;
}

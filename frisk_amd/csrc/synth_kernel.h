// synth_kernel.h - synthetic scaffolds generated on the device (bench / property tests).
//
// "Synthetic scaffolds of the named shape" (BASELINE.json): order-3 Markov background, compositional
// islands drawn from a second, more skewed table, runs of N at two scales, soft-masked runs and - the "repeats"
// shape - simple repeats at a primate-like density: poly-A / poly-T tails and short-period microsatellites (these are what
// makes an 8-mer occur 16+ times in 5 kb), soft-masked like the rest when the assembly is soft-masked at all (lower_frac > 0:
// as RepeatMasker / TRF leave a released assembly), uppercase in an unmasked one.  Every
// property of a base is a pure function of (seed, scaffold index, position), so generation is parallel
// over 4096-base blocks and frisk_amd/synth.py reproduces the same bytes on the host with numpy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SYNTH_BLOCK 4096
#define SYNTH_ISLAND_BLOCKS 5      // island decision per 5 blocks  (20 480 bases)
#define SYNTH_NBIG_BLOCKS 8        // large N runs: units of 8 blocks (32 768 bases)
#define SYNTH_NSMALL 1024          // small N runs: units of 1 024 bases
#define SYNTH_LOWER 512            // soft-masked runs: units of 512 bases
#define SYNTH_REP 512              // simple repeats: at most one per unit of 512 bases, 12 .. 91 bases long
#define SYNTH_SAT_UNIT 131072      // satellite arrays: whole units of 32 blocks (131 072 bases) ...
#define SYNTH_SAT_GROUP 8          // ... the first 1..8 units of a group of 8 (1 Mi bases): arrays of 0.13 .. 1.05 Mb, neighbours merge
#define SYNTH_SAT_MONOMER 171      // alpha-satellite-like monomer length
#define SYNTH_SAT_DIV 0.03         // per-base divergence between copies of the monomer
#define SYNTH_GOLDEN 0x9E3779B97F4A7C15ull

// the simple repeat of one 512-base unit, if it has one: start (absolute position), length and motif - a pure function of the
// unit's hash.  Motifs (period, bases as 2-bit digits A=0,T=1,G=2,C=3): poly-A x3, poly-T x2, (CA)n, (TG)n, (AAAT)n.
// With period_mix > 0 a share period_mix of the repeats (second hash h2 < thr_mix) is instead one of (CAG)n, (AAT)n, (AAAAT)n,
// (TTAGGG)n - periods 3, 5 and 6, which no table of the scan kernel is shaped after - with a longer tail: 12 .. 31 bases, one
// in four up to 210 (a period-6 run needs ~100 bases before one of its 8-mers occurs 16 times), cut at the unit's end.
struct SynthRepeat { int64_t start; int32_t len, period; uint32_t motif; };
__host__ __device__ inline SynthRepeat synth_repeat_of(uint32_t h, int64_t unit, uint32_t h2 = 0xFFFFFFFFu, uint32_t thr_mix = 0u) {
    SynthRepeat r;
    r.len = 12 + int32_t((h >> 3) % 20u) + ((((h >> 8) & 3u) == 0u) ? int32_t((h >> 12) % 60u) : 0);
    r.start = unit * SYNTH_REP + int64_t((h >> 20) % uint32_t(SYNTH_REP - 96));
    const uint32_t kind = h & 7u;
    r.period = kind < 5u ? 1 : (kind < 7u ? 2 : 4);
    r.motif = kind < 3u ? 0x0u : (kind < 5u ? 0x1u : (kind == 5u ? 0xCu /* C,A */ : (kind == 6u ? 0x6u /* T,G */ : 0x01u /* A,A,A,T */)));
    if (thr_mix != 0u && h2 < thr_mix) {
        const uint32_t which = (h >> 1) & 3u;
        r.period = which < 2u ? 3 : (which == 2u ? 5 : 6);
        r.motif = which == 0u ? 0x32u /* C,A,G */ : (which == 1u ? 0x01u /* A,A,T */ : (which == 2u ? 0x001u /* A,A,A,A,T */ : 0x52Au /* T,T,A,G,G,G */));
        r.len = 12 + int32_t((h >> 3) % 20u) + ((((h >> 8) & 3u) == 0u) ? int32_t((h >> 12) % 180u) : 0);
        const int64_t room = (unit + 1) * SYNTH_REP - r.start;
        if (r.len > room) r.len = int32_t(room);
    }
    return r;
}
__host__ __device__ inline uint32_t synth_repeat_base(const SynthRepeat& r, int64_t p) {
    const int32_t k = int32_t((p - r.start) % r.period);
    return (r.motif >> (2 * (r.period - 1 - k))) & 3u;
}

struct SynthTables {
    uint32_t bg[64][4];            // cumulative 32-bit thresholds per order-3 context
    uint32_t isl[64][4];
};

__host__ __device__ inline uint64_t synth_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__host__ __device__ inline uint32_t synth_unit_hash(uint64_t seed, uint32_t scaf, uint64_t unit, uint64_t salt) {
    return uint32_t(synth_mix(seed ^ synth_mix((uint64_t(scaf) << 40) ^ unit ^ (salt << 56))) >> 32);
}

inline uint32_t synth_frac_to_u32(double f) {
    if (!(f > 0.0)) return 0u;
    if (f >= 1.0) return 0xFFFFFFFFu;
    return uint32_t(f * 4294967296.0);
}

inline void synth_make_tables(uint64_t seed, SynthTables& T) {
    for (int ctx = 0; ctx < 64; ++ctx) {
        uint64_t wb[4], wi[4], tb = 0, ti = 0;
        for (int b = 0; b < 4; ++b) {
            const uint64_t h = synth_mix(seed ^ synth_mix(0xB5ull + uint64_t(ctx) * 4 + uint64_t(b)));
            wb[b] = 64 + (h % 192);                 // background: mild skew (<= 4:1)
            wi[b] = 16 + ((h >> 20) % 1009);        // islands: strong skew
            tb += wb[b];
            ti += wi[b];
        }
        uint64_t cb = 0, ci = 0;
        for (int b = 0; b < 4; ++b) {
            cb += wb[b];
            ci += wi[b];
            T.bg[ctx][b] = (b == 3) ? 0xFFFFFFFFu : uint32_t((cb << 32) / tb);
            T.isl[ctx][b] = (b == 3) ? 0xFFFFFFFFu : uint32_t((ci << 32) / ti);
        }
    }
}

// one thread per 4096-base block of one scaffold
__global__ __launch_bounds__(64) void synth_kernel(uint8_t* __restrict__ out, int64_t len, uint64_t seed, uint32_t scaf,
                                                    const SynthTables T, uint32_t thr_island, uint32_t thr_nbig,
                                                    uint32_t thr_nsmall, uint32_t thr_low, uint32_t thr_rep, uint32_t thr_mix,
                                                    uint32_t thr_sat, uint32_t thr_div) {
    const int64_t nblk = (len + SYNTH_BLOCK - 1) / SYNTH_BLOCK;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t blk = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; blk < nblk; blk += stride) {
        const bool island = synth_unit_hash(seed, scaf, uint64_t(blk / SYNTH_ISLAND_BLOCKS), 1) < thr_island;
        const bool nbig = synth_unit_hash(seed, scaf, uint64_t(blk / SYNTH_NBIG_BLOCKS), 2) < thr_nbig;
        uint64_t state = synth_mix(seed ^ synth_mix((uint64_t(scaf) << 40) ^ uint64_t(blk)));
        uint32_t ctx = 0;
        const int64_t base = blk * SYNTH_BLOCK;
        // satellite array: a tandem array of one 171-base monomer (its own per group of units) with 3 % divergence between copies
        const int64_t sat_unit = base / SYNTH_SAT_UNIT, sat_group = sat_unit / SYNTH_SAT_GROUP;
        const bool sat = thr_sat != 0u && synth_unit_hash(seed, scaf, uint64_t(sat_group), 8) < thr_sat &&
                         uint32_t(sat_unit % SYNTH_SAT_GROUP) < 1u + ((synth_unit_hash(seed, scaf, uint64_t(sat_group), 9) >> 8) % uint32_t(SYNTH_SAT_GROUP));
        const int64_t end = (base + SYNTH_BLOCK < len) ? base + SYNTH_BLOCK : len;
        bool nsmall = false, lowr = false, rep = false;
        SynthRepeat R;
        R.start = 0; R.len = 0; R.period = 1; R.motif = 0;
        for (int64_t p = base; p < end; ++p) {
            if ((p & (SYNTH_LOWER - 1)) == 0 || p == base) {
                nsmall = synth_unit_hash(seed, scaf, uint64_t(p / SYNTH_NSMALL), 3) < thr_nsmall;
                lowr = synth_unit_hash(seed, scaf, uint64_t(p / SYNTH_LOWER), 4) < thr_low;
                rep = thr_rep != 0u && synth_unit_hash(seed, scaf, uint64_t(p / SYNTH_REP), 5) < thr_rep;
                if (rep) R = synth_repeat_of(synth_unit_hash(seed, scaf, uint64_t(p / SYNTH_REP), 6), p / SYNTH_REP,
                                             thr_mix ? synth_unit_hash(seed, scaf, uint64_t(p / SYNTH_REP), 7) : 0xFFFFFFFFu, thr_mix);
            }
            state += SYNTH_GOLDEN;
            const uint32_t r = uint32_t(synth_mix(state) >> 32);
            const uint32_t* row = island ? T.isl[ctx] : T.bg[ctx];
            uint32_t b = (r >= row[0]) + (r >= row[1]) + (r >= row[2]);
            const bool in_rep = rep && p >= R.start && p < R.start + R.len;
            if (in_rep) b = synth_repeat_base(R, p);
            if (sat) {
                const uint32_t j = uint32_t((p - sat_group * (int64_t(SYNTH_SAT_UNIT) * SYNTH_SAT_GROUP)) % SYNTH_SAT_MONOMER);
                b = synth_unit_hash(seed, scaf, uint64_t(sat_group) * 256u + j, 10) >> 30;
                const uint32_t hd = synth_unit_hash(seed, scaf, uint64_t(p), 11);
                if (hd < thr_div) b = hd & 3u;
            }
            ctx = ((ctx << 2) | b) & 63u;
            uint8_t ch = uint8_t("ATGC"[b]);
            if (lowr || ((in_rep || sat) && thr_low != 0u)) ch |= 0x20;      // (an assembly is soft-masked - repeats included - or it is not)
            if (nbig || nsmall) ch = 'N';
            out[p] = ch;
        }
    }
}

// frisk_device.h - shared device-side definitions for libfrisk_hip (gfx950 only).
//
// HBM layout of a resident batch of scaffolds ("padded coordinates"):
//   scaffold s occupies padded base positions [off[s], off[s]+len[s]) followed by at least one
//   PAD position; the batch ends on a multiple of 32 positions.  Three bit-packed arrays index
//   the same padded positions, big-endian inside each 32-bit word so that a k-mer read as
//   "the next 2k bits" is already the reference's canonical index (A=0,T=1,G=2,C=3, first base
//   most significant; reference frisk/__init__.py L70, L253-274):
//     codes : 2 bits/base, 16 bases/word, base p in bits [31-2r-1 .. 31-2r],  r = p & 15
//     inv   : 1 bit/base,  32 bases/word, base p in bit 31-(p&31); 1 = not one of ACGTacgt
//     low   : 1 bit/base,  same order;    1 = lowercase acgt (soft-masked)
//   PAD positions have inv = 1 AND low = 1 (impossible for a real base).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FRISK_MAX_K 12             // orders above 8 take the global-memory paths (scan_big_kernel, profile_add_big_kernel)
#define FRISK_PAD_BYTE 0

struct ScafDesc {
    int64_t off;     // padded position of the first RESIDENT base of the scaffold
    int64_t size;    // scaffold length in bases (the whole scaffold, also when only a tile of it is resident)
    int64_t cand0;   // index of the first candidate window described here
    int64_t ncand;   // number of candidate windows (0: none on this device)
    int64_t base0;   // position inside the scaffold of that first resident base (0 unless the batch holds a tile)
    int64_t j0;      // window index inside the scaffold of candidate cand0 (0 unless the batch holds a tile)
    int32_t kind;    // 0 = regular windows (L226-246), 1 = whole scaffold as one window (L211-221)
    int32_t pad_;
};

// offset (in bins) of the order-x table inside a concatenation that starts at order kmin
// = (4^x - 4^kmin) / 3 = sum_{i=kmin}^{x-1} 4^i: the bits 2i of 0b...0101 for kmin <= i < x (no division)
__host__ __device__ inline int64_t table_offset(int kmin, int x) {
    const uint64_t ones = 0x5555555555555555ull;
    return int64_t((ones & ((uint64_t(1) << (2 * x)) - 1)) & ~((uint64_t(1) << (2 * kmin)) - 1));
}

// 16 bits = the 8 bases starting at padded position g (base g most significant)
__device__ inline uint32_t fetch_codes16(const uint32_t* __restrict__ codes, int64_t g) {
    const int64_t wi = g >> 4;
    const int sh = int(g & 15) * 2;
    const uint64_t cat = (uint64_t(codes[wi]) << 32) | codes[wi + 1];
    return uint32_t(cat >> (48 - sh)) & 0xFFFFu;
}

// 8 mask bits for the 8 positions starting at g (position g = bit 7)
__device__ inline uint32_t fetch_mask8(const uint32_t* __restrict__ mask, int64_t g) {
    const int64_t wi = g >> 5;
    const int sh = int(g & 31);
    const uint64_t cat = (uint64_t(mask[wi]) << 32) | mask[wi + 1];
    return uint32_t(cat >> (56 - sh)) & 0xFFu;
}

__device__ inline uint32_t fetch_mask1(const uint32_t* __restrict__ mask, int64_t g) {
    return (mask[g >> 5] >> (31 - int(g & 31))) & 1u;
}

__device__ inline uint32_t fetch_code2(const uint32_t* __restrict__ codes, int64_t g) {
    return (codes[g >> 4] >> (30 - 2 * int(g & 15))) & 3u;
}

// 24 bits = the 12 bases starting at padded position g (orders above 8)
__device__ inline uint32_t fetch_codes24(const uint32_t* __restrict__ codes, int64_t g) {
    const int64_t wi = g >> 4;
    const int sh = int(g & 15) * 2;
    const uint64_t cat = (uint64_t(codes[wi]) << 32) | codes[wi + 1];
    return uint32_t(cat >> (40 - sh)) & 0xFFFFFFu;
}

// 16 mask bits for the 16 positions starting at g (position g = bit 15)
__device__ inline uint32_t fetch_mask16(const uint32_t* __restrict__ mask, int64_t g) {
    const int64_t wi = g >> 5;
    const int sh = int(g & 31);
    const uint64_t cat = (uint64_t(mask[wi]) << 32) | mask[wi + 1];
    return uint32_t(cat >> (48 - sh)) & 0xFFFFu;
}

// number of leading clear bits of a 16-bit field (16 if the field is 0)
__device__ inline int lead_clear16(uint32_t m16) { return m16 ? (__clz(int(m16)) - 16) : 16; }

// number of leading clear bits of an 8-bit field (8 if the field is 0)
__device__ inline int lead_clear8(uint32_t m8) { return m8 ? (__clz(int(m8)) - 24) : 8; }

// reverse complement of an x-mer code: complement = XOR 1 on every digit (A<->T, G<->C)
__host__ __device__ inline uint32_t revcomp_code(uint32_t c, int x) {
    uint32_t r = 0;
    for (int p = 0; p < x; ++p) {
        r = (r << 2) | ((c & 3u) ^ 1u);
        c >>= 2;
    }
    return r;
}

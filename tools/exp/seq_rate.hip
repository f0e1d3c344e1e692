// seq_rate.hip - cost of the scoring loop's arithmetic pieces on gfx950 as the compiler emits them (no memory traffic): shader cycles
// per wave per "position" with W workgroups of 256 threads per CU (= W waves per SIMD), two independent positions in flight per trip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#pragma clang fp contract(off)

__device__ inline double log_poly(double x) {          // log_tab_n<64, 6> with the table entry made from the bin index
    const int k = __builtin_amdgcn_frexp_exp(x);
    const double m = __builtin_amdgcn_frexp_mant(x);
    const uint32_t off = (uint32_t(__double2hiint(m)) >> 10) & (63u << 4);
    const double ex = 1.0 + double(off) * 0x1p-11, ey = double(off) * -0x1p-11;
    const double r = __builtin_fma(m, ex, -1.0);
    double p = 1.0 / 6;
    p = __builtin_fma(r, p, 1.0 / 5); p = __builtin_fma(r, p, -1.0 / 4); p = __builtin_fma(r, p, 1.0 / 3); p = __builtin_fma(r, p, -1.0 / 2);
    return __builtin_fma(double(k), 0.69314718055994530942, ey) + __builtin_fma(r * r, p, r);
}

enum Piece { LOG = 0, RCP, COUNTS, FULL, NPIECE };
static const char* NAMES[NPIECE] = {"log (15 instr)", "rcp + Newton + ratio (6)", "counts + W + A (int part)", "whole position (no memory)"};

template <int PIECE>
__global__ void piece_kernel(unsigned long long* out, int iters, double seed, uint32_t useed) {
    double sw = 0, sg = 0, st = 0;
    double x0 = seed + threadIdx.x * 1e-3, x1 = seed * 1.5 + threadIdx.x * 1e-3;
    uint32_t c0 = useed + threadIdx.x, c1 = useed * 3u + threadIdx.x;
    const double r6 = seed * 0.25, r7 = seed * 0.5, r8 = seed;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            double& x = u ? x1 : x0;
            uint32_t& c = u ? c1 : c0;
            if (PIECE == LOG) { const double l = log_poly(x); st += l; x = __builtin_fma(l, 1e-9, x); }
            if (PIECE == RCP) {
                const double den = x * 3.0;
                double rr = __builtin_amdgcn_rcp(den);
                rr = __builtin_fma(rr, __builtin_fma(-den, rr, 1.0), rr);
                const double ratio = seed * rr;
                st += ratio; x = __builtin_fma(ratio, 1e-9, x);
            }
            if (PIECE == COUNTS || PIECE == FULL) {
                c = c * 1664525u + 1013904223u;
                const uint32_t c16 = c >> 16;
                const uint64_t w64 = (uint64_t(c) << 32) | (c * 7u);
                const uint32_t w7 = uint32_t(w64 >> ((c16 & 12u) << 2));
                uint32_t c8 = __builtin_amdgcn_ubfe(w7, (c16 & 3u) * 4u, 4u);
                uint32_t c7 = __builtin_amdgcn_udot8(w7, 0x1111u, 0u, false);
                uint32_t c6 = __builtin_amdgcn_udot8(uint32_t(w64), 0x11111111u, __builtin_amdgcn_udot8(uint32_t(w64 >> 32), 0x11111111u, 0u, false), false);
                const uint32_t q6 = c16 >> 4, q7 = c16 >> 2;
                c7 += (q7 == useed) ? 1u : 0u;
                c6 += (q6 == useed) ? 1u : 0u;
                c6 += (q6 == useed + 1) ? 1u : 0u;
                const uint32_t W = (c8 << 16) + (c7 << 14) + (c6 << 12) + (c16 | 1u);
                double A = __builtin_fma(double(__umul24(c6, c6)), r6, double(c16 >> 6));
                A = __builtin_fma(double(__umul24(c7, c7)), r7, A);
                A = A + double(c8 + 1u) * r8;
                if (PIECE == COUNTS) { st += A; sw += double(W); }
                if (PIECE == FULL) {
                    const double Ig = x;
                    const double den = double(W) * Ig;
                    double rr = __builtin_amdgcn_rcp(den);
                    rr = __builtin_fma(rr, __builtin_fma(-den, rr, 1.0), rr);
                    const double ratio = A * rr;
                    const double Igr = Ig * r8, Iwr = ratio * Igr;
                    const double ln = log_poly(ratio);
                    sw += Iwr; sg += Igr; st = __builtin_fma(Iwr, ln, st);
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (sw + sg + st + x0 + x1 == 12345.678 && c0 + c1 == 77u) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

template <int PIECE>
void run(unsigned long long* d, int w) {
    const int grid = 256 * w, iters = 4000;
    std::vector<unsigned long long> h(1 + grid * 4);
    (void)hipMemset(d, 0, h.size() * 8);
    piece_kernel<PIECE><<<grid, 256>>>(d, iters, 1.0000001, 12345u);
    piece_kernel<PIECE><<<grid, 256>>>(d, iters, 1.0000001, 12345u);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin() + 1, h.end());
    const double med = double(h[1 + (h.size() - 1) / 2]) / (iters * 2.0);
    std::printf("%-30s %d waves/SIMD: %7.1f cycles per position per wave, %7.1f per SIMD\n", NAMES[PIECE], w, med, med / w);
}

int main() {
    unsigned long long* d;
    (void)hipMalloc(&d, (1 + 256 * 8 * 4) * 8);
    for (int w : {1, 2, 3, 4, 6, 8}) { run<LOG>(d, w); run<RCP>(d, w); run<COUNTS>(d, w); run<FULL>(d, w); }
    return 0;
}

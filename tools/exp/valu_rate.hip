// valu_rate.hip - issue cost of the vector instructions of scan8_kernel's scoring loop on gfx950, in shader cycles per
// wave-instruction, with 1, 2 and 3 waves per SIMD (independent streams: 8 accumulators per lane, so dependencies do not bind).
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_rate tools/exp/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Op { FMA64, MUL64, ADD64, RCP64, CVT64U, FREXPM, FREXPE, SHR64, DOT8, BFE, AND32, MUL24, LSHLADD, CMPADDC, FMA32, MIX, NOPS };
static const char* NAMES[NOPS] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_cvt_f64_u32", "v_frexp_mant_f64", "v_frexp_exp_i32_f64",
                                  "v_lshrrev_b64", "v_dot8_u32_u4", "v_bfe_u32", "v_and_b32", "v_mul_u32_u24", "v_lshl_add_u32",
                                  "v_cmp_eq+v_addc (pair)", "v_fma_f32", "mix: 2 fma64 + bfe + and (x8)"};

template <int OP>
__global__ void rate_kernel(unsigned long long* out, int iters, double seed) {
    double a[8], b = seed, c = seed * 0.5;
    uint32_t u[8], v = uint32_t(seed) + threadIdx.x;
    float f[8];
    uint64_t q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; u[i] = v + i; f[i] = float(a[i]); q[i] = v * 77u + i; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define X(i)                                                                                                          \
        if (OP == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                      \
        if (OP == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                  \
        if (OP == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                  \
        if (OP == RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));                                               \
        if (OP == CVT64U) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a[i]) : "v"(u[i]));                              \
        if (OP == FREXPM) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(a[i]));                                       \
        if (OP == FREXPE) asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(u[i]) : "v"(a[i]));                        \
        if (OP == SHR64) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(q[i]) : "v"(v));                              \
        if (OP == DOT8) asm volatile("v_dot8_u32_u4 %0, %0, %1, %0" : "+v"(u[i]) : "v"(v));                           \
        if (OP == BFE) asm volatile("v_bfe_u32 %0, %0, %1, 4" : "+v"(u[i]) : "v"(v));                                 \
        if (OP == AND32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(v));                                  \
        if (OP == MUL24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(v));                              \
        if (OP == LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(u[i]) : "v"(v));                        \
        if (OP == CMPADDC) asm volatile("v_cmp_eq_u32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(u[i]) : "v"(v) : "vcc"); \
        if (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));                   \
        if (OP == MIX) asm volatile("v_fma_f64 %0, %0, %2, %3\n v_bfe_u32 %1, %1, %4, 4\n v_fma_f64 %0, %0, %3, %2\n v_and_b32 %1, %1, %4" : "+v"(a[i]), "+v"(u[i]) : "v"(b), "v"(c), "v"(v));
        REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    uint32_t us = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s += a[i] + f[i]; us += u[i] + uint32_t(q[i]); }
    if (s == 12345.678 && us == 77u) out[0] = 1;          // keep everything alive
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
void run(unsigned long long* d, int waves_per_simd) {
    const int nt = 256, grid = 256 * waves_per_simd, iters = 2000;      // one 256-thread workgroup per wave slot: w workgroups per CU
    std::vector<unsigned long long> h(1 + grid * (nt / 64));
    hipMemset(d, 0, h.size() * 8);
    rate_kernel<OP><<<grid, nt>>>(d, iters, 1.0000001);
    rate_kernel<OP><<<grid, nt>>>(d, iters, 1.0000001);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin() + 1, h.end());
    const double med = double(h[1 + (h.size() - 1) / 2]);
    const double per = med / (double(iters) * 32.0 * (OP == CMPADDC ? 1 : 1));
    std::printf("%-26s %d waves/SIMD: %6.2f cycles per wave-instruction per wave, %6.2f per SIMD slot\n", NAMES[OP], waves_per_simd, per,
                per / waves_per_simd);
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, (1 + 256 * 8 * 16) * 8);
    for (int w = 1; w <= 8; ++w) {
        run<FMA64>(d, w); run<ADD64>(d, w); run<RCP64>(d, w); run<CVT64U>(d, w);
        run<DOT8>(d, w); run<BFE>(d, w); run<AND32>(d, w); run<LSHLADD>(d, w); run<CMPADDC>(d, w);
        run<FMA32>(d, w); run<MIX>(d, w);
    }
    return 0;
}

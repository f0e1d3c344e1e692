#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#pragma clang fp contract(off)
__global__ void k(const double* n, const double* d, double* r0, double* q1, double* q2, int N) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double dd = d[i], nn = n[i];
    double y = __builtin_amdgcn_rcp(dd);
    r0[i] = y;
    double e = __builtin_fma(-dd, y, 1.0);
    double y1 = __builtin_fma(y, e, y);
    {   // one Newton step + residual correction
        double q = nn * y1;
        double r = __builtin_fma(-dd, q, nn);
        q1[i] = __builtin_fma(r, y1, q);
    }
    e = __builtin_fma(-dd, y1, 1.0);
    double y2 = __builtin_fma(y1, e, y1);
    double q = nn * y2;
    double r = __builtin_fma(-dd, q, nn);
    q2[i] = __builtin_fma(r, y2, q);
}
int main() {
    const int N = 1 << 22;
    std::vector<double> n(N), d(N), r0(N), q1(N), q2(N);
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (s >> 11) * (1.0 / 9007199254740992.0); };
    for (int i = 0; i < N; ++i) { n[i] = std::ldexp(0.5 + rnd(), int(rnd() * 40) - 20); d[i] = std::ldexp(0.5 + rnd(), int(rnd() * 40) - 20); }
    double *dn, *dd, *d0, *d1, *d2;
    hipMalloc(&dn, N * 8); hipMalloc(&dd, N * 8); hipMalloc(&d0, N * 8); hipMalloc(&d1, N * 8); hipMalloc(&d2, N * 8);
    hipMemcpy(dn, n.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(dd, d.data(), N * 8, hipMemcpyHostToDevice);
    k<<<N / 256, 256>>>(dn, dd, d0, d1, d2, N);
    hipMemcpy(r0.data(), d0, N * 8, hipMemcpyDeviceToHost); hipMemcpy(q1.data(), d1, N * 8, hipMemcpyDeviceToHost); hipMemcpy(q2.data(), d2, N * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0; long diff1 = 0, diff2 = 0;
    for (int i = 0; i < N; ++i) {
        long double ex = (long double)n[i] / (long double)d[i];
        double exd = n[i] / d[i];
        e0 = std::fmax(e0, std::fabs((double)(((long double)r0[i] * d[i]) - 1.0L)));
        e1 = std::fmax(e1, std::fabs((double)((q1[i] - ex) / ex)));
        e2 = std::fmax(e2, std::fabs((double)((q2[i] - ex) / ex)));
        diff1 += q1[i] != exd; diff2 += q2[i] != exd;
    }
    printf("rcp rel err max %.3e (%.1f bits); 1-NR div max rel err %.3e, differs from IEEE in %ld of %d; 2-NR: %.3e, differs %ld\n", e0, -std::log2(e0), e1, diff1, N, e2, diff2);
    return 0;
}

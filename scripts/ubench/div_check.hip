// div_check.hip -- jd_div (jaero_amd/csrc/jd_libm.h) against the compiler's fp64 division ON THE DEVICE, operand pair by operand pair.
// jd_div is the compiler's own expansion of a / b (v_rcp_f64, two Newton steps, q = a r, one residual correction) without v_div_scale / v_div_fixup,
// so inside its operand range the two must agree bit for bit -- this program is the evidence ("the same exhaustive-check
// discipline as scripts/div_const_check.c").  2^33 pairs: random operands with exponents in +-60; quotients within two ulps of a representable
// number; quotients within two ulps of a rounding boundary (the midpoint between two doubles: a = RN(b (q + 2^-53)) by one fma); and operands of
// each call site's kind (jd_tanh's t + 2 and 6 - x t3, the AGC gain 1.414213562 / mean, MSEcalc's (sqrt(2) q) / mu, jd_hypot's (t1 + t2) / 2h).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o div_check scripts/ubench/div_check.hip && ./div_check      (prints one JSON line)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../jaero_amd/csrc/jd_libm.h"

__device__ __forceinline__ uint64_t rng(uint64_t &s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
__device__ __forceinline__ double mant(uint64_t r) { return __longlong_as_double((long long)((r >> 12) | 0x3ff0000000000000ull)); } // [1, 2)
__device__ __forceinline__ double scale(double x, int e) { return __longlong_as_double(__double_as_longlong(x) + ((long long)e << 52)); }
__device__ __forceinline__ double step_ulps(double x, int k) { return __longlong_as_double(__double_as_longlong(x) + k); }

__global__ __launch_bounds__(256) void k_div_check(unsigned long long *bad, double *first, int iters, uint64_t seed)
{
    uint64_t s = seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x + 1));
    for (int i = 0; i < 8; i++) rng(s);
    unsigned long long nbad = 0;
    double fa = 0, fb = 0;
    for (int it = 0; it < iters; it++)
    {
        const uint64_t r0 = rng(s), r1 = rng(s), r2 = rng(s);
        const int kind = (int)(r2 & 7), k = (int)((r2 >> 3) % 5) - 2;
        double b = scale(mant(r0), (int)((r2 >> 8) % 121) - 60), a;
        if (r2 & (1ull << 40)) b = -b;
        const double q = mant(r1);
        switch (kind)
        {
        case 0: case 1: a = scale(q, (int)((r2 >> 16) % 121) - 60); if (r2 & (1ull << 41)) a = -a; break;       // random operands
        case 2: a = step_ulps(b * q, k); break;                                                                  // quotient next to a double
        case 3: case 4: a = step_ulps(__builtin_fma(b, q, b * 0x1p-53), k); break;                                // quotient next to a rounding boundary
        case 5: { const double t = scale(mant(r1), (int)((r2 >> 16) % 72) - 53); b = t + 2.0; a = (r2 & (1ull << 42)) ? 2.0 : ((r2 & (1ull << 43)) ? -t * 0x1p-20 : t); } break; // jd_tanh
        case 6: b = scale(mant(r0), (int)((r2 >> 16) % 24) - 20); a = (r2 & (1ull << 42)) ? 1.414213562 : 1.4142135623730951 * scale(q, (int)((r2 >> 24) % 8) - 4); break; // AGC gain, MSEcalc
        default: { const double h = scale(mant(r0), (int)((r2 >> 16) % 40) - 20); b = 2.0 * h; a = h * h * scale(q, -53 - (int)((r2 >> 24) % 8)); if (r2 & (1ull << 42)) a = -a; if (!(r2 & 0xff0000000000ull)) a = 0.0; } break; // jd_hypot
        }
        const double want = a / b, got = jd_div(a, b);
        if (!(want == got)) { if (!nbad) { fa = a; fb = b; } nbad++; }
    }
    if (nbad)
    {
        if (atomicAdd(bad, nbad) == 0) { first[0] = fa; first[1] = fb; }
    }
}

int main(int argc, char **argv)
{
    const int blocks = 8192, threads = 256, iters = argc > 1 ? atoi(argv[1]) : 4096; // 2^21 threads x 4096 = 2^33 pairs
    unsigned long long *d_bad, h_bad = 0;
    double *d_first, h_first[2] = {0, 0};
    hipMalloc(&d_bad, 8); hipMalloc(&d_first, 16);
    hipMemset(d_bad, 0, 8); hipMemset(d_first, 0, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_div_check, dim3(blocks), dim3(threads), 0, 0, d_bad, d_first, iters, 0x4A41524Full);
    hipEventRecord(e1, 0);
    if (hipDeviceSynchronize() != hipSuccess) { printf("{\"error\": \"kernel failed\"}\n"); return 2; }
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h_bad, d_bad, 8, hipMemcpyDeviceToHost); hipMemcpy(h_first, d_first, 16, hipMemcpyDeviceToHost);
    printf("{\"what\": \"jd_div against the compiler's fp64 division on the device\", \"pairs\": %llu, \"differences\": %llu, \"first_a\": \"%a\", \"first_b\": \"%a\", \"ms\": %.1f}\n",
           (unsigned long long)blocks * threads * iters, h_bad, h_first[0], h_first[1], ms);
    return h_bad ? 1 : 0;
}

// What the device library's atan2 / hypot and jd_atan2 / jd_hypot (jaero_amd/csrc/jd_libm.h) return against the HOST libm on the same
// arguments, and what a call costs on a lone wavefront per SIMD (the sample loops' situation).  Not part of the product library.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o scripts/ubench/atan2_rates scripts/ubench/atan2_rates.hip
//   run:   scripts/ubench/atan2_rates [millions of arguments = 64]
// Arguments: a unit-circle oscillator value times a resonator output pair (the symbol-timing detector's operands,
// JAERO/oqpskdemodulator.cpp:480-484), scales 2^-60 .. 2^20.  Timing: 1024 workgroups of one wavefront, each lane a chain of 4096
// dependent calls (the result perturbs the next argument), so the figure is latency-bound issue, as in the back half of k_oqpsk_fb.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../jaero_amd/csrc/jd_libm.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int WHICH> __global__ void k_eval(const double *a, const double *b, double *o, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; // n is a multiple of the block size: every lane of a wavefront is active
    if (i >= n) return;
    const JdAtanLane T = jd_atan_lane_table(threadIdx.x);
    if (WHICH == 0) o[i] = atan2(a[i], b[i]);
    if (WHICH == 1) o[i] = jd_atan2(a[i], b[i], T);
    if (WHICH == 2) o[i] = hypot(a[i], b[i]);
    if (WHICH == 3) o[i] = jd_hypot(a[i], b[i]);
}
template <int WHICH> __global__ void k_chain(const double *a, const double *b, double *o, int reps)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const JdAtanLane T = jd_atan_lane_table(threadIdx.x);
    double y = a[i], x = b[i], acc = 0;
    for (int r = 0; r < reps; r++)
    {
        double v;
        if (WHICH == 0) v = atan2(y, x);
        if (WHICH == 1) v = jd_atan2(y, x, T);
        if (WHICH == 2) v = hypot(y, x);
        if (WHICH == 3) v = jd_hypot(y, x);
        if (WHICH == 4) v = y * 0.5 + x; // the chain's own cost
        acc += v;
        y = y + v * 0x1p-30; x = x - v * 0x1p-31; // dependent, stays in range
    }
    o[i] = acc;
}

static inline unsigned long long rng(unsigned long long *s) { unsigned long long x = *s; x ^= x << 13; x ^= x >> 7; x ^= x << 17; *s = x; return x; }
static inline double u01(unsigned long long *s) { return (double)(rng(s) >> 11) * 0x1p-53; }

int main(int argc, char **argv)
{
    const size_t n = (size_t)(argc > 1 ? atol(argv[1]) : 64) * (1u << 20);
    std::vector<double> a(n), b(n), o(n);
    unsigned long long s = 0x9e3779b97f4a7c15ull;
    for (size_t i = 0; i < n; i++)
    {
        const double ph = 2 * M_PI * u01(&s), sc = exp2(80 * u01(&s) - 60);
        const double cx = cos(ph), cy = sin(ph);
        const double mre = sc * (2 * u01(&s) - 1), mim = sc * (2 * u01(&s) - 1) * exp2(-6 * u01(&s));
        b[i] = cx * mre - cy * mim; a[i] = cx * mim + cy * mre;
    }
    double *da, *db, *dout;
    CK(hipMalloc(&da, n * 8)); CK(hipMalloc(&db, n * 8)); CK(hipMalloc(&dout, n * 8));
    CK(hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice));
    const char *names[] = {"device library atan2", "jd_atan2 (correctly rounded)", "device library hypot", "jd_hypot (glibc 2.35's steps)"};
    for (int w = 0; w < 4; w++)
    {
        const dim3 grid((unsigned)(n / 256)), blk(256);
        if (w == 0) hipLaunchKernelGGL(k_eval<0>, grid, blk, 0, 0, da, db, dout, n);
        if (w == 1) hipLaunchKernelGGL(k_eval<1>, grid, blk, 0, 0, da, db, dout, n);
        if (w == 2) hipLaunchKernelGGL(k_eval<2>, grid, blk, 0, 0, da, db, dout, n);
        if (w == 3) hipLaunchKernelGGL(k_eval<3>, grid, blk, 0, 0, da, db, dout, n);
        CK(hipDeviceSynchronize()); CK(hipGetLastError());
        CK(hipMemcpy(o.data(), dout, n * 8, hipMemcpyDeviceToHost));
        size_t diff = 0, diff2 = 0;
        for (size_t i = 0; i < n; i++)
        {
            const double want = w < 2 ? atan2(a[i], b[i]) : hypot(a[i], b[i]);
            if (memcmp(&want, &o[i], 8) != 0)
            {
                diff++;
                long long x, y; memcpy(&x, &want, 8); memcpy(&y, &o[i], 8);
                if (llabs(x - y) > 1) diff2++;
            }
        }
        printf("%-32s %zu arguments: %zu results differ from the host libm's (%.3e), %zu by more than one ulp\n", names[w], n, diff, (double)diff / (double)n, diff2);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 4096;
    const char *cn[] = {"device library atan2", "jd_atan2", "device library hypot", "jd_hypot", "(the chain alone)"};
    float base = 0;
    for (int pass = 0; pass < 2; pass++)
        for (int w = 4; w >= 0; w--)
        {
            const dim3 grid(1024), blk(64);
            CK(hipEventRecord(e0, 0));
            if (w == 0) hipLaunchKernelGGL(k_chain<0>, grid, blk, 0, 0, da, db, dout, reps);
            if (w == 1) hipLaunchKernelGGL(k_chain<1>, grid, blk, 0, 0, da, db, dout, reps);
            if (w == 2) hipLaunchKernelGGL(k_chain<2>, grid, blk, 0, 0, da, db, dout, reps);
            if (w == 3) hipLaunchKernelGGL(k_chain<3>, grid, blk, 0, 0, da, db, dout, reps);
            if (w == 4) hipLaunchKernelGGL(k_chain<4>, grid, blk, 0, 0, da, db, dout, reps);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (w == 4) base = ms;
            if (pass == 1) printf("%-24s %7.1f ns per dependent call on a lone wavefront (%7.1f ns with the chain's own %5.1f ns)\n", cn[w], (ms - base) * 1e6 / reps, ms * 1e6 / reps, base * 1e6 / reps);
        }
    return 0;
}

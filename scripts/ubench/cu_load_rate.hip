// What one CU gets out of 16-byte-per-lane loads, by where the lines are: HBM (cold), L2 (the same region read again), and HBM while every other CU
// does the same (the coarse kernel's situation).  One 512-thread workgroup per CU streams REGION bytes of its own with 32 loads per thread in flight.
// Not part of the product library.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/ubench/cu_load_rate scripts/ubench/cu_load_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef double v2 __attribute__((ext_vector_type(2)));

// passes over the workgroup's own region; per pass every thread issues 32 loads (512 threads x 32 x 16 B = 256 KiB); cyc[b * npass + p] = clocks
template <bool NT> __global__ __launch_bounds__(512) void k_stream(const v2 *base, size_t region_v2, int npass, int rotate, unsigned long long *cyc, double *sink)
{
    const int t = threadIdx.x;
    const v2 *my = base + (size_t)blockIdx.x * region_v2;
    double acc = 0;
    for (int p = 0; p < npass; p++)
    {
        const v2 *q = my + (size_t)((p * rotate) % (int)(region_v2 / 16384)) * 16384; // rotate = 0: the same 256 KiB again (L2 hits after pass 0)
        __syncthreads();
        const unsigned long long c0 = wall_clock64();
        v2 v[32];
#pragma unroll
        for (int s = 0; s < 32; s++) v[s] = NT ? __builtin_nontemporal_load(q + s * 512 + t) : q[s * 512 + t];
#pragma unroll
        for (int s = 0; s < 32; s++) acc += v[s].x + v[s].y;
        __syncthreads();
        const unsigned long long c1 = wall_clock64();
        if (t == 0) cyc[(size_t)blockIdx.x * npass + p] = c1 - c0;
    }
    if (acc == 1.2345) sink[0] = acc;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount, npass = 64;
    const size_t region_v2 = (size_t)16384 * 64; // 64 x 256 KiB = 16 MiB per workgroup
    v2 *d; unsigned long long *cyc; double *sink;
    CK(hipMalloc(&d, (size_t)ncu * region_v2 * sizeof(v2))); CK(hipMemset(d, 0, (size_t)ncu * region_v2 * sizeof(v2)));
    CK(hipMalloc(&cyc, (size_t)ncu * npass * 8)); CK(hipMalloc(&sink, 64));
    std::vector<unsigned long long> h((size_t)ncu * npass);
    auto run = [&](const char *name, int grid, int rotate, bool nt) {
        if (nt) hipLaunchKernelGGL(k_stream<true>, dim3(grid), dim3(512), 0, 0, (const v2 *)d, region_v2, npass, rotate, cyc, sink);
        else hipLaunchKernelGGL(k_stream<false>, dim3(grid), dim3(512), 0, 0, (const v2 *)d, region_v2, npass, rotate, cyc, sink);
        CK(hipDeviceSynchronize()); CK(hipGetLastError());
        CK(hipMemcpy(h.data(), cyc, (size_t)grid * npass * 8, hipMemcpyDeviceToHost));
        double sum = 0; long n = 0;
        for (int b = 0; b < grid; b++) for (int p = 8; p < npass; p++) { sum += (double)h[(size_t)b * npass + p]; n++; }
        const double us = sum / n * 0.01;
        printf("%-64s %7.2f us per 256 KiB and CU = %6.1f GB/s per CU (%5.2f TB/s over %d CUs)\n", name, us, 262144.0 / us / 1e3, 262144.0 / us / 1e6 * grid, grid);
    };
    run("one CU alone, fresh lines every pass (HBM)", 1, 1, false);
    run("one CU alone, the same 256 KiB again (L2)", 1, 0, false);
    run("all CUs, fresh lines every pass (HBM)", ncu, 1, false);
    run("all CUs, fresh lines, non-temporal", ncu, 1, true);
    run("all CUs, the same 256 KiB again (L2: 32 CUs x 256 KiB per XCD)", ncu, 0, false);
    run("one CU per XCD (8), fresh lines", 8, 1, false);
    return 0;
}

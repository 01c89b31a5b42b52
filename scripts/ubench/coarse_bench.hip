// A/B harness for the 2^14-point coarse-frequency kernels: k_coarse4 (round 1/2) against k_coarse5 (two streams per thread, round 3),
// on the same rings, with phase ablations of both (no HBM traffic / no transforms) so that what a launch spends where can be read off.
// Not part of the product library; includes the kernels' headers directly.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o scripts/ubench/coarse_bench scripts/ubench/coarse_bench.hip
//   run:   scripts/ubench/coarse_bench [channels = 65536] [launches = 4]
// Every channel's ring holds cos(2 pi 896 n / N) exp(j 2 pi d n / N) + noise (d per channel): its square has lines at 2d and 2d +- 1792,
// so the fold has one clear maximum at N/2 + 2d and both kernels must report the same bin; y[] rows are compared to 1e-9 inside the
// support of the squared signal's spectrum.
#include "k_coarse5.h"
#include "k_coarse4.h"
#include "../../jaero_amd/csrc/k_coarse6.h"
#include "k_coarse7.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int ABL> __global__ __launch_bounds__(C2_THREADS) void k5_abl(const JGeom g, const JPtrs p, const int *cl, int nlist, const double2 *tw)
{
    coarse5_body<false, ABL>(g, p, cl, nlist, tw);
}

__global__ void k_fill(double2 *ring, int nch, unsigned seed)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nch * 16384) return;
    const int ch = (int)(i >> 14), n = (int)(i & 16383);
    unsigned h = (unsigned)(i * 2654435761u) ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    unsigned h2 = h * 1664525u + 1013904223u; h2 ^= h2 >> 15;
    const double nr = ((double)(h & 0xffffff) / 16777216.0 - 0.5), ni = ((double)(h2 & 0xffffff) / 16777216.0 - 0.5);
    const double d = (double)((ch * 37) % 2001 - 1000) * 0.5; // bins
    const double a = 1000.0 * cos(2.0 * M_PI * 896.0 * n / 16384.0);
    const double ph = 2.0 * M_PI * d * n / 16384.0;
    ring[i] = make_double2(a * cos(ph) + 600.0 * nr, a * sin(ph) + 600.0 * ni);
}

// one transform of wg_fft14_2s on x (natural order in global memory), result in natural order
__global__ __launch_bounds__(C2_THREADS) void k_fft_test(const double2 *x, double2 *X, const double2 *tw)
{
    extern __shared__ __attribute__((aligned(16))) double xch[];
    const int t = threadIdx.x;
    CV<16> a, b;
#pragma unroll
    for (int s = 0; s < 16; s++)
    {
        const double2 v0 = x[C5_IDX(0, s, t)], v1 = x[C5_IDX(1, s, t)];
        a.r[s] = v0.x; a.i[s] = v0.y; b.r[s] = v1.x; b.i[s] = v1.y;
    }
    c5_fft(a, b, xch, tw, t);
#pragma unroll
    for (int s = 0; s < 16; s++) { X[C5_IDX(0, s, t)] = make_double2(a.r[s], a.i[s]); X[C5_IDX(1, s, t)] = make_double2(b.r[s], b.i[s]); }
}

__global__ __launch_bounds__(C2_THREADS) void k_fft_test6(const double2 *x, double2 *X, const double2 *tw)
{
    extern __shared__ __attribute__((aligned(16))) double xch[];
    const int t = threadIdx.x;
    CV<32> d;
#pragma unroll
    for (int s = 0; s < 32; s++) { const double2 v = x[s * 512 + t]; d.r[s] = v.x; d.i[s] = v.y; }
    c6_fft<14>(d, xch, tw, t);
#pragma unroll
    for (int s = 0; s < 32; s++) X[s * 512 + t] = make_double2(d.r[s], d.i[s]);
}

__global__ __launch_bounds__(256) void k_fft_test7(const double2 *x, double2 *X, const double2 *tw)
{
    extern __shared__ __attribute__((aligned(16))) double xch[];
    const int t = threadIdx.x;
    CV<64> d;
#pragma unroll
    for (int s = 0; s < 64; s++) { const double2 v = x[s * 256 + t]; d.r[s] = v.x; d.i[s] = v.y; }
    wg_fft14_e64(d, xch, tw, t);
#pragma unroll
    for (int s = 0; s < 64; s++) X[s * 256 + t] = make_double2(d.r[s], d.i[s]);
}

static void host_fft(std::vector<double> &re, std::vector<double> &im)
{
    const int n = (int)re.size();
    for (int i = 1, j = 0; i < n; i++) { int bit = n >> 1; for (; j & bit; bit >>= 1) j ^= bit; j ^= bit; if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); } }
    for (int len = 2; len <= n; len <<= 1)
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < len / 2; k++)
            {
                const double a = -2.0 * M_PI * k / len, wr = cos(a), wi = sin(a);
                const double ur = re[i + k], ui = im[i + k], vr = re[i + k + len / 2] * wr - im[i + k + len / 2] * wi, vi = re[i + k + len / 2] * wi + im[i + k + len / 2] * wr;
                re[i + k] = ur + vr; im[i + k] = ui + vi; re[i + k + len / 2] = ur - vr; im[i + k + len / 2] = ui - vi;
            }
}

struct Side { double *S; int *I; double *y; double *slog; };

int main(int argc, char **argv)
{
    const int nch = argc > 1 ? atoi(argv[1]) : 65536;
    const int nl = argc > 2 ? atoi(argv[2]) : 4;
    const int N = 16384;
    JGeom g = {};
    g.kind = 1; g.nch = nch; g.nchp = nch; g.ngroups = nch / 64; g.Fs = 48000.0; g.fb = 10500.0; g.nfft = N; g.nfft_log2 = 14; g.flags = 2u; g.log_cap = 64;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs; %d channels, %d timed launches per variant\n", prop.gcnArchName, ncu, nch, nl);
    double2 *ring, *tw; CK(hipMalloc(&ring, (size_t)nch * N * sizeof(double2))); CK(hipMalloc(&tw, N * sizeof(double2)));
    {
        std::vector<double2> h(N);
        for (int i = 0; i < N; i++) { const double a = -2.0 * M_PI * i / N; h[i].x = cos(a); h[i].y = sin(a); }
        CK(hipMemcpy(tw, h.data(), N * sizeof(double2), hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_fill, dim3((unsigned)(((size_t)nch * N + 255) / 256)), dim3(256), 0, 0, ring, nch, 12345u);
    std::vector<double> S((size_t)S_NFIELDS * nch, 0.0); std::vector<int> I((size_t)I_NFIELDS * nch, 0);
    for (int c = 0; c < nch; c++)
    {
        S[(size_t)S_LOCKINGBW * nch + c] = 10500.0 - 500.0 * (c % 3);
        S[(size_t)S_THRESH * nch + c] = 0.65; S[(size_t)S_MSE * nch + c] = 1.0;
        S[(size_t)S_M2_FREQ * nch + c] = 8000.0; S[(size_t)S_MC_FREQ * nch + c] = 8000.0;
        I[(size_t)I_BB_PTR * nch + c] = (c * 977) & (N - 1);
    }
    Side sd[3];
    for (int k = 0; k < 3; k++)
    {
        CK(hipMalloc(&sd[k].S, S.size() * 8)); CK(hipMalloc(&sd[k].I, I.size() * 4)); CK(hipMalloc(&sd[k].y, (size_t)nch * N * 8));
        CK(hipMalloc(&sd[k].slog, (size_t)nch * g.log_cap * 6 * 8));
        CK(hipMemcpy(sd[k].S, S.data(), S.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(sd[k].I, I.data(), I.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(sd[k].y, 0, (size_t)nch * N * 8));
    }
    auto ptrs = [&](int k) { JPtrs p = {}; p.S = sd[k].S; p.I = sd[k].I; p.bbring = ring; p.y = sd[k].y; p.slog = sd[k].slog; return p; };
    const int lds4 = 64 * 257 * 8, lds5 = C5_XCH * 8, lds6 = C6_XCH * 8;
    CK(hipFuncSetAttribute((const void *)k_coarse6, hipFuncAttributeMaxDynamicSharedMemorySize, lds6));
    CK(hipFuncSetAttribute((const void *)k_coarse6_w8400, hipFuncAttributeMaxDynamicSharedMemorySize, lds6 + C4_TABN * 8));
    CK(hipFuncSetAttribute((const void *)k_coarse4, hipFuncAttributeMaxDynamicSharedMemorySize, lds4));
    CK(hipFuncSetAttribute((const void *)k_coarse5, hipFuncAttributeMaxDynamicSharedMemorySize, lds5));
    CK(hipFuncSetAttribute((const void *)k5_abl<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds5));
    CK(hipFuncSetAttribute((const void *)k5_abl<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds5));
    CK(hipFuncSetAttribute((const void *)k5_abl<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds5));
    CK(hipFuncSetAttribute((const void *)k5_abl<5>, hipFuncAttributeMaxDynamicSharedMemorySize, lds5));
    CK(hipFuncSetAttribute((const void *)k5_abl<9>, hipFuncAttributeMaxDynamicSharedMemorySize, lds5));
    CK(hipFuncSetAttribute((const void *)k5_abl<13>, hipFuncAttributeMaxDynamicSharedMemorySize, lds5));
    CK(hipFuncSetAttribute((const void *)k_coarse4_w8400, hipFuncAttributeMaxDynamicSharedMemorySize, lds4 + C4_TABN * 8));
    CK(hipFuncSetAttribute((const void *)k_coarse5_w8400, hipFuncAttributeMaxDynamicSharedMemorySize, lds5 + C4_TABN * 8));
    {   // the transform alone against a host FFT
        CK(hipFuncSetAttribute((const void *)k_fft_test, hipFuncAttributeMaxDynamicSharedMemorySize, lds5));
        std::vector<double2> hx(N), hX(N); std::vector<double> re(N), im(N);
        srand(7);
        for (int i = 0; i < N; i++) { hx[i].x = re[i] = rand() / (double)RAND_MAX - 0.5; hx[i].y = im[i] = rand() / (double)RAND_MAX - 0.5; }
        double2 *dx, *dX; CK(hipMalloc(&dx, N * 16)); CK(hipMalloc(&dX, N * 16));
        CK(hipMemcpy(dx, hx.data(), N * 16, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_fft_test, dim3(1), dim3(C2_THREADS), lds5, 0, (const double2 *)dx, dX, (const double2 *)tw);
        CK(hipDeviceSynchronize()); CK(hipGetLastError());
        CK(hipMemcpy(hX.data(), dX, N * 16, hipMemcpyDeviceToHost));
        host_fft(re, im);
        double md = 0; int nbad = 0, firstbad = -1;
        for (int i = 0; i < N; i++) { const double d = fmax(fabs(hX[i].x - re[i]), fabs(hX[i].y - im[i])); md = fmax(md, d); if (d > 1e-9) { nbad++; if (firstbad < 0) firstbad = i; } }
        printf("wg_fft14_2s against a host FFT: max error %.3e, bins off by more than 1e-9: %d (first %d)\n", md, nbad, firstbad);
        CK(hipFuncSetAttribute((const void *)k_fft_test6, hipFuncAttributeMaxDynamicSharedMemorySize, C6_XCH * 8));
        hipLaunchKernelGGL(k_fft_test6, dim3(1), dim3(C2_THREADS), C6_XCH * 8, 0, (const double2 *)dx, dX, (const double2 *)tw);
        CK(hipDeviceSynchronize()); CK(hipGetLastError());
        CK(hipMemcpy(hX.data(), dX, N * 16, hipMemcpyDeviceToHost));
        double md6 = 0; int nbad6 = 0;
        for (int i = 0; i < N; i++) { const double d = fmax(fabs(hX[i].x - re[i]), fabs(hX[i].y - im[i])); md6 = fmax(md6, d); if (d > 1e-9) nbad6++; }
        printf("wg_fft14_e32 against a host FFT: max error %.3e, bins off by more than 1e-9: %d\n", md6, nbad6);
        if (nbad) { for (int i = 0, k = 0; i < N && k < 24; i++) if (fmax(fabs(hX[i].x - re[i]), fabs(hX[i].y - im[i])) > 1e-9) { printf(" %d", i); k++; } printf("\n"); }
    }
    const int grid = nch < ncu ? nch : ncu;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch, int reps) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; r++) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-34s %8.3f ms per launch  (%6.2f us per estimate and CU, alg 512 KiB/estimate -> %6.1f GB/s)\n", name, ms / reps, ms / reps * 1e3 / ((double)nch / grid),
               524288.0 * nch / (ms / reps * 1e-3) / 1e9);
    };
    // parity first: one launch of each on fresh state, then two more (y accumulates), compare
    for (int r = 0; r < 3; r++)
    {
        hipLaunchKernelGGL(k_coarse4, dim3(grid), dim3(C2_THREADS), lds4, 0, g, ptrs(0), (const int *)nullptr, nch, (const double2 *)tw);
        hipLaunchKernelGGL(k_coarse5, dim3(grid), dim3(C2_THREADS), lds5, 0, g, ptrs(1), (const int *)nullptr, nch, (const double2 *)tw);
        hipLaunchKernelGGL(k_coarse6, dim3(grid), dim3(C2_THREADS), lds6, 0, g, ptrs(2), (const int *)nullptr, nch, (const double2 *)tw);
    }
    CK(hipDeviceSynchronize()); CK(hipGetLastError());
    {
        const int ncheck = nch < 512 ? nch : 512;
        std::vector<double> y0((size_t)N), y1((size_t)N), y2((size_t)N), l0((size_t)g.log_cap * 6), l1((size_t)g.log_cap * 6), l2((size_t)g.log_cap * 6);
        double maxd = 0, maxd6 = 0; int badbin = 0, badbin6 = 0, nrow = 0;
        for (int k = 0; k < ncheck; k++)
        {
            const int c = (int)(((long long)k * (nch - 1)) / (ncheck > 1 ? ncheck - 1 : 1));
            CK(hipMemcpy(y0.data(), sd[0].y + (size_t)c * N, N * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(y1.data(), sd[1].y + (size_t)c * N, N * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(y2.data(), sd[2].y + (size_t)c * N, N * 8, hipMemcpyDeviceToHost));
            // the band-limited signal's square occupies |k| < 2 startbin; the bins beyond hold the transform's round-off (~1e-16 of the
            // lines' power) and are never read by the fold (|k| <= startbin + expectedpeakbin + 1): compared inside the support only
            const int hw = 2 * (int)lround((10500.0 - 500.0 * (c % 3)) / (48000.0 / N)) - 8;
            // k_coarse5 keeps y[] in the order its threads hold it: bin i = C5_IDX(H, s, t) ^ N/2 lives at (H*16 + s)*512 + t
            for (int i = N / 2 - hw; i <= N / 2 + hw; i++)
            {
                const int k = i ^ (N / 2), H = (k >> 1) & 1, sl = k >> 10, tt = (((k >> 2) & 255) << 1) | (k & 1);
                maxd = fmax(maxd, fabs(y0[i] - y1[(H * 16 + sl) * 512 + tt]));
                maxd6 = fmax(maxd6, fabs(y0[i] - y2[i]));
            }
            CK(hipMemcpy(l0.data(), sd[0].slog + (size_t)c * g.log_cap * 6, l0.size() * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(l1.data(), sd[1].slog + (size_t)c * g.log_cap * 6, l1.size() * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(l2.data(), sd[2].slog + (size_t)c * g.log_cap * 6, l2.size() * 8, hipMemcpyDeviceToHost));
            for (int r = 0; r < 3; r++) { nrow++; if (l0[r * 6 + 1] != l1[r * 6 + 1] || l0[r * 6 + 0] != l1[r * 6 + 0]) badbin++; if (l0[r * 6 + 1] != l2[r * 6 + 1] || l0[r * 6 + 0] != l2[r * 6 + 0]) badbin6++; }
            if (k < 3) printf("  channel %d: estimates (m2_freq) k_coarse4 %.4f %.4f %.4f | k_coarse5 %.4f %.4f %.4f\n", c, l0[1], l0[7], l0[13], l1[1], l1[7], l1[13]);
        }
        printf("parity over %d channels x 3 estimates: max |y4 - y5| inside the signal's support = %.3e, status rows that differ: %d of %d  -> %s\n", ncheck, maxd, badbin, nrow,
               (maxd < 1e-9 && badbin == 0) ? "OK" : "MISMATCH");
        printf("k_coarse6 against k_coarse4: max |y4 - y6| inside the support = %.3e, status rows that differ: %d of %d  -> %s\n", maxd6, badbin6, nrow, (maxd6 < 1e-9 && badbin6 == 0) ? "OK" : "MISMATCH");
    }
    timeit("k_coarse4", [&] { hipLaunchKernelGGL(k_coarse4, dim3(grid), dim3(C2_THREADS), lds4, 0, g, ptrs(0), (const int *)nullptr, nch, (const double2 *)tw); }, nl);
    timeit("k_coarse5", [&] { hipLaunchKernelGGL(k_coarse5, dim3(grid), dim3(C2_THREADS), lds5, 0, g, ptrs(1), (const int *)nullptr, nch, (const double2 *)tw); }, nl);
    timeit("k_coarse6 (32 x 32 x 16)", [&] { hipLaunchKernelGGL(k_coarse6, dim3(grid), dim3(C2_THREADS), lds6, 0, g, ptrs(2), (const int *)nullptr, nch, (const double2 *)tw); }, nl);
    timeit("k_coarse6_w8400", [&] { hipLaunchKernelGGL(k_coarse6_w8400, dim3(grid), dim3(C2_THREADS), lds6 + C4_TABN * 8, 0, g, ptrs(2), (const int *)nullptr, nch, (const double2 *)tw); }, nl);
    timeit("k_coarse5 no ring/y traffic", [&] { hipLaunchKernelGGL(k5_abl<1>, dim3(grid), dim3(C2_THREADS), lds5, 0, g, ptrs(1), (const int *)nullptr, nch, (const double2 *)tw); }, nl);
    timeit("k_coarse5 no transforms", [&] { hipLaunchKernelGGL(k5_abl<2>, dim3(grid), dim3(C2_THREADS), lds5, 0, g, ptrs(1), (const int *)nullptr, nch, (const double2 *)tw); }, nl);
    timeit("k_coarse5 neither (log10, fold, slot)", [&] { hipLaunchKernelGGL(k5_abl<3>, dim3(grid), dim3(C2_THREADS), lds5, 0, g, ptrs(1), (const int *)nullptr, nch, (const double2 *)tw); }, nl);
    timeit("k_coarse5 no traffic, LDS pieces only", [&] { hipLaunchKernelGGL(k5_abl<5>, dim3(grid), dim3(C2_THREADS), lds5, 0, g, ptrs(1), (const int *)nullptr, nch, (const double2 *)tw); }, nl);
    timeit("k_coarse5 no traffic, VALU pieces only", [&] { hipLaunchKernelGGL(k5_abl<9>, dim3(grid), dim3(C2_THREADS), lds5, 0, g, ptrs(1), (const int *)nullptr, nch, (const double2 *)tw); }, nl);
    timeit("k_coarse5 no traffic, barriers only", [&] { hipLaunchKernelGGL(k5_abl<13>, dim3(grid), dim3(C2_THREADS), lds5, 0, g, ptrs(1), (const int *)nullptr, nch, (const double2 *)tw); }, nl);
    timeit("k_coarse4_w8400", [&] { hipLaunchKernelGGL(k_coarse4_w8400, dim3(grid), dim3(C2_THREADS), lds4 + C4_TABN * 8, 0, g, ptrs(0), (const int *)nullptr, nch, (const double2 *)tw); }, nl);
    timeit("k_coarse5_w8400", [&] { hipLaunchKernelGGL(k_coarse5_w8400, dim3(grid), dim3(C2_THREADS), lds5 + C4_TABN * 8, 0, g, ptrs(1), (const int *)nullptr, nch, (const double2 *)tw); }, nl);
    timeit("k_coarse4 (again)", [&] { hipLaunchKernelGGL(k_coarse4, dim3(grid), dim3(C2_THREADS), lds4, 0, g, ptrs(0), (const int *)nullptr, nch, (const double2 *)tw); }, nl);
    timeit("k_coarse6 (again)", [&] { hipLaunchKernelGGL(k_coarse6, dim3(grid), dim3(C2_THREADS), lds6, 0, g, ptrs(2), (const int *)nullptr, nch, (const double2 *)tw); }, nl);
    timeit("k_coarse5 (again)", [&] { hipLaunchKernelGGL(k_coarse5, dim3(grid), dim3(C2_THREADS), lds5, 0, g, ptrs(1), (const int *)nullptr, nch, (const double2 *)tw); }, nl);
    return 0;
}

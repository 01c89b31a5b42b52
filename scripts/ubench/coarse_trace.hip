// Where one estimate of k_coarse6 spends its time, phase by phase (a 100 MHz clock read by thread 0 of one workgroup at the C6_TRACE points
// of jaero_amd/csrc/k_coarse6.h), and A/B timing + parity of epilogue variants (template parameter EPI of coarse6_body).
// Not part of the product library.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DTRACE] -o scripts/ubench/coarse_trace scripts/ubench/coarse_trace.hip
//   run:   scripts/ubench/coarse_trace [channels = 65536] [launches = 4]
#include <hip/hip_runtime.h>
#ifdef TRACE
__device__ unsigned long long *g_trace;
#define C6_TRACE_WG 37
#define C6_TRACE(i)                                                                                                                                           \
    do {                                                                                                                                                      \
        if (g_trace && threadIdx.x == 0 && blockIdx.x == C6_TRACE_WG)                                                                                         \
        {                                                                                                                                                     \
            const int it_ = (li - (int)blockIdx.x) / (int)gridDim.x;                                                                                          \
            if (it_ < 256) g_trace[it_ * 16 + (i)] = wall_clock64();                                                                                          \
        }                                                                                                                                                     \
    } while (0)
#endif
#include "../../jaero_amd/csrc/k_coarse6.h"
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include <cmath>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int EPI> __global__ __launch_bounds__(C2_THREADS) void k6(const JGeom g, const JPtrs p, const int *cl, int nlist, const double2 *tw)
{
    coarse6_body<false, 14, EPI>(g, p, cl, nlist, tw);
}
template <int EPI> __global__ __launch_bounds__(C2_THREADS) void k6w(const JGeom g, const JPtrs p, const int *cl, int nlist, const double2 *tw)
{
    coarse6_body<true, 14, EPI>(g, p, cl, nlist, tw);
}
template <int EPI> __global__ __launch_bounds__(256, 2) void k613(const JGeom g, const JPtrs p, const int *cl, int nlist, const double2 *tw)
{
    coarse6_body<false, 13, EPI>(g, p, cl, nlist, tw);
}

__global__ void k_fill(double2 *ring, int nch, int N, unsigned seed)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nch * N) return;
    const int ch = (int)(i / N), n = (int)(i % N);
    unsigned h = (unsigned)(i * 2654435761u) ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    unsigned h2 = h * 1664525u + 1013904223u; h2 ^= h2 >> 15;
    const double nr = ((double)(h & 0xffffff) / 16777216.0 - 0.5), ni = ((double)(h2 & 0xffffff) / 16777216.0 - 0.5);
    const double d = (double)((ch * 37) % 2001 - 1000) * 0.5 * N / 16384.0; // bins
    const double a = 1000.0 * cos(2.0 * M_PI * (896.0 * N / 16384.0) * n / N);
    const double ph = 2.0 * M_PI * d * n / N;
    ring[i] = make_double2(a * cos(ph) + 600.0 * nr, a * sin(ph) + 600.0 * ni);
}

struct Side { double *S; int *I; double *y; double *slog; double2 *ring; };

int main(int argc, char **argv)
{
    const int nch = argc > 1 ? atoi(argv[1]) : 65536;
    const int nl = argc > 2 ? atoi(argv[2]) : 4;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs; %d channels, %d timed launches per variant\n", prop.gcnArchName, ncu, nch, nl);
    for (int LOG2N = 14; LOG2N >= 13; LOG2N--)
    {
        const int N = 1 << LOG2N;
        JGeom g = {};
        g.kind = 1; g.nch = nch; g.nchp = nch; g.ngroups = nch / 64; g.Fs = 48000.0; g.fb = LOG2N == 14 ? 10500.0 : 5250.0; g.nfft = N; g.nfft_log2 = LOG2N; g.flags = 2u; g.log_cap = 64;
        double2 *tw; CK(hipMalloc(&tw, N * sizeof(double2)));
        {
            std::vector<double2> h(N);
            for (int i = 0; i < N; i++) { const double a = -2.0 * M_PI * i / N; h[i].x = cos(a); h[i].y = sin(a); }
            CK(hipMemcpy(tw, h.data(), N * sizeof(double2), hipMemcpyHostToDevice));
        }
        std::vector<double> S((size_t)S_NFIELDS * nch, 0.0); std::vector<int> I((size_t)I_NFIELDS * nch, 0);
        for (int c = 0; c < nch; c++)
        {
            S[(size_t)S_LOCKINGBW * nch + c] = (LOG2N == 14 ? 10500.0 : 5250.0) - 500.0 * (c % 3);
            S[(size_t)S_THRESH * nch + c] = 0.65; S[(size_t)S_MSE * nch + c] = (c % 5 == 4) ? 0.2 : 1.0;
            S[(size_t)S_M2_FREQ * nch + c] = 8000.0 + (c % 7); S[(size_t)S_MC_FREQ * nch + c] = 8000.0;
            I[(size_t)I_FLAGS * nch + c] = (c % 11 == 0) ? JF_AFC : 0;
            I[(size_t)I_BB_PTR * nch + c] = (c * 977) & (N - 1);
        }
        const int NV = 2;
        Side sd[NV];
        for (int k = 0; k < NV; k++)
        {
            CK(hipMalloc(&sd[k].S, S.size() * 8)); CK(hipMalloc(&sd[k].I, I.size() * 4)); CK(hipMalloc(&sd[k].y, (size_t)nch * N * 8));
            CK(hipMalloc(&sd[k].slog, (size_t)nch * g.log_cap * 6 * 8));
            CK(hipMemcpy(sd[k].S, S.data(), S.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(sd[k].I, I.data(), I.size() * 4, hipMemcpyHostToDevice));
            CK(hipMemset(sd[k].y, 0, (size_t)nch * N * 8)); CK(hipMemset(sd[k].slog, 0, (size_t)nch * g.log_cap * 6 * 8));
            // a ring per side: an AFC recentre zeroes the channel's ring
            CK(hipMalloc(&sd[k].ring, (size_t)nch * N * sizeof(double2)));
            hipLaunchKernelGGL(k_fill, dim3((unsigned)(((size_t)nch * N + 255) / 256)), dim3(256), 0, 0, sd[k].ring, nch, N, 12345u);
        }
        auto ptrs = [&](int k) { JPtrs p = {}; p.S = sd[k].S; p.I = sd[k].I; p.bbring = sd[k].ring; p.y = sd[k].y; p.slog = sd[k].slog; return p; };
        const int lds = (LOG2N == 14 ? C6_XCH : C6_XCH13) * 8;
        const int grid = LOG2N == 14 ? (nch < ncu ? nch : ncu) : (nch < 2 * ncu ? nch : 2 * ncu);
        const int nthr = LOG2N == 14 ? C2_THREADS : 256;
        CK(hipFuncSetAttribute((const void *)k6<0>, hipFuncAttributeMaxDynamicSharedMemorySize, C6_XCH * 8));
        CK(hipFuncSetAttribute((const void *)k6<1>, hipFuncAttributeMaxDynamicSharedMemorySize, C6_XCH * 8));
        CK(hipFuncSetAttribute((const void *)k6w<0>, hipFuncAttributeMaxDynamicSharedMemorySize, C6_XCH * 8 + C4_TABN * 8));
        CK(hipFuncSetAttribute((const void *)k6w<1>, hipFuncAttributeMaxDynamicSharedMemorySize, C6_XCH * 8 + C4_TABN * 8));
        CK(hipFuncSetAttribute((const void *)k613<0>, hipFuncAttributeMaxDynamicSharedMemorySize, C6_XCH13 * 8));
        CK(hipFuncSetAttribute((const void *)k613<1>, hipFuncAttributeMaxDynamicSharedMemorySize, C6_XCH13 * 8));
        auto launch = [&](int epi, int side) {
            if (LOG2N == 14)
            {
                if (epi == 0) hipLaunchKernelGGL(k6<0>, dim3(grid), dim3(nthr), lds, 0, g, ptrs(side), (const int *)nullptr, nch, (const double2 *)tw);
                else hipLaunchKernelGGL(k6<1>, dim3(grid), dim3(nthr), lds, 0, g, ptrs(side), (const int *)nullptr, nch, (const double2 *)tw);
            }
            else
            {
                if (epi == 0) hipLaunchKernelGGL(k613<0>, dim3(grid), dim3(nthr), lds, 0, g, ptrs(side), (const int *)nullptr, nch, (const double2 *)tw);
                else hipLaunchKernelGGL(k613<1>, dim3(grid), dim3(nthr), lds, 0, g, ptrs(side), (const int *)nullptr, nch, (const double2 *)tw);
            }
        };
        // parity: six launches of each on fresh state (y accumulates, the countdowns run down, AFC channels recentre), then compare everything
        for (int r = 0; r < 6; r++) { launch(0, 0); launch(1, 1); }
        CK(hipDeviceSynchronize()); CK(hipGetLastError());
        {
            std::vector<double> ya((size_t)N), yb((size_t)N), la((size_t)g.log_cap * 6), lb((size_t)g.log_cap * 6);
            std::vector<double> Sa(S.size()), Sb(S.size()); std::vector<int> Ia(I.size()), Ib(I.size());
            CK(hipMemcpy(Sa.data(), sd[0].S, S.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(Sb.data(), sd[1].S, S.size() * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(Ia.data(), sd[0].I, I.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(Ib.data(), sd[1].I, I.size() * 4, hipMemcpyDeviceToHost));
            const bool state_same = memcmp(Sa.data(), Sb.data(), S.size() * 8) == 0 && memcmp(Ia.data(), Ib.data(), I.size() * 4) == 0;
            long ydiff = 0, ldiff = 0; int nbig = 0;
            const int ncheck = nch < 1024 ? nch : 1024;
            for (int k = 0; k < ncheck; k++)
            {
                const int c = (int)(((long long)k * (nch - 1)) / (ncheck > 1 ? ncheck - 1 : 1));
                CK(hipMemcpy(ya.data(), sd[0].y + (size_t)c * N, N * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(yb.data(), sd[1].y + (size_t)c * N, N * 8, hipMemcpyDeviceToHost));
                if (memcmp(ya.data(), yb.data(), N * 8) != 0) ydiff++;
                CK(hipMemcpy(la.data(), sd[0].slog + (size_t)c * g.log_cap * 6, la.size() * 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(lb.data(), sd[1].slog + (size_t)c * g.log_cap * 6, lb.size() * 8, hipMemcpyDeviceToHost));
                if (memcmp(la.data(), lb.data(), la.size() * 8) != 0) ldiff++;
                if (ya[5] == 20.0) nbig++;
            }
            printf("N = 2^%d  EPI 1 against EPI 0 after 6 launches: state arrays %s, y rows that differ %ld of %d, status logs that differ %ld (channels just recentred: %d)  -> %s\n", LOG2N,
                   state_same ? "identical" : "DIFFER", ydiff, ncheck, ldiff, nbig, (state_same && !ydiff && !ldiff) ? "OK" : "MISMATCH");
        }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto timeit = [&](const char *name, int epi) {
            launch(epi, epi); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < nl; r++) launch(epi, epi);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("N = 2^%d  %-28s %8.3f ms per launch  (%6.2f us per estimate and workgroup, alg %d KiB/estimate -> %6.1f GB/s)\n", LOG2N, name, ms / nl, ms / nl * 1e3 / ((double)nch / grid),
                   32 * N / 1024, 32.0 * N * nch / (ms / nl * 1e-3) / 1e9);
        };
        for (int rep = 0; rep < 2; rep++) { timeit("EPI 0 (product)", 0); timeit("EPI 1", 1); }
#ifdef TRACE
        for (int epi = 0; epi < 2; epi++)
        {
            unsigned long long *dtr; CK(hipMalloc(&dtr, 256 * 16 * 8)); CK(hipMemset(dtr, 0, 256 * 16 * 8));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &dtr, sizeof(dtr)));
            launch(epi, epi); CK(hipDeviceSynchronize());
            std::vector<unsigned long long> tr(256 * 16);
            CK(hipMemcpy(tr.data(), dtr, tr.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long *nul = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &nul, sizeof(nul)));
            const int iters = nch / grid < 256 ? nch / grid : 256;
            const char *ph[] = {"FFT 1", "band limit + FFT 2", "square + FFT 3", "|X|^2, y loads, 32 log10, y stores + LDS copy", "barrier", "ring prefetch issue", "fold", "shuffle reduce",
                                "barrier + reduce 8 + slot", "second barrier / recentre", "loop latch -> next top"};
            double acc[11] = {0}; int cnt = 0;
            for (int it = 4; it + 1 < iters; it++)
            {
                bool ok = true;
                for (int k = 0; k <= 10; k++) if (!tr[it * 16 + k]) ok = false;
                if (!ok || !tr[(it + 1) * 16]) continue;
                for (int k = 0; k < 10; k++) acc[k] += (double)(tr[it * 16 + k + 1] - tr[it * 16 + k]) * 0.01;
                acc[10] += (double)(tr[(it + 1) * 16] - tr[it * 16 + 10]) * 0.01;
                cnt++;
            }
            double tot = 0; for (int k = 0; k < 11; k++) tot += acc[k] / (cnt ? cnt : 1);
            printf("N = 2^%d  EPI %d: phases of one estimate on workgroup %d, mean over %d estimates (us), total %.2f\n", LOG2N, epi, C6_TRACE_WG, cnt, tot);
            for (int k = 0; k < 11; k++) printf("    %-48s %6.2f\n", ph[k], acc[k] / (cnt ? cnt : 1));
            CK(hipFree(dtr));
        }
#endif
        for (int k = 0; k < NV; k++) { CK(hipFree(sd[k].S)); CK(hipFree(sd[k].I)); CK(hipFree(sd[k].y)); CK(hipFree(sd[k].slog)); CK(hipFree(sd[k].ring)); }
        CK(hipFree(tw));
    }
    return 0;
}

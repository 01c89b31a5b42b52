// Where one estimate of k_coarse6 spends its time, phase by phase (a 100 MHz clock read by thread 0 of one workgroup at the C6_TRACE points
// of jaero_amd/csrc/k_coarse6.h), and the launch times of the three product kernels.  (Round 4 used it to A/B epilogue variants through a
// template parameter of coarse6_body; what won is the product code, the losers' numbers are in DESIGN 9 item 15.)  Parity of the
// kernel against its predecessors: scripts/ubench/coarse_bench.hip.  Not part of the product library.
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
        Side sd[1];
        const int NV = 1;
        for (int k = 0; k < NV; k++)
        {
            CK(hipMalloc(&sd[k].S, S.size() * 8)); CK(hipMalloc(&sd[k].I, I.size() * 4)); CK(hipMalloc(&sd[k].y, (size_t)nch * N * 8));
            CK(hipMalloc(&sd[k].slog, (size_t)nch * g.log_cap * 6 * 8));
            CK(hipMemcpy(sd[k].S, S.data(), S.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(sd[k].I, I.data(), I.size() * 4, hipMemcpyHostToDevice));
            CK(hipMemset(sd[k].y, 0, (size_t)nch * N * 8)); CK(hipMemset(sd[k].slog, 0, (size_t)nch * g.log_cap * 6 * 8));
            CK(hipMalloc(&sd[k].ring, (size_t)nch * N * sizeof(double2)));
            hipLaunchKernelGGL(k_fill, dim3((unsigned)(((size_t)nch * N + 255) / 256)), dim3(256), 0, 0, sd[k].ring, nch, N, 12345u);
        }
        auto ptrs = [&](int k) { JPtrs p = {}; p.S = sd[k].S; p.I = sd[k].I; p.bbring = sd[k].ring; p.y = sd[k].y; p.slog = sd[k].slog; return p; };
        const int lds = (LOG2N == 14 ? C6_XCH : C6_XCH13) * 8;
        const int grid = LOG2N == 14 ? (nch < ncu ? nch : ncu) : (nch < 2 * ncu ? nch : 2 * ncu);
        const int nthr = LOG2N == 14 ? C2_THREADS : 256;
        CK(hipFuncSetAttribute((const void *)k_coarse6, hipFuncAttributeMaxDynamicSharedMemorySize, C6_XCH * 8));
        CK(hipFuncSetAttribute((const void *)k_coarse6_w8400, hipFuncAttributeMaxDynamicSharedMemorySize, C6_XCH * 8 + C4_TABN * 8));
        CK(hipFuncSetAttribute((const void *)k_coarse6_13, hipFuncAttributeMaxDynamicSharedMemorySize, C6_XCH13 * 8));
        auto launch = [&](int which, int side) {
            if (LOG2N == 13) hipLaunchKernelGGL(k_coarse6_13, dim3(grid), dim3(nthr), lds, 0, g, ptrs(side), (const int *)nullptr, nch, (const double2 *)tw);
            else if (which == 1) hipLaunchKernelGGL(k_coarse6_w8400, dim3(grid), dim3(nthr), lds + C4_TABN * 8, 0, g, ptrs(side), (const int *)nullptr, nch, (const double2 *)tw);
            else hipLaunchKernelGGL(k_coarse6, dim3(grid), dim3(nthr), lds, 0, g, ptrs(side), (const int *)nullptr, nch, (const double2 *)tw);
        };
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto timeit = [&](const char *name, int epi) {
            launch(epi, 0); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < nl; r++) launch(epi, 0);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("N = 2^%d  %-28s %8.3f ms per launch  (%6.2f us per estimate and workgroup, alg %d KiB/estimate -> %6.1f GB/s)\n", LOG2N, name, ms / nl, ms / nl * 1e3 / ((double)nch / grid),
                   32 * N / 1024, 32.0 * N * nch / (ms / nl * 1e-3) / 1e9);
        };
        for (int rep = 0; rep < 2; rep++) { timeit(LOG2N == 14 ? "k_coarse6" : "k_coarse6_13", 0); if (LOG2N == 14) timeit("k_coarse6_w8400", 1); }
#ifdef TRACE
        for (int epi = 0; epi < (LOG2N == 14 ? 2 : 1); epi++)
        {
            unsigned long long *dtr; CK(hipMalloc(&dtr, 256 * 16 * 8)); CK(hipMemset(dtr, 0, 256 * 16 * 8));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &dtr, sizeof(dtr)));
            launch(epi, 0); CK(hipDeviceSynchronize());
            std::vector<unsigned long long> tr(256 * 16);
            CK(hipMemcpy(tr.data(), dtr, tr.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long *nul = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &nul, sizeof(nul)));
            const int iters = nch / grid < 256 ? nch / grid : 256;
            const int order[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10};
            const char *ph[] = {"FFT 1", "band limit + FFT 2", "square + FFT 3", "barrier, |X|^2, y loads, per slot: log10, smooth, y store, LDS copy, next ring load", "barrier",
                                "(fold set-up)", "fold", "wavefront reduce", "barrier + reduce 8 + slot (every thread)", "recentre (rare)", "loop latch -> next top"};
            const int NP = 11;
            double acc[NP] = {0}; int cnt = 0;
            for (int it = 4; it + 1 < iters; it++)
            {
                bool ok = true;
                for (int k = 0; k < NP; k++) if (!tr[it * 16 + order[k]]) ok = false;
                if (!ok || !tr[(it + 1) * 16]) continue;
                for (int k = 0; k + 1 < NP; k++) acc[k] += (double)(tr[it * 16 + order[k + 1]] - tr[it * 16 + order[k]]) * 0.01;
                acc[NP - 1] += (double)(tr[(it + 1) * 16] - tr[it * 16 + order[NP - 1]]) * 0.01;
                cnt++;
            }
            double tot = 0; for (int k = 0; k < NP; k++) tot += acc[k] / (cnt ? cnt : 1);
            printf("N = 2^%d  %s: phases of one estimate on workgroup %d, mean over %d estimates (us), total %.2f\n", LOG2N, LOG2N == 13 ? "k_coarse6_13" : epi ? "k_coarse6_w8400" : "k_coarse6", C6_TRACE_WG, cnt, tot);
            for (int k = 0; k < NP; k++) printf("    %-48s %6.2f\n", ph[k], acc[k] / (cnt ? cnt : 1));
            CK(hipFree(dtr));
        }
#endif
        for (int k = 0; k < NV; k++) { CK(hipFree(sd[k].S)); CK(hipFree(sd[k].I)); CK(hipFree(sd[k].y)); CK(hipFree(sd[k].slog)); CK(hipFree(sd[k].ring)); }
        CK(hipFree(tw));
    }
    return 0;
}

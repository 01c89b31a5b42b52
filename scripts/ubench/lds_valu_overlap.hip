// Do LDS traffic and fp64 VALU work overlap on a gfx950 CU, and how must they be arranged?  One 512-thread workgroup per CU (two
// wavefronts per SIMD, as k_coarse5), per iteration each wavefront has NL conflict-free 64-bit LDS operations and NV fp64 FMAs:
//   lds_only / valu_only          the two floors
//   phased                        every wavefront: NL LDS operations, then NV FMAs (what a barrier-separated exchange does)
//   interleaved<G>                every wavefront: one LDS operation after every G FMAs
//   split                         wavefronts 0-3 do 2 NL LDS operations each, wavefronts 4-7 do 2 NV FMAs each (one of each kind per SIMD)
//   phased_skewed                 wavefronts 0-3: LDS then FMAs, wavefronts 4-7: FMAs then LDS
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/lds_valu_overlap scripts/ubench/lds_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define NL 32
#define NV 160
#define ITER 400

#define FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[(i) & 7]) : "v"(m), "v"(c))
#define LDW(i) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(addr), "v"(wd[(i) & 7]), "n"(((i) & 15) * 4096) : "memory")
#define LDR(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ld[(i) & 7]) : "v"(addr), "n"(((i) & 15) * 4096) : "memory")

template <int MODE, int G>
__global__ __launch_bounds__(512) void k(double *out)
{
    extern __shared__ double lds[];
    const int t = threadIdx.x;
    const unsigned addr = t * 8; // 512 consecutive doubles per operation and workgroup: conflict-free
    double acc[8], ld[8], wd[8]; // the LDS operations never touch the FMAs' registers (as an exchange of one stream beside the other's FFT)
    for (int i = 0; i < 8; i++) { acc[i] = t + i; ld[i] = 0; wd[i] = t - i; asm volatile("" : "+v"(wd[i])); }
    const double m = 0.999999, c = 1e-9;
    const bool grpB = __builtin_amdgcn_readfirstlane(t >> 8) != 0;
    for (int it = 0; it < ITER; it++)
    {
        if (MODE == 0) // lds only
        {
#pragma unroll
            for (int i = 0; i < NL; i++) { if (i & 1) LDR(i); else LDW(i); }
        }
        else if (MODE == 1) // valu only
        {
#pragma unroll
            for (int i = 0; i < NV; i++) FMA(i);
        }
        else if (MODE == 2) // phased
        {
#pragma unroll
            for (int i = 0; i < NL; i++) { if (i & 1) LDR(i); else LDW(i); }
#pragma unroll
            for (int i = 0; i < NV; i++) FMA(i);
        }
        else if (MODE == 3) // interleaved: one LDS op after every G FMAs
        {
#pragma unroll
            for (int i = 0; i < NV; i++)
            {
                FMA(i);
                if ((i % G) == G - 1 && i / G < NL) { if ((i / G) & 1) LDR(i / G); else LDW(i / G); }
            }
        }
        else if (MODE == 4) // split across the two wavefronts of a SIMD
        {
            if (!grpB)
            {
#pragma unroll
                for (int i = 0; i < 2 * NL; i++) { if (i & 1) LDR(i); else LDW(i); }
            }
            else
            {
#pragma unroll
                for (int i = 0; i < 2 * NV; i++) FMA(i);
            }
        }
        else if (MODE == 5) // phased, the two wavefronts of a SIMD in opposite order
        {
            if (!grpB)
            {
#pragma unroll
                for (int i = 0; i < NL; i++) { if (i & 1) LDR(i); else LDW(i); }
#pragma unroll
                for (int i = 0; i < NV; i++) FMA(i);
            }
            else
            {
#pragma unroll
                for (int i = 0; i < NV; i++) FMA(i);
#pragma unroll
                for (int i = 0; i < NL; i++) { if (i & 1) LDR(i); else LDW(i); }
            }
        }
        if (MODE == 2 || MODE == 5 || MODE == 6)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += acc[i] + ld[i];
    out[blockIdx.x * 512 + t] = s;
}

template <int MODE, int G>
static void run(const char *name, double *d)
{
    hipFuncSetAttribute((const void *)k<MODE, G>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, G>), dim3(256), dim3(512), 65536 + 4096, 0, d);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k<MODE, G>), dim3(256), dim3(512), 65536 + 4096, 0, d);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns_it = ms / 5 * 1e6 / ITER;
    printf("%-28s %8.1f ns per iteration (LDS alone would move %d x 512 x 8 B = %d KiB: %.0f B/ns per CU; FMAs alone: %.2f ns per wave-instruction and SIMD)\n",
           name, ns_it, NL, NL * 4, NL * 4096.0 / ns_it, ns_it / (2.0 * NV));
}

int main()
{
    double *d; hipMalloc(&d, 256 * 512 * 8);
    run<0, 1>("lds_only", d);
    run<1, 1>("valu_only", d);
    run<2, 1>("phased (+barrier)", d);
    run<3, 5>("interleaved 1 per 5", d);
    run<3, 2>("interleaved 1 per 2 (front)", d);
    run<4, 1>("split by wavefront", d);
    run<5, 1>("phased skewed (+barrier)", d);
    return 0;
}

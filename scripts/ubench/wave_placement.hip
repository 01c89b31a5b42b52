// Where do the wavefronts of a 512-thread (and a 128-thread) workgroup land?  Prints SIMD_ID / WAVE_ID (HW_REG_HW_ID) per wave for a
// few workgroups of a grid that fills the chip with one 160 KiB-LDS workgroup per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/wave_placement scripts/ubench/wave_placement.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned *out, int spin)
{
    extern __shared__ double lds[];
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = hw;
    // stay resident for a while so that every workgroup of the grid is placed before the first one leaves
    double a = threadIdx.x;
    for (int i = 0; i < spin; i++) a = a * 1.0000001 + 0.5;
    if (a == 123.456) lds[threadIdx.x] = a;
}
int main()
{
    unsigned *d, h[1024 * 16];
    hipMalloc(&d, sizeof h);
    for (int cfg = 0; cfg < 2; cfg++)
    {
        const int threads = cfg == 0 ? 512 : 128, lds = cfg == 0 ? 163840 : 40960, grid = cfg == 0 ? 256 : 1024;
        hipMemset(d, 0xff, sizeof h);
        hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(probe, dim3(grid), dim3(threads), lds, 0, d, 200000);
        hipDeviceSynchronize();
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("threads/WG %d, LDS %d, grid %d\n", threads, lds, grid);
        for (int b = 0; b < (cfg == 0 ? 6 : 12); b++)
        {
            printf(" WG %3d:", b);
            for (int w = 0; w < threads / 64; w++)
            {
                const unsigned v = h[b * 16 + w];
                printf("  w%d simd%u slot%u cu%u", w, (v >> 4) & 3, v & 15, (v >> 8) & 15);
            }
            printf("\n");
        }
        // histogram: for 512-thread WGs, how often do waves w and w+4 share a SIMD; for 128-thread WGs: SIMD pair of (w0, w1)
        int share = 0, tot = 0, hist[4][4] = {{0}};
        for (int b = 0; b < grid; b++)
        {
            if (cfg == 0) for (int w = 0; w < 4; w++) { share += ((h[b * 16 + w] >> 4) & 3) == ((h[b * 16 + w + 4] >> 4) & 3); tot++; }
            else hist[(h[b * 16] >> 4) & 3][(h[b * 16 + 1] >> 4) & 3]++;
        }
        if (cfg == 0) printf(" waves (w, w+4) on the same SIMD: %d of %d\n", share, tot);
        else for (int a = 0; a < 4; a++) printf(" w0 on simd%d: w1 on simd0..3 = %d %d %d %d\n", a, hist[a][0], hist[a][1], hist[a][2], hist[a][3]);
    }
    return 0;
}

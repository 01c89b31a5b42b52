// Shader clock seen by back-to-back kernels after an idle gap (gfx950): clock64() cycles / wall_clock64() (100 MHz) per launch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
__global__ __launch_bounds__(64) void spin(unsigned *out, long long *clk, int iters)
{
    const long long c0 = clock64(), w0 = wall_clock64();
    unsigned a = threadIdx.x, b = 3, c = 5, d = 7;
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
        for (int k = 0; k < 64; k++)
            asm volatile("v_pk_add_u16 %0, %0, %1\n v_pk_min_u16 %1, %1, %2\n v_pk_sub_u16 %2, %2, %3\n v_lshl_or_b32 %3, %3, 1, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
int main()
{
    unsigned *d; long long *clk, h[2];
    hipMalloc(&d, 2048 * 64 * 4); hipMalloc(&clk, 16);
    for (int gap_ms : {0, 1, 5, 50})
        for (int rep = 0; rep < 4; rep++)
        {
            if (rep == 0) { hipDeviceSynchronize(); usleep(gap_ms * 1000); }
            hipLaunchKernelGGL(spin, dim3(1024), dim3(64), 0, 0, d, clk, 3000);
            hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            printf("gap %2d ms, launch %d: %.3f ms, shader clock %.0f MHz\n", gap_ms, rep, h[1] / 1e5, (double)h[0] / h[1] * 100.0);
        }
    return 0;
}

// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on the access patterns of the sample-loop kernels (MI355X_MICROARCH.md: the two
// counters are only calibrated for 16-B-per-lane streaming reads).  Each kernel moves a known number of bytes:
//   rd8_rows   64 lanes x 8 B = one 512-B row per load, rows streamed                  (AGC / EbNo window rows)
//   wr8_rows   the same as stores
//   wr16_own   every lane stores 16 B at the start of ITS OWN 64-B sector (stride 64 B per lane and 16 B per step within the sector)
//              = the coarse-frequency ring fill: 4 consecutive steps complete a sector
//   rd16_stream / wr16_stream  16 B per lane, fully coalesced (the documented reference pattern)
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/hbm_counters scripts/ubench/hbm_counters.hip
// run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./hbm_counters ; rocprofv3 --kernel-trace --pmc WRITE_SIZE -- ./hbm_counters
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ROWS 4096 // per wavefront
__global__ __launch_bounds__(64) void rd8_rows(const double *p, double *out)
{
    const double *q = p + (size_t)blockIdx.x * ROWS * 64 + threadIdx.x;
    double a = 0;
    for (int r = 0; r < ROWS; r++) a += q[(size_t)r * 64];
    if (a == 1.2345) out[0] = a;
}
__global__ __launch_bounds__(64) void wr8_rows(double *p)
{
    double *q = p + (size_t)blockIdx.x * ROWS * 64 + threadIdx.x;
    for (int r = 0; r < ROWS; r++) q[(size_t)r * 64] = (double)r;
}
__global__ __launch_bounds__(64) void wr16_own(double2 *p)
{
    // lane l owns the contiguous range [ (wave*64 + l) * ROWS, +ROWS ) of double2: 16 B per step, a 64-B sector every 4 steps
    double2 *q = p + ((size_t)blockIdx.x * 64 + threadIdx.x) * ROWS;
    for (int r = 0; r < ROWS; r++) q[r] = make_double2((double)r, 1.0);
}
__global__ __launch_bounds__(64) void rd16_stream(const double2 *p, double *out)
{
    const double2 *q = p + (size_t)blockIdx.x * ROWS * 64 + threadIdx.x;
    double a = 0;
    for (int r = 0; r < ROWS; r++) { double2 v = q[(size_t)r * 64]; a += v.x + v.y; }
    if (a == 1.2345) out[0] = a;
}
__global__ __launch_bounds__(64) void wr16_stream(double2 *p)
{
    double2 *q = p + (size_t)blockIdx.x * ROWS * 64 + threadIdx.x;
    for (int r = 0; r < ROWS; r++) q[(size_t)r * 64] = make_double2((double)r, 2.0);
}
// the same rows with the non-temporal hint k_coarse6 gives its streamed accesses since round 4 (y[] read and written once per estimate, the ring read once)
__global__ __launch_bounds__(64) void rd8_rows_nt(const double *p, double *out)
{
    const double *q = p + (size_t)blockIdx.x * ROWS * 64 + threadIdx.x;
    double a = 0;
    for (int r = 0; r < ROWS; r++) a += __builtin_nontemporal_load(q + (size_t)r * 64);
    if (a == 1.2345) out[0] = a;
}
__global__ __launch_bounds__(64) void wr8_rows_nt(double *p)
{
    double *q = p + (size_t)blockIdx.x * ROWS * 64 + threadIdx.x;
    for (int r = 0; r < ROWS; r++) __builtin_nontemporal_store((double)r, q + (size_t)r * 64);
}
__global__ __launch_bounds__(64) void rd16_stream_nt(const double2 *p, double *out)
{
    typedef double v2 __attribute__((ext_vector_type(2)));
    const v2 *q = (const v2 *)p + (size_t)blockIdx.x * ROWS * 64 + threadIdx.x;
    double a = 0;
    for (int r = 0; r < ROWS; r++) { const v2 v = __builtin_nontemporal_load(q + (size_t)r * 64); a += v.x + v.y; }
    if (a == 1.2345) out[0] = a;
}
int main()
{
    const int waves = 4096;
    const size_t n16 = (size_t)waves * ROWS * 64;
    double2 *d; double *o;
    hipMalloc(&d, n16 * sizeof(double2)); hipMalloc(&o, 64);
    hipMemset(d, 0, n16 * sizeof(double2));
    hipDeviceSynchronize();
    printf("bytes moved per kernel: rd8_rows/wr8_rows %.3f GB, wr16_own/rd16_stream/wr16_stream %.3f GB\n", n16 * 8 / 1e9, n16 * 16 / 1e9);
    hipLaunchKernelGGL(rd8_rows, dim3(waves), dim3(64), 0, 0, (const double *)d, o);
    hipLaunchKernelGGL(wr8_rows, dim3(waves), dim3(64), 0, 0, (double *)d);
    hipLaunchKernelGGL(wr16_own, dim3(waves), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(rd16_stream, dim3(waves), dim3(64), 0, 0, (const double2 *)d, o);
    hipLaunchKernelGGL(wr16_stream, dim3(waves), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(rd8_rows_nt, dim3(waves), dim3(64), 0, 0, (const double *)d, o);
    hipLaunchKernelGGL(wr8_rows_nt, dim3(waves), dim3(64), 0, 0, (double *)d);
    hipLaunchKernelGGL(rd16_stream_nt, dim3(waves), dim3(64), 0, 0, (const double2 *)d, o);
    hipDeviceSynchronize();
    return 0;
}

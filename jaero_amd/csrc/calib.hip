// calib.hip -- two fixed calibration kernels for bench.py (measurement infrastructure, NOT part of the demodulator path; its own small library
// libjaero_calib.so so that the product ABI of include/jaero_hip.h stays the reference's surface).
//
// Why: the driver's fresh box ran the same device code 20 % slower than the builder's boxes, and nothing in the bench line could
// tell a clock- or power-limited device from a regression.  bench.py now times these two kernels immediately before and after the timed region
// of every workload and reports calib{fp64_tflops, hbm_gbs}; the headline's roofline fractions are also given normalised to them.
//   k_calib_fp64:     every SIMD of the chip issues nothing but dependent-free v_fma_f64 (8 independent chains per lane, 2 wavefronts per SIMD);
//                     the rate it reaches is the fp64 issue rate THIS box sustains at THIS moment (78.6 TFLOP/s at 2.4 GHz).
//   k_calib_hbm_copy: 16 bytes per lane, grid-stride copy of a buffer far larger than the 256 MB of last-level cache; bytes read + written over time.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

struct JCalib
{
    int device;
    size_t bytes;          // size of EACH of the two copy buffers
    uint4 *src, *dst;
    double *sink;
    hipEvent_t e0, e1;
};

__global__ __launch_bounds__(128) void k_calib_fp64(double *sink, int iters, double seed)
{
    // 8 independent fma chains: with 2 wavefronts per SIMD the 4-cycle issue of a wavefront-wide fp64 instruction is always covered
    double a0 = seed + threadIdx.x, a1 = a0 + 1.0, a2 = a0 + 2.0, a3 = a0 + 3.0, a4 = a0 + 4.0, a5 = a0 + 5.0, a6 = a0 + 6.0, a7 = a0 + 7.0;
    const double m = 0.9999999, c = 1.0e-7;
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
        for (int u = 0; u < 8; u++)
        {
            a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
            a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
        }
    }
    const double s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    if (s == 123.456) sink[0] = s; // never true: keeps the chains alive
}

__global__ __launch_bounds__(256) void k_calib_hbm_copy(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

#define CK(x)                                                                                                                                             \
    do {                                                                                                                                                  \
        hipError_t e_ = (x);                                                                                                                              \
        if (e_ != hipSuccess) { fprintf(stderr, "libjaero_calib: %s -> %s\n", #x, hipGetErrorString(e_)); return -(int)e_; }                              \
    } while (0)

extern "C" {

int jaero_calib_create(int device, size_t copy_bytes, JCalib **out)
{
    CK(hipSetDevice(device));
    JCalib *c = new JCalib();
    c->device = device;
    c->bytes = copy_bytes & ~(size_t)4095;
    CK(hipMalloc(&c->src, c->bytes));
    CK(hipMalloc(&c->dst, c->bytes));
    CK(hipMalloc(&c->sink, 64));
    CK(hipMemset(c->src, 0x5a, c->bytes));
    CK(hipMemset(c->dst, 0, c->bytes));
    CK(hipEventCreate(&c->e0));
    CK(hipEventCreate(&c->e1));
    CK(hipDeviceSynchronize());
    *out = c;
    return 0;
}

// `iters` outer iterations of 64 fma per lane on 256 CUs x 4 SIMDs x 2 wavefronts; returns the kernel's duration (HIP events on `stream`) and the flops it performed
int jaero_calib_fp64(JCalib *c, void *stream, int iters, double *ms, double *flops)
{
    CK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, c->device));
    const int blocks = pr.multiProcessorCount * 4; // 128 threads = 2 wavefronts per block, 4 blocks per CU: 2 wavefronts on every SIMD
    CK(hipEventRecord(c->e0, s));
    hipLaunchKernelGGL(k_calib_fp64, dim3(blocks), dim3(128), 0, s, c->sink, iters, 0.5);
    CK(hipEventRecord(c->e1, s));
    CK(hipEventSynchronize(c->e1));
    float t = 0;
    CK(hipEventElapsedTime(&t, c->e0, c->e1));
    *ms = t;
    *flops = (double)blocks * 128.0 * (double)iters * 64.0 * 2.0;
    return 0;
}

// `reps` copies src -> dst of the whole buffer; returns the mean duration of one copy; bytes moved per copy = 2 x copy_bytes
int jaero_calib_hbm_copy(JCalib *c, void *stream, int reps, double *ms_per_copy, double *bytes_per_copy)
{
    CK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t n = c->bytes / sizeof(uint4);
    CK(hipEventRecord(c->e0, s));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_calib_hbm_copy, dim3(256 * 16), dim3(256), 0, s, c->dst, c->src, n);
    CK(hipEventRecord(c->e1, s));
    CK(hipEventSynchronize(c->e1));
    float t = 0;
    CK(hipEventElapsedTime(&t, c->e0, c->e1));
    *ms_per_copy = t / (reps > 0 ? reps : 1);
    *bytes_per_copy = 2.0 * (double)c->bytes;
    return 0;
}

void jaero_calib_destroy(JCalib *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipFree(c->src); hipFree(c->dst); hipFree(c->sink);
    hipEventDestroy(c->e0); hipEventDestroy(c->e1);
    delete c;
}
}

// Issue rate of the VALU ops the lane-layout Viterbi is built from, one wavefront per SIMD (gfx950).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates scripts/ubench/valu_rates.hip ; prints ns per wave-instruction
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define KERNEL(name, INSTR)                                                                       \
    __global__ __launch_bounds__(64) void name(unsigned *out, int iters)                          \
    {                                                                                             \
        unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = threadIdx.x ^ 5, d = 7, k = out[0]; \
        for (int i = 0; i < iters; i++) { REP64(asm volatile(INSTR : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k));) } \
        out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;                                       \
    }
// four independent chains per asm statement
KERNEL(k_add_u32, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4")
KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4")
KERNEL(k_pk_add_opsel, "v_pk_add_u16 %0, %0, %4 op_sel:[1,0] op_sel_hi:[1,1]\n v_pk_add_u16 %1, %1, %4 op_sel_hi:[0,1]\n v_pk_add_u16 %2, %2, %4 op_sel:[1,0] op_sel_hi:[1,1]\n v_pk_add_u16 %3, %3, %4 op_sel_hi:[0,1]")
KERNEL(k_pk_min_u16, "v_pk_min_u16 %0, %0, %4\n v_pk_min_u16 %1, %1, %4\n v_pk_min_u16 %2, %2, %4\n v_pk_min_u16 %3, %3, %4")
KERNEL(k_pk_sub_u16, "v_pk_sub_u16 %0, %0, %4\n v_pk_sub_u16 %1, %1, %4\n v_pk_sub_u16 %2, %2, %4\n v_pk_sub_u16 %3, %3, %4")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %4\n v_lshl_or_b32 %1, %1, 1, %4\n v_lshl_or_b32 %2, %2, 1, %4\n v_lshl_or_b32 %3, %3, 1, %4")
KERNEL(k_min_u32, "v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4")
KERNEL(k_add_u16, "v_add_u16 %0, %0, %4\n v_add_u16 %1, %1, %4\n v_add_u16 %2, %2, %4\n v_add_u16 %3, %3, %4")
KERNEL(k_min_u16, "v_min_u16 %0, %0, %4\n v_min_u16 %1, %1, %4\n v_min_u16 %2, %2, %4\n v_min_u16 %3, %3, %4")
KERNEL(k_dep_pk, "v_pk_add_u16 %0, %0, %4\n v_pk_min_u16 %0, %0, %1\n v_pk_sub_u16 %0, %0, %2\n v_pk_add_u16 %0, %0, %3")
KERNEL(k_dep_u32, "v_add_u32 %0, %0, %4\n v_min_u32 %0, %0, %1\n v_sub_u32 %0, %0, %2\n v_add_u32 %0, %0, %3")
KERNEL(k_cmp_cnd, "v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_u32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %3, vcc")
KERNEL(k_sad_u16, "v_sad_u16 %0, %0, %4, %1\n v_sad_u16 %1, %1, %4, %2\n v_sad_u16 %2, %2, %4, %3\n v_sad_u16 %3, %3, %4, %0")
KERNEL(k_perm, "v_perm_b32 %0, %0, %4, %1\n v_perm_b32 %1, %1, %4, %2\n v_perm_b32 %2, %2, %4, %3\n v_perm_b32 %3, %3, %4, %0")

// fp64: operands are register pairs
#define KERNEL64(name, INSTR)                                                                     \
    __global__ __launch_bounds__(64) void name(unsigned *out, int iters)                          \
    {                                                                                             \
        double a = threadIdx.x, b = 1.0 + threadIdx.x * 1e-3, c = 0.5, d = 0.25, k = 1.0 + out[0] * 1e-9; \
        for (int i = 0; i < iters; i++) { REP64(asm volatile(INSTR : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k));) } \
        out[blockIdx.x * 64 + threadIdx.x] = (unsigned)(a + b + c + d);                           \
    }
KERNEL64(k_fma_f64, "v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4")
KERNEL64(k_add_f64, "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4")
KERNEL64(k_mul_f64, "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4")
KERNEL64(k_rcp_f64, "v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3")
KERNEL64(k_sqrt_f64, "v_sqrt_f64 %0, %0\n v_sqrt_f64 %1, %1\n v_sqrt_f64 %2, %2\n v_sqrt_f64 %3, %3")
KERNEL64(k_ldexp_f64, "v_ldexp_f64 %0, %0, 1\n v_ldexp_f64 %1, %1, 1\n v_ldexp_f64 %2, %2, 1\n v_ldexp_f64 %3, %3, 1")
KERNEL64(k_divfix_f64, "v_div_fixup_f64 %0, %0, %4, %4\n v_div_fixup_f64 %1, %1, %4, %4\n v_div_fixup_f64 %2, %2, %4, %4\n v_div_fixup_f64 %3, %3, %4, %4")
KERNEL64(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %4\n v_cmp_lt_f64 vcc, %1, %4\n v_cmp_lt_f64 vcc, %2, %4\n v_cmp_lt_f64 vcc, %3, %4")
KERNEL64(k_mov_b64, "v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %4")
KERNEL64(k_max_f64, "v_max_f64 %0, %0, %4\n v_max_f64 %1, %1, %4\n v_max_f64 %2, %2, %4\n v_max_f64 %3, %3, %4")
KERNEL64(k_dep1_f64, "v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %0, %0, %4, %4")
KERNEL64(k_dep2_f64, "v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4")
KERNEL64(k_fma_salu, "v_fma_f64 %0, %0, %4, %4\n s_add_u32 s20, s20, 1\n v_fma_f64 %1, %1, %4, %4\n s_add_u32 s21, s21, 1")
KERNEL64(k_fma_mov32, "v_fma_f64 %0, %0, %4, %4\n v_mov_b32 v200, v201\n v_fma_f64 %1, %1, %4, %4\n v_mov_b32 v202, v203")
KERNEL64(k_dep_f64, "v_fma_f64 %0, %0, %4, %1\n v_add_f64 %0, %0, %2\n v_mul_f64 %0, %0, %4\n v_fma_f64 %0, %0, %4, %3")

template <class K> static void run(const char *name, K kern, unsigned *d, int waves)
{
    const int iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double ninstr = (double)iters * 64 * 4;
    printf("%-16s waves=%5d  %.3f ms  %.3f ns per wave-instruction\n", name, waves, ms, ms * 1e6 / ninstr);
}
int main()
{
    unsigned *d;
    hipMalloc(&d, 8192 * 64 * 4);
    hipMemset(d, 0, 8192 * 64 * 4);
    for (int waves : {1024, 2048, 4096})
    {
        run("v_add_u32", k_add_u32, d, waves);
        run("v_pk_add_u16", k_pk_add_u16, d, waves);
        run("v_pk_add opsel", k_pk_add_opsel, d, waves);
        run("v_pk_min_u16", k_pk_min_u16, d, waves);
        run("v_pk_sub_u16", k_pk_sub_u16, d, waves);
        run("v_lshl_or_b32", k_lshl_or, d, waves);
        run("v_min_u32", k_min_u32, d, waves);
        run("v_add_u16", k_add_u16, d, waves);
        run("v_min_u16", k_min_u16, d, waves);
        run("dep chain pk", k_dep_pk, d, waves);
        run("dep chain u32", k_dep_u32, d, waves);
        run("cmp+cndmask", k_cmp_cnd, d, waves);
        run("v_sad_u16", k_sad_u16, d, waves);
        run("v_perm_b32", k_perm, d, waves);
        run("v_fma_f64", k_fma_f64, d, waves);
        run("v_add_f64", k_add_f64, d, waves);
        run("v_mul_f64", k_mul_f64, d, waves);
        run("v_rcp_f64", k_rcp_f64, d, waves);
        run("v_sqrt_f64", k_sqrt_f64, d, waves);
        run("v_ldexp_f64", k_ldexp_f64, d, waves);
        run("v_div_fixup_f64", k_divfix_f64, d, waves);
        run("v_cmp_lt_f64", k_cmp_f64, d, waves);
        run("v_mov_b64", k_mov_b64, d, waves);
        run("v_max_f64", k_max_f64, d, waves);
        run("dep1 fma_f64", k_dep1_f64, d, waves);
        run("dep2 fma_f64", k_dep2_f64, d, waves);
        run("fma+salu", k_fma_salu, d, waves);
        run("fma+mov32", k_fma_mov32, d, waves);
        run("dep chain f64", k_dep_f64, d, waves);
    }
    return 0;
}

// TEST INFRASTRUCTURE ONLY: what the kernel headers of jaero_amd/csrc need from <hip/hip_runtime.h> to compile as plain host C++
// (tests/host_emul/*.cpp: device code driven thread by thread on the CPU).  Wavefront intrinsics are declared so that kernels which
// use them still compile; the emulations never call those kernels.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__
#define __shared__ static
struct emul_dim3 { int x, y, z; };
inline thread_local emul_dim3 blockIdx = {0, 0, 0}, threadIdx = {0, 0, 0}, blockDim = {64, 1, 1}, gridDim = {1, 1, 1};
struct int4 { int x, y, z, w; };
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
using std::max;
using std::min;
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline unsigned __brev(unsigned v)
{
    unsigned r = 0;
    for (int k = 0; k < 32; k++) r |= ((v >> k) & 1u) << (31 - k);
    return r;
}
static inline unsigned long long __ballot(int p) { return p ? 1ull : 0ull; } // one-lane "wavefront": compiles, never relied on
static inline int __any(int p) { return p; }
#ifndef __builtin_amdgcn_fence
#define __builtin_amdgcn_fence(order, scope) ((void)0) // (k_aerolb_deint: compiles, never run here)
#endif
template <class T> static inline T __shfl_xor(T v, int) { return v; } // (k_aerol_scan: compiles, never run here)
static inline void __syncthreads() {}
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)

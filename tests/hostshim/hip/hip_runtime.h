// Host stand-in for <hip/hip_runtime.h>: lets g++ compile the SERIAL routines of the device headers (avsim_math.hip.h,
// avsim_collide.hip.h) for the CPU, so that their arithmetic can be compared with the oracle without a GPU.
// Test infrastructure only (tests/test_host_collide.py).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define GLB_PTR(T) T*
#define AVS_LDS(T) T*
using std::fabs;
using std::sqrt;
// wave intrinsics used only by the 16-lane routines, which the host build never instantiates
template <typename T> inline T __shfl(T x, int, int) { return x; }
inline unsigned long long __ballot(bool) { return 0; }
inline int __popc(unsigned) { return 0; }
#define __builtin_amdgcn_fence(a, b) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)0)

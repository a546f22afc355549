// avsim_math.hip.h -- small vector helpers shared by the physics device code.
#pragma once
#include <hip/hip_runtime.h>

namespace avs {

#define AVS_DEV __device__ __forceinline__
// Out-of-line device functions.  Without -fgpu-rdc every device function is internalised, so the compiler knows every call site; a callee whose
// calls are not marked `tail` then saves no callee-saved VGPRs in its prologue (30-46 dwords per lane through the wave's private segment per call
// otherwise) and its callers spill what THEY keep across the call instead.  TailCallElim marks every call that passes no pointer to a caller's
// alloca; `not_tail_called` keeps a function out of that.  Which way is faster differs per function (profiles/r05_experiments.txt section 7): the
// mask selects the functions that get the attribute -- bit 0 multi_perturb, 1 noslip_trees / noslip_trees2, 2 pgs_groups, 3 qcqp_slide_octet,
// 4 ndense_chol, 5 newton_solve_coupled.
#ifndef AVS_NTC_MASK
#define AVS_NTC_MASK 6       // noslip_trees / noslip_trees2 and pgs_groups: + 1.5 % on every configuration, 3.09 -> 2.06 GB of L2 <-> fabric traffic per launch
#endif
#define AVS_OUTLINE __attribute__((noinline))
#define AVS_OUTLINE_NTC __attribute__((noinline, not_tail_called))
#if (AVS_NTC_MASK) & 1
#define AVS_OUTLINE_0 AVS_OUTLINE_NTC
#else
#define AVS_OUTLINE_0 AVS_OUTLINE
#endif
#if (AVS_NTC_MASK) & 2
#define AVS_OUTLINE_1 AVS_OUTLINE_NTC
#else
#define AVS_OUTLINE_1 AVS_OUTLINE
#endif
#if (AVS_NTC_MASK) & 4
#define AVS_OUTLINE_2 AVS_OUTLINE_NTC
#else
#define AVS_OUTLINE_2 AVS_OUTLINE
#endif
#if (AVS_NTC_MASK) & 8
#define AVS_OUTLINE_3 AVS_OUTLINE_NTC
#else
#define AVS_OUTLINE_3 AVS_OUTLINE
#endif
#if (AVS_NTC_MASK) & 16
#define AVS_OUTLINE_4 AVS_OUTLINE_NTC
#else
#define AVS_OUTLINE_4 AVS_OUTLINE
#endif
#if (AVS_NTC_MASK) & 32
#define AVS_OUTLINE_5 AVS_OUTLINE_NTC
#else
#define AVS_OUTLINE_5 AVS_OUTLINE
#endif
#ifndef GLB_PTR
#define GLB_PTR(T) __attribute__((address_space(1))) T*   // global memory, so that loads are global_load, not flat_load
#endif

template <typename T> AVS_DEV T dot3(const T* a, const T* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <typename T> AVS_DEV void cross3(const T* a, const T* b, T* c) {
    T t0 = a[1] * b[2] - a[2] * b[1], t1 = a[2] * b[0] - a[0] * b[2], t2 = a[0] * b[1] - a[1] * b[0];
    c[0] = t0; c[1] = t1; c[2] = t2;
}
template <typename T> AVS_DEV void sub3(const T* a, const T* b, T* c) { c[0] = a[0] - b[0]; c[1] = a[1] - b[1]; c[2] = a[2] - b[2]; }
template <typename T> AVS_DEV T normalize3(T* a) {
    T n = sqrt(dot3(a, a));
    if (n > T(0)) { T i = T(1) / n; a[0] *= i; a[1] *= i; a[2] *= i; }
    return n;
}
// o = R v, o = R^T v (R row-major 3x3; R may live in LDS or global)
template <typename T, typename M> AVS_DEV void mulmat(const M* R, const T* v, T* o) {
    T t0 = (T)R[0] * v[0] + (T)R[1] * v[1] + (T)R[2] * v[2], t1 = (T)R[3] * v[0] + (T)R[4] * v[1] + (T)R[5] * v[2],
      t2 = (T)R[6] * v[0] + (T)R[7] * v[1] + (T)R[8] * v[2];
    o[0] = t0; o[1] = t1; o[2] = t2;
}
template <typename T, typename M> AVS_DEV void mulmatT(const M* R, const T* v, T* o) {
    T t0 = (T)R[0] * v[0] + (T)R[3] * v[1] + (T)R[6] * v[2], t1 = (T)R[1] * v[0] + (T)R[4] * v[1] + (T)R[7] * v[2],
      t2 = (T)R[2] * v[0] + (T)R[5] * v[1] + (T)R[8] * v[2];
    o[0] = t0; o[1] = t1; o[2] = t2;
}
template <typename T> AVS_DEV void quat2mat(const T* q, T* R) {
    T w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}
template <typename T> AVS_DEV void quatmul(const T* a, const T* b, T* c) {
    T t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
      t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    c[0] = t0; c[1] = t1; c[2] = t2; c[3] = t3;
}
template <typename T> AVS_DEV void quatnorm(T* q) {
    T n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n < T(1e-15)) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
    T i = T(1) / n;
    q[0] *= i; q[1] *= i; q[2] *= i; q[3] *= i;
}
template <typename T> AVS_DEV T tmax(T a, T b) { return a > b ? a : b; }
template <typename T> AVS_DEV T tmin(T a, T b) { return a < b ? a : b; }
template <typename T> AVS_DEV T tclamp(T x, T lo, T hi) { return x < lo ? lo : (x > hi ? hi : x); }

}  // namespace avs

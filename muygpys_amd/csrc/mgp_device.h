// Shared device helpers for the gfx950 kernels of the MuyGPyS local-GP hot path.
// Wave size is 64 on CDNA4; every cross-lane idiom below is written for that.
#pragma once

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/muygpys_hip.h"
#else
// run-time compile (hiprtc): no libc headers; the few names of the C header the kernels use
typedef long long int64_t;
typedef unsigned long long uint64_t;
typedef unsigned long uintptr_t;
enum { MGP_KERNEL_RBF = 0, MGP_KERNEL_MATERN_05 = 1, MGP_KERNEL_MATERN_15 = 2, MGP_KERNEL_MATERN_25 = 3, MGP_KERNEL_MATERN_INF = 4, MGP_KERNEL_MATERN_GEN = 5 };
enum { MGP_METRIC_L2 = 0, MGP_METRIC_F2 = 1 };
enum { MGP_NOISE_SCALAR = 0, MGP_NOISE_TABLE = 1, MGP_NOISE_BATCH = 2 };
#endif

#define MGP_WAVE 64

#define MGP_HIP_CHECK_LAUNCH()                              \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return -(1000 + (int)e__);       \
  } while (0)

namespace mgp {

template <typename T> struct num;
template <> struct num<float> {
  static __device__ __forceinline__ float exp(float x) { return expf(x); }
  static __device__ __forceinline__ float sqrt(float x) { return sqrtf(x); }
  static __device__ __forceinline__ float rsqrt(float x) { return 1.0f / sqrtf(x); }
  static __device__ __forceinline__ float nan() { return __builtin_nanf(""); }
  static __device__ __forceinline__ float inf() { return __builtin_inff(); }
};
template <> struct num<double> {
  static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
  static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
  static __device__ __forceinline__ double rsqrt(double x) { return 1.0 / ::sqrt(x); }
  static __device__ __forceinline__ double nan() { return __builtin_nan(""); }
  static __device__ __forceinline__ double inf() { return __builtin_inf(); }
};

// acc = sum_d (diff)^2 [anisotropic: sum_d (diff / l_d)^2].  Returns the kernel's
// argument: Isotropy l2: sqrt(acc)/l; F2: acc/l^2 (gp/deformation/metric.py:241,264);
// Anisotropy: sqrt(acc) / acc (anisotropy.py:70).  post_scale = 1/l or 1/l^2 or 1.
template <typename T>
__device__ __forceinline__ T metric_arg(T acc, int metric_id, T post_scale) {
  return (metric_id == MGP_METRIC_L2 ? num<T>::sqrt(acc) : acc) * post_scale;
}

// _src/gp/kernels/numpy.py:12-31
template <typename T>
__device__ __forceinline__ T kernel_eval(int kernel_id, T x) {
  switch (kernel_id) {
    case MGP_KERNEL_RBF:
      return num<T>::exp(-x * T(0.5));
    case MGP_KERNEL_MATERN_05:
      return num<T>::exp(-x);
    case MGP_KERNEL_MATERN_15: {
      T t = x * T(1.7320508075688772935);
      return (T(1) + t) * num<T>::exp(-t);
    }
    case MGP_KERNEL_MATERN_25: {
      T t = x * T(2.2360679774997896964);
      return (T(1) + t + t * t * T(1.0 / 3.0)) * num<T>::exp(-t);
    }
    default:  // MGP_KERNEL_MATERN_INF
      return num<T>::exp(-x * x * T(0.5));
  }
}

// d kernel_eval / d x (the kernel's own argument), for the backward pass.
template <typename T>
__device__ __forceinline__ T kernel_deriv(int kernel_id, T x) {
  switch (kernel_id) {
    case MGP_KERNEL_RBF:
      return T(-0.5) * num<T>::exp(-x * T(0.5));
    case MGP_KERNEL_MATERN_05:
      return -num<T>::exp(-x);
    case MGP_KERNEL_MATERN_15:
      return T(-3) * x * num<T>::exp(-x * T(1.7320508075688772935));
    case MGP_KERNEL_MATERN_25: {
      T t = x * T(2.2360679774997896964);
      return T(-5.0 / 3.0) * x * (T(1) + t) * num<T>::exp(-t);
    }
    default:  // MGP_KERNEL_MATERN_INF
      return -x * num<T>::exp(-x * x * T(0.5));
  }
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Sum over the 64 lanes, valid in LANE 63 only, on the VALU's data-parallel-primitive path (no LDS
// crossbar round trips like __shfl_xor = ds_bpermute): neighbours, pairs of neighbours, then the four
// quads of a 16-lane row by rotation, then the row totals handed down rows 0 -> 1, 2 -> 3 and 1 -> 2, 3.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_term(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_lane63(float v) {
  v += dpp_term<0xb1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_term<0x4e, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_term<0x124, 0xf>(v);  // row_ror:4
  v += dpp_term<0x128, 0xf>(v);  // row_ror:8   -> every lane of a row holds the row's total
  v += dpp_term<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_term<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ double wave_sum_lane63(double v) { return wave_sum(v); }

__host__ __device__ inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace mgp

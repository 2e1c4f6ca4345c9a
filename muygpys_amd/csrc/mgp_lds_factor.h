// LDS-resident augmented Cholesky used by the generic (any nn_count) kernels.
//
// The local system of one neighbourhood is held as the lower triangle of
//
//        [ K + diag(eps)                ]   rows 0 .. k-1      (neighbours)
//   S =  [ c^T            (kout)        ]   row  k             (query / Kcross)
//        [ Y^T                          ]   rows k+1 .. k+R    (responses)
//
// Running k columns of a left-looking Cholesky over ALL rows leaves
//   S[i][j] = L[i][j]               (i < k)
//   S[k][j] = z[j],  z = L^-1 c     (forward substitution comes for free)
//   S[k+1+r][j] = (L^-1 y_r)[j]
// so that  var = kout - |z|^2,  mean_r = z . (L^-1 y_r),  y_r^T K^-1 y_r = |L^-1 y_r|^2
// i.e. S1/S2/S3 of SURVEY.md sec. 8a from ONE factorisation and no back-substitution
// (the reference does three LU solves: _src/gp/muygps/numpy.py:37,63 and
// _src/optimize/scale/numpy.py:14).
#pragma once

#include "mgp_device.h"

namespace mgp {

// Odd row stride (in elements) -> lane i reading S[i][m] hits distinct banks.
__host__ __device__ inline int lds_row_stride(int k) { return (k + 1) | 1; }

// S: rows x SP in LDS.  piv: k entries of scratch.  All NT threads of the block call this.
// Returns (to every thread) whether a non-positive / NaN pivot was met.
template <typename T>
__device__ inline bool factor_augmented_lds(T* S, int SP, int k, int rows, T* piv, int* bad_flag, int tid, int NT) {
  if (tid == 0) *bad_flag = 0;
  __syncthreads();
  for (int j = 0; j < k; ++j) {
    const T* rowj = S + j * SP;
    for (int i = j + tid; i < rows; i += NT) {
      T* rowi = S + i * SP;
      T s = rowi[j];
#pragma unroll 4
      for (int m = 0; m < j; ++m) s -= rowi[m] * rowj[m];
      if (i == j) {
        if (!(s > T(0))) *bad_flag = 1;
        piv[j] = num<T>::rsqrt(s);
        rowi[j] = num<T>::sqrt(s);
      } else {
        rowi[j] = s;
      }
    }
    __syncthreads();
    const T inv = piv[j];
    for (int i = j + 1 + tid; i < rows; i += NT) S[i * SP + j] *= inv;
    __syncthreads();
  }
  return *bad_flag != 0;
}

// mean (R), var, ykinvy (R) from the factored S; one output per thread, strided.
template <typename T>
__device__ inline void emit_outputs_lds(const T* S, int SP, int k, int R, T kout, bool bad, bool has_cross,
                                        T* mean, T* var, T* ykinvy, int tid, int NT) {
  const T* z = S + k * SP;
  for (int o = tid; o < 1 + 2 * R; o += NT) {
    if (o == 0) {
      if (var != nullptr && has_cross) {
        T s = T(0);
        for (int m = 0; m < k; ++m) s += z[m] * z[m];
        *var = bad ? num<T>::nan() : kout - s;
      }
    } else if (o <= R) {
      if (mean != nullptr && has_cross) {
        const T* zy = S + (k + o) * SP;
        T s = T(0);
        for (int m = 0; m < k; ++m) s += z[m] * zy[m];
        mean[o - 1] = bad ? num<T>::nan() : s;
      }
    } else if (ykinvy != nullptr) {
      const int r = o - R - 1;
      const T* zy = S + (k + 1 + r) * SP;
      T s = T(0);
      for (int m = 0; m < k; ++m) s += zy[m] * zy[m];
      ykinvy[r] = bad ? num<T>::nan() : s;
    }
  }
}

}  // namespace mgp

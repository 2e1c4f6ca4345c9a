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

// Row stride (in elements) of the LDS-resident system: rows start on 16-byte boundaries and span an
// ODD number of 16-byte slots, so that thread i reading 16 bytes of row i at a common column hits
// distinct banks (the same rule as the feature tiles of the register kernels).
template <typename T>
__host__ __device__ inline int lds_row_stride(int k) {
  constexpr int E = 16 / (int)sizeof(T);
  int slots = (k + 1 + E - 1) / E;
  slots |= 1;
  return slots * E;
}

template <typename T> struct lds_vec;
template <> struct lds_vec<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct lds_vec<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <typename V> __device__ __forceinline__ auto vec_dot(const V& a, const V& b) {
  auto s = a[0] * b[0];
#pragma unroll
  for (int e = 1; e < (int)(sizeof(V) / sizeof(a[0])); ++e) s += a[e] * b[e];
  return s;
}

// Gather the (k+1) x w feature tile of one neighbourhood (rows idx[0..k-1] of feat_nn, row idx[k] of
// feat_q, features d0 .. d0+w-1) into X (row stride XP, 16-byte aligned, zero-padded to wp columns).
// Every thread first issues up to GU 16-byte loads and only then stores them: the loads of a tile are
// in flight together instead of one memory latency per element.  vec_ok: rows are 16-byte aligned
// (d a multiple of the vector width, bases aligned); otherwise the pieces are assembled element-wise.
template <typename T>
__device__ inline void gather_tile_lds(T* X, int XP, const T* feat_q, const T* feat_nn, const int64_t* idx, int k,
                                       int d, int d0, int w, int wp, bool vec_ok, int tid, int NT) {
  using V = typename lds_vec<T>::type;
  constexpr int E = 16 / (int)sizeof(T);
  constexpr int GU = 4;
  const int pieces = wp / E, total = (k + 1) * pieces;
  for (int t0 = tid; t0 < total; t0 += NT * GU) {
    V v[GU];
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int t = t0 + u * NT;
      v[u] = V(0);
      if (t < total) {
        const int r = t / pieces, c0 = (t - r * pieces) * E;
        const T* src = (r < k ? feat_nn : feat_q) + idx[r] * (int64_t)d + d0 + c0;
        if (vec_ok && c0 + E <= w) {
          v[u] = *reinterpret_cast<const V*>(src);
        } else {
#pragma unroll
          for (int e = 0; e < E; ++e)
            if (c0 + e < w) v[u][e] = src[e];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int t = t0 + u * NT;
      if (t < total) {
        const int r = t / pieces, c0 = (t - r * pieces) * E;
        *reinterpret_cast<V*>(X + r * XP + c0) = v[u];
      }
    }
  }
}

// S: rows x SP in LDS (SP = lds_row_stride<T>(k), S 16-byte aligned).  piv: k entries of scratch.
// All NT threads of the block call this.  Returns (to every thread) whether a non-positive / NaN
// pivot was met.
//
// Left-looking by rows, blocked by JB = 4 columns: the bulk of a column's inner product -- the part
// over the columns of earlier blocks -- is taken for four columns at once with 16-byte LDS reads
// (one read of the thread's own row serves four columns; the four pivot rows are broadcasts), i.e.
// 5 LDS instructions per 16 (fp32) multiply-adds instead of 32; the diagonal block is then
// factorised redundantly by every thread (see below).
template <typename T>
__device__ inline bool factor_augmented_lds(T* S, int SP, int k, int rows, T* piv, int* bad_flag, int tid, int NT) {
  using V = typename lds_vec<T>::type;
  constexpr int E = 16 / (int)sizeof(T);
  constexpr int JB = 4;
  if (tid == 0) *bad_flag = 0;
  __syncthreads();
  for (int j0 = 0; j0 < k; j0 += JB) {
    const int jb = min(JB, k - j0);
    if (j0 > 0) {
      for (int i = j0 + tid; i < rows; i += NT) {
        T* rowi = S + i * SP;
        const T* r0 = S + (j0 + 0) * SP;
        const T* r1 = S + (j0 + (jb > 1 ? 1 : 0)) * SP;
        const T* r2 = S + (j0 + (jb > 2 ? 2 : 0)) * SP;
        const T* r3 = S + (j0 + (jb > 3 ? 3 : 0)) * SP;
        // vector accumulators (packed FMAs, one horizontal sum at the end) and two 16-byte steps per
        // iteration with all ten reads issued before the first FMA: the one-step scalar form waited a
        // full LDS round trip for 20 plain VALU instructions
        V a0 = V(0), a1 = V(0), a2 = V(0), a3 = V(0);
        int m = 0;
        for (; m + 2 * E <= j0; m += 2 * E) {  // j0 is a multiple of JB, hence of E
          const V li = *reinterpret_cast<const V*>(rowi + m), lj = *reinterpret_cast<const V*>(rowi + m + E);
          const V p0 = *reinterpret_cast<const V*>(r0 + m), q0 = *reinterpret_cast<const V*>(r0 + m + E);
          const V p1 = *reinterpret_cast<const V*>(r1 + m), q1 = *reinterpret_cast<const V*>(r1 + m + E);
          const V p2 = *reinterpret_cast<const V*>(r2 + m), q2 = *reinterpret_cast<const V*>(r2 + m + E);
          const V p3 = *reinterpret_cast<const V*>(r3 + m), q3 = *reinterpret_cast<const V*>(r3 + m + E);
          a0 = li * p0 + a0;
          a1 = li * p1 + a1;
          a2 = li * p2 + a2;
          a3 = li * p3 + a3;
          a0 = lj * q0 + a0;
          a1 = lj * q1 + a1;
          a2 = lj * q2 + a2;
          a3 = lj * q3 + a3;
        }
        if (m < j0) {
          const V li = *reinterpret_cast<const V*>(rowi + m);
          a0 = li * *reinterpret_cast<const V*>(r0 + m) + a0;
          a1 = li * *reinterpret_cast<const V*>(r1 + m) + a1;
          a2 = li * *reinterpret_cast<const V*>(r2 + m) + a2;
          a3 = li * *reinterpret_cast<const V*>(r3 + m) + a3;
        }
        T s0 = a0[0], s1 = a1[0], s2 = a2[0], s3 = a3[0];
#pragma unroll
        for (int e = 1; e < E; ++e) {
          s0 += a0[e];
          s1 += a1[e];
          s2 += a2[e];
          s3 += a3[e];
        }
        rowi[j0] -= s0;
        if (jb > 1) rowi[j0 + 1] -= s1;
        if (jb > 2) rowi[j0 + 2] -= s2;
        if (jb > 3) rowi[j0 + 3] -= s3;
      }
      __syncthreads();
    }
    // The jb x jb diagonal block (already reduced by the bulk phase) is factorised by EVERY thread,
    // redundantly, in registers -- 20 flops instead of two barriers per column -- and each thread
    // then forward-substitutes its own row's jb panel entries: three barriers per block of four
    // columns instead of eight.
    T D[JB][JB];
#pragma unroll
    for (int r = 0; r < JB; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) D[r][c] = (r < jb) ? S[(j0 + r) * SP + j0 + c] : (r == c ? T(1) : T(0));
    bool bad = false;
    T inv[JB];
#pragma unroll
    for (int c = 0; c < JB; ++c) {
      T p = D[c][c];
#pragma unroll
      for (int m = 0; m < c; ++m) p -= D[c][m] * D[c][m];
      bad = bad || !(p > T(0));
      inv[c] = num<T>::rsqrt(p);
      D[c][c] = num<T>::sqrt(p);
#pragma unroll
      for (int r = c + 1; r < JB; ++r) {
        T v = D[r][c];
#pragma unroll
        for (int m = 0; m < c; ++m) v -= D[r][m] * D[c][m];
        D[r][c] = v * inv[c];
      }
    }
    __syncthreads();  // everyone holds the block: its rows may now be overwritten
    if (tid == 0) {
      if (bad) *bad_flag = 1;
      for (int c = 0; c < jb; ++c) piv[j0 + c] = inv[c];
    }
    for (int i = j0 + tid; i < rows; i += NT) {
      T* rowi = S + i * SP;
      if (i < j0 + jb) {
        const int r = i - j0;
#pragma unroll
        for (int c = 0; c < JB; ++c)
          if (c <= r) {
            T v = T(0);
#pragma unroll
            for (int rr = 0; rr < JB; ++rr) v = rr == r ? D[rr][c] : v;  // row r of the factor, no dynamic indexing
            rowi[j0 + c] = v;
          }
      } else {
        T y[JB];
#pragma unroll
        for (int c = 0; c < JB; ++c) {
          T v = c < jb ? rowi[j0 + c] : T(0);
#pragma unroll
          for (int m = 0; m < c; ++m) v -= y[m] * D[c][m];
          y[c] = v * inv[c];
        }
#pragma unroll
        for (int c = 0; c < JB; ++c)
          if (c < jb) rowi[j0 + c] = y[c];
      }
    }
    __syncthreads();
  }
  return *bad_flag != 0;
}

// mean (R), var, ykinvy (R) from the factored S: 1 + 2 R inner products of length k, each taken by
// the whole first wave (lanes stride over the k entries, one cross-lane sum) -- one thread per output
// walked its k entries alone, a serial chain of k LDS round trips at the end of every neighbourhood.
template <typename T>
__device__ inline void emit_outputs_lds(const T* S, int SP, int k, int R, T kout, bool bad, bool has_cross,
                                        T* mean, T* var, T* ykinvy, int tid, int NT) {
  if (tid >= MGP_WAVE) return;
  const T* z = S + k * SP;
  for (int o = 0; o < 1 + 2 * R; ++o) {
    const T* x = z;
    const T* y = z;
    T* dst = nullptr;
    if (o == 0) {
      if (var != nullptr && has_cross) dst = var;
    } else if (o <= R) {
      y = S + (k + o) * SP;
      if (mean != nullptr && has_cross) dst = mean + (o - 1);
    } else {
      x = y = S + (k + 1 + (o - R - 1)) * SP;
      if (ykinvy != nullptr) dst = ykinvy + (o - R - 1);
    }
    if (dst == nullptr) continue;  // uniform
    T s = T(0);
    for (int m = tid; m < k; m += MGP_WAVE) s += x[m] * y[m];
    s = wave_sum(s);
    if (tid == 0) *dst = bad ? num<T>::nan() : (o == 0 ? kout - s : s);
  }
}

}  // namespace mgp

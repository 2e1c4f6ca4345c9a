// The two selection steps around the exact k-NN scan (csrc/mgp_knn.hip), off torch's topk / sort (round 6):
//
//   topk_rows_kernel    the k smallest entries of every row of a (rows, cols) fp32 matrix, unordered -- the initial k-best
//                       lists of the scan, taken from the Gram-form distances to the first rows of the table.  torch.topk
//                       on a million rows of ~2 000 columns runs its multi-block radix select (four passes of four
//                       kernels each): 52 of the 330 ms of a 1 M x 1 M, d = 40 search (profiles/r05_knn_d40_kernel_stats.csv).
//   knn_finish_kernel   the winners of the scan re-measured exactly (difference form, fp32) and put in order
//                       (distance, then position in the list: a stable sort), indices mapped back to the caller's row
//                       numbers: was a gather of (m, k, d) rows, a subtract, a square, a sum, an argsort and two more
//                       gathers through (m, k, d)- and (m, k)-sized temporaries.
//
// Reference: scikit-learn's brute-force `kneighbors` behind `NN_Wrapper.get_nns` / `get_batch_nns`
// (src/MuyGPyS/neighbors.py:129-211): exact neighbours, ascending distance.
#include <cstdint>

#include "mgp_args.h"

namespace mgp {

// order-preserving map of a float to an unsigned integer (NaN sorts last)
__device__ __forceinline__ unsigned float_key(float x) {
  const unsigned u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ int wave_sum_int(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// One wave per row.  The row (cols <= 64 * VPL) sits in registers as keys; the k-th smallest key is found by bisection on
// its 32 bits (count of keys below the trial value: a per-lane count and one wave sum per bit), then everything below
// it is emitted, and as many entries equal to it as are missing (lowest columns first).  Output order: by lane, then
// by position -- the scan's lists are unordered.
template <int VPL>
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* x, int64_t rows, int cols, int64_t stride, int k, float* out_v,
                                                        int* out_i) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * stride;
  unsigned key[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int c = v * 64 + lane;
    key[v] = c < cols ? float_key(xr[c]) : 0xFFFFFFFFu;
  }
  // the largest T with count(key < T) < k, i.e. T = the k-th smallest key (bit by bit from the top)
  unsigned T = 0;
#pragma unroll 1
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned trial = T | (1u << bit);
    int cnt = 0;
#pragma unroll
    for (int v = 0; v < VPL; ++v) cnt += key[v] < trial ? 1 : 0;
    if (wave_sum_int(cnt) < k) T = trial;  // fewer than k keys below the trial value: the k-th is at or above it
  }
  // ranks: strictly-below entries first, then the ties in column order
  int below = 0, equal = 0;
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    below += key[v] < T ? 1 : 0;
    equal += key[v] == T ? 1 : 0;
  }
  // exclusive prefix sums over the lanes (column order within a lane is v-major: ties are taken lowest lane first, which
  // is not lowest column first -- any k of the tied entries are a valid answer)
  int pb = below, pe = equal;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int tb = __shfl_up(pb, off, 64), te = __shfl_up(pe, off, 64);
    if (lane >= off) {
      pb += tb;
      pe += te;
    }
  }
  const int total_below = __shfl(pb, 63, 64);
  int wb = pb - below, we = total_below + (pe - equal);
  float* ov = out_v + row * (int64_t)k;
  int* oi = out_i + row * (int64_t)k;
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int c = v * 64 + lane;
    if (c >= cols) continue;  // (padding keys: never emitted, whatever the threshold)
    if (key[v] < T) {
      ov[wb] = xr[c];
      oi[wb] = c;
      ++wb;
    } else if (key[v] == T) {
      if (we < k) {
        ov[we] = xr[c];
        oi[we] = c;
      }
      ++we;
    }
  }
}

// L = lanes per query (the power of two >= k); lane j of a query's group re-measures candidate j in the difference
// form and ranks it among the group's (distance, position).
template <int L>
__global__ __launch_bounds__(256) void knn_finish_kernel(const float* queries, const float* train, int d, const int* cand, int64_t m,
                                                         int k, const int64_t* perm, int64_t* out_idx, float* out_dist) {
  constexpr int QPW = 64 / L;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int g = lane / L, j = lane % L;
  const int64_t q = wave * QPW + g;
  const bool on = q < m && j < k;
  int ci = 0;
  float dd = __builtin_inff();
  if (on) {
    ci = cand[q * k + j];
    const float* qr = queries + q * (int64_t)d;
    const float* xr = train + ci * (int64_t)d;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int c = 0; c < d; c += 4) {
      const float4 a = *reinterpret_cast<const float4*>(qr + c), b = *reinterpret_cast<const float4*>(xr + c);
      const float e0 = a.x - b.x, e1 = a.y - b.y, e2 = a.z - b.z, e3 = a.w - b.w;
      s0 = __builtin_fmaf(e0, e0, s0);
      s1 = __builtin_fmaf(e1, e1, s1);
      s2 = __builtin_fmaf(e2, e2, s2);
      s3 = __builtin_fmaf(e3, e3, s3);
    }
    dd = (s0 + s1) + (s2 + s3);
  }
  // rank = entries of the group that come before this one (smaller distance, or equal distance at an earlier position)
  int rank = 0;
#pragma unroll 8
  for (int t = 0; t < L; ++t) {
    const float o = __shfl(dd, g * L + t, 64);
    rank += (o < dd || (o == dd && t < j)) ? 1 : 0;
  }
  if (on) {
    out_idx[q * k + rank] = perm ? perm[ci] : (int64_t)ci;
    out_dist[q * k + rank] = dd;
  }
}

int launch_topk_rows(const float* x, int64_t rows, int cols, int64_t stride, int k, float* out_v, int* out_i, hipStream_t s) {
  if (rows == 0) return MGP_OK;
  if (k < 1 || k > cols || cols > 64 * 64) return MGP_EUNSUPPORTED;
  const unsigned grid = (unsigned)((rows + 3) / 4);
  const int vpl = (cols + 63) / 64;
#define MGP_TOPK(V)                                                                                              \
  if (vpl <= V) {                                                                                                \
    hipLaunchKernelGGL(topk_rows_kernel<V>, dim3(grid), dim3(256), 0, s, x, rows, cols, stride, k, out_v, out_i); \
    MGP_HIP_CHECK_LAUNCH();                                                                                      \
    return MGP_OK;                                                                                               \
  }
  MGP_TOPK(8)
  MGP_TOPK(16)
  MGP_TOPK(32)
  MGP_TOPK(64)
#undef MGP_TOPK
  return MGP_EUNSUPPORTED;
}

int launch_knn_finish(const float* queries, const float* train, int d, const int* cand, int64_t m, int k, const int64_t* perm,
                      int64_t* out_idx, float* out_dist, hipStream_t s) {
  if (m == 0) return MGP_OK;
  if (k < 1 || k > 64 || d % 4 != 0) return MGP_EUNSUPPORTED;
  const uintptr_t al = (uintptr_t)queries | (uintptr_t)train;
  if (al % 16 != 0) return MGP_EUNSUPPORTED;
  const int L = k <= 16 ? 16 : (k <= 32 ? 32 : 64);
  const int64_t waves = (m + 64 / L - 1) / (64 / L);
  const unsigned grid = (unsigned)((waves + 3) / 4);
  if (L == 16) hipLaunchKernelGGL(knn_finish_kernel<16>, dim3(grid), dim3(256), 0, s, queries, train, d, cand, m, k, perm, out_idx, out_dist);
  else if (L == 32) hipLaunchKernelGGL(knn_finish_kernel<32>, dim3(grid), dim3(256), 0, s, queries, train, d, cand, m, k, perm, out_idx, out_dist);
  else hipLaunchKernelGGL(knn_finish_kernel<64>, dim3(grid), dim3(256), 0, s, queries, train, d, cand, m, k, perm, out_idx, out_dist);
  MGP_HIP_CHECK_LAUNCH();
  return MGP_OK;
}

}  // namespace mgp

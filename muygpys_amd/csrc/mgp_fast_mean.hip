// Fused fast posterior mean: prediction from precomputed coefficients.
//
// Reference workflow (src/MuyGPyS/examples/fast_posterior_mean.py:317-400,
// _src/gp/muygps/numpy.py:70-95, _src/gp/tensors/numpy.py:18-37,97-108): for every training
// point i the coefficients C_i = (K_i + eps)^-1 y_i of its self-including neighbourhood are
// computed once; a test point t then takes its closest training point c(t), that point's
// neighbourhood N = nn_fast[c(t)] and predicts
//
//      mean_t = sum_j kappa(dist(x_t, x_{N_j})) * C[c(t)][j]        (einsum 'ij,ijk->ik').
//
// The reference materialises crosswise differences (b,k,d), distances, Kcross (b,k) and the
// gathered coefficients; here one launch reads (k+1) feature rows + k*R coefficients per test
// point and writes R values: a pure gather kernel, HBM-bound by construction
// (algorithmic bytes per test point: (k+1) d s + k R s + 8 (k+2) + R s).
//
// One wave owns 64/NP test points at a time (NP = 32 or 64 slots >= k+1): rows are staged in
// LDS with the same row-walking 16-byte gather as the fused kernel, lane i then owns neighbour
// i: difference-form distance to the query row, kernel, product with its coefficient(s), and a
// cross-lane sum.
#include "mgp_args.h"

namespace mgp {

template <typename T> struct fv16;
template <> struct fv16<float> { typedef float type __attribute__((ext_vector_type(4))); static constexpr int N = 4; };
template <> struct fv16<double> { typedef double type __attribute__((ext_vector_type(2))); static constexpr int N = 2; };

struct FastArgs {
  const void* feat_q;
  const void* feat_nn;
  const int64_t* batch_idx;   // (b) rows of feat_q, NULL = identity
  const int64_t* nn_idx;      // (b, k) rows of feat_nn
  const void* coeffs;         // (n_train, k, R)
  const int64_t* coeff_row;   // (b) row of coeffs per test point (the closest training point)
  const void* length_scale;
  void* mean;                 // (b, R)
  int64_t b;
  int d, k, R, kernel_id, metric_id, ls_count;
  int dst, xs, vec_ok;
};

template <typename T, int NP>
__global__ __launch_bounds__(64) void fast_mean_kernel(FastArgs a) {
  constexpr int NH = 64 / NP;
  constexpr int E = fv16<T>::N;
  constexpr int CH = 2 * E;
  using V = typename fv16<T>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int k = a.k, d = a.d, R = a.R, xs = a.xs, dst = a.dst;
  T* tile = reinterpret_cast<T*>(smem);                       // NH*NP rows x xs
  T* ilbuf = tile + NH * NP * xs;                             // dst
  int64_t* idxbuf = reinterpret_cast<int64_t*>(ilbuf + dst + (dst & 1));  // 64

  const T* feat_q = static_cast<const T*>(a.feat_q);
  const T* feat_nn = static_cast<const T*>(a.feat_nn);
  const T* coeffs = static_cast<const T*>(a.coeffs);
  const T* ls = static_cast<const T*>(a.length_scale);
  T* mean = static_cast<T*>(a.mean);
  const bool aniso = a.ls_count > 1;
  T post_scale = T(1);
  if (!aniso) {
    const T l = ls[0];
    post_scale = a.metric_id == MGP_METRIC_L2 ? T(1) / l : T(1) / (l * l);
  }

  const int lane = threadIdx.x;
  const int h = NH == 1 ? 0 : lane / NP;
  const int i = lane & (NP - 1);
  T* Xh = tile + h * NP * xs;
  int64_t* idxh = idxbuf + h * NP;
  const int64_t ntasks = (a.b + NH - 1) / NH;

  for (int64_t task = blockIdx.x; task < ntasks; task += gridDim.x) {
    const int64_t nb_raw = task * NH + h;
    const bool live = nb_raw < a.b;
    const int64_t nb = live ? nb_raw : a.b - 1;
    int64_t myidx = 0;
    if (i < k) myidx = a.nn_idx[nb * k + i];
    else if (i == k) myidx = a.batch_idx ? a.batch_idx[nb] : nb;
    const int64_t crow = a.coeff_row[nb];
    __syncthreads();
    idxh[i] = myidx * (int64_t)d;
    T acc = T(0);
    for (int d0 = 0; d0 < d; d0 += dst) {
      const int w = min(dst, d - d0);
      const int wp = (w + CH - 1) / CH * CH;
      __syncthreads();
      if (a.vec_ok) {
        const int c16 = w / E, c16p = wp / E;
        const int rpr = NP / c16p;
        const int sub = (int)(((unsigned)i * ((1u << 16) / (unsigned)c16p + 1u)) >> 16);
        const int c = i - sub * c16p;
        const bool lane_on = sub < rpr;
        constexpr int U = 6;
        for (int r0 = 0; r0 <= k; r0 += U * rpr) {
          V v[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int row = r0 + u * rpr + sub;
            v[u] = V(0);
            if (lane_on && row <= k && c < c16)
              v[u] = *reinterpret_cast<const V*>((row < k ? feat_nn : feat_q) + idxh[row] + d0 + c * E);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int row = r0 + u * rpr + sub;
            if (lane_on && row <= k) *reinterpret_cast<V*>(Xh + row * xs + c * E) = v[u];
          }
        }
      } else {
        const unsigned magic = (1u << 20) / (unsigned)wp + 1u;
        for (int t = i; t < (k + 1) * wp; t += NP) {
          const int row = (int)(((unsigned)t * magic) >> 20);
          const int c = t - row * wp;
          Xh[row * xs + c] = c < w ? ((row < k ? feat_nn : feat_q) + idxh[row] + d0)[c] : T(0);
        }
      }
      if (aniso)
        for (int c = lane; c < wp; c += 64) ilbuf[c] = c < w ? T(1) / ls[d0 + c] : T(0);
      __syncthreads();
      if (i < k) {
        const T* xo = Xh + i * xs;
        const T* xq = Xh + k * xs;
        for (int c0 = 0; c0 < wp; c0 += E) {
          V df = *reinterpret_cast<const V*>(xq + c0) - *reinterpret_cast<const V*>(xo + c0);
          if (aniso) df = df * *reinterpret_cast<const V*>(ilbuf + c0);
#pragma unroll
          for (int e = 0; e < E; ++e) acc += df[e] * df[e];
        }
      }
    }
    T kv = T(0);
    if (i < k) kv = kernel_eval<T>(a.kernel_id, metric_arg<T>(acc, a.metric_id, post_scale));
    for (int r = 0; r < R; ++r) {
      T part = i < k ? kv * coeffs[(crow * k + i) * (int64_t)R + r] : T(0);
#pragma unroll
      for (int off = NP / 2; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
      if (live && i == 0) mean[nb * R + r] = part;
    }
  }
}

template <typename T, int NP>
static int launch_fast_np(FastArgs a, hipStream_t stream) {
  constexpr int NH = 64 / NP;
  constexpr int E = fv16<T>::N;
  constexpr int CH = 2 * E;
  const int dpad = (a.d + CH - 1) / CH * CH;
  a.dst = dpad < 64 ? dpad : 64;
  a.xs = a.dst + E;
  const uintptr_t align = (uintptr_t)a.feat_q | (uintptr_t)a.feat_nn;
  a.vec_ok = (a.d % E == 0) && (align % 16 == 0);
  size_t lds = ((size_t)NH * NP * a.xs + a.dst + (a.dst & 1)) * sizeof(T) + 64 * sizeof(int64_t);
  lds = (lds + 15) & ~(size_t)15;
  const int64_t ntasks = (a.b + NH - 1) / NH;
  int64_t grid = 256LL * 16;  // memory-bound: grid-stride over plenty of resident waves
  if (grid > ntasks) grid = ntasks;
  hipLaunchKernelGGL((fast_mean_kernel<T, NP>), dim3((unsigned)grid), dim3(64), lds, stream, a);
  MGP_HIP_CHECK_LAUNCH();
  return MGP_OK;
}

template <typename T>
int launch_fast_mean(const void* fq, const void* fn, int d, const int64_t* bi, const int64_t* ni, int64_t b, int k,
                     const void* coeffs, const int64_t* crow, int R, int kernel_id, int metric_id, const void* ls,
                     int ls_count, void* mean, hipStream_t stream) {
  FastArgs a{fq, fn, bi, ni, coeffs, crow, ls, mean, b, d, k, R, kernel_id, metric_id, ls_count, 0, 0, 0};
  if (k + 1 <= 32) return launch_fast_np<T, 32>(a, stream);
  if (k + 1 <= 64) return launch_fast_np<T, 64>(a, stream);
  return MGP_EUNSUPPORTED;
}

template int launch_fast_mean<float>(const void*, const void*, int, const int64_t*, const int64_t*, int64_t, int,
                                     const void*, const int64_t*, int, int, int, const void*, int, void*, hipStream_t);
template int launch_fast_mean<double>(const void*, const void*, int, const int64_t*, const int64_t*, int64_t, int,
                                      const void*, const int64_t*, int, int, int, const void*, int, void*, hipStream_t);

}  // namespace mgp

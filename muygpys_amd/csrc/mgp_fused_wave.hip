// Register-resident wave-per-neighbourhood fused kernel (the roofline path).
//
// One wavefront (64 lanes, one workgroup) owns NH = 64/NP neighbourhoods at a time; a
// neighbourhood is a set of NP "slots", one per lane:
//
//     slot 0 .. k-1      neighbour rows            (features gathered by nn_idx)
//     slot k .. q-1      padding (identity rows)    q = NP-1-R
//     slot q             the query point            (features gathered by batch_idx)
//     slot q+1 .. NP-1   the R response rows        (no features)
//
// Phase 1  gather: the (k+1) feature rows are staged once in LDS with coalesced 16-byte
//          loads (consecutive lanes walk a row; rows padded to an odd number of 16-B slots
//          so that ds_read_b128 of a column of rows is bank-conflict free).
// Phase 2  distances, difference form sum((x-y)^2) (never the Gram trick: fp32 parity),
//          each unordered pair ONCE: lane i accumulates d(i, (i+s) mod NP) for s = 1..NP/2
//          in registers (packed-f32 FMAs), reading the partner row from LDS.
// Phase 3  kernel function + nugget, exchanged through a small LDS matrix so that lane i
//          ends up holding row i of the augmented system
//                [ K+eps  .   . ]
//                [ c^T    1   . ]     (lower triangle; see mgp_lds_factor.h for the algebra)
//                [ Y^T    0   0 ]
//          in NP registers.
// Phase 4  right-looking Cholesky, row per lane, all in registers: step j broadcasts
//          column j through a 64-entry LDS buffer (one ds_write_b32, a few uniform
//          ds_read_b128), then a chain of FMAs on the trailing registers.  After k steps
//          the Schur complement of the (query, responses) block holds
//          var = S[q][q],  mean_r = -S[q+1+r][q],  y_r^T K^-1 y_r = -S[q+1+r][q+1+r].
//
// No inter-wave communication, no barriers that wait on other waves (one wave per
// workgroup: __syncthreads() is a compiler/wait-count fence only).
#include "mgp_args.h"

namespace mgp {

template <typename T> struct v16;
template <> struct v16<float> {
  typedef float type __attribute__((ext_vector_type(4)));
  typedef float acc __attribute__((ext_vector_type(2)));
  static constexpr int N = 4;
};
template <> struct v16<double> {
  typedef double type __attribute__((ext_vector_type(2)));
  typedef double acc;
  static constexpr int N = 2;
};

__device__ __forceinline__ void accum(v16<float>::acc& a, const v16<float>::type& df) {
  a = df.xy * df.xy + a;  // v_pk_fma_f32
  a = df.zw * df.zw + a;
}
__device__ __forceinline__ void accum(double& a, const v16<double>::type& df) {
  a = __builtin_fma(df.x, df.x, a);
  a = __builtin_fma(df.y, df.y, a);
}
__device__ __forceinline__ float acc_total(const v16<float>::acc& a) { return a.x + a.y; }
__device__ __forceinline__ double acc_total(const double& a) { return a; }

__device__ __forceinline__ float fast_rcp(float p) {
  float r = __builtin_amdgcn_rcpf(p);
  const float e = __builtin_fmaf(-p, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ double fast_rcp(double p) {
  double r = __builtin_amdgcn_rcp(p);
  double e = __builtin_fma(-p, r, 1.0);
  r = __builtin_fma(e, r, r);
  e = __builtin_fma(-p, r, 1.0);
  return __builtin_fma(e, r, r);
}

struct WaveGeom {
  int q;         // query slot
  int dst;       // feature stage width (elements, multiple of the chunk)
  int xs;        // LDS row stride of the feature tile (elements)
  int vec_ok;    // 16-byte gathers allowed (d % (16/sizeof T) == 0, bases aligned)
  int64_t ntasks;
};

template <typename T, int NP>
__global__ __launch_bounds__(64) void fused_wave_kernel(FusedArgs a, WaveGeom g) {
  constexpr int NH = 64 / NP;     // neighbourhoods per wave
  constexpr int NS = NP / 2;      // cyclic offsets
  constexpr int E = v16<T>::N;    // elements per 16 bytes
  constexpr int CH = 2 * E;       // feature chunk per inner iteration (two 16-B reads per row)
  constexpr int KS = NP + E;      // row stride of the exchange matrix: NP/E + 1 (odd) 16-B slots
  using V = typename v16<T>::type;
  using ACC = typename v16<T>::acc;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int k = a.k, d = a.d, R = a.R, q = g.q, xs = g.xs;
  const int tile_elems = NH * NP * (xs > KS ? xs : KS);
  T* tile = reinterpret_cast<T*>(smem);               // feature tile, later the exchange matrix
  T* colbuf = tile + tile_elems;                      // 64 entries
  T* ilbuf = colbuf + 64;                             // g.dst entries (Anisotropy)
  int64_t* idxbuf = reinterpret_cast<int64_t*>(ilbuf + g.dst + (g.dst & 1));  // 64 entries

  const int lane = threadIdx.x;
  const int h = NH == 1 ? 0 : lane / NP;
  const int i = lane & (NP - 1);
  T* Xh = tile + h * NP * xs;
  T* Kh = tile + h * NP * KS;
  T* colh = colbuf + h * NP;
  int64_t* idxh = idxbuf + h * NP;

  const T* feat_q = static_cast<const T*>(a.feat_q);
  const T* feat_nn = static_cast<const T*>(a.feat_nn);
  const T* targets = static_cast<const T*>(a.targets);
  const T* noise_dev = static_cast<const T*>(a.noise_dev);
  const T* ls = static_cast<const T*>(a.length_scale);
  const bool aniso = a.ls_count > 1;
  T post_scale = T(1);
  if (!aniso) {
    const T l = ls[0];
    post_scale = a.metric_id == MGP_METRIC_L2 ? T(1) / l : T(1) / (l * l);
  }

  // XCD-aware task order: workgroups b and b+8 share an XCD (round-robin dispatch), so give
  // each XCD one contiguous eighth of the neighbourhoods -> neighbouring neighbourhoods
  // (which share rows under a spatially sorted kNN) meet in the same L2.
  const int64_t ntasks = g.ntasks;
  const int64_t per_xcd = (ntasks + 7) / 8;
  const int xcd = blockIdx.x & 7;
  const int64_t t_hi = (xcd + 1) * per_xcd;
  const int64_t t_end = t_hi < ntasks ? t_hi : ntasks;
  const int64_t t_step = gridDim.x >> 3;

  for (int64_t task = xcd * per_xcd + (blockIdx.x >> 3); task < t_end; task += t_step) {
    const int64_t nb_raw = task * NH + h;
    const bool live = nb_raw < a.b;
    const int64_t nb = live ? nb_raw : a.b - 1;

    // ---- phase 0: indices ------------------------------------------------------------
    int64_t myidx = 0;
    if (i < k) myidx = a.nn_idx[nb * k + i];
    else if (i == q) myidx = a.batch_idx ? a.batch_idx[nb] : nb;
    __syncthreads();  // previous task's LDS reads are complete
    idxh[i] = myidx;
    T myeps = T(0);
    if (i < k) {
      if (a.noise_mode == MGP_NOISE_SCALAR) myeps = (T)a.noise_scalar;
      else if (a.noise_mode == MGP_NOISE_TABLE) myeps = noise_dev[myidx];
      else myeps = noise_dev[nb * k + i];
    }

    ACC acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = ACC(0);

    // ---- phases 1+2: stage features, accumulate squared distances ---------------------
    for (int d0 = 0; d0 < d; d0 += g.dst) {
      const int w = min(g.dst, d - d0);
      const int wp = (w + CH - 1) / CH * CH;
      __syncthreads();
      if (g.vec_ok) {
        const int c16 = w / E, c16p = wp / E;
        const unsigned magic = (1u << 20) / (unsigned)c16p + 1u;
        for (int t = i; t < NP * c16p; t += NP) {
          const int row = (int)(((unsigned)t * magic) >> 20);
          const int c = t - row * c16p;
          V v = V(0);
          if (c < c16 && (row < k || row == q)) {
            const T* src = (row < k ? feat_nn : feat_q) + idxh[row] * (int64_t)d + d0 + c * E;
            v = *reinterpret_cast<const V*>(src);
          }
          *reinterpret_cast<V*>(Xh + row * xs + c * E) = v;
        }
      } else {
        const unsigned magic = (1u << 20) / (unsigned)wp + 1u;
        for (int t = i; t < NP * wp; t += NP) {
          const int row = (int)(((unsigned)t * magic) >> 20);
          const int c = t - row * wp;
          T v = T(0);
          if (c < w && (row < k || row == q)) v = ((row < k ? feat_nn : feat_q) + idxh[row] * (int64_t)d + d0)[c];
          Xh[row * xs + c] = v;
        }
      }
      if (aniso)
        for (int c = lane; c < wp; c += 64) ilbuf[c] = c < w ? T(1) / ls[d0 + c] : T(0);
      __syncthreads();

      const T* xown = Xh + i * xs;
      for (int c0 = 0; c0 < wp; c0 += CH) {
        const V own0 = *reinterpret_cast<const V*>(xown + c0);
        const V own1 = *reinterpret_cast<const V*>(xown + c0 + E);
        if (aniso) {
          const V il0 = *reinterpret_cast<const V*>(ilbuf + c0);
          const V il1 = *reinterpret_cast<const V*>(ilbuf + c0 + E);
#pragma unroll
          for (int s = 1; s <= NS; ++s) {
            const T* xo = Xh + ((i + s) & (NP - 1)) * xs + c0;
            const V o0 = *reinterpret_cast<const V*>(xo);
            const V o1 = *reinterpret_cast<const V*>(xo + E);
            accum(acc[s - 1], (own0 - o0) * il0);
            accum(acc[s - 1], (own1 - o1) * il1);
          }
        } else {
#pragma unroll
          for (int s = 1; s <= NS; ++s) {
            const T* xo = Xh + ((i + s) & (NP - 1)) * xs + c0;
            const V o0 = *reinterpret_cast<const V*>(xo);
            const V o1 = *reinterpret_cast<const V*>(xo + E);
            accum(acc[s - 1], own0 - o0);
            accum(acc[s - 1], own1 - o1);
          }
        }
      }
    }

    // ---- phase 3: covariances, nugget, responses -> exchange matrix -> row per lane ----
    __syncthreads();  // every lane is done reading the feature tile (Kh aliases it)
#pragma unroll
    for (int s = 1; s <= NS; ++s) {
      const int c = (i + s) & (NP - 1);
      const int hi = max(i, c), lo = min(i, c);
      const bool valid = lo < k && (hi < k || hi == q);
      const T kv = kernel_eval<T>(a.kernel_id, metric_arg<T>(acc_total(acc[s - 1]), a.metric_id, post_scale));
      if (hi <= q) Kh[hi * KS + lo] = valid ? kv : T(0);
    }
    Kh[i * KS + i] = i < k ? T(1) + myeps : (i <= q ? T(1) : T(0));
    for (int r = 0; r < R; ++r) Kh[(q + 1 + r) * KS + i] = i < k ? targets[myidx * (int64_t)R + r] : T(0);
    __syncthreads();
    T A[NP];
#pragma unroll
    for (int c4 = 0; c4 < NP / E; ++c4) {
      const V v = *reinterpret_cast<const V*>(Kh + i * KS + c4 * E);
#pragma unroll
      for (int e = 0; e < E; ++e) A[c4 * E + e] = v[e];
    }

    // ---- phase 4: Cholesky, row per lane, column broadcast through LDS ----------------
    bool bad = false;
#pragma unroll
    for (int j = 0; j < NP - 2; ++j) {
      if (j < k) {
        colh[i] = A[j];
        T col[NP];
#pragma unroll
        for (int c4 = j / E; c4 < NP / E; ++c4) {
          const V v = *reinterpret_cast<const V*>(colh + c4 * E);
#pragma unroll
          for (int e = 0; e < E; ++e) col[c4 * E + e] = v[e];
        }
        const T p = col[j];
        bad = bad || !(p > T(0));
        const T t = A[j] * fast_rcp(p);
#pragma unroll
        for (int c = j + 1; c < NP; ++c) A[c] = __builtin_fma(-t, col[c], A[c]);
      }
    }

    // ---- phase 5: Schur block -> outputs ----------------------------------------------
    __syncthreads();
#pragma unroll
    for (int c4 = 0; c4 < NP / E; ++c4) {
      V v;
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = A[c4 * E + e];
      *reinterpret_cast<V*>(Kh + i * KS + c4 * E) = v;
    }
    __syncthreads();
    if (live) {
      T* mean = static_cast<T*>(a.mean);
      T* var = static_cast<T*>(a.var);
      T* yk = static_cast<T*>(a.ykinvy);
      if (i == q) {
        var[nb] = bad ? num<T>::nan() : Kh[q * KS + q];
        if (bad && a.info) atomicAdd(a.info, 1);
      } else if (i > q) {
        const int r = i - q - 1;
        mean[nb * R + r] = bad ? num<T>::nan() : -Kh[i * KS + q];
        if (yk) yk[nb * R + r] = bad ? num<T>::nan() : -Kh[i * KS + i];
      }
    }
  }
}

template <typename T, int NP>
static int launch_np(const FusedArgs& a, hipStream_t stream) {
  constexpr int NH = 64 / NP;
  constexpr int E = v16<T>::N;
  constexpr int CH = 2 * E;
  constexpr int KS = NP + E;
  WaveGeom g;
  g.q = NP - 1 - a.R;
  const int dpad = (a.d + CH - 1) / CH * CH;
  g.dst = dpad < 64 ? dpad : 64;
  g.xs = g.dst + E;  // dst/E is even -> dst/E + 1 slots: odd
  const uintptr_t align = (uintptr_t)a.feat_q | (uintptr_t)a.feat_nn;
  g.vec_ok = (a.d % E == 0) && (align % 16 == 0);
  g.ntasks = (a.b + NH - 1) / NH;
  const int rowmax = g.xs > KS ? g.xs : KS;
  size_t lds = ((size_t)NH * NP * rowmax + 64 + g.dst + (g.dst & 1)) * sizeof(T) + 64 * sizeof(int64_t);
  lds = (lds + 15) & ~(size_t)15;
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 32) per_cu = 32;
  if (per_cu < 1) return MGP_EUNSUPPORTED;
  int64_t grid = 256LL * per_cu;  // resident waves; every workgroup grid-strides its XCD's range
  if (grid > g.ntasks) grid = (g.ntasks + 7) / 8 * 8;
  hipLaunchKernelGGL((fused_wave_kernel<T, NP>), dim3((unsigned)grid), dim3(64), lds, stream, a, g);
  MGP_HIP_CHECK_LAUNCH();
  return MGP_OK;
}

template <typename T>
int launch_fused_wave(const FusedArgs& a, hipStream_t stream) {
  const int rows = a.k + 1 + a.R;
  if (rows <= 32) return launch_np<T, 32>(a, stream);
  if (rows <= 64) return launch_np<T, 64>(a, stream);
  return MGP_EUNSUPPORTED;
}

template int launch_fused_wave<float>(const FusedArgs&, hipStream_t);
template int launch_fused_wave<double>(const FusedArgs&, hipStream_t);

}  // namespace mgp

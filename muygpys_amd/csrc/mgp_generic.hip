// Generic (any nn_count that fits LDS) kernels of the local-GP hot path:
//   * fused gather -> distances -> kernel -> nugget -> factor -> mean/var/yKinvy
//   * the same factorisation on a materialised Kin (API parity: S1/S2/S3 + fast precompute)
// One workgroup owns one neighbourhood at a time; its whole local system lives in LDS
// (see mgp_lds_factor.h).  The specialised register-resident kernels in mgp_fused_wave.hip
// take over for the shapes they cover; this file is the path for everything else.
#include "mgp_args.h"
#include "mgp_lds_factor.h"

namespace mgp {

// feature-tile row stride for a chunk of dc features (dc a multiple of 2E): 16-byte aligned rows, odd
// number of 16-byte slots
template <typename T>
__host__ __device__ inline int tile_row_stride(int dc) { return dc + 16 / (int)sizeof(T); }

// dynamic LDS carve (every array 16-byte aligned):
// [idx: (k+2 & ~1) int64][S: rows*SP T][X: (k+1)*XP T][il: dc T][piv: k T][flag]
template <typename T>
__global__ void fused_generic_kernel(FusedArgs a) {
  using V = typename lds_vec<T>::type;
  constexpr int E = 16 / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int k = a.k, d = a.d, R = a.R, dc = a.dc;
  const int rows = k + 1 + R;
  const int SP = lds_row_stride<T>(k);
  const int XP = tile_row_stride<T>(dc);
  int64_t* idx = reinterpret_cast<int64_t*>(smem);
  T* S = reinterpret_cast<T*>(idx + ((k + 2) & ~1));
  T* X = S + rows * SP;
  T* il = X + (k + 1) * XP;
  T* piv = il + dc;
  int* flag = reinterpret_cast<int*>(piv + k);

  const T* feat_q = static_cast<const T*>(a.feat_q);
  const T* feat_nn = static_cast<const T*>(a.feat_nn);
  const T* targets = static_cast<const T*>(a.targets);
  const T* noise_dev = static_cast<const T*>(a.noise_dev);
  const T* ls = static_cast<const T*>(a.length_scale);
  const int tid = threadIdx.x, NT = blockDim.x;
  const bool aniso = a.ls_count > 1;
  const int npairs = (k + 1) * k / 2;
  const bool vec_ok = d % E == 0 && dc % E == 0 && ((uintptr_t)feat_q | (uintptr_t)feat_nn) % 16 == 0;

  T post_scale = T(1);
  if (!aniso) {
    const T l = ls[0];
    post_scale = a.metric_id == MGP_METRIC_L2 ? T(1) / l : T(1) / (l * l);
  }

  for (int64_t nb = blockIdx.x; nb < a.b; nb += gridDim.x) {
    __syncthreads();  // previous neighbourhood fully consumed
    for (int r = tid; r <= k; r += NT)
      idx[r] = r < k ? a.nn_idx[nb * k + r] : (a.batch_idx ? a.batch_idx[nb] : nb);
    __syncthreads();

    for (int d0 = 0; d0 < d; d0 += dc) {
      const int w = min(dc, d - d0);
      const int wp = (w + E - 1) / E * E;  // zero-padded to whole 16-byte pieces
      // coalesced gather of the (k+1) x w feature tile: consecutive lanes walk a row
      gather_tile_lds<T>(X, XP, feat_q, feat_nn, idx, k, d, d0, w, wp, vec_ok, tid, NT);
      if (aniso)
        for (int c = tid; c < wp; c += NT) il[c] = c < w ? T(1) / ls[d0 + c] : T(0);
      __syncthreads();
      for (int p = tid; p < npairs; p += NT) {
        // p -> (row a_, col c_) of the strict lower triangle of the (k+1)-point set
        int a_ = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
        while (a_ * (a_ - 1) / 2 > p) --a_;
        while ((a_ + 1) * a_ / 2 <= p) ++a_;
        const int c_ = p - a_ * (a_ - 1) / 2;
        const T* xa = X + a_ * XP;
        const T* xc = X + c_ * XP;
        T acc = T(0);
        if (aniso) {
#pragma unroll 2
          for (int j = 0; j < wp; j += E) {
            const V df = (*reinterpret_cast<const V*>(xa + j) - *reinterpret_cast<const V*>(xc + j)) *
                         *reinterpret_cast<const V*>(il + j);
            acc += vec_dot(df, df);
          }
        } else {
#pragma unroll 2
          for (int j = 0; j < wp; j += E) {
            const V df = *reinterpret_cast<const V*>(xa + j) - *reinterpret_cast<const V*>(xc + j);
            acc += vec_dot(df, df);
          }
        }
        T* dst = S + a_ * SP + c_;
        *dst = d0 == 0 ? acc : *dst + acc;
      }
      __syncthreads();
    }
    // distances -> covariances (in place), nugget on the diagonal, responses into the tail rows
    for (int p = tid; p < npairs; p += NT) {
      int a_ = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
      while (a_ * (a_ - 1) / 2 > p) --a_;
      while ((a_ + 1) * a_ / 2 <= p) ++a_;
      const int c_ = p - a_ * (a_ - 1) / 2;
      T* dst = S + a_ * SP + c_;
      *dst = kernel_eval<T>(a.kernel_id, metric_arg<T>(*dst, a.metric_id, post_scale));
    }
    for (int r = tid; r < k; r += NT) {
      T eps;
      if (a.noise_mode == MGP_NOISE_SCALAR) eps = (T)a.noise_scalar;
      else if (a.noise_mode == MGP_NOISE_TABLE) eps = noise_dev[idx[r]];
      else eps = noise_dev[nb * k + r];
      S[r * SP + r] = T(1) + eps;  // kernel(0) == 1 for every kernel on the path
    }
    for (int t = tid; t < k * R; t += NT) {
      const int c = t / R, r = t - c * R;
      S[(k + 1 + r) * SP + c] = targets[(a.targets_batch ? nb * k + c : idx[c]) * (int64_t)R + r];
    }
    __syncthreads();
    const bool bad = factor_augmented_lds<T>(S, SP, k, rows, piv, flag, tid, NT);
    T* mean = static_cast<T*>(a.mean);
    T* var = static_cast<T*>(a.var);
    T* yk = static_cast<T*>(a.ykinvy);
    emit_outputs_lds<T>(S, SP, k, R, T(1), bad, true, mean ? mean + nb * R : nullptr, var ? var + nb : nullptr,
                        yk ? yk + nb * R : nullptr, tid, NT);
    if (bad && tid == 0 && a.info) atomicAdd(a.info, 1);
  }
}

// LDS carve: [S: rows*SP T][piv: k T][flag]
template <typename T>
__global__ void solve_generic_kernel(SolveArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int k = a.k, R = a.R;
  const int rows = k + 1 + R;
  const int SP = lds_row_stride<T>(k);
  T* S = reinterpret_cast<T*>(smem);
  T* piv = S + rows * SP;
  int* flag = reinterpret_cast<int*>(piv + k);
  const T* Kin = static_cast<const T*>(a.Kin);
  const T* Kcross = static_cast<const T*>(a.Kcross);
  const T* Y = static_cast<const T*>(a.Y);
  const int tid = threadIdx.x, NT = blockDim.x;

  for (int64_t nb = blockIdx.x; nb < a.b; nb += gridDim.x) {
    __syncthreads();
    const T* Kb = Kin + nb * (int64_t)k * k;
    for (int t = tid; t < k * k; t += NT) {
      const int i = t / k, j = t - i * k;
      if (j <= i) S[i * SP + j] = Kb[t];  // lower triangle, as LAPACK 'L' would read it
    }
    for (int c = tid; c < k; c += NT) S[k * SP + c] = Kcross ? Kcross[nb * k + c] : T(0);
    for (int t = tid; t < k * R; t += NT) {
      const int c = t / R, r = t - c * R;
      S[(k + 1 + r) * SP + c] = Y[(nb * k + c) * (int64_t)R + r];
    }
    __syncthreads();
    const bool bad = factor_augmented_lds<T>(S, SP, k, rows, piv, flag, tid, NT);
    T* mean = static_cast<T*>(a.mean);
    T* var = static_cast<T*>(a.var);
    T* yk = static_cast<T*>(a.ykinvy);
    emit_outputs_lds<T>(S, SP, k, R, (T)a.kout, bad, Kcross != nullptr, mean ? mean + nb * R : nullptr,
                        var ? var + nb : nullptr, yk ? yk + nb * R : nullptr, tid, NT);
    if (a.coeffs) {
      // back-substitution L^T x = (L^-1 y_r), one response per thread, in place in row k+1+r
      T* co = static_cast<T*>(a.coeffs) + nb * (int64_t)k * R;
      __syncthreads();
      for (int r = tid; r < R; r += NT) {
        T* x = S + (k + 1 + r) * SP;
        for (int j = k - 1; j >= 0; --j) {
          T s = x[j];
          for (int m = j + 1; m < k; ++m) s -= S[m * SP + j] * x[m];
          x[j] = s * piv[j];
        }
        for (int j = 0; j < k; ++j) co[j * (int64_t)R + r] = bad ? num<T>::nan() : x[j];
      }
    }
    if (bad && tid == 0 && a.info) atomicAdd(a.info, 1);
  }
}

static int block_threads(int rows) { return rows <= 64 ? 64 : 256; }
static const size_t kMaxLds = 160 * 1024;

template <typename T>
static size_t fused_lds_bytes(int k, int R, int dc) {
  const int rows = k + 1 + R, SP = lds_row_stride<T>(k);
  size_t n = (size_t)((k + 2) & ~1) * sizeof(int64_t);
  n += ((size_t)rows * SP + (size_t)(k + 1) * tile_row_stride<T>(dc) + dc + k) * sizeof(T) + 16;
  return (n + 15) & ~(size_t)15;
}
template <typename T>
static size_t solve_lds_bytes(int k, int R) {
  const int rows = k + 1 + R, SP = lds_row_stride<T>(k);
  size_t n = ((size_t)rows * SP + k) * sizeof(T) + 16;
  return (n + 15) & ~(size_t)15;
}

static int grid_for(int64_t b, size_t lds) {
  // Persistent grid = the resident capacity (the kernels grid-stride over the rest): LDS is handed
  // out in 1280-byte granules of the CU's 160 KiB (measured, see mgp_fused_wave.hip); one workgroup
  // more than fits would run as a second, nearly empty round.
  const size_t granules = (lds + 1279) / 1280;
  int per_cu = (int)(kMaxLds / ((granules ? granules : 1) * 1280));
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 16) per_cu = 16;
  int64_t g = 256LL * per_cu;
  return (int)(b < g ? b : g);
}

template <typename T>
int launch_fused_generic(const FusedArgs& in, hipStream_t stream) {
  FusedArgs a = in;
  // feature chunk: as wide as fits next to the factor (bounded so small problems stay small)
  // feature chunk: a multiple of 8 (whole 16-byte pieces, odd slot count with the pad), as wide as fits
  int dc = a.d < 64 ? (a.d + 7) / 8 * 8 : 64;
  while (dc > 8 && fused_lds_bytes<T>(a.k, a.R, dc) > kMaxLds) dc -= 8;
  const size_t lds = fused_lds_bytes<T>(a.k, a.R, dc);
  if (lds > kMaxLds) return MGP_EUNSUPPORTED;
  a.dc = dc;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_generic_kernel<T>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return -(1000 + (int)e);
  }
  hipLaunchKernelGGL(fused_generic_kernel<T>, dim3(grid_for(a.b, lds)), dim3(block_threads(a.k + 1 + a.R)), lds,
                     stream, a);
  MGP_HIP_CHECK_LAUNCH();
  note_launch("mgp::fused_generic_kernel<%s>", sizeof(T) == 4 ? "float" : "double");
  return MGP_OK;
}

template <typename T>
int launch_solve_generic(const SolveArgs& a, hipStream_t stream) {
  const size_t lds = solve_lds_bytes<T>(a.k, a.R);
  if (lds > kMaxLds) return MGP_EUNSUPPORTED;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_generic_kernel<T>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return -(1000 + (int)e);
  }
  hipLaunchKernelGGL(solve_generic_kernel<T>, dim3(grid_for(a.b, lds)), dim3(block_threads(a.k + 1 + a.R)), lds,
                     stream, a);
  MGP_HIP_CHECK_LAUNCH();
  return MGP_OK;
}

template int launch_fused_generic<float>(const FusedArgs&, hipStream_t);
template int launch_fused_generic<double>(const FusedArgs&, hipStream_t);
template int launch_solve_generic<float>(const SolveArgs&, hipStream_t);
template int launch_solve_generic<double>(const SolveArgs&, hipStream_t);

int max_nn_count(int elem_size, int R) {
  int k = 1;
  while (true) {
    const size_t lds = elem_size == 4 ? fused_lds_bytes<float>(k + 1, R, 8) : fused_lds_bytes<double>(k + 1, R, 8);
    if (lds > kMaxLds) break;
    ++k;
  }
  return k;
}

}  // namespace mgp

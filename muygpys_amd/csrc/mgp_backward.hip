// Backward pass (vector-Jacobian product) of the fused local-GP path.
//
// The reference gets these gradients from torch autograd over its torch backend
// (torch/muygps_layer.py:129-164 builds crosswise/pairwise tensors -> kernel -> posterior mean
// and variance; examples/muygps_torch.py:425-437 calls loss.sum().backward()), which keeps every
// (b,k,k,d) intermediate alive for the backward sweep.  Here one workgroup recomputes the local
// system of a neighbourhood in LDS and contracts the cotangents analytically:
//
//   mean_r = c^T K^-1 y_r,  var = 1 - c^T K^-1 c,   a = K^-1 c,   w = K^-1 (Y gm)
//   dL = dc . (w - 2 gv a) + sum_ij dK_ij (gv a_i a_j - a_i w_j) + sum_r gm_r a . dy_r
//
// so two solves (rows k and k+1 of the augmented factor, then a back-substitution) serve any
// response count.  K_ij = kappa(x(acc_ij)) with acc = sum_d ((x_i - x_j) / l_d)^2; the chain through
// kappa and the metric is applied per pair, and the per-point feature cotangents leave with one
// atomic add per (point, feature).
#include "mgp_args.h"
#include "mgp_lds_factor.h"

namespace mgp {

#ifdef MGP_DEBUG_HOOKS
int g_bwd_stage = 0;  // timing ablations only: stop every neighbourhood after stage N (0 = run all)
#else
static const int g_bwd_stage = 0;
#endif

__device__ __forceinline__ void tri_decode(int p, int& row, int& col) {
  int a_ = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
  while (a_ * (a_ - 1) / 2 > p) --a_;
  while ((a_ + 1) * a_ / 2 <= p) ++a_;
  row = a_;
  col = p - a_ * (a_ - 1) / 2;
}

// Single-wave form of "factor, then solve for two right-hand sides" (k + 2 <= 64 rows, 64 threads): the
// row-per-lane register elimination of mgp_fused_wave.hip (column broadcast through a 64-entry LDS
// buffer) on the system assembled in S, multipliers kept in S's lower triangle, then the
// back-substitution L^T [a w] = D^-1 L^-1 [c yt] with the finished component handed down by v_readlane.
// Leaves a = K^-1 c in row k and w = K^-1 yt in row k+1 of S, like the LDS path.  Returns "bad pivot".
template <typename T>
__device__ inline bool wave_factor_solve2(T* S, int SP, int k, T* colb, int lane) {
  constexpr int NP = 64;
  constexpr int E = 16 / (int)sizeof(T);
  using V = typename lds_vec<T>::type;
  const int rows = k + 2;
  V A[NP / E];
#pragma unroll
  for (int c4 = 0; c4 < NP / E; ++c4) A[c4] = V(0);
  if (lane < rows) {
    const T* src = S + lane * SP;
#pragma unroll
    for (int c4 = 0; c4 < NP / E; ++c4)
      if (c4 * E < k) A[c4] = *reinterpret_cast<const V*>(src + c4 * E);  // rows are 16-byte aligned, SP >= k + 1
  }
  bool bad = false;
#pragma unroll
  for (int j = 0; j < NP - 2; ++j) {
    if (j < k) {
      const T ajj = A[j / E][j % E];
      __syncthreads();
      colb[lane] = ajj;
      __syncthreads();
      const V cp = *reinterpret_cast<const V*>(colb + (j / E) * E);
      const T p = cp[j % E];
      bad = bad || !(p > T(0));
      const T t = lane > j && lane < rows ? ajj / p : T(0);
      if (lane > j && lane < rows) S[lane * SP + j] = t;  // multiplier l_ij (rows k, k+1: D^-1 L^-1 of the rhs)
      const V nt = V(-t);
      A[j / E] = cp * nt + A[j / E];
      constexpr int GC = sizeof(T) == 4 ? 8 : 4;
#pragma unroll
      for (int c0 = j / E + 1; c0 < NP / E; c0 += GC) {
        V cv[GC];
#pragma unroll
        for (int u = 0; u < GC; ++u)
          if (c0 + u < NP / E) cv[u] = *reinterpret_cast<const V*>(colb + (c0 + u) * E);
#pragma unroll
        for (int u = 0; u < GC; ++u)
          if (c0 + u < NP / E) A[c0 + u] = cv[u] * nt + A[c0 + u];
      }
    }
  }
  __syncthreads();
  // back-substitution, both vectors at once: lane j < k holds x_a(j), x_w(j)
  T xa = lane < k ? S[k * SP + lane] : T(0);
  T xw = lane < k ? S[(k + 1) * SP + lane] : T(0);
  for (int m = k - 1; m >= 1; --m) {
    const T am = __shfl(xa, m, 64), wm = __shfl(xw, m, 64);
    if (lane < m) {
      const T l = S[m * SP + lane];
      xa -= l * am;
      xw -= l * wm;
    }
  }
  __syncthreads();
  if (lane < k) {
    S[k * SP + lane] = xa;
    S[(k + 1) * SP + lane] = xw;
  }
  __syncthreads();
  return bad;
}

// dynamic LDS carve (every array 16-byte aligned; dc a multiple of 8):
// [idx (k+2 & ~1) i64][S (k+2)*SP][Q (k+1)*SP][X (k+1)*XP][il dc][lacc dc][colb 64][piv k][red 2][flag]
template <typename T>
__global__ __launch_bounds__(256) void backward_kernel(BackwardArgs g, int stage) {
  using V = typename lds_vec<T>::type;
  constexpr int E = 16 / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const FusedArgs& a = g.f;
  const int k = a.k, d = a.d, R = a.R, dc = a.dc;
  const int rows = k + 2;
  const int SP = lds_row_stride<T>(k);
  const int XP = dc + E;
  int64_t* idx = reinterpret_cast<int64_t*>(smem);
  T* S = reinterpret_cast<T*>(idx + ((k + 2) & ~1));
  T* Q = S + rows * SP;
  T* X = Q + (k + 1) * SP;
  T* il = X + (k + 1) * XP;
  T* lacc = il + dc;
  T* colb = lacc + dc;  // 64-entry column broadcast buffer of the single-wave factorisation
  T* piv = colb + 64;
  T* red = piv + k;
  int* flag = reinterpret_cast<int*>(red + 2);

  const T* feat_q = static_cast<const T*>(a.feat_q);
  const T* feat_nn = static_cast<const T*>(a.feat_nn);
  const T* targets = static_cast<const T*>(a.targets);
  const T* noise_dev = static_cast<const T*>(a.noise_dev);
  const T* ls = static_cast<const T*>(a.length_scale);
  const T* gmean = static_cast<const T*>(g.grad_mean);
  const T* gvar = static_cast<const T*>(g.grad_var);
  const T* gyk = static_cast<const T*>(g.grad_yk);  // (LOOCV, R = 1: the right-hand side is y, w = gm u below)
  T* gq = static_cast<T*>(g.grad_feat_q);
  T* gnn = static_cast<T*>(g.grad_feat_nn);
  T* gtg = static_cast<T*>(g.grad_targets);
  T* gls = static_cast<T*>(g.grad_ls);
  T* gnz = static_cast<T*>(g.grad_noise);
  const int tid = threadIdx.x, NT = blockDim.x;
  const bool aniso = a.ls_count > 1;
  const bool l2 = a.metric_id == MGP_METRIC_L2;
  const int npairs = (k + 1) * k / 2;
  const bool vec_ok = d % E == 0 && dc % E == 0 && ((uintptr_t)feat_q | (uintptr_t)feat_nn) % 16 == 0;

  T post_scale = T(1), inv_l = T(1);
  if (!aniso) {
    inv_l = T(1) / ls[0];
    post_scale = l2 ? inv_l : inv_l * inv_l;
  }

  for (int64_t nb = blockIdx.x; nb < a.b; nb += gridDim.x) {
    __syncthreads();
    for (int r = tid; r <= k; r += NT)
      idx[r] = r < k ? a.nn_idx[nb * k + r] : (a.batch_idx ? a.batch_idx[nb] : nb);
    if (tid == 0) red[0] = T(0);
    __syncthreads();

    // ---- forward recomputation: acc (kept in Q), covariances (S), combined right-hand side ----
    for (int d0 = 0; d0 < d; d0 += dc) {
      const int w = min(dc, d - d0);
      const int wp = (w + E - 1) / E * E;  // zero-padded to whole 16-byte pieces
      gather_tile_lds<T>(X, XP, feat_q, feat_nn, idx, k, d, d0, w, wp, vec_ok, tid, NT);
      if (aniso)
        for (int c = tid; c < wp; c += NT) il[c] = c < w ? T(1) / ls[d0 + c] : T(0);
      __syncthreads();
      for (int p = tid; p < npairs; p += NT) {
        int a_, c_;
        tri_decode(p, a_, c_);
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
        T* dst = Q + a_ * SP + c_;
        *dst = d0 == 0 ? acc : *dst + acc;
      }
      __syncthreads();
    }
    for (int p = tid; p < npairs; p += NT) {
      int a_, c_;
      tri_decode(p, a_, c_);
      S[a_ * SP + c_] = kernel_eval<T>(a.kernel_id, metric_arg<T>(Q[a_ * SP + c_], a.metric_id, post_scale));
    }
    for (int r = tid; r < k; r += NT) {
      T eps;
      if (a.noise_mode == MGP_NOISE_SCALAR) eps = (T)a.noise_scalar;
      else if (a.noise_mode == MGP_NOISE_TABLE) eps = noise_dev[idx[r]];
      else eps = noise_dev[nb * k + r];
      S[r * SP + r] = T(1) + eps;
      T yt = T(0);
      if (gyk)
        yt = targets[idx[r] * (int64_t)R];
      else if (gmean)
        for (int q = 0; q < R; ++q) yt += gmean[nb * R + q] * targets[idx[r] * (int64_t)R + q];
      S[(k + 1) * SP + r] = yt;
    }
    __syncthreads();
    if (stage == 1) continue;
    const bool wave_path = NT == 64 && rows <= 64;  // uniform
    bool bad;
    if (wave_path) {
      bad = wave_factor_solve2<T>(S, SP, k, colb, tid);
    } else {
      bad = factor_augmented_lds<T>(S, SP, k, rows, piv, flag, tid, NT);
    }
    if (stage == 2) continue;
    if (bad) {
      if (tid == 0 && a.info) atomicAdd(a.info, 1);
      continue;  // cotangents of a non-SPD neighbourhood are left untouched
    }
    T* av = S + k * SP;
    T* wv = S + (k + 1) * SP;
    if (!wave_path) {
      // ---- back-substitution L^T [a w] = [z zy], both vectors at once, column oriented ----
      for (int j = k - 1; j >= 0; --j) {
        if (tid < 2) S[(k + tid) * SP + j] *= piv[j];
        __syncthreads();
        const T aj = av[j], wj = wv[j];
        const T* Lj = S + j * SP;
        for (int m = tid; m < j; m += NT) {
          const T l = Lj[m];
          av[m] -= l * aj;
          wv[m] -= l * wj;
        }
        __syncthreads();
      }
    }
    if (stage == 3) continue;
    const T gv = gvar ? gvar[nb] : T(0);
    // (LOOCV: wv holds u = K^-1 y; the mean's cotangent enters as the factor gm, y^T K^-1 y's as gyv)
    const T gyv = gyk ? gyk[nb] : T(0);
    const T gm1 = gyk ? (gmean ? gmean[nb] : T(0)) : T(1);

    // ---- cotangent of every pair's acc (overwrites Q), isotropic length-scale partial ----
    T liso = T(0);
    for (int p = tid; p < npairs; p += NT) {
      int a_, c_;
      tri_decode(p, a_, c_);
      T gK;
      if (a_ < k) gK = T(2) * gv * av[a_] * av[c_] - gm1 * (av[a_] * wv[c_] + av[c_] * wv[a_]) - T(2) * gyv * wv[a_] * wv[c_];
      else gK = gm1 * wv[c_] - T(2) * gv * av[c_];
      const T acc = Q[a_ * SP + c_];
      const T x = metric_arg<T>(acc, a.metric_id, post_scale);
      const T kp = kernel_deriv<T>(a.kernel_id, x);
      T dk_dacc;
      if (l2) dk_dacc = x > T(0) ? kp * post_scale * post_scale / (T(2) * x) : T(0);
      else dk_dacc = kp * post_scale;
      const T q = gK * dk_dacc;
      Q[a_ * SP + c_] = q;  // symmetric, zero diagonal: the per-point sweep below runs a uniform loop
      Q[c_ * SP + a_] = q;
      liso += gK * kp * x;
    }
    for (int r = tid; r <= k; r += NT) Q[r * SP + r] = T(0);
    if (stage == 4) continue;
    if (gls && !aniso) {
      liso = wave_sum(liso);
      if ((tid & 63) == 0) atomicAdd(&red[0], liso);
    }
    if (gnz)
      for (int r = tid; r < k; r += NT) gnz[nb * k + r] = gv * av[r] * av[r] - gm1 * av[r] * wv[r] - gyv * wv[r] * wv[r];
    if (gtg && gmean)
      for (int t = tid; t < k * R; t += NT) {
        const int c = t / R, r = t - c * R;
        unsafeAtomicAdd(gtg + idx[c] * (int64_t)R + r, gmean[nb * R + r] * av[c] + (gyk ? T(2) * gyv * wv[c] : T(0)));
      }
    __syncthreads();
    if (gls && !aniso && tid == 0) gls[nb] = -(l2 ? T(1) : T(2)) * inv_l * red[0];  // dx/dl = -x/l | -2x/l

    if (stage == 5) continue;
    // ---- per-point feature cotangents (and anisotropic length-scale partials) ----
    if (!gq && !gnn && !(gls && aniso)) continue;
    for (int d0 = 0; d0 < d; d0 += dc) {
      const int w = min(dc, d - d0);
      const int wp = (w + E - 1) / E * E;
      __syncthreads();
      gather_tile_lds<T>(X, XP, feat_q, feat_nn, idx, k, d, d0, w, wp, vec_ok, tid, NT);
      for (int c = tid; c < wp; c += NT) {
        il[c] = c < w ? (aniso ? T(1) / ls[d0 + c] : T(1)) : T(0);
        lacc[c] = T(0);
      }
      __syncthreads();
      // one thread per (point, 16-byte piece of its row): q_ij is read once per four features
      const int pieces = wp / E;
      for (int t = tid; t < (k + 1) * pieces; t += NT) {
        const int i = t / pieces, c0 = (t - i * pieces) * E;
        const V xi = *reinterpret_cast<const V*>(X + i * XP + c0);
        const T* qi = Q + i * SP;
        V s = V(0), s2 = V(0);
#pragma unroll 2
        for (int j = 0; j <= k; ++j) {
          const T qv = qi[j];
          const V df = xi - *reinterpret_cast<const V*>(X + j * XP + c0);
          s += qv * df;
          s2 += qv * df * df;
        }
        const V ilv = *reinterpret_cast<const V*>(il + c0);
        const V gx = T(2) * s * ilv * ilv;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int c = c0 + e;
          if (c >= w) continue;
          if (stage == 6) {  // ablation: plain store instead of the atomic
            if (gnn) gnn[idx[i] * (int64_t)d + d0 + c] = gx[e];
          } else if (i < k) {
            if (gnn) unsafeAtomicAdd(gnn + idx[i] * (int64_t)d + d0 + c, gx[e]);
          } else if (gq) {
            unsafeAtomicAdd(gq + idx[k] * (int64_t)d + d0 + c, gx[e]);
          }
          if (gls && aniso) atomicAdd(&lacc[c], T(0.5) * s2[e]);  // every pair was visited from both ends
        }
      }
      __syncthreads();
      if (gls && aniso)
        for (int c = tid; c < w; c += NT)
          gls[nb * (int64_t)a.ls_count + d0 + c] = T(-2) * il[c] * il[c] * il[c] * lacc[c];
    }
  }
}

static const size_t kMaxLdsBwd = 160 * 1024;

template <typename T>
static size_t backward_lds_bytes(int k, int dc) {
  const int SP = lds_row_stride<T>(k);
  size_t n = (size_t)((k + 2) & ~1) * sizeof(int64_t);
  n += ((size_t)(k + 2) * SP + (size_t)(k + 1) * SP + (size_t)(k + 1) * (dc + 16 / sizeof(T)) + 2 * (size_t)dc + 64 + k +
        2) * sizeof(T) + 16;
  return (n + 15) & ~(size_t)15;
}

template <typename T>
int launch_backward(const BackwardArgs& in, hipStream_t stream) {
  BackwardArgs g = in;
  int dc = g.f.d < 64 ? (g.f.d + 7) / 8 * 8 : 64;  // multiple of 8: whole 16-byte pieces, odd slot count
  while (dc > 8 && backward_lds_bytes<T>(g.f.k, dc) > kMaxLdsBwd) dc -= 8;
  const size_t lds = backward_lds_bytes<T>(g.f.k, dc);
  if (lds > kMaxLdsBwd) return MGP_EUNSUPPORTED;
  g.f.dc = dc;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&backward_kernel<T>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return -(1000 + (int)e);
  }
  int per_cu = (int)(kMaxLdsBwd / (((lds + 1279) / 1280) * 1280));  // LDS comes in 1280-byte granules
  per_cu = per_cu < 1 ? 1 : (per_cu > 16 ? 16 : per_cu);
  const int64_t full = 256LL * per_cu;
  const int threads = g.f.k + 2 <= 64 ? 64 : (g.f.k + 2 <= 128 ? 128 : 256);
  hipLaunchKernelGGL(backward_kernel<T>, dim3((unsigned)(g.f.b < full ? g.f.b : full)), dim3(threads), lds, stream, g, g_bwd_stage);
  MGP_HIP_CHECK_LAUNCH();
  return MGP_OK;
}

template int launch_backward<float>(const BackwardArgs&, hipStream_t);
template int launch_backward<double>(const BackwardArgs&, hipStream_t);

int max_nn_count_backward(int elem_size) {
  int k = 1;
  while ((elem_size == 4 ? backward_lds_bytes<float>(k + 1, 8) : backward_lds_bytes<double>(k + 1, 8)) <= kMaxLdsBwd)
    ++k;
  return k;
}

}  // namespace mgp

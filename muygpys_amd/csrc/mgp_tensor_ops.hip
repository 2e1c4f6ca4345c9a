// Materialising per-function kernels: the HIP counterparts of the reference's backend
// functions when a caller insists on the intermediate tensors (API parity).  All are
// HBM-bound elementwise / gather kernels: one output element (or one feature chunk) per
// lane, consecutive lanes on consecutive addresses.
#include <cmath>

#include "mgp_device.h"
#include "mgp_loocv_tree.h"

namespace mgp {

static const int kBlock = 256;

static inline int grid_1d(int64_t n) {
  int64_t g = ceil_div(n, kBlock);
  const int64_t cap = 256LL * 32;  // 256 CUs x 32 resident waves; grid-stride beyond
  return (int)(g < cap ? (g < 1 ? 1 : g) : cap);
}

// T1: out[b,j,:] = q[batch_idx[b],:] - x[nn_idx[b,j],:]      _src/gp/tensors/numpy.py:47-58
template <typename T>
__global__ void crosswise_diffs_kernel(const T* fq, const T* fn, int d, const int64_t* bidx, const int64_t* nidx,
                                       int64_t b, int k, T* out) {
  const int64_t n = b * k * (int64_t)d;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / d;
    const int c = (int)(t - row * d);
    const int64_t bi = row / k;
    const int64_t q = bidx ? bidx[bi] : bi;
    out[t] = fq[q * d + c] - fn[nidx[row] * d + c];
  }
}

// 16-byte form of the two difference kernels (d a multiple of the vector width, 16-byte aligned tables):
// one divide chain per four (two) elements instead of per element, 16-byte loads and stores
template <typename T>
__global__ void crosswise_diffs_vec_kernel(const T* fq, const T* fn, int dv, const int64_t* bidx, const int64_t* nidx,
                                           int64_t b, int k, T* out) {
  constexpr int E = 16 / (int)sizeof(T);
  using V = T __attribute__((ext_vector_type(E)));
  const int64_t n = b * k * (int64_t)dv;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / dv;
    const int c = (int)(t - row * dv);
    const int64_t bi = row / k;
    const int64_t q = bidx ? bidx[bi] : bi;
    *reinterpret_cast<V*>(out + t * E) = *reinterpret_cast<const V*>(fq + (q * dv + c) * E) -
                                         *reinterpret_cast<const V*>(fn + (nidx[row] * dv + c) * E);
  }
}
template <typename T>
__global__ void pairwise_diffs_vec_kernel(const T* f, int dv, const int64_t* nidx, int64_t b, int k, T* out) {
  constexpr int E = 16 / (int)sizeof(T);
  using V = T __attribute__((ext_vector_type(E)));
  const int64_t n = b * k * (int64_t)k * dv;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pr = t / dv;
    const int c = (int)(t - pr * dv);
    const int64_t bi = pr / ((int64_t)k * k);
    const int ij = (int)(pr - bi * k * k);
    const int i = ij / k, j = ij - i * k;
    *reinterpret_cast<V*>(out + t * E) = *reinterpret_cast<const V*>(f + (nidx[bi * k + i] * dv + c) * E) -
                                         *reinterpret_cast<const V*>(f + (nidx[bi * k + j] * dv + c) * E);
  }
}

// T2: out[b,i,j,:] = x[nn[b,i],:] - x[nn[b,j],:]              _src/gp/tensors/numpy.py:61-69
template <typename T>
__global__ void pairwise_diffs_kernel(const T* f, int d, const int64_t* nidx, int64_t b, int k, T* out) {
  const int64_t n = b * k * (int64_t)k * d;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pr = t / d;
    const int c = (int)(t - pr * d);
    const int64_t bi = pr / ((int64_t)k * k);
    const int ij = (int)(pr - bi * k * k);
    const int i = ij / k, j = ij - i * k;
    out[t] = f[nidx[bi * k + i] * d + c] - f[nidx[bi * k + j] * d + c];
  }
}

// Sum over a group of 8 consecutive lanes (a row is walked by 8 lanes, 16 bytes each per step, so the
// rows of a wave are read as contiguous 128-byte pieces).
template <typename T>
__device__ __forceinline__ T group8_sum(T v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  return v;
}

// T1+T3: out[b,j] = metric(q - x_j): 8 lanes per output
template <typename T>
__global__ void crosswise_dists_kernel(const T* fq, const T* fn, int d, const int64_t* bidx, const int64_t* nidx,
                                       int64_t b, int k, int metric_id, T* out) {
  const int64_t n = b * k;
  const int sub = threadIdx.x & 7;
  const int64_t rows_per_pass = (int64_t)gridDim.x * blockDim.x / 8;
  // the 8 lanes of a group share t: a group is wholly inside or outside the loop (shuffles stay in-group)
  for (int64_t t = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / 8; t < n; t += rows_per_pass) {
    const int64_t bi = t / k;
    const T* q = fq + (bidx ? bidx[bi] : bi) * d;
    const T* x = fn + nidx[t] * d;
    T acc = T(0);
    for (int c = sub; c < d; c += 8) {
      const T df = q[c] - x[c];
      acc += df * df;
    }
    acc = group8_sum(acc);
    if (sub == 0) out[t] = metric_id == MGP_METRIC_L2 ? num<T>::sqrt(acc) : acc;
  }
}

// T2+T3: out[b,i,j] = metric(x_i - x_j).  Fallback form, one thread per output (any k, d).
template <typename T>
__global__ void pairwise_dists_kernel(const T* f, int d, const int64_t* nidx, int64_t b, int k, int metric_id,
                                      T* out) {
  const int64_t n = b * k * (int64_t)k;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t bi = t / ((int64_t)k * k);
    const int ij = (int)(t - bi * k * k);
    const int i = ij / k, j = ij - i * k;
    const T* xi = f + nidx[bi * k + i] * d;
    const T* xj = f + nidx[bi * k + j] * d;
    T acc = T(0);
    for (int c = 0; c < d; ++c) {
      const T df = xi[c] - xj[c];
      acc += df * df;
    }
    out[t] = metric_id == MGP_METRIC_L2 ? num<T>::sqrt(acc) : acc;
  }
}

// T2+T3, tiled: one wave per neighbourhood; its k rows are gathered once into LDS (consecutive lanes
// walk a row) and every output reads both of its rows from there -- the one-thread-per-output form
// above fetches 2 k d elements per neighbourhood ROW from the table (k-fold redundant).  Row stride
// d | 1 elements: lanes of consecutive j read consecutive rows, an odd stride keeps them on distinct banks.
template <typename T>
__global__ void pairwise_dists_tile_kernel(const T* f, int d, const int64_t* nidx, int64_t b, int k, int metric_id,
                                           T* out) {
  extern __shared__ __attribute__((aligned(16))) char smem_pd[];
  T* X = reinterpret_cast<T*>(smem_pd);
  const int xs = d | 1;
  const int lane = threadIdx.x;
  for (int64_t nb = blockIdx.x; nb < b; nb += gridDim.x) {
    __syncthreads();
    for (int t = lane; t < k * d; t += 64) {
      const int row = t / d, c = t - row * d;
      X[row * xs + c] = f[nidx[nb * k + row] * d + c];
    }
    __syncthreads();
    T* o = out + nb * (int64_t)k * k;
    for (int t = lane; t < k * k; t += 64) {
      const int i = t / k, j = t - i * k;
      const T* xi = X + i * xs;
      const T* xj = X + j * xs;
      T acc = T(0);
      for (int c = 0; c < d; ++c) {
        const T df = xi[c] - xj[c];
        acc += df * df;
      }
      o[t] = metric_id == MGP_METRIC_L2 ? num<T>::sqrt(acc) : acc;
    }
  }
}

template <int V> struct icn { static constexpr int value = V; };

// Register form of the same (k <= 64, d <= 64): one wave owns 64 / NP neighbourhoods, lane i keeps row i
// of its neighbourhood in registers (DG 16-byte groups, zero padded) and walks the rows j of the LDS tile
// as uniform 16-byte broadcasts; entry (j, i) is stored -- the matrix is symmetric -- so that the lanes of
// a wave write consecutive addresses.  k d / 4 broadcast reads per neighbourhood instead of 2 k^2 d / 64
// scalar ones per lane.
template <typename T, int NP, int DG>
__global__ __launch_bounds__(64) void pairwise_dists_wave_kernel(const T* f, int d, const int64_t* nidx, int64_t b, int k,
                                                                 int metric_id, T* out, int vec_ok) {
  constexpr int NH = 64 / NP;
  constexpr int E = 16 / (int)sizeof(T);
  using V = T __attribute__((ext_vector_type(E)));
  constexpr int XS = DG * E + E;  // odd number of 16-byte slots per row
  extern __shared__ __attribute__((aligned(16))) char smem_pw[];
  T* X = reinterpret_cast<T*>(smem_pw);
  __shared__ int64_t idxs[64];
  const int lane = threadIdx.x;
  const int h = NH == 1 ? 0 : lane / NP, i = lane & (NP - 1);
  const int64_t ntasks = (b + NH - 1) / NH;
  for (int64_t task = blockIdx.x; task < ntasks; task += gridDim.x) {
    __syncthreads();
    // gather: row offsets first (one per lane), then the lanes of the wave walk the 64 rows in 16-byte
    // pieces with all DG loads of a lane in flight before the first LDS store
    const int dv = (d + E - 1) / E;
    {
      const int64_t nbh = task * NH + h;
      idxs[lane] = (i < k && nbh < b) ? nidx[nbh * k + i] * (int64_t)d : (int64_t)-1;
    }
    __syncthreads();
    {
      V v[DG];
#pragma unroll
      for (int u = 0; u < DG; ++u) {
        const int t = lane + 64 * u;
        const int row = t / DG, c = t - row * DG;
        const int64_t off = idxs[row];
        v[u] = V(0);
        if (off >= 0 && c < dv) {
          const T* src = f + off + c * E;
          if (vec_ok) {
            v[u] = *reinterpret_cast<const V*>(src);
          } else {
#pragma unroll
            for (int e = 0; e < E; ++e)
              if (c * E + e < d) v[u][e] = src[e];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < DG; ++u) {
        const int t = lane + 64 * u;
        const int row = t / DG, c = t - row * DG;
        *reinterpret_cast<V*>(X + row * XS + c * E) = v[u];
      }
    }
    __syncthreads();
    V own[DG];
#pragma unroll
    for (int c = 0; c < DG; ++c) own[c] = *reinterpret_cast<const V*>(X + (h * NP + i) * XS + c * E);
    const int64_t nb = task * NH + h;
    T* o = out + nb * (int64_t)k * k;
#pragma unroll 4
    for (int j = 0; j < k; ++j) {
      const T* xj = X + (h * NP + j) * XS;
      V a0 = V(0), a1 = V(0);
#pragma unroll
      for (int c = 0; c < DG; c += 2) {
        const V d0 = own[c] - *reinterpret_cast<const V*>(xj + c * E);
        a0 = d0 * d0 + a0;
        if (c + 1 < DG) {
          const V d1 = own[c + 1] - *reinterpret_cast<const V*>(xj + (c + 1) * E);
          a1 = d1 * d1 + a1;
        }
      }
      const V a = a0 + a1;
      T acc = a[0];
#pragma unroll
      for (int e = 1; e < E; ++e) acc += a[e];
      if (i < k && nb < b) o[j * (int64_t)k + i] = metric_id == MGP_METRIC_L2 ? num<T>::sqrt(acc) : acc;
    }
  }
}

// T3 (+D2): out[n] = metric(diffs[n,:] / ls[:]): 8 lanes per row
template <typename T>
__global__ void reduce_diffs_kernel(const T* diffs, int64_t n, int d, const T* ls, int metric_id, T* out) {
  const int sub = threadIdx.x & 7;
  const int64_t rows_per_pass = (int64_t)gridDim.x * blockDim.x / 8;
  for (int64_t t = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / 8; t < n; t += rows_per_pass) {
    const T* x = diffs + t * d;
    T acc = T(0);
    for (int c = sub; c < d; c += 8) {
      const T df = ls ? x[c] / ls[c] : x[c];
      acc += df * df;
    }
    acc = group8_sum(acc);
    if (sub == 0) out[t] = metric_id == MGP_METRIC_L2 ? num<T>::sqrt(acc) : acc;
  }
}

// D1 + K1/K2
template <typename T>
__global__ void kernel_apply_kernel(const T* in, int64_t n, int kernel_id, T in_scale, T* out) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
    out[t] = kernel_eval<T>(kernel_id, in[t] * in_scale);
}

// K3: Matern kernel of general smoothness nu (_src/gp/kernels/numpy.py:34-43):
//     k(r) = 2^(1-nu) / Gamma(nu) * (sqrt(2 nu) r)^nu * K_nu(sqrt(2 nu) r),   r = 0 -> r = eps
// The modified Bessel function of the second kind K_nu(x), real nu > 0, in fp64 by the published
// algorithm of Temme (J. Comput. Phys. 19, 1975) as arranged in Numerical Recipes' bessik: with
// nu = mu + nl, |mu| <= 1/2, K_mu and K_(mu+1) come from Temme's series (x < 2) or from Steed's
// evaluation of the second continued fraction (x >= 2), then nl steps of the stable forward
// recurrence K_(m+1) = (2 m / x) K_m + K_(m-1).  Everything that depends on nu only -- the
// leading coefficient and the four gamma-function combinations of the series -- is computed once on
// the host (MaternGenConst).
struct MaternGenConst {
  double nu, mu, coef, gam1, gam2, gampl, gammi;
  int nl;
};

__device__ inline double bessel_k_scaled_matern(double x, const MaternGenConst c) {
  const double kEps = 1e-16;
  const double mu = c.mu, mu2 = mu * mu, xi2 = 2.0 / x;
  double rkmu, rk1;
  if (x < 2.0) {
    const double b = 0.5 * x;
    double d = -::log(b);
    double e = mu * d;
    const double fact2 = ::fabs(e) < kEps ? 1.0 : ::sinh(e) / e;
    const double pimu = 3.14159265358979323846 * mu;
    const double fact = ::fabs(pimu) < kEps ? 1.0 : pimu / ::sin(pimu);
    double ff = fact * (c.gam1 * ::cosh(e) + c.gam2 * fact2 * d);
    double sum = ff;
    e = ::exp(e);
    double p = 0.5 * e / c.gampl;
    double q = 0.5 / (e * c.gammi);
    double cc = 1.0;
    d = b * b;
    double sum1 = p;
    for (int i = 1; i <= 500; ++i) {
      ff = (i * ff + p + q) / (i * (double)i - mu2);
      cc *= d / i;
      p /= (i - mu);
      q /= (i + mu);
      const double del = cc * ff;
      sum += del;
      sum1 += cc * (p - i * ff);
      if (::fabs(del) < ::fabs(sum) * kEps) break;
    }
    rkmu = sum;
    rk1 = sum1 * xi2;
  } else {
    double b = 2.0 * (1.0 + x);
    double d = 1.0 / b;
    double h = d, delh = d;
    double q1 = 0.0, q2 = 1.0;
    const double a1 = 0.25 - mu2;
    double q = a1, cc = a1;
    double a = -a1;
    double s = 1.0 + q * delh;
    for (int i = 2; i <= 500; ++i) {
      a -= 2 * (i - 1);
      cc = -a * cc / i;
      const double qnew = (q1 - b * q2) / a;
      q1 = q2;
      q2 = qnew;
      q += cc * qnew;
      b += 2.0;
      d = 1.0 / (b + a * d);
      delh = (b * d - 1.0) * delh;
      h += delh;
      const double dels = q * delh;
      s += dels;
      if (::fabs(dels / s) < kEps) break;
    }
    h = a1 * h;
    rkmu = ::sqrt(3.14159265358979323846 / (2.0 * x)) * ::exp(-x) / s;
    rk1 = rkmu * (mu + x + 0.5 - h) / x;
  }
  for (int i = 1; i <= c.nl; ++i) {
    const double rktemp = (mu + i) * xi2 * rk1 + rkmu;
    rkmu = rk1;
    rk1 = rktemp;
  }
  return rkmu;
}

template <typename T>
__global__ void matern_gen_kernel(const T* in, int64_t n, double in_scale, const MaternGenConst c, T* out) {
  const double s2nu = ::sqrt(2.0 * c.nu);
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    double r = (double)in[t] * in_scale;
    if (r == 0.0) r = 2.220446049250313e-16;  // reference: zeros become eps (numpy.py:38)
    const double x = s2nu * r;
    double k = 0.0;
    if (x < 700.0) k = c.coef * ::pow(x, c.nu) * bessel_k_scaled_matern(x, c);
    out[t] = (T)k;
  }
}

// nu-only constants of the general Matern kernel (host).  gam1 = (1/G(1-mu) - 1/G(1+mu)) / (2 mu),
// gam2 = (1/G(1-mu) + 1/G(1+mu)) / 2; near mu = 0 gam1 comes from the even part of the series
// 1/Gamma(z) = sum c_k z^k (Abramowitz & Stegun 6.1.34) instead of the cancelling difference.
static MaternGenConst matern_gen_constants(double nu) {
  MaternGenConst c;
  c.nu = nu;
  c.nl = (int)(nu + 0.5);
  c.mu = nu - c.nl;
  c.gampl = 1.0 / std::tgamma(1.0 + c.mu);
  c.gammi = 1.0 / std::tgamma(1.0 - c.mu);
  c.gam2 = 0.5 * (c.gammi + c.gampl);
  const double m2 = c.mu * c.mu;
  if (std::fabs(c.mu) < 0.05)
    c.gam1 = -(0.5772156649015329 + m2 * (-0.0420026350340952 + m2 * (-0.0421977345555443 +
               m2 * (0.0072189432466630 + m2 * -0.0002152416741149))));
  else
    c.gam1 = (c.gammi - c.gampl) / (2.0 * c.mu);
  c.coef = std::exp((1.0 - nu) * 0.6931471805599453 - std::lgamma(nu));
  return c;
}

template <typename T>
int launch_matern_gen(const T* in, int64_t n, double in_scale, double nu, T* out, hipStream_t s) {
  if (!(nu > 0.0) || !std::isfinite(nu)) return MGP_EINVAL;
  if (n == 0) return MGP_OK;
  hipLaunchKernelGGL(matern_gen_kernel<T>, dim3(grid_1d(n)), dim3(kBlock), 0, s, in, n, in_scale,
                     matern_gen_constants(nu), out);
  MGP_HIP_CHECK_LAUNCH();
  return MGP_OK;
}
void matern_gen_constants_host(double nu, double* out7) {
  const MaternGenConst c = matern_gen_constants(nu);
  out7[0] = c.mu; out7[1] = (double)c.nl; out7[2] = c.coef; out7[3] = c.gam1; out7[4] = c.gam2; out7[5] = c.gampl;
  out7[6] = c.gammi;
}

// N1/N2
template <typename T>
__global__ void perturb_kernel(const T* Kin, int64_t b, int k, int noise_mode, T noise_scalar, const T* noise_dev,
                               T* out) {
  const int64_t n = b * k * (int64_t)k;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t bi = t / ((int64_t)k * k);
    const int ij = (int)(t - bi * k * k);
    const int i = ij / k, j = ij - i * k;
    T v = Kin[t];
    if (i == j) v += noise_mode == MGP_NOISE_SCALAR ? noise_scalar : noise_dev[bi * k + i];
    out[t] = v;
  }
}

// ---- fp64 reductions --------------------------------------------------------------------
// Deterministic two-stage form when the caller provides scratch (kReduceBlocks x N doubles): every
// workgroup stores its partial sums, a second one-workgroup launch adds them in workgroup order,
// so the result does not depend on arrival order and "sharded == serial" holds bit for bit for
// equal shard sizes.  Without scratch the partials meet in fp64 atomics (order-dependent rounding).

static const int kReduceBlocks = 1024;

template <int N>
__device__ inline void block_reduce_store(double (&v)[N], double* out, double* scratch) {
  __shared__ double red[N][kBlock / MGP_WAVE];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double s = wave_sum(v[i]);
    if (lane == 0) red[i][w] = s;
  }
  __syncthreads();
  if (threadIdx.x < N) {
    double s = 0;
    for (int j = 0; j < kBlock / MGP_WAVE; ++j) s += red[threadIdx.x][j];
    if (scratch) scratch[(size_t)blockIdx.x * N + threadIdx.x] = s;
    else atomicAdd(out + threadIdx.x, s);
  }
  __syncthreads();
}

// out[i] = sum over blocks of scratch[block][i], in block order (one workgroup)
__global__ void reduce_partials_kernel(const double* scratch, int blocks, int N, double* out) {
  __shared__ double red[kBlock];
  for (int i = 0; i < N; ++i) {
    double s = 0;
    for (int b = threadIdx.x; b < blocks; b += kBlock) s += scratch[(size_t)b * N + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[i] = red[0];
    __syncthreads();
  }
}

// _src/optimize/loss/numpy.py:22-117 in one pass
template <typename T>
__global__ void loss_sums_kernel(const T* pred, const T* target, const T* var, int64_t n, const double* scale_dev,
                                 double hd, double ld, double* out, double* scratch) {
  const double s = scale_dev ? *scale_dev : 1.0;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    const double r = (double)pred[t] - (double)target[t];
    const double r2 = r * r;
    acc[0] += r2;
    acc[2] += hd * hd * (::sqrt(1.0 + (r / hd) * (r / hd)) - 1.0);
    if (var) {
      const double v = (double)var[t];
      const double sv = s * v;
      acc[1] += r2 / sv + ::log(sv);
      acc[3] += 2.0 * ld * ld * (::sqrt(1.0 + r2 / (ld * ld * sv)) - 1.0) + ::log(sv);
      acc[4] += r2 / v;
      acc[5] += ::log(v);
    }
  }
  block_reduce_store<6>(acc, out, scratch);
}

// LOOCV partial sums of one shard (optimize/loss.py:159-168 with scale/numpy.py:11-18 folded in):
// [sum r^2/v, sum log v, sum r^2, n, sum pseudo-Huber(r), sum y^T K^-1 y], r = mean - y(batch row), as the fixed
// reduction tree of mgp_loocv_tree.h.  The fused wave kernels walk that tree themselves (one launch per evaluation);
// these three kernels walk it behind any other kernel family: one wave per leaf / block, the same device functions.
template <typename T>
__global__ void loocv_level1_kernel(LoocvTree tr, const T* mean, const T* var, const T* yk, const int64_t* batch_idx, int64_t b) {
  const int w = blockIdx.x * (kBlock / MGP_WAVE) + (threadIdx.x >> 6);
  if (w >= tr.grid) return;
  const int lane = threadIdx.x & 63;
  double t[6];
  tree_level1<T>(tr, mean, var, yk, batch_idx, b, w, lane, t);
  if (lane == 0)
    for (int i = 0; i < 6; ++i) tr.part1[6 * w + i] = t[i];
}
__global__ void loocv_level2_kernel(LoocvTree tr) {
  const int j2 = blockIdx.x * (kBlock / MGP_WAVE) + (threadIdx.x >> 6);
  if (j2 >= (int)tree_nb2(tr.grid)) return;
  const int lane = threadIdx.x & 63;
  double t[6];
  tree_level2<false>(tr, j2, lane, t);
  if (lane == 0)
    for (int i = 0; i < 6; ++i) tr.part2[6 * j2 + i] = t[i];
}
__global__ void loocv_level3_kernel(LoocvTree tr) {
  double t[6];
  tree_level3<false>(tr, (int)threadIdx.x, t);
  tree_result(tr.out, t, (int)threadIdx.x);  // (`out` may be mapped host memory: the count last)
}

template <typename T>
__global__ void column_sums_kernel(const T* x, int64_t n, int R, double* out, double* scratch) {
  // one column at a time (R is small); rows strided over the whole grid
  for (int r = 0; r < R; ++r) {
    double acc[1] = {0};
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
      acc[0] += (double)x[t * R + r];
    block_reduce_store<1>(acc, out + r, scratch ? scratch + (size_t)r * gridDim.x : nullptr);
  }
}

// prepared table: out row = [features d | responses R | zero pad] at `stride` bytes
template <typename T>
__global__ void table_pack_kernel(const T* feat, const T* targets, int64_t n, int d, int R, T* out, int64_t se) {
  const int64_t total = n * se;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / se;
    const int c = (int)(t - row * se);
    T v = T(0);
    if (c < d) v = feat[row * d + c];
    else if (c < d + R && targets) v = targets[row * R + (c - d)];
    out[t] = v;
  }
}

template <typename T>
int launch_crosswise_diffs(const T* fq, const T* fn, int d, const int64_t* bi, const int64_t* ni, int64_t b, int k,
                           T* out, hipStream_t s) {
  if (b * k * (int64_t)d == 0) return MGP_OK;
  constexpr int E = 16 / (int)sizeof(T);
  if (d % E == 0 && (reinterpret_cast<uintptr_t>(fq) | reinterpret_cast<uintptr_t>(fn) | reinterpret_cast<uintptr_t>(out)) % 16 == 0) {
    hipLaunchKernelGGL(crosswise_diffs_vec_kernel<T>, dim3(grid_1d(b * k * (d / E))), dim3(kBlock), 0, s, fq, fn, d / E, bi,
                       ni, b, k, out);
    MGP_HIP_CHECK_LAUNCH();
    return MGP_OK;
  }
  hipLaunchKernelGGL(crosswise_diffs_kernel<T>, dim3(grid_1d(b * k * d)), dim3(kBlock), 0, s, fq, fn, d, bi, ni, b,
                     k, out);
  MGP_HIP_CHECK_LAUNCH();
  return MGP_OK;
}
template <typename T>
int launch_pairwise_diffs(const T* f, int d, const int64_t* ni, int64_t b, int k, T* out, hipStream_t s) {
  if (b * k * (int64_t)d == 0) return MGP_OK;
  constexpr int E = 16 / (int)sizeof(T);
  if (d % E == 0 && (reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(out)) % 16 == 0) {
    hipLaunchKernelGGL(pairwise_diffs_vec_kernel<T>, dim3(grid_1d(b * k * k * (d / E))), dim3(kBlock), 0, s, f, d / E, ni, b,
                       k, out);
    MGP_HIP_CHECK_LAUNCH();
    return MGP_OK;
  }
  hipLaunchKernelGGL(pairwise_diffs_kernel<T>, dim3(grid_1d(b * k * k * d)), dim3(kBlock), 0, s, f, d, ni, b, k,
                     out);
  MGP_HIP_CHECK_LAUNCH();
  return MGP_OK;
}
template <typename T>
int launch_crosswise_dists(const T* fq, const T* fn, int d, const int64_t* bi, const int64_t* ni, int64_t b, int k,
                           int metric, T* out, hipStream_t s) {
  if (b * k == 0) return MGP_OK;
  hipLaunchKernelGGL(crosswise_dists_kernel<T>, dim3(grid_1d(b * k * 8)), dim3(kBlock), 0, s, fq, fn, d, bi, ni, b, k,
                     metric, out);
  MGP_HIP_CHECK_LAUNCH();
  return MGP_OK;
}
template <typename T>
int launch_pairwise_dists(const T* f, int d, const int64_t* ni, int64_t b, int k, int metric, T* out,
                          hipStream_t s) {
  if (b * k == 0) return MGP_OK;
  if (k <= 64 && d <= 64) {
    constexpr int E = 16 / (int)sizeof(T);
    const int vec_ok = (d % E == 0) && (reinterpret_cast<uintptr_t>(f) % 16 == 0);
    const int dg = (d + E - 1) / E;  // 16-byte groups per row
    auto go = [&](auto npc, auto dgc) {
      constexpr int NP = decltype(npc)::value, DG = decltype(dgc)::value;
      const size_t lds = (size_t)64 * (DG * E + E) * sizeof(T);
      const int64_t ntasks = (b + 64 / NP - 1) / (64 / NP);
      int64_t g = 256LL * (int64_t)((160 * 1024) / (((lds + 1279) / 1280) * 1280));
      if (g > 256LL * 16) g = 256LL * 16;
      if (g > ntasks) g = ntasks;
      hipLaunchKernelGGL((pairwise_dists_wave_kernel<T, NP, DG>), dim3((unsigned)g), dim3(64), lds, s, f, d, ni, b, k, metric,
                         out, vec_ok);
    };
    constexpr int G16 = 16 / E * 1;  // groups per 16 features: 4 (fp32) / 8 (fp64)
    const int steps = (dg + G16 - 1) / G16;  // 1 .. 4 blocks of 16 features
    if (k <= 32) {
      if (steps == 1) go(icn<32>{}, icn<G16>{});
      else if (steps == 2) go(icn<32>{}, icn<2 * G16>{});
      else if (steps == 3) go(icn<32>{}, icn<3 * G16>{});
      else go(icn<32>{}, icn<4 * G16>{});
    } else {
      if (steps == 1) go(icn<64>{}, icn<G16>{});
      else if (steps == 2) go(icn<64>{}, icn<2 * G16>{});
      else if (steps == 3) go(icn<64>{}, icn<3 * G16>{});
      else go(icn<64>{}, icn<4 * G16>{});
    }
    MGP_HIP_CHECK_LAUNCH();
    return MGP_OK;
  }
  const size_t lds = (size_t)k * (d | 1) * sizeof(T);
  if (lds <= 40 * 1024) {  // the neighbourhood's rows fit LDS: gather them once
    const int64_t cap = 256LL * (lds ? (int64_t)((160 * 1024) / (((lds + 1279) / 1280) * 1280)) : 32);
    int64_t g = cap < 256LL * 32 ? cap : 256LL * 32;
    if (g > b) g = b;
    hipLaunchKernelGGL(pairwise_dists_tile_kernel<T>, dim3((unsigned)g), dim3(64), lds, s, f, d, ni, b, k, metric, out);
    MGP_HIP_CHECK_LAUNCH();
    return MGP_OK;
  }
  hipLaunchKernelGGL(pairwise_dists_kernel<T>, dim3(grid_1d(b * k * k)), dim3(kBlock), 0, s, f, d, ni, b, k, metric,
                     out);
  MGP_HIP_CHECK_LAUNCH();
  return MGP_OK;
}
template <typename T>
int launch_reduce_diffs(const T* diffs, int64_t n, int d, const T* ls, int metric, T* out, hipStream_t s) {
  if (n == 0) return MGP_OK;
  hipLaunchKernelGGL(reduce_diffs_kernel<T>, dim3(grid_1d(n * 8)), dim3(kBlock), 0, s, diffs, n, d, ls, metric, out);
  MGP_HIP_CHECK_LAUNCH();
  return MGP_OK;
}
template <typename T>
int launch_kernel_apply(const T* in, int64_t n, int kernel_id, double scale, T* out, hipStream_t s) {
  if (n == 0) return MGP_OK;
  hipLaunchKernelGGL(kernel_apply_kernel<T>, dim3(grid_1d(n)), dim3(kBlock), 0, s, in, n, kernel_id, (T)scale, out);
  MGP_HIP_CHECK_LAUNCH();
  return MGP_OK;
}
template <typename T>
int launch_perturb(const T* Kin, int64_t b, int k, int mode, double eps, const T* nd, T* out, hipStream_t s) {
  if (b * k == 0) return MGP_OK;
  hipLaunchKernelGGL(perturb_kernel<T>, dim3(grid_1d(b * k * k)), dim3(kBlock), 0, s, Kin, b, k, mode, (T)eps, nd,
                     out);
  MGP_HIP_CHECK_LAUNCH();
  return MGP_OK;
}
template <typename T>
int launch_loss_sums(const T* pred, const T* target, const T* var, int64_t n, const double* scale_dev, double hd,
                     double ld, double* out, double* scratch, hipStream_t s) {
  hipError_t e = hipMemsetAsync(out, 0, 6 * sizeof(double), s);
  if (e != hipSuccess) return -(1000 + (int)e);
  if (n == 0) return MGP_OK;
  int g = grid_1d(n);
  if (g > kReduceBlocks) g = kReduceBlocks;
  hipLaunchKernelGGL(loss_sums_kernel<T>, dim3(g), dim3(kBlock), 0, s, pred, target, var, n, scale_dev, hd, ld, out,
                     scratch);
  MGP_HIP_CHECK_LAUNCH();
  if (scratch) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(kBlock), 0, s, scratch, g, 6, out);
    MGP_HIP_CHECK_LAUNCH();
  }
  return MGP_OK;
}
template <typename T>
int launch_column_sums(const T* x, int64_t n, int R, double* out, double* scratch, hipStream_t s) {
  hipError_t e = hipMemsetAsync(out, 0, (size_t)R * sizeof(double), s);
  if (e != hipSuccess) return -(1000 + (int)e);
  if (n == 0 || R == 0) return MGP_OK;
  int g = grid_1d(n);
  if (g > kReduceBlocks) g = kReduceBlocks;
  if (scratch && (int64_t)g * R > (int64_t)kReduceBlocks * 6) g = kReduceBlocks * 6 / R;  // scratch capacity
  if (g < 1) return MGP_EUNSUPPORTED;
  hipLaunchKernelGGL(column_sums_kernel<T>, dim3(g), dim3(kBlock), 0, s, x, n, R, out, scratch);
  MGP_HIP_CHECK_LAUNCH();
  if (scratch) {
    // scratch holds R runs of g partials: column r at scratch[r * g + block]
    for (int r = 0; r < R; ++r) {
      hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(kBlock), 0, s, scratch + (size_t)r * g, g, 1, out + r);
      MGP_HIP_CHECK_LAUNCH();
    }
  }
  return MGP_OK;
}
LoocvTree loocv_tree_layout(void* scratch, double* out, const void* resp, int64_t resp_stride, double huber_delta) {
  char* base = static_cast<char*>(scratch);
  LoocvTree tr;
  tr.out = out;
  tr.ctrl = reinterpret_cast<unsigned*>(base);
  tr.cnt2 = reinterpret_cast<unsigned*>(base + tree_off_cnt2());
  tr.part1 = reinterpret_cast<double*>(base + tree_off_part1());
  tr.part2 = reinterpret_cast<double*>(base + tree_off_part2());
  tr.resp = static_cast<const char*>(resp);
  tr.resp_stride = resp_stride;
  tr.huber_delta = huber_delta;
  return tr;
}
// (grid, nh): the leaves -- those of the fused launch whose outputs these are, for equal bits with its own walk; 0, 0:
// the canonical ones (kTreeCanonGrid leaves of one neighbourhood per task)
template <typename T>
int launch_loocv_tree(const LoocvTree& tr0, int grid, int nh, const T* mean, const T* var, const T* yk, const int64_t* batch_idx,
                      int64_t b, hipStream_t s) {
  LoocvTree tr = tr0;
  if (!tr.out || !tr.part1 || !tr.part2) return MGP_EINVAL;
  if (grid == 0 && nh == 0) grid = kTreeCanonGrid, nh = 1;
  if (grid < 8 || grid % 8 != 0 || grid > kTreeMaxLeaves || (nh != 1 && nh != 2 && nh != 4)) return MGP_EINVAL;
  tr.grid = grid;
  tr.nh = nh;
  if (b == 0) {
    hipError_t e = hipMemsetAsync(tr.out, 0, 6 * sizeof(double), s);
    return e == hipSuccess ? MGP_OK : -(1000 + (int)e);
  }
  constexpr int W = kBlock / MGP_WAVE;
  hipLaunchKernelGGL(loocv_level1_kernel<T>, dim3((unsigned)((grid + W - 1) / W)), dim3(kBlock), 0, s, tr, mean, var, yk, batch_idx, b);
  MGP_HIP_CHECK_LAUNCH();
  hipLaunchKernelGGL(loocv_level2_kernel, dim3((unsigned)((tree_nb2(grid) + W - 1) / W)), dim3(kBlock), 0, s, tr);
  MGP_HIP_CHECK_LAUNCH();
  hipLaunchKernelGGL(loocv_level3_kernel, dim3(1), dim3(MGP_WAVE), 0, s, tr);
  MGP_HIP_CHECK_LAUNCH();
  return MGP_OK;
}
int reduce_scratch_doubles() { return kReduceBlocks * 6; }

// 16 bytes of output per lane (rows of whole 16-byte groups of features, features 16-byte aligned):
// a group is features, responses (+ zero pad), or zero pad
template <typename T>
__global__ void table_pack_vec_kernel(const T* feat, const T* targets, int64_t n, int d, int R, T* out, int spr) {
  constexpr int E = 16 / (int)sizeof(T);
  using V = T __attribute__((ext_vector_type(E)));
  const int dv = d / E;
  const int64_t total = n * spr;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / spr;
    const int c = (int)(t - row * spr);
    V v = V(0);
    if (c < dv) {
      v = *reinterpret_cast<const V*>(feat + row * d + c * E);
    } else if (targets) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int r = (c - dv) * E + e;
        if (r < R) v[e] = targets[row * R + r];
      }
    }
    *reinterpret_cast<V*>(out + t * E) = v;
  }
}

template <typename T>
int launch_table_pack(const T* feat, const T* targets, int64_t n, int d, int R, void* out, int64_t stride,
                      hipStream_t s) {
  if (n == 0) return MGP_OK;
  if (stride % (int64_t)sizeof(T) != 0 || stride < (int64_t)((d + R) * sizeof(T))) return MGP_EINVAL;
  const int64_t se = stride / (int64_t)sizeof(T);
  constexpr int E = 16 / (int)sizeof(T);
  if (d % E == 0 && stride % 16 == 0 && reinterpret_cast<uintptr_t>(feat) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0) {
    const int spr = (int)(stride / 16);
    hipLaunchKernelGGL(table_pack_vec_kernel<T>, dim3(grid_1d(n * spr)), dim3(kBlock), 0, s, feat, targets, n, d, R,
                       static_cast<T*>(out), spr);
    MGP_HIP_CHECK_LAUNCH();
    return MGP_OK;
  }
  hipLaunchKernelGGL(table_pack_kernel<T>, dim3(grid_1d(n * se)), dim3(kBlock), 0, s, feat, targets, n, d, R,
                     static_cast<T*>(out), se);
  MGP_HIP_CHECK_LAUNCH();
  return MGP_OK;
}

#define MGP_INSTANTIATE(T)                                                                                         \
  template int launch_crosswise_diffs<T>(const T*, const T*, int, const int64_t*, const int64_t*, int64_t, int, T*, \
                                         hipStream_t);                                                             \
  template int launch_pairwise_diffs<T>(const T*, int, const int64_t*, int64_t, int, T*, hipStream_t);             \
  template int launch_crosswise_dists<T>(const T*, const T*, int, const int64_t*, const int64_t*, int64_t, int, int, \
                                         T*, hipStream_t);                                                         \
  template int launch_pairwise_dists<T>(const T*, int, const int64_t*, int64_t, int, int, T*, hipStream_t);        \
  template int launch_reduce_diffs<T>(const T*, int64_t, int, const T*, int, T*, hipStream_t);                     \
  template int launch_kernel_apply<T>(const T*, int64_t, int, double, T*, hipStream_t);                            \
  template int launch_perturb<T>(const T*, int64_t, int, int, double, const T*, T*, hipStream_t);                  \
  template int launch_loss_sums<T>(const T*, const T*, const T*, int64_t, const double*, double, double, double*,  \
                                   double*, hipStream_t);                                                          \
  template int launch_column_sums<T>(const T*, int64_t, int, double*, double*, hipStream_t);                        \
  template int launch_loocv_tree<T>(const LoocvTree&, int, int, const T*, const T*, const T*, const int64_t*, int64_t, hipStream_t); \
  template int launch_table_pack<T>(const T*, const T*, int64_t, int, int, void*, int64_t, hipStream_t);          \
  template int launch_matern_gen<T>(const T*, int64_t, double, double, T*, hipStream_t);
MGP_INSTANTIATE(float)
MGP_INSTANTIATE(double)

}  // namespace mgp

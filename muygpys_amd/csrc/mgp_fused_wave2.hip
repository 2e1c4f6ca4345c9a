// Two-rows-per-lane variant of the register-resident fused kernel, static shapes only
// (float, nn_count 30, one response, feature_count 40: BASELINE configs 2/3).
//
// Why: in mgp_fused_wave.hip (one slot per lane, two neighbourhoods per wave) the LDS pipe is
// as loaded as the VALUs -- every partner row read in the distance phase feeds ONE pair, and
// every column broadcast of the factorisation feeds 32 rows.  Here a neighbourhood's 32 slots
// sit on 16 lanes (lane l owns slots l and l+16) and a wave carries FOUR neighbourhoods:
//   * distances: a partner "super point" {m, m+16} is read once and paired with both own rows
//     (4 pairs per 2 row reads instead of 1 per read)                    -> LDS reads x 0.56
//   * factorisation: one broadcast read serves the rows of 4 neighbourhoods, and the low rows
//     (slots < 16) only ever need columns 0..15 and die after step 14   -> LDS x 0.57, FMAs x 0.85
// The algebra, slot roles and phases are those of mgp_fused_wave.hip (see its header):
// slots 0..29 neighbours, 30 the query, 31 the response row; Schur block in lanes 14/15.
#include "mgp_args.h"

namespace mgp {

typedef float w2v4 __attribute__((ext_vector_type(4)));
typedef float w2v2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ w2v2 w2_pk_sub(w2v2 x, w2v2 y) {
  w2v2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ void w2_acc(w2v2& a, const w2v4& x, const w2v4& y) {
  const w2v2 d0 = w2_pk_sub(x.xy, y.xy), d1 = w2_pk_sub(x.zw, y.zw);
  a = d0 * d0 + a;
  a = d1 * d1 + a;
}
__device__ __forceinline__ float w2_exp_neg(float t) {
  const float hi = -t * 1.44269502162933349609375f;
  const float lo = __builtin_fmaf(-t, 1.44269502162933349609375f, -hi) - t * 1.925963033500011e-08f;
  const float e = __builtin_amdgcn_exp2f(hi);
  return __builtin_fmaf(e * lo, 0.693147180559945f, e);
}
template <int KID, int MID>
__device__ __forceinline__ float w2_cov(float acc, float post_scale) {
  const float x = (MID == MGP_METRIC_L2 ? __builtin_amdgcn_sqrtf(acc) : acc) * post_scale;
  if (KID == MGP_KERNEL_RBF) return w2_exp_neg(x * 0.5f);
  if (KID == MGP_KERNEL_MATERN_05) return w2_exp_neg(x);
  if (KID == MGP_KERNEL_MATERN_15) {
    const float t = x * 1.7320508075688772935f;
    return (1.0f + t) * w2_exp_neg(t);
  }
  if (KID == MGP_KERNEL_MATERN_25) {
    const float t = x * 2.2360679774997896964f;
    return (1.0f + t + t * t * (1.0f / 3.0f)) * w2_exp_neg(t);
  }
  return w2_exp_neg(x * x * 0.5f);
}
__device__ __forceinline__ void w2_glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct Wave2Geom {
  int64_t ntasks;
  int mask;
};

template <int KF, int DF>
__global__ __launch_bounds__(64, 2) void fused_wave2_kernel(FusedArgs a, Wave2Geom g) {
  static_assert(KF == 30 && DF % 8 == 0 && DF <= 64, "static shape: 30 neighbours + query + response = 32 slots");
  constexpr int NG = 4;             // neighbourhoods per wave
  constexpr int SPR = DF / 4 + 1;   // 16-byte slots per staged row (last one padding; odd)
  constexpr int XS = SPR * 4;       // row stride of the feature tile (floats)
  constexpr int KS = 36;            // row stride of the exchange matrix (9 slots: odd)
  constexpr int QS = 30, YS = 31;   // query slot, response slot
  constexpr int NACC = 31;          // pairs per lane
  static_assert((NG * 32 * SPR) % 64 == 0, "tile = whole 1-KiB pieces");
  constexpr int NGL = NG * 32 * SPR / 64;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = reinterpret_cast<float*>(smem);     // NG*32 rows x XS; later NG x 32 x KS exchange
  float* colbuf = tile + NG * 32 * XS;              // NG x 32
  int* idx32 = reinterpret_cast<int*>(colbuf);      // NG x 32, only live while the gather is issued

  const float* feat_q = static_cast<const float*>(a.feat_q);
  const float* feat_nn = static_cast<const float*>(a.feat_nn);
  const float* targets = static_cast<const float*>(a.targets);
  const float* noise_dev = static_cast<const float*>(a.noise_dev);
  float* mean = static_cast<float*>(a.mean);
  float* var = static_cast<float*>(a.var);
  float* yk = static_cast<float*>(a.ykinvy);
  const float ls0 = static_cast<const float*>(a.length_scale)[0];
  const float post_scale = a.metric_id == MGP_METRIC_L2 ? 1.0f / ls0 : 1.0f / (ls0 * ls0);

  const int64_t ntasks = g.ntasks;
  const int64_t per_xcd = (ntasks + 7) / 8;
  const int xcd = blockIdx.x & 7;
  const int64_t t_hi = (xcd + 1) * per_xcd;
  const int64_t t_end = t_hi < ntasks ? t_hi : ntasks;
  const int64_t t_step = gridDim.x >> 3;
  const int64_t task0 = xcd * per_xcd + (blockIdx.x >> 3);

  // one load per owned slot, branch-free (see mgp_fused_wave.hip): raw index values
  auto load_idx = [&](int64_t task, int gi, int l, int64_t& ra, int64_t& rb) {
    int64_t n = task * NG + gi;
    n = n < a.b ? n : a.b - 1;
    const int64_t* row = a.nn_idx + n * KF;
    ra = row[l];
    const int64_t* pb = row + (l < 14 ? 16 + l : 0);
    if (a.batch_idx != nullptr && l == 14) pb = a.batch_idx + n;
    rb = *pb;
  };
  auto fix_b = [&](int64_t rb, int64_t task, int gi, int l) -> int64_t {
    int64_t n = task * NG + gi;
    n = n < a.b ? n : a.b - 1;
    if (l < 14) return rb;
    if (l == 14) return a.batch_idx != nullptr ? rb : n;
    return 0;
  };
  float pre_ya = 0.f, pre_yb = 0.f, pre_ea = 0.f, pre_eb = 0.f;
  auto pipe_issue = [&](int64_t task_n, int64_t ia, int64_t ib, int lane_) {
    const int gi = lane_ >> 4, l = lane_ & 15;
    idx32[gi * 32 + l] = (int)ia;
    idx32[gi * 32 + 16 + l] = (int)ib;
    __syncthreads();
    if (g.mask & 1) {
#pragma unroll
      for (int n = 0; n < NGL; ++n) {
        const int sigma = 64 * n + lane_;
        const int row = sigma / SPR;              // 0 .. NG*32-1
        int c = sigma - row * SPR;
        c = c < DF / 4 ? c : DF / 4 - 1;          // padding slot: re-read the last data slot
        const float* base = (row & 31) == QS ? feat_q : feat_nn;
        w2_glds16(base + (int64_t)idx32[row] * DF + c * 4, reinterpret_cast<char*>(tile) + n * 1024);
      }
    }
    pre_ya = targets[ia];
    pre_yb = targets[ib];
    pre_ea = pre_eb = (float)a.noise_scalar;
    if (a.noise_mode != MGP_NOISE_SCALAR) {
      int64_t n = task_n * NG + gi;
      n = n < a.b ? n : a.b - 1;
      const float* pa = a.noise_mode == MGP_NOISE_TABLE ? noise_dev + ia : noise_dev + n * KF + l;
      const float* pb = a.noise_mode == MGP_NOISE_TABLE ? noise_dev + ib : noise_dev + n * KF + (l < 14 ? 16 + l : 0);
      pre_ea = *pa;
      pre_eb = *pb;
    }
  };

  int64_t nxa = 0, nxb = 0;
  if (task0 < t_end) {
    load_idx(task0, threadIdx.x >> 4, threadIdx.x & 15, nxa, nxb);
    pipe_issue(task0, nxa, fix_b(nxb, task0, threadIdx.x >> 4, threadIdx.x & 15), threadIdx.x);
    if (task0 + t_step < t_end) load_idx(task0 + t_step, threadIdx.x >> 4, threadIdx.x & 15, nxa, nxb);
  }

  for (int64_t task = task0; task < t_end; task += t_step) {
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));  // keep per-lane addresses/masks out of LICM (register pressure)
    const int gi = lane >> 4, l = lane & 15;
    float* Xg = tile + gi * 32 * XS;
    float* Kg = tile + gi * 32 * KS;
    float* colg = colbuf + gi * 32;
    const int64_t n_raw = task * NG + gi;
    const bool live = n_raw < a.b;

    const float ya = pre_ya, yb = l < 14 ? pre_yb : 0.f;
    const float ea = pre_ea, eb = pre_eb;

    // ---- distances ---------------------------------------------------------------------
    __syncthreads();  // the tile requested during the previous factorisation has landed
    w2v2 acc[NACC];
#pragma unroll
    for (int s = 0; s < NACC; ++s) acc[s] = w2v2(0.f);
    {
      const float* xa = Xg + l * XS;
      const float* xb = Xg + (l + 16) * XS;
      const bool lt = l < 8;
      const int m8 = (l + 8) & 15;
      const float* x8 = Xg + (lt ? l : l + 16) * XS;         // sp = 8: first pair's own row
      const float* y81 = Xg + m8 * XS;                        //         its partner
      const float* y82 = Xg + (lt ? m8 : m8 + 16) * XS;       //         second pair's partner (own = b)
      if (g.mask & 2) {
#pragma unroll
        for (int c0 = 0; c0 < DF; c0 += 8) {
          const w2v4 a0 = *reinterpret_cast<const w2v4*>(xa + c0), a1 = *reinterpret_cast<const w2v4*>(xa + c0 + 4);
          const w2v4 b0 = *reinterpret_cast<const w2v4*>(xb + c0), b1 = *reinterpret_cast<const w2v4*>(xb + c0 + 4);
          w2_acc(acc[30], a0, b0);
          w2_acc(acc[30], a1, b1);
#pragma unroll
          for (int sp = 1; sp <= 7; ++sp) {
            const float* xm = Xg + ((l + sp) & 15) * XS + c0;
            const w2v4 m0 = *reinterpret_cast<const w2v4*>(xm), m1 = *reinterpret_cast<const w2v4*>(xm + 4);
            const w2v4 n0 = *reinterpret_cast<const w2v4*>(xm + 16 * XS), n1 = *reinterpret_cast<const w2v4*>(xm + 16 * XS + 4);
            w2_acc(acc[4 * (sp - 1) + 0], a0, m0);
            w2_acc(acc[4 * (sp - 1) + 0], a1, m1);
            w2_acc(acc[4 * (sp - 1) + 1], a0, n0);
            w2_acc(acc[4 * (sp - 1) + 1], a1, n1);
            w2_acc(acc[4 * (sp - 1) + 2], b0, m0);
            w2_acc(acc[4 * (sp - 1) + 2], b1, m1);
            w2_acc(acc[4 * (sp - 1) + 3], b0, n0);
            w2_acc(acc[4 * (sp - 1) + 3], b1, n1);
          }
          {
            const w2v4 p0 = *reinterpret_cast<const w2v4*>(x8 + c0), p1 = *reinterpret_cast<const w2v4*>(x8 + c0 + 4);
            const w2v4 q0 = *reinterpret_cast<const w2v4*>(y81 + c0), q1 = *reinterpret_cast<const w2v4*>(y81 + c0 + 4);
            const w2v4 r0 = *reinterpret_cast<const w2v4*>(y82 + c0), r1 = *reinterpret_cast<const w2v4*>(y82 + c0 + 4);
            w2_acc(acc[28], p0, q0);
            w2_acc(acc[28], p1, q1);
            w2_acc(acc[29], b0, r0);
            w2_acc(acc[29], b1, r1);
          }
        }
      }
    }

    // ---- covariances -> exchange matrix -> two rows per lane ------------------------------
    __syncthreads();  // all reads of the feature tile are done (the exchange matrix aliases it)
    {
      int l3 = l;
      asm volatile("" : "+v"(l3));
      float* Kg3 = tile + (lane >> 4) * 32 * KS;
      float kv[NACC];
      if (g.mask & 4) {
        auto eval = [&](auto kid, auto mid) {
          constexpr int KID = decltype(kid)::value, MID = decltype(mid)::value;
#pragma unroll
          for (int s = 0; s < NACC; ++s) kv[s] = w2_cov<KID, MID>(acc[s].x + acc[s].y, post_scale);
        };
#define W2_CASE(K_)                                                                         \
  case K_:                                                                                  \
    if (a.metric_id == MGP_METRIC_L2) eval(std::integral_constant<int, K_>{}, std::integral_constant<int, MGP_METRIC_L2>{}); \
    else eval(std::integral_constant<int, K_>{}, std::integral_constant<int, MGP_METRIC_F2>{});                              \
    break;
        switch (a.kernel_id) {
          W2_CASE(MGP_KERNEL_RBF)
          W2_CASE(MGP_KERNEL_MATERN_05)
          W2_CASE(MGP_KERNEL_MATERN_15)
          W2_CASE(MGP_KERNEL_MATERN_25)
          default:
            if (a.metric_id == MGP_METRIC_L2) eval(std::integral_constant<int, MGP_KERNEL_MATERN_INF>{}, std::integral_constant<int, MGP_METRIC_L2>{});
            else eval(std::integral_constant<int, MGP_KERNEL_MATERN_INF>{}, std::integral_constant<int, MGP_METRIC_F2>{});
        }
#undef W2_CASE
        const int dump = YS * KS + 32;  // padding columns of the last row
        auto put = [&](int x, int y, float v) {
          const int hi = max(x, y), lo = min(x, y);
          Kg3[hi < YS ? hi * KS + lo : dump] = v;  // pairs with the response slot carry no distance
        };
#pragma unroll
        for (int sp = 1; sp <= 7; ++sp) {
          const int m = (l3 + sp) & 15;
          put(l3, m, kv[4 * (sp - 1) + 0]);
          put(l3, m + 16, kv[4 * (sp - 1) + 1]);
          put(l3 + 16, m, kv[4 * (sp - 1) + 2]);
          put(l3 + 16, m + 16, kv[4 * (sp - 1) + 3]);
        }
        const bool lt3 = l3 < 8;
        const int m83 = (l3 + 8) & 15;
        put(lt3 ? l3 : l3 + 16, m83, kv[28]);
        put(l3 + 16, lt3 ? m83 : m83 + 16, kv[29]);
        put(l3, l3 + 16, kv[30]);
      }
      // diagonal (kernel(0) = 1, + nugget for neighbour slots; Kout = 1 for the query; 0 for the
      // response slot) and the response row
      Kg3[l3 * KS + l3] = 1.0f + ea;
      const int sb = l3 + 16;
      Kg3[sb * KS + sb] = sb < QS ? 1.0f + eb : (sb == QS ? 1.0f : 0.0f);
      Kg3[YS * KS + l3] = ya;
      if (sb < YS) Kg3[YS * KS + sb] = yb;  // (the query column of the response row starts at 0)
    }
    __syncthreads();
    w2v4 Aa[4], Ab[8];
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) Aa[c4] = *reinterpret_cast<const w2v4*>(Kg + l * KS + c4 * 4);
#pragma unroll
    for (int c4 = 0; c4 < 8; ++c4) Ab[c4] = *reinterpret_cast<const w2v4*>(Kg + (l + 16) * KS + c4 * 4);

    // request the next task's tile: the region is free now, the factorisation hides the latency
    if (task + t_step < t_end) {
      pipe_issue(task + t_step, nxa, fix_b(nxb, task + t_step, gi, l), lane);
      if (task + 2 * t_step < t_end) load_idx(task + 2 * t_step, gi, l, nxa, nxb);
    }

    // ---- factorisation: rows a (slots < 16: columns 0..15, dead after step 14) and b ---------
    // Look-ahead: step j first updates only the 16-byte groups that hold column j+1, posts that
    // column and issues the broadcast reads of step j+1; the rest of step j's trailing update
    // runs while the LDS round trip is in flight.  (With 7 waves per CU the round trip is what
    // the factorisation would otherwise wait for, step after step.)
    bool bad = false;
    w2v4 colc[8];
    if (g.mask & 8) {
      colg[l] = Aa[0][0];
      colg[16 + l] = Ab[0][0];
#pragma unroll
      for (int c4 = 0; c4 < 8; ++c4) colc[c4] = *reinterpret_cast<const w2v4*>(colg + c4 * 4);
    }
#pragma unroll
    for (int j = 0; j < KF; ++j) {
      if (g.mask & 8) {
        const float p = colc[j / 4][j % 4];
        bad = bad || !(p > 0.0f);
        const float r = __builtin_amdgcn_rcpf(p);
        const w2v4 nta = w2v4(-(j < 16 ? Aa[(j < 16 ? j : 0) / 4][j % 4] : 0.0f) * r);
        const w2v4 ntb = w2v4(-Ab[j / 4][j % 4] * r);
        constexpr int dummy = 0;
        (void)dummy;
        const int jn = j + 1;
        const int g1 = jn / 4;
        w2v4 coln[8];
        if (jn < KF) {
          // the groups of column j+1 first, then post it and request its broadcast
          if (j < 15 && g1 < 4) Aa[g1] = colc[g1] * nta + Aa[g1];
          Ab[g1] = colc[g1] * ntb + Ab[g1];
          if (jn < 16) colg[l] = Aa[(jn < 16 ? jn : 0) / 4][jn % 4];
          colg[16 + l] = Ab[g1][jn % 4];
#pragma unroll
          for (int c4 = g1; c4 < 8; ++c4) coln[c4] = *reinterpret_cast<const w2v4*>(colg + c4 * 4);
        }
        // the rest of step j
        if (j < 15) {
#pragma unroll
          for (int c4 = j / 4; c4 < 4; ++c4)
            if (!(jn < KF && c4 == g1)) Aa[c4] = colc[c4] * nta + Aa[c4];
        }
#pragma unroll
        for (int c4 = j / 4; c4 < 8; ++c4)
          if (!(jn < KF && c4 == g1)) Ab[c4] = colc[c4] * ntb + Ab[c4];
        if (jn < KF) {
#pragma unroll
          for (int c4 = g1; c4 < 8; ++c4) colc[c4] = coln[c4];
        }
      }
    }

    // ---- Schur block: slot 30 (lane 14) and slot 31 (lane 15), columns 30 and 31 ---------------
    if (live) {
      const float sq = Ab[7][2], sy = Ab[7][3];
      if (l == 14) {
        var[n_raw] = bad ? num<float>::nan() : sq;
        if (bad && a.info) atomicAdd(a.info, 1);
      } else if (l == 15) {
        mean[n_raw] = bad ? num<float>::nan() : -sq;
        if (yk) yk[n_raw] = bad ? num<float>::nan() : -sy;
      }
    }
  }
}

int g_wave2_enable = 0;  // opt-in (mgp_debug_enable_wave2): same speed as the one-row kernel today, see DESIGN.md sec. 4.1
extern int g_phase_mask;
extern int g_grid_per_cu;

int launch_fused_wave2_f32(const FusedArgs& a, hipStream_t stream) {
  if (!g_wave2_enable) return MGP_EUNSUPPORTED;
  if (!(a.k == 30 && a.R == 1 && a.d == 40 && a.ls_count == 1)) return MGP_EUNSUPPORTED;
  if ((((uintptr_t)a.feat_q | (uintptr_t)a.feat_nn) & 15) != 0) return MGP_EUNSUPPORTED;
  constexpr int NG = 4, XS = 44;
  Wave2Geom g;
  g.ntasks = (a.b + NG - 1) / NG;
  g.mask = g_phase_mask;
  const size_t lds = (size_t)NG * 32 * XS * 4 + NG * 32 * 4;  // tile + column buffer = 23040 B
  static int cached_per_cu = 0, cached_cus = 0;
  if (cached_per_cu == 0) {
    int dev = 0, n = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return MGP_EHIP;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &n, reinterpret_cast<const void*>(&fused_wave2_kernel<30, 40>), 64, lds);
    if (e != hipSuccess) return -(1000 + (int)e);
    const int by_lds = (int)((160 * 1024) / (((lds + 1279) / 1280) * 1280));  // 1280-B LDS granules
    cached_per_cu = n < by_lds ? n : by_lds;
    cached_cus = prop.multiProcessorCount;
    if (cached_per_cu < 1) return MGP_EUNSUPPORTED;
  }
  int per_cu = g_grid_per_cu > 0 ? g_grid_per_cu : cached_per_cu;
  int64_t grid = (int64_t)cached_cus * per_cu / 8 * 8;
  if (grid < 8) grid = 8;
  if (grid > g.ntasks) grid = (g.ntasks + 7) / 8 * 8;
  hipLaunchKernelGGL((fused_wave2_kernel<30, 40>), dim3((unsigned)grid), dim3(64), lds, stream, a, g);
  MGP_HIP_CHECK_LAUNCH();
  return MGP_OK;
}

}  // namespace mgp

// Backward pass (vector-Jacobian product) on the phases of the wave kernel: k + 2 <= 64 slots,
// d <= 64; Isotropy or (round 3) Anisotropy: the rows are scaled by the inverse length scales in place, after
// which everything is the isotropic computation at l = 1, the feature cotangents pick the factor up again and
// the per-feature length-scale partials are  dL/dl_f = -1/l_f sum_ij q_ij (z_if - z_jf)^2  (z = x / l, every
// ordered pair once: the sweep of phase 7 visits each unordered pair from both ends and the factor 2 of
// d(z^2)/dl cancels the 1/2).  mgp_backward.hip states the maths (and remains the path for everything else):
//
//   a = K^-1 c,  w = K^-1 (Y gm),   gK_ij = 2 gv a_i a_j - (a_i w_j + a_j w_i),  gc_j = w_j - 2 gv a_j
//   q_ij = gK_ij dkappa/dacc_ij ;  gx_i = 2 / l^2 sum_j q_ij (x_i - x_j)        (reference: torch autograd
//   over torch/muygps_layer.py:129-164, examples/muygps_torch.py:425-437)
//
// One wavefront owns 64 / NP neighbourhoods; slot i < k a neighbour row, slot k the query, slot k + 1
// the combined right-hand side Y gm.  Phases: gather (LDS tile, kept for the last phase) -> cyclic
// register-blocked pair distances (kept in registers: the derivative needs them) -> covariances ->
// exchange -> row-per-lane elimination with the multipliers kept in LDS -> back-substitution for the
// two vectors (finished components handed down by v_readlane) -> pair cotangents q (each pair once,
// written symmetrically to LDS) -> per-row feature sweep (row i in registers, rows j as uniform LDS
// broadcasts) -> one atomic add per (point, feature).
#include "mgp_wave_common.h"

namespace mgp {

// DG: 16-byte groups of a (zero-padded) feature row: the tile rows hold DG groups + one pad slot, so the
// loops over features are compile-time and read zeros past d
// KFIX > 0: nn_count known at compile time (the elimination, the back-substitution and the sweep lose their
// per-step run-time tests: one basic block each, counted LDS waits)
// HYPER (round 5): hyper-parameter gradients only -- no feature cotangents are asked for (the LOOCV gradient of
// L-BFGS-B, mgp_loocv_backward_*).  The per-feature length-scale partials are then accumulated where the pair
// cotangents q_ij are formed, by the lane that owns the pair, from the two rows of the tile:
// dL/dl_f = -(2 / l_f) sum_{pairs} q_ij (z_if - z_jf)^2; the q matrix is not written and the feature sweep -- the one
// phase that needs the whole q row (NP registers) next to two feature rows -- does not exist in the instantiation.
template <typename T, int NP, int DG, int KFIX = 0, bool HYPER = false>
// (fp64 with 64 slots: the system's row -- 64 doubles -- and the 32 kept distance accumulators alone are 256 registers:
// one wave per SIMD and the whole register file; at two waves it spills 358 .. 793 registers)
__global__ __launch_bounds__(64, (sizeof(T) == 8 && NP == 64) ? 1 : 2) void backward_wave_kernel(BackwardArgs g, int vec_ok) {
  constexpr int NH = 64 / NP;
  constexpr int NS = NP / 2, BA = 4, BP = NS / BA;
  auto own_offset = [](int j) { return j == 0 ? 0 : (j + 1) * BP + 1; };
  constexpr int E = v16<T>::N, CH = 2 * E, KS = NP + E;
  constexpr int xs = DG * E + E;  // odd number of 16-byte slots per tile row
  using V = typename v16<T>::type;
  using ACC = typename v16<T>::acc;
  const FusedArgs& a = g.f;
  const int k = KFIX > 0 ? KFIX : a.k, d = a.d, R = a.R;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* X = reinterpret_cast<T*>(smem);             // NH * NP rows x xs
  T* M = X + NH * NP * xs;                       // NH x NP x KS: exchange matrix, multipliers, then q
  T* colbuf = M + NH * NP * KS;                  // 64
  T* avec = colbuf + 64;                         // 64
  T* wvec = avec + 64;                           // 64
  T* ilb = wvec + 64;                            // 64: inverse length scales (Anisotropy; zero past d)
  int64_t* idxbuf = reinterpret_cast<int64_t*>(ilb + 64);  // 64 row offsets (elements)

  const T* feat_q = static_cast<const T*>(a.feat_q);
  const T* feat_nn = static_cast<const T*>(a.feat_nn);
  const T* targets = static_cast<const T*>(a.targets);
  const T* noise_dev = static_cast<const T*>(a.noise_dev);
  const T* gmean = static_cast<const T*>(g.grad_mean);
  const T* gvar = static_cast<const T*>(g.grad_var);
  const T* gyk = static_cast<const T*>(g.grad_yk);  // (LOOCV, R = 1: the right-hand side is y, w = gm u below)
  T* gq = static_cast<T*>(g.grad_feat_q);
  T* gnn = static_cast<T*>(g.grad_feat_nn);
  T* gtg = static_cast<T*>(g.grad_targets);
  T* gls = static_cast<T*>(g.grad_ls);
  T* gnz = static_cast<T*>(g.grad_noise);
  const bool l2 = a.metric_id == MGP_METRIC_L2;
  const bool aniso = a.ls_count > 1;
  const T inv_l = aniso ? T(1) : T(1) / static_cast<const T*>(a.length_scale)[0];
  const T post_scale = l2 ? inv_l : inv_l * inv_l;
  if (aniso) {
    const int f = threadIdx.x;
    ilb[f] = f < a.d ? T(1) / static_cast<const T*>(a.length_scale)[f] : T(0);
  }
  constexpr int wp = DG * E;                     // padded feature count (a multiple of the distance loop's chunk)
  static_assert(wp % CH == 0, "DG must be even");
  const int dv = (d + E - 1) / E;
  const int64_t ntasks = (a.b + NH - 1) / NH;

  for (int64_t task = blockIdx.x; task < ntasks; task += gridDim.x) {
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));
    const int h = NH == 1 ? 0 : lane / NP;
    const int i = lane & (NP - 1);
    T* Xh = X + h * NP * xs;
    T* Mh = M + h * NP * KS;
    T* colh = colbuf + h * NP;
    const int64_t nb = task * NH + h;
    const bool live = nb < a.b;
    const int64_t nbb = live ? nb : task * NH;

    // ---- phase 0: indices, nugget, combined right-hand side -----------------------------------------
    int64_t myidx = 0;
    T myeps = T(0), myyt = T(0);
    if (i < k) {
      myidx = a.nn_idx[nbb * k + i];
      if (a.noise_mode == MGP_NOISE_SCALAR) myeps = (T)a.noise_scalar;
      else if (a.noise_mode == MGP_NOISE_TABLE) myeps = noise_dev[myidx];
      else myeps = noise_dev[nbb * k + i];
      if (gyk) {
        myyt = targets[(a.targets_batch ? nbb * k + i : myidx) * (int64_t)R];
      } else if (gmean) {
        const T* ty = targets + (a.targets_batch ? nbb * k + i : myidx) * (int64_t)R;
        for (int r = 0; r < R; ++r) myyt += gmean[nbb * R + r] * ty[r];
      }
    } else if (i == k) {
      myidx = a.batch_idx ? a.batch_idx[nbb] : nbb;
    }
    __syncthreads();
    idxbuf[lane] = myidx * (int64_t)d;
    __syncthreads();

    // ---- phase 1: feature tile (rows 0 .. k of each half; other slots zero rows) ---------------------
    {
      constexpr int c16p = DG;
      for (int t = lane; t < NH * NP * c16p; t += 64) {
        const int row = t / c16p, c = t - row * c16p;
        const int slot = row & (NP - 1);
        V v = V(0);
        if (slot <= k && c < dv) {
          const T* src = (slot < k ? feat_nn : feat_q) + idxbuf[row] + c * E;
          if (vec_ok) {
            v = *reinterpret_cast<const V*>(src);
          } else {
#pragma unroll
            for (int e = 0; e < E; ++e)
              if (c * E + e < d) v[e] = src[e];
          }
        }
        *reinterpret_cast<V*>(X + row * xs + c * E) = v;
      }
    }
    __syncthreads();
    if (aniso) {  // z = x / l, in place (slot rows only: the right-hand-side slot is a zero row)
      T* xrow = Xh + i * xs;
#pragma unroll
      for (int c4 = 0; c4 < DG; ++c4) {
        V x = *reinterpret_cast<const V*>(xrow + c4 * E);
        x = x * *reinterpret_cast<const V*>(ilb + c4 * E);
        *reinterpret_cast<V*>(xrow + c4 * E) = x;
      }
      __syncthreads();
    }

    // ---- phase 2: squared distances of the lane's NS pairs (kept: the derivative needs them) --------
    ACC acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = ACC(0);
#pragma unroll
    for (int c0 = 0; c0 < wp; c0 += CH) {
      V own0[BA], own1[BA];
#pragma unroll
      for (int j = 0; j < BA; ++j) {
        const T* xj = Xh + ((i + own_offset(j)) & (NP - 1)) * xs + c0;
        own0[j] = *reinterpret_cast<const V*>(xj);
        own1[j] = *reinterpret_cast<const V*>(xj + E);
      }
#pragma unroll
      for (int s = 1; s <= BP; ++s) {
        const T* xo = Xh + ((i + s) & (NP - 1)) * xs + c0;
        const V o0 = *reinterpret_cast<const V*>(xo);
        const V o1 = *reinterpret_cast<const V*>(xo + E);
        if constexpr (sizeof(T) == 4 && BA == 4) {
          // hand-ordered packed blocks: no instruction reads its predecessor's result (mgp_wave_common.h)
          dist_block4(acc[s - 1], acc[BP + s - 1], acc[2 * BP + s - 1], acc[3 * BP + s - 1], own0[0], own0[1], own0[2],
                      own0[3], o0);
          dist_block4(acc[s - 1], acc[BP + s - 1], acc[2 * BP + s - 1], acc[3 * BP + s - 1], own1[0], own1[1], own1[2],
                      own1[3], o1);
        } else {
#pragma unroll
          for (int j = 0; j < BA; ++j) {
            accum(acc[j * BP + s - 1], vsub(own0[j], o0));
            accum(acc[j * BP + s - 1], vsub(own1[j], o1));
          }
        }
      }
    }

    // ---- phase 3: covariances -> exchange matrix -> row per lane ---------------------------------------
    {
      T kv[NS];
      kernel_dispatch(a.kernel_id, a.metric_id, [&](auto kid, auto mid) {
        constexpr int KID = decltype(kid)::value, MID = decltype(mid)::value;
        if constexpr (sizeof(T) == 4) {
#pragma unroll
          for (int s = 0; s < NS; s += 2) {  // two covariances per packed instruction
            const f2 kk = cov_from_sqdist2(f2{acc_total(acc[s]), acc_total(acc[s + 1])}, KID, MID, post_scale);
            kv[s] = kk.x;
            kv[s + 1] = kk.y;
          }
        } else {
#pragma unroll
          for (int s = 0; s < NS; ++s) kv[s] = cov_from_sqdist<T>(acc_total(acc[s]), KID, MID, post_scale);
        }
      });
      const int dump = (NP - 1) * KS + NP;  // padding behind the last row
#pragma unroll
      for (int s = 1; s <= NS; ++s) {
        const int r1 = (i + own_offset((s - 1) / BP)) & (NP - 1);
        const int c = (i + (s - 1) % BP + 1) & (NP - 1);
        const int hi = max(r1, c), lo = min(r1, c);
        const bool real = lo < k && hi <= k;   // neighbour x neighbour, or neighbour x query
        Mh[hi <= k ? hi * KS + lo : dump] = real ? kv[s - 1] : T(0);
      }
      Mh[i * KS + i] = i < k ? T(1) + myeps : T(1);
    }
    __syncthreads();
    if (i < k) Mh[(k + 1) * KS + i] = myyt;      // the right-hand-side row (no pair store touches it)
    else if (i <= k + 1) Mh[(k + 1) * KS + i] = i == k + 1 ? T(1) : T(0);
    __syncthreads();
    V A[NP / E];
#pragma unroll
    for (int c4 = 0; c4 < NP / E; ++c4)
      A[c4] = i <= k + 1 ? *reinterpret_cast<const V*>(Mh + i * KS + c4 * E) : V(0);  // slots behind the rhs row: zero rows
    __syncthreads();

    // ---- phase 4: elimination, multipliers l_ij kept in the exchange matrix ---------------------------
    __builtin_amdgcn_s_setprio(2);  // chains of short dependent steps go first
    bool bad = false;
#pragma unroll
    for (int j = 0; j < (KFIX > 0 ? KFIX : NP - 2); ++j) {
      if (KFIX > 0 || j < k) {
        const T ajj = A[j / E][j % E];
        colh[i] = ajj;
        __syncthreads();
        const V cp = *reinterpret_cast<const V*>(colh + (j / E) * E);
        const T p = cp[j % E];
        bad = bad || !(p > T(0));
        const T t = ajj * pivot_rcp(p);
        Mh[i * KS + j] = t;
        const V nt = V(-t);
        A[j / E] = cp * nt + A[j / E];
        constexpr int GC = sizeof(T) == 4 ? 8 : 4;
#pragma unroll
        for (int c0 = j / E + 1; c0 < NP / E; c0 += GC) {
          V cv[GC];
#pragma unroll
          for (int u = 0; u < GC; ++u)
            if (c0 + u < NP / E) cv[u] = *reinterpret_cast<const V*>(colh + (c0 + u) * E);
#pragma unroll
          for (int u = 0; u < GC; ++u)
            if (c0 + u < NP / E) A[c0 + u] = cv[u] * nt + A[c0 + u];
        }
        __syncthreads();
      }
    }

    // ---- phase 5: back-substitution L^T [a w] = D^-1 L^-1 [c yt] ----------------------------------------
    T xa = i < k ? Mh[k * KS + i] : T(0);
    T xw = i < k ? Mh[(k + 1) * KS + i] : T(0);
    // Eight steps per block: the multipliers do not depend on the solution and are requested together, the updates
    // are selects, not branches -- a step that reads its multiplier inside a divergent `if` is a basic block of its
    // own with an LDS round trip in the open (29 of them per neighbourhood; the same cure as in mgp_fused_rhs.hip).
    {
      constexpr int BB = 8;
      constexpr int KTOP = KFIX > 0 ? KFIX : NP - 2;  // steps m = KTOP - 1 .. 1
#pragma unroll
      for (int mb = (KTOP - 1) / BB * BB; mb >= 0; mb -= BB) {
        if (KFIX > 0 || mb < k) {  // (uniform)
          T lm[BB];
#pragma unroll
          for (int e = 0; e < BB; ++e) lm[e] = Mh[(mb + e) * KS + i];
#pragma unroll
          for (int e = BB - 1; e >= 0; --e) {
            const int m = mb + e;
            if (m >= 1 && m < KTOP) {
              T am = lane_value(xa, m), wm = lane_value(xw, m);
              if constexpr (NH == 2) {
                const T am1 = lane_value(xa, m + NP), wm1 = lane_value(xw, m + NP);
                am = h == 0 ? am : am1;
                wm = h == 0 ? wm : wm1;
              }
              const bool on = i < m && (KFIX > 0 || m < k);
              xa = on ? fma_t(-lm[e], am, xa) : xa;
              xw = on ? fma_t(-lm[e], wm, xw) : xw;
            }
          }
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();  // every lane is done with the multipliers: M becomes the q matrix
    // (LOOCV: xw is u = K^-1 y so far; w = gm u from here on, u kept for the y^T K^-1 y terms)
    const T xu = xw;
    const T gyv = gyk ? gyk[nbb] : T(0);
    if (gyk) xw = (gmean ? gmean[nbb] : T(0)) * xu;
    T* uvec = colbuf;  // (free between the back-substitution and the next task's elimination)
    avec[lane] = i < k ? xa : T(0);
    wvec[lane] = i < k ? xw : T(0);
    if (gyk) uvec[lane] = i < k ? xu : T(0);
    __syncthreads();
    const bool skip = bad || !live;  // cotangents of a non-SPD neighbourhood are left untouched

    // ---- phase 6: pair cotangents q_ij = gK_ij dkappa/dacc_ij, symmetric, zero diagonal ---------------
    const T gv = gvar ? gvar[nbb] : T(0);
    T liso = T(0);
    V s2h[HYPER ? DG : 1];  // (HYPER) per-feature sums of q_ij dz^2 over the lane's own pairs
    T qown[HYPER ? NS : 1];  // (HYPER) q of the lane's own pairs (zero for a pair that is not real, or met twice)
    {
      const T* ah = avec + h * NP;
      const T* wh = wvec + h * NP;
      const int dump = (NP - 1) * KS + NP;
      kernel_dispatch(a.kernel_id, a.metric_id, [&](auto kid, auto mid) {
        constexpr int KID = decltype(kid)::value, MID = decltype(mid)::value;
#pragma unroll
        for (int s = 1; s <= NS; ++s) {
          const int r1 = (i + own_offset((s - 1) / BP)) & (NP - 1);
          const int c = (i + (s - 1) % BP + 1) & (NP - 1);
          const int hi = max(r1, c), lo = min(r1, c);
          const bool real = lo < k && hi <= k;
          const T alo = ah[lo], wlo = wh[lo], ahi = ah[hi], whi = wh[hi];
          T gK = hi < k ? T(2) * gv * ahi * alo - (ahi * wlo + alo * whi) : wlo - T(2) * gv * alo;
          if (gyk && hi < k) gK -= T(2) * gyv * uvec[h * NP + hi] * uvec[h * NP + lo];
          const T accv = acc_total(acc[s - 1]);
          const T x = (MID == MGP_METRIC_L2 ? sqrt_fast(accv) : accv) * post_scale;
          // d kappa / d x with the fast exponential of the forward kernels (mgp_device.h: kernel_deriv)
          T kp;
          if constexpr (KID == MGP_KERNEL_RBF) kp = T(-0.5) * exp_neg(x * T(0.5));
          else if constexpr (KID == MGP_KERNEL_MATERN_05) kp = -exp_neg(x);
          else if constexpr (KID == MGP_KERNEL_MATERN_15) kp = T(-3) * x * exp_neg(x * T(1.7320508075688772935));
          else if constexpr (KID == MGP_KERNEL_MATERN_25) {
            const T t = x * T(2.2360679774997896964);
            kp = T(-5.0 / 3.0) * x * (T(1) + t) * exp_neg(t);
          } else kp = -x * exp_neg(x * x * T(0.5));
          T dk_dacc;
          if constexpr (MID == MGP_METRIC_L2) dk_dacc = x > T(0) ? kp * post_scale * post_scale * pivot_rcp(T(2) * x) : T(0);
          else dk_dacc = kp * post_scale;
          const T q = real ? gK * dk_dacc : T(0);
          // the pair at cyclic distance NP / 2 is met from both ends: count it once
          const bool twice = ((r1 - c) & (NP - 1)) == NP / 2 && r1 < c;
          if (real && !twice) liso += gK * kp * x;
          if constexpr (HYPER) {
            qown[s - 1] = (real && !twice) ? q : T(0);
          } else {
            Mh[hi <= k ? hi * KS + lo : dump] = q;
            Mh[hi <= k ? lo * KS + hi : dump] = q;
          }
        }
      });
      if constexpr (!HYPER) Mh[i * KS + i] = T(0);
    }
    if constexpr (HYPER) {
      // the pairs' shares of the per-feature length-scale partials, pair by pair (the distance accumulators are dead
      // by now; one copy of this loop, not one per covariance function)
#pragma unroll
      for (int c4 = 0; c4 < DG; ++c4) s2h[c4] = V(0);
      if (gls && aniso) {  // (uniform)
#pragma unroll
        for (int s = 1; s <= NS; ++s) {
          const int r1 = (i + own_offset((s - 1) / BP)) & (NP - 1);
          const int c = (i + (s - 1) % BP + 1) & (NP - 1);
          const V qv = V(qown[s - 1]);
          const T* xa_ = Xh + r1 * xs;
          const T* xb_ = Xh + c * xs;
#pragma unroll
          for (int c4 = 0; c4 < DG; ++c4) {
            const V dz = *reinterpret_cast<const V*>(xa_ + c4 * E) - *reinterpret_cast<const V*>(xb_ + c4 * E);
            s2h[c4] = (dz * qv) * dz + s2h[c4];
          }
          if ((s & 3) == 0) __builtin_amdgcn_sched_barrier(0);  // (four pairs' reads in flight; all hoisted, they spill)
        }
      }
    }
    // per-neighbourhood outputs that need a and w only
    if (!skip) {
      if (gnz && i < k) gnz[nb * k + i] = gv * xa * xa - xa * xw - gyv * xu * xu;
      if (gtg && gmean && i < k)
        for (int r = 0; r < R; ++r) unsafeAtomicAdd(gtg + myidx * (int64_t)R + r, gmean[nb * R + r] * xa + (gyk ? T(2) * gyv * xu : T(0)));
    }
    if (gls && !aniso) {
      // sum over the lanes of the half
      T s = liso;
      for (int off = NP / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      if (!skip && i == 0) gls[nb] = -(l2 ? T(1) : T(2)) * inv_l * s;  // dx/dl = -x/l | -2x/l
    }
    __syncthreads();

    // ---- phase 7: feature cotangents: gx_i = 2 sum_j q_ij (x_i - x_j) (the metric's 1/l^2 is in q) ------
    const bool want_gl = gls && aniso;
    if constexpr (HYPER) {
      if (want_gl) {
        // every lane's pair sums through the tile, lane f adds column f (each unordered pair once: twice the sum)
        __syncthreads();
#pragma unroll
        for (int c4 = 0; c4 < DG; ++c4) *reinterpret_cast<V*>(Xh + i * xs + c4 * E) = s2h[c4];
        __syncthreads();
        for (int f = i; f < d; f += NP) {
          T acc = T(0);
          for (int j = 0; j < NP; ++j) acc += Xh[j * xs + f];
          if (!skip) gls[nb * (int64_t)d + f] = T(-2) * ilb[f] * acc;
        }
      }
    } else
    if (gq || gnn || want_gl) {
      // (rows of more than ten 16-byte groups: the sweep in two passes over half the groups each -- its own row, the
      // sums and a row j in flight are 3 DG registers x 4, and at DG = 12 / 16 the single pass spilled 441 / 2 199
      // registers: 9.0 / 18.4 ms per 500 k neighbourhoods at d = 48 / 64 where d = 40 takes 6.2)
      constexpr int NPASS = DG > 10 ? 2 : 1, DGH = DG / NPASS;
      V s2[DG], qrow[NP / E];
#pragma unroll
      for (int c4 = 0; c4 < DG; ++c4) s2[c4] = V(0);
#pragma unroll
      for (int c4 = 0; c4 < NP / E; ++c4) qrow[c4] = *reinterpret_cast<const V*>(Mh + i * KS + c4 * E);
      const int k_rt = a.k;  // (the run-time value on purpose: with every test folded the compiler hoists the 310
                             //  row reads of the unrolled sweep to its top and spills -- measured 88 instead of 13 ms)
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        V xi[DGH], sv[DGH];
#pragma unroll
        for (int c4 = 0; c4 < DGH; ++c4) {
          sv[c4] = V(0);
          xi[c4] = *reinterpret_cast<const V*>(Xh + i * xs + (ps * DGH + c4) * E);
        }
#pragma unroll
        for (int j = 0; j < (KFIX > 0 ? KFIX + 1 : NP - 1); ++j) {
          if (j <= k_rt) {  // uniform; the DGH reads of a row are issued together
            const V qv = V(qrow[j / E][j % E]);
            const T* xj = Xh + j * xs + ps * DGH * E;
            V xr[DGH];
#pragma unroll
            for (int c4 = 0; c4 < DGH; ++c4) xr[c4] = *reinterpret_cast<const V*>(xj + c4 * E);
            if (want_gl) {  // (uniform)
#pragma unroll
              for (int c4 = 0; c4 < DGH; ++c4) {
                const V t = (xi[c4] - xr[c4]) * qv;
                sv[c4] = sv[c4] + t;
                s2[ps * DGH + c4] = t * (xi[c4] - xr[c4]) + s2[ps * DGH + c4];
              }
            } else {
#pragma unroll
              for (int c4 = 0; c4 < DGH; ++c4) sv[c4] = (xi[c4] - xr[c4]) * qv + sv[c4];
            }
          }
        }
        // out through the tile (every lane is done reading this pass's columns; a later pass reads other columns):
        // consecutive lanes then add consecutive features of a row -- a lane adding its own row's features one by one
        // touches 62 different cache lines per instruction (measured: 57 ms of a 65 ms launch)
        __syncthreads();
#pragma unroll
        for (int c4 = 0; c4 < DGH; ++c4) {
          V o = (skip || i > k) ? V(0) : V(T(2)) * sv[c4];
          if (aniso) o = o * *reinterpret_cast<const V*>(ilb + (ps * DGH + c4) * E);  // dz/dx = 1 / l
          *reinterpret_cast<V*>(Xh + i * xs + (ps * DGH + c4) * E) = o;
        }
      }
      __syncthreads();
      // consecutive lanes add consecutive features (row-major over the rows of the task): an atomic
      // instruction then touches 2-3 cache lines; (row, feature) advance incrementally, no division
      {
        int row = lane / d, f = lane - row * d;
        for (int t = lane; t < NH * NP * d; t += 64) {
          const int slot = row & (NP - 1);
          if (slot <= k) {
            T* dstp = slot < k ? gnn : gq;
            const T v = X[row * xs + f];
            if (dstp != nullptr && v != T(0)) unsafeAtomicAdd(dstp + idxbuf[row] + f, v);
          }
          f += 64;
          while (f >= d) {
            f -= d;
            ++row;
          }
        }
      }
      if (want_gl) {
        // per-feature length-scale partials: the rows' sums go through the tile once more, lane f adds column f
        __syncthreads();
#pragma unroll
        for (int c4 = 0; c4 < DG; ++c4) *reinterpret_cast<V*>(Xh + i * xs + c4 * E) = i > k ? V(0) : s2[c4];
        __syncthreads();
        for (int f = i; f < d; f += NP) {
          T acc = T(0);
          for (int j = 0; j <= k; ++j) acc += Xh[j * xs + f];
          if (!skip) gls[nb * (int64_t)d + f] = -ilb[f] * acc;
        }
      }
    }
    if (bad && live && i == 0 && a.info) atomicAdd(a.info, 1);
  }
}

template <typename T, int NP, int DG, int KFIX = 0, bool HYPER = false>
static int launch_bwd_np_impl(const BackwardArgs& g, hipStream_t stream) {
  constexpr int NH = 64 / NP;
  constexpr int E = v16<T>::N, KS = NP + E, xs = DG * E + E;
  const uintptr_t align = (uintptr_t)g.f.feat_q | (uintptr_t)g.f.feat_nn;
  const int vec_ok = (g.f.d % E == 0) && (align % 16 == 0);
  const size_t lds = ((size_t)NH * NP * xs + (size_t)NH * NP * KS + 4 * 64) * sizeof(T) + 64 * sizeof(int64_t);
  static Residency res;
  int per_cu = 0, cus = 0;
  const int rc = res.lookup(reinterpret_cast<const void*>(&backward_wave_kernel<T, NP, DG, KFIX, HYPER>), 64, lds, &per_cu, &cus);
  if (rc != MGP_OK) return rc;
  const int64_t ntasks = (g.f.b + NH - 1) / NH;
  int64_t grid = (int64_t)cus * per_cu;
  if (grid > ntasks) grid = ntasks;
  hipLaunchKernelGGL((backward_wave_kernel<T, NP, DG, KFIX, HYPER>), dim3((unsigned)grid), dim3(64), lds, stream, g, vec_ok);
  MGP_HIP_CHECK_LAUNCH();
  return MGP_OK;
}
// (no feature cotangent asked for: the instantiation without the sweep)
template <typename T, int NP, int DG, int KFIX = 0>
static int launch_bwd_np(const BackwardArgs& g, hipStream_t stream) {
  // (Anisotropy only: under Isotropy the one length-scale partial needs no sweep in either instantiation, and the
  // sweep-less one spills more at the headline shape: 6.3 against 4.9 ms per 1 M neighbourhoods)
  if (!g.grad_feat_q && !g.grad_feat_nn && g.f.ls_count > 1) return launch_bwd_np_impl<T, NP, DG, KFIX, true>(g, stream);
  return launch_bwd_np_impl<T, NP, DG, KFIX, false>(g, stream);
}

template <typename T, int NP>
static int launch_bwd_dg(const BackwardArgs& g, hipStream_t stream) {
  constexpr int E = v16<T>::N;
  const int dv = (g.f.d + E - 1) / E;
  if (dv <= 4) return launch_bwd_np<T, NP, 4>(g, stream);
  if (dv <= 8) return launch_bwd_np<T, NP, 8>(g, stream);
  if constexpr (sizeof(T) == 8) {
    return MGP_EUNSUPPORTED;  // fp64 rows of more than 16 features: the sweep's registers spill
  } else {
    if (dv <= 10) {  // d = 40: 2 KB of LDS less than DG = 12, one more wave per CU
      if constexpr (NP == 32) {
        if (g.f.k == 30 && dv == 10) return launch_bwd_np<T, NP, 10, 30>(g, stream);  // the headline shape, static
      }
      return launch_bwd_np<T, NP, 10>(g, stream);
    }
    if (dv <= 12) return launch_bwd_np<T, NP, 12>(g, stream);
    return launch_bwd_np<T, NP, 16>(g, stream);
  }
}

// d <= 64 (fp32) / 16 (fp64), k + 2 <= 64 (fp32) / 32 (fp64); everything else: MGP_EUNSUPPORTED (the LDS workgroup kernel)
template <typename T>
int launch_backward_wave(const BackwardArgs& g, hipStream_t stream) {
  const int rows = g.f.k + 2;
  if ((g.f.ls_count != 1 && g.f.ls_count != g.f.d) || g.f.d > 16 * (16 / (int)sizeof(T)) || rows > 64) return MGP_EUNSUPPORTED;
  if (rows <= 32) return launch_bwd_dg<T, 32>(g, stream);
  if constexpr (sizeof(T) == 8) {
    // rows of at most 8 features (round 5; BASELINE config 4: k = 50, d = 8, whose L-BFGS-B gradient is this launch):
    // hyper-parameter gradients without the sweep at two waves per SIMD; with feature cotangents the sweep's 64-double
    // q row takes the whole register file (one wave per SIMD) -- still ahead of the LDS workgroup kernel
    if (g.f.d <= 8) return launch_bwd_np<T, 64, 4>(g, stream);
    return MGP_EUNSUPPORTED;
  } else {
    return launch_bwd_dg<T, 64>(g, stream);
  }
}

template int launch_backward_wave<float>(const BackwardArgs&, hipStream_t);
template int launch_backward_wave<double>(const BackwardArgs&, hipStream_t);

}  // namespace mgp

// Prediction with up to 16 responses and nn_count <= 64 (BASELINE config 5), fp32: the rhs-column kernel of
// mgp_fused_rhs.hip rebuilt around the MATRIX cores' output layout (round 4).
//
// Reference path: src/MuyGPyS/_src/gp/tensors/numpy.py:47-94 (differences, F2 / l2), _src/gp/kernels/numpy.py:12-31,
// _src/gp/noise/numpy.py:9-27, _src/gp/muygps/numpy.py:17-67 (posterior mean and variance) -- one wave per
// neighbourhood, nothing materialised.
//
// The Gram matrix of the query-centred rows comes from v_mfma_f32_32x32x2_f32 (plain fp32 arithmetic): three 32 x 32
// tiles -- G00 = A0 A0^T, G01 = A0 A1^T, G11 = A1 A1^T, A0 / A1 the rows 0..31 / 32..63 of the tile.  The instruction
// leaves entry (8 q + 4 h + e, c) of a tile in register 4 q + e of lane 32 h + c.  By symmetry that is ROW c (G00) or
// ROW 32 + c (G01, G11) of the system at the columns S_h = { 8 q + 4 h + e } (+ 32 for G11): lane (c, h) holds half of
// row c -- of which only columns 0..31 matter, the row is above the diagonal beyond -- and half of row 32 + c, its
// partner lane (c, 1 - h) the other halves, in whole 16-byte groups of consecutive columns.  That IS an elimination
// layout: row-per-lane-pair, every lane busy on 12 register groups instead of 16 half-dead ones.  So
//   * the 640 packed FMAs of the pair scheme become 60 matrix instructions on an otherwise idle pipe,
//   * covariances are evaluated where the entries land and NOTHING is exchanged through LDS (the exchange matrix,
//     its nine-instruction address arithmetic per entry and the row read-back are gone),
//   * the system is 48 registers per lane and the kernel's LDS is the feature tile: twelve workgroups per CU, three
//     waves per SIMD (the row-per-lane kernels: 64-row registers + a 17.4 KB exchange matrix, two waves),
//   * per elimination step 344 / 64 group updates on average instead of 544 / 64 (the folded two-neighbourhood variant
//     reaches the same count at 96 registers per lane and two waves).
//   * the first 32 steps' updates of the trailing block (rows and columns 32 .. 63) are a rank-32 update whose
//     accumulator those registers already are: one matrix instruction per two steps (MGP_RHS_MF_TRAIL) instead of 16
//     packed FMAs and 8 column-group reads.
// Elimination: column j is posted to LDS by the half that holds it (look-ahead: right after the group with column
// j + 1 is updated), every lane reads the pivot's group (broadcast), its two rows' entries of the column (they are
// the multipliers' numerators; the partner half holds them) and the column groups its own registers need.  The
// multipliers of four steps are collected in registers and written to the packed L (the feature tile is dead by then)
// where the back-substitution reads them column-wise; right-hand side, back-substitution and outputs as in
// mgp_fused_rhs.hip (prediction variant).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "mgp_wave_common.h"

#ifndef MGP_RHS_MF_TIMING
#define MGP_RHS_MF_TIMING 0
#endif
#if MGP_RHS_MF_TIMING
// phase timing (experiments only; tools/rhs_timing.py --mf): s_memtime differences summed per wave and phase
__device__ unsigned long long g_rhs_mf_timing[8];
#define MGP_MF_T(slot)                                             \
  {                                                                \
    const unsigned long long tnow_ = __builtin_readcyclecounter(); \
    tacc_[slot] += tnow_ - tlast_;                                 \
    tlast_ = tnow_;                                                \
  }
extern "C" int mgp_debug_rhs_mf_timing(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rhs_mf_timing), sizeof(g_rhs_mf_timing)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_rhs_mf_timing), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#else
#define MGP_MF_T(slot)
#endif

namespace mgp {

struct RhsMfGeom {
  int dst, xs;
  int resp_vec;  // all RC responses of a row in 16-byte loads (R == RC, rows 16-byte aligned)
  int64_t ntasks;
  // element strides of the feature rows and of the response rows, and where they start: the plain tables (d, d, R) or
  // -- round 5 -- prepared tables (mgp_table_pack_*: rows [features d | responses R | pad] at a 64-byte multiple
  // stride, 256 B at d = 40, R = 16): a neighbour's sixteen responses then lie in the second of the two 128-byte lines
  // its feature row is gathered from, where the plain tables cost a third line per neighbour
  // (32-bit, and the bases in FusedArgs' own feat_q / feat_nn / targets fields -- the launcher hands the kernel a copy
  // with the prepared tables' addresses there: scalar registers are what this kernel spills)
  int row_nn, row_q, row_resp;
};

#ifndef MGP_RHS_MF_BLOCK
#define MGP_RHS_MF_BLOCK 32
#endif
#ifndef MGP_RHS_MF_SCHED
#define MGP_RHS_MF_SCHED 1
#endif
#ifndef MGP_RHS_MF_PRIO
#define MGP_RHS_MF_PRIO 2  // issue priority raised: covariances .. elimination (1), elimination (2), never (0), (3) / (4): as 1 / 2 + the back-substitution
#endif
#ifndef MGP_RHS_MF_WRITELANE
#define MGP_RHS_MF_WRITELANE 1
#endif
#ifndef MGP_RHS_MF_TRAIL
#define MGP_RHS_MF_TRAIL 1
#endif
#ifndef MGP_RHS_MF_WAVES
#define MGP_RHS_MF_WAVES 3
#endif

template <int RC>
__global__ __launch_bounds__(64, MGP_RHS_MF_WAVES) void fused_rhs_mf_kernel(FusedArgs a, RhsMfGeom g) {
  using T = float;
  using V = v16<float>::type;
  using ACC = v16<float>::acc;
  using F16 = float __attribute__((ext_vector_type(16)));
  constexpr int NP = 64, HALF = 32, E = 4, CH = 8;
  auto troff = [](int r) { return E * (r / E + 1) * (E * (r / E) / 2 + r % E); };  // packed lower triangle, rows in whole groups
  constexpr int KTRI = E * ((NP - 1) / E + 1) * (E * ((NP - 1) / E) / 2 + (NP - 1) % E) + NP + E;

  extern __shared__ __attribute__((aligned(16))) char smem[];
#if MGP_RHS_MF_TIMING
  unsigned long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast_ = __builtin_readcyclecounter();
#endif
  const int k = a.k, d = a.d, R = a.R, xs = g.xs, dst = g.dst;
  const int rows_x = NP + 1;  // 64 slots + the query
  const int tile_elems = rows_x * xs > KTRI ? rows_x * xs : KTRI;
  T* tile = reinterpret_cast<T*>(smem);  // feature rows; from the elimination on the packed L
  T* colbuf = tile + tile_elems;         // two column buffers (the second one holds the squared norms before)
  T* cq = colbuf + 2 * NP;               // 64: cross-covariances on their way to the row owners
  T* epsb = cq + NP;                     // 64: nuggets, likewise
  T* ilbuf = epsb + NP;                  // dst: inverse length scales (Anisotropy)
  int64_t* idxbuf = reinterpret_cast<int64_t*>(colbuf);  // 65 row offsets: only alive during the gather
  T* normb = colbuf + NP;

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
  const int w = d, wp = (d + CH - 1) / CH * CH;  // one feature stage

  int64_t next_idx = 0, next_q = 0;
  if ((int64_t)blockIdx.x < g.ntasks) {
    next_idx = a.nn_idx[(int64_t)blockIdx.x * k + ((int)threadIdx.x < k ? (int)threadIdx.x : 0)];
    next_q = a.batch_idx ? a.batch_idx[blockIdx.x] : (int64_t)blockIdx.x;
  }
  for (int64_t nb = blockIdx.x; nb < g.ntasks; nb += gridDim.x) {
    int i = threadIdx.x;
    asm volatile("" : "+v"(i));  // keep per-lane addresses out of LICM (register pressure)
    const int c = i & (HALF - 1), h = i >> 5;

    // ---- indices, nugget ------------------------------------------------------------------------------
    const int64_t myidx = i < k ? next_idx : 0;
    const int64_t qidx = next_q;
    if (nb + gridDim.x < g.ntasks) {
      const int64_t nn = nb + gridDim.x;
      next_idx = a.nn_idx[nn * k + (i < k ? i : 0)];
      next_q = a.batch_idx ? a.batch_idx[nn] : nn;
    }
    __syncthreads();  // the previous task's LDS reads are done
    idxbuf[i] = myidx * (int64_t)g.row_nn;
    if (i == 0) idxbuf[NP] = qidx * (int64_t)g.row_q;
    T myeps = T(0);
    if (i < k) {
      if (a.noise_mode == MGP_NOISE_SCALAR) myeps = (T)a.noise_scalar;
      else if (a.noise_mode == MGP_NOISE_TABLE) myeps = noise_dev[myidx];
      else myeps = noise_dev[nb * k + i];
    }
    const int64_t yrow = a.targets_batch ? nb * k + i : myidx;
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();

    // ---- gather: rows straight from global memory into LDS (mgp_fused_rhs.hip, folded variant) --------------
    {
      const int SPR = xs / E;
      const unsigned total = (unsigned)(rows_x * SPR);
      const unsigned smagic = (1u << 20) / (unsigned)SPR + 1u;
      const int c16 = w / E;
      for (unsigned n = 0; n * 64u < total; ++n) {
        const unsigned sigma = 64u * n + (unsigned)i;
        const unsigned row = (sigma * smagic) >> 20;
        const unsigned cc = sigma - row * (unsigned)SPR;
        const bool on = sigma < total && ((int)row < k || row == (unsigned)NP);
        const T* src = ((int)row < k ? feat_nn : feat_q) + idxbuf[(int)row < k ? row : NP] + min((int)cc, c16 - 1) * E;
        if (on) glds16_lds(src, smem, (int)n * 1024);
      }
      for (int t = i; t < (NP - k) * SPR; t += NP)  // zero rows for the unused slots
        *reinterpret_cast<V*>(tile + (k + t / SPR) * xs + (t % SPR) * E) = V(0);
      if (wp > w) {  // (uniform) the padding group of the 8-wide loops: zero, once the rows have landed
        lds_dma_wait();
        *reinterpret_cast<V*>(tile + i * xs + w) = V(0);
        if (i == 0) *reinterpret_cast<V*>(tile + NP * xs + w) = V(0);
      }
      if (aniso)
        for (int f = i; f < wp; f += 64) ilbuf[f] = f < w ? T(1) / ls[f] : T(0);
      lds_dma_wait();
    }
    __syncthreads();
    MGP_MF_T(0)

    // ---- centre row i on the query (times the inverse length scales), in place; squared norms ------------
    T nrm;
    {
      const T* xq = tile + NP * xs;
      T* xrow = tile + i * xs;
      ACC n2[2] = {ACC(0), ACC(0)};
      for (int c0 = 0; c0 < wp; c0 += CH) {
        V x0 = *reinterpret_cast<const V*>(xrow + c0), x1 = *reinterpret_cast<const V*>(xrow + c0 + E);
        x0 = vsub(x0, *reinterpret_cast<const V*>(xq + c0));
        x1 = vsub(x1, *reinterpret_cast<const V*>(xq + c0 + E));
        if (aniso) {
          x0 = x0 * *reinterpret_cast<const V*>(ilbuf + c0);
          x1 = x1 * *reinterpret_cast<const V*>(ilbuf + c0 + E);
        }
        norm_accum(n2[0], x0);
        norm_accum(n2[1], x1);
        *reinterpret_cast<V*>(xrow + c0) = x0;
        *reinterpret_cast<V*>(xrow + c0 + E) = x1;
      }
      nrm = acc_total(n2[0] + n2[1]);  // = the crosswise squared distance of row i
    }
    __syncthreads();  // (the row offsets, which share the column buffers' space, are dead)
    normb[i] = i < k ? nrm : num<T>::inf();  // (unused slots: an infinite norm keeps their pairs away from the guard)
    epsb[i] = myeps;

    // ---- Gram matrix on the matrix cores ------------------------------------------------------------------------
    F16 g00 = F16(0), g01 = F16(0), g11 = F16(0);
    {
      // K = 2 per instruction: lanes 0 .. 31 supply feature t of their row, lanes 32 .. 63 feature wp / 2 + t
      const T* r0 = tile + c * xs + h * (wp / 2);
      const T* r1 = tile + (HALF + c) * xs + h * (wp / 2);
      for (int cc = 0; cc < wp / (2 * E); ++cc) {
        const V a0 = *reinterpret_cast<const V*>(r0 + cc * E), a1 = *reinterpret_cast<const V*>(r1 + cc * E);
#pragma unroll
        for (int e = 0; e < E; ++e) {
          g00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], a0[e], g00, 0, 0, 0);
          g01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], a1[e], g01, 0, 0, 0);
          g11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], a1[e], g11, 0, 0, 0);
        }
      }
    }
    __syncthreads();
    MGP_MF_T(1)
    // squared distances |a'|^2 + |b'|^2 - 2 a'.b' with the cancellation guard (mgp_wave_common.h); the two diagonal
    // entries a lane may hold (distance zero against twice the norm) stay out of the guard
    const int rdiag = ((c >> 2) & 1) == h ? 4 * (c >> 3) + (c & 3) : -1;  // the register with row == column (tiles 00, 11)
    // (round 6: the 24 pairs of squared distances are NOT kept between the guard and the covariances -- 48 registers on
    // top of the three accumulators were what spilled at three waves per SIMD (13 registers, 0.68 GB of scratch writes
    // per launch).  The guard pass keeps its minimum only; the covariance pass forms each pair again from the
    // accumulators and the norms: one packed FMA.)
    bool diff_form = false;  // (wave-uniform) the guard tripped: the accumulators hold difference-form distances
    {
      const T nc0 = normb[c], nc1 = normb[HALF + c];
      T guard = T(1);
      static_for<4>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const V nra = *reinterpret_cast<const V*>(normb + 8 * q + 4 * h);
        const V nrb = *reinterpret_cast<const V*>(normb + HALF + 8 * q + 4 * h);
        static_for<6>([&](auto pc) {
          constexpr int tl = decltype(pc)::value / 2, r = 4 * q + 2 * (decltype(pc)::value % 2);
          const F16& gg = tl == 0 ? g00 : (tl == 1 ? g01 : g11);
          const V& nr = tl == 2 ? nrb : nra;
          const T nc = tl == 0 ? nc0 : nc1;
          const f2 ns = f2{nr[r % 4] + nc, nr[r % 4 + 1] + nc};
          const f2 dd = f2{gg[r], gg[r + 1]} * f2{-2.0f, -2.0f} + ns;
          f2 tt = dd * f2{MGP_GRAM_GUARD, MGP_GRAM_GUARD} - ns;
          if constexpr (tl != 1) {
            tt.x = rdiag == r ? T(1) : tt.x;
            tt.y = rdiag == r + 1 ? T(1) : tt.y;
          }
          guard = __builtin_fminf(__builtin_fminf(guard, tt.x), tt.y);
        });
      });
      if (gram_guard_tripped(guard)) {
        // this neighbourhood's distances again in the difference form, entry by entry (rare: registers before speed),
        // INTO the accumulators
        diff_form = true;
        static_for<48>([&](auto ec) {
          constexpr int en = decltype(ec)::value, tl = en / 16, r = en % 16;
          const T* xa = tile + ((tl == 2 ? HALF : 0) + 8 * (r / 4) + 4 * h + r % 4) * xs;
          const T* xb = tile + ((tl == 0 ? 0 : HALF) + c) * xs;
          ACC sum = ACC(0);
#pragma nounroll
          for (int c0 = 0; c0 < wp; c0 += E)
            accum(sum, vsub(*reinterpret_cast<const V*>(xa + c0), *reinterpret_cast<const V*>(xb + c0)));
          F16& gg = tl == 0 ? g00 : (tl == 1 ? g01 : g11);
          gg[r] = acc_total(sum);
        });
      }
    }
    __syncthreads();  // the feature rows are dead from here: the packed L takes their place

    // ---- covariances where the entries are: KS = row c at columns 8 q + 4 h + e, KL = row 32 + c at the same
    //      columns (groups 0 .. 3) and at 32 + them (groups 4 .. 7) ------------------------------------------------
#if MGP_RHS_MF_PRIO == 1
    __builtin_amdgcn_s_setprio(2);
#elif MGP_RHS_MF_PRIO == 3
    __builtin_amdgcn_s_setprio(1);
#endif
    V KS[4], KL[8];
    T rvS, rvL;
    {
      const T cscale = post_scale;  // (1 under Anisotropy: the rows are scaled)
      T kq = T(0);
      const T nc0 = normb[c], nc1 = normb[HALF + c];
      kernel_dispatch(a.kernel_id, a.metric_id, [&](auto kid, auto mid) {
        constexpr int KID = decltype(kid)::value, MID = decltype(mid)::value;
        static_for<4>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          const V nra = *reinterpret_cast<const V*>(normb + 8 * q + 4 * h);
          const V nrb = *reinterpret_cast<const V*>(normb + HALF + 8 * q + 4 * h);
          static_for<6>([&](auto pc) {
            constexpr int tl = decltype(pc)::value / 2, r = 4 * q + 2 * (decltype(pc)::value % 2);
            const F16& gg = tl == 0 ? g00 : (tl == 1 ? g01 : g11);
            const V& nr = tl == 2 ? nrb : nra;
            const T nc = tl == 0 ? nc0 : nc1;
            const f2 ns = f2{nr[r % 4] + nc, nr[r % 4 + 1] + nc};
            f2 dd = f2{gg[r], gg[r + 1]} * f2{-2.0f, -2.0f} + ns;
            dd = f2{__builtin_fmaxf(dd.x, 0.0f), __builtin_fmaxf(dd.y, 0.0f)};
            if (diff_form) dd = f2{gg[r], gg[r + 1]};
            const f2 kk = cov_from_sqdist2(dd, KID, MID, cscale);
            if constexpr (tl == 0) {
              KS[r / 4][r % 4] = kk.x;
              KS[r / 4][r % 4 + 1] = kk.y;
            } else {
              KL[(tl == 2 ? 4 : 0) + r / 4][r % 4] = kk.x;
              KL[(tl == 2 ? 4 : 0) + r / 4][r % 4 + 1] = kk.y;
            }
          });
        });
        kq = cov_from_sqdist<T>(nrm, KID, MID, cscale);
      });
      cq[i] = i < k ? kq : T(0);
      if (k < NP) {  // (uniform) unused slots: identity rows
        const bool sk = c < k, lk = HALF + c < k;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const bool colk = 8 * q + 4 * h + e < k, colk2 = HALF + 8 * q + 4 * h + e < k;
            KS[q][e] = sk && colk ? KS[q][e] : T(0);
            KL[q][e] = lk && colk ? KL[q][e] : T(0);
            KL[4 + q][e] = lk && colk2 ? KL[4 + q][e] : T(0);
          }
      }
      // the diagonal: 1 + nugget (1 for the unused slots)
      const T dS = c < k ? T(1) + epsb[c] : T(1), dL = HALF + c < k ? T(1) + epsb[HALF + c] : T(1);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < E; ++e) {
          KS[q][e] = rdiag == 4 * q + e ? dS : KS[q][e];
          KL[4 + q][e] = rdiag == 4 * q + e ? dL : KL[4 + q][e];
        }
      __syncthreads();
      rvS = cq[c];
      rvL = cq[HALF + c];
    }

    MGP_MF_T(2)
    // ---- elimination ------------------------------------------------------------------------------------------
    // column j: held by the lanes with h == (j / 4) % 2 -- short rows KS[(j % 32) / 8][j % 4] while j < 32, long rows
    // KL[4 (j / 32) + (j % 32) / 8][j % 4].  Column buffer j & 1 holds it: the look-ahead posts column j + 1 into the
    // other one while this step's groups are still being read.
#if MGP_RHS_MF_PRIO >= 2
    __builtin_amdgcn_s_setprio(2);
#endif
    T myu = T(0), myw = T(0);  // u_i = (L^-1 c)_i and u_i / p_i of row i = this lane
    T pmin = num<T>::inf();
    V mL = V(0), mS = V(0);
    const bool h0 = h == 0;
    auto colgrp = [&](int j) { return 4 * (j / HALF) + (j % HALF) / 8; };  // index into KL (KS: the same without the 4)
    // where a half posts its entries of a column: into the column buffer when it holds the column, else into the dead
    // cross-covariance / nugget arrays (branch-free; two per-lane bases, everything else immediate offsets)
    T* const post0 = (h0 ? colbuf : cq) + c;  // columns held by the lower half
    T* const post1 = (h0 ? cq : colbuf) + c;  // ... by the upper half
    post0[0] = KS[0][0];  // column 0
    post0[HALF] = KL[0][0];
#if MGP_RHS_MF_TRAIL
    // The first 32 steps' updates of the trailing block (rows and columns 32 .. 63) are a rank-32 update, A22 -= W L^T with
    // W the numerators and L the multipliers of the long rows: GEMM-shaped, and KL[4 .. 7] IS a 32 x 32 accumulator in
    // the matrix instruction's layout (that is how it was made).  One v_mfma_f32_32x32x2_f32 per two steps -- the lower
    // half supplies the even step's (numerator, -multiplier), the upper half the odd step's -- instead of 16 packed FMAs
    // and 8 column-group reads: -256 VALU and -128 LDS instructions per neighbourhood, on a pipe that idles.
    F16 KH;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < E; ++e) KH[4 * q + e] = KL[4 + q][e];
    T aE = T(0), tE = T(0);
#endif
    f2 rv2 = f2{rvL, rvS};
    V pg = *reinterpret_cast<const V*>(colbuf);
    T aS = colbuf[c], aL = colbuf[HALF + c];
    constexpr int JB = 8;
#pragma unroll
    for (int jb = 0; jb < NP; jb += JB) {
      if (jb < k)
#pragma unroll
      for (int j = jb; j < jb + JB; ++j) {
        const bool sh = j < HALF;  // (compile-time after unrolling) the short rows are still being eliminated
        T* cb = colbuf + (j & 1) * NP;
        T* cbn = colbuf + ((j + 1) & 1) * NP;
        const T bj = lane_value(sh ? rv2.y : rv2.x, j % HALF);  // right-hand side of row j (both halves carry both rows')
        const T p = pg[j % E];
#if !MGP_RHS_MF_WRITELANE
        pmin = __builtin_fminf(pmin, p);
#endif
        const T rp = pivot_rcp(p);
        const T tL = aL * rp, tS = sh ? aS * rp : T(0);
        const V ntL = V(-tL), ntS = V(-tS);
#if MGP_RHS_MF_WRITELANE
        // u_j and p_j are wave-uniform: into lane j of the capture registers by v_writelane (u_j / p_j follows after
        // the loop) instead of a compare, two selects and a multiply
        {
          const int ub = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, bj));
          const int pb = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p));
          asm("v_writelane_b32 %0, %1, %2" : "+v"(myu) : "s"(ub), "n"(j));
          asm("v_writelane_b32 %0, %1, %2" : "+v"(myw) : "s"(pb), "n"(j));
        }
#else
        myu = i == j ? bj : myu;
        myw = i == j ? bj * rp : myw;
#endif
        // the lane's column groups: long group G covers columns 32 (G / 4) + 8 (G % 4) + 4 h + (0..3), short group q
        // the same below 32.  A group is touched while one of its columns (in either half) lies right of the pivot.
        const int j1 = j + 1 < NP ? j + 1 : j;
        const int G1 = colgrp(j1);
#if MGP_RHS_MF_TRAIL
        if (j == HALF) {  // the trailing block is complete: back to the row groups
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < E; ++e) KL[4 + q][e] = KH[4 * q + e];
        }
        if (sh) {
          if (j % 2 == 0) {
            aE = aL;
            tE = tL;
          } else {
            KH = __builtin_amdgcn_mfma_f32_32x32x2f32(h0 ? aE : aL, h0 ? -tE : -tL, KH, 0, 0, 0);
          }
        }
        constexpr bool TRAIL = true;
#else
        constexpr bool TRAIL = false;
#endif
        auto colv = [&](int G) { return *reinterpret_cast<const V*>(cb + HALF * (G / 4) + 8 * (G % 4) + 4 * h); };
        if (!(TRAIL && sh && G1 >= 4)) {  // (TRAIL, step 31: column 32 belongs to the trailing block -- the MFMA above)
          const V cv = colv(G1);
          KL[G1] = cv * ntL + KL[G1];
          if (sh && G1 < 4) KS[G1 < 4 ? G1 : 0] = cv * ntS + KS[G1 < 4 ? G1 : 0];
        }
        if (j + 1 < NP) {  // look-ahead: column j + 1 is complete -- post it, ask for its pivot group and own entries
          T* const post = (j1 / E) % 2 == 0 ? post0 : post1;
          if (TRAIL && sh && G1 >= 4) post[(j1 & 1) * NP + HALF] = KH[4 * (G1 - 4) + j1 % E];
          else post[(j1 & 1) * NP + HALF] = KL[G1][j1 % E];
          if (j1 < HALF) post[(j1 & 1) * NP] = KS[G1 < 4 ? G1 : 0][j1 % E];
          pg = *reinterpret_cast<const V*>(cbn + (j1 / E) * E);
          aL = cbn[HALF + c];
          if (j1 < HALF) aS = cbn[c];
        }
#pragma unroll
        for (int G = 0; G < 8; ++G) {
          if (G != G1 && HALF * (G / 4) + 8 * (G % 4) + 7 > j && !(TRAIL && sh && G >= 4)) {
            const V cv = colv(G);
            KL[G] = cv * ntL + KL[G];
            if (sh && G < 4) KS[G < 4 ? G : 0] = cv * ntS + KS[G < 4 ? G : 0];
          }
        }
        rv2 = rv2 - f2{tL, tS} * f2{bj, bj};  // (tS = 0 once the short rows are done)
        mL[j % E] = tL;
        mS[j % E] = tS;
        if (j % E == E - 1) {
          // four steps' multipliers of the lane's two rows to the packed L (both lanes of a pair write the same words;
          // a row that ends left of the group writes to the 16 spare bytes behind the last row: branch-free)
          const int g0 = j / E;
          T* dump = tile + KTRI - E;
          *reinterpret_cast<V*>(g0 * E < HALF || g0 * E < HALF + c ? tile + troff(HALF + c) + g0 * E : dump) = mL;
          if (sh) *reinterpret_cast<V*>(g0 * E < c ? tile + troff(c) + g0 * E : dump) = mS;
        }
#if MGP_RHS_MF_SCHED
        // (keeps the scheduler from interleaving the bulk updates of several steps: their column copies are what
        // spills at three waves per SIMD; the other waves supply the parallelism)
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
    }
#if MGP_RHS_MF_PRIO < 3
    __builtin_amdgcn_s_setprio(0);
#endif
#if MGP_RHS_MF_WRITELANE
    // lane i holds its row's pivot: a non-positive (or NaN) one anywhere marks the neighbourhood (no running minimum)
    if (__builtin_amdgcn_ballot_w64(i < k && !(myw > T(0))) != 0) pmin = T(-1);
    myw = i < k ? myu * pivot_rcp(myw) : T(0);  // (unused slots: their steps may have been skipped)
#endif
    __syncthreads();

    MGP_MF_T(3)
    // ---- outputs: lane i is row i again -------------------------------------------------------------------------
    T y[RC];
#pragma unroll
    for (int r = 0; r < RC; ++r) y[r] = T(0);
    if (i < k) {  // the responses fly under the back-substitution
      const T* ty = targets + yrow * (int64_t)g.row_resp;
      if (g.resp_vec) {
#pragma unroll
        for (int r4 = 0; r4 < RC / E; ++r4) {
          const V v = *reinterpret_cast<const V*>(ty + r4 * E);
#pragma unroll
          for (int e = 0; e < E; ++e) y[r4 * E + e] = v[e];
        }
      } else {
#pragma unroll
        for (int r = 0; r < RC; ++r)
          if (r < R) y[r] = ty[r];
      }
    }
    T* mean = static_cast<T*>(a.mean);
    T* var = static_cast<T*>(a.var);
    const T sv = wave_sum_lane63(myu * myw);
    const bool bad = !(pmin > T(0)) || !(sv == sv);
    // w = K^-1 c: L^T w = D^-1 u, from the last row up; lane i takes l_mi w_m off for every m > i (mgp_fused_rhs.hip)
    T wv = myw;
    constexpr int BB = MGP_RHS_MF_BLOCK;
#pragma unroll
    for (int mb = NP - BB; mb >= 0; mb -= BB) {
      if (mb < k) {  // (uniform)
        T lm[BB];
#pragma unroll
        for (int e = 0; e < BB; ++e) lm[e] = tile[troff(mb + e) + i];  // (junk for i >= mb + e or i >= k: masked below)
#pragma unroll
        for (int e = BB - 1; e >= 0; --e) {
          const int m = mb + e;
          if (m >= 1) {
            const T wm = lane_value(wv, m);
            if (i < min(m, k)) wv = fma_t(-lm[e], wm, wv);
          }
        }
      }
    }
#if MGP_RHS_MF_PRIO >= 3
    __builtin_amdgcn_s_setprio(0);
#endif
    MGP_MF_T(4)
    T sm[RC];
#pragma unroll
    for (int r = 0; r < RC; ++r) sm[r] = wave_sum_lane63(wv * y[r]);  // (rows >= k and responses >= R carry zeros)
    if (i == NP - 1) {
      var[nb] = bad ? num<T>::nan() : T(1) - sv;
      if (bad && a.info) atomicAdd(a.info, 1);
      const bool vec = g.resp_vec && (reinterpret_cast<uintptr_t>(mean) % 16 == 0);  // (then R == RC, whole 16-byte groups)
      if (vec) {
#pragma unroll
        for (int r4 = 0; r4 < RC / E; ++r4) {
          V v;
#pragma unroll
          for (int e = 0; e < E; ++e) v[e] = bad ? num<T>::nan() : sm[r4 * E + e];
          *reinterpret_cast<V*>(mean + nb * R + r4 * E) = v;
        }
      } else {
#pragma unroll
        for (int r = 0; r < RC; ++r)
          if (r < R) mean[nb * R + r] = bad ? num<T>::nan() : sm[r];
      }
    }
    MGP_MF_T(5)
  }
#if MGP_RHS_MF_TIMING
  if (threadIdx.x == 0)
    for (int t = 0; t < 8; ++t) atomicAdd(&g_rhs_mf_timing[t], tacc_[t]);
#endif
}

// -> MGP_OK, or MGP_EUNSUPPORTED when the shape is not this kernel's (the caller goes on to the other variants)
int launch_fused_rhs_mf(const FusedArgs& a, hipStream_t stream) {
  constexpr int NP = 64, E = 4, CH = 8, RC = 16;
  constexpr int KTRI = E * ((NP - 1) / E + 1) * (E * ((NP - 1) / E) / 2 + (NP - 1) % E) + NP + E;
  const int dpad = (a.d + CH - 1) / CH * CH;
  const bool packed = a.packed_nn != nullptr;
  const uintptr_t align = packed ? ((uintptr_t)a.packed_q | (uintptr_t)a.packed_nn | (uintptr_t)a.q_stride | (uintptr_t)a.nn_stride)
                                 : ((uintptr_t)a.feat_q | (uintptr_t)a.feat_nn);
  if (a.k > NP || a.R > RC || a.ykinvy != nullptr || a.d % E != 0 || a.d < CH || dpad > 64 ||
      align % 16 != 0 || a.kernel_id == MGP_KERNEL_MATERN_05 || a.kernel_id == MGP_KERNEL_MATERN_GEN)
    return MGP_EUNSUPPORTED;
  if (packed && (!a.packed_q || (!a.targets_batch && a.nn_stride < (int64_t)((a.d + a.R) * sizeof(float))))) return MGP_EINVAL;
  // rows of 41 .. 48 features: the tile is 13.5 KB, eleven workgroups per CU (one SIMD a wave short) -- measured 86.8
  // M/s against 94.2 of the folded variant (mgp_fused_rhs.hip), which takes them; d <= 40: 104.6 against 101.9; d = 64
  // (eight workgroups per CU either way): 79.8 against 74.5 of the three-wave variant
  if (dpad == 48 && a.b >= 2) return MGP_EUNSUPPORTED;
  RhsMfGeom g;
  g.dst = dpad;
  g.xs = g.dst + E;
  g.ntasks = a.b;
  FusedArgs ak = a;  // (the kernel's copy: table bases in feat_q / feat_nn / targets whatever the table form)
  if (packed) {
    if (a.nn_stride / 4 > INT32_MAX || a.q_stride / 4 > INT32_MAX) return MGP_EUNSUPPORTED;
    ak.feat_nn = a.packed_nn;
    ak.feat_q = a.packed_q;
    g.row_nn = (int)(a.nn_stride / (int64_t)sizeof(float));
    g.row_q = (int)(a.q_stride / (int64_t)sizeof(float));
  } else {
    g.row_nn = g.row_q = a.d;
  }
  if (packed && !a.targets_batch) {  // the responses ride behind the features of the neighbour's row
    ak.targets = static_cast<const float*>(a.packed_nn) + a.d;
    g.row_resp = g.row_nn;
  } else {
    g.row_resp = a.R;
  }
  g.resp_vec = a.R == RC && (uintptr_t)ak.targets % 16 == 0 && ((size_t)g.row_resp * sizeof(float)) % 16 == 0;
  const size_t tile_elems = (size_t)(NP + 1) * g.xs > (size_t)KTRI ? (size_t)(NP + 1) * g.xs : (size_t)KTRI;
  size_t lds = (tile_elems + 4 * NP + g.dst + (g.dst & 1)) * sizeof(float);
  lds = (lds + 15) & ~(size_t)15;
  static Residency res;
  int per_cu = 0, cus = 0;
  const int rc = res.lookup(reinterpret_cast<const void*>(&fused_rhs_mf_kernel<RC>), 64, lds, &per_cu, &cus);
  if (rc != MGP_OK) return rc;
  static const int env_per_cu = getenv("MGP_RHS_PER_CU") ? atoi(getenv("MGP_RHS_PER_CU")) : 0;  // occupancy experiments
  if (env_per_cu > 0 && env_per_cu < per_cu) per_cu = env_per_cu;
  static const bool trace = getenv("MGP_TRACE") != nullptr;
  if (trace) fprintf(stderr, "[mgp] fused_rhs_mf_kernel<%d>: lds %zu B, %d workgroups per CU\n", RC, lds, per_cu);
  int64_t grid = (int64_t)cus * per_cu;
  if (grid > g.ntasks) grid = g.ntasks;
  hipLaunchKernelGGL((fused_rhs_mf_kernel<RC>), dim3((unsigned)grid), dim3(64), lds, stream, ak, g);
  MGP_HIP_CHECK_LAUNCH();
  note_launch("mgp::fused_rhs_mf_kernel<%d>%s", RC, packed ? " [prepared tables]" : "");  // (the symbol rocprofv3 lists)
  note_launch_geometry(grid, lds);
  return MGP_OK;
}

}  // namespace mgp

// Register-resident fused kernel, "right-hand sides as columns" form: nn_count up to 64 with
// any number of responses (BASELINE config 5: k = 64, R = 16).
//
// mgp_fused_wave.hip appends the query and the R responses as extra ROWS of the local system,
// which needs k + 1 + R lanes.  Here all 64 lanes are neighbour rows and the cross-covariance
// c and the responses Y are carried as 1 + R extra COLUMNS, i.e. 1 + R registers per lane
// (lane i holds c_i and Y[i][:]).  Elimination step j (LDL^T, right-looking, row per lane):
//
//     all lanes post A_i[j]            -> column broadcast (as in the row form)
//     lane j posts its 1+R rhs values  -> rhs broadcast
//     lane i > j:  t = A_i[j] / d_j;   A_i[c] -= t A_c[j] (c > j);   rhs_i[:] -= t rhs_j[:]
//
// After k steps lane j holds u_j = (L^-1 c)_j and (L^-1 y_r)_j, and with its pivot d_j
//     var = Kout - sum_j u_j^2 / d_j,  mean_r = sum_j u_j uy_jr / d_j,  y_r^T K^-1 y_r = sum_j uy_jr^2 / d_j
// are cross-lane sums.  Same maths as SURVEY.md sec. 8a rows S1-S3
// (_src/gp/muygps/numpy.py:17-67, _src/optimize/scale/numpy.py:9-15), one factorisation.
//
// One wave (= one workgroup) per neighbourhood; phases 0-3 as in mgp_fused_wave.hip (register
// staged row-walking gather; cyclic difference-form distances among the k rows plus one
// crosswise distance per lane; covariances exchanged through LDS into row-per-lane registers).
#include <cstdio>
#include <cstdlib>
#include "mgp_wave_common.h"

// issue priorities per phase: covariances / exchange / elimination before the other wave's distances
// (measured here: the three-level order of the wave kernels is 1.6 % slower for this kernel)
#ifndef MGP_RHS_PRIO
#define MGP_RHS_PRIO 2
#endif
#ifndef MGP_RHS_DIST_PRIO
#define MGP_RHS_DIST_PRIO 0
#endif
#ifndef MGP_RHS_XCHG_PRIO
#define MGP_RHS_XCHG_PRIO 2
#endif

#ifndef MGP_RHS_BACK
#define MGP_RHS_BACK 1
#endif
#ifndef MGP_RHS_BACK_BLOCK
#define MGP_RHS_BACK_BLOCK 32
#endif

#ifndef MGP_RHS_TIMING
#define MGP_RHS_TIMING 0
#endif
#if MGP_RHS_TIMING
// phase timing (experiments only; tools/rhs_timing.py): s_memtime differences summed per wave and phase
__device__ unsigned long long g_rhs_timing[8];
#define MGP_RHS_T(slot)                                      \
  {                                                          \
    const unsigned long long tnow_ = __builtin_readcyclecounter(); \
    tacc_[slot] += tnow_ - tlast_;                           \
    tlast_ = tnow_;                                          \
  }
extern "C" int mgp_debug_rhs_timing(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rhs_timing), sizeof(g_rhs_timing)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_rhs_timing), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#else
#define MGP_RHS_T(slot)
#endif

namespace mgp {

struct RhsGeom {
  int dst, xs, vec_ok;
  int resp_vec;  // all RC responses of a row in 16-byte loads (R == RC, rows 16-byte aligned)
  int64_t ntasks;
};

// BACK (prediction: no y^T K^-1 y requested, more than one response): only the cross-covariance column is
// carried through the elimination; the multipliers are kept (one coalesced ds_write_b32 per step into
// the by then free exchange matrix, row = step, odd stride -> the transposed reads of the back-
// substitution are conflict-free), w = K^-1 c follows from L^T w = D^-1 L^-1 c in k - 1 v_readlane + FMA
// steps, and mean_r = w . y_r are R plain dot products.  Per elimination step that is 1 instead of 1 + R
// v_readlane_b32 and 1 instead of (1 + R) / 2 packed FMAs on the right-hand sides: at R = 16, 24 of ~90
// instructions per step.
// GRAM (fp32, one feature stage, not Matern-1/2): rows centred on the query in place, pair
// distances as |a'|^2 + |b'|^2 - 2 a'.b' -- one packed FMA per two features of a pair instead of a packed
// subtract and a packed FMA -- and the crosswise distance of a lane is its row's norm, exactly.
// W3 (round 4; fp32 prediction variant with the Gram form; serves rows longer than d = 48, FOLD below the others):
// held to three waves per SIMD.  BOTH limits of the plain kernel sat at eight workgroups per CU (registers and 18.5 KB
// of LDS).  Here (i) the exchange matrix is packed lower-triangular (8.7 instead of 17.4 KB: a lane's row of the
// system only ever needs its lower triangle), (ii) the multipliers are not stored to LDS step by step but kept IN PLACE
// -- entry (i, j) of a lane's row is dead once column j is eliminated -- and the rows are dumped once, in the same
// packed layout, where the back-substitution reads L column-wise with consecutive lanes on consecutive addresses,
// (iii) the column of a step is consumed in two halves instead of copied whole.  Measured on config 5 (MGP_RHS_PER_CU
// = 4 / 6 / 8 / 10 / 12 workgroups per CU): 54.0 / 68.2 / 85.8 / 80.7 / 91.5 M/s -- the third wave is worth 6.6 %.
#ifndef MGP_RHS_W3_WAVES
#define MGP_RHS_W3_WAVES 3
#endif
#ifndef MGP_RHS_W3_SCHED
#define MGP_RHS_W3_SCHED 0
#endif
#ifndef MGP_RHS_W3_HG
#define MGP_RHS_W3_HG 8
#endif
#ifndef MGP_RHS_ATTR
#define MGP_RHS_ATTR
#endif
#ifndef MGP_RHS_FOLD_SCHED
#define MGP_RHS_FOLD_SCHED 0
#endif
#ifndef MGP_RHS_FOLD_UNROLL
#define MGP_RHS_FOLD_UNROLL 1
#endif
#ifndef MGP_RHS_FOLD_U
#define MGP_RHS_FOLD_U 6
#endif
#ifndef MGP_RHS_FOLD_HG
#define MGP_RHS_FOLD_HG 8
#endif
#ifndef MGP_RHS_FOLD_BLOCK
#define MGP_RHS_FOLD_BLOCK 16
#endif
// FOLD (round 4; same variant): TWO neighbourhoods per wave share one elimination.  Row-per-lane elimination keeps
// all 64 lanes busy on every column right of the pivot although a row only needs its lower triangle and rows above
// the pivot are finished: 544 16-byte group updates per neighbourhood where the factorisation needs a third of that.
// Folded (as the 32-slot wave kernels, mgp_fused_wave_kernel.h phase 4F): lane l of a half-wave owns row l -- columns
// 0 .. 31 matter, 8 groups -- and row 32 + l (16 groups) of ONE neighbourhood; the two half-waves hold two
// neighbourhoods.  A step serves both with the same instructions (every LDS read is one broadcast per half-wave):
// 688 group updates per pair instead of 1088, and the per-step overhead (reciprocal, multipliers, posts, right-hand
// side) once per pair.  The first neighbourhood of a pair runs its gather / distance / covariance phases on all 64
// lanes as before, parks its two rows per lane in lanes 0 .. 31 (96 registers) and the wave goes on to the second.
// Nothing is masked during the elimination: finished rows keep computing on dead entries (their posts only ever
// reach columns that are finished too), the right-hand side of a row is captured the step it is broadcast, the
// multipliers stay in place (W3) and are dumped once per neighbourhood for the two interleaved back-substitutions.
template <typename T, int RC, bool BACK = false, bool GRAM = false, bool W3 = false, bool FOLD = false>  // RC: compiled number of response columns (run-time R <= RC)
__global__ __launch_bounds__(64, (sizeof(T) == 4 ? (W3 ? MGP_RHS_W3_WAVES : 2) : 1)) MGP_RHS_ATTR void fused_rhs_kernel(FusedArgs a, RhsGeom g) {
  static_assert(!W3 || (sizeof(T) == 4 && BACK), "W3: the fp32 prediction variant");
  static_assert(!FOLD || (sizeof(T) == 4 && BACK && GRAM && !W3), "FOLD: the fp32 prediction variant with the Gram form");
  constexpr int NP = 64;
  constexpr int NS = NP / 2;
  // register blocking of the pair scheme (mgp_fused_wave.hip, phase 2): BA own rows x BP partners
  constexpr int BA = 4;
  constexpr int BP = NS / BA;
  static_assert(BA == 4, "the guarded Gram finish below is written for four own rows");
  auto own_offset = [](int j) { return j == 0 ? 0 : (j + 1) * BP + 1; };
  constexpr int E = v16<T>::N;
  constexpr int CH = 2 * E;
  constexpr int KS = NP + E;
  // (W3) packed lower-triangular exchange matrix: row r holds its r + 1 entries padded to whole 16-byte groups, at
  // rowoff(r) = E (a + 1) (E a / 2 + r % E), a = r / E (as the 64-slot wave kernels, mgp_fused_wave_kernel.h)
  auto rowoff = [](int r) {
    if constexpr (W3) {
      const int a_ = r / E;
      return E * (a_ + 1) * (E * a_ / 2 + r % E);
    } else {
      return r * KS;
    }
  };
  auto troff = [](int r) { return E * (r / E + 1) * (E * (r / E) / 2 + r % E); };  // the packed layout, whatever the variant
  constexpr int KTRI = E * ((NP - 1) / E + 1) * (E * ((NP - 1) / E) / 2 + (NP - 1) % E) + NP + E;
  constexpr int KMAT = W3 ? KTRI : (FOLD && 2 * KTRI > NP * KS ? 2 * KTRI : NP * KS);  // (FOLD: later the two packed L)
  constexpr int HALF = NP / 2;
  constexpr int NSYS = FOLD ? 2 : 1;  // neighbourhoods per pass of the persistent loop
  constexpr int NR = 1 + RC;                       // rhs columns: cross-covariance + responses
  constexpr int NRV = (NR + E - 1) / E;            // ... in 16-byte groups
  using V = typename v16<T>::type;
  using ACC = typename v16<T>::acc;

  extern __shared__ __attribute__((aligned(16))) char smem[];
#if MGP_RHS_TIMING
  unsigned long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast_ = __builtin_readcyclecounter();
#endif
  const int k = a.k, d = a.d, R = a.R, xs = g.xs, dst = g.dst;
  const int rows_x = NP + 1;                       // tile rows: 64 slots + the query
  const int tile_need = rows_x * xs + (FOLD ? HALF * HALF : 0);  // (FOLD: + the parked short rows)
  const int tile_elems = tile_need > KMAT ? tile_need : KMAT;
  T* tile = reinterpret_cast<T*>(smem);            // feature tile, later the exchange matrix
  T* colbuf = tile + tile_elems;                   // 64 (W3: two column buffers; FOLD: one per neighbourhood)
  T* rhsbuf = colbuf + (FOLD ? 4 * NP : W3 ? 2 * NP : NP); // NRV * E (FOLD: 2 * 64 -- the cross-covariances on their way to the row owners)
  T* ilbuf = rhsbuf + (FOLD ? 2 * NP : NRV * E);   // dst
  int64_t* idxbuf = reinterpret_cast<int64_t*>(ilbuf + dst + (dst & 1));  // 65

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

  // Index prefetch (as in the wave kernels): the neighbour index of task t + 1 is requested at the top of task t, so
  // that the row gather of a task does not start with a dependent global round trip (two waves per SIMD: ~1.5 us of a
  // ~22 us task that nothing hid).  One branch-free load per lane (lanes behind k read a valid dummy entry).
  // (FOLD) a pass takes the pair of tasks 2 un, 2 un + 1; an odd last task is paired with itself (and written once)
  const int64_t nunits = FOLD ? (g.ntasks + 1) / 2 : g.ntasks;
  auto task_of = [&](int64_t un, int sys) {
    const int64_t t = un * NSYS + sys;
    return t < g.ntasks ? t : g.ntasks - 1;
  };
  int64_t next_idx = 0, next_q = 0;
  if ((int64_t)blockIdx.x < nunits) {
    const int64_t t0 = task_of(blockIdx.x, 0);
    next_idx = a.nn_idx[t0 * k + ((int)threadIdx.x < k ? (int)threadIdx.x : 0)];
    next_q = a.batch_idx ? a.batch_idx[t0] : t0;
  }
  for (int64_t un = blockIdx.x; un < nunits; un += gridDim.x) {
    int i = threadIdx.x;
    asm volatile("" : "+v"(i));  // keep per-lane addresses out of LICM (register pressure)
    T rhs[NR];
    V A[FOLD ? 1 : NP / E];
    // (FOLD) the folded rows of the lane's neighbourhood -- FS: row l, columns 0 .. 31; FL: row 32 + l -- their
    // right-hand sides, and the rows of the response table of the two neighbourhoods
    V FL[FOLD ? NP / E : 1], FS[FOLD ? HALF / E : 1];
    T rvS = T(0), rvL = T(0);
    int64_t yrow0 = 0, yrow1 = 0;
    int64_t nb = 0;
#if MGP_RHS_FOLD_UNROLL
#pragma unroll
#else
#pragma nounroll
#endif
    for (int sys = 0; sys < NSYS; ++sys) {
    nb = task_of(un, sys);

    // ---- indices, nugget, responses ------------------------------------------------------
    const int64_t myidx = i < k ? next_idx : 0;
    const int64_t qidx = next_q;
    {
      // the next task of this workgroup's sequence: the pair's second, or the first of the next pass
      const bool in_unit = sys + 1 < NSYS;
      const int64_t un2 = in_unit ? un : un + gridDim.x;
      if (un2 < nunits) {
        const int64_t nn = task_of(un2, in_unit ? sys + 1 : 0);
        next_idx = a.nn_idx[nn * k + (i < k ? i : 0)];
        next_q = a.batch_idx ? a.batch_idx[nn] : nn;
      }
    }
    __syncthreads();
    idxbuf[i] = myidx * (int64_t)d;
    if (i == 0) idxbuf[NP] = qidx * (int64_t)d;
    T myeps = T(0);
    if (i < k) {
      if (a.noise_mode == MGP_NOISE_SCALAR) myeps = (T)a.noise_scalar;
      else if (a.noise_mode == MGP_NOISE_TABLE) myeps = noise_dev[myidx];
      else myeps = noise_dev[nb * k + i];
    }

#if MGP_RHS_PRIO
    __builtin_amdgcn_s_setprio(MGP_RHS_DIST_PRIO);
#endif
    // ---- gather + distances (pairwise: cyclic scheme; crosswise: lane i vs the query) ------
    ACC acc[NS];
    ACC accq = ACC(0);
    // (zeroed HERE, not under `d0 == 0` inside the loop: a conditional first store makes the 64 accumulator registers
    // a value carried around the persistent loop -- live, untouched, through the whole elimination)
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = ACC(0);
    for (int d0 = 0; d0 < d; d0 += dst) {
      const int w = min(dst, d - d0);
      const int wp = (w + CH - 1) / CH * CH;
      __syncthreads();
      if constexpr (FOLD) {
        // Rows straight from global memory into LDS (global_load_lds, as the wave kernels: no VGPR round trip, no
        // ds_write, and every 1-KiB piece of the tile in flight at once -- one round trip instead of two dependent ones).
        // 16-byte slot sigma = 64 n + lane of the tile: row = sigma / SPR, column = sigma % SPR; the slots behind the
        // features of a row re-read its last group (the norm overwrites the first of them), the slots of unused rows
        // and behind the query row are masked off (what lies behind the tile rows are parked half rows).
        const int SPR = xs / E;
        const unsigned total = (unsigned)(rows_x * SPR);
        const unsigned smagic = (1u << 20) / (unsigned)SPR + 1u;
        const int c16 = w / E;
        for (unsigned n = 0; n * 64u < total; ++n) {
          const unsigned sigma = 64u * n + (unsigned)i;
          const unsigned row = (sigma * smagic) >> 20;
          const unsigned c = sigma - row * (unsigned)SPR;
          const bool on = sigma < total && ((int)row < k || row == (unsigned)NP);
          const T* src = ((int)row < k ? feat_nn : feat_q) + idxbuf[(int)row < k ? row : NP] + d0 + min((int)c, c16 - 1) * E;
          if (on) glds16_lds(src, smem, (int)n * 1024);
        }
        for (int t = i; t < (NP - k) * SPR; t += NP)  // zero rows for the unused slots
          *reinterpret_cast<V*>(tile + (k + t / SPR) * xs + (t % SPR) * E) = V(0);
        if (wp > w) {  // (uniform) the padding group of the 8-wide inner loop: zero, once the rows have landed
          lds_dma_wait();
          *reinterpret_cast<V*>(tile + i * xs + w) = V(0);
          if (i == 0) *reinterpret_cast<V*>(tile + NP * xs + w) = V(0);
        }
      } else if (g.vec_ok) {
        const int c16 = w / E, c16p = wp / E;
        const int rpr = NP / c16p;
        const int sub = (int)(((unsigned)i * ((1u << 16) / (unsigned)c16p + 1u)) >> 16);
        const int c = i - sub * c16p;
        const bool lane_on = sub < rpr;
        // rows in flight per lane and round.  (FOLD: twelve would make the 65 rows of d = 40 ONE round trip instead of
        // two -- measured 93.1 against 98.4 M/s with six: the gather's share of a wave's life, a fifth
        // (tools/rhs_timing.py), is time the other wave uses, and the 24 extra registers spill elsewhere)
        constexpr int U = FOLD ? MGP_RHS_FOLD_U : 6;
        for (int r0 = 0; r0 <= k; r0 += U * rpr) {  // row k of this loop is the query (tile row NP)
          V v[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int row = r0 + u * rpr + sub;
            v[u] = V(0);
            if (lane_on && row <= k && c < c16)
              v[u] = *reinterpret_cast<const V*>((row < k ? feat_nn : feat_q) + idxbuf[row < k ? row : NP] + d0 + c * E);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int row = r0 + u * rpr + sub;
            if (lane_on && row <= k) *reinterpret_cast<V*>(tile + (row < k ? row : NP) * xs + c * E) = v[u];
          }
        }
        for (int t = i; t < (NP - k) * c16p; t += NP)  // zero rows for the unused slots
          *reinterpret_cast<V*>(tile + (k + t / c16p) * xs + (t % c16p) * E) = V(0);
      } else {
        const unsigned magic = (1u << 20) / (unsigned)wp + 1u;
        for (int t = i; t < rows_x * wp; t += NP) {
          const int row = (int)(((unsigned)t * magic) >> 20);
          const int c = t - row * wp;
          T v = T(0);
          if (c < w && (row < k || row == NP)) v = ((row < k ? feat_nn : feat_q) + idxbuf[row] + d0)[c];
          tile[row * xs + c] = v;
        }
      }
      if (aniso)
        for (int c = i; c < wp; c += 64) ilbuf[c] = c < w ? T(1) / ls[d0 + c] : T(0);
      if constexpr (FOLD) lds_dma_wait();  // the rows have landed
      __syncthreads();
      MGP_RHS_T(0)
      const T* xq = tile + NP * xs;
      if constexpr (GRAM) {
        // centre row i on the query (times the inverse length scales), in place; |a'|^2 behind the row (column dst)
        {
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
          const T nrm = acc_total(n2[0] + n2[1]);
          xrow[dst] = i < k ? nrm : num<T>::inf();  // (unused slots: an infinite norm keeps their pairs away from the cancellation guard)
          accq = ACC(0);
          accq.x = nrm;  // the crosswise squared distance
        }
        __syncthreads();
        for (int c0 = 0; c0 < wp; c0 += CH) {
          V own0[BA], own1[BA];
#pragma unroll
          for (int j = 0; j < BA; ++j) {
            const T* xj = tile + ((i + own_offset(j)) & (NP - 1)) * xs + c0;
            own0[j] = *reinterpret_cast<const V*>(xj);
            own1[j] = *reinterpret_cast<const V*>(xj + E);
          }
          static_for<BP>([&](auto sc) {
            constexpr int s = decltype(sc)::value + 1;
            const T* xo = tile + ((i + s) & (NP - 1)) * xs + c0;
            const V o0 = *reinterpret_cast<const V*>(xo), o1 = *reinterpret_cast<const V*>(xo + E);
            gram_block<BA, BP>(&acc[s - 1], own0, o0);
            gram_block<BA, BP>(&acc[s - 1], own1, o1);
          });
        }
        T nown[BA];
#pragma unroll
        for (int j = 0; j < BA; ++j) nown[j] = tile[((i + own_offset(j)) & (NP - 1)) * xs + dst];
        T guard = T(1);
        static_for<BP>([&](auto sc) {
          constexpr int s = decltype(sc)::value;
          const T npar = tile[((i + s + 1) & (NP - 1)) * xs + dst];
          gram_finish2(acc[s], acc[BP + s], nown[0] + npar, nown[1] + npar, guard);
          gram_finish2(acc[2 * BP + s], acc[3 * BP + s], nown[2] + npar, nown[3] + npar, guard);
        });
        // the cancellation guard tripped (mgp_wave_common.h): this neighbourhood's pair distances
        // again in the difference form, on the same centred rows, pair by pair (rare: registers before speed)
        if (gram_guard_tripped(guard)) {
          static_for<NS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            const T* xa = tile + ((i + own_offset(s / BP)) & (NP - 1)) * xs;
            const T* xb = tile + ((i + s % BP + 1) & (NP - 1)) * xs;
            ACC sum = ACC(0);
#pragma nounroll
            for (int c0 = 0; c0 < wp; c0 += E)
              accum(sum, vsub(*reinterpret_cast<const V*>(xa + c0), *reinterpret_cast<const V*>(xb + c0)));
            gram_from_diff(sum);
            acc[s] = sum;
          });
        }
      } else
      for (int c0 = 0; c0 < wp; c0 += CH) {
        V own0[BA], own1[BA];
#pragma unroll
        for (int j = 0; j < BA; ++j) {
          const T* xj = tile + ((i + own_offset(j)) & (NP - 1)) * xs + c0;
          own0[j] = *reinterpret_cast<const V*>(xj);
          own1[j] = *reinterpret_cast<const V*>(xj + E);
        }
        V il0 = V(1), il1 = V(1);
        if (aniso) {
          il0 = *reinterpret_cast<const V*>(ilbuf + c0);
          il1 = *reinterpret_cast<const V*>(ilbuf + c0 + E);
        }
        const V q0 = *reinterpret_cast<const V*>(xq + c0), q1 = *reinterpret_cast<const V*>(xq + c0 + E);
        if (aniso) {
          accum(accq, vsub(own0[0], q0) * il0);
          accum(accq, vsub(own1[0], q1) * il1);
          // compile-time pair indices (an unroll the compiler declines would put acc[] in scratch)
          static_for<BP>([&](auto sc) {
            constexpr int s = decltype(sc)::value + 1;
            const T* xo = tile + ((i + s) & (NP - 1)) * xs + c0;
            const V o0 = *reinterpret_cast<const V*>(xo), o1 = *reinterpret_cast<const V*>(xo + E);
            static_for<BA>([&](auto jc) {
              constexpr int j = decltype(jc)::value;
              accum(acc[j * BP + s - 1], vsub(own0[j], o0) * il0);
              accum(acc[j * BP + s - 1], vsub(own1[j], o1) * il1);
            });
          });
        } else {
          accum(accq, vsub(own0[0], q0));
          accum(accq, vsub(own1[0], q1));
          static_for<BP>([&](auto sc) {
            constexpr int s = decltype(sc)::value + 1;
            const T* xo = tile + ((i + s) & (NP - 1)) * xs + c0;
            const V o0 = *reinterpret_cast<const V*>(xo), o1 = *reinterpret_cast<const V*>(xo + E);
            if constexpr (sizeof(T) == 4 && BA == 4) {
              // hand-ordered packed blocks: no instruction reads its predecessor's result (mgp_wave_common.h)
              dist_block4(acc[s - 1], acc[BP + s - 1], acc[2 * BP + s - 1], acc[3 * BP + s - 1], own0[0], own0[1], own0[2],
                          own0[3], o0);
              dist_block4(acc[s - 1], acc[BP + s - 1], acc[2 * BP + s - 1], acc[3 * BP + s - 1], own1[0], own1[1], own1[2],
                          own1[3], o1);
            } else {
              static_for<BA>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                accum(acc[j * BP + s - 1], vsub(own0[j], o0));
                accum(acc[j * BP + s - 1], vsub(own1[j], o1));
              });
            }
          });
        }
      }
    }

    // ---- responses of the lane's row: requested only now (1 + R registers the distance phase has
    // no room for); the loads fly while the covariances are evaluated and exchanged -----------------
    MGP_RHS_T(1)
#pragma unroll
    for (int r = 0; r < NR; ++r) rhs[r] = T(0);
    if constexpr (FOLD) {  // (requested after both neighbourhoods are parked: no room for them here)
      const int64_t yr = a.targets_batch ? nb * k + i : myidx;
      if (sys == 0) yrow0 = yr;
      else yrow1 = yr;
    } else if (i < k) {
      const T* ty = targets + (a.targets_batch ? nb * k + i : myidx) * (int64_t)R;
      if (g.resp_vec) {  // (uniform) RC / E loads of 16 bytes instead of RC single ones under RC tests
#pragma unroll
        for (int r4 = 0; r4 < RC / E; ++r4) {
          const V v = *reinterpret_cast<const V*>(ty + r4 * E);
#pragma unroll
          for (int e = 0; e < E; ++e) rhs[1 + r4 * E + e] = v[e];
        }
      } else {
#pragma unroll
        for (int r = 0; r < RC; ++r)
          if (r < R) rhs[1 + r] = ty[r];
      }
    }

    // ---- covariances -> exchange matrix -> row per lane; cross-covariance stays in the lane ----
    // from here to the end of the elimination the wave runs chains of short dependent steps: it goes
    // before the other wave's distance phase (long independent streams) at the issue arbiter
#if MGP_RHS_PRIO
    __builtin_amdgcn_s_setprio(MGP_RHS_XCHG_PRIO);
#endif
    __syncthreads();
    {
      T kv[NS];
      T kq = T(0);
      auto sqd = [](const ACC& v) {
        if constexpr (GRAM) return gram_sq(v);
        else return acc_total(v);
      };
      const T cscale = post_scale;  // (1 under Anisotropy: the rows are scaled)
      kernel_dispatch(a.kernel_id, a.metric_id, [&](auto kid, auto mid) {
        constexpr int KID = decltype(kid)::value, MID = decltype(mid)::value;
        if constexpr (sizeof(T) == 4) {
          static_for<NS / 2>([&](auto sc) {  // two covariances per packed instruction
            constexpr int s = 2 * decltype(sc)::value;
            const f2 kk = cov_from_sqdist2(f2{sqd(acc[s]), sqd(acc[s + 1])}, KID, MID, cscale);
            kv[s] = kk.x;
            kv[s + 1] = kk.y;
          });
        } else {
          static_for<NS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            kv[s] = cov_from_sqdist<T>(sqd(acc[s]), KID, MID, cscale);
          });
        }
        kq = cov_from_sqdist<T>(sqd(accq), KID, MID, cscale);
      });
      if constexpr (FOLD) {
        // the short rows of the pair's first neighbourhood come back from their parking space (below) before the
        // exchange matrix takes it; every lane reads, so that the registers were free during the distance phase
        if (sys == 1) {
#pragma unroll
          for (int c4 = 0; c4 < HALF / E; ++c4)
            FS[c4] = *reinterpret_cast<const V*>(tile + rows_x * xs + (c4 * HALF + (i & (HALF - 1))) * E);
        }
      }
      static_for<NS>([&](auto sc) {
        constexpr int s = decltype(sc)::value + 1;
        const int r1 = (i + own_offset((s - 1) / BP)) & (NP - 1);  // pair j * BP + p - 1: (own row j, partner p)
        const int c = (i + (s - 1) % BP + 1) & (NP - 1);
        const int hi = max(r1, c), lo = min(r1, c);
        tile[rowoff(hi) + lo] = hi < k ? kv[s - 1] : T(0);  // unused slots: identity rows
      });
      tile[rowoff(i) + i] = i < k ? T(1) + myeps : T(1);
      rhs[0] = i < k ? kq : T(0);
      if constexpr (FOLD) rhsbuf[i] = rhs[0];
    }
    __syncthreads();
    if constexpr (FOLD) {
      // the half-wave of this neighbourhood picks its rows up (what lies right of a row's diagonal was never written:
      // junk that only ever meets dead entries)
      // (the first of a pair: BOTH half-waves read -- no merge with older register contents -- and the short rows
      // are parked in LDS behind the feature rows of the second neighbourhood: 32 registers less under its distance
      // phase, which otherwise spills)
      if (sys == 0 || (i >> 5) == 1) {
        const int lh = i & (HALF - 1);
#pragma unroll
        for (int c4 = 0; c4 < HALF / E; ++c4) FS[c4] = *reinterpret_cast<const V*>(tile + rowoff(lh) + c4 * E);
#pragma unroll
        for (int c4 = 0; c4 < NP / E; ++c4) FL[c4] = *reinterpret_cast<const V*>(tile + rowoff(HALF + lh) + c4 * E);
        rvS = rhsbuf[lh];
        rvL = rhsbuf[HALF + lh];
      }
      if (sys == 0 && i < HALF) {
#pragma unroll
        for (int c4 = 0; c4 < HALF / E; ++c4) *reinterpret_cast<V*>(tile + rows_x * xs + (c4 * HALF + i) * E) = FS[c4];
      }
    } else {
#pragma unroll
    for (int c4 = 0; c4 < NP / E; ++c4) A[c4] = *reinterpret_cast<const V*>(tile + rowoff(i) + c4 * E);  // (W3: beyond the diagonal belongs to later rows -- never used)
    }
    MGP_RHS_T(2)
    }  // (sys)

    if constexpr (FOLD) {
      // ---- folded elimination of the pair, back-substitutions, outputs ----------------------------------------
      const int sub = i >> 5, lh = i & (HALF - 1);
      const int64_t nbA = un * 2, nbB = un * 2 + 1;
      const bool have_b = nbB < g.ntasks;
      T yA[RC], yB[RC];  // the responses of row i of both neighbourhoods
      auto load_y = [&](T (&y)[RC], int64_t row) {
#pragma unroll
        for (int r = 0; r < RC; ++r) y[r] = T(0);
        if (i < k) {
          const T* ty = targets + row * (int64_t)R;
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
      };

      T* colq = colbuf + sub * 2 * NP;  // the neighbourhood's two column buffers
      T* Lq = tile + sub * KTRI;    // ... and its packed L
      colq[HALF + lh] = FL[0][0];
      colq[lh] = FS[0][0];
      V piv = *reinterpret_cast<const V*>(colq);
      V mL = V(0), mS = V(0);  // the multipliers of the current 16-byte group of columns
      f2 rv2 = f2{rvL, rvS};
      T uS = T(0), uL = T(0), wS = T(0), wL = T(0);  // u_r = (L^-1 c)_r and u_r / p_r of the lane's two rows
      T pmin = num<T>::inf();
      const int bpa = (i & 32) << 2;  // ds_bpermute: byte address of lane 0 of the half-wave
      constexpr int JB = 8;
#pragma unroll
      for (int jb = 0; jb < NP; jb += JB) {
        if (jb < k)
#pragma unroll
        for (int j = jb; j < jb + JB; ++j) {
          constexpr int NG = NP / E, NGS = HALF / E;
          const int g0 = j / E, e0 = j % E;
          const bool sh = j < HALF;  // (compile-time after unrolling) the short rows are still being eliminated
          const T aL = FL[g0][e0];
          const T aS = sh ? FS[g0 < NGS ? g0 : 0][e0] : T(0);
          // right-hand side of row j, to every lane of its half-wave
          const T bj = __builtin_bit_cast(T, __builtin_amdgcn_ds_bpermute(bpa + 4 * (sh ? j : j - HALF),
                                                                         __builtin_bit_cast(int, sh ? rv2.y : rv2.x)));
          // the column in chunks of HG groups (a whole-column copy on top of the 96 row registers spills); column j
          // lives in buffer j & 1, so its later chunks can still be read after the look-ahead has posted column j + 1
          constexpr int HG = MGP_RHS_FOLD_HG;
          T* cb = colq + (j & 1) * NP;
          T* cbn = colq + ((j + 1) & 1) * NP;
          const T p = piv[e0];
          pmin = __builtin_fminf(pmin, p);
          const T rp = pivot_rcp(p);
          const T tL = aL * rp, tS = aS * rp;
          const V ntL = V(-tL), ntS = V(-tS);
          const bool own = lh == (sh ? j : j - HALF);  // this lane owns row j
          const T w0 = bj * rp;
          if (sh) {
            uS = own ? bj : uS;
            wS = own ? w0 : wS;
          } else {
            uL = own ? bj : uL;
            wL = own ? w0 : wL;
          }
          const int g1 = (j + 1 < NP ? j + 1 : j) / E;
          {
            V col[HG];
            col[0] = piv;
#pragma unroll
            for (int u = 1; u < HG; ++u)
              if (g0 + u < NG) col[u] = *reinterpret_cast<const V*>(cb + (g0 + u) * E);
            FL[g1] = col[g1 - g0] * ntL + FL[g1];
            if (sh && g1 < NGS) FS[g1 < NGS ? g1 : 0] = col[g1 - g0] * ntS + FS[g1 < NGS ? g1 : 0];
            if (j + 1 < NP) {  // look-ahead: column j + 1 is complete, post it and ask for its pivot group
              cbn[HALF + lh] = FL[g1][(j + 1) % E];
              if (j + 1 < HALF) cbn[lh] = FS[g1 < NGS ? g1 : 0][(j + 1) % E];
              piv = *reinterpret_cast<const V*>(cbn + g1 * E);
            }
#pragma unroll
            for (int u = 0; u < HG; ++u)
              if (g0 + u < NG && g0 + u != g1) {
                FL[g0 + u] = col[u] * ntL + FL[g0 + u];
                if (sh && g0 + u < NGS) FS[g0 + u < NGS ? g0 + u : 0] = col[u] * ntS + FS[g0 + u < NGS ? g0 + u : 0];
              }
          }
#pragma unroll
          for (int h = g0 + HG; h < NG; h += HG) {
            V col[HG];
#pragma unroll
            for (int u = 0; u < HG; ++u)
              if (h + u < NG) col[u] = *reinterpret_cast<const V*>(cb + (h + u) * E);
#pragma unroll
            for (int u = 0; u < HG; ++u)
              if (h + u < NG) {
                FL[h + u] = col[u] * ntL + FL[h + u];
                if (sh && h + u < NGS) FS[h + u < NGS ? h + u : 0] = col[u] * ntS + FS[h + u < NGS ? h + u : 0];
              }
          }
          rv2 = f2{bj, bj} * f2{-tL, -tS} + rv2;
          mL[e0] = tL;
          mS[e0] = tS;
          if (e0 == E - 1) {
            // group g0 is finished: nobody updates it again.  Its multipliers go to the packed L of the neighbourhood
            // (row r: the groups that hold columns < r; the exchange matrix is dead) and its registers are free.
            // (branch-free: a lane whose row ends left of the group writes to the 16 spare bytes behind the last row)
            T* dump = Lq + KTRI - E;
            *reinterpret_cast<V*>(g0 * E < HALF || g0 * E < HALF + lh ? Lq + troff(HALF + lh) + g0 * E : dump) = mL;
            if (sh) *reinterpret_cast<V*>(g0 * E < lh ? Lq + troff(lh) + g0 * E : dump) = mS;
          }
#if MGP_RHS_FOLD_SCHED
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
      }
#if MGP_RHS_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      MGP_RHS_T(3)
      __syncthreads();
      {
        colbuf[sub * NP + lh] = wS;
        colbuf[sub * NP + HALF + lh] = wL;
        rhsbuf[sub * NP + lh] = uS;
        rhsbuf[sub * NP + HALF + lh] = uL;
      }
      __syncthreads();
      // from here lane i is row i of BOTH neighbourhoods; its responses fly under the back-substitution
      load_y(yA, yrow0);
      load_y(yB, yrow1);
      T wA = colbuf[i], wB = colbuf[NP + i];
      const T uA = rhsbuf[i], uB = rhsbuf[NP + i];
      T* mean = static_cast<T*>(a.mean);
      T* var = static_cast<T*>(a.var);
      const T svA = wave_sum_lane63(uA * wA), svB = wave_sum_lane63(uB * wB);
      const bool badA = !(lane_value(pmin, 0) > T(0)) || !(svA == svA);  // (lane 63's is the one that counts)
      const bool badB = !(lane_value(pmin, 32) > T(0)) || !(svB == svB);
      if (i == NP - 1) {
        var[nbA] = badA ? num<T>::nan() : T(1) - svA;
        if (badA && a.info) atomicAdd(a.info, 1);
        if (have_b) {
          var[nbB] = badB ? num<T>::nan() : T(1) - svB;
          if (badB && a.info) atomicAdd(a.info, 1);
        }
      }
      MGP_RHS_T(4)
      // L^T w = D^-1 u from the last row up, two independent chains interleaved
      constexpr int BB = MGP_RHS_FOLD_BLOCK;
      const T* LA = tile;
      const T* LB = tile + KTRI;
#pragma unroll
      for (int mb = NP - BB; mb >= 0; mb -= BB) {
        if (mb < k) {  // (uniform)
          T la[BB], lb[BB];
#pragma unroll
          for (int e = 0; e < BB; ++e) {  // multipliers of row mb + e at step i (junk for i >= mb + e or i >= k: masked below)
            la[e] = LA[troff(mb + e) + i];
            lb[e] = LB[troff(mb + e) + i];
          }
#pragma unroll
          for (int e = BB - 1; e >= 0; --e) {
            const int m = mb + e;
            if (m >= 1) {
              const T wmA = lane_value(wA, m), wmB = lane_value(wB, m);
              if (i < min(m, k)) {
                wA = fma_t(-la[e], wmA, wA);
                wB = fma_t(-lb[e], wmB, wB);
              }
            }
          }
        }
      }
      MGP_RHS_T(5)
      // all the sums first (independent chains the scheduler can interleave), then one block of stores by lane 63
      T smA[RC], smB[RC];
#pragma unroll
      for (int r = 0; r < RC; ++r) {
        smA[r] = wave_sum_lane63(wA * yA[r]);  // (rows >= k and responses >= R carry zeros)
        smB[r] = wave_sum_lane63(wB * yB[r]);
      }
      if (i == NP - 1) {
        const bool vec = g.resp_vec && (reinterpret_cast<uintptr_t>(mean) % 16 == 0);  // (then R == RC, whole 16-byte groups)
        auto put_row = [&](int64_t nbx, const T (&sm)[RC], bool bad) {
          if (vec) {
#pragma unroll
            for (int r4 = 0; r4 < RC / E; ++r4) {
              V v;
#pragma unroll
              for (int e = 0; e < E; ++e) v[e] = bad ? num<T>::nan() : sm[r4 * E + e];
              *reinterpret_cast<V*>(mean + nbx * R + r4 * E) = v;
            }
          } else {
#pragma unroll
            for (int r = 0; r < RC; ++r)
              if (r < R) mean[nbx * R + r] = bad ? num<T>::nan() : sm[r];
          }
        };
        put_row(nbA, smA, badA);
        if (have_b) put_row(nbB, smB, badB);
      }
      MGP_RHS_T(6)
      continue;
    }

#if MGP_RHS_PRIO
    __builtin_amdgcn_s_setprio(MGP_RHS_PRIO);
#endif
    // ---- elimination with rhs columns ----------------------------------------------------------
    // Per step: every lane posts its column-j entry (one ds_write_b32), all trailing 16-byte groups of
    // the column are requested at once (the reads stay in flight together: serialised read -> wait ->
    // FMA pairs cost an LDS round trip per group), and lane j's 1 + R right-hand sides are broadcast
    // through SGPRs (v_readlane, j is a compile-time lane) instead of a second LDS round trip.
    constexpr int NRE = BACK ? 1 : NRV;  // right-hand-side groups carried through the elimination
    constexpr int LS = NP + 1;           // (BACK) row stride of the multiplier matrix: odd
    T* Lm = tile;                        // (BACK) multipliers l_ij at Lm[j * LS + i]; the exchange matrix is free by now
    V rv[NRE];
#pragma unroll
    for (int r4 = 0; r4 < NRE; ++r4)
#pragma unroll
      for (int e = 0; e < E; ++e) rv[r4][e] = r4 * E + e < (BACK ? 1 : NR) ? rhs[r4 * E + e] : T(0);
    bool bad = false;
    T mypiv = T(1);
    // unused slots (k .. 63) are identity rows: eliminating them changes nothing, so the k test is
    // made once per JB steps (a branch per step costs a register shuffle at every merge point)
    constexpr int JB = 8;
    // fp32: LOOK-AHEAD -- step j first finishes the one register group that holds column j + 1, posts
    // that column and requests the 16 bytes with its pivot, and only then updates the rest of the row:
    // the LDS round trip in front of the next step's reciprocal runs under the packed FMAs of this one.
    // (Requesting the whole next column that early would need a second 64-register copy: spills.)
    V piv = V(0);
    V mreg = V(0);  // (W3) the multipliers of the current 16-byte group of columns
    if constexpr (sizeof(T) == 4) {
      colbuf[i] = A[0][0];
      piv = *reinterpret_cast<const V*>(colbuf);
    }
#pragma unroll
    for (int jb = 0; jb < NP; jb += JB) {
      if (jb < k)
#pragma unroll
      for (int j = jb; j < jb + JB; ++j) {
        const T ajj = A[j / E][j % E];
        V bj[NRE];
#pragma unroll
        for (int r4 = 0; r4 < NRE; ++r4)
#pragma unroll
          for (int e = 0; e < E; ++e) bj[r4][e] = r4 * E + e < (BACK ? 1 : NR) ? lane_value(rv[r4][e], j) : T(0);
        if constexpr (W3) {
          // The column in two chunks of HG groups instead of a whole-column copy (64 registers).  The column buffer is
          // double: column j lives in buffer j & 1, the look-ahead posts column j + 1 into the other one, so the second
          // chunk of column j can still be read after that post.  The multiplier of this step goes to `mreg` and, once
          // its 16-byte group of columns is finished (no later step updates that group), into the row's own dead
          // group: the row ends up holding its row of L.
          T* cb = colbuf + (j & 1) * NP;
          T* cbn = colbuf + ((j + 1) & 1) * NP;
          const T p = piv[j % E];
          bad = bad || !(p > T(0));
          if (i == j) mypiv = p;
          const T t = i > j ? ajj * pivot_rcp(p) : T(0);  // rows <= j are finished: leave them alone
          const V nt = V(-t);
          constexpr int JN = NP - 1, NGR = NP / E, HG = MGP_RHS_W3_HG;
          const int g0 = j / E, g1 = (j < JN ? j + 1 : j) / E;
          {
            V col[HG];
            col[0] = piv;
#pragma unroll
            for (int u = 1; u < HG; ++u)
              if (g0 + u < NGR) col[u] = *reinterpret_cast<const V*>(cb + (g0 + u) * E);
            A[g1] = col[g1 - g0] * nt + A[g1];
            if (j < JN) {
              cbn[i] = A[g1][(j + 1) % E];
              piv = *reinterpret_cast<const V*>(cbn + g1 * E);
            }
#pragma unroll
            for (int u = 0; u < HG; ++u)
              if (g0 + u < NGR && g0 + u != g1) A[g0 + u] = col[u] * nt + A[g0 + u];
          }
#pragma unroll
          for (int h = g0 + HG; h < NGR; h += HG) {
#if MGP_RHS_W3_SCHED & 1
            __builtin_amdgcn_sched_barrier(0);
#endif
            V col[HG];
#pragma unroll
            for (int u = 0; u < HG; ++u)
              if (h + u < NGR) col[u] = *reinterpret_cast<const V*>(cb + (h + u) * E);
#pragma unroll
            for (int u = 0; u < HG; ++u)
              if (h + u < NGR) A[h + u] = col[u] * nt + A[h + u];
          }
          rv[0] = bj[0] * nt + rv[0];
          mreg[j % E] = t;
          if (j % E == E - 1) A[g0] = mreg;  // group g0 is finished: nobody updates it again
#if MGP_RHS_W3_SCHED & 2
          __builtin_amdgcn_sched_barrier(0);
#endif
        } else if constexpr (sizeof(T) == 4) {
          V col[NP / E];
          col[j / E] = piv;
#pragma unroll
          for (int c4 = j / E + 1; c4 < NP / E; ++c4) col[c4] = *reinterpret_cast<const V*>(colbuf + c4 * E);
          const T p = piv[j % E];
          bad = bad || !(p > T(0));
          if (i == j) mypiv = p;
          const T t = i > j ? ajj * pivot_rcp(p) : T(0);  // rows <= j are finished: leave them alone
          const V nt = V(-t);
          if constexpr (BACK && !W3) Lm[j * LS + i] = t;
          constexpr int JN = NP - 1;
          const int g1 = (j < JN ? j + 1 : j) / E;  // group of the next column (compile-time after unrolling)
          A[g1] = col[g1] * nt + A[g1];
          if (j < JN) {
            colbuf[i] = A[g1][(j + 1) % E];
            piv = *reinterpret_cast<const V*>(colbuf + g1 * E);
          }
#pragma unroll
          for (int c4 = j / E; c4 < NP / E; ++c4)
            if (c4 != g1) A[c4] = col[c4] * nt + A[c4];
#pragma unroll
          for (int r4 = 0; r4 < NRE; ++r4) rv[r4] = bj[r4] * nt + rv[r4];
        } else {
          // fp64: a full copy of the column would not fit beside the 128-register row; streamed
          colbuf[i] = ajj;
          const V cp = *reinterpret_cast<const V*>(colbuf + (j / E) * E);
          const T p = cp[j % E];
          bad = bad || !(p > T(0));
          if (i == j) mypiv = p;
          const T t = i > j ? ajj * pivot_rcp(p) : T(0);
          const V nt = V(-t);
          if constexpr (BACK) Lm[j * LS + i] = t;
          A[j / E] = cp * nt + A[j / E];
#pragma unroll
          for (int c4 = j / E + 1; c4 < NP / E; ++c4) {
            const V cv = *reinterpret_cast<const V*>(colbuf + c4 * E);
            A[c4] = cv * nt + A[c4];
          }
#pragma unroll
          for (int r4 = 0; r4 < NRE; ++r4) rv[r4] = bj[r4] * nt + rv[r4];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < (BACK ? 1 : NR); ++r) rhs[r] = rv[r / E][r % E];
    MGP_RHS_T(3)

#if MGP_RHS_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    // ---- outputs: cross-lane sums over the k rows ---------------------------------------------
    const T inv_d = i < k ? pivot_rcp(mypiv) : T(0);
    const T u = rhs[0];
    T* mean = static_cast<T*>(a.mean);
    T* var = static_cast<T*>(a.var);
    T* yk = static_cast<T*>(a.ykinvy);
    // 1 + 2 R cross-lane sums, each landing in lane 63
    const T sv = wave_sum_lane63(u * u * inv_d);
    if (i == NP - 1) {
      var[nb] = bad ? num<T>::nan() : T(1) - sv;
      if (bad && a.info) atomicAdd(a.info, 1);
    }
    if constexpr (BACK) {
      // w = K^-1 c: L^T w = D^-1 u, from the last row up; lane j takes l_mj w_m off for every m > j
      // MGP_RHS_BACK_BLOCK (32; measured 4 / 8 / 16 / 32 / 64: 93.0 / 93.2 / 94.2 / 94.9 / 93.4 M/s on config 5) steps per
      // block: their multipliers are requested together (they do not depend on w), so the chain of
      // dependent steps waits for LDS once per block instead of once per step (a test and an LDS round trip per step
      // were 63 exposed latencies per neighbourhood).  The k test is per block, as in the elimination: the unused
      // slots behind k are identity rows (zero multipliers).  The elimination writes multiplier rows in blocks
      // of JB = 8 steps and skips whole blocks behind k, so ROW i of Lm is written (by all 64 lanes) only for
      // i < roundup8(k): a lane i >= k must not touch its row -- it would read never-written LDS (stale bits
      // of an earlier launch; NaN x 0 = NaN), and its w is 0 anyway.  Hence the `i < min(m, k)` guard below.
      T w = u * inv_d;
      constexpr int BB = MGP_RHS_BACK_BLOCK;
      if constexpr (W3) {
        // the rows of L, kept in place during the elimination, go to LDS once, in the packed layout of the exchange
        // matrix (which is dead by now): row i, the groups that hold columns < i.  Every row is written -- also the
        // identity rows behind k (zero multipliers) -- so nothing stale is ever read; what lies beyond a row's
        // diagonal inside its last group is junk that the `i < min(m, k)` mask below keeps out.
        __syncthreads();
#pragma unroll
        for (int c4 = 0; c4 < NP / E; ++c4)
          if (c4 * E < i) *reinterpret_cast<V*>(tile + rowoff(i) + c4 * E) = A[c4];
        __syncthreads();
      }
#pragma unroll
      for (int mb = NP - BB; mb >= 0; mb -= BB) {
        if (mb < k) {  // (uniform)
          T lm[BB];
#pragma unroll
          for (int e = 0; e < BB; ++e)  // multiplier of row mb + e at step i (junk for i >= mb + e or i >= k: masked below)
            lm[e] = W3 ? tile[rowoff(mb + e) + i] : Lm[i * LS + mb + e];  // (W3: lanes read L column-wise, consecutive addresses)
#pragma unroll
          for (int e = BB - 1; e >= 0; --e) {
            const int m = mb + e;
            if (m >= 1) {
              const T wm = lane_value(w, m);
              if (i < min(m, k)) w = fma_t(-lm[e], wm, w);
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < RC; ++r) {
        if (r < R) {
          const T sm = wave_sum_lane63(w * rhs[1 + r]);  // (rows >= k carry zero responses and w = 0)
          if (i == NP - 1) mean[nb * R + r] = bad ? num<T>::nan() : sm;
        }
      }
    } else {
#pragma unroll
    for (int r = 0; r < RC; ++r) {
      if (r < R) {
        const T sm = wave_sum_lane63(u * rhs[1 + r] * inv_d);
        if (i == NP - 1) mean[nb * R + r] = bad ? num<T>::nan() : sm;
        if (yk) {
          const T sy = wave_sum_lane63(rhs[1 + r] * rhs[1 + r] * inv_d);
          if (i == NP - 1) yk[nb * R + r] = bad ? num<T>::nan() : sy;
        }
      }
    }
    }
    MGP_RHS_T(6)
  }
#if MGP_RHS_TIMING
  if (threadIdx.x == 0)
    for (int t = 0; t < 8; ++t) atomicAdd(&g_rhs_timing[t], tacc_[t]);
#endif
}

template <typename T, int RC, bool BACK = false, bool GRAM = false, bool W3 = false, bool FOLD = false>
static int launch_rhs_impl(const FusedArgs& a, hipStream_t stream) {
  constexpr int NP = 64;
  constexpr int E = v16<T>::N;
  constexpr int CH = 2 * E;
  constexpr int KS = NP + E;
  constexpr int KTRI = E * ((NP - 1) / E + 1) * (E * ((NP - 1) / E) / 2 + (NP - 1) % E) + NP + E;
  constexpr int KMAT = W3 ? KTRI : (FOLD && 2 * KTRI > NP * KS ? 2 * KTRI : NP * KS);
  constexpr int NRV = (1 + RC + E - 1) / E;
  RhsGeom g;
  const int dpad = (a.d + CH - 1) / CH * CH;
  g.dst = dpad < 64 ? dpad : 64;
  g.xs = g.dst + E;
  const uintptr_t align = (uintptr_t)a.feat_q | (uintptr_t)a.feat_nn;
  g.vec_ok = (a.d % E == 0) && (align % 16 == 0);
  g.resp_vec = a.R == RC && RC % E == 0 && (uintptr_t)a.targets % 16 == 0;
  g.ntasks = a.b;
  const size_t tile_need = (size_t)(NP + 1) * g.xs + (FOLD ? (NP / 2) * (NP / 2) : 0);
  const size_t tile_elems = tile_need > (size_t)KMAT ? tile_need : (size_t)KMAT;
  size_t lds = (tile_elems + (FOLD ? 4 * NP : W3 ? 2 * NP : NP) + (FOLD ? 2 * NP : NRV * E) + g.dst + (g.dst & 1)) * sizeof(T) + 66 * sizeof(int64_t);
  lds = (lds + 15) & ~(size_t)15;
  static Residency res;
  int per_cu = 0, cus = 0;
  const int rc = res.lookup(reinterpret_cast<const void*>(&fused_rhs_kernel<T, RC, BACK, GRAM, W3, FOLD>), 64, lds, &per_cu, &cus);
  if (rc != MGP_OK) return rc;
  static const int env_per_cu = getenv("MGP_RHS_PER_CU") ? atoi(getenv("MGP_RHS_PER_CU")) : 0;  // occupancy experiments
  if (env_per_cu > 0 && env_per_cu < per_cu) per_cu = env_per_cu;
  static const bool trace = getenv("MGP_TRACE") != nullptr;
  if (trace)
    fprintf(stderr, "[mgp] fused_rhs_kernel<%d,%d,%d,%d,%d>: lds %zu B, %d workgroups per CU\n", RC, (int)BACK, (int)GRAM, (int)W3, (int)FOLD, lds, per_cu);
  int64_t grid = (int64_t)cus * per_cu;
  const int64_t nunits = FOLD ? (g.ntasks + 1) / 2 : g.ntasks;  // (FOLD: a pass of the persistent loop takes two tasks)
  if (grid > nunits) grid = nunits;
  hipLaunchKernelGGL((fused_rhs_kernel<T, RC, BACK, GRAM, W3, FOLD>), dim3((unsigned)grid), dim3(64), lds, stream, a, g);
  MGP_HIP_CHECK_LAUNCH();
  note_launch("mgp::fused_rhs_kernel<%s,%d,%s,%s%s>", sizeof(T) == 4 ? "float" : "double", RC, BACK ? "true" : "false",
              GRAM ? "true" : "false", W3 ? ",w3" : (FOLD ? ",fold" : ""));
  return MGP_OK;
}

#ifndef MGP_RHS_GRAM
#define MGP_RHS_GRAM 1
#endif
template <typename T, int RC, bool BACK = false>
static int launch_rhs(const FusedArgs& a, hipStream_t stream) {
  if constexpr (sizeof(T) == 4 && MGP_RHS_GRAM) {
    constexpr int CH = 2 * v16<T>::N;
    // one feature stage (the norms live behind the staged row), and not the Matern-1/2 kernel: its slope at
    // zero distance turns the cancellation error of the Gram form into covariance error
    // (and rows of whole 16-byte groups, at least two: the tiny-d fixtures gain nothing and the 1-d one is
    // ill-conditioned enough for the cancellation error to show)
    if ((a.d + CH - 1) / CH * CH <= 64 && a.d % v16<T>::N == 0 && a.d >= CH && a.kernel_id != MGP_KERNEL_MATERN_05) {
#ifndef MGP_RHS_W3
#define MGP_RHS_W3 1
#endif
#ifndef MGP_RHS_FOLD
#define MGP_RHS_FOLD 1
#endif
      // (FOLD: aligned rows -- the vector gather -- at least one pair, and rows short enough for eight workgroups per CU
      // with the parked half rows behind them: d <= 48; longer rows take the three-wave variant)
      if constexpr (BACK && MGP_RHS_FOLD) {
        const uintptr_t align = (uintptr_t)a.feat_q | (uintptr_t)a.feat_nn;
        if (align % 16 == 0 && a.b >= 2 && (a.d + CH - 1) / CH * CH <= 48) return launch_rhs_impl<T, RC, BACK, true, false, true>(a, stream);
      }
      if constexpr (BACK && MGP_RHS_W3) return launch_rhs_impl<T, RC, BACK, true, true>(a, stream);
      return launch_rhs_impl<T, RC, BACK, true>(a, stream);
    }
  }
  return launch_rhs_impl<T, RC, BACK, false>(a, stream);
}

template <typename T>
int launch_fused_rhs(const FusedArgs& a, hipStream_t stream) {
  if (a.k > 64 || a.packed_nn != nullptr) return MGP_EUNSUPPORTED;
  if (a.R <= 4) return launch_rhs<T, 4>(a, stream);
  // many responses without y^T K^-1 y (prediction): one right-hand side + back-substitution
#ifndef MGP_RHS_MF
#define MGP_RHS_MF 1
#endif
  if constexpr (sizeof(T) == 4 && MGP_RHS_MF) {
    // fp32: the kernel built on the matrix cores' output layout first (mgp_fused_rhs_mf.hip); shapes it declines go on
    if (a.R <= 16 && a.ykinvy == nullptr && MGP_RHS_BACK) {
      const int rc = launch_fused_rhs_mf(a, stream);
      if (rc != MGP_EUNSUPPORTED) return rc;
    }
  }
  if (a.R <= 16 && a.ykinvy == nullptr && MGP_RHS_BACK) return launch_rhs<T, 16, true>(a, stream);
  if (a.R <= 16) return launch_rhs<T, 16>(a, stream);
  return MGP_EUNSUPPORTED;
}

template int launch_fused_rhs<float>(const FusedArgs&, hipStream_t);
template int launch_fused_rhs<double>(const FusedArgs&, hipStream_t);

}  // namespace mgp

// Register-resident fused kernel for wide neighbourhoods: 64 < k + 1 + R <= 128 (fp32).
//
// Same algebra and slot layout as mgp_fused_wave.hip -- slots 0..k-1 neighbour rows, padding,
// slot q = 127 - R the query, slots q+1..127 the R response rows; k steps of a row-per-lane
// right-looking Cholesky on the lower triangle of the augmented system leave
//     var = S[q][q],  mean_r = -S[q+1+r][q],  y_r^T K^-1 y_r = -S[q+1+r][q+1+r]
// -- but one neighbourhood is 128 slots, i.e. TWO wavefronts (one workgroup of 128 threads), and a
// lane's row is 128 registers.  What that changes:
//
//   * distances: the cyclic pair scheme with NP = 128: 64 pairs per lane, register blocked
//     4 own rows x 16 partner rows (pair {i+p, i+o_j}, o = 0, 33, 49, 65: cyclic distances
//     1..64, each once, 64 twice);
//   * exchange into row-per-lane registers through a PACKED lower-triangular matrix (34 KB; the
//     square would be 68 KB per workgroup), one pass;
//   * Cholesky: blocked by four columns -- two workgroup barriers per block instead of one per
//     column (the barrier + LDS round trip per step is what two waves per SIMD cannot hide); the row
//     is updated by streaming the eliminated block columns (load 16 B, two packed FMAs), so no copy
//     of a column is held next to the 128 row registers.
//
// 2 waves per SIMD (<= 256 VGPRs), 4 workgroups per CU (38 KB of LDS each).  The LDS workgroup kernel
// (mgp_generic.hip) remains the path for fp64 and for more than 128 slots.
#include "mgp_wave_common.h"

#include <utility>

#ifndef MGP_WIDE_MIN_ROWS
#define MGP_WIDE_MIN_ROWS 65
#endif
#ifndef MGP_WIDE_PRIO
#define MGP_WIDE_PRIO 2
#endif
#ifndef MGP_WIDE_DIST_PRIO
#define MGP_WIDE_DIST_PRIO 1
#endif
#ifndef MGP_WIDE_XCHG_PRIO
#define MGP_WIDE_XCHG_PRIO 0
#endif
#ifndef MGP_WIDE_GC
#define MGP_WIDE_GC 2
#endif
#ifndef MGP_WIDE_SPLIT
#define MGP_WIDE_SPLIT 1  // the first wave stops at its last lower-triangle group (0: both waves run every group)
#endif

namespace mgp {

struct WideGeom {
  int q, dst, xs, vec_ok;
};

// NG (round 5): the 16-byte column groups of the system, 4 NG >= k + 1 + R slots -- query and responses sit in the last
// 1 + R of THEM (q = 4 NG - 1 - R, the launcher's choice), the row registers and every elimination step's trailing
// update end there (k = 100: 26 groups instead of 32; the lanes past 4 NG carry rows nobody reads).  And the first
// wave -- rows 0 .. 63 -- stops at group 15 and at block 15: its rows have no lower-triangle entry beyond, it only
// keeps the workgroup's barriers from there on (two copies of the elimination, one per wave, compile-time limits:
// a uniform run-time test inside the unrolled trailing update costs more than it saves, see below).
template <int NG>
__global__ __launch_bounds__(128, 2) void fused_wide_kernel(FusedArgs a, WideGeom g) {
  using T = float;
  constexpr int NP = 128;
  // Pair scheme on a ring of M = k + 1 rows (the neighbours and the query; round 5 -- the ring used to be all 128
  // slots): lane i < M takes the pairs {i + o_j, i + s} mod M, o = 0, 2 BP + 1, 3 BP + 1, 4 BP + 1, s = 1 .. BP --
  // cyclic distances 1 .. 4 BP, each pair of rows at least once as 4 BP >= (M - 1) / 2 (a distance past that meets a
  // pair from its other end: the same value to the same place).  4 BP = 2 NG + (0 .. 2) covers every M that NG serves;
  // k = 68: 36 pairs per lane instead of 64.  Lanes past M repeat lane i - M.
  constexpr int BA = 4, BP = (NG + 1) / 2;
  constexpr int NS = BA * BP;  // pairs per lane
  auto own_offset = [](int j) { return j == 0 ? 0 : (j + 1) * BP + 1; };
  constexpr int E = 4, CH = 8;
  constexpr int TRI = 8 * 32 * 33 + 2 * NP + 2 * E;  // packed lower-triangular exchange matrix (+ over-read pad)
  using V = v16<T>::type;
  using ACC = v16<T>::acc;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int k = a.k, R = a.R, d = a.d, q = g.q, dst = g.dst, xs = g.xs;
  const int tile_elems = NP * xs > TRI ? NP * xs : TRI;
  T* tile = reinterpret_cast<T*>(smem);   // feature tile, later the exchange matrix
  T* colbuf = tile + tile_elems;          // 2 x 512: raw / eliminated block entries of a Cholesky block
  T* ilbuf = colbuf + 2 * NP * E;         // dst inverse length scales (Anisotropy)
  T* outbuf = ilbuf + dst;                // (R + 1) x 2: Schur-block entries on their way out
  int64_t* idxbuf = reinterpret_cast<int64_t*>(outbuf + 2 * 17 + ((dst + 2 * 17) & 1));  // 128 row offsets

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

  for (int64_t nb = blockIdx.x; nb < a.b; nb += gridDim.x) {
    int i = threadIdx.x;
    asm volatile("" : "+v"(i));  // keep per-lane addresses of the unrolled phases out of LICM
    // ---- phase 0: indices, nugget ------------------------------------------------------------
    int64_t myidx = 0;
    T myeps = T(0);
    if (i < k) {
      myidx = a.nn_idx[nb * k + i];
      if (a.noise_mode == MGP_NOISE_SCALAR) myeps = (T)a.noise_scalar;
      else if (a.noise_mode == MGP_NOISE_TABLE) myeps = noise_dev[myidx];
      else myeps = noise_dev[nb * k + i];
    } else if (i == q) {
      myidx = a.batch_idx ? a.batch_idx[nb] : nb;
    }
    const int64_t mytg = a.targets_batch ? nb * k + (i < k ? i : 0) : myidx;
    __syncthreads();  // the previous neighbourhood's LDS reads are complete
    idxbuf[i] = myidx * (int64_t)d;

    // ---- phase 1: stage the features (one stage: d <= 64) -----------------------------------------
    const int w = d, wp = (d + CH - 1) / CH * CH;
    __syncthreads();
    if (g.vec_ok) {
      // consecutive lanes walk a row in 16-byte pieces; slots without features are zero rows
      const int c16 = w / E, c16p = wp / E;
      // (tile row r = ring position r: the neighbours, then the query at row k)
      for (int t = i; t < (k + 1) * c16p; t += NP) {
        const int row = t / c16p, c = t - row * c16p;
        V v = V(0);
        if (c < c16) v = *reinterpret_cast<const V*>((row < k ? feat_nn + idxbuf[row] : feat_q + idxbuf[q]) + c * E);
        *reinterpret_cast<V*>(tile + row * xs + c * E) = v;
      }
    } else {
      for (int t = i; t < (k + 1) * wp; t += NP) {
        const int row = t / wp, c = t - row * wp;
        T v = T(0);
        if (c < w) v = (row < k ? feat_nn + idxbuf[row] : feat_q + idxbuf[q])[c];
        tile[row * xs + c] = v;
      }
    }
    if (aniso)
      for (int c = i; c < wp; c += NP) ilbuf[c] = c < w ? T(1) / ls[c] : T(0);
    __syncthreads();

#if MGP_WIDE_PRIO
    __builtin_amdgcn_s_setprio(MGP_WIDE_DIST_PRIO);
#endif
    // ---- phase 2: squared distances, then covariances; two halves of the own rows so that the
    //      64 packed accumulators of a lane never coexist (register budget: 256 with the 128-entry row)
    T kv[NS];
    const int M = k + 1;
    const int ir = i < M ? i : i - M;  // ring position of the lane (M >= 65: one wrap)
    auto wrap = [&](int x) { return x >= M ? x - M : x; };
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      constexpr int HB = BA / 2;
      ACC acc[HB * BP];
#pragma unroll
      for (int s = 0; s < HB * BP; ++s) acc[s] = ACC(0);
      // one instance per deformation (the Anisotropy test inside the pair loop is a branch per
      // accumulate); partner rows are requested PB at a time, then consumed
      auto chunks = [&](auto anis) {
        constexpr bool ANISO = decltype(anis)::value != 0;
        constexpr int PB = 4;
        for (int c0 = 0; c0 < wp; c0 += CH) {
          V own0[HB], own1[HB];
#pragma unroll
          for (int j = 0; j < HB; ++j) {
            const T* xj = tile + wrap(ir + own_offset(half * HB + j)) * xs + c0;
            own0[j] = *reinterpret_cast<const V*>(xj);
            own1[j] = *reinterpret_cast<const V*>(xj + E);
          }
          V il0 = V(1), il1 = V(1);
          if constexpr (ANISO) {
            il0 = *reinterpret_cast<const V*>(ilbuf + c0);
            il1 = *reinterpret_cast<const V*>(ilbuf + c0 + E);
          }
#pragma unroll
          for (int s0 = 1; s0 <= BP; s0 += PB) {
            V o0[PB], o1[PB];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
              if (s0 + u > BP) continue;  // (compile-time: BP need not be a multiple of PB)
              const T* xo = tile + wrap(ir + s0 + u) * xs + c0;
              o0[u] = *reinterpret_cast<const V*>(xo);
              o1[u] = *reinterpret_cast<const V*>(xo + E);
            }
#pragma unroll
            for (int u = 0; u < PB; ++u)
#pragma unroll
              for (int j = 0; j < HB; ++j) {
                if (s0 + u > BP) continue;
                if constexpr (ANISO) {
                  accum(acc[j * BP + s0 + u - 1], vsub(own0[j], o0[u]) * il0);
                  accum(acc[j * BP + s0 + u - 1], vsub(own1[j], o1[u]) * il1);
                } else {
                  accum(acc[j * BP + s0 + u - 1], vsub(own0[j], o0[u]));
                  accum(acc[j * BP + s0 + u - 1], vsub(own1[j], o1[u]));
                }
              }
          }
        }
      };
      if (aniso) chunks(ic<1>{});
      else chunks(ic<0>{});
      kernel_dispatch(a.kernel_id, a.metric_id, [&](auto kid, auto mid) {
        constexpr int KID = decltype(kid)::value, MID = decltype(mid)::value;
#pragma unroll
        for (int s = 0; s < HB * BP; s += 2) {
          const f2 kk = cov_from_sqdist2(f2{acc_total(acc[s]), acc_total(acc[s + 1])}, KID, MID, post_scale);
          kv[half * HB * BP + s] = kk.x;
          kv[half * HB * BP + s + 1] = kk.y;
        }
      });
    }

    // ---- phase 3: covariances -> packed lower-triangular exchange matrix -> row per lane ------------
    // Row r of the exchange matrix holds its r + 1 lower-triangle entries and starts at
    // tri(r) = 8 a (a + 1) + 4 (r & 3) (a + 1), a = r >> 2 (every row padded to whole 16-byte groups):
    // 34 KB for 128 rows instead of 68 KB for the square, written in ONE pass.  A lane then reads 32
    // groups from the start of its row: what lies beyond column i belongs to later rows (upper-triangle
    // garbage the elimination never uses).
#if MGP_WIDE_PRIO
    __builtin_amdgcn_s_setprio(MGP_WIDE_XCHG_PRIO);
#endif
    const T mydiag = i < k ? T(1) + myeps : (i <= q ? T(1) : T(0));
    auto tri = [](int r) { const int a = r >> 2; return (a + 1) * (8 * a + 4 * (r & 3)); };
    V A[NG];
    __syncthreads();  // every lane is done reading the feature tile (the exchange matrix aliases it)
    {
      int i3 = i;
      asm volatile("" : "+v"(i3));
      const int dump = tri(NP - 1) + NP + E;  // behind the last row
      const int ir3 = i3 < M ? i3 : i3 - M;
#pragma unroll
      for (int s = 1; s <= NS; ++s) {
        // ring positions -> slots: the query (position k) is slot q; every pair of ring rows is an entry of the system
        const int r1 = wrap(ir3 + own_offset((s - 1) / BP)), c = wrap(ir3 + (s - 1) % BP + 1);
        const int s1 = r1 < k ? r1 : q, sc = c < k ? c : q;
        const int hi = max(s1, sc), lo = min(s1, sc);
        tile[hi != lo ? tri(hi) + lo : dump] = kv[s - 1];
      }
      // what no pair writes: the padding slots k .. q - 1 (rows of zeros) and their columns in the query's row
      if (i3 >= k && i3 <= q) {
        const int rowz = tri(i3);
        for (int cz = i3 == q ? k : 0; cz < i3; ++cz) tile[rowz + cz] = T(0);
      }
      const int myrow = tri(i3);
      tile[myrow + i3] = mydiag;
      for (int r = 0; r < R; ++r)
        if (i3 <= q + 1 + r) tile[tri(q + 1 + r) + i3] = i3 < k ? targets[mytg * (int64_t)R + r] : T(0);
      __syncthreads();
#pragma unroll
      for (int c4 = 0; c4 < NG; ++c4) A[c4] = *reinterpret_cast<const V*>(tile + myrow + c4 * E);
    }

#if MGP_WIDE_PRIO
    __builtin_amdgcn_s_setprio(MGP_WIDE_PRIO);  // elimination > distances > covariances / exchange
#endif
    // ---- phase 4: blocked Cholesky, row per lane, FOUR columns per exchange ------------------------
    // A step-by-step elimination costs a workgroup barrier and an LDS round trip per column, and with
    // two waves per SIMD that latency is what the kernel waits on.  A block of four columns J0..J0+3
    // (one 16-byte register group) takes two barriers instead:
    //   1. every lane posts its raw block entries a_i[0..3]; all read the four rows of the diagonal
    //      block and factor it redundantly (pivots p_m, and u_m'(J0+m) for m' < m), then eliminate
    //      the block inside their own row:  u_m(i) = a_i[m] - sum_{m'<m} u_m'(i) u_m'(J0+m) / p_m';
    //   2. every lane posts u_0..3(i) (transposed: ubuf[m][i]); the trailing groups take
    //      A[i][c] -= sum_m (u_m(i) / p_m) u_m(c), four independent packed-FMA streams per group.
    // Same arithmetic as the column-at-a-time form (the sum over m is applied in the same order).
    // Fully unrolled (compile-time block index: no register-group switches, the loads of a block issue
    // together).
    bool bad = false;
    T* rawbuf = colbuf;            // 128 x 4 raw block entries (row-major: lane i at rawbuf + 4 i)
    T* ubuf = colbuf + NP * E;     // 4 x 128 eliminated entries, transposed: ubuf[m * 128 + i]
    auto eliminate = [&](auto glc) {
    constexpr int GL = decltype(glc)::value;  // this wave's rows have lower-triangle entries in groups 0 .. GL - 1
    static_for<NG>([&](auto jbc) {
      constexpr int jb = decltype(jbc)::value;
      constexpr int J0 = jb * E;
      if constexpr (jb >= GL) {
        if (J0 < k) {  // (uniform) the other wave's block: keep its two barriers
          __syncthreads();
          __syncthreads();
        }
      } else if (J0 < k) {  // uniform
        V pg = A[jb];
        const int mlim = min(E, k - J0);  // columns of this block that are eliminated (the last block may be short)
        *reinterpret_cast<V*>(rawbuf + i * E) = pg;
        __syncthreads();
        // diagonal block rows J0 .. J0+3 (raw), factored redundantly by every lane
        V dr[E];
#pragma unroll
        for (int r = 0; r < E; ++r) dr[r] = *reinterpret_cast<const V*>(rawbuf + (J0 + r) * E);
        T rp[E];
#pragma unroll
        for (int m = 0; m < E; ++m) {
          // dr[r][m] for r >= m now holds u_m(J0 + r)
          const T pm = dr[m][m];
          const bool on = m < mlim;
          bad = bad || (on && !(pm > T(0)));
          rp[m] = on ? pivot_rcp(pm) : T(0);
#pragma unroll
          for (int r = m + 1; r < E; ++r) {
            const T t = dr[r][m] * rp[m];
#pragma unroll
            for (int c = m + 1; c <= r; ++c) dr[r][c] = fma_t(-t, dr[c][m], dr[r][c]);
          }
        }
        // own row: u_m(i) and the multipliers nt_m(i) = -u_m(i) / p_m
        T nt[E];
#pragma unroll
        for (int m = 0; m < E; ++m) {
          // a select, not a product with rp = 0: columns behind the last eliminated one hold, in the rows
          // above them, whatever the packed matrix had beyond the row's diagonal (possibly NaN)
          nt[m] = m < mlim ? -pg[m] * rp[m] : T(0);
#pragma unroll
          for (int c = m + 1; c < E; ++c) pg[c] = fma_t(nt[m], dr[c][m], pg[c]);
        }
#pragma unroll
        for (int m = 0; m < E; ++m) ubuf[m * NP + i] = m < mlim ? pg[m] : T(0);  // 0 x 0, never 0 x garbage
        __syncthreads();
        // trailing groups, the block's own included (it stays current: the outputs read the last
        // groups).  Groups that hold padding columns only are updated too: a uniform test around them
        // splits the block's loads into separately scheduled pieces and costs more than it saves
        // (measured: k = 100 20.0 -> 22.1 ms per 200 k neighbourhoods).
        // GC groups at a time: their 4 GC loads are issued together, then the 8 GC packed FMAs (a load
        // followed at once by its two FMAs pays a full LDS round trip per 16 bytes)
        constexpr int GC = MGP_WIDE_GC;
#pragma unroll
        for (int c0 = jb; c0 < GL; c0 += GC) {
          V cv[GC][E];
#pragma unroll
          for (int g2 = 0; g2 < GC; ++g2)
#pragma unroll
            for (int m = 0; m < E; ++m)
              if (c0 + g2 < GL) cv[g2][m] = *reinterpret_cast<const V*>(ubuf + m * NP + (c0 + g2) * E);
#pragma unroll
          for (int g2 = 0; g2 < GC; ++g2)
#pragma unroll
            for (int m = 0; m < E; ++m)
              if (c0 + g2 < GL) A[c0 + g2] = cv[g2][m] * V(nt[m]) + A[c0 + g2];
        }
      }
    });
    };
    if (MGP_WIDE_SPLIT && i < 64) eliminate(ic<(NG < 16 ? NG : 16)>{});  // (wave-uniform: the workgroup's first wave)
    else eliminate(ic<NG>{});

#if MGP_WIDE_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    // ---- phase 5: Schur block -> outputs --------------------------------------------------------
    // lane q holds S[q][q]; lane q+1+r holds S[q+1+r][q] and S[q+1+r][q+1+r]: picked out of the
    // registers by a compare-select sweep over the last 32 columns (q >= 4 NG - 17 as R <= 16)
    T aq = T(0), aii = T(0);
#pragma unroll
    for (int c = NG * E - 32; c < NG * E; ++c) {
      const T v = A[c / E][c % E];
      aq = c == q ? v : aq;
      aii = c == i ? v : aii;
    }
    T* mean = static_cast<T*>(a.mean);
    T* var = static_cast<T*>(a.var);
    T* yk = static_cast<T*>(a.ykinvy);
    if (i == q) {
      var[nb] = bad ? num<T>::nan() : aq;
      if (bad && a.info) atomicAdd(a.info, 1);
    } else if (i > q && i <= q + R) {  // (the lanes past q + R hold no row of the system when 4 NG < 128)
      const int r = i - q - 1;
      mean[nb * R + r] = bad ? num<T>::nan() : -aq;
      if (yk) yk[nb * R + r] = bad ? num<T>::nan() : -aii;
    }
  }
}

template <typename T>
int launch_fused_wide(const FusedArgs& a, hipStream_t stream) {
  if constexpr (sizeof(T) != 4) {
    return MGP_EUNSUPPORTED;  // a row of 128 doubles does not fit the register file
  } else {
    constexpr int NP = 128, E = 4, CH = 8, TRI = 8 * 32 * 33 + 2 * NP + 2 * E;
    const int rows = a.k + 1 + a.R;
    // every shape past the 64 slots of a wave that the rhs-column kernel (k <= 64) did not take: faster
    // than the LDS workgroup kernel from the first row on (k = 68: 19.8 vs 15.1, k = 75: 18.8 vs 13.0,
    // k = 100: 17.6 vs 4.7, k = 126: 15.6 vs 1.7 M neighbourhoods/s) although it pays for all 128 slots
    // (a.k < 64: the pair ring of k + 1 rows wraps ONCE -- `ir`, wrap() -- which needs 2 (k + 1) >= 128 slots; the
    // rhs-column kernel takes every k <= 64 before this launcher is asked, and must: k = 48 .. 55 with 16 responses
    // would read tile rows the gather no longer loads)
    if (rows < MGP_WIDE_MIN_ROWS || a.k < 64 || rows > NP || a.R > 16 || a.d > 64 || a.packed_nn != nullptr || a.coeffs != nullptr)
      return MGP_EUNSUPPORTED;  // more slots / responses / feature stages: the LDS workgroup kernel
    // column groups: the smallest instantiation that holds the rows (17 .. 32 in steps of 2 or 3)
#ifdef MGP_WIDE_FORCE_NG32  // (A/B builds: the 128-slot form of rounds 2-4)
    const int ng = 32;
#else
    const int ng = rows <= 68 ? 17 : rows <= 80 ? 20 : rows <= 88 ? 22 : rows <= 96 ? 24 : rows <= 104 ? 26
                   : rows <= 112 ? 28 : rows <= 120 ? 30 : 32;
#endif
    WideGeom g;
    g.q = ng * E - 1 - a.R;
    const int dpad = (a.d + CH - 1) / CH * CH;
    g.dst = dpad < 64 ? dpad : 64;
    g.xs = g.dst + E;
    const uintptr_t align = (uintptr_t)a.feat_q | (uintptr_t)a.feat_nn;
    g.vec_ok = (a.d % E == 0) && (align % 16 == 0);
    const size_t tile_elems = (size_t)NP * g.xs > TRI ? (size_t)NP * g.xs : TRI;
    size_t lds = (tile_elems + 2 * NP * E + g.dst + 2 * 17 + ((g.dst + 2 * 17) & 1)) * sizeof(float) + NP * sizeof(int64_t);
    lds = (lds + 15) & ~(size_t)15;
    const void* fn = nullptr;
    static Residency res[8];
    int ri = 0;
    switch (ng) {
#define MGP_WIDE_CASE(N, I) case N: fn = reinterpret_cast<const void*>(&fused_wide_kernel<N>); ri = I; break;
      MGP_WIDE_CASE(17, 0) MGP_WIDE_CASE(20, 1) MGP_WIDE_CASE(22, 2) MGP_WIDE_CASE(24, 3)
      MGP_WIDE_CASE(26, 4) MGP_WIDE_CASE(28, 5) MGP_WIDE_CASE(30, 6)
      default: fn = reinterpret_cast<const void*>(&fused_wide_kernel<32>); ri = 7; break;
#undef MGP_WIDE_CASE
    }
    int per_cu = 0, cus = 0;
    const int rc = res[ri].lookup(fn, NP, lds, &per_cu, &cus);
    if (rc != MGP_OK) return rc;
    int64_t grid = (int64_t)cus * per_cu;
    if (grid > a.b) grid = a.b;
    static const bool trace = getenv("MGP_TRACE") != nullptr;
    if (trace)
      fprintf(stderr, "mgp: fused_wide_kernel b=%lld k=%d d=%d R=%d grid=%lld lds=%zu per_cu=%d\n", (long long)a.b, a.k, a.d,
              a.R, (long long)grid, lds, per_cu);
    void* kargs[] = {const_cast<FusedArgs*>(&a), &g};
    {
      const hipError_t le = hipLaunchKernel(fn, dim3((unsigned)grid), dim3(NP), kargs, lds, stream);
      if (le != hipSuccess) return -(1000 + (int)le);
    }
    MGP_HIP_CHECK_LAUNCH();
    note_launch("mgp::fused_wide_kernel<%d>", ng);
    note_launch_geometry(grid, lds);
    return MGP_OK;
  }
}

template int launch_fused_wide<float>(const FusedArgs&, hipStream_t);
template int launch_fused_wide<double>(const FusedArgs&, hipStream_t);

}  // namespace mgp

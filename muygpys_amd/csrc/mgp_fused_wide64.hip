// Register-resident fused kernel for wide neighbourhoods in fp64: 65 <= k + 1 + R <= 128, k > 64.
//
// mgp_fused_wide.hip keeps a lane's 128-entry row in 128 registers; 128 doubles are 256.  Here a row
// is shared by TWO lanes -- one neighbourhood is 128 slots x 2 = 256 threads (four wavefronts, one
// workgroup): lane (i, h), i = tid & 127, h = tid >> 7 (uniform per wave), owns the 4-column blocks
// b = h, h + 2, h + 4, ... of row i (cyclic, so that both lanes keep trailing work until the end of
// the elimination): 16 blocks = 32 two-double groups = 128 registers.
//
//   * distances: cyclic pair scheme with NP = 128 as in the fp32 kernel (4 own rows x 16 partner
//     rows per slot); lane (i, h) takes the partner rows p = 8 h + 1 .. 8 h + 8: 32 pairs per lane;
//   * exchange through the packed lower-triangular LDS matrix (rows padded to whole 16-byte groups:
//     row r starts at tri(r) = 2 (a + 1)(a + (r & 1)), a = r >> 1; 67 KB), then every lane reads the
//     blocks it owns;
//   * blocked Cholesky, four columns per exchange, exactly the fp32 kernel's steps; what changes is
//     who holds what: the owner lanes of block b (h = b & 1) post the raw block entries and the
//     eliminated ones, the other lane of the row takes its row's raw entries from LDS; the trailing
//     update runs over the blocks a lane owns at or behind b.
//
// Same algebra as everywhere: after k steps var = S[q][q], mean_r = -S[q+1+r][q],
// y_r^T K^-1 y_r = -S[q+1+r][q+1+r] (_src/gp/muygps/numpy.py:17-67,
// _src/optimize/scale/numpy.py:9-15).  2 workgroups per CU (78 KB of LDS each), 2 waves per SIMD.
#include "mgp_wave_common.h"

#include <cstdio>
#include <cstdlib>

#ifndef MGP_WIDE64_PRIO
#define MGP_WIDE64_PRIO 2
#endif

#ifndef MGP_WIDE64_SPLIT
#define MGP_WIDE64_SPLIT 1  // the waves of rows 0 .. 63 stop at their last lower-triangle block
#endif

namespace mgp {

struct Wide64Geom {
  int q, dst, xs, vec_ok;
};

// NB (round 5, as NG of mgp_fused_wide.hip): the 4-column blocks of the system, even, 4 NB >= k + 1 + R -- query and
// responses in the last 1 + R of those slots, rows and trailing updates end there; and the waves that hold rows
// 0 .. 63 stop at block 15 (their rows have no lower-triangle entry beyond) and only keep the barriers after it.
template <int NB>
__global__ __launch_bounds__(256, 2) void fused_wide64_kernel(FusedArgs a, Wide64Geom g) {
  using T = double;
  constexpr int NP = 128;        // slots
  constexpr int NT = 256;        // threads
  // NB <= 24 (k <= 94): the pair ring of mgp_fused_wide.hip -- the k + 1 real rows instead of all 128 slots, BP partners
  // per own row (even: the two lanes of a slot take half each; 4 BP >= 2 NB >= (k + 1) / 2): k = 70 13.0 -> 15.0 M/s.
  // Larger systems keep the 128-slot ring, as a SEPARATE copy of the three places it touches: there the ring saves an
  // eighth of the pairs at most and its addressing (a compare and a select per row address where the 128-slot ring
  // has a mask) costs more -- k = 100 equal, k = 126 8.3 -> 7.7 -- and a version that folded both forms into one body
  // ran 10-15 % slower than either (round 5, measured).
  constexpr bool RING = NB <= 24;
  constexpr int BA = 4, BP = RING ? 2 * ((NB + 3) / 4) : NP / 8, BPH = BP / 2;  // own rows, partner rows per slot / per lane
  auto own_offset = [](int j) { return j == 0 ? 0 : (j + 1) * BP + 1; };
  constexpr int E = 2, CH = 4;
  static_assert(NB % 2 == 0 && NB >= 18 && NB <= NP / 4, "4-column blocks of a row, shared by two lanes");
  constexpr int LB = NB / 2;     // ... owned by one lane
  constexpr int TRI = 2 * 64 * 65 + 2 * NP;  // packed lower-triangular exchange matrix + over-read pad
  using V = v16<T>::type;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int k = a.k, R = a.R, d = a.d, q = g.q, dst = g.dst, xs = g.xs;
  const int tile_elems = NP * xs > TRI ? NP * xs : TRI;
  T* tile = reinterpret_cast<T*>(smem);   // feature tile, later the exchange matrix
  T* rawbuf = tile + tile_elems;          // 128 x 4 raw block entries (row i at rawbuf + 4 i)
  T* ubuf = rawbuf + NP * 4;              // 4 x 128 eliminated entries, transposed: ubuf[m * 128 + i]
  T* ilbuf = ubuf + NP * 4;               // dst inverse length scales (Anisotropy)
  int64_t* idxbuf = reinterpret_cast<int64_t*>(ilbuf + dst);  // 128 row offsets

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
  auto tri = [](int r) { const int aa = r >> 1; return 2 * (aa + 1) * (aa + (r & 1)); };

  for (int64_t nb = blockIdx.x; nb < a.b; nb += gridDim.x) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));  // keep per-lane addresses of the unrolled phases out of LICM
    const int i = tid & (NP - 1);
    const int h = __builtin_amdgcn_readfirstlane(tid >> 7);  // wave-uniform
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
    if (h == 0) idxbuf[i] = myidx * (int64_t)d;

    // ---- phase 1: stage the features (one stage: d <= 64) -----------------------------------------
    const int w = d, wp = (d + CH - 1) / CH * CH;
    __syncthreads();
    if constexpr (RING) {
      // tile row r = ring position r: the neighbours, then the query at row k
      if (g.vec_ok) {
        const int c16 = w / E, c16p = wp / E;
        for (int t = tid; t < (k + 1) * c16p; t += NT) {
          const int row = t / c16p, c = t - row * c16p;
          V v = V(0);
          if (c < c16) v = *reinterpret_cast<const V*>((row < k ? feat_nn + idxbuf[row] : feat_q + idxbuf[q]) + c * E);
          *reinterpret_cast<V*>(tile + row * xs + c * E) = v;
        }
      } else {
        for (int t = tid; t < (k + 1) * wp; t += NT) {
          const int row = t / wp, c = t - row * wp;
          T v = T(0);
          if (c < w) v = (row < k ? feat_nn + idxbuf[row] : feat_q + idxbuf[q])[c];
          tile[row * xs + c] = v;
        }
      }
    } else if (g.vec_ok) {
      const int c16 = w / E, c16p = wp / E;
      for (int t = tid; t < NP * c16p; t += NT) {
        const int row = t / c16p, c = t - row * c16p;
        V v = V(0);
        if (c < c16 && (row < k || row == q))
          v = *reinterpret_cast<const V*>((row < k ? feat_nn : feat_q) + idxbuf[row] + c * E);
        *reinterpret_cast<V*>(tile + row * xs + c * E) = v;
      }
    } else {
      for (int t = tid; t < NP * wp; t += NT) {
        const int row = t / wp, c = t - row * wp;
        T v = T(0);
        if (c < w && (row < k || row == q)) v = ((row < k ? feat_nn : feat_q) + idxbuf[row])[c];
        tile[row * xs + c] = v;
      }
    }
    if (aniso)
      for (int c = tid; c < wp; c += NT) ilbuf[c] = c < w ? T(1) / ls[c] : T(0);
    __syncthreads();

    // ---- phase 2: squared distances of the lane's 32 pairs (own row j, partner 8 h + p), then
    //      covariances; two halves of the own rows so that 16 accumulators are live at a time -------
#if MGP_WIDE64_PRIO
    __builtin_amdgcn_s_setprio(1);
#endif
    T kv[BA * BPH];
    const int M = k + 1;                                  // (RING) rows of the pair ring
    const int ir = RING ? (i < M ? i : i - M) : i;         // ring position of the slot (M >= 66: one wrap)
    auto ringrow = [&](int x) {                            // row of the tile at ring position x (< 2 M)
      if constexpr (RING) return x >= M ? x - M : x;
      else return x & (NP - 1);
    };
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      constexpr int HB = BA / 2;
      T acc[HB * BPH];
#pragma unroll
      for (int s = 0; s < HB * BPH; ++s) acc[s] = T(0);
      auto chunks = [&](auto anis) {
        constexpr bool ANISO = decltype(anis)::value != 0;
        // partner rows requested together: four, or -- RING, where a lane's 5 or 6 partners are no multiple of four --
        // all of them (a guard inside the unrolled loops instead cost the 128-slot shapes 10-15 %: measured)
        constexpr int PB = RING ? BPH : 4;
        static_assert(BPH % PB == 0, "whole partner batches");
        for (int c0 = 0; c0 < wp; c0 += CH) {
          V own0[HB], own1[HB];
#pragma unroll
          for (int j = 0; j < HB; ++j) {
            const T* xj = tile + ringrow(ir + own_offset(half * HB + j)) * xs + c0;
            own0[j] = *reinterpret_cast<const V*>(xj);
            own1[j] = *reinterpret_cast<const V*>(xj + E);
          }
          V il0 = V(1), il1 = V(1);
          if constexpr (ANISO) {
            il0 = *reinterpret_cast<const V*>(ilbuf + c0);
            il1 = *reinterpret_cast<const V*>(ilbuf + c0 + E);
          }
#pragma unroll
          for (int s0 = 0; s0 < BPH; s0 += PB) {
            V o0[PB], o1[PB];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
              const T* xo = tile + ringrow(ir + h * BPH + s0 + u + 1) * xs + c0;
              o0[u] = *reinterpret_cast<const V*>(xo);
              o1[u] = *reinterpret_cast<const V*>(xo + E);
            }
#pragma unroll
            for (int u = 0; u < PB; ++u)
#pragma unroll
              for (int j = 0; j < HB; ++j) {
                if constexpr (ANISO) {
                  accum(acc[j * BPH + s0 + u], (own0[j] - o0[u]) * il0);
                  accum(acc[j * BPH + s0 + u], (own1[j] - o1[u]) * il1);
                } else {
                  accum(acc[j * BPH + s0 + u], own0[j] - o0[u]);
                  accum(acc[j * BPH + s0 + u], own1[j] - o1[u]);
                }
              }
          }
        }
      };
      if (aniso) chunks(ic<1>{});
      else chunks(ic<0>{});
      kernel_dispatch(a.kernel_id, a.metric_id, [&](auto kid, auto mid) {
        constexpr int KID = decltype(kid)::value, MID = decltype(mid)::value;
        static_for<HB * BPH>([&](auto sc) {
          constexpr int s = decltype(sc)::value;
          kv[half * HB * BPH + s] = cov_from_sqdist<T>(acc[s], KID, MID, post_scale);
        });
      });
    }

    // ---- phase 3: covariances -> packed lower-triangular exchange matrix -> the lane's blocks -------
#if MGP_WIDE64_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    const T mydiag = i < k ? T(1) + myeps : (i <= q ? T(1) : T(0));
    V A[2 * LB];
    __syncthreads();  // every lane is done reading the feature tile (the exchange matrix aliases it)
    {
      int i3 = i;
      asm volatile("" : "+v"(i3));
      const int dump = tri(NP - 1) + NP;  // behind the last row
      if constexpr (RING) {
        const int ir3 = i3 < M ? i3 : i3 - M;
        static_for<BA * BPH>([&](auto sc) {
          constexpr int s = decltype(sc)::value;  // pair (own row j = s / BPH, partner BPH h + s % BPH + 1)
          // ring positions -> slots: the query (position k) is slot q; a pair met from both ends writes the same value twice
          const int r1 = ringrow(ir3 + own_offset(s / BPH)), c = ringrow(ir3 + h * BPH + s % BPH + 1);
          const int s1 = r1 < k ? r1 : q, sc2 = c < k ? c : q;
          const int hi = max(s1, sc2), lo = min(s1, sc2);
          tile[hi != lo ? tri(hi) + lo : dump] = kv[s];
        });
        // what no pair writes: the padding slots k .. q - 1 (rows of zeros) and their columns in the query's row
        if (h == 1 && i3 >= k && i3 <= q) {
          const int rowz = tri(i3);
          for (int cz = i3 == q ? k : 0; cz < i3; ++cz) tile[rowz + cz] = T(0);
        }
      } else {
      static_for<BA * BPH>([&](auto sc) {
        constexpr int s = decltype(sc)::value;  // pair (own row j = s / BPH, partner 8 h + s % BPH + 1)
        const int r1 = (i3 + own_offset(s / BPH)) & (NP - 1);
        const int c = (i3 + h * BPH + s % BPH + 1) & (NP - 1);
        const int hi = max(r1, c), lo = min(r1, c);
        const T v = (lo < k && (hi < k || hi == q)) ? kv[s] : T(0);
        // cyclic distance 64 is met from both ends (own row 3, partner 16 <-> own row 0 ...): both write the same value
        tile[hi <= q ? tri(hi) + lo : dump] = v;
      });
      }
      const int myrow = tri(i3);
      if (h == 0) {
        tile[myrow + i3] = mydiag;
        for (int r = 0; r < R; ++r)
          if (i3 <= q + 1 + r) tile[tri(q + 1 + r) + i3] = i3 < k ? targets[mytg * (int64_t)R + r] : T(0);
      }
      __syncthreads();
      // local group g2 = 2 lb + e  <->  global block b = 2 lb + h, columns 4 b + 2 e .. + 1
#pragma unroll
      for (int lb = 0; lb < LB; ++lb)
#pragma unroll
        for (int e = 0; e < 2; ++e)
          A[2 * lb + e] = *reinterpret_cast<const V*>(tile + myrow + 4 * (2 * lb + h) + 2 * e);
    }

    // ---- phase 4: blocked Cholesky, FOUR columns per exchange ------------------------------------
#if MGP_WIDE64_PRIO
    __builtin_amdgcn_s_setprio(MGP_WIDE64_PRIO);
#endif
    bool bad = false;
    auto eliminate = [&](auto llc) {
    constexpr int LBL = decltype(llc)::value;  // this wave's rows have lower-triangle entries in local blocks 0 .. LBL - 1
    static_for<NB>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      constexpr int J0 = 4 * b;
      constexpr int lb = b >> 1;      // the owner's local block
      constexpr int ho = b & 1;       // the owner half
      if constexpr (b >= 2 * LBL) {
        if (J0 < k) {  // (uniform) a block of the lower rows only: keep its two barriers
          __syncthreads();
          __syncthreads();
        }
      } else if (J0 < k) {  // uniform
        const int mlim = min(4, k - J0);
        T pg[4];
        if (h == ho) {  // wave-uniform: the owner posts its raw block entries
          *reinterpret_cast<V*>(rawbuf + i * 4) = A[2 * lb];
          *reinterpret_cast<V*>(rawbuf + i * 4 + 2) = A[2 * lb + 1];
        }
        __syncthreads();
        // diagonal block rows J0 .. J0+3 (raw), factored redundantly by every lane
        T dr[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const V lo2 = *reinterpret_cast<const V*>(rawbuf + (J0 + r) * 4);
          const V hi2 = *reinterpret_cast<const V*>(rawbuf + (J0 + r) * 4 + 2);
          dr[r][0] = lo2[0];
          dr[r][1] = lo2[1];
          dr[r][2] = hi2[0];
          dr[r][3] = hi2[1];
        }
        {
          const V lo2 = *reinterpret_cast<const V*>(rawbuf + i * 4);
          const V hi2 = *reinterpret_cast<const V*>(rawbuf + i * 4 + 2);
          pg[0] = lo2[0];
          pg[1] = lo2[1];
          pg[2] = hi2[0];
          pg[3] = hi2[1];
        }
        T rp[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const T pm = dr[m][m];
          const bool on = m < mlim;
          bad = bad || (on && !(pm > T(0)));
          rp[m] = on ? pivot_rcp(pm) : T(0);
#pragma unroll
          for (int r = m + 1; r < 4; ++r) {
            const T t = dr[r][m] * rp[m];
#pragma unroll
            for (int c = m + 1; c <= r; ++c) dr[r][c] = fma_t(-t, dr[c][m], dr[r][c]);
          }
        }
        T nt[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          // a select, not a product with rp = 0 (stale LDS beyond a row's diagonal may hold NaN bits)
          nt[m] = m < mlim ? -pg[m] * rp[m] : T(0);
#pragma unroll
          for (int c = m + 1; c < 4; ++c) pg[c] = fma_t(nt[m], dr[c][m], pg[c]);
        }
        if (h == ho) {
#pragma unroll
          for (int m = 0; m < 4; ++m) ubuf[m * NP + i] = m < mlim ? pg[m] : T(0);
        }
        __syncthreads();
        // trailing blocks the lane owns, the block itself included for its owner (it stays current:
        // the outputs read the last blocks); loads of a block issued together, then its FMAs
        constexpr int first_owner = lb;                       // owner: local blocks lb ..
        constexpr int first_other = (b + 1) >> 1;             // other lane (h != ho): global block 2 lb' + h >= b
        const int lb0 = h == ho ? first_owner : first_other;  // wave-uniform
#pragma unroll
        for (int lbb = 0; lbb < LBL; ++lbb) {
          if (lbb >= (first_owner < first_other ? first_owner : first_other) && lbb >= lb0) {
            const int gb = 2 * lbb + h;  // global block
            V cv[2][4];
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
              for (int m = 0; m < 4; ++m) cv[e][m] = *reinterpret_cast<const V*>(ubuf + m * NP + 4 * gb + 2 * e);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
              for (int m = 0; m < 4; ++m) A[2 * lbb + e] = cv[e][m] * V(nt[m]) + A[2 * lbb + e];
          }
        }
      }
    });
    };
    if (MGP_WIDE64_SPLIT && i < 64) eliminate(ic<(LB < 8 ? LB : 8)>{});  // (wave-uniform: waves 0 and 2 hold rows 0 .. 63)
    else eliminate(ic<LB>{});

    // ---- phase 5: Schur block -> outputs --------------------------------------------------------
#if MGP_WIDE64_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    // column c of row i is held by lane (i, (c >> 2) & 1) in local group 2 (c >> 3) + ((c >> 1) & 1),
    // element c & 1; q >= 4 NB - 17, so a compare-select sweep over the lane's last four blocks (the
    // last 32 columns) finds column q and the diagonal
    T aq = T(0), aii = T(0);
    bool has_q = false, has_i = false;
#pragma unroll
    for (int lbb = LB - 4; lbb < LB; ++lbb)
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          const int c = 4 * (2 * lbb + h) + 2 * e + x;
          const T v = A[2 * lbb + e][x];
          if (c == q) { aq = v; has_q = true; }
          if (c == i) { aii = v; has_i = true; }
        }
    T* mean = static_cast<T*>(a.mean);
    T* var = static_cast<T*>(a.var);
    T* yk = static_cast<T*>(a.ykinvy);
    if (i == q && has_q) {
      var[nb] = bad ? num<T>::nan() : aq;
      if (bad && a.info) atomicAdd(a.info, 1);
    } else if (i > q && i <= q + R) {  // (lanes past q + R hold no row of the system when 4 NB < 128)
      const int r = i - q - 1;
      if (has_q) mean[nb * R + r] = bad ? num<T>::nan() : -aq;
      if (has_i && yk) yk[nb * R + r] = bad ? num<T>::nan() : -aii;
    }
  }
}

int launch_fused_wide64(const FusedArgs& a, hipStream_t stream) {
  constexpr int NP = 128, E = 2, CH = 4, TRI = 2 * 64 * 65 + 2 * NP;
  const int rows = a.k + 1 + a.R;
  if (rows < 65 || rows > NP || a.k <= 64 || a.R > 16 || a.d > 64 || a.packed_nn != nullptr || a.coeffs != nullptr)
    return MGP_EUNSUPPORTED;
#ifdef MGP_WIDE_FORCE_NG32
  const int nb = 32;
#else
  const int nb = rows <= 72 ? 18 : rows <= 80 ? 20 : rows <= 88 ? 22 : rows <= 96 ? 24 : rows <= 104 ? 26
                 : rows <= 112 ? 28 : rows <= 120 ? 30 : 32;
#endif
  Wide64Geom g;
  g.q = 4 * nb - 1 - a.R;
  const int dpad = (a.d + CH - 1) / CH * CH;
  g.dst = dpad < 64 ? dpad : 64;
  g.xs = g.dst + E;
  const uintptr_t align = (uintptr_t)a.feat_q | (uintptr_t)a.feat_nn;
  g.vec_ok = (a.d % E == 0) && (align % 16 == 0);
  const size_t tile_elems = (size_t)NP * g.xs > TRI ? (size_t)NP * g.xs : TRI;
  size_t lds = (tile_elems + 2 * NP * 4 + g.dst) * sizeof(double) + NP * sizeof(int64_t);
  lds = (lds + 15) & ~(size_t)15;
  const void* fn = nullptr;
  static Residency res[8];
  int ri = 0;
  switch (nb) {
#define MGP_WIDE64_CASE(N, I) case N: fn = reinterpret_cast<const void*>(&fused_wide64_kernel<N>); ri = I; break;
    MGP_WIDE64_CASE(18, 0) MGP_WIDE64_CASE(20, 1) MGP_WIDE64_CASE(22, 2) MGP_WIDE64_CASE(24, 3)
    MGP_WIDE64_CASE(26, 4) MGP_WIDE64_CASE(28, 5) MGP_WIDE64_CASE(30, 6)
    default: fn = reinterpret_cast<const void*>(&fused_wide64_kernel<32>); ri = 7; break;
#undef MGP_WIDE64_CASE
  }
  int per_cu = 0, cus = 0;
  const int rc = res[ri].lookup(fn, 256, lds, &per_cu, &cus);
  if (rc != MGP_OK) return rc;
  int64_t grid = (int64_t)cus * per_cu;
  if (grid > a.b) grid = a.b;
  static const bool trace = getenv("MGP_TRACE") != nullptr;
  if (trace)
    fprintf(stderr, "mgp: fused_wide64_kernel b=%lld k=%d d=%d R=%d grid=%lld lds=%zu per_cu=%d\n", (long long)a.b, a.k, a.d,
            a.R, (long long)grid, lds, per_cu);
  {
    void* kargs[] = {const_cast<FusedArgs*>(&a), &g};
    const hipError_t le = hipLaunchKernel(fn, dim3((unsigned)grid), dim3(256), kargs, lds, stream);
    if (le != hipSuccess) return -(1000 + (int)le);
  }
  MGP_HIP_CHECK_LAUNCH();
  note_launch("mgp::fused_wide64_kernel<%d>", nb);
  note_launch_geometry(grid, lds);
  return MGP_OK;
}

}  // namespace mgp

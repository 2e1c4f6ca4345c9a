// Exact brute-force k-nearest-neighbour scan on the matrix cores.
//
// The reference finds neighbours on the CPU (scikit-learn / hnswlib behind NN_Wrapper,
// src/MuyGPyS/neighbors.py:32-262).  This kernel is the GPU counterpart of the exact search: the
// one GEMM-shaped step around the hot path, so it runs on MFMA (fp32 in / fp32 accumulate,
// v_mfma_f32_32x32x2_f32: bit-for-bit an fmaf chain).
//
// Work split: a workgroup of four waves owns 128 queries (32 per wave, their feature rows
// resident in registers as the MFMA A operand) and streams the training table through LDS in
// double-buffered tiles of 64 rows (global_load_lds, the next tile in flight under the MFMAs).  Per 32x32 block of (query, training point) pairs a wave issues DP/2 MFMAs
// that accumulate  q.x - |x|^2/2  (the accumulator starts at -|x|^2/2 of the lane's column), so a
// pair is closer than the query's current k-th best distance tau exactly when
//      acc > (|q|^2 - tau) / 2,
// ONE compare per pair.  Selection is two-level: the rare pairs that pass are appended to a small
// per-query queue in LDS; after every tile the owning wave drains the queues into the query's
// k-best list (one list element per lane, replace-the-maximum with two cross-lane reductions per
// accepted candidate) and tightens tau.  The lists start from an exact top-k over the first rows of
// the table (host side), so the expected number of queue entries over a whole scan is about
// k ln(N / N0) per query.  A queue that overflows marks its query; the host recomputes those
// queries on the dense path, so the result is exact in every case.
//
// Distances in the lists are the Gram form |q|^2 + |x|^2 - 2 q.x; the host re-measures the k
// winners in difference form and sorts them (as the dense path does).
#include "mgp_args.h"

namespace mgp {

typedef float f16x __attribute__((ext_vector_type(16)));
typedef float f4x __attribute__((ext_vector_type(4)));

constexpr int KNN_TN = 64;    // training rows per staged tile
constexpr int KNN_QW = 32;    // queries per wave
constexpr int KNN_QB = 128;   // queries per workgroup
constexpr int KNN_CAP = 16;   // queue entries per query

template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return v;
}

template <int DP>
__global__ __launch_bounds__(256) void knn_scan_kernel(KnnArgs a) {
  constexpr int XS = DP + 4;      // LDS row stride: an odd number of 16-byte slots
  constexpr int HD = DP / 2;      // features per lane half
  __shared__ __attribute__((aligned(16))) float tile[2][KNN_TN * XS + 64];  // + slack of the last partial pass
  __shared__ __attribute__((aligned(16))) float xn_tile[2][KNN_TN];
  __shared__ float q_d[KNN_QB * KNN_CAP];
  __shared__ int q_i[KNN_QB * KNN_CAP];
  __shared__ int q_cnt[KNN_QB];
  __shared__ float tau_s[KNN_QB];

  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int half = lane >> 5, r32 = lane & 31;
  const int64_t qbase = (int64_t)blockIdx.x * KNN_QB;
  const int d = a.d, k = a.k;

  // ---- per-query state -----------------------------------------------------------------------
  if (tid < KNN_QB) {
    const int64_t q = qbase + tid;
    float t = -__builtin_inff();  // rows past the end never accept anything
    if (q < a.m) {
      t = a.best_d[q * k];
      for (int j = 1; j < k; ++j) t = fmaxf(t, a.best_d[q * k + j]);
    }
    tau_s[tid] = t;
    q_cnt[tid] = 0;
  }
  // A operand: lane (r32, half) holds features [half*HD, half*HD + HD) of query qbase + 32 w + r32
  float aq[HD];
  {
    const int64_t q = qbase + w * KNN_QW + r32;
    const float* qrow = a.queries + (q < a.m ? q : 0) * (int64_t)d;
#pragma unroll
    for (int j = 0; j < HD / 4; ++j) {
      const int c = half * HD + 4 * j;
      f4x v = {0.f, 0.f, 0.f, 0.f};
      if (c < d) v = *reinterpret_cast<const f4x*>(qrow + c);
      aq[4 * j] = v.x, aq[4 * j + 1] = v.y, aq[4 * j + 2] = v.z, aq[4 * j + 3] = v.w;
    }
  }
  __syncthreads();
  // accumulator register v of this lane belongs to query row  8 (v/4) + 4 half + v%4  of the wave
  float qn[16], thr[16];
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    const int row = 8 * (v >> 2) + 4 * half + (v & 3);
    const int64_t q = qbase + w * KNN_QW + row;
    qn[v] = q < a.m ? a.query_sqn[q] : 0.f;
    thr[v] = 0.5f * (qn[v] - tau_s[w * KNN_QW + row]);
  }

  // Double-buffered staging with direct global->LDS loads: the tile after the current one is in
  // flight while the MFMAs run, one barrier per tile.  A wave-instruction fills 64 consecutive
  // 16-byte slots; slot sigma of a tile is row sigma / SPR, column sigma % SPR (the padding column
  // and the columns past d repeat the row's last data slot -- finite values the zero-padded query
  // operand cancels).  Rows past n repeat row n-1; their |x|^2 is +inf (host-padded table).
  constexpr int SPR = XS / 4;
  constexpr int NPASS = (KNN_TN * SPR + 63) / 64;
  const int dslots = d / 4;
  auto issue_tile = [&](int buf, int64_t t0) {
    for (int p = w; p < NPASS; p += 4) {
      const int sigma = 64 * p + lane;
      const int row = min(sigma / SPR, KNN_TN - 1);
      const int c = min(sigma - row * SPR, dslots - 1);
      const int64_t grow = min(t0 + row, a.n - 1);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(a.train + grow * (int64_t)d + c * 4),
          (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(tile[buf]) + p * 1024), 16, 0, 0);
    }
    if (w == 3 && lane < KNN_TN / 4)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(a.train_sqn + t0 + lane * 4),
          (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(xn_tile[buf])), 16, 0, 0);
  };
  issue_tile(0, a.start);
  int buf = 0;
  for (int64_t t0 = a.start; t0 < a.n; t0 += KNN_TN, buf ^= 1) {
    __builtin_amdgcn_s_waitcnt(0);  // this wave's share of tile `buf` has landed
    __syncthreads();                // ... everyone's has, and nobody still reads the other buffer
    if (t0 + KNN_TN < a.n) issue_tile(buf ^ 1, t0 + KNN_TN);
    const float* tl = tile[buf];
    const float* xnt = xn_tile[buf];

    // ---- two 32 x 32 blocks per wave: MFMA, one compare per pair, rare queue pushes ---------
#pragma unroll
    for (int ct = 0; ct < KNN_TN / 32; ++ct) {
      const int col = ct * 32 + r32;
      const float* xrow = tl + col * XS + half * HD;
      float bx[HD];
#pragma unroll
      for (int j = 0; j < HD / 4; ++j) {
        const f4x v = *reinterpret_cast<const f4x*>(xrow + 4 * j);
        bx[4 * j] = v.x, bx[4 * j + 1] = v.y, bx[4 * j + 2] = v.z, bx[4 * j + 3] = v.w;
      }
      const float xn = xnt[col];
      const float c0 = -0.5f * xn;
      f16x acc = {c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0};
#pragma unroll
      for (int t = 0; t < HD; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[t], bx[t], acc, 0, 0, 0);
      bool any = false;
#pragma unroll
      for (int v = 0; v < 16; ++v) any = any || acc[v] > thr[v];
      if (any) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          if (acc[v] > thr[v]) {
            const int qslot = w * KNN_QW + 8 * (v >> 2) + 4 * half + (v & 3);
            const int pos = atomicAdd(&q_cnt[qslot], 1);
            if (pos < KNN_CAP) {
              q_d[qslot * KNN_CAP + pos] = qn[v] - 2.0f * acc[v];
              q_i[qslot * KNN_CAP + pos] = (int)(t0 + col);
            }
          }
        }
      }
    }

    // ---- drain this wave's queues into the k-best lists (wave-private state: no barrier) ------
    const int mycnt = lane < KNN_QW ? q_cnt[w * KNN_QW + lane] : 0;
    unsigned long long pending = __ballot(mycnt > 0);
    bool drained = false;
    while (pending) {
      const int r = __builtin_ctzll(pending);
      pending &= pending - 1;
      const int qslot = w * KNN_QW + r;
      const int64_t q = qbase + qslot;
      int cnt = q_cnt[qslot];
      if (q >= a.m) continue;
      if (cnt > KNN_CAP) {
        if (lane == 0) a.overflow[q] = 1;
        cnt = KNN_CAP;
      }
      const int self = a.self_idx ? (int)a.self_idx[q] : -1;
      float bd = lane < k ? a.best_d[q * k + lane] : -__builtin_inff();
      int bi = lane < k ? a.best_i[q * k + lane] : -1;
      float tau = wave_max(bd);
      for (int c = 0; c < cnt; ++c) {
        const float cd = q_d[qslot * KNN_CAP + c];
        const int ci = q_i[qslot * KNN_CAP + c];
        if (ci == self || !(cd < tau)) continue;
        // replace the (first) lane that holds the current maximum
        const unsigned long long at_max = __ballot(bd == tau);
        if (lane == __builtin_ctzll(at_max)) bd = cd, bi = ci;
        tau = wave_max(bd);
      }
      if (lane < k) {
        a.best_d[q * k + lane] = bd;
        a.best_i[q * k + lane] = bi;
      }
      if (lane == 0) {
        tau_s[qslot] = tau;
        q_cnt[qslot] = 0;
      }
      drained = true;
    }
    if (__any(drained)) {
      // only this wave's rows can have changed (wave-private slots of tau_s): no barrier needed
      // beyond making the lane-0 writes visible to the wave
      __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
      for (int v = 0; v < 16; ++v)
        thr[v] = 0.5f * (qn[v] - tau_s[w * KNN_QW + 8 * (v >> 2) + 4 * half + (v & 3)]);
    }
  }
}

template <int DP>
static int launch_knn_dp(const KnnArgs& a, hipStream_t stream) {
  const int64_t grid = (a.m + KNN_QB - 1) / KNN_QB;
  hipLaunchKernelGGL(knn_scan_kernel<DP>, dim3((unsigned)grid), dim3(256), 0, stream, a);
  MGP_HIP_CHECK_LAUNCH();
  return MGP_OK;
}

int launch_knn_scan(const KnnArgs& a, hipStream_t stream) {
  if (a.k < 1 || a.k > 64 || a.d < 4 || a.d % 4 != 0 || a.d > 64) return MGP_EUNSUPPORTED;
  if (((uintptr_t)a.train | (uintptr_t)a.queries | (uintptr_t)(a.train_sqn + a.start)) % 16 != 0)
    return MGP_EUNSUPPORTED;
  if (a.n >= (int64_t)1 << 31) return MGP_EUNSUPPORTED;  // list indices are 32-bit
  const int dp = (a.d + 7) / 8 * 8;
  switch (dp) {
    case 8: return launch_knn_dp<8>(a, stream);
    case 16: return launch_knn_dp<16>(a, stream);
    case 24: return launch_knn_dp<24>(a, stream);
    case 32: return launch_knn_dp<32>(a, stream);
    case 40: return launch_knn_dp<40>(a, stream);
    case 48: return launch_knn_dp<48>(a, stream);
    case 56: return launch_knn_dp<56>(a, stream);
    default: return launch_knn_dp<64>(a, stream);
  }
}

}  // namespace mgp

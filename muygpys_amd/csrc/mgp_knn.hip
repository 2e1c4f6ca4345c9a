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

// 16 bytes per lane from global memory straight into LDS at (wave-uniform byte offset) + 16 * lane.
// Raw instruction on purpose: through the builtin the compiler orders every later LDS write (the
// candidate queues) behind the transfer with s_waitcnt vmcnt(0), i.e. a full memory latency inside
// the compare loop for one block in six; the kernels wait for their tiles explicitly instead.
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_byte_offset) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off"
               :
               : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_byte_offset))
               : "memory");
}
__device__ __forceinline__ unsigned lds_offset(const void* p) { return (unsigned)(uintptr_t)p; }

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
  // dynamic LDS (one extern array, carved by hand: the compiler then does not tie the LDS reads of
  // the tile being consumed to the direct-to-LDS loads filling the other buffer -- with separate
  // static arrays it inserts s_waitcnt vmcnt(0) before the first read and the prefetch never overlaps)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile0 = reinterpret_cast<float*>(smem);                 // [2][KNN_TN * XS]
  float* xn0 = tile0 + 2 * KNN_TN * XS;                          // [2][KNN_TN]
  float* q_d = xn0 + 2 * KNN_TN;                                 // [KNN_QB * KNN_CAP]
  int* q_i = reinterpret_cast<int*>(q_d + KNN_QB * KNN_CAP);     // [KNN_QB * KNN_CAP]
  int* q_cnt = q_i + KNN_QB * KNN_CAP;                           // [KNN_QB]
  float* tau_s = reinterpret_cast<float*>(q_cnt + KNN_QB);       // [KNN_QB]

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
      lds_dma16(a.train + grow * (int64_t)d + c * 4, lds_offset(tile0 + buf * KNN_TN * XS) + p * 1024);
    }
    if (w == 3 && lane < KNN_TN / 4) lds_dma16(a.train_sqn + t0 + lane * 4, lds_offset(xn0 + buf * KNN_TN));
  };
  // Every workgroup walks the whole table, but starts at its own tile and wraps around: in lockstep
  // all workgroups would hammer the same few L2 channels (measured: ~3 us per tile and workgroup,
  // whatever the arithmetic), staggered they spread over the memory system.
  const int64_t ntiles = (a.n - a.start + KNN_TN - 1) / KNN_TN;
  const int64_t tile_off = ((int64_t)blockIdx.x * 7919) % ntiles;
  // (running tile index instead of (tj + tile_off) % ntiles: see knn_scan_bf16x3_kernel)
  auto tile_after = [&](int64_t t) { return t + 1 == ntiles ? (int64_t)0 : t + 1; };
  int64_t tcur = tile_off;
  issue_tile(0, a.start + tcur * KNN_TN);
  int buf = 0;
  int64_t next_drain = 0;
  for (int64_t tj = 0; tj < ntiles; ++tj, buf ^= 1, tcur = tile_after(tcur)) {
    const int64_t t0 = a.start + tcur * KNN_TN;
    __builtin_amdgcn_s_waitcnt(0);  // this wave's share of tile `buf` has landed
    __syncthreads();                // ... everyone's has, and nobody still reads the other buffer
    if (tj + 1 < ntiles) issue_tile(buf ^ 1, a.start + tile_after(tcur) * KNN_TN);
    const float* tl = tile0 + buf * KNN_TN * XS;
    const float* xnt = xn0 + buf * KNN_TN;

    // ---- two 32 x 32 blocks per wave: both MFMA chains are issued before the first accumulator is
    // read, so the compares of block 0 run under the MFMAs of block 1 ----------------------------
    f16x acc2[KNN_TN / 32];
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
      const float c0 = -0.5f * xnt[col];
      f16x acc = {c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0};
#pragma unroll
      for (int t = 0; t < HD; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[t], bx[t], acc, 0, 0, 0);
      acc2[ct] = acc;
    }
#pragma unroll
    for (int ct = 0; ct < KNN_TN / 32; ++ct) {
      const int col = ct * 32 + r32;
      const f16x acc = acc2[ct];
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
    // batched, with an interval that grows with the rows already seen: see knn_scan_bf16x3_kernel
    if (tj < next_drain && tj + 1 < ntiles) continue;
    next_drain = tj + (tj < 64 ? 1 : tj < 256 ? 4 : tj < 1024 ? 8 : 16);
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
  const size_t lds = (2 * KNN_TN * (DP + 4) + 2 * KNN_TN + 2 * KNN_QB * KNN_CAP + 2 * KNN_QB) * sizeof(float);
  hipLaunchKernelGGL(knn_scan_kernel<DP>, dim3((unsigned)grid), dim3(256), lds, stream, a);
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


// ------------------------------------------------------------------------------------------------
// Split-bf16 pre-filter variant: 4.4 x fewer matrix-pipe cycles and almost no vector work per pair,
// still exact.
//
// Every fp32 feature is split x = hi + lo + r with hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-16 |x|
// (host side, once per table: rows packed as [hi(KP) | lo(KP)]).  Three bf16 MFMA chains
// (hi.hi + hi.lo + lo.hi, v_mfma_f32_32x32x16_bf16, 32 cycles each) give q.x with an error below
// 2^-14 |q| |x| (dropped lo.lo and r terms <= 3.1 * 2^-16 sum|q_i x_i|, fp32 accumulation of
// <= 240 products, Cauchy-Schwarz).  The whole test  q.x - |x|^2/2 + margin > (|q|^2 - tau)/2  is
// evaluated BY the MFMAs: the last two of the KP packed slots are not features but
//      query row:  [ ..., 1, -thr ]      thr = (|q|^2 - tau)/2, nudged down by its own bf16x2 error
//      table row:  [ ..., c,  1   ]      c = -|x|^2/2 + 2^-14 QMAX |x| (+ its own bf16x2 error)
// so the accumulator IS  q.x + c - thr  and a pair survives iff it is positive: per 32 x 32 block the
// vector unit only takes a max over the 16 accumulators of a lane.  When tau of a query tightens,
// the lane that holds the query's last operand piece rewrites the -thr slot in place.
// Survivors are re-measured EXACTLY (fp32 difference form, from the fp32 tables) when their queue
// is drained, so the lists -- and tau -- only ever hold exact distances.
// A workgroup owns 256 queries (64 per wave, two MFMA row blocks).
// ------------------------------------------------------------------------------------------------

#ifndef MGP_KNN_TIMING
#define MGP_KNN_TIMING 0
#endif
#if MGP_KNN_TIMING
// phase timing (experiments only; tools/knn_timing.py): cycle-counter differences summed per wave and phase --
// 0 tile wait + barrier, 1 tile issue, 2 matrix blocks + survivor queues, 3 drain, 4 prologue
__device__ unsigned long long g_knn_timing[8];
#define MGP_KNN_T(slot)                                            \
  {                                                                \
    const unsigned long long tnow_ = __builtin_readcyclecounter(); \
    tacc_[slot] += tnow_ - tlast_;                                 \
    tlast_ = tnow_;                                                \
  }
}  // namespace mgp
extern "C" int mgp_debug_knn_timing(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mgp::g_knn_timing), sizeof(mgp::g_knn_timing)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mgp::g_knn_timing), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
namespace mgp {
#else
#define MGP_KNN_T(slot)
#endif

typedef __bf16 bf8x __attribute__((ext_vector_type(8)));
typedef unsigned int u4x __attribute__((ext_vector_type(4)));

constexpr int KB_RB = 32;     // queries per row block (one 32 x 32 matrix tile's rows)
constexpr int KB_CAP = 16;    // queue entries per query
#ifndef MGP_KNN_STAGGER
#define MGP_KNN_STAGGER 64      // 0: every workgroup starts at the first tile, 1: scattered over the table, n > 1: over a window of n tiles
#endif
#ifndef MGP_KNN_SCAN_BLOCK
#define MGP_KNN_SCAN_BLOCK 16   // list entries loaded together when a survivor is inserted (32: the drain spills more, slower)
#endif
#ifndef MGP_KNN_CADENCE
#define MGP_KNN_CADENCE 64      // drain interval = rows seen / this (0: the fixed schedule of rounds 1-4)
#endif
#ifndef MGP_KNN_REGSTAGE
#define MGP_KNN_REGSTAGE 1
#endif
#ifndef MGP_KNN_SWIZZLE
#define MGP_KNN_SWIZZLE 1
#endif
#ifndef MGP_KNN_PIPE_KP
#define MGP_KNN_PIPE_KP 32     // packed row lengths up to this run the survivor test one column block behind the matrix instructions (32: +2-3 % at d = 16 / 24; 48 spills 38 registers)
#endif
#ifndef MGP_KNN_CADENCE_MAX
#define MGP_KNN_CADENCE_MAX 16384
#endif

__device__ __forceinline__ unsigned bf16_rne(float x) {  // bits of bf16(x), round to nearest even
  const unsigned u = __float_as_uint(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
// -thr as (hi, lo) bf16 bits, with thr first lowered by 2^-14 |thr| (covers the split's own error)
__device__ __forceinline__ void neg_thr_split(float qn, float tau, unsigned& hi, unsigned& lo) {
  float thr = fminf(0.5f * (qn - tau), 1e38f);  // tau = -inf (rows past the end): nothing ever passes
  thr -= 6.103515625e-05f * fabsf(thr);
  const float v = -thr;
  hi = bf16_rne(v);
  lo = bf16_rne(v - __uint_as_float(hi << 16));
}

// TN: training rows per staged tile.  64 for the wider rows; 128 at KP = 16 (d <= 14: BASELINE config 4's d = 8), where a
// 64-row tile is only 12 MFMAs per wave between two workgroup barriers (round 4: the 10 M-point search of config 4 ran at
// 18 % matrix-pipe occupancy, barrier-bound).
// RBN: row blocks of 32 queries per wave.  Two everywhere but at KP = 8, where four fit the registers: a staged tile, its
// barrier, its wait and its LDS-DMA instructions -- a third of a wave's cycles at two -- then serve twice the matrix work
template <int KP, int TN, int NW, int NBUF, int RBN>
__global__ __launch_bounds__(64 * NW, (RBN > 2 ? 3 : KP <= 16 ? 4 : KP <= 48 ? 3 : 2)) void knn_scan_bf16x3_kernel(KnnPackedArgs a) {
  constexpr int KB_QW = KB_RB * RBN;    // queries per wave
  constexpr int KB_QB = KB_QW * NW;     // queries per workgroup
  // KP == 8 (d <= 8: BASELINE config 4): rows [hi(8) | lo(8) | T(8)] and TWO chains per block instead of three -- K = 16
  // takes eight slots from each half of the wave, so  A = [q_hi | q_lo] x B = [x_hi | x_hi]  is q_hi.x_hi + q_lo.x_hi in
  // one instruction and  A = [q_hi | T_q] x B = [x_lo | T_x]  adds q_hi.x_lo and the threshold terms
  // (T_q = [1, 1, -thr_hi, -thr_lo, 0 ..], T_x = [c_hi, c_lo, 1, 1, 0 ..]): the same sum as the three-chain form with
  // six of its sixteen slots padding, a third less matrix work and 48 instead of 64 staged bytes per row
  constexpr bool K8 = KP == 8;
  constexpr int RB = K8 ? 48 : 4 * KP;  // bytes of a packed row: KP bf16 hi + KP bf16 lo (K8: + the eight T slots)
  // KP = 16 (MGP_KNN_SWIZZLE): the staged row is its four 16-byte slots and nothing else -- slot s of row r lies at
  // slot s ^ ((r >> 2) & 3), the transfer's SOURCE address does the permuting (its LDS side is lane-linear) -- which
  // reads as conflict-free as the padded rows (lanes r, r + 4, r + 8, r + 12 of a 16-lane phase share a 16-bank group
  // and now take its four different slots) at 8 transfers per 128-row tile instead of 10 (12 issued)
  constexpr bool SWZ = MGP_KNN_SWIZZLE && KP == 16;
  constexpr int SPR = SWZ || K8 ? RB / 16 : RB / 16 + 1;  // 16-byte slots of a staged row (odd when padded; K8: three)
  constexpr int XSB = SPR * 16;         // LDS row stride in bytes
  constexpr int NQ = K8 ? 1 : KP / 16;  // 16-byte operand pieces per half row = MFMAs per chain
  constexpr int NPASS = TN * SPR / 64;
  static_assert(TN * SPR % 64 == 0, "tile must be whole wave passes");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // carved by hand, see knn_scan_kernel
  char* tile0 = smem;                                                     // [NBUF][TN * XSB]
  int* q_i = reinterpret_cast<int*>(smem + NBUF * TN * XSB);           // [KB_QB * KB_CAP]
  int* q_cnt = q_i + KB_QB * KB_CAP;                                      // [KB_QB]
  float* tau_s = reinterpret_cast<float*>(q_cnt + KB_QB);                 // [KB_QB]

  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int half = lane >> 5, r32 = lane & 31;
  const int64_t qbase = (int64_t)blockIdx.x * KB_QB;
  const int d = a.d, k = a.k;

  for (int t = tid; t < KB_QB; t += 64 * NW) {
    const int64_t q = qbase + t;
    float tau = -__builtin_inff();
    if (q < a.m) {
      tau = a.best_d[q * k];
      for (int j = 1; j < k; ++j) tau = fmaxf(tau, a.best_d[q * k + j]);
    }
    tau_s[t] = tau;
    q_cnt[t] = 0;
  }
  __syncthreads();
  // A operands: row block rb, lane (r32, half): packed slots [half*KP/2, half*KP/2 + KP/2) of query
  // qbase + 64 w + 32 rb + r32, hi and lo parts.  The -thr slot (packed slot KP-1) is the upper
  // half-word of the last dword of the last piece of the half-1 lanes.
  u4x ahi[RBN][NQ], alo[RBN][NQ];
  float myqn[RBN];
  auto set_thr = [&](int rb, float tau) {  // (upper-half lanes) -thr of the row block's query into its operand slots
    unsigned hi, lo;
    neg_thr_split(myqn[rb], tau, hi, lo);
    if constexpr (K8) {
      alo[rb][0].y = hi | (lo << 16);  // T slots 2, 3
    } else {
      ahi[rb][NQ - 1].w = (ahi[rb][NQ - 1].w & 0xFFFFu) | (hi << 16);
      alo[rb][NQ - 1].w = (alo[rb][NQ - 1].w & 0xFFFFu) | (lo << 16);
    }
  };
#pragma unroll
  for (int rb = 0; rb < RBN; ++rb) {
    const int row = w * KB_QW + rb * 32 + r32;
    const int64_t q = qbase + row;
    const char* prow = reinterpret_cast<const char*>(a.packed_queries) + (q < a.m ? q : 0) * (int64_t)RB;
    if constexpr (K8) {  // ahi = [q_hi | q_lo], alo = [q_hi | T_q] (lower half | upper half of the wave)
      ahi[rb][0] = *reinterpret_cast<const u4x*>(prow + 16 * half);
      alo[rb][0] = *reinterpret_cast<const u4x*>(prow + 32 * half);
    } else {
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        ahi[rb][j] = *reinterpret_cast<const u4x*>(prow + half * KP + 16 * j);
        alo[rb][j] = *reinterpret_cast<const u4x*>(prow + 2 * KP + half * KP + 16 * j);
      }
    }
    myqn[rb] = q < a.m ? a.query_sqn[q] : 0.f;
    if (half == 1) set_thr(rb, tau_s[row]);
  }

  // more than two buffers: every wave issues the same number of transfers per tile (a pass index past the end repeats
  // the last pass: the same bytes to the same place), so "all but the NBUF - 2 newest tiles have landed" is one
  // immediate vmcnt; two buffers (the build's setting: three measured slower at every row length) wait for everything
  constexpr int IPW = (NPASS + NW - 1) / NW;
  auto issue_tile = [&](int buf, int64_t t0) {
#pragma unroll
    for (int it = 0; it < IPW; ++it) {
      if (NBUF == 2 && w + NW * it >= NPASS) break;
      const int p = min(w + NW * it, NPASS - 1);
      const int sigma = 64 * p + lane;
      const int row = sigma / SPR;
      const int c = SWZ ? (sigma - row * SPR) ^ ((row >> 2) & 3) : K8 ? sigma - row * SPR : min(sigma - row * SPR, SPR - 2);
      const int64_t grow = min(t0 + row, a.n - 1);
      lds_dma16(reinterpret_cast<const char*>(a.packed_train) + grow * (int64_t)RB + c * 16,
                lds_offset(tile0 + buf * TN * XSB) + p * 1024);
    }
  };
  // MGP_KNN_REGSTAGE: the next tile through registers instead -- plain 16-byte loads requested right behind the barrier,
  // written to the other buffer after this tile's matrix blocks.  An LDS-DMA instruction next to the other waves' matrix
  // instructions takes a few hundred cycles to ISSUE (tools/knn_timing.py: at d = 40 a wave spent as long issuing its
  // three or four per tile as in the tile's 36 matrix instructions); a load into registers does not
  constexpr bool RSTG = MGP_KNN_REGSTAGE && NBUF == 2;
  u4x stg[RSTG ? IPW : 1];
  auto load_tile = [&](int64_t t0) {
#pragma unroll
    for (int it = 0; it < IPW; ++it) {
      const int p = min(w + NW * it, NPASS - 1);
      const int sigma = 64 * p + lane;
      const int row = sigma / SPR;
      const int c = SWZ ? (sigma - row * SPR) ^ ((row >> 2) & 3) : K8 ? sigma - row * SPR : min(sigma - row * SPR, SPR - 2);
      const int64_t grow = min(t0 + row, a.n - 1);
      stg[RSTG ? it : 0] = *reinterpret_cast<const u4x*>(reinterpret_cast<const char*>(a.packed_train) + grow * (int64_t)RB + c * 16);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int it = 0; it < IPW; ++it)
      if (w + NW * it < NPASS)
        *reinterpret_cast<u4x*>(tile0 + buf * TN * XSB + (w + NW * it) * 1024 + lane * 16) = stg[RSTG ? it : 0];
  };
  const int64_t ntiles = (a.n - a.start + TN - 1) / TN;  // staggered walk, see knn_scan_kernel
#if MGP_KNN_STAGGER == 1
  const int64_t tile_off = ((int64_t)blockIdx.x * 7919) % ntiles;
#elif MGP_KNN_STAGGER > 1
  // a window of MGP_KNN_STAGGER tiles: neighbours in the grid start a tile apart, so the workgroups of an XCD are spread
  // over the memory channels but walk the same few hundred KB at any time -- the tile one of them brought in is in L2
  // (or the Infinity Cache) for the ones behind it
  const int64_t tile_off = (int64_t)(blockIdx.x % MGP_KNN_STAGGER) % ntiles;
#else
  const int64_t tile_off = 0;
#endif
  // (the tile of step j is (j + tile_off) mod ntiles, kept as two running indices: the 64-bit remainder is ~150 scalar
  // instructions, twice per tile -- most of the 19 % of a wave's cycles the tile issue took at KP = 16, round 5)
  auto tile_after = [&](int64_t t) { return t + 1 == ntiles ? (int64_t)0 : t + 1; };
  int64_t tcur = tile_off, tahead = tile_off;  // tile index of step tj / of the next tile to request
  if constexpr (RSTG) {
    load_tile(a.start + tahead * TN);
    tahead = tile_after(tahead);
    store_tile(0);
  } else {
    for (int t = 0; t < NBUF - 1; ++t)
      if (t < ntiles) {
        issue_tile(t, a.start + tahead * TN);
        tahead = tile_after(tahead);
      }
  }
  int buf = 0;
  int64_t next_drain = 0;
#if MGP_KNN_TIMING
  unsigned long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast_ = __builtin_readcyclecounter();
#endif
  for (int64_t tj = 0; tj < ntiles; ++tj, buf = buf + 1 == NBUF ? 0 : buf + 1, tcur = tile_after(tcur)) {
    const int64_t t0 = a.start + tcur * TN;
    // tile tj has landed: at most the NBUF - 2 tiles behind it are still on their way
    if constexpr (RSTG) {
      // (the barrier below orders this wave's LDS writes of the tile before everyone's reads)
    } else if (NBUF > 2 && tj + NBUF - 2 < ntiles) {
      constexpr int N = (NBUF - 2) * IPW;
      static_assert(N < 64, "vmcnt is six bits");
      __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (0x7 << 4) | (0xF << 8));
    } else {
      __builtin_amdgcn_s_waitcnt(0);
    }
    __syncthreads();  // ... everyone's share has, and nobody still reads the buffer tile tj - 1 was in
    MGP_KNN_T(0)
    const bool more = tj + NBUF - 1 < ntiles;
    if (more) {
      if constexpr (RSTG) load_tile(a.start + tahead * TN);
      else issue_tile(buf == 0 ? NBUF - 1 : buf - 1, a.start + tahead * TN);
      tahead = tile_after(tahead);
    }
    MGP_KNN_T(1)
    const char* tl = tile0 + buf * TN * XSB;

    const int nvalid = (int)(a.n - t0 < (int64_t)TN ? a.n - t0 : (int64_t)TN);  // rows of this tile that exist
    // one column block of 32 table rows against the wave's two row blocks of 32 queries: three chains per block
    // (rb0: the first of the two row blocks the call serves)
    auto blocks = [&](int ct, int rb0, f16x (&accr)[2]) {
      const char* xrow = tl + (ct * 32 + r32) * XSB;
      u4x bhi[NQ], blo[NQ];
      if constexpr (K8) {  // bhi = [x_hi | x_hi], blo = [x_lo | T_x]; 48-byte rows read conflict-free as they are
        bhi[0] = *reinterpret_cast<const u4x*>(xrow);
        blo[0] = *reinterpret_cast<const u4x*>(xrow + 16 + 16 * half);
      } else if constexpr (SWZ) {  // (NQ == 1; (row >> 2) & 3 == (r32 >> 2) & 3 for every column block)
        const int off = (half ^ ((r32 >> 2) & 3)) << 4;
        bhi[0] = *reinterpret_cast<const u4x*>(xrow + off);
        blo[0] = *reinterpret_cast<const u4x*>(xrow + (off ^ 32));
      } else {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
          bhi[j] = *reinterpret_cast<const u4x*>(xrow + half * KP + 16 * j);
          blo[j] = *reinterpret_cast<const u4x*>(xrow + 2 * KP + half * KP + 16 * j);
        }
      }
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2) {
        const int rb = rb0 + r2;
        f16x acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (K8) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8x, ahi[rb][0]),
                                                        __builtin_bit_cast(bf8x, bhi[0]), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8x, alo[rb][0]),
                                                        __builtin_bit_cast(bf8x, blo[0]), acc, 0, 0, 0);
        } else {
#pragma unroll
          for (int j = 0; j < NQ; ++j)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8x, ahi[rb][j]),
                                                          __builtin_bit_cast(bf8x, bhi[j]), acc, 0, 0, 0);
#pragma unroll
          for (int j = 0; j < NQ; ++j)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8x, ahi[rb][j]),
                                                          __builtin_bit_cast(bf8x, blo[j]), acc, 0, 0, 0);
#pragma unroll
          for (int j = 0; j < NQ; ++j)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8x, alo[rb][j]),
                                                          __builtin_bit_cast(bf8x, bhi[j]), acc, 0, 0, 0);
        }
        accr[r2] = acc;
      }
    };
    auto survivors = [&](int ct, int rb0, const f16x (&accr)[2]) {
      const int col = ct * 32 + r32;
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2) {
        const int rb = rb0 + r2;
        const f16x acc = accr[r2];
        // any accumulator positive?  as integers (a positive float is a positive int32): v_max3_i32
        // without the NaN-quieting canonicalisation fmaxf would put in front of every operand
        int mx = max(__float_as_int(acc[0]), __float_as_int(acc[1]));
#pragma unroll
        for (int v = 2; v < 16; v += 2) mx = max(mx, max(__float_as_int(acc[v]), __float_as_int(acc[v + 1])));
        if (mx > 0 && col < nvalid) {
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            if (acc[v] > 0.f) {
              const int qslot = w * KB_QW + rb * 32 + 8 * (v >> 2) + 4 * half + (v & 3);
              const int pos = atomicAdd(&q_cnt[qslot], 1);
              if (pos < KB_CAP) q_i[qslot * KB_CAP + pos] = (int)(t0 + col);
            }
          }
        }
      }
    };
    if constexpr (RBN == 4) {
      // two pairs of row blocks, each pair's vector work behind the other pair's matrix instructions
      f16x accA[2], accB[2];
#pragma unroll
      for (int ct = 0; ct < TN / 32; ++ct) {
        blocks(ct, 0, accA);
        if (ct > 0) survivors(ct - 1, 2, accB);
        blocks(ct, 2, accB);
        survivors(ct, 0, accA);
      }
      survivors(TN / 32 - 1, 2, accB);
    } else if constexpr (KP <= MGP_KNN_PIPE_KP) {
      // short rows (one matrix instruction per chain): the vector work on a column block's accumulators -- the max over
      // 16 registers, the survivor branch -- sits behind the NEXT block's matrix instructions instead of waiting for its
      // own (a second pair of accumulators: the registers are there up to KP = 32)
      f16x accp[2][2];
      blocks(0, 0, accp[0]);
#pragma unroll
      for (int ct = 1; ct < TN / 32; ++ct) {
        blocks(ct, 0, accp[ct & 1]);
        survivors(ct - 1, 0, accp[(ct - 1) & 1]);
      }
      survivors(TN / 32 - 1, 0, accp[(TN / 32 - 1) & 1]);
    } else {
#pragma unroll
      for (int ct = 0; ct < TN / 32; ++ct) {
        f16x accr[2];
        blocks(ct, 0, accr);
        survivors(ct, 0, accr);
      }
    }

    if constexpr (RSTG) {
      if (more) store_tile(buf ^ 1);  // (everyone passed this step's barrier: nobody reads that buffer any more)
    }
    MGP_KNN_T(2)
    // ---- drain: exact re-measurement of the survivors, then the k-best update ------------------
    // Not after every tile: a drain is a chain of dependent global loads (~2 us), and the four
    // waves of the workgroup meet at the next tile's barrier, so one wave draining stalls all;
    // batching makes the waves drain together.  A query collects ~ 64 k / rows_seen candidates per
    // tile, so the interval grows with the rows already seen (queues hold KB_CAP entries).
    if (tj < next_drain && tj + 1 < ntiles) continue;
    {
#if MGP_KNN_CADENCE
      // cadence in ROWS the lists have seen (the a.start rows they were made from + this workgroup's): a query collects
      // ~ k / rows_seen candidates per row, so an interval of rows_seen / MGP_KNN_CADENCE rows brings k / MGP_KNN_CADENCE
      // (<= 1) per query -- the queues hold KB_CAP = 16 -- and the whole walk ~ MGP_KNN_CADENCE ln(n / start) drains
      // instead of n / 1 024
      const int64_t rows = a.start + tj * TN;
      const int64_t every = rows / MGP_KNN_CADENCE < 64 ? 64 : rows / MGP_KNN_CADENCE > MGP_KNN_CADENCE_MAX ? MGP_KNN_CADENCE_MAX : rows / MGP_KNN_CADENCE;
#else
      // (cadence in ROWS seen by this workgroup: every 64 up to 4 096, then 256 / 512 / 1 024 -- whatever the tile size)
      const int64_t rows = tj * TN;
      const int every = rows < 4096 ? 64 : rows < 16384 ? 256 : rows < 65536 ? 512 : 1024;
#endif
      next_drain = tj + (every / TN > 1 ? every / TN : 1);
    }
    // Lane-per-query: lane r merges the queue of the wave's query r into that query's k-best list,
    // so all 64 queries drain at once and a drain costs a few memory latencies however many
    // queries are pending.  Round e handles every lane's e-th queue entry: exact squared distance
    // (d/4 independent 16-byte loads from each of the two rows), then one pass over the list that
    // finds its largest and second largest entry; a closer candidate replaces the largest.
    bool drained = false;
#pragma nounroll
    for (int qh = 0; qh < RBN / 2; ++qh) {  // 64 of the wave's queries at a time
    const int qslot = w * KB_QW + 64 * qh + lane;
    const int64_t q = qbase + qslot;
    int cnt = q < a.m ? q_cnt[qslot] : 0;
    if (__any(cnt > 0)) {
      drained = true;
      if (cnt > KB_CAP) {
        a.overflow[q] = 1;
        cnt = KB_CAP;
      }
      const int self = (cnt > 0 && a.self_idx) ? (int)a.self_idx[q] : -1;
      const float* qrow = a.queries + (q < a.m ? q : 0) * (int64_t)d;
      float* ld = a.best_d + (q < a.m ? q : 0) * (int64_t)k;
      int* li = a.best_i + (q < a.m ? q : 0) * (int64_t)k;
      float tau = tau_s[qslot];
      const int rounds = wave_max(cnt);
      for (int e = 0; e < rounds; ++e) {
        if (e < cnt) {
          const int ci = q_i[qslot * KB_CAP + e];
          if (ci != self) {
            const float* xrow = a.train + ci * (int64_t)d;
            float cd = 0.f;
            for (int c = 0; c < d; c += 4) {
              const f4x df = *reinterpret_cast<const f4x*>(qrow + c) - *reinterpret_cast<const f4x*>(xrow + c);
              cd += df.x * df.x + df.y * df.y + df.z * df.z + df.w * df.w;
            }
            if (cd < tau) {
              float m1 = -__builtin_inff(), m2 = -__builtin_inff();
              int at = 0;
              // the list in blocks of MGP_KNN_SCAN_BLOCK loads that are all in flight together: one at a time (k is a
              // run-time value, the loop stays rolled) an insertion was a chain of k memory latencies -- 50 x ~0.7 us
              // with every other wave of the workgroup waiting at the next tile's barrier, which was most of the
              // 10 M-row scan's time (round 5)
              for (int j0 = 0; j0 < k; j0 += MGP_KNN_SCAN_BLOCK) {
                float lv[MGP_KNN_SCAN_BLOCK];
#pragma unroll
                for (int u = 0; u < MGP_KNN_SCAN_BLOCK; ++u) lv[u] = ld[min(j0 + u, k - 1)];
#pragma unroll
                for (int u = 0; u < MGP_KNN_SCAN_BLOCK; ++u) {
                  const float v = j0 + u < k ? lv[u] : -__builtin_inff();
                  if (v > m1) {
                    m2 = m1;
                    m1 = v;
                    at = j0 + u;
                  } else if (v > m2) {
                    m2 = v;
                  }
                }
              }
              ld[at] = cd;
              li[at] = ci;
              tau = fmaxf(m2, cd);
            }
          }
        }
      }
      if (cnt > 0) {
        tau_s[qslot] = tau;
        q_cnt[qslot] = 0;
      }
    }
    }
    if (drained) {  // (wave-uniform)
      __builtin_amdgcn_s_waitcnt(0);
      if (half == 1) {
#pragma unroll
        for (int rb = 0; rb < RBN; ++rb) set_thr(rb, tau_s[w * KB_QW + rb * 32 + r32]);
      }
    }
    MGP_KNN_T(3)
  }
#if MGP_KNN_TIMING
  if (lane == 0)
    for (int t = 0; t < 8; ++t) atomicAdd(&g_knn_timing[t], tacc_[t]);
#endif
}

#ifndef MGP_KNN_TN16
#define MGP_KNN_TN16 128
#endif
#ifndef MGP_KNN_TNL
#define MGP_KNN_TNL KNN_TN  // rows per staged tile of the longer packed rows (128 = two workgroups per CU at KP = 48: 586 against 318 ms at d = 40, 290 against 241 at d = 24)
#endif
#ifndef MGP_KNN_NWL
#define MGP_KNN_NWL 4  // waves per workgroup, longer packed rows (6 = two workgroups per CU: 452 against 329 ms at d = 40)
#endif
#ifndef MGP_KNN_NW16
#define MGP_KNN_NW16 4
#endif
#ifndef MGP_KNN_NBUF16
#define MGP_KNN_NBUF16 2
#endif
#ifndef MGP_KNN_RB4_MIN_QUERIES
#define MGP_KNN_RB4_MIN_QUERIES (1 << 20)
#endif
template <int KP, int RBN = 2>
static int launch_knn_packed_kp(const KnnPackedArgs& a, hipStream_t stream) {
  constexpr int TN = KP <= 16 ? MGP_KNN_TN16 : MGP_KNN_TNL;
  constexpr int NW = KP <= 16 ? MGP_KNN_NW16 : MGP_KNN_NWL, NBUF = KP <= 16 ? MGP_KNN_NBUF16 : 2;
  constexpr int KB_QB = KB_RB * RBN * NW;
  const int64_t grid = (a.m + KB_QB - 1) / KB_QB;
  constexpr int XSB = KP == 8 ? 48 : 4 * KP + (MGP_KNN_SWIZZLE && KP == 16 ? 0 : 16);  // staged row (see the kernel)
  const size_t lds = NBUF * TN * XSB + (KB_QB * KB_CAP + 2 * KB_QB) * sizeof(float);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_scan_bf16x3_kernel<KP, TN, NW, NBUF, RBN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return -(1000 + (int)e);
  }
  hipLaunchKernelGGL((knn_scan_bf16x3_kernel<KP, TN, NW, NBUF, RBN>), dim3((unsigned)grid), dim3(64 * NW), lds, stream, a);
  MGP_HIP_CHECK_LAUNCH();
  return MGP_OK;
}

int launch_knn_scan_packed(const KnnPackedArgs& a, hipStream_t stream) {
  if (a.k < 1 || a.k > 64 || a.d < 4 || a.d % 4 != 0 || a.d > 64) return MGP_EUNSUPPORTED;
  const uintptr_t al = (uintptr_t)a.train | (uintptr_t)a.queries | (uintptr_t)a.packed_train |
                       (uintptr_t)a.packed_queries;
  if (al % 16 != 0) return MGP_EUNSUPPORTED;
  if (a.n >= (int64_t)1 << 31) return MGP_EUNSUPPORTED;
  if (a.layout == 1) {
    if (a.d > 8) return MGP_EUNSUPPORTED;
    // four row blocks per wave (512 queries per workgroup, three workgroups per CU): 5-6 % faster once the grid is many
    // rounds of the chip's 768 slots (2 M queries x 10 M rows: 1.78 -> 1.68 s), slower below (400 k queries = 782
    // workgroups: 387 -> 463 ms, a second round for fourteen of them)
    const char* e = getenv("MUYGPYS_HIP_KNN_RB4_MIN");  // (tests: 0 forces the four-block kernel)
    const int64_t rb4_min = e ? atoll(e) : (int64_t)MGP_KNN_RB4_MIN_QUERIES;
    return a.m >= rb4_min ? launch_knn_packed_kp<8, 4>(a, stream) : launch_knn_packed_kp<8, 2>(a, stream);
  }
  switch ((a.d + 2 + 15) / 16 * 16) {  // features + the two threshold slots
    case 16: return launch_knn_packed_kp<16>(a, stream);
    case 32: return launch_knn_packed_kp<32>(a, stream);
    case 48: return launch_knn_packed_kp<48>(a, stream);
    case 64: return launch_knn_packed_kp<64>(a, stream);
    default: return launch_knn_packed_kp<80>(a, stream);
  }
}

}  // namespace mgp

// The LOOCV partial sums of a shard as a FIXED tree over the neighbourhood index -- and the same tree walked either
// by the fused kernel itself (one launch per objective evaluation: the workgroup that completes a block reduces it)
// or by two small kernels behind a fused kernel that cannot (mgp_tensor_ops.hip).  Both walks execute the functions
// below on the same values in the same order, so their sums are equal bit for bit, whatever workgroup finished what
// when (tests/test_gpu_properties.py).
//
//     level 1   block j1 = neighbourhoods 64 j1 .. 64 j1 + 63: one neighbourhood per lane, the six terms
//               [r^2/v, log v, r^2, 1, pseudo-Huber(r), y^T K^-1 y], r = mean - y(batch row)
//               (reference: _src/optimize/loss/numpy.py:22-72, _src/optimize/scale/numpy.py:9-15), butterfly sum
//     level 2   block j2 = level-1 partials 64 j2 .. 64 j2 + 63: one partial per lane, butterfly sum
//     level 3   lane l sums the level-2 partials l, l + 64, ... in order, butterfly sum -> out[6]
//
// In the fused kernel a level is entered by whoever arrives last (an agent-scope ticket per block): hand-off by
// write-through (sc1) stores, a drained vmcnt and sc1 loads -- no fence, no spin, nobody ever waits for another
// workgroup (MI355X_MICROARCH.md, "inter-workgroup visibility": ticket form, the adder whose add came last reads).
// Replaces the reference's three host all-reduce inputs per evaluation (loss/mpi.py:57, scale/mpi.py:35-36) with six
// doubles that stay on the device.
//
// Scratch (caller's, mgp_loocv_scratch_bytes(b)): [control 2 KiB | cnt1 | cnt2 | part1 | part2 | deferred].  Control and the
// counters must be ZERO when a call starts; every call leaves them zero (the last arriver of a block resets its
// counter, the last workgroup out resets the task queue).
#pragma once

#include "mgp_device.h"

namespace mgp {

struct LoocvTree {
  double* out = nullptr;        // [6] the shard's sums; nullptr: no tree (plain prediction launch)
  unsigned* ctrl = nullptr;     // control block: word 32 x = dequeue head of XCD x (x < 8), word 256 = workgroups out,
                                //                word 288 = level-2 blocks done, word 320 = deferred level-1 blocks
  unsigned* cnt1 = nullptr;     // [nb1] neighbourhoods arrived per level-1 block
  unsigned* cnt2 = nullptr;     // [nb2] level-1 blocks arrived per level-2 block
  double* part1 = nullptr;      // [nb1][6]
  double* part2 = nullptr;      // [nb2][6]
  unsigned* deferred = nullptr; // [nb1] level-1 blocks whose completer's list was full
  const char* resp = nullptr;   // response of table row i at resp + i * resp_stride (the tensor, or a prepared table)
  int64_t resp_stride = 0;
  double huber_delta = 1.5;
};

constexpr int kTreeCtrlBytes = 2048;
constexpr int kTreeWordOut = 256, kTreeWordL3 = 288, kTreeWordDeferred = 320;
// completed level-1 blocks a workgroup keeps for the end of its task loop (LDS, behind everything else); one more goes
// to the scratch's deferred list, which the last workgroup out reduces
constexpr int kTreeListCap = 250;
constexpr int kTreeListBytes = 4 * (kTreeListCap + 6);
__host__ __device__ constexpr int64_t tree_nb1(int64_t b) { return (b + 63) >> 6; }
__host__ __device__ constexpr int64_t tree_nb2(int64_t b) { return (tree_nb1(b) + 63) >> 6; }
// byte offsets of the scratch regions (each a multiple of 128)
__host__ __device__ constexpr int64_t tree_align(int64_t x) { return (x + 127) & ~(int64_t)127; }
__host__ __device__ constexpr int64_t tree_off_cnt1(int64_t) { return kTreeCtrlBytes; }
__host__ __device__ constexpr int64_t tree_off_cnt2(int64_t b) { return tree_off_cnt1(b) + tree_align(4 * tree_nb1(b)); }
__host__ __device__ constexpr int64_t tree_zero_bytes(int64_t b) { return tree_off_cnt2(b) + tree_align(4 * tree_nb2(b)); }
__host__ __device__ constexpr int64_t tree_off_part1(int64_t b) { return tree_zero_bytes(b); }
__host__ __device__ constexpr int64_t tree_off_part2(int64_t b) { return tree_off_part1(b) + tree_align(48 * tree_nb1(b)); }
__host__ __device__ constexpr int64_t tree_off_deferred(int64_t b) { return tree_off_part2(b) + tree_align(48 * tree_nb2(b)); }
__host__ __device__ constexpr int64_t tree_scratch_bytes(int64_t b) { return tree_off_deferred(b) + tree_align(4 * tree_nb1(b)); }

// write-through (sc1) accesses of words other workgroups read or wrote in this launch
template <typename U>
__device__ __forceinline__ void st_agent(U* p, U v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename U>
__device__ __forceinline__ U ld_agent(const U* p) {
  return __hip_atomic_load(const_cast<U*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent_f(float* p, float v) { st_agent(reinterpret_cast<unsigned*>(p), __float_as_uint(v)); }
__device__ __forceinline__ void st_agent_f(double* p, double v) {
  st_agent(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v));
}
__device__ __forceinline__ float ld_agent_f(const float* p) { return __uint_as_float(ld_agent(reinterpret_cast<const unsigned*>(p))); }
__device__ __forceinline__ double ld_agent_f(const double* p) {
  return __longlong_as_double((long long)ld_agent(reinterpret_cast<const unsigned long long*>(p)));
}
// every store this wave has issued has left the CU (what a ticket add must come behind)
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// six sums over the wave in a fixed order (butterfly: every lane ends with the same bits)
__device__ __forceinline__ void tree_wave_sum(double (&t)[6]) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int i = 0; i < 6; ++i) t[i] += __shfl_xor(t[i], off, 64);
  }
}

// the six terms of one neighbourhood (no contraction: both walks must round alike)
__device__ __forceinline__ void tree_terms(double mean, double var, double yky, double y, double hd, double (&t)[6]) {
#pragma clang fp contract(off)
  const double r = mean - y;
  const double r2 = r * r;
  const double rh = r / hd;
  t[0] = r2 / var;
  t[1] = ::log(var);
  t[2] = r2;
  t[3] = 1.0;
  t[4] = hd * hd * (::sqrt(1.0 + rh * rh) - 1.0);
  t[5] = yky;
}

// level 1: block j1 of a batch of b neighbourhoods.  AGENT: mean / var / ykinvy were written by other workgroups of
// this launch (sc1 loads); otherwise by an earlier launch (plain loads).  Every lane returns the block's sums.
template <typename T, bool AGENT>
__device__ __forceinline__ void tree_level1(const LoocvTree& tr, const T* mean, const T* var, const T* yk, const int64_t* batch_idx,
                                            int64_t b, int64_t j1, int lane, double (&t)[6]) {
  const int64_t n = (j1 << 6) + lane;
#pragma unroll
  for (int i = 0; i < 6; ++i) t[i] = 0.0;
  if (n < b) {
    const int64_t row = batch_idx ? batch_idx[n] : n;
    const double y = (double)*reinterpret_cast<const T*>(tr.resp + row * tr.resp_stride);
    double m, v, q;
    if constexpr (AGENT) {
      m = (double)ld_agent_f(mean + n), v = (double)ld_agent_f(var + n), q = (double)ld_agent_f(yk + n);
    } else {
      m = (double)mean[n], v = (double)var[n], q = (double)yk[n];
    }
    tree_terms(m, v, q, y, tr.huber_delta, t);
  }
  tree_wave_sum(t);
}
// level 2: block j2 of nb1 level-1 partials
template <bool AGENT>
__device__ __forceinline__ void tree_level2(const LoocvTree& tr, int64_t nb1, int64_t j2, int lane, double (&t)[6]) {
  const int64_t j1 = (j2 << 6) + lane;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    t[i] = 0.0;
    if (j1 < nb1) t[i] = AGENT ? ld_agent_f(tr.part1 + 6 * j1 + i) : tr.part1[6 * j1 + i];
  }
  tree_wave_sum(t);
}
// level 3: all nb2 level-2 partials
template <bool AGENT>
__device__ __forceinline__ void tree_level3(const LoocvTree& tr, int64_t nb2, int lane, double (&t)[6]) {
#pragma unroll
  for (int i = 0; i < 6; ++i) t[i] = 0.0;
  for (int64_t j2 = lane; j2 < nb2; j2 += 64) {
#pragma unroll
    for (int i = 0; i < 6; ++i) t[i] += AGENT ? ld_agent_f(tr.part2 + 6 * j2 + i) : tr.part2[6 * j2 + i];
  }
  tree_wave_sum(t);
}

// In the fused kernel: level-1 block j1 is complete -- this wave drew the ticket that says so (every neighbourhood of the
// block stored write-through and drained by its storing wave before that wave's ticket add).  Reduce it and walk up
// as far as this wave's tickets are the last ones.  One wave per workgroup; uniform control flow.  Called after the
// workgroup's task loop (the blocks it completed wait in a short LDS list), so that none of this is live inside it.
template <typename T>
__device__ __forceinline__ void tree_reduce_block(const LoocvTree& tr, const T* mean, const T* var, const T* yk,
                                                  const int64_t* batch_idx, int64_t b, int64_t j1, int lane) {
  double t[6];
  tree_level1<T, true>(tr, mean, var, yk, batch_idx, b, j1, lane, t);
  if (lane < 6) {
    double mine = t[0];
#pragma unroll
    for (int i = 1; i < 6; ++i) mine = lane == i ? t[i] : mine;
    st_agent_f(tr.part1 + 6 * j1 + lane, mine);
  }
  if (lane == 0) st_agent(tr.cnt1 + j1, 0u);  // (the counter is this call's no longer: left zero for the next)
  drain_stores();
  const int64_t nb1 = tree_nb1(b), nb2 = tree_nb2(b), j2 = j1 >> 6;
  unsigned old2 = 0;
  if (lane == 0) old2 = __hip_atomic_fetch_add(tr.cnt2 + j2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old2 = __builtin_amdgcn_readfirstlane(old2);
  const int64_t left2 = nb1 - (j2 << 6);
  if (old2 + 1u != (left2 < 64 ? (unsigned)left2 : 64u)) return;
  tree_level2<true>(tr, nb1, j2, lane, t);
  if (lane < 6) {
    double mine = t[0];
#pragma unroll
    for (int i = 1; i < 6; ++i) mine = lane == i ? t[i] : mine;
    st_agent_f(tr.part2 + 6 * j2 + lane, mine);
  }
  if (lane == 0) st_agent(tr.cnt2 + j2, 0u);  // (the counter is this call's no longer: left zero for the next)
  drain_stores();
  unsigned old3 = 0;
  if (lane == 0) old3 = __hip_atomic_fetch_add(tr.ctrl + kTreeWordL3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old3 = __builtin_amdgcn_readfirstlane(old3);
  if ((int64_t)old3 + 1 != nb2) return;
  tree_level3<true>(tr, nb2, lane, t);
  if (lane < 6) {
    double mine = t[0];
#pragma unroll
    for (int i = 1; i < 6; ++i) mine = lane == i ? t[i] : mine;
    st_agent_f(tr.out + lane, mine);
  }
  if (lane == 0) st_agent(tr.ctrl + kTreeWordL3, 0u);
}

}  // namespace mgp

// The LOOCV partial sums of a shard as a FIXED tree -- walked by the fused wave kernel itself (one launch per objective
// evaluation) or by three small kernels behind a fused kernel of another family (mgp_tensor_ops.hip).  Both walks run
// the functions below on the same values in the same order: equal sums, bit for bit (tests/test_gpu_loocv_tree.py).
//
//   level 1   leaf w = what workgroup w of a launch of `grid` persistent workgroups, `nh` neighbourhoods per task,
//             evaluates: tasks xcd * ceil(ntasks / 8) + (w >> 3) + n * (grid / 8), xcd = w & 7 (the wave kernels' task
//             order, mgp_fused_wave_kernel.h); its p-th neighbourhood (task order, then position in the task) belongs to
//             lane p & 63, which adds the six terms [r^2/v, log v, r^2, 1, pseudo-Huber(r), y^T K^-1 y], r = mean - y(batch
//             row) (reference: _src/optimize/loss/numpy.py:22-72, _src/optimize/scale/numpy.py:9-15) of its neighbourhoods
//             in order; then a butterfly sum over the lanes.
//   level 2   block j2 = leaves 64 j2 .. 64 j2 + 63: one leaf per lane, butterfly sum
//   level 3   lane l sums the level-2 partials l, l + 64, ... in order, butterfly sum -> out[6]
//
// In the fused kernel level 1 is the workgroup's OWN outputs, read back when it has run out of tasks: nothing of the tree
// lives in the task loop, no output is handed from one workgroup to another.  (Round 5 first built a tree over the
// neighbourhood index -- blocks of 64 neighbourhoods, an arrival ticket per task, the completer reducing the block -- and
// measured it out: with the static task stride the workgroups of a block run in lockstep and the same one arrives
// last every time, 163 block reductions in a row on one wave at 1 M neighbourhoods, +380 us; with tasks drawn from
// dequeue heads instead the completers spread, but the per-task tickets, write-through stores and per-block butterflies
// still cost 4-5 %, and at ten folded pairs per workgroup whole pairs quantise worse than 20-or-21 tasks do.)  Levels
// 2 and 3 are entered by whoever arrives last (an agent-scope ticket per block): hand-off by write-through (sc1)
// stores, a drained vmcnt and sc1 loads -- no fence, no spin, nobody ever waits for another workgroup
// (MI355X_MICROARCH.md, "inter-workgroup visibility": ticket form, the adder whose add came last reads).
// Replaces the reference's three host all-reduce inputs per evaluation (loss/mpi.py:57, scale/mpi.py:35-36) with six
// doubles that stay on the device -- or go straight to pinned host memory (tree_result).
//
// The sums depend on (grid, nh) in their last bits: the same batch on the same device (the same kernel, the same grid)
// gives the same bits, which is what sharded evaluations need (tests/test_gpu_properties.py).
//
// Scratch (caller's, mgp_loocv_scratch_bytes()): [control 128 B | cnt2 | part1 | part2].  Control and the counters must
// be ZERO when a call starts; every call leaves them zero (the last arriver of a block resets its counter).
#pragma once

#include "mgp_device.h"

namespace mgp {

constexpr int kTreeMaxLeaves = 16384;  // persistent workgroups of a launch (256 CUs x 16 resident waves, with room)
constexpr int kTreeCanonGrid = 2048;   // the leaves of a walk behind a kernel that has no persistent grid of its own

struct LoocvTree {
  double* out = nullptr;        // [6] the shard's sums (device memory, or mapped host memory); nullptr: no tree
  unsigned* ctrl = nullptr;     // control block: word 0 = level-2 blocks done
  unsigned* cnt2 = nullptr;     // [leaves / 64] leaves arrived per level-2 block
  double* part1 = nullptr;      // [leaves][6]
  double* part2 = nullptr;      // [leaves / 64][6]
  const char* resp = nullptr;   // response of table row i at resp + i * resp_stride (the tensor, or a prepared table)
  int64_t resp_stride = 0;
  double huber_delta = 1.5;
  int grid = 0, nh = 0;         // the leaves: persistent workgroups of the launch, neighbourhoods per task
  int mode = 0;                 // kTreeTickets / kTreeFenced (how the in-kernel walk hands sums over) / kTreeThreeLaunch
};

// How one workgroup's sums reach the workgroup that adds them up (MUYGPYS_HIP_LOOCV_TREE, mgp_loocv_tree_mode_set):
//   tickets       write-through stores, a drained vmcnt, a RELAXED agent-scope ticket, sc1 loads -- no fence (the default:
//                 the form the microarchitecture guide gives for "the adder whose add came last reads")
//   fenced        the textbook form: release fence, ACQ_REL ticket, acquire fence -- what the memory model guarantees on
//                 any driver / compiler; the same sums bit for bit, a few hundred cycles more per workgroup
//   three_launch  the fused kernel does not walk the tree at all; three small kernels do, behind it, over the SAME
//                 leaves (grid, nh) -- kernel boundaries are the only synchronisation; again the same bits
constexpr int kTreeTickets = 0, kTreeFenced = 1, kTreeThreeLaunch = 2;

constexpr int kTreeCtrlBytes = 128;
constexpr int kTreeWordL3 = 0;
constexpr int64_t tree_nb2(int64_t leaves) { return (leaves + 63) >> 6; }
// byte offsets of the scratch regions (each a multiple of 128), sized for kTreeMaxLeaves
constexpr int64_t tree_off_cnt2() { return kTreeCtrlBytes; }
constexpr int64_t tree_zero_bytes() { return tree_off_cnt2() + 4 * tree_nb2(kTreeMaxLeaves); }
constexpr int64_t tree_off_part1() { return tree_zero_bytes(); }
constexpr int64_t tree_off_part2() { return tree_off_part1() + 48 * (int64_t)kTreeMaxLeaves; }
constexpr int64_t tree_scratch_bytes() { return tree_off_part2() + 48 * tree_nb2(kTreeMaxLeaves); }

// write-through (sc1) accesses of words other workgroups read or wrote in this launch
template <typename U>
__device__ __forceinline__ void st_agent(U* p, U v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename U>
__device__ __forceinline__ U ld_agent(const U* p) {
  return __hip_atomic_load(const_cast<U*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent_f(double* p, double v) {
  st_agent(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v));
}
__device__ __forceinline__ double ld_agent_f(const double* p) {
  return __longlong_as_double((long long)ld_agent(reinterpret_cast<const unsigned long long*>(p)));
}
__device__ __forceinline__ float ld_agent_f(const float* p) { return __uint_as_float(ld_agent(reinterpret_cast<const unsigned*>(p))); }
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

// the six terms of one neighbourhood, added to the lane's sums (no contraction: both walks must round alike)
__device__ __forceinline__ void tree_add_terms(double mean, double var, double yky, double y, double hd, double (&t)[6]) {
#pragma clang fp contract(off)
  const double r = mean - y;
  const double r2 = r * r;
  const double rh = r / hd;
  t[0] += r2 / var;
  t[1] += ::log(var);
  t[2] += r2;
  t[3] += 1.0;
  t[4] += hd * hd * (::sqrt(1.0 + rh * rh) - 1.0);
  t[5] += yky;
}

// level 1: leaf w (see above).  The outputs are the calling workgroup's own (the fused kernel, after its task loop: read
// past the vector L1, from the L2 its drained stores went to) or an earlier launch's (the kernel walk).  Four rounds of
// loads -- a dependent index -> response chain and three output reads each -- are in flight together.  Every lane
// returns the leaf's sums.
template <typename T>
__device__ __forceinline__ void tree_level1(const LoocvTree& tr, const T* mean, const T* var, const T* yk, const int64_t* batch_idx,
                                            int64_t b, int w, int lane, double (&t)[6]) {
  const int nh = tr.nh, sh = nh == 4 ? 2 : (nh == 2 ? 1 : 0);  // (neighbourhoods per task: 1, 2 or 4)
  const int64_t ntasks = (b + nh - 1) >> sh, per_xcd = (ntasks + 7) / 8, step = tr.grid >> 3;
  const int xcd = w & 7;
  const int64_t first = xcd * per_xcd + (w >> 3), hi = (xcd + 1) * per_xcd, end = hi < ntasks ? hi : ntasks;
  const int64_t count = first < end ? ((end - first + step - 1) / step) << sh : 0;  // neighbourhood slots of the leaf
#pragma unroll
  for (int i = 0; i < 6; ++i) t[i] = 0.0;
  constexpr int U = 4;
  for (int64_t p0 = lane; p0 < count; p0 += 64 * U) {
    T m[U], v[U], q[U], y[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = p0 + 64 * u;
      const int64_t nb = ((first + (p >> sh) * step) << sh) + (p & (nh - 1));
      ok[u] = p < count && nb < b;
      m[u] = v[u] = q[u] = y[u] = T(0);
      if (ok[u]) {
        const int64_t row = batch_idx ? batch_idx[nb] : nb;
        m[u] = ld_agent_f(mean + nb), v[u] = ld_agent_f(var + nb), q[u] = ld_agent_f(yk + nb);
        y[u] = *reinterpret_cast<const T*>(tr.resp + row * tr.resp_stride);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (ok[u]) tree_add_terms((double)m[u], (double)v[u], (double)q[u], (double)y[u], tr.huber_delta, t);
  }
  tree_wave_sum(t);
}
// level 2: block j2 of the leaves' partials
template <bool AGENT>
__device__ __forceinline__ void tree_level2(const LoocvTree& tr, int j2, int lane, double (&t)[6]) {
  const int w = (j2 << 6) + lane;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    t[i] = 0.0;
    if (w < tr.grid) t[i] = AGENT ? ld_agent_f(tr.part1 + 6 * w + i) : tr.part1[6 * w + i];
  }
  tree_wave_sum(t);
}
// level 3: all level-2 partials
template <bool AGENT>
__device__ __forceinline__ void tree_level3(const LoocvTree& tr, int lane, double (&t)[6]) {
  const int nb2 = (int)tree_nb2(tr.grid);
#pragma unroll
  for (int i = 0; i < 6; ++i) t[i] = 0.0;
  for (int j2 = lane; j2 < nb2; j2 += 64) {
#pragma unroll
    for (int i = 0; i < 6; ++i) t[i] += AGENT ? ld_agent_f(tr.part2 + 6 * j2 + i) : tr.part2[6 * j2 + i];
  }
  tree_wave_sum(t);
}

// six sums another workgroup will read in this launch (lane 0, six write-through stores: a `lane == i ? t[i]` select
// chain is turned into an indexed load of t[] from scratch memory)
__device__ __forceinline__ void tree_publish(double* dst, const double (&t)[6], int lane) {
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) st_agent_f(dst + i, t[i]);
  }
}
// the shard's six sums, for whoever reads them -- a later kernel, or THE HOST: `out` may be pinned host memory mapped
// into the device's address space, so that an optimiser's loop learns the value of an evaluation by polling instead of
// a stream synchronisation and a copy (system-scope stores).  The count (element 3, = b > 0) goes LAST, behind a
// drain of the others: a host that zeroed it before the launch and sees b has all six.
__device__ __forceinline__ void tree_result(double* out, const double (&t)[6], int lane) {
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
      if (i != 3) __hip_atomic_store(out + i, t[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    drain_stores();
    __hip_atomic_store(out + 3, t[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// In the fused kernel, when workgroup w has run out of tasks (its output stores drained): its leaf, and up the tree as far
// as this wave's tickets are the last ones.  One wave per workgroup; uniform control flow.
template <bool FENCED>
__device__ __forceinline__ void tree_handoff() {  // this wave's published sums are visible before its ticket is
  if constexpr (FENCED) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  else drain_stores();
}
template <bool FENCED>
__device__ __forceinline__ unsigned tree_ticket(unsigned* counter, int lane) {
  unsigned old = 0;
  if (lane == 0)
    old = FENCED ? __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
                 : __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __builtin_amdgcn_readfirstlane(old);
}
template <typename T, bool FENCED>
__device__ __forceinline__ void tree_climb(const LoocvTree& tr, int w, int lane, double (&t)[6]) {
  tree_publish(tr.part1 + 6 * w, t, lane);
  tree_handoff<FENCED>();
  const int j2 = w >> 6, left2 = tr.grid - (j2 << 6);
  const unsigned old2 = tree_ticket<FENCED>(tr.cnt2 + j2, lane);
  if (old2 + 1u != (unsigned)(left2 < 64 ? left2 : 64)) return;
  if constexpr (FENCED) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  tree_level2<true>(tr, j2, lane, t);
  tree_publish(tr.part2 + 6 * j2, t, lane);
  if (lane == 0) st_agent(tr.cnt2 + j2, 0u);  // (the counter is this call's no longer: left zero for the next)
  tree_handoff<FENCED>();
  const unsigned old3 = tree_ticket<FENCED>(tr.ctrl + kTreeWordL3, lane);
  if ((int64_t)old3 + 1 != tree_nb2(tr.grid)) return;
  if constexpr (FENCED) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  tree_level3<true>(tr, lane, t);  // the last level-2 block of the launch: the shard's sums
  tree_result(tr.out, t, lane);
  if (lane == 0) st_agent(tr.ctrl + kTreeWordL3, 0u);
}

// In the fused kernel, when workgroup w has run out of tasks (its output stores drained): its leaf, and up the tree as far
// as this wave's tickets are the last ones.  One wave per workgroup; uniform control flow.
template <typename T>
__device__ __forceinline__ void tree_leaf_done(const LoocvTree& tr, const T* mean, const T* var, const T* yk,
                                               const int64_t* batch_idx, int64_t b, int w, int lane) {
  double t[6];
  tree_level1<T>(tr, mean, var, yk, batch_idx, b, w, lane, t);
  if (tr.mode == kTreeFenced) tree_climb<T, true>(tr, w, lane, t);
  else tree_climb<T, false>(tr, w, lane, t);
}

}  // namespace mgp

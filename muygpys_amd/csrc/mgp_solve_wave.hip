// Register-resident solve on MATERIALISED tensors: mgp_solve_* for k + 1 + R <= 64 slots.
//
// The per-function form of SURVEY.md sec. 8a rows S1-S3 (_muygps_posterior_mean,
// _muygps_diagonal_variance, _analytic_scale_optim_unnormalized: _src/gp/muygps/numpy.py:17-67,
// _src/optimize/scale/numpy.py:9-15) takes Kin (b, k, k) -- nugget already added by the caller --,
// Kcross (b, k) and the gathered responses (b, k, R).  It used to go through the LDS workgroup kernel
// (mgp_generic.hip); this is the elimination and output phase of mgp_fused_wave.hip fed from memory
// instead: slot layout, row-per-lane registers, column broadcast through a 64-entry LDS buffer and
// the Schur-block outputs are the same, so are the results.
//
//     slot 0 .. k-1      row i of Kin (lower triangle used)
//     slot k .. q-1      padding (identity rows)          q = NP-1-R
//     slot q             Kcross, kout on the diagonal
//     slot q+1 .. NP-1   the R response columns
//
// One wavefront (= one workgroup) owns 64 / NP neighbourhoods at a time; a lane reads its own row
// (a neighbourhood's Kin block is contiguous, so the 64 rows of a wave are one 64 k s-byte span).
// The coefficient output (K^-1 Y) stays with the LDS workgroup kernel.
#include "mgp_wave_common.h"

namespace mgp {

template <typename T, int NP>
__global__ __launch_bounds__(64, (sizeof(T) == 4 ? (NP == 32 ? 4 : 2) : 2)) void solve_wave_kernel(SolveArgs a, int q,
                                                                                                  int vec) {
  constexpr int NH = 64 / NP;
  constexpr int E = v16<T>::N;
  using V = typename v16<T>::type;
  __shared__ __attribute__((aligned(16))) T colbuf[64];
  const int k = a.k, R = a.R;
  const T* Kin = static_cast<const T*>(a.Kin);
  const T* Kcross = static_cast<const T*>(a.Kcross);
  const T* Y = static_cast<const T*>(a.Y);
  const int64_t ntasks = (a.b + NH - 1) / NH;

  // The lane's row of one task (rows of the next task are requested before the elimination of the
  // current one starts: their latency hides behind it).  vec: 2 = 16-byte loads (k s % 16 == 0),
  // 1 = 8-byte loads (k s % 8 == 0), 0 = element loads; Kcross / Y rows are strided: element loads.
  typedef T V8 __attribute__((ext_vector_type(8 / sizeof(T) > 1 ? 8 / sizeof(T) : 1)));
  auto load_row = [&](int64_t task, auto& A) {
    const int lane = threadIdx.x;
    const int h = NH == 1 ? 0 : lane / NP;
    const int i = lane & (NP - 1);
    const int64_t nb = task * NH + h;
    const int64_t nbb = nb < a.b ? nb : task * NH;  // odd tail: replay the first neighbourhood
#pragma unroll
    for (int c4 = 0; c4 < NP / E; ++c4) A[c4] = V(0);
    if (i < k) {
      const T* src = Kin + (nbb * k + i) * (int64_t)k;
      if (vec == 2) {
#pragma unroll
        for (int c4 = 0; c4 < NP / E; ++c4)
          if (c4 * E < k) A[c4] = *reinterpret_cast<const V*>(src + c4 * E);  // k % E == 0: whole groups
      } else if (sizeof(T) == 4 && vec == 1) {
#pragma unroll
        for (int c2 = 0; c2 < NP / 2; ++c2)
          if (c2 * 2 < k) {  // k even: whole pairs
            const V8 p = *reinterpret_cast<const V8*>(src + c2 * 2);
            A[(c2 * 2) / E][(c2 * 2) % E] = p[0];
            A[(c2 * 2 + 1) / E][(c2 * 2 + 1) % E] = p[sizeof(T) == 4 ? 1 : 0];
          }
      } else {
#pragma unroll
        for (int c4 = 0; c4 < NP / E; ++c4)
#pragma unroll
          for (int e = 0; e < E; ++e)
            if (c4 * E + e < k) A[c4][e] = src[c4 * E + e];
      }
    } else if (i == q) {
      if (Kcross != nullptr) {
        const T* src = Kcross + nbb * k;
#pragma unroll
        for (int c4 = 0; c4 < NP / E; ++c4)
#pragma unroll
          for (int e = 0; e < E; ++e)
            if (c4 * E + e < k) A[c4][e] = src[c4 * E + e];
      }
    } else if (i > q && i - q - 1 < R) {
      const T* src = Y + nbb * k * (int64_t)R + (i - q - 1);
#pragma unroll
      for (int c4 = 0; c4 < NP / E; ++c4)
#pragma unroll
        for (int e = 0; e < E; ++e)
          if (c4 * E + e < k) A[c4][e] = src[(c4 * E + e) * (int64_t)R];
    }
  };

  constexpr bool PREFETCH = !(sizeof(T) == 8 && NP == 64);  // a second 128-register row does not fit
  V Anext[PREFETCH ? NP / E : 1];
  if constexpr (PREFETCH)
    if ((int64_t)blockIdx.x < ntasks) load_row(blockIdx.x, Anext);
  for (int64_t task = blockIdx.x; task < ntasks; task += gridDim.x) {
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));
    const int h = NH == 1 ? 0 : lane / NP;
    const int i = lane & (NP - 1);
    T* colh = colbuf + h * NP;
    const int64_t nb = task * NH + h;
    const bool live = nb < a.b;

    V A[NP / E];
    if constexpr (PREFETCH) {
#pragma unroll
      for (int c4 = 0; c4 < NP / E; ++c4) A[c4] = Anext[c4];
      if (task + gridDim.x < ntasks) load_row(task + gridDim.x, Anext);
    } else {
      load_row(task, A);
    }
    // diagonal of the padding rows and of the query row (compare-select: no dynamic register index)
    {
      const T dv = (i >= k && i < q) ? T(1) : (i == q ? (T)a.kout : T(0));
      const bool set = i >= k && i <= q;
#pragma unroll
      for (int c = 0; c < NP; ++c) A[c / E][c % E] = (set && c == i) ? dv : A[c / E][c % E];
    }

    // ---- elimination (mgp_fused_wave.hip, phase 4) ----------------------------------------------
    __builtin_amdgcn_s_setprio(1);
    bool bad = false;
#pragma unroll
    for (int j = 0; j < NP - 2; ++j) {
      if (j < k) {
        const T ajj = A[j / E][j % E];
        __syncthreads();  // single wave: orders this write after the previous step's reads
        colh[i] = ajj;
        __syncthreads();
        if constexpr (sizeof(T) == 4) {
          V col[NP / E];
#pragma unroll
          for (int c4 = j / E; c4 < NP / E; ++c4) col[c4] = *reinterpret_cast<const V*>(colh + c4 * E);
          const T p = col[j / E][j % E];
          bad = bad || !(p > T(0));
          const V nt = V(-ajj * pivot_rcp(p));
#pragma unroll
          for (int c4 = j / E; c4 < NP / E; ++c4) A[c4] = col[c4] * nt + A[c4];
        } else {
          // fp64: no full copy of the column beside the row; the trailing groups in batches of GC
          const V cp = *reinterpret_cast<const V*>(colh + (j / E) * E);
          const T p = cp[j % E];
          bad = bad || !(p > T(0));
          const V nt = V(-ajj * pivot_rcp(p));
          A[j / E] = cp * nt + A[j / E];
          constexpr int GC = 6;
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
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);

    // ---- Schur block -> outputs ------------------------------------------------------------------
    T aq = T(0), aii = T(0);
#pragma unroll
    for (int c = 0; c < NP; ++c) {
      const T v = A[c / E][c % E];
      aq = c == q ? v : aq;
      aii = c == i ? v : aii;
    }
    T* mean = static_cast<T*>(a.mean);
    T* var = static_cast<T*>(a.var);
    T* yk = static_cast<T*>(a.ykinvy);
    if (live) {
      if (i == q) {
        if (var != nullptr && Kcross != nullptr) var[nb] = bad ? num<T>::nan() : aq;
        if (bad && a.info) atomicAdd(a.info, 1);
      } else if (i > q && i - q - 1 < R) {
        const int r = i - q - 1;
        if (mean != nullptr && Kcross != nullptr) mean[nb * R + r] = bad ? num<T>::nan() : -aq;
        if (yk != nullptr) yk[nb * R + r] = bad ? num<T>::nan() : -aii;
      }
    }
  }
}

template <typename T, int NP>
static int launch_solve_np(const SolveArgs& a, hipStream_t stream) {
  constexpr int NH = 64 / NP;
  constexpr int E = v16<T>::N;
  const int q = NP - 1 - (a.R > 0 ? a.R : 1);  // no responses (variance only): one empty response slot
  const size_t row_bytes = (size_t)a.k * sizeof(T);
  const uintptr_t base = reinterpret_cast<uintptr_t>(a.Kin);
  const int vec_ok = (row_bytes % 16 == 0 && base % 16 == 0) ? 2 : ((row_bytes % 8 == 0 && base % 8 == 0) ? 1 : 0);
  static Residency res;
  int per_cu = 0, cus = 0;
  const int rc = res.lookup(reinterpret_cast<const void*>(&solve_wave_kernel<T, NP>), 64, 0, &per_cu, &cus);
  if (rc != MGP_OK) return rc;
  const int64_t ntasks = (a.b + NH - 1) / NH;
  int64_t grid = (int64_t)cus * per_cu;
  if (grid > ntasks) grid = ntasks;
  hipLaunchKernelGGL((solve_wave_kernel<T, NP>), dim3((unsigned)grid), dim3(64), 0, stream, a, q, vec_ok);
  MGP_HIP_CHECK_LAUNCH();
  return MGP_OK;
}

template <typename T>
int launch_solve_wave(const SolveArgs& a, hipStream_t stream) {
  const int rows = a.k + 1 + (a.R > 0 ? a.R : 1);
  if (a.coeffs != nullptr || rows > 64) return MGP_EUNSUPPORTED;
  if (rows <= 32) return launch_solve_np<T, 32>(a, stream);
  return launch_solve_np<T, 64>(a, stream);
}

template int launch_solve_wave<float>(const SolveArgs&, hipStream_t);
template int launch_solve_wave<double>(const SolveArgs&, hipStream_t);

}  // namespace mgp

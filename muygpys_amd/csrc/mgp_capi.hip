// extern "C" boundary: argument validation + dispatch to the kernels.  See
// include/muygpys_hip.h for the contract and the reference functions each entry replaces.
#include <cstdlib>
#include <cstdio>

#include "mgp_args.h"

#include <cstdarg>
#include <cstring>

namespace mgp {

static thread_local char g_last_kernel[256] = "";
static thread_local int g_tree_grid = 0, g_tree_nh = 0;
// how the one-launch LOOCV evaluation hands sums between workgroups (mgp_loocv_tree.h); process-wide
static int tree_mode_from_env() {
  const char* e = getenv("MUYGPYS_HIP_LOOCV_TREE");
  if (!e || !*e || !strcmp(e, "tickets")) return kTreeTickets;
  if (!strcmp(e, "fenced")) return kTreeFenced;
  if (!strcmp(e, "three_launch")) return kTreeThreeLaunch;
  fprintf(stderr, "mgp: MUYGPYS_HIP_LOOCV_TREE=%s is none of tickets / fenced / three_launch; using three_launch\n", e);
  return kTreeThreeLaunch;  // (an unknown word must not select the least conservative form)
}
static int g_tree_mode = tree_mode_from_env();
void note_tree_geometry(int grid, int nh) { g_tree_grid = grid, g_tree_nh = nh; }
static thread_local int64_t g_launch_grid = 0;
static thread_local int g_launch_lds = 0;
void note_launch_geometry(int64_t grid, size_t lds_bytes) { g_launch_grid = grid, g_launch_lds = (int)lds_bytes; }
void note_launch(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_kernel, sizeof(g_last_kernel), fmt, ap);
  va_end(ap);
}

static bool valid_kernel(int id) { return id >= MGP_KERNEL_RBF && id <= MGP_KERNEL_MATERN_INF; }
static bool valid_metric(int id) { return id == MGP_METRIC_L2 || id == MGP_METRIC_F2; }

#ifdef MGP_DEBUG_HOOKS
extern int g_phase_mask;  // mgp_fused_wave.hip (timing ablations, debug builds only)
extern int g_grid_per_cu;
extern int g_lds_pad;
extern int g_bwd_stage;  // mgp_backward.hip
#endif

// which kernel family serves a fused call: the dispatcher's choice, or one family named by the
// caller (mgp_posterior_generic_* / mgp_posterior_rhs_*: tests and A/B timing)
enum { PATH_AUTO = 0, PATH_GENERIC = 1, PATH_RHS = 2 };

template <typename T>
int posterior(const T* fq, const T* fn, int d, const int64_t* bi, const int64_t* ni, int64_t b, int k, const T* tg,
              int R, int noise_mode, double eps, const T* nd, int kernel_id, int metric_id, const T* ls,
              int ls_count, T* mean, T* var, T* yk, int* info, void* stream, int path = PATH_AUTO,
              const void* packed_q = nullptr, int64_t q_stride = 0, const void* packed_nn = nullptr,
              int64_t nn_stride = 0, int targets_batch = 0, const LoocvTree* tree = nullptr, bool* tree_served = nullptr) {
  if (b < 0 || k < 1 || d < 1 || R < 1) return MGP_EINVAL;
  if (b == 0) return MGP_OK;  // empty shard: nothing to read or write (outputs may be NULL)
  const bool packed = packed_nn != nullptr;
  if (packed) {
    // both strides cover the features and the 16-byte response slot the gather reads with every row
    const int64_t need = (int64_t)(d * sizeof(T)) + 16;
    if (!packed_q || q_stride < need || nn_stride < need) return MGP_EINVAL;
    // the neighbour table carries the responses, unless they come already gathered (b, k, R)
    if (targets_batch ? tg == nullptr : nn_stride < (int64_t)((d + R) * sizeof(T))) return MGP_EINVAL;
  } else if (!fq || !fn || !tg) {
    return MGP_EINVAL;
  }
  if (!ni || !ls || !mean || !var) return MGP_EINVAL;
  if (!valid_kernel(kernel_id) || !valid_metric(metric_id)) return MGP_EINVAL;
  if (noise_mode < MGP_NOISE_SCALAR || noise_mode > MGP_NOISE_BATCH) return MGP_EINVAL;
  if (noise_mode != MGP_NOISE_SCALAR && !nd) return MGP_EINVAL;
  if (ls_count != 1 && ls_count != d) return MGP_EINVAL;
  FusedArgs a{fq, fn, bi, ni, tg, nd, ls, mean, var, yk, info, b, eps, d, k, R, noise_mode, kernel_id, metric_id,
              ls_count, 0};
  a.targets_batch = targets_batch;
  a.packed_q = packed_q;
  a.packed_nn = packed_nn;
  a.q_stride = q_stride;
  a.nn_stride = nn_stride;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // a LOOCV evaluation (mgp_loocv_*): the wave kernels walk the reduction tree themselves (one launch); behind the
  // other families the caller does
  if (tree) a.tree = *tree;
  if (tree_served) *tree_served = false;
  if (packed || path == PATH_AUTO) {
    int rc = launch_fused_wave<T>(a, s);
    if (rc == MGP_OK && tree && tree_served) *tree_served = true;
    if (rc == MGP_EUNSUPPORTED && tree) {  // (a grid beyond the tree's leaves: the plain launch, the tree by kernels)
      a.tree = LoocvTree{};
      rc = launch_fused_wave<T>(a, s);
    }
    // prepared tables with up to sixteen responses behind the features (round 5): the fp32 prediction kernel on the
    // matrix cores' layout reads them too (BASELINE config 5: two 128-byte lines per neighbour instead of three)
    if constexpr (sizeof(T) == 4) {
      if (packed && rc == MGP_EUNSUPPORTED && !tree && !yk) rc = launch_fused_rhs_mf(a, s);
    }
    if (packed || rc != MGP_EUNSUPPORTED) return rc;  // (prepared tables, MGP_EUNSUPPORTED: the caller uses the plain tables)
  }
  a.tree = LoocvTree{};
  if (path == PATH_RHS) return launch_fused_rhs<T>(a, s);
  if (path == PATH_AUTO) {
    int rc;
    rc = launch_fused_rhs<T>(a, s);
    if (rc != MGP_EUNSUPPORTED) return rc;
    rc = launch_fused_wide<T>(a, s);
    if (rc != MGP_EUNSUPPORTED) return rc;
    if constexpr (sizeof(T) == 8) {
      rc = launch_fused_wide64(a, s);
      if (rc != MGP_EUNSUPPORTED) return rc;
    }
  }
  return launch_fused_generic<T>(a, s);
}

// fused fast-mean precompute: coefficients K^-1 y of every neighbourhood (no query involved)
template <typename T>
int fast_coefficients(const T* fn, int d, const int64_t* ni, int64_t b, int k, const T* tg, int noise_mode, double eps,
                      const T* nd, int kernel_id, int metric_id, const T* ls, int ls_count, T* coeffs, int* info,
                      void* stream) {
  if (b < 0 || k < 1 || d < 1) return MGP_EINVAL;
  if (b == 0) return MGP_OK;
  if (!fn || !ni || !tg || !ls || !coeffs) return MGP_EINVAL;
  if (!valid_kernel(kernel_id) || !valid_metric(metric_id)) return MGP_EINVAL;
  if (noise_mode < MGP_NOISE_SCALAR || noise_mode > MGP_NOISE_BATCH) return MGP_EINVAL;
  if (noise_mode != MGP_NOISE_SCALAR && !nd) return MGP_EINVAL;
  if (ls_count != 1 && ls_count != d) return MGP_EINVAL;
  // the query slot is fed the first neighbour of each row: its outputs are not stored
  FusedArgs a{fn, fn, nullptr, ni, tg, nd, ls, nullptr, nullptr, nullptr, info, b, eps, d, k, 1, noise_mode, kernel_id,
              metric_id, ls_count, 0};
  a.coeffs = coeffs;
  return launch_fused_wave<T>(a, static_cast<hipStream_t>(stream));
}

template <typename T>
int posterior_backward(const T* fq, const T* fn, int d, const int64_t* bi, const int64_t* ni, int64_t b, int k,
                       const T* tg, int R, int noise_mode, double eps, const T* nd, int kernel_id, int metric_id,
                       const T* ls, int ls_count, const T* gmean, const T* gvar, T* gfq, T* gfn, T* gtg, T* gls,
                       T* gnz, int* info, void* stream, const T* gyk = nullptr) {
  if (b < 0 || k < 1 || d < 1 || R < 1) return MGP_EINVAL;
  if (b == 0) return MGP_OK;
  if (!fq || !fn || !ni || !tg || !ls) return MGP_EINVAL;
  if (!gmean && !gvar && !gyk) return MGP_EINVAL;
  if (gyk && R != 1) return MGP_EINVAL;  // (the LOOCV losses are defined for one response)
  if (!valid_kernel(kernel_id) || !valid_metric(metric_id)) return MGP_EINVAL;
  if (noise_mode < MGP_NOISE_SCALAR || noise_mode > MGP_NOISE_BATCH) return MGP_EINVAL;
  if (noise_mode != MGP_NOISE_SCALAR && !nd) return MGP_EINVAL;
  if (ls_count != 1 && ls_count != d) return MGP_EINVAL;
  BackwardArgs g{{fq, fn, bi, ni, tg, nd, ls, nullptr, nullptr, nullptr, info, b, eps, d, k, R, noise_mode, kernel_id,
                  metric_id, ls_count, 0},
                 gmean, gvar, gfq, gfn, gtg, gls, gnz};
  g.grad_yk = gyk;
  static const bool lds_only = getenv("MGP_BACKWARD_LDS") != nullptr;  // A/B switch (timing only)
  // hyper-parameter gradients on the forward kernel itself (round 6: the dealt-triangle shapes of BASELINE config 4 and
  // the 32-slot shapes of config 3)
  if (!lds_only) {
    const int rc = launch_backward_fwd<T>(g, static_cast<hipStream_t>(stream));
    if (rc != MGP_EUNSUPPORTED) return rc;
  }
  if (!lds_only) {
    const int rc = launch_backward_wave<T>(g, static_cast<hipStream_t>(stream));
    if (rc != MGP_EUNSUPPORTED) return rc;
  }
  return launch_backward<T>(g, static_cast<hipStream_t>(stream));
}

template <typename T>
int solve(const T* Kin, const T* Kc, const T* Y, int64_t b, int k, int R, double kout, T* mean, T* var, T* yk,
          T* coeffs, int* info, void* stream) {
  if (b < 0 || k < 1 || R < 0) return MGP_EINVAL;
  if (b == 0) return MGP_OK;
  if (!Kin) return MGP_EINVAL;
  if (R > 0 && !Y) return MGP_EINVAL;
  if ((mean || var) && !Kc) return MGP_EINVAL;
  if ((mean || yk || coeffs) && R == 0) return MGP_EINVAL;
  if (b == 0) return MGP_OK;
  SolveArgs a{Kin, Kc, Y, mean, var, yk, coeffs, info, b, kout, k, R};
  const int rc = launch_solve_wave<T>(a, static_cast<hipStream_t>(stream));
  if (rc != MGP_EUNSUPPORTED) return rc;
  return launch_solve_generic<T>(a, static_cast<hipStream_t>(stream));
}
// behind the fused launch of a LOOCV evaluation: nothing when a wave kernel walked the tree itself; the walk by kernels
// over the SAME leaves when it was launched without (three_launch); over the canonical leaves behind another family
template <typename T>
int loocv_finish(int rc, bool served, const LoocvTree& tr, const T* mean, const T* var, const T* yk, const int64_t* bi,
                 int64_t b, hipStream_t s) {
  if (rc != MGP_OK) return rc;
  if (served && tr.mode != kTreeThreeLaunch) return MGP_OK;
  const int grid = served ? g_tree_grid : 0, nh = served ? g_tree_nh : 0;
  return launch_loocv_tree<T>(tr, grid, nh, mean, var, yk, bi, b, s);
}
}  // namespace mgp

using namespace mgp;
#define S_(x) static_cast<hipStream_t>(x)

// General-smoothness Matern through the fused wave kernels (fp32, k + 1 + R <= 32 or a static shape);
// MGP_EUNSUPPORTED elsewhere: the caller evaluates mgp_matern_gen_* on materialised distances instead.
template <typename T>
static int posterior_gen(const T* fq, const T* fn, const void* packed_q, int64_t q_stride, const void* packed_nn,
                         int64_t nn_stride, int d, const int64_t* bi, const int64_t* ni, int64_t b, int k, const T* tg,
                         int R, int targets_batch, int noise_mode, double eps, const T* nd, double smoothness,
                         int metric_id, const T* ls, int ls_count, T* mean, T* var, T* yk, int* info, void* stream) {
  if (b < 0 || k < 1 || d < 1 || R < 1 || !(smoothness > 0.0)) return MGP_EINVAL;
  if (b == 0) return MGP_OK;
  const bool packed = packed_nn != nullptr;
  if (packed) {
    const int64_t need = (int64_t)(d * sizeof(T)) + 16;
    if (!packed_q || q_stride < need || nn_stride < need) return MGP_EINVAL;
    if (targets_batch ? tg == nullptr : nn_stride < (int64_t)((d + R) * sizeof(T))) return MGP_EINVAL;
  } else if (!fq || !fn || !tg) {
    return MGP_EINVAL;
  }
  if (!ni || !ls || !mean || !var || !valid_metric(metric_id)) return MGP_EINVAL;
  if (noise_mode < MGP_NOISE_SCALAR || noise_mode > MGP_NOISE_BATCH) return MGP_EINVAL;
  if (noise_mode != MGP_NOISE_SCALAR && !nd) return MGP_EINVAL;
  if (ls_count != 1 && ls_count != d) return MGP_EINVAL;
  if (smoothness > 30.0) return MGP_EUNSUPPORTED;  // beyond nu = 30 the RBF limit is the better model anyway
  FusedArgs a{fq, fn, bi, ni, tg, nd, ls, mean, var, yk, info, b, eps, d, k, R, noise_mode, MGP_KERNEL_MATERN_GEN, metric_id,
              ls_count, 0};
  a.targets_batch = targets_batch;
  a.packed_q = packed_q;
  a.packed_nn = packed_nn;
  a.q_stride = q_stride;
  a.nn_stride = nn_stride;
  a.smoothness = smoothness;
  return launch_fused_wave<T>(a, static_cast<hipStream_t>(stream));
}

extern "C" {

const char* mgp_version(void) { return "muygpys_amd-hip 0.1 (gfx950)"; }
int mgp_max_nn_count(int elem_size, int R) { return max_nn_count(elem_size, R); }
int mgp_reduce_scratch_doubles(void) { return reduce_scratch_doubles(); }
int64_t mgp_loocv_scratch_bytes(void) { return tree_scratch_bytes(); }
int64_t mgp_loocv_scratch_zero_bytes(void) { return tree_zero_bytes(); }
int mgp_last_launch_geometry(int64_t* workgroups, int* lds_bytes) {
  if (!workgroups || !lds_bytes) return MGP_EINVAL;
  *workgroups = g_launch_grid, *lds_bytes = g_launch_lds;
  return MGP_OK;
}
int mgp_loocv_tree_mode_get(void) { return g_tree_mode; }
int mgp_loocv_tree_mode_set(int mode) {
  if (mode != kTreeTickets && mode != kTreeFenced && mode != kTreeThreeLaunch) return MGP_EINVAL;
  g_tree_mode = mode;
  return MGP_OK;
}
int mgp_last_loocv_geometry(int* grid, int* nh) {
  if (!grid || !nh) return MGP_EINVAL;
  *grid = g_tree_grid, *nh = g_tree_nh;
  return MGP_OK;
}
int mgp_matern_gen_constants(double smoothness, double* out7) {
  if (!out7 || !(smoothness > 0.0)) return MGP_EINVAL;
  matern_gen_constants_host(smoothness, out7);
  return MGP_OK;
}
int mgp_posterior_kernel_name(int elem_size, int d, int k, int R, int packed, int path, char* buf, int len) {
  if (!buf || len < 1 || (elem_size != 4 && elem_size != 8)) return MGP_EINVAL;
  if (path == PATH_AUTO && describe_fused_wave(elem_size, d, k, R, packed, buf, len) > 0) return MGP_OK;
  const char* t = elem_size == 4 ? "float" : "double";
  if (packed) return snprintf(buf, len, "%s", "") < 0 ? MGP_EINVAL : MGP_EUNSUPPORTED;
  if (path != PATH_GENERIC && k <= 64 && R <= 16)
    snprintf(buf, len, "mgp::fused_rhs_kernel<%s,%d>", t, R <= 4 ? 4 : 16);
  else if (path == PATH_AUTO && k + 1 + R >= 65 && k + 1 + R <= 128 && R <= 16 && d <= 64)
    snprintf(buf, len, elem_size == 4 ? "mgp::fused_wide_kernel" : "mgp::fused_wide64_kernel");
  else
    snprintf(buf, len, "mgp::fused_generic_kernel<%s>", t);
  return MGP_OK;
}
int mgp_last_kernel_name(char* buf, int len) {
  if (!buf || len < 1) return MGP_EINVAL;
  snprintf(buf, len, "%s", mgp::g_last_kernel);
  return MGP_OK;
}
int mgp_allreduce_partials(double* partials, int count, void* nccl_comm, void* st) {
  return allreduce_partials(partials, count, nccl_comm, S_(st));
}
int mgp_jit_prepare(int elem_size, int k, int R, int d, int packed, int kernel_id) {
  if ((elem_size != 4 && elem_size != 8) || k < 1 || R < 1 || d < 1 || !(valid_kernel(kernel_id) || kernel_id == MGP_KERNEL_MATERN_GEN))
    return MGP_EINVAL;
  return prepare_fused_wave(elem_size, d, k, R, packed, kernel_id);
}
int mgp_jit_prepare_backward(int elem_size, int k, int d, int kernel_id) {
  if ((elem_size != 4 && elem_size != 8) || k < 1 || d < 1 || !valid_kernel(kernel_id)) return MGP_EINVAL;
  return prepare_backward_fwd(elem_size, k, d, kernel_id);
}
int mgp_jit_mode(void) { return jit_mode(); }
int mgp_jit_loaded_count(void) { return jit_loaded_count(); }
int mgp_jit_source_hash(char* buf, int len) {
  if (!buf || len < 17) return MGP_EINVAL;
  snprintf(buf, len, "%016llx", (unsigned long long)jit_source_hash());
  return MGP_OK;
}
int64_t mgp_packed_row_bytes(int d, int R, int elem_size) {
  if (d < 1 || R < 0 || (elem_size != 4 && elem_size != 8)) return MGP_EINVAL;
  // the gather always reads the 16-byte slot behind the features (the responses), also from a table
  // packed without responses (a query table): the stride reserves it, so the read stays inside the row
  const int64_t resp = (int64_t)R * elem_size > 16 ? (int64_t)R * elem_size : 16;
  return ((int64_t)d * elem_size + resp + 63) / 64 * 64;
}
#ifdef MGP_DEBUG_HOOKS
/* timing-ablation hooks of debug builds (tools/kbench.py, tools/bwdbench.py); absent from the shipped library */
void mgp_debug_set_phase_mask(int mask) { mgp::g_phase_mask = mask; }
void mgp_debug_set_grid_per_cu(int n) { mgp::g_grid_per_cu = n; }
void mgp_debug_set_lds_pad(int n) { mgp::g_lds_pad = n; }
void mgp_debug_set_bwd_stage(int n) { mgp::g_bwd_stage = n; }
#endif

int mgp_posterior_f32(const float* fq, const float* fn, int d, const int64_t* bi, const int64_t* ni, int64_t b, int k,
                      const float* tg, int R, int nm, double eps, const float* nd, int kid, int mid, const float* ls,
                      int lsc, float* mean, float* var, float* yk, int* info, void* st) {
  return posterior<float>(fq, fn, d, bi, ni, b, k, tg, R, nm, eps, nd, kid, mid, ls, lsc, mean, var, yk, info, st);
}
int mgp_posterior_f64(const double* fq, const double* fn, int d, const int64_t* bi, const int64_t* ni, int64_t b,
                      int k, const double* tg, int R, int nm, double eps, const double* nd, int kid, int mid,
                      const double* ls, int lsc, double* mean, double* var, double* yk, int* info, void* st) {
  return posterior<double>(fq, fn, d, bi, ni, b, k, tg, R, nm, eps, nd, kid, mid, ls, lsc, mean, var, yk, info, st);
}

#define MGP_DEFINE_PATHS(SUF, T)                                                                                    \
  int mgp_posterior_generic_##SUF(const T* fq, const T* fn, int d, const int64_t* bi, const int64_t* ni, int64_t b,  \
                                  int k, const T* tg, int R, int nm, double eps, const T* nd, int kid, int mid,      \
                                  const T* ls, int lsc, T* mean, T* var, T* yk, int* info, void* st) {               \
    return posterior<T>(fq, fn, d, bi, ni, b, k, tg, R, nm, eps, nd, kid, mid, ls, lsc, mean, var, yk, info, st,    \
                        PATH_GENERIC);                                                                               \
  }                                                                                                                  \
  int mgp_posterior_rhs_##SUF(const T* fq, const T* fn, int d, const int64_t* bi, const int64_t* ni, int64_t b,      \
                              int k, const T* tg, int R, int nm, double eps, const T* nd, int kid, int mid,          \
                              const T* ls, int lsc, T* mean, T* var, T* yk, int* info, void* st) {                   \
    return posterior<T>(fq, fn, d, bi, ni, b, k, tg, R, nm, eps, nd, kid, mid, ls, lsc, mean, var, yk, info, st,    \
                        PATH_RHS);                                                                                   \
  }                                                                                                                  \
  int mgp_posterior_gathered_##SUF(const T* fq, const T* fn, int d, const int64_t* bi, const int64_t* ni, int64_t b, \
                                   int k, const T* nn_tg, int R, int nm, double eps, const T* nd, int kid, int mid,  \
                                   const T* ls, int lsc, T* mean, T* var, T* yk, int* info, void* st) {              \
    return posterior<T>(fq, fn, d, bi, ni, b, k, nn_tg, R, nm, eps, nd, kid, mid, ls, lsc, mean, var, yk, info, st, \
                        PATH_AUTO, nullptr, 0, nullptr, 0, 1);                                                       \
  }                                                                                                                  \
  int mgp_table_pack_##SUF(const T* feat, const T* targets, int64_t n, int d, int R, void* packed,                   \
                           int64_t stride_bytes, void* st) {                                                         \
    if (!feat || !packed || n < 0 || d < 1 || R < 0) return MGP_EINVAL;                                              \
    return launch_table_pack<T>(feat, targets, n, d, R, packed, stride_bytes, S_(st));                               \
  }                                                                                                                  \
  int mgp_posterior_packed_##SUF(const void* packed_q, int64_t q_stride, const void* packed_nn, int64_t nn_stride,   \
                                 int d, const int64_t* bi, const int64_t* ni, int64_t b, int k, int R, int nm,       \
                                 double eps, const T* nd, int kid, int mid, const T* ls, int lsc, T* mean, T* var,   \
                                 T* yk, int* info, void* st) {                                                       \
    return posterior<T>(nullptr, nullptr, d, bi, ni, b, k, nullptr, R, nm, eps, nd, kid, mid, ls, lsc, mean, var,   \
                        yk, info, st, PATH_AUTO, packed_q, q_stride, packed_nn, nn_stride);                          \
  }                                                                                                                  \
  int mgp_posterior_packed_gathered_##SUF(const void* packed_q, int64_t q_stride, const void* packed_nn,            \
                                          int64_t nn_stride, int d, const int64_t* bi, const int64_t* ni, int64_t b, \
                                          int k, const T* nn_tg, int R, int nm, double eps, const T* nd, int kid,    \
                                          int mid, const T* ls, int lsc, T* mean, T* var, T* yk, int* info,          \
                                          void* st) {                                                                \
    return posterior<T>(nullptr, nullptr, d, bi, ni, b, k, nn_tg, R, nm, eps, nd, kid, mid, ls, lsc, mean, var, yk, \
                        info, st, PATH_AUTO, packed_q, q_stride, packed_nn, nn_stride, 1);                           \
  }                                                                                                                  \
  int mgp_posterior_gen_##SUF(const T* fq, const T* fn, const void* packed_q, int64_t q_stride,                     \
                              const void* packed_nn, int64_t nn_stride, int d, const int64_t* bi, const int64_t* ni, \
                              int64_t b, int k, const T* tg, int R, int targets_batch, int nm, double eps,          \
                              const T* nd, double smoothness, int mid, const T* ls, int lsc, T* mean, T* var,       \
                              T* yk, int* info, void* st) {                                                          \
    return posterior_gen<T>(fq, fn, packed_q, q_stride, packed_nn, nn_stride, d, bi, ni, b, k, tg, R, targets_batch, \
                            nm, eps, nd, smoothness, mid, ls, lsc, mean, var, yk, info, st);                         \
  }                                                                                                                  \
  int mgp_loocv_##SUF(const T* feat, int d, const int64_t* bi, const int64_t* ni, int64_t b, int k, const T* tg,     \
                      int nm, double eps, const T* nd, int kid, int mid, const T* ls, int lsc, T* mean, T* var,      \
                      T* yk, int* info, double huber_delta, double* partials, void* scratch, void* st) {             \
    if ((b > 0 && (!mean || !var || !yk)) || !partials || !scratch || !(huber_delta > 0)) return MGP_EINVAL;         \
    LoocvTree tr = loocv_tree_layout(scratch, partials, tg, (int64_t)sizeof(T), huber_delta);                        \
    tr.mode = g_tree_mode;                                                                                           \
    bool served = false;                                                                                             \
    note_tree_geometry(0, 0);                                                                                        \
    const int rc = posterior<T>(feat, feat, d, bi, ni, b, k, tg, 1, nm, eps, nd, kid, mid, ls, lsc, mean, var, yk,  \
                                info, st, PATH_AUTO, nullptr, 0, nullptr, 0, 0, &tr, &served);                       \
    return loocv_finish<T>(rc, served, tr, mean, var, yk, bi, b, S_(st));                                            \
  }                                                                                                                  \
  int mgp_loocv_tree_##SUF(const T* mean, const T* var, const T* yk, const void* resp, int64_t resp_stride,          \
                           const int64_t* bi, int64_t b, double huber_delta, int grid, int nh, double* partials,     \
                           void* scratch, void* st) {                                                                \
    if (b < 0 || (b > 0 && (!mean || !var || !yk || !resp)) || !partials || !scratch || !(huber_delta > 0))          \
      return MGP_EINVAL;                                                                                             \
    return launch_loocv_tree<T>(loocv_tree_layout(scratch, partials, resp, resp_stride, huber_delta), grid, nh,      \
                                mean, var, yk, bi, b, S_(st));                                                       \
  }                                                                                                                  \
  int mgp_loocv_packed_##SUF(const void* packed, int64_t stride, int d, const int64_t* bi, const int64_t* ni,        \
                             int64_t b, int k, int nm, double eps, const T* nd, int kid, int mid, const T* ls,       \
                             int lsc, T* mean, T* var, T* yk, int* info, double huber_delta, double* partials,       \
                             void* scratch, void* st) {                                                              \
    if ((b > 0 && (!mean || !var || !yk)) || !partials || !scratch || !(huber_delta > 0)) return MGP_EINVAL;         \
    LoocvTree tr = loocv_tree_layout(scratch, partials, static_cast<const char*>(packed) + (size_t)d * sizeof(T),    \
                                     stride, huber_delta);                                                           \
    tr.mode = g_tree_mode;                                                                                           \
    bool served = false;                                                                                             \
    note_tree_geometry(0, 0);                                                                                        \
    const int rc = posterior<T>(nullptr, nullptr, d, bi, ni, b, k, nullptr, 1, nm, eps, nd, kid, mid, ls, lsc, mean, \
                                var, yk, info, st, PATH_AUTO, packed, stride, packed, stride, 0, &tr, &served);      \
    return loocv_finish<T>(rc, served, tr, mean, var, yk, bi, b, S_(st));                                            \
  }
MGP_DEFINE_PATHS(f32, float)
MGP_DEFINE_PATHS(f64, double)

int mgp_topk_rows_f32(const float* x, int64_t rows, int cols, int64_t row_stride, int k, float* out_values, int32_t* out_cols,
                      void* st) {
  if (rows < 0 || cols < 1 || k < 1 || row_stride < cols) return MGP_EINVAL;
  if (rows == 0) return MGP_OK;
  if (!x || !out_values || !out_cols) return MGP_EINVAL;
  return launch_topk_rows(x, rows, cols, row_stride, k, out_values, out_cols, S_(st));
}
int mgp_knn_finish_f32(const float* queries, const float* train, int d, const int32_t* candidates, int64_t m, int k,
                       const int64_t* row_map, int64_t* out_idx, float* out_dist, void* st) {
  if (m < 0 || d < 1 || k < 1) return MGP_EINVAL;
  if (m == 0) return MGP_OK;
  if (!queries || !train || !candidates || !out_idx || !out_dist) return MGP_EINVAL;
  return launch_knn_finish(queries, train, d, candidates, m, k, row_map, out_idx, out_dist, S_(st));
}

int mgp_knn_scan_f32(const float* train, const float* train_sqn, int64_t n, int d, const float* queries,
                     const float* query_sqn, const int64_t* self_idx, int64_t m, int k, int64_t start, float* best_d,
                     int32_t* best_i, int32_t* overflow, void* st) {
  if (n < 0 || m < 0 || d < 1 || k < 1 || start < 0) return MGP_EINVAL;
  if (m == 0 || start >= n) return MGP_OK;
  if (!train || !train_sqn || !queries || !query_sqn || !best_d || !best_i || !overflow) return MGP_EINVAL;
  KnnArgs a{train, train_sqn, queries, query_sqn, self_idx, best_d, best_i, overflow, n, m, start, d, k};
  return launch_knn_scan(a, S_(st));
}

int mgp_knn_scan_bf16x3(const float* train, const void* packed_train, const float* train_sqn, int64_t n, int d,
                        const float* queries, const void* packed_queries, const float* query_sqn,
                        const int64_t* self_idx, int64_t m, int k, int64_t start, float* best_d, int32_t* best_i,
                        int32_t* overflow, void* st) {
  if (n < 0 || m < 0 || d < 1 || k < 1 || start < 0) return MGP_EINVAL;
  if (m == 0 || start >= n) return MGP_OK;
  if (!train || !packed_train || !train_sqn || !queries || !packed_queries || !query_sqn || !best_d || !best_i ||
      !overflow)
    return MGP_EINVAL;
  KnnPackedArgs a{train, packed_train, train_sqn, queries, packed_queries, query_sqn, self_idx, best_d, best_i,
                  overflow, n, m, start, d, k};
  return launch_knn_scan_packed(a, S_(st));
}

int mgp_knn_scan_bf16x2_d8(const float* train, const void* packed_train, const float* train_sqn, int64_t n, int d,
                           const float* queries, const void* packed_queries, const float* query_sqn,
                           const int64_t* self_idx, int64_t m, int k, int64_t start, float* best_d, int32_t* best_i,
                           int32_t* overflow, void* st) {
  if (n < 0 || m < 0 || d < 1 || k < 1 || start < 0) return MGP_EINVAL;
  if (d > 8) return MGP_EUNSUPPORTED;
  if (m == 0 || start >= n) return MGP_OK;
  if (!train || !packed_train || !train_sqn || !queries || !packed_queries || !query_sqn || !best_d || !best_i ||
      !overflow)
    return MGP_EINVAL;
  KnnPackedArgs a{train, packed_train, train_sqn, queries, packed_queries, query_sqn, self_idx, best_d, best_i,
                  overflow, n, m, start, d, k};
  a.layout = 1;
  return launch_knn_scan_packed(a, S_(st));
}

#define MGP_DEFINE_COEF(SUF, T)                                                                                     \
  int mgp_fast_coefficients_##SUF(const T* fn, int d, const int64_t* ni, int64_t b, int k, const T* tg, int nm,      \
                                  double eps, const T* nd, int kid, int mid, const T* ls, int lsc, T* coeffs,        \
                                  int* info, void* st) {                                                             \
    return fast_coefficients<T>(fn, d, ni, b, k, tg, nm, eps, nd, kid, mid, ls, lsc, coeffs, info, st);              \
  }
MGP_DEFINE_COEF(f32, float)
MGP_DEFINE_COEF(f64, double)

#define MGP_DEFINE_BWD(SUF, T)                                                                                      \
  int mgp_posterior_backward_##SUF(const T* fq, const T* fn, int d, const int64_t* bi, const int64_t* ni, int64_t b, \
                                   int k, const T* tg, int R, int nm, double eps, const T* nd, int kid, int mid,     \
                                   const T* ls, int lsc, const T* gmean, const T* gvar, T* gfq, T* gfn, T* gtg,     \
                                   T* gls, T* gnz, int* info, void* st) {                                            \
    return posterior_backward<T>(fq, fn, d, bi, ni, b, k, tg, R, nm, eps, nd, kid, mid, ls, lsc, gmean, gvar, gfq,  \
                                 gfn, gtg, gls, gnz, info, st);                                                      \
  }                                                                                                                  \
  int mgp_loocv_backward_##SUF(const T* feat, int d, const int64_t* bi, const int64_t* ni, int64_t b, int k,        \
                               const T* tg, int nm, double eps, const T* nd, int kid, int mid, const T* ls, int lsc, \
                               const T* gmean, const T* gvar, const T* gyk, T* gls, T* gnz, int* info, void* st) {   \
    if (b > 0 && (!gmean || !gvar || !gyk || (!gls && !gnz))) return MGP_EINVAL;                                     \
    return posterior_backward<T>(feat, feat, d, bi, ni, b, k, tg, 1, nm, eps, nd, kid, mid, ls, lsc, gmean, gvar,    \
                                 nullptr, nullptr, nullptr, gls, gnz, info, st, gyk);                                \
  }
MGP_DEFINE_BWD(f32, float)
MGP_DEFINE_BWD(f64, double)
int mgp_max_nn_count_backward(int elem_size) { return max_nn_count_backward(elem_size); }

#define MGP_DEFINE(SUF, T)                                                                                           \
  int mgp_crosswise_diffs_##SUF(const T* fq, const T* fn, int d, const int64_t* bi, const int64_t* ni, int64_t b,    \
                                int k, T* out, void* st) {                                                           \
    if (!fq || !fn || !ni || !out || b < 0 || k < 1 || d < 1) return MGP_EINVAL;                                     \
    return launch_crosswise_diffs<T>(fq, fn, d, bi, ni, b, k, out, S_(st));                                          \
  }                                                                                                                  \
  int mgp_pairwise_diffs_##SUF(const T* f, int d, const int64_t* ni, int64_t b, int k, T* out, void* st) {           \
    if (!f || !ni || !out || b < 0 || k < 1 || d < 1) return MGP_EINVAL;                                             \
    return launch_pairwise_diffs<T>(f, d, ni, b, k, out, S_(st));                                                    \
  }                                                                                                                  \
  int mgp_crosswise_dists_##SUF(const T* fq, const T* fn, int d, const int64_t* bi, const int64_t* ni, int64_t b,    \
                                int k, int mid, T* out, void* st) {                                                  \
    if (!fq || !fn || !ni || !out || b < 0 || k < 1 || d < 1 || !valid_metric(mid)) return MGP_EINVAL;               \
    return launch_crosswise_dists<T>(fq, fn, d, bi, ni, b, k, mid, out, S_(st));                                     \
  }                                                                                                                  \
  int mgp_pairwise_dists_##SUF(const T* f, int d, const int64_t* ni, int64_t b, int k, int mid, T* out, void* st) {  \
    if (!f || !ni || !out || b < 0 || k < 1 || d < 1 || !valid_metric(mid)) return MGP_EINVAL;                       \
    return launch_pairwise_dists<T>(f, d, ni, b, k, mid, out, S_(st));                                               \
  }                                                                                                                  \
  int mgp_reduce_diffs_##SUF(const T* diffs, int64_t n, int d, const T* ls, int mid, T* out, void* st) {             \
    if (!diffs || !out || n < 0 || d < 1 || !valid_metric(mid)) return MGP_EINVAL;                                   \
    return launch_reduce_diffs<T>(diffs, n, d, ls, mid, out, S_(st));                                                \
  }                                                                                                                  \
  int mgp_kernel_apply_##SUF(const T* in, int64_t n, int kid, double scale, T* out, void* st) {                      \
    if (!in || !out || n < 0 || !valid_kernel(kid)) return MGP_EINVAL;                                               \
    return launch_kernel_apply<T>(in, n, kid, scale, out, S_(st));                                                   \
  }                                                                                                                  \
  int mgp_matern_gen_##SUF(const T* in, int64_t n, double in_scale, double smoothness, T* out, void* st) {           \
    if (!in || !out || n < 0) return MGP_EINVAL;                                                                     \
    return launch_matern_gen<T>(in, n, in_scale, smoothness, out, S_(st));                                           \
  }                                                                                                                  \
  int mgp_perturb_##SUF(const T* Kin, int64_t b, int k, int nm, double eps, const T* nd, T* out, void* st) {         \
    if (!Kin || !out || b < 0 || k < 1) return MGP_EINVAL;                                                           \
    if (nm != MGP_NOISE_SCALAR && nm != MGP_NOISE_BATCH) return MGP_EINVAL;                                          \
    if (nm == MGP_NOISE_BATCH && !nd) return MGP_EINVAL;                                                             \
    return launch_perturb<T>(Kin, b, k, nm, eps, nd, out, S_(st));                                                   \
  }                                                                                                                  \
  int mgp_solve_##SUF(const T* Kin, const T* Kc, const T* Y, int64_t b, int k, int R, double kout, T* mean, T* var,  \
                      T* yk, T* coeffs, int* info, void* st) {                                                       \
    return solve<T>(Kin, Kc, Y, b, k, R, kout, mean, var, yk, coeffs, info, st);                                     \
  }                                                                                                                  \
  int mgp_loss_sums_##SUF(const T* pred, const T* target, const T* var, int64_t n, const double* scale_dev,          \
                          double hd, double ld, double* out, double* scratch, void* st) {                            \
    if (!pred || !target || !out || n < 0 || !(hd > 0) || !(ld > 0)) return MGP_EINVAL;                              \
    return launch_loss_sums<T>(pred, target, var, n, scale_dev, hd, ld, out, scratch, S_(st));                       \
  }                                                                                                                  \
  int mgp_column_sums_##SUF(const T* x, int64_t n, int R, double* out, double* scratch, void* st) {                  \
    if (!x || !out || n < 0 || R < 0) return MGP_EINVAL;                                                             \
    return launch_column_sums<T>(x, n, R, out, scratch, S_(st));                                                     \
  }

#define MGP_DEFINE_FAST(SUF, T)                                                                                     \
  int mgp_fast_posterior_mean_##SUF(const T* fq, const T* fn, int d, const int64_t* bi, const int64_t* ni, int64_t b, \
                                    int k, const T* coeffs, const int64_t* crow, int R, int kid, int mid,            \
                                    const T* ls, int lsc, T* mean, void* st) {                                       \
    if (b < 0 || k < 1 || d < 1 || R < 1) return MGP_EINVAL;                                                         \
    if (b == 0) return MGP_OK;                                                                                       \
    if (!fq || !fn || !ni || !coeffs || !crow || !ls || !mean) return MGP_EINVAL;                                    \
    if (!valid_kernel(kid) || !valid_metric(mid) || (lsc != 1 && lsc != d)) return MGP_EINVAL;                       \
    return launch_fast_mean<T>(fq, fn, d, bi, ni, b, k, coeffs, crow, R, kid, mid, ls, lsc, mean, S_(st));           \
  }
MGP_DEFINE_FAST(f32, float)
MGP_DEFINE_FAST(f64, double)

MGP_DEFINE(f32, float)
MGP_DEFINE(f64, double)

}  // extern "C"

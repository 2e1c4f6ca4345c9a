// Argument blocks + launcher declarations shared by the translation units.
#pragma once

#include "mgp_device.h"
#include "mgp_loocv_tree.h"

namespace mgp {

struct FusedArgs {
  const void* feat_q;
  const void* feat_nn;
  const int64_t* batch_idx;
  const int64_t* nn_idx;
  const void* targets;
  const void* noise_dev;
  const void* length_scale;
  void* mean;
  void* var;
  void* ykinvy;
  int* info;
  int64_t b;
  double noise_scalar;
  int d, k, R, noise_mode, kernel_id, metric_id, ls_count, dc;
  void* coeffs = nullptr;  // (b, k) K^-1 y per neighbourhood (fused fast-mean precompute), or nullptr
  // prepared tables (mgp_table_pack_*): rows [features d | responses R | pad], byte strides; when
  // packed_nn is set feat_q / feat_nn / targets are not read by the pipelined wave kernels
  // targets_batch != 0: `targets` is the already gathered (b, k, R) tensor (what the reference's
  // make_*_tensors hand to the solve functions, gp/muygps.py:474,543) instead of the (n, R) table
  int targets_batch = 0;
  const void* packed_q = nullptr;
  const void* packed_nn = nullptr;
  int64_t q_stride = 0, nn_stride = 0;
  double smoothness = 0.0;  // kernel_id == MGP_KERNEL_MATERN_GEN: the Matern smoothness nu
  // One-launch LOOCV evaluation (mgp_loocv_*; wave kernels): tree.out != nullptr -- a workgroup that has run out of
  // tasks reduces its own outputs (its leaf of the reduction tree) and walks up as far as its tickets are the last
  // ones (mgp_loocv_tree.h).  Only launch_fused_wave serves it (and fills in tree.grid / tree.nh); behind every other
  // kernel family the caller walks the same tree with launch_loocv_tree (mgp_tensor_ops.hip).
  LoocvTree tree;
  // Backward on the forward kernels (BWD instantiations of fused_wave_kernel,
  // mgp_backward_dlt.hip; nullptr everywhere else): upstream cotangents of mean / variance / y^T K^-1 y, (b) each (any
  // may be null = zero), and the per-neighbourhood partials they produce -- d/d length scale(s) (b, ls_count), the
  // diagonal (noise) cotangent (b, k), and the cotangent of the neighbours' responses (n, 1; atomic adds)
  const void* bwd_gmean = nullptr;
  const void* bwd_gvar = nullptr;
  const void* bwd_gyk = nullptr;
  void* bwd_gls = nullptr;
  void* bwd_gnz = nullptr;
  void* bwd_gtg = nullptr;
  void* bwd_gnn = nullptr;  // (n_nn, d) += feature cotangents of the neighbour rows (row-per-lane form only)
  void* bwd_gq = nullptr;   // (n_q, d)  += ... of the query rows
};

#ifndef __HIPCC_RTC__  // the rest is host side: other argument blocks and the launcher declarations
#define MGP_MAX_DEVICES 64

struct SolveArgs {
  const void* Kin;
  const void* Kcross;
  const void* Y;
  void* mean;
  void* var;
  void* ykinvy;
  void* coeffs;
  int* info;
  int64_t b;
  double kout;
  int k, R;
};

// Vector-Jacobian product of the fused path (mgp_backward.hip); every grad_* may be NULL.
struct BackwardArgs {
  FusedArgs f;             // forward inputs (mean/var/ykinvy unused)
  const void* grad_mean;   // (b, R) upstream
  const void* grad_var;    // (b)    upstream
  void* grad_feat_q;       // (n_q, d)   += (atomic)
  void* grad_feat_nn;      // (n_nn, d)  += (atomic)
  void* grad_targets;      // (n_nn, R)  += (atomic)
  void* grad_ls;           // (b, ls_count) per-neighbourhood partials
  void* grad_noise;        // (b, k) per-neighbourhood diagonal cotangent
  // LOOCV objective (round 5; one response): upstream cotangent of y^T K^-1 y, (b).  When set, the solved right-hand
  // side is y itself (u = K^-1 y) and grad_mean enters as a factor: K-bar = gv a a^T - gm a u^T - gyk u u^T.
  const void* grad_yk = nullptr;
};
template <typename T> int launch_backward(const BackwardArgs&, hipStream_t);
template <typename T> int launch_backward_wave(const BackwardArgs&, hipStream_t);  // Isotropy, k + 2 <= 64: on the wave kernel's phases
// the backward on the forward kernel's own phases + the saved factor (mgp_backward_dlt.hip): the row-per-lane static
// shapes of either type (config 3: k = 30, d = 40; every cotangent, the feature cotangents included, any response
// count) and the fp64 dealt-triangle shapes (BASELINE config 4: k = 50, d = 8; hyper-parameter gradients of one
// response); MGP_EUNSUPPORTED for everything else
template <typename T> int launch_backward_fwd(const BackwardArgs&, hipStream_t);
int max_nn_count_backward(int elem_size);

// Exact k-NN scan on the matrix cores (mgp_knn.hip)
struct KnnArgs {
  const float* train;       // (n, d)
  const float* train_sqn;   // (n)  |x|^2
  const float* queries;     // (m, d)
  const float* query_sqn;   // (m)
  const int64_t* self_idx;  // (m) training row to exclude per query, or nullptr
  float* best_d;            // (m, k) in/out: current k best (Gram-form squared distances)
  int* best_i;              // (m, k) in/out
  int* overflow;            // (m) out: 1 = queue overflowed, recompute this query
  int64_t n, m;
  int64_t start;            // first training row to scan (earlier rows are already in the lists)
  int d, k;
};
int launch_knn_scan(const KnnArgs&, hipStream_t);
// split-bf16 pre-filter variant: packed rows [bf16 hi (KP) | bf16 lo (KP)], KP = 16 ceil(d / 16)
struct KnnPackedArgs {
  const float* train;           // (n, d) fp32 (exact re-measurement of survivors)
  const void* packed_train;     // (n, 2 KP) bf16
  const float* train_sqn;       // (n rounded up to 64, +inf past n)
  const float* queries;         // (m, d) fp32
  const void* packed_queries;   // (m, 2 KP) bf16
  const float* query_sqn;       // (m)
  const int64_t* self_idx;      // (m) or nullptr
  float* best_d;                // (m, k) in/out, exact squared distances
  int* best_i;                  // (m, k) in/out
  int* overflow;                // (m)
  int64_t n, m, start;
  int d, k;
  int layout = 0;               // 0: rows [hi(KP) | lo(KP)], KP = 16 ceil((d + 2) / 16); 1: d <= 8, rows [hi(8) | lo(8) | T(8)]
};
int launch_knn_scan_packed(const KnnPackedArgs&, hipStream_t);
// the selection steps around the scan (mgp_knn_select.hip): k smallest of every row, unordered; exact re-measurement +
// stable order + row-number mapping of the scan's winners
int launch_topk_rows(const float* x, int64_t rows, int cols, int64_t stride, int k, float* out_v, int* out_i, hipStream_t);
int launch_knn_finish(const float* queries, const float* train, int d, const int* cand, int64_t m, int k, const int64_t* perm,
                      int64_t* out_idx, float* out_dist, hipStream_t);

template <typename T> int launch_fused_generic(const FusedArgs&, hipStream_t);
template <typename T> int launch_solve_generic(const SolveArgs&, hipStream_t);
template <typename T> int launch_solve_wave(const SolveArgs&, hipStream_t);  // k + 1 + R <= 64, no coefficients
// register-resident wave-per-neighbourhood kernels; MGP_EUNSUPPORTED when the shape is not covered
template <typename T> int launch_fused_wave(const FusedArgs&, hipStream_t);
// run-time specialisation of the wave kernel (mgp_jit.hip)
int jit_wave_function(int es, int np, int k, int R, int d, bool packed, bool gram, hipFunction_t* fn, bool allow_compile = true,
                      bool gen64 = false, bool bwd = false);
int jit_wave_prepare(int es, int np, int k, int R, int d, bool packed, bool gram, bool gen64 = false, bool bwd = false);
int prepare_backward_fwd(int elem_size, int k, int d, int kernel_id);  // compile the backward instantiation of a shape into the disk cache (mgp_backward_dlt.hip)
// one instantiation of the wave kernel (mgp_fused_wave_launch.h; instantiated in mgp_fused_wave_inst_*.hip)
template <typename T, int NP, int KFIX, int RFIX, int DFIX, bool PIPED, bool COEFF, bool PACKED, bool GRAM, bool GEN64>
int launch_np_impl(const FusedArgs& a, hipStream_t stream);
int jit_mode();
int jit_loaded_count();
uint64_t jit_source_hash();
int64_t jit_min_batch();
int64_t jit_cached_min_batch();
int prepare_fused_wave(int elem_size, int d, int k, int R, int packed, int kernel_id);  // compile into the disk cache
// k <= 64 with up to 16 responses carried as right-hand-side columns (mgp_fused_rhs.hip)
template <typename T> int launch_fused_rhs(const FusedArgs&, hipStream_t);
int launch_fused_rhs_mf(const FusedArgs&, hipStream_t);  // fp32 prediction variant on the matrix cores' layout (mgp_fused_rhs_mf.hip)
// 64 < k + 1 + R <= 128, fp32: two waves per neighbourhood, rows in registers (mgp_fused_wide.hip)
template <typename T> int launch_fused_wide(const FusedArgs&, hipStream_t);
int launch_fused_wide64(const FusedArgs&, hipStream_t);  // fp64, two lanes per row
int max_nn_count(int elem_size, int R);
int describe_fused_wave(int elem_size, int d, int k, int R, int packed, char* buf, int len);
// the instantiation a posterior launcher actually put on the stream, per calling thread (mgp_last_kernel_name):
// what served a call depends on more than the shape (batch thresholds of the run-time compiler, its disk cache,
// alignment, the kernel's Gram-form eligibility), so tests and bench.py report THIS, not a description of the shape
void note_launch(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
// ... and the leaves of the reduction tree it walked (grid = 0: it walked none; mgp_last_loocv_geometry)
void note_tree_geometry(int grid, int nh);
// ... and its launch geometry: workgroups and dynamic LDS bytes per workgroup (mgp_last_launch_geometry)
void note_launch_geometry(int64_t grid, size_t lds_bytes);

template <typename T>
int launch_crosswise_diffs(const T*, const T*, int, const int64_t*, const int64_t*, int64_t, int, T*, hipStream_t);
template <typename T> int launch_pairwise_diffs(const T*, int, const int64_t*, int64_t, int, T*, hipStream_t);
template <typename T>
int launch_crosswise_dists(const T*, const T*, int, const int64_t*, const int64_t*, int64_t, int, int, T*,
                           hipStream_t);
template <typename T> int launch_pairwise_dists(const T*, int, const int64_t*, int64_t, int, int, T*, hipStream_t);
template <typename T> int launch_reduce_diffs(const T*, int64_t, int, const T*, int, T*, hipStream_t);
template <typename T> int launch_kernel_apply(const T*, int64_t, int, double, T*, hipStream_t);
template <typename T> int launch_perturb(const T*, int64_t, int, int, double, const T*, T*, hipStream_t);
template <typename T>
int launch_loss_sums(const T*, const T*, const T*, int64_t, const double*, double, double, double*, double*, hipStream_t);
template <typename T> int launch_column_sums(const T*, int64_t, int, double*, double*, hipStream_t);
// the reduction tree of mgp_loocv_tree.h over finished outputs, as three small launches (same functions, same bits
// as the walk inside the fused wave kernels)
template <typename T>
int launch_loocv_tree(const LoocvTree&, int grid, int nh, const T* mean, const T* var, const T* yk, const int64_t* batch_idx, int64_t b,
                      hipStream_t);
LoocvTree loocv_tree_layout(void* scratch, double* out, const void* resp, int64_t resp_stride, double huber_delta);
int reduce_scratch_doubles();
int allreduce_partials(double* partials_dev, int count, void* nccl_comm, hipStream_t stream);  // mgp_collective.hip
template <typename T> int launch_matern_gen(const T*, int64_t, double, double, T*, hipStream_t);
void matern_gen_constants_host(double nu, double* out7);
template <typename T> int launch_table_pack(const T*, const T*, int64_t, int, int, void*, int64_t, hipStream_t);
template <typename T>
int launch_fast_mean(const void*, const void*, int, const int64_t*, const int64_t*, int64_t, int, const void*,
                     const int64_t*, int, int, int, const void*, int, void*, hipStream_t);

#endif  // !__HIPCC_RTC__

}  // namespace mgp

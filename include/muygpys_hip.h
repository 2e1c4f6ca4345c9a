/*
 * muygpys_hip.h -- C ABI of the MI355X-native MuyGPyS local-GP hot path.
 *
 * The reference (LLNL/MuyGPyS 0.9.0) is pure Python and has no FFI; its
 * "interface for this path" is the set of free functions each backend module
 * exports (src/MuyGPyS/_src/<family>/{numpy,torch,jax,mpi}.py), resolved by
 * _collect_implementation (src/MuyGPyS/_src/util.py:9-32).  Every entry point
 * below names the reference function(s) it replaces; the Python binding a
 * maintainer would add is shown in INTEGRATION.md and implemented in
 * muygpys_amd/_src/<family>/hip.py via ctypes.
 *
 * Conventions
 *   - all data pointers are DEVICE pointers on the current HIP device, row-major,
 *     contiguous; indices are int64 (reference: itype = int64,
 *     _src/math/numpy.py:92-94); T is float (_f32) or double (_f64)
 *     (MUYGPYS_FTYPE, _src/config.py:254-261).
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  Entry
 *     points only enqueue work; they never synchronise, allocate or free.
 *   - return value: 0 on success, a negative MGP_E* code on bad arguments, or
 *     -(1000 + hipError_t) when the HIP runtime reports an error.  Nothing throws.
 *   - inputs are never written; outputs never alias inputs.
 *   - entry points are re-entrant: the library keeps no mutable state besides per-device
 *     launch geometry computed once under a mutex.
 *   - non-SPD neighbourhoods (non-positive Cholesky pivot; the reference's LU
 *     would raise numpy.linalg.LinAlgError only for an exactly singular matrix)
 *     produce NaN outputs and are counted in `*info` (device int32, may be NULL).
 */
#ifndef MUYGPYS_HIP_H
#define MUYGPYS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* kernel ids -- _src/gp/kernels/numpy.py:12-31 */
enum mgp_kernel {
  MGP_KERNEL_RBF = 0,        /* exp(-x/2),  x = F2 / l^2             :12-13 */
  MGP_KERNEL_MATERN_05 = 1,  /* exp(-r),    r = l2 / l               :16-17 */
  MGP_KERNEL_MATERN_15 = 2,  /* (1+s3 r) exp(-s3 r)                  :20-22 */
  MGP_KERNEL_MATERN_25 = 3,  /* (1+s5 r+5r^2/3) exp(-s5 r)           :25-27 */
  MGP_KERNEL_MATERN_INF = 4, /* exp(-r^2/2)                          :30-31 */
  MGP_KERNEL_MATERN_GEN = 5  /* 2^(1-nu)/Gamma(nu) x^nu K_nu(x), x = sqrt(2 nu) r :34-43; only the entry points
                                that take a smoothness accept it (mgp_posterior_gen_*, mgp_matern_gen_*) */
};

/* metric ids -- _src/gp/tensors/numpy.py:89-94, gp/deformation/metric.py:237-265 */
enum mgp_metric { MGP_METRIC_L2 = 0, MGP_METRIC_F2 = 1 };

/* noise model -- _src/gp/noise/numpy.py:9-14 (homoscedastic), :56-67 (heteroscedastic) */
enum mgp_noise_mode {
  MGP_NOISE_SCALAR = 0,  /* eps * I, eps = noise_scalar                               */
  MGP_NOISE_TABLE = 1,   /* diag(noise_dev[nn_idx[b, :]]): per-training-point table    */
  MGP_NOISE_BATCH = 2    /* diag(noise_dev[b, :]): already gathered (b,k) tensor       */
};

enum mgp_status {
  MGP_OK = 0,
  MGP_EINVAL = -1,       /* null pointer / negative size / unknown enum      */
  MGP_EUNSUPPORTED = -2, /* shape outside what the kernels were built for    */
  MGP_EHIP = -1000       /* -(1000 + hipError_t)                             */
};

const char* mgp_version(void);
/* Largest nn_count the fused / solve kernels accept for a given float width and
 * feature/response count (LDS-resident factorisation). */
int mgp_max_nn_count(int elem_size, int response_count);

/* ---------------------------------------------------------------------------
 * Fused hot path.  Replaces, in one launch and without materialising anything:
 *   T1 _crosswise_tensor + T2 _pairwise_tensor   _src/gp/tensors/numpy.py:47-69
 *   T3 _F2 / _l2                                 :89-94
 *   T4 target / noise gathers                    gp/muygps.py:474,543,545
 *   D1 Isotropy.__call__ / D2 Anisotropy.__call__ gp/deformation/isotropy.py:60-89,
 *                                                 anisotropy.py:43-70
 *   K1/K2 kernel functions                       _src/gp/kernels/numpy.py:12-31
 *   N1/N2 perturb                                _src/gp/noise/numpy.py:9-14,56-67
 *   S1 _muygps_posterior_mean                    _src/gp/muygps/numpy.py:17-41
 *   S2 _muygps_diagonal_variance (Kout = 1)      :44-67
 *   S3 _analytic_scale_optim_unnormalized terms  _src/optimize/scale/numpy.py:9-15
 *
 *   feat_q   (n_q, d)   query table (rows selected by batch_idx)
 *   feat_nn  (n_nn, d)  neighbour table (rows selected by nn_idx)
 *   batch_idx (b)       may be NULL = identity (row i of feat_q)
 *   nn_idx   (b, k)
 *   targets  (n_nn, R)  neighbour responses (rows selected by nn_idx)
 *   length_scale (ls_count) device; ls_count == 1 Isotropy, == d Anisotropy
 *   mean (b, R), var (b) [unscaled: 1 - c^T K^-1 c], ykinvy (b, R) or NULL
 *   [y_r^T K^-1 y_r per neighbourhood], info device int32 or NULL.
 * ------------------------------------------------------------------------- */
int mgp_posterior_f32(const float* feat_q, const float* feat_nn, int d,
                      const int64_t* batch_idx, const int64_t* nn_idx, int64_t b, int k,
                      const float* targets, int R,
                      int noise_mode, double noise_scalar, const float* noise_dev,
                      int kernel_id, int metric_id, const float* length_scale, int ls_count,
                      float* mean, float* var, float* ykinvy, int* info, void* stream);
int mgp_posterior_f64(const double* feat_q, const double* feat_nn, int d,
                      const int64_t* batch_idx, const int64_t* nn_idx, int64_t b, int k,
                      const double* targets, int R,
                      int noise_mode, double noise_scalar, const double* noise_dev,
                      int kernel_id, int metric_id, const double* length_scale, int ls_count,
                      double* mean, double* var, double* ykinvy, int* info, void* stream);

/* The same call served by one named kernel family instead of the dispatcher's choice (parity
 * tests compare the families with each other; same arguments, same results):
 *   _generic: one workgroup per neighbourhood, system in LDS (any k up to mgp_max_nn_count)
 *   _rhs:     one wave per neighbourhood, responses as right-hand-side columns (k <= 64, R <= 16);
 *             MGP_EUNSUPPORTED outside that. */
int mgp_posterior_generic_f32(const float* feat_q, const float* feat_nn, int d,
                              const int64_t* batch_idx, const int64_t* nn_idx, int64_t b, int k,
                              const float* targets, int R,
                              int noise_mode, double noise_scalar, const float* noise_dev,
                              int kernel_id, int metric_id, const float* length_scale, int ls_count,
                              float* mean, float* var, float* ykinvy, int* info, void* stream);
int mgp_posterior_generic_f64(const double* feat_q, const double* feat_nn, int d,
                              const int64_t* batch_idx, const int64_t* nn_idx, int64_t b, int k,
                              const double* targets, int R,
                              int noise_mode, double noise_scalar, const double* noise_dev,
                              int kernel_id, int metric_id, const double* length_scale, int ls_count,
                              double* mean, double* var, double* ykinvy, int* info, void* stream);
int mgp_posterior_rhs_f32(const float* feat_q, const float* feat_nn, int d,
                          const int64_t* batch_idx, const int64_t* nn_idx, int64_t b, int k,
                          const float* targets, int R,
                          int noise_mode, double noise_scalar, const float* noise_dev,
                          int kernel_id, int metric_id, const float* length_scale, int ls_count,
                          float* mean, float* var, float* ykinvy, int* info, void* stream);
int mgp_posterior_rhs_f64(const double* feat_q, const double* feat_nn, int d,
                          const int64_t* batch_idx, const int64_t* nn_idx, int64_t b, int k,
                          const double* targets, int R,
                          int noise_mode, double noise_scalar, const double* noise_dev,
                          int kernel_id, int metric_id, const double* length_scale, int ls_count,
                          double* mean, double* var, double* ykinvy, int* info, void* stream);

/* mgp_posterior_* with the neighbour responses ALREADY GATHERED: nn_targets (b, k, R) =
 * targets[nn_idx] -- the tensor the reference's MuyGPS.make_predict_tensors / make_train_tensors
 * return (gp/muygps.py:474,543,545) and pass to _muygps_posterior_mean / _analytic_scale_optim.
 * Lets the reference's own functor layer, which gathers the responses itself, reach the fused
 * launch through the lazy family functions.  Everything else as mgp_posterior_*. */
int mgp_posterior_gathered_f32(const float* feat_q, const float* feat_nn, int d,
                               const int64_t* batch_idx, const int64_t* nn_idx, int64_t b, int k,
                               const float* nn_targets, int R,
                               int noise_mode, double noise_scalar, const float* noise_dev,
                               int kernel_id, int metric_id, const float* length_scale, int ls_count,
                               float* mean, float* var, float* ykinvy, int* info, void* stream);
int mgp_posterior_gathered_f64(const double* feat_q, const double* feat_nn, int d,
                               const int64_t* batch_idx, const int64_t* nn_idx, int64_t b, int k,
                               const double* nn_targets, int R,
                               int noise_mode, double noise_scalar, const double* noise_dev,
                               int kernel_id, int metric_id, const double* length_scale, int ls_count,
                               double* mean, double* var, double* ykinvy, int* info, void* stream);

/* Name of the kernel instantiation that serves mgp_posterior_* (path 0), mgp_posterior_generic_*
 * (path 1) or mgp_posterior_rhs_* (path 2) for a shape, for 16-byte aligned tables; packed != 0:
 * mgp_posterior_packed_*.  A pure function of its arguments (benchmarks name the kernel whose
 * duration they report).  Writes a NUL-terminated string into buf. */
int mgp_posterior_kernel_name(int elem_size, int d, int k, int R, int packed, int path, char* buf, int len);

/* Name of the kernel instantiation the calling thread's most recent mgp_posterior_* / mgp_loocv_* call actually
 * launched ("" before the first).  Unlike mgp_posterior_kernel_name this reflects everything the dispatch looked at:
 * batch thresholds and disk cache of the run-time compiler, table alignment, Gram-form eligibility of the
 * covariance function.  Tests use it to prove that the kernel a benchmark line times is one the oracle has checked. */
int mgp_last_kernel_name(char* buf, int len);

/* ---------------------------------------------------------------------------
 * Prepared tables.  The gathers of T1/T2/T4 (_src/gp/tensors/numpy.py:47-69,
 * gp/muygps.py:474,543,545) read, per neighbour, a feature row and a 4/8-byte
 * response from two separate arrays; at d = 40 fp32 that is two 128-byte lines
 * for the row plus one whole line for the response.  A prepared table stores
 *     row i = [ features (d) | responses (R) | zero pad ]   at stride mgp_packed_row_bytes(d, R, s)
 * (a multiple of 64 bytes that always covers d * s + max(16, R * s): the 16-byte slot behind the
 * features is read with every row, also of a table packed with R = 0), so row and responses
 * arrive together.  The tables
 * are constant across all objective evaluations of a hyper-parameter search
 * (the reference rebuilds its difference tensors never, its kernels every
 * evaluation: optimize/objective.py:95-103), so the table is packed once.
 *   mgp_table_pack_*: features (n, d), targets (n, R) or NULL (R = 0 / zeros: a query
 *       table) -> packed (n * stride bytes).
 *   mgp_posterior_packed_*: mgp_posterior_* reading two prepared tables (query
 *       table: rows selected by batch_idx; neighbour table: rows and responses selected
 *       by nn_idx; the two may be the same table).  Needs d * sizeof(T) % 16 == 0,
 *       R * sizeof(T) <= 16 and k + 1 + R <= 64; otherwise MGP_EUNSUPPORTED and the caller
 *       uses mgp_posterior_* on the plain tables.  Results are identical to
 *       mgp_posterior_* (same kernels, same arithmetic).
 * ------------------------------------------------------------------------- */
int64_t mgp_packed_row_bytes(int d, int R, int elem_size);

/* ---------------------------------------------------------------------------
 * General-smoothness Matern on the fused path (K3: _matern_gen_fn, _src/gp/kernels/numpy.py:34-43, selected
 * by gp/kernels/matern.py:61-81 whenever the smoothness is not one of 1/2, 3/2, 5/2, inf -- e.g. while it
 * is being optimised).  One entry point for every table form: plain tables (feat_q / feat_nn / targets;
 * packed_* NULL) or prepared tables (packed_q / packed_nn with strides; feat_* NULL), responses from the
 * table or already gathered (targets_batch != 0: targets = (b, k, R)).  The covariance is evaluated inside
 * the fused kernel (trapezoidal rule on the integral representation of K_nu, csrc/mgp_wave_common.h), so a
 * free-smoothness objective evaluation is ONE launch with nothing materialised.  fp32 tables,
 * k + 1 + R <= 32 or a shape the static kernels serve, nu <= 30; MGP_EUNSUPPORTED otherwise (the caller
 * falls back to mgp_*_dists_* + mgp_matern_gen_* + mgp_solve_*).
 * ------------------------------------------------------------------------- */
int mgp_posterior_gen_f32(const float* feat_q, const float* feat_nn, const void* packed_q, int64_t q_stride_bytes,
                          const void* packed_nn, int64_t nn_stride_bytes, int d,
                          const int64_t* batch_idx, const int64_t* nn_idx, int64_t b, int k,
                          const float* targets, int R, int targets_batch,
                          int noise_mode, double noise_scalar, const float* noise_dev,
                          double smoothness, int metric_id, const float* length_scale, int ls_count,
                          float* mean, float* var, float* ykinvy, int* info, void* stream);
int mgp_posterior_gen_f64(const double* feat_q, const double* feat_nn, const void* packed_q, int64_t q_stride_bytes,
                          const void* packed_nn, int64_t nn_stride_bytes, int d,
                          const int64_t* batch_idx, const int64_t* nn_idx, int64_t b, int k,
                          const double* targets, int R, int targets_batch,
                          int noise_mode, double noise_scalar, const double* noise_dev,
                          double smoothness, int metric_id, const double* length_scale, int ls_count,
                          double* mean, double* var, double* ykinvy, int* info, void* stream);

/* ---------------------------------------------------------------------------
 * Run-time specialisation.  mgp_posterior_* / mgp_loocv_* serve a shape (k, R, d) with a kernel whose
 * loops are compiled for exactly that shape: built into the library for the BASELINE shapes, compiled
 * on first use (hiprtc, ~1 s, cached on disk next to the library) for any other shape with
 * k + 1 + R <= 64, 16-byte rows and d <= 64 (fp32) / 32 (fp64) -- when the call has at least
 * MUYGPYS_HIP_JIT_MIN_BATCH neighbourhoods (default 65536); a shape whose kernel is loaded or in the disk
 * cache already (the build compiles 78 common ones) is served by it from MUYGPYS_HIP_JIT_CACHED_MIN_BATCH
 * neighbourhoods on (default 4096), without ever compiling.  MUYGPYS_HIP_JIT=0 turns it off (the
 * run-time-shape kernels serve every call), =force applies it to every call.  Results do not depend
 * on which kernel served a call beyond the rounding of a different summation order.
 *   mgp_jit_prepare: compile one shape into the disk cache ahead of time (no GPU needed);
 *                    MGP_OK, or MGP_EUNSUPPORTED (shape outside the static kernels, or no hiprtc).
 *   mgp_jit_prepare_backward: the same for the backward instantiation of the forward kernel (round 6) that serves
 *                    mgp_posterior_backward_* / mgp_loocv_backward_*: row per lane -- either type with
 *                    5 <= nn_count + 2 <= 32, fp32 up to 64; every cotangent, feature cotangents included, any response
 *                    count; rows of whole 16-byte groups, up to 64 features (fp32: 128) -- or, fp64 with
 *                    33 <= nn_count + 2 <= 64, on the dealt triangle (no feature cotangents).
 *   mgp_jit_mode:    0 off, 1 automatic, 2 forced.
 *   mgp_jit_loaded_count: run-time compiled kernels loaded in this process so far.
 *   mgp_jit_source_hash:  the 16 hex digits every cache file name of THIS build ends in (kernel sources, compile
 *                         options, hiprtc version); files with another suffix belong to other builds.
 * ------------------------------------------------------------------------- */
int mgp_jit_prepare(int elem_size, int k, int R, int d, int packed, int kernel_id);
int mgp_jit_prepare_backward(int elem_size, int k, int d, int kernel_id);
int mgp_jit_mode(void);
int mgp_jit_loaded_count(void);
int mgp_jit_source_hash(char* buf, int len);
int mgp_table_pack_f32(const float* features, const float* targets, int64_t n, int d, int R, void* packed,
                       int64_t stride_bytes, void* stream);
int mgp_table_pack_f64(const double* features, const double* targets, int64_t n, int d, int R, void* packed,
                       int64_t stride_bytes, void* stream);
int mgp_posterior_packed_f32(const void* packed_q, int64_t q_stride_bytes, const void* packed_nn,
                             int64_t nn_stride_bytes, int d,
                             const int64_t* batch_idx, const int64_t* nn_idx, int64_t b, int k, int R,
                             int noise_mode, double noise_scalar, const float* noise_dev,
                             int kernel_id, int metric_id, const float* length_scale, int ls_count,
                             float* mean, float* var, float* ykinvy, int* info, void* stream);
int mgp_posterior_packed_f64(const void* packed_q, int64_t q_stride_bytes, const void* packed_nn,
                             int64_t nn_stride_bytes, int d,
                             const int64_t* batch_idx, const int64_t* nn_idx, int64_t b, int k, int R,
                             int noise_mode, double noise_scalar, const double* noise_dev,
                             int kernel_id, int metric_id, const double* length_scale, int ls_count,
                             double* mean, double* var, double* ykinvy, int* info, void* stream);
/* The same with the neighbour responses ALREADY GATHERED, nn_targets (b, k, R) = targets[nn_idx]: what the
 * reference's functor layer hands to posterior_mean (MuyGPS.make_predict_tensors gathers them itself,
 * gp/muygps.py:474,543).  The feature rows still come from the prepared tables (two cache lines per
 * neighbour); the table may be packed without responses (R = 0).  Any R the kernels support. */
int mgp_posterior_packed_gathered_f32(const void* packed_q, int64_t q_stride_bytes, const void* packed_nn,
                                      int64_t nn_stride_bytes, int d,
                                      const int64_t* batch_idx, const int64_t* nn_idx, int64_t b, int k,
                                      const float* nn_targets, int R,
                                      int noise_mode, double noise_scalar, const float* noise_dev,
                                      int kernel_id, int metric_id, const float* length_scale, int ls_count,
                                      float* mean, float* var, float* ykinvy, int* info, void* stream);
int mgp_posterior_packed_gathered_f64(const void* packed_q, int64_t q_stride_bytes, const void* packed_nn,
                                      int64_t nn_stride_bytes, int d,
                                      const int64_t* batch_idx, const int64_t* nn_idx, int64_t b, int k,
                                      const double* nn_targets, int R,
                                      int noise_mode, double noise_scalar, const double* noise_dev,
                                      int kernel_id, int metric_id, const double* length_scale, int ls_count,
                                      double* mean, double* var, double* ykinvy, int* info, void* stream);

/* ---------------------------------------------------------------------------
 * One LOOCV objective evaluation of a shard in one call: mgp_posterior_* with
 * the training table on both sides (gp/muygps.py:406-551 under
 * optimize/objective.py:95-103), followed -- on the same stream, without the
 * host touching mean / var -- by the fp64 partial sums the losses and the
 * analytic scale are made of (optimize/loss.py:159-168, _src/optimize/loss/
 * numpy.py:22-61, _src/optimize/scale/numpy.py:11-18):
 *     partials[6] = { sum r^2/v, sum log v, sum r^2, b, sum pseudo-Huber(r; huber_delta),
 *                     sum y^T K^-1 y },   r = mean - y[batch row]
 * from which sigma^2 = partials[5] / (b k) and lool = partials[0] / sigma^2 +
 * partials[1] + b log sigma^2 (mse = partials[2] / b) follow on the host -- and,
 * sharded, after ONE all-reduce of the six numbers (_src/optimize/loss/mpi.py:
 * 57-70, scale/mpi.py:35-36).  One response (the losses' domain).  mean / var /
 * ykinvy (b each) are written as by mgp_posterior_*.
 *
 * ONE launch (round 5): for every shape the register-resident wave kernels serve
 * (k + 2 <= 64) the fused kernel walks the reduction tree of the sums itself
 * (csrc/mgp_loocv_tree.h).  A leaf of the tree is what ONE workgroup of the
 * persistent launch evaluates: out of tasks, it reads its own outputs back and
 * reduces them; the workgroup that completes a block of 64 leaves reduces those,
 * the last one writes partials[] (agent-scope tickets, write-through hand-off,
 * nobody waits; nothing of it inside the task loop).  Behind the other kernel
 * families the SAME tree is walked by three small launches (mgp_loocv_tree_*,
 * also callable on the outputs of any mgp_posterior_*): equal sums bit for bit
 * for equal leaves (mgp_last_loocv_geometry).  The last bits of the sums depend
 * on the leaves, i.e. on the kernel and the device -- not on timing.
 *
 * partials may be DEVICE memory or PINNED HOST memory mapped into the device's
 * address space: the count, partials[3], is written last, behind a drain of the
 * other five, so a host that zeroed it before the call and polls it reading b
 * has all six without a stream synchronisation or a copy.
 *
 * scratch: mgp_loocv_scratch_bytes() bytes of device memory, 128-byte aligned,
 * whose first mgp_loocv_scratch_zero_bytes() bytes are ZERO when the call
 * starts; every call leaves them zero again, so one buffer serves any number of
 * consecutive calls on one stream (zero it once; again after a failed launch).
 * mgp_loocv_packed_* reads one prepared table (MGP_EUNSUPPORTED where
 * mgp_posterior_packed_* is).
 * ------------------------------------------------------------------------- */
int64_t mgp_loocv_scratch_bytes(void);
int64_t mgp_loocv_scratch_zero_bytes(void);
/* the leaves of the tree the calling thread's most recent mgp_loocv_* call walked inside its fused launch
 * (persistent workgroups, neighbourhoods per task); grid = 0: it walked none (the three-launch walk on the
 * canonical leaves served it) */
int mgp_last_loocv_geometry(int* grid, int* nh);
/* How the one-launch evaluation hands sums from workgroup to workgroup (process-wide; initial value from the
 * environment variable MUYGPYS_HIP_LOOCV_TREE = tickets | fenced | three_launch, default tickets):
 *   0 tickets       write-through stores, drained vmcnt, relaxed agent-scope ticket, sc1 loads (no fence)
 *   1 fenced        release fence / ACQ_REL ticket / acquire fence: what the memory model guarantees anywhere
 *   2 three_launch  the fused kernel does not walk; three kernels walk the SAME leaves behind it
 * All three give the same sums bit for bit (same values, same order).  The Python host runs a start-up self-check
 * of the default form against three_launch and falls back to it, for the rest of the process, if they ever differ
 * (muygpys_amd/_lib.py: loocv_tree_selfcheck).  Replaces the host-side reductions of the reference's
 * _src/optimize/loss/mpi.py:57. */
int mgp_loocv_tree_mode_get(void);
int mgp_loocv_tree_mode_set(int mode);
/* launch geometry of the calling thread's most recent wave-kernel / matrix-core-layout launch: workgroups (the
 * persistent grid) and dynamic LDS bytes per workgroup -- what profiles/ quote next to the compiler's register
 * record (lib/kernel_resources.json); diagnostic */
int mgp_last_launch_geometry(int64_t* workgroups, int* lds_bytes);
int mgp_loocv_f32(const float* features, int d, const int64_t* batch_idx, const int64_t* nn_idx, int64_t b, int k,
                  const float* targets, int noise_mode, double noise_scalar, const float* noise_dev,
                  int kernel_id, int metric_id, const float* length_scale, int ls_count,
                  float* mean, float* var, float* ykinvy, int* info,
                  double huber_delta, double* partials, void* scratch, void* stream);
int mgp_loocv_f64(const double* features, int d, const int64_t* batch_idx, const int64_t* nn_idx, int64_t b, int k,
                  const double* targets, int noise_mode, double noise_scalar, const double* noise_dev,
                  int kernel_id, int metric_id, const double* length_scale, int ls_count,
                  double* mean, double* var, double* ykinvy, int* info,
                  double huber_delta, double* partials, void* scratch, void* stream);
int mgp_loocv_packed_f32(const void* packed, int64_t stride_bytes, int d, const int64_t* batch_idx,
                         const int64_t* nn_idx, int64_t b, int k,
                         int noise_mode, double noise_scalar, const float* noise_dev,
                         int kernel_id, int metric_id, const float* length_scale, int ls_count,
                         float* mean, float* var, float* ykinvy, int* info,
                         double huber_delta, double* partials, void* scratch, void* stream);
int mgp_loocv_packed_f64(const void* packed, int64_t stride_bytes, int d, const int64_t* batch_idx,
                         const int64_t* nn_idx, int64_t b, int k,
                         int noise_mode, double noise_scalar, const double* noise_dev,
                         int kernel_id, int metric_id, const double* length_scale, int ls_count,
                         double* mean, double* var, double* ykinvy, int* info,
                         double huber_delta, double* partials, void* scratch, void* stream);

/* The reduction tree alone, over finished outputs (mean / var / ykinvy of b neighbourhoods as written by any
 * mgp_posterior_* call with one response): partials[6] as above.  resp + row * resp_stride_bytes is the response
 * of table row `row` (the response tensor: stride sizeof(T); a prepared table: its row stride, resp = table +
 * d * sizeof(T)); batch_idx may be NULL (row = neighbourhood index).  (grid, nh): the leaves -- what
 * mgp_last_loocv_geometry reported, for the bits of that call's own walk; 0, 0: the canonical ones.  Same scratch
 * as mgp_loocv_* (its counters are not used here: no zeroing needed).
 * Reference: _src/optimize/loss/numpy.py:22-72, scale/numpy.py:9-15. */
int mgp_loocv_tree_f32(const float* mean, const float* var, const float* ykinvy, const void* resp,
                       int64_t resp_stride_bytes, const int64_t* batch_idx, int64_t b, double huber_delta,
                       int grid, int nh, double* partials, void* scratch, void* stream);
int mgp_loocv_tree_f64(const double* mean, const double* var, const double* ykinvy, const void* resp,
                       int64_t resp_stride_bytes, const int64_t* batch_idx, int64_t b, double huber_delta,
                       int grid, int nh, double* partials, void* scratch, void* stream);

/* Fused coefficient precompute of the fast posterior mean: coeffs (b, k) = (K_b + eps)^-1 y_b
 * for the neighbourhoods nn_idx (b, k) of one table (gather -> distances -> kernel -> nugget ->
 * LDL^T -> back-substitution in one launch).  Replaces _muygps_fast_posterior_mean_precompute
 * on materialised tensors (_src/gp/muygps/numpy.py:88-95, called from
 * MuyGPS.fast_coefficients, gp/muygps.py:261-298, in examples/fast_posterior_mean.py:373-386).
 * One response column (targets (n)), k <= 62; anything else returns
 * MGP_EUNSUPPORTED and the caller uses the mgp_pairwise_dists_*, mgp_kernel_apply_*, mgp_perturb_*
 * and mgp_solve_* entry points (the last with its coeffs output). */
int mgp_fast_coefficients_f32(const float* feat, int d, const int64_t* nn_idx, int64_t b, int k,
                              const float* targets, int noise_mode, double noise_scalar, const float* noise_dev,
                              int kernel_id, int metric_id, const float* length_scale, int ls_count,
                              float* coeffs, int* info, void* stream);
int mgp_fast_coefficients_f64(const double* feat, int d, const int64_t* nn_idx, int64_t b, int k,
                              const double* targets, int noise_mode, double noise_scalar, const double* noise_dev,
                              int kernel_id, int metric_id, const double* length_scale, int ls_count,
                              double* coeffs, int* info, void* stream);

/* The selection steps around the scans (round 6; torch's topk / argsort did them until then: a quarter of a search's time).
 * Reference: scikit-learn's brute-force kneighbors behind NN_Wrapper.get_nns / get_batch_nns
 * (src/MuyGPyS/neighbors.py:129-211): exact neighbours in ascending order of distance.
 *   mgp_topk_rows_f32   the k smallest entries of every row of x (rows, cols <= 4096; row_stride elements apart):
 *                       values and column numbers, UNORDERED (the scans' initial lists); ties at the k-th value: any of them.
 *   mgp_knn_finish_f32  candidates (m, k) int32 rows of `train`: squared distances to the queries re-measured in the
 *                       difference form, each query's k candidates put in ascending order (ties: position in the
 *                       list), row numbers through row_map (int64, or NULL) -> out_idx (m, k) int64, out_dist (m, k).
 *                       d % 4 == 0, 16-byte aligned rows, k <= 64; MGP_EUNSUPPORTED otherwise. */
int mgp_topk_rows_f32(const float* x, int64_t rows, int cols, int64_t row_stride, int k, float* out_values,
                      int32_t* out_cols, void* stream);
int mgp_knn_finish_f32(const float* queries, const float* train, int d, const int32_t* candidates, int64_t m, int k,
                       const int64_t* row_map, int64_t* out_idx, float* out_dist, void* stream);

/* ---------------------------------------------------------------------------
 * Exact k-nearest-neighbour scan (the step upstream of the hot path).  Replaces
 * the exact search behind NN_Wrapper (src/MuyGPyS/neighbors.py:106-107 builds a
 * scikit-learn NearestNeighbors, :129-167 get_nns / :169-211 get_batch_nns query
 * it; squared-l2 convention :246-250).
 *
 *   train (n, d), queries (m, d): fp32, rows 16-byte aligned, d % 4 == 0, d <= 64
 *   train_sqn (n rounded up to a multiple of 64; +inf past n), query_sqn (m):
 *       squared norms of the rows; train_sqn + start must be 16-byte aligned
 *       (start % 4 == 0)
 *   self_idx (m) or NULL: training row each query must not return (batch queries
 *       drop the self match, neighbors.py:207-211)
 *   best_d / best_i (m, k), k <= 64: IN: an exact k-best list over training rows
 *       [0, start) (Gram-form squared distances |q|^2+|x|^2-2q.x, any order);
 *       OUT: the exact k-best over all n rows (unordered; the caller re-measures
 *       the winners in difference form and sorts them).
 *   overflow (m), zero-filled by the caller: set to 1 for a query whose candidate
 *       queue overflowed (adversarial row order); its list is then incomplete
 *       and the caller recomputes that query on its dense path.
 *   Returns MGP_EUNSUPPORTED for shapes outside the above (caller falls back).
 * ------------------------------------------------------------------------- */
int mgp_knn_scan_f32(const float* train, const float* train_sqn, int64_t n, int d,
                     const float* queries, const float* query_sqn, const int64_t* self_idx, int64_t m,
                     int k, int64_t start, float* best_d, int32_t* best_i, int32_t* overflow,
                     void* stream);

/* Same search with a split-bf16 pre-filter on the matrix cores (3 bf16 MFMA chains
 * hi.hi + hi.lo + lo.hi, error < 2^-14 |q||x|, folded into the threshold) and an
 * exact fp32 difference-form re-measurement of every survivor: exact results,
 * ~4.4 x fewer matrix-pipe cycles than mgp_knn_scan_f32.
 *   packed_train (n, 2 KP) / packed_queries (m, 2 KP), KP = 16 ceil((d + 2) / 16): per row
 *       [bf16(x) zero-padded to KP | bf16(x - float(bf16(x))) zero-padded to KP]; slots KP - 2, KP - 1 of each part hold
 *       the split of (c, 1) in a table row, c = -|x|^2/2 + 2^-14 QMAX |x| (raised by its own split error), and of
 *       (1, 0) in a query row (the kernel writes -thr into the second one)
 *   best_d IN/OUT: exact squared distances (difference form); other arguments and
 *   the overflow contract as for mgp_knn_scan_f32. */
int mgp_knn_scan_bf16x3(const float* train, const void* packed_train, const float* train_sqn, int64_t n, int d,
                        const float* queries, const void* packed_queries, const float* query_sqn,
                        const int64_t* self_idx, int64_t m, int k, int64_t start,
                        float* best_d, int32_t* best_i, int32_t* overflow, void* stream);
/* The same for d <= 8 (BASELINE config 4: d = 8) with TWO chains per block: K = 16 takes eight slots from each half of
 * the wave, so [q_hi | q_lo] x [x_hi | x_hi] and [q_hi | T_q] x [x_lo | T_x] give hi.hi + lo.hi + hi.lo + the threshold
 * terms in two matrix instructions (round 5).
 *   packed_train (n, 24) / packed_queries (m, 24) bf16: per row [hi(8) | lo(8) | T(8)], features zero-padded to 8;
 *       T of a table row = [hi(c), lo(c), 1, 1, 0, 0, 0, 0] with c = -|x|^2/2 + 2^-14 QMAX |x| (raised by its own
 *       split error), T of a query row = [1, 1, 0, 0, 0, 0, 0, 0] (the kernel fills slots 2, 3 with -thr).
 *   Everything else as for mgp_knn_scan_bf16x3; MGP_EUNSUPPORTED for d > 8. */
int mgp_knn_scan_bf16x2_d8(const float* train, const void* packed_train, const float* train_sqn, int64_t n, int d,
                           const float* queries, const void* packed_queries, const float* query_sqn,
                           const int64_t* self_idx, int64_t m, int k, int64_t start,
                           float* best_d, int32_t* best_i, int32_t* overflow, void* stream);

/* ---------------------------------------------------------------------------
 * Backward pass of the fused hot path (vector-Jacobian product).  Replaces what
 * torch autograd derives for the reference's torch backend when a deep-kernel
 * model trains through MuyGPs_layer (torch/muygps_layer.py:129-164,
 * torch/multivariate_muygps_layer.py:117-154; loss.sum().backward() at
 * examples/muygps_torch.py:425-437): the cotangents of
 *   mean (b,R) = Kcross K^-1 Y   and   var (b) = 1 - Kcross K^-1 Kcross
 * with respect to the feature tables, the neighbour responses, the length
 * scale(s) and the noise diagonal, without materialising any (b,k,k,d) tensor.
 *
 *   forward arguments: as mgp_posterior_*.
 *   grad_mean (b, R) / grad_var (b): upstream cotangents; either may be NULL (= 0).
 *   grad_feat_q  (n_q, d)  and  grad_feat_nn (n_nn, d): ACCUMULATED into with
 *       atomic adds (caller zero-fills; the two may be the same buffer when the
 *       query and neighbour tables are the same tensor, as in LOOCV training).
 *   grad_targets (n_nn, R): accumulated likewise.
 *   grad_ls (b, ls_count): per-neighbourhood partials of d/d length_scale
 *       (sum over b on the caller's side, mgp_column_sums_*).
 *   grad_noise (b, k): cotangent of each neighbourhood's noise diagonal (sum
 *       everything for a homoscedastic eps; scatter by nn_idx for a table).
 *   Any grad_* output may be NULL (skipped).  Pairs at zero distance contribute
 *   no distance gradient (torch.norm's subgradient, _src/gp/tensors/torch.py:85-86).
 *   Non-SPD neighbourhoods leave their cotangents untouched and count in *info.
 * ------------------------------------------------------------------------- */
int mgp_posterior_backward_f32(const float* feat_q, const float* feat_nn, int d,
                               const int64_t* batch_idx, const int64_t* nn_idx, int64_t b, int k,
                               const float* targets, int R,
                               int noise_mode, double noise_scalar, const float* noise_dev,
                               int kernel_id, int metric_id, const float* length_scale, int ls_count,
                               const float* grad_mean, const float* grad_var,
                               float* grad_feat_q, float* grad_feat_nn, float* grad_targets,
                               float* grad_ls, float* grad_noise, int* info, void* stream);
int mgp_posterior_backward_f64(const double* feat_q, const double* feat_nn, int d,
                               const int64_t* batch_idx, const int64_t* nn_idx, int64_t b, int k,
                               const double* targets, int R,
                               int noise_mode, double noise_scalar, const double* noise_dev,
                               int kernel_id, int metric_id, const double* length_scale, int ls_count,
                               const double* grad_mean, const double* grad_var,
                               double* grad_feat_q, double* grad_feat_nn, double* grad_targets,
                               double* grad_ls, double* grad_noise, int* info, void* stream);
/* Gradient of a LOOCV objective with respect to the length scale(s) and the noise diagonal (round 5): the
 * vector-Jacobian product above with the training table on both sides, one response, and a third upstream cotangent
 * -- grad_ykinvy (b), of y^T K^-1 y, through which the analytic sigma^2 enters every LOOCV loss
 * (gp/hyperparameter/scale.py:205-217 inside optimize/loss.py:159-176):
 *     K-bar = gv a a^T - gm a u^T - gyk u u^T,   a = K^-1 c,  u = K^-1 y.
 * grad_mean / grad_var / grad_ykinvy (b each) are d loss / d (mean_i, var_i, y_i^T K_i^-1 y_i) of the caller's loss;
 * grad_ls (b, ls_count) and grad_noise (b, k) as above (either may be NULL).  What the reference obtains from
 * torch autograd over its torch backend (torch/muygps_layer.py:129-164); here it gives scipy's L-BFGS-B an analytic
 * gradient instead of p + 1 finite-difference evaluations per iteration (_src/optimize/chassis/numpy.py:57-81). */
int mgp_loocv_backward_f32(const float* features, int d, const int64_t* batch_idx, const int64_t* nn_idx, int64_t b, int k,
                           const float* targets, int noise_mode, double noise_scalar, const float* noise_dev,
                           int kernel_id, int metric_id, const float* length_scale, int ls_count,
                           const float* grad_mean, const float* grad_var, const float* grad_ykinvy,
                           float* grad_ls, float* grad_noise, int* info, void* stream);
int mgp_loocv_backward_f64(const double* features, int d, const int64_t* batch_idx, const int64_t* nn_idx, int64_t b, int k,
                           const double* targets, int noise_mode, double noise_scalar, const double* noise_dev,
                           int kernel_id, int metric_id, const double* length_scale, int ls_count,
                           const double* grad_mean, const double* grad_var, const double* grad_ykinvy,
                           double* grad_ls, double* grad_noise, int* info, void* stream);
/* Largest nn_count the backward kernel accepts (two LDS-resident k x k triangles). */
int mgp_max_nn_count_backward(int elem_size);

/* ---------------------------------------------------------------------------
 * Materialising per-function kernels (API parity with the backend modules).
 * ------------------------------------------------------------------------- */

/* T1 _crosswise_tensor, _src/gp/tensors/numpy.py:47-58: out (b,k,d) = q - x_nn */
int mgp_crosswise_diffs_f32(const float* feat_q, const float* feat_nn, int d, const int64_t* batch_idx,
                            const int64_t* nn_idx, int64_t b, int k, float* out, void* stream);
int mgp_crosswise_diffs_f64(const double* feat_q, const double* feat_nn, int d, const int64_t* batch_idx,
                            const int64_t* nn_idx, int64_t b, int k, double* out, void* stream);
/* T2 _pairwise_tensor, :61-69: out (b,k,k,d)[b,i,j,:] = x[nn[b,i]] - x[nn[b,j]] */
int mgp_pairwise_diffs_f32(const float* feat, int d, const int64_t* nn_idx, int64_t b, int k, float* out,
                           void* stream);
int mgp_pairwise_diffs_f64(const double* feat, int d, const int64_t* nn_idx, int64_t b, int k, double* out,
                           void* stream);
/* T1+T3 fused: out (b,k) = metric(q - x_nn); T2+T3 fused: out (b,k,k).  What
 * Isotropy.crosswise_tensor / pairwise_tensor return (isotropy.py:92-161). */
int mgp_crosswise_dists_f32(const float* feat_q, const float* feat_nn, int d, const int64_t* batch_idx,
                            const int64_t* nn_idx, int64_t b, int k, int metric_id, float* out, void* stream);
int mgp_crosswise_dists_f64(const double* feat_q, const double* feat_nn, int d, const int64_t* batch_idx,
                            const int64_t* nn_idx, int64_t b, int k, int metric_id, double* out, void* stream);
int mgp_pairwise_dists_f32(const float* feat, int d, const int64_t* nn_idx, int64_t b, int k, int metric_id,
                           float* out, void* stream);
int mgp_pairwise_dists_f64(const double* feat, int d, const int64_t* nn_idx, int64_t b, int k, int metric_id,
                           double* out, void* stream);
/* T3 _F2/_l2 (:89-94) with the optional Anisotropy division (anisotropy.py:70):
 * out[n] = metric(diffs[n,:] / length_scale[:]); length_scale NULL = no division. */
int mgp_reduce_diffs_f32(const float* diffs, int64_t n, int d, const float* length_scale, int metric_id,
                         float* out, void* stream);
int mgp_reduce_diffs_f64(const double* diffs, int64_t n, int d, const double* length_scale, int metric_id,
                         double* out, void* stream);
/* D1 + K1/K2: out[i] = kernel(in[i] * in_scale).  in_scale = 1/l (l2) or 1/l^2 (F2)
 * (metric.py:241,264); _src/gp/kernels/numpy.py:12-31. */
int mgp_kernel_apply_f32(const float* in, int64_t n, int kernel_id, double in_scale, float* out, void* stream);
int mgp_kernel_apply_f64(const double* in, int64_t n, int kernel_id, double in_scale, double* out, void* stream);
/* K3 _matern_gen_fn, _src/gp/kernels/numpy.py:34-43 (selected whenever the smoothness is not fixed
 * at 1/2, 3/2, 5/2 or inf: gp/kernels/matern.py:61-81): out[i] = 2^(1-nu)/Gamma(nu) x^nu K_nu(x),
 * x = sqrt(2 nu) in[i] in_scale, zeros replaced by eps like the reference.  K_nu is evaluated in
 * fp64 on the device (Temme's series below x = 2, Steed's continued fraction above, forward
 * recurrence in the order); the reference calls scipy.special.kv / gamma.  Unlike the reference the
 * input is not modified.  mgp_matern_gen_constants writes the nu-only host constants
 * {mu, nl, coef, gam1, gam2, 1/Gamma(1+mu), 1/Gamma(1-mu)} (a CPU-side known-answer check). */
int mgp_matern_gen_f32(const float* in, int64_t n, double in_scale, double smoothness, float* out, void* stream);
int mgp_matern_gen_f64(const double* in, int64_t n, double in_scale, double smoothness, double* out, void* stream);
int mgp_matern_gen_constants(double smoothness, double* out7);
/* N1/N2 perturb, _src/gp/noise/numpy.py:9-14,56-67: out = Kin + diag(noise). noise_dev
 * is (b,k) when noise_mode == MGP_NOISE_BATCH. */
int mgp_perturb_f32(const float* Kin, int64_t b, int k, int noise_mode, double noise_scalar,
                    const float* noise_dev, float* out, void* stream);
int mgp_perturb_f64(const double* Kin, int64_t b, int k, int noise_mode, double noise_scalar,
                    const double* noise_dev, double* out, void* stream);
/* S1/S2/S3 + fast-mean precompute on a materialised (already perturbed) Kin (b,k,k):
 *   mean (b,R)   = Kcross^T Kin^-1 Y           _src/gp/muygps/numpy.py:17-41
 *   var (b)      = kout - Kcross^T Kin^-1 Kcross   :44-67
 *   ykinvy (b,R) = y_r^T Kin^-1 y_r            _src/optimize/scale/numpy.py:9-15
 *   coeffs (b,k,R) = Kin^-1 Y                  _src/gp/muygps/numpy.py:88-95
 * Any of Kcross / Y and the outputs that need them may be NULL. */
int mgp_solve_f32(const float* Kin, const float* Kcross, const float* Y, int64_t b, int k, int R, double kout,
                  float* mean, float* var, float* ykinvy, float* coeffs, int* info, void* stream);
int mgp_solve_f64(const double* Kin, const double* Kcross, const double* Y, int64_t b, int k, int R, double kout,
                  double* mean, double* var, double* ykinvy, double* coeffs, int* info, void* stream);

/* ---------------------------------------------------------------------------
 * Fused fast posterior mean (prediction from precomputed coefficients).  Replaces
 *   _crosswise_tensor + metric + kernel + _muygps_fast_posterior_mean
 *   (_src/gp/tensors/numpy.py:47-58,89-94, _src/gp/kernels/numpy.py:12-31,
 *    _src/gp/muygps/numpy.py:70-77) as chained by examples/fast_posterior_mean.py:317-400:
 *   mean[t, r] = sum_j kernel(dist(q_t, x_{nn_idx[t, j]})) * coeffs[coeff_row[t], j, r]
 * coeffs (n_train, k, R) = mgp_solve_* `coeffs` output of the self-including neighbourhoods
 * (_muygps_fast_posterior_mean_precompute, numpy.py:88-95); coeff_row (b) = the closest
 * training point of each test point; nn_idx (b, k) = that point's neighbourhood.
 * ------------------------------------------------------------------------- */
int mgp_fast_posterior_mean_f32(const float* feat_q, const float* feat_nn, int d, const int64_t* batch_idx,
                                const int64_t* nn_idx, int64_t b, int k, const float* coeffs,
                                const int64_t* coeff_row, int R, int kernel_id, int metric_id,
                                const float* length_scale, int ls_count, float* mean, void* stream);
int mgp_fast_posterior_mean_f64(const double* feat_q, const double* feat_nn, int d, const int64_t* batch_idx,
                                const int64_t* nn_idx, int64_t b, int k, const double* coeffs,
                                const int64_t* coeff_row, int R, int kernel_id, int metric_id,
                                const double* length_scale, int ls_count, double* mean, void* stream);

/* L1/L2 loss sums in fp64, _src/optimize/loss/numpy.py:22-117.  For n residuals
 * r = pred - target and variances v (may be NULL), with s = *scale_dev (device
 * double, NULL = 1) writes
 *   out[0] = sum r^2                      (mse numerator, :22-31)
 *   out[1] = sum r^2 / (s v) + log(s v)   (lool, :34-61)
 *   out[2] = sum huber_delta^2 (sqrt(1 + (r/huber_delta)^2) - 1)            (:64-72)
 *   out[3] = sum 2 looph_delta^2 (sqrt(1 + r^2/(looph_delta^2 s v)) - 1) + log(s v)  (:75-117)
 *   out[4] = sum r^2 / v,  out[5] = sum log v   (separable lool pieces for a single allreduce)
 * `out` is a device double[6], overwritten (not accumulated).
 * `scratch`: a device double[mgp_reduce_scratch_doubles()] owned by the call, or NULL.
 * With scratch the sum is a deterministic two-stage reduction (per-workgroup partials
 * added in workgroup order: the same inputs always give the same bits); with NULL the
 * partials meet in fp64 atomics. */
int mgp_reduce_scratch_doubles(void);
int mgp_loss_sums_f32(const float* pred, const float* target, const float* var, int64_t n,
                      const double* scale_dev, double huber_delta, double looph_delta, double* out,
                      double* scratch, void* stream);
int mgp_loss_sums_f64(const double* pred, const double* target, const double* var, int64_t n,
                      const double* scale_dev, double huber_delta, double looph_delta, double* out,
                      double* scratch, void* stream);
/* The one collective of the path: in-place SUM all-reduce of `count` device doubles (the partial sums mgp_loocv_* /
 * mgp_loss_sums_* / mgp_column_sums_* leave behind) over the caller's RCCL communicator (an ncclComm_t passed as
 * void*), enqueued on `stream`.  Replaces comm_world.allreduce(..., op=MPI.SUM) of _src/optimize/loss/mpi.py:23-24,57
 * and _src/optimize/scale/mpi.py:35-36.  RCCL is opened with dlopen on first use (MUYGPYS_HIP_RCCL overrides the
 * library name); MGP_EUNSUPPORTED when none is found, MGP_EINVAL for a NULL pointer / communicator or count < 1,
 * -(2000 + ncclResult_t) when RCCL reports an error. */
int mgp_allreduce_partials(double* partials, int count, void* nccl_comm, void* stream);

/* Column sums in fp64: out[r] = sum_i x[i, r]  (x is (n, R)); used for
 * sum_b y^T K^-1 y (scale/numpy.py:9-15).  out is a device double[R]; scratch as above. */
int mgp_column_sums_f32(const float* x, int64_t n, int R, double* out, double* scratch, void* stream);
int mgp_column_sums_f64(const double* x, int64_t n, int R, double* out, double* scratch, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MUYGPYS_HIP_H */

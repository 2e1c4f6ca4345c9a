// Backward (vector-Jacobian product) of one response's posterior mean / variance / y^T K^-1 y with respect to the
// HYPER-PARAMETERS -- length scale(s), the noise diagonal, the neighbours' responses -- for the fp64 static shapes of the
// dealt-triangle forward kernels (BASELINE config 4: anisotropic Matern, k = 50, d = 8: the gradient of the LOOCV
// objective that L-BFGS-B and the torch layer ask for; round 6), and, further down, for the row-per-lane static shapes
// of either element type (config 3: k = 30, d = 40) -- those with the FEATURE cotangents as well.
//
// Reference: torch autograd over src/MuyGPyS/torch/muygps_layer.py:129-164 (the reference's way to these gradients);
// the numpy chassis has only finite differences (src/MuyGPyS/_src/optimize/chassis/numpy.py:57-81).  The maths is
// mgp_backward.hip's:  a = K^-1 c, u = K^-1 y,  K-bar_ij = 2 gv a_i a_j - gm (a_i u_j + a_j u_i) - 2 gy u_i u_j,
// c-bar_j = gm u_j - 2 gv a_j,  q_ij = K-bar_ij dk/dacc_ij,  dL/dl_f = -(2 / l_f) sum_pairs q_ij (z_if - z_jf)^2.
//
// The kernel IS the forward kernel (mgp_fused_wave_kernel.h, BWD instantiation): gather, pair distances (kept in
// registers), covariances, the dealt lower triangle and its elimination -- 434 FMAs per neighbourhood instead of the
// 1 400 of a row per lane -- with every finished column written back into the dealt image, which so becomes the
// factor; then back-substitution for the two vectors on that image, the pair cotangents in the pair scheme's own
// layout, and the length-scale partials from the tile.  Two waves per SIMD; round 5's row-per-lane kernel
// (mgp_backward_wave.hip: a 64-double row, 32 kept distances, the multipliers in a 34 KB LDS matrix) ran at one:
// 117.7 ms per 2 M neighbourhoods against a 12.9 ms forward.
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "mgp_fused_wave_launch.h"

namespace mgp {

static void bwd_args(const BackwardArgs& b, FusedArgs* a);

template <int KFIX, int DFIX>
static int launch_bwd_dlt_impl(const BackwardArgs& b, hipStream_t stream) {
  using T = double;
  constexpr int NP = 64, RFIX = 1;
  constexpr WaveDims WD = wave_dims(sizeof(T), NP, KFIX, RFIX, DFIX, false, false);
  static_assert(WD.DLT && WD.NH == 1, "the dealt-triangle shapes");
  constexpr int E = WD.E, CH = WD.CH, KMAT = WD.KMAT;
  FusedArgs a;
  bwd_args(b, &a);
  WaveGeom g;
  g.mask = 0xF;
  g.q = KFIX;
  const int dpad = (a.d + CH - 1) / CH * CH;
  g.dst = dpad;
  g.xs = g.dst + E;
  g.vec_ok = 1;
  g.ntasks = a.b;
  size_t lds_unused = 0;
  gen_geometry(a, &g, &lds_unused, (int)sizeof(T));
  const size_t tile_feat = (size_t)wave_tile_rows(WD, NP, KFIX, g.xs) * g.xs + wave_stage_elems(WD);
  size_t lds = (tile_feat + KMAT) * sizeof(T) + wave_colbuf_bytes(sizeof(T), NP, false);
  lds = (lds + 15) & ~(size_t)15;
  static Residency res;
  int per_cu = 0, cus = 0;
  const void* fn = reinterpret_cast<const void*>(&fused_wave_kernel<T, NP, KFIX, RFIX, DFIX, true, false, false, false, false, true>);
  const int rrc = res.lookup(fn, 64, lds, &per_cu, &cus);
  if (rrc != MGP_OK) return rrc;
  static const int env_per_cu = getenv("MGP_BWD_DLT_PER_CU") ? atoi(getenv("MGP_BWD_DLT_PER_CU")) : 0;  // occupancy experiments
  if (env_per_cu > 0 && env_per_cu < per_cu) per_cu = env_per_cu;
  int64_t grid = (int64_t)cus * per_cu / 8 * 8;
  if (grid < 8) grid = 8;
  if (grid > g.ntasks) grid = (g.ntasks + 7) / 8 * 8;
  static const bool trace = getenv("MGP_TRACE") != nullptr;
  if (trace)
    fprintf(stderr, "mgp: backward on the dealt triangle, k = %d, d = %d: b = %lld, grid %lld, lds %zu B, %d workgroups per CU\n", KFIX,
            DFIX, (long long)a.b, (long long)grid, lds, per_cu);
  hipLaunchKernelGGL((fused_wave_kernel<T, NP, KFIX, RFIX, DFIX, true, false, false, false, false, true>), dim3((unsigned)grid), dim3(64),
                     lds, stream, a, g);
  MGP_HIP_CHECK_LAUNCH();
  note_launch("mgp::fused_wave_kernel<double,%d,%d,%d,%d,true,false,false,false,false,backward>", NP, KFIX, RFIX, DFIX);
  note_launch_geometry(grid, lds);
  return MGP_OK;
}

// The same for a static shape the library was not built with: the BWD instantiation compiled at run time (mgp_jit.hip:
// one second per shape, cached on disk), geometry from wave_dims() evaluated at run time -- as launch_jit of
// mgp_fused_wave.hip does for the forward kernels.
static bool dlt_shape(int k, int d) {  // fp64, one response: 33 .. 64 slots, rows of whole 16-byte groups, one feature stage
  return k + 2 >= 33 && k + 2 <= 64 && d >= 2 && d % 2 == 0 && (d + 3) / 4 * 4 <= 64;
}
static void bwd_args(const BackwardArgs& b, FusedArgs* a) {
  *a = b.f;
  a->mean = a->var = a->ykinvy = nullptr;
  a->coeffs = nullptr;
  a->tree = LoocvTree{};
  a->packed_q = a->packed_nn = nullptr;
  a->bwd_gmean = b.grad_mean;
  a->bwd_gvar = b.grad_var;
  a->bwd_gyk = b.grad_yk;
  a->bwd_gls = b.grad_ls;
  a->bwd_gnz = b.grad_noise;
  a->bwd_gtg = b.grad_targets;
  a->bwd_gnn = b.grad_feat_nn;
  a->bwd_gq = b.grad_feat_q;
}
static int launch_bwd_dlt_jit(const BackwardArgs& b, hipStream_t stream) {
  using T = double;
  constexpr int NP = 64;
  FusedArgs a;
  bwd_args(b, &a);
  if (!dlt_shape(a.k, a.d) || jit_mode() == 0) return MGP_EUNSUPPORTED;
  const WaveDims WD = wave_dims(sizeof(T), NP, a.k, 1, a.d, false, false);
  if (!WD.DLT) return MGP_EUNSUPPORTED;
  hipFunction_t fn = nullptr;
  const int jrc = jit_wave_function(sizeof(T), NP, a.k, 1, a.d, false, false, &fn, jit_mode() == 2 || a.b >= jit_min_batch(), false, true);
  if (jrc != MGP_OK) return jrc;
  WaveGeom g;
  g.mask = 0xF;
  g.q = a.k;
  g.dst = (a.d + WD.CH - 1) / WD.CH * WD.CH;
  g.xs = g.dst + WD.E;
  g.vec_ok = 1;
  g.ntasks = a.b;
  size_t lds_unused = 0;
  gen_geometry(a, &g, &lds_unused, (int)sizeof(T));
  const size_t tile_feat = (size_t)wave_tile_rows(WD, NP, a.k, g.xs) * g.xs + wave_stage_elems(WD);
  size_t lds = (tile_feat + WD.KMAT) * sizeof(T) + wave_colbuf_bytes(sizeof(T), NP, false);
  lds = (lds + 15) & ~(size_t)15;
  static Residency res;
  int per_cu = 0, cus = 0;
  const int rrc = res.lookup(fn, 64, lds, &per_cu, &cus);
  if (rrc != MGP_OK) return rrc;
  int64_t grid = (int64_t)cus * per_cu / 8 * 8;
  if (grid < 8) grid = 8;
  if (grid > g.ntasks) grid = (g.ntasks + 7) / 8 * 8;
  void* params[] = {&a, &g};
  const hipError_t err = hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 64, 1, 1, (unsigned)lds, stream, params, nullptr);
  if (err != hipSuccess) return -(1000 + (int)err);
  note_launch("mgp::fused_wave_kernel<double,%d,%d,1,%d,true,false,false,false,false,backward> [run-time compiled]", NP, a.k, a.d);
  note_launch_geometry(grid, lds);
  return MGP_OK;
}

// ---- row-per-lane form: 32-slot static shapes of either element type (BASELINE config 3's k = 30, d = 40 built in) ----
// slots of the row-per-lane instantiation that serves a shape (0: none): 32 for up to 32 rows (smaller neighbourhoods
// ride in the 32-slot kernel: lanes idle, nothing else changes), 64 for fp32 beyond; rows of whole 16-byte groups, one
// feature stage
template <typename T>
static int row_slots(int k, int d) {
  constexpr int E = 16 / (int)sizeof(T), CH = 2 * E;
  constexpr int DCAP = sizeof(T) == 4 ? 128 : 64;  // (fp32: rows of up to 128 features in one stage)
  if (k < 3 || d < E || d % E != 0 || (d + CH - 1) / CH * CH > DCAP) return 0;
  if (k + 2 <= 32) return 32;
  return (sizeof(T) == 4 && k + 2 <= 64) ? 64 : 0;
}
template <typename T>
static bool row_shape(int k, int d) { return row_slots<T>(k, d) != 0; }
template <typename T>
static bool row_gram(const FusedArgs& f) {  // (the forward kernels' rule: fp32, not the Matern-1/2 kernel)
  return sizeof(T) == 4 && MGP_GRAM && f.kernel_id != MGP_KERNEL_MATERN_05;
}
template <typename T>
static void row_geometry(const FusedArgs& a, const WaveDims& WD, int NP, WaveGeom* g, size_t* lds) {
  g->mask = 0xF;
  g->q = a.k;
  g->dst = (a.d + WD.CH - 1) / WD.CH * WD.CH;
  g->xs = g->dst + WD.E;
  g->vec_ok = 1;
  g->ntasks = (a.b + WD.NH - 1) / WD.NH;
  size_t unused = 0;
  gen_geometry(a, g, &unused, (int)sizeof(T));
  const size_t tile_feat = (size_t)wave_tile_rows(WD, NP, a.k, g->xs) * g->xs + wave_stage_elems(WD);
  // (behind tile and exchange images: the column buffers / norm array / row addresses, and -- BWD -- the two solved
  // vectors of both neighbourhoods: 128 entries)
  size_t tail = wave_colbuf_bytes(sizeof(T), NP, false) > 128 * sizeof(T) ? wave_colbuf_bytes(sizeof(T), NP, false) : 128 * sizeof(T);
  if (wave_bwd_tailfree((int)sizeof(T), NP, row_gram<T>(a), g->dst)) tail = 0;  // (everything that lived there has another home)
  const size_t kmat = (size_t)NP * (NP + WD.E);  // (whole rows: the packed triangle of the 64-slot forward does not apply)
  *lds = ((tile_feat + (size_t)WD.NH * kmat) * sizeof(T) + tail + 15) & ~(size_t)15;
}
static int64_t row_grid(int cus, int per_cu, int64_t ntasks) {
  static const int env_per_cu = getenv("MGP_BWD_ROW_PER_CU") ? atoi(getenv("MGP_BWD_ROW_PER_CU")) : 0;  // occupancy experiments
  if (env_per_cu > 0 && env_per_cu < per_cu) per_cu = env_per_cu;
  int64_t grid = (int64_t)cus * per_cu / 8 * 8;
  if (grid < 8) grid = 8;
  if (grid > ntasks) grid = (ntasks + 7) / 8 * 8;
  return grid;
}
template <typename T, int KFIX, int DFIX, bool GRAM>
static int launch_bwd_row_impl(const BackwardArgs& b, hipStream_t stream) {
  constexpr int NP = 32;
  constexpr WaveDims WD = wave_dims(sizeof(T), NP, KFIX, 1, DFIX, false, GRAM);
  static_assert(!WD.DLT && WD.STAT, "a 32-slot static shape");
  FusedArgs a;
  bwd_args(b, &a);
  WaveGeom g;
  size_t lds = 0;
  row_geometry<T>(a, WD, NP, &g, &lds);
  static Residency res;
  int per_cu = 0, cus = 0;
  const void* fn = reinterpret_cast<const void*>(&fused_wave_kernel<T, NP, KFIX, 1, DFIX, true, false, false, GRAM, false, true>);
  const int rrc = res.lookup(fn, 64, lds, &per_cu, &cus);
  if (rrc != MGP_OK) return rrc;
  const int64_t grid = row_grid(cus, per_cu, g.ntasks);
  hipLaunchKernelGGL((fused_wave_kernel<T, NP, KFIX, 1, DFIX, true, false, false, GRAM, false, true>), dim3((unsigned)grid), dim3(64), lds,
                     stream, a, g);
  MGP_HIP_CHECK_LAUNCH();
  note_launch("mgp::fused_wave_kernel<%s,%d,%d,1,%d,true,false,false,%s,false,backward>", sizeof(T) == 4 ? "float" : "double", NP, KFIX,
              DFIX, GRAM ? "true" : "false");
  note_launch_geometry(grid, lds);
  return MGP_OK;
}
template <typename T>
static int launch_bwd_row_jit(const BackwardArgs& b, hipStream_t stream) {
  FusedArgs a;
  bwd_args(b, &a);
  const int NP = row_slots<T>(a.k, a.d);
  if (NP == 0 || jit_mode() == 0) return MGP_EUNSUPPORTED;
  const bool gram = row_gram<T>(a);
  const WaveDims WD = wave_dims(sizeof(T), NP, a.k, 1, a.d, false, gram);
  if (WD.DLT || !WD.STAT) return MGP_EUNSUPPORTED;
  hipFunction_t fn = nullptr;
  const int jrc = jit_wave_function(sizeof(T), NP, a.k, 1, a.d, false, gram, &fn, jit_mode() == 2 || a.b >= jit_min_batch(), false, true);
  if (jrc != MGP_OK) return jrc;
  WaveGeom g;
  size_t lds = 0;
  row_geometry<T>(a, WD, NP, &g, &lds);
  static Residency res;
  int per_cu = 0, cus = 0;
  const int rrc = res.lookup(fn, 64, lds, &per_cu, &cus);
  if (rrc != MGP_OK) return rrc;
  const int64_t grid = row_grid(cus, per_cu, g.ntasks);
  void* params[] = {&a, &g};
  const hipError_t err = hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 64, 1, 1, (unsigned)lds, stream, params, nullptr);
  if (err != hipSuccess) return -(1000 + (int)err);
  note_launch("mgp::fused_wave_kernel<%s,%d,%d,1,%d,true,false,false,%s,false,backward> [run-time compiled]", sizeof(T) == 4 ? "float" : "double",
              NP, a.k, a.d, gram ? "true" : "false");
  note_launch_geometry(grid, lds);
  return MGP_OK;
}

int prepare_backward_fwd(int elem_size, int k, int d, int kernel_id) {
  if (elem_size == 8 && dlt_shape(k, d) && wave_dims(8, 64, k, 1, d, false, false).DLT) {
    if (k == 50 && d == 8) return MGP_OK;  // built into the library
    return jit_wave_prepare(8, 64, k, 1, d, false, false, false, true);
  }
  const bool ok = elem_size == 4 ? row_shape<float>(k, d) : (elem_size == 8 && row_shape<double>(k, d));
  if (!ok) return MGP_EUNSUPPORTED;
  const bool gram = elem_size == 4 && MGP_GRAM && kernel_id != MGP_KERNEL_MATERN_05;
  if (elem_size == 4 && k == 30 && d == 40 && gram) return MGP_OK;  // built in
  const int np = elem_size == 4 ? row_slots<float>(k, d) : row_slots<double>(k, d);
  return jit_wave_prepare(elem_size, np, k, 1, d, false, gram, false, true);
}

// gradients of one response; plain tables, 16-byte aligned rows (feature cotangents: the row-per-lane shapes only)
template <typename T>
int launch_backward_fwd(const BackwardArgs& b, hipStream_t stream) {
  const FusedArgs& f = b.f;
  // (several responses: one combined right-hand side Y g_mean, formed as the rows' responses are fetched; the LOOCV form
  // with its y^T K^-1 y is one response by definition)
  if (f.R < 1 || (f.R != 1 && b.grad_yk) || f.targets_batch || f.kernel_id == MGP_KERNEL_MATERN_GEN) return MGP_EUNSUPPORTED;
  const bool feat = b.grad_feat_q != nullptr || b.grad_feat_nn != nullptr;  // feature cotangents: the row-per-lane form has them
  if (f.ls_count != 1 && f.ls_count != f.d) return MGP_EUNSUPPORTED;
  const uintptr_t align = (uintptr_t)f.feat_q | (uintptr_t)f.feat_nn;
  if (align % 16 != 0 || f.b >= ((int64_t)1 << 31)) return MGP_EUNSUPPORTED;
  static const bool off = getenv("MGP_BACKWARD_DLT") != nullptr && atoi(getenv("MGP_BACKWARD_DLT")) == 0;  // A/B switch (timing only)
  if (off) return MGP_EUNSUPPORTED;
  if constexpr (sizeof(T) == 8) {
    if (feat && !row_shape<T>(f.k, f.d)) return MGP_EUNSUPPORTED;
    if (f.k == 50 && f.d == 8) return launch_bwd_dlt_impl<50, 8>(b, stream);
    if (dlt_shape(f.k, f.d)) return launch_bwd_dlt_jit(b, stream);
  } else {
    if (f.k == 30 && f.d == 40 && row_gram<T>(f)) return launch_bwd_row_impl<float, 30, 40, true>(b, stream);
  }
  return launch_bwd_row_jit<T>(b, stream);
}
template int launch_backward_fwd<float>(const BackwardArgs&, hipStream_t);
template int launch_backward_fwd<double>(const BackwardArgs&, hipStream_t);

}  // namespace mgp

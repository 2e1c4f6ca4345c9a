// Host side of the register-resident wave kernels: launch geometry, the precompiled instantiations,
// dispatch.  The kernel itself lives in mgp_fused_wave_kernel.h.
#include <cstdio>
#include <cstdlib>

#include "mgp_fused_wave_launch.h"

namespace mgp {

#ifdef MGP_DEBUG_HOOKS
int g_phase_mask = 0xF;
int g_grid_per_cu = 0;  // override of resident workgroups per CU
int g_lds_pad = 0;      // extra dynamic LDS bytes per workgroup
#endif

// fp32 pipelined kernels compute the squared distances in the Gram form, except for the Matern-1/2
// kernel (and the general Matern below nu = 1): exp(-r) has a kink at r = 0, so the absolute error a
// cancelling Gram form leaves in a tiny squared distance (duplicated training points) would show up at
// first order there.
// In fp64 the Gram form is used for every kernel: its absolute error in a squared distance, ~1e-16 r^2, is
// eleven orders below the 1e-5 the results are held to.
template <typename T>
static bool gram_allowed(const FusedArgs& a) {
  if (sizeof(T) == 8) return MGP_GRAM64 != 0;
  return a.kernel_id != MGP_KERNEL_MATERN_05 && !(a.kernel_id == MGP_KERNEL_MATERN_GEN && a.smoothness < 1.0);
}

// (launch_np_impl is instantiated in mgp_fused_wave_inst_*.hip; here only the calls)
template <typename T, int NP, int KFIX, int RFIX, int DFIX, bool PIPED, bool COEFF, bool PACKED, bool GRAM, bool GEN64 = false>
static int launch_inst(const FusedArgs& a, hipStream_t stream) {
  return launch_np_impl<T, NP, KFIX, RFIX, DFIX, PIPED, COEFF, PACKED, GRAM, GEN64>(a, stream);
}

template <typename T, int NP, int KFIX, int RFIX, int DFIX, bool PIPED, bool COEFF = false, bool PACKED = false>
static int launch_np(const FusedArgs& a, hipStream_t stream) {
  // fp64 general-smoothness Matern: its own instantiations, for the run-time-shape 32-slot kernels (static shapes:
  // compiled at run time, launch_jit); the difference form throughout
  if constexpr (sizeof(T) == 8) {
    if (a.kernel_id == MGP_KERNEL_MATERN_GEN) {
      if constexpr (NP == 32 && KFIX == 0 && !COEFF) return launch_inst<T, NP, KFIX, RFIX, DFIX, PIPED, COEFF, PACKED, false, true>(a, stream);
      else return MGP_EUNSUPPORTED;
    }
  }
  if constexpr (PIPED && !COEFF && MGP_GRAM && (sizeof(T) == 4 || MGP_GRAM64)) {
    if (gram_allowed<T>(a)) return launch_inst<T, NP, KFIX, RFIX, DFIX, PIPED, COEFF, PACKED, true>(a, stream);
  }
  return launch_inst<T, NP, KFIX, RFIX, DFIX, PIPED, COEFF, PACKED, false>(a, stream);
}

// Slots per neighbourhood of the static instantiation that serves a shape (0: none does): 16 / 32 / 64,
// pipelined gather (16-byte aligned rows, one feature stage), at most four responses in a prepared table.
static int static_slots(int es, int d, int k, int R, bool packed, bool gathered) {
  const int E = 16 / es, CH = 2 * E, rows = k + 1 + R;
  if (k < 1 || R < 1 || rows > 64 || d % E != 0 || (d + CH - 1) / CH * CH > 64) return 0;
  if (packed && !gathered && R > E) return 0;
  return rows <= 16 ? 16 : (rows <= 32 ? 32 : 64);
}

// A static shape the library was not built with: the instantiation compiled at run time (mgp_jit.hip).
// Same geometry code as launch_np_impl, from wave_dims() evaluated at run time.
template <typename T>
static int launch_jit(const FusedArgs& a, hipStream_t stream) {
  const bool packed = a.packed_nn != nullptr;
  const int NP = static_slots(sizeof(T), a.d, a.k, a.R, packed, a.targets_batch != 0);
  if (NP == 0 || a.coeffs != nullptr) return MGP_EUNSUPPORTED;
  const uintptr_t align = packed ? ((uintptr_t)a.packed_q | (uintptr_t)a.packed_nn | (uintptr_t)a.q_stride | (uintptr_t)a.nn_stride)
                                 : ((uintptr_t)a.feat_q | (uintptr_t)a.feat_nn);
  if (align % 16 != 0) return MGP_EUNSUPPORTED;
  const bool gen64 = sizeof(T) == 8 && a.kernel_id == MGP_KERNEL_MATERN_GEN;
  const bool gram = MGP_GRAM && gram_allowed<T>(a) && !gen64;
  hipFunction_t fn = nullptr;
  const int jrc = jit_wave_function(sizeof(T), NP, a.k, a.R, a.d, packed, gram, &fn, jit_mode() == 2 || a.b >= jit_min_batch(), gen64);
  if (jrc != MGP_OK) return jrc;
  const WaveDims WD = wave_dims(sizeof(T), NP, a.k, a.R, a.d, false, gram);
  WaveGeom g;
  g.mask = 0xF;
  g.q = a.k;
  const int dpad = (a.d + WD.CH - 1) / WD.CH * WD.CH;
  g.dst = dpad;
  g.xs = g.dst + WD.E;
  g.vec_ok = 1;
  g.ntasks = (a.b + WD.NH - 1) / WD.NH;
  const size_t tile_feat = (size_t)wave_tile_rows(WD, NP, a.k, g.xs) * g.xs + wave_stage_elems(WD), tile_mat = (size_t)WD.NH * WD.KMAT;
  const size_t tile_elems = tile_feat > tile_mat ? tile_feat : tile_mat;
  size_t lds = tile_elems * sizeof(T) + wave_colbuf_bytes(sizeof(T), NP, wave_fold(sizeof(T), NP, a.k, a.R, a.d, true, false, gram));
  lds = (lds + 15) & ~(size_t)15;
  gen_geometry(a, &g, &lds, (int)sizeof(T));
  static Residency res;
  int per_cu = 0, cus = 0;
  const int rrc = res.lookup(fn, 64, lds, &per_cu, &cus);
  if (rrc != MGP_OK) return rrc;
#ifdef MGP_DEBUG_HOOKS
  if (g_grid_per_cu > 0) per_cu = g_grid_per_cu;
#endif
  static const int env_per_cu = getenv("MGP_JIT_PER_CU") ? atoi(getenv("MGP_JIT_PER_CU")) : 0;  // occupancy experiments
  if (env_per_cu > 0 && env_per_cu < per_cu) per_cu = env_per_cu;
  int64_t grid = (int64_t)cus * per_cu / 8 * 8;
  if (grid < 8) grid = 8;
  if (grid > g.ntasks) grid = (g.ntasks + 7) / 8 * 8;
  if (a.tree.out && grid > kTreeMaxLeaves) return MGP_EUNSUPPORTED;  // (as launch_np_impl)
  static const bool trace = getenv("MGP_TRACE") != nullptr;
  if (trace)
    fprintf(stderr, "mgp: [run-time compiled] fused_wave_kernel<%s,%d,%d,%d,%d,pipe%s%s> b=%lld grid=%lld lds=%zu\n",
            sizeof(T) == 4 ? "float" : "double", NP, a.k, a.R, a.d, packed ? ",packed" : "", gram ? ",gram" : "", (long long)a.b,
            (long long)grid, lds);
  FusedArgs args = a;
  args.tree.grid = (int)grid;  // (the leaves of the reduction tree are this launch's workgroups)
  args.tree.nh = WD.NH;
  if (a.tree.mode == kTreeThreeLaunch) args.tree.out = nullptr;  // (as launch_np_impl)
  void* params[] = {&args, &g};
  const hipError_t err = hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 64, 1, 1, (unsigned)lds, stream, params, nullptr);
  if (err != hipSuccess) return -(1000 + (int)err);
  note_launch("mgp::fused_wave_kernel<%s,%d,%d,%d,%d,true,false,%s,%s%s> [run-time compiled]", sizeof(T) == 4 ? "float" : "double", NP,
              a.k, a.R, a.d, packed ? "true" : "false", gram ? "true" : "false", gen64 ? ",gen64" : "");
  note_tree_geometry(a.tree.out ? (int)grid : 0, WD.NH);
  note_launch_geometry(grid, lds);
  return MGP_OK;
}

template <typename T>
int launch_fused_wave(const FusedArgs& a, hipStream_t stream) {
  const int rows = a.k + 1 + a.R;
  if (a.b >= (int64_t)1 << 31) return MGP_EUNSUPPORTED;  // (task and chunk numbers are 32-bit in the kernel)
  if (a.tree.out && (a.R != 1 || a.coeffs || !a.ykinvy || !a.tree.ctrl || !a.tree.resp)) return MGP_EINVAL;
  if (a.coeffs != nullptr) {  // fused fast-mean precompute: one response
    if (a.packed_nn != nullptr) return MGP_EUNSUPPORTED;
    if (a.R == 1 && rows <= 32) return launch_np<T, 32, 0, 0, 0, false, true>(a, stream);
    if (a.R == 1 && rows <= 64) return launch_np<T, 64, 0, 0, 0, false, true>(a, stream);
    return MGP_EUNSUPPORTED;
  }
  // other static shapes: the instantiation compiled at run time, when allowed and worth it (mgp_jit.hip)
  const bool gen64 = sizeof(T) == 8 && a.kernel_id == MGP_KERNEL_MATERN_GEN;  // (no built-in static instantiation)
  const bool builtin = !gen64 && a.R == 1 && ((a.k == 30 && a.d == 40) || (a.k == 50 && a.d == 8));
  // (calls from MUYGPYS_HIP_JIT_MIN_BATCH neighbourhoods on may compile; shorter ones, from
  // MUYGPYS_HIP_JIT_CACHED_MIN_BATCH on, take a kernel that is loaded or in the disk cache already)
  const bool try_jit = !builtin && jit_mode() != 0 && (jit_mode() == 2 || a.b >= jit_cached_min_batch());
  if (a.packed_nn != nullptr) {  // prepared tables: the pipelined kernels only
    if (builtin && a.k == 30) return launch_np<T, 32, 30, 1, 40, true, false, true>(a, stream);
    if (builtin && a.k == 50) return launch_np<T, 64, 50, 1, 8, true, false, true>(a, stream);
    if (try_jit) {
      const int rc = launch_jit<T>(a, stream);
      if (rc != MGP_EUNSUPPORTED) return rc;
    }
    if (rows <= 32) return launch_np<T, 32, 0, 0, 0, true, false, true>(a, stream);
    if (rows <= 64) return launch_np<T, 64, 0, 0, 0, true, false, true>(a, stream);
    return MGP_EUNSUPPORTED;
  }
  if (builtin && a.k == 30) {  // BASELINE configs 2/3 (and their fp64 form), all shapes static
    const int rc = launch_np<T, 32, 30, 1, 40, true>(a, stream);
    if (rc != MGP_EUNSUPPORTED) return rc;
  }
  if (builtin && a.k == 50) {  // BASELINE config 4 shape, all shapes static
    const int rc = launch_np<T, 64, 50, 1, 8, true>(a, stream);
    if (rc != MGP_EUNSUPPORTED) return rc;
  }
  if (try_jit) {
    const int rc = launch_jit<T>(a, stream);
    if (rc != MGP_EUNSUPPORTED) return rc;
  }
  // run-time shapes: the pipelined direct-to-LDS gather when the rows allow it (16-byte aligned,
  // d a multiple of 16 bytes, one feature stage), the register-staged gather otherwise
  if (rows <= 32) {
    const int rc = launch_np<T, 32, 0, 0, 0, true>(a, stream);
    return rc != MGP_EUNSUPPORTED ? rc : launch_np<T, 32, 0, 0, 0, false>(a, stream);
  }
  if (rows <= 64) {
    const int rc = launch_np<T, 64, 0, 0, 0, true>(a, stream);
    return rc != MGP_EUNSUPPORTED ? rc : launch_np<T, 64, 0, 0, 0, false>(a, stream);
  }
  return MGP_EUNSUPPORTED;
}

// The instantiation launch_fused_wave would pick for a shape (pure function of its arguments;
// mirrors the dispatch above for 16-byte aligned tables).  Empty string: not a wave-kernel shape.
int describe_fused_wave(int elem_size, int d, int k, int R, int packed, char* buf, int len) {
  const int rows = k + 1 + R, E = 16 / elem_size;
  const char* t = elem_size == 4 ? "float" : "double";
  const bool vec = d % E == 0, one_stage = ((d + 2 * E - 1) / (2 * E)) * (2 * E) <= 64;
  int np = rows <= 32 ? 32 : (rows <= 64 ? 64 : 0), kf = 0, rf = 0, df = 0;
  bool piped = vec && one_stage;
  if (k == 30 && R == 1 && d == 40) kf = 30, rf = 1, df = 40, np = 32, piped = true;
  else if (k == 50 && R == 1 && d == 8) kf = 50, rf = 1, df = 8, np = 64, piped = true;
  else if (jit_mode() != 0 && static_slots(elem_size, d, k, R, packed != 0, false) != 0)  // (large batches: run-time compiled)
    kf = k, rf = R, df = d, np = static_slots(elem_size, d, k, R, packed != 0, false), piped = true;
  if (np == 0 || (packed && (!piped || R > E))) return snprintf(buf, len, "%s", "");
  // (the Gram-form instantiation serves fp32 pipelined shapes for every kernel except Matern-1/2)
  const bool gram = piped && MGP_GRAM && (elem_size == 4 || MGP_GRAM64);
  return snprintf(buf, len, "mgp::fused_wave_kernel<%s,%d,%d,%d,%d,%s,false,%s,%s>", t, np, kf, rf, df,
                  piped ? "true" : "false", packed ? "true" : "false", gram ? "true" : "false");
}

// compile the static instantiation of a shape into the disk cache (no GPU needed)
int prepare_fused_wave(int elem_size, int d, int k, int R, int packed, int kernel_id) {
  const bool gen64 = elem_size == 8 && kernel_id == MGP_KERNEL_MATERN_GEN;  // (fp64 general Matern: an instantiation of its own)
  if (!gen64 && R == 1 && ((k == 30 && d == 40) || (k == 50 && d == 8))) return MGP_OK;  // built into the library
  const int np = static_slots(elem_size, d, k, R, packed != 0, false);
  if (np == 0) return MGP_EUNSUPPORTED;
  const bool gram = !gen64 && MGP_GRAM && (elem_size == 8 ? MGP_GRAM64 != 0 : kernel_id != MGP_KERNEL_MATERN_05);
  return jit_wave_prepare(elem_size, np, k, R, d, packed != 0, gram, gen64);
}

template int launch_fused_wave<float>(const FusedArgs&, hipStream_t);
template int launch_fused_wave<double>(const FusedArgs&, hipStream_t);

}  // namespace mgp

#if MGP_WAVE_TIMING
extern "C" int mgp_debug_wave_timing(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mgp::g_wave_timing), sizeof(mgp::g_wave_timing)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mgp::g_wave_timing), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

// Host side of one instantiation of the register-resident wave kernel: launch geometry and the launch itself.
// Included by the translation units that instantiate launch_np_impl explicitly (mgp_fused_wave_inst_*.hip: the
// instantiations of mgp_fused_wave_list.h, split by element type so that they compile in parallel) and -- for the geometry helpers -- by the dispatcher (mgp_fused_wave.hip, launch_jit).
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "mgp_fused_wave_kernel.h"

namespace mgp {

#ifdef MGP_DEBUG_HOOKS
extern int g_phase_mask;
extern int g_grid_per_cu;  // override of resident workgroups per CU
extern int g_lds_pad;      // extra dynamic LDS bytes per workgroup
#endif

// general-smoothness Matern: the node table (2 x MGP_GEN_NODES floats) goes behind everything else in
// LDS; spacing and the log2 of h 2^(1-nu)/Gamma(nu) are launch constants
inline void gen_geometry(const FusedArgs& a, WaveGeom* g, size_t* lds, int elem_size) {
  g->gen_tab = 0;
  g->gen_h = 0.5f;
  g->gen_lc = 0.0f;
  g->gen_h64 = 0.3;
  g->gen_lc64 = 0.0;
  g->gen_xmin64 = 1e-12;
  if (a.kernel_id != MGP_KERNEL_MATERN_GEN) return;
  const double nu = a.smoothness, h = gen_step(nu);
  g->gen_tab = (int)*lds;
  g->gen_h = (float)h;
  g->gen_lc = (float)((log(h) + (1.0 - nu) * log(2.0) - lgamma(nu)) / log(2.0));
  if (elem_size == 8) {  // fp64: finer step, natural logarithms, a larger table
    const double h64 = gen_step64(nu);
    g->gen_h64 = h64;
    g->gen_lc64 = log(h64) + (1.0 - nu) * log(2.0) - lgamma(nu);
    g->gen_xmin64 = gen_xmin64(nu);
    *lds += 2 * MGP_GEN_NODES64 * sizeof(double);
  } else {
    *lds += 2 * MGP_GEN_NODES * sizeof(float);
  }
}

template <typename T, int NP, int KFIX, int RFIX, int DFIX, bool PIPED, bool COEFF, bool PACKED, bool GRAM, bool GEN64>
int launch_np_impl(const FusedArgs& a, hipStream_t stream) {
  if (COEFF && a.tree.out) return MGP_EINVAL;
  constexpr WaveDims WD = wave_dims(sizeof(T), NP, KFIX, RFIX, DFIX, COEFF, GRAM);
  constexpr int NH = WD.NH;
  constexpr int E = WD.E;
  constexpr int CH = WD.CH;
  constexpr int KMAT = WD.KMAT;
  WaveGeom g;
#ifdef MGP_DEBUG_HOOKS
  g.mask = g_phase_mask;
#else
  g.mask = 0xF;
#endif
  g.q = NP - 1 - a.R;
  const int dpad = (a.d + CH - 1) / CH * CH;
  g.dst = dpad < 64 ? dpad : 64;
  g.xs = g.dst + E;  // dst/E is even -> dst/E + 1 slots: odd
  const uintptr_t align = PACKED ? ((uintptr_t)a.packed_q | (uintptr_t)a.packed_nn | (uintptr_t)a.q_stride | (uintptr_t)a.nn_stride)
                                 : ((uintptr_t)a.feat_q | (uintptr_t)a.feat_nn);
  g.vec_ok = (a.d % E == 0) && (align % 16 == 0);
  if (PACKED && a.R > E && !a.targets_batch) return MGP_EUNSUPPORTED;  // the responses ride in one 16-byte slot
  if ((DFIX > 0 || PIPED) && !g.vec_ok) return MGP_EUNSUPPORTED;
  if (PIPED && a.d > g.dst) return MGP_EUNSUPPORTED;  // more than one feature stage
  g.ntasks = (a.b + NH - 1) / NH;
  const size_t tile_feat = (size_t)wave_tile_rows(WD, NP, KFIX, g.xs) * g.xs + wave_stage_elems(WD), tile_mat = (size_t)NH * KMAT;
  const size_t tile_elems = tile_feat > tile_mat ? tile_feat : tile_mat;
  constexpr bool PIPE = PIPED;
  size_t lds = PIPE ? tile_elems * sizeof(T) +
                          wave_colbuf_bytes(sizeof(T), NP, wave_fold(sizeof(T), NP, KFIX, RFIX, DFIX, PIPED, COEFF, GRAM))
                    : (tile_elems + 64 + g.dst + (g.dst & 1)) * sizeof(T) + 64 * sizeof(int64_t);
  lds = (lds + 15) & ~(size_t)15;
  // (the general Matern needs the per-lane pair tables: 32-slot or static shapes; fp64 -- round 4 -- in the GEN64
  // instantiations only)
  if (a.kernel_id == MGP_KERNEL_MATERN_GEN && !((NP <= 32 || KFIX > 0) && !COEFF && (sizeof(T) == 4 || GEN64))) return MGP_EUNSUPPORTED;
  if (GEN64 && a.kernel_id != MGP_KERNEL_MATERN_GEN) return MGP_EUNSUPPORTED;
  gen_geometry(a, &g, &lds, (int)sizeof(T));
#ifdef MGP_DEBUG_HOOKS
  lds += (size_t)g_lds_pad;
#endif
  // Persistent grid = exactly the resident capacity: every workgroup owns a fixed share of the
  // tasks, so one workgroup more than fits runs as a second, nearly empty round (measured: 13
  // instead of 12 per CU costs 40 %).  Residency comes from the occupancy query for this kernel
  // at this LDS size; the CU count from the device.
  static Residency res;
  int per_cu = 0, cus = 0;
  const int rrc = res.lookup(
      reinterpret_cast<const void*>(&fused_wave_kernel<T, NP, KFIX, RFIX, DFIX, PIPED, COEFF, PACKED, GRAM, GEN64>), 64, lds, &per_cu,
      &cus);
  if (rrc != MGP_OK) return rrc;
#ifdef MGP_DEBUG_HOOKS
  if (g_grid_per_cu > 0) per_cu = g_grid_per_cu;
#endif
  // (fp64, 32 slots, run-time shape: two waves per SIMD although three would fit -- measured, mgp_fused_wave_kernel.h)
  if (sizeof(T) == 8 && NP == 32 && KFIX == 0 && per_cu > 8) per_cu = 8;
  static const int env_per_cu = getenv("MGP_WAVE_PER_CU") ? atoi(getenv("MGP_WAVE_PER_CU")) : 0;  // occupancy experiments
  if (env_per_cu > 0 && env_per_cu < per_cu) per_cu = env_per_cu;
  int64_t grid = (int64_t)cus * per_cu / 8 * 8;
  if (grid < 8) grid = 8;
  if (grid > g.ntasks) grid = (g.ntasks + 7) / 8 * 8;
  // (one-launch LOOCV evaluation: the leaves of the reduction tree are this launch's workgroups)
  if (a.tree.out && grid > kTreeMaxLeaves) return MGP_EUNSUPPORTED;
  FusedArgs al = a;
  al.tree.grid = (int)grid;
  al.tree.nh = NH;
  if (a.tree.mode == kTreeThreeLaunch) al.tree.out = nullptr;  // (the caller walks these leaves by kernels behind the launch)
  static const bool trace = getenv("MGP_TRACE") != nullptr;  // which instantiation served a call
  if (trace)
    fprintf(stderr, "mgp: fused_wave_kernel<%s,%d,%d,%d,%d,%s%s%s> b=%lld k=%d d=%d R=%d grid=%lld lds=%zu\n",
            sizeof(T) == 4 ? "float" : "double", NP, KFIX, RFIX, DFIX, PIPED ? "pipe" : "stage", PACKED ? ",packed" : "",
            GRAM ? ",gram" : "",
            (long long)a.b, a.k, a.d, a.R, (long long)grid, lds);
  hipLaunchKernelGGL((fused_wave_kernel<T, NP, KFIX, RFIX, DFIX, PIPED, COEFF, PACKED, GRAM, GEN64>), dim3((unsigned)grid), dim3(64),
                     lds, stream, al, g);
  MGP_HIP_CHECK_LAUNCH();
  note_launch("mgp::fused_wave_kernel<%s,%d,%d,%d,%d,%s,%s,%s,%s%s>", sizeof(T) == 4 ? "float" : "double", NP, KFIX, RFIX, DFIX,
              PIPED ? "true" : "false", COEFF ? "true" : "false", PACKED ? "true" : "false", GRAM ? "true" : "false",
              GEN64 ? ",gen64" : "");
  note_tree_geometry(a.tree.out ? (int)grid : 0, NH);
  note_launch_geometry(grid, lds);
  return MGP_OK;
}


}  // namespace mgp

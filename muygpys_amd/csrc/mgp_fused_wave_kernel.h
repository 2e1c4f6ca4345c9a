// Register-resident wave-per-neighbourhood fused kernel (the roofline path).
//
// One wavefront (64 lanes, one workgroup) owns NH = 64/NP neighbourhoods at a time; a
// neighbourhood is a set of NP "slots", one per lane:
//
//     slot 0 .. k-1      neighbour rows            (features gathered by nn_idx)
//     slot k .. q-1      padding (identity rows)    q = NP-1-R
//     slot q             the query point            (features gathered by batch_idx)
//     slot q+1 .. NP-1   the R response rows        (no features)
//
// Phase 0  indices (prefetched one task ahead), responses and nugget of the neighbours.
// Phase 1  gather: the (k+1) feature rows are staged once in LDS with coalesced 16-byte
//          loads (d/4 consecutive lanes walk one row, all loads of a task in flight together;
//          rows padded to an odd number of 16-B slots so that ds_read_b128 of a column of
//          rows is bank-conflict free).
// Phase 2  distances, difference form sum((x-y)^2) (never the Gram trick: fp32 parity),
//          each unordered pair ONCE, NP/2 pairs per lane in registers (v_pk_add_f32 /
//          v_pk_fma_f32).  LDS read bandwidth is what this phase is short of (measured: halving
//          the partner reads at equal VALU work saves 21 % of the kernel), so the pairs are
//          register-blocked BA x BP (BA = 4 own rows, BP = NP/8 partner rows per lane): lane i
//          keeps own rows i + o_j, o_0 = 0, o_j = (j+1) BP + 1, and every partner row i+p
//          (p = 1..BP) it reads serves all of them -- pair {i+p, i+o_j} has cyclic distance
//          p for j = 0 and o_j - p in (j BP, (j+1) BP] otherwise, so the BA x BP pairs of all
//          lanes cover the distances 1..NP/2 exactly like the one-row scheme, with 8 instead
//          of 17 row reads at NP = 32.
// Phase 3  kernel function + nugget, exchanged through a small LDS matrix so that lane i
//          ends up holding row i of the augmented system
//                [ K+eps  .   . ]
//                [ c^T    1   . ]     (lower triangle; see mgp_lds_factor.h for the algebra)
//                [ Y^T    0   0 ]
//          in NP registers.
// Phase 4  right-looking Cholesky, row per lane, all in registers: step j broadcasts
//          column j through a 64-entry LDS buffer (one ds_write_b32, a few uniform
//          ds_read_b128), then packed FMAs on the trailing registers.  After k steps the
//          Schur complement of the (query, responses) block holds
//          var = S[q][q],  mean_r = -S[q+1+r][q],  y_r^T K^-1 y_r = -S[q+1+r][q+1+r].
//
// No inter-wave communication, no barrier that waits on another wave (one wave per
// workgroup: __syncthreads() is a compiler / wait-count fence only).  The kernel is
// VALU-issue bound, so the code below spends its effort on instruction
// count: packed f32 math, compile-time shapes for the headline configuration
// (KFIX/RFIX/DFIX), and a grid sized to exactly the resident capacity.
// (This header is the DEVICE side: the kernel template and its compile-time knobs.  It is compiled into
// the library for the shapes instantiated in mgp_fused_wave.hip and, at run time, by hiprtc for any
// other static shape -- mgp_jit.hip -- so it includes nothing a device-only compile cannot see.)
#pragma once

#include "mgp_wave_common.h"

// Phase tests: WaveGeom::mask is a kernel argument, always 0xF in the shipped library (only builds
// with -DMGP_DEBUG_HOOKS export a setter, for the timing ablations of tools/kbench.py).  The test
// stays a run-time one on purpose: with the phases fused into one straight-line region the
// compiler hoists per-lane addresses of all phases to the top of the task loop and the static
// shapes spill (168 VGPRs + 164 B scratch instead of 134 and none; 2.78 instead of 2.22 ms).
#define MGP_PHASE(g, bit) ((g).mask & (bit))

#ifndef MGP_F64_SAME_PRIO
#define MGP_F64_SAME_PRIO 0
#endif
#ifndef MGP_DIST_PRIO
#define MGP_DIST_PRIO 1
#endif
#ifndef MGP_PRIO_EARLY_RAISE
#define MGP_PRIO_EARLY_RAISE 0
#endif
#ifndef MGP_W4
#define MGP_W4 0
#endif
#ifndef MGP_PRIO_LATE_DROP
#define MGP_PRIO_LATE_DROP 0
#endif
#ifndef MGP_XCHG_PRIO
#define MGP_XCHG_PRIO 0
#endif
#ifndef MGP_CHOL_PRIO
#define MGP_CHOL_PRIO 2
#endif
#ifndef MGP_LOOKAHEAD
#define MGP_LOOKAHEAD 1
#endif
#ifndef MGP_CHOL_ONE_BLOCK
#define MGP_CHOL_ONE_BLOCK 1
#endif
#ifndef MGP_F64_COV_BATCH
#define MGP_F64_COV_BATCH 5
#endif
#ifndef MGP_OWN_REG
#define MGP_OWN_REG 0
#endif
#ifndef MGP_C4_W3
#define MGP_C4_W3 1
#endif
#ifndef MGP_MODM
#define MGP_MODM 1
#endif
#ifndef MGP_GRAM
#define MGP_GRAM 1
#endif
// (the Gram form in fp64: precision is no concern there, and the code path exists -- but the fp64 kernels are
// bound by the elimination, not by the distances: config 4 114.8 vs 114.9 M/s, fp64 headline 241.6 vs 245.2,
// k = 20 / d = 16 340 vs 332.  Off.)
#ifndef MGP_GRAM64
#define MGP_GRAM64 0
#endif
#ifndef MGP_DMA_ASM
#define MGP_DMA_ASM 1
#endif
#ifndef MGP_DIST_ASM
#define MGP_DIST_ASM 1
#endif
#ifndef MGP_F64_GC
#define MGP_F64_GC 6
#endif
// fp32, 32 slots, static shapes with k >= 16 and one response: the elimination FOLDED -- sixteen lanes per
// neighbourhood, lane l holding the lower-triangle parts of rows l and 16 + l, four neighbourhoods (two
// consecutive tasks of the workgroup) per elimination.  See phase 4F.
#ifndef MGP_FOLD
#define MGP_FOLD 1
#endif
// fp64, 64 slots, static shapes: the lower triangle of the augmented system DEALT over the lanes
// in vertical pairs (rows 2r, 2r+1 of one column per 16-byte register group, column-major, 64 consecutive pairs per
// slot) instead of a row per lane.  See phase 4D.
#ifndef MGP_DLT
#define MGP_DLT 1
#endif
#ifndef MGP_F64_DIST_HALF
#define MGP_F64_DIST_HALF 0
#endif

// (BWD keeps all NS squared distances through the covariance phase, where the forward retires them batch by batch:
// five interleaved exp chains on top of that spill 26 registers at two waves per SIMD)
#ifndef MGP_BWD_COV_BATCH
#define MGP_BWD_COV_BATCH 3
#endif
#ifndef MGP_BWD_SWEEP_LATE
#define MGP_BWD_SWEEP_LATE -1  // length-scale sweep behind the pair phase (1), inside it (0), by shape (-1)
#endif
#ifndef MGP_BWD_EXP
#define MGP_BWD_EXP 0  // (experiments: bits switch parts of the BWD instantiation off -- 1 length-scale partials, 2 pair cotangents, 4 back-substitution, 8 factor write-back)
#endif
#ifndef MGP_FEAT_EXP
#define MGP_FEAT_EXP 0  // (experiments on the feature-cotangent phase: 1 no sweep, 2 no pair stores, 4 no output, 8 plain stores for the atomic adds)
#endif

namespace mgp {

struct WaveGeom {
  int q;         // query slot
  int dst;       // feature stage width (elements, multiple of the chunk)
  int xs;        // LDS row stride of the feature tile (elements)
  int vec_ok;    // 16-byte gathers allowed (d % (16/sizeof T) == 0, bases aligned)
  int64_t ntasks;
  int mask;      // debug: phases to execute (bit0 gather, 1 distances, 2 kernel+exchange, 3 factor)
  // general-smoothness Matern (kernel_id == MGP_KERNEL_MATERN_GEN, fp32): byte offset of the node table
  // in LDS (2 x MGP_GEN_NODES floats), node spacing, log2(h 2^(1-nu) / Gamma(nu))
  int gen_tab;
  float gen_h, gen_lc;
  // ... and in fp64 (2 x MGP_GEN_NODES64 doubles; natural logarithms; smallest scaled distance the table covers)
  double gen_h64, gen_lc64, gen_xmin64;
};

// Sizes shared by the kernel and its launchers.  Plain constexpr functions of the shape (element size es,
// slots NP, and the static k / R / d or 0), so that the precompiled instantiations (template arguments)
// and the run-time compiled ones (mgp_fused_wave.hip, launch_jit: the same numbers at run time) cannot
// drift apart.
struct WaveDims {
  int E, CH, NH, NPL, NG, KS, KMAT, BA, BP, NS, M;
  bool STAT, TRI, MODM;
  bool DLT;             // dealt lower triangle (phase 4D)
  int NR2, NPAIR, NSL;  // (DLT) row pairs per column 0, pairs in all, 64-pair slots = register groups per lane
};
// (DLT) first pair of column c: columns 2m and 2m + 1 start at row pair m and hold NR2 - m pairs each
constexpr int dlt_col_start(int c, int NR2) {
  const int m = c >> 1, b = c & 1;
  return 2 * m * NR2 - m * (m - 1) + b * (NR2 - m);
}
// The row-per-lane backward at the headline shape (fp32, 32 slots, Gram form, 33 .. 40 padded features) without the
// 512 bytes behind tile and image -- 20 480 instead of 20 992 bytes of LDS per wave: the eighth wave of a CU, and the
// kernel's rate follows the waves in flight.  What lived there moves into space that is dead at the time: row pointers
// of the gather and the norms into the image (not yet / no longer in use), the elimination's pivot buffer into the
// tile's query row (all zeros in the Gram form: re-zeroed afterwards), the two solved vectors into the padding columns
// of the image, the rows' element offsets of the scatter into the image's last rows.
constexpr bool wave_bwd_tailfree(int es, int NP, bool gram, int dst) { return es == 4 && NP == 32 && gram && dst > 32 && dst <= 40; }
constexpr WaveDims wave_dims(int es, int NP, int KFIX, int RFIX, int DFIX, bool COEFF, bool GRAM) {
  WaveDims w{};
  w.E = 16 / es;
  w.CH = 2 * w.E;
  w.NH = 64 / NP;
  w.STAT = KFIX > 0 && RFIX > 0 && DFIX > 0 && !COEFF;  // all shapes static
  w.NPL = w.STAT ? KFIX + 1 + RFIX : NP;                 // live slots
  w.NG = (w.NPL + w.E - 1) / w.E;                        // 16-byte groups of a lane's row
  w.KS = NP + w.E;
  w.TRI = NP == 64 && !COEFF;
  w.DLT = MGP_DLT && es == 8 && NP == 64 && w.STAT && RFIX >= 1;
  w.NR2 = (w.NPL + 1) / 2;
  w.NPAIR = dlt_col_start(w.NPL, w.NR2);
  w.NSL = (w.NPAIR + 63) / 64;
  // elements per exchange matrix (packed: up to the last live row, the NG groups a lane reads from
  // there, a dump slot)
  w.KMAT = w.TRI ? w.E * ((w.NPL - 1) / w.E + 1) * (w.E * ((w.NPL - 1) / w.E) / 2 + (w.NPL - 1) % w.E) + w.NG * w.E + w.E
                 : NP * w.KS;
  if (w.DLT) w.KMAT = 2 * 64 * w.NSL + 2;  // NSL slots of 64 pairs, and a dump pair
  // Pair scheme.  Generic: the NP slots form the cycle, NP/2 pairs per lane in 4 x NP/8 blocks (the
  // half-way distance twice; pairs with a response or padding slot computed and discarded).  Static
  // shapes run the cycle modulo the feature-row count M = k + 1 instead when that is shorter: cyclic
  // distances 1 .. M/2, BA own rows x BP partner rows per lane with BA x BP >= M/2 (BA of 3, 4 or 5,
  // whichever wastes least; a distance computed twice -- t and M - t, or the surplus of the blocking --
  // writes the same value to the same slot).  k = 30: 15 pairs (3 x 5) instead of 16; k = 50: 25
  // (5 x 5) instead of 32; k = 10: 5 instead of 16.  With the Gram form the modulo scheme must save at
  // least two pairs per lane: the wrap at M breaks the bank spread of the tile rows, measured 1.89 vs
  // 1.83 ms at k = 30.
  w.M = NP;
  w.BA = 4;
  w.BP = NP / 8;
  w.MODM = false;
  if (MGP_MODM && w.STAT) {
    const int need = (KFIX + 1) / 2;
    int ba = 4, bp = (need + 3) / 4;
    for (int c = 3; c <= 5; c += 2) {
      const int q = (need + c - 1) / c;
      if (c * q < ba * bp) ba = c, bp = q;
    }
    if (ba * bp + (GRAM ? 2 : 1) <= NP / 2) {
      w.MODM = true;
      w.M = KFIX + 1;
      w.BA = ba;
      w.BP = bp;
    }
  }
  w.NS = w.BA * w.BP;
  return w;
}
// (BWD) pair s = j * BP + (p - 1) of a lane joins rows i + o_j and i + p (cyclic, modulo M): cyclic distance
// delta = (o_j - p) mod M.  Sums over unordered pairs must meet every pair ONCE: a pair is counted by the first s whose
// distance class {delta, M - delta} it is (the surplus of the blocking, and t / M - t of the modulo scheme, repeat
// classes); the half-way class 2 delta = M is met from both ends by the same s -- the kernel keeps the end with r1 < c.
constexpr int wave_pair_delta(int s, int BP, int M) {
  const int j = s / BP, p = s % BP + 1, o = j == 0 ? 0 : (j + 1) * BP + 1;
  return ((o - p) % M + M) % M;
}
constexpr bool wave_pair_first(int s, int BP, int M) {
  const int d = wave_pair_delta(s, BP, M);
  if (d == 0) return false;
  for (int t = 0; t < s; ++t) {
    const int e = wave_pair_delta(t, BP, M);
    if (e == d || e == M - d) return false;
  }
  return true;
}
constexpr unsigned long long wave_pair_first_mask(int NS, int BP, int M) {  // bit s: pair s counts
  unsigned long long m = 0;
  for (int s = 0; s < NS; ++s)
    if (wave_pair_first(s, BP, M)) m |= 1ull << s;
  return m;
}
constexpr unsigned long long wave_pair_half_mask(int NS, int BP, int M) {  // bit s: pair s is of the half-way class
  unsigned long long m = 0;
  for (int s = 0; s < NS; ++s)
    if (2 * wave_pair_delta(s, BP, M) == M) m |= 1ull << s;
  return m;
}
// rows of the feature tile: all NP slots of every neighbourhood of the wave, or -- one static
// neighbourhood per wave -- the live slots only (the direct-to-LDS gather stops after the query row;
// its last 1-KiB piece may run into the first response row, which holds no features)
constexpr int wave_tile_rows(const WaveDims& w, int NP, int KFIX, int xs) {
  if (!(w.STAT && w.NH == 1)) return w.NH * NP;
  const int spr = xs / w.E, pieces = ((KFIX + 1) * spr + 63) / 64;
  const int covered = (pieces * 64 + spr - 1) / spr;
  // (without the modulo-M pair scheme the pairs cycle over all NP slots: the rows behind the live ones must exist --
  // the Gram path parks an infinite norm in them so that their pairs cannot trip the cancellation guard)
  const int rows = w.MODM ? w.NPL : NP;
  return covered > rows ? covered : rows;
}
constexpr int wave_gather_pieces(const WaveDims& w, int KFIX, int xs) {
  const int spr = xs / w.E;
  return (w.STAT && w.NH == 1) ? ((KFIX + 1) * spr + 63) / 64 : spr;
}
// (DLT) elements of the posted-column staging area behind the feature tile: two slots of 64 pairs and a bias of 32
// pairs in front (a consumer addresses it relative to row pair j / 2 of the pivot column)
constexpr int wave_stage_elems(const WaveDims& w) { return w.DLT ? 2 * (32 + 128) : 0; }
// waves per SIMD the register allocation is held to
// (fp64, 64 slots: the dealt-lower-triangle kernels -- 44 instead of 104 registers of matrix -- fit three waves:
// config 4 158.0 vs 153.1 M/s; the row-per-lane ones spill there, measured 10 % slower)
// (run-time shapes, round 4: without the accumulators' phantom live range -- sec. 4.1b (v) -- the fp32 64-slot kernels
// need 155..176 registers: held to three waves, MGP_RT_W3: k = 45, d = 24: 121 -> 134 M/s, k = 40, d = 8: 132 -> 157.
// The fp64 32-slot ones would fit three waves as well (152..164) and are SLOWER there -- k = 20, d = 8 / 24 / 32: 340 /
// 255 / 223 M/s at twelve workgroups per CU against 377 / 297 / 269 at eight, the same from d = 40 on: their launcher
// caps the residency at eight, mgp_fused_wave.hip)
#ifndef MGP_RT_W3
#define MGP_RT_W3 1
#endif
constexpr int wave_min_waves(int es, int NP, int KFIX, int RFIX = 0, int DFIX = 0) {
  return es == 4 ? (NP <= 32 ? (KFIX == 30 && MGP_W4 ? 4 : 3) : (MGP_RT_W3 && KFIX == 0 ? 3 : 2))
                 : (NP <= 32 ? 2 : (MGP_C4_W3 && wave_dims(es, NP, KFIX, RFIX, DFIX, false, false).DLT ? 3 : 2));
}

// Folded elimination (phase 4F): which instantiations use it, and the bytes of column buffers behind the tile
// (one per neighbourhood in flight; never less than the 64 row addresses of the gather that share the space).
#ifndef MGP_FOLD64
#define MGP_FOLD64 1
#endif
// (16 slots, eight lanes per neighbourhood: k = 10, d = 8 / 40 / 64: 0.140 / 0.289 / 0.460 vs 0.138 / 0.292 / 0.463 ms -- nothing)
#ifndef MGP_FOLD16
#define MGP_FOLD16 0
#endif
// (fp64, 32 slots, folded: 96 parked VGPRs, 4-6 spilled at the headline shape -- measured 3.894 vs 3.899 ms: nothing)
#ifndef MGP_FOLD_F64
#define MGP_FOLD_F64 0
#endif
constexpr bool wave_fold(int es, int NP, int KFIX, int RFIX, int DFIX, bool PIPED, bool COEFF, bool GRAM) {
  // (fp64, 64 slots: 168 parked VGPRs -- spills; the difference-form distance phase needs more registers than
  // the Gram form and spills too)
  return MGP_FOLD && PIPED && KFIX > 0 && RFIX >= 1 && DFIX > 0 && !COEFF && KFIX >= NP / 2 &&
         ((es == 4 && GRAM && (NP == 32 || (NP == 64 && MGP_FOLD64) || (NP == 16 && MGP_FOLD16))) || (es == 8 && NP == 32 && MGP_FOLD_F64));
}
constexpr int wave_colbuf_bytes(int es, int NP, bool fold) {
  const int need = fold ? (64 / (NP / 2)) * NP * es : 64 * es;
  return need > 512 ? need : 512;
}

// (DLT) per (slot, lane): element offsets into the posted column of the pair's two rows (2 r) and of its column's
// entry (c).  A compile-time table in the code object's constant data, fetched per task right before the elimination
// (11 coalesced 8-byte loads at k = 50; L2-resident) -- as 22 registers computed once per kernel they were live across
// the distance phase, the kernel's register peak.
template <int NPL_>
struct DltMeta {
  static constexpr int NR2 = (NPL_ + 1) / 2;
  static constexpr int NPAIR = dlt_col_start(NPL_, NR2);
  static constexpr int NSL = (NPAIR + 63) / 64;
  int rc[NSL * 64][2];
  constexpr DltMeta() : rc{} {
    int c = 0;
    for (int e = 0; e < NSL * 64; ++e) {
      while (c + 1 < NPL_ && dlt_col_start(c + 1, NR2) <= e) ++c;
      rc[e][0] = e < NPAIR ? 2 * (e - dlt_col_start(c, NR2) + (c >> 1)) : 0;  // (behind the last pair: any valid offset)
      rc[e][1] = c;
    }
  }
};
template <int NPL_>
__device__ const DltMeta<NPL_> g_dlt_meta{};

// KFIX / RFIX / DFIX > 0: nn_count / response_count / feature_count known at compile time.
// PIPED: software-pipelined direct-to-LDS gather (one feature stage, 16-byte aligned rows).
// COEFF: also emit K^-1 y (the fast-posterior-mean coefficients): multipliers kept, back-substitution.
// PACKED: the tables are prepared tables (mgp_table_pack_*): rows of [features | responses | pad] at a
//        64-byte multiple stride, so a row and its response arrive with the same two cache lines
//        and no separate 4-byte response read (a whole line each) is issued.
// GRAM: (pipelined kernels) squared distances as |a'|^2 + |b'|^2 - 2 a'.b' on rows centred on the query
//        in place (a' = a - q, times the inverse length scales under Anisotropy): one packed FMA per
//        two features of a pair instead of a packed subtract + a packed FMA.  See phase 1b / 2.
// GEN64: (fp64) the general-smoothness Matern is the covariance function of this instantiation (the trapezoidal
//        K_nu of mgp_wave_common.h in software exp: a body of its own, kept out of the fixed-smoothness kernels, where
//        it cost the config-4 kernel spilled registers).  fp32 kernels carry their (hardware-exp) form in every
//        instantiation.
#ifndef MGP_WAVE_TIMING
#define MGP_WAVE_TIMING 0
#endif
#if MGP_WAVE_TIMING
// phase timing (experiments only; tools/wave_timing.py): s_memtime differences summed per wave and phase
__device__ unsigned long long g_wave_timing[8];
#define MGP_WAVE_T(slot)                                           \
  {                                                                \
    const unsigned long long tnow_ = __builtin_readcyclecounter(); \
    tacc_[slot] += tnow_ - tlast_;                                 \
    tlast_ = tnow_;                                                \
  }
#else
#define MGP_WAVE_T(slot)
#endif
// BWD: (dealt-triangle instantiations, one response) the BACKWARD of the instantiation's outputs with respect to the
//        hyper-parameters -- the forward phases, with the squared distances kept, the feature tile and the dealt image of
//        the system in separate LDS regions (the image becomes the factor: every column is written back when it is
//        posted), then back-substitution, pair cotangents and length-scale partials.  See "phase 6B" below and
//        mgp_backward_dlt.hip.  Two waves per SIMD.
template <typename T, int NP, int KFIX, int RFIX, int DFIX, bool PIPED, bool COEFF = false, bool PACKED = false,
          bool GRAM = false, bool GEN64 = false, bool BWD = false>
__global__ __launch_bounds__(64, BWD ? 2 : wave_min_waves(sizeof(T), NP, KFIX, RFIX, DFIX))
void fused_wave_kernel(FusedArgs a, WaveGeom g) {
  static_assert(!GEN64 || sizeof(T) == 8, "GEN64 is the fp64 general-smoothness instantiation");
  static_assert(!PACKED || PIPED, "prepared tables are gathered by the direct-to-LDS pipeline");
  static_assert(!GRAM || (PIPED && !COEFF), "Gram form: one feature stage");
  constexpr WaveDims WD = wave_dims(sizeof(T), NP, KFIX, RFIX, DFIX, COEFF, GRAM);
  constexpr int NH = WD.NH;       // neighbourhoods per wave
  constexpr bool STAT = WD.STAT;  // all shapes static: q = k, NPL live slots, lanes NPL .. NP-1 idle; everything
  constexpr int NPL = WD.NPL;     // downstream is sized by NPL (k = 50, R = 1: 52 of 64 slots, 26 of 32 fp64 groups)
  static_assert(NPL <= NP, "live slots");
  constexpr bool MODM = WD.MODM;  // pair scheme modulo M feature rows (wave_dims)
  constexpr int M = WD.M;
  constexpr int NS = WD.NS;       // pairs per lane
  // (32 slots, fp32, Gram form; or -- MGP_FOLD64 -- 64 slots: one neighbourhood per half-wave.  The query and
  // response rows must be among the long rows: k >= NP / 2)
  constexpr bool FOLD = !BWD && wave_fold(sizeof(T), NP, KFIX, RFIX, DFIX, PIPED, COEFF, GRAM);  // (BWD keeps the factor: row per lane)
  constexpr int HALF = NP / 2;    // (FOLD) lanes per neighbourhood; lane l owns rows l and HALF + l
  constexpr int NGS = HALF / WD.E;  // (FOLD) 16-byte groups of a short row
  constexpr int LOGH = NP == 64 ? 5 : (NP == 32 ? 4 : 3);
  constexpr int BA = WD.BA;       // own rows per lane        } register blocking of the pair scheme,
  constexpr int BP = WD.BP;       // partner rows per lane    } see phase 2
  auto own_offset = [](int j) { return j == 0 ? 0 : (j + 1) * BP + 1; };
  auto wrap = [](int r) { return MODM ? r % M : r & (NP - 1); };
  constexpr int E = v16<T>::N;    // elements per 16 bytes
  constexpr int CH = 2 * E;       // feature chunk per inner iteration (two 16-B reads per row)
  constexpr int KS = NP + E;      // row stride of the (square) exchange matrix: NP/E + 1 (odd) 16-B slots
  // 64-slot neighbourhoods keep the exchange matrix PACKED lower-triangular: row r holds its r + 1
  // entries, padded to whole 16-byte groups, at rowoff(r) = E (a + 1) (E a / 2 + r % E), a = r / E --
  // 17 KB instead of 34 KB in fp64, so LDS no longer holds these shapes at one wave per SIMD.  A lane
  // reads NP entries from the start of its row; what lies beyond its diagonal belongs to later rows
  // (upper-triangle garbage the elimination never uses).
  constexpr bool TRI = NP == 64 && !COEFF && !(BWD && !WD.DLT);  // (the row-per-lane backward keeps whole rows of multipliers)
  auto rowoff = [](int r) {
    if constexpr (TRI) {
      const int a = r / E;
      return E * (a + 1) * (E * a / 2 + r % E);
    } else {
      return r * KS;
    }
  };
  constexpr int NG = WD.NG;      // 16-byte groups of a lane's row
  constexpr int KMAT = (BWD && !WD.DLT) ? NP * KS : WD.KMAT;  // elements per exchange matrix
  // Dealt lower triangle (phase 4D): entry (i, c), i >= c, lives in pair cs2(c) + i/2 - c/2, element i & 1; lane l holds
  // pairs 64 s + l, s = 0 .. NSL-1.  The exchange matrix is stored in exactly that order, so the read-back is NSL
  // lane-linear ds_read_b128.
  constexpr bool DLT = WD.DLT;
  // BWD comes in two layouts: on the dealt triangle (fp64, 64 slots) and -- round 6, second half -- row per lane for the
  // 32-slot static shapes of either element type (BASELINE config 3's shape: the multipliers kept in the exchange image
  // as the COEFF variant keeps them, two neighbourhoods per wave)
  constexpr bool BWD_ROW = BWD && !DLT;
  static_assert(!BWD || (RFIX == 1 && PIPED && !PACKED && !GEN64 && !COEFF && WD.STAT), "BWD: static shapes on plain tables, one response");
  static_assert(!BWD || DLT || NP == 32 || (NP == 64 && sizeof(T) == 4), "BWD: the dealt-triangle shapes, the 32-slot ones, fp32 with 64 slots");
  static_assert(!(BWD && DLT && GRAM), "BWD on the dealt triangle: difference form");
  constexpr int NR2 = WD.NR2, NPAIR = WD.NPAIR, NSL = WD.NSL;
  auto cs2 = [](int c) { return dlt_col_start(c, NR2); };
  auto eoff = [&](int hi, int lo) {  // element offset of entry (hi, lo) of a neighbourhood's exchange matrix, hi >= lo
    if constexpr (DLT) return 2 * (cs2(lo) + (hi >> 1) - (lo >> 1)) + (hi & 1);
    else return rowoff(hi) + lo;
  };
  constexpr int DPADFIX = (DFIX + CH - 1) / CH * CH;
  // (one feature stage of up to 64 columns; the row-per-lane backward takes rows of up to 128 -- the reference's torch
  // tutorial embeds into 100 dimensions -- at 37 KB of LDS per wave)
  constexpr int DCAP = (BWD && !WD.DLT && sizeof(T) == 4) ? 128 : 64;
  constexpr int DSTFIX = DFIX > 0 ? (DPADFIX < DCAP ? DPADFIX : DCAP) : CH;
  using V = typename v16<T>::type;
  using ACC = typename v16<T>::acc;

  extern __shared__ __attribute__((aligned(16))) char smem[];
#if MGP_WAVE_TIMING
  unsigned long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast_ = __builtin_readcyclecounter();
#endif
  const int k = KFIX > 0 ? KFIX : a.k;
  const int R = RFIX > 0 ? RFIX : a.R;
  const int d = DFIX > 0 ? DFIX : a.d;
  const int q = STAT ? KFIX : (RFIX > 0 ? NP - 1 - RFIX : g.q);
  const int dst = DFIX > 0 ? DSTFIX : g.dst;
  const int xs = DFIX > 0 ? DSTFIX + E : g.xs;
  const int tile_rows = wave_tile_rows(WD, NP, KFIX, xs);
  const int tile_feat = tile_rows * xs + wave_stage_elems(WD);  // (DLT: the staging area lies behind the rows the gather fills)
  // (BWD: the dealt image of the system lies BEHIND the feature tile, which the last phase still reads)
  const int kmat_base = BWD ? tile_feat : 0;
  const int tile_elems = BWD ? tile_feat + NH * KMAT : (tile_feat > NH * KMAT ? tile_feat : NH * KMAT);
  T* tile = reinterpret_cast<T*>(smem);               // feature tile, later the exchange matrix
  // Plain kernels: [tile][colbuf 64][ilbuf dst][idxbuf 64 x int64].  Pipelined kernels keep LDS
  // at tile + 256 bytes (9 allocation granules of 1280 B -> 14 workgroups per CU): the 32-bit row
  // indices live in the column buffer (only read while the gather is issued, before the
  // factorisation writes there) and the inverse length scales in the tile row of the last slot
  // (a response slot: it has no features, and its distances are never used).
  constexpr bool PIPE_ = PIPED;
  T* colbuf = tile + tile_elems;                      // 64 entries
  T* ilbuf = PIPE_ ? tile + (NPL - 1) * xs : colbuf + 64;  // dst entries (Anisotropy)
  int64_t* idxbuf = reinterpret_cast<int64_t*>(ilbuf + dst + (dst & 1));  // 64 entries (plain kernels)
  constexpr bool TF = BWD && !WD.DLT && wave_bwd_tailfree(sizeof(T), NP, GRAM, DSTFIX);  // (no space behind tile and image)
  const T** rowaddr = reinterpret_cast<const T**>(TF ? tile + kmat_base : colbuf);  // 64 row pointers (pipelined; overlays colbuf + 256 B)

  const float* gtab = reinterpret_cast<const float*>(smem + g.gen_tab);
  if (a.kernel_id == MGP_KERNEL_MATERN_GEN) {  // node table of the launch's smoothness, once per workgroup
    if constexpr (sizeof(T) == 4) gen_build_table(reinterpret_cast<float*>(smem + g.gen_tab), (float)a.smoothness, g.gen_h, (int)threadIdx.x);
    else if constexpr (GEN64) gen_build_table64(reinterpret_cast<double*>(smem + g.gen_tab), a.smoothness, g.gen_h64, (int)threadIdx.x);
  }

  const T* feat_q = static_cast<const T*>(a.feat_q);
  const T* feat_nn = static_cast<const T*>(a.feat_nn);
  const T* targets = static_cast<const T*>(a.targets);
  const T* noise_dev = static_cast<const T*>(a.noise_dev);
  const T* ls = static_cast<const T*>(a.length_scale);
  const bool aniso = a.ls_count > 1;
  T post_scale = T(1);
  if (!aniso) {
    const T l = ls[0];
    post_scale = a.metric_id == MGP_METRIC_L2 ? T(1) / l : T(1) / (l * l);
  }
  const bool nopad = k == q;  // every slot below q is a real neighbour

  // XCD-aware task order: workgroups b and b+8 share an XCD (round-robin dispatch), so give
  // each XCD one contiguous eighth of the neighbourhoods -> neighbouring neighbourhoods
  // (which share rows under a spatially sorted kNN) meet in the same L2.
  const int64_t ntasks = g.ntasks;
  const int64_t per_xcd = (ntasks + 7) / 8;
  const int xcd = blockIdx.x & 7;
  const int64_t t_hi = (xcd + 1) * per_xcd;
  const int64_t t_end = t_hi < ntasks ? t_hi : ntasks;
  const int64_t t_step = gridDim.x >> 3;

  // Index prefetch: the (dependent) index load of task t+1 is issued at the top of task t.
  // The row of the index tensor is addressed as uniform 64-bit base + 32-bit lane offset.
  // Branch-free on purpose: ONE load instruction per call whatever the slot, so that the value
  // can stay in flight across the task (loads under divergent branches that write the same
  // register make the compiler wait for the first before issuing the second).  Slots without
  // an index read a valid dummy element and are zeroed when the value is consumed.
  auto load_index = [&](int task, int h, int i) -> int64_t {
    const int64_t nb0 = (int64_t)task * NH;                       // uniform
    const int hh = (NH > 1 && nb0 + h >= a.b) ? 0 : h;   // odd tail: replay the first half
    const int64_t* row = a.nn_idx + nb0 * k;
    const int64_t* p = row + (hh * k + (i < k ? i : 0));
    if (a.batch_idx != nullptr && i == q) p = a.batch_idx + nb0 + hh;
    return *p;
  };
  auto fix_index = [&](int64_t raw, int task, int h, int i) -> int64_t {
    const int64_t nb0 = (int64_t)task * NH;
    const int hh = (NH > 1 && nb0 + h >= a.b) ? 0 : h;
    if (i < k) return raw;
    if (i == q) return a.batch_idx != nullptr ? raw : nb0 + hh;
    return 0;
  };

  const int64_t task0 = xcd * per_xcd + (blockIdx.x >> 3);

  // ---- the workgroup's task sequence T0, T1, ...: T_n = task0 + n t_step inside the XCD's eighth; the index row of
  // T_{n+2} is requested while T_n runs.  (Round 5 also built a dynamic sequence for the LOOCV instantiations -- chunks
  // of a folded pair drawn from per-XCD dequeue heads, draws consumed a task later -- and measured it out: at config 3's
  // strong-scaling shard, ten pairs per workgroup, whole pairs quantise worse than 20-or-21 tasks do, +12 us on 196;
  // at 1 M neighbourhoods -0.5 %.)
  int g_pos = 0;  // tasks dealt
  auto seq_gen = [&]() -> int {
    const int64_t ts = task0 + (int64_t)g_pos * t_step;
    ++g_pos;
    return ts < t_end ? (int)ts : -1;
  };
  int task = seq_gen(), t1 = seq_gen(), t2 = -1;  // (task numbers are 32-bit: the launcher refuses more)
  int64_t next_idx = 0;
  if (task >= 0) next_idx = load_index(task, NH == 1 ? 0 : (int)threadIdx.x / NP, threadIdx.x & (NP - 1));

  // Software-pipelined gather (static shapes, one feature stage): the feature tile of task
  // t+1 is requested right before the factorisation of task t -- by then the tile region of
  // LDS is free (row i of the system sits in registers) -- with direct global->LDS loads, so it
  // costs no registers and its latency hides behind the Cholesky.  One load instruction fills
  // 64 consecutive 16-byte slots of the tile (SPR slots per row, the last one padding).
  constexpr bool PIPE = PIPED;
  static_assert(!PIPED || DFIX <= DCAP, "the pipelined gather stages all features at once");
  const int SPR = xs / E;                                   // 16-byte slots per staged row (NH * NP = 64 rows
  const int C16V = d / E;                                   //  -> SPR loads per task); C16V of them hold data
  constexpr int GB = 11;                                    // loads issued per batch (= SPR at d = 40, fp32)
  const unsigned spr_magic = (1u << 20) / (unsigned)SPR + 1u;  // sigma / SPR for sigma < 64 * 33
  T pre_y = T(0), pre_eps = T(0);
  int64_t pre_idx = 0, pre_tg = 0;  // pre_tg: row of the response tensor (table row, or b * k + slot when gathered)
  auto pipe_issue = [&](int task_n, int64_t idx_n, int lane_) {
    const int h = NH == 1 ? 0 : lane_ / NP;
    const int i = lane_ & (NP - 1);
    // pointer to this slot's feature row; slots without one (idx_n = 0) point at row 0 -- any
    // valid row will do, their tile rows are never used
    if constexpr (PACKED)
      rowaddr[lane_] = reinterpret_cast<const T*>(
          (i == q ? static_cast<const char*>(a.packed_q) + idx_n * a.q_stride
                  : static_cast<const char*>(a.packed_nn) + idx_n * a.nn_stride));
    else
      rowaddr[lane_] = (i == q ? feat_q : feat_nn) + idx_n * (int64_t)d;
    pre_idx = idx_n;
    __syncthreads();
    if (MGP_PHASE(g, 1)) {
      // 16-byte slot sigma = 64 n + lane of the tile: row = sigma / SPR, column = sigma % SPR
      // (unsigned 32-bit arithmetic throughout; padding slots re-read the last data slot, the
      // ones inside the padded feature range are zeroed when the tile is consumed)
      const int NPC = wave_gather_pieces(WD, KFIX, xs);  // 1-KiB pieces per task (SPR when all 64 rows are gathered)
      for (int n0 = 0; n0 < NPC; n0 += GB) {
        const T* src[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          if (n0 + u < NPC) {  // uniform; folded for the static shapes
            const unsigned sigma = 64u * (unsigned)(n0 + u) + (unsigned)lane_;
            const unsigned row = (sigma * spr_magic) >> 20;
            // prepared tables: the slot after the features holds the responses
            const unsigned c = min(sigma - row * (unsigned)SPR, (unsigned)(PACKED ? C16V : C16V - 1));
            src[u] = rowaddr[row] + c * E;
          }
        }
#pragma unroll
        for (int u = 0; u < GB; ++u)
          if (n0 + u < NPC) {  // the tile starts the dynamic LDS
#if MGP_DMA_ASM
            glds16_asm(src[u], smem, (n0 + u) * 1024);
#else
            glds16_lds(src[u], smem, (n0 + u) * 1024);
#endif
          }
      }
    }
    // response and nugget of the slot's row: unconditional single loads (idx_n is 0, a valid
    // row, for slots without one; the values are masked where they are consumed)
    {
      const int64_t nb0 = (int64_t)task_n * NH;
      const int hh = (NH > 1 && nb0 + h >= a.b) ? 0 : h;
      // (only neighbour slots index the response / noise tables: the query slot's row number belongs to
      // the QUERY table, which may be the longer one)
      pre_tg = a.targets_batch ? (nb0 + hh) * k + (i < k ? i : 0) : (i < k ? idx_n : 0);
    }
    // (prepared tables: the responses arrive in the tile, unless the caller hands them over gathered)
    if (!PACKED || a.targets_batch) pre_y = targets[pre_tg * (int64_t)R];
    if constexpr (BWD) {
      // several responses (a.R > 1; the system itself has one right-hand side): the combined column Y g_mean of the
      // row -- mean-bar . mean = a^T K^-1 (Y g_mean), so one solve serves any response count (mgp_backward.hip)
      if (a.R > 1) {  // (uniform)
        const T* gmp = static_cast<const T*>(a.bwd_gmean);
        const int64_t nbq = (int64_t)task_n * NH + ((NH > 1 && (int64_t)task_n * NH + h >= a.b) ? 0 : h);
        T ys = T(0);
        for (int r = 0; r < a.R; ++r) ys = fma_t(gmp ? gmp[nbq * a.R + r] : T(0), targets[pre_tg * (int64_t)a.R + r], ys);
        pre_y = ys;
      }
    }
    pre_eps = (T)a.noise_scalar;
    if (a.noise_mode != MGP_NOISE_SCALAR) {
      const int64_t nb0 = (int64_t)task_n * NH;
      const int hh = (NH > 1 && nb0 + h >= a.b) ? 0 : h;
      const T* pn = a.noise_mode == MGP_NOISE_TABLE ? noise_dev + (i < k ? idx_n : 0) : noise_dev + nb0 * k + (hh * k + (i < k ? i : 0));
      pre_eps = *pn;
    }
  };
  if (PIPE && task >= 0) {
    pipe_issue(task, fix_index(next_idx, task, NH == 1 ? 0 : (int)threadIdx.x / NP, threadIdx.x & (NP - 1)),
               threadIdx.x);
    if (t1 >= 0) next_idx = load_index(t1, NH == 1 ? 0 : (int)threadIdx.x / NP, threadIdx.x & (NP - 1));
  }

  // Exchange-matrix slot of each of the lane's NS pairs (phase 3) and which of them are real
  // (both rows neighbours, or neighbour x query): functions of the lane only, so computed once per
  // kernel instead of once per task (8 integer instructions per pair: 9 % of the headline kernel's
  // VALU work).  NS registers: the 32-slot kernels and the static shapes have them to spare.
  constexpr bool XPRE = (NP <= 32 || KFIX > 0) && !COEFF;
  // (more than 16 pairs per lane: two 16-bit element offsets per register -- the 25 offsets of the
  // k = 50 kernel are what stands between it and a third wave per SIMD)
#ifndef MGP_FOLD_XPK
#define MGP_FOLD_XPK 0
#endif
#ifndef MGP_FOLD_DPRE
#define MGP_FOLD_DPRE 0
#endif
  constexpr bool XPK = XPRE && (NS > 16 || (FOLD && (MGP_FOLD_XPK || sizeof(T) == 8)));  // (FOLD, fp64: the parked rows need the registers)
  static_assert(!XPK || NH * KMAT + (BWD ? 4096 : 0) < 65536, "packed exchange offsets are 16-bit");
  int xoff[XPRE ? (XPK ? (NS + 1) / 2 : NS) : 1];
  unsigned xkeep = 0;
  if constexpr (XPRE) {
    const int i0 = threadIdx.x & (NP - 1);
    const int hbase = (NH == 1 ? 0 : (int)threadIdx.x / NP) * KMAT + kmat_base;
    const int dump0 = DLT ? KMAT - 2 : (TRI ? KMAT - E : (NP - 1) * KS + NP);
#pragma unroll
    for (int s = 1; s <= NS; ++s) {
      const int r1 = wrap(i0 + own_offset((s - 1) / BP));
      const int c = wrap(i0 + (s - 1) % BP + 1);
      const int hi = max(r1, c), lo = min(r1, c);
      // (dropped: a response / padding slot, a lane that repeats another, and -- tiny M -- a surplus
      // distance that wraps onto the row itself)
      const int xo = hbase + (hi <= q && hi != lo && i0 < M ? eoff(hi, lo) : dump0);
      if constexpr (XPK) {
        if ((s - 1) % 2 == 0) xoff[(s - 1) / 2] = xo;
        else xoff[(s - 1) / 2] |= xo << 16;
      } else {
        xoff[s - 1] = xo;
      }
      if (lo < k && (hi < k || hi == q)) xkeep |= 1u << (s - 1);
    }
  }

  // feature-tile rows of the lane's own and partner rows (phase 2), static shapes: lane-only as well
  constexpr bool DPRE = XPRE && DFIX > 0 && NP <= 32 && (!FOLD || MGP_FOLD_DPRE);  // (the 64-slot static kernels have no registers left for it)
  int down[DPRE ? BA : 1], dpar[DPRE ? BP : 1];
  if constexpr (DPRE) {
    const int i0 = threadIdx.x & (NP - 1);
    const int hb = (NH == 1 ? 0 : (int)threadIdx.x / NP) * NP * xs;
#pragma unroll
    for (int j = 0; j < BA; ++j) down[j] = hb + wrap(i0 + own_offset(j)) * xs;
#pragma unroll
    for (int s2 = 1; s2 <= BP; ++s2) dpar[s2 - 1] = hb + wrap(i0 + s2) * xs;
  }
  static_assert(!MODM || XPRE, "the modulo-M pair scheme relies on the per-lane exchange offsets");
  static_assert(!DLT || XPRE, "the dealt layout is written through the per-lane exchange offsets");

  // (FOLD) the folded rows: FS = row l (columns 0 .. 15), FL = row 16 + l, of the neighbourhood of the lane's
  // quarter; quarters 0 / 1 belong to the first task of a pair, 2 / 3 to the second
  V FL[FOLD ? NG : 1], FS[FOLD ? NGS : 1];
  int fold_sub = 0;
  int fold_task_a = 0;
  for (; task >= 0; task = t1, t1 = t2) {
    // The lane id is made opaque per task: otherwise LICM hoists every per-lane address, mask
    // and index of the unrolled phases out of this loop and the kernel runs out of registers.
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));
    const int h = NH == 1 ? 0 : lane / NP;
    const int i = lane & (NP - 1);
    T* Xh = tile + h * NP * xs;
    T* Kh = tile + kmat_base + h * KMAT;
    T* colh = TF ? Xh + KFIX * xs : colbuf + h * NP;  // (TF: the query row of the neighbourhood's tile)
    int64_t* idxh = idxbuf + h * NP;
    const int64_t nb0 = (int64_t)task * NH;
    const bool live = nb0 + h < a.b;
    const int hh = live ? h : 0;


    // ---- phase 0: indices, responses, nugget -------------------------------------------
    int64_t myidx = 0, mytg = 0;
    T myeps = T(0), myy0 = T(0);
    V myyv = V(0);  // prepared tables: the row's responses, taken from the tile
    if (PIPE) {
      // the tile of this task was requested during the previous task's factorisation; the
      // barrier at the top of the stage loop below waits for it (vmcnt) before anyone reads it
      myidx = pre_idx;
      mytg = pre_tg;
      myy0 = i < k ? pre_y : T(0);
      myeps = pre_eps;
    } else {
      myidx = fix_index(next_idx, task, h, i);
      if (t1 >= 0) next_idx = load_index(t1, h, i);
      __syncthreads();  // previous task's LDS reads are complete
      idxh[i] = myidx * (int64_t)d;  // element offset of the row
      mytg = a.targets_batch ? (nb0 + hh) * k + (i < k ? i : 0) : myidx;
      if (i < k) {
        myy0 = targets[mytg * (int64_t)R];
        if (a.noise_mode == MGP_NOISE_SCALAR) myeps = (T)a.noise_scalar;
        else if (a.noise_mode == MGP_NOISE_TABLE) myeps = noise_dev[myidx];
        else myeps = (noise_dev + nb0 * k)[hh * k + i];
      }
    }

    ACC acc[NS];
    // (zeroed here, not under `d0 == 0` inside the feature loop: with a run-time feature count a conditional first
    // store makes the accumulators a value carried around the persistent loop -- NS registers (pairs) live, untouched,
    // through the whole elimination; found in the rhs-column kernel, round 4)
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = ACC(0);
#if MGP_CHOL_PRIO && MGP_PRIO_LATE_DROP
    __builtin_amdgcn_s_setprio(0);  // the distance phase: long independent streams, lowest priority
#endif
#if MGP_DIST_PRIO
    // fp32: elimination (2) > distances (1) > covariances / exchange (0); fp64 (software exp in the
    // covariances, few distance instructions at small d): elimination (2) > exchange (1) > distances (0)
    __builtin_amdgcn_s_setprio(sizeof(T) == 4 || MGP_F64_SAME_PRIO ? MGP_DIST_PRIO : MGP_XCHG_PRIO);
#endif

    // ---- phases 1+2: stage features, accumulate squared distances ---------------------
    for (int d0 = 0; d0 < d; d0 += dst) {
      const int w = min(dst, d - d0);
      const int wp = (w + CH - 1) / CH * CH;
      // (Anisotropy, pipelined kernels: the inverse length scales are REQUESTED before the wait for the tile -- they go
      // into a tile row the gather has just overwritten, every task; loaded where they are stored, behind the barrier,
      // the round trip to L2 is exposed once per task: 5 % of the headline kernel, 10 % of its backward)
      constexpr int ILN = PIPED ? (DSTFIX + 63) / 64 : 1;
      T il_pre[ILN];
      if constexpr (PIPED) {
#pragma unroll
        for (int u = 0; u < ILN; ++u) {
          const int c = u * 64 + lane;
          il_pre[u] = (aniso && c < w) ? T(1) / ls[d0 + c] : T(0);
        }
      }
#if MGP_DMA_ASM
      if (PIPE) lds_dma_wait();  // this task's tile (requested during the previous task's elimination) has landed
#endif
      __syncthreads();
      MGP_WAVE_T(0)
      if (PIPE) {
        if constexpr (PACKED) {
          // the slot behind the features carries the row's responses (before it is zeroed as padding)
          if (!a.targets_batch) {
            myyv = *reinterpret_cast<const V*>(Xh + (NPL == NP ? i : min(i, NPL - 1)) * xs + d);
            myy0 = i < k ? myyv[0] : T(0);
          }
        }
        // feature columns w .. wp-1 of the staged rows are padding of the 8-wide inner loop: the
        // direct-to-LDS gather filled them with a repeat of the last data slot
        if (wp > w && (NPL == NP || i < NPL)) *reinterpret_cast<V*>(Xh + i * xs + w) = V(0);
      } else if (!MGP_PHASE(g, 1)) {
      } else if (DFIX > 0 || g.vec_ok) {
        // c16p consecutive lanes walk one row; rpr rows per round; all rounds of a task in flight
        const int c16 = w / E, c16p = wp / E;
        const int rpr = NP / c16p;
        const int sub = DFIX > 0 ? i / c16p : (int)(((unsigned)i * ((1u << 16) / (unsigned)c16p + 1u)) >> 16);
        const int c = i - sub * c16p;
        const bool lane_on = sub < rpr;
        constexpr int RPRFIX = NP / (DSTFIX / E);
        constexpr int U = (KFIX > 0 && DFIX > 0) ? (KFIX + RPRFIX - 1) / RPRFIX : 6;
        const int64_t* idxl = idxh + sub;
        T* xdst = Xh + sub * xs + c * E;
        for (int r0 = 0; r0 < k; r0 += U * rpr) {
          V v[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int row = r0 + u * rpr + sub;
            v[u] = V(0);
            if (lane_on && row < k && c < c16)
              v[u] = *reinterpret_cast<const V*>(feat_nn + idxl[r0 + u * rpr] + d0 + c * E);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int row = r0 + u * rpr + sub;
            if (lane_on && row < k) *reinterpret_cast<V*>(xdst + (r0 + u * rpr) * xs) = v[u];
          }
        }
        // the query row, and zero rows for the padding slots (their distances are masked later,
        // zeros only keep them finite)
        if (i < c16p) {
          V vq = V(0);
          if (i < c16) vq = *reinterpret_cast<const V*>(feat_q + idxh[q] + d0 + i * E);
          *reinterpret_cast<V*>(Xh + q * xs + i * E) = vq;
        }
        if (!nopad)
          for (int t = i; t < (q - k) * c16p; t += NP) {
            const int row = k + t / c16p;
            *reinterpret_cast<V*>(Xh + row * xs + (t % c16p) * E) = V(0);
          }
      } else {
        const unsigned magic = (1u << 20) / (unsigned)wp + 1u;
        for (int t = i; t < NP * wp; t += NP) {
          const int row = (int)(((unsigned)t * magic) >> 20);
          const int c = t - row * wp;
          T v = T(0);
          if (c < w && (row < k || row == q)) v = ((row < k ? feat_nn : feat_q) + idxh[row] + d0)[c];
          Xh[row * xs + c] = v;
        }
      }
      if (aniso) {
        if constexpr (PIPED) {
#pragma unroll
          for (int u = 0; u < ILN; ++u)
            if (u * 64 + lane < wp) ilbuf[u * 64 + lane] = il_pre[u];
        } else {
          for (int c = lane; c < wp; c += 64) ilbuf[c] = c < w ? T(1) / ls[d0 + c] : T(0);
        }
      }
      __syncthreads();

      if constexpr (GRAM) {
        // ---- phase 1b: centre the rows on the query, in place; squared norms --------------------
        // Row i becomes a' = (a - q) [x inverse length scales]; |a'|^2 goes to the norm array (below), where the
        // lanes that pair with the row pick it up.  The query
        // row becomes exactly zero (norm 0), so a pair with the query is |a'|^2: the cross-covariances
        // keep the difference form.  All reads of the query row are issued before any lane's writes (one
        // wave, LDS executes in order).  Slots without features (responses, padding) are left alone:
        // their pairs are dropped or masked.
        // (-DMGP_OWN_REG=1: the centred row of the lane's own slot stays in registers -- it is own row 0 of
        // the pair scheme, so the Gram loop below need not read it back: 10 of 330 LDS instructions at
        // d = 40 for 38 more VGPRs.  Measured 1.826 vs 1.819 ms: nothing; off.)
        // The squared norms go to a compact array (one entry per slot, in the column-buffer space, which is free
        // between two eliminations): behind the rows -- stride 44 floats at d = 40 -- eight of their 32 banks served
        // a half-wave's ds_read_b32, a four-way conflict on each of the eight norm reads of a lane (~50 of the 136
        // conflict cycles per task of round 3).
        // fp32: every norm is stored TWICE, a cycle length apart, so that the norm of row (i + o) mod cycle is entry
        // i + o: one lane-linear address and immediate offsets instead of eight wrapped per-lane addresses (which
        // cost the headline kernel five spilled registers).
        constexpr bool NDUP = sizeof(T) == 4;
        constexpr int NCYC = MODM ? M : NP;
        T* nrmh = (TF ? tile + kmat_base : colbuf) + h * (NDUP ? 2 * NP : NP);
        const int iw = MODM ? wrap(i) : i;
        auto put_norm = [&](bool has, T n2) {
          if (MODM && !has) return;  // (modulo scheme: idle lanes repeat a live one; they store nothing)
          const T v = has ? n2 : num<T>::inf();
          nrmh[iw] = v;
          if constexpr (NDUP) nrmh[iw + NCYC] = v;
        };
        constexpr int NCF = DFIX > 0 ? DSTFIX / E : 1;  // 16-byte groups per row (static shapes)
        constexpr bool OWNREG = MGP_OWN_REG && DFIX > 0;
        V x[DFIX > 0 ? NCF : 1];
        if (MGP_PHASE(g, 2)) {
          const bool has = i < k || i == q;
          T* xrow = Xh + (MODM ? wrap(i) : i) * xs;  // (modulo scheme: idle lanes repeat a live one; they store nothing)
          const T* qrow = Xh + q * xs;
          if constexpr (DFIX > 0 && (FOLD || (BWD && NCF > 16)) && NCF > 10 && !OWNREG) {
            // (FOLD, long rows: the parked rows of the pair's first task leave no room for a whole row and a
            // whole query row in registers -- four groups at a time; a chunk of the query row is overwritten
            // (by lane q) only after every lane has read it)
            ACC n4[4] = {ACC(0), ACC(0), ACC(0), ACC(0)};
#pragma unroll
            for (int c0 = 0; c0 < NCF; c0 += 4) {
              V xc[4], qc[4];
#pragma unroll
              for (int u = 0; u < 4; ++u)
                if (c0 + u < NCF) {
                  xc[u] = *reinterpret_cast<const V*>(xrow + (c0 + u) * E);
                  qc[u] = *reinterpret_cast<const V*>(qrow + (c0 + u) * E);
                }
#pragma unroll
              for (int u = 0; u < 4; ++u)
                if (c0 + u < NCF) {
                  xc[u] = vsub(xc[u], qc[u]);
                  if (aniso) xc[u] = xc[u] * *reinterpret_cast<const V*>(ilbuf + (c0 + u) * E);
                  norm_accum(n4[u], xc[u]);
                }
              if (has) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                  if (c0 + u < NCF) *reinterpret_cast<V*>(xrow + (c0 + u) * E) = xc[u];
              }
              __builtin_amdgcn_sched_barrier(0);  // keep the chunks apart (hoisted loads are what spills)
            }
            const ACC n2 = (n4[0] + n4[1]) + (n4[2] + n4[3]);
            put_norm(has, acc_total(n2));  // (no features: infinite -- its pairs must not trip the cancellation guard)
          } else if constexpr (DFIX > 0) {
            V qv[NCF];
#pragma unroll
            for (int c = 0; c < NCF; ++c) {
              x[c] = *reinterpret_cast<const V*>(xrow + c * E);
              qv[c] = *reinterpret_cast<const V*>(qrow + c * E);
            }
#pragma unroll
            for (int c = 0; c < NCF; ++c) {
              x[c] = vsub(x[c], qv[c]);  // (v_pk_add_f32 with neg modifiers; a plain vector subtract is split into v_sub_f32)
              if (aniso) x[c] = x[c] * *reinterpret_cast<const V*>(ilbuf + c * E);
            }
            if (has) {
#pragma unroll
              for (int c = 0; c < NCF; ++c) *reinterpret_cast<V*>(xrow + c * E) = x[c];
            }
            // four independent partial sums (a single chain of 2 NCF dependent packed FMAs costs a wait state each)
            ACC n4[4] = {ACC(0), ACC(0), ACC(0), ACC(0)};
#pragma unroll
            for (int c = 0; c < NCF; ++c) norm_accum(n4[c & 3], x[c]);
            const ACC n2 = (n4[0] + n4[1]) + (n4[2] + n4[3]);
            put_norm(has, acc_total(n2));
          } else {
            ACC n2 = ACC(0);
            for (int c0 = 0; c0 < wp; c0 += CH) {
              V x0 = *reinterpret_cast<const V*>(xrow + c0), x1 = *reinterpret_cast<const V*>(xrow + c0 + E);
              const V q0 = *reinterpret_cast<const V*>(qrow + c0), q1 = *reinterpret_cast<const V*>(qrow + c0 + E);
              x0 = vsub(x0, q0);
              x1 = vsub(x1, q1);
              if (aniso) {
                x0 = x0 * *reinterpret_cast<const V*>(ilbuf + c0);
                x1 = x1 * *reinterpret_cast<const V*>(ilbuf + c0 + E);
              }
              norm_accum(n2, x0);
              norm_accum(n2, x1);
              // (lane q writes zeros over the chunk every lane has just read; the next chunk's reads come
              // after this store in program order)
              if (has) {
                *reinterpret_cast<V*>(xrow + c0) = x0;
                *reinterpret_cast<V*>(xrow + c0 + E) = x1;
              }
            }
            put_norm(has, acc_total(n2));
          }
        }
        __syncthreads();
        MGP_WAVE_T(1)
        // ---- phase 2 (Gram form): acc = a'.b' per pair ------------------------------------------
        if (MGP_PHASE(g, 2)) {
#pragma unroll
          for (int c0 = 0; c0 < (DFIX > 0 ? DSTFIX : wp); c0 += CH) {
            V own0[BA], own1[BA];
#pragma unroll
            for (int j = 0; j < BA; ++j) {
              if (OWNREG && j == 0) {  // own row 0 is the lane's own slot (o_0 = 0): still in registers
                own0[0] = x[OWNREG ? c0 / E : 0];
                own1[0] = x[OWNREG ? c0 / E + 1 : 0];
                continue;
              }
              const T* xj = DPRE ? tile + down[DPRE ? j : 0] + c0 : Xh + wrap(i + own_offset(j)) * xs + c0;
              own0[j] = *reinterpret_cast<const V*>(xj);
              own1[j] = *reinterpret_cast<const V*>(xj + E);
            }
#pragma unroll
            for (int s = 1; s <= BP; ++s) {
              const T* xo = DPRE ? tile + dpar[DPRE ? s - 1 : 0] + c0 : Xh + wrap(i + s) * xs + c0;
              const V o0 = *reinterpret_cast<const V*>(xo);
              const V o1 = *reinterpret_cast<const V*>(xo + E);
              gram_block<BA, BP>(&acc[s - 1], own0, o0);
              gram_block<BA, BP>(&acc[s - 1], own1, o1);
            }
          }
          // squared distance of a pair: |a'|^2 + |b'|^2 - 2 a'.b', clamped at zero; left in acc[].x
          // (acc[].y = 0) so that the covariance stage below reads it like a difference-form sum
          // (partner-major: one partner norm live at a time -- loading all BA + BP norms first cost the d = 64
          // instantiation a spilled register)
          T nown[BA];
#pragma unroll
          for (int j = 0; j < BA; ++j) nown[j] = NDUP ? nrmh[iw + own_offset(j)] : nrmh[wrap(i + own_offset(j))];
          T guard = T(1);
#pragma unroll
          for (int s = 0; s < BP; ++s) {
            const T np1 = NDUP ? nrmh[iw + s + 1] : nrmh[wrap(i + s + 1)];
#pragma unroll
            for (int j = 0; j + 1 < BA; j += 2)
              gram_finish2(acc[j * BP + s], acc[(j + 1) * BP + s], nown[j] + np1, nown[j + 1] + np1, guard);
            if constexpr (BA % 2 == 1) gram_finish1(acc[(BA - 1) * BP + s], nown[BA - 1] + np1, guard);
          }
          // ---- phase 2G: the cancellation guard tripped (mgp_wave_common.h) -- some pair of this task lies much
          // closer together than its rows lie to the query.  Rare (a query far outside a tight cluster), and
          // wave-uniform: the task's distances again, in the difference form, on the same centred rows
          // (a' - b' = a - b; under Anisotropy the rows are scaled already).  One 16-byte group per iteration,
          // rolled: this path must cost the common one no registers.
          const unsigned long long tripped = gram_guard_lanes(guard);
          if (tripped != 0) {
            // (pair by pair, one 16-byte group per iteration: two row addresses and three groups of registers
            // live -- blocked like the common path it costs the headline kernel five spilled registers.  Only the
            // neighbourhood(s) of the wave whose own guard tripped take the new values: what a neighbourhood
            // returns must not depend on its wave-mates)
            const bool mine = NH == 1 || ((tripped >> (h * NP)) & ((NP == 32 ? 0xFFFFFFFFull : 0xFFFFull))) != 0;
            const int ngrp = (DFIX > 0 ? DSTFIX : wp) / E;
            if (mine) {  // (the other neighbourhoods' lanes sit this out)
#pragma unroll
              for (int s = 0; s < NS; ++s) {
                const T* xa = DPRE ? tile + down[DPRE ? s / BP : 0] : Xh + wrap(i + own_offset(s / BP)) * xs;
                const T* xb = DPRE ? tile + dpar[DPRE ? s % BP : 0] : Xh + wrap(i + s % BP + 1) * xs;
                ACC sum = ACC(0);
#pragma nounroll
                for (int c = 0; c < ngrp; ++c)
                  accum(sum, vsub(*reinterpret_cast<const V*>(xa + c * E), *reinterpret_cast<const V*>(xb + c * E)));
                gram_from_diff(sum);
                acc[s] = sum;
              }
            }
          }
        }
      } else
      if (MGP_PHASE(g, 2)) {
        if (aniso) {
          // Anisotropy: every row is scaled by the inverse length scales ONCE, in place (x / l), so that
          // the pair loop below is the isotropic one: one multiply per row element instead of one per
          // pair element (k = 50, d = 8: 8 instead of 200 multiplies per lane and neighbourhood).
          if (i < k || i == q) {  // rows with features (the pipelined kernels keep ilbuf in a response slot's row)
            T* xrow = Xh + i * xs;
            for (int c0 = 0; c0 < wp; c0 += E) {
              V x = *reinterpret_cast<const V*>(xrow + c0);
              x = x * *reinterpret_cast<const V*>(ilbuf + c0);
              *reinterpret_cast<V*>(xrow + c0) = x;
            }
          }
          __syncthreads();
        }
        if constexpr (sizeof(T) == 8 && MGP_F64_DIST_HALF) {
          // fp64: one 16-byte group of the own rows at a time (the same reads and arithmetic as the two-group form
          // below with half the own-row registers: 20 instead of 40 at five own rows)
#pragma unroll
          for (int c0 = 0; c0 < (DFIX > 0 ? DSTFIX : wp); c0 += E) {
            V own[BA];
#pragma unroll
            for (int j = 0; j < BA; ++j)
              own[j] = *reinterpret_cast<const V*>((DPRE ? tile + down[DPRE ? j : 0] : Xh + wrap(i + own_offset(j)) * xs) + c0);
#pragma unroll
            for (int s = 1; s <= BP; ++s) {
              const V o = *reinterpret_cast<const V*>((DPRE ? tile + dpar[DPRE ? s - 1 : 0] : Xh + wrap(i + s) * xs) + c0);
#pragma unroll
              for (int j = 0; j < BA; ++j) accum(acc[j * BP + s - 1], vsub(own[j], o));
            }
            if (MGP_F64_DIST_HALF == 2) __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll
          for (int c0 = 0; c0 < (DFIX > 0 ? DSTFIX : wp); c0 += CH) {
            V own0[BA], own1[BA];
#pragma unroll
            for (int j = 0; j < BA; ++j) {
              const T* xj = DPRE ? tile + down[DPRE ? j : 0] + c0 : Xh + wrap(i + own_offset(j)) * xs + c0;
              own0[j] = *reinterpret_cast<const V*>(xj);
              own1[j] = *reinterpret_cast<const V*>(xj + E);
            }
#pragma unroll
            for (int s = 1; s <= BP; ++s) {
              const T* xo = DPRE ? tile + dpar[DPRE ? s - 1 : 0] + c0 : Xh + wrap(i + s) * xs + c0;
              const V o0 = *reinterpret_cast<const V*>(xo);
              const V o1 = *reinterpret_cast<const V*>(xo + E);
              if constexpr (sizeof(T) == 4 && BA == 4 && MGP_DIST_ASM) {
                dist_block4(acc[s - 1], acc[BP + s - 1], acc[2 * BP + s - 1], acc[3 * BP + s - 1], own0[0], own0[1],
                            own0[2], own0[3], o0);
                dist_block4(acc[s - 1], acc[BP + s - 1], acc[2 * BP + s - 1], acc[3 * BP + s - 1], own1[0], own1[1],
                            own1[2], own1[3], o1);
              } else {
#pragma unroll
                for (int j = 0; j < BA; ++j) {
                  accum(acc[j * BP + s - 1], vsub(own0[j], o0));
                  accum(acc[j * BP + s - 1], vsub(own1[j], o1));
                }
              }
            }
          }
        }
      }
    }

    MGP_WAVE_T(2)
    // ---- phase 3: covariances, nugget, responses -> exchange matrix -> row per lane ----
#if MGP_XCHG_PRIO || MGP_DIST_PRIO
    __builtin_amdgcn_s_setprio(sizeof(T) == 4 || MGP_F64_SAME_PRIO ? MGP_XCHG_PRIO : MGP_DIST_PRIO);
#endif
    __syncthreads();  // every lane is done reading the feature tile (Kh aliases it)
    V A[DLT ? 1 : NG];
    V Dp[DLT ? NSL : 1];
    int wrow[DLT ? NSL : 1], wcol[DLT ? NSL : 1];
    {
      // re-materialise the slot index here so that the per-offset masks/addresses of this phase
      // are computed now and not kept alive (or spilled) across the distance loop
      int i3 = i;
      asm volatile("" : "+v"(i3));
      T* Kh3 = tile + kmat_base + (NH == 1 ? 0 : (lane / NP) * KMAT);
      if (MGP_PHASE(g, 4)) {
        // fp32: all NS covariances first (independent chains the scheduler can interleave), then the
        // stores -- unconditional: an entry whose row is a response slot (hi > q) goes to the
        // unused padding column of the last row instead of being branched around.
        // fp64 evaluates exp and sqrt in software (~40 instructions and a dozen temporaries per chain):
        // interleaving all NS chains is what the register file cannot hold, so they go in batches of CB,
        // each batch stored before the next starts.
        const int dump = TRI ? KMAT - E : (NP - 1) * KS + NP;  // padding behind the last row / columns NP .. KS-1
        auto put = [&](int s, T v) {  // s = 1 .. NS: pair j * BP + p - 1 = (own row j, partner p)
          if constexpr (XPRE) {
            if (!nopad) v = (xkeep >> (s - 1)) & 1u ? v : T(0);
            if constexpr (XPK) tile[(s - 1) % 2 == 0 ? (xoff[(s - 1) / 2] & 0xFFFF) : ((unsigned)xoff[(s - 1) / 2] >> 16)] = v;
            else tile[xoff[s - 1]] = v;
          } else {
            const int r1 = wrap(i3 + own_offset((s - 1) / BP));
            const int c = wrap(i3 + (s - 1) % BP + 1);
            const int hi = max(r1, c), lo = min(r1, c);
            if (!nopad) v = (lo < k && (hi < k || hi == q)) ? v : T(0);
            Kh3[hi <= q && hi != lo ? rowoff(hi) + lo : dump] = v;
          }
        };
        auto covariances = [&](auto kid, auto mid) {
          constexpr int KID = decltype(kid)::value, MID = decltype(mid)::value;
          if constexpr (KID == MGP_KERNEL_MATERN_GEN) {
            // general smoothness (fp32 kernels with per-lane pair tables; the launcher admits nothing else):
            // metric arguments of all pairs, then the node loop shared by the wave (mgp_wave_common.h)
            if constexpr (sizeof(T) == 4 && XPRE) {
              float kv[NS];
#pragma unroll
              for (int s = 0; s < NS; ++s) {
                float sqd;
                if constexpr (GRAM) sqd = gram_sq(acc[s]);
                else sqd = acc_total(acc[s]);
                kv[s] = (MID == MGP_METRIC_L2 ? sqrt_fast(sqd) : sqd) * post_scale;
              }
              matern_gen_eval<NS>(kv, xkeep, gtab, (float)a.smoothness, g.gen_h, g.gen_lc);
#pragma unroll
              for (int s = 1; s <= NS; ++s) put(s, kv[s - 1]);
            } else if constexpr (sizeof(T) == 8 && XPRE) {
              // fp64 (round 4): the same rule with a finer step and the software exp, CB pairs at a time
#ifndef MGP_F64_GEN_BATCH
#define MGP_F64_GEN_BATCH 3  // (seven doubles per chain in the node loop: five chains cost the config-4 kernel two spilled registers)
#endif
              constexpr int CB = MGP_F64_GEN_BATCH;
              const double* gtab64 = reinterpret_cast<const double*>(smem + g.gen_tab);
              // which of the lane's pairs are real (both rows neighbours, or neighbour x query): recomputed here -- as a
              // register kept across the task loop it would cost the fixed-smoothness kernels two spilled registers
              unsigned keep = 0;
#pragma unroll
              for (int s = 1; s <= NS; ++s) {
                const int r1 = wrap(i3 + own_offset((s - 1) / BP)), c = wrap(i3 + (s - 1) % BP + 1);
                const int hi = max(r1, c), lo = min(r1, c);
                if (lo < k && (hi < k || hi == q) && hi != lo) keep |= 1u << (s - 1);
              }
#pragma unroll
              for (int s0 = 0; s0 < NS; s0 += CB) {
                double kv[CB];
#pragma unroll
                for (int u = 0; u < CB; ++u) {
                  const int su = s0 + u < NS ? s0 + u : NS - 1;
                  kv[u] = GRAM ? gram_sq(acc[su]) : acc_total(acc[su]);
                }
                metric_batch64<CB, MID>(kv, post_scale);
                matern_gen_batch64<CB>(kv, (keep >> s0) & ((1u << CB) - 1u), gtab64, a.smoothness, g.gen_h64, g.gen_lc64,
                                       g.gen_xmin64);
#pragma unroll
                for (int u = 0; u < CB; ++u)
                  if (s0 + u < NS) put(s0 + u + 1, kv[u]);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          } else if constexpr (sizeof(T) == 4) {
            T kv[NS];
            // (Gram form: the squared distance already sits in acc[].x)
            auto sq = [&](int s) {
              if constexpr (GRAM) return gram_sq(acc[s]);
              else return acc_total(acc[s]);
            };
#pragma unroll
            for (int s = 0; s + 1 < NS; s += 2) {
              const f2 kk = cov_from_sqdist2(f2{sq(s), sq(s + 1)}, KID, MID, post_scale);
              kv[s] = kk.x;
              kv[s + 1] = kk.y;
            }
            if constexpr (NS % 2 == 1) kv[NS - 1] = cov_from_sqdist<T>(sq(NS - 1), KID, MID, post_scale);
#pragma unroll
            for (int s = 1; s <= NS; ++s) put(s, kv[s - 1]);
          } else {
            constexpr int CB = BWD ? MGP_BWD_COV_BATCH : MGP_F64_COV_BATCH;
#pragma unroll
            for (int s0 = 0; s0 < NS; s0 += CB) {
              T kv[CB];
#ifndef MGP_COV_BATCH64
#define MGP_COV_BATCH64 1
#endif
              if constexpr (sizeof(T) == 8 && MGP_COV_BATCH64) {
#pragma unroll
                for (int u = 0; u < CB; ++u) {
                  const int su = s0 + u < NS ? s0 + u : NS - 1;  // (a short last batch repeats its last pair)
                  kv[u] = GRAM ? gram_sq(acc[su]) : acc_total(acc[su]);
                }
                cov_batch64<CB, KID, MID>(kv, post_scale);
              } else {
#pragma unroll
              for (int u = 0; u < CB; ++u)
                if (s0 + u < NS) kv[u] = cov_from_sqdist<T>(GRAM ? gram_sq(acc[s0 + u]) : acc_total(acc[s0 + u]), KID, MID, post_scale);
              }
#pragma unroll
              for (int u = 0; u < CB; ++u)
                if (s0 + u < NS) put(s0 + u + 1, kv[u]);
              __builtin_amdgcn_sched_barrier(0);  // keep the batches apart
            }
          }
        };
        if constexpr (GEN64) {  // this instantiation serves kernel_id == MGP_KERNEL_MATERN_GEN only
          if (a.metric_id == MGP_METRIC_L2) covariances(ic<MGP_KERNEL_MATERN_GEN>{}, ic<MGP_METRIC_L2>{});
          else covariances(ic<MGP_KERNEL_MATERN_GEN>{}, ic<MGP_METRIC_F2>{});
        } else if constexpr (sizeof(T) == 4) {
          kernel_dispatch_gen(a.kernel_id, a.metric_id, covariances);
        } else {
          kernel_dispatch(a.kernel_id, a.metric_id, covariances);
        }
      }
      if (NPL == NP || i3 < NPL) Kh3[eoff(i3, i3)] = i3 < k ? T(1) + myeps : (i3 <= q ? T(1) : T(0));
      // response rows: lower-triangle columns only (a packed row ends at its diagonal)
      if (!TRI || i3 <= q + 1) Kh3[eoff(q + 1, i3)] = myy0;
      if (PACKED && !a.targets_batch) {
#pragma unroll
        for (int r = 1; r < E; ++r)
          if (r < R && (!TRI || i3 <= q + 1 + r)) Kh3[eoff(q + 1 + r, i3)] = i3 < k ? myyv[r] : T(0);
      } else {
        for (int r = 1; r < R; ++r)
          if (!TRI || i3 <= q + 1 + r) Kh3[eoff(q + 1 + r, i3)] = i3 < k ? targets[mytg * (int64_t)R + r] : T(0);
      }
    }
    __syncthreads();
#if MGP_CHOL_PRIO && MGP_PRIO_EARLY_RAISE
    __builtin_amdgcn_s_setprio(MGP_CHOL_PRIO);  // row read-back and the next task's gather already at elimination priority
#endif
    if constexpr (FOLD) {
      // the half-wave of this task of the pair picks its rows up, sixteen lanes per neighbourhood
      if ((lane >> 5) == fold_sub) {
        const T* Kq = tile + (NH == 1 ? 0 : (lane >> LOGH) & (NH - 1)) * KMAT;
        const int lh = lane & (HALF - 1);
#pragma unroll
        for (int c4 = 0; c4 < NGS; ++c4) FS[c4] = *reinterpret_cast<const V*>(Kq + rowoff(lh) + c4 * E);
#pragma unroll
        for (int c4 = 0; c4 < NG; ++c4) FL[c4] = *reinterpret_cast<const V*>(Kq + rowoff(HALF + lh) + c4 * E);
      }
    } else if constexpr (DLT) {
      // the lane's pairs: slot s holds pair 64 s + lane of the dealt order (lane-linear, conflict-free); and what each
      // needs in an elimination step (g_dlt_meta above)
#pragma unroll
      for (int s = 0; s < NSL; ++s) {
        typedef int i2 __attribute__((ext_vector_type(2)));
        const i2 m = *reinterpret_cast<const i2*>(&g_dlt_meta<DLT ? NPL : 2>.rc[64 * s + lane][0]);
        wrow[s] = m.x;
        wcol[s] = m.y;
      }
#pragma unroll
      for (int s = 0; s < NSL; ++s) Dp[s] = *reinterpret_cast<const V*>(Kh + 2 * (64 * s) + 2 * lane);
    } else {
      const T* myrow = Kh + rowoff(NPL == NP ? i : min(i, NPL - 1));  // (idle lanes re-read the last live row)
#pragma unroll
      for (int c4 = 0; c4 < NG; ++c4) A[c4] = *reinterpret_cast<const V*>(myrow + c4 * E);
    }

    // the next task's rows are requested now: their latency hides behind the factorisation, and
    // the registers they land in are not live during the (register-hungry) distance phase
    t2 = seq_gen();  // the task after the next: its index row is requested now
    if (PIPE && !BWD && t1 >= 0) {
      pipe_issue(t1, fix_index(next_idx, t1, h, i), lane);
      if (t2 >= 0) next_idx = load_index(t2, h, i);
    }

    if constexpr (FOLD) {
      // ---- phase 4F: folded elimination of a PAIR of tasks ---------------------------------------------
      // Row-per-lane elimination spends half its lanes on rows that are finished and half of the rest on the
      // upper triangle.  Folded: lane l of a quarter-wave owns rows l (columns 0 .. 15 matter) and 16 + l of one
      // neighbourhood -- 12 instead of 8 register groups, but four neighbourhoods per wave: per step and
      // neighbourhood half the column reads and, summed over the steps, 182 instead of 284 group updates.
      // The first task of a pair parks its rows in lanes 0 .. 31 and goes on to the second; the elimination
      // runs when both are there (or the workgroup has no task left).
      const bool lastt = t1 < 0;
      MGP_WAVE_T(3)
      if (fold_sub == 0 && !lastt) {
        fold_sub = 1;
        fold_task_a = task;
        continue;
      }
      const bool have_b = fold_sub == 1;
      if (!have_b) fold_task_a = task;
      fold_sub = 0;
      bool bad = false;
#if MGP_CHOL_PRIO
      __builtin_amdgcn_s_setprio(MGP_CHOL_PRIO);
#endif
      if (MGP_PHASE(g, 8)) {
        const int lh = lane & (HALF - 1);
        T* colq = colbuf + (lane >> LOGH) * NP;  // the neighbourhood's column buffer (together: the row-address array's space)
        if constexpr (sizeof(T) == 4) {
          colq[HALF + lh] = FL[0][0];
          colq[lh] = FS[0][0];
          V piv = *reinterpret_cast<const V*>(colq);
#pragma unroll
          for (int j = 0; j < KFIX; ++j) {
            const int g0 = j / E, e0 = j % E;
            const bool sh = j < HALF;  // (compile-time after unrolling) the short rows are still being eliminated
            const T aL = FL[g0][e0];
            const T aS = sh ? FS[g0 < NGS ? g0 : 0][e0] : T(0);
            V col[NG];
            col[g0] = piv;
#pragma unroll
            for (int c4 = g0 + 1; c4 < NG; ++c4) col[c4] = *reinterpret_cast<const V*>(colq + c4 * E);
            const T p = piv[e0];
            bad = bad || !(p > T(0));
            const T rp = pivot_rcp(p);
            const V ntL = V(-aL * rp), ntS = V(-aS * rp);
            const int g1 = (j + 1 < KFIX ? j + 1 : j) / E;
            FL[g1] = col[g1] * ntL + FL[g1];
            if (sh && g1 < NGS) FS[g1] = col[g1] * ntS + FS[g1];
            if (j + 1 < KFIX) {  // look-ahead: column j + 1 is complete, post it and ask for its pivot group
              colq[HALF + lh] = FL[g1][(j + 1) % E];
              if (j + 1 < HALF) colq[lh] = FS[g1][(j + 1) % E];
              piv = *reinterpret_cast<const V*>(colq + g1 * E);
            }
#pragma unroll
            for (int c4 = g0; c4 < NG; ++c4)
              if (c4 != g1) FL[c4] = col[c4] * ntL + FL[c4];
            if (sh) {
#pragma unroll
              for (int c4 = g0; c4 < NGS; ++c4)
                if (c4 != g1) FS[c4] = col[c4] * ntS + FS[c4];
            }
          }
        } else {
          // fp64: no look-ahead, the column streamed GC groups at a time (as in phase 4), each group serving
          // the long and -- while it has one -- the short row
#pragma unroll
          for (int j = 0; j < KFIX; ++j) {
            const int g0 = j / E, e0 = j % E;
            const bool sh = j < HALF;
            const T aL = FL[g0][e0];
            const T aS = sh ? FS[g0 < NGS ? g0 : 0][e0] : T(0);
            colq[HALF + lh] = aL;
            if (sh) colq[lh] = aS;
            const V cp = *reinterpret_cast<const V*>(colq + g0 * E);
            const T p = cp[e0];
            bad = bad || !(p > T(0));
            const T rp = pivot_rcp(p);
            const V ntL = V(-aL * rp), ntS = V(-aS * rp);
            FL[g0] = cp * ntL + FL[g0];
            if (sh) FS[g0 < NGS ? g0 : 0] = cp * ntS + FS[g0 < NGS ? g0 : 0];
            constexpr int GC = MGP_F64_GC;
#pragma unroll
            for (int c0 = g0 + 1; c0 < NG; c0 += GC) {
              V cv[GC];
#pragma unroll
              for (int u = 0; u < GC; ++u)
                if (c0 + u < NG) cv[u] = *reinterpret_cast<const V*>(colq + (c0 + u) * E);
#pragma unroll
              for (int u = 0; u < GC; ++u)
                if (c0 + u < NG) {
                  FL[c0 + u] = cv[u] * ntL + FL[c0 + u];
                  if (sh && c0 + u < NGS) FS[c0 + u] = cv[u] * ntS + FS[c0 + u];
                }
            }
          }
        }
      }
#if MGP_CHOL_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      MGP_WAVE_T(4)
      // Schur block: the query row KFIX and the response rows behind it are long rows (lanes KFIX - HALF ...)
      if constexpr (RFIX > 1) {
        constexpr int QF = KFIX;
        const int l16 = lane & (HALF - 1);
        const bool second = (lane >> 5) != 0;
        const int64_t nbq = (int64_t)(second ? task : fold_task_a) * NH + (NH == 1 ? 0 : (lane >> LOGH) & (NH - 1));
        const bool liveq = nbq < a.b && (!second || have_b);
        T* mean = static_cast<T*>(a.mean);
        T* var = static_cast<T*>(a.var);
        T* yk = static_cast<T*>(a.ykinvy);
        const T sq = FL[QF / E][QF % E];  // column q of the lane's long row
        T sd = T(0);                       // its diagonal: column HALF + l, picked out by a compare-select sweep
        if (yk) {
#pragma unroll
          for (int c = QF + 1; c < NG * E; ++c) sd = c == HALF + l16 ? FL[c / E][c % E] : sd;
        }
        if (liveq) {
          if (l16 == QF - HALF) {
            *(var + nbq) = bad ? num<T>::nan() : sq;
            if (bad && a.info) atomicAdd(a.info, 1);
          } else if (l16 > QF - HALF && l16 <= QF - HALF + RFIX) {
            const int r = l16 - (QF - HALF) - 1;
            *(mean + (nbq * RFIX + r)) = bad ? num<T>::nan() : -sq;
            if (yk) *(yk + (nbq * RFIX + r)) = bad ? num<T>::nan() : -sd;
          }
        }
      } else {
        constexpr int QF = KFIX, YF = KFIX + 1;
        const int l16 = lane & (HALF - 1);
        const bool second = (lane >> 5) != 0;
        const int64_t nbq = (int64_t)(second ? task : fold_task_a) * NH + (NH == 1 ? 0 : (lane >> LOGH) & (NH - 1));
        const bool liveq = nbq < a.b && (!second || have_b);
        T* mean = static_cast<T*>(a.mean);
        T* var = static_cast<T*>(a.var);
        T* yk = static_cast<T*>(a.ykinvy);
        const T sq = FL[QF / E][QF % E], sy = FL[YF / E][YF % E];
        if (liveq) {
          if (l16 == QF - HALF) {
            *(var + nbq) = bad ? num<T>::nan() : sq;
            if (bad && a.info) atomicAdd(a.info, 1);
          } else if (l16 == YF - HALF) {
            *(mean + nbq) = bad ? num<T>::nan() : -sq;
            if (yk) *(yk + nbq) = bad ? num<T>::nan() : -sy;
          }
        }
      }
      MGP_WAVE_T(5)
      continue;
    }
    if constexpr (DLT) {
      MGP_WAVE_T(3)
      // ---- phase 4D: elimination on the DEALT lower triangle (fp64, one neighbourhood per wave) --------------
      // Row-per-lane elimination issues an FMA for every column right of the pivot in every lane: 3.8 x the
      // arithmetic of the factorisation (finished rows idle, the upper triangle is updated too), and a 104-register
      // row.  Here the lower triangle of the augmented system is dealt over the lanes in column-major order, two
      // rows of one column per 16-byte group: NSL (11 at k = 50) groups per lane, and because a slot holds 64
      // CONSECUTIVE pairs, all of its columns are finished together -- step j touches slots cs2(j + 1) / 64 .. only:
      // 217 slot updates (434 FMAs) per neighbourhood at k = 50 instead of 688 group updates (1 376 FMAs).
      // What a pair (rows 2r, 2r + 1; column c) needs in step j -- l_2r,j, l_2r+1,j and a_c,j = l_c,j p_j -- it
      // reads from the posted column: one ds_read_b128 and one ds_read_b64 at per-lane addresses (wrow / wcol),
      // i.e. 6 LDS-array cycles per two FMAs against 4 per two for the broadcast reads of the row form -- on
      // a third of the FMAs.  Per step: pivot from its (compile-time) lane by v_readlane, one reciprocal, the
      // column's lanes post l_.,j = a_.,j / p_j (one ds_write_b128), every live slot reads, multiplies once and
      // updates twice.  Entries of finished columns and the odd upper-triangle elements that ride along in
      // pairs receive junk updates; nobody reads them again.  The Schur block lands in three compile-time lanes.
      bool bad = false;
#if MGP_CHOL_PRIO
      __builtin_amdgcn_s_setprio(MGP_CHOL_PRIO);
#endif
      if (MGP_PHASE(g, 8)) {
        // staging area of the posted column: behind the tile rows the next task's gather is filling meanwhile.  The
        // slot(s) that hold column j write ALL their pairs, scaled, in lane order (no predicate, no per-lane store
        // address: pairs of other columns land where nobody reads); pair (r, j) then lies cs2(j) - 64 s0 + r - j/2
        // pairs in, so a consumer's addresses are its own 2 r / c plus a per-step constant.
        T* stg = tile + tile_rows * xs + 64;
        // Column j is posted RAW (a copy of the registers: nothing on the path from the last update of the column to
        // the LDS write), its pivot is fetched from its compile-time lane and the reciprocal chain runs while the
        // write-to-read turnaround is under way; a consumer forms l_2r,j a_c,j = a_2r,j (a_c,j / p_j).
        // Pivot test on the scalar unit: p > 0 and finite <=> 0 < high dword < 0x7FF00000 as an integer (a positive
        // denormal pivot counts as singular too), accumulated as an unsigned maximum of (high dword - 1).
        T nrp = T(0);
        unsigned badm = 0;
        auto post = [&](int j) {
          const int ep = cs2(j);  // pair of (j, j): the first of column j
          const int s0 = ep >> 6, s1 = (cs2(j + 1) - 1) >> 6;  // the slot(s) that hold column j
#pragma unroll
          for (int s = s0; s <= s1; ++s) *reinterpret_cast<V*>(stg + 2 * (64 * (s - s0)) + 2 * lane) = Dp[s];
          if constexpr (BWD && !(MGP_BWD_EXP & 8)) {
            // the column is final: back into the dealt image (its own pairs only -- earlier columns' pairs of the same
            // slot have taken junk updates since they were saved, later ones are not finished)
            // Branch-free (a predicated store makes every step a basic block of its own: 544 spilled registers): the
            // lanes that hold other columns' pairs repeat their staging-area write instead.
#pragma unroll
            for (int s = s0; s <= s1; ++s) {
              T* dstp = wcol[s] == j ? Kh + 2 * (64 * s) + 2 * lane : stg + 2 * (64 * (s - s0)) + 2 * lane;
              *reinterpret_cast<V*>(dstp) = Dp[s];
            }
          }
          const long long pb = __double_as_longlong(Dp[s0][j & 1]);
          const unsigned plo = __builtin_amdgcn_readlane((int)(pb & 0xFFFFFFFFll), ep & 63);
          const unsigned phi = __builtin_amdgcn_readlane((int)(pb >> 32), ep & 63);
          badm = max(badm, phi - 1u);
          nrp = -pivot_rcp(__longlong_as_double(((long long)phi << 32) | plo));
        };
        post(0);
#pragma unroll
        for (int j = 0; j < KFIX; ++j) {
          const T nrpj = nrp;
          const int cj = 2 * (cs2(j) - 64 * (cs2(j) >> 6) - (j >> 1));  // (>= -64: the bias in front of stg)
          const int sl = cs2(j + 1) >> 6;                                 // first slot with a pair of a column right of j
          // all operands of the step are requested together (a read -> wait -> FMA chain per slot would expose an LDS
          // round trip per slot) ...
          // (MGP_DLT_CHUNK slots at a time: the whole step in flight needs 6 registers per slot)
#ifndef MGP_DLT_CHUNK
#define MGP_DLT_CHUNK 0  // slots per chunk; 0 = by shape
#endif
          // (tools/jit_sweep.py at three waves per SIMD, 2 / 4 slots per chunk: k = 49, d = 8: 157.0 / 153.5 M/s -- BASELINE
          // config 4's neighbourhood --, k = 32 and 40: the same, k = 50, d = 16: 124.0 / 125.0, k = 62, d = 16: 84.4 /
          // 87.4; 6: 154, the whole step: 149; profiles/r06_c4_param_sweep.txt)
          constexpr int CHK = MGP_DLT_CHUNK > 0 ? MGP_DLT_CHUNK : (DFIX <= 8 && KFIX <= 52 ? 2 : 4);
          const int n1 = j + 1 < KFIX ? (cs2(j + 2) - 1) >> 6 : sl - 1;  // last slot of column j + 1
#pragma unroll
          for (int sc = sl; sc < NSL; sc += CHK) {
            V wp[CHK];
            T wc[CHK];
#pragma unroll
            for (int u = 0; u < CHK; ++u)
              if (sc + u < NSL) {
                wp[u] = *reinterpret_cast<const V*>(stg + cj + wrow[sc + u]);
                wc[u] = (stg + cj)[wcol[sc + u]];
              }
            // ... LOOK-AHEAD: the slot(s) of column j + 1 are updated first and the column posted (the reads of the
            // first chunk were issued before that write, and a later chunk's reads of the CURRENT column would come
            // after it: the post therefore waits until the last chunk has been requested), the other slots follow
            // while the reciprocal chain and the write-to-read turnaround of the next step are under way
            auto update = [&](int u) {
              const T ngq = wc[u] * nrpj;  // -a_c,j / p_j
              Dp[sc + u][0] = fma_t(wp[u][0], ngq, Dp[sc + u][0]);
              Dp[sc + u][1] = fma_t(wp[u][1], ngq, Dp[sc + u][1]);
            };
            const bool last_chunk = sc + CHK >= NSL;
#pragma unroll
            for (int u = 0; u < CHK; ++u)
              if (sc + u < NSL && sc + u <= n1) update(u);
            if (last_chunk && j + 1 < KFIX) post(j + 1);
#pragma unroll
            for (int u = 0; u < CHK; ++u)
              if (sc + u < NSL && sc + u > n1) update(u);
          }
        }
        bad = badm >= 0x7FEFFFFFu;
      }
#if MGP_CHOL_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      MGP_WAVE_T(4)
      if constexpr (BWD) {
        // ---- phase 5B: a = K^-1 c and u = K^-1 y by back-substitution on the saved factor -------------------------
        // K = L D L^T (unit L, D = the pivots): the dealt image now holds, raw, a_m,j = l_m,j p_j for m >= j -- the
        // diagonal is p_j, rows q and q + 1 are (D L^-1 c)^T and (D L^-1 y)^T.  Lane i owns column i: entry (m, i) lies
        // at lbase + m.  L^T x = D^-1 L^-1 rhs from the last row up: x_m is final when the rows above it are done; it is
        // handed down by v_readlane and lane i < m takes l_m,i x_m off (mgp_backward_wave.hip, phase 5).
        __syncthreads();
        const T* Lh = Kh;
        const int ic = i < KFIX ? i : KFIX - 1;  // (lanes without a neighbour row: any valid column)
        const int lbase = 2 * (cs2(ic) - (ic >> 1));
        const T rp = pivot_rcp(Lh[lbase + ic]);
        T xa = i < KFIX ? Lh[lbase + KFIX] * rp : T(0);
        T xu = i < KFIX ? Lh[lbase + KFIX + 1] * rp : T(0);
        if (!(MGP_BWD_EXP & 4)) {
          constexpr int BB = 8;
#pragma unroll
          for (int mb = (KFIX - 1) / BB * BB; mb >= 0; mb -= BB) {
            T lm[BB];
#pragma unroll
            for (int e = 0; e < BB; ++e)
              if (mb + e >= 1 && mb + e < KFIX) lm[e] = Lh[lbase + mb + e];  // (junk for m <= i: masked below)
#pragma unroll
            for (int e = BB - 1; e >= 0; --e) {
              const int m = mb + e;
              if (m >= 1 && m < KFIX) {
                const T am = lane_value(xa, m), um = lane_value(xu, m);
                const T t = i < m ? lm[e] * rp : T(0);
                xa = fma_t(-t, am, xa);
                xu = fma_t(-t, um, xu);
              }
            }
          }
        }
        // the two vectors where the pair phase picks them up by row.  The query row enters as a = -1, u = 0: with
        //   gK_rc = a_r P_c - u_r Q_c,  P_c = 2 gv a_c - gm u_c,  Q_c = gm a_c + 2 gy u_c
        // (= 2 gv a_r a_c - gm (a_r u_c + a_c u_r) - 2 gy u_r u_c, symmetric) a pair with the query row gives
        // gm u_c - 2 gv a_c, the cotangent of the cross-covariance, with no case distinction.
        T* avec = tile + tile_rows * xs;  // (the staging area of the posted columns: dead)
        T* uvec = avec + 64;
        __syncthreads();
        avec[lane] = i < KFIX ? xa : (i == KFIX ? T(-1) : T(0));
        uvec[lane] = i < KFIX ? xu : T(0);
        __syncthreads();
        const T gmv = (live && a.bwd_gmean) ? (a.R > 1 ? T(1) : static_cast<const T*>(a.bwd_gmean)[nb0]) : T(0);  // (R > 1: folded into the column)
        const T gvv = (live && a.bwd_gvar) ? static_cast<const T*>(a.bwd_gvar)[nb0] : T(0);
        const T gyv = (live && a.bwd_gyk) ? static_cast<const T*>(a.bwd_gyk)[nb0] : T(0);
        const bool skip = bad || !live;  // (cotangents of a neighbourhood that did not factorise are left untouched)

        // ---- phase 6B: pair cotangents q_rc = gK_rc dk/dacc_rc, in the pair scheme's own layout (the kept squared
        // distances acc[s] are replaced by q) -- each unordered pair once ------------------------------------------
        T liso = T(0);
        // (Anisotropy) per-feature sums of q_rc (z_rf - z_cf)^2 over the lane's pairs, taken on the scaled rows the tile
        // still holds RIGHT WHERE q_rc is formed: kept for a separate sweep, the NS q values cost 2 NS registers
        // through the covariance-derivative batches -- 128 spill slots and 30.4 instead of 27.0 ms at k = 50, d = 8
        constexpr int DGF = DSTFIX / E;
        V s2h[DGF];  // (unused under SWEEP_LATE)
        const bool sweep_any = a.bwd_gls != nullptr && aniso && !(MGP_BWD_EXP & 1);  // (uniform)
        // ... except where keeping the q values and sweeping afterwards -- the partner rows of a lane read once for all
        // its own rows -- fits the registers: BASELINE config 4's own shape 27.0 against 30.8 ms per 2 M; by shape
        // (tools/jit_sweep.py, 1 M neighbourhoods, late / in the loop): k = 40, d = 8: 10.5 / 10.9 ms, k = 40, d = 16:
        // 13.2 / 15.1, k = 50, d = 16: 19.0 / 22.6 -- but k = 62, d = 8: 36.7 / 30.6, and from d = 24 on ten times slower
        // (the q values and 2 d sums: scratch)
        constexpr bool SWEEP_LATE = MGP_BWD_SWEEP_LATE < 0 ? (KFIX <= 52 && DFIX <= 16) : MGP_BWD_SWEEP_LATE != 0;
        const bool sweep = sweep_any && !SWEEP_LATE;
        if constexpr (!SWEEP_LATE) {
#pragma unroll
          for (int c4 = 0; c4 < DGF; ++c4) s2h[c4] = V(0);
        }
        {
          T ar[BA], ur[BA], Pc[BP], Qc[BP];
#pragma unroll
          for (int j = 0; j < BA; ++j) {
            const int r = wrap(i + own_offset(j));
            ar[j] = avec[r];
            ur[j] = uvec[r];
          }
#pragma unroll
          for (int p = 0; p < BP; ++p) {
            const int c = wrap(i + p + 1);
            const T ac = avec[c], uc = uvec[c];
            Pc[p] = T(2) * gvv * ac - gmv * uc;
            Qc[p] = gmv * ac + T(2) * gyv * uc;
          }
          const bool lane_on = i < M;  // (modulo scheme: the idle lanes repeat live ones)
          constexpr unsigned long long FIRSTM = wave_pair_first_mask(NS, BP, M), HALFM = wave_pair_half_mask(NS, BP, M);
          if (!(MGP_BWD_EXP & 2))
          kernel_dispatch(a.kernel_id, a.metric_id, [&](auto kid, auto mid) {
            constexpr int KID = decltype(kid)::value, MID = decltype(mid)::value;
            constexpr int CB = MGP_BWD_COV_BATCH;
#pragma unroll
            for (int s0 = 0; s0 < NS; s0 += CB) {
              T dv[CB];
#pragma unroll
              for (int u = 0; u < CB; ++u) dv[u] = acc[s0 + u < NS ? s0 + u : NS - 1];
              dcov_batch64<CB, KID, MID>(dv, post_scale);
#pragma unroll
              for (int u = 0; u < CB; ++u) {
                if (s0 + u < NS) {
                  const int sidx = s0 + u;
                  const int jj = sidx / BP, pp = sidx % BP;
                  bool on = lane_on && ((xkeep >> sidx) & 1u) != 0 && ((FIRSTM >> sidx) & 1ull) != 0;
                  if ((HALFM >> sidx) & 1ull)  // (the half-way class: the end with r1 < c)
                    on = on && wrap(i + own_offset(jj)) < wrap(i + pp + 1);
                  const T gK = ar[jj] * Pc[pp] - ur[jj] * Qc[pp];
                  const T qv = on ? gK * dv[u] : T(0);
                  liso = fma_t(qv, acc[sidx], liso);  // (difference form: every kept distance is finite, 0 x finite = 0)
                  if constexpr (SWEEP_LATE) acc[sidx] = qv;
                  if constexpr (!SWEEP_LATE)
                  if (sweep) {
                    const T* xa_ = Xh + wrap(i + own_offset(jj)) * xs;
                    const T* xb_ = Xh + wrap(i + pp + 1) * xs;
                    const V qq = V(qv);
#pragma unroll
                    for (int c4 = 0; c4 < DGF; ++c4) {
                      const V dz = *reinterpret_cast<const V*>(xa_ + c4 * E) - *reinterpret_cast<const V*>(xb_ + c4 * E);
                      s2h[c4] = (dz * qq) * dz + s2h[c4];
                    }
                  }
                }
              }
              __builtin_amdgcn_sched_barrier(0);  // keep the batches apart
            }
          });
        }
        // per-neighbourhood outputs that need a and u only
        if (!skip && i < KFIX) {
          if (a.bwd_gnz) static_cast<T*>(a.bwd_gnz)[nb0 * KFIX + i] = gvv * xa * xa - gmv * xa * xu - gyv * xu * xu;
          if (a.bwd_gtg) {
            if (a.R > 1) {  // (uniform) y-bar_j,r = g_mean,r a_j
              const T* gmp = static_cast<const T*>(a.bwd_gmean);
              for (int r = 0; gmp && r < a.R; ++r)
                __hip_atomic_fetch_add(static_cast<T*>(a.bwd_gtg) + myidx * a.R + r, gmp[nb0 * a.R + r] * xa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
              __hip_atomic_fetch_add(static_cast<T*>(a.bwd_gtg) + myidx, gmv * xa + T(2) * gyv * xu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
        // ---- phase 7B: length-scale partials ------------------------------------------------------------------
        if (a.bwd_gls && !(MGP_BWD_EXP & 1)) {  // (uniform)
          T* gls = static_cast<T*>(a.bwd_gls);
          if (!aniso) {
            // Isotropy: dL/dl = -(1 | 2) / l sum gK k'(x) x, and k'(x) x = dk/dacc acc (2 | 1) for (l2 | F2)
            T sum = liso;
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
            if (!skip && lane == 0) gls[nb0] = T(-2) / ls[0] * sum;  // (-(1 | 2) / l times the (2 | 1) above)
          } else {
            // Anisotropy: dL/dl_f = -(2 / l_f) sum_pairs q_rc (z_rf - z_cf)^2 (the lane's share: s2h, phase 6B)
            V s2r[DGF];  // (the sums the reduction below takes: a local of this phase under SWEEP_LATE)
#pragma unroll
            for (int c4 = 0; c4 < DGF; ++c4) s2r[c4] = SWEEP_LATE ? V(0) : s2h[SWEEP_LATE ? 0 : c4];
            if constexpr (SWEEP_LATE) {
#pragma unroll
              for (int j = 0; j < BA; ++j) {
                const T* xa_ = Xh + wrap(i + own_offset(j)) * xs;
                V own[DGF];
#pragma unroll
                for (int c4 = 0; c4 < DGF; ++c4) own[c4] = *reinterpret_cast<const V*>(xa_ + c4 * E);
#pragma unroll
                for (int p = 0; p < BP; ++p) {
                  const T* xb_ = Xh + wrap(i + p + 1) * xs;
                  const V qv = V(acc[j * BP + p]);
#pragma unroll
                  for (int c4 = 0; c4 < DGF; ++c4) {
                    const V dz = own[c4] - *reinterpret_cast<const V*>(xb_ + c4 * E);
                    s2r[c4] = (dz * qv) * dz + s2r[c4];
                  }
                  if ((p & 1) == 1 || p == BP - 1) __builtin_amdgcn_sched_barrier(0);
                }
              }
            }
            // sum over the lanes through the tile (every lane is done reading it): lane i writes its DSTFIX sums over its
            // own row, lane (g, f) adds rows 64 / G g .. of feature f, a butterfly over g finishes.  (Rows >= M hold no
            // sums -- idle lanes, zero -- and the last slot's row holds the inverse length scales: read first.)
            constexpr int FP = DSTFIX <= 2 ? 2 : DSTFIX <= 4 ? 4 : DSTFIX <= 8 ? 8 : DSTFIX <= 16 ? 16 : DSTFIX <= 32 ? 32 : 64;
            constexpr int G = 64 / FP, RPG = 64 / G;
            constexpr int RROWS = M < NPL ? M : NPL;  // rows that carry sums (all of them exist in the tile)
            const T il = ilbuf[lane < d ? lane : 0];
            __syncthreads();
            // (through the dead dealt image where it is large enough: the built-in shape -- through the tile the same
            // kernel spills 476 registers instead of 8)
            constexpr bool RED_IMG = 64 * DSTFIX <= KMAT;
            T* redb = RED_IMG ? Kh : Xh;
            const int rstride = RED_IMG ? DSTFIX : xs;
            if (RED_IMG || i < RROWS) {
#pragma unroll
              for (int c4 = 0; c4 < DGF; ++c4) *reinterpret_cast<V*>(redb + i * rstride + c4 * E) = s2r[c4];
            }
            __syncthreads();
            const int f = lane & (FP - 1), gq = lane / FP;
            T tot = T(0);
            if (f < DSTFIX) {
#pragma unroll
              for (int r = 0; r < RPG; ++r)
                if (RED_IMG || gq * RPG + r < RROWS) tot += redb[(gq * RPG + r) * rstride + f];
            }
            for (int off = FP; off < 64; off <<= 1) tot += __shfl_xor(tot, off, 64);
            if (!skip && lane < d) gls[nb0 * (int64_t)d + lane] = T(-2) * il * tot;
          }
        }
        if (bad && live && lane == 0 && a.info) atomicAdd(a.info, 1);
        // the next task's rows: only now is the tile free
        __syncthreads();
        if (PIPE && t1 >= 0) {
          pipe_issue(t1, fix_index(next_idx, t1, h, i), lane);
          if (t2 >= 0) next_idx = load_index(t2, h, i);
        }
        MGP_WAVE_T(5)
        continue;
      }
      {
        // the Schur block sits in compile-time lanes: (q, q) = variance, (q + 1 + r, q) = -mean_r, (q + 1 + r, q + 1 + r) = -y_r^T K^-1 y_r
        constexpr int QF = KFIX;
        const int eq = cs2(QF);
        T* mean = static_cast<T*>(a.mean);
        T* var = static_cast<T*>(a.var);
        T* yk = static_cast<T*>(a.ykinvy);
        const int64_t nb = nb0;
        if (live) {
          if (lane == (eq & 63)) {
            *(var + nb) = bad ? num<T>::nan() : Dp[eq >> 6][QF & 1];
            if (bad && a.info) atomicAdd(a.info, 1);
          }
#pragma unroll
          for (int r = 0; r < RFIX; ++r) {
            const int YF = QF + 1 + r;
            const int em = cs2(QF) + (YF >> 1) - (QF >> 1), ey = cs2(YF);
            if (lane == (em & 63)) *(mean + (nb * RFIX + r)) = bad ? num<T>::nan() : -Dp[em >> 6][YF & 1];
            if (yk && lane == (ey & 63)) *(yk + (nb * RFIX + r)) = bad ? num<T>::nan() : -Dp[ey >> 6][YF & 1];
          }
        }
      }
      MGP_WAVE_T(5)
      continue;
    }
    // ---- phase 4: Cholesky, row per lane, column broadcast through LDS ----------------
    // Whole 16-byte groups are updated from the pivot's group on: entries of columns <= j
    // inside that group are dead by then (right-looking: column j is never read again).
    bool bad = false;
#if MGP_CHOL_PRIO
    // the elimination is a chain of short dependent steps: let its instructions go first, the other
    // waves' distance phases (long independent streams) fill the gaps
    __builtin_amdgcn_s_setprio(MGP_CHOL_PRIO);
#endif
    // fp32: LOOK-AHEAD.  Step j first finishes the register group that holds column j + 1, posts that
    // column and requests the 16 bytes with its pivot; the rest of the row is updated while that LDS round
    // trip is under way, so the reciprocal of the next step does not wait for it.
    constexpr bool LOOK = sizeof(T) == 4 && KFIX > 0 && MGP_LOOKAHEAD;  // (the run-time shapes gain nothing measurable and one of them spills)
    V piv = V(0);
    if constexpr (LOOK) {
      if (MGP_PHASE(g, 8)) {
        colh[i] = A[0][0];
        piv = *reinterpret_cast<const V*>(colh);
      }
    }
    // ONE test of the phase bit around the whole elimination (not one per step): with a test per step every
    // step is its own basic block -- two taken branches per step, and the wait-count pass, which starts
    // each block pessimistic, puts `s_waitcnt lgkmcnt(0)` in front of the trailing update, i.e. waits for
    // the look-ahead pivot it has just requested.  In one block the waits are counted.
    // (static shapes only: with a run-time k every step stays a block of its own anyway, and the hoisted test costs
    // the 64-slot fp64 run-time kernel 120 more spilled registers)
    constexpr bool ONEB = MGP_CHOL_ONE_BLOCK && STAT;
    if (!ONEB || MGP_PHASE(g, 8))
#pragma unroll
    for (int j = 0; j < (STAT ? KFIX : NP - 2); ++j) {
      if (j < k && (ONEB || MGP_PHASE(g, 8))) {
        const T ajj = A[j / E][j % E];
        if constexpr (!LOOK) colh[i] = ajj;
        if constexpr (sizeof(T) == 8) {
          // fp64: the pivot first (one group), then the trailing groups streamed: load, FMA,
          // next -- no full copy of the column is held, which keeps the 128-register rows of
          // the big shapes out of the spill zone (2 waves/SIMD instead of 1)
          const V cp = *reinterpret_cast<const V*>(colh + (j / E) * E);
          const T p = cp[j % E];
          bad = bad || !(p > T(0));
          const V nt = V(-ajj * pivot_rcp(p));
          if constexpr (COEFF || BWD_ROW) Kh[i * KS + j] = -nt[0];
          A[j / E] = cp * nt + A[j / E];
          // trailing groups GC at a time: the GC loads are in flight together, then the 2 GC FMAs (a
          // single wave needs ~8 outstanding 16-byte reads to cover the LDS latency with FMAs)
          constexpr int GC = MGP_F64_GC;
#pragma unroll
          for (int c0 = j / E + 1; c0 < NG; c0 += GC) {
            V cv[GC];
#pragma unroll
            for (int u = 0; u < GC; ++u)
              if (c0 + u < NG) cv[u] = *reinterpret_cast<const V*>(colh + (c0 + u) * E);
#pragma unroll
            for (int u = 0; u < GC; ++u)
              if (c0 + u < NG) A[c0 + u] = cv[u] * nt + A[c0 + u];
          }
        } else if constexpr (LOOK) {
          V col[NG];
          col[j / E] = piv;
#pragma unroll
          for (int c4 = j / E + 1; c4 < NG; ++c4) col[c4] = *reinterpret_cast<const V*>(colh + c4 * E);
          const T p = piv[j % E];
          bad = bad || !(p > T(0));
          const V nt = V(-ajj * pivot_rcp(p));
          if constexpr (COEFF || BWD_ROW) Kh[i * KS + j] = -nt[0];  // multiplier l_ij, kept for the back-substitution
          constexpr int JL = (STAT ? KFIX : NP - 2) - 1;    // last step of the loop
          const int g1 = (j < JL ? j + 1 : j) / E;         // compile-time after unrolling
          A[g1] = col[g1] * nt + A[g1];
          if (j < JL && j + 1 < k) {
            colh[i] = A[g1][(j + 1) % E];
            piv = *reinterpret_cast<const V*>(colh + g1 * E);
          }
#pragma unroll
          for (int c4 = j / E; c4 < NG; ++c4)
            if (c4 != g1) A[c4] = col[c4] * nt + A[c4];
        } else {
          V col[NG];
#pragma unroll
          for (int c4 = j / E; c4 < NG; ++c4) col[c4] = *reinterpret_cast<const V*>(colh + c4 * E);
          const T p = col[j / E][j % E];
          bad = bad || !(p > T(0));
          const V nt = V(-ajj * pivot_rcp(p));
          if constexpr (COEFF) Kh[i * KS + j] = -nt[0];  // multiplier l_ij, kept for the back-substitution
#pragma unroll
          for (int c4 = j / E; c4 < NG; ++c4) A[c4] = col[c4] * nt + A[c4];
        }
      }
    }

#if MGP_CHOL_PRIO && !MGP_PRIO_LATE_DROP
    __builtin_amdgcn_s_setprio(0);
#endif
    if constexpr (BWD_ROW) {
      // ---- phases 5B-7B, row-per-lane form (32-slot static shapes, NH neighbourhoods per wave): the multipliers
      // l_m,j sit in the exchange image (Kh[m * KS + j], written step by step above); see the dealt-triangle form of
      // these phases for the algebra.  Reference: torch autograd over torch/muygps_layer.py:129-164. ----------------
      if constexpr (TF) {  // the query row held the pivot buffer: zeros again, as the Gram form left it
        if (i < DSTFIX / E) *reinterpret_cast<V*>(Xh + KFIX * xs + i * E) = V(0);
      }
      __syncthreads();
      const int hoff = NH == 1 ? 0 : h * NP;
      T xa = i < KFIX ? Kh[KFIX * KS + i] : T(0);        // l_q,i = (D^-1 L^-1 c)_i
      T xu = i < KFIX ? Kh[(KFIX + 1) * KS + i] : T(0);  // l_y,i
      {
        constexpr int BB = 8;
#pragma unroll
        for (int mb = (KFIX - 1) / BB * BB; mb >= 0; mb -= BB) {
          T lm[BB];
#pragma unroll
          for (int e = 0; e < BB; ++e)
            if (mb + e >= 1 && mb + e < KFIX) lm[e] = Kh[(mb + e) * KS + i];  // (junk for m <= i: masked below)
#pragma unroll
          for (int e = BB - 1; e >= 0; --e) {
            const int m = mb + e;
            if (m >= 1 && m < KFIX) {
              T am = lane_value(xa, m), um = lane_value(xu, m);
              if constexpr (NH == 2) {
                const T am1 = lane_value(xa, m + NP), um1 = lane_value(xu, m + NP);
                am = h == 0 ? am : am1;
                um = h == 0 ? um : um1;
              }
              const T t = i < m ? lm[e] : T(0);
              xa = fma_t(-t, am, xa);
              xu = fma_t(-t, um, xu);
            }
          }
        }
      }
      // the two solved vectors by slot: 64 + 64 entries in the norm array's space (dead since the distance phase), or
      // (TF) in the padding columns NP, NP + 1 of the slot's image row
      T* avec = TF ? tile + kmat_base + NP : colbuf;
      T* uvec = TF ? tile + kmat_base + NP + 1 : colbuf + 64;
      auto vslot = [&](int s64) { return TF ? (s64 / NP) * KMAT + (s64 % NP) * KS : s64; };  // (entry of slot s64 = h NP + r)
      // feature cotangents asked for (uniform): the pair cotangents q_rc are also laid out as a symmetric NP x NP image
      // per neighbourhood where the multipliers were (dead from here on) -- phase 8B below
      // With them the per-feature length-scale sums come out of the same sweep: sum over pairs of q_rc (z_rf - z_cf)^2
      //   = sum_i z_if g_if  with g_i = sum_j q_ij (z_i - z_j), the unscaled feature cotangent (the g_i add up to zero, so
      // any common offset of the rows drops out) -- one product per row and feature instead of 40 subtractions and FMAs
      // per PAIR inside the pair loop: every cotangent with 40 length scales 9.4 -> 8.0 ms per 1 M.  Without feature
      // cotangents the pair loop keeps its own sweep (the image would cost as much at d = 40 and more at d = 8).
      const bool feat = a.bwd_gnn != nullptr || a.bwd_gq != nullptr;
      const bool lsweep = a.bwd_gls != nullptr && aniso;  // (uniform)
      const bool img = feat;
      const bool sweep = lsweep && !feat;
      __syncthreads();
      if (img) {  // (before the vectors: TF keeps them in the image's padding)
        T* Q0 = tile + kmat_base;  // (both neighbourhoods' images are contiguous)
#pragma unroll
        for (int e = 0; e < NH * KMAT; e += 64 * E)
          if (e + lane * E < NH * KMAT) *reinterpret_cast<V*>(Q0 + e + lane * E) = V(0);
      }
      avec[vslot(lane)] = i < KFIX ? xa : (i == KFIX ? T(-1) : T(0));
      uvec[vslot(lane)] = i < KFIX ? xu : T(0);
      __syncthreads();
      const int64_t nbw = nb0 + h;
      const T gmv = (live && a.bwd_gmean) ? (a.R > 1 ? T(1) : static_cast<const T*>(a.bwd_gmean)[nbw]) : T(0);  // (R > 1: folded into the column)
      const T gvv = (live && a.bwd_gvar) ? static_cast<const T*>(a.bwd_gvar)[nbw] : T(0);
      const T gyv = (live && a.bwd_gyk) ? static_cast<const T*>(a.bwd_gyk)[nbw] : T(0);
      const bool skip = bad || !live;
      T liso = T(0);
      constexpr int DGF = DSTFIX / E;
      V s2p[DGF];  // (pair-loop sweep only)
#pragma unroll
      for (int c4 = 0; c4 < DGF; ++c4) s2p[c4] = V(0);
      {
        T ar[BA], ur[BA], Pc[BP], Qc[BP];
#pragma unroll
        for (int j = 0; j < BA; ++j) {
          const int r = wrap(i + own_offset(j));
          ar[j] = avec[vslot(hoff + r)];
          ur[j] = uvec[vslot(hoff + r)];
        }
#pragma unroll
        for (int p = 0; p < BP; ++p) {
          const int c = wrap(i + p + 1);
          const T ac = avec[vslot(hoff + c)], uc = uvec[vslot(hoff + c)];
          Pc[p] = T(2) * gvv * ac - gmv * uc;
          Qc[p] = gmv * ac + T(2) * gyv * uc;
        }
        const bool lane_on = i < M;
        constexpr unsigned long long FIRSTM = wave_pair_first_mask(NS, BP, M), HALFM = wave_pair_half_mask(NS, BP, M);
        kernel_dispatch(a.kernel_id, a.metric_id, [&](auto kid, auto mid) {
          constexpr int KID = decltype(kid)::value, MID = decltype(mid)::value;
          auto pairs = [&](auto featc) {
          constexpr bool FEAT = decltype(featc)::value != 0;
#pragma unroll
          for (int sidx = 0; sidx < NS; ++sidx) {
            const int jj = sidx / BP, pp = sidx % BP;
            T sq;
            if constexpr (GRAM) sq = gram_sq(acc[sidx]);
            else sq = acc_total(acc[sidx]);
            const T dk = dcov_dacc<T, KID, MID>(sq, post_scale);
            bool on = lane_on && ((xkeep >> sidx) & 1u) != 0 && ((FIRSTM >> sidx) & 1ull) != 0;
            if ((HALFM >> sidx) & 1ull) on = on && wrap(i + own_offset(jj)) < wrap(i + pp + 1);
            const T gK = ar[jj] * Pc[pp] - ur[jj] * Qc[pp];
            const T qv = on ? gK * dk : T(0);
            // (a pair with a slot that has no features carries an infinite / NaN Gram-form distance: 0 x inf)
            liso = on ? fma_t(qv, sq, liso) : liso;
            if constexpr (FEAT && !(MGP_FEAT_EXP & 2)) {
              // both mirror entries of the image; a pair that is not counted here (its twin is, or it has no
              // features) goes to a padding element -- branch-free, as in the dealt form's write-back
              const int r_ = wrap(i + own_offset(jj)), c_ = wrap(i + pp + 1);
              constexpr int QDUMP = (NP - 1) * KS + NP;
              Kh[on ? r_ * KS + c_ : QDUMP] = qv;
              Kh[on ? c_ * KS + r_ : QDUMP] = qv;
            }
            if constexpr (!FEAT) {
              if (sweep) {
                const T* xa_ = Xh + wrap(i + own_offset(jj)) * xs;
                const T* xb_ = Xh + wrap(i + pp + 1) * xs;
                const V qq = V(qv);
#pragma unroll
                for (int c4 = 0; c4 < DGF; ++c4) {
                  const V dz = vsub(*reinterpret_cast<const V*>(xa_ + c4 * E), *reinterpret_cast<const V*>(xb_ + c4 * E));
                  s2p[c4] = (dz * qq) * dz + s2p[c4];
                }
              }
            }
          }
          };
          if (img) pairs(ic<1>{});
          else pairs(ic<0>{});
        });
      }
      if (!skip && i < KFIX) {
        if (a.bwd_gnz) static_cast<T*>(a.bwd_gnz)[nbw * KFIX + i] = gvv * xa * xa - gmv * xa * xu - gyv * xu * xu;
        if (a.bwd_gtg && a.R == 1)
          __hip_atomic_fetch_add(static_cast<T*>(a.bwd_gtg) + myidx, gmv * xa + T(2) * gyv * xu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (a.bwd_gtg && a.R > 1 && a.bwd_gmean) {  // (uniform)
        // several responses: y-bar_j,r = g_mean,r a_j, the lanes along (row, response) -- R consecutive elements per row
        // (a lane per row and a loop over r: 5 of 10.9 ms per 500 k neighbourhoods at k = 30, R = 10)
        const T* gmp = static_cast<const T*>(a.bwd_gmean) + nbw * a.R;
        const int RR = a.R, nel = KFIX * RR;
        for (int t0 = 0; t0 < nel; t0 += NP) {
          const int t = t0 + i;
          const int tt = t < nel ? t : 0;
          const int j = tt / RR, r = tt - j * RR;
          const int64_t rowj = __shfl(myidx, h * NP + j, 64);  // (slot j's table row)
          if (t < nel && !skip)
            __hip_atomic_fetch_add(static_cast<T*>(a.bwd_gtg) + rowj * RR + r, gmp[r] * avec[vslot(hoff + j)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      // per-feature length-scale sums -> gradient partials of the neighbourhood
      auto ls_reduce = [&](const V (&s2h)[DGF]) {
          T* gls = static_cast<T*>(a.bwd_gls);
          // per-feature sums over the neighbourhood's lanes, through its (dead) exchange image, FPP features per pass:
          // lane i writes its sums of the pass, lane f adds the NP rows of feature f
          constexpr int FPP = (KMAT / NP) / E * E;  // features of a pass (whole 16-byte groups)
          static_assert(FPP >= E, "exchange image too small for the length-scale reduction");
#pragma unroll
          for (int f0 = 0; f0 < DSTFIX; f0 += FPP) {
            const int nf = DSTFIX - f0 < FPP ? DSTFIX - f0 : FPP;  // (compile-time after unrolling)
            __syncthreads();
#pragma unroll
            for (int c4 = 0; c4 < FPP / E; ++c4)
              if (f0 / E + c4 < DGF) *reinterpret_cast<V*>(Kh + i * FPP + c4 * E) = s2h[f0 / E + c4];
            __syncthreads();
#pragma unroll
            for (int ff = 0; ff < (FPP + NP - 1) / NP; ++ff) {
              const int f = ff * NP + i;  // feature of the pass this lane sums
              if (ff * NP < nf) {
                T t0 = T(0), t1 = T(0), t2 = T(0), t3 = T(0);
                if (f < nf) {
#pragma unroll
                  for (int r = 0; r < NP; r += 4) {
                    t0 += Kh[(r + 0) * FPP + f];
                    t1 += Kh[(r + 1) * FPP + f];
                    t2 += Kh[(r + 2) * FPP + f];
                    t3 += Kh[(r + 3) * FPP + f];
                  }
                  const int fg = f0 + f;  // the feature
                  if (!skip && fg < d) gls[nbw * (int64_t)d + fg] = T(-2) * ilbuf[fg] * ((t0 + t1) + (t2 + t3));
                }
              }
            }
          }
      };
      if (sweep) ls_reduce(s2p);  // (before the registers of the feature sweep are needed)
      V gxv[DGF];
      if (img) {
        // ---- phase 8B: feature cotangents  x-bar_i,f = 2 / l_f  sum_j q_ij (z_i,f - z_j,f)  on the (scaled, GRAM: centred)
        // rows the tile still holds.  Lane i takes row i of the image into registers and walks the rows j of the tile
        // (every lane of a neighbourhood reads the same 16 bytes: broadcast reads); the sums then go back into row i of
        // the tile -- dead by then -- from where they leave as atomic adds with the lanes along the FEATURES of a row
        // (one instruction covers 32 / 64 consecutive elements of the gradient table instead of as many rows).
        // Reference: autograd through torch/muygps_layer.py:129-164 (crosswise / pairwise differences of the embedded
        // features); the same sums as mgp_backward.hip's last stage.
        __syncthreads();
        constexpr int QG = (KFIX + 1 + E - 1) / E;  // 16-byte groups of the image row that hold q_i,0 .. q_i,KFIX
        V qrow[QG];
#pragma unroll
        for (int c4 = 0; c4 < QG; ++c4) qrow[c4] = *reinterpret_cast<const V*>(Kh + i * KS + c4 * E);
        // GRAM: the rows are centred on the query (|z| is of the size of the differences), so the sum is taken as
        //   z_i sum_j q_ij - sum_j q_ij z_j  -- one packed FMA per two features and row instead of a subtraction and an FMA;
        // difference form otherwise (raw rows: the products would cancel)
        const T* xi_ = Xh + i * xs;
#pragma unroll
        for (int c4 = 0; c4 < DGF; ++c4) gxv[c4] = V(0);
        if constexpr (GRAM) {
          T qs = T(0);
#pragma unroll
          for (int j = 0; j <= ((MGP_FEAT_EXP & 1) ? -1 : KFIX); ++j) {
            const T qj = qrow[j / E][j % E];
            qs += qj;
            const V qq = V(-qj);
            const T* xj_ = Xh + j * xs;
#pragma unroll
            for (int c4 = 0; c4 < DGF; ++c4) gxv[c4] = *reinterpret_cast<const V*>(xj_ + c4 * E) * qq + gxv[c4];
            if (j % 2 == 1) __builtin_amdgcn_sched_barrier(0);  // (two rows in flight; all of them hoisted is what spills)
          }
          const V qsv = V(qs);
#pragma unroll
          for (int c4 = 0; c4 < DGF; ++c4) gxv[c4] = *reinterpret_cast<const V*>(xi_ + c4 * E) * qsv + gxv[c4];
        } else {
          constexpr int FCH = DGF <= 10 ? DGF : 10;  // groups of the own row in registers at a time
#pragma unroll
          for (int c0 = 0; c0 < DGF; c0 += FCH) {
            V xi[FCH];
#pragma unroll
            for (int u = 0; u < FCH; ++u)
              if (c0 + u < DGF) xi[u] = *reinterpret_cast<const V*>(xi_ + (c0 + u) * E);
#pragma unroll
            for (int j = 0; j <= ((MGP_FEAT_EXP & 1) ? -1 : KFIX); ++j) {
              const V qq = V(qrow[j / E][j % E]);
              const T* xj_ = Xh + j * xs;
#pragma unroll
              for (int u = 0; u < FCH; ++u)
                if (c0 + u < DGF) gxv[c0 + u] = (xi[u] - *reinterpret_cast<const V*>(xj_ + (c0 + u) * E)) * qq + gxv[c0 + u];
              if (j % 2 == 1) __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        // (the sums are pinned HERE: left to itself the compiler sinks each group's chain of FMAs to its use -- into
        // the branches behind the next barrier, which the loads cannot cross -- and keeps all 310 loaded groups alive in
        // between: 1 150 spilled registers)
#pragma unroll
        for (int c4 = 0; c4 < DGF; ++c4) asm volatile("" : "+v"(gxv[c4]));
      }
      if (a.bwd_gls) {  // (uniform)
        T* gls = static_cast<T*>(a.bwd_gls);
        if (!aniso) {
          T sum = liso;
          for (int off = NP / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
          if (!skip && i == 0) gls[nbw] = T(-2) / ls[0] * sum;
        } else if (feat) {
          V s2h[DGF];
#pragma unroll
          for (int c4 = 0; c4 < DGF; ++c4) {
            const V zi = *reinterpret_cast<const V*>(Xh + i * xs + c4 * E);
            s2h[c4] = i <= KFIX ? zi * gxv[c4] : V(0);
          }
          ls_reduce(s2h);
        }
      }
      if (feat) {
#pragma unroll
        for (int c4 = 0; c4 < DGF; ++c4) {
          const V ilv = *reinterpret_cast<const V*>(ilbuf + c4 * E);  // (Isotropy: whatever the row holds; not used)
          gxv[c4] = gxv[c4] * (aniso ? ilv * V(T(2)) : V(T(2)));
        }
        __syncthreads();  // every lane is done with the rows
        T* gnn_ = static_cast<T*>(a.bwd_gnn);
        T* gq_ = static_cast<T*>(a.bwd_gq);
        // element offset of every slot's row: where the two vectors were, or (TF) in the last 512 bytes of the images
        int64_t* roff = reinterpret_cast<int64_t*>(TF ? tile + kmat_base + NH * KMAT - 128 : colbuf);
        // (Measured out: requesting the NEXT task's rows here, ahead of the scatter -- the sums then staged through the
        // dead image in two passes instead of the tile: 9.2 against 7.3 ms per 1 M at the headline shape.)
        {
          T* gdst = i <= KFIX ? Xh + i * xs : Kh;  // (lanes without a row: into the dead image -- no branch)
#pragma unroll
          for (int c4 = 0; c4 < DGF; ++c4) *reinterpret_cast<V*>(gdst + c4 * E) = gxv[c4];
        }
        roff[lane] = myidx * (int64_t)d;
        __syncthreads();
        constexpr int NOUT = (KFIX + 1) * DFIX;  // elements of a neighbourhood's cotangent block, row-major
#pragma unroll 4
        for (int t0 = 0; t0 < NOUT; t0 += NP) {
          const int t = t0 + i;
          const int r = t / DFIX, f = t - r * DFIX;
          if (t < NOUT && !skip && !(MGP_FEAT_EXP & 4)) {
            T* base = r < KFIX ? gnn_ : gq_;
            if (base) {
              if constexpr ((MGP_FEAT_EXP & 8) != 0) base[roff[h * NP + r] + f] = Xh[r * xs + f];  // (timing: plain stores)
              else unsafeAtomicAdd(base + roff[h * NP + r] + f, Xh[r * xs + f]);
            }
          }
        }
      }
      if (bad && live && i == 0 && a.info) atomicAdd(a.info, 1);
      // the next task's rows: only now is the tile free
      __syncthreads();
      if (PIPE && t1 >= 0) {
        pipe_issue(t1, fix_index(next_idx, t1, h, i), lane);
        if (t2 >= 0) next_idx = load_index(t2, h, i);
      }
      continue;
    }
    // ---- phase 4b (COEFF): x = K^-1 y by back-substitution -------------------------------
    // K = L D L^T with unit lower L; the forward sweep formed l_ij (j < i) in lane i, and the
    // response row's multipliers are w_j = (D^-1 L^-1 y)_j.  Solve L^T x = w: the multipliers are
    // written to the (by then free) exchange matrix as they are formed, so lane j can read column j
    // of L; then for m = k-1 .. 1 the finished x_m is read from lane m (v_readlane) and every lane
    // j < m takes l_mj x_m off.
    if constexpr (COEFF) {
      static_assert(!PIPED, "coefficient variant: register-staged gather (the tile must be free)");
      __syncthreads();
      T x = Kh[(q + 1) * KS + i];  // w_i, from the response row
#pragma unroll
      for (int m = NP - 3; m >= 1; --m) {
        if (m < k) {
          const T lmj = Kh[m * KS + i];  // l_mi
          T xm = lane_value(x, m);
          if constexpr (NH == 2) xm = h == 0 ? xm : lane_value(x, m + NP);
          if (i < m) x = fma_t(-lmj, xm, x);
        }
      }
      if (live && i < k) static_cast<T*>(a.coeffs)[(nb0 + h) * (int64_t)k + i] = bad ? num<T>::nan() : x;
    }

    // ---- phase 5: Schur block -> outputs ----------------------------------------------
    T* mean = static_cast<T*>(a.mean);
    T* var = static_cast<T*>(a.var);
    T* yk = static_cast<T*>(a.ykinvy);
    const int64_t nb = nb0 + h;
    if constexpr (COEFF) {
      if (live && bad && i == q && a.info) atomicAdd(a.info, 1);  // mean / variance are not emitted
    } else if (RFIX == 1) {
      // q = NP-2 and the response row NP-1 are compile-time: the Schur block sits in fixed registers
      constexpr int QF = STAT ? KFIX : NP - 2, YF = QF + 1;
      const T sq = A[QF / E][QF % E], sy = A[YF / E][YF % E];
      if (live) {
        if (i == QF) {
          *(var + nb) = bad ? num<T>::nan() : sq;
          if (bad && a.info) atomicAdd(a.info, 1);
        } else if (i == YF) {
          *(mean + nb) = bad ? num<T>::nan() : -sq;
          if (yk) *(yk + nb) = bad ? num<T>::nan() : -sy;
        }
      }
    } else if constexpr (PIPED || TRI) {
      // The tile (which the exchange matrix aliases) already receives the next task's rows (and a
      // packed exchange matrix has no room for whole rows), so
      // the two entries a lane emits -- column q and the diagonal of its own row -- are picked out
      // of the registers by a compare-select sweep instead of a round trip through LDS.
      T aq = T(0), aii = T(0);
#pragma unroll
      for (int c = 0; c < NG * E; ++c) {
        const T v = A[c / E][c % E];
        aq = c == q ? v : aq;
        aii = c == i ? v : aii;
      }
      if (live) {
        if (i == q) {
          *(var + nb) = bad ? num<T>::nan() : aq;
          if (bad && a.info) atomicAdd(a.info, 1);
        } else if (i > q && i <= q + R) {  // (static shapes: lanes behind the last response row are idle)
          const int r = i - q - 1;
          *(mean + (nb * R + r)) = bad ? num<T>::nan() : -aq;
          if (yk) *(yk + (nb * R + r)) = bad ? num<T>::nan() : -aii;
        }
      }
    } else {
      __syncthreads();
#pragma unroll
      for (int c4 = 0; c4 < NG; ++c4) *reinterpret_cast<V*>(Kh + i * KS + c4 * E) = A[c4];
      __syncthreads();
      if (live) {
        if (i == q) {
          *(var + nb) = bad ? num<T>::nan() : Kh[q * KS + q];
          if (bad && a.info) atomicAdd(a.info, 1);
        } else if (i > q && i <= q + R) {  // (static shapes: lanes behind the last response row are idle)
          const int r = i - q - 1;
          *(mean + (nb * R + r)) = bad ? num<T>::nan() : -Kh[i * KS + q];
          if (yk) *(yk + (nb * R + r)) = bad ? num<T>::nan() : -Kh[i * KS + i];
        }
      }
    }
  }
  // ---- one-launch LOOCV evaluation (mgp_loocv_*): the workgroup is out of tasks -- its leaf of the reduction tree (its
  // own outputs, read back) and up the tree as far as its tickets are the last ones (mgp_loocv_tree.h) ---------------
#ifndef MGP_EXP_NO_TREE  // (-DMGP_EXP_NO_TREE: A/B control, tools/loocv_ab.py -- the kernel without the walk)
  if constexpr (!COEFF) {
    if (a.tree.out != nullptr) {
      drain_stores();
      tree_leaf_done<T>(a.tree, static_cast<const T*>(a.mean), static_cast<const T*>(a.var), static_cast<const T*>(a.ykinvy),
                        a.batch_idx, a.b, (int)blockIdx.x, (int)threadIdx.x);
    }
  }
#endif
#if MGP_WAVE_TIMING
  if (threadIdx.x == 0)
    for (int t = 0; t < 8; ++t) atomicAdd(&g_wave_timing[t], tacc_[t]);
#endif
}

}  // namespace mgp

// Device helpers shared by the register-resident wave kernels (mgp_fused_wave.hip,
// mgp_fused_rhs.hip): 16-byte vector types, packed-f32 difference/accumulate, fast exp / sqrt /
// reciprocal, squared distance -> covariance, direct-to-LDS load, compile-time kernel dispatch.
#pragma once

#ifndef __HIPCC_RTC__
#include <mutex>
#include <utility>
#endif

#include "mgp_args.h"

namespace mgp {

#ifndef __HIPCC_RTC__  // host side (a run-time compile of the kernels sees the device helpers only)

// Resident workgroups per CU of one kernel instantiation, per device and per (LDS size, thread count):
// the run-time-shape instantiations change their LDS size with d, so a few recent geometries are kept
// (alternating shapes do not re-run the occupancy query); the CU count is read once per device.
// Queried under a mutex: the entry points stay re-entrant and a second device gets its own numbers.
struct Residency {
  static constexpr int WAYS = 16;
  std::mutex mu;
  struct Entry { int lds = -1, threads = 0, per_cu = 0; uintptr_t who = 0; };
  struct Dev { int cus = 0, next = 0; Entry e[WAYS]; } dev[MGP_MAX_DEVICES];
  // -> MGP_OK and (per_cu, cus), or an error status
  int lookup(const void* kernel, int threads, size_t lds, int* per_cu, int* cus) {
    return lookup_by([&](int* n) { return hipOccupancyMaxActiveBlocksPerMultiprocessor(n, kernel, threads, lds); }, 0, threads,
                     lds, per_cu, cus);
  }
  // (a kernel of a run-time compiled module: one Residency serves them all, so the function is part of the key)
  int lookup(hipFunction_t fn, int threads, size_t lds, int* per_cu, int* cus) {
    return lookup_by([&](int* n) { return hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(n, fn, threads, lds); },
                     reinterpret_cast<uintptr_t>(fn), threads, lds, per_cu, cus);
  }
  template <typename Query>
  int lookup_by(Query&& query, uintptr_t who, int threads, size_t lds, int* per_cu, int* cus) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MGP_MAX_DEVICES) return MGP_EHIP;
    std::lock_guard<std::mutex> lock(mu);
    Dev& v = dev[d];
    if (v.cus == 0) {
      int n = 0;
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n < 1) return MGP_EHIP;
      v.cus = n;
    }
    for (int w = 0; w < WAYS; ++w)
      if (v.e[w].lds == (int)lds && v.e[w].threads == threads && v.e[w].who == who) {
        *per_cu = v.e[w].per_cu;
        *cus = v.cus;
        return MGP_OK;
      }
    int n = 0;
    hipError_t err = query(&n);
    if (err != hipSuccess) return -(1000 + (int)err);
    if (n < 1) return MGP_EUNSUPPORTED;
    // the occupancy query over-reports for LDS-bound shapes: measured on gfx950, LDS is handed out
    // in 1280-byte granules of the CU's 160 KiB (13 x 12192 B is refused, 12 x 12704 B fits)
    const int by_lds = lds == 0 ? n : (int)((160 * 1024) / (((lds + 1279) / 1280) * 1280));
    Entry& e = v.e[v.next];
    v.next = (v.next + 1) % WAYS;
    e.per_cu = n < by_lds ? n : by_lds;
    if (e.per_cu < 1) {
      e.lds = -1;
      return MGP_EUNSUPPORTED;
    }
    e.lds = (int)lds;
    e.threads = threads;
    e.who = who;
    *per_cu = e.per_cu;
    *cus = v.cus;
    return MGP_OK;
  }
};

#endif  // !__HIPCC_RTC__

template <typename T> struct v16;
template <> struct v16<float> {
  typedef float type __attribute__((ext_vector_type(4)));
  typedef float acc __attribute__((ext_vector_type(2)));
  static constexpr int N = 4;
};
template <> struct v16<double> {
  typedef double type __attribute__((ext_vector_type(2)));
  typedef double acc;
  static constexpr int N = 2;
};

typedef float f2 __attribute__((ext_vector_type(2)));
// packed subtract in ONE instruction (hipcc lowers a <2 x float> fsub to two v_sub_f32)
__device__ __forceinline__ f2 pk_sub(f2 x, f2 y) {
  f2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ v16<float>::type vsub(const v16<float>::type& x, const v16<float>::type& y) {
  v16<float>::type r;
  r.xy = pk_sub(x.xy, y.xy);
  r.zw = pk_sub(x.zw, y.zw);
  return r;
}
__device__ __forceinline__ v16<double>::type vsub(const v16<double>::type& x, const v16<double>::type& y) {
  return x - y;
}
__device__ __forceinline__ void accum(v16<float>::acc& a, const v16<float>::type& df) {
  a = df.xy * df.xy + a;  // v_pk_fma_f32
  a = df.zw * df.zw + a;
}
__device__ __forceinline__ void accum(double& a, const v16<double>::type& df) {
  a = __builtin_fma(df.x, df.x, a);
  a = __builtin_fma(df.y, df.y, a);
}
// One partner 16-byte group against FOUR own rows: acc_j += (w_j - o).xy^2 + (w_j - o).zw^2, as one
// hand-ordered block of 16 packed instructions.  gfx950 needs one wait state between a packed-f32
// instruction and an instruction that reads its result; left to the compiler's scheduler, which
// does not model that, every second instruction of this stream was followed by an `s_nop 0`
// (~300 per task).  Here no instruction reads the result of its predecessor: 4 differences, then
// squares-and-adds interleaved with the second half's differences.  Same operation order per
// accumulator as accum(acc, vsub(w, o)) -- results are bit-identical.
__device__ __forceinline__ void dist_block4(f2& a0, f2& a1, f2& a2, f2& a3, const v16<float>::type& w0,
                                            const v16<float>::type& w1, const v16<float>::type& w2,
                                            const v16<float>::type& w3, const v16<float>::type& o) {
  f2 t0, t1, t2, t3;
  asm volatile(
      "v_pk_add_f32 %4, %8, %16 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %5, %10, %16 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %6, %12, %16 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %7, %14, %16 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_fma_f32 %0, %4, %4, %0\n\t"
      "v_pk_add_f32 %4, %9, %17 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_fma_f32 %1, %5, %5, %1\n\t"
      "v_pk_add_f32 %5, %11, %17 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_fma_f32 %2, %6, %6, %2\n\t"
      "v_pk_add_f32 %6, %13, %17 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_fma_f32 %3, %7, %7, %3\n\t"
      "v_pk_add_f32 %7, %15, %17 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_fma_f32 %0, %4, %4, %0\n\t"
      "v_pk_fma_f32 %1, %5, %5, %1\n\t"
      "v_pk_fma_f32 %2, %6, %6, %2\n\t"
      "v_pk_fma_f32 %3, %7, %7, %3"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(w0.xy), "v"(w0.zw), "v"(w1.xy), "v"(w1.zw), "v"(w2.xy), "v"(w2.zw), "v"(w3.xy), "v"(w3.zw), "v"(o.xy),
        "v"(o.zw));
}
__device__ __forceinline__ void dist_block4(double&, double&, double&, double&, const v16<double>::type&,
                                            const v16<double>::type&, const v16<double>::type&,
                                            const v16<double>::type&, const v16<double>::type&) {}  // fp32 only
// Gram form: one partner 16-byte group against BA own rows, acc[j * BP] += w_j.xy * o.xy + w_j.zw * o.zw,
// ordered so that no packed instruction reads its predecessor's result (see dist_block4).
template <int BA, int BP>
__device__ __forceinline__ void gram_block(f2* acc, const v16<float>::type (&w)[BA], const v16<float>::type& o) {
  if constexpr (BA == 3) {
    asm volatile(
        "v_pk_fma_f32 %0, %3, %9, %0\n\t"
        "v_pk_fma_f32 %1, %5, %9, %1\n\t"
        "v_pk_fma_f32 %2, %7, %9, %2\n\t"
        "v_pk_fma_f32 %0, %4, %10, %0\n\t"
        "v_pk_fma_f32 %1, %6, %10, %1\n\t"
        "v_pk_fma_f32 %2, %8, %10, %2"
        : "+v"(acc[0]), "+v"(acc[BP]), "+v"(acc[2 * BP])
        : "v"(w[0].xy), "v"(w[0].zw), "v"(w[1].xy), "v"(w[1].zw), "v"(w[2].xy), "v"(w[2].zw), "v"(o.xy), "v"(o.zw));
  } else if constexpr (BA == 4) {
    asm volatile(
        "v_pk_fma_f32 %0, %4, %12, %0\n\t"
        "v_pk_fma_f32 %1, %6, %12, %1\n\t"
        "v_pk_fma_f32 %2, %8, %12, %2\n\t"
        "v_pk_fma_f32 %3, %10, %12, %3\n\t"
        "v_pk_fma_f32 %0, %5, %13, %0\n\t"
        "v_pk_fma_f32 %1, %7, %13, %1\n\t"
        "v_pk_fma_f32 %2, %9, %13, %2\n\t"
        "v_pk_fma_f32 %3, %11, %13, %3"
        : "+v"(acc[0]), "+v"(acc[BP]), "+v"(acc[2 * BP]), "+v"(acc[3 * BP])
        : "v"(w[0].xy), "v"(w[0].zw), "v"(w[1].xy), "v"(w[1].zw), "v"(w[2].xy), "v"(w[2].zw), "v"(w[3].xy), "v"(w[3].zw),
          "v"(o.xy), "v"(o.zw));
  } else if constexpr (BA == 5) {
    asm volatile(
        "v_pk_fma_f32 %0, %5, %15, %0\n\t"
        "v_pk_fma_f32 %1, %7, %15, %1\n\t"
        "v_pk_fma_f32 %2, %9, %15, %2\n\t"
        "v_pk_fma_f32 %3, %11, %15, %3\n\t"
        "v_pk_fma_f32 %4, %13, %15, %4\n\t"
        "v_pk_fma_f32 %0, %6, %16, %0\n\t"
        "v_pk_fma_f32 %1, %8, %16, %1\n\t"
        "v_pk_fma_f32 %2, %10, %16, %2\n\t"
        "v_pk_fma_f32 %3, %12, %16, %3\n\t"
        "v_pk_fma_f32 %4, %14, %16, %4"
        : "+v"(acc[0]), "+v"(acc[BP]), "+v"(acc[2 * BP]), "+v"(acc[3 * BP]), "+v"(acc[4 * BP])
        : "v"(w[0].xy), "v"(w[0].zw), "v"(w[1].xy), "v"(w[1].zw), "v"(w[2].xy), "v"(w[2].zw), "v"(w[3].xy), "v"(w[3].zw),
          "v"(w[4].xy), "v"(w[4].zw), "v"(o.xy), "v"(o.zw));
  } else {
#pragma unroll
    for (int j = 0; j < BA; ++j) acc[j * BP] = w[j].xy * o.xy + acc[j * BP];
#pragma unroll
    for (int j = 0; j < BA; ++j) acc[j * BP] = w[j].zw * o.zw + acc[j * BP];
  }
}
// fp64: the same block as plain FMAs (no packed fp64; two accumulators' worth of independent chains per row)
template <int BA, int BP>
__device__ __forceinline__ void gram_block(double* acc, const v16<double>::type (&w)[BA], const v16<double>::type& o) {
#pragma unroll
  for (int j = 0; j < BA; ++j) acc[j * BP] = __builtin_fma(w[j].x, o.x, acc[j * BP]);
#pragma unroll
  for (int j = 0; j < BA; ++j) acc[j * BP] = __builtin_fma(w[j].y, o.y, acc[j * BP]);
}
// |x|^2 of one 16-byte group into a partial sum; squared distance out of a Gram accumulator
__device__ __forceinline__ void norm_accum(f2& n, const v16<float>::type& x) {
  n = x.xy * x.xy + n;
  n = x.zw * x.zw + n;
}
__device__ __forceinline__ void norm_accum(double& n, const v16<double>::type& x) {
  n = __builtin_fma(x.x, x.x, n);
  n = __builtin_fma(x.y, x.y, n);
}
__device__ __forceinline__ void gram_finish(f2& a, float nsum) {  // |a'|^2 + |b'|^2 - 2 a'.b', clamped; left in a.x
  a.x = __builtin_fmaxf(__builtin_fmaf(-2.0f, a.x + a.y, nsum), 0.0f);
  a.y = 0.0f;
}
__device__ __forceinline__ void gram_finish(double& a, double nsum) { a = __builtin_fmax(__builtin_fma(-2.0, a, nsum), 0.0); }
// The same with the CANCELLATION GUARD (fp32).  |a'|^2 + |b'|^2 - 2 a'.b' loses log2((|a'|^2 + |b'|^2) / d^2) bits: harmless
// when the neighbours are about as far from each other as from the query (k-NN neighbourhoods, random rows), fatal when
// they form a tight cluster far from the query -- d^2 = 2e-4 l^2 under norms of 9 l^2 leaves the covariances with ~1e-5
// absolute error and the posterior mean of a nugget-1e-3 model with 6e-3 (tests/test_gpu_gram_stress.py; the difference
// form on the same data: 4e-5).  `guard` goes negative when a pair's squared distance comes out below 1 / MGP_GRAM_GUARD
// of the norm sum it was subtracted from (more than 5 bits cancelled: relative error of d^2 above ~1e-5); the wave then
// recomputes the task's distances in the difference form (wave kernels: phase 2G).  A pair with a slot that has no
// features carries an infinite norm (phase 1b writes it): its test value is NaN, which v_min_f32 ignores.
#ifndef MGP_GRAM_GUARD
#define MGP_GRAM_GUARD 32.0f
#endif
__device__ __forceinline__ void gram_finish2(f2& a, f2& b, float nsa, float nsb, float& guard) {
  const float da = __builtin_fmaf(-2.0f, a.x + a.y, nsa), db = __builtin_fmaf(-2.0f, b.x + b.y, nsb);
  const float ta = __builtin_fmaf(da, MGP_GRAM_GUARD, -nsa), tb = __builtin_fmaf(db, MGP_GRAM_GUARD, -nsb);
  guard = __builtin_fminf(__builtin_fminf(guard, ta), tb);  // v_min3_f32
  a.x = __builtin_fmaxf(da, 0.0f);
  a.y = 0.0f;
  b.x = __builtin_fmaxf(db, 0.0f);
  b.y = 0.0f;
}
__device__ __forceinline__ void gram_finish1(f2& a, float nsa, float& guard) {
  const float da = __builtin_fmaf(-2.0f, a.x + a.y, nsa);
  guard = __builtin_fminf(guard, __builtin_fmaf(da, MGP_GRAM_GUARD, -nsa));
  a.x = __builtin_fmaxf(da, 0.0f);
  a.y = 0.0f;
}
__device__ __forceinline__ void gram_finish2(double& a, double& b, double nsa, double nsb, double&) {  // fp64: no guard needed
  gram_finish(a, nsa);
  gram_finish(b, nsb);
}
__device__ __forceinline__ void gram_finish1(double& a, double nsa, double&) { gram_finish(a, nsa); }
__device__ __forceinline__ bool gram_guard_tripped(float guard) { return __builtin_amdgcn_ballot_w64(guard < 0.0f) != 0; }
__device__ __forceinline__ bool gram_guard_tripped(double) { return false; }
// the lanes whose guard tripped (a wave kernel with several neighbourhoods per wave redoes only the tripped ones: a
// neighbourhood's bits must not depend on which other neighbourhood shares its wave -- shards of a batch concatenate
// bit for bit, tests/test_gpu_properties.py)
__device__ __forceinline__ unsigned long long gram_guard_lanes(float guard) { return __builtin_amdgcn_ballot_w64(guard < 0.0f); }
__device__ __forceinline__ unsigned long long gram_guard_lanes(double) { return 0ull; }
// difference-form result (two partial sums) -> the layout the Gram path leaves behind (squared distance in .x)
__device__ __forceinline__ void gram_from_diff(f2& a) {
  a.x = a.x + a.y;
  a.y = 0.0f;
}
__device__ __forceinline__ void gram_from_diff(double&) {}
__device__ __forceinline__ float gram_sq(const f2& a) { return a.x; }
__device__ __forceinline__ double gram_sq(const double& a) { return a; }

__device__ __forceinline__ float acc_total(const v16<float>::acc& a) { return a.x + a.y; }
__device__ __forceinline__ double acc_total(const double& a) { return a; }

__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

// 1/p for the pivot: v_rcp_f32 is 1 ulp (as good as the FMAs it feeds); f64 needs refining
__device__ __forceinline__ float pivot_rcp(float p) { return __builtin_amdgcn_rcpf(p); }
__device__ __forceinline__ double pivot_rcp(double p) {
  double r = __builtin_amdgcn_rcp(p);
  double e = __builtin_fma(-p, r, 1.0);
  r = __builtin_fma(e, r, r);
  e = __builtin_fma(-p, r, 1.0);
  return __builtin_fma(e, r, r);
}

// e^{-t} (t >= 0) on v_exp_f32 with a two-term log2(e) so that the argument's rounding
// error does not scale with t: ~2 ulp, 6 instructions (libm expf is ~25).
__device__ __forceinline__ float exp_neg(float t) {
  const float hi = -t * 1.44269502162933349609375f;
  const float lo = __builtin_fmaf(-t, 1.44269502162933349609375f, -hi) - t * 1.925963033500011e-08f;
  const float e = __builtin_amdgcn_exp2f(hi);
  return __builtin_fmaf(e * lo, 0.693147180559945f, e);
}
// exp(-t) in fp64, t >= 0 (every caller passes a scaled distance).  The library exp is the same algorithm -- n =
// rint(-t log2 e), r = -t - n ln 2 in two pieces, a degree-11 polynomial, ldexp -- but the compiler emits its
// Horner steps as v_fmac_f64 into a register that has to be re-loaded with the coefficient first: 18 v_mov_b32
// per evaluation next to 14 FMAs.  Here the coefficients are SGPR operands of v_fma_f64 (inline assembly: the
// "s" constraint), written by the scalar unit.  Coefficients: the library's (degree 12, minimax on [-ln2/2, ln2/2]); the
// overflow side needs no test (the argument is <= 0), underflow is ldexp's.
#ifndef MGP_EXP64_LIB
#define MGP_EXP64_LIB 0
#endif
__device__ __forceinline__ double fma_sc(double a, double b, double c_uniform) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_uniform));
  return d;
}
__device__ __forceinline__ double exp_neg(double t) {
#if MGP_EXP64_LIB
  return ::exp(-t);
#else
  const double n = __builtin_rint(t * -1.4426950408889634074);
  double r = __builtin_fma(n, -6.93147180369123816490e-01, -t);
  r = __builtin_fma(n, -1.90821492927058770002e-10, r);
  constexpr auto C = [](unsigned long long bits) { return __builtin_bit_cast(double, bits); };
  double p = fma_sc(r, C(0x3e5ae64567f544e4ull), C(0x3e928af3fca7ab0cull));  // ~ 1/12!, 1/11!
  p = fma_sc(p, r, C(0x3ec71dee623fde64ull));
  p = fma_sc(p, r, C(0x3efa01997c89e6b0ull));
  p = fma_sc(p, r, C(0x3f2a01a014761f6eull));
  p = fma_sc(p, r, C(0x3f56c16c1852b7b0ull));
  p = fma_sc(p, r, C(0x3f81111111122322ull));
  p = fma_sc(p, r, C(0x3fa55555555502a1ull));
  p = fma_sc(p, r, C(0x3fc5555555555511ull));
  p = fma_sc(p, r, C(0x3fe000000000000bull));
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)__builtin_fmax(n, -1100.0));
#endif
}
__device__ __forceinline__ float sqrt_fast(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ double sqrt_fast(double x) { return ::sqrt(x); }

// squared distance -> covariance; same formulas as kernel_eval/metric_arg in mgp_device.h
// (_src/gp/kernels/numpy.py:12-31, gp/deformation/metric.py:241,264)
template <typename T>
__device__ __forceinline__ T cov_from_sqdist(T acc, int kernel_id, int metric_id, T post_scale) {
  const T x = (metric_id == MGP_METRIC_L2 ? sqrt_fast(acc) : acc) * post_scale;
  switch (kernel_id) {  // callers pass compile-time ids (see KERNEL_DISPATCH): no branch survives
    case MGP_KERNEL_RBF:
      return exp_neg(x * T(0.5));
    case MGP_KERNEL_MATERN_05:
      return exp_neg(x);
    case MGP_KERNEL_MATERN_15: {
      const T t = x * T(1.7320508075688772935);
      return (T(1) + t) * exp_neg(t);
    }
    case MGP_KERNEL_MATERN_25: {
      const T t = x * T(2.2360679774997896964);
      return (T(1) + t + t * t * T(1.0 / 3.0)) * exp_neg(t);
    }
    default:
      return exp_neg(x * x * T(0.5));
  }
}

// CB fp64 covariances evaluated STAGE BY STAGE (every stage a loop over the batch), so that the CB dependent chains --
// a software square root and a software exp of ~30 instructions between them -- are interleaved in the instruction
// stream: one evaluation after the other (what the compiler makes of CB separate calls) leaves every v_fma_f64 waiting
// for its predecessor's result at two waves per SIMD.  The square root is the hardware's reciprocal-square-root
// estimate with one Goldschmidt step and one Newton correction (the library's version adds denormal scaling, a second
// correction and a class test: 18 instructions; here 9): its argument is a squared distance -- zero (identical rows)
// is lifted by adding 1e-280 (a no-op for every x >= 1e-264), whose root, 1e-140, is zero for every covariance function; relative error < 2^-50.  The exp is
// exp_neg() above without the clamp on the exponent (v_cvt_i32_f64 saturates, v_ldexp_f64 underflows to zero).
// v_cvt_i32_f64 saturates; the C cast `(int)x` of an out-of-range double is poison to the optimiser (advisor, round 4).
__device__ __forceinline__ int cvt_i32_sat(double x) {
  int i;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(i) : "v"(x));
  return i;
}
template <int CB, int KID, int MID>
__device__ __forceinline__ void cov_batch64(double (&v)[CB], double post_scale) {
  constexpr auto C = [](unsigned long long bits) { return __builtin_bit_cast(double, bits); };
  double t[CB], w[CB];
  if constexpr (MID == MGP_METRIC_L2) {
    double y[CB], h[CB];
#pragma unroll
    for (int u = 0; u < CB; ++u) v[u] = v[u] + 1e-280;  // (one v_add_f64; fmax() costs a canonicalising second instruction)
#pragma unroll
    for (int u = 0; u < CB; ++u) y[u] = __builtin_amdgcn_rsq(v[u]);
#pragma unroll
    for (int u = 0; u < CB; ++u) t[u] = v[u] * y[u];  // g ~ sqrt(x)
#pragma unroll
    for (int u = 0; u < CB; ++u) h[u] = y[u] * 0.5;
#pragma unroll
    for (int u = 0; u < CB; ++u) y[u] = __builtin_fma(-h[u], t[u], 0.5);
#pragma unroll
    for (int u = 0; u < CB; ++u) t[u] = __builtin_fma(t[u], y[u], t[u]);
#pragma unroll
    for (int u = 0; u < CB; ++u) h[u] = __builtin_fma(h[u], y[u], h[u]);
#pragma unroll
    for (int u = 0; u < CB; ++u) y[u] = __builtin_fma(-t[u], t[u], v[u]);
#pragma unroll
    for (int u = 0; u < CB; ++u) v[u] = __builtin_fma(y[u], h[u], t[u]);
  }
  // kernel argument x = metric * post_scale; t = what goes into exp(-t); w = the polynomial factor
#pragma unroll
  for (int u = 0; u < CB; ++u) {
    const double x = v[u] * post_scale;
    if constexpr (KID == MGP_KERNEL_RBF) {
      t[u] = x * 0.5;
      w[u] = 1.0;
    } else if constexpr (KID == MGP_KERNEL_MATERN_05) {
      t[u] = x;
      w[u] = 1.0;
    } else if constexpr (KID == MGP_KERNEL_MATERN_15) {
      t[u] = x * 1.7320508075688772935;
      w[u] = 1.0 + t[u];
    } else if constexpr (KID == MGP_KERNEL_MATERN_25) {
      t[u] = x * 2.2360679774997896964;
      w[u] = 1.0 + t[u] + t[u] * t[u] * (1.0 / 3.0);
    } else {
      t[u] = x * x * 0.5;
      w[u] = 1.0;
    }
  }
  // exp(-800) = 0 in fp64: the clamp keeps an extreme scaled distance (length scale at a 1e-5 bound, an infinite
  // feature) a plain number through the range reduction below (inf - inf otherwise) and returns covariance 0 like the
  // reference's exp(-inf); v_min_f64 also drops a NaN from inf * rsq(inf) = inf * 0.
  double n[CB], r[CB], q[CB];
#pragma unroll
  for (int u = 0; u < CB; ++u) t[u] = __builtin_fmin(t[u], 800.0);
#pragma unroll
  for (int u = 0; u < CB; ++u) n[u] = __builtin_rint(t[u] * -1.4426950408889634074);
#pragma unroll
  for (int u = 0; u < CB; ++u) r[u] = __builtin_fma(n[u], -6.93147180369123816490e-01, -t[u]);
#pragma unroll
  for (int u = 0; u < CB; ++u) r[u] = __builtin_fma(n[u], -1.90821492927058770002e-10, r[u]);
#pragma unroll
  for (int u = 0; u < CB; ++u) q[u] = fma_sc(r[u], C(0x3e5ae64567f544e4ull), C(0x3e928af3fca7ab0cull));
  constexpr unsigned long long coef[8] = {0x3ec71dee623fde64ull, 0x3efa01997c89e6b0ull, 0x3f2a01a014761f6eull, 0x3f56c16c1852b7b0ull,
                                          0x3f81111111122322ull, 0x3fa55555555502a1ull, 0x3fc5555555555511ull, 0x3fe000000000000bull};
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int u = 0; u < CB; ++u) q[u] = fma_sc(q[u], r[u], C(coef[c]));
#pragma unroll
  for (int u = 0; u < CB; ++u) q[u] = __builtin_fma(q[u], r[u], 1.0);
#pragma unroll
  for (int u = 0; u < CB; ++u) q[u] = __builtin_fma(q[u], r[u], 1.0);
#pragma unroll
  for (int u = 0; u < CB; ++u) q[u] = __builtin_amdgcn_ldexp(q[u], cvt_i32_sat(n[u]));
#pragma unroll
  for (int u = 0; u < CB; ++u) {
    if constexpr (KID == MGP_KERNEL_MATERN_15 || KID == MGP_KERNEL_MATERN_25) v[u] = w[u] * q[u];
    else v[u] = q[u];
  }
}

// d covariance / d (squared distance accumulator), one pair, any element type (the fp32 backward instantiations of the
// wave kernel: hardware exp / rsq; fp64 callers use the staged dcov_batch64 below).  Same formulas.
template <typename T, int KID, int MID>
__device__ __forceinline__ T dcov_dacc(T acc, T post_scale) {
  if constexpr (MID == MGP_METRIC_L2) {
    const T ps2 = post_scale * post_scale;
    const T x = sqrt_fast(acc) * post_scale;
    if constexpr (KID == MGP_KERNEL_MATERN_15) {
      return T(-1.5) * ps2 * exp_neg(x * T(1.7320508075688772935));
    } else if constexpr (KID == MGP_KERNEL_MATERN_25) {
      const T t = x * T(2.2360679774997896964);
      return T(-5.0 / 6.0) * ps2 * (T(1) + t) * exp_neg(t);
    } else if constexpr (KID == MGP_KERNEL_MATERN_INF) {
      return T(-0.5) * ps2 * exp_neg(x * x * T(0.5));
    } else {
      // k'(x) / (2 sqrt(acc)): singular at acc = 0, where the forward's kink leaves no derivative either -> 0
      const T h = acc > T(0) ? T(0.5) * post_scale / sqrt_fast(acc) : T(0);
      if constexpr (KID == MGP_KERNEL_RBF) return T(-0.5) * exp_neg(x * T(0.5)) * h;
      else return -exp_neg(x) * h;
    }
  } else {
    const T x = acc * post_scale;
    if constexpr (KID == MGP_KERNEL_RBF) {
      return T(-0.5) * post_scale * exp_neg(x * T(0.5));
    } else if constexpr (KID == MGP_KERNEL_MATERN_05) {
      return -post_scale * exp_neg(x);
    } else if constexpr (KID == MGP_KERNEL_MATERN_15) {
      return T(-3) * x * post_scale * exp_neg(x * T(1.7320508075688772935));
    } else if constexpr (KID == MGP_KERNEL_MATERN_25) {
      const T t = x * T(2.2360679774997896964);
      return T(-5.0 / 3.0) * x * (T(1) + t) * post_scale * exp_neg(t);
    } else {
      return -x * post_scale * exp_neg(x * x * T(0.5));
    }
  }
}

// d covariance / d (squared distance accumulator) of CB pairs, fp64, stage by stage like cov_batch64() (the backward of
// the dealt-triangle kernels, mgp_backward_dlt.hip).  in: v[u] = acc (the squared distance of the rows as the kernel
// holds them: scaled by the inverse length scales under Anisotropy, raw under Isotropy); out: v[u] = dk / d acc, with
// x = metric(acc) * post_scale the kernel's argument:
//   dk/dacc = k'(x) dx/dacc,  dx/dacc = post_scale (F2)  |  post_scale / (2 sqrt(acc)) (l2)
//   k'(x):  RBF -1/2 e^(-x/2) | Matern-1/2 -e^(-x) | 3/2 -3 x e^(-sqrt3 x) | 5/2 -5/3 x (1 + t) e^(-t), t = sqrt5 x |
//           inf -x e^(-x^2/2)
// Under l2 the factor x of the last three cancels 1 / sqrt(acc): no division and no singularity at acc = 0; the first
// two keep h ~ 1 / (2 sqrt(acc)) from the square-root iteration and return 0 at acc = 0 (mgp_backward.hip does too).
// (The chain-rule term of the isotropic length scale, gK k'(x) x, is dk/dacc * acc * (l2 ? 2 : 1): the caller's.)
template <int CB, int KID, int MID>
__device__ __forceinline__ void dcov_batch64(double (&v)[CB], double post_scale) {
  constexpr auto C = [](unsigned long long bits) { return __builtin_bit_cast(double, bits); };
  double t[CB], w[CB];
  constexpr bool L2 = MID == MGP_METRIC_L2;
  constexpr bool XCANCELS = KID == MGP_KERNEL_MATERN_15 || KID == MGP_KERNEL_MATERN_25 || KID == MGP_KERNEL_MATERN_INF;
  const double ps2 = post_scale * post_scale;
  if constexpr (L2) {
    double y[CB], h[CB], a0[CB];
#pragma unroll
    for (int u = 0; u < CB; ++u) a0[u] = v[u];
#pragma unroll
    for (int u = 0; u < CB; ++u) v[u] = v[u] + 1e-280;
#pragma unroll
    for (int u = 0; u < CB; ++u) y[u] = __builtin_amdgcn_rsq(v[u]);
#pragma unroll
    for (int u = 0; u < CB; ++u) t[u] = v[u] * y[u];
#pragma unroll
    for (int u = 0; u < CB; ++u) h[u] = y[u] * 0.5;
#pragma unroll
    for (int u = 0; u < CB; ++u) y[u] = __builtin_fma(-h[u], t[u], 0.5);
#pragma unroll
    for (int u = 0; u < CB; ++u) t[u] = __builtin_fma(t[u], y[u], t[u]);
#pragma unroll
    for (int u = 0; u < CB; ++u) h[u] = __builtin_fma(h[u], y[u], h[u]);
#pragma unroll
    for (int u = 0; u < CB; ++u) y[u] = __builtin_fma(-t[u], t[u], v[u]);
#pragma unroll
    for (int u = 0; u < CB; ++u) v[u] = __builtin_fma(y[u], h[u], t[u]);  // sqrt(acc)
    // w = k'(x) dx/dacc without its exponential
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      const double x = v[u] * post_scale;
      if constexpr (KID == MGP_KERNEL_RBF) {
        t[u] = x * 0.5;
        w[u] = a0[u] > 0.0 ? -0.5 * post_scale * h[u] : 0.0;
      } else if constexpr (KID == MGP_KERNEL_MATERN_05) {
        t[u] = x;
        w[u] = a0[u] > 0.0 ? -post_scale * h[u] : 0.0;
      } else if constexpr (KID == MGP_KERNEL_MATERN_15) {
        t[u] = x * 1.7320508075688772935;
        w[u] = -1.5 * ps2;
      } else if constexpr (KID == MGP_KERNEL_MATERN_25) {
        t[u] = x * 2.2360679774997896964;
        w[u] = (-5.0 / 6.0) * ps2 * (1.0 + t[u]);
      } else {
        t[u] = x * x * 0.5;
        w[u] = -0.5 * ps2;
      }
    }
  } else {
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      const double x = v[u] * post_scale;
      if constexpr (KID == MGP_KERNEL_RBF) {
        t[u] = x * 0.5;
        w[u] = -0.5 * post_scale;
      } else if constexpr (KID == MGP_KERNEL_MATERN_05) {
        t[u] = x;
        w[u] = -post_scale;
      } else if constexpr (KID == MGP_KERNEL_MATERN_15) {
        t[u] = x * 1.7320508075688772935;
        w[u] = -3.0 * x * post_scale;
      } else if constexpr (KID == MGP_KERNEL_MATERN_25) {
        t[u] = x * 2.2360679774997896964;
        w[u] = (-5.0 / 3.0) * x * (1.0 + t[u]) * post_scale;
      } else {
        t[u] = x * x * 0.5;
        w[u] = -x * post_scale;
      }
    }
  }
  (void)XCANCELS;
  double n[CB], r[CB], q[CB];
#pragma unroll
  for (int u = 0; u < CB; ++u) t[u] = __builtin_fmin(t[u], 800.0);
#pragma unroll
  for (int u = 0; u < CB; ++u) n[u] = __builtin_rint(t[u] * -1.4426950408889634074);
#pragma unroll
  for (int u = 0; u < CB; ++u) r[u] = __builtin_fma(n[u], -6.93147180369123816490e-01, -t[u]);
#pragma unroll
  for (int u = 0; u < CB; ++u) r[u] = __builtin_fma(n[u], -1.90821492927058770002e-10, r[u]);
#pragma unroll
  for (int u = 0; u < CB; ++u) q[u] = fma_sc(r[u], C(0x3e5ae64567f544e4ull), C(0x3e928af3fca7ab0cull));
  constexpr unsigned long long coef[8] = {0x3ec71dee623fde64ull, 0x3efa01997c89e6b0ull, 0x3f2a01a014761f6eull, 0x3f56c16c1852b7b0ull,
                                          0x3f81111111122322ull, 0x3fa55555555502a1ull, 0x3fc5555555555511ull, 0x3fe000000000000bull};
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int u = 0; u < CB; ++u) q[u] = fma_sc(q[u], r[u], C(coef[c]));
#pragma unroll
  for (int u = 0; u < CB; ++u) q[u] = __builtin_fma(q[u], r[u], 1.0);
#pragma unroll
  for (int u = 0; u < CB; ++u) q[u] = __builtin_fma(q[u], r[u], 1.0);
#pragma unroll
  for (int u = 0; u < CB; ++u) q[u] = __builtin_amdgcn_ldexp(q[u], cvt_i32_sat(n[u]));
#pragma unroll
  for (int u = 0; u < CB; ++u) v[u] = w[u] * q[u];
}

// Two covariances at once (fp32): the polynomial / scaling arithmetic runs as packed ops
// (v_pk_mul_f32 / v_pk_fma_f32), only v_sqrt_f32 and v_exp_f32 stay scalar.  Same formulas and
// the same two-term log2(e) as the scalar form above.
__device__ __forceinline__ f2 exp_neg2(f2 t) {
#ifndef MGP_EXP_ONE_TERM
  const f2 c_hi = {-1.44269502162933349609375f, -1.44269502162933349609375f};
  const f2 c_lo = {-1.925963033500011e-08f, -1.925963033500011e-08f};
  const f2 hi = t * c_hi;
  f2 lo = t * c_hi - hi;  // contracted to v_pk_fma_f32: the rounding error of hi
  lo = t * c_lo + lo;
  f2 e;
  e.x = __builtin_amdgcn_exp2f(hi.x);
  e.y = __builtin_amdgcn_exp2f(hi.y);
  const f2 ln2 = {0.693147180559945f, 0.693147180559945f};
  return (e * lo) * ln2 + e;
#else
  // -DMGP_EXP_ONE_TERM: one packed multiply + v_exp_f32 (relative error |t log2 e| 2^-24 ln 2, i.e.
  // absolute error < 3e-8 for every t).  Passes every parity test, saves 5 packed instructions per
  // pair -- and measured no faster on the headline shape (2.10-2.12 vs 2.08-2.11 ms: the covariance
  // phase is not on the critical path), so the ~2 ulp two-term form above stays the default.
  const f2 a = t * f2{-1.44269502162933349609375f, -1.44269502162933349609375f};
  f2 e;
  e.x = __builtin_amdgcn_exp2f(a.x);
  e.y = __builtin_amdgcn_exp2f(a.y);
  return e;
#endif
}
__device__ __forceinline__ f2 cov_from_sqdist2(f2 acc, int kernel_id, int metric_id, float post_scale) {
  f2 x = acc;
  if (metric_id == MGP_METRIC_L2) {
    x.x = sqrt_fast(acc.x);
    x.y = sqrt_fast(acc.y);
  }
  x = x * f2{post_scale, post_scale};
  switch (kernel_id) {
    case MGP_KERNEL_RBF:
      return exp_neg2(x * f2{0.5f, 0.5f});
    case MGP_KERNEL_MATERN_05:
      return exp_neg2(x);
    case MGP_KERNEL_MATERN_15: {
      const f2 t = x * f2{1.7320508075688772935f, 1.7320508075688772935f};
      return (f2{1.0f, 1.0f} + t) * exp_neg2(t);
    }
    case MGP_KERNEL_MATERN_25: {
      const f2 t = x * f2{2.2360679774997896964f, 2.2360679774997896964f};
      const f2 third = {1.0f / 3.0f, 1.0f / 3.0f};
      return (f2{1.0f, 1.0f} + t + t * t * third) * exp_neg2(t);
    }
    default:
      return exp_neg2(x * x * f2{0.5f, 0.5f});
  }
}

// ---- general-smoothness Matern inside the fused kernels (fp32) ------------------------------------------
// k(r) = 2^(1-nu)/Gamma(nu) x^nu K_nu(x), x = sqrt(2 nu) r (_src/gp/kernels/numpy.py:34-43).  nu is uniform
// per launch, so K_nu(x) = int_0^inf exp(-x cosh t) cosh(nu t) dt is evaluated by the trapezoidal rule on
// nodes t_n = n h that every lane shares: the rule converges exponentially for this entire integrand
// (error ~ exp(-pi^2 / h) at moderate nu; h shrinks with nu, gen_step()), and with everything kept in the
// log domain,
//     term_n = exp2( nu log2 x + lc + l_n - x c_n ),   c_n = cosh(t_n) log2 e,  l_n = log2 cosh(nu t_n) [- 1 at n = 0],
// no term exceeds the result (<= 1), whatever nu.  The node table (-c_n, l_n) is built once per workgroup
// in LDS (gen_build_table); a covariance costs N(x) nodes x (one packed add, one packed FMA, one v_exp_f32,
// one packed add per TWO covariances), N = 5..13 for x in [0.5, 10] -- against ~10^3 fp64 instructions of
// the Temme / Steed evaluation the per-function kernel uses (mgp_tensor_ops.hip).  Checked against scipy
// in tests/test_matern_gen_cpu.py (the same arithmetic restated in numpy float32): <= 4e-6 absolute for
// nu <= 4, <= 3e-5 up to nu = 25, x in [1e-5, 80].
#define MGP_GEN_NODES 128
__host__ __device__ inline float gen_step(double nu) { return nu <= 2.0 ? 0.5f : (nu <= 4.0 ? 0.4f : (nu <= 10.0 ? 0.3f : 0.22f)); }

__device__ __forceinline__ void gen_build_table(float* tab, float nu, float h, int lane) {
  for (int n = lane; n < MGP_GEN_NODES; n += 64) {
    const float t = n * h, a = nu * t;
    tab[2 * n] = -coshf(t) * 1.44269504088896f;
    tab[2 * n + 1] = (a + log1pf(expf(-2.0f * a)) - 0.693147180559945f) * 1.44269504088896f - (n == 0 ? 1.0f : 0.0f);
  }
}

// in: r[s] = metric argument (scaled distance) of the lane's NS pairs; out: the covariances.  `live`: bit s
// set for pairs whose value is used (the others are evaluated at x = 1: their distance may be garbage and
// must not drive the node count).  lc = log2(h 2^(1-nu) / Gamma(nu)).
template <int NS>
__device__ __forceinline__ void matern_gen_eval(float (&r)[NS], unsigned live, const float* tab, float nu, float h, float lc) {
  constexpr int NP2 = (NS + 1) / 2;
  f2 x2[NP2], L2[NP2], acc2[NP2];
  const float s2nu = __builtin_amdgcn_sqrtf(2.0f * nu);
  float nmaxf = 1.0f;
  unsigned zero = 0;
#pragma unroll
  for (int s = 0; s < NP2 * 2; ++s) {
    float x = 1.0f;
    if (s < NS && ((live >> s) & 1u)) {
      if (r[s] == 0.0f) zero |= 1u << s;  // identical rows: k(0) = 1 (the reference nudges zeros to eps); a NaN distance stays NaN
      x = r[s] != r[s] ? r[s] : __builtin_fminf(__builtin_fmaxf(r[s] * s2nu, 1e-6f), 3.0e4f);
    }
    const float lx = __builtin_amdgcn_logf(x);  // log2
    // nodes until x cosh t - nu t > ~22: T = ln(2 (22 + nu max(1, ln(44 / x))) / x)
    const float inner = 22.0f + nu * __builtin_fmaxf(1.0f, (5.4594316f - lx) * 0.693147181f);
    const float T = (__builtin_amdgcn_logf(2.0f * inner) - lx) * 0.693147181f;
    nmaxf = __builtin_fmaxf(nmaxf, T);
    const float Ls = __builtin_fmaf(nu, lx, lc);
    if (s & 1) {
      x2[s / 2].y = x;
      L2[s / 2].y = Ls;
    } else {
      x2[s / 2].x = x;
      L2[s / 2].x = Ls;
    }
  }
  int N = (int)(nmaxf / h) + 3;
  N = N < MGP_GEN_NODES ? N : MGP_GEN_NODES;
#pragma unroll
  for (int p = 0; p < NP2; ++p) acc2[p] = f2{0.0f, 0.0f};
  for (int n = 0; n < MGP_GEN_NODES; ++n) {
    if (__builtin_amdgcn_ballot_w64(n < N) == 0) break;  // (uniform: the longest tail of the wave)
    const f2 cl = *reinterpret_cast<const f2*>(tab + 2 * n);  // uniform LDS read: (-c_n, l_n)
    const f2 negc = {cl.x, cl.x}, l = {cl.y, cl.y};
#pragma unroll
    for (int p = 0; p < NP2; ++p) {
      const f2 arg = x2[p] * negc + (L2[p] + l);
      f2 e;
      e.x = __builtin_amdgcn_exp2f(arg.x);
      e.y = __builtin_amdgcn_exp2f(arg.y);
      acc2[p] = acc2[p] + e;
    }
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) r[s] = (zero >> s) & 1u ? 1.0f : (s & 1 ? acc2[s / 2].y : acc2[s / 2].x);
}

// ---- the same in fp64 (round 4) ----------------------------------------------------------------------------------
// Natural logarithms, the software exp of exp_neg(), the trapezoidal rule with a finer step (error ~ exp(-pi^2 / h):
// h = 0.30 / 0.25 / 0.18 / 0.13 for nu <= 2 / 4 / 10 / 30 -> <= 6e-13 absolute against scipy over r in [1e-7, 20],
// tests/test_matern_gen_cpu.py) and nodes until x cosh t - nu t > 37.  CB covariances per call, stage by stage like
// cov_batch64(); the node count is the wave's maximum.  A zero distance is the reference's eps
// (_src/gp/kernels/numpy.py:38); x is clamped from below to what the node table covers (`xmin`, a launch constant:
// ~1e-11 at nu = 30, far smaller below -- |k(x) - k(xmin)| is of that order at most).
#define MGP_GEN_NODES64 256
__host__ __device__ inline double gen_step64(double nu) { return nu <= 2.0 ? 0.30 : (nu <= 4.0 ? 0.25 : (nu <= 10.0 ? 0.18 : 0.13)); }
// nodes needed at scaled distance x: T(x) / h + 3 with T = ln(2 (37 + nu max(1, ln(74 / x))) / x)
__host__ __device__ inline double gen_span64(double x, double nu) {
  const double lx = log(x);
  const double inner = 37.0 + nu * fmax(1.0, 4.30406509320417 - lx);
  return log(2.0 * inner) - lx;
}
#ifndef __HIPCC_RTC__
inline double gen_xmin64(double nu) {  // smallest x the MGP_GEN_NODES64-node table integrates to the end (bisection, host)
  const double reach = (MGP_GEN_NODES64 - 4) * gen_step64(nu);
  double lo = -300.0, hi = 0.0;  // log10 x
  for (int it = 0; it < 80; ++it) {
    const double mid = 0.5 * (lo + hi);
    (gen_span64(pow(10.0, mid), nu) > reach ? lo : hi) = mid;
  }
  return pow(10.0, hi);
}
#endif
__device__ __forceinline__ void gen_build_table64(double* tab, double nu, double h, int lane) {
  for (int n = lane; n < MGP_GEN_NODES64; n += 64) {
    const double t = n * h, a = nu * t;
    tab[2 * n] = -cosh(t);
    tab[2 * n + 1] = a + log1p(exp(-2.0 * a)) - 0.693147180559945309417 - (n == 0 ? 0.693147180559945309417 : 0.0);
  }
}
// in: v[u] = metric argument (scaled distance) of CB pairs, `live` bit u set for pairs whose value is used; out: covariances
template <int CB>
__device__ __forceinline__ void matern_gen_batch64(double (&v)[CB], unsigned live, const double* tab, double nu, double h, double lc,
                                                   double xmin) {
  constexpr auto C = [](unsigned long long bits) { return __builtin_bit_cast(double, bits); };
  double x[CB], L[CB], acc[CB];
  const double s2nu = ::sqrt(2.0 * nu);
  double tmax = 1.0;
#pragma unroll
  for (int u = 0; u < CB; ++u) {
    double xx = 1.0;
    if ((live >> u) & 1u) {
      const double r = v[u] == 0.0 ? 2.220446049250313e-16 : v[u];
      xx = r != r ? r : __builtin_fmin(__builtin_fmax(r * s2nu, xmin), 1.0e5);
    }
    const double lx = ::log(xx);
    tmax = __builtin_fmax(tmax, ::log(2.0 * (37.0 + nu * __builtin_fmax(1.0, 4.30406509320417 - lx))) - lx);
    x[u] = xx;
    L[u] = __builtin_fma(nu, lx, lc);
    acc[u] = 0.0;
  }
  int N = (int)(tmax / h) + 3;
  N = N < MGP_GEN_NODES64 ? N : MGP_GEN_NODES64;
  for (int n = 0; n < MGP_GEN_NODES64; ++n) {
    if (__builtin_amdgcn_ballot_w64(n < N) == 0) break;  // (uniform: the longest tail of the wave)
    const double negc = tab[2 * n], l = tab[2 * n + 1];  // uniform LDS reads: (-cosh t_n, ln cosh(nu t_n) [- ln 2 at n = 0])
    // exp(L + l - x cosh t), CB chains interleaved (exp_neg() of mgp_wave_common.h, stage by stage)
    double t[CB], m[CB], r[CB], q[CB];
#pragma unroll
    for (int u = 0; u < CB; ++u) t[u] = -__builtin_fma(x[u], negc, L[u] + l);
#pragma unroll
    for (int u = 0; u < CB; ++u) m[u] = __builtin_rint(t[u] * -1.4426950408889634074);
#pragma unroll
    for (int u = 0; u < CB; ++u) r[u] = __builtin_fma(m[u], -6.93147180369123816490e-01, -t[u]);
#pragma unroll
    for (int u = 0; u < CB; ++u) r[u] = __builtin_fma(m[u], -1.90821492927058770002e-10, r[u]);
#pragma unroll
    for (int u = 0; u < CB; ++u) q[u] = fma_sc(r[u], C(0x3e5ae64567f544e4ull), C(0x3e928af3fca7ab0cull));
    constexpr unsigned long long coef[8] = {0x3ec71dee623fde64ull, 0x3efa01997c89e6b0ull, 0x3f2a01a014761f6eull, 0x3f56c16c1852b7b0ull,
                                            0x3f81111111122322ull, 0x3fa55555555502a1ull, 0x3fc5555555555511ull, 0x3fe000000000000bull};
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int u = 0; u < CB; ++u) q[u] = fma_sc(q[u], r[u], C(coef[c]));
#pragma unroll
    for (int u = 0; u < CB; ++u) q[u] = __builtin_fma(q[u], r[u], 1.0);
#pragma unroll
    for (int u = 0; u < CB; ++u) q[u] = __builtin_fma(q[u], r[u], 1.0);
#pragma unroll
    for (int u = 0; u < CB; ++u) acc[u] += __builtin_amdgcn_ldexp(q[u], cvt_i32_sat(m[u]));
  }
#pragma unroll
  for (int u = 0; u < CB; ++u) v[u] = acc[u];
}
// metric argument of CB squared distances in fp64: the lean square root of cov_batch64() under the l2 metric
template <int CB, int MID>
__device__ __forceinline__ void metric_batch64(double (&v)[CB], double post_scale) {
  if constexpr (MID == MGP_METRIC_L2) {
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      if (v[u] > 0.0) {  // (an exact zero stays zero: the general Matern treats it as the reference does)
        const double y = __builtin_amdgcn_rsq(v[u]);
        double g = v[u] * y, hh = y * 0.5;
        const double e = __builtin_fma(-hh, g, 0.5);
        g = __builtin_fma(g, e, g);
        hh = __builtin_fma(hh, e, hh);
        v[u] = __builtin_fma(__builtin_fma(-g, g, v[u]), hh, g);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < CB; ++u) v[u] = v[u] * post_scale;
}

// value of x in a given (wave-uniform) lane
__device__ __forceinline__ float lane_value(float x, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane));
}
__device__ __forceinline__ double lane_value(double x, int lane) {
  const long long b = __double_as_longlong(x);
  const unsigned lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), lane);
  const unsigned hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | lo);
}

// 16 bytes per lane straight from global memory into LDS (no VGPR round trip): the LDS
// destination is the wave-uniform pointer + lane * 16, the global source is per lane.
// same, destination given as the (address-space-3) shared array itself + a byte offset: no generic ->
// LDS pointer conversion (a null test and four scalar instructions per load) is left in the code
template <typename SharedArray>
__device__ __forceinline__ void glds16_lds(const void* gsrc, SharedArray& shared, int byte_offset) {
  typedef __attribute__((address_space(3))) char lds_char;
  lds_char* base = (lds_char*)(&shared[0]);
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)(base + byte_offset), 16, 0, 0);
}
// Same transfer issued as inline assembly.  The compiler books `global_load_lds` as a FLAT operation
// that touches both memory and LDS; while one is pending, its wait-count pass turns EVERY LDS wait
// into `s_waitcnt lgkmcnt(0)` (and every memory wait into vmcnt(0)) -- for a kernel that keeps the
// next tile in flight during the whole elimination that means no counted waits at all: each step
// waited for the pivot group it had just requested for the NEXT step.  Issued from here the compiler
// does not see the transfer; the consumer must wait for it itself (lds_dma_wait) before reading the
// tile.  M0 carries the LDS byte address of lane 0's 16 bytes (one wait state between the write of
// M0 and the transfer); the compiler keeps nothing in M0 across statements on gfx950.
template <typename SharedArray>
__device__ __forceinline__ void glds16_asm(const void* gsrc, SharedArray& shared, int byte_offset) {
  typedef __attribute__((address_space(3))) char lds_char;
  const unsigned lds = (unsigned)(size_t)(lds_char*)(&shared[0]) + (unsigned)byte_offset;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds) : "memory");
}
__device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Call f(ic<KID>, ic<MID>) with kernel id and metric id as compile-time constants: the uniform
// switch is taken once per task instead of once per matrix entry.
template <int V> struct ic { static constexpr int value = V; };
template <int MID, typename F>
__device__ __forceinline__ void kernel_dispatch_m(int kernel_id, F&& f) {
  switch (kernel_id) {
    case MGP_KERNEL_RBF: f(ic<MGP_KERNEL_RBF>{}, ic<MID>{}); break;
    case MGP_KERNEL_MATERN_05: f(ic<MGP_KERNEL_MATERN_05>{}, ic<MID>{}); break;
    case MGP_KERNEL_MATERN_15: f(ic<MGP_KERNEL_MATERN_15>{}, ic<MID>{}); break;
    case MGP_KERNEL_MATERN_25: f(ic<MGP_KERNEL_MATERN_25>{}, ic<MID>{}); break;
    default: f(ic<MGP_KERNEL_MATERN_INF>{}, ic<MID>{}); break;
  }
}
template <typename F>
__device__ __forceinline__ void kernel_dispatch(int kernel_id, int metric_id, F&& f) {
  if (metric_id == MGP_METRIC_L2) kernel_dispatch_m<MGP_METRIC_L2>(kernel_id, f);
  else kernel_dispatch_m<MGP_METRIC_F2>(kernel_id, f);
}

// the same with the general-smoothness Matern as a sixth kernel id (the wave kernels, fp32)
template <typename F>
__device__ __forceinline__ void kernel_dispatch_gen(int kernel_id, int metric_id, F&& f) {
  if (kernel_id == MGP_KERNEL_MATERN_GEN) {
    if (metric_id == MGP_METRIC_L2) f(ic<MGP_KERNEL_MATERN_GEN>{}, ic<MGP_METRIC_L2>{});
    else f(ic<MGP_KERNEL_MATERN_GEN>{}, ic<MGP_METRIC_F2>{});
  } else {
    kernel_dispatch(kernel_id, metric_id, f);
  }
}

#ifndef __HIPCC_RTC__
// f(ic<0>{}), f(ic<1>{}), ... f(ic<N-1>{}): a loop whose index is a compile-time constant in every
// iteration (a `#pragma unroll` the compiler declines leaves the 128-register row indexed at run
// time, i.e. in scratch memory)
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(ic<I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

#endif  // !__HIPCC_RTC__

}  // namespace mgp

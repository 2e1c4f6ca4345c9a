// Microbenchmarks that settle design questions for the fused wave kernel on gfx950 (MI355X).
// Stand-alone: hipcc --offload-arch=gfx950 -O3 -o ubench ubench.hip ; ./ubench [test ...]
//
//   valu    cycles per wave64 VALU instruction (v_fma_f32, v_pk_fma_f32, v_pk_add_f32, v_exp_f32 ...)
//           at 1..4 waves per SIMD                           -> is a wave64 op 2 or 4 SIMD cycles?
//   mfma    cycles per f32-input MFMA of every shape (one wave per SIMD)
//   coexec  an MFMA wave and a VALU wave on the same SIMD: what each loses
//   mix     one wave alternating 1 MFMA + n VALU: how many VALU hide under one MFMA
//   lds     ds_read_b128 (broadcast / conflict-free), ds_write_b32, ds_write_b128 per CU
//   gather  row-gather floor (direct-to-LDS) for candidate table layouts: 160-B rows + separate
//           target table, packed rows of 176 / 192 / 256 B carrying the target
//
// Cycle counts are s_memtime deltas taken inside the kernel (shader clock), median over waves.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f32v __attribute__((ext_vector_type(32)));
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long memtime() { return __builtin_amdgcn_s_memtime(); }

// ------------------------------------------------------------------------------------------------
// VALU issue rate
// ------------------------------------------------------------------------------------------------
enum { OP_FMA, OP_PKFMA, OP_PKADD, OP_PKMUL, OP_EXP, OP_SQRT, OP_RCP, OP_MUL, OP_FMA64, OP_ADD, OP_COUNT };
static const char* op_name[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_exp_f32",
                                "v_sqrt_f32", "v_rcp_f32", "v_mul_f32", "v_fma_f64", "v_add_f32"};

#define R8(X) X X X X X X X X

template <int OP>
__device__ __forceinline__ void valu_body(float (&a)[16], f2 (&p)[16], double (&q)[8], float s, f2 s2, double sd) {
  // 16 independent instructions per call (dependent distance 16)
  if constexpr (OP == OP_FMA) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s));
  } else if constexpr (OP == OP_ADD) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
  } else if constexpr (OP == OP_MUL) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
  } else if constexpr (OP == OP_EXP) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
  } else if constexpr (OP == OP_SQRT) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
  } else if constexpr (OP == OP_RCP) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
  } else if constexpr (OP == OP_PKFMA) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(s2));
  } else if constexpr (OP == OP_PKADD) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
  } else if constexpr (OP == OP_PKMUL) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
  } else if constexpr (OP == OP_FMA64) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(q[i]) : "v"(sd));
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(q[i]) : "v"(sd));
  }
}

template <int OP>
__global__ void valu_kernel(float* out, int iters, unsigned long long* cyc) {
  extern __shared__ char smem[];  // sized to force one workgroup per CU
  float a[16];
  f2 p[16];
  double q[8];
  for (int i = 0; i < 16; ++i) {
    a[i] = 1.0f + threadIdx.x * 1e-6f + i;
    p[i] = f2{a[i], a[i] + 0.5f};
  }
  for (int i = 0; i < 8; ++i) q[i] = a[i];
  const float s = 0.999f;
  const f2 s2 = {0.999f, 0.998f};
  const double sd = 0.999;
  __syncthreads();
  const unsigned long long t0 = memtime();
  for (int it = 0; it < iters; ++it) {
    valu_body<OP>(a, p, q, s, s2, sd);
    valu_body<OP>(a, p, q, s, s2, sd);
    valu_body<OP>(a, p, q, s, s2, sd);
    valu_body<OP>(a, p, q, s, s2, sd);
  }
  const unsigned long long t1 = memtime();
  float r = 0;
  for (int i = 0; i < 16; ++i) r += a[i] + p[i].x + p[i].y;
  for (int i = 0; i < 8; ++i) r += (float)q[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

static double median(std::vector<unsigned long long>& v) {
  std::sort(v.begin(), v.end());
  return (double)v[v.size() / 2];
}

struct Bufs {
  float* out;
  unsigned long long* cyc;
  std::vector<unsigned long long> h;
};

static int g_cus = 256;

template <int OP>
static void run_valu(Bufs& B) {
  const int iters = 4000;
  for (int w = 1; w <= 4; ++w) {
    const int threads = 256 * w;
    hipLaunchKernelGGL(valu_kernel<OP>, dim3(g_cus), dim3(threads), 96 * 1024, 0, B.out, 10, B.cyc);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(valu_kernel<OP>, dim3(g_cus), dim3(threads), 96 * 1024, 0, B.out, iters, B.cyc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const int nw = g_cus * threads / 64;
    B.h.resize(nw);
    CK(hipMemcpy(B.h.data(), B.cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const double c = median(B.h);
    const double n = (double)iters * 64;
    printf("{\"test\":\"valu\",\"op\":\"%s\",\"waves_per_simd\":%d,\"cycles_per_instr_per_wave\":%.3f,"
           "\"simd_cycles_per_instr\":%.3f,\"kernel_ms\":%.3f,\"clock_ghz\":%.3f}\n",
           op_name[OP], w, c / n, c / n / w, ms, c / (ms * 1e6));
  }
}

// ------------------------------------------------------------------------------------------------
// MFMA shapes (f32 in, exact)
// ------------------------------------------------------------------------------------------------
enum { M_32x32x2, M_16x16x4, M_32x32x1_2B, M_16x16x1_4B, M_4x4x1_16B, M_F64_16x16x4, M_F64_4x4x4, M_BF16_32x32x16, M_BF16_16x16x32, M_COUNT };
static const char* mfma_name[] = {"v_mfma_f32_32x32x2_f32", "v_mfma_f32_16x16x4_f32", "v_mfma_f32_32x32x1_2b_f32",
                                  "v_mfma_f32_16x16x1_4b_f32", "v_mfma_f32_4x4x1_16b_f32", "v_mfma_f64_16x16x4_f64",
                                  "v_mfma_f64_4x4x4_4b_f64", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_16x16x32_bf16"};

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;

template <int M>
__device__ __forceinline__ float mfma_loop(int iters, float x, float y) {
  float r = 0;
  if constexpr (M == M_32x32x2) {
    f16v c0 = {0}, c1 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, c1, 0, 0, 0);
      }
    }
    for (int i = 0; i < 16; ++i) r += c0[i] + c1[i];
  } else if constexpr (M == M_16x16x4) {
    f4 c[4] = {{0}, {0}, {0}, {0}};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c[j], 0, 0, 0);
    }
    for (int j = 0; j < 4; ++j) r += c[j][0] + c[j][1] + c[j][2] + c[j][3];
  } else if constexpr (M == M_32x32x1_2B) {
    f32v c0 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) c0 = __builtin_amdgcn_mfma_f32_32x32x1f32(x, y, c0, 0, 0, 0);
    }
    for (int i = 0; i < 32; ++i) r += c0[i];
  } else if constexpr (M == M_16x16x1_4B) {
    f16v c0 = {0}, c1 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x1f32(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x1f32(y, x, c1, 0, 0, 0);
      }
    }
    for (int i = 0; i < 16; ++i) r += c0[i] + c1[i];
  } else if constexpr (M == M_4x4x1_16B) {
    f4 c[4] = {{0}, {0}, {0}, {0}};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, c[j], 0, 0, 0);
    }
    for (int j = 0; j < 4; ++j) r += c[j][0] + c[j][1] + c[j][2] + c[j][3];
  } else if constexpr (M == M_F64_16x16x4) {
    d4 c[4] = {{0}, {0}, {0}, {0}};
    const double xd = x, yd = y;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(xd, yd, c[j], 0, 0, 0);
    }
    for (int j = 0; j < 4; ++j) r += (float)(c[j][0] + c[j][1] + c[j][2] + c[j][3]);
  } else if constexpr (M == M_F64_4x4x4) {
    double c[8] = {0};
    const double xd = x, yd = y;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(xd, yd, c[j], 0, 0, 0);
    }
    for (int j = 0; j < 8; ++j) r += (float)c[j];
  } else if constexpr (M == M_BF16_32x32x16) {
    f16v c0 = {0}, c1 = {0};
    bf16x8 av, bv;
    for (int i = 0; i < 8; ++i) {
      av[i] = (__bf16)(x + i);
      bv[i] = (__bf16)(y - i);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv, av, c1, 0, 0, 0);
      }
    }
    for (int i = 0; i < 16; ++i) r += c0[i] + c1[i];
  } else if constexpr (M == M_BF16_16x16x32) {
    f4 c[4] = {{0}, {0}, {0}, {0}};
    bf16x8 av, bv;
    for (int i = 0; i < 8; ++i) {
      av[i] = (__bf16)(x + i);
      bv[i] = (__bf16)(y - i);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c[j], 0, 0, 0);
    }
    for (int j = 0; j < 4; ++j) r += c[j][0] + c[j][1] + c[j][2] + c[j][3];
  }
  return r;
}

template <int M>
__global__ void mfma_kernel(float* out, int iters, unsigned long long* cyc) {
  extern __shared__ char smem[];
  const float x = 1.0f + threadIdx.x * 1e-3f, y = 0.5f - threadIdx.x * 1e-3f;
  __syncthreads();
  const unsigned long long t0 = memtime();
  const float r = mfma_loop<M>(iters, x, y);
  const unsigned long long t1 = memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int M>
static void run_mfma(Bufs& B) {
  const int iters = 2000;
  for (int w = 1; w <= 2; ++w) {
    const int threads = 256 * w;
    hipLaunchKernelGGL(mfma_kernel<M>, dim3(g_cus), dim3(threads), 96 * 1024, 0, B.out, 10, B.cyc);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(mfma_kernel<M>, dim3(g_cus), dim3(threads), 96 * 1024, 0, B.out, iters, B.cyc);
    CK(hipDeviceSynchronize());
    const int nw = g_cus * threads / 64;
    B.h.resize(nw);
    CK(hipMemcpy(B.h.data(), B.cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const double c = median(B.h);
    printf("{\"test\":\"mfma\",\"op\":\"%s\",\"waves_per_simd\":%d,\"cycles_per_instr_per_wave\":%.2f,"
           "\"simd_cycles_per_instr\":%.2f}\n",
           mfma_name[M], w, c / (iters * 8.0), c / (iters * 8.0) / w);
  }
}

// ------------------------------------------------------------------------------------------------
// co-execution: waves 0-3 run MFMAs, waves 4-7 (same SIMDs) run VALU
// ------------------------------------------------------------------------------------------------
template <int M, int OP>
__global__ void coexec_kernel(float* out, int iters_m, int iters_v, int mode, unsigned long long* cyc) {
  extern __shared__ char smem[];
  const int wave = threadIdx.x / 64;
  float r = 0;
  __syncthreads();
  const unsigned long long t0 = memtime();
  if (wave < 4) {
    if (mode & 1) r = mfma_loop<M>(iters_m, 1.0f + threadIdx.x * 1e-3f, 0.5f);
  } else {
    if (mode & 2) {
      float a[16];
      f2 p[16];
      double q[8];
      for (int i = 0; i < 16; ++i) {
        a[i] = 1.0f + threadIdx.x * 1e-6f + i;
        p[i] = f2{a[i], a[i] + 0.5f};
      }
      for (int i = 0; i < 8; ++i) q[i] = a[i];
      for (int it = 0; it < iters_v; ++it) {
        valu_body<OP>(a, p, q, 0.999f, f2{0.999f, 0.998f}, 0.999);
        valu_body<OP>(a, p, q, 0.999f, f2{0.999f, 0.998f}, 0.999);
        valu_body<OP>(a, p, q, 0.999f, f2{0.999f, 0.998f}, 0.999);
        valu_body<OP>(a, p, q, 0.999f, f2{0.999f, 0.998f}, 0.999);
      }
      for (int i = 0; i < 16; ++i) r += a[i] + p[i].x;
    }
  }
  const unsigned long long t1 = memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int M, int OP>
static void run_coexec(Bufs& B) {
  // iteration counts chosen so both halves run about equally long alone
  const int iters_m = 2000;                       // x 8 MFMA
  for (int mode = 1; mode <= 3; ++mode) {
    const int iters_v = 4000;                     // x 64 VALU
    hipLaunchKernelGGL((coexec_kernel<M, OP>), dim3(g_cus), dim3(512), 96 * 1024, 0, B.out, iters_m, iters_v, mode, B.cyc);
    CK(hipDeviceSynchronize());
    B.h.resize(g_cus * 8);
    CK(hipMemcpy(B.h.data(), B.cyc, g_cus * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<unsigned long long> m, v;
    for (int b = 0; b < g_cus; ++b)
      for (int w = 0; w < 8; ++w) (w < 4 ? m : v).push_back(B.h[b * 8 + w]);
    printf("{\"test\":\"coexec\",\"mfma\":\"%s\",\"valu\":\"%s\",\"mode\":\"%s\",\"mfma_wave_cycles_per_mfma\":%.2f,"
           "\"valu_wave_cycles_per_instr\":%.3f}\n",
           mfma_name[M], op_name[OP], mode == 1 ? "mfma alone" : mode == 2 ? "valu alone" : "both",
           median(m) / (iters_m * 8.0), median(v) / (iters_v * 64.0));
  }
}

// one wave: per iteration 1 MFMA 32x32x2 + NV independent v_fma_f32
template <int NV, int WAVES>
__global__ void mix_kernel(float* out, int iters, unsigned long long* cyc) {
  extern __shared__ char smem[];
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = 1.0f + threadIdx.x * 1e-6f + i;
  f16v c0 = {0}, c1 = {0};
  const float x = 1.0f + threadIdx.x * 1e-3f, y = 0.5f;
  __syncthreads();
  const unsigned long long t0 = memtime();
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i % 16]) : "v"(y));
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, c1, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i % 16]) : "v"(y));
  }
  const unsigned long long t1 = memtime();
  float r = 0;
  for (int i = 0; i < 16; ++i) r += a[i] + c0[i] + c1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NV, int WAVES>
static void run_mix(Bufs& B) {
  const int iters = 2000;
  hipLaunchKernelGGL((mix_kernel<NV, WAVES>), dim3(g_cus), dim3(256 * WAVES), 96 * 1024, 0, B.out, iters, B.cyc);
  CK(hipDeviceSynchronize());
  const int nw = g_cus * 4 * WAVES;
  B.h.resize(nw);
  CK(hipMemcpy(B.h.data(), B.cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  const double c = median(B.h);
  printf("{\"test\":\"mix\",\"valu_per_mfma\":%d,\"waves_per_simd\":%d,\"wave_cycles_per_mfma\":%.2f,"
         "\"simd_cycles_per_mfma\":%.2f}\n",
         NV, WAVES, c / (iters * 2.0), c / (iters * 2.0) / WAVES);
}

// ------------------------------------------------------------------------------------------------
// LDS
// ------------------------------------------------------------------------------------------------
enum { L_R128_BCAST, L_R128_ROWS, L_W32, L_W128, L_R32, L_R64_BCAST, L_COUNT };
static const char* lds_name[] = {"ds_read_b128 broadcast", "ds_read_b128 rows (176-B stride)", "ds_write_b32",
                                 "ds_write_b128", "ds_read_b32 linear", "ds_read_b64 broadcast"};

template <int L>
__global__ void lds_kernel(float* out, int iters, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x / 64, lane = threadIdx.x & 63;
  char* base = smem + wave * 12288;
  for (int i = lane; i < 12288 / 4; i += 64) reinterpret_cast<float*>(base)[i] = i;
  __syncthreads();
  f4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if constexpr (L == L_R128_BCAST) {
        // different uniform address per u, all lanes equal
        acc += *reinterpret_cast<volatile f4*>(base + u * 16);
      } else if constexpr (L == L_R128_ROWS) {
        acc += *reinterpret_cast<volatile f4*>(base + lane * 176 + (u % 11) * 16);
      } else if constexpr (L == L_W32) {
        *reinterpret_cast<volatile float*>(base + u * 256 + lane * 4) = acc.x;
      } else if constexpr (L == L_W128) {
        *reinterpret_cast<volatile f4*>(base + (u % 11) * 1024 + lane * 16) = acc;
      } else if constexpr (L == L_R32) {
        acc.x += *reinterpret_cast<volatile float*>(base + u * 256 + lane * 4);
      } else if constexpr (L == L_R64_BCAST) {
        f2 v = *reinterpret_cast<volatile f2*>(base + u * 8);
        acc.x += v.x;
        acc.y += v.y;
      }
    }
  }
  const unsigned long long t1 = memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

template <int L>
static void run_lds(Bufs& B) {
  const int iters = 1000;
  for (int waves : {4, 8, 12}) {
    hipLaunchKernelGGL(lds_kernel<L>, dim3(g_cus), dim3(64 * waves), 12 * 12288 + 4096, 0, B.out, iters, B.cyc);
    CK(hipDeviceSynchronize());
    const int nw = g_cus * waves;
    B.h.resize(nw);
    CK(hipMemcpy(B.h.data(), B.cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const double c = median(B.h);
    const int per_iter = 16;
    printf("{\"test\":\"lds\",\"op\":\"%s\",\"waves_per_cu\":%d,\"wave_cycles_per_instr\":%.2f,\"cu_cycles_per_instr\":%.2f}\n",
           lds_name[L], waves, c / (iters * (double)per_iter), c / (iters * (double)per_iter) / waves);
  }
}


// ------------------------------------------------------------------------------------------------
// mix2: MFMA with AGPR accumulators (inline asm), VALU on unrelated registers
// ------------------------------------------------------------------------------------------------
template <int NV, int KIND>   // KIND 0: 32x32x2 acc in AGPR, 1: 16x16x4 acc in AGPR, 2: 32x32x2 acc in VGPR (asm)
__global__ void mix2_kernel(float* out, int iters, unsigned long long* cyc) {
  extern __shared__ char smem[];
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = 1.0f + threadIdx.x * 1e-6f + i;
  float x = 1.0f + threadIdx.x * 1e-3f, y = 0.5f, z = 0.25f;
  __syncthreads();
  const unsigned long long t0 = memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0) asm volatile("v_mfma_f32_32x32x2_f32 a[0:15], %0, %1, a[0:15]" ::"v"(x), "v"(y) : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15");
    if constexpr (KIND == 1) asm volatile("v_mfma_f32_16x16x4_f32 a[0:3], %0, %1, a[0:3]" ::"v"(x), "v"(y) : "a0","a1","a2","a3");
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i % 16]) : "v"(z));
    if constexpr (KIND == 0) asm volatile("v_mfma_f32_32x32x2_f32 a[16:31], %0, %1, a[16:31]" ::"v"(y), "v"(x) : "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31");
    if constexpr (KIND == 1) asm volatile("v_mfma_f32_16x16x4_f32 a[4:7], %0, %1, a[4:7]" ::"v"(y), "v"(x) : "a4","a5","a6","a7");
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i % 16]) : "v"(z));
  }
  const unsigned long long t1 = memtime();
  float r = 0;
  for (int i = 0; i < 16; ++i) r += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NV, int KIND>
static void run_mix2(Bufs& B) {
  const int iters = 2000;
  for (int waves = 1; waves <= 3; ++waves) {
    hipLaunchKernelGGL((mix2_kernel<NV, KIND>), dim3(g_cus), dim3(256 * waves), 96 * 1024, 0, B.out, iters, B.cyc);
    CK(hipDeviceSynchronize());
    const int nw = g_cus * 4 * waves;
    B.h.resize(nw);
    CK(hipMemcpy(B.h.data(), B.cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const double c = median(B.h);
    printf("{\"test\":\"mix2\",\"mfma\":\"%s\",\"valu_per_mfma\":%d,\"waves_per_simd\":%d,\"wave_cycles_per_mfma\":%.2f,"
           "\"simd_cycles_per_mfma\":%.2f}\n",
           KIND == 0 ? "32x32x2 agpr" : "16x16x4 agpr", NV, waves, c / (iters * 2.0), c / (iters * 2.0) / waves);
  }
}

// ------------------------------------------------------------------------------------------------
// coexec3: per SIMD one MFMA wave (waves 0-3) + NVW VALU waves (pk_fma); VALU throughput with / without
// ------------------------------------------------------------------------------------------------
template <int M, int OP>
__global__ void coexec3_kernel(float* out, int iters_m, int iters_v, int mode, unsigned long long* cyc) {
  extern __shared__ char smem[];
  const int wave = threadIdx.x / 64;
  float r = 0;
  __syncthreads();
  const unsigned long long t0 = memtime();
  if (wave < 4) {
    if (mode & 1) r = mfma_loop<M>(iters_m, 1.0f + threadIdx.x * 1e-3f, 0.5f);
  } else {
    float a[16];
    f2 p[16];
    double q[8];
    for (int i = 0; i < 16; ++i) {
      a[i] = 1.0f + threadIdx.x * 1e-6f + i;
      p[i] = f2{a[i], a[i] + 0.5f};
    }
    for (int i = 0; i < 8; ++i) q[i] = a[i];
    for (int it = 0; it < iters_v; ++it) {
      valu_body<OP>(a, p, q, 0.999f, f2{0.999f, 0.998f}, 0.999);
      valu_body<OP>(a, p, q, 0.999f, f2{0.999f, 0.998f}, 0.999);
      valu_body<OP>(a, p, q, 0.999f, f2{0.999f, 0.998f}, 0.999);
      valu_body<OP>(a, p, q, 0.999f, f2{0.999f, 0.998f}, 0.999);
    }
    for (int i = 0; i < 16; ++i) r += a[i] + p[i].x;
  }
  const unsigned long long t1 = memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int M, int OP>
static void run_coexec3(Bufs& B) {
  for (int vw = 1; vw <= 3; ++vw) {       // VALU waves per SIMD
    for (int mode = 0; mode <= 1; ++mode) {
      const int iters_v = 2000;
      const int iters_m = (M == M_16x16x4 || M == M_BF16_32x32x16 ? 2000 * 2 * vw : M == M_BF16_16x16x32 ? 2000 * 4 * vw : 2000 * vw);  // MFMA wave outlasts the VALU waves
      const int waves = 4 + 4 * vw;
      hipLaunchKernelGGL((coexec3_kernel<M, OP>), dim3(g_cus), dim3(64 * waves), 96 * 1024, 0, B.out, iters_m, iters_v, mode, B.cyc);
      CK(hipDeviceSynchronize());
      B.h.resize(g_cus * 16);
      CK(hipMemcpy(B.h.data(), B.cyc, g_cus * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      std::vector<unsigned long long> m, v;
      for (int b = 0; b < g_cus; ++b)
        for (int w = 0; w < waves; ++w) (w < 4 ? m : v).push_back(B.h[b * 16 + w]);
      printf("{\"test\":\"coexec3\",\"mfma\":\"%s\",\"valu\":\"%s\",\"valu_waves_per_simd\":%d,\"mfma_running\":%d,"
             "\"mfma_wave_cycles_per_mfma\":%.2f,\"valu_simd_cycles_per_instr\":%.3f}\n",
             mfma_name[M], op_name[OP], vw, mode, median(m) / (iters_m * 8.0), median(v) / (iters_v * 64.0) / vw);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LDS throughput (reads issued back to back into independent registers, one wait per 16)
// ------------------------------------------------------------------------------------------------
enum { T_R128_BCAST, T_R128_ROWS176, T_R128_ROWS192, T_R64_ROWS176, T_R32, T_W32, T_W128, T_BPERM, T_SWAP32, T_CNDMASK_S, T_DPP, T_READLANE };
static const char* lds2_name[] = {"ds_read_b128 broadcast", "ds_read_b128 rows stride 176", "ds_read_b128 rows stride 192",
                                  "ds_read_b64 rows stride 176", "ds_read_b32 linear", "ds_write_b32 linear", "ds_write_b128 linear",
                                  "ds_bpermute_b32", "v_permlane32_swap_b32", "v_cndmask_b32 (sgpr mask)", "v_mov_b32 dpp row_shr",
                                  "v_readlane_b32"};

template <int L>
__global__ void lds2_kernel(float* out, int iters, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x / 64, lane = threadIdx.x & 63;
  char* base = smem + wave * 12288;
  for (int i = lane; i < 12288 / 4; i += 64) reinterpret_cast<float*>(base)[i] = i;
  __syncthreads();
  const unsigned b0 = (unsigned)(size_t)(base - smem);
  f4 v[8];
  f2 w2[8];
  float s[16];
  for (int i = 0; i < 8; ++i) { v[i] = f4{0, 0, 0, 0}; w2[i] = f2{0, 0}; }
  for (int i = 0; i < 16; ++i) s[i] = lane + i;
  const unsigned long long mask = 0x00000001000000ffull << (iters & 3);
  unsigned a_rows176 = b0 + (lane & 31) * 176 + (lane >> 5) * 80;
  unsigned a_rows192 = b0 + (lane & 31) * 192 + (lane >> 5) * 80;
  unsigned a_lin4 = b0 + lane * 4, a_lin16 = b0 + lane * 16, a_b = b0;
  const unsigned long long t0 = memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if constexpr (L == T_R128_BCAST) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[u % 8]) : "v"(a_b), "n"(u * 16));
      if constexpr (L == T_R128_ROWS176) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[u % 8]) : "v"(a_rows176), "n"((u % 5) * 16));
      if constexpr (L == T_R128_ROWS192) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[u % 8]) : "v"(a_rows192), "n"((u % 5) * 16));
      if constexpr (L == T_R64_ROWS176) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(w2[u % 8]) : "v"(a_rows176), "n"((u % 10) * 8));
      if constexpr (L == T_R32) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(s[u]) : "v"(a_lin4), "n"(u * 256));
      if constexpr (L == T_W32) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(a_lin4), "v"(s[u]), "n"(u * 256));
      if constexpr (L == T_W128) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a_lin16), "v"(v[u % 8]), "n"((u % 8) * 1024));
      if constexpr (L == T_BPERM) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(s[u]) : "v"(a_lin4 ^ 64u), "v"(s[(u + 1) % 16]));
      if constexpr (L == T_SWAP32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(s[u]), "+v"(s[(u + 8) % 16]));
      if constexpr (L == T_CNDMASK_S) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(s[u]) : "v"(s[u]), "v"(s[(u + 1) % 16]), "s"(mask));
      if constexpr (L == T_DPP) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(s[u]) : "v"(s[(u + 1) % 16]));
      if constexpr (L == T_READLANE) {
        int sg;
        asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(sg) : "v"(s[u]));
        asm volatile("" ::"s"(sg));
      }
    }
    if constexpr (L <= T_BPERM) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  const unsigned long long t1 = memtime();
  float r = 0;
  for (int i = 0; i < 8; ++i) r += v[i].x + v[i].w + w2[i].x;
  for (int i = 0; i < 16; ++i) r += s[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

template <int L>
static void run_lds2(Bufs& B) {
  const int iters = 1000;
  for (int waves : {4, 8, 12}) {
    hipLaunchKernelGGL(lds2_kernel<L>, dim3(g_cus), dim3(64 * waves), 12 * 12288 + 4096, 0, B.out, iters, B.cyc);
    CK(hipDeviceSynchronize());
    const int nw = g_cus * waves;
    B.h.resize(nw);
    CK(hipMemcpy(B.h.data(), B.cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const double c = median(B.h);
    printf("{\"test\":\"lds2\",\"op\":\"%s\",\"waves_per_cu\":%d,\"wave_cycles_per_instr\":%.2f,\"cu_cycles_per_instr\":%.2f,"
           "\"simd_cycles_per_instr\":%.2f}\n",
           lds2_name[L], waves, c / (iters * 16.0), c / (iters * 16.0) / waves, c / (iters * 16.0) / (waves / 4.0));
  }
}

// ------------------------------------------------------------------------------------------------
// gather floor for candidate table layouts
// ------------------------------------------------------------------------------------------------
// A wave (one workgroup, LDS sized for `per_cu` resident workgroups) gathers 62 rows per task
// (two neighbourhoods of 30 neighbours + query) with direct-to-LDS 16-byte loads, SPR slots per
// row; layouts with a separate target table also read one 4-byte target per neighbour.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int SPR, int DSLOTS, bool SEP_TARGET>
__global__ __launch_bounds__(64) void gather_kernel(const char* table, int stride, const float* targets,
                                                    const int64_t* idx, int64_t ntasks, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const char** rowaddr = reinterpret_cast<const char**>(smem + SPR * 1024);
  const int lane = threadIdx.x;
  float ysum = 0;
  int64_t task = blockIdx.x;
  int64_t nidx = task < ntasks ? idx[task * 64 + lane] : 0;
  for (; task < ntasks; task += gridDim.x) {
    const int64_t myidx = nidx;
    if (task + gridDim.x < ntasks) nidx = idx[(task + gridDim.x) * 64 + lane];
    __syncthreads();
    rowaddr[lane] = table + myidx * stride;
    __syncthreads();
#pragma unroll
    for (int n = 0; n < SPR; ++n) {
      const unsigned sigma = 64u * n + lane;
      const unsigned row = sigma / SPR;
      const unsigned c = min(sigma - row * SPR, (unsigned)(DSLOTS - 1));
      glds16(rowaddr[row] + c * 16, smem + n * 1024);
    }
    if (SEP_TARGET) {
      if ((lane & 31) < 30) ysum += targets[myidx];
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SPR + 2));  // previous task's loads have landed
  }
  asm volatile("s_waitcnt vmcnt(0)");
  __syncthreads();
  out[blockIdx.x * 64 + lane] = ysum + reinterpret_cast<float*>(smem)[lane];
}

// Column-major variant: lane = row, load n takes 16-byte column n of all 64 rows (no row / column
// arithmetic, no pointer table in LDS; every lane of a load touches a different cache line).
template <int SPR>
__global__ __launch_bounds__(64) void gather_colmajor_kernel(const char* table, int stride, const float* targets,
                                                             const int64_t* idx, int64_t ntasks, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  int64_t task = blockIdx.x;
  int64_t nidx = task < ntasks ? idx[task * 64 + lane] : 0;
  for (; task < ntasks; task += gridDim.x) {
    const char* row = table + nidx * stride;
    if (task + gridDim.x < ntasks) nidx = idx[(task + gridDim.x) * 64 + lane];
    __syncthreads();
#pragma unroll
    for (int n = 0; n < SPR; ++n) glds16(row + n * 16, smem + n * 1024);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SPR + 1));
  }
  asm volatile("s_waitcnt vmcnt(0)");
  __syncthreads();
  out[blockIdx.x * 64 + lane] = reinterpret_cast<float*>(smem)[lane];
}

template <int SPR, int DSLOTS, bool SEP>
static void run_gather(const char* name, int stride, int64_t nrows, int64_t ntasks, int per_cu, const int64_t* didx) {
  char* table;
  float* targets;
  float* out;
  CK(hipMalloc(&table, (size_t)nrows * stride + 256));
  CK(hipMemset(table, 0, (size_t)nrows * stride + 256));
  CK(hipMalloc(&targets, nrows * 4));
  CK(hipMemset(targets, 0, nrows * 4));
  CK(hipMalloc(&out, (size_t)g_cus * per_cu * 64 * 4));
  const size_t lds = SPR * 1024 + 64 * 8;
  const int grid = g_cus * per_cu;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e9f, sum = 0;
  const int reps = 5;
  for (int r = 0; r <= reps; ++r) {
    CK(hipEventRecord(e0));
    if (DSLOTS == 0)
      hipLaunchKernelGGL((gather_colmajor_kernel<SPR>), dim3(grid), dim3(64), lds, 0, table, stride, targets, didx, ntasks, out);
    else
      hipLaunchKernelGGL((gather_kernel<SPR, (DSLOTS > 0 ? DSLOTS : 1), SEP>), dim3(grid), dim3(64), lds, 0, table, stride, targets, didx, ntasks, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0) {
      best = std::min(best, ms);
      sum += ms;
    }
  }
  const double rows = (double)ntasks * 62;  // rows actually used per task (2 pad slots re-read row 0.. whatever)
  printf("{\"test\":\"gather\",\"layout\":\"%s\",\"stride\":%d,\"slots_per_row\":%d,\"rows_in_table\":%lld,\"per_cu\":%d,"
         "\"ms_per_1M_nbhd\":%.3f,\"best_ms\":%.3f,\"payload_TBps\":%.3f}\n",
         name, stride, SPR, (long long)nrows, per_cu, sum / reps * (1e6 / (ntasks * 2.0)), best * (1e6 / (ntasks * 2.0)),
         rows * 164.0 / (sum / reps * 1e-3) / 1e12);
  CK(hipFree(table));
  CK(hipFree(targets));
  CK(hipFree(out));
}

static int64_t* make_indices(int64_t nrows, int64_t ntasks) {
  std::vector<int64_t> h((size_t)ntasks * 64);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for (size_t i = 0; i < h.size(); ++i) {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    h[i] = (int64_t)(s % (uint64_t)nrows);
  }
  int64_t* d;
  CK(hipMalloc(&d, h.size() * 8));
  CK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  g_cus = prop.multiProcessorCount;
  fprintf(stderr, "device: %s, %d CUs, clock %d kHz\n", prop.name, g_cus, prop.clockRate);
  auto want = [&](const char* t) {
    if (argc < 2) return true;
    for (int i = 1; i < argc; ++i)
      if (!strcmp(argv[i], t)) return true;
    return false;
  };
  Bufs B;
  CK(hipMalloc(&B.out, (size_t)g_cus * 1024 * 4));
  CK(hipMalloc(&B.cyc, (size_t)g_cus * 32 * 8));
  if (want("valu")) {
    run_valu<OP_FMA>(B);
    run_valu<OP_ADD>(B);
    run_valu<OP_MUL>(B);
    run_valu<OP_PKFMA>(B);
    run_valu<OP_PKADD>(B);
    run_valu<OP_PKMUL>(B);
    run_valu<OP_EXP>(B);
    run_valu<OP_SQRT>(B);
    run_valu<OP_RCP>(B);
    run_valu<OP_FMA64>(B);
  }
  if (want("mfma")) {
    run_mfma<M_32x32x2>(B);
    run_mfma<M_16x16x4>(B);
    run_mfma<M_32x32x1_2B>(B);
    run_mfma<M_16x16x1_4B>(B);
    run_mfma<M_4x4x1_16B>(B);
    run_mfma<M_F64_16x16x4>(B);
    run_mfma<M_F64_4x4x4>(B);
    run_mfma<M_BF16_32x32x16>(B);
    run_mfma<M_BF16_16x16x32>(B);
  }
  if (want("coexec")) {
    run_coexec<M_32x32x2, OP_FMA>(B);
    run_coexec<M_32x32x2, OP_PKFMA>(B);
    run_coexec<M_16x16x4, OP_FMA>(B);
    run_coexec<M_32x32x1_2B, OP_FMA>(B);
    run_coexec<M_F64_16x16x4, OP_FMA64>(B);
  }
  if (want("mix")) {
    run_mix<0, 1>(B);
    run_mix<4, 1>(B);
    run_mix<8, 1>(B);
    run_mix<12, 1>(B);
    run_mix<16, 1>(B);
    run_mix<24, 1>(B);
    run_mix<32, 1>(B);
    run_mix<8, 2>(B);
    run_mix<16, 2>(B);
    run_mix<24, 2>(B);
    run_mix<32, 2>(B);
    run_mix<16, 3>(B);
    run_mix<32, 3>(B);
  }
  if (want("lds")) {
    run_lds<L_R128_BCAST>(B);
    run_lds<L_R128_ROWS>(B);
    run_lds<L_R64_BCAST>(B);
    run_lds<L_R32>(B);
    run_lds<L_W32>(B);
    run_lds<L_W128>(B);
  }
  if (want("mix2")) {
    run_mix2<0, 0>(B);
    run_mix2<4, 0>(B);
    run_mix2<8, 0>(B);
    run_mix2<16, 0>(B);
    run_mix2<0, 1>(B);
    run_mix2<2, 1>(B);
    run_mix2<4, 1>(B);
    run_mix2<8, 1>(B);
  }
  if (want("coexec3")) {
    run_coexec3<M_32x32x2, OP_PKFMA>(B);
    run_coexec3<M_16x16x4, OP_PKFMA>(B);
    run_coexec3<M_32x32x2, OP_ADD>(B);
    run_coexec3<M_BF16_32x32x16, OP_PKFMA>(B);
    run_coexec3<M_BF16_32x32x16, OP_ADD>(B);
    run_coexec3<M_BF16_16x16x32, OP_PKFMA>(B);
    run_coexec3<M_BF16_16x16x32, OP_ADD>(B);
    run_coexec3<M_BF16_32x32x16, OP_EXP>(B);
    run_coexec3<M_32x32x2, OP_EXP>(B);
  }
  if (want("lds2")) {
    run_lds2<T_R128_BCAST>(B);
    run_lds2<T_R128_ROWS176>(B);
    run_lds2<T_R128_ROWS192>(B);
    run_lds2<T_R64_ROWS176>(B);
    run_lds2<T_R32>(B);
    run_lds2<T_W32>(B);
    run_lds2<T_W128>(B);
    run_lds2<T_BPERM>(B);
    run_lds2<T_SWAP32>(B);
    run_lds2<T_CNDMASK_S>(B);
    run_lds2<T_DPP>(B);
    run_lds2<T_READLANE>(B);
  }
  if (want("gather")) {
    for (int64_t nrows : {(int64_t)1000000, (int64_t)8000000}) {
      const int64_t ntasks = 500000;
      int64_t* didx = make_indices(nrows, ntasks);
      for (int per_cu : {12, 14}) {
        run_gather<11, 10, true>("rows160+targets", 160, nrows, ntasks, per_cu, didx);   // 10 data slots + 1 re-read
        run_gather<11, 11, false>("packed176", 176, nrows, ntasks, per_cu, didx);
        run_gather<11, 11, false>("packed192", 192, nrows, ntasks, per_cu, didx);
        run_gather<11, 11, false>("packed256", 256, nrows, ntasks, per_cu, didx);
        run_gather<12, 12, false>("packed192x12", 192, nrows, ntasks, per_cu, didx);
        run_gather<11, 0, false>("packed192-colmajor", 192, nrows, ntasks, per_cu, didx);
      }
      CK(hipFree(didx));
    }
  }
  return 0;
}

// Every instantiation of fused_wave_kernel the library is built with:
//   X(T, NP, KFIX, RFIX, DFIX, PIPED, COEFF, PACKED, GRAM, GEN64)
// (what launch_np / launch_fused_wave of mgp_fused_wave.hip can ask for; GRAM: fp32 pipelined kernels only while
// MGP_GRAM64 is 0; GEN64: the fp64 32-slot run-time-shape kernels), built by mgp_fused_wave_inst_{f32,f64}.hip.
#pragma once
#define MGP_WAVE_LIST_F32(X) \
  X(float, 32, 0, 0, 0, false, false, false, false, false) \
  X(float, 32, 0, 0, 0, false, true, false, false, false) \
  X(float, 32, 0, 0, 0, true, false, false, false, false) \
  X(float, 32, 0, 0, 0, true, false, false, true, false) \
  X(float, 32, 0, 0, 0, true, false, true, false, false) \
  X(float, 32, 0, 0, 0, true, false, true, true, false) \
  X(float, 32, 30, 1, 40, true, false, false, false, false) \
  X(float, 32, 30, 1, 40, true, false, false, true, false) \
  X(float, 32, 30, 1, 40, true, false, true, false, false) \
  X(float, 32, 30, 1, 40, true, false, true, true, false) \
  X(float, 64, 0, 0, 0, false, false, false, false, false) \
  X(float, 64, 0, 0, 0, false, true, false, false, false) \
  X(float, 64, 0, 0, 0, true, false, false, false, false) \
  X(float, 64, 0, 0, 0, true, false, false, true, false) \
  X(float, 64, 0, 0, 0, true, false, true, false, false) \
  X(float, 64, 0, 0, 0, true, false, true, true, false) \
  X(float, 64, 50, 1, 8, true, false, false, false, false) \
  X(float, 64, 50, 1, 8, true, false, false, true, false) \
  X(float, 64, 50, 1, 8, true, false, true, false, false) \
  X(float, 64, 50, 1, 8, true, false, true, true, false)
#define MGP_WAVE_LIST_F64(X) \
  X(double, 32, 0, 0, 0, false, false, false, false, false) \
  X(double, 32, 0, 0, 0, false, true, false, false, false) \
  X(double, 32, 0, 0, 0, true, false, false, false, false) \
  X(double, 32, 0, 0, 0, true, false, true, false, false) \
  X(double, 32, 0, 0, 0, false, false, false, false, true) \
  X(double, 32, 0, 0, 0, true, false, false, false, true) \
  X(double, 32, 0, 0, 0, true, false, true, false, true) \
  X(double, 32, 30, 1, 40, true, false, false, false, false) \
  X(double, 32, 30, 1, 40, true, false, true, false, false) \
  X(double, 64, 0, 0, 0, false, false, false, false, false) \
  X(double, 64, 0, 0, 0, false, true, false, false, false) \
  X(double, 64, 0, 0, 0, true, false, false, false, false) \
  X(double, 64, 0, 0, 0, true, false, true, false, false) \
  X(double, 64, 50, 1, 8, true, false, false, false, false) \
  X(double, 64, 50, 1, 8, true, false, true, false, false)

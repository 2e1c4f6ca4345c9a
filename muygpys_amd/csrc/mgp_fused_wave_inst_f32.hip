// Explicit instantiations of the wave kernel's launchers, fp32 (mgp_fused_wave_list.h; a translation unit per
// element type so that the two groups compile in parallel).
#include "mgp_fused_wave_launch.h"
#include "mgp_fused_wave_list.h"

namespace mgp {
#define MGP_X(T, NP, K, R, D, PIPED, COEFF, PACKED, GRAM, GEN64) \
  template int launch_np_impl<T, NP, K, R, D, PIPED, COEFF, PACKED, GRAM, GEN64>(const FusedArgs&, hipStream_t);
MGP_WAVE_LIST_F32(MGP_X)
}  // namespace mgp

#if MGP_WAVE_TIMING
// (experiments only, tools/wave_timing.py: the phase counters of THIS translation unit's kernels -- a __device__
// variable defined in a header is one variable per translation unit)
extern "C" int mgp_debug_wave_timing_f32(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mgp::g_wave_timing), sizeof(mgp::g_wave_timing)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mgp::g_wave_timing), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

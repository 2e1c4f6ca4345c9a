// Explicit instantiations of the wave kernel's launchers, fp32 (mgp_fused_wave_list.h; a translation unit per
// element type so that the two groups compile in parallel).
#include "mgp_fused_wave_launch.h"
#include "mgp_fused_wave_list.h"

namespace mgp {
#define MGP_X(T, NP, K, R, D, PIPED, COEFF, PACKED, GRAM, GEN64) \
  template int launch_np_impl<T, NP, K, R, D, PIPED, COEFF, PACKED, GRAM, GEN64>(const FusedArgs&, hipStream_t);
MGP_WAVE_LIST_F32(MGP_X)
}  // namespace mgp

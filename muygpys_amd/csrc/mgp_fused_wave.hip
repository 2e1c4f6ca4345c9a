// placeholder until the register-resident kernels land
#include "mgp_args.h"
namespace mgp {
template <typename T> int launch_fused_wave(const FusedArgs&, hipStream_t) { return MGP_EUNSUPPORTED; }
template int launch_fused_wave<float>(const FusedArgs&, hipStream_t);
template int launch_fused_wave<double>(const FusedArgs&, hipStream_t);
}

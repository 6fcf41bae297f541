// Kernel instantiation: Connect-4, 128 channels, 6 boards per workgroup, 2 x 4 waves, statically scheduled tower
// (BASELINE.json configs[3]'s network).
#include "azsp_resnet_kernel.h"
namespace azsp {
void fwd_c4_128_big(const FwdArgs &a) { launch_forward<Connect4, 4, 4, 0, 4, 2>(a); }
}  // namespace azsp

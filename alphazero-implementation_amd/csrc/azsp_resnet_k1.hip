// Kernel instantiations: the headline fp32 kernel (Connect-4, 64 channels, 6 boards per workgroup, 2 x 2 waves, statically
// scheduled tower) and its stamped diagnostic form.  Own translation unit: the static schedule is slow to compile.
#include "azsp_resnet_kernel.h"
namespace azsp {
void fwd_c4_64_big(const FwdArgs &a) { launch_forward<Connect4, 2, 4, 0, 2, 2, true>(a); }
}  // namespace azsp

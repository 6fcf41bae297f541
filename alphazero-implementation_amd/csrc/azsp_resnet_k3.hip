// Kernel instantiations: the other Connect-4 variants (small and mid-size batches, other channel counts).
#include "azsp_resnet_kernel.h"
namespace azsp {
void fwd_c4_32_small(const FwdArgs &a) { launch_forward<Connect4, 1, 1, 1>(a); }
void fwd_c4_32_big(const FwdArgs &a) { launch_forward<Connect4, 1, 2, 0>(a); }
void fwd_c4_64_one(const FwdArgs &a) { launch_forward<Connect4, 2, 1, 2, 2, 2, true>(a); }       // (PF = 2: the tower on three 16-row tiles)
void fwd_c4_64_mid(const FwdArgs &a) { launch_forward<Connect4, 2, 2, 3, 2, 2, true>(a); }       // (PF = 3: eight 16-row tiles, static tap skipping)
void fwd_c4_96(const FwdArgs &a) { launch_forward<Connect4, 3, 1, 1>(a); }
void fwd_c4_128_one(const FwdArgs &a) { launch_forward<Connect4, 4, 1, 2, 4, 2>(a); }
void fwd_c4_128_mid(const FwdArgs &a) { launch_forward<Connect4, 4, 2, 3, 4, 2>(a); }
void fwd_c4_192(const FwdArgs &a) { launch_forward<Connect4, 6, 2, 1, 3, 2>(a); }
void fwd_c4_256(const FwdArgs &a) { launch_forward<Connect4, 8, 2, 1, 4, 2>(a); }
}  // namespace azsp

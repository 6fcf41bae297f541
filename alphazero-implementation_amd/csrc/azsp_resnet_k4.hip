// Kernel instantiations: 9x9 Gomoku (the Game plugin of BASELINE.json configs[4]).
#include "azsp_resnet_kernel.h"
namespace azsp {
void fwd_g9_32(const FwdArgs &a) { launch_forward<Gomoku9, 1, 2, 0>(a); }
void fwd_g9_64(const FwdArgs &a) { launch_forward<Gomoku9, 2, 2, 0>(a); }
void fwd_g9_96(const FwdArgs &a) { launch_forward<Gomoku9, 3, 3, 0, 3, 1>(a); }
void fwd_g9_128(const FwdArgs &a) { launch_forward<Gomoku9, 4, 3, 0, 4, 1>(a); }
void fwd_g9_192(const FwdArgs &a) { launch_forward<Gomoku9, 6, 3, 0, 6, 1>(a); }
void fwd_g9_256(const FwdArgs &a) { launch_forward<Gomoku9, 8, 3, 0, 8, 1, true>(a); }      // + the traced form (AZSP_RESNET_TRACE)
}  // namespace azsp

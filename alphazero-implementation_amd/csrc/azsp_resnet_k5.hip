// Kernel instantiations: the one-board Connect-4 kernels with the tree step fused in (FUSED rounds, runs of up to 32
// games: the workgroup that evaluated a game's leaf expands it, backs it up and finds the game's next leaf --
// azsp_tree_dev.h advance_game; there a simulation round is launch latency and one launch per round is the default).
#include "azsp_resnet_kernel.h"
namespace azsp {
void fwd_c4_64_one_fused(const FwdArgs &a) { launch_forward<Connect4, 2, 1, 2, 2, 2, false, true>(a); }
void fwd_c4_128_one_fused(const FwdArgs &a) { launch_forward<Connect4, 4, 1, 2, 4, 2, false, true>(a); }
}  // namespace azsp

// Dispatch of the 256-thread backward block kernels to the translation unit of their input width (tu_bwd32 / 48 / 64.hip).
#define MWW_BLOCK_TU 1
#include "block_launch.hip.h"

namespace mww {

bool k_launch_bwd_block_cin32(hipStream_t st, int mode, int cout, int k, bool last, const BwdBlockArgs& a, int grid);
bool k_launch_bwd_block_cin48(hipStream_t st, int mode, int cout, int k, bool last, const BwdBlockArgs& a, int grid);
bool k_launch_bwd_block_cin64(hipStream_t st, int mode, int cout, int k, bool last, const BwdBlockArgs& a, int grid);

bool k_launch_bwd_block(hipStream_t st, int mode, int cin, int cout, int k, bool last, const BwdBlockArgs& a, int grid) {
  if (cin == 32) return k_launch_bwd_block_cin32(st, mode, cout, k, last, a, grid);
  if (cin == 48) return k_launch_bwd_block_cin48(st, mode, cout, k, last, a, grid);
  if (cin == 64) return k_launch_bwd_block_cin64(st, mode, cout, k, last, a, grid);
  return false;
}

}  // namespace mww

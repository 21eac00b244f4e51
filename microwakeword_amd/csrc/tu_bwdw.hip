// Translation unit of the wide-workgroup backward block kernels (kernels_bwdw.hip.h; see block_launch.hip.h).
#define MWW_BLOCK_TU 1
#include "block_launch.hip.h"
#include "kernels_bwdw.hip.h"

namespace mww {

template <int C, int K, int NTH>
static void launch_w(hipStream_t st, bool last, const BwdBlockArgs& a, int grid) {
  if (last) hipLaunchKernelGGL((bwd_blockw_kernel<C, C, K, true, NTH>), dim3(grid), dim3(NTH), 0, st, a);
  else hipLaunchKernelGGL((bwd_blockw_kernel<C, C, K, false, NTH>), dim3(grid), dim3(NTH), 0, st, a);
}

bool k_launch_bwd_blockw(hipStream_t st, int cin, int cout, int k, bool last, const BwdBlockArgs& a, int grid) {
  if (cin != cout) return false;
#define X(CI, CO, K)                                                                                           \
  if (cin == CI && k == K) {                                                                                   \
    if constexpr (CI == CO && (CI == 48 || CI == 64)) {                                                        \
      launch_w<CI, K, 512>(st, last, a, grid);                                                                 \
      return true;                                                                                             \
    }                                                                                                          \
  }
  MWW_BLOCK_SHAPES(X)
#undef X
  return false;
}

}  // namespace mww

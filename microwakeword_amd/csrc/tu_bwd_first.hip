// Translation unit of the 256-thread first-block backward kernels (see block_launch.hip.h).
#define MWW_BLOCK_TU 1
#include "block_launch.hip.h"

namespace mww {

bool k_launch_bwd_first(hipStream_t st, int mode, int k1, int c1, int cout, int k, int stride, const BwdFirstArgs& a, int grid) {
  if (mode != 0) {
#define X(K1, C1, CO, K, S)                                                                                    \
    if (k1 == K1 && c1 == C1 && cout == CO && k == K && stride == S) {                                         \
      if (mode == 2)                                                                                           \
        hipLaunchKernelGGL((bwd_first_kernel<K1, C1, CO, K, S, true, true>), dim3(grid), dim3(kThreads), 0, st, a); \
      else                                                                                                     \
        hipLaunchKernelGGL((bwd_first_kernel<K1, C1, CO, K, S, true>), dim3(grid), dim3(kThreads), 0, st, a);  \
      return true;                                                                                             \
    }
    MWW_FIRST_SHAPES_BF16(X)
#undef X
    return false;
  }
#define X(K1, C1, CO, K, S)                                                                                    \
  if (k1 == K1 && c1 == C1 && cout == CO && k == K && stride == S) {                                           \
    hipLaunchKernelGGL((bwd_first_kernel<K1, C1, CO, K, S, false>), dim3(grid), dim3(kThreads), 0, st, a);     \
    return true;                                                                                               \
  }
  MWW_FIRST_SHAPES(X)
#undef X
  return false;
}

}  // namespace mww

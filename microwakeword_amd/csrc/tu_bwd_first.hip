// Translation unit of the 256-thread first-block backward kernels (see block_launch.hip.h).
#define MWW_BLOCK_TU 1
#include "block_launch.hip.h"

namespace mww {

// the x6 form of the conv1 weight gradient exists for the stride-1 shapes (in a template: the other branch is not instantiated)
template <int K1, int C1, int CO, int K, int S, bool BF = false, bool SB = false>
static bool launch_bwd_first_x6(hipStream_t st, const BwdFirstArgs& a, int grid) {
  if constexpr (S == 1) {
    hipLaunchKernelGGL((bwd_first_kernel<K1, C1, CO, K, S, BF, SB, true>), dim3(grid), dim3(kThreads), 0, st, a);
    return true;
  } else {
    return false;
  }
}

bool k_launch_bwd_first(hipStream_t st, int mode, int k1, int c1, int cout, int k, int stride, const BwdFirstArgs& a, int grid, bool x6) {
  if (mode != 0) {
#define X(K1, C1, CO, K, S)                                                                                    \
    if (k1 == K1 && c1 == C1 && cout == CO && k == K && stride == S) {                                         \
      /* (the bf16 modes round the operands of the 1x1 contractions; the conv1 weight gradient stays fp32-grade: x6) */ \
      if (x6 && mode == 2 && launch_bwd_first_x6<K1, C1, CO, K, S, true, true>(st, a, grid)) return true;     \
      if (x6 && mode == 1 && launch_bwd_first_x6<K1, C1, CO, K, S, true, false>(st, a, grid)) return true;    \
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
    if (x6 && launch_bwd_first_x6<K1, C1, CO, K, S>(st, a, grid)) return true;                                 \
    hipLaunchKernelGGL((bwd_first_kernel<K1, C1, CO, K, S, false>), dim3(grid), dim3(kThreads), 0, st, a);     \
    return true;                                                                                               \
  }
  MWW_FIRST_SHAPES(X)
#undef X
  return false;
}

}  // namespace mww

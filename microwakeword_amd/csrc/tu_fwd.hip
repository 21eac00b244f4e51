// Translation unit of the forward block kernels (see block_launch.hip.h).
#define MWW_BLOCK_TU 1
#include "block_launch.hip.h"

namespace mww {

// the x6 form of the first convolution exists for the stride-1 shapes (in a template: the other branch is not instantiated)
template <int K1, int C1, int CO, int K, int S>
static bool launch_fwd_first_x6(hipStream_t st, const FwdFirstArgs& a, int grid) {
  if constexpr (S == 1) {
    hipLaunchKernelGGL((fwd_first_kernel<K1, C1, CO, K, S, false, false, true>), dim3(grid), dim3(kThreads), 0, st, a);
    return true;
  } else {
    return false;
  }
}

bool k_launch_fwd_first(hipStream_t st, int mode, int k1, int c1, int cout, int k, int stride, const FwdFirstArgs& a, int grid, bool x6) {
  if (mode != 0) {
#define X(K1, C1, CO, K, S)                                                                                    \
    if (k1 == K1 && c1 == C1 && cout == CO && k == K && stride == S) {                                         \
      if (mode == 2)                                                                                           \
        hipLaunchKernelGGL((fwd_first_kernel<K1, C1, CO, K, S, true, true>), dim3(grid), dim3(kThreads), 0, st, a); \
      else                                                                                                     \
        hipLaunchKernelGGL((fwd_first_kernel<K1, C1, CO, K, S, true>), dim3(grid), dim3(kThreads), 0, st, a);  \
      return true;                                                                                             \
    }
    MWW_FIRST_SHAPES_BF16(X)
#undef X
    return false;
  }
#define X(K1, C1, CO, K, S)                                                                                    \
  if (k1 == K1 && c1 == C1 && cout == CO && k == K && stride == S) {                                           \
    if (x6 && launch_fwd_first_x6<K1, C1, CO, K, S>(st, a, grid)) return true;                                 \
    hipLaunchKernelGGL((fwd_first_kernel<K1, C1, CO, K, S, false>), dim3(grid), dim3(kThreads), 0, st, a);     \
    return true;                                                                                               \
  }
  MWW_FIRST_SHAPES(X)
#undef X
  return false;
}

bool k_launch_fwd_block(hipStream_t st, int mode, int cin, int cout, int k, const FwdBlockArgs& a, int grid) {
  if (mode != 0) {
#define X(CI, CO, K)                                                                                           \
    if (cin == CI && cout == CO && k == K) {                                                                   \
      if (mode == 2)                                                                                           \
        hipLaunchKernelGGL((fwd_block_kernel<CI, CO, K, true, true>), dim3(grid), dim3(kThreads), 0, st, a);   \
      else                                                                                                     \
        hipLaunchKernelGGL((fwd_block_kernel<CI, CO, K, true>), dim3(grid), dim3(kThreads), 0, st, a);         \
      return true;                                                                                             \
    }
    MWW_BLOCK_SHAPES_BF16(X)
#undef X
    return false;
  }
#define X(CI, CO, K)                                                                                           \
  if (cin == CI && cout == CO && k == K) {                                                                     \
    hipLaunchKernelGGL((fwd_block_kernel<CI, CO, K, false>), dim3(grid), dim3(kThreads), 0, st, a);            \
    return true;                                                                                               \
  }
  MWW_BLOCK_SHAPES(X)
#undef X
  return false;
}

}  // namespace mww

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
template <int C, int K, int NTH>
static void launch_w_bf16(hipStream_t st, int mode, bool last, const BwdBlockArgs& a, int grid) {
  if (mode == 2) {
    if (last) hipLaunchKernelGGL((bwd_blockw_kernel<C, C, K, true, NTH, true, true>), dim3(grid), dim3(NTH), 0, st, a);
    else hipLaunchKernelGGL((bwd_blockw_kernel<C, C, K, false, NTH, true, true>), dim3(grid), dim3(NTH), 0, st, a);
  } else {
    if (last) hipLaunchKernelGGL((bwd_blockw_kernel<C, C, K, true, NTH, true>), dim3(grid), dim3(NTH), 0, st, a);
    else hipLaunchKernelGGL((bwd_blockw_kernel<C, C, K, false, NTH, true>), dim3(grid), dim3(NTH), 0, st, a);
  }
}

// square 48- and 64-wide blocks only (kernels_bwdw.hip.h: WidePitch / WideRoles); false = use the 256-thread kernel
bool k_launch_bwd_blockw(hipStream_t st, int mode, int cin, int cout, int k, bool last, const BwdBlockArgs& a, int grid) {
  if (cin != cout) return false;
  if (mode != 0) {
#define X(CI, CO, K)                                                                                           \
    if (cin == CI && k == K) {                                                                                 \
      if constexpr (CI == CO && (CI == 48 || CI == 64)) {                                                      \
        launch_w_bf16<CI, K, 512>(st, mode, last, a, grid);                                                    \
        return true;                                                                                           \
      }                                                                                                        \
    }
    MWW_BLOCK_SHAPES_BF16(X)
#undef X
    return false;
  }
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

// first block: the stride-3 first convolutions only (194+ staged x rows: one workgroup per CU whatever its size, so 512
// threads double the waves: 45.8 -> 37.5 us at the notebook topology).  The default first block was built and measured too
// (two 512-thread workgroups per CU at 128 registers): 53.0-53.3 us against 51.2-51.8 for bwd_first_kernel in the same session
// (profiles/round5_bwd_first_forms_ab.txt) - its launch is bound by the exact-fp32 MFMAs of the conv1 weight gradient, which
// more waves do not make cheaper - and the stride-1 crosses would trade two 256-thread workgroups for one of 512.
// (round 6: with the conv1 weight gradient as bf16 slice products - x6 - the default first block is no longer bound by the
// matrix pipe, and the wide form was measured again for stride 1: option "bwd_first_wide", profiles/round6_first_wide_and_kmap_ab.txt)
template <int K1, int C1, int CO, int K, int S>
static bool launch_bwd_firstw_x6(hipStream_t st, const BwdFirstArgs& a, int grid) {
  if constexpr (S == 1 && K1 == 3 && CO <= 64) {
    hipLaunchKernelGGL((bwd_firstw_kernel<K1, C1, CO, K, S, 512, true>), dim3(grid), dim3(512), 0, st, a);
    return true;
  } else {
    return false;
  }
}

bool k_launch_bwd_firstw(hipStream_t st, int k1, int c1, int cout, int k, int stride, const BwdFirstArgs& a, int grid, bool wide_x6) {
#define X(K1, C1, CO, K, S)                                                                                    \
  if (k1 == K1 && c1 == C1 && cout == CO && k == K && stride == S) {                                           \
    if (wide_x6 && launch_bwd_firstw_x6<K1, C1, CO, K, S>(st, a, grid)) return true;                           \
    if constexpr (S > 1) {                                                                                     \
      hipLaunchKernelGGL((bwd_firstw_kernel<K1, C1, CO, K, S, 512>), dim3(grid), dim3(512), 0, st, a);         \
      return true;                                                                                             \
    }                                                                                                          \
  }
  MWW_FIRST_SHAPES(X)
#undef X
  return false;
}

}  // namespace mww

// Launchers of the specialised block kernels.  Each family is instantiated in its own translation unit (tu_fwd.hip,
// tu_bwd.hip, tu_bwdw.hip) so that build() compiles them in parallel: the hot kernels are fully unrolled templates and
// one translation unit holding every instantiation took twelve minutes to compile.  The launchers return false when the
// shape has no instantiation (mww_lib.hip turns that into MWW_ERR_UNSUPPORTED or falls back to the graph kernels).
#pragma once
#include "kernels_bwd.hip.h"

// (conv1 kernel, conv1 filters, block-1 pointwise filters, block-1 depthwise kernel, conv1 stride)
// Shapes with specialised block kernels: the reference's argparse defaults (3x1 first conv, 48 filters, [5],[9],[13],[21]),
// its training notebook (5x1 first conv stride 3, 64 filters, [5],[7,11],[9,15],[23] - multi-kernel groups are fused to
// their longest kernel) and the crosses of the two (either width with either kernel set, either first conv); everything
// else runs on the conv / depthwise graph kernels.
#ifdef MWW_SLIM   // kernel-tuning builds (tools/build_variant.sh): the default topology only, compiles in a quarter of the time
#define MWW_FIRST_SHAPES(X) X(3, 32, 48, 5, 1)
#define MWW_BLOCK_SHAPES(X) X(48, 48, 9) X(48, 48, 13) X(48, 48, 21)
#else
#define MWW_FIRST_SHAPES(X) X(3, 32, 48, 5, 1) X(5, 32, 64, 5, 3) X(5, 32, 64, 5, 1) X(3, 32, 64, 5, 1) X(5, 32, 48, 5, 3) X(5, 32, 48, 5, 1)
#define MWW_BLOCK_SHAPES(X)                                                                               \
  X(48, 48, 5) X(48, 48, 9) X(48, 48, 13) X(48, 48, 21) X(48, 48, 11) X(48, 48, 15) X(48, 48, 23)         \
  X(64, 64, 11) X(64, 64, 15) X(64, 64, 23) X(64, 64, 5) X(64, 64, 9) X(64, 64, 13) X(64, 64, 21)
#endif

namespace mww {

// mode: 0 = fp32, 1 = bf16 operands of the 1x1 contractions, 2 = bf16 operands and bf16 storage of p_k / g_k
bool k_launch_fwd_first(hipStream_t st, int mode, int k1, int c1, int cout, int k, int stride, const FwdFirstArgs& a, int grid);
bool k_launch_fwd_block(hipStream_t st, int mode, int cin, int cout, int k, const FwdBlockArgs& a, int grid);
bool k_launch_bwd_first(hipStream_t st, int mode, int k1, int c1, int cout, int k, int stride, const BwdFirstArgs& a, int grid);
bool k_launch_bwd_block(hipStream_t st, int mode, int cin, int cout, int k, bool last, const BwdBlockArgs& a, int grid);
// wide-workgroup form of the block backward (kernels_bwdw.hip.h: 512 threads per workgroup; every mode)
bool k_launch_bwd_blockw(hipStream_t st, int mode, int cin, int cout, int k, bool last, const BwdBlockArgs& a, int grid);
bool k_launch_bwd_firstw(hipStream_t st, int k1, int c1, int cout, int k, int stride, const BwdFirstArgs& a, int grid);

}  // namespace mww

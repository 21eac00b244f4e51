// Launchers of the specialised block kernels.  Each family is instantiated in its own translation unit (tu_fwd.hip,
// tu_bwd.hip, tu_bwdw.hip) so that build() compiles them in parallel: the hot kernels are fully unrolled templates and
// one translation unit holding every instantiation took twelve minutes to compile.  The launchers return false when the
// shape has no instantiation (mww_lib.hip turns that into MWW_ERR_UNSUPPORTED or falls back to the graph kernels).
#pragma once
#include "kernels_bwd.hip.h"

// Shapes with specialised block kernels.  MWW_BLOCK_SHAPES(X): X(cin, cout, depthwise kernel) - every pair of the widths 32 /
// 48 / 64 with every odd kernel length 3..23 (MixConv groups are fused to their longest kernel, so these are the lengths
// `--mixconv_kernel_sizes` reaches).  MWW_FIRST_SHAPES(X): X(conv1 kernel, conv1 filters, block-1 pointwise filters, block-1
// depthwise kernel, conv1 stride).  The reference's argparse defaults (3x1 first conv, 48 filters, [5],[9],[13],[21]) and its
// training notebook (5x1 first conv stride 3, 64 filters, [5],[7,11],[9,15],[23]) are two points of the table; the
// *_BF16 tables list the shapes that also have the bf16 modes (BASELINE configs[4]: the two documented topologies and their
// crosses).  Anything else - other widths, even or longer kernels, no first convolution, residual / repeated / attention /
// pooled blocks - runs on the conv / depthwise graph kernels (mww_block_kernels_cover() tells which family a model gets).
// The wide backward form exists for the square 48- and 64-wide blocks (kernels_bwdw.hip.h); the others keep 256 threads.
#define MWW_ODD_KS(X, CI, CO) X(CI, CO, 3) X(CI, CO, 5) X(CI, CO, 7) X(CI, CO, 9) X(CI, CO, 11) X(CI, CO, 13) X(CI, CO, 15) \
  X(CI, CO, 17) X(CI, CO, 19) X(CI, CO, 21) X(CI, CO, 23)
#define MWW_FIRST_KS(X, K1, CO, S) X(K1, 32, CO, 3, S) X(K1, 32, CO, 5, S) X(K1, 32, CO, 7, S)
#define MWW_FIRST_STRIDES(X, K1, CO) MWW_FIRST_KS(X, K1, CO, 1) MWW_FIRST_KS(X, K1, CO, 2) MWW_FIRST_KS(X, K1, CO, 3)
#ifdef MWW_SLIM   // kernel-tuning builds (tools/build_variant.sh): the default topology only, compiles in a quarter of the time
#define MWW_FIRST_SHAPES(X) X(3, 32, 48, 5, 1)
#define MWW_BLOCK_SHAPES(X) X(48, 48, 9) X(48, 48, 13) X(48, 48, 21)
#define MWW_FIRST_SHAPES_BF16(X) MWW_FIRST_SHAPES(X)
#define MWW_BLOCK_SHAPES_BF16(X) MWW_BLOCK_SHAPES(X)
#else
#define MWW_FIRST_SHAPES(X)                                                                                   \
  MWW_FIRST_STRIDES(X, 3, 32) MWW_FIRST_STRIDES(X, 3, 48) MWW_FIRST_STRIDES(X, 3, 64)                         \
  MWW_FIRST_STRIDES(X, 5, 32) MWW_FIRST_STRIDES(X, 5, 48) MWW_FIRST_STRIDES(X, 5, 64)
#define MWW_BLOCK_SHAPES(X)                                                                                   \
  MWW_ODD_KS(X, 32, 32) MWW_ODD_KS(X, 48, 48) MWW_ODD_KS(X, 64, 64) MWW_ODD_KS(X, 32, 48) MWW_ODD_KS(X, 32, 64)   \
  MWW_ODD_KS(X, 48, 64) MWW_ODD_KS(X, 48, 32) MWW_ODD_KS(X, 64, 48) MWW_ODD_KS(X, 64, 32)
#define MWW_FIRST_SHAPES_BF16(X) X(3, 32, 48, 5, 1) X(5, 32, 64, 5, 3) X(5, 32, 64, 5, 1) X(3, 32, 64, 5, 1) X(5, 32, 48, 5, 3) X(5, 32, 48, 5, 1)
#define MWW_BLOCK_SHAPES_BF16(X)                                                                              \
  X(48, 48, 5) X(48, 48, 9) X(48, 48, 13) X(48, 48, 21) X(48, 48, 11) X(48, 48, 15) X(48, 48, 23)             \
  X(64, 64, 11) X(64, 64, 15) X(64, 64, 23) X(64, 64, 5) X(64, 64, 9) X(64, 64, 13) X(64, 64, 21)
#endif

namespace mww {

// mode: 0 = fp32, 1 = bf16 operands of the 1x1 contractions, 2 = bf16 operands and bf16 storage of p_k / g_k
// x6: the first convolution (and, backward, its weight gradient) as bf16 slice products (common.hip.h; fp32 mode, stride 1)
bool k_launch_fwd_first(hipStream_t st, int mode, int k1, int c1, int cout, int k, int stride, const FwdFirstArgs& a, int grid, bool x6);
bool k_launch_fwd_block(hipStream_t st, int mode, int cin, int cout, int k, const FwdBlockArgs& a, int grid);
bool k_launch_bwd_first(hipStream_t st, int mode, int k1, int c1, int cout, int k, int stride, const BwdFirstArgs& a, int grid, bool x6);
bool k_launch_bwd_block(hipStream_t st, int mode, int cin, int cout, int k, bool last, const BwdBlockArgs& a, int grid);
// wide-workgroup form of the block backward (kernels_bwdw.hip.h: 512 threads per workgroup; every mode)
bool k_launch_bwd_blockw(hipStream_t st, int mode, int cin, int cout, int k, bool last, const BwdBlockArgs& a, int grid);
bool k_launch_bwd_firstw(hipStream_t st, int k1, int c1, int cout, int k, int stride, const BwdFirstArgs& a, int grid, bool wide_x6);

}  // namespace mww

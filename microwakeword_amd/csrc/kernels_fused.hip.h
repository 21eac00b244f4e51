// Several layers of the MixedNet train step in one launch (common.hip.h "several layers in one launch").
//
//   bwd_fused_kernel : block 4 -> | -> block 3 -> | -> block 2 -> | -> first block      (| = grid_sync: BN gradient statistics)
//
// (Round 3 also had the forward counterpart; it measured 1 % slower than four launches - 0.3395 against 0.3365 ms per step,
// profiles/round3_bench_fused_stages_*.json - and was removed when the forward launches took grids of their own)
//
// The stages are the bodies of the one-launch-per-layer kernels (fwd_first_kernel ... bwd_first_kernel: the same text,
// *_body.inc), run by persistent workgroups over the same windows in every stage; the classifier head (another window ->
// workgroup mapping) and the gradient assembly (reads every workgroup's partial rows) stay launches of their own, so the
// train step is seven launches instead of ten.  The reference has no counterpart: TF runs ~120 kernels per step
// (microwakeword/train.py:295-299 model.train_on_batch); what is fused here are the layers of mixednet.py:307-360.
#pragma once
#include "kernels_bwd.hip.h"

namespace mww {

constexpr int kFusedBlocks = 4;   // fused launches exist for four-block topologies (the reference's default and its notebook's)

struct BwdFusedArgs {
  BwdBlockArgs blk[kFusedBlocks - 1];   // last block first
  BwdFirstArgs first;
  GridSync sync;
};

__host__ __device__ constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int K1, int C1, int S, int CW, int KA, int KB, int KC, int KD>
struct FusedLds {
  static constexpr int BWD = cmax(cmax(BwdFirstLds<K1, C1, CW, KA, S>::END, BwdBlockLds<CW, CW, KB>::END),
                                  cmax(BwdBlockLds<CW, CW, KC>::END, BwdBlockLds<CW, CW, KD>::END));
};

template <int K1, int C1, int S, int CW, int KA, int KB, int KC, int KD, bool BF, bool SB>
__global__ __launch_bounds__(kThreads, (CW > 48 || S > 1 ? 1 : 2)) void bwd_fused_kernel(BwdFusedArgs a) {
  HIP_DYNAMIC_SHARED(float4, lds4)
  float* lds = reinterpret_cast<float*>(lds4);
  bwd_block_stage<CW, CW, KD, true, BF, SB>(a.blk[0], lds, nullptr, 0);
  stage_end();
  bwd_block_stage<CW, CW, KC, false, BF, SB>(a.blk[1], lds, &a.sync, 1);
  stage_end();
  bwd_block_stage<CW, CW, KB, false, BF, SB>(a.blk[2], lds, &a.sync, 2);
  stage_end();
  bwd_first_stage<K1, C1, CW, KA, S, BF, SB>(a.first, lds, &a.sync, 3);
  grid_sync_finish(a.sync);
}

}  // namespace mww

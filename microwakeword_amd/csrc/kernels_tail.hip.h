// One launch for the three independent pieces of work that follow the classifier head in a MixedNet train
// step: the BN backward coefficients of the last block (bn_bwd_finalize, C workgroups), the dense-weight
// gradient (dense_grad, batch-chunked) and the metric update (one workgroup).  Each is latency-bound on
// its own (5-10 us for a few microseconds of work); as roles of one grid they overlap.
#pragma once
#include "kernels_bwd.hip.h"
#include "kernels_head.hip.h"

namespace mww {

struct HeadTailArgs {
  BnBwdFinalizeArgs fin;
  DenseGradArgs dense;
  MetricsArgs met;
  int n_fin;        // workgroups [0, n_fin): finalize channel blockIdx.x
  int ndx, ndy;     // workgroups [n_fin, n_fin + ndx*ndy): dense-gradient tile (x + ndx*y)
  int do_metrics;   // one more workgroup: metric update
};

__global__ __launch_bounds__(kThreads) void head_tail_kernel(HeadTailArgs a) {
  __shared__ __attribute__((aligned(16))) double sAcc[256 + 16];
  __shared__ double sOut[2];
  __shared__ unsigned sH101[2][101];
  __shared__ unsigned sH200[2][200];
  __shared__ unsigned sCnt[8];
  const int bid = blockIdx.x, tid = threadIdx.x;
  if (bid < a.n_fin) {
    bn_bwd_finalize_body(a.fin, bid, sAcc, sOut, tid);
  } else if (bid < a.n_fin + a.ndx * a.ndy) {
    const int i = bid - a.n_fin;
    dense_grad_body(a.dense, i % a.ndx, i / a.ndx, tid);
  } else if (a.do_metrics) {
    metrics_body<kThreads>(a.met, sH101, sH200, sCnt, sAcc, tid);   // sAcc doubles as the per-thread BCE partials
  }
}

}  // namespace mww

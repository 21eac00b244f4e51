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

// MixedNet train step with the statistics hand-over: the dense-weight gradient and the metric update have no
// consumer before the gradient finish, so they ride in the gradient-reduction launch as extra z-slices of its grid
// (slices [0, kGradSplit) reduce the weight-gradient partials; the blocks of the slices above are numbered linearly:
// ndx * kGradSplit dense-gradient tiles, then one metric workgroup).  The dense tiles write their batch-chunk sums
// straight into the staging slices the finish kernel adds up (chunk by = slice by), so no partial rows and no segment.
struct GradReduceTailArgs {
  DenseGradArgs dense;   // part = stage + offset of the dense kernel, stride = P, chunk = ceil(B / kGradSplit)
  MetricsArgs met;
  int ndx;
  int do_metrics;
};

__global__ __launch_bounds__(kThreads) void grad_reduce_tail_kernel(GradReduceArgs a, GradReduceTailArgs t) {
  __shared__ __attribute__((aligned(16))) double sAcc[256 + 16];
  __shared__ unsigned sH101[2][101];
  __shared__ unsigned sH200[2][200];
  __shared__ unsigned sCnt[8];
  if ((int)blockIdx.z < kGradSplit) {
    grad_reduce_body(a);
    return;
  }
  const int id = (((int)blockIdx.z - kGradSplit) * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x;
  const int n_dense = t.ndx * kGradSplit;
  if (id < n_dense) dense_grad_body(t.dense, id % t.ndx, id / t.ndx, threadIdx.x);
  else if (id == n_dense && t.do_metrics) metrics_body<kThreads>(t.met, sH101, sH200, sCnt, sAcc, threadIdx.x);
}

}  // namespace mww

// One launch for the three independent pieces of work that follow the classifier head in a MixedNet train
// step: the BN backward coefficients of the last block (bn_bwd_finalize, C workgroups), the dense-weight
// gradient (dense_grad, batch-chunked) and the metric update (one workgroup).  Each is latency-bound on
// its own (5-10 us for a few microseconds of work); as roles of one grid they overlap.
#pragma once
#include <type_traits>

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
    metrics_body<kThreads>(a.met, sH101, sH200, sCnt, sAcc, tid);   // sAcc doubles as the per-wave BCE partials
  }
}

// ---- gradient assembly (+ Adam) in one launch -------------------------------------------------------------------
// Every workgroup owns kFinalCols consecutive parameters of one segment and finishes them completely: its 256 threads
// are kFinalSlices row slices x kFinalCols parameters; a thread sums its slice of the segment's rows in a fixed order
// (16 loads in flight, 128-byte coalesced), the slices are added in a fixed order through LDS => bit-reproducible
// gradients; then the structural mask and the gradient scale and - in the single-device step - Keras Adam (SURVEY
// A.6) are applied in place.  (Round 1 used a two-level reduction through a [32][P] staging buffer in three
// launches: partial sums, finish + Adam; each tiny launch costs ~6 us on the stream.)
// Rows of a segment:  kind 0  the per-workgroup partial rows a backward kernel wrote;
//                     kind 1  the dense kernel's gradient, one row per window: dz_b * relu(bn(p_L[b, e])) and, as the
//                             last element, the bias gradient sum_b dz_b (second read of p_L);
//                     kind 2  values that are already final in grad[] (BN gamma / beta, written by the folding kernels);
//                     kind 3  parameters nothing contributes to (gradient 0).
// One more workgroup updates the metric counters (train.py:209-221) - they have no consumer on the device.
constexpr int kFinalCols = 32, kFinalSlices = kThreads / kFinalCols;
#ifndef MWW_FINAL_ROWS_IN_FLIGHT
#define MWW_FINAL_ROWS_IN_FLIGHT 64
#endif
#ifndef MWW_FINAL_DENSE_PAIR
#define MWW_FINAL_DENSE_PAIR 1
#endif
constexpr int kMaxFinalSegments = 56;
enum { kSegPartials = 0, kSegDense = 1, kSegDirect = 2, kSegZero = 3 };
struct FinalSegment {
  const float* part;   // [G][stride] (kind 0)
  int G, stride, n, dst, kind;
  int block0;          // first workgroup of this segment
};
struct GradFinalArgs {
  // first workgroup of every segment, ascending, INT_MAX past nseg: a workgroup finds its segment by counting - 56 scalar
  // compares on four wide scalar loads - instead of walking seg[].block0 (one dependent scalar-cache round trip per two
  // segments in the round-2 ISA: ~14 of them in front of every workgroup's first vector load)
  int block0[kMaxFinalSegments];
  FinalSegment seg[kMaxFinalSegments];
  int nseg, nblocks;       // nblocks = workgroups of all segments (the metric workgroup, if any, is block 0 in front of them)
  DenseGradArgs dense;     // kind 1: p, scale, shift, dz, B, n, C, keep, residual (part / stride / chunk unused)
  MetricsArgs met;
  int do_metrics;
  const float* mask;       // [P] 1 = trainable tap, 0 = structural zero (MixConv padding)
  float* grad;             // [P]
  float scale;
  AdamArgs adam;
  int apply_adam;
};

// The dense-weight gradient of one column over the batch chunks [c0, c1) (grad_final_kernel's dense role).  Storage mode,
// residual branch and dropout scale are template arguments and the row index is clamped instead of predicated: with
// run-time selects around them the 32 rows of a chunk became 32 dependent round trips (the round-2 ISA waited after every
// row); now every load of a batch is issued before the first use.  Same arithmetic as dense_grad_kernel's batch chunks
// summed as partial rows: an fma chain per chunk of d.chunk windows, the chunk sums added in order.
template <bool SB, bool RES, bool KEEP>
__device__ __forceinline__ float dense_role_chunks(const DenseGradArgs& d, int e, int c0, int c1, float sc, float sh, float rsc,
                                                   float rsh, size_t roff, size_t rstride) {
  constexpr int U = (RES || KEEP) ? 8 : 32;   // rows in flight per thread (a chunk of the headline batch is one round trip)
  float acc = 0.f;
  if constexpr (!RES && !KEEP && MWW_FINAL_DENSE_PAIR != 0) {
    if (d.chunk <= U) {
      // chunks of at most U windows (the headline batch: 32): the rows of TWO chunks are requested together - a slice's four
      // chunks are two memory round trips instead of four.  Same sums in the same order (sub per chunk, chunks in order).
      for (int ci = c0; ci < c1; ci += 2) {
        float v[2][U], dz[2][U];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int b0 = min(ci + k, c1 - 1) * d.chunk, b1 = min(d.B, b0 + d.chunk);
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const size_t row = (size_t)min(b0 + u, b1 - 1);
            v[k][u] = load_elem<SB>(d.p, row * d.n + e);
            dz[k][u] = d.dz[row];
          }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          if (ci + k < c1) {
            const int b0 = (ci + k) * d.chunk, b1 = min(d.B, b0 + d.chunk);
            float sub = 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u)
              if (b0 + u < b1) sub = fmaf(dz[k][u], fmaxf(fmaf(v[k][u], sc, sh), 0.f), sub);
            acc += sub;
          }
        }
      }
      return acc;
    }
  }
  for (int ci = c0; ci < c1; ++ci) {
    const int b0 = ci * d.chunk, b1 = min(d.B, b0 + d.chunk);
    float sub = 0.f;
    for (int bb = b0; bb < b1; bb += U) {
      float v[U], dz[U], r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t row = (size_t)min(bb + u, b1 - 1);   // rows past the chunk re-read its last row (and are not summed)
        v[u] = load_elem<SB>(d.p, row * d.n + e);
        dz[u] = d.dz[row];
        r[u] = 0.f;
        if constexpr (RES) r[u] = d.rp[row * rstride + roff];
        if constexpr (KEEP) dz[u] *= d.keep[row * d.n + e];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if constexpr (RES) r[u] = fmaf(r[u], rsc, rsh);
        if (bb + u < b1) sub = fmaf(dz[u], fmaxf(fmaf(v[u], sc, sh) + r[u], 0.f), sub);
      }
    }
    acc += sub;
  }
  return acc;
}

__device__ __forceinline__ float dense_role_dispatch(const DenseGradArgs& d, int e, int c0, int c1, float sc, float sh, float rsc,
                                                     float rsh, size_t roff, size_t rstride) {
#define MWW_DR(SB, RES, KEEP) return dense_role_chunks<SB, RES, KEEP>(d, e, c0, c1, sc, sh, rsc, rsh, roff, rstride)
  if (d.p_bf16) {
    if (d.rp) { if (d.keep) MWW_DR(true, true, true); else MWW_DR(true, true, false); }
    else { if (d.keep) MWW_DR(true, false, true); else MWW_DR(true, false, false); }
  } else {
    if (d.rp) { if (d.keep) MWW_DR(false, true, true); else MWW_DR(false, true, false); }
    else { if (d.keep) MWW_DR(false, false, true); else MWW_DR(false, false, false); }
  }
#undef MWW_DR
}

__global__ __launch_bounds__(kThreads) void grad_final_kernel(GradFinalArgs a) {
  __shared__ __attribute__((aligned(16))) double sAcc[256 + 16];
  __shared__ unsigned sH101[2][101];
  __shared__ unsigned sH200[2][200];
  __shared__ unsigned sCnt[8];
  const int tid = threadIdx.x;
  // (the metric workgroup, if any, is block 0: its chain - probabilities in, histograms, counters out - is the longest
  // after the dense role's)
  const int bid = (int)blockIdx.x - a.do_metrics;
  if (bid < 0) {
    metrics_body<kThreads>(a.met, sH101, sH200, sCnt, sAcc, tid);   // sAcc doubles as the per-wave BCE partials
    return;
  }
  int si = 0;
#pragma unroll
  for (int i = 1; i < kMaxFinalSegments; ++i) si += (int)((unsigned)(a.block0[i] - 1 - bid) >> 31);   // bid >= block0[i], as sign arithmetic
  const FinalSegment s = a.seg[si];
  const int pl = tid % kFinalCols, sl = tid / kFinalCols;
  const int e = (bid - s.block0) * kFinalCols + pl;
  const bool in = e < s.n;
  float acc = 0.f;
  if (s.kind == kSegPartials) {
    const int per = (s.G + kFinalSlices - 1) / kFinalSlices;
    const int j0 = sl * per, j1 = min(s.G, j0 + per);
    // a slice of the headline grids (512 partial rows / 8 slices) is ONE batch of 64 loads: four dependent batches of 16
    // were four memory round trips in front of every workgroup's sum (MWW_FINAL_ROWS_IN_FLIGHT: tuning builds)
    constexpr int UB = MWW_FINAL_ROWS_IN_FLIGHT;
    for (int jb = j0; jb < j1; jb += UB) {
      float v[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) v[u] = (in && jb + u < j1) ? s.part[(size_t)(jb + u) * s.stride + e] : 0.f;
#pragma unroll
      for (int u = 0; u < UB; ++u) acc += v[u];
    }
  } else if (s.kind == kSegDense) {
    // same arithmetic as dense_grad_kernel's batch chunks summed as partial rows (the path without the ride-along role):
    // an fma chain per chunk of d.chunk windows, the chunk sums added in order => bit-identical gradients either way
    const DenseGradArgs& d = a.dense;
    const int nchunk = (d.B + d.chunk - 1) / d.chunk;
    const int per = (nchunk + kFinalSlices - 1) / kFinalSlices;
    const int c0 = sl * per, c1 = min(nchunk, c0 + per);
    if (in && e < d.n) {
      const int c = e % d.C;
      const float sc = d.scale[c], sh = d.shift[c];
      const float rsc = d.rp ? d.rscale[c] : 0.f, rsh = d.rp ? d.rshift[c] : 0.f;
      const size_t roff = d.rp ? (size_t)d.rdrop * d.C + e : 0, rstride = (size_t)d.rT * d.C;
      acc = dense_role_dispatch(d, e, c0, c1, sc, sh, rsc, rsh, roff, rstride);
    } else if (in) {   // e == d.n: the dense bias
      for (int ci = c0; ci < c1; ++ci) {
        const int b0 = ci * d.chunk, b1 = min(d.B, b0 + d.chunk);
        float sub = 0.f;
        for (int b = b0; b < b1; ++b) sub += d.dz[b];
        acc += sub;
      }
    }
  } else if (s.kind == kSegDirect) {
    if (in && sl == 0) acc = a.grad[s.dst + e];
  }
  float* sSum = reinterpret_cast<float*>(sAcc);
  sSum[tid] = acc;
  __syncthreads();
  if (sl == 0 && in) {
    float g = 0.f;
#pragma unroll
    for (int j = 0; j < kFinalSlices; ++j) g += sSum[j * kFinalCols + pl];
    const int p = s.dst + e;
    g = g * a.mask[p] * a.scale;
    a.grad[p] = g;
    if (a.apply_adam) {
      const AdamArgs& ad = a.adam;
      const float alpha = ad.hyper[0];
      const float gg = g * ad.hyper[1];
      float m = ad.m[p], v = ad.v[p];
      m += (gg - m) * (1.0f - ad.beta1);
      v += (gg * gg - v) * (1.0f - ad.beta2);
      ad.m[p] = m;
      ad.v[p] = v;
      ad.param[p] -= alpha * m / (sqrtf(v) + ad.eps);
    }
  }
}

}  // namespace mww

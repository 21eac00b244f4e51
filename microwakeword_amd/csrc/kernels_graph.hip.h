// Conv -> (sub-spectral) BatchNorm -> ReLU graphs with time-aligned concatenation: the layer
// vocabulary of the reference's Inception model (microwakeword/inception.py:46-141 conv2d_bn /
// conv2d_bn_delay / the three-branch block with StridedDrop + Concatenate, :232-340 model) and of
// SubSpectralNormalization (layers/sub_spectral_normalization.py:49-61).
//
// An op = valid k x 1 convolution (dilation d, no bias) over the channel-concatenation of up to
// three sources, each aligned by dropping its leading frames, followed by BN (or SSN with g slots:
// channel c uses slot c % g) and ReLU.  As in the MixedNet kernels only the pre-BN output p of
// every op is materialised; consumers apply the producer's folded BN + ReLU on load, and the
// backward pass recomputes activations from p.
//
// Mapping (all kernels: 256-thread workgroups that own whole windows, grid-stride over the batch):
//   * staging / epilogues: thread -> (channel c = tid % C, frame group tid / C): a wave touches
//     consecutive channels of consecutive frames => contiguous HBM segments;
//   * convolution: lane <-> output frame, NC accumulators per lane, the activation tile of the whole
//     window in LDS (odd row pitch => conflict-free column walks), weights through scalar loads
//     (uniform addresses), so the inner loop is one ds_read + NC v_fma per input element;
//   * the data gradient is the same convolution run on the BN-backward-transformed output gradient
//     with reversed taps and transposed weights (gweights_transpose_kernel), its epilogue scatters
//     into the sources' gradient tensors (first consumer stores, later ones accumulate, the last one
//     also produces the BN statistics partials of that source);
//   * the weight gradient maps thread -> (tap j, input channel ci, frame subset q) with NC
//     accumulators held across the workgroup's windows; partials are summed by grad_reduce_kernel.
// These channel counts (10..48) are far from MFMA tile shapes and the work is a few MFLOP per
// window, so the kernels are VALU/LDS kernels by design.
#pragma once
#include "common.hip.h"

namespace mww {

constexpr int kGMaxSrc = 3;
constexpr int kGB = 8;   // rows a thread keeps in flight in the staging / epilogue loops
enum { GSRC_IDENTITY = 1, GSRC_ACCUM = 2, GSRC_STATS = 4, GSRC_GRAD = 8, GSRC_LINEAR = 16 };   // LINEAR: affine only, no ReLU

struct GSrc {
  const float* p;       // [B][T][C] pre-BN output of the producer (or the spectrogram)
  const float* scale;   // producer's folded BN (unused with GSRC_IDENTITY)
  const float* shift;
  const float* mean;    // producer's batch statistics (backward only)
  const float* rstd;
  float* g;             // gradient at the producer's BN output, ReLU mask applied [B][T][C]
  float* gstat_part;    // [grid][2][ld]
  int T, C, toff, flags;
  int ld, c0;           // the source is channels [c0, c0+C) of a producer tensor with ld channels per frame
  // residual branch added to the producer's normalised output before its activation (mixednet.py:340-358:
  // residual = BN(conv1x1(block input)); net = relu(BN(...) + StridedDrop(residual))): frame t of the producer
  // pairs with frame t + rdrop of the residual op's pre-BN tensor rp [B][rT][C]; null = none
  const float* rp;
  const float* rscale;
  const float* rshift;
  int rT, rdrop;
};

// value of a source element before its activation clamp
__device__ __forceinline__ float src_affine(const GSrc& s, float v, float sc, float sh, float rv, float rsc, float rsh) {
  float y = fmaf(v, sc, sh);
  if (s.rp) y += fmaf(rv, rsc, rsh);
  return y;
}

struct GBnBwd {           // BN backward of the op itself: dp = c1 * (g - mg - xhat * mgx)
  const float* g;         // [B][Tout][C]
  const float* p;
  const float *mean, *rstd, *c1, *mg, *mgx;
};

// ---------------------------------------------------------------------------------------------
// staging helpers
__device__ __forceinline__ void stage_sources(const GSrc* src, int n_src, int b, int rows, float* sIn, int PI, int tid) {
  int c0 = 0;
  for (int i = 0; i < n_src; ++i) {
    const GSrc& s = src[i];
    const int C = s.C, nrg = kThreads / C, c = tid % C, rg = tid / C;
    if (rg < nrg) {
      const bool ident = (s.flags & GSRC_IDENTITY) != 0;
      const float sc = ident ? 1.f : s.scale[s.c0 + c], sh = ident ? 0.f : s.shift[s.c0 + c];
      const float lo = (s.flags & (GSRC_IDENTITY | GSRC_LINEAR)) ? -3.0e38f : 0.f;   // ReLU as a clamp from below
      const float* base = s.p + ((size_t)b * s.T + s.toff) * s.ld + s.c0 + c;
      const float rsc = s.rp ? s.rscale[c] : 0.f, rsh = s.rp ? s.rshift[c] : 0.f;
      const float* rbase = s.rp ? s.rp + ((size_t)b * s.rT + s.toff + s.rdrop) * C + c : nullptr;
      // kGB rows per thread in flight (a rolled loop with one load and one LDS write per row is one memory round trip
      // per row: ~rows * C / 256 of them per window and source)
      for (int t0 = rg; t0 < rows; t0 += kGB * nrg) {
        float v[kGB], rv[kGB];
#pragma unroll
        for (int u = 0; u < kGB; ++u) {
          const int t = t0 + u * nrg;
          v[u] = t < rows ? base[(size_t)t * s.ld] : 0.f;
          rv[u] = (t < rows && rbase) ? rbase[(size_t)t * C] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < kGB; ++u) {
          const int t = t0 + u * nrg;
          if (t < rows) sIn[t * PI + c0 + c] = fmaxf(src_affine(s, v[u], sc, sh, rv[u], rsc, rsh), lo);
        }
      }
    }
    c0 += C;
  }
}

__device__ __forceinline__ void stage_dp(const GBnBwd& y, int C, int b, int rows, float* dst, int ld, int tid) {
  const int nrg = kThreads / C, c = tid % C, rg = tid / C;
  if (rg < nrg) {
    const float mu = y.mean[c], rs = y.rstd[c], c1 = y.c1[c], mg = y.mg[c], mgx = y.mgx[c];
    const size_t base = (size_t)b * rows * C + c;
    for (int t0 = rg; t0 < rows; t0 += kGB * nrg) {
      float g[kGB], p[kGB];
#pragma unroll
      for (int u = 0; u < kGB; ++u) {
        const int t = t0 + u * nrg;
        g[u] = t < rows ? y.g[base + (size_t)t * C] : 0.f;
        p[u] = t < rows ? y.p[base + (size_t)t * C] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < kGB; ++u) {
        const int t = t0 + u * nrg;
        if (t < rows) dst[t * ld + c] = c1 * (g[u] - mg - (p[u] - mu) * rs * mgx);
      }
    }
  }
}

// per-thread (s1, s2) of channel c = tid % C, frame group tid / C  ->  part[2][ld] of this workgroup
// (columns [0, C) of each statistic's row; ld = C unless the channels are a slice of a wider tensor)
__device__ __forceinline__ void write_channel_partials(float s1, float s2, int C, float* sRed, float* part, int tid, int ld) {
  const int nrg = kThreads / C, c = tid % C, rg = tid / C;
  __syncthreads();
  if (rg < nrg) {
    sRed[(rg * 2 + 0) * C + c] = s1;
    sRed[(rg * 2 + 1) * C + c] = s2;
  }
  __syncthreads();
  if (tid < 2 * C) {
    float v = 0.f;
    for (int r = 0; r < nrg; ++r) v += sRed[r * 2 * C + tid];
    part[(tid / C) * ld + (tid % C)] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// MODE 0: forward convolution.  MODE 1: data gradient.
struct GConvArgs {
  GSrc src[kGMaxSrc];
  int n_src;
  const float* w;       // MODE 0: [k][cin][NC];  MODE 1: reversed taps, transposed: [k][cin = fwd cout][NC = fwd cin]
  int k, dil, cin;      // cin = channels reduced over
  int stride;           // MODE 0 only: time stride (ops fed by the spectrogram, e.g. MixedNet's first conv); else 1
  int B;
  int Tin;              // MODE 0: aligned input frames;   MODE 1: frames of the op's output (dp)
  int Tout;             // MODE 0: (Tin - (k-1)*dil - 1)/stride + 1;   MODE 1: Tin + (k-1)*dil (= aligned input frames)
  float* out;           // MODE 0: pre-BN output [B][Tout][NC]
  float* stat_part;     // MODE 0: [grid][2][NC], or null when the op has no batch statistics
  GBnBwd y;             // MODE 1
};

// bid / nb: this workgroup's index and the number of workgroups sharing the batch (a launch may hold several roles)
template <int NC, int MODE>
__device__ __forceinline__ void gconv_body(const GConvArgs& a, const int bid, const int nb) {
  HIP_DYNAMIC_SHARED(float4, g_smem4)
  float* g_smem = reinterpret_cast<float*>(g_smem4);
  __shared__ float sRed[2 * kThreads];
  constexpr int NCP = (NC + 3) / 4 * 4;   // weight rows padded to whole float4 (broadcast ds_read_b128)
  const int tid = threadIdx.x;
  const int PI = a.cin | 1, PO = NC | 1;
  const int pad = MODE == 1 ? (a.k - 1) * a.dil : 0;
  const int rows_in = a.Tin + 2 * pad;
  float* sW = g_smem;                       // [k*cin][NCP], loaded once per workgroup
  float* sIn = sW + a.k * a.cin * NCP;
  float* sOut = sIn + rows_in * PI;
  // running (sum, sum of squares) of this thread's channel: MODE 0 of the output, MODE 1 one pair per source, kept
  // in LDS so that the sources can be walked by a real loop (their descriptors stay in the kernel-argument
  // segment instead of pinning ~100 SGPRs)
  float s1o = 0.f, s2o = 0.f;
  __shared__ float sSrcAcc[MODE == 1 ? kGMaxSrc * 2 * kThreads : 1];
  if (MODE == 1)
    for (int i = 0; i < kGMaxSrc * 2; ++i) sSrcAcc[i * kThreads + tid] = 0.f;

  for (int i = tid; i < a.k * a.cin * NC; i += kThreads) sW[(i / NC) * NCP + (i % NC)] = a.w[i];
  if (MODE == 1) {
    // the zero frames around dp are written once: staging only touches the middle
    for (int i = tid; i < pad * PI; i += kThreads) {
      sIn[i] = 0.f;
      sIn[(pad + a.Tin) * PI + i] = 0.f;
    }
  }
  for (int b = bid; b < a.B; b += nb) {
    __syncthreads();   // the previous window's epilogue is done with sOut / the conv with sIn
    if (MODE == 0) stage_sources(a.src, a.n_src, b, a.Tin, sIn, PI, tid);
    else stage_dp(a.y, a.cin, b, a.Tin, sIn + pad * PI, PI, tid);
    __syncthreads();
    for (int t = tid; t < a.Tout; t += kThreads) {
      float acc[NCP];
#pragma unroll
      for (int co = 0; co < NCP; ++co) acc[co] = 0.f;
      for (int j = 0; j < a.k; ++j) {
        const float* row = sIn + ((MODE == 0 ? t * a.stride : t) + j * a.dil) * PI;
        const float4* wj = reinterpret_cast<const float4*>(sW + j * a.cin * NCP);
#pragma unroll 2
        for (int ci = 0; ci < a.cin; ++ci) {
          const float v = row[ci];
#pragma unroll
          for (int c4 = 0; c4 < NCP / 4; ++c4) {
            const float4 w = wj[ci * (NCP / 4) + c4];
            acc[c4 * 4 + 0] = fmaf(v, w.x, acc[c4 * 4 + 0]);
            if (c4 * 4 + 1 < NC) acc[c4 * 4 + 1] = fmaf(v, w.y, acc[c4 * 4 + 1]);
            if (c4 * 4 + 2 < NC) acc[c4 * 4 + 2] = fmaf(v, w.z, acc[c4 * 4 + 2]);
            if (c4 * 4 + 3 < NC) acc[c4 * 4 + 3] = fmaf(v, w.w, acc[c4 * 4 + 3]);
          }
        }
      }
#pragma unroll
      for (int co = 0; co < NC; ++co) sOut[t * PO + co] = acc[co];
    }
    __syncthreads();
    if (MODE == 0) {
      const int nrg = kThreads / NC, c = tid % NC, rg = tid / NC;
      if (rg < nrg) {
        float* dst = a.out + (size_t)b * a.Tout * NC + c;
        for (int t = rg; t < a.Tout; t += nrg) {
          const float v = sOut[t * PO + c];
          dst[(size_t)t * NC] = v;
          s1o += v;
          s2o = fmaf(v, v, s2o);
        }
      }
    } else {
      int c0 = 0;
      for (int i = 0; i < a.n_src; ++i) {
        const GSrc& s = a.src[i];
        const int C = s.C;
        if (s.flags & GSRC_GRAD) {
          const int nrg = kThreads / C, c = tid % C, rg = tid / C;
          if (rg < nrg) {
            const float sc = s.scale[s.c0 + c], sh = s.shift[s.c0 + c];
            const bool accum = (s.flags & GSRC_ACCUM) != 0, stats = (s.flags & GSRC_STATS) != 0;
            const bool linear = (s.flags & GSRC_LINEAR) != 0;
            const float mu = stats ? s.mean[s.c0 + c] : 0.f, rs = stats ? s.rstd[s.c0 + c] : 0.f;
            const size_t base = (size_t)b * s.T * s.ld + s.c0 + c;
            const float rsc = s.rp ? s.rscale[c] : 0.f, rsh = s.rp ? s.rshift[c] : 0.f;
            const float* rbase = s.rp ? s.rp + ((size_t)b * s.rT + s.rdrop) * C + c : nullptr;
            float t1 = 0.f, t2 = 0.f;
            for (int tb = rg; tb < s.T; tb += kGB * nrg) {
              float pv[kGB], rvv[kGB], gold[kGB];
#pragma unroll
              for (int u = 0; u < kGB; ++u) {
                const int t = tb + u * nrg;
                const bool ok = t < s.T;
                const size_t idx = base + (size_t)(ok ? t : 0) * s.ld;
                pv[u] = ok ? s.p[idx] : 0.f;
                rvv[u] = (ok && rbase) ? rbase[(size_t)t * C] : 0.f;
                gold[u] = (ok && accum) ? s.g[idx] : 0.f;
              }
#pragma unroll
              for (int u = 0; u < kGB; ++u) {
                const int t = tb + u * nrg;
                if (t < s.T) {
                  const size_t idx = base + (size_t)t * s.ld;
                  const float p = pv[u];
                  const int r = t - s.toff;
                  float gv = (r >= 0 && (linear || src_affine(s, p, sc, sh, rvv[u], rsc, rsh) > 0.f)) ? sOut[r * PO + c0 + c] : 0.f;
                  if (accum) gv += gold[u];
                  s.g[idx] = gv;
                  t1 += gv;
                  t2 = fmaf(gv, (p - mu) * rs, t2);
                }
              }
            }
            sSrcAcc[(i * 2 + 0) * kThreads + tid] += t1;
            sSrcAcc[(i * 2 + 1) * kThreads + tid] += t2;
          }
        }
        c0 += C;
      }
    }
  }
  if (MODE == 0) {
    if (a.stat_part) write_channel_partials(s1o, s2o, NC, sRed, a.stat_part + (size_t)bid * 2 * NC, tid, NC);
  } else {
    for (int i = 0; i < a.n_src; ++i) {
      const GSrc& s = a.src[i];
      if ((s.flags & GSRC_GRAD) && (s.flags & GSRC_STATS))
        write_channel_partials(sSrcAcc[(i * 2 + 0) * kThreads + tid], sSrcAcc[(i * 2 + 1) * kThreads + tid], s.C, sRed,
                               s.gstat_part + (size_t)bid * 2 * s.ld + s.c0, tid, s.ld);
    }
  }
}

template <int NC, int MODE>
__global__ __launch_bounds__(kThreads) void gconv_kernel(GConvArgs a) {
  gconv_body<NC, MODE>(a, blockIdx.x, gridDim.x);
}

// Two independent ops of the same shape ("twins": Inception's k x 1 convs of branch 2 and branch 3) as the two
// halves of one launch: workgroups [0, nb) run op a0, [nb, 2 nb) op a1.
template <int NC>
__global__ __launch_bounds__(kThreads) void gconv_fwd2_kernel(GConvArgs a0, GConvArgs a1, int nb) {
  if ((int)blockIdx.x < nb) gconv_body<NC, 0>(a0, blockIdx.x, nb);
  else gconv_body<NC, 0>(a1, blockIdx.x - nb, nb);
}

// ---------------------------------------------------------------------------------------------
// weight gradient: dW[j][ci][co] = sum_{b,t} act[b][t + j*dil][ci] * dp[b][t][co]
struct GWgradArgs {
  GSrc src[kGMaxSrc];
  int n_src;
  GBnBwd y;
  int k, dil, cin, B, Tin, Tout;
  int stride;           // time stride of the forward convolution
  int nq;               // frame subsets per (tap, channel) task: nq * k * cin <= 256
  float* grad_part;     // [grid * nq][k*cin*NC]
};

constexpr int kGWgChunk = 4;   // output channels per pass of the in-workgroup reduction over the frame subsets

template <int NC>
__device__ __forceinline__ void gconv_wgrad_body(const GWgradArgs& a, const int bid, const int nb) {
  HIP_DYNAMIC_SHARED(float4, g_smem4)
  float* g_smem = reinterpret_cast<float*>(g_smem4);
  constexpr int PO = (NC + 3) / 4 * 4;
  const int tid = threadIdx.x;
  const int PI = a.cin | 1;
  float* sA = g_smem;
  float* sDP = g_smem + (a.Tin * PI + 3) / 4 * 4;
  const int tasks = a.k * a.cin, nq = a.nq;
  const int task = tid % tasks, q = tid / tasks;
  const bool active = q < nq;
  const int j = task / a.cin, ci = task % a.cin;
  float acc[NC];
#pragma unroll
  for (int co = 0; co < NC; ++co) acc[co] = 0.f;
  if (PO != NC) {
    // columns NC..PO-1 of dp stay zero
    for (int i = tid; i < a.Tout * PO; i += kThreads) sDP[i] = 0.f;
  }
  for (int b = bid; b < a.B; b += nb) {
    __syncthreads();
    stage_sources(a.src, a.n_src, b, a.Tin, sA, PI, tid);
    stage_dp(a.y, NC, b, a.Tout, sDP, PO, tid);
    __syncthreads();
    if (active) {
      const float* col = sA + j * a.dil * PI + ci;
      for (int t = q; t < a.Tout; t += nq) {
        const float v = col[t * a.stride * PI];
        const float4* row = reinterpret_cast<const float4*>(sDP + t * PO);
#pragma unroll
        for (int c4 = 0; c4 < PO / 4; ++c4) {
          const float4 d = row[c4];
          if (c4 * 4 + 0 < NC) acc[c4 * 4 + 0] = fmaf(v, d.x, acc[c4 * 4 + 0]);
          if (c4 * 4 + 1 < NC) acc[c4 * 4 + 1] = fmaf(v, d.y, acc[c4 * 4 + 1]);
          if (c4 * 4 + 2 < NC) acc[c4 * 4 + 2] = fmaf(v, d.z, acc[c4 * 4 + 2]);
          if (c4 * 4 + 3 < NC) acc[c4 * 4 + 3] = fmaf(v, d.w, acc[c4 * 4 + 3]);
        }
      }
    }
  }
  // sum over the frame subsets inside the workgroup (fixed order), kGWgChunk output channels at a time through
  // LDS: one partial row per workgroup instead of nq of them (the rows are what grad_reduce_kernel has to read)
  if (nq > 1) {
    float* sQ = g_smem;   // [nq][tasks][kGWgChunk]
#pragma unroll
    for (int c0 = 0; c0 < NC; c0 += kGWgChunk) {
      __syncthreads();
      if (active) {
#pragma unroll
        for (int u = 0; u < kGWgChunk; ++u)
          if (c0 + u < NC) sQ[(q * tasks + task) * kGWgChunk + u] = acc[c0 + u];
      }
      __syncthreads();
      if (active && q == 0) {
#pragma unroll
        for (int u = 0; u < kGWgChunk; ++u)
          if (c0 + u < NC) {
            float v = 0.f;
            for (int qq = 0; qq < nq; ++qq) v += sQ[(qq * tasks + task) * kGWgChunk + u];
            acc[c0 + u] = v;
          }
      }
    }
  }
  if (active && q == 0) {
    float* dst = a.grad_part + (size_t)bid * ((size_t)tasks * NC) + (size_t)task * NC;
#pragma unroll
    for (int co = 0; co < NC; ++co) dst[co] = acc[co];
  }
}

template <int NC>
__global__ __launch_bounds__(kThreads) void gconv_wgrad_kernel(GWgradArgs a) {
  gconv_wgrad_body<NC>(a, blockIdx.x, gridDim.x);
}

// Both halves of an op's backward in one launch: workgroups [0, nb) form the weight gradient, [nb, 2 nb) the data
// gradient.  They are independent (both only read the op's output gradient) and each is latency-bound on its
// own, so sharing the launch hides one of the two.
template <int NCO, int NCI>
__global__ __launch_bounds__(kThreads) void gconv_bwd_kernel(GWgradArgs w, GConvArgs d, int nb) {
  if ((int)blockIdx.x < nb) gconv_wgrad_body<NCO>(w, blockIdx.x, nb);
  else gconv_body<NCI, 1>(d, blockIdx.x - nb, nb);
}

// ... and of twin ops: four roles
template <int NCO, int NCI>
__global__ __launch_bounds__(kThreads) void gconv_bwd2_kernel(GWgradArgs w0, GConvArgs d0, GWgradArgs w1, GConvArgs d1, int nb) {
  const int role = blockIdx.x / nb, bid = blockIdx.x - role * nb;
  if (role == 0) gconv_wgrad_body<NCO>(w0, bid, nb);
  else if (role == 1) gconv_body<NCI, 1>(d0, bid, nb);
  else if (role == 2) gconv_wgrad_body<NCO>(w1, bid, nb);
  else gconv_body<NCI, 1>(d1, bid, nb);
}

// ---------------------------------------------------------------------------------------------
// Depthwise k x 1 convolution ops (MixedNet's MixConv, mixednet.py:168-231, for shapes the specialised
// block kernels do not cover): one source, C channels in and out, taps w[k][C] (multi-kernel MixConv
// groups arrive fused: right-aligned, zero leading taps, gradient mask), bias handled as the op's "shift".
// thread <-> (channel, frame group) everywhere, so reads of a wave are C-contiguous rows.
struct GDwArgs {
  GSrc src;
  const float* w;       // [k][C]
  int k, C, B, Tin, Tout;
  float* out;           // MODE 0: [B][Tout][C]
  GBnBwd y;             // MODE 1 / weight gradient: the op's output gradient (coefficients are the constants 1, 0, 0)
  float* grad_part;     // weight gradient: [grid][k*C]
};

// MODE 0: forward.  MODE 1: data gradient da[s] = sum_j w[j] dp[s-j], scattered into the source's gradient.
template <int MODE>
__global__ __launch_bounds__(kThreads) void gdw_kernel(GDwArgs a) {
  HIP_DYNAMIC_SHARED(float4, g_smem4)
  float* g_smem = reinterpret_cast<float*>(g_smem4);
  __shared__ float sRed[2 * kThreads];
  const int tid = threadIdx.x, C = a.C, PI = C | 1;
  const int pad = MODE == 1 ? a.k - 1 : 0;
  const int rows_in = (MODE == 0 ? a.Tin : a.Tout + 2 * pad);
  float* sW = g_smem;               // [k][C]
  float* sIn = sW + a.k * C;        // MODE 0: activated source rows; MODE 1: zero-padded dp rows
  const int nrg = kThreads / C, c = tid % C, rg = tid / C;
  float s1 = 0.f, s2 = 0.f;
  for (int i = tid; i < a.k * C; i += kThreads) sW[i] = a.w[i];
  if (MODE == 1)
    for (int i = tid; i < pad * PI; i += kThreads) {
      sIn[i] = 0.f;
      sIn[(pad + a.Tout) * PI + i] = 0.f;
    }
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    __syncthreads();
    if (MODE == 0) stage_sources(&a.src, 1, b, a.Tin, sIn, PI, tid);
    else stage_dp(a.y, C, b, a.Tout, sIn + pad * PI, PI, tid);
    __syncthreads();
    if (rg < nrg) {
      if (MODE == 0) {
        float* dst = a.out + (size_t)b * a.Tout * C + c;
        for (int t = rg; t < a.Tout; t += nrg) {
          float acc = 0.f;
          for (int j = 0; j < a.k; ++j) acc = fmaf(sW[j * C + c], sIn[(t + j) * PI + c], acc);
          dst[(size_t)t * C] = acc;
        }
      } else {
        const GSrc& s = a.src;
        const float sc = s.scale[s.c0 + c], sh = s.shift[s.c0 + c];
        const bool accum = (s.flags & GSRC_ACCUM) != 0, stats = (s.flags & GSRC_STATS) != 0, linear = (s.flags & GSRC_LINEAR) != 0;
        const float mu = stats ? s.mean[s.c0 + c] : 0.f, rs = stats ? s.rstd[s.c0 + c] : 0.f;
        const size_t base = (size_t)b * s.T * s.ld + s.c0 + c;
        const float rsc = s.rp ? s.rscale[c] : 0.f, rsh = s.rp ? s.rshift[c] : 0.f;
        const float* rbase = s.rp ? s.rp + ((size_t)b * s.rT + s.rdrop) * C + c : nullptr;
        for (int t = rg; t < s.T; t += nrg) {
          const size_t idx = base + (size_t)t * s.ld;
          const float p = s.p[idx];
          const float rv = rbase ? rbase[(size_t)t * C] : 0.f;
          const int r = t - s.toff;   // frame of the (aligned) input; da[r] = sum_j w[j] dp[r - j]
          float gv = 0.f;
          if (r >= 0 && r < a.Tin && (linear || src_affine(s, p, sc, sh, rv, rsc, rsh) > 0.f)) {
            float acc = 0.f;
            for (int j = 0; j < a.k; ++j) acc = fmaf(sW[j * C + c], sIn[(r - j + pad) * PI + c], acc);
            gv = acc;
          }
          if (accum) gv += s.g[idx];
          s.g[idx] = gv;
          s1 += gv;
          s2 = fmaf(gv, (p - mu) * rs, s2);
        }
      }
    }
  }
  if (MODE == 1 && (a.src.flags & GSRC_STATS))
    write_channel_partials(s1, s2, C, sRed, a.src.gstat_part + (size_t)blockIdx.x * 2 * a.src.ld + a.src.c0, tid, a.src.ld);
}

// dw[j][c] = sum_{b,t} act[b][t+j][c] * dp[b][t][c]; task (j, c) -> thread (task % 256), up to kGDwTasks per thread
constexpr int kGDwTasks = 8;
__global__ __launch_bounds__(kThreads) void gdw_wgrad_kernel(GDwArgs a) {
  HIP_DYNAMIC_SHARED(float4, g_smem4)
  float* g_smem = reinterpret_cast<float*>(g_smem4);
  const int tid = threadIdx.x, C = a.C, PI = C | 1;
  float* sA = g_smem;
  float* sDP = g_smem + a.Tin * PI;
  const int tasks = a.k * C;
  float acc[kGDwTasks];
#pragma unroll
  for (int u = 0; u < kGDwTasks; ++u) acc[u] = 0.f;
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    __syncthreads();
    stage_sources(&a.src, 1, b, a.Tin, sA, PI, tid);
    stage_dp(a.y, C, b, a.Tout, sDP, PI, tid);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kGDwTasks; ++u) {
      const int task = tid + u * kThreads;
      if (task < tasks) {
        const int j = task / C, c = task - j * C;
        float v = acc[u];
        for (int t = 0; t < a.Tout; ++t) v = fmaf(sA[(t + j) * PI + c], sDP[t * PI + c], v);
        acc[u] = v;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < kGDwTasks; ++u) {
    const int task = tid + u * kThreads;
    if (task < tasks) a.grad_part[(size_t)blockIdx.x * tasks + task] = acc[u];
  }
}

// Gradient of a residual op R: the ops that add it (one per repeat of the block) already hold the gradient at
// their own outputs, masked by their ReLU; R's gradient is their sum, frame t of adder X landing on frame
// t + drop_X of R.  Also emits R's BN-backward statistics partials.
constexpr int kGMaxAdders = 8;
struct GResGatherArgs {
  const float* gx[kGMaxAdders];   // [B][Tx][C] gradients of the adders
  int Tx[kGMaxAdders], drop[kGMaxAdders];
  int n;
  const float* p;                 // R's pre-BN output [B][T][C]
  const float *mean, *rstd;
  float* g;                       // [B][T][C]
  float* gstat_part;              // [grid][2][C]
  int B, T, C;
};
__global__ __launch_bounds__(kThreads) void gres_gather_kernel(GResGatherArgs a) {
  __shared__ float sRed[2 * kThreads];
  const int tid = threadIdx.x, C = a.C, nrg = kThreads / C, c = tid % C, rg = tid / C;
  float s1 = 0.f, s2 = 0.f;
  if (rg < nrg) {
    const float mu = a.mean[c], rs = a.rstd[c];
    for (int b = blockIdx.x; b < a.B; b += gridDim.x)
      for (int t = rg; t < a.T; t += nrg) {
        float gv = 0.f;
        for (int i = 0; i < a.n; ++i) {
          const int tx = t - a.drop[i];
          if (tx >= 0 && tx < a.Tx[i]) gv += a.gx[i][((size_t)b * a.Tx[i] + tx) * C + c];
        }
        const size_t idx = ((size_t)b * a.T + t) * C + c;
        a.g[idx] = gv;
        s1 += gv;
        s2 = fmaf(gv, (a.p[idx] - mu) * rs, s2);
      }
  }
  write_channel_partials(s1, s2, C, sRed, a.gstat_part + (size_t)blockIdx.x * 2 * C, tid, C);
}

// W[k][cin][cout] -> WT[k][cout][cin] with reversed taps, for every op that needs a data gradient
struct GTransposeItem { int src, dst, k, cin, cout; };
constexpr int kGMaxOps = 48;
struct GTransposeArgs {
  GTransposeItem item[kGMaxOps];
  const float* params;
  float* wt;
};
// grid = (ceil(max n / 256), items)
__global__ __launch_bounds__(kThreads) void gweights_transpose_kernel(GTransposeArgs a) {
  const GTransposeItem it = a.item[blockIdx.y];
  const int e = blockIdx.x * kThreads + threadIdx.x;
  if (e >= it.k * it.cin * it.cout) return;
  const int ci = e % it.cin, co = (e / it.cin) % it.cout, j = e / (it.cin * it.cout);
  a.wt[it.dst + e] = a.params[it.src + ((it.k - 1 - j) * it.cin + ci) * it.cout + co];
}

// ---------------------------------------------------------------------------------------------
// BN / SSN finalize: one workgroup per slot (slot s owns channels s, s+g, s+2g, ... when g > 1)
__device__ __forceinline__ void block_sum2(double& t1, double& t2, double* sAcc, int tid) {
  sAcc[tid] = t1;
  sAcc[kThreads + tid] = t2;
  __syncthreads();
  for (int w = kThreads / 2; w > 0; w >>= 1) {
    if (tid < w) {
      sAcc[tid] += sAcc[tid + w];
      sAcc[kThreads + tid] += sAcc[kThreads + tid + w];
    }
    __syncthreads();
  }
  t1 = sAcc[0];
  t2 = sAcc[kThreads];
}

struct GBnFwdArgs {
  const float* stat_part;   // [G][2][C]
  int G, C, groups;
  float inv_n;              // 1 / (B * T * channels per slot)
  const float *gamma, *beta;   // [slots]
  float *moving_mean, *moving_var;
  float *scale, *shift, *mean, *rstd;   // [C] expanded per channel
  int update_moving;
};

__device__ __forceinline__ void gbn_fwd_finalize_body(const GBnFwdArgs& a, int slot, double* sAcc) {
  const int tid = threadIdx.x;
  const int members = a.groups > 1 ? a.C / a.groups : 1, cstride = a.groups > 1 ? a.groups : 0;
  double t1 = 0.0, t2 = 0.0;
  for (int it = tid; it < a.G * members; it += kThreads) {
    const int jj = it / members, c = slot + (it % members) * cstride;
    t1 += (double)a.stat_part[(size_t)jj * 2 * a.C + c];
    t2 += (double)a.stat_part[(size_t)jj * 2 * a.C + a.C + c];
  }
  block_sum2(t1, t2, sAcc, tid);
  const double m = t1 * (double)a.inv_n;
  double var = t2 * (double)a.inv_n - m * m;   // biased batch variance
  if (var < 0.0) var = 0.0;
  const float meanf = (float)m, varf = (float)var;
  const float rstd = 1.0f / sqrtf(varf + kBnEps);
  const float sc = a.gamma[slot] * rstd, sh = a.beta[slot] - meanf * sc;
  if (tid < members) {
    const int c = slot + tid * cstride;
    a.scale[c] = sc;
    a.shift[c] = sh;
    a.mean[c] = meanf;
    a.rstd[c] = rstd;
  }
  if (tid == 0 && a.update_moving) {
    a.moving_mean[slot] = a.moving_mean[slot] * kBnMomentum + meanf * (1.0f - kBnMomentum);
    a.moving_var[slot] = a.moving_var[slot] * kBnMomentum + varf * (1.0f - kBnMomentum);
  }
}

__global__ __launch_bounds__(kThreads) void gbn_fwd_finalize_kernel(GBnFwdArgs a) {
  __shared__ double sAcc[2 * kThreads];
  gbn_fwd_finalize_body(a, blockIdx.x, sAcc);
}
// twin ops: slots of a0 first, then those of a1
__global__ __launch_bounds__(kThreads) void gbn_fwd_finalize2_kernel(GBnFwdArgs a0, GBnFwdArgs a1, int n0) {
  __shared__ double sAcc[2 * kThreads];
  if ((int)blockIdx.x < n0) gbn_fwd_finalize_body(a0, blockIdx.x, sAcc);
  else gbn_fwd_finalize_body(a1, blockIdx.x - n0, sAcc);
}

struct GBnEvalArgs {
  const float *gamma, *beta, *moving_mean, *moving_var;
  float *scale, *shift;
  int C, groups;
};
__global__ __launch_bounds__(kThreads) void gbn_eval_prepare_kernel(GBnEvalArgs a) {
  for (int c = threadIdx.x; c < a.C; c += kThreads) {
    const int slot = a.groups > 1 ? c % a.groups : c;
    const float sc = a.gamma[slot] / sqrtf(a.moving_var[slot] + kBnEps);
    a.scale[c] = sc;
    a.shift[c] = a.beta[slot] - a.moving_mean[slot] * sc;
  }
}

struct GBnBwdArgs {
  const float* gstat_part;   // [G][2][C]
  int G, C, groups;
  float inv_n;
  const float* gamma;        // [slots]
  const float* rstd;         // [C]
  float *c1, *mg, *mgx;      // [C]
  float *dgamma, *dbeta;     // [slots] -> flat gradient
  float dscale;              // 1, or 1/W when the sums were all-reduced (see BnBwdFinalizeArgs)
  int bias_only;             // the op has a bias instead of a BN (depthwise convolution): only dbeta = sum g is needed,
                             // the backward coefficients are the constants c1 = 1, mg = mgx = 0
};
__device__ __forceinline__ void gbn_bwd_finalize_body(const GBnBwdArgs& a, int slot, double* sAcc) {
  const int tid = threadIdx.x;
  const int members = a.groups > 1 ? a.C / a.groups : 1, cstride = a.groups > 1 ? a.groups : 0;
  double t1 = 0.0, t2 = 0.0;
  for (int it = tid; it < a.G * members; it += kThreads) {
    const int jj = it / members, c = slot + (it % members) * cstride;
    t1 += (double)a.gstat_part[(size_t)jj * 2 * a.C + c];
    t2 += (double)a.gstat_part[(size_t)jj * 2 * a.C + a.C + c];
  }
  block_sum2(t1, t2, sAcc, tid);
  if (a.bias_only) {
    if (tid == 0) a.dbeta[slot] = (float)t1 * a.dscale;
    return;
  }
  if (tid < members) {
    const int c = slot + tid * cstride;
    a.c1[c] = a.gamma[slot] * a.rstd[c];
    a.mg[c] = (float)(t1 * (double)a.inv_n);
    a.mgx[c] = (float)(t2 * (double)a.inv_n);
  }
  if (tid == 0) {
    a.dbeta[slot] = (float)t1 * a.dscale;
    a.dgamma[slot] = (float)t2 * a.dscale;
  }
}

__global__ __launch_bounds__(kThreads) void gbn_bwd_finalize_kernel(GBnBwdArgs a) {
  __shared__ double sAcc[2 * kThreads];
  gbn_bwd_finalize_body(a, blockIdx.x, sAcc);
}
__global__ __launch_bounds__(kThreads) void gbn_bwd_finalize2_kernel(GBnBwdArgs a0, GBnBwdArgs a1, int n0) {
  __shared__ double sAcc[2 * kThreads];
  if ((int)blockIdx.x < n0) gbn_bwd_finalize_body(a0, blockIdx.x, sAcc);
  else gbn_bwd_finalize_body(a1, blockIdx.x - n0, sAcc);
}

// ---------------------------------------------------------------------------------------------
// head: last op -> BN + ReLU -> Flatten -> Dropout -> Dense(1) -> sigmoid (inception.py:330-338)
struct GHeadArgs {
  const float* p;          // [B][T][C]
  const float *scale, *shift, *mean, *rstd;
  const float* wd;         // [T*C]
  const float* bd;
  const float* y;
  const float* sw;
  const float* keep;       // [B][T*C] 0 or 1/(1-rate); null = no dropout
  float *z, *prob, *dz, *loss_part;
  float* g;                // [B][T][C] gradient at the BN output of the last op
  float* gstat_part;       // [grid][2][C]
  int B, T, C;
  float inv_b;
  int training;
  const float *rp, *rscale, *rshift;   // residual branch of the last op (see GSrc), or null
  int rT, rdrop;
  // keep_gen set: this step's dropout mask is generated here (and written to keep_gen for the dense-weight gradient)
  float* keep_gen;
  unsigned long long seed;
  const unsigned* counter;   // [2] low / high word of this step's counter (mapped mailbox)
  float rate;
};

// Dropout keep value of element e in step `step`: counter-based hash of (seed, step, element) -> 0 or 1/(1-rate).
// (Keras draws its mask from a stateful generator that is not reproducible across frameworks; the parity tests
// inject an explicit mask instead.)
__device__ __forceinline__ float dropout_keep(unsigned long long seed, unsigned long long step, unsigned long long e, float rate) {
  unsigned long long h = seed * 0x9E3779B97F4A7C15ull + step * 0xD1B54A32D192ED03ull + e;
  h ^= h >> 30; h *= 0xBF58476D1CE4E5B9ull;
  h ^= h >> 27; h *= 0x94D049BB133111EBull;
  h ^= h >> 31;
  const float u = (float)(h >> 40) * (1.0f / 16777216.0f);
  return u >= rate ? 1.0f / (1.0f - rate) : 0.f;
}

__global__ __launch_bounds__(kThreads) void ghead_kernel(GHeadArgs a) {
  __shared__ float sRed[8];
  __shared__ float sBcast[2];
  __shared__ float sStat[2 * kThreads];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = a.C, nrg = kThreads / C, c = tid % C, rg = tid / C;
  const bool active = rg < nrg;
  const int n = a.T * C;
  const float sc = active ? a.scale[c] : 0.f, sh = active ? a.shift[c] : 0.f;
  const float mu = (active && (a.training & kHeadTraining)) ? a.mean[c] : 0.f, rs = (active && (a.training & kHeadTraining)) ? a.rstd[c] : 0.f;
  const float rsc = (active && a.rp) ? a.rscale[c] : 0.f, rsh = (active && a.rp) ? a.rshift[c] : 0.f;
  const float bias = a.bd[0];
  float g1 = 0.f, g2 = 0.f;
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    const float* pb = a.p + (size_t)b * n;
    const float* kb = a.keep ? a.keep + (size_t)b * n : nullptr;
    const float* rb = a.rp ? a.rp + ((size_t)b * a.rT + a.rdrop) * C : nullptr;
    float* kgen = a.keep_gen ? a.keep_gen + (size_t)b * n : nullptr;
    float dot = 0.f;
    // rows in batches of kHB per thread: every load of a batch is issued before the first use (one load, one wait per
    // row made this kernel a chain of ~2 T C / 256 memory round trips per window: 51 us per launch for 10 MB in round 2)
    constexpr int kHB = 8;
    if (active) {
      const unsigned long long step = kgen ? (((unsigned long long)a.counter[1] << 32) | a.counter[0]) : 0ull;
      for (int t0 = rg; t0 < a.T; t0 += kHB * nrg) {
        float pv[kHB], wv[kHB], rv[kHB], kv[kHB];
#pragma unroll
        for (int u = 0; u < kHB; ++u) {
          const int t = t0 + u * nrg;
          const bool ok = t < a.T;
          const int i = ok ? t * C + c : c;
          pv[u] = ok ? pb[i] : 0.f;
          wv[u] = ok ? a.wd[i] : 0.f;
          rv[u] = (ok && rb) ? rb[i] : 0.f;
          kv[u] = (ok && kb && !kgen) ? kb[i] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < kHB; ++u) {
          const int t = t0 + u * nrg;
          if (t < a.T) {
            const int i = t * C + c;
            const float act = fmaxf(fmaf(pv[u], sc, sh) + (rb ? fmaf(rv[u], rsc, rsh) : 0.f), 0.f);
            float k1 = kv[u];
            if (kgen) {
              k1 = dropout_keep(a.seed, step, (unsigned long long)b * n + i, a.rate);
              kgen[i] = k1;   // read back by this thread in the backward part below and by the dense-weight gradient
            }
            dot = fmaf((kgen || kb) ? act * k1 : act, wv[u], dot);
          }
        }
      }
    }
    if (kgen) kb = kgen;
    dot = wave_sum(dot);
    if (lane == 0) sRed[wave] = dot;
    __syncthreads();
    if (tid == 0) {
      const float zz = ((sRed[0] + sRed[1]) + (sRed[2] + sRed[3])) + bias;
      const float pr = 1.0f / (1.0f + expf(-zz));
      a.z[b] = zz;
      a.prob[b] = pr;
      float dzz = 0.f;
      if (a.y != nullptr) {
        const float yy = a.y[b];
        const bool clipped_form = (a.training & kHeadClippedLoss) != 0;
        const float bce = bce_value(zz, pr, yy, clipped_form);
        if (a.training & kHeadTraining) {
          const float w = a.sw[b];
          a.loss_part[b] = w * bce * a.inv_b;
          dzz = w * bce_dz(pr, yy, clipped_form) * a.inv_b;
          a.dz[b] = dzz;
        }
      }
      sBcast[0] = dzz;
    }
    __syncthreads();
    if ((a.training & kHeadTraining) && active) {
      const float dzz = sBcast[0];
      float* gb = a.g + (size_t)b * n;
      for (int t0 = rg; t0 < a.T; t0 += kHB * nrg) {
        float pv[kHB], wv[kHB], rv[kHB], kv[kHB];
#pragma unroll
        for (int u = 0; u < kHB; ++u) {
          const int t = t0 + u * nrg;
          const bool ok = t < a.T;
          const int i = ok ? t * C + c : c;
          pv[u] = ok ? pb[i] : 0.f;
          wv[u] = ok ? a.wd[i] : 0.f;
          rv[u] = (ok && rb) ? rb[i] : 0.f;
          kv[u] = (ok && kb) ? kb[i] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < kHB; ++u) {
          const int t = t0 + u * nrg;
          if (t < a.T) {
            const int i = t * C + c;
            const float raw = pv[u];
            float gv = (fmaf(raw, sc, sh) + (rb ? fmaf(rv[u], rsc, rsh) : 0.f)) > 0.f ? dzz * wv[u] : 0.f;
            if (kb) gv *= kv[u];
            gb[i] = gv;
            g1 += gv;
            g2 = fmaf(gv, (raw - mu) * rs, g2);
          }
        }
      }
    }
  }
  if (a.training & kHeadTraining) write_channel_partials(g1, g2, C, sStat, a.gstat_part + (size_t)blockIdx.x * 2 * C, tid, C);
}

// ---------------------------------------------------------------------------------------------
// MixedNet's optional heads (mixednet.py:234-275 SpatialAttention, :362-384): on the last op's activations a
//   attention : per frame mean and max over the channels -> Conv2D(1, (4,1), valid, no bias, sigmoid) over time
//               -> gate s[t'] for the LAST T-3 frames:  out[t'][c] = a[t'+3][c] * s[t']
//   pooling   : average or max of out over all remaining frames -> [C]     (else Flatten of out)
//   Dense(1, sigmoid), Keras BCE, and the complete backward of the above down to the gradient at the last
//   op's BN output.  One workgroup per window with the window's activations in LDS.  What the dense layer
//   sees is written to hact so that dense_grad_kernel can form the dense-weight gradient from it.
struct GHead2Args {
  GHeadArgs h;
  const float* watt;     // [4][2] attention taps (tap j: weight of the mean, weight of the max); null = no attention
  int pool;              // 0 Flatten, 1 average, 2 max
  float* hact;           // [B][n_dense]
  float* watt_part;      // [grid][8]
};

__global__ __launch_bounds__(kThreads) void ghead_att_kernel(GHead2Args a) {
  HIP_DYNAMIC_SHARED(float4, g_smem4)
  float* sm = reinterpret_cast<float*>(g_smem4);
  __shared__ float sStat[2 * kThreads];
  __shared__ float sRed[8];
  __shared__ float sBcast[2];
  const GHeadArgs& h = a.h;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = h.C, T = h.T, PA = C | 1;
  const bool att = a.watt != nullptr;
  const int Toff = att ? 3 : 0, To = T - Toff;
  const int nd = a.pool ? C : To * C;
  float* sA = sm;
  float* sAvg = sA + T * PA;
  float* sMax = sAvg + T;
  float* sDavg = sMax + T;
  float* sDmx = sDavg + T;
  float* sS = sDmx + T;
  float* sDpre = sS + T;
  int* sArg = reinterpret_cast<int*>(sDpre + T);
  float* sV = reinterpret_cast<float*>(sArg + T);
  int* sArgT = reinterpret_cast<int*>(sV + C);
  const int nrg = kThreads / C, c = tid % C, rg = tid / C;
  const bool active = rg < nrg;
  const float sc = active ? h.scale[c] : 0.f, sh = active ? h.shift[c] : 0.f;
  const float mu = (active && (h.training & kHeadTraining)) ? h.mean[c] : 0.f, rs = (active && (h.training & kHeadTraining)) ? h.rstd[c] : 0.f;
  const float rsc = (active && h.rp) ? h.rscale[c] : 0.f, rsh = (active && h.rp) ? h.rshift[c] : 0.f;
  float w8[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) w8[i] = att ? a.watt[i] : 0.f;
  const float bias = h.bd[0];
  float g1 = 0.f, g2 = 0.f, wacc = 0.f;
  // gradient of the dense layer's input element (t', cc) per unit of dL/dz
  auto dout = [&](int tp, int cc) -> float {
    if (a.pool == 0) return h.wd[tp * C + cc];
    if (a.pool == 1) return h.wd[cc] / (float)To;
    return sArgT[cc] == tp ? h.wd[cc] : 0.f;
  };
  for (int b = blockIdx.x; b < h.B; b += gridDim.x) {
    const float* pb = h.p + (size_t)b * T * C;
    const float* rb = h.rp ? h.rp + ((size_t)b * h.rT + h.rdrop) * C : nullptr;
    __syncthreads();
    if (active)
      for (int t = rg; t < T; t += nrg) {
        const int i = t * C + c;
        sA[t * PA + c] = fmaxf(fmaf(pb[i], sc, sh) + (rb ? fmaf(rb[i], rsc, rsh) : 0.f), 0.f);
      }
    __syncthreads();
    if (att) {
      for (int t = tid; t < T; t += kThreads) {
        float sum = 0.f, mx = -3.0e38f;
        int arg = 0;
        for (int cc = 0; cc < C; ++cc) {
          const float v = sA[t * PA + cc];
          sum += v;
          if (v > mx) { mx = v; arg = cc; }
        }
        sAvg[t] = sum / (float)C;
        sMax[t] = mx;
        sArg[t] = arg;
      }
      __syncthreads();
      for (int t = tid; t < To; t += kThreads) {
        float pre = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) pre += w8[2 * j] * sAvg[t + j] + w8[2 * j + 1] * sMax[t + j];
        sS[t] = 1.0f / (1.0f + expf(-pre));
      }
      __syncthreads();
    }
    float dot = 0.f;
    if (a.pool) {
      if (tid < C) {
        float acc = a.pool == 1 ? 0.f : -3.0e38f;
        int arg = 0;
        for (int t = 0; t < To; ++t) {
          const float v = sA[(t + Toff) * PA + tid] * (att ? sS[t] : 1.f);
          if (a.pool == 1) acc += v;
          else if (v > acc) { acc = v; arg = t; }
        }
        if (a.pool == 1) acc /= (float)To;
        sV[tid] = acc;
        sArgT[tid] = arg;
        dot = acc * h.wd[tid];
        if (h.training & kHeadTraining) a.hact[(size_t)b * nd + tid] = acc;
      }
    } else if (active) {
      for (int t = rg; t < To; t += nrg) {
        const float v = sA[(t + Toff) * PA + c] * (att ? sS[t] : 1.f);
        dot = fmaf(v, h.wd[t * C + c], dot);
        if (h.training & kHeadTraining) a.hact[(size_t)b * nd + t * C + c] = v;
      }
    }
    dot = wave_sum(dot);
    if (lane == 0) sRed[wave] = dot;
    __syncthreads();
    if (tid == 0) {
      const float zz = ((sRed[0] + sRed[1]) + (sRed[2] + sRed[3])) + bias;
      const float pr = 1.0f / (1.0f + expf(-zz));
      h.z[b] = zz;
      h.prob[b] = pr;
      float dzz = 0.f;
      if (h.y != nullptr) {
        const float yy = h.y[b];
        const bool clipped_form = (h.training & kHeadClippedLoss) != 0;
        const float bce = bce_value(zz, pr, yy, clipped_form);
        if (h.training & kHeadTraining) {
          const float w = h.sw[b];
          h.loss_part[b] = w * bce * h.inv_b;
          dzz = w * bce_dz(pr, yy, clipped_form) * h.inv_b;
          h.dz[b] = dzz;
        }
      }
      sBcast[0] = dzz;
    }
    __syncthreads();
    if (!(h.training & kHeadTraining)) continue;
    const float dzz = sBcast[0];
    if (att) {
      for (int t = tid; t < To; t += kThreads) {
        float ds = 0.f;
        for (int cc = 0; cc < C; ++cc) ds = fmaf(dout(t, cc), sA[(t + 3) * PA + cc], ds);
        const float s = sS[t];
        sDpre[t] = dzz * ds * s * (1.0f - s);
      }
      __syncthreads();
      if (tid < 8) {
        const int j = tid >> 1;
        const float* src = (tid & 1) ? sMax : sAvg;
        float acc = 0.f;
        for (int t = 0; t < To; ++t) acc = fmaf(sDpre[t], src[t + j], acc);
        wacc += acc;
      }
      for (int t = tid; t < T; t += kThreads) {
        float da = 0.f, dm = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int tp = t - j;
          if (tp >= 0 && tp < To) {
            da = fmaf(sDpre[tp], w8[2 * j], da);
            dm = fmaf(sDpre[tp], w8[2 * j + 1], dm);
          }
        }
        sDavg[t] = da / (float)C;
        sDmx[t] = dm;
      }
      __syncthreads();
    }
    if (active) {
      float* gb = h.g + (size_t)b * T * C;
      for (int t = rg; t < T; t += nrg) {
        const int tp = t - Toff;
        float da = tp >= 0 ? dzz * dout(tp, c) * (att ? sS[tp] : 1.f) : 0.f;
        if (att) da += sDavg[t] + (sArg[t] == c ? sDmx[t] : 0.f);
        const float gv = sA[t * PA + c] > 0.f ? da : 0.f;
        gb[t * C + c] = gv;
        g1 += gv;
        g2 = fmaf(gv, (pb[t * C + c] - mu) * rs, g2);
      }
    }
  }
  if (h.training & kHeadTraining) {
    write_channel_partials(g1, g2, C, sStat, h.gstat_part + (size_t)blockIdx.x * 2 * C, tid, C);
    if (att && tid < 8) a.watt_part[(size_t)blockIdx.x * 8 + tid] = wacc;
  }
}

// Dropout keep-mask of one step as its own launch (the attention / pooled heads; ghead_kernel generates it inline)
struct DropoutMaskArgs {
  float* keep;
  long long n;
  unsigned long long seed;
  const unsigned* counter;   // [2] low / high word of this step's counter (mapped mailbox)
  float rate;
};
__global__ __launch_bounds__(kThreads) void dropout_mask_kernel(DropoutMaskArgs a) {
  const long long e = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (e >= a.n) return;
  const unsigned long long step = ((unsigned long long)a.counter[1] << 32) | a.counter[0];
  a.keep[e] = dropout_keep(a.seed, step, (unsigned long long)e, a.rate);
}

}  // namespace mww

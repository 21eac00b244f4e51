// Classifier head of MixedNet (reference microwakeword/mixednet.py:383-384 after :352,:360 of the
// last block) and the loss of train.py:206,295-299:
//
//   head_kernel       : one workgroup per window.  p_L -> BN_L + ReLU -> Flatten -> Dense(1) -> sigmoid,
//                       Keras BCE (probabilities clipped to [1e-7, 1-1e-7]) times the per-sample
//                       weight, dL/dz, the (sum g, sum g*xhat) partials of BN_L's backward (g rebuilt
//                       from the per-sample scalar dL/dz).  The window's 7104 activations stay in registers between
//                       the dot product and the backward partials: p_L is read from HBM once.
//   dense_grad_kernel : dW_dense[t,c] = sum_b dL/dz_b * relu(bn(p_L[b,t,c])) as a batch-chunked
//                       reduction (fixed order) + the dense bias gradient.
#pragma once
#include <type_traits>

#include "common.hip.h"

namespace mww {

// INVARIANT: the cumulative state has ONE writer at a time - a single metric workgroup per launch (metrics_kernel, or the
// metric role of head_tail_kernel / grad_final_kernel), launches ordered on the context's stream.  metrics_body reads the
// counters at its start and stores them back at its end with plain accesses (no atomics): two metric workgroups in flight
// - a second stream, two launches of a step carrying do_metrics - would lose counts silently.  The host asserts the
// per-step half (mww_lib.hip: at most one launch of a step carries the metric role).
struct MetricState {            // device-resident cumulative metric counters (train.py:209-221)
  unsigned long long hist101[2][101];
  unsigned long long hist200[2][200];
  unsigned long long n, correct, tp5, fp5, fn5, pos, neg;
  double bce_sum;
};

struct HeadArgs {
  const float* p;          // p_L [B][T][C]
  const float* scale;      // BN_L folded
  const float* shift;
  const float* mean;       // BN_L batch mean / rstd (training only)
  const float* rstd;
  const float* wd;         // [T*C]
  const float* bd;         // [1]
  const float* y;          // [B] labels (training / metrics) or null
  const float* sw;         // [B] per-sample weight (penalty * class weight)
  float* z;                // [B] logits
  float* prob;             // [B]
  float* dz;               // [B] dL/dz (training)
  float* loss_part;        // [B] weighted loss / B per sample (training)
  float* gstat_part;       // [gridDim.x][2][C]   sum g, sum g*xhat of the BN_L input gradient
  int B, T;
  float inv_b;
  int training;            // kHeadTraining | kHeadClippedLoss
  BnFoldArgs fold;         // fold.acc set: BN_L's scale / shift / mean / rstd are folded here from the accumulator rows
  StatAcc gacc;            // gacc.acc set: the (sum g, sum g*xhat) partials go to accumulator rows instead of gstat_part
};

template <int C, int JMAX, bool SB = false>
__global__ __launch_bounds__(kThreads, 2) void head_kernel(HeadArgs a) {
  constexpr int Q = C / 4;                 // float4 per frame
  constexpr int NRG = kThreads / Q;        // frame groups
  __shared__ __attribute__((aligned(16))) float sRed[8];
  __shared__ __attribute__((aligned(16))) float sBcast[2];
  __shared__ __attribute__((aligned(16))) float sStat[NRG * 2 * C];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = tid % Q, rg = tid / Q;
  const bool active = rg < NRG;
  // The dense kernel rows of this thread are the same for every window: loaded once.  A window's rows are fetched while
  // the previous window is reduced (its label / weight too: thread 0 used to load them after the reduction, a second
  // round trip per window on the critical path).  Rows past T read as zeros through the buffer resource.  The dense rows
  // and the first window are requested before BN_L's statistics are folded (they do not depend on them: one round trip
  // less in front of the first window).
  float4 wdv[JMAX];
  {
    const BufRsrc wr = tile_rsrc(a.wd, active ? a.T * C * 4 : 0);
#pragma unroll
    for (int j = 0; j < JMAX; ++j) wdv[j] = tile_load4(wr, ((rg + NRG * j) * C + q * 4) * 4);
  }
  auto fetch = [&](int b, float4 (&dst)[JMAX], float& yy, float& ww) {
    const bool ok = b < a.B;
    const BufRsrc pr = tile_rsrc(elem_ptr<SB>(a.p, (size_t)(ok ? b : 0) * a.T * C), (ok && active) ? a.T * C * elem_bytes(SB) : 0);
#pragma unroll
    for (int j = 0; j < JMAX; ++j) dst[j] = tile_load4s<SB, MWW_AUX_LD_HP>(pr, (rg + NRG * j) * Q + q);
    yy = (ok && a.y != nullptr) ? a.y[b] : 0.f;
    ww = (ok && a.y != nullptr && (a.training & kHeadTraining)) ? a.sw[b] : 0.f;
  };
  float4 raw[JMAX], nxt[JMAX];
  float y_cur = 0.f, w_cur = 0.f, y_nxt = 0.f, w_nxt = 0.f;
  fetch(blockIdx.x, raw, y_cur, w_cur);
  float4 sc = make_float4(0, 0, 0, 0), sh = sc, mu = sc, rs = sc;
  if (a.fold.acc) {
    float* sFold = sStat;   // [4][C], free until the first window's partials
    if (tid < C) bn_fold_channel(a.fold, C, tid, sFold[tid], sFold[C + tid], sFold[2 * C + tid], sFold[3 * C + tid]);
    __syncthreads();
    if (active) {
      sc = *reinterpret_cast<const float4*>(sFold + q * 4);
      sh = *reinterpret_cast<const float4*>(sFold + C + q * 4);
      mu = *reinterpret_cast<const float4*>(sFold + 2 * C + q * 4);
      rs = *reinterpret_cast<const float4*>(sFold + 3 * C + q * 4);
    }
    __syncthreads();
  } else if (active) {
    sc = *reinterpret_cast<const float4*>(a.scale + q * 4);
    sh = *reinterpret_cast<const float4*>(a.shift + q * 4);
    if (a.training & kHeadTraining) {
      mu = *reinterpret_cast<const float4*>(a.mean + q * 4);
      rs = *reinterpret_cast<const float4*>(a.rstd + q * 4);
    }
  }
  const float bias = a.bd[0];
  float4 g1 = make_float4(0, 0, 0, 0), g2 = g1;

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    float dot = 0.f;
    fetch(b + gridDim.x, nxt, y_nxt, w_nxt);
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
      // rows past T hold raw = 0 and wd = 0: relu(shift)*0 contributes nothing
      dot = fmaf(fmaxf(fmaf(raw[j].x, sc.x, sh.x), 0.f), wdv[j].x, dot);
      dot = fmaf(fmaxf(fmaf(raw[j].y, sc.y, sh.y), 0.f), wdv[j].y, dot);
      dot = fmaf(fmaxf(fmaf(raw[j].z, sc.z, sh.z), 0.f), wdv[j].z, dot);
      dot = fmaf(fmaxf(fmaf(raw[j].w, sc.w, sh.w), 0.f), wdv[j].w, dot);
    }
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
        const float yy = y_cur;
        const bool clipped_form = (a.training & kHeadClippedLoss) != 0;
        const float bce = bce_value(zz, pr, yy, clipped_form);
        if (a.training & kHeadTraining) {
          const float w = w_cur;
          a.loss_part[b] = w * bce * a.inv_b;
          dzz = w * bce_dz(pr, yy, clipped_form) * a.inv_b;
          a.dz[b] = dzz;
        }
      }
      sBcast[0] = dzz;
    }
    __syncthreads();
    if (a.training & kHeadTraining) {
      const float dzz = sBcast[0];
#pragma unroll
      for (int j = 0; j < JMAX; ++j) {
        // gradient entering BN_L: g = dz * wd * relu'(.) ; partial sums of g and g*xhat
        const float gx = fmaf(raw[j].x, sc.x, sh.x) > 0.f ? dzz * wdv[j].x : 0.f;
        const float gy = fmaf(raw[j].y, sc.y, sh.y) > 0.f ? dzz * wdv[j].y : 0.f;
        const float gz = fmaf(raw[j].z, sc.z, sh.z) > 0.f ? dzz * wdv[j].z : 0.f;
        const float gw = fmaf(raw[j].w, sc.w, sh.w) > 0.f ? dzz * wdv[j].w : 0.f;
        g1.x += gx; g1.y += gy; g1.z += gz; g1.w += gw;
        g2.x = fmaf(gx, (raw[j].x - mu.x) * rs.x, g2.x);
        g2.y = fmaf(gy, (raw[j].y - mu.y) * rs.y, g2.y);
        g2.z = fmaf(gz, (raw[j].z - mu.z) * rs.z, g2.z);
        g2.w = fmaf(gw, (raw[j].w - mu.w) * rs.w, g2.w);
      }
    }
    // sRed / sBcast are rewritten only after the next window's first barrier
#pragma unroll
    for (int j = 0; j < JMAX; ++j) raw[j] = nxt[j];
    y_cur = y_nxt;
    w_cur = w_nxt;
  }
  if (a.training & kHeadTraining) {
    if (active) {
      *reinterpret_cast<float4*>(sStat + (rg * 2 + 0) * C + q * 4) = g1;
      *reinterpret_cast<float4*>(sStat + (rg * 2 + 1) * C + q * 4) = g2;
    }
    __syncthreads();
    if (tid < 2 * C) {
      float v = 0.f;
      for (int r = 0; r < NRG; ++r) v += sStat[r * 2 * C + tid];
      publish_stat(a.gacc, a.gstat_part + (size_t)blockIdx.x * 2 * C, 2 * C, tid, v);
    }
  }
}

// Metric update of train.py:209-221 as one workgroup: LDS histograms of the Keras threshold buckets
// (evenly spaced thresholds => bucket = ceil(p*(n-1)) - 1), then one global add per non-empty bin.
struct MetricsArgs {
  const float* prob;   // [B]
  const float* y;      // [B]
  MetricState* m;
  int B;
  const float* z;      // [B] logits for the loss metric (null: probability form with the Keras clip)
};

template <int NT>
__device__ __forceinline__ void metrics_body(const MetricsArgs& a, unsigned (*sH101)[101], unsigned (*sH200)[200], unsigned* sCnt,
                                             double* sBce, int tid) {
  // This workgroup is the only writer of the metric state while it runs (one stream, one workgroup): the cumulative counters
  // are read at the start - beside the probabilities, one memory round trip for both - and stored back at the end, instead of
  // ~150 64-bit atomics and a read-modify-write of the loss sum behind the LDS phase.
  constexpr int N101 = (202 + NT - 1) / NT, N200 = (400 + NT - 1) / NT;
  unsigned long long old101[N101], old200[N200], old_cnt = 0ull;
  double old_bce = 0.0;
#pragma unroll
  for (int k = 0; k < N101; ++k) old101[k] = tid + k * NT < 202 ? (&a.m->hist101[0][0])[tid + k * NT] : 0ull;
#pragma unroll
  for (int k = 0; k < N200; ++k) old200[k] = tid + k * NT < 400 ? (&a.m->hist200[0][0])[tid + k * NT] : 0ull;
  if (tid < 7) old_cnt = (&a.m->n)[tid];
  if (tid == 0) old_bce = a.m->bce_sum;
  for (int i = tid; i < 202; i += NT) (&sH101[0][0])[i] = 0u;
  for (int i = tid; i < 400; i += NT) (&sH200[0][0])[i] = 0u;
  if (tid < 8) sCnt[tid] = 0u;
  __syncthreads();
  // counters and the loss sum per thread, then per wave (shuffles), then one LDS add per wave: the first version sent every
  // window's seven counter updates to the same LDS words and let thread 0 add the NT loss partials one after the other -
  // ~8 us of the gradient-assembly launch this workgroup rides in, which was that launch's critical path
  double bce = 0.0;
  int cnt[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int b = tid; b < a.B; b += NT) {
    const float pr = a.prob[b], yy = a.y[b];
    const int lab = yy > 0.5f ? 1 : 0;
    const float p01 = fminf(fmaxf(pr, 0.f), 1.f);
    const int b101 = (int)ceilf(p01 * 100.0f) - 1;
    int b200 = (int)ceilf(p01 * 199.0f) - 1;
    if (b200 < 0) b200 = 0;                       // AUC thresholds carry epsilon ends
    if (b101 >= 0) atomicAdd(&sH101[lab][b101], 1u);   // p == 0 exceeds no threshold (0.0 included): strict '>' of train.py's metrics
    atomicAdd(&sH200[lab][b200], 1u);
    const bool ppos = pr > 0.5f;
    cnt[0] += 1;
    cnt[1] += (ppos == (lab == 1)) ? 1 : 0;
    cnt[2] += (ppos && lab) ? 1 : 0;
    cnt[3] += (ppos && !lab) ? 1 : 0;
    cnt[4] += (!ppos && lab) ? 1 : 0;
    cnt[5] += lab ? 1 : 0;
    cnt[6] += lab ? 0 : 1;
    bce += (double)bce_value(a.z ? a.z[b] : 0.f, pr, yy, a.z == nullptr);
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    int v = cnt[k];
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    if ((tid & 63) == 0 && v) atomicAdd(&sCnt[k], (unsigned)v);
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    union { double d; int i[2]; } u;
    u.d = bce;
    u.i[0] = __shfl_xor(u.i[0], m);
    u.i[1] = __shfl_xor(u.i[1], m);
    bce += u.d;
  }
  if ((tid & 63) == 0) sBce[tid >> 6] = bce;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N101; ++k) {
    const int i = tid + k * NT;
    const unsigned v = i < 202 ? (&sH101[0][0])[i] : 0u;
    if (v) (&a.m->hist101[0][0])[i] = old101[k] + v;
  }
#pragma unroll
  for (int k = 0; k < N200; ++k) {
    const int i = tid + k * NT;
    const unsigned v = i < 400 ? (&sH200[0][0])[i] : 0u;
    if (v) (&a.m->hist200[0][0])[i] = old200[k] + v;
  }
  if (tid < 7 && sCnt[tid]) (&a.m->n)[tid] = old_cnt + sCnt[tid];
  if (tid == 0) {
    double s = 0.0;
    for (int i = 0; i < NT / 64; ++i) s += sBce[i];
    a.m->bce_sum = old_bce + s;
  }
}

__global__ __launch_bounds__(1024) void metrics_kernel(MetricsArgs a) {
  __shared__ unsigned sH101[2][101];
  __shared__ unsigned sH200[2][200];
  __shared__ unsigned sCnt[8];       // n, correct, tp5, fp5, fn5, pos, neg
  __shared__ double sBce[1024];
  metrics_body<1024>(a, sH101, sH200, sCnt, sBce, threadIdx.x);
}

// Dense-weight gradient of the classifier head (second read of p_L) as a batch-chunked reduction
struct DenseGradArgs {
  const float* p;       // p_L [B][T*C]
  const float* scale;   // [C]
  const float* shift;   // [C]
  const float* dz;      // [B]
  float* part;          // [n_chunks][stride]   (element n = dense bias gradient)
  int B, n, C, stride, chunk;
  const float* keep;    // [B][n] dropout keep-scale in front of the dense layer, or null
  const float *rp, *rscale, *rshift;   // residual branch added before the last ReLU ([B][rT][C], frame t + rdrop), or null
  int rT, rdrop;
  int p_bf16;           // p is stored as bf16 ("storage_bf16")
};

__device__ __forceinline__ void dense_grad_body(const DenseGradArgs& a, int bx, int by, int tid) {
  const int e = bx * kThreads + tid;
  const int b0 = by * a.chunk, b1 = min(a.B, b0 + a.chunk);
  if (e < a.n) {
    const int c = e % a.C;
    const float sc = a.scale[c], sh = a.shift[c];
    const float rsc = a.rp ? a.rscale[c] : 0.f, rsh = a.rp ? a.rshift[c] : 0.f;
    const size_t roff = a.rp ? (size_t)a.rdrop * a.C + e : 0, rstride = (size_t)a.rT * a.C;
    float acc = 0.f;
    // storage mode / residual branch / dropout scale as template arguments, clamped row index instead of predicates:
    // all loads of a batch of 8 rows are issued before the first use (see grad_final_kernel's dense role)
    auto rows = [&](auto stored, auto has_res, auto has_keep) {
      constexpr bool SB = decltype(stored)::value, RES = decltype(has_res)::value, KEEP = decltype(has_keep)::value;
      for (int bb = b0; bb < b1; bb += 8) {
        float v[8], d[8], r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const size_t row = (size_t)min(bb + u, b1 - 1);
          v[u] = load_elem<SB>(a.p, row * a.n + e);
          d[u] = a.dz[row];
          r[u] = 0.f;
          if constexpr (RES) r[u] = a.rp[row * rstride + roff];
          if constexpr (KEEP) d[u] *= a.keep[row * a.n + e];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if constexpr (RES) r[u] = fmaf(r[u], rsc, rsh);
          if (bb + u >= b1) d[u] = 0.f;
          acc = fmaf(d[u], fmaxf(fmaf(v[u], sc, sh) + r[u], 0.f), acc);
        }
      }
    };
    auto pick_keep = [&](auto stored, auto has_res) {
      if (a.keep) rows(stored, has_res, std::true_type{});
      else rows(stored, has_res, std::false_type{});
    };
    auto pick_res = [&](auto stored) {
      if (a.rp) pick_keep(stored, std::true_type{});
      else pick_keep(stored, std::false_type{});
    };
    if (a.p_bf16) pick_res(std::true_type{});
    else pick_res(std::false_type{});
    a.part[(size_t)by * a.stride + e] = acc;
  } else if (e == a.n) {
    float s = 0.f;
    for (int b = b0; b < b1; ++b) s += a.dz[b];
    a.part[(size_t)by * a.stride + a.n] = s;
  }
}

__global__ __launch_bounds__(kThreads) void dense_grad_kernel(DenseGradArgs a) {
  dense_grad_body(a, blockIdx.x, blockIdx.y, threadIdx.x);
}

}  // namespace mww

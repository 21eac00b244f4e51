// Forward kernels of the MixedNet train step (reference graph: microwakeword/mixednet.py:307-386).
//
//   fwd_first_kernel : x -> Conv2D(K1x1, valid, no bias) -> ReLU -> DepthwiseConv(Kx1)+bias
//                        -> 1x1 Conv -> p_1 (pre-BN) + per-workgroup (sum, sum^2) partials
//                      (mixednet.py:317-331 then :209-211 and :349-351; no BN in between, so one kernel)
//   fwd_block_kernel : p_{k-1} -> [BN_{k-1} + ReLU on load] -> Depthwise(Kx1)+bias -> 1x1 -> p_k + partials
//                      (mixednet.py:352,360 of block k-1, then :209-211,:349-351 of block k)
//   bn_fwd_finalize_kernel : partials -> batch mean / biased variance -> folded scale/shift,
//                      saved mean/rstd for backward, Keras moving-average update
//   bn_eval_prepare_kernel : moving stats -> folded scale/shift (inference)
//   head_kernel      : p_L -> BN_L + ReLU -> Flatten -> Dense(1) -> sigmoid, Keras BCE (clipped
//                      probability form, train.py:206), dL/dz, dense-weight gradient partials,
//                      BN_L backward partials, metric histograms (train.py:209-221)
#pragma once
#include "common.hip.h"

namespace mww {

struct FwdFirstArgs {
  const float* x;        // [B][T][40]
  const float* w1;       // [K1*40][C1]   (Keras [K1,1,40,C1] flattened)
  const float* dw_w;     // [K][C1]
  const float* dw_b;     // [C1]
  const float* pw_w;     // [C1][COUT]
  float* out;            // p_1 [B][Tout][COUT]
  float* stat_part;      // [gridDim.x][2][COUT]
  int B, T, Tout;        // Tout = T - (K1-1) - (K-1)
};

struct FwdBlockArgs {
  const float* in;       // p_{k-1} [B][Tin][CIN]
  const float* in_scale; // [CIN]  gamma*rstd of BN_{k-1}
  const float* in_shift; // [CIN]  beta - mean*gamma*rstd
  const float* dw_w;     // [K][CIN]
  const float* dw_b;     // [CIN]
  const float* pw_w;     // [CIN][COUT]
  float* out;            // p_k [B][Tout][COUT]
  float* stat_part;      // [gridDim.x][2][COUT]
  int B, Tin, Tout;      // Tout = Tin - (K-1)
};

// depthwise conv over one (channel, chunk): out[t] = bias + sum_i w[i]*src[t+i], t in [0,L)
// src rows are read from LDS once into a register window (fully unrolled, static indices).
//   REV : use the taps reversed (w[K-1-i]) — the transposed (input-gradient) depthwise
//   ACT : apply y = relu(src*sc + sh) while loading (LDS holds the raw pre-BN tensor)
template <int K, int L, bool REV = false, bool ACT = false>
__device__ __forceinline__ void dw_chunk(const float* src, int src_pitch, int row0, int row_limit, int c,
                                         const float (&w)[K], float bias, float (&out)[L], float sc = 1.f,
                                         float sh = 0.f) {
  float win[L + K - 1];
#pragma unroll
  for (int j = 0; j < L + K - 1; ++j) {
    const int r = row0 + j;
    float v = (r < row_limit) ? src[r * src_pitch + c] : 0.f;
    if (ACT) v = (r < row_limit) ? fmaxf(fmaf(v, sc, sh), 0.f) : 0.f;
    win[j] = v;
  }
#pragma unroll
  for (int t = 0; t < L; ++t) {
    float acc = bias;
#pragma unroll
    for (int i = 0; i < K; ++i) acc = fmaf(REV ? w[K - 1 - i] : w[i], win[t + i], acc);
    out[t] = acc;
  }
}

// 1x1 conv of one 16-row tile held in LDS (rows row0.., pitch CPI) against register-resident weights.
template <int KS, int NT>
__device__ __forceinline__ void pw_rowtile(const float* sU, int cpi, int row0, int r16, int g,
                                           const float (&bfrag)[KS][NT], f32x4 (&acc)[NT]) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = zero4();
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    const float av = sU[(row0 + r16) * cpi + kk * 4 + g];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma4(av, bfrag[kk][nt], acc[nt]);
  }
}

// store a 16 x (NT*16) accumulator tile to global rows and accumulate per-channel sum / sum^2
template <int NT, int COUT>
__device__ __forceinline__ void store_tile_stats(const f32x4 (&acc)[NT], float* out_rows, int row0, int rows_valid,
                                                 int r16, int g, float (&s1)[NT], float (&s2)[NT]) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + g * 4 + r;
      const float v = acc[nt][r];
      if (row < rows_valid) {
        out_rows[(size_t)row * COUT + nt * 16 + r16] = v;
        s1[nt] += v;
        s2[nt] = fmaf(v, v, s2[nt]);
      }
    }
  }
}

template <int NT, int COUT>
__device__ __forceinline__ void write_stat_partials(float (&s1)[NT], float (&s2)[NT], float* sRed, float* dst,
                                                    int tid, int wave, int r16, int g) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    s1[nt] = sum_over_groups(s1[nt]);
    s2[nt] = sum_over_groups(s2[nt]);
    if (g == 0) {
      sRed[(wave * 2 + 0) * COUT + nt * 16 + r16] = s1[nt];
      sRed[(wave * 2 + 1) * COUT + nt * 16 + r16] = s2[nt];
    }
  }
  __syncthreads();
  if (tid < 2 * COUT) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += sRed[w * 2 * COUT + tid];
    dst[tid] = v;
  }
}

// ------------------------------------------------------------------------------------------
template <int K1, int C1, int COUT, int K>
__global__ __launch_bounds__(kThreads) void fwd_first_kernel(FwdFirstArgs a) {
  constexpr int CP1 = pitch(C1);
  constexpr int RA = TT + K - 1;               // a0 rows per tile
  constexpr int RT1 = (RA + 15) / 16;          // MFMA row tiles of the first conv
  constexpr int XR = RT1 * 16 + K1 - 1;        // x rows staged (zero filled past the valid ones)
  constexpr int KS1 = K1 * FBINS / 4;          // k-steps of the im2col GEMM
  constexpr int NT1 = C1 / 16;
  constexpr int KS = C1 / 4, NT = COUT / 16;
  constexpr int NCH = nchunks(C1), L = chunk_len(C1);
  static_assert(4 % NT1 == 0, "first-conv filters must be 16, 32 or 64");
  static_assert((K1 * FBINS) % 4 == 0 && C1 % 16 == 0 && COUT % 16 == 0, "shape");

  __shared__ __attribute__((aligned(16))) float sX[XR * FBINS];
  __shared__ __attribute__((aligned(16))) float sA[RA * CP1];
  __shared__ __attribute__((aligned(16))) float sU[TT * CP1];
  __shared__ __attribute__((aligned(16))) float sRed[4 * 2 * COUT];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
  const int c = tid % C1, chunk = tid / C1;
  const bool dw_active = chunk < NCH;

  // register-resident weights
  const int nt1 = wave % NT1;
  float w1frag[KS1];
#pragma unroll
  for (int kk = 0; kk < KS1; ++kk) w1frag[kk] = a.w1[(kk * 4 + g) * C1 + nt1 * 16 + r16];
  float bfrag[KS][NT];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bfrag[kk][nt] = a.pw_w[(kk * 4 + g) * COUT + nt * 16 + r16];
  float dww[K];
  float dwb = 0.f;
  if (dw_active) {
#pragma unroll
    for (int i = 0; i < K; ++i) dww[i] = a.dw_w[i * C1 + c];
    dwb = a.dw_b[c];
  } else {
#pragma unroll
    for (int i = 0; i < K; ++i) dww[i] = 0.f;
  }
  float s1[NT], s2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) s1[nt] = s2[nt] = 0.f;

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    for (int t0 = 0; t0 < a.Tout; t0 += TT) {
      const int rows_out = min(TT, a.Tout - t0);
      const int rows_a = rows_out + K - 1;
      const int rows_x = rows_a + K1 - 1;
      // stage x rows [t0, t0+rows_x) (contiguous in HBM), zero fill the rest
      const float4* src = reinterpret_cast<const float4*>(a.x + ((size_t)b * a.T + t0) * FBINS);
      float4* dst = reinterpret_cast<float4*>(sX);
      const int nvalid = rows_x * FBINS / 4;
      for (int i = tid; i < XR * FBINS / 4; i += kThreads) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nvalid) v = src[i];
        dst[i] = v;
      }
      __syncthreads();
      // first conv as im2col GEMM: A[row][k] = sX[row*40 + k], k = j*40 + f
      for (int rt = wave / NT1; rt < RT1; rt += 4 / NT1) {
        f32x4 acc = zero4();
#pragma unroll
        for (int kk = 0; kk < KS1; ++kk) acc = mfma4(sX[(rt * 16 + r16) * FBINS + kk * 4 + g], w1frag[kk], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rt * 16 + g * 4 + r;
          if (row < RA) sA[row * CP1 + nt1 * 16 + r16] = fmaxf(acc[r], 0.f);
        }
      }
      __syncthreads();
      // depthwise
      if (dw_active) {
        float o[L];
        dw_chunk<K, L>(sA, CP1, chunk * L, rows_a, c, dww, dwb, o);
#pragma unroll
        for (int t = 0; t < L; ++t) {
          const int tl = chunk * L + t;
          if (tl < TT) sU[tl * CP1 + c] = (tl < rows_out) ? o[t] : 0.f;
        }
      }
      __syncthreads();
      // pointwise
      f32x4 acc[NT];
      pw_rowtile<KS, NT>(sU, CP1, wave * 16, r16, g, bfrag, acc);
      store_tile_stats<NT, COUT>(acc, a.out + ((size_t)b * a.Tout + t0) * COUT, wave * 16, rows_out, r16, g, s1, s2);
      __syncthreads();
    }
  }
  write_stat_partials<NT, COUT>(s1, s2, sRed, a.stat_part + (size_t)blockIdx.x * 2 * COUT, tid, wave, r16, g);
}

// ------------------------------------------------------------------------------------------
template <int CIN, int COUT, int K>
__global__ __launch_bounds__(kThreads) void fwd_block_kernel(FwdBlockArgs a) {
  constexpr int CPI = pitch(CIN);
  constexpr int RA = TT + K - 1;
  constexpr int KS = CIN / 4, NT = COUT / 16;
  constexpr int NCH = nchunks(CIN), L = chunk_len(CIN);
  constexpr int Q = CIN / 4;
  static_assert(CIN % 16 == 0 && COUT % 16 == 0, "channel counts must be multiples of 16");

  __shared__ __attribute__((aligned(16))) float sA[RA * CPI];
  __shared__ __attribute__((aligned(16))) float sU[TT * CPI];
  __shared__ __attribute__((aligned(16))) float sRed[4 * 2 * COUT];
  __shared__ __attribute__((aligned(16))) float sScale[CIN];
  __shared__ __attribute__((aligned(16))) float sShift[CIN];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
  const int c = tid % CIN, chunk = tid / CIN;
  const bool dw_active = chunk < NCH;

  if (tid < CIN) {
    sScale[tid] = a.in_scale[tid];
    sShift[tid] = a.in_shift[tid];
  }
  float bfrag[KS][NT];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bfrag[kk][nt] = a.pw_w[(kk * 4 + g) * COUT + nt * 16 + r16];
  float dww[K];
  float dwb = 0.f;
  if (dw_active) {
#pragma unroll
    for (int i = 0; i < K; ++i) dww[i] = a.dw_w[i * CIN + c];
    dwb = a.dw_b[c];
  } else {
#pragma unroll
    for (int i = 0; i < K; ++i) dww[i] = 0.f;
  }
  float s1[NT], s2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) s1[nt] = s2[nt] = 0.f;
  __syncthreads();

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    for (int t0 = 0; t0 < a.Tout; t0 += TT) {
      const int rows_out = min(TT, a.Tout - t0);
      const int rows_in = rows_out + K - 1;
      const float* src = a.in + ((size_t)b * a.Tin + t0) * CIN;
      for (int i = tid; i < rows_in * Q; i += kThreads) {
        const int r = i / Q, q = i - r * Q;
        float4 v = *reinterpret_cast<const float4*>(src + (size_t)r * CIN + q * 4);
        const float4 sc = *reinterpret_cast<const float4*>(sScale + q * 4);
        const float4 sh = *reinterpret_cast<const float4*>(sShift + q * 4);
        v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f);
        v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
        v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f);
        v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
        *reinterpret_cast<float4*>(sA + r * CPI + q * 4) = v;
      }
      __syncthreads();
      if (dw_active) {
        float o[L];
        dw_chunk<K, L>(sA, CPI, chunk * L, rows_in, c, dww, dwb, o);
#pragma unroll
        for (int t = 0; t < L; ++t) {
          const int tl = chunk * L + t;
          if (tl < TT) sU[tl * CPI + c] = (tl < rows_out) ? o[t] : 0.f;
        }
      }
      __syncthreads();
      f32x4 acc[NT];
      pw_rowtile<KS, NT>(sU, CPI, wave * 16, r16, g, bfrag, acc);
      store_tile_stats<NT, COUT>(acc, a.out + ((size_t)b * a.Tout + t0) * COUT, wave * 16, rows_out, r16, g, s1, s2);
      __syncthreads();
    }
  }
  write_stat_partials<NT, COUT>(s1, s2, sRed, a.stat_part + (size_t)blockIdx.x * 2 * COUT, tid, wave, r16, g);
}

// ------------------------------------------------------------------------------------------
// One workgroup of 1024 threads: 8 partial-groups x 128 (stat,channel) slots, fp64 combine.
struct BnFwdFinalizeArgs {
  const float* stat_part;  // [G][2][C]
  int G, C;
  float inv_n;             // 1 / (B*T)
  const float* gamma;      // [C]
  const float* beta;       // [C]
  float* moving_mean;      // [C] (updated in place when update_moving)
  float* moving_var;       // [C]
  float* scale;            // [C] out: gamma*rstd
  float* shift;            // [C] out: beta - mean*gamma*rstd
  float* mean;             // [C] out
  float* rstd;             // [C] out
  int update_moving;
};

__global__ __launch_bounds__(1024) void bn_fwd_finalize_kernel(BnFwdFinalizeArgs a) {
  __shared__ __attribute__((aligned(16))) double sAcc[8 * 128];
  const int tid = threadIdx.x, slot = tid & 127, grp = tid >> 7;
  double acc = 0.0;
  if (slot < 2 * a.C)
    for (int j = grp; j < a.G; j += 8) acc += (double)a.stat_part[(size_t)j * 2 * a.C + slot];
  sAcc[grp * 128 + slot] = acc;
  __syncthreads();
  if (tid < a.C) {
    double s1 = 0.0, s2 = 0.0;
    for (int j = 0; j < 8; ++j) {
      s1 += sAcc[j * 128 + tid];
      s2 += sAcc[j * 128 + a.C + tid];
    }
    const double m = s1 * (double)a.inv_n;
    double var = s2 * (double)a.inv_n - m * m;  // biased batch variance (Keras BN)
    if (var < 0.0) var = 0.0;
    const float meanf = (float)m, varf = (float)var;
    const float rstd = 1.0f / sqrtf(varf + kBnEps);
    const float sc = a.gamma[tid] * rstd;
    a.scale[tid] = sc;
    a.shift[tid] = a.beta[tid] - meanf * sc;
    a.mean[tid] = meanf;
    a.rstd[tid] = rstd;
    if (a.update_moving) {
      a.moving_mean[tid] = a.moving_mean[tid] * kBnMomentum + meanf * (1.0f - kBnMomentum);
      a.moving_var[tid] = a.moving_var[tid] * kBnMomentum + varf * (1.0f - kBnMomentum);
    }
  }
}

// inference: fold the moving statistics of every BN layer (state = [mean|var] per layer, packed)
struct BnEvalPrepareArgs {
  const float* gamma;
  const float* beta;
  const float* moving_mean;
  const float* moving_var;
  float* scale;
  float* shift;
  int C;
};

__global__ void bn_eval_prepare_kernel(BnEvalPrepareArgs a) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < a.C) {
    const float rstd = 1.0f / sqrtf(a.moving_var[c] + kBnEps);
    const float sc = a.gamma[c] * rstd;
    a.scale[c] = sc;
    a.shift[c] = a.beta[c] - a.moving_mean[c] * sc;
  }
}

// ------------------------------------------------------------------------------------------
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
  const float* y;          // [B] labels (training / metrics)
  const float* sw;         // [B] per-sample weight (penalty * class weight)
  float* z;                // [B] logits
  float* prob;             // [B]
  float* dz;               // [B] dL/dz (training)
  float* loss_part;        // [B] weighted loss / B per sample (training)
  float* dwd_part;         // [gridDim.x][dwd_stride], dwd_stride = T*C + 4 (element T*C = dense bias gradient)
  float* gstat_part;       // [gridDim.x][2][C]   sum g, sum g*xhat of the BN_L input gradient
  MetricState* metrics;    // may be null
  int B, T;
  int dwd_stride;
  float inv_b;
  int training;
};

template <int C, int JMAX>
__global__ __launch_bounds__(kThreads) void head_kernel(HeadArgs a) {
  constexpr int Q = C / 4;                 // float4 per frame
  constexpr int NRG = kThreads / Q;        // frame groups
  __shared__ __attribute__((aligned(16))) float sRed[8];
  __shared__ __attribute__((aligned(16))) float sBcast[2];
  __shared__ __attribute__((aligned(16))) float sStat[NRG * 2 * C];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = tid % Q, rg = tid / Q;
  const bool active = rg < NRG;
  float4 sc = make_float4(0, 0, 0, 0), sh = sc, mu = sc, rs = sc;
  if (active) {
    sc = *reinterpret_cast<const float4*>(a.scale + q * 4);
    sh = *reinterpret_cast<const float4*>(a.shift + q * 4);
    if (a.training) {
      mu = *reinterpret_cast<const float4*>(a.mean + q * 4);
      rs = *reinterpret_cast<const float4*>(a.rstd + q * 4);
    }
  }
  float4 wdv[JMAX], dwd[JMAX];
#pragma unroll
  for (int j = 0; j < JMAX; ++j) {
    const int t = rg + NRG * j;
    wdv[j] = (active && t < a.T) ? *reinterpret_cast<const float4*>(a.wd + (size_t)t * C + q * 4) : make_float4(0, 0, 0, 0);
    dwd[j] = make_float4(0, 0, 0, 0);
  }
  const float bias = a.bd[0];
  float4 g1 = make_float4(0, 0, 0, 0), g2 = g1;
  float dbias = 0.f;

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    float4 raw[JMAX], act[JMAX];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
      const int t = rg + NRG * j;
      raw[j] = make_float4(0, 0, 0, 0);
      act[j] = raw[j];
      if (active && t < a.T) {
        raw[j] = *reinterpret_cast<const float4*>(a.p + ((size_t)b * a.T + t) * C + q * 4);
        act[j].x = fmaxf(fmaf(raw[j].x, sc.x, sh.x), 0.f);
        act[j].y = fmaxf(fmaf(raw[j].y, sc.y, sh.y), 0.f);
        act[j].z = fmaxf(fmaf(raw[j].z, sc.z, sh.z), 0.f);
        act[j].w = fmaxf(fmaf(raw[j].w, sc.w, sh.w), 0.f);
        dot = fmaf(act[j].x, wdv[j].x, dot);
        dot = fmaf(act[j].y, wdv[j].y, dot);
        dot = fmaf(act[j].z, wdv[j].z, dot);
        dot = fmaf(act[j].w, wdv[j].w, dot);
      }
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
        const float yy = a.y[b];
        // Keras binary_crossentropy(from_logits=False): clip to [eps, 1-eps], probability form
        const float pc = fminf(fmaxf(pr, kKerasEps), 1.0f - kKerasEps);
        const float bce = -(yy * logf(pc) + (1.0f - yy) * logf(1.0f - pc));
        if (a.training) {
          const float w = a.sw[b];
          a.loss_part[b] = w * bce * a.inv_b;
          const bool clipped = (pr < kKerasEps) || (pr > 1.0f - kKerasEps);
          dzz = clipped ? 0.f : w * (pr - yy) * a.inv_b;
          a.dz[b] = dzz;
        }
        if (a.metrics != nullptr) {
          MetricState* m = a.metrics;
          const int lab = yy > 0.5f ? 1 : 0;
          const float p01 = fminf(fmaxf(pr, 0.f), 1.f);
          const int b101 = (int)ceilf(p01 * 100.0f) - 1;                 // Keras evenly-spaced bucketing
          int b200 = (int)ceilf(p01 * 199.0f) - 1;
          if (b200 < 0) b200 = 0;                                         // AUC thresholds carry epsilon ends
          if (b101 >= 0) atomicAdd(&m->hist101[lab][b101], 1ull);
          atomicAdd(&m->hist200[lab][b200], 1ull);
          const bool ppos = pr > 0.5f;
          atomicAdd(&m->n, 1ull);
          if (ppos == (lab == 1)) atomicAdd(&m->correct, 1ull);
          if (ppos && lab) atomicAdd(&m->tp5, 1ull);
          if (ppos && !lab) atomicAdd(&m->fp5, 1ull);
          if (!ppos && lab) atomicAdd(&m->fn5, 1ull);
          atomicAdd(lab ? &m->pos : &m->neg, 1ull);
          atomicAdd(&m->bce_sum, (double)bce);
        }
      }
      sBcast[0] = dzz;
    }
    __syncthreads();
    if (a.training) {
      const float dzz = sBcast[0];
      if (tid == 0) dbias += dzz;
#pragma unroll
      for (int j = 0; j < JMAX; ++j) {
        dwd[j].x = fmaf(dzz, act[j].x, dwd[j].x);
        dwd[j].y = fmaf(dzz, act[j].y, dwd[j].y);
        dwd[j].z = fmaf(dzz, act[j].z, dwd[j].z);
        dwd[j].w = fmaf(dzz, act[j].w, dwd[j].w);
        // gradient entering BN_L: g = dz * wd * relu'(.) ; partial sums of g and g*xhat
        const float gx = act[j].x > 0.f ? dzz * wdv[j].x : 0.f;
        const float gy = act[j].y > 0.f ? dzz * wdv[j].y : 0.f;
        const float gz = act[j].z > 0.f ? dzz * wdv[j].z : 0.f;
        const float gw = act[j].w > 0.f ? dzz * wdv[j].w : 0.f;
        g1.x += gx; g1.y += gy; g1.z += gz; g1.w += gw;
        g2.x = fmaf(gx, (raw[j].x - mu.x) * rs.x, g2.x);
        g2.y = fmaf(gy, (raw[j].y - mu.y) * rs.y, g2.y);
        g2.z = fmaf(gz, (raw[j].z - mu.z) * rs.z, g2.z);
        g2.w = fmaf(gw, (raw[j].w - mu.w) * rs.w, g2.w);
      }
    }
    // sRed / sBcast are rewritten only after the next sample's first barrier
  }
  if (a.training) {
    float* dst = a.dwd_part + (size_t)blockIdx.x * (size_t)a.dwd_stride;
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
      const int t = rg + NRG * j;
      if (active && t < a.T) *reinterpret_cast<float4*>(dst + (size_t)t * C + q * 4) = dwd[j];
    }
    if (tid == 0) dst[(size_t)a.T * C] = dbias;
    if (active) {
      *reinterpret_cast<float4*>(sStat + (rg * 2 + 0) * C + q * 4) = g1;
      *reinterpret_cast<float4*>(sStat + (rg * 2 + 1) * C + q * 4) = g2;
    }
    __syncthreads();
    if (tid < 2 * C) {
      float v = 0.f;
      for (int r = 0; r < NRG; ++r) v += sStat[r * 2 * C + tid];
      a.gstat_part[(size_t)blockIdx.x * 2 * C + tid] = v;
    }
  }
}

}  // namespace mww

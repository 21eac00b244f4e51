// Backward kernels of the MixedNet train step.  The reference gets these from TF autodiff of
// the graph in microwakeword/mixednet.py:307-386 (Keras train_on_batch, train.py:295-299); here
// each block's backward is one fused kernel that RECOMPUTES the block's activations from the
// stored pre-BN tensor p_{k-1} (SURVEY §8d "minimal materialisation"):
//
//   inputs  : p_{k-1} (raw), p_k (raw), g_k = dL/d(BN_k output) already ReLU-masked
//             (block L: rebuilt from the per-sample scalar dL/dz and the dense weights)
//   BN_k bwd: dp_k = gamma*rstd * (g_k - mean(g_k) - xhat_k * mean(g_k*xhat_k))
//   1x1 bwd : dW_pw += u^T dp_k           (MFMA, accumulated in registers over the whole grid-stride loop)
//             du     = dp_k W_pw^T        (MFMA)
//   dw  bwd : dW_dw[i,c] += sum_t du[t,c] a[t+i,c] ; db[c] += sum_t du[t,c]
//             da[s,c]    = sum_i du[s-i,c] w[i,c]    (K-1 rows of du carried across time tiles in LDS)
//   output  : g_{k-1} = da * relu'(.)  -> HBM, plus per-workgroup partials of sum g_{k-1}, sum g_{k-1} xhat_{k-1}
//
// Weight-gradient partials are written once per workgroup ([grid][params of the block]) and
// summed in a fixed order by grad_reduce_kernel => bit-reproducible gradients.
#pragma once
#include "kernels_fwd.hip.h"

namespace mww {

struct BwdBlockArgs {
  const float* in;        // p_{k-1} [B][Tin][CIN] raw
  const float* in_scale;  // BN_{k-1} folded scale/shift (activation recompute)
  const float* in_shift;
  const float* in_mean;   // BN_{k-1} batch mean / rstd (xhat for the stats of g_{k-1})
  const float* in_rstd;
  const float* pk;        // p_k [B][Tout][COUT] raw
  const float* gk;        // g_k [B][Tout][COUT]            (unused when LAST)
  const float* k_mean;    // BN_k: batch mean, rstd
  const float* k_rstd;
  const float* k_c1;      // gamma_k * rstd_k
  const float* k_mg;      // mean(g_k)
  const float* k_mgx;     // mean(g_k * xhat_k)
  const float* k_scale;   // LAST only: BN_k folded (ReLU mask of the head input)
  const float* k_shift;
  const float* wd;        // LAST only: dense kernel [Tout*COUT]
  const float* dz;        // LAST only: dL/dz [B]
  const float* dw_w;      // [K][CIN]
  const float* dw_b;      // [CIN]
  const float* pw_w;      // [CIN][COUT]
  float* g_out;           // g_{k-1} [B][Tin][CIN]
  float* gstat_part;      // [gridDim.x][2][CIN]
  float* grad_part;       // [gridDim.x][K*CIN + CIN + CIN*COUT]  (dW_dw, db, dW_pw)
  int B, Tin, Tout;
  int ablate;             // profiling only: bit0 skip P1, bit1 skip MFMA, bit2 skip P4 (results invalid); bit 16: phase clocks;
                          // parts of P4: 32 skip the g_{k-1} stores, 64 skip the depthwise weight gradient, 128 skip the input gradient
  unsigned long long* phase_clk;   // [gridDim.x][8]
  StatAcc gacc;           // (sum g, sum g*xhat) of g_{k-1} go to the accumulator rows instead of gstat_part when set
  BnGradFoldArgs gfold;   // gfold.acc set: k_c1 / k_mg / k_mgx are folded here from the producer's accumulator rows
};

// dp tile: rows [t0, t0+TT) of (p_k, g_k) are fetched into registers early (issue) and turned into
// dp = BN_k backward of g_k while being written to LDS late (commit).  A sample's rows are
// contiguous, so float4 i of the tile sits at offset 4*i from the tile start.
template <int COUT, bool LAST, bool SB = false>
struct DpStage {
  static constexpr int QO = COUT / 4, CPO = pitch(COUT), N = (TT * QO + kThreads - 1) / kThreads;
  float4 pk[N], gg[N];

  // both slices hold nvalid float4s; float4s past them come back as zeros
  // (LAST: g_base are rows of the dense kernel, a parameter: always fp32)
  __device__ __forceinline__ void issue(const float* pk_base, const float* g_base, int nvalid, int tid) {
    constexpr bool SG = SB && !LAST;
    const BufRsrc rp = tile_rsrc(pk_base, nvalid * 4 * elem_bytes(SB)), rg = tile_rsrc(g_base, nvalid * 4 * elem_bytes(SG));
#pragma unroll
    for (int j = 0; j < N; ++j) {
      pk[j] = tile_load4s<SB>(rp, tid + j * kThreads);
      gg[j] = tile_load4s<SG>(rg, tid + j * kThreads);
    }
  }

  // g_base rows are g_k (middle blocks) or the dense kernel rows (LAST: g = dz * wd * relu'(bn_k(p_k)))
  __device__ __forceinline__ void commit(float* sDP, const float* sKp, float dzb, int nvalid, int tid) const {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int i = tid + j * kThreads;
      if (i < TT * QO) {
        const int r = i / QO, q = i - r * QO;
        float4 dp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nvalid) {
          // sKp rows: 0 = c1, 1 = kA, 2 = kB with dp = c1*(g - mg - (p - mean)*rstd*mgx) = c1*g + (kA*p + kB)
          // (kA = -c1*rstd*mgx, kB = -c1*mg - kA*mean, formed once per channel in the prologue: two fma per element)
          const float4 p = pk[j];
          const float4 c1 = *reinterpret_cast<const float4*>(sKp + 0 * COUT + q * 4);
          const float4 kA = *reinterpret_cast<const float4*>(sKp + 1 * COUT + q * 4);
          const float4 kB = *reinterpret_cast<const float4*>(sKp + 2 * COUT + q * 4);
          float4 g = gg[j];
          if (LAST) {
            const float4 sc = *reinterpret_cast<const float4*>(sKp + 5 * COUT + q * 4);
            const float4 sh = *reinterpret_cast<const float4*>(sKp + 6 * COUT + q * 4);
            g.x = fmaf(p.x, sc.x, sh.x) > 0.f ? dzb * g.x : 0.f;
            g.y = fmaf(p.y, sc.y, sh.y) > 0.f ? dzb * g.y : 0.f;
            g.z = fmaf(p.z, sc.z, sh.z) > 0.f ? dzb * g.z : 0.f;
            g.w = fmaf(p.w, sc.w, sh.w) > 0.f ? dzb * g.w : 0.f;
          }
          dp.x = fmaf(g.x, c1.x, fmaf(p.x, kA.x, kB.x));
          dp.y = fmaf(g.y, c1.y, fmaf(p.y, kA.y, kB.y));
          dp.z = fmaf(g.z, c1.z, fmaf(p.z, kA.z, kB.z));
          dp.w = fmaf(g.w, c1.w, fmaf(p.w, kA.w, kB.w));
        }
        *reinterpret_cast<float4*>(sDP + r * CPO + q * 4) = dp;
      }
    }
  }
};

// Weight staging of the backward kernels, in two halves so that the prologue is ONE memory round trip deep: load()
// issues every global load of the thread (W_pw for the transposed LDS copy, the depthwise taps), the BN fold that
// follows issues its own, and store() writes LDS once everything has arrived.  (Rolled loops with a load and an LDS
// write per iteration cost one L2 round trip per element - nine in a row for W_pw^T: the round-2 timeline showed a
// 6 us prologue per backward kernel.)
template <int CIN, int COUT, int KD>
struct WeightStage {
  static constexpr int CPI = pitch(CIN), NW = (CIN * COUT + kThreads - 1) / kThreads, ND = KD > 0 ? (KD * CIN + kThreads - 1) / kThreads : 1;
  float w[NW], d[ND];
  __device__ __forceinline__ void load(const float* pw_w, const float* dw_w, int tid) {
#pragma unroll
    for (int j = 0; j < NW; ++j) w[j] = (tid + j * kThreads < CIN * COUT) ? pw_w[tid + j * kThreads] : 0.f;
#pragma unroll
    for (int j = 0; j < ND; ++j) d[j] = (KD > 0 && tid + j * kThreads < KD * CIN) ? dw_w[tid + j * kThreads] : 0.f;
  }
  // sWt [COUT][pitch(CIN)] = W_pw^T; sDW [KD][CIN] (KD = 0: the taps stay in registers, nothing to stage)
  __device__ __forceinline__ void store(float* sWt, float* sDW, int tid) const {
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int i = tid + j * kThreads;
      if (i < CIN * COUT) sWt[(i % COUT) * CPI + i / COUT] = w[j];
    }
    if (KD > 0) {
#pragma unroll
      for (int j = 0; j < ND; ++j)
        if (tid + j * kThreads < KD * CIN) sDW[tid + j * kThreads] = d[j];
    }
  }
};

// carry the last K-1 rows of du to the front of the ring (or clear them at the start of a sample)
template <int K, int CPI>
__device__ __forceinline__ void carry_du(float* sDU, bool first_tile, int tid) {
  for (int i = tid; i < (K - 1) * CPI; i += kThreads) sDU[i] = first_tile ? 0.f : sDU[TT * CPI + i];
}

// REPS x ([READS LDS reads] [MFMAS MFMAs]) in the instruction schedule of the enclosing block
template <int REPS, int READS, int MFMAS>
__device__ __forceinline__ void sched_read_mfma_groups() {
  if constexpr (REPS > 0) {
    if constexpr (READS > 0) __builtin_amdgcn_sched_group_barrier(0x100, READS, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, MFMAS, 0);
    sched_read_mfma_groups<REPS - 1, READS, MFMAS>();
  }
}

// MFMA part shared by both backward kernels:
//   dwacc[mt][nt] += U^T DP over this wave's 16 rows;  du = DP W^T -> sDU rows [K-1+16*wave, ...)
//   sWt = W_pw^T staged in LDS as [COUT][pitch(CIN)] (B[k=co][n=ci] = W[ci][co])
template <int CIN, int COUT, int K, bool BF>
__device__ __forceinline__ void pointwise_backward_tile(const float* sU, const float* sDP, float* sDU, int wave, int r16,
                                                        int g, const float* sWt,
                                                        f32x4 (&dwacc)[CIN / 16][COUT / 16], bool live) {
  constexpr int CPI = pitch(CIN), CPO = pitch(COUT), MT = CIN / 16, NT = COUT / 16, KSO = COUT / 4;
  f32x4 du[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) du[mt] = zero4();
  if (!live) {
    // (wave-uniform) this wave's 16 rows lie past the sample: dp = 0 there, so dW gains nothing and du = 0
  } else if constexpr (!BF) {
    // The operands of k-step kk+1 are read from LDS before the MFMAs of k-step kk are issued, and the schedule is pinned
    // that way (read group, MFMA group, ...): left to itself the scheduler sank every read to just in front of its first
    // use (register pressure), so each group of 2-3 MFMAs waited for an LDS round trip (27 waits for 72 MFMAs in the
    // round-2 ISA; the phase took 3.7k cycles alone for 2.3k of MFMA issue).
    {
      float av[2][MT], bv[2][NT];
      auto load_dw = [&](int kk, int s) {
        const int row = wave * 16 + kk * 4 + g;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[s][mt] = sU[row * CPI + mt * 16 + r16];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[s][nt] = sDP[row * CPO + nt * 16 + r16];
      };
      float a2[2], b2[2][MT];
      auto load_du = [&](int kk, int s) {
        a2[s] = sDP[(wave * 16 + r16) * CPO + kk * 4 + g];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) b2[s][mt] = sWt[(kk * 4 + g) * CPI + mt * 16 + r16];
      };
      load_dw(0, 0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk + 1 < 4) load_dw(kk + 1, (kk + 1) & 1);
        else load_du(0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) dwacc[mt][nt] = mfma4(av[kk & 1][mt], bv[kk & 1][nt], dwacc[mt][nt]);
      }
#pragma unroll
      for (int kk = 0; kk < KSO; ++kk) {
        if (kk + 1 < KSO) load_du(kk + 1, (kk + 1) & 1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) du[mt] = mfma4(a2[kk & 1], b2[kk & 1][mt], du[mt]);
      }
      // pinned order: [reads k0] ([reads k+1][MFMAs k]) x 4, then ([reads k+1][MFMAs k]) x KSO
      __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
      sched_read_mfma_groups<3, MT + NT, MT * NT>();
      sched_read_mfma_groups<1, 1 + MT, MT * NT>();
      sched_read_mfma_groups<KSO - 1, 1 + MT, MT>();
      sched_read_mfma_groups<1, 0, MT>();
    }
  } else {
    // bf16 operands: one MFMA spans the wave's 16 rows (dW) / 16 output channels (du)
    const int row = wave * 16 + 4 * g;
    bf16x4 av[MT], bv[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const float* col = sU + row * CPI + mt * 16 + r16;
      av[mt] = to_bf16x4(col[0], col[CPI], col[2 * CPI], col[3 * CPI]);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float* col = sDP + row * CPO + nt * 16 + r16;
      bv[nt] = to_bf16x4(col[0], col[CPO], col[2 * CPO], col[3 * CPO]);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) dwacc[mt][nt] = mfma_bf16(av[mt], bv[nt], dwacc[mt][nt]);
#pragma unroll
    for (int kk = 0; kk < COUT / 16; ++kk) {
      const float4 v = *reinterpret_cast<const float4*>(sDP + (wave * 16 + r16) * CPO + kk * 16 + 4 * g);
      const bf16x4 a4 = to_bf16x4(v.x, v.y, v.z, v.w);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float* col = sWt + (kk * 16 + 4 * g) * CPI + mt * 16 + r16;
        du[mt] = mfma_bf16(a4, to_bf16x4(col[0], col[CPI], col[2 * CPI], col[3 * CPI]), du[mt]);
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) sDU[(K - 1 + wave * 16 + g * 4 + r) * CPI + mt * 16 + r16] = du[mt][r];
}

// final per-workgroup write of dW_pw (cross-wave sum), dW_dw / db (cross-chunk sum)
template <int CIN, int COUT, int K>
__device__ __forceinline__ void write_block_grad_partials(float* scratch, float* dst, const f32x4 (&dwacc)[CIN / 16][COUT / 16],
                                                          const float (&accw)[K], float accb, bool dw_active, int c,
                                                          int chunk, int tid, int wave, int r16, int g) {
  constexpr int MT = CIN / 16, NT = COUT / 16, NCH = nchunks(CIN);
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        scratch[(wave * CIN + mt * 16 + g * 4 + r) * COUT + nt * 16 + r16] = dwacc[mt][nt][r];
  __syncthreads();
  for (int e = tid; e < CIN * COUT; e += kThreads)
    dst[(K + 1) * CIN + e] = (scratch[e] + scratch[CIN * COUT + e]) + (scratch[2 * CIN * COUT + e] + scratch[3 * CIN * COUT + e]);
  __syncthreads();
  if (dw_active) {
#pragma unroll
    for (int i = 0; i < K; ++i) scratch[(chunk * (K + 1) + i) * CIN + c] = accw[i];
    scratch[(chunk * (K + 1) + K) * CIN + c] = accb;
  }
  __syncthreads();
  for (int e = tid; e < (K + 1) * CIN; e += kThreads) {
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) v += scratch[j * (K + 1) * CIN + e];
    dst[e] = v;
  }
  __syncthreads();
}

// depthwise backward of one (channel, chunk) for one time tile, in two steps so that the input
// gradient can be consumed (masked, stored) before the weight-gradient window is loaded:
//   da[sl]      = sum_j w[K-1-j] * du_ring[sl + j]                     (sl local input row)
//   dW_dw[i]   += sum_t du[t] * a[t+i] ;  db += sum_t du[t]            (t local output row)
// `a_at(row)` returns the activation of local input row `row` for channel c.
template <int K, int L, int CPI>
__device__ __forceinline__ void depthwise_input_grad_chunk(const float* sDU, int chunk, int c, const float (&dww)[K],
                                                           float (&da)[L]) {
  dw_chunk<K, L, true, false>(sDU, CPI, chunk * L, c, dww, 0.f, da);
}

template <int K, int L, int CPI, typename ActFn>
__device__ __forceinline__ void depthwise_weight_grad_chunk(const float* sDU, int chunk, int c, float (&accw)[K],
                                                            float& accb, ActFn a_at) {
  // every LDS read of the phase is issued before the first FMA (read -> wait -> use per element exposes one LDS
  // round trip per output row: the round-2 ISA of this phase was a chain of lgkmcnt(0) waits)
  float win[L + K - 1], du[L];
#pragma unroll
  for (int t = 0; t < L; ++t) du[t] = sDU[(K - 1 + chunk * L + t) * CPI + c];   // ring rows past the tile are allocated and zero
#pragma unroll
  for (int j = 0; j < L + K - 1; ++j) win[j] = a_at(chunk * L + j);
  lds_reads_first();
#pragma unroll
  for (int t = 0; t < L; ++t) {
    accb += du[t];
#pragma unroll
    for (int i = 0; i < K; ++i) accw[i] = fmaf(du[t], win[t + i], accw[i]);
  }
}

// ------------------------------------------------------------------------------------------
template <int CIN, int COUT, int K, bool LAST, bool BF, bool SB = false>
__global__ __launch_bounds__(kThreads, 2) void bwd_block_kernel(BwdBlockArgs a) {
  constexpr int CPI = pitch(CIN), CPO = pitch(COUT);
  constexpr int RA = TT + K - 1;
  constexpr int MT = CIN / 16, NT = COUT / 16, KSO = COUT / 4;
  constexpr int NCH = nchunks(CIN), L = chunk_len(CIN);
  constexpr int QI = CIN / 4;
  constexpr int RAP = halo_rows_padded(CIN, K), TTP = tile_rows_padded(CIN);
  constexpr int OFF_P = 0, OFF_DP = OFF_P + RAP * CPI, OFF_U = OFF_DP + TT * CPO, OFF_DU = OFF_U + TTP * CPI;
  constexpr int OFF_END = OFF_DU + RAP * CPI;
  static_assert(OFF_END >= 4 * CIN * COUT && OFF_END >= NCH * (K + 1) * CIN && OFF_END >= NCH * 2 * CIN, "scratch aliasing");
  static_assert(TT >= K - 1, "carry rows must not overlap");

  __shared__ __attribute__((aligned(16))) float smem[OFF_END];
  __shared__ __attribute__((aligned(16))) float sKp[7 * COUT];
  __shared__ __attribute__((aligned(16))) float sWt[COUT * CPI];   // W_pw^T
  __shared__ __attribute__((aligned(16))) float sDW[K * CIN];      // depthwise taps
  __shared__ __attribute__((aligned(16))) float sAct[2 * CIN];     // BN_{k-1} folded scale / shift (activation at commit)
  float* sP = smem + OFF_P;
  float* sDP = smem + OFF_DP;
  float* sU = smem + OFF_U;
  float* sDU = smem + OFF_DU;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
  const int c = tid % CIN, chunk = tid / CIN;
  const bool dw_active = chunk < NCH;
  MWW_PC_DECL
  MWW_PC_AT(0);   // kernel entry

  // work items = (sample, input-row tile); the next item's rows travel HBM -> registers while the
  // current one is computed
  const int ntiles = (a.Tin + TT - 1) / TT;
  const int nsamp = (int)blockIdx.x < a.B ? (a.B - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const int nitems = nsamp * ntiles;
  constexpr int NP = (RA * QI + kThreads - 1) / kThreads;
  float4 pre_p[NP];
  DpStage<COUT, LAST, SB> dps;
  float pre_dz = 0.f;
  auto issue = [&](int it) {
    const int b = blockIdx.x + (it / ntiles) * gridDim.x, t0 = (it % ntiles) * TT;
    const int nvp = min(RA, a.Tin - t0) * QI;
    const BufRsrc src = tile_rsrc(elem_ptr<SB>(a.in, ((size_t)b * a.Tin + t0) * CIN), nvp * 4 * elem_bytes(SB));
#pragma unroll
    for (int j = 0; j < NP; ++j) pre_p[j] = tile_load4s<SB>(src, tid + j * kThreads);
    const int nvk = max(0, min(TT, a.Tout - t0)) * (COUT / 4);
    const size_t koff = ((size_t)b * a.Tout + t0) * COUT;
    dps.issue(elem_ptr<SB>(a.pk, koff), LAST ? a.wd + (size_t)t0 * COUT : elem_ptr<SB>(a.gk, koff), nvk, tid);
    if (LAST) pre_dz = tile_load1(tile_rsrc(a.dz, a.B * 4), b * 4);
  };
  if (nitems > 0) issue(0);

  // every global load of the prologue first ...
  WeightStage<CIN, COUT, K> wst;
  wst.load(a.pw_w, a.dw_w, tid);
  float accw[K];
  float accb = 0.f, gs1 = 0.f, gs2 = 0.f, dwb = 0.f;
  float sc_c = 0.f, sh_c = 0.f, mu_c = 0.f, rs_c = 0.f;
#pragma unroll
  for (int i = 0; i < K; ++i) accw[i] = 0.f;
  if (dw_active) {
    dwb = a.dw_b[c];
    sc_c = a.in_scale[c];
    sh_c = a.in_shift[c];
    mu_c = a.in_mean[c];
    rs_c = a.in_rstd[c];
  }
  // ... then BN_k's backward coefficients (their loads are the last ones issued: when they have arrived, all have)
  for (int i = tid; i < COUT; i += kThreads) {
    const float krs = a.k_rstd[i];
    const float kmean = a.k_mean[i];
    const float ksc = LAST ? a.k_scale[i] : 0.f, ksh = LAST ? a.k_shift[i] : 0.f;
    float c1, mg, mgx;
    if (a.gfold.acc) {
      bn_grad_fold_channel(a.gfold, COUT, i, krs, c1, mg, mgx);
    } else {
      c1 = a.k_c1[i];
      mg = a.k_mg[i];
      mgx = a.k_mgx[i];
    }
    const float kA = -c1 * krs * mgx;
    sKp[0 * COUT + i] = c1;
    sKp[1 * COUT + i] = kA;
    sKp[2 * COUT + i] = -c1 * mg - kA * kmean;
    sKp[5 * COUT + i] = ksc;
    sKp[6 * COUT + i] = ksh;
  }
  wst.store(sWt, sDW, tid);
  for (int i = RA * CPI + tid; i < RAP * CPI; i += kThreads) {   // rows only the padded windows touch
    sP[i] = 0.f;
    sDU[i] = 0.f;
  }
  f32x4 dwacc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) dwacc[mt][nt] = zero4();
  // The block input is activated ONCE, while its rows are committed to LDS: sP holds a = relu(BN_{k-1}(p_{k-1})), so the
  // depthwise recompute (P1) and the depthwise weight gradient (P4) read their windows as they are (before, each of the
  // L + K - 1 window rows went through the fma + max twice per (channel, chunk): 2 x 66 of ~1150 VALU instructions per
  // wave and tile at K = 21).  What P4 still needs of the raw tensor follows from a for the units it matters for (a > 0,
  // i.e. a = gamma * xhat + beta):  the ReLU decision is a > 0, and xhat = (a - beta) / gamma = a * xk1 + xk0.
  // (gamma = 0 gives xk1 = 0: such a channel passes no gradient to p_{k-1}; its own d gamma = sum g * xhat is then formed
  // with xhat = 0 - the one deviation from the two-pass form, for a value of gamma training does not produce.)
  if (chunk == 0) {
    sAct[c] = sc_c;
    sAct[CIN + c] = sh_c;
  }
  float xk1 = sc_c != 0.f ? rs_c / sc_c : 0.f;
  float xk0 = -(sh_c + mu_c * sc_c) * xk1;
  pin(dwb); pin(xk1); pin(xk0);
  __syncthreads();

  MWW_PC_AT(1);   // prologue done
  MWW_PC_START(MWW_ABLATE(a, 16) && tid == 0);
  for (int it = 0; it < nitems; ++it) {
    rotate_priority(it, 2);
    const int b = blockIdx.x + (it / ntiles) * gridDim.x, t0 = (it % ntiles) * TT;
    const int nrows_new = max(0, min(TT, a.Tout - t0));  // du rows produced by this tile
    const int rows_da = min(TT, a.Tin - t0);             // input-gradient rows finalised by this tile
    // ---- P0: commit raw p_{k-1} rows [t0, t0+RA) (zero past the sample), dp rows; roll the du ring
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int i = tid + j * kThreads;
      if (i < RA * QI) {
        const int r = i / QI, q = i - r * QI;
        const float4 s4 = *reinterpret_cast<const float4*>(sAct + q * 4), h4 = *reinterpret_cast<const float4*>(sAct + CIN + q * 4);
        float4 v = pre_p[j];
        v.x = fmaxf(fmaf(v.x, s4.x, h4.x), 0.f);
        v.y = fmaxf(fmaf(v.y, s4.y, h4.y), 0.f);
        v.z = fmaxf(fmaf(v.z, s4.z, h4.z), 0.f);
        v.w = fmaxf(fmaf(v.w, s4.w, h4.w), 0.f);
        *reinterpret_cast<float4*>(sP + r * CPI + q * 4) = v;
      }
    }
    dps.commit(sDP, sKp, pre_dz, nrows_new * (COUT / 4), tid);
    carry_du<K, CPI>(sDU, t0 == 0, tid);
    MWW_PC_MARK(0);   // commit (incl. wait for the prefetch)
    if (!MWW_ABLATE(a, 8)) __syncthreads();
    MWW_PC_MARK(1);   // barrier 1
    if (it + 1 < nitems) issue(it + 1);
    // ---- P1: recompute u = depthwise(relu(bn(p_{k-1}))) + bias for the tile's output rows
    // (chunks / row tiles past the sample's last row only write their zero rows, see fwd_block_kernel)
    if (dw_active && !MWW_ABLATE(a, 1)) {
      if (chunk * L < nrows_new) {
        float o[L], dww[K];
#pragma unroll
        for (int i = 0; i < K; ++i) dww[i] = sDW[i * CIN + c];
        dw_chunk<K, L, false, false>(sP, CPI, chunk * L, c, dww, dwb, o);
#pragma unroll
        for (int t = 0; t < L; ++t) {
          const int tl = chunk * L + t;
          sU[tl * CPI + c] = (tl < nrows_new) ? o[t] : 0.f;
        }
      } else {
#pragma unroll
        for (int t = 0; t < L; ++t) sU[(chunk * L + t) * CPI + c] = 0.f;
      }
    }
    MWW_PC_MARK(2);   // issue + P1 (u recompute)
    if (!MWW_ABLATE(a, 8)) __syncthreads();
    MWW_PC_MARK(3);   // barrier 2
    // ---- P2/P3: dW_pw += u^T dp ; du = dp W^T -> ring rows [K-1, K-1+TT)
    if (!MWW_ABLATE(a, 2)) pointwise_backward_tile<CIN, COUT, K, BF>(sU, sDP, sDU, wave, r16, g, sWt, dwacc, wave * 16 < nrows_new);
    MWW_PC_MARK(4);   // MFMA (dW_pw, du)
    if (!MWW_ABLATE(a, 8)) __syncthreads();
    MWW_PC_MARK(5);   // barrier 3
    // ---- P4: depthwise backward, ReLU mask, stats, store g_{k-1}.  No divergent branch around the global stores (the
    // wait-count pass would have to assume the skipped path at the next commit): the lanes past the last chunk shadow
    // it, their stores are out of range and their sums are dropped by the epilogue.
    if (!MWW_ABLATE(a, 4)) {
      const int cch = dw_active ? chunk : NCH - 1;
      // the tile's slice of g_{k-1}: rows past rows_da are dropped by the address unit
      const BufRsrc gtile = tile_rsrc(elem_ptr<SB>(a.g_out, ((size_t)b * a.Tin + t0) * CIN), rows_da * CIN * elem_bytes(SB));
      const int goff = (dw_active ? 0 : kOobOffset / 4) + cch * L * CIN + c;   // element index (out of range for the shadow lanes)
      {
        float da[L], raw[L];
#pragma unroll
        for (int t = 0; t < L; ++t) raw[t] = sP[(cch * L + t) * CPI + c];   // in flight under the da FMAs
        if (cch * L < rows_da && !MWW_ABLATE(a, 128)) {   // chunks past the sample's last row: da = 0, nothing to compute
          float dww[K];
#pragma unroll
          for (int i = 0; i < K; ++i) dww[i] = sDW[i * CIN + c];
          depthwise_input_grad_chunk<K, L, CPI>(sDU, cch, c, dww, da);
        } else {
#pragma unroll
          for (int t = 0; t < L; ++t) da[t] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < L; ++t) {
          const int sl = cch * L + t;
          // (row TT of the last chunk belongs to the next tile: its da is still partial)
          const float gg = (sl < rows_da && raw[t] > 0.f) ? da[t] : 0.f;   // raw = the activated value here
          if (!MWW_ABLATE(a, 32)) tile_store1s<SB>(gtile, goff + t * CIN, gg);
          gs1 += gg;
          gs2 = fmaf(gg, fmaf(raw[t], xk1, xk0), gs2);
        }
      }
      if (cch * L < nrows_new && !MWW_ABLATE(a, 64))   // du = 0 past the sample's last output row
        depthwise_weight_grad_chunk<K, L, CPI>(sDU, cch, c, accw, accb,
                                               [&](int row) { return sP[row * CPI + c]; });
    }
    MWW_PC_MARK(6);   // P4 (depthwise backward, stores)
    if (!MWW_ABLATE(a, 8)) __syncthreads();
    MWW_PC_MARK(7);   // barrier 4
  }
  MWW_PC_AT(2);   // tile loop done
  float* gdst = a.grad_part + (size_t)blockIdx.x * ((K + 1) * CIN + CIN * COUT);
  write_block_grad_partials<CIN, COUT, K>(smem, gdst, dwacc, accw, accb, dw_active, c, chunk, tid, wave, r16, g);
  if (dw_active) {
    smem[(chunk * 2 + 0) * CIN + c] = gs1;
    smem[(chunk * 2 + 1) * CIN + c] = gs2;
  }
  __syncthreads();
  if (tid < 2 * CIN) {
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) v += smem[j * 2 * CIN + tid];
    publish_stat(a.gacc, a.gstat_part + (size_t)blockIdx.x * 2 * CIN, 2 * CIN, tid, v);
  }
  MWW_PC_AT(3);   // epilogue done
  MWW_PC_DUMP(a.phase_clk ? a.phase_clk + (size_t)blockIdx.x * kClkSlots : nullptr);
}

// ------------------------------------------------------------------------------------------
// First block: the block input a0 = relu(conv1(x)) has no BN in front and is read back from the tensor fwd_first_kernel
// stored (49 KB/window of traffic instead of recomputing the K1*40-deep im2col GEMM, which was 40 % of this kernel's
// MFMA work and made it the longest launch of the step); instead of an input gradient tensor the kernel produces the
// first-conv weight gradient
//   dW1[j*40+f][c1] += sum_s x[s*S+j][f] * g0[s][c1]      (im2col^T x g0 on MFMA)
struct BwdFirstArgs {
  const float* x;         // [B][T][40]
  const float* a0;        // relu(conv1(x)) [B][Ta][C1]
  const float* pk;        // p_1 [B][Tout][COUT]
  const float* gk;        // g_1 [B][Tout][COUT]
  const float* k_mean;
  const float* k_rstd;
  const float* k_c1;
  const float* k_mg;
  const float* k_mgx;
  const float* dw_w;      // [K][C1]
  const float* dw_b;      // [C1]
  const float* pw_w;      // [C1][COUT]
  float* grad_part;       // [gridDim.x][K1*40*C1 + K*C1 + C1 + C1*COUT]
  int B, T, Tout;         // a0 frames Ta = (T-K1)/S+1 ; Tout = Ta-(K-1)
  BnGradFoldArgs gfold;   // gfold.acc set: k_c1 / k_mg / k_mgx are folded here from the producer's accumulator rows
  XGather xg;             // xg.win set: x rows are gathered from the feature stores (see kernels_fwd.hip.h)
};

template <int K1, int C1, int COUT, int K, int S, bool BF, bool SB = false>
__global__ __launch_bounds__(kThreads, 2) void bwd_first_kernel(BwdFirstArgs a) {
  constexpr int CIN = C1;
  constexpr int CPI = pitch(CIN), CPO = pitch(COUT);
  constexpr int RA = TT + K - 1;
  constexpr int XR = (TT - 1) * S + K1;          // x rows the dW1 contraction of one tile reads
  constexpr int NT1 = C1 / 16;
  constexpr int M1 = K1 * FBINS;                 // rows of W1
  constexpr int MT1 = (M1 + 15) / 16;            // m-tiles of dW1
  constexpr int MPW = (MT1 + 3) / 4;             // m-tiles per wave
  constexpr int MT = CIN / 16, NT = COUT / 16, KSO = COUT / 4;
  constexpr int NCH = nchunks(CIN), L = chunk_len(CIN);
  constexpr int QI = CIN / 4;
  constexpr int RAP = halo_rows_padded(CIN, K), TTP = tile_rows_padded(CIN);
  constexpr int OFF_A = 0, OFF_DP = OFF_A + RAP * CPI, OFF_U = OFF_DP + TT * CPO, OFF_DU = OFF_U + TTP * CPI;
  constexpr int OFF_G0 = OFF_DU + RAP * CPI, OFF_END = OFF_G0 + TTP * CPI;
  constexpr int PX = FBINS + 1;                  // odd pitch of the staged x rows (see fwd_first_kernel)
  static_assert(OFF_END >= 4 * CIN * COUT && OFF_END >= NCH * (K + 1) * CIN, "scratch aliasing");
  static_assert(TT >= K - 1, "shape");

  __shared__ __attribute__((aligned(16))) float sX[XR * PX];
  __shared__ XShared sXg;
  __shared__ __attribute__((aligned(16))) float smem[OFF_END];
  __shared__ __attribute__((aligned(16))) float sKp[7 * COUT];
  __shared__ __attribute__((aligned(16))) float sWt[COUT * CPI];   // W_pw^T
  float* sA = smem + OFF_A;
  float* sDP = smem + OFF_DP;
  float* sU = smem + OFF_U;
  float* sDU = smem + OFF_DU;
  float* sG0 = smem + OFF_G0;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
  const int c = tid % CIN, chunk = tid / CIN;
  const bool dw_active = chunk < NCH;
  const int Ta = (a.T - K1) / S + 1;

  const int ntiles = (Ta + TT - 1) / TT;
  const int nsamp = (int)blockIdx.x < a.B ? (a.B - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const int nitems = nsamp * ntiles;
  XStage<XR, PX> xs;
  DpStage<COUT, false, SB> dps;
  constexpr int NA = (RA * QI + kThreads - 1) / kThreads;
  float4 pre_a[NA];
  auto issue = [&](int it) {
    const int s = it / ntiles, b = blockIdx.x + s * gridDim.x, t0 = (it % ntiles) * TT;
    const int nrx = (min(TT, Ta - t0) - 1) * S + K1;
    xs.issue(a.x, a.xg, sXg, s, b, a.T, t0 * S, nrx, tid);
    const BufRsrc ra = tile_rsrc(a.a0 + ((size_t)b * Ta + t0) * CIN, min(RA, Ta - t0) * QI * 16);
#pragma unroll
    for (int j = 0; j < NA; ++j) pre_a[j] = tile_load4(ra, (tid + j * kThreads) * 16);
    const int nvk = max(0, min(TT, a.Tout - t0)) * (COUT / 4);
    const size_t koff = ((size_t)b * a.Tout + t0) * COUT;
    dps.issue(elem_ptr<SB>(a.pk, koff), elem_ptr<SB>(a.gk, koff), nvk, tid);
  };
  // per-lane offsets of the dW1 rows this wave owns: row m = j*40+f of W1 reads x[s*S+j][f]
  int offm[MPW];
  bool okm[MPW];
#pragma unroll
  for (int mi = 0; mi < MPW; ++mi) {
    const int m = (wave * MPW + mi) * 16 + r16;
    okm[mi] = m < M1;
    offm[mi] = okm[mi] ? (m / FBINS) * PX + (m % FBINS) : 0;
  }
  if (a.xg.win) xgather_setup(a.xg, sXg, nsamp, tid);
  if (nitems > 0) issue(0);

  // every global load of the prologue first (see WeightStage) ...
  WeightStage<CIN, COUT, 0> wst;
  wst.load(a.pw_w, nullptr, tid);
  float dww[K], accw[K];
  float accb = 0.f, dwb = 0.f;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    dww[i] = dw_active ? a.dw_w[i * CIN + c] : 0.f;
    accw[i] = 0.f;
  }
  if (dw_active) dwb = a.dw_b[c];
  // ... then BN_1's backward coefficients
  for (int i = tid; i < COUT; i += kThreads) {
    const float krs = a.k_rstd[i];
    const float kmean = a.k_mean[i];
    float c1, mg, mgx;
    if (a.gfold.acc) {
      bn_grad_fold_channel(a.gfold, COUT, i, krs, c1, mg, mgx);
    } else {
      c1 = a.k_c1[i];
      mg = a.k_mg[i];
      mgx = a.k_mgx[i];
    }
    const float kA = -c1 * krs * mgx;
    sKp[0 * COUT + i] = c1;
    sKp[1 * COUT + i] = kA;
    sKp[2 * COUT + i] = -c1 * mg - kA * kmean;
    sKp[5 * COUT + i] = 0.f;
    sKp[6 * COUT + i] = 0.f;
  }
  wst.store(sWt, nullptr, tid);
  for (int i = RA * CPI + tid; i < RAP * CPI; i += kThreads) {
    sA[i] = 0.f;
    sDU[i] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < K; ++i) pin(dww[i]);
  pin(dwb);
  f32x4 dwacc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) dwacc[mt][nt] = zero4();
  f32x4 w1acc[MPW][NT1];
#pragma unroll
  for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt) w1acc[mi][nt] = zero4();
  __syncthreads();

  for (int it = 0; it < nitems; ++it) {
    rotate_priority(it, 2);
    const int t0 = (it % ntiles) * TT;
    const int nrows_new = max(0, min(TT, a.Tout - t0));
    const int rows_da = min(TT, Ta - t0);
    // ---- P0: commit x (odd pitch), a0 rows [t0, t0+RA) (zero past the sample), dp; roll the du ring
    xs.commit(sX, a.xg, sXg, it / ntiles, t0 * S, tid);
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int i = tid + j * kThreads;
      if (i < RA * QI) {
        const int r = i / QI, q = i - r * QI;
        *reinterpret_cast<float4*>(sA + r * CPI + q * 4) = pre_a[j];
      }
    }
    dps.commit(sDP, sKp, 0.f, nrows_new * (COUT / 4), tid);
    carry_du<K, CPI>(sDU, t0 == 0, tid);
    __syncthreads();
    if (it + 1 < nitems) issue(it + 1);
    // ---- P1: u = depthwise(a0) + bias
    if (dw_active) {
      if (chunk * L < nrows_new) {
        float o[L];
        dw_chunk<K, L>(sA, CPI, chunk * L, c, dww, dwb, o);
#pragma unroll
        for (int t = 0; t < L; ++t) {
          const int tl = chunk * L + t;
          sU[tl * CPI + c] = (tl < nrows_new) ? o[t] : 0.f;
        }
      } else {
#pragma unroll
        for (int t = 0; t < L; ++t) sU[(chunk * L + t) * CPI + c] = 0.f;
      }
    }
    __syncthreads();
    pointwise_backward_tile<CIN, COUT, K, BF>(sU, sDP, sDU, wave, r16, g, sWt, dwacc, wave * 16 < nrows_new);
    __syncthreads();
    // ---- P4: depthwise backward -> g0 = da * relu'(a0) kept in LDS
    if (dw_active) {
      {
        float da[L];
        depthwise_input_grad_chunk<K, L, CPI>(sDU, chunk, c, dww, da);
#pragma unroll
        for (int t = 0; t < L; ++t) {
          const int sl = chunk * L + t;
          sG0[sl * CPI + c] = (sl < rows_da && sA[sl * CPI + c] > 0.f) ? da[t] : 0.f;
        }
      }
      if (chunk * L < nrows_new)
        depthwise_weight_grad_chunk<K, L, CPI>(sDU, chunk, c, accw, accb, [&](int row) { return sA[row * CPI + c]; });
    }
    __syncthreads();
    // ---- dW1 += im2col(x)^T g0 : A[m][k=s] = x[s*S + m/40][m%40], B[k=s][n] = g0[s][n]
    // (operands one k-step ahead of their MFMAs, schedule pinned: see pointwise_backward_tile)
    {
      float av[2][MPW], bv[2][NT1];
      auto load_w1 = [&](int kk, int sl) {
        const int s = kk * 4 + g;
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) bv[sl][nt] = sG0[s * CPI + nt * 16 + r16];
#pragma unroll
        for (int mi = 0; mi < MPW; ++mi) {
          const float v = sX[s * S * PX + offm[mi]];   // (rows past M1 read offset 0 and are zeroed: no conditional load)
          av[sl][mi] = okm[mi] ? v : 0.f;
        }
      };
      load_w1(0, 0);
#pragma unroll
      for (int kk = 0; kk < TT / 4; ++kk) {
        if (kk + 1 < TT / 4) load_w1(kk + 1, (kk + 1) & 1);
#pragma unroll
        for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
          for (int nt = 0; nt < NT1; ++nt) w1acc[mi][nt] = mfma4(av[kk & 1][mi], bv[kk & 1][nt], w1acc[mi][nt]);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, MPW + NT1, 0);
      sched_read_mfma_groups<TT / 4 - 1, MPW + NT1, MPW * NT1>();
      sched_read_mfma_groups<1, 0, MPW * NT1>();
    }
    __syncthreads();
  }
  float* gdst = a.grad_part + (size_t)blockIdx.x * (M1 * C1 + (K + 1) * CIN + CIN * COUT);
#pragma unroll
  for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = (wave * MPW + mi) * 16 + g * 4 + r;
        if (m < M1) gdst[m * C1 + nt * 16 + r16] = w1acc[mi][nt][r];
      }
  write_block_grad_partials<CIN, COUT, K>(smem, gdst + M1 * C1, dwacc, accw, accb, dw_active, c, chunk, tid, wave, r16, g);
}

// ------------------------------------------------------------------------------------------
// BN backward coefficients from the (sum g, sum g*xhat) partials; also emits dgamma / dbeta.
struct BnBwdFinalizeArgs {
  const float* gstat_part;  // [G][2][C]
  int G, C;
  float inv_n;              // 1/(B*T)
  const float* gamma;
  const float* rstd;
  float* c1;                // gamma*rstd
  float* mg;                // mean g
  float* mgx;               // mean g*xhat
  float* dgamma;            // -> flat gradient
  float* dbeta;
  float dscale;             // 1 (local statistics) or 1/W (sums already all-reduced: the gradient all-reduce adds them W times)
};

__device__ __forceinline__ void bn_bwd_finalize_body(const BnBwdFinalizeArgs& a, int c, double* sAcc, double* sOut, int tid) {
  float gam = 0.f, rs = 0.f;
  if (tid == 0) {
    gam = a.gamma[c];
    rs = a.rstd[c];
  }
  const double r = reduce_partials_256(a.gstat_part, a.G, a.C, c, sAcc, tid);
  if ((tid & 127) == 0) sOut[tid >> 7] = r;
  __syncthreads();
  if (tid == 0) {
    const double s1 = sOut[0], s2 = sOut[1];
    a.dbeta[c] = (float)s1 * a.dscale;
    a.dgamma[c] = (float)s2 * a.dscale;
    a.c1[c] = gam * rs;
    a.mg[c] = (float)(s1 * (double)a.inv_n);
    a.mgx[c] = (float)(s2 * (double)a.inv_n);
  }
}

__global__ __launch_bounds__(kThreads) void bn_bwd_finalize_kernel(BnBwdFinalizeArgs a) {
  __shared__ __attribute__((aligned(16))) double sAcc[256 + 16];
  __shared__ double sOut[2];
  bn_bwd_finalize_body(a, blockIdx.x, sAcc, sOut, threadIdx.x);
}

// ------------------------------------------------------------------------------------------
// Gradient assembly: the per-workgroup partial rows of every segment are summed in a fixed order by
// grad_final_kernel (kernels_tail.hip.h); the host lists the segments of a step here.
struct GradSegment {
  const float* part;   // [G][stride]
  int G;
  int stride;          // floats between consecutive workgroups' partials
  int n;               // parameters in this segment
  int dst;             // offset in the flat gradient
};
constexpr int kMaxSegments = 112;
struct GradReduceArgs {   // host-side list (not a kernel argument)
  GradSegment seg[kMaxSegments];
  int nseg;
};

// Keras Adam: alpha = lr*sqrt(1-b2^t)/(1-b1^t) computed on the host per step.
struct AdamArgs {
  float* param;
  const float* grad;
  float* m;
  float* v;
  const float* hyper;   // mailbox (mapped host memory): [0] = alpha, [1] = grad scale (1/world for averaged all-reduce)
  int P;
  float beta1, beta2, eps;
};

__global__ __launch_bounds__(kThreads) void adam_kernel(AdamArgs a) {
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= a.P) return;
  const float alpha = a.hyper[0];
  const float gg = a.grad[p] * a.hyper[1];
  float m = a.m[p], v = a.v[p];
  m += (gg - m) * (1.0f - a.beta1);
  v += (gg * gg - v) * (1.0f - a.beta2);
  a.m[p] = m;
  a.v[p] = v;
  a.param[p] -= alpha * m / (sqrtf(v) + a.eps);
}

}  // namespace mww

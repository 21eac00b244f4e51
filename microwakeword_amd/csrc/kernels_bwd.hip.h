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
      pk[j] = tile_load4s<SB, MWW_AUX_LD_PK>(rp, tid + j * kThreads);
      gg[j] = tile_load4s<SG, MWW_AUX_LD_GK>(rg, tid + j * kThreads);
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
template <int CIN, int COUT, int KD, int PWT = pitch(CIN)>
struct WeightStage {
  static constexpr int CPI = PWT, NW = (CIN * COUT + kThreads - 1) / kThreads, ND = KD > 0 ? (KD * CIN + kThreads - 1) / kThreads : 1;
  float w[NW], d[ND];
  __device__ __forceinline__ void load(const float* pw_w, const float* dw_w, int tid) {
#pragma unroll
    for (int j = 0; j < NW; ++j) w[j] = (tid + j * kThreads < CIN * COUT) ? pw_w[tid + j * kThreads] : 0.f;
#pragma unroll
    for (int j = 0; j < ND; ++j) d[j] = (KD > 0 && tid + j * kThreads < KD * CIN) ? dw_w[tid + j * kThreads] : 0.f;
  }
  // sWt [COUT][PWT] = W_pw^T; sDW [KD][CIN] (KD = 0: the taps stay in registers, nothing to stage)
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
  constexpr int PWT = pitch_wt(CIN, BF);
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
        // k-step kk meets the rows kk, kk + 4, kk + 8, kk + 12 of the wave's 16: the two rows of a 32-lane LDS group are four
        // apart, and 4 * (c + 4) = 16 (mod 32) for c = 32, 48, 64 puts them on disjoint banks (rows one apart: two-way on every read)
        const int row = wave * 16 + kk + 4 * g;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[s][mt] = sU[row * CPI + mt * 16 + r16];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[s][nt] = sDP[row * CPO + nt * 16 + r16];
      };
      float a2[2], b2[2][MT];
      auto load_du = [&](int kk, int s) {
        a2[s] = sDP[(wave * 16 + r16) * CPO + kk * 4 + g];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) b2[s][mt] = sWt[(kk * 4 + g) * PWT + mt * 16 + r16];
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
        const float* col = sWt + (kk * 16 + 4 * g) * PWT + mt * 16 + r16;
        du[mt] = mfma_bf16(a4, to_bf16x4(col[0], col[PWT], col[2 * PWT], col[3 * PWT]), du[mt]);
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
    store_stream<MWW_AUX_ST_GP>(dst + (K + 1) * CIN + e, (scratch[e] + scratch[CIN * COUT + e]) + (scratch[2 * CIN * COUT + e] + scratch[3 * CIN * COUT + e]));
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
    store_stream<MWW_AUX_ST_GP>(dst + e, v);
  }
  __syncthreads();
}

// depthwise backward of one (channel, chunk) for one time tile (phase P4 of the two backward bodies, written out there):
//   da[sl]      = sum_j w[K-1-j] * du_ring[sl + j]                     (sl local input row)
//   dW_dw[i]   += sum_t du[t] * a[t+i] ;  db += sum_t du[t]            (t local output row)
// Every LDS read of the phase is issued before the first FMA (read -> wait -> use per element exposes one LDS round trip
// per output row: the round-2 ISA of this phase was a chain of lgkmcnt(0) waits), and the du-ring / activation windows
// are read once and serve both sums.

// ------------------------------------------------------------------------------------------
// LDS of the block / first-block backward stages as float offsets into a fused launch's LDS array
template <int CIN, int COUT, int K>
struct BwdBlockLds {
  static constexpr int CPI = pitch(CIN), CPO = pitch(COUT);
  static constexpr int RAP = halo_rows_padded(CIN, K), TTP = tile_rows_padded(CIN);
  static constexpr int OFF_END = RAP * CPI + TT * CPO + TTP * CPI + RAP * CPI;
  static constexpr int up4(int v) { return (v + 3) / 4 * 4; }
  static constexpr int KP = OFF_END, WT = KP + up4(7 * COUT), DW = WT + COUT * pitch_wt(CIN, false), ACT = DW + up4(K * CIN), END = ACT + 2 * CIN;
};
template <int K1, int C1, int COUT, int K, int S>
struct BwdFirstLds {
  static constexpr int CIN = C1, CPI = pitch(CIN), CPO = pitch(COUT);
  static constexpr int RAP = halo_rows_padded(CIN, K), TTP = tile_rows_padded(CIN);
  static constexpr int TAIL = S > 1 ? K - 1 : 0;   // extra a0 / g0 rows of a single-tile window (bwd_first_body.inc "tail rows")
  static constexpr int TAILK = (TAIL + 3) / 4 * 4;   // the tail k-steps of dW1 read whole k-steps of four rows
  static constexpr int OFF_END = RAP * CPI + TT * CPO + TTP * CPI + RAP * CPI + (TTP + TAILK) * CPI;
  static constexpr int XR = (TT + TAILK - 1) * S + K1, PX = FBINS + 1;
  static constexpr int up4(int v) { return (v + 3) / 4 * 4; }
  static constexpr int KP = OFF_END, WT = KP + up4(7 * COUT), X = WT + COUT * pitch_wt(CIN, false), XG = X + up4(XR * PX);
  static constexpr int END = XG + up4((int)(sizeof(XShared) + 3) / 4);
};

// (64-wide blocks - and 48 -> 64 blocks with long kernels - hold 81-107 KB of LDS per workgroup: one workgroup per CU whatever
// the register count, so they are compiled for one - the bound of 2 they carried until round 3 could not be met and only produced "failed to meet occupancy target")
template <int CIN, int COUT, int K, bool LAST, bool BF, bool SB = false>
__global__ __launch_bounds__(kThreads, ((CIN > 48 || BwdBlockLds<CIN, COUT, K>::END * 4 > 80 * 1024) ? 1 : 2)) void bwd_block_kernel(BwdBlockArgs a) {
  typedef BwdBlockLds<CIN, COUT, K> Lds;
  __shared__ __attribute__((aligned(16))) float smem[Lds::OFF_END];
  __shared__ __attribute__((aligned(16))) float sKp[7 * COUT];
  __shared__ __attribute__((aligned(16))) float sWt[COUT * pitch_wt(CIN, BF)];   // W_pw^T
  __shared__ __attribute__((aligned(16))) float sDW[K * CIN];      // depthwise taps
  __shared__ __attribute__((aligned(16))) float sAct[2 * CIN];     // BN_{k-1} folded scale / shift (activation at commit)
#include "bwd_block_body.inc"
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

// (stride-3 first convolutions stage 194 x rows per tile: 92-99 KB of LDS, one workgroup per CU)
template <int K1, int C1, int COUT, int K, int S, bool BF, bool SB = false, bool X6 = false>
__global__ __launch_bounds__(kThreads, (S > 1 ? 1 : 2)) void bwd_first_kernel(BwdFirstArgs a) {
  typedef BwdFirstLds<K1, C1, COUT, K, S> Lds;
  __shared__ __attribute__((aligned(16))) float sX[X6 ? 3 * Lds::XR * 96 / 4 : Lds::XR * Lds::PX];   // X6: three bf16 planes, 96-byte rows
  __shared__ XShared sXg;
  __shared__ __attribute__((aligned(16))) float smem[X6 ? Lds::OFF_END - (Lds::TTP + Lds::TAILK) * Lds::CPI : Lds::OFF_END];   // X6: g0 lives in the dp tile
  __shared__ __attribute__((aligned(16))) float sKp[7 * COUT];
  __shared__ __attribute__((aligned(16))) float sWt[COUT * pitch_wt(C1, BF)];   // W_pw^T
#include "bwd_first_body.inc"
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

#ifndef MWW_BLOCK_TU   // defined once, in mww_lib.hip (the block-kernel translation units skip it)
__global__ __launch_bounds__(kThreads) void bn_bwd_finalize_kernel(BnBwdFinalizeArgs a) {
  __shared__ __attribute__((aligned(16))) double sAcc[256 + 16];
  __shared__ double sOut[2];
  bn_bwd_finalize_body(a, blockIdx.x, sAcc, sOut, threadIdx.x);
}
#endif

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

#ifndef MWW_BLOCK_TU   // defined once, in mww_lib.hip (the block-kernel translation units skip it)
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
#endif

}  // namespace mww

// Wide-workgroup form of the block backward kernel (round 5).  Same arithmetic, same inputs / outputs / partial-row
// layout as bwd_block_kernel (kernels_bwd.hip.h, reference: TF autodiff of microwakeword/mixednet.py:334-360), but a
// 64-row time tile is shared by NTH = 512 threads instead of 256, with a register budget (128) that lets a CU hold four
// waves per SIMD instead of two (a 384-thread form - six waves, three per SIMD - was built and lost: 56-63 us per launch
// against 44-54, profiles/round5_bwd_forms_ab.txt; six waves leave two of the four SIMDs with twice the MFMA work):
//
//   * VALU phases: thread -> (channel, chunk) with NTH / C chunks of L = 7 (C = 48) or 8 (C = 64) rows (256 threads: 13 / 16): the register
//     windows of the depthwise phases shrink from L + K - 1 = 33 to 27 values at K = 21, and the one-pass depthwise
//     backward is split into an input-gradient pass and a weight-gradient pass that reuse the same registers;
//   * MFMA phase: waves own OUTPUT tiles instead of row slices.  A wave of the 256-thread kernel accumulates the whole
//     C x C weight gradient over its 16 rows (36 accumulator registers at C = 48); here a wave owns at most three
//     16 x 16 tiles of dW_pw over 32 or 64 rows (12 registers) and one to three 16 x 16 tiles of du (WideRoles below);
//   * LDS pitches chosen per access pattern against the bank model of tools/lds_banks.py (WidePitch below): the
//     (channel, chunk) windows, the row-pattern MFMA operands and the column-pattern operand are conflict-free or at
//     the conflict-free rate (C = 48 / 512 threads: pitch 48 = no padding, the du A operand read as one float4 per lane).
//
// The weight-gradient partials keep a fixed summation order (per-wave tiles, then row halves, then the grid's partial
// rows in grad_final_kernel): gradients stay bit-reproducible from run to run.  The k-order of the contractions differs
// from the 256-thread kernel's (other rows meet in one MFMA k-step), so the two forms agree to rounding, not bitwise.
#pragma once
#include "kernels_bwd.hip.h"

// tuning switches of the wide block backward (see the tile loop of bwd_blockw_kernel and TapGroups)
#ifndef MWW_WIDE_RELANE_K
#define MWW_WIDE_RELANE_K 15
#endif
#ifndef MWW_WIDE_TWO_GROUPS_UPTO   // longest depthwise kernel whose phases run in two tap groups (three beyond)
#define MWW_WIDE_TWO_GROUPS_UPTO 19
#endif

namespace mww {

// P: row pitch of the activation / gradient tiles; D: row distance between the four k-values of one dW k-step
// (rows s + D g of a block of 4 D rows, g = lane >> 4: a 32-lane LDS group holds two of them and D * P = 16 mod 32 keeps
// their banks disjoint).  The du A operand (lanes across rows, fixed column) is read as one float4 per lane = the lane's k of
// four k-steps (k = 16 kb + 4 g + s: two-way conflicts at P = 48, i.e. the rate of four conflict-free dword reads; none at
// 72); PW: pitch of W^T, whose rows are then read 4 apart (4 PW = 16 mod 32).
template <int C, int NTH>
struct WidePitch;
template <>
struct WidePitch<48, 512> {
  static constexpr int P = 48, D = 1, PW = 52;
};
template <>
struct WidePitch<64, 512> {
  static constexpr int P = 72, D = 2, PW = 68;
};

// dp tile of the wide kernel (DpStage with the thread count and the pitch as parameters)
template <int C, bool LAST, bool SB, int NTH, int P>
struct DpStageW {
  static constexpr int Q = C / 4, N = (TT * Q + NTH - 1) / NTH;
  float4 pk[N], gg[N];

  __device__ __forceinline__ void issue(const float* pk_base, const float* g_base, int nvalid, int tid) {
    constexpr bool SG = SB && !LAST;
    const BufRsrc rp = tile_rsrc(pk_base, nvalid * 4 * elem_bytes(SB)), rg = tile_rsrc(g_base, nvalid * 4 * elem_bytes(SG));
#pragma unroll
    for (int j = 0; j < N; ++j) {
      pk[j] = tile_load4s<SB, MWW_AUX_LD_PK>(rp, tid + j * NTH);
      gg[j] = tile_load4s<SG, MWW_AUX_LD_GK>(rg, tid + j * NTH);
    }
  }

  // sKp rows: 0 = c1, 1 = kA, 2 = kB (dp = c1 g + kA p + kB), 5 / 6 = BN_k scale / shift (LAST: ReLU mask of the head input)
  __device__ __forceinline__ void commit(float* sDP, const float* sKp, float dzb, int nvalid, int tid) const {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int i = tid + j * NTH;
      if (i < TT * Q) {
        const int r = i / Q, q = i - r * Q;
        float4 dp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nvalid) {
          const float4 p = pk[j];
          const float4 c1 = *reinterpret_cast<const float4*>(sKp + 0 * C + q * 4);
          const float4 kA = *reinterpret_cast<const float4*>(sKp + 1 * C + q * 4);
          const float4 kB = *reinterpret_cast<const float4*>(sKp + 2 * C + q * 4);
          float4 g = gg[j];
          if (LAST) {
            const float4 sc = *reinterpret_cast<const float4*>(sKp + 5 * C + q * 4);
            const float4 sh = *reinterpret_cast<const float4*>(sKp + 6 * C + q * 4);
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
        *reinterpret_cast<float4*>(sDP + r * P + q * 4) = dp;
      }
    }
  }
};

// dW tiles of one wave: acc[nt] += U[rows, mcol..]^T DP[rows, ncol0 + 16 nt ..] over NBLK blocks of BLK rows from row0;
// blocks that start past the sample's last output row are skipped (u and dp are zero there).  The operands of k-step
// kk + 1 are read before the MFMAs of k-step kk.
template <int NTW, int NBLK, int BLK, int D, int PA, int PB = PA>
__device__ __forceinline__ void wide_dw_rows(const float* sU, const float* sDP, int row0, int nrows, int mcol, int ncol0,
                                             int r16, int g, f32x4 (&acc)[NTW]) {
  constexpr int KSB = BLK / 4;
  static_assert(BLK % (4 * D) == 0, "a block holds whole groups of 4 D rows");
  const float* pu = sU + (row0 + D * g) * PA + mcol + r16;
  const float* pd = sDP + (row0 + D * g) * PB + ncol0 + r16;
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk) {
    if (row0 + blk * BLK < nrows) {   // wave-uniform
      float av[2], bv[2][NTW];
      auto ld = [&](int kk, int s) {
        const int ro = blk * BLK + (kk / D) * 4 * D + (kk % D);
        av[s] = pu[ro * PA];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) bv[s][nt] = pd[ro * PB + nt * 16];
      };
      ld(0, 0);
#pragma unroll
      for (int kk = 0; kk < KSB; ++kk) {
        if (kk + 1 < KSB) ld(kk + 1, (kk + 1) & 1);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[nt] = mfma4(av[kk & 1], bv[kk & 1][nt], acc[nt]);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 1 + NTW, 0);
      sched_read_mfma_groups<KSB - 1, 1 + NTW, NTW>();
      sched_read_mfma_groups<1, 0, NTW>();
    }
  }
}

// du tiles of one wave: du[mu] = DP[16 rows of tile rt, 0..C) W^T[0..C, mcol0 + 16 mu ..] (C = the block's output
// channels = the contraction length; P / PW / PD = pitches of dp, W^T and the du ring), stored to the ring rows
// [K-1 + 16 rt, +16).  A tile past the sample's last output row only writes its zeros.
template <int C, int NMU, int K, int P, int PW, int PD = P>
__device__ __forceinline__ void wide_du_tiles(const float* sDP, const float* sWt, float* sDU, int rt, int nrows, int mcol0,
                                              int r16, int g) {
  f32x4 du[NMU];
#pragma unroll
  for (int mu = 0; mu < NMU; ++mu) du[mu] = zero4();
  if (rt * 16 < nrows) {   // wave-uniform
    const float* pa = sDP + (rt * 16 + r16) * P + 4 * g;
    const float* pb = sWt + (4 * g) * PW + mcol0 + r16;
    float4 a4[2];
    a4[0] = *reinterpret_cast<const float4*>(pa);
#pragma unroll
    for (int kb = 0; kb < C / 16; ++kb) {
      if (kb + 1 < C / 16) a4[(kb + 1) & 1] = *reinterpret_cast<const float4*>(pa + (kb + 1) * 16);
      float bv[4][NMU];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mu = 0; mu < NMU; ++mu) bv[s][mu] = pb[(kb * 16 + s) * PW + mu * 16];
      const float4 av = a4[kb & 1];
      const float as[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mu = 0; mu < NMU; ++mu) du[mu] = mfma4(as[s], bv[s][mu], du[mu]);
    }
  }
  float* pd = sDU + (K - 1 + rt * 16 + g * 4) * PD + mcol0 + r16;
#pragma unroll
  for (int mu = 0; mu < NMU; ++mu)
#pragma unroll
    for (int r = 0; r < 4; ++r) pd[r * PD + mu * 16] = du[mu][r];
}

// bf16-operand forms of the two contractions (BASELINE configs[4], "pointwise_bf16": operands rounded to bf16, products
// exact, fp32 accumulation; one v_mfma_f32_16x16x16_bf16 covers 16 rows / 16 channels of the contraction).  The lane's four
// k-values of an MFMA are rows g + 4 j of the 16-row block (dW: one row per read instruction, the two g of a 32-lane
// group one row apart - conflict-free at a pitch of 16 mod 32) or the channels 4 g + j of a float4 (du).
template <int NTW, int NBLK, int PA, int PB = PA>
__device__ __forceinline__ void wide_dw_rows_bf16(const float* sU, const float* sDP, int row0, int nrows, int mcol, int ncol0,
                                                  int r16, int g, f32x4 (&acc)[NTW]) {
  const float* pu = sU + (row0 + g) * PA + mcol + r16;
  const float* pd = sDP + (row0 + g) * PB + ncol0 + r16;
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk) {
    if (row0 + blk * 16 < nrows) {   // wave-uniform
      const float* cu = pu + blk * 16 * PA;
      const bf16x4 av = to_bf16x4(cu[0], cu[4 * PA], cu[8 * PA], cu[12 * PA]);
      bf16x4 bv[NTW];
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const float* cd = pd + blk * 16 * PB + nt * 16;
        bv[nt] = to_bf16x4(cd[0], cd[4 * PB], cd[8 * PB], cd[12 * PB]);
      }
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) acc[nt] = mfma_bf16(av, bv[nt], acc[nt]);
    }
  }
}

template <int C, int NMU, int K, int P, int PW, int PD = P>
__device__ __forceinline__ void wide_du_tiles_bf16(const float* sDP, const float* sWt, float* sDU, int rt, int nrows, int mcol0,
                                                   int r16, int g) {
  f32x4 du[NMU];
#pragma unroll
  for (int mu = 0; mu < NMU; ++mu) du[mu] = zero4();
  if (rt * 16 < nrows) {   // wave-uniform
    const float* pa = sDP + (rt * 16 + r16) * P + 4 * g;
    const float* pb = sWt + (4 * g) * PW + mcol0 + r16;
#pragma unroll
    for (int kb = 0; kb < C / 16; ++kb) {
      const float4 v = *reinterpret_cast<const float4*>(pa + kb * 16);
      const bf16x4 a4 = to_bf16x4(v.x, v.y, v.z, v.w);
#pragma unroll
      for (int mu = 0; mu < NMU; ++mu) {
        const float* col = pb + kb * 16 * PW + mu * 16;
        du[mu] = mfma_bf16(a4, to_bf16x4(col[0], col[PW], col[2 * PW], col[3 * PW]), du[mu]);
      }
    }
  }
  float* pd = sDU + (K - 1 + rt * 16 + g * 4) * PD + mcol0 + r16;
#pragma unroll
  for (int mu = 0; mu < NMU; ++mu)
#pragma unroll
    for (int r = 0; r < 4; ++r) pd[r * PD + mu * 16] = du[mu][r];
}

// Depthwise sums over a sub-range [I0, I1) of the K taps, for one (channel, chunk): the register window of a phase is
// L + (I1 - I0) - 1 rows instead of L + K - 1, so long kernels run their phases in two or three tap groups.
//   dw_tap_group     : acc[t] += sum_i w(i) * src[t + i]      (w(i) = taps[i], or taps[K-1-i] when REV: the input gradient)
//   dw_wgrad_group   : accw[i] += sum_t du[t] * src[t + i]    (the depthwise weight gradient)
// src_c / taps_c point at the thread's channel (row 0 of its window, tap 0).
template <int K, int L, int I0, int I1, bool REV>
__device__ __forceinline__ void dw_tap_group(const float* src_c, int pitch, const float* taps_c, int tap_pitch, float (&acc)[L]) {
  constexpr int N = I1 - I0;
  float win[L + N - 1], w[N];
#pragma unroll
  for (int j = 0; j < L + N - 1; ++j) win[j] = src_c[(I0 + j) * pitch];
#pragma unroll
  for (int i = 0; i < N; ++i) w[i] = taps_c[(REV ? K - 1 - (I0 + i) : I0 + i) * tap_pitch];
#ifndef MWW_WIDE_NOREADSFIRST
  lds_reads_first();
#endif
#pragma unroll
  for (int t = 0; t < L; ++t)
#pragma unroll
    for (int i = 0; i < N; ++i) acc[t] = fmaf(w[i], win[t + i], acc[t]);
}
template <int K, int L, int I0, int I1>
__device__ __forceinline__ void dw_wgrad_group(const float* src_c, int pitch, const float (&du)[L], float (&accw)[K]) {
  constexpr int N = I1 - I0;
  float win[L + N - 1];
#pragma unroll
  for (int j = 0; j < L + N - 1; ++j) win[j] = src_c[(I0 + j) * pitch];
#ifndef MWW_WIDE_NOREADSFIRST
  lds_reads_first();
#endif
#pragma unroll
  for (int t = 0; t < L; ++t)
#pragma unroll
    for (int i = 0; i < N; ++i) accw[I0 + i] = fmaf(du[t], win[t + i], accw[I0 + i]);
}
// tap groups of a K-tap kernel: one up to 11 taps, two up to 19, three beyond
template <int K>
struct TapGroups {
#ifdef MWW_WIDE_FEWGROUPS   // tuning builds: larger register windows, fewer scheduling fences
  static constexpr int N = K <= 15 ? 1 : 2;
#else
  static constexpr int N = K <= 11 ? 1 : (K <= MWW_WIDE_TWO_GROUPS_UPTO ? 2 : 3);
#endif
  static constexpr int lo(int gidx) { return gidx * K / N; }
};
template <int K, int L, bool REV, int G = 0>
__device__ __forceinline__ void dw_all_groups(const float* src_c, int pitch, const float* taps_c, int tap_pitch, float (&acc)[L]) {
  if constexpr (G < TapGroups<K>::N) {
    dw_tap_group<K, L, TapGroups<K>::lo(G), TapGroups<K>::lo(G + 1), REV>(src_c, pitch, taps_c, tap_pitch, acc);
    if constexpr (G + 1 < TapGroups<K>::N) __builtin_amdgcn_sched_barrier(0);
    dw_all_groups<K, L, REV, G + 1>(src_c, pitch, taps_c, tap_pitch, acc);
  }
}
template <int K, int L, int G = 0>
__device__ __forceinline__ void dw_wgrad_all_groups(const float* src_c, int pitch, const float (&du)[L], float (&accw)[K]) {
  if constexpr (G < TapGroups<K>::N) {
    dw_wgrad_group<K, L, TapGroups<K>::lo(G), TapGroups<K>::lo(G + 1)>(src_c, pitch, du, accw);
    if constexpr (G + 1 < TapGroups<K>::N) __builtin_amdgcn_sched_barrier(0);
    dw_wgrad_all_groups<K, L, G + 1>(src_c, pitch, du, accw);
  }
}

// MFMA work of the waves of one workgroup (288 MFMAs per tile at C = 48, 512 at C = 64, dealt evenly):
//   C = 48, 8 waves: waves 0-5 own the dW tiles (mt = w % 3, nt = 0..2) over the row half w / 3 (24 MFMAs) and the du
//                    tile (rt = 2 + w / 3, mt = w % 3) (12); waves 6, 7 own the du tiles (rt = w - 6, mt = 0..2) (36)
//   C = 64, 8 waves: every wave owns the dW tiles (mt = w / 2, nt = 2 (w % 2) + {0, 1}) over all rows (32) and the du
//                    tiles (rt = w / 2, mt = 2 (w % 2) + {0, 1}) (32)
template <int C, int NW>
struct WideRoles;
template <>
struct WideRoles<48, 8> {
  static constexpr int NTW = 3, NPART = 2;
  static __device__ __forceinline__ bool has_dw(int w) { return w < 6; }
  static __device__ __forceinline__ int part(int w) { return w / 3; }
  static __device__ __forceinline__ int mt(int w) { return w % 3; }
  static __device__ __forceinline__ int nt0(int) { return 0; }
};
template <>
struct WideRoles<64, 8> {
  static constexpr int NTW = 2, NPART = 1;
  static __device__ __forceinline__ bool has_dw(int) { return true; }
  static __device__ __forceinline__ int part(int) { return 0; }
  static __device__ __forceinline__ int mt(int w) { return w / 2; }
  static __device__ __forceinline__ int nt0(int w) { return 2 * (w % 2); }
};

template <int C, int K, int NTH>
struct BwdWideLds {
  typedef WidePitch<C, NTH> WP;
  static constexpr int NCH = NTH / C, L = (TT + NCH - 1) / NCH, TTP = NCH * L, RAP = TTP + K - 1;
  static constexpr int OFF_P = 0, OFF_DP = OFF_P + RAP * WP::P, OFF_U = OFF_DP + TT * WP::P, OFF_DU = OFF_U + TTP * WP::P;
  static constexpr int OFF_END = OFF_DU + RAP * WP::P;
  static constexpr int BYTES = (OFF_END + 7 * C + C * WP::PW + K * C + 2 * C) * 4;
};

// waves per SIMD the kernel is compiled for: two workgroups per CU where their LDS allows it
template <int C, int K, int NTH>
constexpr int wide_waves_per_simd() {
  return (BwdWideLds<C, K, NTH>::BYTES <= 80 * 1024 ? 2 : 1) * (NTH / 64) / 4;
}

template <int CIN, int COUT, int K, bool LAST, int NTH, bool BF = false, bool SB = false>
__global__ __launch_bounds__(NTH, (wide_waves_per_simd<CIN, K, NTH>())) void bwd_blockw_kernel(BwdBlockArgs a) {
  static_assert(CIN == COUT, "the wide form is instantiated for square blocks");
  constexpr int C = CIN;
  typedef WidePitch<C, NTH> WP;
  typedef BwdWideLds<C, K, NTH> Lds;
  constexpr int NW = NTH / 64;
  typedef WideRoles<C, NW> Roles;
  constexpr int P = WP::P, PW = WP::PW, D = WP::D;
  constexpr int NCH = Lds::NCH, L = Lds::L, TTP = Lds::TTP, RA = TT + K - 1, RAP = Lds::RAP;
  constexpr int Q = C / 4, MT = C / 16;
  static_assert(NCH * L >= TT && TT >= K - 1, "tile geometry");
  static_assert(Lds::OFF_END >= Roles::NPART * C * C && Lds::OFF_END >= NCH * (K + 1) * C, "scratch aliasing");

  __shared__ __attribute__((aligned(16))) float smem[Lds::OFF_END];
  __shared__ __attribute__((aligned(16))) float sKp[7 * C];
  __shared__ __attribute__((aligned(16))) float sWt[C * PW];   // W_pw^T
  __shared__ __attribute__((aligned(16))) float sDW[K * C];    // depthwise taps
  __shared__ __attribute__((aligned(16))) float sAct[2 * C];   // BN_{k-1} folded scale / shift
  float* sP = smem + Lds::OFF_P;
  float* sDP = smem + Lds::OFF_DP;
  float* sU = smem + Lds::OFF_U;
  float* sDU = smem + Lds::OFF_DU;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r16_ = lane & 15, g_ = lane >> 4;
  const int c = tid % C, chunk = tid / C;
  const bool dw_active = chunk < NCH;
  MWW_PC_DECL
  MWW_PC_AT(0);   // kernel entry (profiling builds only: tools/phase_clocks.py)

  const int ntiles = (a.Tin + TT - 1) / TT;
  const int nsamp = (int)blockIdx.x < a.B ? (a.B - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const int nitems = nsamp * ntiles;
  constexpr int NP = (RA * Q + NTH - 1) / NTH;
  float4 pre_p[NP];
  DpStageW<C, LAST, SB, NTH, P> dps;
  float pre_dz = 0.f;
  auto issue = [&](int it) {
    const int b = blockIdx.x + (it / ntiles) * gridDim.x, t0 = (it % ntiles) * TT;
    const int nvp = min(RA, a.Tin - t0) * Q;
    const BufRsrc src = tile_rsrc(elem_ptr<SB>(a.in, ((size_t)b * a.Tin + t0) * C), nvp * 4 * elem_bytes(SB));
#pragma unroll
    for (int j = 0; j < NP; ++j) pre_p[j] = tile_load4s<SB, MWW_AUX_LD_BP>(src, tid + j * NTH);
    const int nvk = max(0, min(TT, a.Tout - t0)) * Q;
    const size_t koff = ((size_t)b * a.Tout + t0) * C;
    dps.issue(elem_ptr<SB>(a.pk, koff), LAST ? a.wd + (size_t)t0 * C : elem_ptr<SB>(a.gk, koff), nvk, tid);
    if (LAST) pre_dz = tile_load1(tile_rsrc(a.dz, a.B * 4), b * 4);
  };
  stagger_start<MWW_STAGGER_BWD>();
  if (nitems > 0) issue(0);

  // ---- prologue: every global load first (one memory round trip), then the LDS copies
  constexpr int NWL = (C * C + NTH - 1) / NTH, NDL = (K * C + NTH - 1) / NTH;
  float wl[NWL], dl[NDL];
#pragma unroll
  for (int j = 0; j < NWL; ++j) wl[j] = (tid + j * NTH < C * C) ? a.pw_w[tid + j * NTH] : 0.f;
#pragma unroll
  for (int j = 0; j < NDL; ++j) dl[j] = (tid + j * NTH < K * C) ? a.dw_w[tid + j * NTH] : 0.f;
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
  for (int i = tid; i < C; i += NTH) {
    const float krs = a.k_rstd[i];
    const float kmean = a.k_mean[i];
    const float ksc = LAST ? a.k_scale[i] : 0.f, ksh = LAST ? a.k_shift[i] : 0.f;
    float c1, mg, mgx;
    if (a.gfold.acc) {
      bn_grad_fold_channel(a.gfold, C, i, krs, c1, mg, mgx);
    } else {
      c1 = a.k_c1[i];
      mg = a.k_mg[i];
      mgx = a.k_mgx[i];
    }
    const float kA = -c1 * krs * mgx;
    sKp[0 * C + i] = c1;
    sKp[1 * C + i] = kA;
    sKp[2 * C + i] = -c1 * mg - kA * kmean;
    sKp[5 * C + i] = ksc;
    sKp[6 * C + i] = ksh;
  }
#pragma unroll
  for (int j = 0; j < NWL; ++j) {
    const int i = tid + j * NTH;
    if (i < C * C) sWt[(i % C) * PW + i / C] = wl[j];   // W[ci][co] -> W^T[co][ci]
  }
#pragma unroll
  for (int j = 0; j < NDL; ++j)
    if (tid + j * NTH < K * C) sDW[tid + j * NTH] = dl[j];
  for (int i = RA * P + tid; i < RAP * P; i += NTH) {   // rows only the padded windows touch
    sP[i] = 0.f;
    sDU[i] = 0.f;
  }
  f32x4 dwacc[Roles::NTW];
#pragma unroll
  for (int nt = 0; nt < Roles::NTW; ++nt) dwacc[nt] = zero4();
  if (chunk == 0) {
    sAct[c] = sc_c;
    sAct[C + c] = sh_c;
  }
  // sP holds a = relu(BN_{k-1}(p_{k-1})); the ReLU decision of a unit is a > 0 and its x-hat = a * xk1 + xk0 (see bwd_block_body.inc)
  float xk1 = sc_c != 0.f ? rs_c / sc_c : 0.f;
  float xk0 = -(sh_c + mu_c * sc_c) * xk1;
  pin(dwb); pin(xk1); pin(xk0);
  __syncthreads();

  MWW_PC_AT(1);   // prologue done
  MWW_PC_START(MWW_ABLATE(a, 16) && tid == 0);
  for (int it = 0; it < nitems; ++it) {
#ifndef MWW_WIDE_NOPRIO
    rotate_priority(it, 2);
#endif
    const int b = blockIdx.x + (it / ntiles) * gridDim.x, t0 = (it % ntiles) * TT;
    const int nrows_new = max(0, min(TT, a.Tout - t0));  // du rows produced by this tile
    const int rows_da = min(TT, a.Tin - t0);             // input-gradient rows finalised by this tile
    // ---- P0: commit the activated input rows [t0, t0 + RA) (zero past the sample), the dp rows; roll the du ring
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int i = tid + j * NTH;
      if (i < RA * Q) {
        const int r = i / Q, q = i - r * Q;
        const float4 s4 = *reinterpret_cast<const float4*>(sAct + q * 4), h4 = *reinterpret_cast<const float4*>(sAct + C + q * 4);
        float4 v = pre_p[j];
        v.x = fmaxf(fmaf(v.x, s4.x, h4.x), 0.f);
        v.y = fmaxf(fmaf(v.y, s4.y, h4.y), 0.f);
        v.z = fmaxf(fmaf(v.z, s4.z, h4.z), 0.f);
        v.w = fmaxf(fmaf(v.w, s4.w, h4.w), 0.f);
        *reinterpret_cast<float4*>(sP + r * P + q * 4) = v;
      }
    }
    dps.commit(sDP, sKp, pre_dz, nrows_new * Q, tid);
    for (int i = tid; i < (K - 1) * P; i += NTH) sDU[i] = (t0 == 0) ? 0.f : sDU[TT * P + i];
    MWW_PC_MARK(0);   // commit (incl. the wait for the prefetched rows)
    __syncthreads();
    MWW_PC_MARK(1);   // barrier 1
    if (it + 1 < nitems) issue(it + 1);
    // ---- P1: recompute u = depthwise(a) + bias for the tile's output rows
    if (dw_active) {
      if (chunk * L < nrows_new) {
        float o[L];
#pragma unroll
        for (int t = 0; t < L; ++t) o[t] = dwb;
        dw_all_groups<K, L, false>(sP + chunk * L * P + c, P, sDW + c, C, o);
#pragma unroll
        for (int t = 0; t < L; ++t) {
          const int tl = chunk * L + t;
          sU[tl * P + c] = (tl < nrows_new) ? o[t] : 0.f;
        }
      } else {
#pragma unroll
        for (int t = 0; t < L; ++t) sU[(chunk * L + t) * P + c] = 0.f;
      }
    }
    MWW_PC_MARK(2);   // issue + P1 (u recompute)
    __syncthreads();
    MWW_PC_MARK(3);   // barrier 2
    // ---- P2/P3: dW_pw += u^T dp ; du = dp W^T -> ring rows [K-1, K-1+TT)
    // Long depthwise kernels (K = 21: 21 tap-gradient accumulators + a 27-value window per lane) left no register for the
    // loop-invariant fragment addresses of this phase: the compiler spilled two of them and reloaded them HERE, and a scratch
    // reload waits with vmcnt(0) - i.e. for the rows of the next tile that were requested a phase ago.  Deriving the fragment
    // coordinates from a lane id the compiler cannot hoist rebuilds the addresses per tile (a dozen integer instructions)
    // instead (MWW_WIDE_RELANE_K: kernels at least this long; 0 = never).
    int r16 = r16_, g = g_;
    if constexpr (MWW_WIDE_RELANE_K > 0 && K >= MWW_WIDE_RELANE_K) {
      int lm = lane;
      pin(lm);
      r16 = lm & 15;
      g = lm >> 4;
    }
    if constexpr (C == 48 && NW == 8) {
      if (wave < 6) {
        if constexpr (BF) {
          wide_dw_rows_bf16<3, 2, P>(sU, sDP, 32 * (wave / 3), nrows_new, 16 * (wave % 3), 0, r16, g, dwacc);
          wide_du_tiles_bf16<C, 1, K, P, PW>(sDP, sWt, sDU, 2 + wave / 3, nrows_new, 16 * (wave % 3), r16, g);
        } else {
          wide_dw_rows<3, 2, 16, D, P>(sU, sDP, 32 * (wave / 3), nrows_new, 16 * (wave % 3), 0, r16, g, dwacc);
          wide_du_tiles<C, 1, K, P, PW>(sDP, sWt, sDU, 2 + wave / 3, nrows_new, 16 * (wave % 3), r16, g);
        }
      } else {
        if constexpr (BF) wide_du_tiles_bf16<C, 3, K, P, PW>(sDP, sWt, sDU, wave - 6, nrows_new, 0, r16, g);
        else wide_du_tiles<C, 3, K, P, PW>(sDP, sWt, sDU, wave - 6, nrows_new, 0, r16, g);
      }
    } else if constexpr (BF) {
      wide_dw_rows_bf16<2, 4, P>(sU, sDP, 0, nrows_new, 16 * (wave / 2), 32 * (wave % 2), r16, g, dwacc);
      wide_du_tiles_bf16<C, 2, K, P, PW>(sDP, sWt, sDU, wave / 2, nrows_new, 32 * (wave % 2), r16, g);
    } else {
      wide_dw_rows<2, 4, 16, D, P>(sU, sDP, 0, nrows_new, 16 * (wave / 2), 32 * (wave % 2), r16, g, dwacc);
      wide_du_tiles<C, 2, K, P, PW>(sDP, sWt, sDU, wave / 2, nrows_new, 32 * (wave % 2), r16, g);
    }
    MWW_PC_MARK(4);   // MFMA (dW_pw, du)
    __syncthreads();
    MWW_PC_MARK(5);   // barrier 3
    // ---- P4: depthwise backward in two passes over the same registers.  No divergent branch around the global stores:
    // the lanes past the last chunk shadow it, their stores are out of range and their sums are dropped by the epilogue.
    {
      const int cch = dw_active ? chunk : NCH - 1;
      const BufRsrc gtile = tile_rsrc(elem_ptr<SB>(a.g_out, ((size_t)b * a.Tin + t0) * C), rows_da * C * elem_bytes(SB));
      const int goff = (dw_active ? 0 : kOobOffset / 4) + cch * L * C + c;
      const float* wdu_p = sDU + cch * L * P + c;
      const float* wa_p = sP + cch * L * P + c;
      {
        // pass A: da[sl] = sum_j w[K-1-j] * du_ring[sl + j] (in tap groups), ReLU mask, store g_{k-1}, its BN sums
        float da[L], am[L];
#pragma unroll
        for (int t = 0; t < L; ++t) da[t] = 0.f;
        if (cch * L < rows_da) dw_all_groups<K, L, true>(wdu_p, P, sDW + c, C, da);   // ring rows past the tile are allocated and zero
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < L; ++t) am[t] = wa_p[t * P];
#pragma unroll
        for (int t = 0; t < L; ++t) {
          const int sl = cch * L + t;
          // (rows past TT of the last chunk belong to the next tile: their da is still partial, rows_da <= TT drops them)
          const float gg = (sl < rows_da && am[t] > 0.f) ? da[t] : 0.f;
          tile_store1s<SB, MWW_AUX_ST_G>(gtile, goff + t * C, gg);
          gs1 += gg;
          gs2 = fmaf(gg, fmaf(am[t], xk1, xk0), gs2);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (cch * L < nrows_new) {
        // pass B: dW_dw[i] += sum_t du[t] * a[t + i] ; db += sum_t du[t]   (du[t] = ring row K-1 + chunk L + t)
        float duk[L];
#pragma unroll
        for (int t = 0; t < L; ++t) duk[t] = wdu_p[(K - 1 + t) * P];
#pragma unroll
        for (int t = 0; t < L; ++t) accb += duk[t];
        dw_wgrad_all_groups<K, L>(wa_p, P, duk, accw);
      }
    }
    MWW_PC_MARK(6);   // P4 (depthwise backward, stores)
    __syncthreads();
    MWW_PC_MARK(7);   // barrier 4
  }
  MWW_PC_AT(2);   // tile loop done

  // ---- epilogue: per-workgroup partial rows [K*C (dW_dw) | C (db) | C*C (dW_pw)] and the BN sums of g_{k-1}
  float* gdst = a.grad_part + (size_t)blockIdx.x * ((K + 1) * C + C * C);
  float* scratch = smem;
  if (Roles::has_dw(wave)) {
    float* sp = scratch + Roles::part(wave) * C * C + (Roles::mt(wave) * 16 + g_ * 4) * C + Roles::nt0(wave) * 16 + r16_;
#pragma unroll
    for (int nt = 0; nt < Roles::NTW; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sp[r * C + nt * 16] = dwacc[nt][r];
  }
  __syncthreads();
  for (int e = tid; e < C * C; e += NTH) {
    float v = scratch[e];
    if constexpr (Roles::NPART == 2) v += scratch[C * C + e];
    store_stream<MWW_AUX_ST_GP>(gdst + (K + 1) * C + e, v);
  }
  __syncthreads();
  if (dw_active) {
#pragma unroll
    for (int i = 0; i < K; ++i) scratch[(chunk * (K + 1) + i) * C + c] = accw[i];
    scratch[(chunk * (K + 1) + K) * C + c] = accb;
  }
  __syncthreads();
  for (int e = tid; e < (K + 1) * C; e += NTH) {
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) v += scratch[j * (K + 1) * C + e];
    store_stream<MWW_AUX_ST_GP>(gdst + e, v);
  }
  __syncthreads();
  if (dw_active) {
    scratch[(chunk * 2 + 0) * C + c] = gs1;
    scratch[(chunk * 2 + 1) * C + c] = gs2;
  }
  __syncthreads();
  if (tid < 2 * C) {
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) v += scratch[j * 2 * C + tid];
    publish_stat(a.gacc, a.gstat_part + (size_t)blockIdx.x * 2 * C, 2 * C, tid, v);
  }
  MWW_PC_AT(3);   // epilogue done
  MWW_PC_DUMP(a.phase_clk ? a.phase_clk + (size_t)blockIdx.x * kClkSlots : nullptr);
}

// ------------------------------------------------------------------------------------------
// Wide-workgroup form of the first block's backward kernel (bwd_first_kernel, kernels_bwd.hip.h: a0 = relu(conv1(x)) read
// back, no input gradient; instead dW1 += im2col(x)^T g0 on MFMA).  512 threads per 64-row tile of a0:
//   * VALU phases: C1 = 32 channels x 16 chunks of L = 4 rows;
//   * pointwise phase: waves 0-3 own the du tiles of row tile w (both 16-channel halves), waves 4-7 the dW_pw tiles
//     (mt = v % 2, nt = 0..NT-1) over the row half v / 2, v = w - 4;
//   * dW1 phase: m-tile mi * 8 + w of W1 (both 16-column halves of C1) per wave, mi < ceil(MT1 / 8), rows met in the
//     order s = kk + 16 g: the x rows of a k-step are 16 S rows apart (16 S * 41 = 16 mod 32 for S = 1, 3: conflict-free
//     with the odd x pitch) and the g0 tile has an odd pitch of its own for the same reason.
template <int K1, int C1, int COUT, int K, int S, int NTH, bool X6 = false>
struct BwdFirstWideLds {
  static constexpr int PI = C1 + 4;                        // a0 / du ring / u: row-pattern MFMA reads 4 rows apart (4 * 36 = 16 mod 32)
  static constexpr int PO = COUT + 4;                      // dp: the same rows (4 * 52 = 4 * 68 = 16 mod 32), float4 column reads
  static constexpr int PG = C1 + 5;                        // g0: rows 16 apart need an odd pitch
  static constexpr int PW = C1 + 4;                        // W_pw^T [COUT][C1]: rows 4 apart
  static constexpr int PX = FBINS + 1;
  static constexpr int NCH = NTH / C1, L = (TT + NCH - 1) / NCH, TTP = NCH * L, RAP = TTP + K - 1;
  static constexpr int TAIL = S > 1 ? K - 1 : 0;
  static constexpr int TAILK = (TAIL + 3) / 4 * 4;   // the tail k-steps of dW1 read x / g0 rows [TT, TT + TAILK): whole k-steps of four
  static constexpr int XR = (TT + TAILK - 1) * S + K1;
  static constexpr int up4(int v) { return (v + 3) / 4 * 4; }
  static constexpr int OFF_A = 0, OFF_DP = OFF_A + up4(RAP * PI), OFF_U = OFF_DP + TT * PO, OFF_DU = OFF_U + up4(TTP * PI);
  // X6 (common.hip.h; bwd_first_body.inc): x as three bf16 slice planes of 96-byte rows, g0 as three planes of 64-byte rows in the
  // dp (+ u) tile's space - no g0 tile of its own
  static constexpr int PB = 96, PLB = XR * PB, GPB = 2 * C1, GPL = TT * GPB;
  static constexpr int XFLOATS = X6 ? 3 * PLB / 4 : up4(XR * PX);
  static constexpr int OFF_G0 = OFF_DU + up4(RAP * PI), OFF_END = X6 ? OFF_G0 : OFF_G0 + up4((TTP + TAILK) * PG);
  static constexpr int BYTES = (OFF_END + XFLOATS + 7 * COUT + COUT * PW) * 4 + (int)sizeof(XShared);
};

template <int K1, int C1, int COUT, int K, int S, int NTH, bool X6 = false>
constexpr int wide_first_waves_per_simd() {
  return (BwdFirstWideLds<K1, C1, COUT, K, S, NTH, X6>::BYTES <= 80 * 1024 ? 2 : 1) * (NTH / 64) / 4;
}

template <int K1, int C1, int COUT, int K, int S, int NTH, bool X6 = false>
__global__ __launch_bounds__(NTH, (wide_first_waves_per_simd<K1, C1, COUT, K, S, NTH, X6>())) void bwd_firstw_kernel(BwdFirstArgs a) {
  typedef BwdFirstWideLds<K1, C1, COUT, K, S, NTH, X6> Lds;
  constexpr bool SB = false;
  constexpr int CIN = C1;
  constexpr int PI = Lds::PI, PO = Lds::PO, PG = Lds::PG, PW = Lds::PW, PX = Lds::PX;
  constexpr int NW = NTH / 64, NCH = Lds::NCH, L = Lds::L, TTP = Lds::TTP, RA = TT + K - 1, RAP = Lds::RAP;
  constexpr int TAIL = Lds::TAIL, XR = Lds::XR;
  constexpr int NT1 = C1 / 16, M1 = K1 * FBINS, MT1 = (M1 + 15) / 16, MPW = (MT1 + NW - 1) / NW;
  constexpr int MT = CIN / 16, NT = COUT / 16, QI = CIN / 4, QO = COUT / 4;
  static_assert(NW == 8 && MT == 2 && NCH * L == TT && TT >= K - 1, "geometry of the wide first-block kernel");
  static_assert(TAIL <= NCH && TTP + TAIL <= RAP, "tail rows");
  static_assert(Lds::OFF_END >= 2 * CIN * COUT && Lds::OFF_END >= NCH * (K + 1) * CIN, "scratch aliasing");
  constexpr int PB = Lds::PB, PLB = Lds::PLB, GPB = Lds::GPB, GPL = Lds::GPL;
  static_assert(!X6 || (S == 1 && CIN == 32 && (K1 * FBINS) % 8 == 0 && 3 * GPL <= (TT * PO + TTP * PI) * 4 && TTP == TT), "the x6 form serves the stride-1 first convolutions");

  __shared__ __attribute__((aligned(16))) float sX[Lds::XFLOATS];
  __shared__ XShared sXg;
  __shared__ __attribute__((aligned(16))) float smem[Lds::OFF_END];
  __shared__ __attribute__((aligned(16))) float sKp[7 * COUT];
  __shared__ __attribute__((aligned(16))) float sWt[COUT * PW];   // W_pw^T
  float* sA = smem + Lds::OFF_A;
  float* sDP = smem + Lds::OFF_DP;
  float* sU = smem + Lds::OFF_U;
  float* sDU = smem + Lds::OFF_DU;
  float* sG0 = smem + Lds::OFF_G0;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, g = lane >> 4;
  const int c = tid % CIN, chunk = tid / CIN;
  const int Ta = (a.T - K1) / S + 1;

  const bool tailmode = TAIL > 0 && Ta > TT && Ta <= TT + TAIL;   // the window is one tile: rows [TT, Ta) ride behind the TT tile rows
  const int ntiles = tailmode ? 1 : (Ta + TT - 1) / TT;
  const int nsamp = (int)blockIdx.x < a.B ? (a.B - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const int nitems = nsamp * ntiles;
  XStage<XR, PX, MWW_AUX_LD_XB, NTH> xs;
  DpStageW<COUT, false, SB, NTH, PO> dps;
  constexpr int NA = (RA * QI + NTH - 1) / NTH;
  float4 pre_a[NA];
  auto issue = [&](int it) {
    const int s = it / ntiles, b = blockIdx.x + s * gridDim.x, t0 = (it % ntiles) * TT;
    const int nrx = ((tailmode ? Ta : min(TT, Ta - t0)) - 1) * S + K1;
    xs.issue(a.x, a.xg, sXg, s, b, a.T, t0 * S, nrx, tid);
    const BufRsrc ra = tile_rsrc(a.a0 + ((size_t)b * Ta + t0) * CIN, min(RA, Ta - t0) * QI * 16);
#pragma unroll
    for (int j = 0; j < NA; ++j) pre_a[j] = tile_load4<MWW_AUX_LD_A0>(ra, (tid + j * NTH) * 16);
    const int nvk = max(0, min(TT, a.Tout - t0)) * QO;
    const size_t koff = ((size_t)b * a.Tout + t0) * COUT;
    dps.issue(elem_ptr<SB>(a.pk, koff), elem_ptr<SB>(a.gk, koff), nvk, tid);
  };
  // per-lane offsets of the dW1 rows this wave owns: row m = j*40+f of W1 reads x[s*S+j][f]
  int offm[MPW];
  bool okm[MPW];
  int xoff[MPW], goff[NT1];   // X6: this lane's share of the transpose reads (bwd_first_body.inc)
  if constexpr (X6) {
    const int rowl = 4 * g + (r16 >> 2), cg = 4 * (r16 & 3);
#pragma unroll
    for (int mi = 0; mi < MPW; ++mi) {
      const int m = min((mi * NW + wave) * 16 + cg, M1 - 4);
      xoff[mi] = (rowl + m / FBINS) * PB + (m % FBINS) * 2;
    }
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt) goff[nt] = rowl * GPB + ((nt * 16 + cg) ^ ((rowl & 4) ? 16 : 0)) * 2;
  } else {
#pragma unroll
  for (int mi = 0; mi < MPW; ++mi) {
    const int m = (mi * NW + wave) * 16 + r16;
    okm[mi] = m < M1;
    offm[mi] = okm[mi] ? (m / FBINS) * PX + (m % FBINS) : 0;
  }
  }
  if (a.xg.win) xgather_setup(a.xg, sXg, nsamp, tid);
  stagger_start<MWW_STAGGER_BWD>();
  if (nitems > 0) issue(0);

  // every global load of the prologue first, then the LDS copies
  constexpr int NWL = (CIN * COUT + NTH - 1) / NTH;
  float wl[NWL];
#pragma unroll
  for (int j = 0; j < NWL; ++j) wl[j] = (tid + j * NTH < CIN * COUT) ? a.pw_w[tid + j * NTH] : 0.f;
  float accw[K];
  float accb = 0.f;
#pragma unroll
  for (int i = 0; i < K; ++i) accw[i] = 0.f;
  float dwb = a.dw_b[c];
  // the depthwise taps stay in LDS (read per phase): scratch rows behind the g0 tile are not needed, sKp rows 3-4 are free
  // (2 * COUT >= K * CIN / ... does not hold in general: the taps get their own rows of the x-stage array's padding) - kept
  // simple: K * CIN floats at the end of sKp's unused rows 3..4 when they fit, else registers
  float dww[K];
#pragma unroll
  for (int i = 0; i < K; ++i) dww[i] = a.dw_w[i * CIN + c];
  for (int i = tid; i < COUT; i += NTH) {
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
#pragma unroll
  for (int j = 0; j < NWL; ++j) {
    const int i = tid + j * NTH;
    if (i < CIN * COUT) sWt[(i % COUT) * PW + i / COUT] = wl[j];   // W[ci][co] -> W^T[co][ci]
  }
  for (int i = RA * PI + tid; i < RAP * PI; i += NTH) {
    sA[i] = 0.f;
    sDU[i] = 0.f;
  }
  if constexpr (!X6)
    for (int i = TT * PG + tid; i < (TTP + Lds::TAILK) * PG; i += NTH) sG0[i] = 0.f;   // rows the tail k-step may read
#pragma unroll
  for (int i = 0; i < K; ++i) pin(dww[i]);
  pin(dwb);
  f32x4 dwacc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) dwacc[nt] = zero4();
  f32x4 w1acc[MPW][NT1];
#pragma unroll
  for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt) w1acc[mi][nt] = zero4();
  __syncthreads();

  for (int it = 0; it < nitems; ++it) {
#ifndef MWW_WIDE_NOPRIO
    rotate_priority(it, 2);
#endif
    const int t0 = (it % ntiles) * TT;
    const int nrows_new = max(0, min(TT, a.Tout - t0));
    const int rows_da = min(TT, Ta - t0);
    // ---- P0: commit x (odd pitch), a0 rows [t0, t0+RA) (zero past the sample), dp; roll the du ring
    if constexpr (X6) xs.template commit_planes<PB, PLB>(reinterpret_cast<unsigned*>(sX), a.xg, sXg, it / ntiles, t0 * S, tid);
    else xs.commit(sX, a.xg, sXg, it / ntiles, t0 * S, tid);
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int i = tid + j * NTH;
      if (i < RA * QI) {
        const int r = i / QI, q = i - r * QI;
        *reinterpret_cast<float4*>(sA + r * PI + q * 4) = pre_a[j];
      }
    }
    dps.commit(sDP, sKp, 0.f, nrows_new * QO, tid);
    for (int i = tid; i < (K - 1) * PI; i += NTH) sDU[i] = (t0 == 0) ? 0.f : sDU[TT * PI + i];
    __syncthreads();
    if (it + 1 < nitems) issue(it + 1);
    // ---- P1: u = depthwise(a0) + bias
    if (chunk * L < nrows_new) {
      float o[L];
      dw_chunk<K, L>(sA, PI, chunk * L, c, dww, dwb, o);
#pragma unroll
      for (int t = 0; t < L; ++t) {
        const int tl = chunk * L + t;
        sU[tl * PI + c] = (tl < nrows_new) ? o[t] : 0.f;
      }
    } else {
#pragma unroll
      for (int t = 0; t < L; ++t) sU[(chunk * L + t) * PI + c] = 0.f;
    }
    __syncthreads();
    // ---- P2/P3: waves 0-3: du tiles of row tile w; waves 4-7: dW_pw tiles (mt = v % 2, all nt) over the row half v / 2
    if (wave < 4) {
      wide_du_tiles<COUT, MT, K, PO, PW, PI>(sDP, sWt, sDU, wave, nrows_new, 0, r16, g);
    } else {
      const int v = wave - 4;
      wide_dw_rows<NT, 2, 16, 4, PI, PO>(sU, sDP, 32 * (v / 2), nrows_new, 16 * (v % 2), 0, r16, g, dwacc);
    }
    __syncthreads();
    // ---- P4: depthwise backward -> g0 = da * relu'(a0) kept in LDS; dW_dw, db
    {
      float wdu[L + K - 1], wa[L + K - 1];
#pragma unroll
      for (int j = 0; j < L + K - 1; ++j) wdu[j] = sDU[(chunk * L + j) * PI + c];
#pragma unroll
      for (int j = 0; j < L + K - 1; ++j) wa[j] = sA[(chunk * L + j) * PI + c];
      lds_reads_first();
#pragma unroll
      for (int t = 0; t < L; ++t) {
        const int sl = chunk * L + t;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < K; ++i) acc = fmaf(dww[K - 1 - i], wdu[t + i], acc);
        const float gv = (sl < rows_da && wa[t] > 0.f) ? acc : 0.f;
        if constexpr (X6) {
          unsigned h0, h1, h2;
          split3(gv, h0, h1, h2);
          unsigned short* gp = reinterpret_cast<unsigned short*>(sDP) + sl * CIN + (c ^ ((sl & 4) ? 16 : 0));
          gp[0] = (unsigned short)(h0 >> 16);
          gp[GPL / 2] = (unsigned short)(h1 >> 16);
          gp[GPL] = (unsigned short)(h2 >> 16);
        } else {
          sG0[sl * PG + c] = gv;
        }
      }
      if (chunk * L < nrows_new) {
#pragma unroll
        for (int t = 0; t < L; ++t) {
          const float dut = wdu[K - 1 + t];
          accb += dut;
#pragma unroll
          for (int i = 0; i < K; ++i) accw[i] = fmaf(dut, wa[t + i], accw[i]);
        }
      }
      if constexpr (TAIL > 0) {
        if (tailmode && chunk < TAIL) {
          // input-gradient rows [TT, Ta) of a single-tile window: one row per chunk (du rows past the tile are zero)
          const int sl = TT + chunk;
          float acc = 0.f;
#pragma unroll
          for (int j = 0; j < K; ++j) acc = fmaf(dww[K - 1 - j], sl + j < RAP ? sDU[(sl + j) * PI + c] : 0.f, acc);
          sG0[sl * PG + c] = (sl < Ta && sA[sl * PI + c] > 0.f) ? acc : 0.f;
        }
      }
    }
    __syncthreads();
    // ---- dW1 += im2col(x)^T g0 : A[m][k=s] = x[s*S + m/40][m%40], B[k=s][n] = g0[s][n]; k-step kk holds rows kk + 16 g
    if constexpr (X6) {
      // six bf16 slice products per fp32 product, operands through ds_read_b64_tr_b16 (bwd_first_body.inc)
      const char* xb = reinterpret_cast<const char*>(sX);
      const char* gb = reinterpret_cast<const char*>(sDP);
#pragma unroll
      for (int kb = 0; kb < TT / 32; ++kb)
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) {
          u32x4v gq[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            const u32x2v lo = lds_read_tr16(gb + goff[nt] + p * GPL + (32 * kb) * GPB), hi = lds_read_tr16(gb + goff[nt] + p * GPL + (32 * kb + 16) * GPB);
            gq[p] = u32x4v{lo.x, lo.y, hi.x, hi.y};
          }
#pragma unroll
          for (int mi = 0; mi < MPW; ++mi)
            if (mi * NW + wave < MT1) {   // wave-uniform
              u32x4v xq[3];
#pragma unroll
              for (int p = 0; p < 3; ++p) {
                const u32x2v lo = lds_read_tr16(xb + xoff[mi] + p * PLB + (32 * kb) * S * PB), hi = lds_read_tr16(xb + xoff[mi] + p * PLB + (32 * kb + 16) * S * PB);
                xq[p] = u32x4v{lo.x, lo.y, hi.x, hi.y};
              }
              w1acc[mi][nt] = mfma_x6(xq, gq, w1acc[mi][nt]);
            }
        }
    } else {
      float av[2][MPW], bv[2][NT1];
      auto load_w1 = [&](int srow, int sl) {   // srow = this lane's row of the k-step
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) bv[sl][nt] = sG0[srow * PG + nt * 16 + r16];
#pragma unroll
        for (int mi = 0; mi < MPW; ++mi) {
          const float v = sX[srow * S * PX + offm[mi]];   // (rows past M1 read offset 0 and are zeroed: no conditional load)
          av[sl][mi] = okm[mi] ? v : 0.f;
        }
      };
      load_w1(16 * g, 0);
#pragma unroll
      for (int kk = 0; kk < TT / 4; ++kk) {
        if (kk + 1 < TT / 4) load_w1(kk + 1 + 16 * g, (kk + 1) & 1);
#pragma unroll
        for (int mi = 0; mi < MPW; ++mi)
          if (mi * NW + wave < MT1) {   // wave-uniform
#pragma unroll
            for (int nt = 0; nt < NT1; ++nt) w1acc[mi][nt] = mfma4(av[kk & 1][mi], bv[kk & 1][nt], w1acc[mi][nt]);
          }
      }
      if constexpr (TAIL > 0) {
        if (tailmode) {   // the tail rows [TT, Ta): one or two more k-steps of four rows (g0 rows past Ta are zero)
#pragma unroll
          for (int ks = 0; ks < Lds::TAILK / 4; ++ks) {
            load_w1(TT + 4 * ks + g, 0);
#pragma unroll
            for (int mi = 0; mi < MPW; ++mi)
              if (mi * NW + wave < MT1) {
#pragma unroll
                for (int nt = 0; nt < NT1; ++nt) w1acc[mi][nt] = mfma4(av[0][mi], bv[0][nt], w1acc[mi][nt]);
              }
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue: partial row [M1*C1 (dW1) | K*CIN (dW_dw) | CIN (db) | CIN*COUT (dW_pw)]
  float* gdst = a.grad_part + (size_t)blockIdx.x * (M1 * C1 + (K + 1) * CIN + CIN * COUT);
#pragma unroll
  for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = (mi * NW + wave) * 16 + g * 4 + r;
        if (m < M1) store_stream<MWW_AUX_ST_GP>(gdst + m * C1 + nt * 16 + r16, w1acc[mi][nt][r]);
      }
  float* gblk = gdst + M1 * C1;
  float* scratch = smem;
  if (wave >= 4) {
    const int v = wave - 4;
    float* sp = scratch + (v / 2) * CIN * COUT + ((v % 2) * 16 + g * 4) * COUT + r16;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sp[r * COUT + nt * 16] = dwacc[nt][r];
  }
  __syncthreads();
  for (int e = tid; e < CIN * COUT; e += NTH) store_stream<MWW_AUX_ST_GP>(gblk + (K + 1) * CIN + e, scratch[e] + scratch[CIN * COUT + e]);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < K; ++i) scratch[(chunk * (K + 1) + i) * CIN + c] = accw[i];
  scratch[(chunk * (K + 1) + K) * CIN + c] = accb;
  __syncthreads();
  for (int e = tid; e < (K + 1) * CIN; e += NTH) {
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) v += scratch[j * (K + 1) * CIN + e];
    store_stream<MWW_AUX_ST_GP>(gblk + e, v);
  }
}

}  // namespace mww

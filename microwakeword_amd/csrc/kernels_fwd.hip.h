// Forward kernels of the MixedNet train step (reference graph: microwakeword/mixednet.py:307-386).
//
//   fwd_first_kernel : x (dense, or gathered from the feature stores: XGather / XStage below) -> Conv2D(K1x1, valid,
//                      no bias) -> ReLU -> DepthwiseConv(Kx1)+bias -> 1x1 Conv -> p_1 (pre-BN) + (sum, sum^2) of p_1
//                      (mixednet.py:317-331 then :209-211 and :349-351; no BN in between, so one kernel)
//   fwd_block_kernel : p_{k-1} -> [BN_{k-1} + ReLU on load] -> Depthwise(Kx1)+bias -> 1x1 -> p_k + its sums
//                      (mixednet.py:352,360 of block k-1, then :209-211,:349-351 of block k)
//   The sums leave a kernel either as fp64 atomic adds to replicated accumulator rows, folded by the next kernel's
//   prologue (common.hip.h "BN statistics hand-over"; the default), or as per-workgroup partial rows for
//   bn_fwd_finalize_kernel : partials -> batch mean / biased variance -> folded scale/shift,
//                      saved mean/rstd for backward, Keras moving-average update
//   bn_eval_prepare_kernel : moving stats -> folded scale/shift (inference)
//   (classifier head and loss: kernels_head.hip.h)
#pragma once
#include "common.hip.h"
#include "../../include/mww.h"

namespace mww {

// ---- first-block input straight from the feature stores ("fused_input") ---------------------------------
// With XGather::win set, fwd_first_kernel / bwd_first_kernel do what assemble_kernel does (reference
// microwakeword/data.py:74-118 pad / truncate, :268-269 uint16 scaling, :32-71 SpecAugment zeroing) while
// staging their x rows: the window descriptors and masks of the workgroup's samples are fetched once in
// the prologue (descriptors to LDS, masks as row / column bitmaps), each staged float4 group is gathered
// from its store (uint16: 8 bytes instead of 16) and converted + masked on its way into LDS.  The x buffer
// is then neither written nor read: 93 KB less HBM traffic per window and one launch less per step.
constexpr int kXMaxSamples = 8;   // windows per workgroup the gather mode keeps descriptors for
constexpr int kXRowWords = 8;     // row bitmap: up to 256 frames
constexpr int kXMaxMasks = 8;     // SpecAugment masks per window in gather mode (more: the batch is materialised)

struct XGather {
  const mww_window* win;   // [B] (null: dense float32 x)
  const int* masks;        // [B][ntm + nfm][2]
  const void* store[MWW_MAX_STORES];
  int dtype[MWW_MAX_STORES];
  int ntm, nfm, T;
};

struct XShared {
  const void* base[kXMaxSamples];
  long long src_elem[kXMaxSamples];
  int pad_rows[kXMaxSamples], copy_rows[kXMaxSamples], dtype[kXMaxSamples];
  unsigned rowbits[kXMaxSamples][kXRowWords];
  unsigned colbits[kXMaxSamples][2];
};

__device__ __forceinline__ unsigned bits_of_range(int start, int width, int word) {
  const int lo = max(start, 32 * word) - 32 * word, hi = min(start + width, 32 * word + 32) - 32 * word;
  if (hi <= lo) return 0u;
  const unsigned run = (hi - lo >= 32) ? 0xffffffffu : ((1u << (hi - lo)) - 1u);
  return run << lo;
}

// prologue (all threads; ends with a barrier): descriptors, store bases and mask bitmaps of this workgroup's samples
__device__ __forceinline__ void xgather_setup(const XGather& g, XShared& sh, int nsamp, int tid) {
  if (tid < nsamp) {
    const mww_window w = g.win[blockIdx.x + tid * gridDim.x];
    const void* base = g.store[0];
    int dt = g.dtype[0];
#pragma unroll
    for (int i = 1; i < MWW_MAX_STORES; ++i)
      if (w.store == i) {
        base = g.store[i];
        dt = g.dtype[i];
      }
    sh.base[tid] = base;
    sh.dtype[tid] = dt;
    sh.src_elem[tid] = w.src_elem;
    sh.pad_rows[tid] = w.pad_rows;
    sh.copy_rows[tid] = w.copy_rows;
  }
  constexpr int WPS = kXRowWords + 2;   // bitmap words per sample
  if (tid < nsamp * WPS) {
    const int s = tid / WPS, word = tid - s * WPS;
    const int nm = g.ntm + g.nfm;
    const int* mk = g.masks + (size_t)(blockIdx.x + s * gridDim.x) * nm * 2;
    // all (start, width) pairs are fetched before the first one is used: one memory round trip
    int mv[2 * kXMaxMasks];
#pragma unroll
    for (int m = 0; m < 2 * kXMaxMasks; ++m) mv[m] = (m < 2 * nm) ? mk[m] : 0;
    const bool row = word < kXRowWords;
    const int w = row ? word : word - kXRowWords;
    unsigned bits = 0u;
#pragma unroll
    for (int m = 0; m < kXMaxMasks; ++m)
      if (m < nm && (m < g.ntm) == row) bits |= bits_of_range(mv[2 * m], mv[2 * m + 1], w);
    if (row) sh.rowbits[s][w] = bits;
    else sh.colbits[s][w] = bits;
  }
  __syncthreads();
}

// register stage of the x rows of one (sample, tile) item.  Thread -> (row group rq = tid / 10, float4 column
// q = tid % 10) for tid < 250 (NTH = 256 threads; 510 of 512); pass j handles row rq + 25 j (51 j): every per-pass address is a compile-time offset
// from a per-thread base, in HBM (float4 index tid + 250 j of the tile) and in LDS (odd row pitch PX).
// issue() fetches rows [row0, row0 + nrows) of the sample (dense) or gathers them (store rows row0 - pad_rows ...);
// commit() writes them to LDS, converting / masking in gather mode.
template <int XROWS, int PX, int AUX = 0, int NTH = kThreads>
struct XStage {
  static constexpr int QX = FBINS / 4, RPP = NTH / QX, ACT = RPP * QX, NJ = (XROWS + RPP - 1) / RPP;
  float4 pre[NJ];

  // `tid` is laundered in both methods: the per-lane row / offset arithmetic is a handful of VALU ops per tile,
  // hoisted out of the tile loop it would cost long-lived registers in kernels that have none to spare.
  // One straight-line path for dense x and for both store dtypes (no conditional loads, see common.hip.h): the rows
  // [r_lo, r_hi) of the tile that exist in the source are one contiguous slice; it is read through two descriptors in
  // 8-byte halves of a float4 group (uint16 stores: a group IS 8 bytes and the second descriptor is empty), lanes in
  // front of the slice wrap to huge unsigned offsets and, like those behind it, get zeros.
  __device__ __forceinline__ void issue(const float* x, const XGather& g, const XShared& sh, int s, int b, int T, int row0,
                                        int nrows, int tid) {
    asm volatile("" : "+v"(tid));
    const bool gather = g.win != nullptr;
    // the descriptor is workgroup-uniform: keep it in scalar registers
    const int pad = gather ? uniform_int(sh.pad_rows[s]) : 0, copy = gather ? uniform_int(sh.copy_rows[s]) : T;
    const bool u16 = gather && uniform_int(sh.dtype[s]) == MWW_DTYPE_U16;
    const int gb = u16 ? 8 : 16;                                         // bytes of one float4 group in the source
    const int r_lo = max(pad - row0, 0), r_hi = min(pad + copy - row0, nrows);
    const char* base = gather ? reinterpret_cast<const char*>(uniform_ptr(sh.base[s])) + uniform_i64(sh.src_elem[s]) * (u16 ? 2 : 4)
                              : reinterpret_cast<const char*>(x + (size_t)b * T * FBINS);
    base += (long long)(row0 - pad + r_lo) * (QX * gb);
    const int nbytes = (r_hi - r_lo) * (QX * gb);
    const BufRsrc lo = tile_rsrc(base, nbytes), hi = tile_rsrc(base + 8, u16 ? 0 : nbytes - 8);
    const int off0 = tid < ACT ? (tid - r_lo * QX) * gb : kOobOffset;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const uint2 a = tile_load2<AUX>(lo, off0 + ACT * j * gb), c = tile_load2<AUX>(hi, off0 + ACT * j * gb);
      pre[j] = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(c.x), __uint_as_float(c.y));
    }
  }

  __device__ __forceinline__ void commit(float* sX, const XGather& g, const XShared& sh, int s, int row0, int tid) const {
    asm volatile("" : "+v"(tid));
    const int rq = tid / QX, q = tid - rq * QX;
    if (tid >= ACT) return;
    const bool gather = g.win != nullptr;
    const bool u16 = gather && uniform_int(sh.dtype[s]) == MWW_DTYPE_U16;
    // mask words of this thread's rows are fetched together (one LDS round trip), then applied without branches
    unsigned cm = 0u, rb[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) rb[j] = 0u;
    if (gather) {
      cm = (sh.colbits[s][(4 * q) >> 5] >> ((4 * q) & 31)) & 0xfu;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int t = min(row0 + rq + RPP * j, g.T - 1);   // rows past the window are zero already
        rb[j] = sh.rowbits[s][t >> 5] >> (t & 31);
      }
    }
    float* dst = sX + rq * PX + q * 4;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int r = rq + RPP * j;
      if (r < XROWS) {
        float4 v = pre[j];
        if (u16) {
          const unsigned lo = __float_as_uint(v.x), hi = __float_as_uint(v.y);
          v.x = (float)(lo & 0xffffu) * 0.0390625f;   // data.py:268-269
          v.y = (float)(lo >> 16) * 0.0390625f;
          v.z = (float)(hi & 0xffffu) * 0.0390625f;
          v.w = (float)(hi >> 16) * 0.0390625f;
        }
        if (gather) {
          const unsigned m4 = (rb[j] & 1u) ? 0xfu : cm;
          v.x = (m4 & 1u) ? 0.f : v.x;
          v.y = (m4 & 2u) ? 0.f : v.y;
          v.z = (m4 & 4u) ? 0.f : v.z;
          v.w = (m4 & 8u) ? 0.f : v.w;
        }
        float* d = dst + RPP * j * PX;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    }
  }

  // The "x6" kernels (common.hip.h "fp32-grade products on the bf16 matrix pipe") stage the rows as three bf16 slice planes
  // instead: plane p holds slice p of every value, rows of PB bytes (FBINS bf16 + padding), planes PLB bytes apart; the same
  // pad / truncate / scale / mask decisions as commit(), then and / sub / and / sub per value and one 8-byte store per plane.
  template <int PB, int PLB>
  __device__ __forceinline__ void commit_planes(unsigned* sXb, const XGather& g, const XShared& sh, int s, int row0, int tid) const {
    asm volatile("" : "+v"(tid));
    const int rq = tid / QX, q = tid - rq * QX;
    if (tid >= ACT) return;
    const bool gather = g.win != nullptr;
    const bool u16 = gather && uniform_int(sh.dtype[s]) == MWW_DTYPE_U16;
    unsigned cm = 0u, rb[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) rb[j] = 0u;
    if (gather) {
      cm = (sh.colbits[s][(4 * q) >> 5] >> ((4 * q) & 31)) & 0xfu;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int t = min(row0 + rq + RPP * j, g.T - 1);   // rows past the window are zero already
        rb[j] = sh.rowbits[s][t >> 5] >> (t & 31);
      }
    }
    char* dst = reinterpret_cast<char*>(sXb) + rq * PB + q * 8;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int r = rq + RPP * j;
      if (r < XROWS) {
        float4 v = pre[j];
        if (u16) {
          const unsigned lo = __float_as_uint(v.x), hi = __float_as_uint(v.y);
          v.x = (float)(lo & 0xffffu) * 0.0390625f;   // data.py:268-269
          v.y = (float)(lo >> 16) * 0.0390625f;
          v.z = (float)(hi & 0xffffu) * 0.0390625f;
          v.w = (float)(hi >> 16) * 0.0390625f;
        }
        if (gather) {
          const unsigned m4 = (rb[j] & 1u) ? 0xfu : cm;
          v.x = (m4 & 1u) ? 0.f : v.x;
          v.y = (m4 & 2u) ? 0.f : v.y;
          v.z = (m4 & 4u) ? 0.f : v.z;
          v.w = (m4 & 8u) ? 0.f : v.w;
        }
        unsigned h[3][4];
        split3(v.x, h[0][0], h[1][0], h[2][0]);
        split3(v.y, h[0][1], h[1][1], h[2][1]);
        split3(v.z, h[0][2], h[1][2], h[2][2]);
        split3(v.w, h[0][3], h[1][3], h[2][3]);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          u32x2v w2 = {pack_hi2(h[p][0], h[p][1]), pack_hi2(h[p][2], h[p][3])};
          *reinterpret_cast<u32x2v*>(dst + p * PLB + RPP * j * PB) = w2;
        }
      }
    }
  }
};

// k-step map of the first convolution's im2col GEMM in the exact-fp32 form (fwd_first_body.inc): which (x row, first bin) k-step
// kk meets, and whether it is a "wide" one (lane group g then adds (g >> 1) + 16 (g & 1) instead of g)
#ifndef MWW_CONV1_KMAP   // tuning builds: 0 = the plain order k = 4 kk + g at every stride
#define MWW_CONV1_KMAP 1
#endif
template <int S>
struct Conv1KMap {
  static constexpr bool permuted = MWW_CONV1_KMAP && (S % 2) == 1;
  static constexpr int row(int kk) { return kk / (FBINS / 4); }
  static constexpr int sub(int kk) { return kk % (FBINS / 4); }
  static constexpr bool wide(int kk) { return permuted && sub(kk) < 8; }
  static constexpr int col(int kk) { return !permuted ? sub(kk) * 4 : (sub(kk) < 8 ? 2 * sub(kk) : 32 + 4 * (sub(kk) - 8)); }
};

struct FwdFirstArgs {
  const float* x;        // [B][T][40]
  const float* w1;       // [K1*40][C1]   (Keras [K1,1,40,C1] flattened)
  const float* dw_w;     // [K][C1]
  const float* dw_b;     // [C1]
  const float* pw_w;     // [C1][COUT]
  float* out;            // p_1 [B][Tout][COUT]
  float* stat_part;      // [gridDim.x][2][COUT]
  int B, T, Tout;        // Tout = (T - K1)/S + 1 - (K-1)
  int ablate;            // profiling only: bit0 skip depthwise, bit1 skip MFMA, bit2 skip stores (results invalid)
  StatAcc sacc;          // statistics go to the accumulator rows instead of stat_part when set
  XGather xg;            // xg.win set: x rows are gathered from the feature stores (x is not read)
  float* a0;             // relu(conv1(x)) [B][Ta][C1] kept for bwd_first_kernel (null: not stored, e.g. inference)
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
  int ablate;            // profiling only (see FwdFirstArgs); bit 16: per-phase clocks of thread 0 -> phase_clk
  unsigned long long* phase_clk;   // [gridDim.x][8]
  StatAcc sacc;          // statistics go to the accumulator rows instead of stat_part when set
  BnFoldArgs fold;       // fold.acc set: in_scale / in_shift are computed here from the producer's accumulator rows
};

// depthwise conv over one (channel, chunk): out[t] = bias + sum_i w[i]*src[t+i], t in [0,L)
// src rows are read from LDS once into a register window (fully unrolled, static indices).
//   REV : use the taps reversed (w[K-1-i]) — the transposed (input-gradient) depthwise
//   ACT : apply y = relu(src*sc + sh) while loading (LDS holds the raw pre-BN tensor)
template <int K, int L, bool REV = false, bool ACT = false>
__device__ __forceinline__ void dw_chunk(const float* src, int src_pitch, int row0, int c, const float (&w)[K],
                                         float bias, float (&out)[L], float sc = 1.f, float sh = 0.f) {
  float win[L + K - 1];
#pragma unroll
  for (int j = 0; j < L + K - 1; ++j) {
    float v = src[(row0 + j) * src_pitch + c];   // rows past the valid ones are allocated and zero
    if (ACT) v = fmaxf(fmaf(v, sc, sh), 0.f);
    win[j] = v;
  }
  lds_reads_first();
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

// register-resident 1x1 weights of a workgroup + the row-tile contraction, in exact fp32 (BF = false) or
// with bf16 operands (BF = true)
template <int CIN, int NT, bool BF>
struct PwWeights;

template <int CIN, int NT>
struct PwWeights<CIN, NT, false> {
  float f[CIN / 4][NT];
  __device__ __forceinline__ void load(const float* w, int cout, int g, int r16) {
#pragma unroll
    for (int kk = 0; kk < CIN / 4; ++kk)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) f[kk][nt] = w[(kk * 4 + g) * cout + nt * 16 + r16];
  }
  __device__ __forceinline__ void retire() {
#pragma unroll
    for (int kk = 0; kk < CIN / 4; ++kk)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) pin(f[kk][nt]);
  }
  __device__ __forceinline__ void tile(const float* sU, int cpi, int row0, int r16, int g, f32x4 (&acc)[NT]) const {
    pw_rowtile<CIN / 4, NT>(sU, cpi, row0, r16, g, f, acc);
  }
};

template <int CIN, int NT>
struct PwWeights<CIN, NT, true> {
  bf16x4 f[CIN / 16][NT];
  __device__ __forceinline__ void load(const float* w, int cout, int g, int r16) {
#pragma unroll
    for (int kk = 0; kk < CIN / 16; ++kk)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float* col = w + (size_t)(kk * 16 + 4 * g) * cout + nt * 16 + r16;
        f[kk][nt] = to_bf16x4(col[0], col[cout], col[2 * cout], col[3 * cout]);
      }
  }
  __device__ __forceinline__ void retire() {
#pragma unroll
    for (int kk = 0; kk < CIN / 16; ++kk)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 t = __builtin_bit_cast(f32x2, f[kk][nt]);
        float lo = t.x, hi = t.y;
        pin(lo);
        pin(hi);
        const f32x2 u = {lo, hi};
        f[kk][nt] = __builtin_bit_cast(bf16x4, u);
      }
  }
  __device__ __forceinline__ void tile(const float* sU, int cpi, int row0, int r16, int g, f32x4 (&acc)[NT]) const {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = zero4();
#pragma unroll
    for (int kk = 0; kk < CIN / 16; ++kk) {
      const float4 v = *reinterpret_cast<const float4*>(sU + (row0 + r16) * cpi + kk * 16 + 4 * g);
      const bf16x4 av = to_bf16x4(v.x, v.y, v.z, v.w);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma_bf16(av, f[kk][nt], acc[nt]);
    }
  }
};

// store a 16 x (NT*16) accumulator tile to the (sample, tile) slice `out` of the output tensor (rows past the slice
// are dropped by the address unit) and accumulate per-channel sum / sum^2.  Rows past the slice hold exact zeros (their
// u rows are zero), so the sums need no mask either.
template <int NT, int COUT, bool SB = false>
__device__ __forceinline__ void store_tile_stats(const f32x4 (&acc)[NT], BufRsrc out, int row0, int r16, int g,
                                                 float (&s1)[NT], float (&s2)[NT], int row_base = 0) {
  // row_base < 0 (rows in front of the slice) gives negative = huge unsigned offsets: dropped like the rows behind it.
  // The sums are those of the fp32 values (bf16 storage rounds only what goes to HBM).
  const int off = (row_base + row0 + g * 4) * COUT + r16;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = acc[nt][r];
      tile_store1s<SB, MWW_AUX_ST_P>(out, off + r * COUT + nt * 16, v);
      s1[nt] += v;
      s2[nt] = fmaf(v, v, s2[nt]);
    }
}

template <int NT, int COUT>
__device__ __forceinline__ void write_stat_partials(float (&s1)[NT], float (&s2)[NT], float* sRed, float* dst,
                                                    int tid, int wave, int r16, int g, const StatAcc& sacc) {
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
    publish_stat(sacc, dst, 2 * COUT, tid, v);
  }
}

// ------------------------------------------------------------------------------------------
// Tiles follow the first conv's output a0 = relu(conv1(x)): tile i computes a0 rows [64 i, 64 i + 64) exactly once
// (four 16-row MFMA tiles, no halo recompute: the im2col GEMM with K = K1*40 is the most expensive contraction of the
// network), keeps them in an LDS ring behind the K-1 rows carried over from the previous tile, and stores them to
// HBM for bwd_first_kernel.  The depthwise / pointwise part of the tile then produces the output rows
// [64 i - (K-1), 64 i + 64 - (K-1)) that those ring rows complete.
// LDS of the first-block / block stages as float offsets into a fused launch's LDS array
template <int K1, int C1, int COUT, int K, int S, bool BF = false>
struct FwdFirstLds {
  static constexpr int CP1 = pitch(C1), CPU = pitch_fu(C1, BF), RAP = halo_rows_padded(C1, K), TTP = tile_rows_padded(C1);
  static constexpr int TAIL = S > 1 ? K - 1 : 0;   // extra a0 rows of a single-tile window (fwd_first_body.inc "tail rows")
  static constexpr int XR = (TT + TAIL - 1) * S + K1, PX = FBINS + 1;
  static constexpr int up4(int v) { return (v + 3) / 4 * 4; }
};
template <int CIN, int COUT, int K, bool BF = false>
struct FwdBlockLds {
  static constexpr int CPA = pitch_fa(CIN), CPU = pitch_fu(CIN, BF), RAP = halo_rows_padded(CIN, K), TTP = tile_rows_padded(CIN);
};

template <int K1, int C1, int COUT, int K, int S, bool BF, bool SB = false, bool X6 = false>
__global__ __launch_bounds__(kThreads, ((S > 1 || COUT > 48 || K1 > 3) ? 2 : 4)) void fwd_first_kernel(FwdFirstArgs a) {
  typedef FwdFirstLds<K1, C1, COUT, K, S, BF> Lds;
  __shared__ __attribute__((aligned(16))) float sX[X6 ? 3 * Lds::XR * 96 / 4 : Lds::XR * Lds::PX];   // X6: three bf16 planes, 96-byte rows
  static_assert(!X6 || 3 * Lds::XR * 96 / 4 >= Lds::TTP * Lds::CPU, "the u tile of the x6 form lives in the x planes");
  __shared__ __attribute__((aligned(16))) float sA[Lds::RAP * Lds::CP1];
  __shared__ __attribute__((aligned(16))) float sU[X6 ? 4 : Lds::TTP * Lds::CPU];
  __shared__ __attribute__((aligned(16))) float sPw[X6 ? C1 * COUT : 4];   // X6: the 1x1 weights
  __shared__ __attribute__((aligned(16))) float sRed[4 * 2 * COUT];
  __shared__ XShared sXg;
#include "fwd_first_body.inc"
}

// ------------------------------------------------------------------------------------------
template <int CIN, int COUT, int K, bool BF, bool SB = false>
__global__ __launch_bounds__(kThreads, (CIN > 48 ? 2 : (K > 13 ? 3 : 4))) void fwd_block_kernel(FwdBlockArgs a) {
  typedef FwdBlockLds<CIN, COUT, K, BF> Lds;
  __shared__ __attribute__((aligned(16))) float sA[Lds::RAP * Lds::CPA];
  __shared__ __attribute__((aligned(16))) float sU[Lds::TTP * Lds::CPU];
  __shared__ __attribute__((aligned(16))) float sRed[4 * 2 * COUT];
  // BN_{k-1} folded scale | shift, typed float4: the commit reads them as ONE ds_read_b128 per lane (64 banks: the 12 or 16
  // distinct float4 of a wave do not collide).  As float arrays the compiler could not prove the 16-byte alignment and split every
  // read into three dword accesses, which at 48 channels wrap onto the 32 dword banks two-way (tools/ubench/lds_patterns "table4").
  __shared__ float4 sAff4[2 * CIN / 4];
  float* const sScale = reinterpret_cast<float*>(sAff4);
  float* const sShift = sScale + CIN;
#include "fwd_block_body.inc"
}

// ------------------------------------------------------------------------------------------
// Per-channel reduction of the per-workgroup partials: one workgroup per channel, 2 x 128 threads
// stride over the G partials (fp64, fixed order => bit-reproducible), then thread 0 folds the result.
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

// sum of part[j][stat][c] over j for stat = tid>>7, returned to threads 0 (stat 0) and 128 (stat 1)
__device__ __forceinline__ double reduce_partials_256(const float* part, int G, int C, int c, double* sAcc, int tid) {
  const int stat = tid >> 7, jp = tid & 127;
  // all loads of a thread are issued before the first use (one memory round trip for G <= 1024)
  double total = 0.0;
  for (int j0 = jp; j0 < G; j0 += 128 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + 128 * u;
      v[u] = (j < G) ? part[(size_t)j * 2 * C + stat * C + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) total += (double)v[u];
  }
  sAcc[tid] = total;
  __syncthreads();
  if (jp < 8) {
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) v += sAcc[stat * 128 + jp * 16 + i];
    sAcc[256 + stat * 8 + jp] = v;
  }
  __syncthreads();
  double r = 0.0;
  if (jp == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) r += sAcc[256 + stat * 8 + i];
  }
  return r;
}

#ifndef MWW_BLOCK_TU   // defined once, in mww_lib.hip (the block-kernel translation units skip it)
__global__ __launch_bounds__(kThreads) void bn_fwd_finalize_kernel(BnFwdFinalizeArgs a) {
  __shared__ __attribute__((aligned(16))) double sAcc[256 + 16];
  __shared__ double sOut[2];
  const int tid = threadIdx.x, c = blockIdx.x;
  // per-channel parameters are fetched together with the partials (one memory round trip, not two)
  float gam = 0.f, bet = 0.f, mm = 0.f, mv = 0.f;
  if (tid == 0) {
    gam = a.gamma[c];
    bet = a.beta[c];
    mm = a.moving_mean[c];
    mv = a.moving_var[c];
  }
  const double r = reduce_partials_256(a.stat_part, a.G, a.C, c, sAcc, tid);
  if ((tid & 127) == 0) sOut[tid >> 7] = r;
  __syncthreads();
  if (tid == 0) {
    const double m = sOut[0] * (double)a.inv_n;
    double var = sOut[1] * (double)a.inv_n - m * m;  // biased batch variance (Keras BN)
    if (var < 0.0) var = 0.0;
    const float meanf = (float)m, varf = (float)var;
    const float rstd = 1.0f / sqrtf(varf + kBnEps);
    const float sc = gam * rstd;
    a.scale[c] = sc;
    a.shift[c] = bet - meanf * sc;
    a.mean[c] = meanf;
    a.rstd[c] = rstd;
    if (a.update_moving) {
      a.moving_mean[c] = mm * kBnMomentum + meanf * (1.0f - kBnMomentum);
      a.moving_var[c] = mv * kBnMomentum + varf * (1.0f - kBnMomentum);
    }
  }
}
#endif

// sync-BN: this rank's partials [G][2][C] -> sums [2][C] in a fixed order, ready to be all-reduced
struct StatCollapseArgs {
  const float* part;
  int G, C;
  float* out;   // [2][C]
};
#ifndef MWW_BLOCK_TU   // defined once, in mww_lib.hip (the block-kernel translation units skip it)
__global__ __launch_bounds__(kThreads) void stat_collapse_kernel(StatCollapseArgs a) {
  __shared__ __attribute__((aligned(16))) double sAcc[256 + 16];
  const int tid = threadIdx.x, c = blockIdx.x;
  const double r = reduce_partials_256(a.part, a.G, a.C, c, sAcc, tid);
  if ((tid & 127) == 0) a.out[(tid >> 7) * a.C + c] = (float)r;
}
#endif

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

#ifndef MWW_BLOCK_TU   // defined once, in mww_lib.hip (the block-kernel translation units skip it)
__global__ void bn_eval_prepare_kernel(BnEvalPrepareArgs a) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < a.C) {
    const float rstd = 1.0f / sqrtf(a.moving_var[c] + kBnEps);
    const float sc = a.gamma[c] * rstd;
    a.scale[c] = sc;
    a.shift[c] = a.beta[c] - a.moving_mean[c] * sc;
  }
}
#endif

}  // namespace mww

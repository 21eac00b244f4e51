// Batch assembly kernel: ragged gather + left pad / truncate + uint16->f32 scale + SpecAugment
// zeroing, one workgroup per output window.  Replaces the per-sample Python loop of reference
// microwakeword/data.py:555-569 (fixed_length_spectrogram :74-118, scaling :268-269,
// spec_augment :32-71) and the shuffle copy :591-597 (the host hands the windows over already
// in shuffled order).  Pure data movement: 4 elements per lane, rows of 40 bins are contiguous.
#pragma once
#include "common.hip.h"
#include "../../include/mww.h"

namespace mww {

struct AssembleArgs {
  const void* store[MWW_MAX_STORES];
  int dtype[MWW_MAX_STORES];
  const mww_window* win;   // [B] in output order
  const int* masks;        // [B][ntm+nfm][2]
  float* x;                // [B][T][40]
  int B, T, ntm, nfm;
  // labels / per-sample weights that arrived in the same mailbox: copied to HBM by the first n_targets workgroups
  const float* y_src;
  const float* sw_src;
  float* y_dst;
  float* sw_dst;
  int n_targets;
  int split;               // workgroups per window (each gathers a contiguous share of the window's rows)
};

constexpr int kMaxMasks = 16;

__global__ __launch_bounds__(kThreads) void assemble_kernel(AssembleArgs a) {
  __shared__ __attribute__((aligned(16))) int sMask[kMaxMasks * 2];
  const int j = blockIdx.x / a.split, part = blockIdx.x - j * a.split;
  const int tid = threadIdx.x;
  const int nm = a.ntm + a.nfm;
  const mww_window w = a.win[j];   // issued together with the mask loads: one memory round trip, not two
  if (tid < nm * 2) sMask[tid] = a.masks[(size_t)j * nm * 2 + tid];
  if (part == 0 && j < a.n_targets && tid == 64) a.y_dst[j] = a.y_src[j];
  if (part == 0 && j < a.n_targets && tid == 128) a.sw_dst[j] = a.sw_src[j];
  __syncthreads();
  const int dtype = a.dtype[w.store];
  const unsigned short* s16 = reinterpret_cast<const unsigned short*>(a.store[w.store]);
  const float* s32 = reinterpret_cast<const float*>(a.store[w.store]);
  float* dst = a.x + (size_t)j * a.T * FBINS;
  constexpr int Q = FBINS / 4;
  constexpr int U = 4;   // float4 groups in flight per thread: all loads of a workgroup's share are issued before the first store
  const int n_all = a.T * Q, share = (n_all + a.split - 1) / a.split;
  const int i_end = min(n_all, (part + 1) * share);
  for (int i0 = part * share + tid; i0 < i_end; i0 += kThreads * U) {
    float4 v[U];
    int tt[U], qq[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * kThreads;
      const int t = i / Q, q = i - t * Q;
      tt[u] = t;
      qq[u] = q;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int r = t - w.pad_rows;
      if (i < i_end && r >= 0 && r < w.copy_rows) {
        const size_t e = (size_t)w.src_elem + (size_t)r * FBINS + q * 4;
        if (dtype == MWW_DTYPE_U16) {
          const ushort4 u16 = *reinterpret_cast<const ushort4*>(s16 + e);
          v[u].x = (float)u16.x * 0.0390625f;   // data.py:268-269
          v[u].y = (float)u16.y * 0.0390625f;
          v[u].z = (float)u16.z * 0.0390625f;
          v[u].w = (float)u16.w * 0.0390625f;
        } else {
          v[u] = *reinterpret_cast<const float4*>(s32 + e);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * kThreads;
      if (i >= i_end) continue;
      const int t = tt[u], q = qq[u];
      bool row_masked = false;
      for (int m = 0; m < a.ntm; ++m) {
        const int t0 = sMask[2 * m], tw = sMask[2 * m + 1];
        row_masked = row_masked || (t >= t0 && t < t0 + tw);
      }
      if (row_masked) {
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        for (int m = a.ntm; m < nm; ++m) {
          const int f0 = sMask[2 * m], fw = sMask[2 * m + 1];
          const int f = q * 4;
          if (f + 0 >= f0 && f + 0 < f0 + fw) v[u].x = 0.f;
          if (f + 1 >= f0 && f + 1 < f0 + fw) v[u].y = 0.f;
          if (f + 2 >= f0 && f + 2 < f0 + fw) v[u].z = 0.f;
          if (f + 3 >= f0 && f + 3 < f0 + fw) v[u].w = 0.f;
        }
      }
      *reinterpret_cast<float4*>(dst + (size_t)t * FBINS + q * 4) = v[u];
    }
  }
}

}  // namespace mww

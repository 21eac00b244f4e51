// Shared definitions for the gfx950 kernels of the microWakeWord train step.
//
// Geometry used by every conv kernel:
//   * one workgroup = 256 threads = 4 wave64; it owns whole samples (grid-stride over the batch)
//     and walks each sample in time tiles of TT = 64 output frames, so the only cross-tile state
//     (the depthwise halo) stays inside the workgroup's LDS / L1.
//   * VALU phases (BN+ReLU, depthwise, masks, stats) map thread -> (channel c, time chunk):
//     lanes of a wave cover consecutive channels of one frame => 128/192-byte coalesced rows in
//     HBM and conflict-free LDS rows.
//   * GEMM-shaped phases (3x1 first conv as im2col, 1x1 pointwise, their transposes and weight
//     gradients) run on v_mfma_f32_16x16x4_f32 (exact fp32 fma chain): lane l supplies
//     A[i=l&15][k=l>>4], B[k=l>>4][j=l&15] and owns C/D[row=(l>>4)*4+r][col=l&15].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mww {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int TT = 64;        // output frames per time tile (4 waves x one 16-row MFMA tile)
constexpr int FBINS = 40;     // mel bins of the micro-frontend features
constexpr float kBnEps = 1e-3f;
constexpr float kBnMomentum = 0.99f;
constexpr float kKerasEps = 1e-7f;

// Loss of train.py:206, BinaryCrossentropy(from_logits=False) on a Dense(1, activation="sigmoid") output.  Keras 3
// (the reference needs TF >= 2.16) caches the logits on the sigmoid output (`_keras_logits`, keras/src/activations) and
// the TensorFlow backend's binary_crossentropy evaluates tf.nn.sigmoid_cross_entropy_with_logits on them: no clipping,
// dL/dz = p - y everywhere - the default here.  The probability form with the [1e-7, 1-1e-7] clip (zero
// gradient for saturated samples) is what a graph without the cached logits computes: option "bce_from_logits" 0.
constexpr int kHeadTraining = 1, kHeadClippedLoss = 2;   // bits of the head kernels' `training` argument
__device__ __forceinline__ float bce_value(float z, float pr, float yy, bool clipped_form) {
  if (!clipped_form) return fmaxf(z, 0.f) - z * yy + log1pf(expf(-fabsf(z)));
  const float pc = fminf(fmaxf(pr, kKerasEps), 1.0f - kKerasEps);
  return -(yy * logf(pc) + (1.0f - yy) * logf(1.0f - pc));
}
__device__ __forceinline__ float bce_dz(float pr, float yy, bool clipped_form) {
  if (clipped_form && ((pr < kKerasEps) || (pr > 1.0f - kKerasEps))) return 0.f;
  return pr - yy;
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Optional reduced-precision mode of the 1x1 contractions (BASELINE configs[4] "bf16 with MFMA pointwise"):
// operands rounded to bf16 (RNE, v_cvt_pk_bf16_f32), products exact, fp32 accumulation.  One
// v_mfma_f32_16x16x16_bf16 covers K = 16 in the time the f32 form covers K = 4.  Lane l supplies
// A[i = l&15][k = 4*(l>>4) + j] and B[k = 4*(l>>4) + j][n = l&15], j < 4; C/D as the f32 form.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma_bf16(bf16x4 a, bf16x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}

__device__ __forceinline__ bf16x4 to_bf16x4(float a, float b, float c, float d) {
  bf16x4 v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
  return v;
}

// ---- fp32-grade products on the bf16 matrix pipe ("x6": the first convolution and its weight gradient, round 6) -----------
// v = s0 + s1 + s2 exactly, the three 8-bit slices of the fp32 significand as bf16 values (by truncation: and / sub / and /
// sub).  A product x * w is then the sum of nine slice products, each exact in fp32; the six with i + j <= 2 carry everything
// above 2^-24 of it (tools/ubench/mfma_bf16x9: max error 1.08e-7 of sum |x w| against 1.19e-7 for the f32 MFMA's fma chain),
// and six v_mfma_f32_16x16x32_bf16 (K = 32, ~17-19 cycles each) replace eight v_mfma_f32_16x16x4_f32 (K = 4, 32 cycles each).
// Lane l of the 16x16x32 form supplies A[i = l&15][k = 8*(l>>4) + e] and B[k = 8*(l>>4) + e][n = l&15], e < 8, as one
// 16-byte register quad; C/D as the f32 form.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 mfma_bf16k32(u32x4v a, u32x4v b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// the slices in the HIGH halves of three dwords (low halves zero)
__device__ __forceinline__ void split3(float v, unsigned& h0, unsigned& h1, unsigned& h2) {
  h0 = __float_as_uint(v) & 0xffff0000u;
  const float r1 = v - __uint_as_float(h0);
  h1 = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(h1);   // at most 8 significant bits are left: a bf16 value
  h2 = __float_as_uint(r2) & 0xffff0000u;
}
// two slices (high halves) -> one dword of two bf16: element 0 = a, element 1 = b
__device__ __forceinline__ unsigned pack_hi2(unsigned a, unsigned b) { return (a >> 16) | b; }
// slices of eight values as the three operand quads of the 16x16x32 form
__device__ __forceinline__ void split3x8(const float (&v)[8], u32x4v (&q)[3]) {
  unsigned h[3][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) split3(v[e], h[0][e], h[1][e], h[2][e]);
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int d = 0; d < 4; ++d) q[p][d] = pack_hi2(h[p][2 * d], h[p][2 * d + 1]);
}
// acc += x * w for operand slices xs[0..2], ws[0..2]: the six products with i + j <= 2, small terms first
__device__ __forceinline__ f32x4 mfma_x6(const u32x4v (&xs)[3], const u32x4v (&ws)[3], f32x4 acc) {
  acc = mfma_bf16k32(xs[2], ws[0], acc);
  acc = mfma_bf16k32(xs[1], ws[1], acc);
  acc = mfma_bf16k32(xs[0], ws[2], acc);
  acc = mfma_bf16k32(xs[1], ws[0], acc);
  acc = mfma_bf16k32(xs[0], ws[1], acc);
  acc = mfma_bf16k32(xs[0], ws[0], acc);
  return acc;
}
__device__ __forceinline__ void pin4(u32x4v& v) { asm volatile("" : "+v"(v)); }
// ds_read_b64_tr_b16 (gfx950): within each group of 16 lanes, lane p supplies the 8-byte-aligned LDS address of four 16-bit
// elements in[p][0..3] and lane i receives out[j] = in[4 j + (i >> 2)][i & 3], j < 4 (probed: tools/ubench/tr16_probe.hip).
// With lane p pointing at row k0 + (p >> 2), columns c0 + 4 (p & 3) .. + 3 of a row-major bf16 tile, lane i gets column
// c0 + i of rows k0 .. k0 + 3: a k-contiguous MFMA operand from a tile whose k runs over rows.  Returns two dwords.
__device__ __forceinline__ u32x2v lds_read_tr16(const void* lds_ptr) {
  typedef short s16x4t __attribute__((ext_vector_type(4)));
  const s16x4t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4t*)(lds_ptr));
  return __builtin_bit_cast(u32x2v, v);
}

__device__ __forceinline__ f32x4 zero4() {
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  return z;
}

// LDS row pitch: +4 floats keeps 16-byte alignment of float4 rows and breaks 2^n strides.
__host__ __device__ constexpr int pitch(int c) { return c + 4; }
// Forward kernels, per array (tools/lds_banks.py; SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE was 34 % with c + 4 everywhere):
//   input tile (float4 commits, (channel, chunk) window reads): at c = 48 the 64 lanes of a wave straddle two chunks L = 13
//   rows apart - a pitch of 48 (= 16 mod 32, L odd) puts the second chunk's 16 lanes on the banks the first leaves free;
//   u tile (written per (channel, chunk), read as the MFMA A operand: lane (r16, g) reads row r16, column 4 kk + g):
//   c + 2 = 2 or 18 mod 32 spreads the 16 rows x 2 columns of a 32-lane group over 32 banks (c + 4: two-way).  The bf16
//   mode reads that operand as float4 and keeps 16-byte rows.
#ifndef MWW_FWD_PITCH_A   // tuning builds: 0 = c + 4 on the input tile
#define MWW_FWD_PITCH_A 1
#endif
#ifndef MWW_FWD_PITCH_U   // tuning builds: 0 = c + 4 on the u tile
#define MWW_FWD_PITCH_U 1
#endif
__host__ __device__ constexpr int pitch_fa(int c) { return (MWW_FWD_PITCH_A && c == 48) ? 48 : c + 4; }
__host__ __device__ constexpr int pitch_fu(int c, bool bf) { return (MWW_FWD_PITCH_U && !bf) ? c + 2 : c + 4; }

// pitch of W_pw^T in the 256-thread backward kernels: its rows are read one apart by the fp32 contraction (the two rows of a
// 32-lane LDS group land on disjoint banks at 16 mod 32: 48 / 48 / 80 for 32 / 48 / 64 channels; c + 4 was two-way) and four
// apart by the bf16 one (where c + 4 already is conflict-free)
__host__ __device__ constexpr int pitch_wt(int c, bool bf) { return bf ? c + 4 : (c % 32 == 16 ? c : c + 16); }

// number of time chunks / chunk length of the (channel, chunk) VALU mapping
__host__ __device__ constexpr int nchunks(int c) { return kThreads / c; }
__host__ __device__ constexpr int chunk_len(int c) { return (TT + nchunks(c) - 1) / nchunks(c); }
// rows the (channel, chunk) mapping may touch: tile rows rounded up to whole chunks (+ halo).  LDS
// tiles are allocated with this many rows and kept zero past the valid ones, so the register
// windows are loaded without per-element bounds checks (those compile to exec-mask branches).
__host__ __device__ constexpr int tile_rows_padded(int c) { return nchunks(c) * chunk_len(c); }
__host__ __device__ constexpr int halo_rows_padded(int c, int k) { return tile_rows_padded(c) + k - 1; }

// ---- bounds-checked tile access through buffer resources ------------------------------------------------
// A (sample, time tile) slice of a tensor is addressed through a buffer resource whose num_records is the slice's
// valid byte count: loads past it return 0 and stores past it are dropped by the address unit.  The tile loops
// therefore carry no per-lane bounds branches.  That is not cosmetic: a conditional load (`v = 0; if (ok) v = *p`)
// reaches its consumer through a phi whose copies the register allocator places right behind the load, and the
// wait-count pass then drains the whole memory queue there (`s_waitcnt vmcnt(0)` in the middle of the prefetch,
// seen in the round-1 ISA of every block kernel) - the "register prefetch" of the next tile became a synchronous
// load.  Straight-line buffer loads keep every wait at the point of first use with an exact count.
typedef __amdgpu_buffer_rsrc_t BufRsrc;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// base must be workgroup-uniform (the descriptor lives in scalar registers); bytes <= 0 makes every access a no-op
__device__ __forceinline__ BufRsrc tile_rsrc(const void* base, int bytes) {
  // the count is pinned to a scalar register: clamps like max(0, min(n, TT)) are otherwise selected as v_med3_i32
  // and a descriptor with a vector-register word is applied through a waterfall loop
  const int n = __builtin_amdgcn_readfirstlane(bytes > 0 ? bytes : 0);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, n, 0x00020000);
}
// ---- cache policies of the activation / gradient streams (round 3) --------------------------------------------
// A train step moves ~300 MB of activations and gradients through 8 x 4 MB of L2 and 256 MB of memory-side cache (MALL).
// With the default policy every store leaves a dirty L2 line (written back in a burst when the kernel ends and its release
// fence runs) and every load allocates in the MALL, so tensors that are dead after the read push out the ones the backward
// pass comes back for.  Per stream (tools/gpu_variants.sh sweeps r3t-r3w, same session, +-0.0015 ms):
//   stores of p_k (forward) and g_k (backward): sc1 = written through to memory as they are produced      -1.5 us per launch
//   store of a0 (read again only by the very last backward launch): nt                                     -0.5 us
//   the weight-gradient partial rows a backward launch writes in its last microsecond: written through     -0.5 us per launch
//   backward loads of p_k / g_k / a0 (last use of each): nt = no allocation                                -1...2 us per launch,
//                                                  and the NEXT step's forward launches find their inputs: -1.5 us each
//   forward loads (p_{k-1}: the backward pass reads it again) and the head's read of p_L: default (nt: +2.5 us per launch);
//   sc1|nt on the forward stores: +2 us per forward launch; nt on the first block's gathers from the feature stores: no effect.
// 0.329 -> 0.312 ms per step in total.  The macros exist for the sweeps (tools/build_variant.sh -DMWW_AUX_...=n).
// Conv/BN graph kernels (Inception, 1.2 GB of traffic per step - nothing survives in the MALL): forward stores written
// through -1.2 % (0.889 -> 0.879 ms); written-through gradient stores +1 %, nt loads of (g, p) +1.7 %: left at the default.
#ifndef MWW_AUX_ST_P
#define MWW_AUX_ST_P 16
#endif
#ifndef MWW_AUX_ST_G
#define MWW_AUX_ST_G 16
#endif
#ifndef MWW_AUX_ST_A0
#define MWW_AUX_ST_A0 2
#endif
#ifndef MWW_AUX_LD_FP
#define MWW_AUX_LD_FP 0
#endif
#ifndef MWW_AUX_LD_BP
#define MWW_AUX_LD_BP 2
#endif
#ifndef MWW_AUX_LD_A0
#define MWW_AUX_LD_A0 2
#endif
#ifndef MWW_AUX_LD_PK
#define MWW_AUX_LD_PK 2
#endif
#ifndef MWW_AUX_LD_GK
#define MWW_AUX_LD_GK 2
#endif
#ifndef MWW_AUX_LD_XF
#define MWW_AUX_LD_XF 0
#endif
#ifndef MWW_AUX_LD_XB
#define MWW_AUX_LD_XB 0
#endif
#ifndef MWW_AUX_LD_HP
#define MWW_AUX_LD_HP 0
#endif
#ifndef MWW_AUX_ST_GP
#define MWW_AUX_ST_GP 1
#endif
#ifndef MWW_AUX_GR_ST_P
#define MWW_AUX_GR_ST_P 1
#endif
#ifndef MWW_AUX_GR_ST_G
#define MWW_AUX_GR_ST_G 0
#endif
#ifndef MWW_AUX_GR_LD_GOLD
#define MWW_AUX_GR_LD_GOLD 0
#endif
#ifndef MWW_AUX_GR_ST_GP
#define MWW_AUX_GR_ST_GP 0
#endif
#ifndef MWW_AUX_GR_LD_DP
#define MWW_AUX_GR_LD_DP 0
#endif
// a plain store written through to memory (sc1) when WT, for the epilogues that address with pointers
template <int WT>
__device__ __forceinline__ void store_stream(float* p, float v) {
  if constexpr (WT != 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
// AUX = cache policy of the access (gfx940+ buffer instructions: 1 = sc0, 2 = nt, 16 = sc1; see "cache policies" below)
template <int AUX = 0>
__device__ __forceinline__ float4 tile_load4(BufRsrc r, int byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, AUX);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
template <int AUX = 0>
__device__ __forceinline__ uint2 tile_load2(BufRsrc r, int byte_off) {
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, AUX);
  uint2 o;
  o.x = v.x;
  o.y = v.y;
  return o;
}
template <int AUX = 0>
__device__ __forceinline__ float tile_load1(BufRsrc r, int byte_off) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ void tile_store1(BufRsrc r, int byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, byte_off, 0, AUX);
}
// ---- bf16 storage of the block outputs p_k and the stashed gradients g_k ("storage_bf16", BASELINE configs[4]) ----
// SB = true: the tensor holds bf16 (RNE on store, exact widening on load); arithmetic, accumulation and the BN sums
// stay fp32.  Element / float4-group indices are the same in both modes, only the byte offsets differ.
template <bool SB, int AUX = 0>
__device__ __forceinline__ float4 tile_load4s(BufRsrc r, int group) {   // group = index of a 4-channel group in the slice
  if constexpr (!SB) {
    return tile_load4<AUX>(r, group * 16);
  } else {
    const uint2 v = tile_load2<AUX>(r, group * 8);
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                       __uint_as_float(v.y & 0xffff0000u));
  }
}
template <bool SB, int AUX = 0>
__device__ __forceinline__ void tile_store1s(BufRsrc r, int elem, float v) {   // elem = element index in the slice
  if constexpr (!SB) {
    tile_store1<AUX>(r, elem * 4, v);
  } else {
    const __bf16 h = (__bf16)v;
    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h), r, elem * 2, 0, AUX);
  }
}
template <bool SB>
__device__ __forceinline__ float load_elem(const float* base, size_t elem) {   // scalar read of a stored tensor
  if constexpr (!SB) {
    return base[elem];
  } else {
    // through the aligned dword that holds the element: 16-bit loads fill register halves one after the other
    // (d16 / d16_hi chains wait for each other), a batch of dword loads stays independent
    const unsigned dw = reinterpret_cast<const unsigned*>(base)[elem >> 1];
    return __uint_as_float((elem & 1) ? (dw & 0xffff0000u) : (dw << 16));
  }
}
__host__ __device__ constexpr int elem_bytes(bool sb) { return sb ? 2 : 4; }
// the stored tensor at element offset `elem` (both modes keep fp32-sized allocations; bf16 uses the first half)
template <bool SB>
__device__ __forceinline__ const float* elem_ptr(const float* base, size_t elem) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + elem * (SB ? 2 : 4));
}
template <bool SB>
__device__ __forceinline__ float* elem_ptr(float* base, size_t elem) {
  return reinterpret_cast<float*>(reinterpret_cast<char*>(base) + elem * (SB ? 2 : 4));
}

// lane offset that is out of range for every resource (lanes that take no part in an access)
constexpr int kOobOffset = 0x40000000;

// scheduling hint for a VALU phase that starts with a window of LDS reads: put every DS read of the region in front of
// its VALU work (the default schedule interleaves "two reads, wait, four VALU" to save registers, which exposes one LDS
// round trip per pair at two waves per SIMD)
__device__ __forceinline__ void lds_reads_first() {
  __builtin_amdgcn_sched_group_barrier(0x100, 64, 0);   // DS reads
  __builtin_amdgcn_sched_group_barrier(0x002, 2048, 0);  // then VALU
}

// The workgroups that share a CU are not served equally: the SIMD arbiter prefers the oldest wave, so the workgroup
// dispatched first (blockIdx < 256 on the 256-CU part) ran its tile loop 14-16 % faster than the one dispatched into
// the same CU after it (round-2 timeline: 76.8k vs 89.3k cycles in bwd_block), and the launch lasts as long as the
// slowest.  Rotating the wave priority per work item - co-resident workgroups differ in (blockIdx >> 8) - gives each
// its turn in front: the two halves of the backward grids then finish together (-1...-2 us per backward launch).  The
// forward kernels (four workgroups per CU) measured no gain for K = 9 / 13 and +3 us for K = 21: not used there.
__device__ __forceinline__ void rotate_priority(int item, int levels) {
  const int p = ((int)(blockIdx.x >> 8) + item) & (levels - 1);
  if (levels == 2) {
    if (p) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
  } else {
    if (p == 0) __builtin_amdgcn_s_setprio(0);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
  }
}

// Staggered start (round 5; MWW_STAGGER_* = s_sleep units of 64 clocks per position, 0 = off).  Every workgroup of a launch
// requests its first tile at once - 27 MB in ~5 us with nothing to compute - and the workgroups that share a CU then run
// their phases in lockstep.  Delaying the workgroup dispatched `pos`-th into its CU (blockIdx >> 8 on the 256-CU part) by a
// fraction of that burst lets the earlier one start computing while the later one's rows arrive.  Backward kernels (two per
// CU), same-session A/B (profiles/round5_stagger_ab.txt): 47 units (~1.3 us) 0.3030 -> 0.2983 ms per step, 94 units +-0; the
// forward kernels (four per CU) gained nothing at 31 units per position.
#ifndef MWW_STAGGER_BWD
#define MWW_STAGGER_BWD 47
#endif
#ifndef MWW_STAGGER_FWD
#define MWW_STAGGER_FWD 0
#endif
#ifndef MWW_STAGGER_GRAPH   // the conv / BN graph kernels (3-4 workgroups per CU and launch)
#define MWW_STAGGER_GRAPH 0
#endif
// Conv / BN graph kernels, static shapes (round 5, kernels_graph.hip.h):
//   MWW_G_FWD_DIRECT  1 = the stem shape (5 x 40 input bins), 2 = every static forward convolution: output rows and BN sums
//                     straight from the MFMA accumulators (no output tile in LDS, one barrier less per window);
//   MWW_G_WGRAD_EVEN  the task tiles a weight gradient's four waves cannot share out evenly are cut into filter tiles.
#ifndef MWW_G_FWD_DIRECT
#define MWW_G_FWD_DIRECT 1
#endif
#ifndef MWW_G_WGRAD_EVEN
#define MWW_G_WGRAD_EVEN 1
#endif
#ifndef MWW_G_STEM_BREG
#define MWW_G_STEM_BREG 0
#endif
#ifndef MWW_G_WGRAD_XG_NARROW
#define MWW_G_WGRAD_XG_NARROW 0
#endif
template <int UNITS>
__device__ __forceinline__ void stagger_start() {
  if constexpr (UNITS > 0) {
    const int pos = (int)(blockIdx.x >> 8);
    for (int i = 0; i < pos; ++i) __builtin_amdgcn_s_sleep(UNITS);
  }
}

// keep a value (and the loads that produce it) from sinking below this point: used to retire the
// prologue's weight loads before the tile loop, so waits inside the loop never drain the prefetch
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }
// the same for an index: what is derived from it afterwards cannot be hoisted above this point (kernels_bwdw.hip.h: MWW_WIDE_RELANE_K)
__device__ __forceinline__ void pin(int& v) { asm volatile("" : "+v"(v)); }

// profiling only (build with -DMWW_PROFILE, see tools/phase_clocks.py): phase ablation by the "ablate" option bits
// (results invalid) and per-phase shader-clock accounting of one thread ("ablate" bit 16).  Compiled out of the
// shipped library: the default build neither reads `ablate` nor carries the ~17 VGPRs of the counters.
#ifdef MWW_PROFILE
#define MWW_ABLATE(a, bits) ((a).ablate & (bits))
#else
#define MWW_ABLATE(a, bits) 0
#endif
#ifdef MWW_PROFILE
#define MWW_PC_DECL PhaseClock pc;
#define MWW_PC_START(en) pc.start(en)
#define MWW_PC_MARK(i) pc.mark(i)
#define MWW_PC_DUMP(p) do { if (p) pc.dump(p); } while (0)
#define MWW_PC_AT(i) pc.at(i)
#else
#define MWW_PC_AT(i) ((void)0)
#define MWW_PC_DECL
#define MWW_PC_START(en) ((void)0)
#define MWW_PC_MARK(i) ((void)0)
#define MWW_PC_DUMP(p) ((void)0)
#endif
constexpr int kClkSlots = 12;   // per workgroup: 8 phase accumulators + entry / loop start / loop end / exit timestamps
struct PhaseClock {
  unsigned long long last = 0, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp[4] = {0, 0, 0, 0};
  bool on = false;
  __device__ __forceinline__ void at(int i) { stamp[i] = __builtin_amdgcn_s_memtime(); }
  __device__ __forceinline__ void start(bool enable) {
    on = enable;
    if (on) last = __builtin_amdgcn_s_memtime();
  }
  __device__ __forceinline__ void mark(int slot) {
    if (on) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      acc[slot] += t - last;
      last = t;
    }
  }
  __device__ __forceinline__ void dump(unsigned long long* dst) const {
    if (on) {
      for (int i = 0; i < 8; ++i) dst[i] = acc[i];
      for (int i = 0; i < 4; ++i) dst[8 + i] = stamp[i];
    }
  }
};

// q = a / b, r = a % b for 0 <= a < 2^22 and b > 0 known only at run time: a reciprocal, a multiply and a one-step
// correction (~10 VALU) instead of the ~35-instruction integer division the compiler emits; the conv/BN graph kernels
// decompose thread / element indices by run-time channel counts dozens of times per window.
__device__ __forceinline__ void fast_divmod(int a, int b, int& q, int& r) {
  q = (int)((float)a * __frcp_rn((float)b));   // within one of the quotient
  r = a - q * b;
  if (r >= b) {
    ++q;
    r -= b;
  } else if (r < 0) {
    --q;
    r += b;
  }
}
__device__ __forceinline__ int fast_div(int a, int b) {
  int q, r;
  fast_divmod(a, b, q, r);
  return q;
}

// workgroup-uniform values read from LDS / memory: move them to scalar registers
__device__ __forceinline__ int uniform_int(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uniform_i64(long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
  return reinterpret_cast<const void*>((uintptr_t)uniform_i64((long long)(uintptr_t)p));
}

// sum over the four 16-lane groups of a wave (lanes l, l^16, l^32, l^48)
__device__ __forceinline__ float sum_over_groups(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// ---- BN statistics hand-over without a finalize launch (DESIGN §4 "statistics hand-over") ---------------
// Producer: every workgroup adds its partial sums to one of kStatRows replicated fp64 accumulator rows with
// memory-side atomics (no fence, no ticket: the kernel boundary publishes them).  First consumer: every
// workgroup sums the kStatRows rows in its prologue and folds them exactly like bn_*_finalize_kernel does;
// workgroup 0 also writes the folded arrays / moving statistics / dgamma, dbeta for the later kernels of the
// step.  The accumulators are double-buffered by launch parity: a producer clears the rows its next-but-one
// launch will add to.  fp64 sums of fp32 partials are exact unless the partials span more than 2^29 in
// magnitude, so the result does not depend on the arrival order in practice (see DESIGN).
constexpr int kStatRows = 8;

struct StatAcc {
  double* acc;     // [kStatRows][2C] rows this launch adds to (null: write the per-workgroup partial row instead)
  double* clear;   // the other parity's rows
};

// called by threads tid < C2 with their column's per-workgroup sum
__device__ __forceinline__ void publish_stat(const StatAcc& s, float* row, int C2, int tid, float v) {
  if (s.acc) {
    unsafeAtomicAdd(s.acc + (size_t)(blockIdx.x % kStatRows) * C2 + tid, (double)v);
    // every row of the other parity is cleared whatever the grid size (a launch with fewer than kStatRows workgroups
    // must not leave rows of an earlier, larger launch behind: the consumers always sum all kStatRows rows)
    // (agent-scope stores like every other access to these rows: no XCD's L2 keeps a line of them)
    for (int r = blockIdx.x; r < kStatRows; r += gridDim.x)
      __hip_atomic_store(s.clear + (size_t)r * C2 + tid, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    row[tid] = v;
  }
}

struct BnFoldArgs {      // forward statistics of a BN layer, folded by the first kernel that consumes them
  const double* acc;     // [kStatRows][2][C] sums of x, x^2 (null: scale / shift were written by a finalize / eval-prepare launch)
  float inv_n;           // 1 / (B*T)
  int update_moving;
  const float* gamma;
  const float* beta;
  float* moving_mean;
  float* moving_var;
  float* scale;          // published by workgroup 0 (same arrays bn_fwd_finalize_kernel writes)
  float* shift;
  float* mean;
  float* rstd;
};

// one thread per channel; same arithmetic as bn_fwd_finalize_kernel
__device__ __forceinline__ void bn_fold_channel(const BnFoldArgs& f, int C, int ch, float& sc, float& sh, float& meanf, float& rstd) {
  double s1 = 0.0, s2 = 0.0;
  double v1[kStatRows], v2[kStatRows];
#pragma unroll
  for (int j = 0; j < kStatRows; ++j) {
    v1[j] = f.acc[(size_t)j * 2 * C + ch];
    v2[j] = f.acc[(size_t)j * 2 * C + C + ch];
  }
  const float gam = f.gamma[ch], bet = f.beta[ch];
  // workgroup 0 also updates the moving statistics: their old values travel with the sums (a load behind the stores
  // below would be a second, dependent round trip that only this workgroup pays - and the launch ends with it)
  float mm_old = 0.f, mv_old = 0.f;
  if (blockIdx.x == 0 && f.update_moving) {
    mm_old = f.moving_mean[ch];
    mv_old = f.moving_var[ch];
  }
#pragma unroll
  for (int j = 0; j < kStatRows; ++j) {
    s1 += v1[j];
    s2 += v2[j];
  }
  const double m = s1 * (double)f.inv_n;
  double var = s2 * (double)f.inv_n - m * m;   // biased batch variance (Keras BN)
  if (var < 0.0) var = 0.0;
  meanf = (float)m;
  const float varf = (float)var;
  rstd = 1.0f / sqrtf(varf + kBnEps);
  sc = gam * rstd;
  sh = bet - meanf * sc;
  if (blockIdx.x == 0) {
    f.scale[ch] = sc;
    f.shift[ch] = sh;
    f.mean[ch] = meanf;
    f.rstd[ch] = rstd;
    if (f.update_moving) {
      f.moving_mean[ch] = mm_old * kBnMomentum + meanf * (1.0f - kBnMomentum);
      f.moving_var[ch] = mv_old * kBnMomentum + varf * (1.0f - kBnMomentum);
    }
  }
}

struct BnGradFoldArgs {  // backward statistics (sum g, sum g*xhat) of a BN layer, folded by the first consumer
  const double* acc;     // [kStatRows][2][C] (null: c1 / mg / mgx were written by a finalize launch)
  float inv_n;
  float dscale;
  const float* gamma;
  float* c1;             // published by workgroup 0 (same arrays bn_bwd_finalize_kernel writes)
  float* mg;
  float* mgx;
  float* dgamma;
  float* dbeta;
};

__device__ __forceinline__ void bn_grad_fold_channel(const BnGradFoldArgs& f, int C, int ch, float rstd, float& c1, float& mg, float& mgx) {
  double s1 = 0.0, s2 = 0.0;
  double v1[kStatRows], v2[kStatRows];
#pragma unroll
  for (int j = 0; j < kStatRows; ++j) {
    v1[j] = f.acc[(size_t)j * 2 * C + ch];
    v2[j] = f.acc[(size_t)j * 2 * C + C + ch];
  }
  const float gam = f.gamma[ch];
#pragma unroll
  for (int j = 0; j < kStatRows; ++j) {
    s1 += v1[j];
    s2 += v2[j];
  }
  c1 = gam * rstd;
  mg = (float)(s1 * (double)f.inv_n);
  mgx = (float)(s2 * (double)f.inv_n);
  if (blockIdx.x == 0) {
    f.dbeta[ch] = (float)s1 * f.dscale;
    f.dgamma[ch] = (float)s2 * f.dscale;
    f.c1[ch] = c1;
    f.mg[ch] = mg;
    f.mgx[ch] = mgx;
  }
}

}  // namespace mww

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
//   * staging: thread -> (group of 4 / 2 / 1 adjacent channels, frame group), buffer loads over the window's slab
//     (workgroup-uniform descriptor, 32-bit lane offsets, kGB rows per thread and tensor in flight: these
//     kernels are chains of memory round trips with a handful of windows per CU, so what counts is how many
//     workgroups are resident - registers are occupancy here);
//   * convolution: v_mfma_f32_16x16x4_f32 (exact fp32), per tap a [16 frames] x [4 channels] A tile of the window in LDS
//     (odd row pitch) against a [4 channels] x [16 filters] B tile of the zero-padded weights in LDS; the four waves
//     deal the window's 16-frame tiles among themselves;
//   * the data gradient is the same convolution run on the BN-backward-transformed output gradient
//     with reversed taps and transposed weights (read in place from the forward weights), its epilogue scatters
//     into the sources' gradient tensors (first consumer stores, later ones accumulate, the last one
//     also produces the BN statistics of that source);
//   * the weight gradient is dW = A^T B on the matrix cores with the (tap, input channel) tasks as rows, the filters as
//     columns and the frames as the contraction; task tiles are dealt to the waves and kept in accumulators across the
//     workgroup's windows; partials are summed by grad_final_kernel;
//   * BN statistics travel through replicated fp64 accumulator rows and are folded by the first launch that consumes
//     them (no finalize launches, see "statistics hand-over" below).
// Measured (profiles/round2_*inception*): the matrix cores do not make these launches faster than the former VALU form
// (one lane per output frame) - at 1024 windows per step a launch is 2-4 windows per CU and 12-30 us of latency chain
// (weights -> slab -> tile -> epilogue) of which the contraction is 1-4 us.  What counts is how many workgroups a CU
// holds: static LDS is kept to the fold tables, every launch takes the grid its own occupancy allows (g_role_grid in
// mww_lib.hip: 1.030 -> 0.884 ms per Inception step), and the depthwise ops of MixedNet graphs are register-blocked.
#pragma once
#include <type_traits>
#include "common.hip.h"
#include "kernels_fwd.hip.h"   // XGather / XShared / XStage: the stem's input straight from the feature stores

namespace mww {

constexpr int kGMaxSrc = 3;
constexpr int kGB = 4;    // rows a thread keeps in flight in the staging loops and in the data-gradient epilogue: same-session A/B of
                          // the Inception step, 2 / 4 / 8 / 12 rows = 1.112 / 1.079 / 1.108 / 1.134 ms (4 keeps the fused backward kernels at
                          // <= 128 VGPRs = four resident workgroups per CU; 32 rows: 1.65 ms)
constexpr int kGE = 4;
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
  // Planar producers (GConvArgs::out_planes): a tensor whose consumers all read equal channel slices is stored as one plane
  // per slice, [plane][B][T][C], so that a consumer reads whole rows instead of 40 bytes out of every 120 (the counters of
  // the default Inception step showed FETCH_SIZE at three times the operand bytes).  Such a source arrives with p / g and
  // the BN arrays already pointing at its plane, ld = C and c0 = 0; what still refers to the producer's channel axis -
  // the folded statistics table, the backward statistics rows - uses scb (the slice's first producer channel) and sld
  // (the producer's channel count).  Interleaved sources: scb = c0, sld = ld.
  int sld, scb;
  // residual branch added to the producer's normalised output before its activation (mixednet.py:340-358:
  // residual = BN(conv1x1(block input)); net = relu(BN(...) + StridedDrop(residual))): frame t of the producer
  // pairs with frame t + rdrop of the residual op's pre-BN tensor rp [B][rT][C]; null = none
  const float* rp;
  const float* rscale;
  const float* rshift;
  int rT, rdrop;
  StatAcc gacc;         // backward sums of this slice go to the producer's accumulator rows ([kStatRows][2][ld]) instead of gstat_part
};

// ---- compile-time shapes ---------------------------------------------------------------------------------------
// The graph kernels take every shape at run time: kernel length, dilation, channel counts, the sources' widths and row
// lengths.  Their launches turned out to be bound by instruction issue, not by memory latency (round-2 counters of the
// default Inception step: 2 800 VALU + 2 400 SALU instructions per wave around 180 MFMAs - index arithmetic, loop control,
// run-time divisions - with four waves per SIMD each active 20-30 % of its time: the SIMDs are saturated).  GShape names
// what the instantiations for a known topology (mww_lib.hip MWW_G_SHAPES: the ops of the reference's default Inception
// flags, inception.py:146-209) know at compile time: kernel length K (dilation and stride 1), the number of sources and
// each one's (width, row length of the tensor it is a channel slice of).  Everything derived from them - LDS pitches, the
// thread <-> (channel group, frame group) maps, trip counts of the MFMA loops - folds to constants, LDS reads take
// immediate offsets.  GShapeDyn (all zero) = the run-time path, for every other topology.
template <int K_, int NSRC_, int C0_, int L0_, int C1_ = 0, int L1_ = 0, int C2_ = 0, int L2_ = 0>
struct GShape {
  static constexpr int K = K_, NSRC = NSRC_, CIN = C0_ + C1_ + C2_;
  static constexpr int srcC(int i) { return i == 0 ? C0_ : (i == 1 ? C1_ : C2_); }
  static constexpr int srcLD(int i) { return i == 0 ? L0_ : (i == 1 ? L1_ : L2_); }
  static constexpr int srcC0(int i) { return i == 0 ? 0 : (i == 1 ? C0_ : C0_ + C1_); }   // first channel in the concatenation
};
typedef GShape<0, 0, 0, 0> GShapeDyn;

// ---- BN statistics hand-over (common.hip.h) in the graph kernels -------------------------------------------
// The producer of a BN'd tensor adds its per-workgroup sums to kStatRows replicated fp64 rows; the FIRST launch that
// consumes the tensor folds them in every workgroup's prologue into an LDS table (its workgroup 0 also publishes the
// folded arrays, the moving statistics and - backward - dgamma / dbeta for the later launches of the step), so no
// finalize launch stands between a producer and its consumer.  Same arithmetic as gbn_*_finalize_body.
constexpr int kGFoldC = 64;   // channels of a folded tensor (the instantiated widths end at 64)

struct GFoldFwd {        // forward statistics of one source's producer
  const double* acc;     // [kStatRows][2][C] sums of p, p^2 per channel; null: nothing to fold (the arrays are current)
  int C, groups;         // producer channels, SubSpectralNormalization slots (1 = plain BN)
  float inv_n;           // 1 / (B * T * channels per slot)
  int publish;           // workgroup 0 of this role writes the arrays below
  int update_moving;
  const float *gamma, *beta;           // [slots]
  float *moving_mean, *moving_var;     // [slots]
  float *scale, *shift, *mean, *rstd;  // [C]
};

struct GFoldBwd {        // backward statistics (sum g, sum g*xhat) of the op a backward launch works on
  const double* acc;     // [kStatRows][2][C]; null: c1 / mg / mgx are current
  int groups;
  float inv_n, dscale;
  int publish;
  const float* gamma;    // [slots]
  float *c1, *mg, *mgx;  // [C]
  float *dgamma, *dbeta; // [slots] -> flat gradient
};

// A fold runs in two halves so that its memory round trip overlaps whatever the prologue stages in between (weights):
// load() issues the loads of thread ch = tid < C (the accumulator rows of its channel's slot members, gamma / beta and what
// the publishing workgroup updates), finish() turns them into the table entries.  Plain BN (one member per slot) keeps the
// kStatRows x 2 raw sums in registers; SubSpectralNormalization sums its members in load().
struct GFoldRegs {
  double v1[kStatRows], v2[kStatRows];
  float p0, p1, m0, m1;   // forward: gamma, beta, old moving mean / variance;  backward: gamma, rstd
};

// lane `src`'s copy of v (all 64 lanes of the wave take part)
__device__ __forceinline__ double gfold_lane(double v, int src) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __shfl(u.i[0], src);
  u.i[1] = __shfl(u.i[1], src);
  return u.d;
}

// The raw sums of channel ch's slot (lane ch of its wave; lanes >= C are idle but take part in the exchange).  Every lane
// reads the kStatRows x 2 sums of its own channel - one round trip - and a SubSpectralNormalization slot's members are then
// collected from their lanes in the order the former per-member loop added them (slot, slot + groups, ...: bit-identical
// sums; as a loop of loads it was members - 1 = five more dependent round trips in front of the stem's consumers, and in a
// kernel short of registers the compiler shared one temporary between an iteration's sixteen loads: eighty round trips).
__device__ __forceinline__ void gfold_load_sums(const double* acc, int C, int groups, int ch, GFoldRegs& r) {
  const bool active = ch < C;
#pragma unroll
  for (int j = 0; j < kStatRows; ++j) {
    r.v1[j] = active ? acc[(size_t)j * 2 * C + ch] : 0.0;
    r.v2[j] = active ? acc[(size_t)j * 2 * C + C + ch] : 0.0;
  }
  if (groups > 1) {   // (uniform)
    const int members = C / groups, slot = active ? ch % groups : 0;
#pragma unroll
    for (int j = 0; j < kStatRows; ++j) {
      double a1 = gfold_lane(r.v1[j], slot), a2 = gfold_lane(r.v2[j], slot);
      for (int m = 1; m < members; ++m) {
        a1 += gfold_lane(r.v1[j], slot + m * groups);
        a2 += gfold_lane(r.v2[j], slot + m * groups);
      }
      r.v1[j] = a1;
      r.v2[j] = a2;
    }
  }
}

// (called by whole waves: gfold_load_sums exchanges between lanes)
__device__ __forceinline__ void gfold_forward_load(const GFoldFwd& f, int bid, int tid, GFoldRegs& r) {
  gfold_load_sums(f.acc, f.C, f.groups, tid, r);
  if (tid >= f.C) return;
  const int slot = f.groups > 1 ? tid % f.groups : tid, nslots = f.groups > 1 ? f.groups : f.C;
  r.p0 = f.gamma[slot];
  r.p1 = f.beta[slot];
  r.m0 = r.m1 = 0.f;
  if (f.publish && bid == 0 && f.update_moving && tid < nslots) {
    r.m0 = f.moving_mean[tid];
    r.m1 = f.moving_var[tid];
  }
}

// tab = [4][kGFoldC]: scale, shift, mean, rstd of every producer channel
__device__ __forceinline__ void gfold_forward_finish(const GFoldFwd& f, float* tab, int bid, int tid, const GFoldRegs& r) {
  if (tid >= f.C) return;
  const int ch = tid, nslots = f.groups > 1 ? f.groups : f.C;
  double t1 = 0.0, t2 = 0.0;
#pragma unroll
  for (int j = 0; j < kStatRows; ++j) {
    t1 += r.v1[j];
    t2 += r.v2[j];
  }
  const double mean = t1 * (double)f.inv_n;
  double var = t2 * (double)f.inv_n - mean * mean;   // biased batch variance
  if (var < 0.0) var = 0.0;
  const float meanf = (float)mean, varf = (float)var;
  const float rstd = 1.0f / sqrtf(varf + kBnEps);
  const float sc = r.p0 * rstd, sh = r.p1 - meanf * sc;
  tab[ch] = sc;
  tab[kGFoldC + ch] = sh;
  tab[2 * kGFoldC + ch] = meanf;
  tab[3 * kGFoldC + ch] = rstd;
  if (f.publish && bid == 0) {
    f.scale[ch] = sc;
    f.shift[ch] = sh;
    f.mean[ch] = meanf;
    f.rstd[ch] = rstd;
    if (ch < nslots && f.update_moving) {   // ch == its slot
      f.moving_mean[ch] = r.m0 * kBnMomentum + meanf * (1.0f - kBnMomentum);
      f.moving_var[ch] = r.m1 * kBnMomentum + varf * (1.0f - kBnMomentum);
    }
  }
}

// C = channels of the op; rstd = its forward statistic (already published)
__device__ __forceinline__ void gfold_backward_load(const GFoldBwd& f, int C, const float* rstd, int tid, GFoldRegs& r) {
  gfold_load_sums(f.acc, C, f.groups, tid, r);
  if (tid >= C) return;
  r.p0 = f.gamma[f.groups > 1 ? tid % f.groups : tid];
  r.p1 = rstd[tid];
}

// tab = [3][kGFoldC]: c1, mg, mgx of every channel of the op
__device__ __forceinline__ void gfold_backward_finish(const GFoldBwd& f, int C, float* tab, int bid, int tid, const GFoldRegs& r) {
  if (tid >= C) return;
  const int ch = tid, nslots = f.groups > 1 ? f.groups : C;
  double t1 = 0.0, t2 = 0.0;
#pragma unroll
  for (int j = 0; j < kStatRows; ++j) {
    t1 += r.v1[j];
    t2 += r.v2[j];
  }
  const float c1 = r.p0 * r.p1;
  const float mg = (float)(t1 * (double)f.inv_n), mgx = (float)(t2 * (double)f.inv_n);
  tab[ch] = c1;
  tab[kGFoldC + ch] = mg;
  tab[2 * kGFoldC + ch] = mgx;
  if (f.publish && bid == 0) {
    f.c1[ch] = c1;
    f.mg[ch] = mg;
    f.mgx[ch] = mgx;
    if (ch < nslots) {
      f.dbeta[ch] = (float)t1 * f.dscale;
      f.dgamma[ch] = (float)t2 * f.dscale;
    }
  }
}

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
  GFoldBwd fold;          // fold.acc set: c1 / mg / mgx come from the accumulator rows (folded in the prologue)
  int planes, pc;         // the op's own tensors g / p are planar (see GSrc): planes of pc channels, pstride floats apart
  long long pstride;
};

// ---------------------------------------------------------------------------------------------
// staging helpers
// A window's slab of a source is staged in ONE memory round trip wherever the layout allows: thread <-> (group of V
// adjacent channels, frame group) with V = 4 / 2 / 1 floats per load (V divides the slice width, its offset and the
// producer's row length, so every load is naturally aligned), kGB rows per thread and tensor in flight, every
// load of a batch issued before the first use (the round-2 profile showed these kernels to be chains of dependent round
// trips: 12-32 us per launch for a single window per workgroup).  The loads go through a buffer resource over the
// window's slab: workgroup-uniform base in scalar registers, one 32-bit offset register per load, elements past the slab
// read as 0 - no clamp, no predicate, no 64-bit address pair per load in flight.
template <int V>
struct GVec {
  float f[V];
};
template <int V, int AUX = 0>
__device__ __forceinline__ GVec<V> gvec_bload(BufRsrc r, int elem) {
  GVec<V> o;
  if constexpr (V == 4) {
    const float4 t = tile_load4<AUX>(r, elem * 4);
    o.f[0] = t.x; o.f[1] = t.y; o.f[2] = t.z; o.f[3] = t.w;
  } else if constexpr (V == 2) {
    const uint2 t = tile_load2<AUX>(r, elem * 4);
    o.f[0] = __uint_as_float(t.x); o.f[1] = __uint_as_float(t.y);
  } else {
    o.f[0] = tile_load1<AUX>(r, elem * 4);
  }
  return o;
}
__device__ __forceinline__ int gvec_width(int C, int ld, int c0) {
  return ((C | ld | c0) & 3) == 0 ? 4 : (((C | ld | c0) & 1) == 0 ? 2 : 1);
}

// one source without a residual branch; coef = [2][...] scale / shift indexed by producer channel (global or LDS)
// f0: first frame of the staged rows within the (aligned) window - frame chunks of 1x1 ops, else 0
// (SC, SLD > 0: the slice width and the producer's row length as the instantiation knows them - GShape)
template <int V, int SC = 0, int SLD = 0>
__device__ __forceinline__ void stage_source_vec(const GSrc& s, int b, int rows, float* sIn, int PI, int c0out, int tid,
                                                 const float* cscale, const float* cshift, int f0 = 0) {
  const int sC = SC > 0 ? SC : s.C, sLD = SLD > 0 ? SLD : s.ld;
  int q, rg, nrg;
  if constexpr (SC > 0) {
    constexpr int NQ = SC / V;
    nrg = kThreads / NQ;
    rg = tid / NQ;
    q = tid - rg * NQ;
  } else {
    const int NQ = sC / V;
    nrg = fast_div(kThreads, NQ);
    fast_divmod(tid, NQ, rg, q);
  }
  if (rg >= nrg) return;
  const bool ident = (s.flags & GSRC_IDENTITY) != 0;
  float sc[V], sh[V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    sc[e] = ident ? 1.f : cscale[q * V + e];   // (cscale / cshift: the slice's first channel, see stage_sources)
    sh[e] = ident ? 0.f : cshift[q * V + e];
  }
  const float lo = (s.flags & (GSRC_IDENTITY | GSRC_LINEAR)) ? -3.0e38f : 0.f;   // ReLU as a clamp from below
  const BufRsrc slab = tile_rsrc(s.p + ((size_t)b * s.T + s.toff + f0) * sLD + s.c0, ((rows - 1) * sLD + sC) * 4);
  float* dst = sIn + c0out + q * V;
  constexpr int NB = kGB;   // rows in flight (these kernels live on
                                               // the number of resident workgroups: registers are occupancy)
  for (int t0 = rg; t0 < rows; t0 += NB * nrg) {
    GVec<V> v[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) v[u] = gvec_bload<V>(slab, (t0 + u * nrg) * sLD + q * V);
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int t = t0 + u * nrg;
      if (t < rows) {
#pragma unroll
        for (int e = 0; e < V; ++e) dst[t * PI + e] = fmaxf(fmaf(v[u].f[e], sc[e], sh[e]), lo);
      }
    }
  }
}

// ---- a window's rows in registers (static shapes) ----------------------------------------------------------------
// With the widths known at compile time a thread's share of a window's slab is a fixed set of registers: ALL its rows are
// requested at once (one memory round trip per window instead of one per kGB rows), the request for the NEXT window
// goes out before this window's contraction (a backward role walks four windows per workgroup at the headline batch),
// and the first window's before the weights are staged and the statistics folded.  A window may have at most kGTmax
// frames for that (the host checks); slices whose share would take more than kGPipeRegs registers keep the direct
// staging (registers are occupancy here).
constexpr int kGTmax = 208;
constexpr int kGPipeRegs = 40;

template <int V, int C, int LD>
struct GSliceRegs {
  static constexpr int NQ = C / V, NRG = kThreads / NQ, NS = (kGTmax + NRG - 1) / NRG, REGS = NS * V;
  GVec<V> v[NS];
  // rows [0, rows) x channels [0, C) of the slab whose first element is `base` (row length LD)
  __device__ __forceinline__ void issue(const float* base, int rows, int tid) {
    const int rg = tid / NQ, q = tid - rg * NQ;
    const BufRsrc slab = tile_rsrc(base, ((rows - 1) * LD + C) * 4);
    const int e0 = rg < NRG ? rg * LD + q * V : kOobOffset / 4;
#pragma unroll
    for (int u = 0; u < NS; ++u) v[u] = gvec_bload<V>(slab, e0 + u * NRG * LD);
  }
};

// per-channel affine (+ clamp) of a source slice as this thread's channels see it
template <int V>
struct GSliceAffine {
  float sc[V], sh[V], lo;
};

template <int V, int C, int LD>
__device__ __forceinline__ void gslice_affine_load(const GSrc& s, const float* cscale, const float* cshift, int tid, GSliceAffine<V>& f) {
  constexpr int NQ = C / V;
  const int q = tid % NQ;
  const bool ident = (s.flags & GSRC_IDENTITY) != 0;
#pragma unroll
  for (int e = 0; e < V; ++e) {
    f.sc[e] = ident ? 1.f : cscale[q * V + e];
    f.sh[e] = ident ? 0.f : cshift[q * V + e];
  }
  f.lo = (s.flags & (GSRC_IDENTITY | GSRC_LINEAR)) ? -3.0e38f : 0.f;
}

template <int V, int C, int LD>
__device__ __forceinline__ void gslice_commit(const GSliceRegs<V, C, LD>& r, const GSliceAffine<V>& f, float* dst, int PI, int rows, int tid) {
  typedef GSliceRegs<V, C, LD> R;
  const int rg = tid / R::NQ, q = tid - rg * R::NQ;
  if (rg >= R::NRG) return;
#pragma unroll
  for (int u = 0; u < R::NS; ++u) {
    const int t = rg + u * R::NRG;
    if (t < rows) {
#pragma unroll
      for (int e = 0; e < V; ++e) dst[t * PI + q * V + e] = fmaxf(fmaf(r.v[u].f[e], f.sc[e], f.sh[e]), f.lo);
    }
  }
}

// the (up to three) sources of a static shape
template <class SH, int I>
struct GSrcTraits {
  static constexpr bool ON = I < SH::NSRC;
  static constexpr int C = ON ? SH::srcC(I) : 4, LD = ON ? SH::srcLD(I) : 4;
  static constexpr int V = ((C | LD) & 3) == 0 ? 4 : (((C | LD) & 1) == 0 ? 2 : 1);
  typedef GSliceRegs<V, C, LD> Regs;
  static constexpr int REGS = ON ? Regs::REGS : 0;
};
template <class SH>
struct GSrcPipe {
  typedef GSrcTraits<SH, 0> T0;
  typedef GSrcTraits<SH, 1> T1;
  typedef GSrcTraits<SH, 2> T2;
  static constexpr int REGS = T0::REGS + T1::REGS + T2::REGS;
  typename T0::Regs r0;
  typename T1::Regs r1;
  typename T2::Regs r2;
  GSliceAffine<T0::V> f0;
  GSliceAffine<T1::V> f1;
  GSliceAffine<T2::V> f2;
  // fold / ftab as in stage_sources: the affine comes from the folded LDS table where this launch folded the source
  __device__ __forceinline__ void load_affine(const GSrc* src, const GFoldFwd* fold, const float* ftab, int tid) {
    auto one = [&](auto tr, int i, auto& f) {
      typedef decltype(tr) T;
      const bool folded = fold != nullptr && ftab != nullptr && fold[i].acc != nullptr;
      const float* tab = ftab + i * 4 * kGFoldC;
      gslice_affine_load<T::V, T::C, T::LD>(src[i], folded ? tab + src[i].scb : src[i].scale + src[i].c0,
                                            folded ? tab + kGFoldC + src[i].scb : src[i].shift + src[i].c0, tid, f);
    };
    if constexpr (T0::ON) one(T0{}, 0, f0);
    if constexpr (T1::ON) one(T1{}, 1, f1);
    if constexpr (T2::ON) one(T2{}, 2, f2);
  }
  __device__ __forceinline__ void issue(const GSrc* src, int b, int rows, int tid) {
    if constexpr (T0::ON) r0.issue(src[0].p + ((size_t)b * src[0].T + src[0].toff) * T0::LD + src[0].c0, rows, tid);
    if constexpr (T1::ON) r1.issue(src[1].p + ((size_t)b * src[1].T + src[1].toff) * T1::LD + src[1].c0, rows, tid);
    if constexpr (T2::ON) r2.issue(src[2].p + ((size_t)b * src[2].T + src[2].toff) * T2::LD + src[2].c0, rows, tid);
  }
  __device__ __forceinline__ void commit(float* sIn, int PI, int rows, int tid) const {
    if constexpr (T0::ON) gslice_commit(r0, f0, sIn + SH::srcC0(0), PI, rows, tid);
    if constexpr (T1::ON) gslice_commit(r1, f1, sIn + SH::srcC0(1), PI, rows, tid);
    if constexpr (T2::ON) gslice_commit(r2, f2, sIn + SH::srcC0(2), PI, rows, tid);
  }
};

// ... and the op's own output gradient on its way to dp = BN backward (g, p of the op: two slabs of C channels)
template <int C>
struct GDpPipe {
  static constexpr int V = C % 4 == 0 ? 4 : (C % 2 == 0 ? 2 : 1);
  typedef GSliceRegs<V, C, C> Regs;
  static constexpr int REGS = 2 * Regs::REGS;
  Regs g, p;
  float mu[V], rs[V], c1[V], mg[V], mgx[V];
  __device__ __forceinline__ void load_coeffs(const GBnBwd& y, const float* btab, int tid) {
    const int q = tid % Regs::NQ;
    const bool folded = btab != nullptr && y.fold.acc != nullptr;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const int c = q * V + e;
      mu[e] = y.mean[c];
      rs[e] = y.rstd[c];
      c1[e] = folded ? btab[c] : y.c1[c];
      mg[e] = folded ? btab[kGFoldC + c] : y.mg[c];
      mgx[e] = folded ? btab[2 * kGFoldC + c] : y.mgx[c];
    }
  }
  __device__ __forceinline__ void issue(const GBnBwd& y, int b, int rows, int tid) {
    const size_t w0 = (size_t)b * rows * C;
    g.issue(y.g + w0, rows, tid);
    p.issue(y.p + w0, rows, tid);
  }
  // same arithmetic as stage_dp_vec
  __device__ __forceinline__ void commit(float* dst, int ld, int rows, int tid) const {
    const int rg = tid / Regs::NQ, q = tid - rg * Regs::NQ;
    if (rg >= Regs::NRG) return;
#pragma unroll
    for (int u = 0; u < Regs::NS; ++u) {
      const int t = rg + u * Regs::NRG;
      if (t < rows) {
#pragma unroll
        for (int e = 0; e < V; ++e) dst[t * ld + q * V + e] = c1[e] * (g.v[u].f[e] - mg[e] - (p.v[u].f[e] - mu[e]) * rs[e] * mgx[e]);
      }
    }
  }
};

// ---- the stem's input straight from the feature stores (XG instantiations) ---------------------------------------------
// The graph form of kernels_fwd.hip.h "fused_input": a static shape whose one source is the spectrogram (40 bins, no
// affine) gathers its windows from the stores while it stages them - pad / truncate offsets, uint16 scaling and the
// SpecAugment masks applied on the way into LDS, bit-identical to assemble_kernel - so the [B][T][40] float32 batch is
// neither written (the assembly launch: 14 us of the Inception step) nor read (twice: forward and weight gradient).
// The register stage is XStage (one window ahead, like GSrcPipe); the descriptors of the workgroup's windows (at most
// kXMaxSamples: the host checks the grid) sit behind the launch's dynamic LDS tiles.
constexpr int kGXRows = 200;   // frames of a gathered window at most: 8 float4 per thread in flight
template <int AUX>
using GXStage = XStage<kGXRows, FBINS + 1, AUX>;
struct GXNone {};

// xgather_setup for a role's workgroup: windows first, first + stride, ...  No barrier of its own: the window loop's first
// one stands between these writes and their readers (gx_commit; gx_window_lds for the second window on).
__device__ __forceinline__ void gx_setup(const XGather& g, XShared& sh, int nsamp, int tid, int first, int stride) {
  if (tid < nsamp) {
    const mww_window w = g.win[first + tid * stride];
    // (g lives in the kernel-argument segment: the store table is indexed in place)
    const int si = (unsigned)w.store < (unsigned)MWW_MAX_STORES ? w.store : 0;
    sh.base[tid] = g.store[si];
    sh.dtype[tid] = g.dtype[si];
    sh.src_elem[tid] = w.src_elem;
    sh.pad_rows[tid] = w.pad_rows;
    sh.copy_rows[tid] = w.copy_rows;
  }
  constexpr int WPS = kXRowWords + 2;
  if (tid < nsamp * WPS) {
    const int s = tid / WPS, word = tid - s * WPS;
    const int nm = g.ntm + g.nfm;
    const int* mk = g.masks + (size_t)(first + s * stride) * nm * 2;
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
}

// Where a window's frames come from: first byte of its first copied frame, zero frames in front, frames copied, element type.
// The first window's descriptor is read straight from HBM through the scalar unit (the address is workgroup-uniform) so that
// its rows are requested at kernel entry like a dense batch's; later windows take theirs from LDS.
struct GXWindow {
  const char* src;
  int pad, copy;
  bool u16;
};
__device__ __forceinline__ GXWindow gx_window_global(const XGather& g, int b) {
  const mww_window w = g.win[b];
  const int si = (unsigned)w.store < (unsigned)MWW_MAX_STORES ? w.store : 0;
  const void* base = g.store[si];
  const int dt = g.dtype[si];
  GXWindow x;
  x.u16 = uniform_int(dt) == MWW_DTYPE_U16;
  x.src = reinterpret_cast<const char*>(uniform_ptr(base)) + uniform_i64(w.src_elem) * (x.u16 ? 2 : 4);
  x.pad = uniform_int(w.pad_rows);
  x.copy = uniform_int(w.copy_rows);
  return x;
}
__device__ __forceinline__ GXWindow gx_window_lds(const XShared& sh, int s) {
  GXWindow x;
  x.u16 = uniform_int(sh.dtype[s]) == MWW_DTYPE_U16;
  x.src = reinterpret_cast<const char*>(uniform_ptr(sh.base[s])) + uniform_i64(sh.src_elem[s]) * (x.u16 ? 2 : 4);
  x.pad = uniform_int(sh.pad_rows[s]);
  x.copy = uniform_int(sh.copy_rows[s]);
  return x;
}
// XStage::issue for a whole window of `rows` frames (row0 = 0) described by `w`: the same loads
template <int AUX>
__device__ __forceinline__ void gx_issue(GXStage<AUX>& xs, const GXWindow& w, int rows, int tid) {
  typedef GXStage<AUX> X;
  asm volatile("" : "+v"(tid));
  const int gb = w.u16 ? 8 : 16;                       // bytes of one float4 group in the source
  const int r_lo = w.pad, r_hi = min(w.pad + w.copy, rows);
  const int nbytes = (r_hi - r_lo) * (X::QX * gb);
  const BufRsrc lo = tile_rsrc(w.src, nbytes), hi = tile_rsrc(w.src + 8, w.u16 ? 0 : nbytes - 8);
  const int off0 = tid < X::ACT ? (tid - r_lo * X::QX) * gb : kOobOffset;
#pragma unroll
  for (int j = 0; j < X::NJ; ++j) {
    const uint2 a = tile_load2<AUX>(lo, off0 + X::ACT * j * gb), c = tile_load2<AUX>(hi, off0 + X::ACT * j * gb);
    xs.pre[j] = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(c.x), __uint_as_float(c.y));
  }
}

// XStage::commit for a window of `rows` <= kGXRows frames: nothing is written behind the window (the tiles that follow it
// in LDS are live)
template <int AUX>
__device__ __forceinline__ void gx_commit(const GXStage<AUX>& xs, float* sX, const XGather& g, const XShared& sh, int s, int rows, int tid) {
  typedef GXStage<AUX> X;
  constexpr int PX = FBINS + 1;
  asm volatile("" : "+v"(tid));
  const int rq = tid / X::QX, q = tid - rq * X::QX;
  if (tid >= X::ACT) return;
  const bool u16 = uniform_int(sh.dtype[s]) == MWW_DTYPE_U16;
  const unsigned cm = (sh.colbits[s][(4 * q) >> 5] >> ((4 * q) & 31)) & 0xfu;
  unsigned rb[X::NJ];
#pragma unroll
  for (int j = 0; j < X::NJ; ++j) {
    const int t = min(rq + X::RPP * j, rows - 1);
    rb[j] = sh.rowbits[s][t >> 5] >> (t & 31);
  }
  float* dst = sX + rq * PX + q * 4;
#pragma unroll
  for (int j = 0; j < X::NJ; ++j) {
    if (rq + X::RPP * j < rows) {
      float4 v = xs.pre[j];
      if (u16) {
        const unsigned lo = __float_as_uint(v.x), hi = __float_as_uint(v.y);
        v.x = (float)(lo & 0xffffu) * 0.0390625f;   // data.py:268-269
        v.y = (float)(lo >> 16) * 0.0390625f;
        v.z = (float)(hi & 0xffffu) * 0.0390625f;
        v.w = (float)(hi >> 16) * 0.0390625f;
      }
      const unsigned m4 = (rb[j] & 1u) ? 0xfu : cm;
      v.x = (m4 & 1u) ? 0.f : v.x;
      v.y = (m4 & 2u) ? 0.f : v.y;
      v.z = (m4 & 4u) ? 0.f : v.z;
      v.w = (m4 & 8u) ? 0.f : v.w;
      float* d = dst + X::RPP * j * PX;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  }
}

// the sources of a static shape: the loop over them is unrolled, every source with its own compile-time width / row length
template <class SH, int I = 0>
__device__ __forceinline__ void stage_sources_static(const GSrc* src, int b, int rows, float* sIn, int PI, int tid, const GFoldFwd* fold,
                                                     const float* ftab, int f0) {
  if constexpr (I < SH::NSRC) {
    constexpr int C = SH::srcC(I), LD = SH::srcLD(I);
    constexpr int V = ((C | LD) & 3) == 0 ? 4 : (((C | LD) & 1) == 0 ? 2 : 1);   // (the host checks that the slice offset is a multiple of it)
    const GSrc& s = src[I];
    const bool folded = fold != nullptr && ftab != nullptr && fold[I].acc != nullptr;
    const float* tab = ftab + I * 4 * kGFoldC;
    stage_source_vec<V, C, LD>(s, b, rows, sIn, PI, SH::srcC0(I), tid, folded ? tab + s.scb : s.scale + s.c0,
                               folded ? tab + kGFoldC + s.scb : s.shift + s.c0, f0);
    stage_sources_static<SH, I + 1>(src, b, rows, sIn, PI, tid, fold, ftab, f0);
  }
}

// ftab: null, or [kGMaxSrc][4][kGFoldC] with the folded (scale, shift, ..) of the sources whose fold[i].acc is set
template <class SH = GShapeDyn>
__device__ __forceinline__ void stage_sources(const GSrc* src, int n_src, int b, int rows, float* sIn, int PI, int tid,
                                              const GFoldFwd* fold = nullptr, const float* ftab = nullptr, int f0 = 0) {
  if constexpr (SH::NSRC > 0) {
    stage_sources_static<SH>(src, b, rows, sIn, PI, tid, fold, ftab, f0);
    return;
  }
  int c0 = 0;
  for (int i = 0; i < n_src; ++i) {
    const GSrc& s = src[i];
    const bool folded = fold != nullptr && ftab != nullptr && fold[i].acc != nullptr;
    const float* tab = ftab + i * 4 * kGFoldC;
    if (!s.rp) {
      const float* cscale = folded ? tab + s.scb : s.scale + s.c0;
      const float* cshift = folded ? tab + kGFoldC + s.scb : s.shift + s.c0;
      const int V = gvec_width(s.C, s.ld, s.c0);
      if (V == 4) stage_source_vec<4>(s, b, rows, sIn, PI, c0, tid, cscale, cshift, f0);
      else if (V == 2) stage_source_vec<2>(s, b, rows, sIn, PI, c0, tid, cscale, cshift, f0);
      else stage_source_vec<1>(s, b, rows, sIn, PI, c0, tid, cscale, cshift, f0);
      c0 += s.C;
      continue;
    }
    // sources with a residual branch (MixedNet residual_connection): one channel per thread, two tensors
    const int C = s.C, nrg = kThreads / C, c = tid % C, rg = tid / C;
    if (rg < nrg) {
      const float sc = folded ? tab[s.scb + c] : s.scale[s.c0 + c];
      const float sh = folded ? tab[kGFoldC + s.scb + c] : s.shift[s.c0 + c];
      const float lo = (s.flags & GSRC_LINEAR) ? -3.0e38f : 0.f;
      const float* base = s.p + ((size_t)b * s.T + s.toff + f0) * s.ld + s.c0 + c;
      const float rsc = s.rscale[c], rsh = s.rshift[c];
      const float* rbase = s.rp + ((size_t)b * s.rT + s.toff + f0 + s.rdrop) * C + c;
      for (int t0 = rg; t0 < rows; t0 += kGB * nrg) {
        float v[kGB], rv[kGB];
#pragma unroll
        for (int u = 0; u < kGB; ++u) {
          const int t = min(t0 + u * nrg, rows - 1);
          v[u] = base[(size_t)t * s.ld];
          rv[u] = rbase[(size_t)t * C];
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

// dp = BN backward of the op's output gradient, rows [0, rows) of window b -> dst[t * ld + c]
// (f0, Ttot: rows [f0, f0 + rows) of a window of Ttot frames - frame chunks of 1x1 ops; Ttot < 0: the whole window)
// (CC > 0: the channel count as the instantiation knows it)
template <int V, int CC = 0>
__device__ __forceinline__ void stage_dp_vec(const GBnBwd& y, int C_, int b, int rows, float* dst, int ld, int tid, const float* btab,
                                             int f0 = 0, int Ttot = -1) {
  const int C = CC > 0 ? CC : C_;
  int q, rg, nrg;
  if constexpr (CC > 0) {
    constexpr int NQ = CC / V;
    nrg = kThreads / NQ;
    rg = tid / NQ;
    q = tid - rg * NQ;
  } else {
    const int NQ = C / V;
    nrg = fast_div(kThreads, NQ);
    fast_divmod(tid, NQ, rg, q);
  }
  if (rg >= nrg) return;
  const bool folded = btab != nullptr && y.fold.acc != nullptr;
  float mu[V], rs[V], c1[V], mg[V], mgx[V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const int c = q * V + e;
    mu[e] = y.mean[c];
    rs[e] = y.rstd[c];
    c1[e] = folded ? btab[c] : y.c1[c];
    mg[e] = folded ? btab[kGFoldC + c] : y.mg[c];
    mgx[e] = folded ? btab[2 * kGFoldC + c] : y.mgx[c];
  }
  const size_t w0 = ((size_t)b * (Ttot < 0 ? rows : Ttot) + f0) * C;
  // planar tensors (GSrc): this thread's channel group lies in plane (q V) / pc; its rows are pc floats long.  One resource
  // over the whole tensor then (the offsets stay below 2^31 bytes for every batch the context holds), rows past the window
  // are not committed
  // (the host makes only 30- and 48-channel tensors planar - mww_create_convnet - so the instantiations that know another
  // width carry none of this: the 16-filter twin backward launch lost 10 us to the extra scalar pressure before)
  constexpr bool kMaybePlanar = CC == 0 || CC == 30 || CC == 48;
  const bool planar = kMaybePlanar && y.planes > 1;
  const int pl = planar ? (q * V) / y.pc : 0, rowc = planar ? y.pc : C;
  const size_t wp = (size_t)pl * (size_t)y.pstride + ((size_t)b * (Ttot < 0 ? rows : Ttot) + f0) * rowc;
  const int tbytes = planar ? (int)((size_t)y.planes * (size_t)y.pstride * 4) : rows * C * 4;
  const BufRsrc gslab = tile_rsrc(planar ? y.g : y.g + w0, tbytes), pslab = tile_rsrc(planar ? y.p : y.p + w0, tbytes);
  const int e0 = planar ? (int)wp + (q * V - pl * rowc) : q * V;
  constexpr int NB = kGB;
  for (int t0 = rg; t0 < rows; t0 += NB * nrg) {
    GVec<V> g[NB], p[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int off = (t0 + u * nrg) * rowc + e0;
      g[u] = gvec_bload<V, MWW_AUX_GR_LD_DP>(gslab, off);
      p[u] = gvec_bload<V, MWW_AUX_GR_LD_DP>(pslab, off);
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int t = t0 + u * nrg;
      if (t < rows) {
#pragma unroll
        for (int e = 0; e < V; ++e) dst[t * ld + q * V + e] = c1[e] * (g[u].f[e] - mg[e] - (p[u].f[e] - mu[e]) * rs[e] * mgx[e]);
      }
    }
  }
}

// btab: [3][kGFoldC] folded (c1, mg, mgx) when y.fold.acc is set.  CW = the op's filter count when the kernel knows it at
// compile time (vector width chosen there: a run-time choice between the three instantiations makes the compiler keep all
// of them in registers at once), 0 = one channel per load.
template <int CW>
__device__ __forceinline__ void stage_dp(const GBnBwd& y, int C, int b, int rows, float* dst, int ld, int tid, const float* btab = nullptr,
                                         int f0 = 0, int Ttot = -1) {
  constexpr int V = CW == 0 ? 1 : (CW % 4 == 0 ? 4 : (CW % 2 == 0 ? 2 : 1));
  stage_dp_vec<V, CW>(y, C, b, rows, dst, ld, tid, btab, f0, Ttot);
}

// per-thread (s1, s2) of channel c = tid % C, frame group tid / C  ->  part[2][ld] of this workgroup
// (columns [0, C) of each statistic's row; ld = C unless the channels are a slice of a wider tensor)
__device__ __forceinline__ void write_channel_partials(float s1, float s2, int C, float* sRed, float* part, int tid, int ld) {
  const int nrg = fast_div(kThreads, C);
  int c, rg;
  fast_divmod(tid, C, rg, c);
  __syncthreads();
  if (rg < nrg) {
    sRed[(rg * 2 + 0) * C + c] = s1;
    sRed[(rg * 2 + 1) * C + c] = s2;
  }
  __syncthreads();
  if (tid < 2 * C) {
    float v = 0.f;
    for (int r = 0; r < nrg; ++r) v += sRed[r * 2 * C + tid];
    part[(tid >= C ? ld - C : 0) + tid] = v;   // (tid / C) * ld + tid % C for tid < 2 C
  }
}

// ... or, with acc.acc set (statistics hand-over), added to row bid % kStatRows of the accumulator rows [kStatRows][2][ld]
// at column offset c0, and the same columns of the other parity's rows cleared (bid / nb: this workgroup's index and the
// number of workgroups of its role, as in gconv_body)
__device__ __forceinline__ void publish_channel_partials(float s1, float s2, int C, float* sRed, float* part, int tid, int ld,
                                                         const StatAcc& acc, int c0, int bid, int nb) {
  if (!acc.acc) {
    write_channel_partials(s1, s2, C, sRed, part, tid, ld);
    return;
  }
  const int nrg = fast_div(kThreads, C);
  int c, rg;
  fast_divmod(tid, C, rg, c);
  __syncthreads();
  if (rg < nrg) {
    sRed[(rg * 2 + 0) * C + c] = s1;
    sRed[(rg * 2 + 1) * C + c] = s2;
  }
  __syncthreads();
  if (tid < 2 * C) {
    float v = 0.f;
    for (int r = 0; r < nrg; ++r) v += sRed[r * 2 * C + tid];
    const size_t col = (size_t)((tid >= C ? ld - C : 0) + c0 + tid);   // (tid / C) * ld + c0 + tid % C for tid < 2 C
    unsafeAtomicAdd(acc.acc + (size_t)(bid % kStatRows) * 2 * ld + col, (double)v);
    for (int r = bid; r < kStatRows; r += nb) acc.clear[(size_t)r * 2 * ld + col] = 0.0;
  }
}

// ---------------------------------------------------------------------------------------------
// MODE 0: forward convolution.  MODE 1: data gradient.
struct GConvArgs {
  GSrc src[kGMaxSrc];
  int n_src;
  const float* w;       // the op's weights [k][fwd cin][fwd cout] (MODE 0: cin = fwd cin, NC = fwd cout; MODE 1: cin = fwd cout, NC = fwd cin)
  int k, dil, cin;      // cin = channels reduced over
  int stride;           // MODE 0 only: time stride (ops fed by the spectrogram, e.g. MixedNet's first conv); else 1
  int B;
  int Tin;              // MODE 0: aligned input frames;   MODE 1: frames of the op's output (dp)
  int Tout;             // MODE 0: (Tin - (k-1)*dil - 1)/stride + 1;   MODE 1: Tin + (k-1)*dil (= aligned input frames)
  float* out;           // MODE 0: pre-BN output [B][Tout][NC]
  float* stat_part;     // MODE 0: [grid][2][NC], or null when the op has no batch statistics
  GBnBwd y;             // MODE 1
  GFoldFwd fold[kGMaxSrc];   // MODE 0: sources whose producer statistics this launch is the first to consume
  StatAcc sacc;         // MODE 0: the output statistics go to these accumulator rows instead of stat_part
  int S, Tc;            // frame chunks (the CH instantiations, 1x1 ops only): a window is S work items of Tc frames (the last one shorter)
  // MODE 0: planar output (see GSrc): channel c goes to plane c / out_pc of `out`, planes out_pstride floats apart, rows of
  // out_pc channels; out_planes <= 1: interleaved [B][Tout][NC]
  int out_planes, out_pc;
  long long out_pstride;
};

// bid / nb: this workgroup's index and the number of workgroups sharing the batch (a launch may hold several roles)
// CDP (MODE 1): the op's filter count = channels of dp, when the launch knows it at compile time
// CH: frame chunks.  A 1x1 op has no halo, so a chunk of its frames is just a shorter window: the LDS tiles shrink to Tc
// frames (a 64 -> 64 op of a MixedNet holds 105 KB for a whole window = one workgroup per CU), the work items are
// (window, chunk) pairs.  The forward convolution and the weight gradient also take k > 1 (a chunk stages its input frames
// plus halo: the 5 x 40 -> 24 stem); the data gradient only k = 1.  Separate instantiations: the whole-window kernels are
// unchanged by it.
// Data-gradient epilogue of a static shape (whole windows, no residual branches): source I's slice of the gradient tile goes
// to g (first consumer: stored, later ones: accumulated) through the ReLU mask of the source, with the slice's backward sums.
// Same arithmetic and order as the run-time loop in gconv_body; the slice width / row length are constants here.
// A thread's channel of every source is the same for all windows: the producers' folded scale / shift (ReLU mask) and
// mean / rstd (x-hat of the backward sums) are read once per workgroup.  (Inside the epilogue - where the stores of the window
// loop keep the compiler from hoisting them - they were a dependent round trip per source and window in front of the rows'
// loads.)
struct GDgradCoef {
  float sc[kGMaxSrc], sh[kGMaxSrc], mu[kGMaxSrc], rs[kGMaxSrc];
};
template <class SH, int I = 0>
__device__ __forceinline__ void dgrad_coefs_static(const GSrc* src, int tid, GDgradCoef& dc) {
  if constexpr (I < SH::NSRC) {
    constexpr int C = SH::srcC(I);
    const GSrc& s = src[I];
    const int rg = tid / C, c = tid - rg * C;
    const bool on = (s.flags & GSRC_GRAD) && rg < kThreads / C, stats = on && (s.flags & GSRC_STATS);
    dc.sc[I] = on ? s.scale[s.c0 + c] : 0.f;
    dc.sh[I] = on ? s.shift[s.c0 + c] : 0.f;
    dc.mu[I] = stats ? s.mean[s.c0 + c] : 0.f;
    dc.rs[I] = stats ? s.rstd[s.c0 + c] : 0.f;
    dgrad_coefs_static<SH, I + 1>(src, tid, dc);
  }
}

template <class SH, int I = 0>
__device__ __forceinline__ void dgrad_sources_static(const GSrc* src, int b, const float* sOut, int PO, int tid, float& s1o, float& s2o,
                                                     float* sSrcAcc, const GDgradCoef& dc) {
  if constexpr (I < SH::NSRC) {
    constexpr int C = SH::srcC(I), LD = SH::srcLD(I), c0 = SH::srcC0(I);
    constexpr int kGES = 8;   // rows in flight per thread (the run-time epilogue keeps kGE = 4: a window's share of a 10- / 16-channel
                              // slice is 8-12 rows per thread, i.e. two or three dependent round trips there)
    const GSrc& s = src[I];
    if (s.flags & GSRC_GRAD) {
      constexpr int nrg = kThreads / C;
      const int rg = tid / C, c = tid - rg * C;
      if (rg < nrg) {
        const float sc = dc.sc[I], sh = dc.sh[I], mu = dc.mu[I], rs = dc.rs[I];
        const bool accum = (s.flags & GSRC_ACCUM) != 0;
        const bool linear = (s.flags & GSRC_LINEAR) != 0;
        float t1 = 0.f, t2 = 0.f;
        const size_t woff = (size_t)b * s.T * LD + s.c0;
        const int wbytes = ((s.T - 1) * LD + C) * 4;
        const BufRsrc pr = tile_rsrc(s.p + woff, wbytes), gr = tile_rsrc(s.g + woff, wbytes), go = tile_rsrc(s.g + woff, accum ? wbytes : 0);
        for (int tb = rg; tb < s.T; tb += kGES * nrg) {
          float pv[kGES], gold[kGES];
#pragma unroll
          for (int u = 0; u < kGES; ++u) {
            const int off = ((tb + u * nrg) * LD + c) * 4;
            pv[u] = tile_load1(pr, off);
            gold[u] = tile_load1<MWW_AUX_GR_LD_GOLD>(go, off);
          }
#pragma unroll
          for (int u = 0; u < kGES; ++u) {
            const int t = tb + u * nrg;
            if (t < s.T) {
              const float p = pv[u];
              const int r = t - s.toff;   // row of the output tile
              const float gv = ((r >= 0 && (linear || fmaf(p, sc, sh) > 0.f)) ? sOut[r * PO + c0 + c] : 0.f) + gold[u];
              tile_store1<MWW_AUX_GR_ST_G>(gr, (t * LD + c) * 4, gv);
              t1 += gv;
              t2 = fmaf(gv, (p - mu) * rs, t2);
            }
          }
        }
        if constexpr (I == 0) {
          s1o += t1;
          s2o += t2;
        } else {
          sSrcAcc[(I * 2 - 2) * kThreads + tid] += t1;
          sSrcAcc[(I * 2 - 1) * kThreads + tid] += t2;
        }
      }
    }
    dgrad_sources_static<SH, I + 1>(src, b, sOut, PO, tid, s1o, s2o, sSrcAcc, dc);
  }
}

// which forward instantiations store from the accumulators (see gconv_body "DIRECT"; the host sizes LDS with g_lds_fwd_direct)
template <class SH, int MODE>
__host__ __device__ constexpr bool g_fwd_direct() {
  return SH::NSRC > 0 && MODE == 0 && (MWW_G_FWD_DIRECT >= 2 || (MWW_G_FWD_DIRECT == 1 && SH::NSRC == 1 && SH::CIN == FBINS));
}
// floats of the direct form's tiles: weights [k * cin4][NCW] + 8 (the wrap of the last row's last fragment) + the window
__host__ __device__ constexpr int g_direct_tiles(int k, int cin, int nc, int rows_in) {
  return k * ((cin + 3) / 4 * 4) * ((nc + 7) / 8 * 8) + 8 + rows_in * (cin | 1);
}

// SH: the op's shape where the instantiation knows it (GShape; static shapes have dilation = stride = 1 and whole windows)
// XG (+ xgp): the op's one source is the spectrogram, gathered from the feature stores (see "the stem's input" above)
template <int NC, int MODE, int CDP = 0, bool CH = false, class SH = GShapeDyn, bool XG = false>
__device__ __forceinline__ void gconv_body(const GConvArgs& a, const int bid, const int nb, const XGather* xgp = nullptr) {
  constexpr bool ST = SH::NSRC > 0;
  static_assert(!ST || !CH, "static shapes run whole windows");
  static_assert(!ST || MODE == 0 || CDP > 0, "the data gradient of a static shape knows its filter count");
  static_assert(!XG || (ST && MODE == 0 && SH::NSRC == 1 && SH::CIN == FBINS), "gathered input: the forward convolution of a static shape over the 40 bins");
  // kernel length, dilation, stride, channels reduced over, number of sources: constants of a static shape
  const int kK = ST ? SH::K : a.k, kDil = ST ? 1 : a.dil, kStride = ST ? 1 : a.stride;
  const int kCin = ST ? (MODE == 0 ? SH::CIN : CDP) : a.cin;
  const int kNsrc = ST ? SH::NSRC : a.n_src;
  // static shapes: the window's rows travel in registers, one window ahead (GSrcPipe / GDpPipe), if they fit
  typedef typename std::conditional<ST, SH, GShape<1, 1, 4, 4> >::type SHX;
  typedef GSrcPipe<SHX> SrcPipe;
  typedef GDpPipe<(CDP > 0 ? CDP : 4)> DpPipe;
  constexpr bool PIPE = ST && (XG || (MODE == 0 ? SrcPipe::REGS <= kGPipeRegs : DpPipe::REGS <= kGPipeRegs));
  SrcPipe spipe;
  DpPipe dpipe;
  typename std::conditional<XG, GXStage<MWW_AUX_LD_XF>, GXNone>::type xs;
  (void)xs;
  if constexpr (XG) {
    if (bid < a.B) gx_issue(xs, gx_window_global(*xgp, bid), a.Tin, (int)threadIdx.x);
  } else if constexpr (PIPE) {
    if (bid < a.B) {
      if constexpr (MODE == 0) spipe.issue(a.src, bid, a.Tin, (int)threadIdx.x);
      else dpipe.issue(a.y, bid, a.Tin, (int)threadIdx.x);
    }
  }
  HIP_DYNAMIC_SHARED(float4, g_smem4)
  float* g_smem = reinterpret_cast<float*>(g_smem4);
  // The convolution runs on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32): per tap j a [16 frames] x [4 channels]
  // A tile of the window against a [4 channels] x [16 filters] B tile of the weights.  Weight rows are padded to whole
  // k-steps (cin4) and whole filter tiles (NCW) with zeros, so the inner loop carries no predicate on B.
  // DIRECT (static forward convolutions, MWW_G_FWD_DIRECT): no output tile - a wave stores its accumulator tiles itself and keeps
  // the BN sums of their columns; the weight rows shrink to whole 8-filter groups (a B fragment of the last filter tile then
  // wraps into the next weight row: those products only reach columns that are neither stored nor summed).  For the stem
  // (5 x 40 -> 24) that is 19 + 32 KB instead of 25 + 32 + 19: three workgroups per CU instead of two.
  constexpr bool DIRECT = g_fwd_direct<SH, MODE>();
  // BREG (the gathering stem, MWW_G_STEM_BREG): the B fragments of every (tap, k-step, filter tile) stay in registers for the
  // launch (100 for 5 x 40 x 24) - no weight tile in LDS, half the LDS reads of the contraction; two waves per SIMD.
  constexpr bool BREG = DIRECT && XG && MWW_G_STEM_BREG != 0;
  constexpr int NT = (NC + 15) / 16, NCW = DIRECT ? (NC + 7) / 8 * 8 : NT * 16;
  const int tid = threadIdx.x;
  const int PI = kCin | 1, PO = DIRECT ? 0 : (NC | 1);
  const int cin4 = (kCin + 3) & ~3;
  const int pad = MODE == 1 ? (kK - 1) * kDil : 0;
  // rows of the input tile / of the output tile.  CH: a chunk of Tc output frames; the forward convolution stages the
  // chunk's input frames with their halo, the data gradient is only chunked for k = 1 (no halo, no zero frames)
  const int rows_in = CH ? (MODE == 0 ? (a.Tc - 1) * kStride + (kK - 1) * kDil + 1 : a.Tc) : a.Tin + 2 * pad;
  const int rows_o = CH ? a.Tc : a.Tout;
  float* sW = g_smem;                       // [k][cin4][NCW], loaded once per workgroup
  float* sIn = sW + kK * cin4 * NCW + (DIRECT ? 8 : 0);
  float* sOut = sIn + rows_in * PI;
  const int tiles_f = kK * cin4 * NCW + (DIRECT ? 8 : 0) + rows_in * PI + rows_o * PO;   // floats of the launch's tiles
  // running (sum, sum of squares) of this thread's channel: MODE 0 of the output, MODE 1 one pair per source - the first
  // source's in registers, the others' in LDS so that the sources can be walked by a real loop (their descriptors stay
  // in the kernel-argument segment instead of pinning ~100 SGPRs).  Static LDS is occupancy here: the windows of the
  // 48-channel ops take 45-47 KB and three workgroups per CU need <= 53 760 B each (1280-byte granules), so the pairs
  // live behind the tiles only for the sources that exist and the reduction scratch of the final publish reuses the
  // weight tile (the host sizes the dynamic segment to match: mww_lib.hip, lds_fwd / lds_dx).
  float s1o = 0.f, s2o = 0.f;
  float* sSrcAcc = g_smem + max(tiles_f, 2 * kThreads);   // [(n_src - 1) * 2][kThreads]
  float* sRed = g_smem;   // [2 * kThreads], after the window loop
  if (MODE == 1)
    for (int i = 0; i < (kNsrc - 1) * 2; ++i) sSrcAcc[i * kThreads + tid] = 0.f;

  // statistics hand-over: fold what this launch is the first to consume.  The loads go out before the weights are staged,
  // the table is written after (visible to the other waves after the loop's first barrier).
  // (the direct form counts its LDS in single KB: one table per source it has, none for the gathered spectrogram)
  constexpr int kFoldTabs = DIRECT ? (XG ? 0 : SH::NSRC) : kGMaxSrc;
  __shared__ float sFold[MODE == 0 ? (kFoldTabs > 0 ? kFoldTabs * 4 * kGFoldC : 4) : 3 * kGFoldC];
  // (wave i folds source i: one set of fold registers per thread)
  GFoldRegs fr;
  const int fsrc = tid >> 6, ftid = tid & 63;
  static_assert(kGFoldC <= 64 && kGMaxSrc <= kThreads / 64, "one wave folds one source");
  if (MODE == 0) {
    if constexpr (!XG)
      if (fsrc < kNsrc && a.fold[fsrc].acc) gfold_forward_load(a.fold[fsrc], bid, ftid, fr);
  } else if (a.y.fold.acc) {
    gfold_backward_load(a.y.fold, kCin, a.y.rstd, tid, fr);
  }
  // gathered input: the descriptors and mask bitmaps of this workgroup's windows (behind the launch's tiles)
  XShared* sXg = nullptr;
  if constexpr (XG) {
    sXg = reinterpret_cast<XShared*>(g_smem + ((max(tiles_f, 2 * kThreads) + 1) & ~1));
    gx_setup(*xgp, *sXg, bid < a.B ? (a.B - bid + nb - 1) / nb : 0, tid, bid, nb);
  }
  constexpr int kBK = BREG ? SH::K : 1, kBS = BREG ? (SH::CIN + 3) / 4 : 1, kBN = BREG ? (NC + 15) / 16 : 1;
  float breg[kBK][kBS][kBN];
  (void)breg;
  if constexpr (BREG) {
    const int lane_ = tid & 63, r16_ = lane_ & 15, g_ = lane_ >> 4;
#pragma unroll
    for (int j = 0; j < kBK; ++j)
#pragma unroll
      for (int ks = 0; ks < kBS; ++ks)
#pragma unroll
        for (int nt = 0; nt < kBN; ++nt) {
          const int ci = ks * 4 + g_, co = nt * 16 + r16_;
          const bool real = ci < SH::CIN && co < NC;
          const float v = a.w[real ? (j * SH::CIN + ci) * NC + co : 0];
          breg[j][ks][nt] = real ? v : 0.f;
        }
  } else
  {
    // (four elements per thread in flight: a rolled load -> LDS-write loop is one memory round trip per element, 25 of
    // them in a row for the 5 x 40 x 24 stem; most ops have two elements per thread, and every predicated-off slot of a
    // wider batch still pays its index arithmetic: 16 / 8 / 4 / 2 / 1 per batch = 1.090 / 1.074 / 1.064 / 1.068 / 1.065 ms
    // per Inception step in same-session A/B)
    // A static shape knows its element count: one batch holds them all (25 per thread for the stem - one round trip where
    // batches of four were seven in a row, the longest chain of its prologue; 2-5 for the other ops, without idle slots).
    constexpr int kWStatic = ST ? (SH::K * (((MODE == 0 ? SH::CIN : CDP) + 3) / 4 * 4) * NCW + kThreads - 1) / kThreads : 0;
    constexpr int kWB = (ST && kWStatic <= 28) ? kWStatic : 4;
    const int nw = kK * cin4 * NCW, nreal = kK * kCin * NC;
    for (int i0 = tid; i0 < nw; i0 += kWB * kThreads) {
      float wv[kWB];
#pragma unroll
      for (int u = 0; u < kWB; ++u) {
        const int i = i0 + u * kThreads;
        const int co = i % NCW, rest = i / NCW;   // (compile-time divisor)
        int ci, j;
        fast_divmod(rest, cin4, j, ci);
        const bool real = i < nw && co < NC && ci < kCin;
        // MODE 1 reads the forward weights [k][fwd cin = NC][fwd cout = cin] as its transposed, tap-reversed operand in
        // place (a separate transpose launch per step used to prepare a copy)
        const int src = min(MODE == 0 ? (j * kCin + ci) * NC + co : ((kK - 1 - j) * NC + co) * kCin + ci, nreal - 1);
        wv[u] = a.w[real ? src : 0];
        if (!real) wv[u] = 0.f;
      }
#pragma unroll
      for (int u = 0; u < kWB; ++u) {
        const int i = i0 + u * kThreads;
        if (i < nw) sW[i] = wv[u];
      }
    }
    if (DIRECT && tid < 8) sW[nw + tid] = 0.f;
  }
  if (MODE == 0) {
    if constexpr (!XG)
      if (fsrc < kNsrc && a.fold[fsrc].acc) gfold_forward_finish(a.fold[fsrc], sFold + fsrc * 4 * kGFoldC, bid, ftid, fr);
  } else if (a.y.fold.acc) {
    gfold_backward_finish(a.y.fold, kCin, sFold, bid, tid, fr);
  }
  if (MODE == 1) {
    // the zero frames around dp are written once: staging only touches the middle
    for (int i = tid; i < pad * PI; i += kThreads) {
      sIn[i] = 0.f;
      sIn[(pad + a.Tin) * PI + i] = 0.f;
    }
  }
  if constexpr (PIPE && !XG) {
    __syncthreads();   // the folded table is complete
    if constexpr (MODE == 0) spipe.load_affine(a.src, a.fold, sFold, tid);
    else dpipe.load_coeffs(a.y, sFold, tid);
  }
  int xsamp = 0;   // (XG) the window's slot in sXg
  // DIRECT: lane (r16, g) owns column nt * 16 + r16 of its waves' accumulator tiles: where that column starts in the output
  // tensor (plane and offset inside the plane's rows), and its running sums
  float* dcol[DIRECT ? NT : 1];
  float s1w[DIRECT ? NT : 1], s2w[DIRECT ? NT : 1];
  const int drowc = (DIRECT && a.out_planes > 1) ? a.out_pc : NC;
  if constexpr (DIRECT) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = min(nt * 16 + (tid & 15), NC - 1);
      const int pl = a.out_planes > 1 ? col / a.out_pc : 0;
      dcol[nt] = a.out + (size_t)pl * a.out_pstride + (col - pl * drowc);
      s1w[nt] = s2w[nt] = 0.f;
    }
  }
  (void)dcol; (void)s1w; (void)s2w; (void)drowc;
  GDgradCoef dcoef;
  if constexpr (ST && MODE == 1) dgrad_coefs_static<SHX>(a.src, tid, dcoef);
  (void)dcoef;
  for (int v = bid; v < (CH ? a.B * a.S : a.B); v += nb) {
    // work item v = window b, frames [f0, f0 + Tout) of its Ttot (whole-window kernels: f0 = 0, Tout = Ttot)
    int b = v, f0 = 0, chunk = 0;
    const int Ttot = MODE == 0 ? a.Tout : a.Tin;
    int Tin = a.Tin, Tout = a.Tout;
    if (CH) {
      fast_divmod(v, a.S, b, chunk);
      f0 = chunk * a.Tc;
      Tout = min(a.Tc, Ttot - f0);
      Tin = MODE == 0 ? (Tout - 1) * kStride + (kK - 1) * kDil + 1 : Tout;
    }
    __syncthreads();   // the previous window's epilogue is done with sOut / the conv with sIn
    if constexpr (XG) {
      gx_commit(xs, sIn, *xgp, *sXg, xsamp, a.Tin, tid);
    } else if constexpr (PIPE) {
      if constexpr (MODE == 0) spipe.commit(sIn, PI, a.Tin, tid);
      else dpipe.commit(sIn + pad * PI, PI, a.Tin, tid);
    } else {
      if (MODE == 0) stage_sources<SH>(a.src, kNsrc, b, Tin, sIn, PI, tid, a.fold, sFold, CH ? f0 * kStride : 0);
      else if (CH) stage_dp<CDP>(a.y, kCin, b, Tin, sIn, PI, tid, sFold, f0, Ttot);
      else stage_dp<CDP>(a.y, kCin, b, a.Tin, sIn + pad * PI, PI, tid, sFold);
    }
    __syncthreads();
    if constexpr (XG) {
      ++xsamp;
      if (v + nb < a.B) gx_issue(xs, gx_window_lds(*sXg, xsamp), a.Tin, tid);
    } else if constexpr (PIPE) {
      if (v + nb < a.B) {   // the next window's rows travel while this one is contracted and written out
        if constexpr (MODE == 0) spipe.issue(a.src, v + nb, a.Tin, tid);
        else dpipe.issue(a.y, v + nb, a.Tin, tid);
      }
    }
    {
      const int lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
      const int ntile = (Tout + 15) >> 4;
      const bool kfull = (kCin & 3) == 0;
      // (static forward convolutions: at most kGTmax / 16 = 13 tiles, i.e. four per wave, written as four guarded bodies - as
      // a loop the compiler put an s_waitcnt vmcnt(0) in front of it, which made every wave wait for the NEXT window's rows,
      // requested a few instructions earlier, before it contracted the current one: -0.5 .. -1.3 us per forward launch.  The
      // data-gradient role keeps the loop: the bodies cost its launches 12-16 VGPRs, and the 16-filter twin backward launch,
      // which sits at the 128-register line of four workgroups per CU, went 40 -> 47 us with them.)
      constexpr int kRtMax = (ST && MODE == 0) ? (kGTmax / 16 + kThreads / 64 - 1) / (kThreads / 64) : 0;
      auto row_tile = [&](const int rt) {
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = zero4();
        // A: lane (r16, g) holds frame rt*16 + r16 (frames past the window repeat its last one: their rows are not
        // stored), channel ci0 + g;  B: lane holds W[ci0 + g][nt*16 + r16]
        const int t = min(rt * 16 + r16, Tout - 1);
        const float* arow = sIn + (MODE == 0 ? t * kStride : t) * PI + g;
        const float* wrow = sW + g * NCW + r16;
        if constexpr (ST) {
          // constant trip counts: every LDS read is one instruction with an immediate offset off two base registers
#pragma unroll
          for (int j = 0; j < SH::K; ++j) {
#pragma unroll
            for (int ci0 = 0; ci0 < ((MODE == 0 ? SH::CIN : CDP) + 3) / 4 * 4; ci0 += 4) {
              float av = arow[j * PI + ci0];
              if (!kfull && ci0 + 4 > kCin && ci0 + g >= kCin) av = 0.f;   // (only the last k-step carries the select)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                if constexpr (BREG) acc[nt] = mfma4(av, breg[j][ci0 / 4][nt], acc[nt]);
                else acc[nt] = mfma4(av, wrow[(j * cin4 + ci0) * NCW + nt * 16], acc[nt]);
              }
            }
          }
        } else {
        for (int j = 0; j < kK; ++j) {
          const float* ar = arow + j * kDil * PI;
          const float* wr = wrow + j * cin4 * NCW;
          for (int ci0 = 0; ci0 < cin4; ci0 += 4) {
            float av = ar[ci0];
            if (!kfull && ci0 + g >= kCin) av = 0.f;   // the last k-step of a channel count that is no multiple of 4
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma4(av, wr[ci0 * NCW + nt * 16], acc[nt]);
          }
        }
        }
        if constexpr (DIRECT) {
          // D: lane (r16, g) holds rows rt*16 + g*4 + r of column nt*16 + r16: four row segments of 16 floats per store
          const size_t wrow = (size_t)b * Ttot + rt * 16 + g * 4;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const bool colok = (NC % 16 == 0) || nt < NT - 1 || r16 < NC - (NT - 1) * 16;
            float* d = dcol[nt] + wrow * drowc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const bool ok = colok && rt * 16 + g * 4 + r < Tout;
              const float v = ok ? acc[nt][r] : 0.f;
              if (ok) store_stream<MWW_AUX_GR_ST_P>(d + r * drowc, v);
              s1w[nt] += v;
              s2w[nt] = fmaf(v, v, s2w[nt]);
            }
          }
        } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = rt * 16 + g * 4 + r, col = nt * 16 + r16;
            if (row < Tout && col < NC) sOut[row * PO + col] = acc[nt][r];
          }
        }
      };
      if constexpr (kRtMax > 0) {
#pragma unroll
        for (int i = 0; i < kRtMax; ++i)
          if (wave + i * (kThreads / 64) < ntile) row_tile(wave + i * (kThreads / 64));
      } else {
        for (int rt = wave; rt < ntile; rt += kThreads / 64) row_tile(rt);
      }
    }
    if constexpr (!DIRECT) __syncthreads();
    if constexpr (DIRECT) {
      // (nothing: the waves stored their tiles)
    } else if (MODE == 0) {
      const int nrg = kThreads / NC, c = tid % NC, rg = tid / NC;
      if (rg < nrg) {
        const bool planar = a.out_planes > 1;
        const int pl = planar ? c / a.out_pc : 0, rowc = planar ? a.out_pc : NC;
        float* dst = a.out + (size_t)pl * a.out_pstride + ((size_t)b * Ttot + f0) * rowc + (c - pl * rowc);
        for (int t = rg; t < Tout; t += nrg) {
          const float v = sOut[t * PO + c];
          store_stream<MWW_AUX_GR_ST_P>(dst + (size_t)t * rowc, v);
          s1o += v;
          s2o = fmaf(v, v, s2o);
        }
      }
    } else if constexpr (ST) {
      dgrad_sources_static<SH>(a.src, b, sOut, PO, tid, s1o, s2o, sSrcAcc, dcoef);
    } else {
      int c0 = 0;
      for (int i = 0; i < kNsrc; ++i) {
        const GSrc& s = a.src[i];
        const int C = s.C;
        if (s.flags & GSRC_GRAD) {
          const int nrg = fast_div(kThreads, C);
          int c, rg;
          fast_divmod(tid, C, rg, c);
          if (rg < nrg) {
            const float sc = s.scale[s.c0 + c], sh = s.shift[s.c0 + c];
            const bool accum = (s.flags & GSRC_ACCUM) != 0, stats = (s.flags & GSRC_STATS) != 0;
            const bool linear = (s.flags & GSRC_LINEAR) != 0;
            const float mu = stats ? s.mean[s.c0 + c] : 0.f, rs = stats ? s.rstd[s.c0 + c] : 0.f;
            float t1 = 0.f, t2 = 0.f;
            if (!s.rp) {
              // the slice's rows of p and g through buffer resources over the window's slab: 32-bit lane offsets, no
              // predicate around a load (rows past the window read 0 and their stores are dropped; a first consumer reads
              // its "old" gradient through an empty resource = 0)
              const size_t woff = (size_t)b * s.T * s.ld + s.c0;
              const int wbytes = ((s.T - 1) * s.ld + C) * 4;
              const BufRsrc pr = tile_rsrc(s.p + woff, wbytes), gr = tile_rsrc(s.g + woff, wbytes), go = tile_rsrc(s.g + woff, accum ? wbytes : 0);
              // rows of the source this work item writes: all of them, or (CH) the chunk's - the first chunk also takes the
              // rows in front of the aligned input, the last one those behind it
              const int tlo = (CH && chunk > 0) ? s.toff + f0 : 0, thi = (CH && chunk < a.S - 1) ? s.toff + f0 + Tout : s.T;
              for (int tb = tlo + rg; tb < thi; tb += kGE * nrg) {
                float pv[kGE], gold[kGE];
#pragma unroll
                for (int u = 0; u < kGE; ++u) {
                  const int off = ((tb + u * nrg) * s.ld + c) * 4;
                  pv[u] = tile_load1(pr, off);
                  gold[u] = tile_load1<MWW_AUX_GR_LD_GOLD>(go, off);
                }
#pragma unroll
                for (int u = 0; u < kGE; ++u) {
                  const int t = tb + u * nrg;
                  if (t < thi) {
                    const float p = pv[u];
                    const int r = t - s.toff - f0;   // row of the output tile
                    const float gv = ((r >= 0 && (!CH || r < Tout) && (linear || fmaf(p, sc, sh) > 0.f)) ? sOut[r * PO + c0 + c] : 0.f) + gold[u];
                    tile_store1<MWW_AUX_GR_ST_G>(gr, (t * s.ld + c) * 4, gv);
                    t1 += gv;
                    t2 = fmaf(gv, (p - mu) * rs, t2);
                  }
                }
              }
            } else {
              // producer with a residual branch (MixedNet residual_connection): two tensors decide the ReLU mask
              // (whole windows only: the host never chunks a graph with residual branches)
              const size_t base = (size_t)b * s.T * s.ld + s.c0 + c;
              const float rsc = s.rscale[c], rsh = s.rshift[c];
              const float* rbase = s.rp + ((size_t)b * s.rT + s.rdrop) * C + c;
              for (int tb = rg; tb < s.T; tb += kGE * nrg) {
                float pv[kGE], rvv[kGE], gold[kGE];
#pragma unroll
                for (int u = 0; u < kGE; ++u) {
                  const int t = min(tb + u * nrg, s.T - 1);
                  const size_t idx = base + (size_t)t * s.ld;
                  pv[u] = s.p[idx];
                  rvv[u] = rbase[(size_t)t * C];
                  gold[u] = accum ? s.g[idx] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < kGE; ++u) {
                  const int t = tb + u * nrg;
                  if (t < s.T) {
                    const size_t idx = base + (size_t)t * s.ld;
                    const float p = pv[u];
                    const int r = t - s.toff;
                    float gv = (r >= 0 && (linear || src_affine(s, p, sc, sh, rvv[u], rsc, rsh) > 0.f)) ? sOut[r * PO + c0 + c] : 0.f;
                    gv += gold[u];
                    s.g[idx] = gv;
                    t1 += gv;
                    t2 = fmaf(gv, (p - mu) * rs, t2);
                  }
                }
              }
            }
            if (i == 0) {
              s1o += t1;
              s2o += t2;
            } else {
              sSrcAcc[(i * 2 - 2) * kThreads + tid] += t1;
              sSrcAcc[(i * 2 - 1) * kThreads + tid] += t2;
            }
          }
        }
        c0 += C;
      }
    }
  }
  if constexpr (DIRECT) {
    // the waves' column sums -> thread c < NC (the layout publish_channel_partials takes: channel tid % NC, row group tid / NC)
    __shared__ float sWS[(kThreads / 64) * 2 * NT * 16];
    if (a.stat_part) {   // (uniform)
      const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float t1 = sum_over_groups(s1w[nt]), t2 = sum_over_groups(s2w[nt]);
        if (lane < 16) {
          sWS[(wave * 2 + 0) * NT * 16 + nt * 16 + lane] = t1;
          sWS[(wave * 2 + 1) * NT * 16 + nt * 16 + lane] = t2;
        }
      }
      __syncthreads();
      if (tid < NC) {
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) {
          s1o += sWS[(w * 2 + 0) * NT * 16 + tid];
          s2o += sWS[(w * 2 + 1) * NT * 16 + tid];
        }
      }
    }
  }
  if (MODE == 0) {
    if (a.stat_part) publish_channel_partials(s1o, s2o, NC, sRed, a.stat_part + (size_t)bid * 2 * NC, tid, NC, a.sacc, 0, bid, nb);
  } else {
    for (int i = 0; i < kNsrc; ++i) {
      const GSrc& s = a.src[i];
      if ((s.flags & GSRC_GRAD) && (s.flags & GSRC_STATS))
        publish_channel_partials(i == 0 ? s1o : sSrcAcc[(i * 2 - 2) * kThreads + tid], i == 0 ? s2o : sSrcAcc[(i * 2 - 1) * kThreads + tid],
                                 s.C, sRed, s.gstat_part + (size_t)bid * 2 * s.sld + s.scb, tid, s.sld, s.gacc, s.scb, bid, nb);
    }
  }
}

template <int NC, int MODE, class SH = GShapeDyn>
__global__ __launch_bounds__(kThreads) void gconv_kernel(GConvArgs a) {
  stagger_start<MWW_STAGGER_GRAPH>();
  gconv_body<NC, MODE, 0, false, SH>(a, blockIdx.x, gridDim.x);
}
template <int NC, class SH>
__global__ __launch_bounds__(kThreads) void gconv_xg_kernel(GConvArgs a, XGather xg) {
  stagger_start<MWW_STAGGER_GRAPH>();
  gconv_body<NC, 0, 0, false, SH, true>(a, blockIdx.x, gridDim.x, &xg);
}
template <int NC, int MODE>
__global__ __launch_bounds__(kThreads) void gconv_chunk_kernel(GConvArgs a) {
  gconv_body<NC, MODE, 0, true>(a, blockIdx.x, gridDim.x);
}

// Two independent ops of the same shape ("twins": Inception's k x 1 convs of branch 2 and branch 3) as the two
// halves of one launch: workgroups [0, nb) run op a0, [nb, 2 nb) op a1.
// (the twin's descriptors are picked by a run-time index into the kernel-argument segment: one copy of the body's code)
struct GConv2Args { GConvArgs op[2]; };
template <int NC, class SH = GShapeDyn>
__global__ __launch_bounds__(kThreads) void gconv_fwd2_kernel(GConv2Args a, int nb) {
  stagger_start<MWW_STAGGER_GRAPH>();
  const int op = (int)blockIdx.x >= nb ? 1 : 0;
  gconv_body<NC, 0, 0, false, SH>(a.op[op], blockIdx.x - op * nb, nb);
}

// ---------------------------------------------------------------------------------------------
// weight gradient: dW[j][ci][co] = sum_{b,t} act[b][t + j*dil][ci] * dp[b][t][co]
struct GWgradArgs {
  GSrc src[kGMaxSrc];
  int n_src;
  GBnBwd y;
  int k, dil, cin, B, Tin, Tout;
  int stride;           // time stride of the forward convolution
  float* grad_part;     // [workgroups of the role][k*cin*NC]
  int S, Tc;            // frame chunks (CH instantiations, 1x1 ops): see GConvArgs
};

// On the matrix cores: dW = A^T B with A[t][m] = act[t*stride + j*dil][ci] (m = j*cin + ci, the "task" axis) and
// B[t][co] = dp[t][co]; the contraction runs over the frames of a window (k-steps of 4 frames) and over the workgroup's
// windows.  16-task tiles are dealt to the waves (every wave keeps all NT filter tiles of its task tiles in accumulators
// for the whole launch); ops with fewer than three task tiles also split the frames over the waves (kparts) and sum the
// parts through LDS at the end.  One partial row [k*cin][NC] per workgroup.
constexpr int kGWgTilesPerWave = 4;   // 16 task tiles (k * cin <= 256) over four waves

__host__ __device__ constexpr int gwg_kparts(int tasks) { return tasks > 32 ? 1 : (tasks > 16 ? 2 : 4); }
// row pitch of the staged dp: the filter tiles, unpadded.  (A pitch of 16 mod 32 keeps the four k-rows of a B fragment on
// disjoint banks, but for 17-32 filters it costs 16 floats per frame = 12 KB of a 190-frame window - the difference between
// two and three resident workgroups for the 24 -> 30 op; same-session A/B of the Inception step: 0.943 against 0.949 ms.
// Going further - 8-float granularity, B fragments of the last tile wrapping into the next row, whose products only reach
// columns that are never stored - puts the stem's weight gradient at three per CU too, but measured 0.891 against 0.887.)
__host__ __device__ inline int gwg_dp_pitch(int nc) { return (nc + 15) / 16 * 16; }

template <int NC, bool CH = false, class SH = GShapeDyn, bool XG = false>
__device__ __forceinline__ void gconv_wgrad_body(const GWgradArgs& a, const int bid, const int nb, const XGather* xgp = nullptr) {
  constexpr bool ST = SH::NSRC > 0;
  static_assert(!ST || !CH, "static shapes run whole windows");
  static_assert(!XG || (ST && SH::NSRC == 1 && SH::CIN == FBINS), "gathered input: a static shape over the 40 bins");
  const int kK = ST ? SH::K : a.k, kDil = ST ? 1 : a.dil, kStride = ST ? 1 : a.stride, kCin = ST ? SH::CIN : a.cin;
  const int kNsrc = ST ? SH::NSRC : a.n_src;
  // static shapes: the window's rows travel in registers, one window ahead (see gconv_body)
  typedef typename std::conditional<ST, SH, GShape<1, 1, 4, 4> >::type SHX;
  typedef GSrcPipe<SHX> SrcPipe;
  typedef GDpPipe<NC> DpPipe;
  constexpr bool PIPE_S = ST && (XG || SrcPipe::REGS <= kGPipeRegs);
  // (the 16-filter backward kernels passed 128 registers with the dp pipeline: three workgroups per CU instead of four; the
  // gathering stem is a launch of its own whose 56 KB of tiles allow two per CU whatever it keeps in registers)
  constexpr bool PIPE_D = PIPE_S && (XG || SrcPipe::REGS + DpPipe::REGS <= 32);
  SrcPipe spipe;
  DpPipe dpipe;
  typename std::conditional<XG, GXStage<MWW_AUX_LD_XB>, GXNone>::type xs;
  (void)xs;
  if constexpr (XG) {
    if (bid < a.B) {
      gx_issue(xs, gx_window_global(*xgp, bid), a.Tin, (int)threadIdx.x);
      dpipe.issue(a.y, bid, a.Tout, (int)threadIdx.x);
    }
  } else if constexpr (PIPE_S) {
    if (bid < a.B) {
      spipe.issue(a.src, bid, a.Tin, (int)threadIdx.x);
      if constexpr (PIPE_D) dpipe.issue(a.y, bid, a.Tout, (int)threadIdx.x);
    }
  }
  HIP_DYNAMIC_SHARED(float4, g_smem4)
  float* g_smem = reinterpret_cast<float*>(g_smem4);
  constexpr int NT = (NC + 15) / 16;
  // (MWW_G_WGRAD_XG_NARROW: the gathering stem's dp rows at 8-float granularity - a B fragment of the last filter tile wraps into
  // the next row, its products only reach columns that are never stored - and the kernel held to 168 registers: 50 KB of
  // tiles, three workgroups per CU)
  const int PO = (XG && MWW_G_WGRAD_XG_NARROW) ? (NC + 7) / 8 * 8 : gwg_dp_pitch(NC);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
  const int PI = kCin | 1;
  const int cap = CH ? a.Tc : a.Tout;                   // frames of the dp tile (CH: a chunk; the A tile holds its input frames + halo)
  const int Tout4 = (cap + 3) & ~3;
  float* sA = g_smem;
  float* sDP = g_smem + ((CH ? (a.Tc - 1) * kStride + (kK - 1) * kDil + 1 : a.Tin) * PI + 3 + 3) / 4 * 4;   // (+3: the clamped A reads of a short last k-step stay in front of it)
  const int tasks = kK * kCin, MT = (tasks + 15) >> 4;
  const int KS = gwg_kparts(tasks), nslot = (kThreads / 64) / KS;
  const int kp = wave % KS, slot = wave / KS;
  // this wave's task tiles: mt = slot, slot + nslot, ...;  lane r16 <-> task m of the tile (tasks past the end repeat the
  // last one: their rows are not written)
  // EVEN (static shapes whose task tiles do not divide by the four waves, e.g. the stem's 13): the whole rounds are dealt as
  // before, the tiles left over are cut into their filter tiles and those units go to the waves (unit + rot) % 4, rot = the
  // workgroup's round of the dispatch (blockIdx / 256: workgroups i and i + 256 share a CU), so that on every SIMD the waves of
  // the resident workgroups add up to the same number of MFMAs per k-step (the stem: 13 where wave 0 of both had 8 = 16).
  constexpr int kTasksS = ST ? SH::K * SH::CIN : 0, kMTS = (kTasksS + 15) / 16;
  constexpr int kFull = kMTS / 4, kRem = kMTS % 4;
  constexpr bool EVEN = ST && MWW_G_WGRAD_EVEN != 0 && gwg_kparts(kTasksS) == 1 && kRem > 0 && kRem * NT <= 4 && kFull < kGWgTilesPerWave;
  // UNI (static shapes whose task tiles do divide by the waves that share them, e.g. 50 tasks = 4 tiles, or the 1x1 ops' one or
  // two tiles per frame part): every wave owns the same number of tiles - the k-step carries no wave-uniform branch either.
  constexpr int kNslotS = 4 / gwg_kparts(kTasksS > 0 ? kTasksS : 64);
  constexpr bool UNI = ST && MWW_G_WGRAD_EVEN != 0 && !EVEN && kMTS % kNslotS == 0 && kMTS / kNslotS <= kGWgTilesPerWave;
  constexpr int kUniTiles = UNI ? kMTS / kNslotS : 0;
  const int my_unit = EVEN ? ((wave - (bid >> 8)) & 3) : 0;       // this wave's left-over unit (if < kRem * NT)
  const bool has_rem = EVEN && my_unit < kRem * NT;
  const int rem_mt = kFull * 4 + my_unit / NT, rem_nt = my_unit % NT;
  int offA[kGWgTilesPerWave];
#pragma unroll
  for (int u = 0; u < kGWgTilesPerWave; ++u) {
    const int mt_u = (EVEN && u == kFull) ? min(rem_mt, kMTS - 1) : slot + u * nslot;
    const int m = min(mt_u * 16 + r16, tasks - 1);
    int mj, mc;
    fast_divmod(m, kCin, mj, mc);
    offA[u] = mj * kDil * PI + mc;
  }
  f32x4 acc[kGWgTilesPerWave][NT];
#pragma unroll
  for (int u = 0; u < kGWgTilesPerWave; ++u)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[u][nt] = zero4();
  // dp columns NC..PO-1 and rows Tout..Tout4-1 stay zero: staging only writes the window's own elements
  if (ST && tid < 8) sA[a.Tin * PI + tid] = 0.f;   // (the gap in front of the dp tile: read - against zero dp rows - by the unclamped k-steps)
  for (int i = tid; i < Tout4 * PO; i += kThreads) sDP[i] = 0.f;
  // statistics hand-over: the op's backward coefficients folded from the accumulator rows; this role publishes them
  __shared__ float sFoldB[3 * kGFoldC];
  XShared* sXg = nullptr;
  if constexpr (XG) {   // the descriptors and mask bitmaps of this workgroup's windows (see gconv_body)
    sXg = reinterpret_cast<XShared*>(sDP + ((Tout4 * PO + 1) & ~1));
    gx_setup(*xgp, *sXg, bid < a.B ? (a.B - bid + nb - 1) / nb : 0, tid, bid, nb);
  }
  if (a.y.fold.acc) {
    GFoldRegs fr;
    gfold_backward_load(a.y.fold, NC, a.y.rstd, tid, fr);
    gfold_backward_finish(a.y.fold, NC, sFoldB, bid, tid, fr);
  }
  int xsamp = 0;
  if constexpr (PIPE_S) {
    if constexpr (!XG) spipe.load_affine(a.src, nullptr, nullptr, tid);
    if constexpr (PIPE_D) {
      __syncthreads();   // the folded coefficients are complete
      dpipe.load_coeffs(a.y, sFoldB, tid);
    }
  }
  for (int v = bid; v < (CH ? a.B * a.S : a.B); v += nb) {
    int b = v, f0 = 0, Tin = a.Tin, Tout = a.Tout;
    if (CH) {
      int chunk;
      fast_divmod(v, a.S, b, chunk);
      f0 = chunk * a.Tc;
      Tout = min(a.Tc, a.Tout - f0);
      Tin = (Tout - 1) * kStride + (kK - 1) * kDil + 1;
    }
    __syncthreads();
    if constexpr (XG) gx_commit(xs, sA, *xgp, *sXg, xsamp, a.Tin, tid);
    else if constexpr (PIPE_S) spipe.commit(sA, PI, a.Tin, tid);
    else stage_sources<SH>(a.src, kNsrc, b, Tin, sA, PI, tid, nullptr, nullptr, CH ? f0 * kStride : 0);
    if constexpr (PIPE_D) {
      dpipe.commit(sDP, PO, a.Tout, tid);
    } else if (CH) {
      stage_dp<NC>(a.y, NC, b, Tout, sDP, PO, tid, sFoldB, f0, a.Tout);
      // a shorter last chunk leaves the previous item's rows behind its own: the k-step that straddles the end reads them
      for (int i = tid; i < (((Tout + 3) & ~3) - Tout) * PO; i += kThreads) sDP[Tout * PO + i] = 0.f;
    } else {
      stage_dp<NC>(a.y, NC, b, a.Tout, sDP, PO, tid, sFoldB);
    }
    __syncthreads();
    if constexpr (XG) {
      ++xsamp;
      if (v + nb < a.B) {
        gx_issue(xs, gx_window_lds(*sXg, xsamp), a.Tin, tid);
        dpipe.issue(a.y, v + nb, a.Tout, tid);
      }
    } else if constexpr (PIPE_S) {
      if (v + nb < a.B) {
        spipe.issue(a.src, v + nb, a.Tin, tid);
        if constexpr (PIPE_D) dpipe.issue(a.y, v + nb, a.Tout, tid);
      }
    }
    if constexpr (ST) {
      // Static shapes: the k-steps walk two base pointers with immediate offsets, four steps per trip (the run-time loop
      // below spends ~12 instructions around every MFMA: clamp, two row addresses, loop control - 47 trips for a 186-frame
      // window, the larger half of what a weight-gradient wave of a 10-filter op issues).  No clamp: the frames a short last
      // k-step reads past the window lie in front of / inside the dp tile (finite: the gap was zeroed above) and their dp
      // rows are zero.
      const int nsteps = (Tout - kp * 4 + 4 * KS - 1) / (4 * KS);
      const float* pa = sA + (kp * 4 + g) * PI;
      const float* pb = sDP + (kp * 4 + g) * PO + r16;
      // (REM: this wave owns a left-over unit - two copies of the loop instead of a branch in every k-step)
      auto run = [&](auto remc) {
        constexpr bool REM = decltype(remc)::value;
        const float* qa0 = pa;
        const float* qb0 = pb;
        auto kstep = [&](const float* qa, const float* qb) {
          float bv[NT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bv[nt] = qb[nt * 16];
          if constexpr (EVEN) {
#pragma unroll
            for (int u = 0; u < kFull; ++u) {
              const float av = qa[offA[u]];
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) acc[u][nt] = mfma4(av, bv[nt], acc[u][nt]);
            }
            if constexpr (REM) {   // the unit's filter tile is picked by a (wave-uniform) select, its products go to acc[kFull][0]
              float bsel = bv[0];
#pragma unroll
              for (int nt = 1; nt < NT; ++nt) bsel = rem_nt == nt ? bv[nt] : bsel;
              acc[kFull][0] = mfma4(qa[offA[kFull]], bsel, acc[kFull][0]);
            }
          } else if constexpr (UNI) {
#pragma unroll
            for (int u = 0; u < kUniTiles; ++u) {
              const float av = qa[offA[u]];
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) acc[u][nt] = mfma4(av, bv[nt], acc[u][nt]);
            }
          } else {
#pragma unroll
            for (int u = 0; u < kGWgTilesPerWave; ++u) {
              if (slot + u * nslot < MT) {   // wave-uniform
                const float av = qa[offA[u]];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[u][nt] = mfma4(av, bv[nt], acc[u][nt]);
              }
            }
          }
        };
        int i = 0;
        for (; i + 4 <= nsteps; i += 4) {
#pragma unroll
          for (int q = 0; q < 4; ++q) kstep(qa0 + q * 4 * KS * PI, qb0 + q * 4 * KS * PO);
          qa0 += 16 * KS * PI;
          qb0 += 16 * KS * PO;
        }
        for (; i < nsteps; ++i) {
          kstep(qa0, qb0);
          qa0 += 4 * KS * PI;
          qb0 += 4 * KS * PO;
        }
      };
      if (EVEN && has_rem) run(std::true_type{});
      else run(std::false_type{});
    } else
    for (int t0 = kp * 4; t0 < Tout; t0 += 4 * KS) {
      // A: lane (r16, g) = task r16 of the tile, frame t0 + g (clamped: the matching dp rows are zero);  B: dp[t0 + g][nt*16 + r16]
      const int tf = min(t0 + g, Tout - 1);
      const float* arow = sA + tf * kStride * PI;
      float bv[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bv[nt] = sDP[(t0 + g) * PO + nt * 16 + r16];
#pragma unroll
      for (int u = 0; u < kGWgTilesPerWave; ++u) {
        if (slot + u * nslot < MT) {   // wave-uniform
          const float av = arow[offA[u]];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[u][nt] = mfma4(av, bv[nt], acc[u][nt]);
        }
      }
    }
  }
  // D: lane (r16, g) holds dW[task g*4 + r of the tile][filter nt*16 + r16]
  float* dst = a.grad_part + (size_t)bid * ((size_t)tasks * NC);
  if constexpr (EVEN) {
#pragma unroll
    for (int u = 0; u < kFull; ++u) {   // whole rounds: tile slot + 4 u is never the last one
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = (slot + u * 4) * 16 + g * 4 + r, co = nt * 16 + r16;
          if (co < NC) store_stream<MWW_AUX_GR_ST_GP>(dst + (size_t)m * NC + co, acc[u][nt][r]);
        }
    }
    if (has_rem) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = rem_mt * 16 + g * 4 + r, co = rem_nt * 16 + r16;
        if (m < tasks && co < NC) store_stream<MWW_AUX_GR_ST_GP>(dst + (size_t)m * NC + co, acc[kFull][0][r]);
      }
    }
  } else if (KS == 1) {
#pragma unroll
    for (int u = 0; u < kGWgTilesPerWave; ++u) {
      const int mt = slot + u * nslot;
      if (mt < MT) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = mt * 16 + g * 4 + r, co = nt * 16 + r16;
            if (m < tasks && co < NC) store_stream<MWW_AUX_GR_ST_GP>(dst + (size_t)m * NC + co, acc[u][nt][r]);
          }
      }
    }
  } else {
    // sum over the frame parts in a fixed order through LDS: sQ[kp][mt][nt][16][16]
    float* sQ = g_smem;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kGWgTilesPerWave; ++u) {
      const int mt = slot + u * nslot;
      if (mt < MT) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) sQ[(((kp * MT + mt) * NT + nt) * 16 + g * 4 + r) * 16 + r16] = acc[u][nt][r];
      }
    }
    __syncthreads();
    for (int e = tid; e < MT * NT * 256; e += kThreads) {
      const int r16e = e & 15, rowe = (e >> 4) & 15, tile = e >> 8, nt = tile % NT, mt = tile / NT;
      const int m = mt * 16 + rowe, co = nt * 16 + r16e;
      if (m < tasks && co < NC) {
        float v = 0.f;
        for (int q = 0; q < KS; ++q) v += sQ[(((q * MT + mt) * NT + nt) * 16 + rowe) * 16 + r16e];
        dst[(size_t)m * NC + co] = v;
      }
    }
  }
}

template <int NC, class SH = GShapeDyn>
__global__ __launch_bounds__(kThreads) void gconv_wgrad_kernel(GWgradArgs a) {
  stagger_start<MWW_STAGGER_GRAPH>();
  gconv_wgrad_body<NC, false, SH>(a, blockIdx.x, gridDim.x);
}
template <int NC, class SH>
__global__ __launch_bounds__(kThreads, (MWW_G_WGRAD_XG_NARROW ? 3 : 1)) void gconv_wgrad_xg_kernel(GWgradArgs a, XGather xg) {
  stagger_start<MWW_STAGGER_GRAPH>();
  gconv_wgrad_body<NC, false, SH, true>(a, blockIdx.x, gridDim.x, &xg);
}
template <int NC>
__global__ __launch_bounds__(kThreads) void gconv_wgrad_chunk_kernel(GWgradArgs a) {
  gconv_wgrad_body<NC, true>(a, blockIdx.x, gridDim.x);
}

// Both halves of an op's backward in one launch: workgroups [0, nb) form the weight gradient, [nb, 2 nb) the data
// gradient.  They are independent (both only read the op's output gradient) and each is latency-bound on its
// own, so sharing the launch hides one of the two.
template <int NCO, int NCI, class SH = GShapeDyn>
__global__ __launch_bounds__(kThreads) void gconv_bwd_kernel(GWgradArgs w, GConvArgs d, int nbw, int nbd) {
  stagger_start<MWW_STAGGER_GRAPH>();
  if ((int)blockIdx.x < nbw) gconv_wgrad_body<NCO, false, SH>(w, blockIdx.x, nbw);
  else gconv_body<NCI, 1, NCO, false, SH>(d, blockIdx.x - nbw, nbd);
}
template <int NCO, int NCI>
__global__ __launch_bounds__(kThreads, 3) void gconv_bwd_chunk_kernel(GWgradArgs w, GConvArgs d, int nbw, int nbd) {
  if ((int)blockIdx.x < nbw) gconv_wgrad_body<NCO, true>(w, blockIdx.x, nbw);
  else gconv_body<NCI, 1, NCO, true>(d, blockIdx.x - nbw, nbd);
}

// ... and of twin ops: four roles
struct GBwd2Args { GWgradArgs w[2]; GConvArgs d[2]; };
template <int NCO, int NCI, class SH = GShapeDyn>
__global__ __launch_bounds__(kThreads) void gconv_bwd2_kernel(GBwd2Args a, int nbw, int nbd) {
  stagger_start<MWW_STAGGER_GRAPH>();
  const int pair = nbw + nbd, op = (int)blockIdx.x >= pair ? 1 : 0, bid = blockIdx.x - op * pair;
  if (bid < nbw) gconv_wgrad_body<NCO, false, SH>(a.w[op], bid, nbw);
  else gconv_body<NCI, 1, NCO, false, SH>(a.d[op], bid - nbw, nbd);
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
  // statistics hand-over (graphs of convolutions + BatchNorm and depthwise ops + bias):
  GFoldFwd fold;            // MODE 0: the source's forward statistics, when this launch is their first consumer
  const double* bias_acc;   // weight gradient: [kStatRows][2][C] rows whose first statistic is sum g of this op (its consumer's data gradient added them)
  float* dbeta;             // ... folded into the bias gradient by workgroup 0
};

// Register blocking (end of round 2; the first version spent two LDS reads per FMA and staged dp one float per load,
// 0.88 of the 1.52 ms of the default MixedNet on these kernels): a thread owns a channel and computes blocks of 8 frames
// x 8 taps from 8 weights (or 8 dp values) + 15 consecutive rows of its channel = 23 LDS reads per 64 FMAs, all with static
// register indices.  Taps are padded to whole blocks of 8 with zero weights and the staged windows are followed by
// kGDwTail zero rows, so no index is clamped or predicated.  Depthwise ops carry no BatchNorm (bias or nothing), so
// their dp is the stored output gradient itself: staged as a plain vector copy.
constexpr int kGDwJ = 8;       // taps per register block = frames per register block
constexpr int kGDwTail = 16;   // zero rows behind a staged window (a block reads up to 13 rows past the window)
__host__ __device__ inline int gdw_kpad(int k) { return (k + kGDwJ - 1) / kGDwJ * kGDwJ; }

// rows [0, rows) of a plain [rows][C] slab -> dst[t * ld + c]
template <int V>
__device__ __forceinline__ void gdw_stage_rows_vec(const float* slab, int rows, int C, float* dst, int ld, int tid) {
  const int NQ = C / V, nrg = fast_div(kThreads, NQ);
  int q, rg;
  fast_divmod(tid, NQ, rg, q);
  if (rg >= nrg) return;
  const BufRsrc r = tile_rsrc(slab, rows * C * 4);
  constexpr int NB = 8;
  for (int t0 = rg; t0 < rows; t0 += NB * nrg) {
    GVec<V> v[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) v[u] = gvec_bload<V>(r, (t0 + u * nrg) * C + q * V);
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int t = t0 + u * nrg;
      if (t < rows) {
#pragma unroll
        for (int e = 0; e < V; ++e) dst[t * ld + q * V + e] = v[u].f[e];
      }
    }
  }
}
__device__ __forceinline__ void gdw_stage_rows(const float* slab, int rows, int C, float* dst, int ld, int tid) {
  if ((C & 3) == 0) gdw_stage_rows_vec<4>(slab, rows, C, dst, ld, tid);
  else if ((C & 1) == 0) gdw_stage_rows_vec<2>(slab, rows, C, dst, ld, tid);
  else gdw_stage_rows_vec<1>(slab, rows, C, dst, ld, tid);
}

// out[i] = sum_j wc[j * C] * xc[(i + j) * PI], i < 8, j < 8 * njb in ascending order
__device__ __forceinline__ void gdw_block(const float* wc, int C, const float* xc, int PI, int njb, float (&out)[kGDwJ]) {
#pragma unroll
  for (int i = 0; i < kGDwJ; ++i) out[i] = 0.f;
  for (int jb = 0; jb < njb; ++jb) {
    float w[kGDwJ], x[2 * kGDwJ - 1];
#pragma unroll
    for (int jj = 0; jj < kGDwJ; ++jj) w[jj] = wc[(jb * kGDwJ + jj) * C];
#pragma unroll
    for (int m = 0; m < 2 * kGDwJ - 1; ++m) x[m] = xc[(jb * kGDwJ + m) * PI];
#pragma unroll
    for (int i = 0; i < kGDwJ; ++i)
#pragma unroll
      for (int jj = 0; jj < kGDwJ; ++jj) out[i] = fmaf(w[jj], x[i + jj], out[i]);
  }
}

// MODE 0: forward.  MODE 1: data gradient da[r] = sum_j w[j] dp[r-j] = sum_j' w[k-1-j'] dpz[r+j'] on the zero-padded dp
// (the forward form with reversed taps), masked and scattered into the source's gradient.
template <int MODE>
__global__ __launch_bounds__(kThreads) void gdw_kernel(GDwArgs a) {
  HIP_DYNAMIC_SHARED(float4, g_smem4)
  float* g_smem = reinterpret_cast<float*>(g_smem4);
  __shared__ float sRed[2 * kThreads];
  const int tid = threadIdx.x, C = a.C, PI = C | 1;
  const int kp = gdw_kpad(a.k), njb = kp / kGDwJ;
  const int pad = MODE == 1 ? a.k - 1 : 0;
  const int rows_in = (MODE == 0 ? a.Tin : a.Tout + 2 * pad);
  const int rows_out = (MODE == 0 ? a.Tout : a.Tin);
  float* sW = g_smem;               // [kp][C]
  float* sIn = sW + kp * C;         // [rows_in + kGDwTail][PI]  MODE 0: activated source rows; MODE 1: zero-padded dp rows
  const int nrg = kThreads / C, c = tid % C, rg = tid / C;
  float s1 = 0.f, s2 = 0.f;
  for (int i = tid; i < kp * C; i += kThreads) {
    int j, cc;
    fast_divmod(i, C, j, cc);
    sW[i] = j < a.k ? a.w[(MODE == 0 ? j : a.k - 1 - j) * C + cc] : 0.f;
  }
  for (int i = tid; i < kGDwTail * PI; i += kThreads) sIn[rows_in * PI + i] = 0.f;
  if (MODE == 1)
    for (int i = tid; i < pad * PI; i += kThreads) {
      sIn[i] = 0.f;
      sIn[(pad + a.Tout) * PI + i] = 0.f;
    }
  // statistics hand-over: the first consumer of the source folds its producer's sums (one wave; the table is visible
  // to the others after the loop's first barrier)
  __shared__ float sFold[MODE == 0 ? 4 * kGFoldC : 1];
  if (MODE == 0 && a.fold.acc && tid < 64) {
    GFoldRegs fr;
    gfold_forward_load(a.fold, blockIdx.x, tid, fr);
    gfold_forward_finish(a.fold, sFold, blockIdx.x, tid, fr);
  }
  const int nblk = (rows_out + kGDwJ - 1) / kGDwJ;
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    __syncthreads();
    if (MODE == 0) stage_sources(&a.src, 1, b, a.Tin, sIn, PI, tid, &a.fold, sFold);
    else gdw_stage_rows(a.y.g + (size_t)b * a.Tout * C, a.Tout, C, sIn + pad * PI, PI, tid);
    __syncthreads();
    if (rg < nrg) {
      if (MODE == 0) {
        float* dst = a.out + (size_t)b * a.Tout * C + c;
        for (int blk = rg; blk < nblk; blk += nrg) {
          const int t0 = blk * kGDwJ;
          float o[kGDwJ];
          gdw_block(sW + c, C, sIn + t0 * PI + c, PI, njb, o);
#pragma unroll
          for (int i = 0; i < kGDwJ; ++i)
            if (t0 + i < a.Tout) store_stream<MWW_AUX_GR_ST_P>(dst + (size_t)(t0 + i) * C, o[i]);
        }
      } else {
        const GSrc& s = a.src;
        const float sc = s.scale[s.c0 + c], sh = s.shift[s.c0 + c];
        const bool accum = (s.flags & GSRC_ACCUM) != 0, stats = (s.flags & GSRC_STATS) != 0, linear = (s.flags & GSRC_LINEAR) != 0;
        const float mu = stats ? s.mean[s.c0 + c] : 0.f, rs = stats ? s.rstd[s.c0 + c] : 0.f;
        const size_t base = (size_t)b * s.T * s.ld + s.c0 + c;
        const float rsc = s.rp ? s.rscale[c] : 0.f, rsh = s.rp ? s.rshift[c] : 0.f;
        const float* rbase = s.rp ? s.rp + ((size_t)b * s.rT + s.rdrop) * C + c : nullptr;
        // source rows in front of / behind the aligned input receive no gradient from this op
        for (int t = rg; t < s.T; t += nrg)
          if (t < s.toff || t >= s.toff + a.Tin) {
            const size_t idx = base + (size_t)t * s.ld;
            const float gv = accum ? s.g[idx] : 0.f;
            s.g[idx] = gv;
            s1 += gv;
            s2 = fmaf(gv, (s.p[idx] - mu) * rs, s2);
          }
        for (int blk = rg; blk < nblk; blk += nrg) {
          const int r0 = blk * kGDwJ;
          // the rows of p (and of the gradient so far) are fetched before the block is computed
          float pv[kGDwJ], rv[kGDwJ], gold[kGDwJ];
#pragma unroll
          for (int i = 0; i < kGDwJ; ++i) {
            const int t = min(r0 + i, a.Tin - 1) + s.toff;
            const size_t idx = base + (size_t)t * s.ld;
            pv[i] = s.p[idx];
            rv[i] = rbase ? rbase[(size_t)t * C] : 0.f;
            gold[i] = accum ? s.g[idx] : 0.f;
          }
          float o[kGDwJ];
          gdw_block(sW + c, C, sIn + r0 * PI + c, PI, njb, o);
#pragma unroll
          for (int i = 0; i < kGDwJ; ++i)
            if (r0 + i < a.Tin) {
              const size_t idx = base + (size_t)(r0 + i + s.toff) * s.ld;
              const float gv = ((linear || src_affine(s, pv[i], sc, sh, rv[i], rsc, rsh) > 0.f) ? o[i] : 0.f) + gold[i];
              s.g[idx] = gv;
              s1 += gv;
              s2 = fmaf(gv, (pv[i] - mu) * rs, s2);
            }
        }
      }
    }
  }
  if (MODE == 1 && (a.src.flags & GSRC_STATS))
    publish_channel_partials(s1, s2, C, sRed, a.src.gstat_part + (size_t)blockIdx.x * 2 * a.src.sld + a.src.scb, tid, a.src.sld, a.src.gacc,
                             a.src.scb, blockIdx.x, gridDim.x);
}

// dw[j][c] = sum_{b,t} act[b][t+j][c] * dp[b][t][c].  thread <-> (channel, tap block, frame part): the 8 taps of a block
// are 8 accumulators kept for the whole launch; a channel's tap blocks (and, when there are fewer tap blocks than
// threads per channel, parts of the frame range) go to the kThreads / C threads of that channel, the frame parts are
// summed in a fixed order through LDS at the end.  taps x channels <= kGDwTasks * kThreads keeps it at two (tap block,
// part) pairs per thread.
constexpr int kGDwTasks = 8;
__global__ __launch_bounds__(kThreads) void gdw_wgrad_kernel(GDwArgs a) {
  HIP_DYNAMIC_SHARED(float4, g_smem4)
  float* g_smem = reinterpret_cast<float*>(g_smem4);
  const int tid = threadIdx.x, C = a.C, PI = C | 1;
  const int njb = gdw_kpad(a.k) / kGDwJ;
  float* sA = g_smem;                               // [Tin + kGDwTail][PI]
  float* sDP = g_smem + (a.Tin + kGDwTail) * PI;    // [Tout + kGDwJ][PI]
  const int nslot = kThreads / C, c = tid % C, slot = tid / C;
  const int FP = max(1, nslot / njb), Q = njb * FP;
  const int nblk = (a.Tout + kGDwJ - 1) / kGDwJ, nbp = (nblk + FP - 1) / FP;
  float acc[2][kGDwJ];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int jj = 0; jj < kGDwJ; ++jj) acc[u][jj] = 0.f;
  for (int i = tid; i < kGDwTail * PI; i += kThreads) sA[a.Tin * PI + i] = 0.f;
  for (int i = tid; i < kGDwJ * PI; i += kThreads) sDP[a.Tout * PI + i] = 0.f;
  if (a.bias_acc && blockIdx.x == 0 && tid < C) {   // d bias = sum of the output gradient, in the fixed order of the rows
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < kStatRows; ++j) t += a.bias_acc[(size_t)j * 2 * C + tid];
    a.dbeta[tid] = (float)t;
  }
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    __syncthreads();
    stage_sources(&a.src, 1, b, a.Tin, sA, PI, tid);
    gdw_stage_rows(a.y.g + (size_t)b * a.Tout * C, a.Tout, C, sDP, PI, tid);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = slot + u * nslot;
      if (slot < nslot && q < Q) {
        int fp, jb;
        fast_divmod(q, njb, fp, jb);
        const int b0 = fp * nbp, b1 = min(nblk, b0 + nbp);
        for (int blk = b0; blk < b1; ++blk) {
          const int t0 = blk * kGDwJ;
          float d[kGDwJ], x[2 * kGDwJ - 1];
#pragma unroll
          for (int i = 0; i < kGDwJ; ++i) d[i] = sDP[(t0 + i) * PI + c];
#pragma unroll
          for (int m = 0; m < 2 * kGDwJ - 1; ++m) x[m] = sA[(t0 + jb * kGDwJ + m) * PI + c];
#pragma unroll
          for (int i = 0; i < kGDwJ; ++i)
#pragma unroll
            for (int jj = 0; jj < kGDwJ; ++jj) acc[u][jj] = fmaf(x[i + jj], d[i], acc[u][jj]);
        }
      }
    }
  }
  // frame parts summed in a fixed order: sQ[q][tap of the block][C]
  float* sQ = g_smem;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int q = slot + u * nslot;
    if (slot < nslot && q < Q) {
#pragma unroll
      for (int jj = 0; jj < kGDwJ; ++jj) sQ[(q * kGDwJ + jj) * C + c] = acc[u][jj];
    }
  }
  __syncthreads();
  const int tasks = a.k * C;
  for (int task = tid; task < tasks; task += kThreads) {
    int j, cc;
    fast_divmod(task, C, j, cc);
    const int jb = j / kGDwJ, jj = j % kGDwJ;
    float v = 0.f;
    for (int fp = 0; fp < FP; ++fp) v += sQ[((fp * njb + jb) * kGDwJ + jj) * C + cc];
    a.grad_part[(size_t)blockIdx.x * tasks + task] = v;
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
  GFoldFwd fold;           // fold.acc set: the last op's statistics are folded here (this is their first consumer)
  StatAcc gacc;            // gacc.acc set: (sum g, sum g*xhat) go to the last op's accumulator rows instead of gstat_part
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
  // Register-resident path (windows with at most kHR rows per thread and no residual branch: Inception's head): a window's
  // rows are loaded once (buffer loads, no predicate) and stay in registers for the backward half; the dense kernel rows
  // of the thread are loaded once per workgroup.  The first window's rows and the dense rows are requested before the
  // statistics fold below (they do not depend on it: one round trip less in front of the first window).
  constexpr int kHR = 12;
  const bool fast = a.rp == nullptr && (a.T + nrg - 1) / nrg <= kHR;
  const bool kgen_on = a.keep_gen != nullptr;
  float wvf[kHR], pvf[kHR], kvf[kHR];
  auto load_rows = [&](int b) {
    const float* kb = a.keep ? a.keep + (size_t)b * n : nullptr;
    const BufRsrc prs = tile_rsrc(a.p + (size_t)b * n, active ? n * 4 : 0), krs = tile_rsrc(kb, (active && kb && !kgen_on) ? n * 4 : 0);
#pragma unroll
    for (int u = 0; u < kHR; ++u) {
      const int off = ((rg + u * nrg) * C + c) * 4;
      pvf[u] = tile_load1(prs, off);
      kvf[u] = tile_load1(krs, off);
    }
  };
  if (fast) {
    const BufRsrc wrs = tile_rsrc(a.wd, active ? n * 4 : 0);
#pragma unroll
    for (int u = 0; u < kHR; ++u) wvf[u] = tile_load1(wrs, ((rg + u * nrg) * C + c) * 4);
    if ((int)blockIdx.x < a.B) load_rows(blockIdx.x);
  }
  __shared__ float sFold[4 * kGFoldC];
  const bool folded = a.fold.acc != nullptr;
  if (folded) {
    GFoldRegs fr;
    gfold_forward_load(a.fold, blockIdx.x, tid, fr);
    gfold_forward_finish(a.fold, sFold, blockIdx.x, tid, fr);
    __syncthreads();
  }
  const float sc = active ? (folded ? sFold[c] : a.scale[c]) : 0.f, sh = active ? (folded ? sFold[kGFoldC + c] : a.shift[c]) : 0.f;
  const float mu = (active && (a.training & kHeadTraining)) ? (folded ? sFold[2 * kGFoldC + c] : a.mean[c]) : 0.f;
  const float rs = (active && (a.training & kHeadTraining)) ? (folded ? sFold[3 * kGFoldC + c] : a.rstd[c]) : 0.f;
  const float rsc = (active && a.rp) ? a.rscale[c] : 0.f, rsh = (active && a.rp) ? a.rshift[c] : 0.f;
  const float bias = a.bd[0];
  float g1 = 0.f, g2 = 0.f;
  // this step's dropout counter, read once (inside the window loop it was a dependent round trip per window: the stores of
  // the loop keep the compiler from hoisting it)
  const unsigned long long step = kgen_on ? (((unsigned long long)a.counter[1] << 32) | a.counter[0]) : 0ull;
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    const float* pb = a.p + (size_t)b * n;
    const float* kb = a.keep ? a.keep + (size_t)b * n : nullptr;
    const float* rb = a.rp ? a.rp + ((size_t)b * a.rT + a.rdrop) * C : nullptr;
    float* kgen = a.keep_gen ? a.keep_gen + (size_t)b * n : nullptr;
    float dot = 0.f;
    // the window's label and weight travel with its rows (thread 0 needs them between the two barriers below: fetched
    // there they were one more memory round trip per window on the workgroup's critical path)
    float y_pre = 0.f, w_pre = 0.f;
    if (tid == 0 && a.y != nullptr) {
      y_pre = a.y[b];
      if (a.training & kHeadTraining) w_pre = a.sw[b];
    }
    // rows in batches of kHB per thread: every load of a batch is issued before the first use (one load, one wait per
    // row made this kernel a chain of ~2 T C / 256 memory round trips per window: 51 us per launch for 10 MB in round 2)
    constexpr int kHB = 8;
    if (fast) {
      if (b != (int)blockIdx.x) load_rows(b);   // (the first window's rows are on their way since the kernel's entry)
#pragma unroll
      for (int u = 0; u < kHR; ++u) {
        const int t = rg + u * nrg;
        if (active && t < a.T) {
          const int i = t * C + c;
          const float act = fmaxf(fmaf(pvf[u], sc, sh), 0.f);
          float k1 = (kb && !kgen) ? kvf[u] : 1.f;
          if (kgen) {
            k1 = dropout_keep(a.seed, step, (unsigned long long)b * n + i, a.rate);
            kgen[i] = k1;   // read by the dense-weight gradient
          }
          kvf[u] = k1;
          dot = fmaf(act * k1, wvf[u], dot);
        }
      }
    } else if (active) {
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
        const float yy = y_pre;
        const bool clipped_form = (a.training & kHeadClippedLoss) != 0;
        const float bce = bce_value(zz, pr, yy, clipped_form);
        if (a.training & kHeadTraining) {
          const float w = w_pre;
          a.loss_part[b] = w * bce * a.inv_b;
          dzz = w * bce_dz(pr, yy, clipped_form) * a.inv_b;
          a.dz[b] = dzz;
        }
      }
      sBcast[0] = dzz;
    }
    __syncthreads();
    if ((a.training & kHeadTraining) && active && fast) {
      const float dzz = sBcast[0];
      const BufRsrc grs = tile_rsrc(a.g + (size_t)b * n, n * 4);
#pragma unroll
      for (int u = 0; u < kHR; ++u) {
        const int t = rg + u * nrg;
        if (t < a.T) {
          const float raw = pvf[u];
          const float gv = (fmaf(raw, sc, sh) > 0.f ? dzz * wvf[u] : 0.f) * kvf[u];
          tile_store1<MWW_AUX_GR_ST_G>(grs, (t * C + c) * 4, gv);
          g1 += gv;
          g2 = fmaf(gv, (raw - mu) * rs, g2);
        }
      }
    } else if ((a.training & kHeadTraining) && active) {
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
  if (a.training & kHeadTraining)
    publish_channel_partials(g1, g2, C, sStat, a.gstat_part + (size_t)blockIdx.x * 2 * C, tid, C, a.gacc, 0, blockIdx.x, gridDim.x);
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

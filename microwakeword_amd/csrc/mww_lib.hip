// libmww_hip.so — context, device memory, launch sequencing and the C ABI of include/mww.h.
// One context = one device + one HIP stream + one model; every call enqueues on that stream.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <climits>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/mww.h"
#include "block_launch.hip.h"
#include "kernels_data.hip.h"
#include "kernels_graph.hip.h"
#include "kernels_head.hip.h"
#include "kernels_tail.hip.h"

// the fp32 block backward runs as the wide-workgroup form (option "bwd_wide"; kernels_bwdw.hip.h) unless told otherwise
// options "conv1_x6" (the conv1 weight gradient in the first block's backward kernel) and "conv1_x6_fwd" (the first convolution
// itself): see common.hip.h "fp32-grade products on the bf16 matrix pipe".  Same-session A/B at B = 1024 (profiles/round6_conv1_x6_ab.txt):
// backward 49.7 -> 44.0 us; forward 33.5 -> 36-37 us (the matrix pipe's 6.5 us are paid back by the slicing of x and W1 in a
// launch whose workgroups see three tiles each) - so the default is backward only.
#ifndef MWW_CONV1_X6_DEFAULT
#define MWW_CONV1_X6_DEFAULT 1
#endif
#ifndef MWW_CONV1_X6_FWD_DEFAULT
#define MWW_CONV1_X6_FWD_DEFAULT 0
#endif
#ifndef MWW_BWD_FIRST_WIDE_DEFAULT   // option "bwd_first_wide"
#define MWW_BWD_FIRST_WIDE_DEFAULT 0
#endif
#ifndef MWW_BWD_WIDE_DEFAULT
#define MWW_BWD_WIDE_DEFAULT 1
#endif

using namespace mww;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      return fail(MWW_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                \
  } while (0)

struct Layer {
  int cin, cout, k, tin, tout;
  // offsets into the flat parameter / state vectors
  int64_t o_dw_w, o_dw_b, o_pw_w, o_gamma, o_beta, o_mm, o_mv;
  // device buffers
  float* p = nullptr;          // pre-BN output [maxB][tout][cout]
  float* g = nullptr;          // gradient at the BN output (masked by ReLU) [maxB][tout][cout]
  float* stat_part = nullptr;  // [grid_fwd][2][cout]
  float* gstat_part = nullptr; // [grid_bwd or grid_head][2][cout]
  float* grad_part = nullptr;  // [grid_bwd][params of the block (+ conv1 for block 0)]
  int grad_part_stride = 0;
  float* bn = nullptr;         // 9 x cout: scale, shift, mean, rstd, c1, mg, mgx, (spare x2)
  // statistics hand-over without finalize launches (common.hip.h): [parity][kStatRows][2][cout] for the forward
  // sums (x, x^2) and the backward sums (g, g*xhat); *_cur = rows the latest producer launch added to
  double* facc[2] = {nullptr, nullptr};
  double* gacc[2] = {nullptr, nullptr};
  double* facc_cur = nullptr;
  double* gacc_cur = nullptr;
};

// one conv -> BN/SSN -> ReLU op of a mww_convnet_desc graph (kernels_graph.hip.h)
struct GOp {
  int n_src = 0, src[kGMaxSrc] = {0, 0, 0}, toff[kGMaxSrc] = {0, 0, 0};
  int sc0[kGMaxSrc] = {0, 0, 0}, scn[kGMaxSrc] = {0, 0, 0};   // channel slice of each source
  bool src_first[kGMaxSrc] = {false, false, false}, src_last[kGMaxSrc] = {false, false, false};   // this op's place among the consumers of that slice (backward order)
  int k = 1, dil = 1, cin = 0, cout = 0, groups = 1, slots = 0, tin = 0, tout = 0;
  int kind = MWW_OP_CONV, stride = 1, norm = MWW_NORM_BN, act = MWW_ACT_RELU;
  int res_src = -1, res_drop = 0;     // residual branch added before this op's activation
  std::vector<int> adders;           // (residual ops) the ops that add this one
  int64_t o_w = 0, o_gamma = 0, o_beta = 0, o_mm = 0, o_mv = 0;
  float *p = nullptr, *g = nullptr, *stat_part = nullptr, *gstat_part = nullptr, *grad_part = nullptr, *bn = nullptr;
  bool needs_dx = false;
  bool twin_next = false;     // op i+1 is an independent op of the same shape: the pair shares its launches
  int planes = 1, pc = 0;     // > 1: every consumer reads one of `planes` equal channel slices of pc channels: the tensors p / g may be
                              // stored one plane per slice (kernels_graph.hip.h GSrc; "graph_planar")
  size_t lds_fwd = 0, lds_dx = 0, lds_wg = 0;
  // statistics hand-over (kernels_graph.hip.h): [parity][kStatRows][2][cout] accumulator rows of the forward / backward sums,
  // *_cur = the rows the latest producer launch added to; first_consumer = the lowest op index that reads this op
  double* facc[2] = {nullptr, nullptr};
  double* gacc[2] = {nullptr, nullptr};
  double* facc_cur = nullptr;
  double* gacc_cur = nullptr;
  int first_consumer = -1;
};

struct ProfileEntry {
  std::string name;
  hipEvent_t a, b;
};

// RCCL bound at run time (dlopen: the library loads on hosts without RCCL and shares the copy a framework in the same
// process has already loaded); only the five entry points the gradient / statistics exchange needs
struct RcclApi {
  void* so = nullptr;
  struct UniqueId { char internal[128]; };
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;   // optional: mww_allreduce_world
  const char* (*GetErrorString)(int) = nullptr;
};

constexpr int kRing = 8;
constexpr int kDenseChunks = 32;  // batch chunks of the dense-weight gradient reduction

}  // namespace

struct mww_ctx {
  mww_mixednet_desc d;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int n_cu = 256;
  int grid_fwd = 0, grid_bwd = 0, grid_head = 0;
  bool conv1_x6 = MWW_CONV1_X6_DEFAULT != 0;   // conv1 weight gradient as six bf16 slice products per fp32 product (stride-1 shapes, fp32 mode)
  bool conv1_x6_fwd = MWW_CONV1_X6_FWD_DEFAULT != 0;
  bool bwd_first_wide = MWW_BWD_FIRST_WIDE_DEFAULT != 0;   // stride-1 first block (3-tap conv1) with conv1_x6: the 512-thread form of its backward kernel   // ... and the first convolution of the forward kernel
  bool bwd_wide = MWW_BWD_WIDE_DEFAULT != 0;   // fp32 block backward kernels: 512 threads per workgroup (bwd_blockw_kernel) or 256 (bwd_block_kernel)
  int64_t P = 0, S = 0;
  int64_t o_conv1 = 0, o_dense_w = 0, o_dense_b = 0;
  int t_last = 0, c_last = 0, dwd_stride = 0;
  std::vector<Layer> L;
  // conv/BN graph models (mww_create_convnet)
  bool generic = false;
  std::vector<GOp> G;
  float dropout = 0.f;
  float* keep = nullptr;            // [max_batch][t_last*c_last] dropout keep-scale
  bool keep_explicit = false;       // set by mww_set_dropout_mask: do not regenerate
  unsigned long long dropout_seed = 0x5EEDull, dropout_counter = 0;
  bool head2 = false;               // attention / pooled head (ghead_att_kernel)
  bool head_att = false;
  int head_pool = 0;
  int64_t o_att = 0;
  float *hact = nullptr, *watt_part = nullptr;
  size_t lds_head2 = 0;
  float *ones = nullptr, *zeros = nullptr;   // [256] constants standing in for the BN arrays of ops without a BN
  int grid_g = 0;
  int g_cap_fwd = 4, g_cap_bwd = 4;   // "graph_fwd_wg_per_cu" / "graph_bwd_wg_per_cu" (g_role_grid)
  int metric_launches = 0;   // launches of the step being enqueued that carry the metric role (kernels_head.hip.h MetricState: one writer)
  bool g_planar = true;   // "graph_planar": tensors read only as equal channel slices are stored one plane per slice
  bool g_static = true;   // "graph_static_shapes": ops whose shape has a compile-time instantiation (MWW_G_SHAPES) use it
  int g_chunks = 0;   // "graph_frame_chunks" (g_chunks())
  int g_dgrad_share = 50;   // "graph_dgrad_share"
  bool grid_g_auto = true;   // per-launch grids from the kernel's occupancy (g_role_grid); "grid_graph" > 0 fixes one grid
  std::map<std::pair<const void*, size_t>, int> g_occ;   // workgroups per CU of (kernel, dynamic LDS)
  // data-parallel exchange hook (mww_set_allreduce_hook)
  mww_allreduce_fn hook = nullptr;
  void* hook_user = nullptr;
  int world = 1;
  bool sync_bn = false, reduce_grads = false;
  struct RcclState* rccl = nullptr;  // mww_allreduce_init: the library's own communicator + side stream (the hook then points at it)
  float* sync_buf = nullptr;        // [layers][fwd 2C | bwd 2C] statistics sums being exchanged
  std::vector<int64_t> sync_off;    // offset of layer i in sync_buf
  float *params = nullptr, *grads = nullptr, *adam_m = nullptr, *adam_v = nullptr, *mask = nullptr;
  unsigned char* direct = nullptr;
  std::vector<unsigned char> direct_host;   // host copy: which parameters' gradients are written directly by a folding kernel
  bool exchange_pending = false;            // a deferred bucket exchange is in flight (data-parallel step)
  int grad_buckets = 1;                     // data-parallel step: gradient exchanged in this many buckets ("grad_buckets" option; 2 = first
                                            // bucket overlapped with the backward tail - slower at W = 1, unmeasured at W > 1, so not the default)
  float* bn_state = nullptr;
  float *x = nullptr, *y = nullptr, *sw = nullptr, *z = nullptr, *prob = nullptr, *dz = nullptr, *loss_part = nullptr;
  float* a0 = nullptr;     // relu(conv1(x)) [max_batch][Ta][conv1_filters]: written by the training forward, read by bwd_first_kernel
  float* gbuf[2] = {nullptr, nullptr};   // the two buffers the blocks' g_k take in turn (block k uses gbuf[k & 1])
  float* dwd_part = nullptr;
  MetricState* metrics = nullptr;
  // "mailboxes": pinned host memory mapped into the device address space.  The host writes one
  // step's descriptors (windows, masks, labels, weights, Adam step size) into mailbox m and the
  // kernels read them in place over PCIe (56 KB/step) — no H2D copy kernels on the stream.  A
  // mailbox is rewritten only after the event of its previous use has completed.
  char* mail_host[kRing] = {};
  char* mail_dev[kRing] = {};
  hipEvent_t mail_ev[kRing] = {};
  int mail_cur = 0;
  bool mail_open = false;
  size_t mail_off_masks = 0, mail_off_y = 0, mail_off_sw = 0, mail_off_hyper = 0, mail_bytes = 0;
  int targets_in_mail = 0;   // rows of (y, sw) sitting in the current mailbox, not yet on the device
  // side stream: work that is off the critical path of the step (metric update, dense-weight gradient)
  // descriptors reach HBM through a DMA copy on their own stream, issued as soon as the host has
  // written the mailbox — it overlaps the previous step's kernels; only the Adam step size is read in
  // place from the mapped mailbox
  hipStream_t copy_stream = nullptr;
  char* mail_hbm[kRing] = {};
  hipEvent_t ev_copy[kRing] = {};
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool side_pending = false;
  int asm_split = 2;        // workgroups per window of the assembly kernel ("assemble_split" option)
  // "fused_input" option (default on, specialised MixedNet kernels only): mww_assemble_batch only uploads the window
  // descriptors; the first block's forward / backward kernels gather their rows from the stores themselves
  // (kernels_fwd.hip.h XGather).  x is materialised (assemble_kernel) only for a reader that needs it.
  bool fused_input = true;
  bool x_lazy = false;
  int lazy_slot = -1;
  AssembleArgs lazy_a;
  const float* y_cur = nullptr;   // labels / sample weights the kernels read: the y / sw buffers, or the rows that
  const float* sw_cur = nullptr;  // travelled in the mailbox of a descriptor-only batch
  // "bn_inline" option (default on): BN statistics travel through replicated fp64 accumulator rows and are folded by
  // their first consumer instead of by a finalize launch (off with sync-BN: the sums must be exchanged in between)
  bool bn_inline = true;
  bool g_role_split = true;   // launches that hold several roles (twin ops, weight + data gradient) divide the workgroups between the
                              // roles instead of multiplying them ("graph_role_split"; needs the statistics hand-over: the partial-row
                              // readers assume one row count per tensor)
  bool g_inline_ok = false;   // conv/BN graph: every op is a convolution with a BatchNorm and no residual branch => hand-over possible
  int fpar = 0, gpar = 0;   // accumulator parity of the next training forward / backward
  bool tail_pending = false, tail_metrics = false;   // dense gradient (+ metrics) ride in the first backward launch
  bool tail_in_reduce = false;   // ... or, with the statistics hand-over, in the gradient-reduction launch ("tail_roles" option)
  bool tail_roles = true;
  void* store[MWW_MAX_STORES] = {};
  int store_dtype[MWW_MAX_STORES] = {};
  int64_t store_elems[MWW_MAX_STORES] = {};
  int64_t step = 0;
  int have_batch = 0, have_targets = 0;
  bool use_graphs = false, profile = false;
  bool profile_split = false;   // "profile_split" option: keep weight- and data-gradient of a graph op in separate launches
  bool use_side = false;  // "side_stream" option: metric update + dense-weight gradient on a second stream (measured: co-running
                          // kernels displace workgroups of the occupancy-tuned block kernels; serial is 8 us/step faster)
  bool pw_bf16 = false;   // 1x1 contractions with bf16 operands (mww_set_option "pointwise_bf16")
  bool st_bf16 = false;   // p_k / g_k stored as bf16 ("storage_bf16", implies pointwise_bf16: BASELINE configs[4])
  bool bce_clipped = false;   // "bce_from_logits" 0: probability-form BCE with the Keras clip instead of the logits form (common.hip.h)
  bool bn_eval_ready = false;   // inside mww_evaluate_windows: the moving statistics are folded once, not per batch
  int ablate = 0;
  unsigned long long* phase_clk = nullptr;   // profiling: [2*layers][2048 workgroups][8 phases]
  std::vector<ProfileEntry> prof;
  // cached graphs keyed by (B, flags)
  struct GraphEntry { int B, flags, mail, par; hipGraphExec_t exec; };
  std::vector<GraphEntry> graphs;
};

namespace {

struct Launcher {
  mww_ctx* c;
  hipEvent_t ea = nullptr;
  const char* name = nullptr;
  size_t idx = 0;   // this bracket's entry (a launch that has to write x out first opens a bracket of its own inside the caller's)
  void begin(const char* n, int layer = -1) {
    if (!c->profile) return;
    name = n;
    idx = c->prof.size();
    ProfileEntry e;
    e.name = n;
    if (layer >= 0) e.name += std::to_string(layer + 1);
    (void)hipEventCreate(&e.a);
    (void)hipEventCreate(&e.b);
    (void)hipEventRecord(e.a, c->stream);
    c->prof.push_back(e);
  }
  void end() {
    if (!c->profile || idx >= c->prof.size()) return;
    (void)hipEventRecord(c->prof[idx].b, c->stream);
  }
};

// ---------------------------------------------------------------------------------- dispatch
// The block kernels are instantiated and launched in their own translation units (tu_fwd.hip, tu_bwd.hip, tu_bwdw.hip:
// compiled in parallel by build()); block_launch.hip.h declares their launchers and the table of specialised shapes.
int launch_fwd_first(mww_ctx* c, int k1, int c1, int cout, int k, int st, const FwdFirstArgs& a, int grid) {
  if (k_launch_fwd_first(c->stream, c->st_bf16 ? 2 : (c->pw_bf16 ? 1 : 0), k1, c1, cout, k, st, a, grid, c->conv1_x6_fwd)) return MWW_OK;
  return fail(MWW_ERR_UNSUPPORTED, "no first-block kernel for this (conv1 kernel, filters, pointwise, depthwise) shape");
}

int launch_bwd_first(mww_ctx* c, int k1, int c1, int cout, int k, int st, const BwdFirstArgs& a, int grid) {
  if (c->bwd_wide && !c->pw_bf16 && !c->st_bf16 && k_launch_bwd_firstw(c->stream, k1, c1, cout, k, st, a, grid, c->conv1_x6 && c->bwd_first_wide)) return MWW_OK;
  if (k_launch_bwd_first(c->stream, c->st_bf16 ? 2 : (c->pw_bf16 ? 1 : 0), k1, c1, cout, k, st, a, grid, c->conv1_x6)) return MWW_OK;
  return fail(MWW_ERR_UNSUPPORTED, "no first-block backward kernel for this shape");
}

int launch_fwd_block(mww_ctx* c, int cin, int cout, int k, const FwdBlockArgs& a, int grid) {
  if (k_launch_fwd_block(c->stream, c->st_bf16 ? 2 : (c->pw_bf16 ? 1 : 0), cin, cout, k, a, grid)) return MWW_OK;
  return fail(MWW_ERR_UNSUPPORTED, "no block kernel for this (cin, cout, depthwise) shape");
}

int launch_bwd_block(mww_ctx* c, int cin, int cout, int k, bool last, const BwdBlockArgs& a, int grid) {
  const int mode = c->st_bf16 ? 2 : (c->pw_bf16 ? 1 : 0);
  if (c->bwd_wide && k_launch_bwd_blockw(c->stream, mode, cin, cout, k, last, a, grid)) return MWW_OK;
  if (k_launch_bwd_block(c->stream, c->st_bf16 ? 2 : (c->pw_bf16 ? 1 : 0), cin, cout, k, last, a, grid)) return MWW_OK;
  return fail(MWW_ERR_UNSUPPORTED, "no block backward kernel for this shape");
}

constexpr int kHeadMaxRows = 24;   // frames per frame group of the widest head_kernel instantiation below
// final frames one head workgroup covers at `ch` channels (a thread keeps one float4 of every frame of its group)
int head_frame_limit(int ch) { return (kThreads / (ch / 4)) * kHeadMaxRows; }

int launch_head(mww_ctx* c, int ch, int jmax, const HeadArgs& a, int grid) {
#define X(C, J)                                                                                                \
  if (ch == C && jmax <= J) {                                                                                  \
    if (c->st_bf16)                                                                                            \
      hipLaunchKernelGGL((head_kernel<C, J, true>), dim3(grid), dim3(kThreads), 0, c->stream, a);              \
    else                                                                                                       \
      hipLaunchKernelGGL((head_kernel<C, J>), dim3(grid), dim3(kThreads), 0, c->stream, a);                    \
    return MWW_OK;                                                                                             \
  }
  static_assert(kHeadMaxRows == 24, "the widest instantiation below");
  X(32, 2) X(32, 4) X(32, 8) X(32, 12) X(32, 16) X(32, 24) X(48, 2) X(48, 4) X(48, 8) X(48, 12) X(48, 16) X(48, 24) X(64, 2) X(64, 4) X(64, 8) X(64, 12) X(64, 16) X(64, 24)
#undef X
  return fail(MWW_ERR_UNSUPPORTED, "no head kernel for this (channels, frames) shape");
}

// does every block of the model have a specialised kernel (bf16: in the bf16 modes too)?
bool shape_supported(const mww_mixednet_desc& d, std::string* why, bool bf16 = false) {
  if (d.n_blocks < 2 || d.n_blocks > MWW_MAX_BLOCKS) { *why = "the block kernels serve 2.." + std::to_string(MWW_MAX_BLOCKS) + " blocks"; return false; }
  bool ok = false;
#define X(K1, C1, CO, K, S) ok = ok || (d.conv1_kernel == K1 && d.conv1_filters == C1 && d.block_filters[0] == CO && d.block_kernel[0] == K && d.conv1_stride == S);
  if (bf16) { MWW_FIRST_SHAPES_BF16(X) } else { MWW_FIRST_SHAPES(X) }
#undef X
  if (!ok) { *why = "first block (conv1 kernel/filters/stride, pointwise filters, depthwise kernel) not instantiated"; return false; }
  for (int i = 1; i < d.n_blocks; ++i) {
    ok = false;
#define X(CI, CO, K) ok = ok || (d.block_filters[i - 1] == CI && d.block_filters[i] == CO && d.block_kernel[i] == K);
    if (bf16) { MWW_BLOCK_SHAPES_BF16(X) } else { MWW_BLOCK_SHAPES(X) }
#undef X
    if (!ok) { *why = "block " + std::to_string(i) + " (cin, cout, depthwise kernel) not instantiated"; return false; }
  }
  const int cl = d.block_filters[d.n_blocks - 1];
  if (cl != 32 && cl != 48 && cl != 64) { *why = "head kernel needs 32, 48 or 64 channels"; return false; }
  // the classifier head keeps a window's final frames in registers: more of them than its widest instantiation holds would only
  // surface as MWW_ERR_UNSUPPORTED at the first forward (found by tools/gpu_x6_fuzz.py case 460: 64 channels x 390 frames)
  int t = d.frames >= d.conv1_kernel && d.conv1_stride > 0 ? (d.frames - d.conv1_kernel) / d.conv1_stride + 1 : 0;
  for (int i = 0; i < d.n_blocks; ++i) t -= d.block_kernel[i] - 1;
  if (t > head_frame_limit(cl)) {
    *why = "head kernel holds at most " + std::to_string(head_frame_limit(cl)) + " final frames at " + std::to_string(cl) + " channels (" + std::to_string(t) + " here)";
    return false;
  }
  return true;
}

float* bn_slot(Layer& l, int i) { return l.bn + (size_t)i * l.cout; }
enum { BN_SCALE = 0, BN_SHIFT, BN_MEAN, BN_RSTD, BN_C1, BN_MG, BN_MGX };

// ---------------------------------------------------------------------------------- sequences
const float* mail_hyper(mww_ctx* c) { return reinterpret_cast<const float*>(c->mail_dev[c->mail_cur] + c->mail_off_hyper); }

// labels / weights read in place from the mailbox of a descriptor-only batch -> the y / sw buffers (before that
// mailbox slot can be rewritten)
int bring_targets(mww_ctx* c) {
  if (c->y_cur == c->y) return MWW_OK;
  const size_t n = (size_t)c->lazy_a.B * sizeof(float);
  HIPCHK(hipMemcpyAsync(c->y, c->y_cur, n, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->sw, c->sw_cur, n, hipMemcpyDeviceToDevice, c->stream));
  c->y_cur = c->y;
  c->sw_cur = c->sw;
  return MWW_OK;
}

// descriptor-only batch -> x, for readers outside the first block's kernels
int materialise_x(mww_ctx* c) {
  int rc = bring_targets(c);
  if (rc || !c->x_lazy) return rc;
  AssembleArgs a = c->lazy_a;
  a.n_targets = 0;
  Launcher lp{c};
  lp.begin("assemble");
  hipLaunchKernelGGL(assemble_kernel, dim3(a.B * a.split), dim3(kThreads), 0, c->stream, a);
  lp.end();
  HIPCHK(hipGetLastError());
  c->x_lazy = false;
  return MWW_OK;
}

XGather x_gather(mww_ctx* c) {
  XGather g;
  memset(&g, 0, sizeof(g));
  if (!c->x_lazy) return g;
  const AssembleArgs& a = c->lazy_a;
  g.win = a.win;
  g.masks = a.masks;
  for (int i = 0; i < MWW_MAX_STORES; ++i) { g.store[i] = a.store[i]; g.dtype[i] = a.dtype[i]; }
  g.ntm = a.ntm;
  g.nfm = a.nfm;
  g.T = a.T;
  return g;
}

int join_side(mww_ctx* c) {
  if (c->side_pending) {
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    c->side_pending = false;
  }
  return MWW_OK;
}

// off the critical path: metric update and (training) the dense-weight gradient run on the side
// stream while the backward chain proceeds; joined before the gradient assembly
int enqueue_side_work(mww_ctx* c, int B, bool metrics, bool loss, const float* p_last, const float* scale,
                      const float* shift, const float* keep) {
  Launcher lp{c};
  if (metrics || loss) {
    const bool inline_side = c->profile || !c->use_side;
    hipStream_t ss = inline_side ? c->stream : c->side;
    if (!inline_side) {
      HIPCHK(hipEventRecord(c->ev_fork, c->stream));
      HIPCHK(hipStreamWaitEvent(c->side, c->ev_fork, 0));
    }
    const bool one_launch = metrics && loss && inline_side;   // both pieces as roles of one launch (head_tail_kernel without a finalize role)
    if (metrics && !one_launch) {
      MetricsArgs ma{c->prob, c->y_cur, c->metrics, B, c->bce_clipped ? nullptr : c->z};
      lp.begin("metrics");
      hipLaunchKernelGGL(metrics_kernel, dim3(1), dim3(1024), 0, ss, ma);
      c->metric_launches += 1;
      lp.end();
    }
    if (loss) {
      const int dchunk = (B + kDenseChunks - 1) / kDenseChunks;
      const int ndchunks = (B + dchunk - 1) / dchunk;
      DenseGradArgs dg{p_last, scale, shift, c->dz, c->dwd_part, B, c->t_last * c->c_last, c->c_last, c->dwd_stride, dchunk, keep,
                       nullptr, nullptr, nullptr, 0, 0, c->st_bf16 ? 1 : 0};
      if (c->generic && !c->head2 && c->G.back().res_src >= 0) {
        GOp& rr = c->G[c->G.back().res_src];
        dg.rp = rr.p;
        dg.rscale = rr.bn + (size_t)BN_SCALE * rr.cout;
        dg.rshift = rr.bn + (size_t)BN_SHIFT * rr.cout;
        dg.rT = rr.tout;
        dg.rdrop = c->G.back().res_drop;
      }
      if (one_launch) {
        HeadTailArgs ht;
        memset(&ht, 0, sizeof(ht));
        ht.dense = dg;
        ht.met = MetricsArgs{c->prob, c->y_cur, c->metrics, B, c->bce_clipped ? nullptr : c->z};
        ht.n_fin = 0;
        ht.ndx = (dg.n + 1 + kThreads - 1) / kThreads;
        ht.ndy = ndchunks;
        ht.do_metrics = 1;
        c->metric_launches += 1;
        lp.begin("dense_grad+metrics");
        hipLaunchKernelGGL(head_tail_kernel, dim3(ht.ndx * ht.ndy + 1), dim3(kThreads), 0, ss, ht);
        lp.end();
      } else {
        lp.begin("dense_grad");
        hipLaunchKernelGGL(dense_grad_kernel, dim3((dg.n + 1 + kThreads - 1) / kThreads, ndchunks), dim3(kThreads), 0, ss, dg);
        lp.end();
      }
    }
    if (!inline_side) {
      HIPCHK(hipEventRecord(c->ev_join, c->side));
      c->side_pending = true;
    }
  }
  return MWW_OK;
}

// sync-BN: collapse this rank's partials, sum them over the ranks through the caller's hook, and hand
// the result to the finalize kernel as a single "partial" row.  Returns the pointer / row count /
// element count the finalize kernel should use.
struct StatSource { const float* part; int G; float inv_n; float dscale; };

int exchange_stats(mww_ctx* c, Launcher& lp, const char* what, int layer, const float* part, int G, int C, int bwd,
                   float local_inv_n, StatSource* out) {
  out->part = part;
  out->G = G;
  out->inv_n = local_inv_n;
  out->dscale = 1.0f;
  if (!(c->hook && c->sync_bn)) return MWW_OK;
  float* buf = c->sync_buf + c->sync_off[layer] + (bwd ? 2 * C : 0);
  StatCollapseArgs a{part, G, C, buf};
  lp.begin(what, layer);
  hipLaunchKernelGGL(stat_collapse_kernel, dim3(C), dim3(kThreads), 0, c->stream, a);
  lp.end();
  if (c->hook(c->hook_user, buf, 2 * C, MWW_EXCHANGE_IN_ORDER) != 0) return fail(MWW_ERR_STATE, "all-reduce hook failed");
  out->part = buf;
  out->G = 1;
  out->inv_n = local_inv_n / (float)c->world;
  out->dscale = 1.0f / (float)c->world;
  return MWW_OK;
}

int g_enqueue_forward(mww_ctx* c, int B, bool training, bool update_moving, bool loss, bool metrics);
int g_enqueue_backward(mww_ctx* c, int B, bool fuse_adam);

// Workgroups of one forward block launch: its (window, time tile) items over at most the workgroups the instantiation
// keeps resident (the __launch_bounds__ of fwd_block_kernel), so that no launch runs a partial second dispatch round.
int fwd_block_grid(const mww_ctx* c, const Layer& l, int B) {
  const int per_cu = l.cin > 48 ? 2 : (l.k > 13 ? 3 : 4);
  const long long items = (long long)B * ((l.tout + TT - 1) / TT);
  return (int)std::min<long long>(items, std::min(c->grid_fwd, c->n_cu * per_cu));
}

int enqueue_forward(mww_ctx* c, int B, bool training, bool update_moving, bool loss, bool metrics) {
  if (c->generic) return g_enqueue_forward(c, B, training, update_moving, loss, metrics);
  Launcher lp{c};
  const mww_mixednet_desc& d = c->d;
  const int nb = d.n_blocks;
  if (!training && !c->bn_eval_ready) {
    for (int i = 0; i < nb; ++i) {
      Layer& l = c->L[i];
      BnEvalPrepareArgs a{c->params + l.o_gamma, c->params + l.o_beta, c->bn_state + l.o_mm, c->bn_state + l.o_mv,
                          bn_slot(l, BN_SCALE), bn_slot(l, BN_SHIFT), l.cout};
      lp.begin("bn_eval_prepare", i);
      hipLaunchKernelGGL(bn_eval_prepare_kernel, dim3(1), dim3(64), 0, c->stream, a);
      lp.end();
    }
  }
  // statistics of BN_i: accumulator rows folded by the next kernel, or partial rows + a finalize launch
  const bool inl = training && c->bn_inline && !(c->hook && c->sync_bn);
  auto fold_of = [&](Layer& pl) {
    BnFoldArgs f;
    memset(&f, 0, sizeof(f));
    if (!inl) return f;
    f.acc = pl.facc_cur;
    f.inv_n = 1.0f / ((float)B * (float)pl.tout);
    f.update_moving = update_moving ? 1 : 0;
    f.gamma = c->params + pl.o_gamma;
    f.beta = c->params + pl.o_beta;
    f.moving_mean = c->bn_state + pl.o_mm;
    f.moving_var = c->bn_state + pl.o_mv;
    f.scale = bn_slot(pl, BN_SCALE);
    f.shift = bn_slot(pl, BN_SHIFT);
    f.mean = bn_slot(pl, BN_MEAN);
    f.rstd = bn_slot(pl, BN_RSTD);
    return f;
  };
  for (int i = 0; i < nb; ++i) {
    Layer& l = c->L[i];
    const int grid = i == 0 ? std::min(B, c->grid_fwd) : fwd_block_grid(c, l, B);
    StatAcc sacc{nullptr, nullptr};
    if (inl) {
      sacc.acc = l.facc[c->fpar];
      sacc.clear = l.facc[c->fpar ^ 1];
      l.facc_cur = sacc.acc;
    }
    if (i == 0) {
      FwdFirstArgs a{c->x, c->params + c->o_conv1, c->params + l.o_dw_w, c->params + l.o_dw_b, c->params + l.o_pw_w,
                     l.p, l.stat_part, B, d.frames, l.tout, 0, sacc, x_gather(c), training ? c->a0 : nullptr};
      lp.begin("fwd_block", i);
      int rc = launch_fwd_first(c, d.conv1_kernel, d.conv1_filters, l.cout, l.k, d.conv1_stride, a, grid);
      lp.end();
      if (rc) return rc;
    } else {
      Layer& pl = c->L[i - 1];
      FwdBlockArgs a{pl.p, bn_slot(pl, BN_SCALE), bn_slot(pl, BN_SHIFT), c->params + l.o_dw_w, c->params + l.o_dw_b,
                     c->params + l.o_pw_w, l.p, l.stat_part, B, l.tin, l.tout, c->ablate, c->phase_clk + (size_t)(2 * i) * 2048 * kClkSlots,
                     sacc, fold_of(pl)};
      lp.begin("fwd_block", i);
      int rc = launch_fwd_block(c, l.cin, l.cout, l.k, a, grid);
      lp.end();
      if (rc) return rc;
    }
    if (training && !inl) {
      StatSource ss;
      int rcs = exchange_stats(c, lp, "bn_stat_exchange", i, l.stat_part, grid, l.cout, 0, 1.0f / ((float)B * (float)l.tout), &ss);
      if (rcs) return rcs;
      BnFwdFinalizeArgs f{ss.part, ss.G, l.cout, ss.inv_n, c->params + l.o_gamma,
                          c->params + l.o_beta, c->bn_state + l.o_mm, c->bn_state + l.o_mv, bn_slot(l, BN_SCALE),
                          bn_slot(l, BN_SHIFT), bn_slot(l, BN_MEAN), bn_slot(l, BN_RSTD), update_moving ? 1 : 0};
      lp.begin("bn_fwd_finalize", i);
      hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(l.cout), dim3(kThreads), 0, c->stream, f);
      lp.end();
    }
  }
  Layer& ll = c->L[nb - 1];
  const int ghead = std::min(B, c->grid_head);
  HeadArgs h;
  h.p = ll.p;
  h.scale = bn_slot(ll, BN_SCALE);
  h.shift = bn_slot(ll, BN_SHIFT);
  h.mean = bn_slot(ll, BN_MEAN);
  h.rstd = bn_slot(ll, BN_RSTD);
  h.wd = c->params + c->o_dense_w;
  h.bd = c->params + c->o_dense_b;
  h.y = (loss || metrics) ? c->y_cur : nullptr;
  h.sw = c->sw_cur;
  h.z = c->z;
  h.prob = c->prob;
  h.dz = c->dz;
  h.loss_part = c->loss_part;
  h.gstat_part = ll.gstat_part;
  h.B = B;
  h.T = ll.tout;
  h.inv_b = 1.0f / (float)B;
  h.training = (loss ? kHeadTraining : 0) | (c->bce_clipped ? kHeadClippedLoss : 0);
  h.fold = fold_of(ll);
  if (inl) c->fpar ^= 1;
  // train step with the statistics hand-over: BN_L's backward sums go to accumulator rows (folded by the last block's
  // backward kernel) and the dense-weight gradient / metric update ride in the gradient-reduction launch
  const bool tail_late = loss && inl && c->tail_roles;
  h.gacc = StatAcc{nullptr, nullptr};
  if (tail_late) {
    h.gacc.acc = ll.gacc[c->gpar];
    h.gacc.clear = ll.gacc[c->gpar ^ 1];
    ll.gacc_cur = h.gacc.acc;
  }
  const int q = ll.cout / 4, nrg = kThreads / q;
  lp.begin("head");
  int rc = launch_head(c, ll.cout, (ll.tout + nrg - 1) / nrg, h, ghead);
  lp.end();
  if (rc) return rc;
  if (tail_late) {
    c->tail_in_reduce = true;
    c->tail_metrics = metrics;
    return MWW_OK;
  }
  if (loss && !(c->hook && c->sync_bn)) {
    // train step: the dense-weight gradient and the metric update share the launch of the last block's
    // BN-backward finalize (head_tail_kernel, first thing in enqueue_backward)
    c->tail_pending = true;
    c->tail_metrics = metrics;
    return MWW_OK;
  }
  return enqueue_side_work(c, B, metrics, loss, ll.p, bn_slot(ll, BN_SCALE), bn_slot(ll, BN_SHIFT), nullptr);
}

// gradient assembly: fixed-order sum of the per-workgroup partials (+ the dense layer's, which come
// from the side stream), structural mask, optionally fused with the Adam update
int enqueue_adam(mww_ctx* c);

// Gradient assembly of the parameter range [lo, hi): one grad_final_kernel launch (kernels_tail.hip.h) finishes every
// parameter of the range - fixed-order sums of the partial rows listed in `ga` (and of the rows the dense / metric
// roles produce when they ride along), the values the folding kernels already wrote (BN gamma / beta), zeros for
// parameters nothing contributes to - and applies the mask, and Adam when `apply_adam`.
int assemble_range(mww_ctx* c, int B, const GradReduceArgs& ga, int64_t lo, int64_t hi, bool tail_dense, bool metrics, bool apply_adam) {
  Launcher lp{c};
  std::vector<FinalSegment> segs;
  for (int i = 0; i < ga.nseg; ++i) {
    const GradSegment& g = ga.seg[i];
    if (g.dst < lo || g.dst >= hi) continue;
    if (g.dst + g.n > hi) return fail(MWW_ERR_STATE, "gradient segment straddles a bucket boundary");
    segs.push_back(FinalSegment{g.part, g.G, g.stride, g.n, g.dst, kSegPartials, 0});
  }
  GradFinalArgs a;
  memset(&a, 0, sizeof(a));
  if (tail_dense && c->o_dense_w >= lo && c->o_dense_w < hi) {
    Layer& ll = c->L[c->d.n_blocks - 1];
    a.dense = DenseGradArgs{ll.p, bn_slot(ll, BN_SCALE), bn_slot(ll, BN_SHIFT), c->dz, nullptr, B, c->t_last * c->c_last,
                            c->c_last, 0, (B + kDenseChunks - 1) / kDenseChunks, nullptr, nullptr, nullptr, nullptr, 0, 0, c->st_bf16 ? 1 : 0};
    segs.push_back(FinalSegment{nullptr, B, 0, c->t_last * c->c_last + 1, (int)c->o_dense_w, kSegDense, 0});
  }
  std::sort(segs.begin(), segs.end(), [](const FinalSegment& x, const FinalSegment& y) { return x.dst < y.dst; });
  // the gaps between the segments: runs of parameters that are final in grad[] already ("direct") or untouched
  std::vector<FinalSegment> all;
  int64_t cur = lo;
  auto fill = [&](int64_t upto) {
    while (cur < upto) {
      const bool dir = c->direct_host[(size_t)cur] != 0;
      int64_t e = cur;
      while (e < upto && (c->direct_host[(size_t)e] != 0) == dir) ++e;
      all.push_back(FinalSegment{nullptr, 1, 0, (int)(e - cur), (int)cur, dir ? kSegDirect : kSegZero, 0});
      cur = e;
    }
  };
  for (const FinalSegment& sgm : segs) {
    if (sgm.dst < cur) return fail(MWW_ERR_STATE, "overlapping gradient segments");
    fill(sgm.dst);
    all.push_back(sgm);
    cur = sgm.dst + sgm.n;
  }
  fill(hi);
  a.met = MetricsArgs{c->prob, c->y_cur, c->metrics, B, c->bce_clipped ? nullptr : c->z};
  a.mask = c->mask;
  a.grad = c->grads;
  a.scale = 1.0f;
  a.adam = AdamArgs{c->params, c->grads, c->adam_m, c->adam_v, mail_hyper(c), (int)c->P, 0.9f, 0.999f, 1e-7f};
  a.apply_adam = apply_adam ? 1 : 0;
  for (size_t first = 0; first < all.size() || (metrics && first == 0); first += kMaxFinalSegments) {
    const int n = (int)std::min<size_t>(kMaxFinalSegments, all.size() - first);
    // workgroups are dispatched in block order: the longest role (the dense kernel's gradient: B strided rows of p_L per
    // parameter) takes the first blocks, so that it starts first
    int nb = 0, ns = 0;
    for (int pass = 0; pass < 2; ++pass)
      for (int i = 0; i < n; ++i) {
        const FinalSegment& sg = all[first + i];
        if ((sg.kind == kSegDense) != (pass == 0)) continue;
        a.seg[ns] = sg;
        a.seg[ns].block0 = nb;
        a.block0[ns] = nb;
        nb += (sg.n + kFinalCols - 1) / kFinalCols;
        ++ns;
      }
    for (int i = ns; i < kMaxFinalSegments; ++i) a.block0[i] = INT_MAX;
    a.nseg = n;
    a.nblocks = nb;
    a.do_metrics = (metrics && first == 0) ? 1 : 0;
    if (nb + a.do_metrics == 0) break;
    c->metric_launches += a.do_metrics;
    lp.begin(apply_adam ? "grad_final+adam" : "grad_final");
    hipLaunchKernelGGL(grad_final_kernel, dim3(nb + a.do_metrics), dim3(kThreads), 0, c->stream, a);
    lp.end();
    if (all.empty()) break;
  }
  return MWW_OK;
}

// gradient exchange of a finished range through the caller's hook (data-parallel step)
int exchange_range(mww_ctx* c, int64_t lo, int64_t hi, int flags) {
  if (c->hook(c->hook_user, flags == MWW_EXCHANGE_FLUSH ? nullptr : c->grads + lo, flags == MWW_EXCHANGE_FLUSH ? 0 : hi - lo, flags) != 0)
    return fail(MWW_ERR_STATE, "all-reduce hook failed");
  return MWW_OK;
}

int enqueue_grad_assembly(mww_ctx* c, int B, GradReduceArgs& ga, bool fuse_adam, int64_t lo = 0, int64_t hi = -1, bool last_range = true) {
  if (hi < 0) hi = c->P;
  int rcj = join_side(c);
  if (rcj) return rcj;
  const bool tail_here = c->tail_in_reduce && c->o_dense_w >= lo && c->o_dense_w < hi;
  if (tail_here) c->tail_in_reduce = false;
  if (!tail_here && c->o_dense_w >= lo && c->o_dense_w < hi) {
    // the dense-weight gradient came as batch-chunk rows from dense_grad_kernel / head_tail_kernel
    const int dchunk = (B + kDenseChunks - 1) / kDenseChunks;
    GradSegment s;
    s.part = c->dwd_part;
    s.G = (B + dchunk - 1) / dchunk;
    s.stride = c->dwd_stride;
    s.n = c->t_last * c->c_last + 1;
    s.dst = (int)c->o_dense_w;
    ga.seg[ga.nseg++] = s;
  }
  const bool exchange = fuse_adam && c->hook && c->reduce_grads;
  // single device: Adam rides in the assembly launch; data-parallel: local gradient -> sum over the ranks (the 1/W
  // factor travels as the gradient scale next to the step size, see mww_train_step) -> Adam on the average
  int rc = assemble_range(c, B, ga, lo, hi, tail_here, tail_here && c->tail_metrics, fuse_adam && !exchange);
  if (rc || !exchange) return rc;
  rc = exchange_range(c, lo, hi, last_range ? MWW_EXCHANGE_IN_ORDER : MWW_EXCHANGE_DEFERRED);
  if (!last_range) c->exchange_pending = true;
  if (rc || !last_range) return rc;
  if (c->exchange_pending) {
    rc = exchange_range(c, 0, 0, MWW_EXCHANGE_FLUSH);
    c->exchange_pending = false;
    if (rc) return rc;
  }
  return enqueue_adam(c);
}

// the weight-gradient partial rows of blocks [b0, b1)
void block_segments(mww_ctx* c, int gbwd, int b0, int b1, GradReduceArgs* ga) {
  memset(ga, 0, sizeof(*ga));
  for (int i = b0; i < b1; ++i) {
    Layer& l = c->L[i];
    GradSegment s;
    s.part = l.grad_part;
    s.G = gbwd;
    s.stride = l.grad_part_stride;
    s.n = l.grad_part_stride;
    s.dst = (int)(i == 0 ? c->o_conv1 : l.o_dw_w);
    ga->seg[ga->nseg++] = s;
  }
}

int enqueue_backward(mww_ctx* c, int B, bool fuse_adam) {
  if (c->generic) return g_enqueue_backward(c, B, fuse_adam);
  Launcher lp{c};
  const mww_mixednet_desc& d = c->d;
  const int nb = d.n_blocks;
  const int gbwd = std::min(B, c->grid_bwd);
  const int ghead = std::min(B, c->grid_head);
  const bool inl = c->bn_inline && !(c->hook && c->sync_bn);
  // data-parallel step: the gradient of [blocks >= split, dense] (a contiguous tail of the flat vector) is final once
  // block `split`'s backward kernel is enqueued; it is assembled and handed to the exchange hook there, so that the
  // all-reduce runs next to the remaining backward kernels.  Needs the statistics hand-over (the BN
  // gamma / beta gradients of a block are then written by that block's own backward kernel).
  const int split = nb >= 3 ? nb - 2 : 0;
  const bool bucketed = fuse_adam && c->hook && c->reduce_grads && !c->sync_bn && inl && c->tail_in_reduce && c->grad_buckets == 2 && split > 0;
  for (int i = nb - 1; i >= 0; --i) {
    Layer& l = c->L[i];
    const bool last = (i == nb - 1);
    // BN_i's backward sums: the last block's come from the head kernel's partial rows (folded by head_tail);
    // the others arrive in accumulator rows and are folded by this block's backward kernel
    const bool fold_here = inl && (!last || c->tail_in_reduce);
    StatSource ss{nullptr, 0, 0.f, 1.0f};
    if (!fold_here) {
      int rcs = exchange_stats(c, lp, "bn_gstat_exchange", i, l.gstat_part, last ? ghead : gbwd, l.cout, 1,
                               1.0f / ((float)B * (float)l.tout), &ss);
      if (rcs) return rcs;
    }
    BnBwdFinalizeArgs f{ss.part, ss.G, l.cout, ss.inv_n,
                        c->params + l.o_gamma, bn_slot(l, BN_RSTD), bn_slot(l, BN_C1), bn_slot(l, BN_MG),
                        bn_slot(l, BN_MGX), c->grads + l.o_gamma, c->grads + l.o_beta, ss.dscale};
    BnGradFoldArgs gf;
    memset(&gf, 0, sizeof(gf));
    if (fold_here) {
      gf.acc = l.gacc_cur;
      gf.inv_n = 1.0f / ((float)B * (float)l.tout);
      gf.dscale = 1.0f;
      gf.gamma = c->params + l.o_gamma;
      gf.c1 = bn_slot(l, BN_C1);
      gf.mg = bn_slot(l, BN_MG);
      gf.mgx = bn_slot(l, BN_MGX);
      gf.dgamma = c->grads + l.o_gamma;
      gf.dbeta = c->grads + l.o_beta;
    }
    if (fold_here) {
      // no launch
    } else if (last && c->tail_pending) {
      c->tail_pending = false;
      const int dchunk = (B + kDenseChunks - 1) / kDenseChunks;
      HeadTailArgs ht;
      ht.fin = f;
      ht.dense = DenseGradArgs{l.p, bn_slot(l, BN_SCALE), bn_slot(l, BN_SHIFT), c->dz, c->dwd_part, B, c->t_last * c->c_last,
                               c->c_last, c->dwd_stride, dchunk, nullptr, nullptr, nullptr, nullptr, 0, 0, c->st_bf16 ? 1 : 0};
      ht.met = MetricsArgs{c->prob, c->y_cur, c->metrics, B, c->bce_clipped ? nullptr : c->z};
      ht.n_fin = l.cout;
      ht.ndx = (ht.dense.n + 1 + kThreads - 1) / kThreads;
      ht.ndy = (B + dchunk - 1) / dchunk;
      ht.do_metrics = c->tail_metrics ? 1 : 0;
      c->metric_launches += ht.do_metrics;
      lp.begin("head_tail");
      hipLaunchKernelGGL(head_tail_kernel, dim3(ht.n_fin + ht.ndx * ht.ndy + ht.do_metrics), dim3(kThreads), 0, c->stream, ht);
      lp.end();
    } else {
      lp.begin("bn_bwd_finalize", i);
      hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(l.cout), dim3(kThreads), 0, c->stream, f);
      lp.end();
    }
    if (i > 0) {
      Layer& pl = c->L[i - 1];
      BwdBlockArgs a;
      a.in = pl.p;
      a.in_scale = bn_slot(pl, BN_SCALE);
      a.in_shift = bn_slot(pl, BN_SHIFT);
      a.in_mean = bn_slot(pl, BN_MEAN);
      a.in_rstd = bn_slot(pl, BN_RSTD);
      a.pk = l.p;
      a.gk = l.g;
      a.k_mean = bn_slot(l, BN_MEAN);
      a.k_rstd = bn_slot(l, BN_RSTD);
      a.k_c1 = bn_slot(l, BN_C1);
      a.k_mg = bn_slot(l, BN_MG);
      a.k_mgx = bn_slot(l, BN_MGX);
      a.k_scale = bn_slot(l, BN_SCALE);
      a.k_shift = bn_slot(l, BN_SHIFT);
      a.wd = c->params + c->o_dense_w;
      a.dz = c->dz;
      a.dw_w = c->params + l.o_dw_w;
      a.dw_b = c->params + l.o_dw_b;
      a.pw_w = c->params + l.o_pw_w;
      a.g_out = pl.g;
      a.gstat_part = pl.gstat_part;
      a.grad_part = l.grad_part;
      a.B = B;
      a.Tin = l.tin;
      a.Tout = l.tout;
      a.ablate = c->ablate;
      a.phase_clk = c->phase_clk + (size_t)(2 * i + 1) * 2048 * kClkSlots;
      a.gacc = StatAcc{nullptr, nullptr};
      if (inl) {
        a.gacc.acc = pl.gacc[c->gpar];
        a.gacc.clear = pl.gacc[c->gpar ^ 1];
        pl.gacc_cur = a.gacc.acc;
      }
      a.gfold = gf;
      lp.begin("bwd_block", i);
      int rc = launch_bwd_block(c, l.cin, l.cout, l.k, last, a, gbwd);
      lp.end();
      if (rc) return rc;
      if (bucketed && i == split) {
        GradReduceArgs gb;
        block_segments(c, gbwd, split, nb, &gb);
        rc = enqueue_grad_assembly(c, B, gb, fuse_adam, l.o_dw_w, c->P, false);
        if (rc) return rc;
      }
    } else {
      if (last) return fail(MWW_ERR_UNSUPPORTED, "single-block models are not supported");
      BwdFirstArgs a{c->x, c->a0, l.p, l.g, bn_slot(l, BN_MEAN), bn_slot(l, BN_RSTD), bn_slot(l, BN_C1),
                     bn_slot(l, BN_MG), bn_slot(l, BN_MGX), c->params + l.o_dw_w, c->params + l.o_dw_b,
                     c->params + l.o_pw_w, l.grad_part, B, d.frames, l.tout, gf, x_gather(c)};
      lp.begin("bwd_block", i);
      int rc = launch_bwd_first(c, d.conv1_kernel, d.conv1_filters, l.cout, l.k, d.conv1_stride, a, gbwd);
      lp.end();
      if (rc) return rc;
    }
  }
  if (inl) c->gpar ^= 1;
  GradReduceArgs ga;
  block_segments(c, gbwd, 0, bucketed ? split : nb, &ga);
  return enqueue_grad_assembly(c, B, ga, fuse_adam, 0, bucketed ? c->L[split].o_dw_w : c->P, true);
}

// ---------------------------------------------------------------------------------- conv/BN graphs
#ifdef MWW_SLIM
#define MWW_G_WIDTHS(X) X(48)
#else
#define MWW_G_WIDTHS(X) X(8) X(10) X(12) X(16) X(20) X(24) X(30) X(32) X(36) X(40) X(48) X(60) X(64)
#endif

bool g_width_supported(int n) {
#define X(N) if (n == N) return true;
  MWW_G_WIDTHS(X)
#undef X
  return false;
}

// gfx950 has 160 KB of LDS per CU; tiles above the 64 KB default need the function attribute
constexpr size_t kMaxDynLds = 144 * 1024;

// Dynamic LDS of the MFMA graph kernels (kernels_graph.hip.h) for an op whose tiles hold rin input rows / rout output rows
// (forward naming; the whole window, or a frame chunk of a 1x1 op): weights [k][cin4][NCW] zero-padded to whole k-steps /
// filter tiles; gconv_body's publish scratch aliases the first 2 * kThreads floats, its MODE 1 keeps the statistics pairs of
// the second and third source behind the tiles.
size_t g_up4(int v) { return (size_t)((v + 3) & ~3); }
size_t g_up16(int v) { return (size_t)((v + 15) / 16 * 16); }
size_t g_lds_body(size_t tiles, int pairs) { return (std::max(tiles, (size_t)2 * kThreads) + (size_t)pairs * 2 * kThreads + 4) * sizeof(float); }
size_t g_lds_fwd(const GOp& o, int rin, int rout) {
  return g_lds_body((size_t)o.k * g_up4(o.cin) * g_up16(o.cout) + (size_t)rin * (o.cin | 1) + (size_t)rout * (o.cout | 1), 0);
}
size_t g_lds_dx(const GOp& o, int rows_dp_padded, int rows_dx) {
  return g_lds_body((size_t)o.k * g_up4(o.cout) * g_up16(o.cin) + (size_t)rows_dp_padded * (o.cout | 1) + (size_t)rows_dx * (o.cin | 1), o.n_src - 1);
}
size_t g_lds_wg(const GOp& o, int rin, int rout) {
  const int tasks = o.k * o.cin, mt = (tasks + 15) / 16, nt = (o.cout + 15) / 16;
  size_t b = (((size_t)rin * (o.cin | 1) + 6) / 4 * 4 + g_up4(rout) * (size_t)gwg_dp_pitch(o.cout)) * sizeof(float);
  if (gwg_kparts(tasks) > 1) b = std::max(b, (size_t)gwg_kparts(tasks) * mt * nt * 256 * sizeof(float));   // scratch of the sum over the frame parts
  return b;
}

// Frame chunks ("graph_frame_chunks"; kernels_graph.hip.h, CH instantiations): S work items of Tc output frames per window -
// the 1x1 ops in all three roles, ops with k > 1 in the forward convolution and in a weight gradient that has no data
// gradient next to it (the stem).  0 = whole windows (the default: the chunked kernels are covered by the parity tests but have not been timed
// on the GPU yet), 1 = as many chunks (<= 4) as it takes for the launch's tiles to fit four times per CU, 2..4 = that many.
// Only with the statistics hand-over (such graphs have no residual branches, which the chunked data gradient does not
// handle) and never for twin launches.
// input frames (with halo) of a chunk of t output frames
int g_chunk_in(const GOp& o, int t) { return (t - 1) * o.stride + (o.k - 1) * o.dil + 1; }

int g_chunks(const mww_ctx* c, const GOp& o, bool inl, bool backward, int* Tc) {
  *Tc = o.tout;
  // the data gradient is only chunked without a halo (k = 1); forward convolution and a weight gradient on its own take any k
  if (!inl || c->g_chunks == 0 || o.kind != MWW_OP_CONV || o.tout < 32 || (backward && o.needs_dx && o.k != 1)) return 1;
  int S = c->g_chunks;
  if (S == 1) {
    for (S = 1; S < 4; ++S) {
      const int t = (o.tout + S - 1) / S, ti = g_chunk_in(o, t);
      const size_t lds = backward ? std::max(g_lds_wg(o, ti, t), o.needs_dx ? g_lds_dx(o, t, t) : 0) : g_lds_fwd(o, ti, t);
      if (lds + 3072 <= 40960) break;
    }
  }
  S = std::min(S, 4);
  *Tc = (o.tout + S - 1) / S;
  return (S - 1) * *Tc < o.tout ? S : 1;   // (every chunk non-empty)
}

// Workgroups per role of a conv/BN graph launch.  The kernels are latency-bound (one wave per SIMD and workgroup, ~15
// cycles per issued instruction), so a launch wants as many resident workgroups as its own LDS tile and registers let a
// CU hold - and no more: a workgroup that has to wait for a free slot costs more than it brings.  With one grid for the
// whole step (3 workgroups per CU, the best single value) the 48-channel ops, whose tiles fit twice, ran a third of their
// workgroups as a second round, and the 10- and 16-channel ops left half of the CU's wave slots empty.  Same-session
// sweeps of the Inception step (B = 1024, tools/gpu_knobs.sh): one grid of 768 = 0.993 ms; per launch
// n_cu x min(occupancy, cap) with caps forward / backward 4 / 2 = 1.035, 4 / 3 = 0.941, 3 / 4 = 0.96 (cap 3 forward),
// 4 / 4 = 0.884 (default), 8 / 4 = 0.882, 4 / 5 with the 10-channel backward kernels compiled for five waves = 0.881 (not
// kept); rounding a role's workgroups down to the fewest that keep the number of windows per workgroup = 0.981 (the
// workgroups with one window fewer leave the CU early: fewer, evenly loaded ones are slower).
// `fixed` > 0 (no statistics hand-over: the partial statistics rows of a tensor are shared by all its launches; or
// "grid_graph" set by the caller) keeps the given grid.
struct GridPick {
  int fixed;          // workgroups per role, or 0: choose
  int B, roles, cap;  // windows; roles sharing the launch's workgroups; workgroups per CU at most
  int* used;          // out: workgroups per role
};

int g_role_grid(mww_ctx* c, const void* func, size_t lds, const GridPick& pk) {
  int grid = pk.fixed;
  if (grid <= 0) {
    const auto key = std::make_pair(func, lds);
    auto it = c->g_occ.find(key);
    if (it == c->g_occ.end()) {
      int occ = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, func, kThreads, lds) != hipSuccess || occ < 1) occ = 3;
      it = c->g_occ.emplace(key, occ).first;
    }
    const int wpc = std::max(1, std::min(it->second, pk.cap));
    grid = std::max(1, std::min(std::min(pk.B, c->n_cu * 4), c->n_cu * wpc / std::max(1, pk.roles)));   // (n_cu * 4 rows of weight-gradient partials)
  }
  if (pk.used) *pk.used = grid;
  return grid;
}

// weight-gradient and data-gradient roles that divide a launch's workgroups (pk.roles > 1) need not take equal halves:
// "graph_dgrad_share" percent of an op's workgroups form the data gradient
void g_share_roles(mww_ctx* c, const GridPick& pk, int* nbw, int* nbd) {
  if (pk.roles > 1 && c->g_dgrad_share != 50) {
    const int pair = *nbw + *nbd;
    *nbd = std::max(1, std::min(pair - 1, (pair * c->g_dgrad_share + 50) / 100));
    *nbw = std::max(1, pair - *nbd);
  }
  if (pk.used) *pk.used = *nbw;
}

// Static shapes (kernels_graph.hip.h GShape): the ops of the reference's default Inception flags (inception.py:146-209:
// 5x1 stem over the 40 spectrogram bins; per block a fused 1x1 head, 5x1 convolutions over channel slices of it and over
// each other, and the 1x1 convolution over the aligned concatenation).  (id, K, sources, C0, LD0, C1, LD1, C2, LD2); dilation
// and stride 1, no residual branches, whole windows.  Any other op takes the run-time kernels.
#ifdef MWW_SLIM
#define MWW_G_SHAPES(X)
#else
#define MWW_G_SHAPES(X)                                                                                                   \
  X(1, 5, 1, 40, 40, 0, 0, 0, 0) X(2, 1, 1, 24, 24, 0, 0, 0, 0) X(3, 5, 1, 10, 30, 0, 0, 0, 0) X(4, 5, 1, 10, 10, 0, 0, 0, 0)    \
  X(5, 1, 3, 10, 30, 10, 10, 10, 10) X(6, 1, 1, 10, 10, 0, 0, 0, 0) X(7, 5, 1, 16, 48, 0, 0, 0, 0) X(8, 5, 1, 16, 16, 0, 0, 0, 0) \
  X(9, 1, 3, 16, 48, 16, 16, 16, 16) X(10, 1, 3, 10, 10, 10, 10, 10, 10) X(11, 1, 3, 16, 16, 16, 16, 16, 16)
#endif
#define X(ID, K, N, C0, L0, C1, L1, C2, L2) typedef GShape<K, N, C0, L0, C1, L1, C2, L2> GSh##ID;
MWW_G_SHAPES(X)
#undef X
// dynamic LDS of a static shape's forward launch: the direct form (gconv_body "DIRECT") has no output tile and narrower weight rows
template <class SH, int NC>
size_t g_lds_fwd_static(size_t lds, const GConvArgs& a) {
  if constexpr (g_fwd_direct<SH, 0>()) return g_lds_body((size_t)g_direct_tiles(SH::K, SH::CIN, NC, a.Tin), 0);
  else return lds;
}
// (shape id, filters) of the forward / weight-gradient instantiations, (id, filters, input channels) of the backward pairs
#ifdef MWW_SLIM
#define MWW_G_SHAPE_FWD(X)
#define MWW_G_SHAPE_FWD2(X)
#define MWW_G_SHAPE_WG(X)
#define MWW_G_SHAPE_XG(X)
#define MWW_G_SHAPE_BWD(X)
#define MWW_G_SHAPE_BWD2(X)
#else
#define MWW_G_SHAPE_FWD(X) X(1, 24) X(2, 30) X(3, 10) X(4, 10) X(5, 10) X(6, 30) X(6, 48) X(7, 16) X(8, 16) X(9, 16) X(10, 10) X(11, 16)
#define MWW_G_SHAPE_FWD2(X) X(3, 10) X(7, 16) X(4, 10) X(8, 16)
#define MWW_G_SHAPE_WG(X) X(1, 24)
#define MWW_G_SHAPE_XG(X) X(1, 24)   // forward + weight gradient with the input gathered from the feature stores (gconv_xg_kernel)
#define MWW_G_SHAPE_BWD(X) X(2, 30, 24) X(3, 10, 10) X(4, 10, 10) X(5, 10, 30) X(6, 30, 10) X(6, 48, 10) X(7, 16, 16) X(8, 16, 16) X(9, 16, 48) X(10, 10, 30) X(11, 16, 48)
#define MWW_G_SHAPE_BWD2(X) X(3, 10) X(7, 16) X(4, 10) X(8, 16)
#endif

// planes of op `o`'s tensors in effect (1 = interleaved) and the distance between two planes in floats
int g_planes(const mww_ctx* c, const GOp& o) { return (c->g_planar && o.planes > 1) ? o.planes : 1; }
// (+ kPlanePad floats: without it two planes lie a multiple of 4-8 KB apart - max_batch x T x pc x 4 bytes - and twin ops that
// walk their planes in step hit the same HBM channels: the 16-channel twin backward launch went 45 -> 55 us)
constexpr long long kPlanePad = 1088;   // 17 x 256 bytes
long long g_pstride(const mww_ctx* c, const GOp& o) { return (long long)c->d.max_batch * o.tout * o.pc + kPlanePad; }

// the static shape of op `o`, or 0
int g_shape_id(const mww_ctx* c, const GOp& o) {
  if (!c->g_static || o.kind != MWW_OP_CONV || o.dil != 1 || o.stride != 1 || o.res_src >= 0 || o.n_src < 1) return 0;
  if (o.tin > kGTmax || o.tout > kGTmax) return 0;   // a window's rows travel in a fixed set of registers (GSliceRegs)
  int C[kGMaxSrc] = {0, 0, 0}, L[kGMaxSrc] = {0, 0, 0};
  for (int i = 0; i < o.n_src; ++i) {
    if (o.src[i] < 0) {
      C[i] = L[i] = MWW_FEATURE_BINS;
    } else {
      const GOp& pr = c->G[o.src[i]];
      if (pr.res_src >= 0) return 0;
      C[i] = o.scn[i];
      L[i] = g_planes(c, pr) > 1 ? o.scn[i] : pr.cout;   // (a plane of a planar producer is a whole tensor of its own)
    }
    const int v = ((C[i] | L[i]) & 3) == 0 ? 4 : (((C[i] | L[i]) & 1) == 0 ? 2 : 1);
    if (o.src[i] >= 0 && L[i] != C[i] && o.sc0[i] % v) return 0;   // the slice must start on the vector width the static staging uses
  }
#define X(ID, K, N, C0, L0, C1, L1, C2, L2)                                                                     \
  if (o.k == K && o.n_src == N && C[0] == C0 && L[0] == L0 && C[1] == C1 && L[1] == L1 && C[2] == C2 && L[2] == L2) return ID;
  MWW_G_SHAPES(X)
#undef X
  return 0;
}

// The stem of a conv/BN graph can read a descriptor-only batch in place ("fused_input", kernels_graph.hip.h XG): exactly one
// op reads the spectrogram, as its only source, and its shape has a gathering instantiation.
bool g_stem_gathers(const mww_ctx* c) {
  if (!c->generic || !c->fused_input || c->d.frames > kGXRows) return false;
  int readers = 0, stem = -1;
  for (size_t i = 0; i < c->G.size(); ++i)
    for (int s = 0; s < c->G[i].n_src; ++s)
      if (c->G[i].src[s] < 0) {
        ++readers;
        stem = (int)i;
      }
  if (readers != 1) return false;
  const GOp& o = c->G[stem];
  if (o.n_src != 1 || o.toff[0] != 0 || o.tin != c->d.frames) return false;
  const int shape = g_shape_id(c, o);
#define XS(ID, N) if (shape == ID && o.cout == N) return true;
  MWW_G_SHAPE_XG(XS)
#undef XS
  return false;
}
bool g_reads_lazy_x(const mww_ctx* c, const GSrc* src, int n) {
  if (!c->x_lazy) return false;
  for (int i = 0; i < n; ++i)
    if (src[i].p == c->x) return true;
  return false;
}

// (CH: the frame-chunk instantiations, a.S > 1)
template <int MODE, bool CH = false>
int launch_gconv(mww_ctx* c, int nc, const GConvArgs& a, const GridPick& pk, size_t lds, int shape = 0) {
  if (MODE == 0 && g_reads_lazy_x(c, a.src, a.n_src)) {
    // descriptor-only batch: the gathering instantiation if there is one and the grid leaves every workgroup at most
    // kXMaxSamples windows; else x is written out first
    if constexpr (MODE == 0 && !CH) {
#define XS(ID, N)                                                                                              \
      if (shape == ID && nc == N && a.n_src == 1 && a.Tin <= kGXRows) {                                        \
        auto k = &gconv_xg_kernel<N, GSh##ID>;                                                                 \
        const void* f = reinterpret_cast<const void*>(k);                                                      \
        const size_t ldx = g_lds_fwd_static<GSh##ID, N>(lds, a) + sizeof(XShared) + 16;                        \
        if (ldx > 64 * 1024) HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldx)); \
        const int grid = g_role_grid(c, f, ldx, pk);                                                           \
        if ((a.B + grid - 1) / grid <= kXMaxSamples) {                                                         \
          hipLaunchKernelGGL(k, dim3(grid), dim3(kThreads), ldx, c->stream, a, x_gather(c));                   \
          return MWW_OK;                                                                                       \
        }                                                                                                      \
      }
      MWW_G_SHAPE_XG(XS)
#undef XS
    }
    int rcx = materialise_x(c);
    if (rcx) return rcx;
  }
  if constexpr (MODE == 0 && !CH) {
#define XS(ID, N)                                                                                              \
    if (shape == ID && nc == N) {                                                                              \
      auto k = &gconv_kernel<N, 0, GSh##ID>;                                                                   \
      const void* f = reinterpret_cast<const void*>(k);                                                        \
      const size_t lds_s = g_lds_fwd_static<GSh##ID, N>(lds, a);                                               \
      if (lds_s > 64 * 1024) HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s)); \
      const int grid = g_role_grid(c, f, lds_s, pk);                                                           \
      hipLaunchKernelGGL(k, dim3(grid), dim3(kThreads), lds_s, c->stream, a);                                  \
      return MWW_OK;                                                                                           \
    }
    MWW_G_SHAPE_FWD(XS)
#undef XS
  }
#define X(N)                                                                                                   \
  if (nc == N) {                                                                                               \
    auto k = CH ? &gconv_chunk_kernel<N, MODE> : &gconv_kernel<N, MODE>;                                       \
    const void* f = reinterpret_cast<const void*>(k);                                                          \
    if (lds > 64 * 1024) HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    const int grid = g_role_grid(c, f, lds, pk);                                                               \
    hipLaunchKernelGGL(k, dim3(grid), dim3(kThreads), lds, c->stream, a);                                      \
    return MWW_OK;                                                                                             \
  }
  MWW_G_WIDTHS(X)
#undef X
  return fail(MWW_ERR_UNSUPPORTED, "conv width not instantiated");
}

template <bool CH = false>
int launch_gwgrad(mww_ctx* c, int nc, const GWgradArgs& a, const GridPick& pk, size_t lds, int shape = 0) {
  if (g_reads_lazy_x(c, a.src, a.n_src)) {   // (as in launch_gconv)
    if constexpr (!CH) {
#define XS(ID, N)                                                                                              \
      if (shape == ID && nc == N && a.n_src == 1 && a.Tin <= kGXRows) {                                        \
        auto k = &gconv_wgrad_xg_kernel<N, GSh##ID>;                                                           \
        const void* f = reinterpret_cast<const void*>(k);                                                      \
        const size_t narrow = MWW_G_WGRAD_XG_NARROW ? g_up4(a.Tout) * (size_t)(gwg_dp_pitch(N) - (N + 7) / 8 * 8) * sizeof(float) : 0; \
        const size_t ldx = lds - narrow + sizeof(XShared) + 16;                                                \
        if (ldx > 64 * 1024) HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldx)); \
        const int grid = g_role_grid(c, f, ldx, pk);                                                           \
        if ((a.B + grid - 1) / grid <= kXMaxSamples) {                                                         \
          hipLaunchKernelGGL(k, dim3(grid), dim3(kThreads), ldx, c->stream, a, x_gather(c));                   \
          return MWW_OK;                                                                                       \
        }                                                                                                      \
      }
      MWW_G_SHAPE_XG(XS)
#undef XS
    }
    int rcx = materialise_x(c);
    if (rcx) return rcx;
  }
  if constexpr (!CH) {
#define XS(ID, N)                                                                                              \
    if (shape == ID && nc == N) {                                                                              \
      auto k = &gconv_wgrad_kernel<N, GSh##ID>;                                                                \
      const void* f = reinterpret_cast<const void*>(k);                                                        \
      if (lds > 64 * 1024) HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      const int grid = g_role_grid(c, f, lds, pk);                                                             \
      hipLaunchKernelGGL(k, dim3(grid), dim3(kThreads), lds, c->stream, a);                                    \
      return MWW_OK;                                                                                           \
    }
    MWW_G_SHAPE_WG(XS)
#undef XS
  }
#define X(N)                                                                                                   \
  if (nc == N) {                                                                                               \
    auto k = CH ? &gconv_wgrad_chunk_kernel<N> : &gconv_wgrad_kernel<N>;                                       \
    const void* f = reinterpret_cast<const void*>(k);                                                          \
    if (lds > 64 * 1024) HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    const int grid = g_role_grid(c, f, lds, pk);                                                               \
    hipLaunchKernelGGL(k, dim3(grid), dim3(kThreads), lds, c->stream, a);                                      \
    return MWW_OK;                                                                                             \
  }
  MWW_G_WIDTHS(X)
#undef X
  return fail(MWW_ERR_UNSUPPORTED, "conv width not instantiated");
}

// (filters, input channels) pairs with a fused weight-gradient + data-gradient launch; others use two launches
#ifdef MWW_SLIM
#define MWW_G_BWD_PAIRS(X) X(48, 48)
#else
#define MWW_G_BWD_PAIRS(X) X(30, 24) X(10, 10) X(10, 30) X(30, 10) X(48, 10) X(16, 16) X(16, 48) X(24, 16) X(16, 24) X(36, 24) X(12, 36) X(48, 32) X(48, 48) X(64, 32) X(64, 64)
#endif

template <bool CH = false>
bool launch_gbwd_fused(mww_ctx* c, int nco, int nci, const GWgradArgs& w, const GConvArgs& d, const GridPick& pk, size_t lds, int shape = 0) {
  if constexpr (!CH) {
#define XS(ID, NCO, NCI)                                                                                       \
    if (shape == ID && nco == NCO && nci == NCI) {                                                             \
      auto k = &gconv_bwd_kernel<NCO, NCI, GSh##ID>;                                                           \
      const void* f = reinterpret_cast<const void*>(k);                                                        \
      if (lds > 64 * 1024) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      int nbw = g_role_grid(c, f, lds, pk), nbd = nbw;                                                         \
      g_share_roles(c, pk, &nbw, &nbd);                                                                        \
      hipLaunchKernelGGL(k, dim3(nbw + nbd), dim3(kThreads), lds, c->stream, w, d, nbw, nbd);                  \
      return true;                                                                                             \
    }
    MWW_G_SHAPE_BWD(XS)
#undef XS
  }
#define X(NCO, NCI)                                                                                            \
  if (nco == NCO && nci == NCI) {                                                                              \
    auto k = CH ? &gconv_bwd_chunk_kernel<NCO, NCI> : &gconv_bwd_kernel<NCO, NCI>;                             \
    const void* f = reinterpret_cast<const void*>(k);                                                          \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
    int nbw = g_role_grid(c, f, lds, pk), nbd = nbw;                                                           \
    g_share_roles(c, pk, &nbw, &nbd);                                                                          \
    hipLaunchKernelGGL(k, dim3(nbw + nbd), dim3(kThreads), lds, c->stream, w, d, nbw, nbd);                    \
    return true;                                                                                               \
  }
  MWW_G_BWD_PAIRS(X)
#undef X
  return false;
}

#ifdef MWW_SLIM
#define MWW_G_TWIN_WIDTHS(X) X(32)
#else
#define MWW_G_TWIN_WIDTHS(X) X(8) X(10) X(12) X(16) X(20) X(24) X(32)
#endif
bool launch_gfwd2(mww_ctx* c, int nc, const GConvArgs& a0, const GConvArgs& a1, const GridPick& pk, size_t lds, int shape = 0) {
#define XS(ID, N)                                                                                              \
  if (shape == ID && nc == N) {                                                                                \
    const void* f = reinterpret_cast<const void*>(&gconv_fwd2_kernel<N, GSh##ID>);                             \
    const size_t lds_s = std::max(g_lds_fwd_static<GSh##ID, N>(lds, a0), g_lds_fwd_static<GSh##ID, N>(lds, a1)); \
    if (lds_s > 64 * 1024) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s); \
    const int grid = g_role_grid(c, f, lds_s, pk);                                                             \
    hipLaunchKernelGGL((gconv_fwd2_kernel<N, GSh##ID>), dim3(2 * grid), dim3(kThreads), lds_s, c->stream, GConv2Args{{a0, a1}}, grid); \
    return true;                                                                                               \
  }
  MWW_G_SHAPE_FWD2(XS)
#undef XS
#define X(N)                                                                                                   \
  if (nc == N) {                                                                                               \
    const void* f = reinterpret_cast<const void*>(&gconv_fwd2_kernel<N>);                                      \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
    const int grid = g_role_grid(c, f, lds, pk);                                                               \
    hipLaunchKernelGGL((gconv_fwd2_kernel<N>), dim3(2 * grid), dim3(kThreads), lds, c->stream, GConv2Args{{a0, a1}}, grid); \
    return true;                                                                                               \
  }
  MWW_G_TWIN_WIDTHS(X)
#undef X
  return false;
}
bool launch_gbwd2(mww_ctx* c, int nc, const GWgradArgs& w0, const GConvArgs& d0, const GWgradArgs& w1, const GConvArgs& d1,
                  const GridPick& pk, size_t lds, int shape = 0) {
#define XS(ID, N)                                                                                              \
  if (shape == ID && nc == N) {                                                                                \
    const void* f = reinterpret_cast<const void*>(&gconv_bwd2_kernel<N, N, GSh##ID>);                          \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
    int nbw = g_role_grid(c, f, lds, pk), nbd = nbw;                                                           \
    g_share_roles(c, pk, &nbw, &nbd);                                                                          \
    hipLaunchKernelGGL((gconv_bwd2_kernel<N, N, GSh##ID>), dim3(2 * (nbw + nbd)), dim3(kThreads), lds, c->stream, GBwd2Args{{w0, w1}, {d0, d1}}, nbw, nbd); \
    return true;                                                                                               \
  }
  MWW_G_SHAPE_BWD2(XS)
#undef XS
#define X(N)                                                                                                   \
  if (nc == N) {                                                                                               \
    const void* f = reinterpret_cast<const void*>(&gconv_bwd2_kernel<N, N>);                                   \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
    int nbw = g_role_grid(c, f, lds, pk), nbd = nbw;                                                           \
    g_share_roles(c, pk, &nbw, &nbd);                                                                          \
    hipLaunchKernelGGL((gconv_bwd2_kernel<N, N>), dim3(2 * (nbw + nbd)), dim3(kThreads), lds, c->stream, GBwd2Args{{w0, w1}, {d0, d1}}, nbw, nbd); \
    return true;                                                                                               \
  }
  MWW_G_TWIN_WIDTHS(X)
#undef X
  return false;
}

float* gbn_slot(GOp& o, int i) { return o.bn + (size_t)i * o.cout; }

// source i of op `oi` as the kernels see it; `backward` adds the gradient routing flags
GSrc g_make_src(mww_ctx* c, int oi, int i, bool backward, bool inl = false) {
  GOp& o = c->G[oi];
  GSrc s;
  memset(&s, 0, sizeof(s));
  s.toff = o.toff[i];
  if (o.src[i] < 0) {
    s.p = c->x;
    s.T = c->d.frames;
    s.C = s.ld = s.sld = MWW_FEATURE_BINS;
    s.flags = GSRC_IDENTITY;
    return s;
  }
  GOp& pr = c->G[o.src[i]];
  s.p = pr.p;
  if (pr.norm == MWW_NORM_BN) {
    s.scale = gbn_slot(pr, BN_SCALE);
    s.shift = gbn_slot(pr, BN_SHIFT);
    s.mean = gbn_slot(pr, BN_MEAN);
    s.rstd = gbn_slot(pr, BN_RSTD);
  } else {   // a bias (or nothing) instead of a BN: y = p * 1 + bias
    s.scale = c->ones;
    s.shift = pr.norm == MWW_NORM_BIAS ? c->params + pr.o_beta : c->zeros;
    s.mean = c->zeros;
    s.rstd = c->ones;
  }
  s.g = pr.g;
  s.gstat_part = pr.gstat_part;
  s.T = pr.tout;
  s.C = o.scn[i];
  s.ld = s.sld = pr.cout;
  s.c0 = s.scb = o.sc0[i];
  if (g_planes(c, pr) > 1) {
    // the producer's tensors are planar and this slice is one of the planes: whole rows of C channels, BN arrays at the plane
    const long long off = (long long)(o.sc0[i] / pr.pc) * g_pstride(c, pr);
    s.p += off;
    s.g += off;
    s.scale += s.c0;
    s.shift += s.c0;
    s.mean += s.c0;
    s.rstd += s.c0;
    s.ld = s.C;
    s.c0 = 0;
  }
  if (pr.act == MWW_ACT_LINEAR) s.flags |= GSRC_LINEAR;
  if (pr.res_src >= 0) {
    GOp& rr = c->G[pr.res_src];
    s.rp = rr.p;
    s.rscale = gbn_slot(rr, BN_SCALE);
    s.rshift = gbn_slot(rr, BN_SHIFT);
    s.rT = rr.tout;
    s.rdrop = pr.res_drop;
  }
  if (backward) s.flags |= GSRC_GRAD | (o.src_first[i] ? 0 : GSRC_ACCUM) | (o.src_last[i] ? GSRC_STATS : 0);
  if (backward && inl && o.src_last[i]) {   // the slice's backward sums go to the producer's accumulator rows
    s.gacc.acc = pr.gacc[c->gpar];
    s.gacc.clear = pr.gacc[c->gpar ^ 1];
    pr.gacc_cur = s.gacc.acc;
  }
  return s;
}

GBnBwd g_make_bnbwd(mww_ctx* c, GOp& o) {
  GBnBwd y;
  memset(&y, 0, sizeof(y));
  y.g = o.g;
  y.p = o.p;
  if (o.norm != MWW_NORM_BN) {   // dp = g
    y.mean = c->zeros; y.rstd = c->ones; y.c1 = c->ones; y.mg = c->zeros; y.mgx = c->zeros;
  } else {
    y.mean = gbn_slot(o, BN_MEAN); y.rstd = gbn_slot(o, BN_RSTD); y.c1 = gbn_slot(o, BN_C1); y.mg = gbn_slot(o, BN_MG); y.mgx = gbn_slot(o, BN_MGX);
  }
  y.planes = g_planes(c, o);
  y.pc = o.pc;
  y.pstride = g_pstride(c, o);
  return y;
}

GDwArgs g_make_dw(mww_ctx* c, int oi, int B, bool backward, bool inl = false) {
  GOp& o = c->G[oi];
  GDwArgs a;
  memset(&a, 0, sizeof(a));
  a.src = g_make_src(c, oi, 0, backward, inl);
  a.w = c->params + o.o_w;
  a.k = o.k;
  a.C = o.cout;
  a.B = B;
  a.Tin = o.tin;
  a.Tout = o.tout;
  a.out = o.p;
  a.y = g_make_bnbwd(c, o);
  a.grad_part = o.grad_part;
  return a;
}

int g_enqueue_forward(mww_ctx* c, int B, bool training, bool update_moving, bool loss, bool metrics) {
  if (c->x_lazy && !g_stem_gathers(c)) {   // (an option changed since the batch was assembled)
    int rcx = materialise_x(c);
    if (rcx) return rcx;
  }
  Launcher lp{c};
  const int n = (int)c->G.size();
  const int gg = std::min(B, c->grid_g);
  // statistics hand-over instead of finalize launches (kernels_graph.hip.h)
  const bool inl = training && c->bn_inline && c->g_inline_ok && !(c->hook && c->sync_bn) && !c->profile_split;
  const bool pick = inl && c->grid_g_auto;   // per-launch grids (g_role_grid)
  auto leader = [&](int oi) { return (oi > 0 && c->G[oi - 1].twin_next) ? oi - 1 : oi; };   // first op of the launch op oi rides in
  auto fold_of = [&](int pi, bool publish) {
    GOp& pr = c->G[pi];
    GFoldFwd f;
    memset(&f, 0, sizeof(f));
    f.acc = pr.facc_cur;
    f.C = pr.cout;
    f.groups = pr.groups;
    f.inv_n = 1.0f / ((float)B * (float)pr.tout * (float)(pr.groups > 1 ? pr.cout / pr.groups : 1));
    f.publish = publish ? 1 : 0;
    f.update_moving = update_moving ? 1 : 0;
    f.gamma = c->params + pr.o_gamma;
    f.beta = c->params + pr.o_beta;
    f.moving_mean = c->bn_state + pr.o_mm;
    f.moving_var = c->bn_state + pr.o_mv;
    f.scale = gbn_slot(pr, BN_SCALE);
    f.shift = gbn_slot(pr, BN_SHIFT);
    f.mean = gbn_slot(pr, BN_MEAN);
    f.rstd = gbn_slot(pr, BN_RSTD);
    return f;
  };
  for (int i = 0; i < n; ++i) {
    GOp& o = c->G[i];
    if (!training && o.norm == MWW_NORM_BN) {
      GBnEvalArgs e{c->params + o.o_gamma, c->params + o.o_beta, c->bn_state + o.o_mm, c->bn_state + o.o_mv,
                    gbn_slot(o, BN_SCALE), gbn_slot(o, BN_SHIFT), o.cout, o.groups};
      lp.begin("bn_eval_prepare", i);
      hipLaunchKernelGGL(gbn_eval_prepare_kernel, dim3(1), dim3(kThreads), 0, c->stream, e);
      lp.end();
    }
    if (o.kind == MWW_OP_DEPTHWISE) {
      GDwArgs dw = g_make_dw(c, i, B, false);
      if (inl && o.src[0] >= 0 && c->G[o.src[0]].norm == MWW_NORM_BN && c->G[o.src[0]].first_consumer == i)
        dw.fold = fold_of(o.src[0], true);
      lp.begin("dw_fwd", i);
      // (no statistics leave this launch: its grid is free to follow its occupancy even without the hand-over)
      const void* f = reinterpret_cast<const void*>(&gdw_kernel<0>);
      if (o.lds_fwd > 64 * 1024) HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)o.lds_fwd));
      const int gdw = g_role_grid(c, f, o.lds_fwd, GridPick{c->grid_g_auto ? 0 : gg, B, 1, c->g_cap_fwd, nullptr});
      hipLaunchKernelGGL(gdw_kernel<0>, dim3(gdw), dim3(kThreads), o.lds_fwd, c->stream, dw);
      lp.end();
      continue;
    }
    auto fwd_args = [&](int oi) {
      GOp& q = c->G[oi];
      GConvArgs a;
      memset(&a, 0, sizeof(a));
      a.n_src = q.n_src;
      for (int s = 0; s < q.n_src; ++s) a.src[s] = g_make_src(c, oi, s, false);
      a.w = c->params + q.o_w;
      a.k = q.k;
      a.dil = q.dil;
      a.cin = q.cin;
      a.stride = q.stride;
      a.B = B;
      a.Tin = q.tin;
      a.Tout = q.tout;
      a.out = q.p;
      a.out_planes = g_planes(c, q);
      a.out_pc = q.pc;
      a.out_pstride = g_pstride(c, q);
      a.stat_part = (training && q.norm == MWW_NORM_BN) ? q.stat_part : nullptr;
      if (inl) {
        a.sacc.acc = q.facc[c->fpar];
        a.sacc.clear = q.facc[c->fpar ^ 1];
        q.facc_cur = a.sacc.acc;
        for (int s = 0; s < q.n_src; ++s) {
          const int pi = q.src[s];
          if (pi < 0 || c->G[pi].norm != MWW_NORM_BN) continue;   // (no statistics to fold)
          const int fc = c->G[pi].first_consumer;
          if (leader(oi) != leader(fc)) continue;   // a later launch: the arrays were published by the first one
          bool first_ref = true;
          for (int s2 = 0; s2 < s; ++s2) first_ref = first_ref && q.src[s2] != pi;
          a.fold[s] = fold_of(pi, oi == fc && first_ref);
        }
      }
      return a;
    };
    auto fin_args = [&](int oi, const StatSource& ss) {
      GOp& q = c->G[oi];
      return GBnFwdArgs{ss.part, ss.G, q.cout, q.groups, ss.inv_n,
                        c->params + q.o_gamma, c->params + q.o_beta, c->bn_state + q.o_mm, c->bn_state + q.o_mv,
                        gbn_slot(q, BN_SCALE), gbn_slot(q, BN_SHIFT), gbn_slot(q, BN_MEAN), gbn_slot(q, BN_RSTD), update_moving ? 1 : 0};
    };
    const bool sync = c->hook && c->sync_bn;
    if (o.twin_next && !sync && !c->profile_split) {
      // twins: one convolution launch and one finalize launch for the pair
      GOp& o2 = c->G[i + 1];
      if (!training) {
        GBnEvalArgs e{c->params + o2.o_gamma, c->params + o2.o_beta, c->bn_state + o2.o_mm, c->bn_state + o2.o_mv,
                      gbn_slot(o2, BN_SCALE), gbn_slot(o2, BN_SHIFT), o2.cout, o2.groups};
        hipLaunchKernelGGL(gbn_eval_prepare_kernel, dim3(1), dim3(kThreads), 0, c->stream, e);
      }
      const GConvArgs fa0 = fwd_args(i), fa1 = fwd_args(i + 1);
      lp.begin("conv_fwd2_", i);
      const bool split2 = inl && c->g_role_split;
      const bool ok = launch_gfwd2(c, o.cout, fa0, fa1, GridPick{pick ? 0 : (split2 ? std::max(1, gg / 2) : gg), B, split2 ? 2 : 1, c->g_cap_fwd, nullptr},
                                   std::max(o.lds_fwd, o2.lds_fwd), g_shape_id(c, o) == g_shape_id(c, o2) ? g_shape_id(c, o) : 0);
      lp.end();
      if (ok) {
        if (training && !inl) {
          const float inv_n = 1.0f / ((float)B * (float)o.tout * (float)(o.groups > 1 ? o.cout / o.groups : 1));
          StatSource s0{o.stat_part, gg, inv_n, 1.0f}, s1{o2.stat_part, gg, inv_n, 1.0f};
          const GBnFwdArgs f0 = fin_args(i, s0), f1 = fin_args(i + 1, s1);
          const int n0 = o.slots;
          lp.begin("bn_fwd_finalize2_", i);
          hipLaunchKernelGGL(gbn_fwd_finalize2_kernel, dim3(o.slots + o2.slots), dim3(kThreads), 0, c->stream, f0, f1, n0);
          lp.end();
        }
        ++i;   // the twin is done
        continue;
      }
      if (c->profile) {   // width not instantiated: nothing was launched, fall through to the single-op route
        (void)hipEventDestroy(c->prof.back().a);
        (void)hipEventDestroy(c->prof.back().b);
        c->prof.pop_back();
      }
    }
    GConvArgs fa = fwd_args(i);
    int Tc = 0;
    const int S = g_chunks(c, o, inl, false, &Tc);
    lp.begin("conv_fwd", i);
    int rc;
    if (S > 1) {
      fa.S = S;
      fa.Tc = Tc;
      rc = launch_gconv<0, true>(c, o.cout, fa, GridPick{pick ? 0 : gg, B * S, 1, c->g_cap_fwd, nullptr}, g_lds_fwd(o, g_chunk_in(o, Tc), Tc));
    } else {
      rc = launch_gconv<0>(c, o.cout, fa, GridPick{pick ? 0 : gg, B, 1, c->g_cap_fwd, nullptr}, o.lds_fwd, g_shape_id(c, o));
    }
    lp.end();
    if (rc) return rc;
    if (training && o.norm == MWW_NORM_BN && !inl) {
      const int members = o.groups > 1 ? o.cout / o.groups : 1;
      StatSource ss;
      int rcs = exchange_stats(c, lp, "bn_stat_exchange", i, o.stat_part, gg, o.cout, 0,
                               1.0f / ((float)B * (float)o.tout * (float)members), &ss);
      if (rcs) return rcs;
      const GBnFwdArgs f = fin_args(i, ss);
      lp.begin("bn_fwd_finalize", i);
      hipLaunchKernelGGL(gbn_fwd_finalize_kernel, dim3(o.slots), dim3(kThreads), 0, c->stream, f);
      lp.end();
    }
  }
  GOp& lo = c->G[n - 1];
  const bool drop = loss && c->dropout > 0.f;   // Dropout is active in the train step only (Keras training=True)
  const bool gen_inline = drop && !c->keep_explicit && !c->head2;   // ghead_kernel draws the mask itself
  if (drop && !c->keep_explicit && !gen_inline) {
    const long long ne = (long long)B * c->t_last * c->c_last;
    DropoutMaskArgs dm{c->keep, ne, c->dropout_seed, reinterpret_cast<const unsigned*>(mail_hyper(c)) + 2, c->dropout};
    lp.begin("dropout_mask");
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((ne + kThreads - 1) / kThreads)), dim3(kThreads), 0, c->stream, dm);
    lp.end();
  }
  const int ghead = std::min(B, c->grid_head);
  GHeadArgs h;
  memset(&h, 0, sizeof(h));
  h.p = lo.p;
  h.scale = gbn_slot(lo, BN_SCALE);
  h.shift = gbn_slot(lo, BN_SHIFT);
  h.mean = gbn_slot(lo, BN_MEAN);
  h.rstd = gbn_slot(lo, BN_RSTD);
  h.wd = c->params + c->o_dense_w;
  h.bd = c->params + c->o_dense_b;
  h.y = (loss || metrics) ? c->y_cur : nullptr;
  h.sw = c->sw_cur;
  h.keep = (drop && !gen_inline) ? c->keep : nullptr;
  if (gen_inline) {
    h.keep_gen = c->keep;
    h.seed = c->dropout_seed;
    h.counter = reinterpret_cast<const unsigned*>(mail_hyper(c)) + 2;
    h.rate = c->dropout;
  }
  h.z = c->z;
  h.prob = c->prob;
  h.dz = c->dz;
  h.loss_part = c->loss_part;
  h.g = lo.g;
  h.gstat_part = lo.gstat_part;
  h.B = B;
  h.T = lo.tout;
  h.C = lo.cout;
  h.inv_b = 1.0f / (float)B;
  h.training = (loss ? kHeadTraining : 0) | (c->bce_clipped ? kHeadClippedLoss : 0);
  if (inl) {
    h.fold = fold_of(n - 1, true);   // the head is the first (and only) consumer of the last op
    c->fpar ^= 1;
    if (loss) {
      h.gacc.acc = lo.gacc[c->gpar];
      h.gacc.clear = lo.gacc[c->gpar ^ 1];
      lo.gacc_cur = h.gacc.acc;
    }
  }
  if (lo.res_src >= 0) {
    GOp& rr = c->G[lo.res_src];
    h.rp = rr.p;
    h.rscale = gbn_slot(rr, BN_SCALE);
    h.rshift = gbn_slot(rr, BN_SHIFT);
    h.rT = rr.tout;
    h.rdrop = lo.res_drop;
  }
  if (c->head2) {
    GHead2Args h2;
    h2.h = h;
    h2.watt = c->head_att ? c->params + c->o_att : nullptr;
    h2.pool = c->head_pool;
    h2.hact = c->hact;
    h2.watt_part = c->watt_part;
    if (c->lds_head2 > 64 * 1024)
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ghead_att_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_head2));
    lp.begin("head");
    hipLaunchKernelGGL(ghead_att_kernel, dim3(ghead), dim3(kThreads), c->lds_head2, c->stream, h2);
    lp.end();
    // the dense layer sees hact (already activated): identity "BN" for the dense-weight gradient
    return enqueue_side_work(c, B, metrics, loss, c->hact, c->ones, c->zeros, nullptr);
  }
  lp.begin("head");
  hipLaunchKernelGGL(ghead_kernel, dim3(ghead), dim3(kThreads), 0, c->stream, h);
  lp.end();
  return enqueue_side_work(c, B, metrics, loss, lo.p, gbn_slot(lo, BN_SCALE), gbn_slot(lo, BN_SHIFT), drop ? c->keep : nullptr);
}

int g_enqueue_backward(mww_ctx* c, int B, bool fuse_adam) {
  Launcher lp{c};
  const int n = (int)c->G.size();
  const int gg = std::min(B, c->grid_g);
  const int ghead = std::min(B, c->grid_head);
  GradReduceArgs ga;
  memset(&ga, 0, sizeof(ga));
  // statistics hand-over: the op's own backward launch folds (sum g, sum g*xhat) from the accumulator rows its consumers
  // (or the head) added to; the weight-gradient role publishes c1 / mg / mgx / dgamma / dbeta
  const bool inl = c->bn_inline && c->g_inline_ok && !(c->hook && c->sync_bn) && !c->profile_split;
  auto bfold = [&](GOp& q, bool publish) {
    GFoldBwd f;
    memset(&f, 0, sizeof(f));
    if (!inl || q.norm != MWW_NORM_BN) return f;
    f.acc = q.gacc_cur;
    f.groups = q.groups;
    f.inv_n = 1.0f / ((float)B * (float)q.tout * (float)(q.groups > 1 ? q.cout / q.groups : 1));
    f.dscale = 1.0f;
    f.publish = publish ? 1 : 0;
    f.gamma = c->params + q.o_gamma;
    f.c1 = gbn_slot(q, BN_C1);
    f.mg = gbn_slot(q, BN_MG);
    f.mgx = gbn_slot(q, BN_MGX);
    f.dgamma = c->grads + q.o_gamma;
    f.dbeta = c->grads + q.o_beta;
    return f;
  };
  auto bwd_fin_args = [&](int oi, const StatSource& ss) {
    GOp& q = c->G[oi];
    return GBnBwdArgs{ss.part, ss.G, q.cout, q.groups, ss.inv_n,
                      c->params + q.o_gamma, gbn_slot(q, BN_RSTD), gbn_slot(q, BN_C1), gbn_slot(q, BN_MG), gbn_slot(q, BN_MGX),
                      c->grads + q.o_gamma, c->grads + q.o_beta, ss.dscale, 0};
  };
  auto wgrad_args = [&](int oi) {
    GOp& q = c->G[oi];
    GWgradArgs w;
    memset(&w, 0, sizeof(w));
    w.n_src = q.n_src;
    for (int s = 0; s < q.n_src; ++s) w.src[s] = g_make_src(c, oi, s, false);
    w.y = g_make_bnbwd(c, q);
    w.y.fold = bfold(q, true);
    w.k = q.k;
    w.dil = q.dil;
    w.cin = q.cin;
    w.stride = q.stride;
    w.B = B;
    w.Tin = q.tin;
    w.Tout = q.tout;
    w.grad_part = q.grad_part;
    return w;
  };
  auto dgrad_args = [&](int oi) {
    GOp& q = c->G[oi];
    GConvArgs a;
    memset(&a, 0, sizeof(a));
    a.n_src = q.n_src;
    for (int s = 0; s < q.n_src; ++s) a.src[s] = g_make_src(c, oi, s, true, inl);
    a.w = c->params + q.o_w;   // (the data-gradient kernel reads them transposed / tap-reversed in place)
    a.k = q.k;
    a.dil = q.dil;
    a.cin = q.cout;
    a.stride = 1;
    a.B = B;
    a.Tin = q.tout;
    a.Tout = q.tin;
    a.y = g_make_bnbwd(c, q);
    a.y.fold = bfold(q, false);
    return a;
  };
  const bool split = inl && c->g_role_split;
  const bool pick = inl && c->grid_g_auto;   // per-launch grids (g_role_grid)
  const int gg2 = split ? std::max(1, gg / 2) : gg, gg4 = split ? std::max(1, gg / 4) : gg;
  auto add_segment = [&](int oi, int rows) {
    GOp& q = c->G[oi];
    GradSegment s;
    s.part = q.grad_part;
    s.G = rows;
    s.stride = q.k * q.cin * q.cout;
    s.n = s.stride;
    s.dst = (int)q.o_w;
    ga.seg[ga.nseg++] = s;
  };
  const bool sync = c->hook && c->sync_bn;
  for (int i = n - 1; i >= 0; --i) {
    GOp& o = c->G[i];
    const int members = o.groups > 1 ? o.cout / o.groups : 1;
    if (i > 0 && c->G[i - 1].twin_next && !sync && !c->profile_split) {
      // twins (i-1, i): one finalize launch and one four-role backward launch for the pair
      GOp& o1 = c->G[i - 1];
      const float inv_n = 1.0f / ((float)B * (float)o.tout * (float)members);
      StatSource s0{o.gstat_part, gg, inv_n, 1.0f}, s1{o1.gstat_part, gg, inv_n, 1.0f};
      const GBnBwdArgs bf0 = bwd_fin_args(i, s0), bf1 = bwd_fin_args(i - 1, s1);
      const GWgradArgs w0 = wgrad_args(i), w1 = wgrad_args(i - 1);
      const GConvArgs d0 = dgrad_args(i), d1 = dgrad_args(i - 1);
      const int n0 = o.slots;
      lp.begin("conv_bwd2_", i);
      if (!inl) hipLaunchKernelGGL(gbn_bwd_finalize2_kernel, dim3(o.slots + o1.slots), dim3(kThreads), 0, c->stream, bf0, bf1, n0);
      int rows = gg4;
      const bool ok = launch_gbwd2(c, o.cout, w0, d0, w1, d1, GridPick{pick ? 0 : gg4, B, split ? 4 : 1, c->g_cap_bwd, &rows},
                                   std::max(std::max(o.lds_wg, o.lds_dx), std::max(o1.lds_wg, o1.lds_dx)),
                                   g_shape_id(c, o) == g_shape_id(c, o1) ? g_shape_id(c, o) : 0);
      lp.end();
      if (!ok) return fail(MWW_ERR_UNSUPPORTED, "twin ops without a fused backward instantiation");
      add_segment(i, rows);
      add_segment(i - 1, rows);
      --i;
      continue;
    }
    if (!o.adders.empty()) {
      GResGatherArgs ra;
      memset(&ra, 0, sizeof(ra));
      ra.n = (int)o.adders.size();
      for (int q = 0; q < ra.n; ++q) {
        GOp& x = c->G[o.adders[q]];
        ra.gx[q] = x.g;
        ra.Tx[q] = x.tout;
        ra.drop[q] = x.res_drop;
      }
      ra.p = o.p;
      ra.mean = gbn_slot(o, BN_MEAN);
      ra.rstd = gbn_slot(o, BN_RSTD);
      ra.g = o.g;
      ra.gstat_part = o.gstat_part;
      ra.B = B;
      ra.T = o.tout;
      ra.C = o.cout;
      lp.begin("residual_gather", i);
      hipLaunchKernelGGL(gres_gather_kernel, dim3(gg), dim3(kThreads), 0, c->stream, ra);
      lp.end();
    }
    if (o.norm == MWW_NORM_BN && !inl) {
      StatSource ss;
      int rcs = exchange_stats(c, lp, "bn_gstat_exchange", i, o.gstat_part, i == n - 1 ? ghead : gg, o.cout, 1,
                               1.0f / ((float)B * (float)o.tout * (float)members), &ss);
      if (rcs) return rcs;
      GBnBwdArgs f{ss.part, ss.G, o.cout, o.groups, ss.inv_n,
                   c->params + o.o_gamma, gbn_slot(o, BN_RSTD), gbn_slot(o, BN_C1), gbn_slot(o, BN_MG), gbn_slot(o, BN_MGX),
                   c->grads + o.o_gamma, c->grads + o.o_beta, ss.dscale, 0};
      lp.begin("bn_bwd_finalize", i);
      hipLaunchKernelGGL(gbn_bwd_finalize_kernel, dim3(o.slots), dim3(kThreads), 0, c->stream, f);
      lp.end();
    } else if (o.norm == MWW_NORM_BIAS && !inl) {
      // d bias = sum of the output gradient = the first statistic the consumers already accumulated
      GBnBwdArgs f{o.gstat_part, gg, o.cout, 1, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, c->grads + o.o_beta, 1.0f, 1};
      lp.begin("bias_grad", i);
      hipLaunchKernelGGL(gbn_bwd_finalize_kernel, dim3(o.cout), dim3(kThreads), 0, c->stream, f);
      lp.end();
    }
    if (o.kind == MWW_OP_DEPTHWISE) {
      GDwArgs dw = g_make_dw(c, i, B, true, inl);
      if (inl && o.norm == MWW_NORM_BIAS) {   // the rows this op's consumer added (sum g, ..) to: folded by the weight-gradient launch
        dw.bias_acc = o.gacc_cur;
        dw.dbeta = c->grads + o.o_beta;
      }
      lp.begin("dw_wgrad", i);
      const void* f = reinterpret_cast<const void*>(&gdw_wgrad_kernel);
      if (o.lds_wg > 64 * 1024) HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)o.lds_wg));
      const int gwg = g_role_grid(c, f, o.lds_wg, GridPick{c->grid_g_auto ? 0 : gg, B, 1, c->g_cap_bwd, nullptr});   // (its partial rows are its own)
      hipLaunchKernelGGL(gdw_wgrad_kernel, dim3(gwg), dim3(kThreads), o.lds_wg, c->stream, dw);
      lp.end();
      if (o.needs_dx) {
        lp.begin("dw_dgrad", i);
        const void* fd = reinterpret_cast<const void*>(&gdw_kernel<1>);
        if (o.lds_dx > 64 * 1024) HIPCHK(hipFuncSetAttribute(fd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)o.lds_dx));
        const int gdx = g_role_grid(c, fd, o.lds_dx, GridPick{pick ? 0 : gg, B, 1, c->g_cap_bwd, nullptr});   // (partial statistics rows are shared without the hand-over)
        hipLaunchKernelGGL(gdw_kernel<1>, dim3(gdx), dim3(kThreads), o.lds_dx, c->stream, dw);
        lp.end();
      }
      GradSegment s;
      s.part = o.grad_part;
      s.G = gwg;
      s.stride = o.k * o.cout;
      s.n = s.stride;
      s.dst = (int)o.o_w;
      ga.seg[ga.nseg++] = s;
      continue;
    }
    GWgradArgs w;
    memset(&w, 0, sizeof(w));
    w.n_src = o.n_src;
    for (int s = 0; s < o.n_src; ++s) w.src[s] = g_make_src(c, i, s, false);
    w.y = g_make_bnbwd(c, o);
    w.y.fold = bfold(o, true);
    w.k = o.k;
    w.dil = o.dil;
    w.cin = o.cin;
    w.stride = o.stride;
    w.B = B;
    w.Tin = o.tin;
    w.Tout = o.tout;
    w.grad_part = o.grad_part;
    GConvArgs a;
    memset(&a, 0, sizeof(a));
    if (o.needs_dx) {
      a.n_src = o.n_src;
      for (int s = 0; s < o.n_src; ++s) a.src[s] = g_make_src(c, i, s, true, inl);
      a.w = c->params + o.o_w;
      a.k = o.k;
      a.dil = o.dil;
      a.cin = o.cout;
      a.stride = 1;
      a.B = B;
      a.Tin = o.tout;
      a.Tout = o.tin;
      a.y = g_make_bnbwd(c, o);
      a.y.fold = bfold(o, false);
    }
    bool fused = false;
    int rows = gg;
    int Tc = 0;
    const int S = g_chunks(c, o, inl, true, &Tc);   // frame chunks (1x1 ops): S work items per window for both roles
    size_t lds_wg = o.lds_wg, lds_dx = o.lds_dx;
    if (S > 1) {
      w.S = a.S = S;
      w.Tc = a.Tc = Tc;
      lds_wg = g_lds_wg(o, g_chunk_in(o, Tc), Tc);
      lds_dx = o.needs_dx ? g_lds_dx(o, Tc, Tc) : 0;
    }
    const int items = B * S;
    if (o.needs_dx && !c->profile_split) {
      lp.begin("conv_bwd", i);
      const GridPick pkf{pick ? 0 : gg2, items, split ? 2 : 1, c->g_cap_bwd, &rows};
      fused = S > 1 ? launch_gbwd_fused<true>(c, o.cout, o.cin, w, a, pkf, std::max(lds_wg, lds_dx))
                    : launch_gbwd_fused(c, o.cout, o.cin, w, a, pkf, std::max(lds_wg, lds_dx), g_shape_id(c, o));
      lp.end();
      if (!fused && c->profile) {   // nothing was launched: drop the empty profile entry
        (void)hipEventDestroy(c->prof.back().a);
        (void)hipEventDestroy(c->prof.back().b);
        c->prof.pop_back();
      }
    }
    if (!fused) {
      lp.begin("conv_wgrad", i);
      const GridPick pkw{pick ? 0 : gg, items, 1, c->g_cap_bwd, &rows};
      int rc = S > 1 ? launch_gwgrad<true>(c, o.cout, w, pkw, lds_wg) : launch_gwgrad(c, o.cout, w, pkw, lds_wg, g_shape_id(c, o));
      lp.end();
      if (rc) return rc;
      if (o.needs_dx) {
        lp.begin("conv_dgrad", i);
        const GridPick pkd{pick ? 0 : gg, items, 1, c->g_cap_bwd, nullptr};
        rc = S > 1 ? launch_gconv<1, true>(c, o.cin, a, pkd, lds_dx) : launch_gconv<1>(c, o.cin, a, pkd, lds_dx);
        lp.end();
        if (rc) return rc;
      }
    }
    GradSegment s;
    s.part = o.grad_part;
    s.G = rows;
    s.stride = o.k * o.cin * o.cout;
    s.n = s.stride;
    s.dst = (int)o.o_w;
    ga.seg[ga.nseg++] = s;
  }
  if (c->head2 && c->head_att) {
    GradSegment s;
    s.part = c->watt_part;
    s.G = ghead;
    s.stride = 8;
    s.n = 8;
    s.dst = (int)c->o_att;
    ga.seg[ga.nseg++] = s;
  }
  if (inl) c->gpar ^= 1;
  return enqueue_grad_assembly(c, B, ga, fuse_adam);
}


int enqueue_adam(mww_ctx* c) {
  Launcher lp{c};
  AdamArgs a{c->params, c->grads, c->adam_m, c->adam_v, mail_hyper(c), (int)c->P, 0.9f, 0.999f, 1e-7f};
  lp.begin("adam");
  hipLaunchKernelGGL(adam_kernel, dim3(((int)c->P + kThreads - 1) / kThreads), dim3(kThreads), 0, c->stream, a);
  lp.end();
  return MWW_OK;
}

// the host may rewrite the current mailbox once the GPU work of its previous use has finished
int mail_begin(mww_ctx* c) {
  if (!c->mail_open) {
    HIPCHK(hipEventSynchronize(c->mail_ev[c->mail_cur]));
    c->mail_open = true;
  }
  return MWW_OK;
}
// everything enqueued so far may read the current mailbox: stamp it and move on to the next one
int mail_commit(mww_ctx* c) {
  HIPCHK(hipEventRecord(c->mail_ev[c->mail_cur], c->stream));
  c->mail_cur = (c->mail_cur + 1) % kRing;
  c->mail_open = false;
  c->targets_in_mail = 0;
  return MWW_OK;
}
int push_hyper(mww_ctx* c, float alpha, float gscale) {
  int rc = mail_begin(c);
  if (rc) return rc;
  float* h = reinterpret_cast<float*>(c->mail_host[c->mail_cur] + c->mail_off_hyper);
  h[0] = alpha;
  h[1] = gscale;
  return MWW_OK;
}
// labels / weights written by mww_set_targets that no assembly kernel carried to the device
int flush_targets(mww_ctx* c) {
  if (c->targets_in_mail > 0) {
    const char* m = c->mail_host[c->mail_cur];
    const size_t n = (size_t)c->targets_in_mail * sizeof(float);
    HIPCHK(hipMemcpyAsync(c->y, m + c->mail_off_y, n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->sw, m + c->mail_off_sw, n, hipMemcpyHostToDevice, c->stream));
    c->targets_in_mail = 0;
    c->y_cur = c->y;
    c->sw_cur = c->sw;
  }
  return MWW_OK;
}

float adam_alpha(float lr, int64_t t) {
  // Keras: alpha = lr * sqrt(1 - beta2^t) / (1 - beta1^t), evaluated in float32 like the variables
  const float b1p = powf(0.9f, (float)t), b2p = powf(0.999f, (float)t);
  return lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
}

int step_sequence(mww_ctx* c, int B, int flags) {
  c->metric_launches = 0;
  int rc = enqueue_forward(c, B, true, true, true, !(flags & MWW_STEP_NO_METRICS));
  if (rc) return rc;
  rc = enqueue_backward(c, B, !(flags & MWW_STEP_NO_APPLY));
  if (rc) return rc;
  // the metric state has one writer per step (kernels_head.hip.h MetricState): a second launch with the role would lose counts
  if (c->metric_launches > 1) return fail(MWW_ERR_STATE, "internal: more than one launch of this step carries the metric update");
  return MWW_OK;
}

template <typename T>
int dev_alloc(T** p, size_t n) {
  HIPCHK(hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T)));
  HIPCHK(hipMemset(*p, 0, std::max<size_t>(n, 1) * sizeof(T)));
  return MWW_OK;
}

struct BnSlots { int64_t o_gamma, o_beta, o_mv; int n; };

// mask = 1 everywhere, direct flags on the BN gamma/beta slots; moving variance starts at 1
int init_defaults(mww_ctx* c, const std::vector<BnSlots>& bn) {
  std::vector<float> ones((size_t)c->P, 1.0f);
  std::vector<unsigned char> dir((size_t)c->P, 0);
  std::vector<float> st((size_t)c->S, 0.0f);
  for (const BnSlots& b : bn)
    for (int j = 0; j < b.n; ++j) {
      dir[(size_t)b.o_gamma + j] = 1;
      dir[(size_t)b.o_beta + j] = 1;
      if (b.o_mv >= 0) st[(size_t)b.o_mv + j] = 1.0f;
    }
  HIPCHK(hipMemcpy(c->mask, ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->direct, dir.data(), dir.size(), hipMemcpyHostToDevice));
  c->direct_host = dir;
  HIPCHK(hipMemcpy(c->bn_state, st.data(), st.size() * sizeof(float), hipMemcpyHostToDevice));
  return MWW_OK;
}

// buffers, mailboxes and streams that do not depend on the topology (needs P, S, t_last, c_last)
int alloc_common(mww_ctx* c) {
  const mww_mixednet_desc& d = c->d;
  const size_t mb = (size_t)d.max_batch;
  int rc = 0;
#define A(call) if ((rc = (call)) != 0) return rc;
#define H(call) if ((call) != hipSuccess) return fail(MWW_ERR_HIP, #call);
  A(dev_alloc(&c->params, c->P));
  A(dev_alloc(&c->grads, c->P));
  A(dev_alloc(&c->adam_m, c->P));
  A(dev_alloc(&c->adam_v, c->P));
  A(dev_alloc(&c->mask, c->P));
  A(dev_alloc(&c->direct, c->P));
  A(dev_alloc(&c->bn_state, c->S));
  A(dev_alloc(&c->x, mb * d.frames * MWW_FEATURE_BINS));
  A(dev_alloc(&c->y, mb));
  A(dev_alloc(&c->sw, mb));
  c->y_cur = c->y;
  c->sw_cur = c->sw;
  A(dev_alloc(&c->z, mb));
  A(dev_alloc(&c->prob, mb));
  A(dev_alloc(&c->dz, mb));
  A(dev_alloc(&c->loss_part, mb));
  A(dev_alloc(&c->dwd_part, (size_t)kDenseChunks * c->dwd_stride));
  A(dev_alloc(&c->metrics, 1));
  A(dev_alloc(&c->phase_clk, (size_t)2 * MWW_MAX_BLOCKS * 2048 * kClkSlots));
  c->mail_off_masks = mb * sizeof(mww_window);
  c->mail_off_y = c->mail_off_masks + mb * kMaxMasks * 2 * sizeof(int);
  c->mail_off_sw = c->mail_off_y + mb * sizeof(float);
  c->mail_off_hyper = c->mail_off_sw + mb * sizeof(float);
  c->mail_bytes = c->mail_off_hyper + 16;
  for (int i = 0; i < kRing; ++i) {
    H(hipHostMalloc((void**)&c->mail_host[i], c->mail_bytes, hipHostMallocMapped));
    memset(c->mail_host[i], 0, c->mail_bytes);
    H(hipHostGetDevicePointer((void**)&c->mail_dev[i], c->mail_host[i], 0));
    H(hipEventCreateWithFlags(&c->mail_ev[i], hipEventDisableTiming));
  }
  H(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  for (int i = 0; i < kRing; ++i) {
    A(dev_alloc(&c->mail_hbm[i], c->mail_bytes));
    H(hipEventCreateWithFlags(&c->ev_copy[i], hipEventDisableTiming));
  }
  H(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
  H(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  H(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
#undef A
#undef H
  return MWW_OK;
}

// device / stream / launch-geometry part of context creation
int open_device(mww_ctx* c, int device, void* stream) {
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) return fail(MWW_ERR_HIP, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(MWW_ERR_INVALID, "device index out of range");
  HIPCHK(hipSetDevice(device));
  c->device = device;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (stream) {
    c->stream = (hipStream_t)stream;
  } else {
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  }
  c->grid_fwd = c->n_cu * 4;
  c->grid_bwd = c->n_cu * 2;
  bool wide64 = false;
  for (int i = 0; i < c->d.n_blocks; ++i) wide64 = wide64 || c->d.block_filters[i] > 48;
  if (!c->generic && wide64) {
    // 64-wide blocks: the backward kernels fit once per CU (LDS), the forward kernels twice - grids of resident workgroups
    // only, no second dispatch round (tools/gpu_r3g.sh: notebook topology grid sweep)
    c->grid_fwd = c->n_cu * 2;
    c->grid_bwd = c->n_cu;
  }
  c->grid_head = c->n_cu * 2;   // one window per workgroup at a time, two resident per CU (177 VGPRs): measured 13.5 us vs 15.4 (x4) / 17.4 (x1)
  c->grid_g = c->n_cu * 3;   // measured on the Inception step: 3 workgroups per CU and launch (roles share them) beats 2 and 4
  return MWW_OK;
}

}  // namespace

// ====================================================================================== C ABI
extern "C" {

// mww_version(): version.cpp (carries the sha256 of the source set the library was built from)
const char* mww_last_error(void) { return g_err.c_str(); }

int mww_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int mww_block_kernels_cover(const mww_mixednet_desc* desc, int bf16) {
  if (!desc) return fail(MWW_ERR_INVALID, "null descriptor");
  std::string why;
  if (shape_supported(*desc, &why, bf16 != 0)) return 1;
  g_err = why;
  return 0;
}

int mww_create(const mww_mixednet_desc* desc, int device, void* stream, mww_ctx** out) {
  if (!desc || !out) return fail(MWW_ERR_INVALID, "null argument");
  const mww_mixednet_desc& d = *desc;
  if (d.n_blocks < 2 || d.n_blocks > MWW_MAX_BLOCKS) return fail(MWW_ERR_INVALID, "n_blocks must be in [2, 8]");
  if (d.conv1_stride < 1 || d.frames < d.conv1_kernel) return fail(MWW_ERR_INVALID, "bad first-conv stride / kernel");
  if (d.conv1_filters <= 0) return fail(MWW_ERR_UNSUPPORTED, "first_conv_filters == 0 is not implemented");
  if (d.max_batch <= 0 || d.frames <= 0) return fail(MWW_ERR_INVALID, "frames and max_batch must be positive");
  std::string why;
  if (!shape_supported(d, &why)) return fail(MWW_ERR_UNSUPPORTED, why);
  mww_ctx* c = new mww_ctx();
  c->d = d;
  {
    int rco = open_device(c, device, stream);
    if (rco) { delete c; return rco; }
  }
  // ---- parameter layout
  int64_t off = 0, soff = 0;
  c->o_conv1 = off;
  off += (int64_t)d.conv1_kernel * MWW_FEATURE_BINS * d.conv1_filters;
  int t = (d.frames - d.conv1_kernel) / d.conv1_stride + 1, ch = d.conv1_filters;
  c->L.resize(d.n_blocks);
  for (int i = 0; i < d.n_blocks; ++i) {
    Layer& l = c->L[i];
    l.cin = ch;
    l.cout = d.block_filters[i];
    l.k = d.block_kernel[i];
    l.tin = t;
    l.tout = t - (l.k - 1);
    if (l.tout <= 0) { mww_destroy(c); return fail(MWW_ERR_INVALID, "spectrogram too short for the kernel sizes"); }
    l.o_dw_w = off; off += (int64_t)l.k * l.cin;
    l.o_dw_b = off; off += l.cin;
    l.o_pw_w = off; off += (int64_t)l.cin * l.cout;
    l.o_gamma = off; off += l.cout;
    l.o_beta = off; off += l.cout;
    l.o_mm = soff; soff += l.cout;
    l.o_mv = soff; soff += l.cout;
    t = l.tout;
    ch = l.cout;
  }
  c->t_last = t;
  c->c_last = ch;
  c->o_dense_w = off; off += (int64_t)t * ch;
  c->o_dense_b = off; off += 1;
  c->P = off;
  c->S = soff;
  c->dwd_stride = t * ch + 4;
  const size_t mb = (size_t)d.max_batch;
  // partial rows are sized for the largest grids the "grid_fwd" / "grid_bwd" / "grid_head" options accept, not for this
  // topology's defaults (until round 3 a 64-wide context - defaults 2 / 1 workgroups per CU - overran them when the options
  // asked for more: found by the shape fuzz on the emulator)
  const int gmax_f = c->n_cu * 4, gmax_b = c->n_cu * 2;
  int rc = 0;
#define A(call) if ((rc = (call)) != 0) { mww_destroy(c); return rc; }
  A(alloc_common(c));
  A(dev_alloc(&c->a0, mb * c->L[0].tin * d.conv1_filters));
#ifndef MWW_G_PINGPONG
#define MWW_G_PINGPONG 1
#endif
  // g_k (the gradient at block k's BN output) is written by the backward launch of block k+1 and read by block k's, once: two
  // buffers taken in turn hold them all (35 MB each at the headline batch instead of one per block - address space the
  // memory-side cache does not have to give up activations for)
  if (MWW_G_PINGPONG) {
    size_t need[2] = {0, 0};
    for (int i = 0; i < d.n_blocks; ++i) need[i & 1] = std::max(need[i & 1], mb * c->L[i].tout * c->L[i].cout);
    for (int par = 0; par < 2; ++par)
      if (need[par]) A(dev_alloc(&c->gbuf[par], need[par]));
  }
  for (int i = 0; i < d.n_blocks; ++i) {
    Layer& l = c->L[i];
    A(dev_alloc(&l.p, mb * l.tout * l.cout));
    if (MWW_G_PINGPONG) l.g = c->gbuf[i & 1];
    else A(dev_alloc(&l.g, mb * l.tout * l.cout));
    A(dev_alloc(&l.stat_part, (size_t)gmax_f * 2 * l.cout));
    A(dev_alloc(&l.gstat_part, (size_t)std::max(gmax_b, c->n_cu * 4) * 2 * l.cout));
    for (int par = 0; par < 2; ++par) {
      A(dev_alloc(&l.facc[par], (size_t)kStatRows * 2 * l.cout));
      A(dev_alloc(&l.gacc[par], (size_t)kStatRows * 2 * l.cout));
    }
    l.grad_part_stride = (l.k + 1) * l.cin + l.cin * l.cout;
    if (i == 0) l.grad_part_stride += d.conv1_kernel * MWW_FEATURE_BINS * d.conv1_filters;
    A(dev_alloc(&l.grad_part, (size_t)gmax_b * l.grad_part_stride));
    A(dev_alloc(&l.bn, (size_t)9 * l.cout));
  }
  {
    std::vector<BnSlots> bn;
    for (int i = 0; i < d.n_blocks; ++i) bn.push_back(BnSlots{c->L[i].o_gamma, c->L[i].o_beta, c->L[i].o_mv, c->L[i].cout});
    A(init_defaults(c, bn));
  }
#undef A
  HIPCHK(hipDeviceSynchronize());
  *out = c;
  return MWW_OK;
}

int mww_create_convnet(const mww_convnet_desc* desc, int device, void* stream, mww_ctx** out) {
  if (!desc || !out) return fail(MWW_ERR_INVALID, "null argument");
  const mww_convnet_desc& d = *desc;
  if (d.n_ops < 1 || d.n_ops > MWW_MAX_GRAPH_OPS) return fail(MWW_ERR_INVALID, "n_ops out of range");
  if (d.max_batch <= 0 || d.frames <= 0) return fail(MWW_ERR_INVALID, "frames and max_batch must be positive");
  if (!(d.dropout >= 0.f && d.dropout < 1.f)) return fail(MWW_ERR_INVALID, "dropout rate must be in [0, 1)");
  std::vector<GOp> ops(d.n_ops);
  std::vector<int> n_consumers(d.n_ops, 0);
  int64_t off = 0, soff = 0;
  for (int i = 0; i < d.n_ops; ++i) {
    const mww_conv_bn_op& s = d.ops[i];
    GOp& o = ops[i];
    const std::string tag = "op " + std::to_string(i) + ": ";
    if (s.n_src < 1 || s.n_src > MWW_MAX_OP_SOURCES) return fail(MWW_ERR_INVALID, tag + "1..3 sources");
    if (s.kernel < 1 || s.dilation < 1 || s.filters < 1 || s.bn_groups < 1) return fail(MWW_ERR_INVALID, tag + "bad kernel / dilation / filters / groups");
    if (s.filters % s.bn_groups) return fail(MWW_ERR_INVALID, tag + "filters must be a multiple of the sub-spectral groups");
    if (s.kind != MWW_OP_CONV && s.kind != MWW_OP_DEPTHWISE) return fail(MWW_ERR_INVALID, tag + "unknown op kind");
    if (s.norm < MWW_NORM_BN || s.norm > MWW_NORM_NONE || (s.act != MWW_ACT_RELU && s.act != MWW_ACT_LINEAR)) return fail(MWW_ERR_INVALID, tag + "unknown norm / activation");
    o.res_src = s.residual > 0 ? s.residual - 1 : -1;
    o.res_drop = s.residual_drop;
    o.kind = s.kind;
    o.stride = s.stride > 1 ? s.stride : 1;
    o.norm = s.norm;
    o.act = s.act;
    o.n_src = s.n_src;
    o.k = s.kernel;
    o.dil = s.dilation;
    o.cout = s.filters;
    o.groups = s.bn_groups;
    o.slots = s.norm == MWW_NORM_BN ? (s.bn_groups > 1 ? s.bn_groups : s.filters) : 0;
    o.cin = 0;
    o.tin = -1;
    for (int j = 0; j < s.n_src; ++j) {
      const int src = s.src[j];
      if (src < -1 || src >= i) return fail(MWW_ERR_INVALID, tag + "sources must be earlier ops (or -1 for the spectrogram)");
      for (int j2 = 0; j2 < j; ++j2)
        if (s.src[j2] == src) return fail(MWW_ERR_UNSUPPORTED, tag + "the same source twice");
      if (s.src_drop[j] < 0) return fail(MWW_ERR_INVALID, tag + "negative frame drop");
      const int T = src < 0 ? d.frames : ops[src].tout, Cfull = src < 0 ? MWW_FEATURE_BINS : ops[src].cout;
      const int c0 = s.src_cn[j] > 0 ? s.src_c0[j] : 0, C = s.src_cn[j] > 0 ? s.src_cn[j] : Cfull;
      if (c0 < 0 || c0 + C > Cfull || (src < 0 && C != Cfull)) return fail(MWW_ERR_INVALID, tag + "bad channel slice");
      const int rows = T - s.src_drop[j];
      if (o.tin >= 0 && rows != o.tin) return fail(MWW_ERR_INVALID, tag + "sources are not aligned to the same number of frames");
      o.tin = rows;
      o.cin += C;
      o.src[j] = src;
      o.toff[j] = s.src_drop[j];
      o.sc0[j] = c0;
      o.scn[j] = C;
      if (src >= 0) {
        o.needs_dx = true;
        n_consumers[src]++;
      }
    }
    const int span = o.tin - (o.k - 1) * o.dil;
    if (span <= 0) return fail(MWW_ERR_INVALID, tag + "spectrogram too short for the kernel sizes");
    o.tout = (span - 1) / o.stride + 1;
    if (o.stride > 1 && o.needs_dx) return fail(MWW_ERR_UNSUPPORTED, tag + "a time stride is only implemented for ops fed by the spectrogram");
    const int pad = (o.k - 1) * o.dil;
    if (o.kind == MWW_OP_DEPTHWISE) {
      if (o.n_src != 1 || o.cin != o.cout || o.dil != 1 || o.stride != 1) return fail(MWW_ERR_INVALID, tag + "a depthwise op has one source with as many channels as filters, no dilation, no stride");
      if (o.norm == MWW_NORM_BN) return fail(MWW_ERR_UNSUPPORTED, tag + "depthwise + BatchNorm is not implemented (bias or nothing)");
      if (o.cout > kThreads || o.k * o.cout > kGDwTasks * kThreads) return fail(MWW_ERR_UNSUPPORTED, tag + "depthwise op too large (channels <= 256, taps x channels <= 2048)");
      // tap blocks of 8 with zero weights, kGDwTail zero rows behind every staged window (kernels_graph.hip.h)
      const size_t pi = (size_t)(o.cout | 1), wsz = (size_t)gdw_kpad(o.k) * o.cout;
      o.lds_fwd = (wsz + (size_t)(o.tin + kGDwTail) * pi) * sizeof(float);
      o.lds_dx = o.needs_dx ? (wsz + (size_t)(o.tout + 2 * pad + kGDwTail) * pi) * sizeof(float) : 0;
      o.lds_wg = std::max((size_t)(o.tin + kGDwTail) * pi + (size_t)(o.tout + kGDwJ) * pi, (size_t)2 * kThreads * kGDwJ) * sizeof(float);   // (.. or the scratch of the final sum)
    } else {
      if (!g_width_supported(o.cout)) return fail(MWW_ERR_UNSUPPORTED, tag + "filter count not instantiated (8,10,12,16,20,24,30,32,36,40,48,60,64)");
      if (o.needs_dx && !g_width_supported(o.cin)) return fail(MWW_ERR_UNSUPPORTED, tag + "input channel count not instantiated");
      if (o.k * o.cin > kThreads) return fail(MWW_ERR_UNSUPPORTED, tag + "kernel x input channels exceeds 256");
      o.lds_fwd = g_lds_fwd(o, o.tin, o.tout);   // (g_lds_*: the LDS tiles of the MFMA kernels)
      o.lds_dx = o.needs_dx ? g_lds_dx(o, o.tout + 2 * pad, o.tin) : 0;
      o.lds_wg = g_lds_wg(o, o.tin, o.tout);
    }
    if (std::max(o.lds_fwd, std::max(o.lds_dx, o.lds_wg)) > kMaxDynLds) return fail(MWW_ERR_UNSUPPORTED, tag + "window does not fit the LDS tile");
    o.o_w = off; off += o.kind == MWW_OP_DEPTHWISE ? (int64_t)o.k * o.cout : (int64_t)o.k * o.cin * o.cout;
    if (o.norm == MWW_NORM_BN) {
      o.o_gamma = off; off += o.slots;
      o.o_beta = off; off += o.slots;
      o.o_mm = soff; soff += o.slots;
      o.o_mv = soff; soff += o.slots;
    } else if (o.norm == MWW_NORM_BIAS) {
      o.o_beta = off; off += o.cout;
    }
  }
  for (int i = 0; i < d.n_ops; ++i) {
    GOp& o = ops[i];
    if (o.res_src < 0) continue;
    const std::string tag = "op " + std::to_string(i) + ": ";
    if (o.res_src >= i) return fail(MWW_ERR_INVALID, tag + "the residual op must come earlier");
    GOp& r = ops[o.res_src];
    if (r.kind != MWW_OP_CONV || r.norm != MWW_NORM_BN || r.act != MWW_ACT_LINEAR || o.norm != MWW_NORM_BN)
      return fail(MWW_ERR_UNSUPPORTED, tag + "a residual is a conv + BatchNorm + linear op added to a BatchNorm output");
    if (r.cout != o.cout || o.res_drop < 0 || r.tout - o.res_drop != o.tout) return fail(MWW_ERR_INVALID, tag + "residual shape does not match");
    if (n_consumers[o.res_src] != 0) return fail(MWW_ERR_UNSUPPORTED, tag + "a residual op cannot also be a regular source");
    if ((int)r.adders.size() >= kGMaxAdders) return fail(MWW_ERR_UNSUPPORTED, tag + "too many ops add the same residual");
    r.adders.push_back(i);
    for (int i2 = i + 1; i2 < d.n_ops; ++i2)
      for (int j2 = 0; j2 < ops[i2].n_src; ++j2)
        if (ops[i2].src[j2] == i && ops[i2].scn[j2] != o.cout) return fail(MWW_ERR_UNSUPPORTED, tag + "an op with a residual must be read whole (no channel slice)");
  }
  for (int i = 0; i + 1 < d.n_ops; ++i)
    if (n_consumers[i] == 0 && ops[i].adders.empty()) return fail(MWW_ERR_INVALID, "op " + std::to_string(i) + " has no consumer");
  // twins: consecutive, mutually independent convolutions of one shape (Inception's second-level k x 1 convs of
  // branch 2 and branch 3) share their forward, finalize and backward launches
  auto twin_width = [](int n) {
#define X(N) if (n == N) return true;
    MWW_G_TWIN_WIDTHS(X)
#undef X
    return false;
  };
  for (int i = 0; i + 2 < d.n_ops; ++i) {
    GOp &a = ops[i], &b = ops[i + 1];
    const bool same = a.kind == MWW_OP_CONV && b.kind == MWW_OP_CONV && a.k == b.k && a.dil == b.dil && a.cin == b.cin && a.cout == b.cout &&
                      a.groups == b.groups && a.norm == MWW_NORM_BN && b.norm == MWW_NORM_BN && a.act == b.act && a.stride == 1 &&
                      b.stride == 1 && a.tin == b.tin && a.n_src == 1 && b.n_src == 1 && a.src[0] >= 0 && b.src[0] >= 0 &&
                      b.src[0] != i && a.res_src < 0 && b.res_src < 0 && a.adders.empty() && b.adders.empty() && a.cin == a.cout &&
                      twin_width(a.cout);
    // twins run concurrently inside one launch: they must not route gradient into the same channels of one producer
    // (store vs accumulate would race; e.g. the unfused 1x1 branch heads of an Inception block with sub-spectral groups)
    const bool shared = a.src[0] == b.src[0] && a.sc0[0] < b.sc0[0] + b.cin && b.sc0[0] < a.sc0[0] + a.cin;
    if (same && !shared && (i == 0 || !ops[i - 1].twin_next)) a.twin_next = true;
  }
  // gradient routing: per producer, the slices its consumers read must be identical or disjoint and cover
  // every channel; in the backward pass (descending op index) the first consumer of a slice stores, later
  // ones accumulate and the last one also emits the BN statistics partials of that slice
  for (int pi = 0; pi + 1 < d.n_ops; ++pi) {
    if (!ops[pi].adders.empty()) continue;   // residual ops: gradient gathered from their adders
    std::vector<int> covered(ops[pi].cout, 0);
    for (int i = d.n_ops - 1; i > pi; --i)
      for (int j = 0; j < ops[i].n_src; ++j) {
        if (ops[i].src[j] != pi) continue;
        const int c0 = ops[i].sc0[j], cn = ops[i].scn[j];
        bool first = true, last = true;
        for (int i2 = pi + 1; i2 < d.n_ops; ++i2)
          for (int j2 = 0; j2 < ops[i2].n_src; ++j2) {
            if (ops[i2].src[j2] != pi || (i2 == i && j2 == j)) continue;
            const int d0 = ops[i2].sc0[j2], dn = ops[i2].scn[j2];
            if (d0 + dn <= c0 || c0 + cn <= d0) continue;   // disjoint
            if (d0 != c0 || dn != cn) return fail(MWW_ERR_UNSUPPORTED, "op " + std::to_string(pi) + ": consumers read overlapping, unequal channel slices");
            if (i2 > i) first = false;
            if (i2 < i) last = false;
          }
        ops[i].src_first[j] = first;
        ops[i].src_last[j] = last;
        for (int cc = c0; cc < c0 + cn; ++cc) covered[cc] = 1;
      }
    for (int cc = 0; cc < ops[pi].cout; ++cc)
      if (!covered[cc]) return fail(MWW_ERR_UNSUPPORTED, "op " + std::to_string(pi) + ": channel " + std::to_string(cc) + " has no consumer");
  }
  // planar tensors: a convolution + BatchNorm op whose consumers are all convolutions that read one of `planes` equal slices
  // each (the fused 1x1 branch heads of an Inception block: 30 = 3 x 10, 48 = 3 x 16 channels).  Only for the widths whose
  // own backward staging is the direct one (kernels_graph.hip.h GDpPipe is not planar-aware: 30 and 48 exceed its registers).
  for (int pi = 0; pi + 1 < d.n_ops; ++pi) {
    GOp& pr = ops[pi];
    if (pr.kind != MWW_OP_CONV || pr.norm != MWW_NORM_BN || pr.res_src >= 0 || !pr.adders.empty() || (pr.cout != 30 && pr.cout != 48)) continue;
    int cn = 0;
    bool ok = true;
    for (int i = pi + 1; i < d.n_ops && ok; ++i)
      for (int j = 0; j < ops[i].n_src; ++j) {
        if (ops[i].src[j] != pi) continue;
        if (ops[i].kind != MWW_OP_CONV || ops[i].res_src >= 0 || ops[i].stride != 1) ok = false;
        if (cn == 0) cn = ops[i].scn[j];
        if (ops[i].scn[j] != cn || ops[i].scn[j] >= pr.cout || ops[i].sc0[j] % cn) ok = false;
      }
    if (ok && cn > 0 && pr.cout % cn == 0 && (cn % 2) == 0) {
      pr.planes = pr.cout / cn;
      pr.pc = cn;
    }
  }
  if (n_consumers[d.n_ops - 1] != 0) return fail(MWW_ERR_INVALID, "the last op feeds the classifier head and cannot have other consumers");
  {
    const GOp& lo = ops[d.n_ops - 1];
    if (lo.kind != MWW_OP_CONV || lo.norm != MWW_NORM_BN || lo.act != MWW_ACT_RELU)
      return fail(MWW_ERR_UNSUPPORTED, "the classifier head expects a convolution + BatchNorm + ReLU as the last op");
  }

  mww_ctx* c = new mww_ctx();
  memset(&c->d, 0, sizeof(c->d));
  c->d.frames = d.frames;
  c->d.max_batch = d.max_batch;
  c->generic = true;
  c->dropout = d.dropout;
  c->G = ops;
  // frame chunks ("graph_frame_chunks"): automatic for graphs with depthwise ops, i.e. MixedNet flag sets on this engine - their
  // wide 1x1 ops hold 45-105 KB of LDS per whole-window workgroup; measured on the default MixedNet forced onto this engine
  // 0.877 -> 0.815 ms/step (3: 0.818).  Off for pure convolution graphs: Inception 0.892 / 0.897 / 0.957 / 0.960 ms for 0 / 1 / 2 / 3
  // (profiles/round3_frame_chunks.txt)
  for (const GOp& o : ops)
    if (o.kind == MWW_OP_DEPTHWISE) c->g_chunks = 1;
  {
    // statistics hand-over: possible when every op is a convolution followed by a BatchNorm / SSN (or by nothing: a
    // MixedNet's first convolution) or a depthwise op with a bias (or nothing), none has a residual branch and every folded
    // tensor fits the kernels' fold table; first_consumer = the op whose launch folds
    bool ok = true;
    for (int i = 0; i < d.n_ops; ++i) {
      GOp& o = c->G[i];
      const bool conv_ok = o.kind == MWW_OP_CONV && (o.norm == MWW_NORM_BN || o.norm == MWW_NORM_NONE);
      const bool dw_ok = o.kind == MWW_OP_DEPTHWISE && (o.norm == MWW_NORM_BIAS || o.norm == MWW_NORM_NONE);
      if (!(conv_ok || dw_ok) || o.res_src >= 0 || !o.adders.empty() || o.cout > kGFoldC) ok = false;
      for (int j = 0; j < o.n_src; ++j)
        if (o.src[j] >= 0 && c->G[o.src[j]].first_consumer < 0) c->G[o.src[j]].first_consumer = i;
    }
    c->g_inline_ok = ok && !d.head_attention && !d.head_pool;
  }
  {
    int rco = open_device(c, device, stream);
    if (rco) { mww_destroy(c); return rco; }
  }
  GOp& lo = c->G.back();
  c->t_last = lo.tout;
  c->c_last = lo.cout;
  if (lo.tout > 1 && (d.head_attention || d.head_pool)) {   // mixednet.py:362: only if more than one frame remains
    if (d.head_pool < 0 || d.head_pool > 2) { mww_destroy(c); return fail(MWW_ERR_INVALID, "head_pool must be 0, 1 or 2"); }
    if (d.head_attention && lo.tout < 4) { mww_destroy(c); return fail(MWW_ERR_INVALID, "spatial attention needs at least 4 frames"); }
    if (d.dropout > 0.f) { mww_destroy(c); return fail(MWW_ERR_UNSUPPORTED, "dropout with the attention / pooled head"); }
    c->head2 = true;
    c->head_att = d.head_attention != 0;
    c->head_pool = d.head_pool;
    const int to = lo.tout - (c->head_att ? 3 : 0);
    c->t_last = c->head_pool ? 1 : to;
    if (c->head_att) { c->o_att = off; off += 8; }
    c->lds_head2 = ((size_t)lo.tout * (lo.cout | 1) + 7 * (size_t)lo.tout + 3 * (size_t)lo.cout) * sizeof(float);
    if (c->lds_head2 > kMaxDynLds) { mww_destroy(c); return fail(MWW_ERR_UNSUPPORTED, "window does not fit the head's LDS tile"); }
  }
  c->o_dense_w = off; off += (int64_t)c->t_last * lo.cout;
  c->o_dense_b = off; off += 1;
  c->P = off;
  c->S = soff;
  c->dwd_stride = c->t_last * lo.cout + 4;
  const size_t mb = (size_t)d.max_batch;
  const int gmax = c->n_cu * 4;
  int rc = 0;
#define A(call) if ((rc = (call)) != 0) { mww_destroy(c); return rc; }
  A(alloc_common(c));
  A(dev_alloc(&c->keep, mb * lo.tout * lo.cout));
  if (c->head2) {
    A(dev_alloc(&c->hact, mb * c->t_last * lo.cout));
    A(dev_alloc(&c->watt_part, (size_t)gmax * 8));
  }
  std::vector<BnSlots> bn;
  for (GOp& o : c->G) {
    A(dev_alloc(&o.p, mb * o.tout * o.cout + (size_t)o.planes * kPlanePad));
    A(dev_alloc(&o.g, mb * o.tout * o.cout + (size_t)o.planes * kPlanePad));
    A(dev_alloc(&o.stat_part, (size_t)gmax * 2 * o.cout));
    A(dev_alloc(&o.gstat_part, (size_t)gmax * 2 * o.cout));
    A(dev_alloc(&o.grad_part, (size_t)gmax * o.k * (o.kind == MWW_OP_DEPTHWISE ? 1 : o.cin) * o.cout));   // ("grid_graph" may be raised to gmax)
    A(dev_alloc(&o.bn, (size_t)9 * o.cout));
    for (int par = 0; par < 2; ++par) {
      A(dev_alloc(&o.facc[par], (size_t)kStatRows * 2 * o.cout));
      A(dev_alloc(&o.gacc[par], (size_t)kStatRows * 2 * o.cout));
    }
    if (o.norm == MWW_NORM_BN) bn.push_back(BnSlots{o.o_gamma, o.o_beta, o.o_mv, o.slots});
    else if (o.norm == MWW_NORM_BIAS) bn.push_back(BnSlots{o.o_beta, o.o_beta, -1, o.cout});   // bias gradient is written directly too
  }
  {
    A(dev_alloc(&c->ones, (size_t)kThreads));
    A(dev_alloc(&c->zeros, (size_t)kThreads));
    std::vector<float> one((size_t)kThreads, 1.0f);
    if (hipMemcpy(c->ones, one.data(), one.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { mww_destroy(c); return fail(MWW_ERR_HIP, "hipMemcpy"); }
  }
  A(init_defaults(c, bn));
#undef A
  HIPCHK(hipDeviceSynchronize());
  *out = c;
  return MWW_OK;
}

int mww_set_allreduce_hook(mww_ctx* c, mww_allreduce_fn fn, void* user, int world_size, int sync_bn, int reduce_grads) {
  if (!c) return fail(MWW_ERR_INVALID, "null context");
  if (fn && world_size < 1) return fail(MWW_ERR_INVALID, "world size must be positive");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->hook = fn;
  c->hook_user = user;
  c->world = fn ? world_size : 1;
  c->sync_bn = fn && sync_bn;
  c->reduce_grads = fn && reduce_grads;
  if (c->sync_bn && !c->sync_buf) {
    int64_t off = 0;
    c->sync_off.clear();
    if (c->generic) for (auto& o : c->G) { c->sync_off.push_back(off); off += 4 * (int64_t)o.cout; }
    else for (auto& l : c->L) { c->sync_off.push_back(off); off += 4 * (int64_t)l.cout; }
    int rc = dev_alloc(&c->sync_buf, (size_t)off);
    if (rc) return rc;
  }
  return MWW_OK;
}

// ---- RCCL inside the library (mww_allreduce_init).  The exchange the caller's hook performed through
// Python / torch.distributed (two ctypes callbacks, two dispatcher round trips and a pair of cross-stream event waits
// per step: +38 us at W = 1 for the two-bucket schedule in round 2) is issued here, from the launching thread:
//   MWW_EXCHANGE_IN_ORDER  ncclAllReduce on the context's stream itself
//   MWW_EXCHANGE_DEFERRED  event on the context's stream -> the library's side stream waits for it -> ncclAllReduce there
//   MWW_EXCHANGE_FLUSH     the context's stream waits for the side stream's last exchange
struct RcclState {
  RcclApi api;
  void* comm = nullptr;
  hipStream_t side = nullptr;
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  mww_ctx* c = nullptr;
  bool deferred = false;
};

namespace {
int rccl_load(RcclApi* a) {
  if (a->so) return MWW_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) {
    a->so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (a->so) break;
  }
  if (!a->so) return fail(MWW_ERR_UNSUPPORTED, std::string("librccl.so not found: ") + dlerror());
  a->GetUniqueId = reinterpret_cast<decltype(a->GetUniqueId)>(dlsym(a->so, "ncclGetUniqueId"));
  a->CommInitRank = reinterpret_cast<decltype(a->CommInitRank)>(dlsym(a->so, "ncclCommInitRank"));
  a->AllReduce = reinterpret_cast<decltype(a->AllReduce)>(dlsym(a->so, "ncclAllReduce"));
  a->CommDestroy = reinterpret_cast<decltype(a->CommDestroy)>(dlsym(a->so, "ncclCommDestroy"));
  a->GetErrorString = reinterpret_cast<decltype(a->GetErrorString)>(dlsym(a->so, "ncclGetErrorString"));
  a->CommCount = reinterpret_cast<decltype(a->CommCount)>(dlsym(a->so, "ncclCommCount"));
  if (!a->GetUniqueId || !a->CommInitRank || !a->AllReduce || !a->CommDestroy || !a->GetErrorString)
    return fail(MWW_ERR_UNSUPPORTED, "librccl.so lacks an entry point");
  return MWW_OK;
}

int rccl_exchange(void* user, float* buf, int64_t n, int flags) {
  RcclState* r = static_cast<RcclState*>(user);
  mww_ctx* c = r->c;
  constexpr int kFloat = 7, kSum = 0;   // ncclFloat32, ncclSum (rccl.h)
  if (flags == MWW_EXCHANGE_FLUSH) {
    if (r->deferred) {
      if (hipStreamWaitEvent(c->stream, r->ev_done, 0) != hipSuccess) return -1;
      r->deferred = false;
    }
    return 0;
  }
  hipStream_t st = c->stream;
  if (flags == MWW_EXCHANGE_DEFERRED) {
    if (hipEventRecord(r->ev_ready, c->stream) != hipSuccess) return -1;
    if (hipStreamWaitEvent(r->side, r->ev_ready, 0) != hipSuccess) return -1;
    st = r->side;
  }
  const int e = r->api.AllReduce(buf, buf, (size_t)n, kFloat, kSum, r->comm, st);
  if (e != 0) {
    g_err = std::string("ncclAllReduce: ") + r->api.GetErrorString(e);
    return -1;
  }
  if (flags == MWW_EXCHANGE_DEFERRED) {
    if (hipEventRecord(r->ev_done, r->side) != hipSuccess) return -1;
    r->deferred = true;
  }
  return 0;
}
}  // namespace

int mww_allreduce_unique_id(void* out_id, int capacity) {
  if (!out_id || capacity < MWW_UNIQUE_ID_BYTES) return fail(MWW_ERR_INVALID, "unique-id buffer too small");
  static RcclApi api;
  int rc = rccl_load(&api);
  if (rc) return rc;
  RcclApi::UniqueId id;
  const int e = api.GetUniqueId(&id);
  if (e != 0) return fail(MWW_ERR_HIP, std::string("ncclGetUniqueId: ") + api.GetErrorString(e));
  memcpy(out_id, id.internal, sizeof(id.internal));
  return MWW_OK;
}

int mww_allreduce_destroy(mww_ctx* c) {
  if (!c) return fail(MWW_ERR_INVALID, "null context");
  RcclState* r = c->rccl;
  if (!r) return MWW_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (r->side) (void)hipStreamSynchronize(r->side);
  if (c->hook == rccl_exchange) {
    c->hook = nullptr;
    c->hook_user = nullptr;
    c->world = 1;
    c->sync_bn = c->reduce_grads = false;
  }
  if (r->comm) (void)r->api.CommDestroy(r->comm);
  if (r->side) (void)hipStreamDestroy(r->side);
  if (r->ev_ready) (void)hipEventDestroy(r->ev_ready);
  if (r->ev_done) (void)hipEventDestroy(r->ev_done);
  delete r;
  c->rccl = nullptr;
  return MWW_OK;
}

int mww_allreduce_world(mww_ctx* c) {
  if (!c) return fail(MWW_ERR_INVALID, "null context");
  if (!c->rccl || !c->rccl->comm) return 0;
  int n = 0;
  if (!c->rccl->api.CommCount) return fail(MWW_ERR_UNSUPPORTED, "librccl.so lacks ncclCommCount");
  const int e = c->rccl->api.CommCount(c->rccl->comm, &n);
  if (e != 0) return fail(MWW_ERR_HIP, std::string("ncclCommCount: ") + c->rccl->api.GetErrorString(e));
  return n;
}

int mww_allreduce_init(mww_ctx* c, int rank, int world, const void* unique_id, int sync_bn) {
  if (!c || !unique_id || world < 1 || rank < 0 || rank >= world) return fail(MWW_ERR_INVALID, "bad rank / world size");
  int rc = mww_allreduce_destroy(c);
  if (rc) return rc;
  HIPCHK(hipSetDevice(c->device));
  RcclState* r = new RcclState();
  r->c = c;
  rc = rccl_load(&r->api);
  if (rc) { delete r; return rc; }
  RcclApi::UniqueId id;
  memcpy(id.internal, unique_id, sizeof(id.internal));
  const int e = r->api.CommInitRank(&r->comm, world, id, rank);
  if (e != 0) {
    const std::string msg = std::string("ncclCommInitRank: ") + r->api.GetErrorString(e);
    delete r;
    return fail(MWW_ERR_HIP, msg);
  }
  c->rccl = r;
  HIPCHK(hipStreamCreateWithFlags(&r->side, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&r->ev_ready, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&r->ev_done, hipEventDisableTiming));
  return mww_set_allreduce_hook(c, rccl_exchange, r, world, sync_bn, 1);
}

int mww_set_dropout_mask(mww_ctx* c, const uint8_t* keep, int B) {
  if (!c || !c->generic) return fail(MWW_ERR_INVALID, "context has no dropout layer");
  if (!keep) { c->keep_explicit = false; return MWW_OK; }
  if (B <= 0 || B > c->d.max_batch) return fail(MWW_ERR_INVALID, "bad batch size");
  if (!(c->dropout > 0.f)) return fail(MWW_ERR_STATE, "model was created with dropout = 0");
  const size_t n = (size_t)B * c->t_last * c->c_last;
  std::vector<float> h(n);
  const float sc = 1.0f / (1.0f - c->dropout);
  for (size_t i = 0; i < n; ++i) h[i] = keep[i] ? sc : 0.f;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpyAsync(c->keep, h.data(), n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->keep_explicit = true;
  return MWW_OK;
}

void mww_destroy(mww_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  (void)mww_allreduce_destroy(c);
  for (auto& g : c->graphs) (void)hipGraphExecDestroy(g.exec);
  for (auto& e : c->prof) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  void* flat[] = {c->params, c->grads, c->adam_m, c->adam_v, c->mask, c->direct, c->bn_state, c->x, c->y, c->sw,
                  c->z, c->prob, c->dz, c->loss_part, c->dwd_part, c->metrics, c->phase_clk, c->a0};
  for (void* p : flat) if (p) (void)hipFree(p);
  for (auto& l : c->L) {
    void* lp[] = {l.p, (c->gbuf[0] || c->gbuf[1]) ? nullptr : l.g, l.stat_part, l.gstat_part, l.grad_part, l.bn, l.facc[0], l.facc[1], l.gacc[0], l.gacc[1]};
    for (void* p : lp) if (p) (void)hipFree(p);
  }
  for (float* p : c->gbuf) if (p) (void)hipFree(p);
  for (auto& o : c->G) {
    void* op[] = {o.p, o.g, o.stat_part, o.gstat_part, o.grad_part, o.bn, o.facc[0], o.facc[1], o.gacc[0], o.gacc[1]};
    for (void* p : op) if (p) (void)hipFree(p);
  }
  if (c->sync_buf) (void)hipFree(c->sync_buf);
  if (c->hact) (void)hipFree(c->hact);
  if (c->watt_part) (void)hipFree(c->watt_part);
  if (c->ones) (void)hipFree(c->ones);
  if (c->zeros) (void)hipFree(c->zeros);
  if (c->keep) (void)hipFree(c->keep);
  for (int i = 0; i < MWW_MAX_STORES; ++i) if (c->store[i]) (void)hipFree(c->store[i]);
  if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
  for (int i = 0; i < kRing; ++i) {
    if (c->mail_host[i]) (void)hipHostFree(c->mail_host[i]);
    if (c->mail_ev[i]) (void)hipEventDestroy(c->mail_ev[i]);
    if (c->mail_hbm[i]) (void)hipFree(c->mail_hbm[i]);
    if (c->ev_copy[i]) (void)hipEventDestroy(c->ev_copy[i]);
  }
  if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int mww_synchronize(mww_ctx* c) {
  if (!c) return fail(MWW_ERR_INVALID, "null context");
  HIPCHK(hipStreamSynchronize(c->stream));
  return MWW_OK;
}

int64_t mww_num_params(const mww_ctx* c) { return c ? c->P : 0; }
int64_t mww_num_bn_state(const mww_ctx* c) { return c ? c->S : 0; }

static int copy_in(mww_ctx* c, void* dst, const void* src, size_t bytes) {
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return MWW_OK;
}
static int copy_out(mww_ctx* c, void* dst, const void* src, size_t bytes) {
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return MWW_OK;
}

int mww_set_params(mww_ctx* c, const float* h, int64_t n) {
  if (!c || !h || n != c->P) return fail(MWW_ERR_INVALID, "parameter vector size mismatch");
  return copy_in(c, c->params, h, (size_t)n * sizeof(float));
}
int mww_get_params(mww_ctx* c, float* h, int64_t n) {
  if (!c || !h || n != c->P) return fail(MWW_ERR_INVALID, "parameter vector size mismatch");
  return copy_out(c, h, c->params, (size_t)n * sizeof(float));
}
int mww_set_bn_state(mww_ctx* c, const float* h, int64_t n) {
  if (!c || !h || n != c->S) return fail(MWW_ERR_INVALID, "BN state size mismatch");
  return copy_in(c, c->bn_state, h, (size_t)n * sizeof(float));
}
int mww_get_bn_state(mww_ctx* c, float* h, int64_t n) {
  if (!c || !h || n != c->S) return fail(MWW_ERR_INVALID, "BN state size mismatch");
  return copy_out(c, h, c->bn_state, (size_t)n * sizeof(float));
}
int mww_set_grad_mask(mww_ctx* c, const float* h, int64_t n) {
  if (!c || !h || n != c->P) return fail(MWW_ERR_INVALID, "mask size mismatch");
  return copy_in(c, c->mask, h, (size_t)n * sizeof(float));
}
int mww_set_opt_state(mww_ctx* c, const float* m, const float* v, int64_t n, int64_t step) {
  if (!c || !m || !v || n != c->P || step < 0) return fail(MWW_ERR_INVALID, "optimizer state size mismatch");
  int rc = copy_in(c, c->adam_m, m, (size_t)n * sizeof(float));
  if (rc) return rc;
  rc = copy_in(c, c->adam_v, v, (size_t)n * sizeof(float));
  c->step = step;
  return rc;
}
int mww_get_opt_state(mww_ctx* c, float* m, float* v, int64_t n, int64_t* step) {
  if (!c || n != c->P) return fail(MWW_ERR_INVALID, "optimizer state size mismatch");
  int rc = 0;
  if (m) rc = copy_out(c, m, c->adam_m, (size_t)n * sizeof(float));
  if (!rc && v) rc = copy_out(c, v, c->adam_v, (size_t)n * sizeof(float));
  if (step) *step = c->step;
  return rc;
}
int mww_get_grads(mww_ctx* c, float* h, int64_t n) {
  if (!c || !h || n != c->P) return fail(MWW_ERR_INVALID, "gradient vector size mismatch");
  return copy_out(c, h, c->grads, (size_t)n * sizeof(float));
}

int mww_upload_store(mww_ctx* c, int id, const void* data, int64_t n, int dtype) {
  if (!c || !data || id < 0 || id >= MWW_MAX_STORES || n <= 0) return fail(MWW_ERR_INVALID, "bad store arguments");
  if (dtype != MWW_DTYPE_U16 && dtype != MWW_DTYPE_F32) return fail(MWW_ERR_INVALID, "store dtype must be uint16 or float32");
  if (n % MWW_FEATURE_BINS) return fail(MWW_ERR_INVALID, "store length is not a multiple of 40 bins");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->store[id]) { HIPCHK(hipFree(c->store[id])); c->store[id] = nullptr; }
  const size_t bytes = (size_t)n * (dtype == MWW_DTYPE_U16 ? 2 : 4);
  HIPCHK(hipMalloc(&c->store[id], bytes + 16));
  c->store_dtype[id] = dtype;
  c->store_elems[id] = n;
  return copy_in(c, c->store[id], data, bytes);
}

int mww_assemble_batch(mww_ctx* c, const mww_window* win, const int32_t* masks, int B, int ntm, int nfm) {
  if (!c || !win || B <= 0 || B > c->d.max_batch) return fail(MWW_ERR_INVALID, "bad batch size");
  const int nm = ntm + nfm;
  if (ntm < 0 || nfm < 0 || nm > kMaxMasks || (nm > 0 && !masks)) return fail(MWW_ERR_INVALID, "too many masks");
  const int T = c->d.frames;
  for (int j = 0; j < B; ++j) {
    const mww_window& w = win[j];
    if (w.store < 0 || w.store >= MWW_MAX_STORES || !c->store[w.store]) return fail(MWW_ERR_INVALID, "window refers to a store that was not uploaded");
    if (w.pad_rows < 0 || w.copy_rows < 0 || w.pad_rows + w.copy_rows != T) return fail(MWW_ERR_INVALID, "window rows do not add up to the spectrogram length");
    if (w.src_elem < 0 || w.src_elem + (int64_t)w.copy_rows * MWW_FEATURE_BINS > c->store_elems[w.store]) return fail(MWW_ERR_INVALID, "window reads past the end of its store");
  }
  HIPCHK(hipSetDevice(c->device));
  int rcm = mail_begin(c);
  if (rcm) return rcm;
  char* mh = c->mail_host[c->mail_cur];
  char* md = c->mail_hbm[c->mail_cur];
  memcpy(mh, win, (size_t)B * sizeof(mww_window));
  if (nm) memcpy(mh + c->mail_off_masks, masks, (size_t)B * nm * 2 * sizeof(int));
  HIPCHK(hipMemcpyAsync(md, mh, c->mail_off_hyper, hipMemcpyHostToDevice, c->copy_stream));
  HIPCHK(hipEventRecord(c->ev_copy[c->mail_cur], c->copy_stream));
  HIPCHK(hipStreamWaitEvent(c->stream, c->ev_copy[c->mail_cur], 0));
  AssembleArgs a;
  for (int i = 0; i < MWW_MAX_STORES; ++i) { a.store[i] = c->store[i]; a.dtype[i] = c->store_dtype[i]; }
  a.win = reinterpret_cast<const mww_window*>(md);
  a.masks = reinterpret_cast<const int*>(md + c->mail_off_masks);
  // labels / weights already in this mailbox ride along: window j's workgroup moves row j to HBM
  a.n_targets = c->targets_in_mail >= B ? B : 0;
  a.y_src = reinterpret_cast<const float*>(md + c->mail_off_y);
  a.sw_src = reinterpret_cast<const float*>(md + c->mail_off_sw);
  a.y_dst = c->y;
  a.sw_dst = c->sw;
  if (a.n_targets) c->targets_in_mail = 0;
  a.x = c->x;
  a.B = B;
  a.T = T;
  a.ntm = ntm;
  a.nfm = nfm;
  a.split = c->asm_split;
  {
    const int per_fwd = (B + std::min(B, c->grid_fwd) - 1) / std::min(B, c->grid_fwd);
    const int per_bwd = (B + std::min(B, c->grid_bwd) - 1) / std::min(B, c->grid_bwd);
    // (conv/BN graphs: the stem's launches check their own grids and write x out themselves if a workgroup would own too many windows)
    const bool lazy_ok = c->generic ? g_stem_gathers(c) : (per_fwd <= kXMaxSamples && per_bwd <= kXMaxSamples);
    if (c->fused_input && lazy_ok && T <= 32 * kXRowWords && nm <= kXMaxMasks) {
      // descriptor-only batch: the first block's kernels gather from the stores (the labels / weights that
      // arrived in this mailbox are read in place)
      if (a.n_targets) {
        c->y_cur = a.y_src;
        c->sw_cur = a.sw_src;
      } else {
        int rcx = bring_targets(c);   // labels of an earlier descriptor-only batch stay valid past their mailbox slot
        if (rcx) return rcx;
      }
      c->lazy_a = a;
      c->x_lazy = true;
      c->lazy_slot = c->mail_cur;
      c->have_batch = B;
      return MWW_OK;
    }
    if (a.n_targets) {   // the assembly kernel below brings this batch's labels / weights into y / sw
      c->y_cur = c->y;
      c->sw_cur = c->sw;
    } else {
      int rcx = bring_targets(c);
      if (rcx) return rcx;
    }
    c->x_lazy = false;
  }
  {
    Launcher lp{c};
    lp.begin("assemble");
    hipLaunchKernelGGL(assemble_kernel, dim3(B * a.split), dim3(kThreads), 0, c->stream, a);
    lp.end();
  }
  HIPCHK(hipGetLastError());
  c->have_batch = B;
  return MWW_OK;
}

int mww_set_batch(mww_ctx* c, const float* hx, int B) {
  if (!c || !hx || B <= 0 || B > c->d.max_batch) return fail(MWW_ERR_INVALID, "bad batch size");
  HIPCHK(hipSetDevice(c->device));
  int rc = bring_targets(c);
  if (rc) return rc;
  c->x_lazy = false;
  rc = copy_in(c, c->x, hx, (size_t)B * c->d.frames * MWW_FEATURE_BINS * sizeof(float));
  if (!rc) c->have_batch = B;
  return rc;
}
int mww_get_batch(mww_ctx* c, float* hx, int B) {
  if (!c || !hx || B <= 0 || B > c->d.max_batch) return fail(MWW_ERR_INVALID, "bad batch size");
  HIPCHK(hipSetDevice(c->device));
  int rc = materialise_x(c);
  if (rc) return rc;
  return copy_out(c, hx, c->x, (size_t)B * c->d.frames * MWW_FEATURE_BINS * sizeof(float));
}

int mww_set_targets(mww_ctx* c, const float* hy, const float* hw, int B) {
  if (!c || !hy || !hw || B <= 0 || B > c->d.max_batch) return fail(MWW_ERR_INVALID, "bad batch size");
  HIPCHK(hipSetDevice(c->device));
  int rc = mail_begin(c);
  if (rc) return rc;
  char* mh = c->mail_host[c->mail_cur];
  memcpy(mh + c->mail_off_y, hy, (size_t)B * sizeof(float));
  memcpy(mh + c->mail_off_sw, hw, (size_t)B * sizeof(float));
  c->targets_in_mail = B;   // picked up by the next mww_assemble_batch, or copied at the next step
  c->have_targets = B;
  return MWW_OK;
}

int mww_assemble_prefetched(mww_ctx* c, mww_prefetcher* p, float* out_labels, float* out_weights) {
  if (!c || !p) return fail(MWW_ERR_INVALID, "null context / prefetcher");
  const mww_window* win = nullptr;
  const int32_t* masks = nullptr;
  const float *y = nullptr, *w = nullptr;
  int rc = mww_prefetch_acquire(p, &win, &masks, &y, &w, nullptr, nullptr);
  if (rc) return fail(rc, "the prefetcher's sampler failed (a provider's truncation strategy cannot form a fixed-length window)");
  int B = 0, ntm = 0, nfm = 0;
  mww_prefetch_shape(p, &B, &ntm, &nfm);
  rc = mww_set_targets(c, y, w, B);
  if (!rc) rc = mww_assemble_batch(c, win, masks, B, ntm, nfm);
  if (!rc && out_labels) memcpy(out_labels, y, (size_t)B * sizeof(float));
  if (!rc && out_weights) memcpy(out_weights, w, (size_t)B * sizeof(float));
  mww_prefetch_release(p);
  return rc;
}

int mww_train_step(mww_ctx* c, int B, float lr, int flags) {
  if (!c || B <= 0 || B > c->d.max_batch) return fail(MWW_ERR_INVALID, "bad batch size");
  if (c->have_batch < B || c->have_targets < B) return fail(MWW_ERR_STATE, "train step needs a batch and targets of at least B rows");
  HIPCHK(hipSetDevice(c->device));
  int rc = flush_targets(c);
  if (rc) return rc;
  if (c->lazy_slot != c->mail_cur) {   // a descriptor-only batch is gathered in place only while its mailbox slot is current
    rc = materialise_x(c);
    if (rc) return rc;
  }
  const bool apply = !(flags & MWW_STEP_NO_APPLY);
  if (apply) {
    c->step += 1;
    rc = push_hyper(c, adam_alpha(lr, c->step), (c->hook && c->reduce_grads) ? 1.0f / (float)c->world : 1.0f);
    if (rc) return rc;
  }
  const bool gen_dropout = c->generic && c->dropout > 0.f && !c->keep_explicit;
  if (gen_dropout) {
    // the mask generator reads this step's counter from the mailbox (a graph node cannot carry it)
    rc = mail_begin(c);
    if (rc) return rc;
    unsigned* h = reinterpret_cast<unsigned*>(c->mail_host[c->mail_cur] + c->mail_off_hyper);
    h[2] = (unsigned)(c->dropout_counter & 0xFFFFFFFFull);
    h[3] = (unsigned)(c->dropout_counter >> 32);
    c->dropout_counter += 1;
  }
  if (c->use_graphs && !c->profile && !c->hook) {   // the exchange hook enqueues foreign work: no capture
    // only the Adam / dropout nodes and the gather of a descriptor-only batch read the mailbox
    const int mail = (apply || gen_dropout || c->x_lazy) ? c->mail_cur : -1;
    // the accumulator parities of the statistics hand-over are baked into the captured kernel arguments
    const bool flips = c->bn_inline && (!c->generic || (c->g_inline_ok && !c->profile_split));
    const int par = (flips ? (4 | c->fpar | (c->gpar << 1)) : 0) | (c->x_lazy ? 8 : 0) | (c->y_cur != c->y ? 16 : 0) | (c->tail_roles ? 32 : 0) | (c->g_role_split ? 64 : 0) | (c->grid_g_auto ? 128 : 0) | (c->g_dgrad_share << 8) | (c->g_cap_fwd << 16) | (c->g_cap_bwd << 20) | (c->g_chunks << 24);
    hipGraphExec_t exec = nullptr;
    for (auto& g : c->graphs)
      if (g.B == B && g.flags == flags && g.mail == mail && g.par == par) exec = g.exec;
    if (exec && flips) {
      c->fpar ^= 1;
      c->gpar ^= 1;
    }
    if (!exec) {
      hipGraph_t graph;
      HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
      rc = step_sequence(c, B, flags);
      hipError_t e = hipStreamEndCapture(c->stream, &graph);
      if (rc) return rc;
      if (e != hipSuccess) return fail(MWW_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
      HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      HIPCHK(hipGraphDestroy(graph));
      c->graphs.push_back({B, flags, mail, par, exec});
    }
    HIPCHK(hipGraphLaunch(exec, c->stream));
    return mail_commit(c);
  }
  rc = step_sequence(c, B, flags);
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  return mail_commit(c);
}

int mww_apply_gradients(mww_ctx* c, float lr, float gscale) {
  if (!c) return fail(MWW_ERR_INVALID, "null context");
  HIPCHK(hipSetDevice(c->device));
  c->step += 1;
  int rc = push_hyper(c, adam_alpha(lr, c->step), gscale);
  if (rc) return rc;
  rc = enqueue_adam(c);
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  return mail_commit(c);
}

int mww_forward(mww_ctx* c, int B, int training, int update_metrics) {
  if (!c || B <= 0 || B > c->d.max_batch) return fail(MWW_ERR_INVALID, "bad batch size");
  if (c->have_batch < B) return fail(MWW_ERR_STATE, "forward needs a batch of at least B rows");
  if (update_metrics && c->have_targets < B) return fail(MWW_ERR_STATE, "metric update needs targets");
  HIPCHK(hipSetDevice(c->device));
  int rc = flush_targets(c);
  if (rc) return rc;
  if (c->lazy_slot != c->mail_cur) {
    rc = materialise_x(c);
    if (rc) return rc;
  }
  rc = enqueue_forward(c, B, training != 0, false, false, update_metrics != 0);
  if (rc) return rc;
  rc = join_side(c);
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  return mail_commit(c);
}

int mww_evaluate_windows(mww_ctx* c, const mww_window* windows, const float* labels, int64_t n, int batch) {
  if (!c || !windows || !labels || n < 0 || batch <= 0 || batch > c->d.max_batch) return fail(MWW_ERR_INVALID, "bad evaluation arguments");
  std::vector<float> ones((size_t)batch, 1.0f);
  int rc = MWW_OK;
  for (int64_t s = 0; s < n && !rc; s += batch) {
    const int b = (int)std::min<int64_t>(batch, n - s);
    rc = mww_set_targets(c, labels + s, ones.data(), b);
    if (!rc) rc = mww_assemble_batch(c, windows + s, nullptr, b, 0, 0);
    if (!rc) rc = mww_forward(c, b, 0, 1);
    c->bn_eval_ready = !c->generic;   // the weights cannot change between the batches of this call
  }
  c->bn_eval_ready = false;
  return rc;
}

int mww_read_outputs(mww_ctx* c, int B, float* probs, float* logits, float* loss) {
  if (!c || B <= 0 || B > c->d.max_batch) return fail(MWW_ERR_INVALID, "bad batch size");
  if (probs) HIPCHK(hipMemcpyAsync(probs, c->prob, (size_t)B * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (logits) HIPCHK(hipMemcpyAsync(logits, c->z, (size_t)B * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  std::vector<float> lp;
  if (loss) {
    lp.resize(B);
    HIPCHK(hipMemcpyAsync(lp.data(), c->loss_part, (size_t)B * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  if (loss) {
    double s = 0.0;
    for (int i = 0; i < B; ++i) s += lp[i];
    *loss = (float)s;
  }
  return MWW_OK;
}

int mww_metrics_read(mww_ctx* c, mww_metrics* out) {
  if (!c || !out) return fail(MWW_ERR_INVALID, "null argument");
  static_assert(sizeof(mww_metrics) == sizeof(MetricState), "metric layouts must match");
  return copy_out(c, out, c->metrics, sizeof(MetricState));
}
int mww_metrics_reset(mww_ctx* c) {
  if (!c) return fail(MWW_ERR_INVALID, "null context");
  HIPCHK(hipMemsetAsync(c->metrics, 0, sizeof(MetricState), c->stream));
  return MWW_OK;
}

void* mww_device_ptr(mww_ctx* c, int which) {
  if (!c) return nullptr;
  switch (which) {
    case MWW_BUF_PARAMS: return c->params;
    case MWW_BUF_GRADS: return c->grads;
    case MWW_BUF_BN_STATE: return c->bn_state;
    case MWW_BUF_X: return c->x;
    case MWW_HANDLE_STREAM: return c->stream;
    default: return nullptr;
  }
}

int64_t mww_debug_read(mww_ctx* c, const char* name, int B, float* host, int64_t cap) {
  if (!c || !name || !host || B <= 0 || B > c->d.max_batch) return fail(MWW_ERR_INVALID, "bad debug_read arguments");
  const float* src = nullptr;
  int64_t n = 0;
  const int nb = c->d.n_blocks;
  auto idx = [&](const char* prefix) -> int {
    const size_t pl = strlen(prefix);
    if (strncmp(name, prefix, pl) != 0) return -1;
    const int k = atoi(name + pl);
    return (k >= 1 && k <= nb && name[pl] >= '0' && name[pl] <= '9') ? k - 1 : -1;
  };
  int k;
  if (c->generic) {
    const int no = (int)c->G.size();
    auto gidx = [&](const char* prefix) -> int {
      const size_t pl = strlen(prefix);
      if (strncmp(name, prefix, pl) != 0 || name[pl] < '0' || name[pl] > '9') return -1;
      const int kk = atoi(name + pl);
      return (kk >= 1 && kk <= no) ? kk - 1 : -1;
    };
    if ((k = gidx("p")) >= 0) { src = c->G[k].p; n = (int64_t)B * c->G[k].tout * c->G[k].cout; }
    else if ((k = gidx("g")) >= 0) { src = c->G[k].g; n = (int64_t)B * c->G[k].tout * c->G[k].cout; }
    else if ((k = gidx("bn")) >= 0) { src = c->G[k].bn; n = (int64_t)9 * c->G[k].cout; }
    else if (!strcmp(name, "dz")) { src = c->dz; n = B; }
    else if (!strcmp(name, "keep")) { src = c->keep; n = (int64_t)B * c->t_last * c->c_last; }
    else if (!strcmp(name, "x")) {
      if (materialise_x(c)) return -1;
      src = c->x;
      n = (int64_t)B * c->d.frames * MWW_FEATURE_BINS;
    }
    else return fail(MWW_ERR_INVALID, std::string("unknown tensor name: ") + name);
    if (n > cap) return fail(MWW_ERR_INVALID, "host buffer too small");
    const int kpl = gidx("p") >= 0 ? gidx("p") : gidx("g");
    if (kpl >= 0 && g_planes(c, c->G[kpl]) > 1) {
      // a planar tensor is handed out interleaved [B][T][C], as the caller expects it
      const GOp& o = c->G[kpl];
      const int planes = g_planes(c, o);
      std::vector<float> tmp((size_t)B * o.tout * o.pc);
      for (int pl = 0; pl < planes; ++pl) {
        int rcp = copy_out(c, tmp.data(), src + (size_t)pl * g_pstride(c, o), tmp.size() * sizeof(float));
        if (rcp) return rcp;
        for (int64_t r = 0; r < (int64_t)B * o.tout; ++r)
          for (int cc = 0; cc < o.pc; ++cc) host[r * o.cout + pl * o.pc + cc] = tmp[(size_t)r * o.pc + cc];
      }
      return n;
    }
    int rcg = copy_out(c, host, src, (size_t)n * sizeof(float));
    return rcg ? rcg : n;
  }
  bool stored = false;   // p_k / g_k: bf16 in HBM under "storage_bf16", widened for the caller
  if ((k = idx("p")) >= 0) { src = c->L[k].p; n = (int64_t)B * c->L[k].tout * c->L[k].cout; stored = true; }
  else if ((k = idx("g")) >= 0) { src = c->L[k].g; n = (int64_t)B * c->L[k].tout * c->L[k].cout; stored = true; }
  else if ((k = idx("bn")) >= 0) { src = c->L[k].bn; n = (int64_t)9 * c->L[k].cout; }
  else if (!strcmp(name, "dz")) { src = c->dz; n = B; }
  else if (!strcmp(name, "a0")) { src = c->a0; n = (int64_t)B * c->L[0].tin * c->d.conv1_filters; }   // relu(conv1(x)) as the first block stored it
  else if (!strncmp(name, "clkf", 4) || !strncmp(name, "clkb", 4)) {
    // phase clocks of layer k (1-based) as raw 64-bit counters viewed as floats: 2048 x 8 x 2 words
    const int kk = atoi(name + 4);
    if (kk < 1 || kk > nb) return fail(MWW_ERR_INVALID, "bad layer");
    src = reinterpret_cast<const float*>(c->phase_clk + (size_t)(2 * (kk - 1) + (name[3] == 'b' ? 1 : 0)) * 2048 * kClkSlots);
    n = 2048 * kClkSlots * 2;
  }
  else if (!strcmp(name, "x")) { src = c->x; n = (int64_t)B * c->d.frames * MWW_FEATURE_BINS; }
  else return fail(MWW_ERR_INVALID, std::string("unknown tensor name: ") + name);
  if (n > cap) return fail(MWW_ERR_INVALID, "host buffer too small");
  if (stored && c->st_bf16) {
    std::vector<unsigned short> half((size_t)n);
    int rch = copy_out(c, half.data(), src, (size_t)n * sizeof(unsigned short));
    if (rch) return rch;
    for (int64_t i = 0; i < n; ++i) {
      const unsigned bits = (unsigned)half[(size_t)i] << 16;
      memcpy(host + i, &bits, 4);
    }
    return n;
  }
  int rc = copy_out(c, host, src, (size_t)n * sizeof(float));
  return rc ? rc : n;
}

int mww_set_option(mww_ctx* c, const char* name, int64_t v) {
  if (!c || !name) return fail(MWW_ERR_INVALID, "null argument");
  if (!strcmp(name, "graphs")) c->use_graphs = v != 0;
  else if (!strcmp(name, "profile")) {
    c->profile = v != 0;
    for (auto& e : c->prof) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    c->prof.clear();
  }
  else if (!strcmp(name, "ablate")) c->ablate = (int)v;
  else if (!strcmp(name, "side_stream")) c->use_side = v != 0;
  else if (!strcmp(name, "bn_inline")) c->bn_inline = v != 0;
  else if (!strcmp(name, "graph_role_split")) c->g_role_split = v != 0;
  else if (!strcmp(name, "graph_static_shapes")) c->g_static = v != 0;
  else if (!strcmp(name, "graph_planar")) c->g_planar = v != 0;
  else if (!strcmp(name, "graph_fwd_wg_per_cu")) { if (v < 1 || v > 8) return fail(MWW_ERR_INVALID, "graph_fwd_wg_per_cu must be 1..8"); c->g_cap_fwd = (int)v; }
  else if (!strcmp(name, "graph_bwd_wg_per_cu")) { if (v < 1 || v > 8) return fail(MWW_ERR_INVALID, "graph_bwd_wg_per_cu must be 1..8"); c->g_cap_bwd = (int)v; }
  else if (!strcmp(name, "graph_frame_chunks")) { if (v < 0 || v > 4) return fail(MWW_ERR_INVALID, "graph_frame_chunks must be 0..4"); c->g_chunks = (int)v; }
  else if (!strcmp(name, "graph_dgrad_share")) { if (v < 10 || v > 90) return fail(MWW_ERR_INVALID, "graph_dgrad_share must be 10..90"); c->g_dgrad_share = (int)v; }
  else if (!strcmp(name, "tail_roles")) c->tail_roles = v != 0;
  else if (!strcmp(name, "bce_from_logits")) c->bce_clipped = v == 0;
  else if (!strcmp(name, "grad_buckets")) { if (v < 1 || v > 2) return fail(MWW_ERR_INVALID, "grad_buckets must be 1 or 2"); c->grad_buckets = (int)v; }
  else if (!strcmp(name, "fused_input")) {
    c->fused_input = v != 0;
    if (!v) { int rc = materialise_x(c); if (rc) return rc; }
  }
  else if (!strcmp(name, "assemble_split")) { if (v < 1 || v > 8) return fail(MWW_ERR_INVALID, "assemble_split out of range"); c->asm_split = (int)v; }
  else if (!strcmp(name, "profile_split")) c->profile_split = v != 0;
  else if (!strcmp(name, "pointwise_bf16")) {
    if (c->generic && v) return fail(MWW_ERR_UNSUPPORTED, "the conv/BN graph kernels have no bf16 mode");
    { std::string why; if (v && !shape_supported(c->d, &why, true)) return fail(MWW_ERR_UNSUPPORTED, "no bf16 mode for this topology: " + why); }
    c->pw_bf16 = v != 0;
    if (!v) c->st_bf16 = false;
  }
  else if (!strcmp(name, "storage_bf16")) {
    if (c->generic && v) return fail(MWW_ERR_UNSUPPORTED, "the conv/BN graph kernels have no bf16 mode");
    { std::string why; if (v && !shape_supported(c->d, &why, true)) return fail(MWW_ERR_UNSUPPORTED, "no bf16 mode for this topology: " + why); }
    c->st_bf16 = v != 0;
    if (v) c->pw_bf16 = true;
  }
  else if (!strcmp(name, "bwd_wide")) c->bwd_wide = v != 0;
  else if (!strcmp(name, "conv1_x6")) c->conv1_x6 = v != 0;
  else if (!strcmp(name, "conv1_x6_fwd")) c->conv1_x6_fwd = v != 0;
  else if (!strcmp(name, "bwd_first_wide")) c->bwd_first_wide = v != 0;
  else if (!strcmp(name, "grid_fwd")) { if (v < 1 || v > c->n_cu * 4) return fail(MWW_ERR_INVALID, "grid_fwd out of range"); c->grid_fwd = (int)v; }
  else if (!strcmp(name, "grid_bwd")) { if (v < 1 || v > c->n_cu * 2) return fail(MWW_ERR_INVALID, "grid_bwd out of range"); c->grid_bwd = (int)v; }
  else if (!strcmp(name, "grid_graph")) {   // 0: per-launch grids by occupancy (default); > 0: this many workgroups per launch
    if (v < 0 || v > c->n_cu * 4) return fail(MWW_ERR_INVALID, "grid_graph out of range");
    c->grid_g_auto = v == 0;
    if (v > 0) c->grid_g = (int)v;
  }
  else if (!strcmp(name, "dropout_seed")) { c->dropout_seed = (unsigned long long)v; c->dropout_counter = 0; }
  else if (!strcmp(name, "grid_head")) { if (v < 1 || v > c->n_cu * 4) return fail(MWW_ERR_INVALID, "grid_head out of range"); c->grid_head = (int)v; }
  else return fail(MWW_ERR_INVALID, std::string("unknown option: ") + name);
  for (auto& g : c->graphs) (void)hipGraphExecDestroy(g.exec);
  c->graphs.clear();
  return MWW_OK;
}

int mww_profile_read(mww_ctx* c, char* names, int names_cap, float* ms, int cap) {
  if (!c) return fail(MWW_ERR_INVALID, "null context");
  HIPCHK(hipStreamSynchronize(c->stream));
  int n = 0, pos = 0;
  for (auto& e : c->prof) {
    if (n >= cap) break;
    float t = 0.f;
    (void)hipEventElapsedTime(&t, e.a, e.b);
    ms[n] = t;
    const int len = (int)e.name.size();
    if (names && pos + len + 1 < names_cap) {
      memcpy(names + pos, e.name.c_str(), len);
      names[pos + len] = '\n';
      pos += len + 1;
    }
    ++n;
  }
  if (names && names_cap > 0) names[pos < names_cap ? pos : names_cap - 1] = 0;
  for (auto& e : c->prof) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  c->prof.clear();
  return n;
}

}  // extern "C"

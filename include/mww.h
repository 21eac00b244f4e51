/* mww.h — C ABI of libmww_hip.so: the MI355X-native (gfx950) train-step engine for
 * microWakeWord's MixedNet and Inception classifiers.
 *
 * The reference (kahrendt/microWakeWord) has no FFI layer: its hot path is duck-typed Python
 * (microwakeword/train.py:249-299) calling numpy batch assembly (microwakeword/data.py:497-597)
 * and Keras `train_on_batch` on the graph built by microwakeword/mixednet.py:278-386 or
 * microwakeword/inception.py:232-340.  This
 * header is the boundary a maintainer would bind (ctypes / cffi / pybind — see INTEGRATION.md):
 * plain pointers and sizes, no torch / Python types.  Each entry point names the reference
 * interface it replaces.
 *
 * Conventions: every call returns MWW_OK (0) or a negative error code; the message is available
 * from mww_last_error() (thread-local).  The caller owns every host pointer; the library owns
 * all device memory.  A context is bound to one device and one HIP stream and is NOT thread
 * safe (one context per rank, one host thread).  Calls enqueue work on the context's stream;
 * only the mww_read_* / mww_get_* / mww_synchronize calls wait for the device.
 */
#ifndef MWW_H
#define MWW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MWW_OK 0
#define MWW_ERR_INVALID -1      /* bad argument / unsupported configuration */
#define MWW_ERR_HIP -2          /* a HIP runtime call failed */
#define MWW_ERR_UNSUPPORTED -3  /* model shape has no compiled kernel instantiation (mww_create: fall back to mww_create_convnet) */
#define MWW_ERR_STATE -4        /* call sequence error (e.g. train step before a batch) */

#define MWW_MAX_BLOCKS 8
#define MWW_MAX_STORES 64
#define MWW_FEATURE_BINS 40

typedef struct mww_ctx mww_ctx;

/* ---- model: replaces mixednet.model(flags, shape, batch_size) (mixednet.py:278-386) with the specialised
 * MFMA block kernels.  Covers conv1 -> ReLU -> n x [MixConv + 1x1 + BN + ReLU] -> Flatten -> Dense for the
 * (filters, kernel size) shapes instantiated in csrc/mww_lib.hip; every other flag combination (repeat_in_block,
 * residual_connection, spatial_attention, pooled, other widths ...) is expressed as a mww_convnet_desc below.
 * `block_kernel[i]` is the aligned depthwise length of block i (= the LAST listed MixConv kernel
 * size, mixednet.py:227); smaller MixConv groups are expressed by the host as zero leading taps
 * plus a gradient mask (mww_set_grad_mask), SURVEY §A.2.
 * Flat trainable-parameter order ("native order" == Keras get_weights() order without the BN
 * moving statistics):
 *   conv1.kernel[K1*40*C1] ; per block: dw.kernel[K*Cin] dw.bias[Cin] pw.kernel[Cin*Cout]
 *   bn.gamma[Cout] bn.beta[Cout] ; dense.kernel[T_last*C_last] dense.bias[1]
 * BN state order: per block moving_mean[Cout] moving_variance[Cout]. */
typedef struct {
  int32_t frames;               /* T: spectrogram_length (model_train_eval.py:82-88) */
  int32_t conv1_filters;        /* --first_conv_filters  (mixednet.py:76-81) */
  int32_t conv1_kernel;         /* --first_conv_kernel_size */
  int32_t conv1_stride;         /* --stride: time stride of the first convolution */
  int32_t n_blocks;
  int32_t block_filters[MWW_MAX_BLOCKS];  /* --pointwise_filters */
  int32_t block_kernel[MWW_MAX_BLOCKS];   /* max/last of --mixconv_kernel_sizes per block */
  int32_t max_batch;            /* largest batch any call will use */
} mww_mixednet_desc;

const char* mww_version(void);
const char* mww_last_error(void);
int mww_device_count(void);

/* 1 if every block of `desc` has a specialised MFMA block kernel (bf16 != 0: in the bf16 modes too), 0 if the model has to
 * run on the conv / depthwise graph kernels of mww_create_convnet (mww_last_error() says which block is not covered, or that the
 * classifier head holds fewer final frames than the model has: 768 / 504 / 384 at 32 / 48 / 64 channels).
 * Needs no device: it answers from the build-time shape table (csrc/block_launch.hip.h).  mww_create refuses the same shapes
 * with MWW_ERR_UNSUPPORTED. */
int mww_block_kernels_cover(const mww_mixednet_desc* desc, int bf16);

/* stream: a hipStream_t to run on (e.g. torch.cuda.current_stream().cuda_stream) or NULL for a
 * private non-blocking stream. */
int mww_create(const mww_mixednet_desc* desc, int device, void* stream, mww_ctx** out);

/* ---- conv -> BN/SSN -> ReLU graphs: replaces inception.model(flags, shape, batch_size)
 * (microwakeword/inception.py:232-340) and its building blocks conv2d_bn / conv2d_bn_delay (:46-141:
 * Conv2D(k x 1, dilation, valid, use_bias=False) -> BatchNormalization or SubSpectralNormalization
 * (layers/sub_spectral_normalization.py:49-61: channel c uses slot c % groups) -> ReLU) and the
 * three-branch block's StridedDrop + Concatenate (:143-209; strided_drop.py:42 drops LEADING frames).
 * The head is Flatten -> Dropout(rate) -> Dense(1, sigmoid) (:330-338) on the LAST op.
 * Ops are listed in layer-creation order, so the flat parameter vector is Keras get_weights() order:
 *   per op: kernel[k*Cin*filters] (depthwise: [k*filters]) then gamma[slots] beta[slots] (MWW_NORM_BN) or
 *   bias[filters] (MWW_NORM_BIAS) ; dense.kernel[T_last*C_last] dense.bias[1]
 *   (slots = bn_groups if bn_groups > 1 else filters); BN state: per BN op moving_mean[slots] moving_variance[slots].
 * Every other entry point of this header works on such a context exactly as on a MixedNet one. */
#define MWW_MAX_GRAPH_OPS 48
#define MWW_MAX_OP_SOURCES 3
typedef struct {
  int32_t n_src;                        /* inputs concatenated along channels, in this order */
  int32_t src[MWW_MAX_OP_SOURCES];      /* producing op (index < this op's) or -1 for the spectrogram */
  int32_t src_drop[MWW_MAX_OP_SOURCES]; /* leading frames of that input dropped to align the branches */
  int32_t src_c0[MWW_MAX_OP_SOURCES];   /* first channel of the slice of that input this op reads ... */
  int32_t src_cn[MWW_MAX_OP_SOURCES];   /* ... and its width (0 = the whole tensor).  Sibling convolutions that read
                                           the same input (Inception's three 1x1 branch heads) can be fused by the caller
                                           into one op with the concatenated filters; each consumer then names its slice.
                                           The slices different consumers take from one producer must be identical or
                                           disjoint and together cover all of its channels. */
  int32_t kernel, dilation, filters;
  int32_t bn_groups;                    /* 1 = BatchNormalization, g > 1 = SubSpectralNormalization(g) */
  /* the four fields below default to 0 = the Inception vocabulary (dense conv, stride 1, BN/SSN, ReLU); the others
   * express any MixedNet flag combination the specialised block kernels do not cover (mixednet.py:307-360:
   * first conv with --stride and no BN, MixConv depthwise + bias with no activation, repeat_in_block > 1, ...) */
  int32_t kind;                         /* MWW_OP_CONV: k x 1 convolution over all input channels;
                                           MWW_OP_DEPTHWISE: per-channel k x 1 taps [k][C] (one source, filters == its channels) */
  int32_t stride;                       /* time stride (0/1 = none); only for ops fed by the spectrogram */
  int32_t norm;                         /* MWW_NORM_BN (gamma, beta + moving statistics), MWW_NORM_BIAS (bias[filters]), MWW_NORM_NONE */
  int32_t act;                          /* MWW_ACT_RELU or MWW_ACT_LINEAR */
  int32_t residual;                     /* 1 + index of an earlier conv + BN + linear op whose normalised output is added to
                                           this op's normalised output before the activation (mixednet.py:340-358); 0 = none */
  int32_t residual_drop;                /* leading frames of that op dropped to align it (StridedDrop) */
} mww_conv_bn_op;
#define MWW_OP_CONV 0
#define MWW_OP_DEPTHWISE 1
#define MWW_NORM_BN 0
#define MWW_NORM_BIAS 1
#define MWW_NORM_NONE 2
#define MWW_ACT_RELU 0
#define MWW_ACT_LINEAR 1
typedef struct {
  int32_t frames;
  int32_t n_ops;
  mww_conv_bn_op ops[MWW_MAX_GRAPH_OPS];
  float dropout;                        /* --dropout (inception.py:330), active in the train step only */
  int32_t max_batch;
  /* MixedNet's optional heads (mixednet.py:234-275,362-384), applied to the last op's activations when more than one
   * frame remains; both 0 = Flatten -> Dense.  With attention the flat parameter vector carries the attention
   * kernel[4*2] (Keras [4,1,2,1]) between the last op and dense.kernel. */
  int32_t head_attention;               /* --spatial_attention: SpatialAttention(kernel_size = 4) */
  int32_t head_pool;                    /* --pooled: 0 none, 1 average (AveragePooling2D), 2 max (--max_pool) over the remaining frames */
} mww_convnet_desc;
int mww_create_convnet(const mww_convnet_desc* desc, int device, void* stream, mww_ctx** out);
/* Dropout keep decisions for the next train steps, [B][T_last*C_last] bytes (0 = dropped); NULL
 * returns to the built-in counter-based generator (mww_set_option "dropout_seed").  Keras draws the
 * mask from a generator that cannot be reproduced outside TensorFlow, so parity tests inject it. */
int mww_set_dropout_mask(mww_ctx* ctx, const uint8_t* keep, int B);
void mww_destroy(mww_ctx* ctx);
int mww_synchronize(mww_ctx* ctx);

int64_t mww_num_params(const mww_ctx* ctx);    /* trainable scalars (22177 for default mixednet) */
int64_t mww_num_bn_state(const mww_ctx* ctx);  /* moving mean/variance scalars (384) */

/* ---- weights: replace model.get_weights()/set_weights()/save_weights()/load_weights()
 * (train.py:336-338,391-399,448-450; model_train_eval.py:420-426) at the array level. */
int mww_set_params(mww_ctx* ctx, const float* host, int64_t n);
int mww_get_params(mww_ctx* ctx, float* host, int64_t n);
int mww_set_bn_state(mww_ctx* ctx, const float* host, int64_t n);
int mww_get_bn_state(mww_ctx* ctx, float* host, int64_t n);
int mww_set_grad_mask(mww_ctx* ctx, const float* host_mask01, int64_t n);
/* optimizer slots of tf.keras.optimizers.Adam (train.py:207): m, v and the iteration count;
 * replaces the tf.train.Checkpoint(optimizer=...) payload of train.py:229-233 */
int mww_set_opt_state(mww_ctx* ctx, const float* m, const float* v, int64_t n, int64_t step);
int mww_get_opt_state(mww_ctx* ctx, float* m, float* v, int64_t n, int64_t* step);
int mww_get_grads(mww_ctx* ctx, float* host, int64_t n);  /* flat gradient of the last step */

/* ---- feature stores: replace the RaggedMmap page-cache reads of data.py:255-258.  A store is the
 * flat concatenation of its samples ([T_i][40], uint16 raw micro-frontend values or float32
 * already scaled), uploaded once and kept resident in HBM. */
#define MWW_DTYPE_U16 0
#define MWW_DTYPE_F32 1
int mww_upload_store(mww_ctx* ctx, int store_id, const void* host_data, int64_t n_elems, int dtype);

/* ---- batch assembly: replaces FeatureHandler.get_data("training") materialisation
 * (data.py:555-597: fixed_length_spectrogram :74-118, uint16->f32 * 0.0390625 :268-269,
 * spec_augment :32-71, final shuffle :591-597).  All random choices are made on the host
 * (mww_sample_training_batch) and arrive here as integers. */
typedef struct {
  int32_t store;      /* store id */
  int32_t pad_rows;   /* zero frames in front (left pad, data.py:107-113) */
  int32_t copy_rows;  /* frames copied */
  int32_t reserved;
  int64_t src_elem;   /* element offset of the first copied frame inside the store */
} mww_window;
/* masks: [B][n_time_masks + n_freq_masks][2] = (start, width); time masks first.
 * Defines x[B][T][40] float32 of the context's batch buffer.  With the "fused_input" option (the default) only the
 * descriptors are uploaded and the kernels that read the spectrogram gather from the stores (the specialised MixedNet
 * first-block kernels; the stem of a conv/BN graph when it is the static 5 x 40 shape of the default Inception and the
 * window has at most 200 frames); the buffer itself is filled when any other reader needs it (other graphs,
 * mww_get_batch, mww_debug_read "x", a batch reused after its mailbox slot has moved on). */
int mww_assemble_batch(mww_ctx* ctx, const mww_window* windows, const int32_t* masks, int B, int n_time_masks,
                       int n_freq_masks);
/* ready-made batch: the `x` argument of Keras train_on_batch / evaluate (train.py:295-299,50-58) */
int mww_set_batch(mww_ctx* ctx, const float* host_x, int B);
int mww_get_batch(mww_ctx* ctx, float* host_x, int B);
/* labels and per-sample weights (penalty_weight * class weight, train.py:288-293) */
int mww_set_targets(mww_ctx* ctx, const float* host_y, const float* host_w, int B);

/* ---- the train step: replaces model.train_on_batch (train.py:295-299): forward(training=True),
 * weighted Keras BCE, backward, Adam(lr), BN moving statistics, metric update.
 * flags: MWW_STEP_NO_APPLY stops after the flat gradient is assembled (data-parallel: all-reduce
 * mww_device_ptr(MWW_BUF_GRADS), then mww_apply_gradients). */
#define MWW_STEP_NO_APPLY 1
#define MWW_STEP_NO_METRICS 2
int mww_train_step(mww_ctx* ctx, int B, float learning_rate, int flags);
int mww_apply_gradients(mww_ctx* ctx, float learning_rate, float grad_scale);

/* ---- data-parallel exchange.  The caller owns the communicator (RCCL through
 * torch.distributed, one process per GPU); the library calls back whenever a buffer has to be summed
 * over the ranks.  The callback must ENQUEUE an in-place sum all-reduce of device_buf[0..n) (no host
 * synchronisation needed) and return 0.  `flags` says how the exchange is ordered:
 *   MWW_EXCHANGE_IN_ORDER  the reduced values are read by the next launch on the context's stream: the all-reduce has to
 *                          be complete, in stream order, when the callback's work is reached (e.g. a blocking-in-stream
 *                          dist.all_reduce issued while the context's stream is the current stream);
 *   MWW_EXCHANGE_DEFERRED  a finished gradient bucket: the all-reduce may run on the communicator's own stream next to
 *                          the backward kernels that follow (SURVEY §8e "two buckets ... overlaps the remaining backward
 *                          blocks"); it only has to be complete when the next MWW_EXCHANGE_FLUSH call returns its work;
 *   MWW_EXCHANGE_FLUSH     device_buf = NULL, n = 0: order the context's stream after every deferred exchange.
 *   sync_bn = 1: BatchNorm statistics are exchanged in every train step — per BN layer one all-reduce of
 *                (sum x, sum x^2) in the forward and one of (sum g, sum g*xhat) in the backward — so the W
 *                ranks normalise over the GLOBAL batch exactly like the single-device reference ("parity
 *                mode"; 2 x layers tiny all-reduces on the critical path, no hipGraph replay).
 *   sync_bn = 0: local-batch statistics ("throughput mode").
 *   reduce_grads = 1: mww_train_step also all-reduces the flat gradient through the callback and applies
 *                Adam to the rank average, i.e. it is the complete data-parallel step.  With the specialised MixedNet
 *                kernels and option "grad_buckets" 2 the gradient goes in two buckets: [dense + the last two
 *                blocks] as soon as their backward kernels are enqueued (deferred), the rest after the first block's
 *                (default 1 = one exchange after the backward pass: the faster schedule at world size 1, and the only
 *                one measured so far).
 * fn = NULL removes the hook. */
#define MWW_EXCHANGE_IN_ORDER 0
#define MWW_EXCHANGE_DEFERRED 1
#define MWW_EXCHANGE_FLUSH 2
typedef int (*mww_allreduce_fn)(void* user, float* device_buf, int64_t n, int flags);
int mww_set_allreduce_hook(mww_ctx* ctx, mww_allreduce_fn fn, void* user, int world_size, int sync_bn, int reduce_grads);

/* ---- the same exchange with RCCL called from the library (SURVEY 8b `mww_allreduce_init`, 8e): one process per GPU,
 * rank 0 obtains a unique id (ncclGetUniqueId) and hands its MWW_UNIQUE_ID_BYTES bytes to every rank by whatever channel
 * the launcher has (torch.distributed store / broadcast, MPI, a file); every rank then calls mww_allreduce_init, which
 * joins the communicator (collective: returns when all ranks have called it) and installs an internal exchange in place of
 * the callback above with reduce_grads = 1: in-order exchanges are ncclAllReduce calls on the context's stream, a deferred
 * bucket runs on a library-owned side stream ordered by events - no callback into the host language per step.
 * librccl.so is bound at run time (dlopen), so a single-GPU user needs no RCCL.  mww_allreduce_destroy leaves the
 * communicator (also done by mww_destroy). */
#define MWW_UNIQUE_ID_BYTES 128
int mww_allreduce_unique_id(void* out_id, int capacity);
int mww_allreduce_init(mww_ctx* ctx, int rank, int world_size, const void* unique_id, int sync_bn);
int mww_allreduce_destroy(mww_ctx* ctx);
/* ranks of the library-owned communicator as RCCL reports them (ncclCommCount); 0 without one: a bench line can state how
 * many GPUs really took part in its collectives. */
int mww_allreduce_world(mww_ctx* ctx);

/* ---- inference forward on the current batch: replaces model(x, training=...) /
 * model.evaluate's per-batch forward (train.py:50-58). training=1 uses batch statistics without
 * touching the moving averages. update_metrics=1 also accumulates the compiled metrics. */
int mww_forward(mww_ctx* ctx, int B, int training, int update_metrics);

/* ---- evaluation of a whole window list: replaces model.evaluate(x, y, batch_size=1024) over the result of
 * FeatureHandler.get_data(<validation / testing mode>) (train.py:42-58,75-96; the ambient sets arrive as the 100 ms-stride
 * windows of data.py:301-311).  `windows` / `labels` ([n]) are host arrays; the library walks them in batches of `batch`:
 * descriptor upload, gather from the HBM-resident stores, inference forward (moving statistics folded once for the whole
 * call), threshold / AUC / loss counters accumulated on the device (read them with mww_metrics_read).  One call per
 * evaluation instead of three per batch. */
int mww_evaluate_windows(mww_ctx* ctx, const mww_window* windows, const float* labels, int64_t n, int batch);

/* outputs of the last train step / forward (waits for the stream). Any pointer may be NULL. */
int mww_read_outputs(mww_ctx* ctx, int B, float* probs, float* logits, float* loss);

/* ---- metrics: the nine compiled Keras metrics of train.py:209-221 as raw cumulative counters;
 * the host derives accuracy/recall/precision/tp..fn@101/auc@200/loss from them. */
typedef struct {
  uint64_t hist101[2][101]; /* [label][bucket], bucket = ceil(p*100)-1 (Keras evenly spaced thresholds) */
  uint64_t hist200[2][200]; /* AUC buckets, ceil(p*199)-1 clamped at 0 */
  uint64_t n, correct, tp5, fp5, fn5, pos, neg;
  double bce_sum;           /* unweighted Keras BCE summed over samples */
} mww_metrics;
int mww_metrics_read(mww_ctx* ctx, mww_metrics* out);
int mww_metrics_reset(mww_ctx* ctx);

/* ---- device buffers for the host's collectives (torch.distributed over RCCL) */
#define MWW_BUF_PARAMS 0
#define MWW_BUF_GRADS 1
#define MWW_BUF_BN_STATE 2
#define MWW_BUF_X 3
#define MWW_HANDLE_STREAM 4   /* not a buffer: the hipStream_t the context enqueues on (to order foreign work against it) */
void* mww_device_ptr(mww_ctx* ctx, int which);

/* named internal tensors for parity tests ("p1".."p8" pre-BN block outputs, "g1".. gradients at
 * the BN outputs, "bn<k>" the folded BN rows of block k, "a0" = relu(conv1(x)) as the first block stored it, "dz", "x");
 * returns the element count or a negative error */
int64_t mww_debug_read(mww_ctx* ctx, const char* name, int B, float* host, int64_t capacity);

/* options: "graphs" (0/1 replay the step from a hipGraph), "grid_fwd", "grid_bwd", "grid_head", "grid_graph",
 * "dropout_seed", "pointwise_bf16" (0 = exact fp32 MFMA, the default; 1 = the MixedNet 1x1 convolutions and
 * their two backward contractions take bf16-rounded operands with fp32 accumulation — BASELINE configs[4]),
 * "storage_bf16" (1 = additionally the block outputs p_k and the stashed gradients g_k live in HBM as bf16, every sum
 * stays fp32 and is taken from the unrounded values; implies pointwise_bf16, cleared by pointwise_bf16 = 0),
 * "fused_input" (default 1: mww_assemble_batch uploads descriptors only and the kernels that read the spectrogram - the
 * specialised MixedNet first-block kernels, the gathering stem of the default Inception graph - gather / scale / mask their
 * rows from the stores; x is materialised on demand — same values),
 * "bn_inline" (default 1: BN sums travel in fp64 accumulator rows and are folded by their first consumer instead of
 * by finalize launches, in the MixedNet block kernels and in conv/BN graphs whose ops are convolutions with a BatchNorm (or
 * nothing) and depthwise ops with a bias (or nothing), without residual branches and attention / pooled heads; forced off by
 * the sync-BN exchange hook), "tail_roles" (default 1, with bn_inline: the dense-weight
 * gradient and the metric update ride in the gradient-assembly launch), "bce_from_logits" (default 1: the loss is the
 * logits form Keras 3 evaluates for a sigmoid output, 0: clipped probability form), "graph_role_split" (default 1, conv/BN
 * graph contexts with bn_inline: launches that hold several roles - twin ops, weight + data gradient - divide the launch's
 * workgroups between the roles; "graph_dgrad_share" = percent of an op's workgroups that form its data gradient, 10..90),
 * "grid_graph" (conv/BN graph contexts: 0 = default, with bn_inline every launch takes the
 * grid its own LDS tile and registers let the CUs hold - at most "graph_fwd_wg_per_cu" / "graph_bwd_wg_per_cu" workgroups per
 * CU, default 4 / 4; > 0 = that many workgroups per launch; without bn_inline one grid of 3 per CU, because a tensor's
 * partial statistics rows are shared by its launches),
 * "graph_frame_chunks" (conv/BN graph contexts with bn_inline: the 1x1 ops - and the forward convolution of any op - process
 * a window as up to 4 frame chunks with correspondingly smaller LDS tiles; 0 = off, 1 = automatic, 2..4 = that many; default 1 for graphs
 * with depthwise ops (MixedNet flag sets: -7 % step time measured), 0 for pure convolution graphs (Inception: +0.5 ... +8 %)),
 * "graph_static_shapes" (conv/BN graph contexts: 1 = ops whose shape - kernel length, sources' widths and row lengths - has a
 * compile-time instantiation use it, the default; 0 = the run-time-shape kernels for every op),
 * "graph_planar" (conv/BN graph contexts: 1 = a 30- / 48-filter op whose consumers all read one of its equal channel slices keeps its
 * tensors one plane per slice, the default; 0 = interleaved),
 * "bwd_wide" (default 1: the block backward kernels of square 48- / 64-wide blocks - and of stride-3 first blocks - as 512-thread
 * workgroups; 0 = 256 threads),
 * "conv1_x6" (default 1, stride-1 first convolutions: the conv1 weight gradient as six bf16 slice products per fp32 product on the
 * bf16 matrix pipe, fp32-grade; 0 = exact-fp32 MFMA), "conv1_x6_fwd" (default 0: the same form for the first convolution of the
 * forward kernel - measured slower), "bwd_first_wide" (default 0, with conv1_x6: the 512-thread form of the stride-1 first block's
 * backward kernel - measured equal),
 * "grad_buckets" (data-parallel step: 1 = one exchange after the backward pass, the default; 2 =
 * overlapped two-bucket gradient exchange), "assemble_split" (workgroups per window of the
 * assembly kernel), "side_stream", "profile", "profile_split", "ablate" (profiling switches) */
int mww_set_option(mww_ctx* ctx, const char* name, int64_t value);

/* per-kernel timing of the last N steps measured with HIP events on the context's stream:
 * enable with mww_set_option("profile", 1); names/ms are returned in launch order. */
int mww_profile_read(mww_ctx* ctx, char* names, int names_capacity, float* ms, int capacity);

/* ---- host sampler: replaces the RNG-consuming part of FeatureHandler.get_data("training")
 * (data.py:540-569,591-595) bit-exactly; see csrc/sampler.cpp. */
#define MWW_STRATEGY_RANDOM 0
#define MWW_STRATEGY_TRUNCATE_START 1
#define MWW_STRATEGY_TRUNCATE_END 2
#define MWW_STRATEGY_FIXED_RIGHT_CUTOFF 3
#define MWW_STRATEGY_NONE 4
typedef struct {
  int32_t n_providers;            /* providers that have training samples, reference order */
  const double* sampling_weight;  /* [n_providers] */
  const int32_t* strategy;        /* [n_providers] MWW_STRATEGY_* */
  const int64_t* set_offsets;     /* [n_providers+1] into the set_* arrays */
  const int32_t* set_store;       /* store id of each entry of feature_sets["training"] (shuffled order) */
  const int64_t* set_src_elem;    /* element offset of the sample inside its store */
  const int32_t* set_len;         /* frames of the sample */
  const int32_t* cutoff_offsets;  /* [n_providers+1] */
  const int32_t* cutoffs;         /* fixed_right_cutoffs, concatenated */
} mww_sampler_desc;
/* py_state / np_state: 624 MT19937 words + position (625 uint32), advanced in place.
 * out_order is the final np.random.shuffle permutation: output slot j of the batch is draw
 * out_order[j].  apply_order = 0 leaves windows/masks/provider/sample in DRAW order, 1 permutes them
 * into the final batch order.  default_strategy < 0 means "default" (per provider). */
int mww_sample_training_batch(const mww_sampler_desc* d, uint32_t* py_state, uint32_t* np_state, int B, int T, int tmax,
                              int tcount, int fmax, int fcount, int32_t default_strategy, int32_t apply_order,
                              mww_window* out_windows, int32_t* out_masks, int32_t* out_provider, int32_t* out_sample,
                              int32_t* out_order);
int mww_rng_selftest(uint32_t* state, int which, int n, double* out_real, uint32_t* out_int, uint32_t bound);

/* ---- batch prefetcher: the draws of FeatureHandler.get_data("training") for the NEXT steps made by a worker thread
 * while the launching thread enqueues the current one (the reference alternates get_data and train_on_batch on one
 * thread, train.py:276-299; at 0.3 ms per step the 0.16 ms the draws of 1024 windows take is most of the host's share).
 * The worker owns private copies of the two MT19937 streams (py_state / np_state: 625 words each, as above) and calls
 * mww_sample_training_batch(..., apply_order = 1) in sequence, so batch n is the batch the synchronous path draws from
 * those states with its n-th call.  provider_label / provider_weight: label and per-sample weight (penalty_weight x
 * class weight, train.py:288-293) of each provider's windows.  depth = batches drawn ahead (1..16).
 * Host-only objects: no device or context is involved until mww_assemble_prefetched. */
typedef struct mww_prefetcher mww_prefetcher;
int mww_prefetch_create(const mww_sampler_desc* d, const float* provider_label, const float* provider_weight,
                        const uint32_t* py_state, const uint32_t* np_state, int B, int T, int tmax, int tcount, int fmax,
                        int fcount, int32_t default_strategy, int depth, mww_prefetcher** out);
/* The same with class and penalty weights kept apart: the reference multiplies penalty[B] by class_weight(y)[B,1]
 * (microwakeword/train.py:288-293), a [B,B] matrix W[i,j] = penalty_j * cw(y_i) that Keras reduces against the [B] per-sample
 * losses.  provider_weight = penalty weights; provider_class_weight = class weight of each provider's label;
 * broadcast 0: w_j = penalty_j * cw_j (then identical to mww_prefetch_create with the products);
 *           1: w_j = penalty_j * mean_i cw_i over the batch ("keras_last_axis": the reference's arithmetic, the default of
 *              microwakeword_amd.train); 2: w_i = cw_i * mean_j penalty_j ("keras_first_axis").  Means in float64. */
int mww_prefetch_create_weighted(const mww_sampler_desc* d, const float* provider_label, const float* provider_weight,
                                 const float* provider_class_weight, int broadcast, const uint32_t* py_state,
                                 const uint32_t* np_state, int B, int T, int tmax, int tcount, int fmax, int fcount,
                                 int32_t default_strategy, int depth, mww_prefetcher** out);
/* waits for the next batch; the arrays ([B] windows, [B][tcount+fcount][2] masks, [B] labels / weights / provider /
 * index of the sample inside its provider's training set) stay valid until mww_prefetch_release.  Any pointer may be NULL. */
int mww_prefetch_acquire(mww_prefetcher* p, const mww_window** windows, const int32_t** masks, const float** labels,
                         const float** weights, const int32_t** provider, const int32_t** sample);
int mww_prefetch_release(mww_prefetcher* p);
/* the two streams as they stood after the last batch handed out (batches drawn ahead of it do not count): what the
 * synchronous sampler continues from when the prefetcher is dropped.  Returns the number of batches handed out. */
int64_t mww_prefetch_rng_state(mww_prefetcher* p, uint32_t* py_state, uint32_t* np_state);
int mww_prefetch_shape(const mww_prefetcher* p, int* B, int* n_time_masks, int* n_freq_masks);   /* as created */
void mww_prefetch_destroy(mww_prefetcher* p);
/* mww_prefetch_acquire + mww_set_targets + mww_assemble_batch + mww_prefetch_release in one call; out_labels /
 * out_weights ([B], may be NULL) receive copies of what went to the device. */
int mww_assemble_prefetched(mww_ctx* ctx, mww_prefetcher* p, float* out_labels, float* out_weights);

#ifdef __cplusplus
}
#endif
#endif /* MWW_H */

/*
 * mshgnn.h -- C-ABI of the MI355X-native MS-HGNN message-passing engine (libmshgnn.so).
 *
 * The reference (lunarlab-gatech/MorphSym-HGNN) has no FFI layer: its hot path sits behind two Python
 * surfaces.  This header is the boundary a maintainer would bind instead of
 *   GRF_HGNN_C2.forward   src/ms_hgnn/lightning_py/hgnn_c2.py:133-182   (+ autograd backward of it)
 *   GRF_HGNN_K4.forward   src/ms_hgnn/lightning_py/hgnn_k4.py:146-196
 *   GRF_HGNN.forward      src/ms_hgnn/lightning_py/hgnn.py:57-62
 *   MSE loss of the Lightning wrapper  gnnLightning.py:633-639, 680-695
 * i.e. everything `torch_geometric.nn.{HeteroDictLinear, HeteroConv, GraphConv, Linear}` + ATen execute for
 * one minibatch of time-window graphs (hgnn_c2.py:3, :88, :100-113, :131).
 *
 * Conventions: plain pointers and sizes, no torch types; every function returns 0 on success or a negative
 * MSHGNN_E* code (never throws across the ABI); mshgnn_last_error() gives the text for the calling thread.
 * All device buffers are caller-allocated (e.g. by the torch caching allocator); no ownership transfer.
 * forward/backward are asynchronous on the caller's HIP stream (passed as void* = hipStream_t).  A plan is
 * immutable after creation, so forward/backward are re-entrant from autograd worker threads as long as each
 * call uses its own workspace.  A workspace carries the activation stash from a forward to its backward: use one
 * workspace per stream, with the device that owns it current (hipSetDevice) when the entry point is called -- two
 * forwards on one workspace overwrite each other's stash, whatever streams they run on.
 *
 * Engines: plan creation picks the LDS-resident kernels (hidden == 128, <= 20 nodes per window, in-degree 1 on 'mean' relations) where
 * they apply, else the generic-width engine (any hidden multiple of 128 up to 2048, any node count / in-degree; bf16 or split-bf16
 * arithmetic -- an MSHGNN_F32 request is served by the latter at the same tolerance).  mshgnn_info.kernel_sets bit 2 tells which.
 *
 * Data layout ("dense window-major"): every window graph of a minibatch has the same tiny topology, so the
 * PyG-batched [B*n_type, F] tensors the reference passes are already [B, n_type, F] contiguous; they are
 * consumed as-is.  dtype selects storage/operand precision of inputs and activations:
 *   MSHGNN_F32  : fp32 storage, exact-fp32 MFMA (v_mfma_f32_16x16x4_f32)           -- parity mode (1e-4 rel).  ONLY on the LDS-resident
 *                 kernels: a plan that falls to the generic-width engine (mshgnn_info.kernel_sets & 4: hidden != 128, > 20 nodes, mean
 *                 aggregation with in-degree > 1) has no fp32-MFMA variant and serves an MSHGNN_F32 request with the split-bf16 arithmetic of
 *                 MSHGNN_BF16X3 (inputs stay fp32; products x_hi w_hi + x_lo w_hi + x_hi w_lo with fp32 accumulation: measured <= 1.4e-5 of the
 *                 fp64 reference, tolerance 1e-4) -- NOT a bit-exact fp32 fma chain.
 *   MSHGNN_BF16 : bf16 storage + bf16 MFMA operands, fp32 accumulate, fp32 weights  -- throughput mode
 *   MSHGNN_BF16X3: fp32 inputs, hi/lo bf16 planes, three bf16 MFMA products per term -- parity mode at bf16-MFMA speed
 * Parameters and parameter gradients are always one flat fp32 buffer (offsets given in the descriptor).
 */
#ifndef MSHGNN_H
#define MSHGNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSHGNN_MAX_TYPES 4
#define MSHGNN_F32 0
#define MSHGNN_BF16 1
#define MSHGNN_BF16X3 2   /* split-bf16 parity plan: fp32 inputs; every activation stored as two bf16 planes (hi + lo, 16 mantissa
                             bits); products on the bf16 MFMA as hi*hi + hi*lo + lo*hi, fp32 accumulate -- 1e-4 relative like
                             MSHGNN_F32 at a multiple of its speed.  LDS-resident for topologies with <= 20 nodes whose two planes fit a
                             CU's LDS: 2 x nodes <= 40 blocks (every A1 / Solo graph; MiniCheetah-K4's 20 nodes exactly, its four
                             base_transform scratch blocks aliasing node blocks) -- anything larger runs on the generic-width engine */

#define MSHGNN_OK 0
#define MSHGNN_EINVAL (-1)      /* bad descriptor / argument */
#define MSHGNN_EUNSUPPORTED (-2) /* valid request the engine cannot run (e.g. hidden not a multiple of 128) */
#define MSHGNN_EHIP (-3)        /* HIP runtime error */
#define MSHGNN_ENOMEM (-4)

#define MSHGNN_FLAG_RESIDUAL 1u      /* X <- f(H) + X            (hgnn_c2.py:161-166)                 */
#define MSHGNN_FLAG_BASE_MLP 2u      /* base_transform on mlp_type (hgnn_c2.py:117-121,156)           */

typedef struct mshgnn_plan mshgnn_plan;

/* One model instance over one robot topology.  Relation r is the reference's edge type
 * (src_type, rel, dst_type); its GraphConv has lin_rel (weight+bias) and lin_root (weight only).
 * All offsets are element offsets into the flat fp32 parameter buffer.                              */
typedef struct mshgnn_desc {
    int32_t n_types;                              /* node types, in data_metadata[0] order            */
    int32_t hidden;                               /* hidden_channels: a multiple of 128 (128: LDS-resident kernels; else generic-width engine) */
    int32_t num_layers;                           /* message-passing layers                           */
    int32_t n_rel;                                /* relations, in data_metadata[1] order             */
    int32_t out_type;                             /* node type the decoder reads ('foot')             */
    int32_t out_channels;                         /* out_channels_per_foot                            */
    int32_t mlp_type;                             /* type with the base_transform epilogue, or -1     */
    uint32_t flags;                               /* MSHGNN_FLAG_*                                    */
    int32_t dtype;                                /* MSHGNN_F32 | MSHGNN_BF16 | MSHGNN_BF16X3          */
    int32_t type_nodes[MSHGNN_MAX_TYPES];         /* nodes of each type per window                    */
    int32_t type_width[MSHGNN_MAX_TYPES];         /* input feature width F_type                       */
    const int32_t* rel_src;                       /* [n_rel] source type                              */
    const int32_t* rel_dst;                       /* [n_rel] destination type                         */
    const int32_t* rel_mean;                      /* [n_rel] 1 = 'mean' aggregation, 0 = 'add'        */
    const int32_t* rel_edge_off;                  /* [n_rel+1] prefix offsets into edges              */
    const int32_t* edges;                         /* [2*E] (src_node, dst_node) per-window pairs      */
    const float* in_mask[MSHGNN_MAX_TYPES];       /* [n_t*F_t] +-1 symmetry mask (apply_symmetry), host */
    const float* out_mask;                        /* [n_out*out_channels] +-1 (ms_foot_decoder), host */
    /* flat-parameter offsets */
    const int64_t* off_enc_w;                     /* [n_types]  encoder.lins.<t>.weight [h, F_t]      */
    const int64_t* off_enc_b;                     /* [n_types]                                        */
    const int64_t* off_rel_w;                     /* [L*n_rel]  convs.l.convs.<r>.lin_rel.weight      */
    const int64_t* off_rel_b;                     /* [L*n_rel]                 ...lin_rel.bias        */
    const int64_t* off_root_w;                    /* [L*n_rel]                 ...lin_root.weight     */
    int64_t off_mlp[4];                           /* base_transform.0.{w,b}, base_transform.2.{w,b}   */
    int64_t off_dec_w, off_dec_b;                 /* decoder.{weight,bias}                            */
    int64_t n_flat;                               /* length of the flat parameter buffer              */
} mshgnn_desc;

/* Work / traffic the plan executes per window (dead nodes of the last layers are skipped).           */
typedef struct mshgnn_info {
    int32_t rows_per_tile;        /* windows per workgroup tile of the layer kernels                   */
    int32_t total_nodes;          /* nodes per window                                                  */
    int64_t lds_bytes;            /* dynamic LDS of the layer kernels                                  */
    double flops_fwd;             /* algorithmic FLOPs / window, forward (2 FLOP per MAC)              */
    double flops_bwd;             /* algorithmic FLOPs / window, backward (dX chain + all dW)          */
    double flops_exec_fwd;        /* FLOPs / window the kernels actually issue (per-edge MACs, padding) */
    double flops_exec_bwd;
    double bytes_in;              /* input bytes / window at the plan dtype                            */
    int32_t n_gradw_workgroups;   /* split-K workgroups of the weight-gradient kernel                  */
    int32_t n_launches_fwd, n_launches_bwd;
    int32_t kernel_sets;          /* bit 0: fused stack kernels (bf16 plan), bit 1: slab stack kernels (two 4-wave workgroups per
                                     CU; used for batches of >= 1.5 tiles per CU), bit 2: generic-width engine -- what the plan allows  */
    int64_t grad_split;           /* two-phase step (mshgnn_step_mse_phase): gradients [grad_split, n_flat) are final after
                                     phase 0, [0, grad_split) (the encoder's) after phase 1; -1: the plan has no split       */
    double bytes_in_live;         /* input bytes / window of the nodes whose inputs can reach the output at this depth (<= bytes_in): what
                                     the plan reads.  A1-C2 at L = 3: the base nodes are four hops from the feet, their 3 600 B never matter */
} mshgnn_info;

/* Offsets (bytes) of the per-layer buffers inside the workspace, for tests / debugging.              */
typedef struct mshgnn_ws_layout {
    size_t total;
    size_t x[17];       /* X_l      [B][NN][h]  l = 0..L   (dtype)      hidden state after the encoder / layer l-1 */
    size_t dx[17];      /* dX_l     [B][NN][h]  l = 0..L   (dtype)                                                  */
    size_t dh[16];      /* dH_l     [B][NN][h]  l = 0..L-1 (dtype)      gradient w.r.t. the HeteroConv output       */
    size_t dd[16];      /* dd[0]: relu bytes of the encoder activation X_0 (layout as mask[l], training only); generic engine: dd[1] / dd[2] =
                           forward / backward sums of more than 32 rows into one node, L buffers of [aggregates][B][h] each (0: none); rest 0 */
    size_t mask[16];    /* relu bits, one byte per (node, window, 8 features): [NN][4][ceil(B/16)][4][16] uint8          */
    size_t hb[16], t1[16], du[16];  /* base_transform stash [B][n_mlp][h] (dtype)                                   */
    size_t wpack;       /* packed weights                                                                          */
    size_t bias;        /* packed biases (fp32)                                                                    */
    size_t dec_slabs;   /* decoder partial gradients (fp32)                                                                         */
    size_t slabs;       /* split-K partial weight gradients (fp32)                                                 */
    size_t loss;        /* 16 floats                                                                               */
} mshgnn_ws_layout;

/* Per-kernel timing (HIP events on the caller's stream) + the algorithmic work of each kernel of a step.   */
#define MSHGNN_BOUND_HBM 0
#define MSHGNN_BOUND_MFMA 1
typedef struct mshgnn_kernel_stat {
    char name[32];
    int32_t launches;             /* launches accumulated since the last read                          */
    int32_t bound;                /* MSHGNN_BOUND_* : the roofline that bounds this kernel             */
    float total_ms;               /* sum of event-measured durations                                   */
    float _pad;
    double flops_per_window;      /* algorithmic FLOPs / window / launch                                */
    double flops_exec_per_window; /* FLOPs the kernel issues / window / launch                          */
    double bytes_per_window;      /* algorithmic HBM bytes / window / launch                            */
} mshgnn_kernel_stat;

const char* mshgnn_last_error(void);
const char* mshgnn_version(void);
/* ABI guard for callers that keep their own copy of this header: MSHGNN_ABI_VERSION changes whenever a struct below changes size or meaning (5: mshgnn_info
 * gained bytes_in_live, kernel_sets bits 3 / 4 retired; 6: mshgnn_window_desc gained run_ptrs_ready), and the sizes the LIBRARY was built with can be compared with the caller's sizeof() before any
 * struct crosses the boundary -- mshgnn_plan_info / mshgnn_workspace_layout write sizeof(struct) bytes through the pointer they are given.                */
#define MSHGNN_ABI_VERSION 6
int mshgnn_abi_version(void);
size_t mshgnn_struct_size(int which);      /* 0: mshgnn_desc, 1: mshgnn_info, 2: mshgnn_ws_layout, 3: mshgnn_window_desc, 4: mshgnn_kernel_stat; else 0 */

int mshgnn_plan_create(const mshgnn_desc* desc, mshgnn_plan** plan_out);
void mshgnn_plan_destroy(mshgnn_plan* plan);
int mshgnn_plan_info(const mshgnn_plan* plan, mshgnn_info* info);
/* Name of the compile-time program ("A1C2_L3", ...) this plan's stack launches run on -- the one-call training steps, the forward alone (evaluation) and the two-call training
 * route; at every batch size (batches that are not whole 16-window tiles: predicated forms of the step and of the evaluation forward) -- or "" when the plan's tables are
 * interpreted at run time.  The kernels are specialised at build time for the topologies BASELINE.json names and the model types the reference's scripts default to
 * (csrc/mshgnn_spec_tables.inc, generated from this library's own plan compiler); a plan takes one only when its tables equal the kernel's, and MSHGNN_SPEC=0 at plan creation
 * keeps the interpreting kernels (same bits either way: tests/test_spec_gpu.py).  Nothing in the reference corresponds: its forward is interpreted by PyTorch
 * (hgnn_c2.py:150-166). */
const char* mshgnn_plan_specialised(const mshgnn_plan* plan);
/* Attach a program compiled for THIS plan after the library was built: `selector` is the `mshgnn_jit_program` symbol of a small shared library made from the library's own
 * kernel source over the plan's own tables (csrc/mshgnn.hip as -DMSHGNN_SPEC_SHARD=99; morphsym_hgnn_amd/jit.py renders the tables, compiles with hipcc and caches the result).
 * For topologies the build has no program for (any URDF through the topology compiler): their stack launches then run straight-line code like the BASELINE topologies' instead of
 * the interpreting kernels -- same MACs, same order, same bits.  LDS-resident bf16 and split-bf16 plans (the latter: `mshgnn_jit_program_x3` of csrc/mshgnn_x3.hip); the tables are compared with the plan's, a mismatch is MSHGNN_EINVAL and changes
 * nothing.  The caller keeps that shared library loaded for the plan's lifetime.  Nothing in the reference corresponds (its forward is interpreted by PyTorch). */
int mshgnn_plan_attach_program(mshgnn_plan* plan, void* selector);
/* The plan compiler alone, on the host: no HIP device is touched, nothing is allocated.  Fills `info` (work counts, LDS bytes, kernel sets: what
 * mshgnn_plan_info would report for a plan of this descriptor) and the number of 32-bit table entries the plan would upload; returns the same error
 * codes / mshgnn_last_error() text as mshgnn_plan_create for a descriptor no engine takes.  Used by the CPU test-suite and by callers that size a
 * deployment before a GPU is attached (replaces nothing in the reference: its models have no compile step, hgnn_c2.py:10-131).  */
int mshgnn_plan_compile_host(const mshgnn_desc* desc, mshgnn_info* info, int32_t* n_tables_out);

/* Profiling: while enabled, forward/backward bracket every kernel with HIP events on the stream (not
 * graph-capturable, not thread-safe).  mshgnn_profile_read synchronises the recorded events, fills up to
 * *n_inout stats (one per kernel of a step, in launch order), sets *n_inout and resets the accumulators.   */
int mshgnn_profile_enable(mshgnn_plan* plan, int on);
int mshgnn_profile_read(mshgnn_plan* plan, mshgnn_kernel_stat* stats, int32_t* n_inout);

/* Workspace for `batch` windows.  training=0: forward-only (no stash beyond what forward needs).    */
int mshgnn_workspace_layout(const mshgnn_plan* plan, int64_t batch, int training, mshgnn_ws_layout* out);

/* Forward.  x[t]: device pointer [batch][n_t][x_pitch[t]] at the plan dtype (x_pitch[t] >= F_t elements;
 * NULL pitch array = dense).  params: device fp32 flat buffer.  out: device fp32 [batch][n_out][out_channels]
 * (== the reference's [B*4, out] and, for C2 3-D GRF, its [B, 12] view; output mask applied).           */
int mshgnn_forward(const mshgnn_plan* plan, const void* const* x, const int64_t* x_pitch, const float* params,
                   float* out, void* workspace, int64_t batch, int training, void* stream);

/* Backward of the forward that last ran on `workspace` with the same x/params.  grad_out: device fp32
 * [batch][n_out][out_channels] (dL/d out).  grad_params: device fp32 flat buffer, fully OVERWRITTEN with
 * dL/d params (parameters that cannot influence the output get exact zeros).                            */
int mshgnn_backward(const mshgnn_plan* plan, const void* const* x, const int64_t* x_pitch, const float* params,
                    const float* grad_out, float* grad_params, void* workspace, int64_t batch, void* stream);

/* Backward with the wrapper's MSE fused in (one launch less, no grad_out round trip): `out` is the forward's output,
 * y the labels in the same [batch][n_out][out_channels] order; loss_out (device float[1]) receives mean((out-y)^2). */
int mshgnn_backward_mse(const mshgnn_plan* plan, const void* const* x, const int64_t* x_pitch, const float* params,
                        const float* out, const float* y, float* loss_out, float* grad_params, void* workspace,
                        int64_t batch, void* stream);

/* One whole training step of the regression wrappers: forward + MSE + backward, == mshgnn_forward(training=1) followed by
 * mshgnn_backward_mse, same results up to fp32 summation order.  On the bf16 plan the decoder, the loss and the decoder
 * backward all run inside the fused forward kernel (one launch less, no X_L / dX_L round trip); other plans run the
 * two-call sequence.  out receives the forward output, loss_out mean((out - y)^2), grad_params every gradient.
 * Batches of at least twice MSHGNN_STEP_CHUNK windows (read at plan creation, default 32768, 0 = never; mshgnn_step_ce alike; not the generic-width engine, not the
 * _src / _series forms) run as equal sub-steps of at least that many windows over contiguous window ranges on the front of the same workspace: same out, loss and gradient up to fp32
 * summation order (sub-step sums are added in order: still deterministic); afterwards the workspace holds the stashes of the LAST sub-step only.             */
int mshgnn_step_mse(const mshgnn_plan* plan, const void* const* x, const int64_t* x_pitch, const float* params, const float* y,
                    float* out, float* loss_out, float* grad_params, void* workspace, int64_t batch, void* stream);

/* One whole training_step of the classification wrappers (forward + CrossEntropyLoss over the per-foot logit pairs + loss.backward(),
 * gnnLightning.py:640-648, 680-722) in one call: on the bf16 plan the decoder, the cross entropy and the decoder backward run in the tail of the
 * fused forward kernel (== mshgnn_forward followed by mshgnn_backward_ce, which the other plans run).  labels: device int32 [batch][n_out] in {0, 1}. */
int mshgnn_step_ce(const mshgnn_plan* plan, const void* const* x, const int64_t* x_pitch, const float* params, const int32_t* labels,
                   float* out, float* loss_out, float* grad_params, void* workspace, int64_t batch, void* stream);

/* The same step in two calls, for overlapping the gradient all-reduce with compute on several GPUs: phase 0 runs the forward,
 * the loss, the backward sweep and every weight gradient except the encoder's -- grad_params[grad_split, n_flat) and
 * loss_out are final when it completes; phase 1 finishes the encoder's gradients, grad_params[0, grad_split).  A caller
 * launches the all-reduce of the first region between the two calls (bench.py, N > 1).  Results are bit-identical to
 * mshgnn_step_mse.  Plans without a split (mshgnn_info.grad_split < 0) return an error.                                   */
int mshgnn_step_mse_phase(const mshgnn_plan* plan, const void* const* x, const int64_t* x_pitch, const float* params,
                          const float* y, float* out, float* loss_out, float* grad_params, void* workspace, int64_t batch,
                          int phase, void* stream);

/* Backward with the classification wrapper's cross entropy fused in (gnnLightning.py:640-648: CrossEntropyLoss over the
 * batch*4 per-foot logit pairs, mean).  `out` = the forward's logits [batch][n_out][2], labels int32 [batch][n_out] in
 * {0,1}; loss_out (device float[1]) receives the mean cross entropy.  Only for plans with out_channels == 2.            */
int mshgnn_backward_ce(const mshgnn_plan* plan, const void* const* x, const int64_t* x_pitch, const float* params,
                       const float* out, const int32_t* labels, float* loss_out, float* grad_params, void* workspace,
                       int64_t batch, void* stream);

/* Adam on the flat fp32 buffers (configure_optimizers, gnnLightning.py:258-265; torch.optim.Adam defaults, no weight
 * decay / amsgrad).  step is 1-based; grads are multiplied by grad_scale first (1/world_size after a sum all-reduce).
 * All four buffers: device fp32, n elements, 16-byte aligned; updated in place.                                     */
int mshgnn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step,
                     float lr, float beta1, float beta2, float eps, float grad_scale, void* stream);
/* The same update with the step count kept on the DEVICE: *step_count = steps taken so far (int64, device memory); the launch uses t = *step_count + 1 for
 * the bias corrections and stores t behind the update.  Nothing that changes from step to step is a launch argument, so a training step (forward, loss,
 * backward, this) can be captured once in a HIP graph and replayed -- what torch.optim.Adam(capturable=True) does for the reference's optimizer
 * (gnnLightning.py:258-265).  morphsym_hgnn_amd.optim.FlatAdam(graph_safe=True), wrappers.GraphedTrainingStep.                                          */
int mshgnn_adam_step_counted(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t* step_count,
                             float lr, float beta1, float beta2, float eps, float grad_scale, void* stream);

/* Loss of the Lightning wrapper (gnnLightning.py:633-639): loss = mean((out - y)^2) over n elements and
 * grad_out = 2 (out - y) / n.  loss_out: device float[1].                                               */
int mshgnn_mse_loss(const float* out, const float* y, int64_t n, float* loss_out, float* grad_out, void* stream);

/* Loss of the classification wrappers (gnnLightning.py:640-648, customMetrics.py:6-25): mean cross entropy over `rows` per-foot logit
 * pairs (logits [rows][2], labels int32 [rows], nonzero = stable contact) and grad_out [rows][2] = (softmax - onehot) / rows
 * (NULL: loss only).  loss_out: device float[1].                                                                                  */
int mshgnn_ce_loss(const float* logits, const int32_t* labels, int64_t rows, float* loss_out, float* grad_out, void* stream);

/* ---- step metrics of the Lightning wrappers ------------------------------------------
 * The reference's torchmetrics states are plain sums across steps (gnnLightning.py:52-63, customMetrics.py:11-54); each
 * call ADDS one step's sums into caller-owned device state (zero it to start an epoch / to get step values).  Math in
 * fp64, a fixed reduction order (bit-reproducible).
 *
 * regression (calculate_losses_step, gnnLightning.py:124-130 / :633-639): state double[3]:
 *   [0] += sum (y_pred - y)^2   [1] += sum |y_pred - y|   [2] += n        => MSE = [0]/[2], RMSE = sqrt(MSE), L1 = [1]/[2] */
int mshgnn_metrics_regression(const float* y_pred, const float* y, int64_t n, double* state, void* stream);

/* classification (gnnLightning.py:132-151, 285-348): logits fp32 [batch*4][2] (per foot: no-contact, contact), labels int32
 * [batch][4] in {0,1}.  ce_state double[2]: [0] += sum of per-foot cross entropies, [1] += 4*batch.  counts int64[18]:
 *   [0] += batch, [1] += windows whose 16-class argmax (classification_conversion_16_class, :306-348) equals the label
 *   state 8 y0 + 4 y1 + 2 y2 + y3, [2 + 4k .. 5 + 4k] += tp, fp, fn, tn of leg k (BinaryF1Score, customMetrics.py:26-54). */
int mshgnn_metrics_classification(const float* logits, const int32_t* y, int64_t batch, double* ce_state, int64_t* counts,
                                  void* stream);

/* The same sums for one training / validation step in ONE multi-workgroup launch (what a wrapper's calculate_losses_step needs every step):
 * batch_* (nullable as a pair) receive this step's sums (overwritten -- no zeroing) followed by the step's published values, so that a
 * wrapper logs them without further launches: batch_state double[8] = sq, abs, n, MSE, RMSE, L1, 0, 0; batch_ce double[8] = ce sum, rows,
 * CE, 16-class accuracy, F1 of leg 0..3 (customMetrics.py:51-54, 0/0 -> 0); epoch_* (nullable as a pair; double[3] / double[2]) have the sums added, and grad_out (nullable) receives the gradient of the step's loss with respect to the predictions -- what
 * `training_step` returns for backward (gnnLightning.py:709-722): regression d MSE / d y_pred = 2 (y_pred - y) / n, fp32 [n];
 * classification d CE / d logits = (softmax - onehot) / (4 batch), fp32 [batch*4][2].  scratch: MSHGNN_METRICS_SCRATCH_BYTES of device
 * memory, 8-byte aligned, zeroed ONCE by the caller and then owned by these calls (per-workgroup partials + a ticket the kernel resets;
 * the partials are added in workgroup order: bit-reproducible); one scratch per stream.                                                  */
#define MSHGNN_METRICS_SCRATCH_BYTES 16384
int mshgnn_metrics_regression_step(const float* y_pred, const float* y, int64_t n, double* batch_state, double* epoch_state,
                                   float* grad_out, void* scratch, void* stream);
int mshgnn_metrics_classification_step(const float* logits, const int32_t* y, int64_t batch, double* batch_ce, int64_t* batch_counts,
                                       double* epoch_ce, int64_t* epoch_counts, float* grad_out, void* scratch, void* stream);

/* The extra sums of the centroidal-momentum wrappers (COM_Base_Lightning.calculate_losses_step, gnnLightning_com.py:96-121;
 * CosineSimilarityMetric, customMetrics.py:56-95): y_pred / y fp32 [batch][n_bases][6], each base node (lin(3) | ang(3)), standardised;
 * y_mean / y_std: HOST double[6] of the dataset's Standarizer (soloDataset.py:12-46).  state double[8] (batch_state overwritten,
 * epoch_state added into, either nullable):  [0] sum sq err of the lin halves  [1] of the ang halves  [2] = [3] 3 n_bases batch
 * [4] sum over windows of cos(lin_pred, lin) of base node 0 after un-standardising  [5] the same for ang  [6] batch  [7] 0.
 * MSE_lin = [0]/[2], MSE_ang = [1]/[3], cos_sim_lin = [4]/[6], cos_sim_ang = [5]/[6].  scratch: as for the two calls above.          */
int mshgnn_metrics_com_step(const float* y_pred, const float* y, int64_t batch, int n_bases, const double* y_mean, const double* y_std,
                            double* batch_state, double* epoch_state, void* scratch, void* stream);

/* body_frame_to_world_frame (gnnLightning.py:663-676) without the per-step CPU/scipy round trip: quat fp32 [batch][4] is
 * the world->body rotation, scalar-last (x, y, z, w) as scipy.Rotation.from_quat takes it; grf fp32 [batch][4][3].       */
int mshgnn_grf_body_to_world(const float* quat, const float* grf_body, float* grf_world, int64_t batch, void* stream);

/* ---- on-device window assembly ---------------------------------------------------------------
 * Replaces the per-window feature building of the dataset classes (quadSDKDataset_Morph.py:304-369, 444-488;
 * flexibleDataset.py:340-400, 563-596) + PyG collate: the sequence's raw series stay in HBM, fp32 COLUMN-major (element
 * (row, col) at src[col * src_cstride + row], so a window of one column is one contiguous stretch), and a batch of
 * windows [start, start + history) is gathered into the engine's inputs [batch][n_t][x_pitch[t]] (dtype).
 * runs (DEVICE int32[n_runs][5], sorted by (type, node)): {type, node, first feature, source << 8 | column (or -1: constant
 * 1), length <= 256}: features first .. first + length - 1 of that node row = src(start + k, column), k = 0 .. length - 1
 * (standardised over the window when normalize, as flexibleDataset.py:390-396).  rows (DEVICE int32[n_rows][2]): the run
 * range [begin, end) of every node row (one wave assembles one node row of one window).  Labels: y[b][k] = src[label_src][start + history - 1][label_cols[k]]
 * (label_cols: DEVICE int32[n_label]); label_rotate: 3-D world-frame GRFs -> body frame with the quaternion
 * src[quat_src] of that row (quadSDKDataset.py, load_data_at_dataset_seq_3d); quat_out (optional) receives it.         */
typedef struct mshgnn_window_desc {
    int32_t n_types, dtype, history, normalize;
    int32_t type_nodes[MSHGNN_MAX_TYPES], type_width[MSHGNN_MAX_TYPES];
    int32_t n_src, n_runs, n_rows;
    int32_t fast_layout;      /* != 0: the caller states that the runs of every node row all have one length and follow each other from the row's first run's
                               * feature on (what a recipe of T-long variables gives): with 16-byte aligned output rows and no standardisation the
                               * gather then runs as one 16-byte store per output chunk (k_assemble_windows_fast)                              */
    const int32_t* runs;
    const int32_t* rows;
    int32_t n_label, label_src, label_rotate, quat_src;      /* quat_src = -1: none */
    const int32_t* label_cols;
    int32_t run_ptrs_ready;   /* mshgnn_step_*_series: != 0 = the caller states that the run_ptrs scratch still holds what an earlier call with THIS descriptor and
                               * THESE source arrays wrote there (the runs' column pointers depend on nothing else): the one-workgroup launch that resolves them
                               * in front of the encoder (4.8 us) is skipped.  0: resolve them (always safe).                                       */
    int32_t reserved_;
} mshgnn_window_desc;

int mshgnn_assemble_windows(const mshgnn_window_desc* desc, const float* const* src, const int64_t* src_cstride,
                            const int64_t* src_rows, const int64_t* starts /* device int64[batch] */, int64_t batch,
                            void* const* x_out, const int64_t* x_pitch, float* y_out, float* quat_out, void* stream);

/* One training step straight from a sequence's resident raw series: mshgnn_assemble_windows + mshgnn_step_mse with the window gather FUSED INTO THE
 * ENCODER (no separate pass that writes and re-reads the batch's windows): the encoder kernel gathers its K chunks from bf16 copies of the series
 * (src_bf16: same shapes / column strides as src, every column followed by >= 8 elements of slack: src_cstride >= src_rows + 8), and writes the
 * materialised windows x_out (bf16, 16-byte aligned rows, pitch a multiple of 8) for the weight-gradient pass and for the caller; labels / quaternions come
 * from the fp32 series as in mshgnn_assemble_windows.  The descriptor must be a bf16, fast_layout, unstandardised recipe whose node types match the
 * plan's; bf16 plan with the fused stack kernels (else MSHGNN_EUNSUPPORTED: assemble, then mshgnn_step_mse).  run_ptrs: device scratch, 8 bytes per run.
 * x_out == NULL (x_pitch ignored): NO materialised windows -- the weight-gradient kernel gathers its raw-input operands from the series as well.
 * Results are bit-identical to mshgnn_assemble_windows followed by mshgnn_step_mse either way.
 * Split plan (MSHGNN_BF16X3): the encoder gathers from the fp32 series themselves (src; src_bf16 may be NULL; the same 8 elements of slack behind
 * every column), the descriptor's dtype is MSHGNN_F32 / MSHGNN_BF16X3, x_out (fp32, 16-byte aligned rows, pitch a multiple of 4) is required.     */
int mshgnn_step_mse_series(const mshgnn_plan* plan, const mshgnn_window_desc* desc, const float* const* src, const void* const* src_bf16,
                           const int64_t* src_cstride, const int64_t* src_rows, const int64_t* starts /* device int64[batch] */, int64_t batch,
                           void* const* x_out, const int64_t* x_pitch, float* y_out, float* quat_out, void* run_ptrs,
                           const float* params, float* out, float* loss_out, float* grad_params, void* workspace, void* stream);

/* The same for the classification wrappers (mshgnn_step_ce): labels_out (device int32 [batch][n_out]) receives the window labels != 0 (the
 * contact flags of the window's last step), y_out the float label rows.                                                                         */
int mshgnn_step_ce_series(const mshgnn_plan* plan, const mshgnn_window_desc* desc, const float* const* src, const void* const* src_bf16,
                          const int64_t* src_cstride, const int64_t* src_rows, const int64_t* starts, int64_t batch,
                          void* const* x_out, const int64_t* x_pitch, float* y_out, int32_t* labels_out, void* run_ptrs,
                          const float* params, float* out, float* loss_out, float* grad_params, void* workspace, void* stream);

/* ---- stand-alone operators behind the four torch_geometric.nn names ----------------------------------
 * For a maintainer who swaps only the PyG import (hgnn_c2.py:3): Linear / HeteroDictLinear (hgnn_c2.py:88,131), GraphConv
 * (hgnn_c2.py:100-112) and their autograd backward on arbitrary graphs and widths, fp32 operands, fp32 MFMA, no float atomics.
 *
 * mshgnn_op_gemm: C[m ldc + n] (+)= sum_k A[m sAm + k sAk] B[n sBn + k sBk] (+ bias[n]) -- element strides, so the three products
 * of a Linear are one entry point: y = x W^T + b (A = x, B = W), dx = dy W (B = W with sBn = 1, sBk = in), dW = dy^T x (A = dy with
 * sAm = 1, sAk = out; B = x with sBn = 1, sBk = in: reduction over the rows, split-K with a fixed-order sum).
 * mshgnn_op_gemm_workspace: bytes of device workspace that shape needs (0: none).  accumulate != 0 adds into C.               */
int64_t mshgnn_op_gemm_workspace(int64_t M, int64_t N, int64_t K, int32_t* splits_out);
int mshgnn_op_gemm(const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBn, int64_t sBk, const float* bias,
                   float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int accumulate, void* workspace, void* stream);

/* GraphConv's aggregation (PyG: propagate with aggr 'add' / 'mean', before lin_rel): out[r][:] = sum over the CSR row r of
 * edge_scale[e] x[col[e]][:] (edge_scale null: 1; 'mean': 1 / max(in-degree, 1) of the edge's destination).  The backward is the
 * same call on the CSR of the transposed graph.  rowptr: int32[n_rows + 1], col / edge_scale: [n_edges], all on the device.   */
int mshgnn_op_aggregate(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col, const float* edge_scale,
                        float* out, int64_t ldo, int64_t n_rows, int64_t width, void* stream);

/* bias gradient: out[n] = sum_m X[m ldx + n], two fixed-order stages; workspace bytes from mshgnn_op_colsum_workspace.      */
int64_t mshgnn_op_colsum_workspace(int64_t M, int64_t N);
int mshgnn_op_colsum(const float* X, int64_t ldx, float* out, int64_t M, int64_t N, void* workspace, void* stream);

/* ---- the reference's own input tensors, uncast ---------------------------------------------------------------------------------------------------
 * The reference's datasets hand the model fp64 tensors at the dense pitch (torch.set_default_dtype(float64), gnnLightning.py:1183; x_dict[type] is
 * [B n_t, F_t]).  These three entry points are mshgnn_forward / mshgnn_step_mse / mshgnn_step_ce for such tensors: src[t] = device pointer to the caller's
 * rows, src_bytes = 8 (fp64) or 4 (fp32), src_pitch[t] in elements (NULL = dense).  The encoder converts in registers (fp64 -> fp32 -> bf16, round to nearest
 * even at each step: exactly the values torch's .to() produces, two roundings included) and WRITES the plan-dtype rows into x_rows[t] ([batch][n_t][x_pitch[t]], 16-byte aligned, pitch a
 * whole number of 16-byte chunks: what mshgnn_backward* and the weight-gradient pass of the step read) for the nodes the plan reads -- instead of a separate
 * cast + re-pitch pass over the batch (A1-C2, 8192 windows: 472 MB read + 118 MB written + 118 MB re-read).  bf16 and split-bf16 plans of the LDS-resident
 * kernels; MSHGNN_EUNSUPPORTED on the fp32 plan and the generic-width engine (cast there).  fp32 rows of an even width must start 8-byte aligned.      */
int mshgnn_forward_src(const mshgnn_plan* plan, int src_bytes, const void* const* src, const int64_t* src_pitch, void* const* x_rows,
                       const int64_t* x_pitch, const float* params, float* out, void* workspace, int64_t batch, int training, void* stream);
int mshgnn_step_mse_src(const mshgnn_plan* plan, int src_bytes, const void* const* src, const int64_t* src_pitch, void* const* x_rows,
                        const int64_t* x_pitch, const float* params, const float* y, float* out, float* loss_out, float* grad_params,
                        void* workspace, int64_t batch, void* stream);
int mshgnn_step_ce_src(const mshgnn_plan* plan, int src_bytes, const void* const* src, const int64_t* src_pitch, void* const* x_rows,
                       const int64_t* x_pitch, const float* params, const int32_t* labels, float* out, float* loss_out, float* grad_params,
                       void* workspace, int64_t batch, void* stream);

/* ---- data-parallel gradient exchange on the caller's stream (replaces the bucketed all-reduce of Lightning-DDP, ---------------------
 * gnnLightning.py:1396-1400) -- one process per GPU, replicated weights, windows sharded: the ONLY collective of a step is the mean of the flat fp32
 * gradient over the ranks.  RCCL is bound at run time (dlopen of `rccl_path`, else librccl.so as the process already has it / as the loader finds it):
 * no link-time dependency, single-GPU callers never load it.  mshgnn_comm_unique_id on rank 0 -> the 128 bytes travel to every rank by any means
 * (torch.distributed's store, MPI, a file) -> mshgnn_comm_create on every rank with its device current (collective: ncclCommInitRank).
 * mshgnn_comm_allreduce_mean enqueues ncclAllReduce(buf, buf, n, float32, avg) on `stream` and returns: stream-ordered behind the step that produced
 * buf, no host synchronisation, capturable in a HIP graph together with the step.  _sum: the same with ncclSum (callers that weight by window counts). */
typedef struct mshgnn_comm mshgnn_comm;
int mshgnn_comm_unique_id(const char* rccl_path, void* id_out_128_bytes);
int mshgnn_comm_create(const char* rccl_path, const void* id_128_bytes, int nranks, int rank, mshgnn_comm** comm_out);
void mshgnn_comm_destroy(mshgnn_comm* comm);
int mshgnn_comm_allreduce_mean(mshgnn_comm* comm, float* buf, int64_t n, void* stream);
int mshgnn_comm_allreduce_sum(mshgnn_comm* comm, float* buf, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MSHGNN_H */

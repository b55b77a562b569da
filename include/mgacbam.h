/*
 * mgacbam.h -- C ABI of the MI355X (gfx950) mask-guided CBAM library (libmgacbam.so).
 *
 * This is the drop-in boundary for ONE hot path of MGA-YOLO: the MaskCBAM block, forward and backward
 * (reference: mga_yolo/nn/modules/masked_cbam.py:10-174).  Nothing like this ABI exists in the
 * reference -- there the block is ~65 eager ATen ops forward / ~90 backward; the Python mirror of the
 * reference module (mga_yolo_amd/module.py) binds these entry points with ctypes and keeps the
 * reference's nn.Module contract (constructor, [feat, mask] list input, state_dict keys, .alpha).
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / C++ types.  `stream` is a hipStream_t passed as void*.
 *   - the caller owns every buffer (features, gradients, ctx, scratch); the library never allocates,
 *     frees, copies to the host or synchronises, so every call may be captured in a hipGraph.
 *   - re-entrant: no mutable globals except a thread-local error string.
 *   - return value: 0 = ok, <0 = argument error (MGACBAM_E_*), >0 = hipError_t from a launch.
 *   - tensors are dense NCHW; `dtype` selects the element type of x / y / gy / gx (mask, parameters,
 *     every accumulator and every saved statistic are fp32).
 *
 * Entry-point families: mgacbam_*  MaskCBAM (the hot path) + mgacbam_eca_* MaskECA + mgacbam_resize_nearest;
 *                       mgahead_*  MGAMaskHead (the mask producer);  mgaseg_* multi-scale segmentation loss + mgakendall_* combine;
 *                       mgapmg_*   ProbMaskGater's Gumbel gate.
 */
#ifndef MGACBAM_H_
#define MGACBAM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGACBAM_ABI_VERSION 14
#define MGACBAM_MAX_LEVELS 8          /* P3/P4/P5 need 3 */

enum { MGACBAM_F32 = 0, MGACBAM_F16 = 1, MGACBAM_BF16 = 2 };

/* Level flags.  The forward can save P[b,j,hw] = sum_c W1[j,c] x[b,c,hw] (hidden <= MGACBAM_PROJ_MAX_HIDDEN planes per sample)
 * while it streams x anyway; the backward then forms the masked-average part of dL/dmask, sum_c g_avg[b,c] x[b,c,hw] =
 * sum_j g_h[b,j] P[b,j,hw], from those planes and its largest kernel does not read x at all (3 -> 2 feature-sized streams).
 * Set MGACBAM_FWD_SAVE_PROJ in the forward of a step whose backward will ask for gmask, and MGACBAM_BWD_HAVE_PROJ in that
 * backward.  Levels with a larger hidden size ignore the flags and read x. */
#define MGACBAM_PROJ_MAX_HIDDEN 4
enum { MGACBAM_FWD_SAVE_PROJ = 1 };
enum { MGACBAM_BWD_HAVE_PROJ = 1 };

enum {
  MGACBAM_E_NULL = -1,        /* required pointer is NULL */
  MGACBAM_E_SHAPE = -2,       /* B,C,H,W,hidden,k out of range */
  MGACBAM_E_DTYPE = -3,
  MGACBAM_E_ALIGN = -4,       /* ctx/scratch not 16-byte aligned */
  MGACBAM_E_LEVELS = -5,      /* n_levels outside 1..MGACBAM_MAX_LEVELS */
  MGACBAM_E_SIZE = -6         /* a caller-owned work buffer (ctx / scratch / ws) is smaller than the matching *_bytes() query answers for
                                 this shape under the library's CURRENT knobs; last_error names the buffer, want and got.  Nothing is
                                 launched.  (ABI 14: every work buffer travels with its capacity -- a binding that caches a size across
                                 mgacbam_reload_env(), or passes a buffer sized for another shape, gets this code instead of
                                 out-of-bounds device writes) */
};

/* Constructor arguments + learnable state of one block (reference masked_cbam.py:34-64).
 * Parameter tensors are the reference's state_dict entries, fp32, contiguous, on the device:
 *   w1  = cam_mlp.0.weight (hidden, C)     b1 = cam_mlp.0.bias (hidden)
 *   w2  = cam_mlp.2.weight (C, hidden)     b2 = cam_mlp.2.bias (C)
 *   wsa = sam_conv.weight  (1, 3, k, k)    beta = beta ()        -- alpha = softplus(beta) is formed on
 * the device so that no host read of a parameter (a sync) is ever needed. */
typedef struct mgacbam_params {
  const float* w1;
  const float* b1;
  const float* w2;
  const float* b2;
  const float* wsa;
  const float* beta;
  int32_t hidden;            /* max(1, C / r) */
  int32_t k;                 /* odd spatial kernel size, 1..15 */
  int32_t use_sigmoid_mask;  /* masked_cbam.py:39 */
  float tiny_thr;            /* masked_cbam.py:40 */
  float eps;                 /* masked_cbam.py:41 */
} mgacbam_params_t;

/* One pyramid level of a forward call: replaces MaskCBAM.forward([feat, mask]) (masked_cbam.py:154-171). */
typedef struct mgacbam_fwd_level {
  const void* x;             /* (B,C,H,W) feature                              */
  const float* mask;         /* (B,1,H,W) fp32 logits, or NULL = vanilla CBAM  */
  void* y;                   /* (B,C,H,W) output, same dtype as x              */
  void* ctx;                 /* mgacbam_ctx_bytes(): statistics saved for backward (or inspection) */
  size_t ctx_bytes;          /* capacity of the buffer at ctx, as the caller allocated it (checked: MGACBAM_E_SIZE) */
  mgacbam_params_t p;
  int32_t B, C, H, W;
  int32_t dtype;
  int32_t flags;             /* MGACBAM_FWD_* */
} mgacbam_fwd_level_t;

/* One pyramid level of a backward call: replaces what autograd derives for the block. */
typedef struct mgacbam_bwd_level {
  const void* x;             /* as in forward                                   */
  const float* mask;         /* as in forward (NULL if it was NULL)             */
  const void* gy;            /* (B,C,H,W) dL/dy                                 */
  const void* ctx;           /* written by the matching forward                 */
  void* scratch;             /* mgacbam_bwd_scratch_bytes(), contents undefined */
  size_t ctx_bytes;          /* capacities of the buffers at ctx / scratch as allocated (checked: MGACBAM_E_SIZE) */
  size_t scratch_bytes;
  void* gx;                  /* (B,C,H,W) dL/dx                                 */
  float* gmask;              /* (B,1,H,W) dL/dmask, or NULL if not wanted       */
  float* gw1;                /* parameter gradients, same shapes as parameters; OVERWRITTEN (not accumulated) */
  float* gb1;
  float* gw2;
  float* gb2;
  float* gwsa;
  float* gbeta;
  mgacbam_params_t p;
  int32_t B, C, H, W;
  int32_t dtype;
  int32_t flags;             /* MGACBAM_BWD_HAVE_PROJ */
} mgacbam_bwd_level_t;

/* Named regions inside ctx (byte offsets), for stage-wise tests and tooling.  All fp32 unless noted. */
typedef struct mgacbam_ctx_layout {
  int64_t S;        /* (B)      sum_hw sigma(mask)                        masked_cbam.py:99  */
  int64_t use;      /* (B)      1.0 if mean sigma(mask) >= tiny_thr       masked_cbam.py:98  */
  int64_t den;      /* (B)      max(S, eps)                               masked_cbam.py:99  */
  int64_t avg;      /* (B,C)    pooled descriptor fed to the MLP          masked_cbam.py:102 */
  int64_t mx;       /* (B,C)    pooled descriptor fed to the MLP          masked_cbam.py:121 */
  int64_t mavg;     /* (B,C)    masked average before the GAP blend       masked_cbam.py:100 */
  int64_t valid;    /* (B,C)    int32: 1 = masked max used, 0 = GAP fallback   masked_cbam.py:120 */
  int64_t amax;     /* (B,C)    int32: first arg-max position in H*W           masked_cbam.py:117 */
  int64_t h_avg;    /* (B,hid)  relu(W1 avg + b1)                          masked_cbam.py:54-56 */
  int64_t h_mx;     /* (B,hid)  relu(W1 mx + b1)                                              */
  int64_t ca;       /* (B,C)    channel gate                              masked_cbam.py:129 */
  int64_t planes;   /* (B,3,HW) [max_c u, mean_c u, sigma(mask)]           masked_cbam.py:146 */
  int64_t cidx;     /* (B,HW)   int32: first arg-max channel of u          masked_cbam.py:135 */
  int64_t sa;       /* (B,HW)   spatial gate                              masked_cbam.py:147 */
  int64_t proj;     /* (B,hid,HW) W1-projection of x, only when hid <= MGACBAM_PROJ_MAX_HIDDEN (else empty)  */
  int64_t sync;     /* int32 hand-off state, generation counters that are never reset: (B, ceil(HW/16)+1) tile flags of MGACBAM_FWD_FUSE
                       (see there), 4 status words ([0] time-out, [1] [2] arrival counters of the dWsa tail roles: 0 between calls), (B) ca
                       flags, 2 x (B, ceil(HW/16)+1) MGACBAM_BWD_FOLD tile / conv-tile flags; the merged backward launch's own 3 x (B, ceil(HW/16)+1)
                       tile / conv-tile / dWsa-tile flags and (B, C) sweep flags */
  int64_t total;    /* == mgacbam_ctx_bytes()                                                 */
  int64_t status;   /* int32 status word inside `sync`: 0 = every in-launch hand-off of every call on this ctx completed; non-zero =
                       one timed out (that tile's outputs were poisoned with NaN).  The caller reads these 4 bytes wherever it
                       synchronises anyway (mga_yolo_amd: PyramidPlan.check_handoff(), and per call with MGACBAM_CHECK_HANDOFF=1) */
} mgacbam_ctx_layout_t;

int mgacbam_abi_version(void);
/* Tuning / test knobs (MGACBAM_* environment variables) are read once, at the first call; this re-reads them.  Not to be called
 * concurrently with other entry points. */
void mgacbam_reload_env(void);
const char* mgacbam_last_error(void);      /* thread-local, valid until the next call on this thread */
const char* mgacbam_build_info(void);      /* "gfx950 hipcc-x.y ..." */

size_t mgacbam_ctx_bytes(int B, int C, int H, int W, int hidden);
size_t mgacbam_bwd_scratch_bytes(int B, int C, int H, int W, int hidden, int k);
int mgacbam_ctx_layout(int B, int C, int H, int W, int hidden, mgacbam_ctx_layout_t* out);

/* Forward / backward over n_levels independent pyramid levels (P3/P4/P5 = 3) enqueued on `stream`. */
int mgacbam_forward(const mgacbam_fwd_level_t* levels, int n_levels, void* stream);
int mgacbam_backward(const mgacbam_bwd_level_t* levels, int n_levels, void* stream);

/* The same work split into its kernels ("stages"), in dependency order.  mgacbam_forward == all forward stages,
 * mgacbam_backward == all backward stages.  Uses:
 *   - data-parallel training: enqueue MGACBAM_BWD_PARAMS (everything the parameter gradients need), start the
 *     gradient all-reduce on another stream, then enqueue MGACBAM_BWD_INPUTS (gx, gmask) so the exchange overlaps it;
 *   - measurement: time one kernel alone with events on `stream` (bench.py roofline).
 * A later stage reads what earlier stages left in ctx / scratch, so both buffers must be kept between calls. */
enum {
  MGACBAM_FWD_POOL = 1,      /* k_pool : masked avg/max pooling over H*W                      reads x           */
  MGACBAM_FWD_CHAN = 2,      /* k_chan : shared MLP + channel gate (prologue), channel max/mean planes  reads x */
  MGACBAM_FWD_APPLY = 4,     /* k_apply: k x k conv + spatial gate (prologue), y = x + alpha (x ca sa - x)      */
  MGACBAM_FWD_ALL = 7,
  MGACBAM_FWD_FUSE = 8       /* CHAN + APPLY as ONE launch (k_gate) that reads x once instead of twice: every workgroup keeps
                                its tile of x (all channels) in registers and hands its plane rows to the neighbouring
                                tiles through generation flags; one role workgroup per sample runs the shared MLP.
                                CONTRACT: the caller zero-fills ctx[sync .. total) once after allocating ctx (and again
                                after a call that was aborted mid-flight); the library keeps that state consistent.
                                Eligibility is decided per launch group from the shape (C <= 4096, tiles >= 16 px and >= one
                                image row, halo rows fit in LDS) AND from the device: a tile waits for tiles up to 8*span
                                workgroup ids ahead, so 2*(8*span+1) workgroups of the chosen kernel must be co-resident
                                (multiProcessorCount x occupancy; MGACBAM_RESIDENT_WGS overrides) -- otherwise the group
                                runs as three launches as without the flag.  Waits are bounded: a hand-off that times out
                                (the device is shared with other work that takes the CUs away) sets the status word
                                (mgacbam_ctx_layout_t.status), POISONS that tile's sa / y with NaN and the launch still drains:
                                a failure is loud in the loss, never a silently stale halo                              */
};
enum {
  MGACBAM_BWD_REDUCE1 = 1,    /* k_bwd_reduce1: sums of gy*x over H*W and over C                reads x, gy       */
  MGACBAM_BWD_CONVT = 2,      /* k_bwd_convT  : transposed conv -> dL/dplanes                   tiny              */
  MGACBAM_BWD_REDUCE2 = 4,    /* k_bwd_reduce2: rest of dL/dca, dL/dz, hidden-gradient partials reads x           */
  MGACBAM_BWD_WSA = 8,        /* k_bwd_wsa    : dWsa tile partials (needs REDUCE1 only)         tiny              */
  MGACBAM_BWD_PARAMGRAD = 16, /* k_bwd_params : dW1 db1 dW2 db2 dWsa dbeta (needs REDUCE2, WSA) tiny              */
  MGACBAM_BWD_APPLY = 32,     /* k_bwd_apply  : gx (+ gmask) (needs REDUCE2, CONVT)              reads gy (+x), writes gx */
  MGACBAM_BWD_FUSE = 64,      /* run WSA as role workgroups inside the REDUCE2 launch and PARAMGRAD inside the APPLY
                                 launch (when both stages of a pair are requested): the tiny latency-bound kernels then
                                 overlap the HBM-bound ones instead of standing on the critical path                    */
  MGACBAM_BWD_PARAMS = 31,    /* every stage the parameter gradients depend on, as separate launches: a data-parallel
                                 caller starts its all-reduce after this and overlaps it with MGACBAM_BWD_INPUTS     */
  MGACBAM_BWD_INPUTS = 32,    /* the stage only the input gradients depend on                   */
  MGACBAM_BWD_ALL = 127,
  MGACBAM_BWD_FOLD = 128      /* with REDUCE1 + CONVT: the transposed-conv tiles run as the LAST workgroups of the REDUCE1 launch and
                                 pick the g_pre rows up inside the launch (one generation counter per REDUCE1 tile and per conv
                                 tile in ctx.sync; contract as MGACBAM_FWD_FUSE: the caller zero-filled ctx[sync .. total) once).
                                 The counters are never reset, so a backward that stops after this launch leaves a consistent
                                 state.  The conv tiles wait only for lower-numbered workgroups that never wait themselves, so
                                 progress does not depend on residency; a time-out still poisons (NaN g_planes -> gx) and sets
                                 the status word.  Not part of MGACBAM_BWD_ALL: mgacbam_backward() works on an un-zeroed ctx.
                                 With REDUCE2 + WSA + FUSE in the same call as well (mgacbam_backward_stages(MGACBAM_BWD_ALL | MGACBAM_BWD_FOLD):
                                 the whole backward), spatial kernel 7: REDUCE1, CONVT, WSA and REDUCE2 are ONE launch (k_bwd_r12) -- the
                                 sweeps of a sample wait for its conv tiles inside the launch; generation counters of its own in ctx.sync,
                                 so both launch forms may alternate on one ctx.  The backward is then 2 launches (4 per step).
                                 Knob MGACBAM_WSA_TAIL=1 (opt-in, measured slower at BASELINE configs[1]): with WSA + PARAMGRAD + APPLY +
                                 FUSE in the same call the flag also moves the dWsa tile partials from the front of the REDUCE2 launch
                                 to the END of the APPLY launch, followed by the workgroups that sum them in the fixed order (bitwise
                                 reproducible) once an arrival counter (status words 1, 2 of ctx.sync, 0 between calls) says so      */
};
int mgacbam_forward_stages(const mgacbam_fwd_level_t* levels, int n_levels, int stages, void* stream);
int mgacbam_backward_stages(const mgacbam_bwd_level_t* levels, int n_levels, int stages, void* stream);

/* Nearest-neighbour resize of (n_planes, in_h, in_w) fp32 planes to (n_planes, out_h, out_w): the integer
 * index path src = min(floor(dst * in/out), in-1) of mga_yolo/nn/losses/segmentation.py:103-110
 * (F.interpolate(mode="nearest")).  Bit-exact with the reference by construction (pure gather). */
int mgacbam_resize_nearest(const float* src, float* dst, int n_planes, int in_h, int in_w, int out_h, int out_w,
                           void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * MaskECA -- the second mask-guided attention block of the reference (mga_yolo/nn/modules/masked_eca.py:68-196), behind the
 * same boundary: masked average pooling (GAP fallback for tiny masks) -> k-tap conv1d over the channel axis -> sigmoid ->
 * y = x * (1 + softplus(beta) * (w - 0.5)).  Same conventions as above (caller-owned buffers, explicit stream, no sync).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct mgacbam_eca_params {
  const float* w;            /* conv1d.weight (1,1,k) fp32                  masked_eca.py:124 */
  const float* beta;         /* beta ()                                     masked_eca.py:127 */
  int32_t k;                 /* odd, 1..15 (eca_kernel_size, :44-54)        */
  int32_t use_sigmoid_mask;  /* :63  */
  float tiny_thr;            /* :64  */
  float eps;                 /* :65  */
} mgacbam_eca_params_t;

typedef struct mgacbam_eca_fwd_level {
  const void* x;             /* (B,C,H,W) */
  const float* mask;         /* (B,1,H,W) fp32 or NULL */
  void* y;
  void* ctx;                 /* mgacbam_eca_ctx_bytes() */
  size_t ctx_bytes;          /* capacity of ctx (checked: MGACBAM_E_SIZE) */
  mgacbam_eca_params_t p;
  int32_t B, C, H, W;
  int32_t dtype;
} mgacbam_eca_fwd_level_t;

typedef struct mgacbam_eca_bwd_level {
  const void* x;
  const float* mask;
  const void* gy;
  const void* ctx;
  void* scratch;             /* mgacbam_eca_scratch_bytes() */
  size_t ctx_bytes;          /* capacities of ctx / scratch (checked: MGACBAM_E_SIZE) */
  size_t scratch_bytes;
  void* gx;
  float* gmask;              /* or NULL */
  float* gw;                 /* (1,1,k), overwritten */
  float* gbeta;
  mgacbam_eca_params_t p;
  int32_t B, C, H, W;
  int32_t dtype;
} mgacbam_eca_bwd_level_t;

size_t mgacbam_eca_ctx_bytes(int B, int C, int H, int W);
size_t mgacbam_eca_scratch_bytes(int B, int C, int H, int W);
int mgacbam_eca_forward(const mgacbam_eca_fwd_level_t* levels, int n_levels, void* stream);    /* 2 launches */
int mgacbam_eca_backward(const mgacbam_eca_bwd_level_t* levels, int n_levels, void* stream);   /* 2 launches */

/* ------------------------------------------------------------------------------------------------
 * Multi-scale segmentation loss on the mask logits: replaces SegmentationLoss.forward in its default mode,
 * mga_yolo/nn/losses/segmentation.py:87-151 (BCEWithLogits(mean) + soft Dice per level, scale weights, loss_lambda) and its
 * autograd backward.  Targets at another resolution are gathered inside the kernels: F.interpolate(mode="nearest")'s index rule
 * (segmentation.py:110) or the 4-tap bilinear rule with align_corners=False (segmentation.py:103-108).  use_unified_focal selects the Unified Focal mode (_lmf :44-63, _lmft :65-85, combine :114-131).
 * ------------------------------------------------------------------------------------------------ */
#define MGASEG_MAX_LEVELS 4
typedef struct mgaseg_level {
  const void* logits;        /* (B,1,H,W) of dtype                                              */
  const float* target;       /* (B,1,Ht,Wt) fp32                                                */
  void* glogits;             /* (B,1,H,W) of dtype: written by mgaseg_backward (NULL in forward) */
  int32_t B, H, W, Ht, Wt;
  int32_t dtype;             /* MGACBAM_F32 / F16 / BF16                                        */
  float scale_weight;        /* SegLossConfig.scale_weights[i]                                  */
  int32_t resize;            /* target at another resolution: MGASEG_NEAREST (segmentation.py:110) or MGASEG_BILINEAR
                                (align_corners=False; the MGA_PROB_MODE branch, segmentation.py:103-108)                      */
} mgaseg_level_t;
enum { MGASEG_NEAREST = 0, MGASEG_BILINEAR = 1 };
typedef struct mgaseg_cfg {                                   /* SegLossConfig, segmentation.py:9-21 */
  float bce_weight, dice_weight, smooth, loss_lambda;
  int32_t use_unified_focal;
  float ufl_lambda, ufl_delta, ufl_gamma;
} mgaseg_cfg_t;

size_t mgaseg_ws_bytes(const mgaseg_level_t* levels, int n_levels);   /* workspace: kept from forward to backward */
/* ws_bytes: capacity of ws as allocated (checked against mgaseg_ws_bytes(): MGACBAM_E_SIZE)
 * out (device, 1 + 3*n floats): [0] total, then per level {bce | l_mf, dice | l_mft, combined} (the reference's log entries) */
int mgaseg_forward(const mgaseg_level_t* levels, int n_levels, const mgaseg_cfg_t* cfg, void* ws, size_t ws_bytes, float* out, void* stream);
/* gout: device scalar dL/d(total) */
int mgaseg_backward(const mgaseg_level_t* levels, int n_levels, const mgaseg_cfg_t* cfg, const void* ws, size_t ws_bytes, const float* gout,
                    void* stream);

/* ------------------------------------------------------------------------------------------------
 * MGAMaskHead: the producer of the mask logits, mga_yolo/nn/modules/segmentation.py:56-110 in its default configuration
 * (norm="bn", act=SiLU, dropout=0, out_channels=1):  Conv1x1(C -> hidden, no bias) -> BatchNorm2d -> SiLU -> Conv3x3(hidden -> 1) + bias,
 * and its autograd backward.  The 1x1 conv and its two backward products run on the matrix cores (fp32 MFMA 16x16x4).  Parameters are
 * the reference's state_dict entries, fp32, contiguous, on the device.  training != 0: BatchNorm uses batch statistics and updates
 * running_mean / running_var (momentum) and num_batches_tracked in place, as torch does; training == 0: running statistics, no update.
 * ------------------------------------------------------------------------------------------------ */
typedef struct mgahead_params {
  const float* w1;             /* proj.0.weight (hidden, C, 1, 1)          segmentation.py:81 */
  const float* bn_weight;      /* proj.1.weight (hidden)                   :83                */
  const float* bn_bias;        /* proj.1.bias (hidden)                                        */
  float* running_mean;         /* proj.1.running_mean (hidden), updated when training         */
  float* running_var;          /* proj.1.running_var (hidden)                                 */
  int64_t* num_batches_tracked;/* proj.1.num_batches_tracked () int64, or NULL                */
  const float* wh;             /* head.weight (1, hidden, 3, 3)            :92                */
  const float* bh;             /* head.bias (1)                                               */
  int32_t hidden;
  float eps, momentum;         /* BatchNorm2d.eps / .momentum (Ultralytics sets 1e-3 / 0.03: U/utils/torch_utils.py:570-572) */
  int32_t training;
} mgahead_params_t;

typedef struct mgahead_fwd_level {
  const void* x;               /* (B,C,H,W) feature of `dtype`                                 */
  void* logits;                /* (B,1,H,W) mask logits of `dtype` (fp32 with MGAHEAD_LOGITS_F32) */
  void* ctx;                   /* mgahead_ctx_bytes(): z, batch statistics (kept for backward) */
  size_t ctx_bytes;            /* capacity of ctx (checked: MGACBAM_E_SIZE)                    */
  mgahead_params_t p;
  int32_t B, C, H, W;
  int32_t dtype;
  int32_t flags;               /* MGAHEAD_LOGITS_F32                                           */
} mgahead_fwd_level_t;

typedef struct mgahead_bwd_level {
  const void* x;               /* as in forward                                                */
  const void* g_logits;        /* (B,1,H,W) dL/dlogits of `dtype` (fp32 with MGAHEAD_LOGITS_F32) */
  const float* g_logits2;      /* optional second addend of dL/dlogits, fp32 (B,1,H,W), or NULL: the logits feed the segmentation loss AND
                                  MaskCBAM (model.py:57-64, 196-202); passing MaskCBAM's dL/dmask here saves the add launch */
  const void* ctx;             /* written by the matching forward                              */
  void* scratch;               /* mgahead_bwd_scratch_bytes(), contents undefined              */
  size_t ctx_bytes;            /* capacities of ctx / scratch (checked: MGACBAM_E_SIZE)        */
  size_t scratch_bytes;
  void* gx;                    /* (B,C,H,W) dL/dx of `dtype`                                   */
  float* gw1;                  /* parameter gradients, shapes of the parameters, OVERWRITTEN   */
  float* gbn_weight;
  float* gbn_bias;
  float* gwh;
  float* gbh;
  mgahead_params_t p;          /* training as in the forward; running statistics are not touched */
  int32_t B, C, H, W;
  int32_t dtype;
  int32_t flags;               /* MGAHEAD_BWD_ACCUM_GX: gx already holds the gradient of the feature's OTHER consumer (in the layer loop the
                                  feature feeds both its mask head and its MaskCBAM, model.py:57-64) and this call ADDS to it in the GEMM
                                  epilogue -- the feature-sized add autograd would do as a pass of its own disappears            */
} mgahead_bwd_level_t;
enum {
  MGAHEAD_BWD_ACCUM_GX = 1,
  MGAHEAD_LOGITS_F32 = 2       /* forward: logits, backward: g_logits are fp32 whatever `dtype` is -- half-precision features with the mask
                                  logits handed to MaskCBAM (whose mask input is fp32) and to the loss without a conversion pass        */
};

size_t mgahead_ctx_bytes(int B, int C, int H, int W, int hidden);
size_t mgahead_bwd_scratch_bytes(int B, int C, int H, int W, int hidden);
int mgahead_forward(const mgahead_fwd_level_t* levels, int n_levels, void* stream);     /* 3 launches for all levels at hidden <= 128 (GEMM, statistics, output); 4 when levels with hidden > 128 are present (their GEMM is a launch of its own) */
int mgahead_backward(const mgahead_bwd_level_t* levels, int n_levels, void* stream);    /* 5 launches for all levels */

/* Kendall multi-task combine of MGAModel.loss (mga_yolo/model/model.py:204-206), on the device so that no loss value has to visit
 * the host:  total[i] = exp(-s_det) * det[i] + s_det + exp(-s_seg) * seg + s_seg   for the n_det entries of the detection-loss vector
 * (box, cls, dfl); log_vars = {s_det, s_seg} (the learnable mtl_log_vars).  Backward: g_det[i], g_seg, g_log_vars[2] from g_total[i].
 * All fp32 device pointers; one tiny launch each. */
int mgakendall_forward(const float* det, int n_det, const float* seg, const float* log_vars, float* total, void* stream);
int mgakendall_backward(const float* det, int n_det, const float* seg, const float* log_vars, const float* g_total,
                        float* g_det, float* g_seg, float* g_log_vars, void* stream);

/* The loss tail of MGAModel.loss in one piece (model.py:196-206): mgaseg_forward followed by the combine on its total, and the backward
 * of both -- same results as the two calls, with the combine riding in the loss's last forward launch / its backward launch (two
 * launches fewer per step; every one of these kernels is launch-latency-bound).  `out` = mgaseg_forward's out (1 + 3 * n_levels floats;
 * out[0] = the seg total the combine reads).  Backward: levels[l].glogits <- d total.sum-weighted / d logits_l, g_det[n_det],
 * g_log_vars[2]; g_seg (dL/d seg total) may be NULL. */
int mgaseg_kendall_forward(const mgaseg_level_t* levels, int n_levels, const mgaseg_cfg_t* cfg, void* ws, size_t ws_bytes, float* out,
                           const float* det, int n_det, const float* log_vars, float* total, void* stream);
int mgaseg_kendall_backward(const mgaseg_level_t* levels, int n_levels, const mgaseg_cfg_t* cfg, const void* ws, size_t ws_bytes, const float* out,
                            const float* det, int n_det, const float* log_vars, const float* g_total,
                            float* g_det, float* g_seg, float* g_log_vars, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ProbMaskGater: mga_yolo/nn/modules/probmaskgater.py:58-98, training mode, 'gumbel' (hard = 0) and 'hard_st'
 * (hard = 1).  u1, u2: the two uniform tensors the reference draws with torch.rand (:53-56, 66-67), drawn by the caller the same
 * way, so the result is a pure function of its inputs.  All tensors fp32 with n elements; msoft is kept for the backward.
 * ------------------------------------------------------------------------------------------------ */
typedef struct mgapmg_cfg { float tau, p_min, threshold; int32_t hard; } mgapmg_cfg_t;
int mgapmg_forward(const float* p, const float* u1, const float* u2, float* out, float* msoft, size_t n, const mgapmg_cfg_t* cfg,
                   void* stream);
int mgapmg_backward(const float* p, const float* msoft, const float* gout, float* gp, size_t n, const mgapmg_cfg_t* cfg, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MGACBAM_H_ */

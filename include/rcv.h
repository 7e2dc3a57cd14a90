/*
 * rcv.h -- C ABI of librcv.so, the MI355X (gfx950) implementation of the RoboCupVision
 * ROBO-UNet / U-Net training hot path.
 *
 * The reference (szemenyeim/RoboCupVision) has no FFI of its own: its hot path bottoms out in
 * PyTorch ATen operators called from model.py / train.py.  This header is therefore the boundary
 * a maintainer would bind *instead of* those operator calls; every entry point names the
 * reference call site it replaces (file:line into the reference repository).
 *
 * Conventions
 *   - extern "C", plain pointers and integers only.  No torch / C++ types cross the boundary.
 *   - Every function returns 0 on success, a negative RCV_E_* code on failure; the text of the
 *     last failure on the calling thread is returned by rcv_last_error().  Nothing throws.
 *   - All pointers are device pointers BORROWED for the duration of the call (they must stay valid
 *     until the work enqueued on `stream` has completed).  The library never allocates device
 *     memory: outputs and workspaces are passed in; sizes come from rcv_op_workspace().
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).  All work is
 *     enqueued asynchronously on it; nothing here synchronises the device (graph-capture safe).
 *   - Activations are fp32 NHWC ([N][H][W][C], C contiguous).  The network input (images) and the
 *     network output (logits) are fp32 NCHW, as the reference's callers hand over / expect.
 *   - All reductions are fixed-order (no float atomics): results are bitwise reproducible.
 *
 * Execution model: the host describes each kernel invocation as one `rcv_op` record; a whole
 * forward or backward pass is an array of records executed by ONE call to rcv_run() (one host
 * call per pass instead of one per layer; the records are plain data and may be cached).
 * The named entry points further down are conveniences that fill a record and run it.
 */
#ifndef RCV_H
#define RCV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RCV_VERSION 100

/* error codes */
#define RCV_OK            0
#define RCV_E_ARG        -1   /* bad argument / unsupported shape */
#define RCV_E_HIP        -2   /* a HIP runtime call failed        */
#define RCV_E_UNSUPPORTED -3

typedef struct rcv_handle rcv_handle;

int  rcv_create(int device, rcv_handle** out);
/* A handle without a device, for hosts that only lay out buffers: rcv_op_workspace / rcv_op_kernel_label answer for a chip with
 * `num_cus` compute units (256 on MI355X); every call that would enqueue work fails with RCV_E_ARG. */
int  rcv_create_planner(int num_cus, rcv_handle** out);
int  rcv_destroy(rcv_handle* h);
const char* rcv_last_error(void);
int  rcv_version(void);
/* Multiprocessor (CU) count of the handle's device; used by callers to size split-K. */
int  rcv_num_cus(const rcv_handle* h);

/* ------------------------------------------------------------------------------------------ */
/* Operation records                                                                           */
/* ------------------------------------------------------------------------------------------ */

/* kinds */
enum {
  RCV_OP_CONV        = 1,  /* 3x3 gather convolution (implicit GEMM on MFMA)                     */
  RCV_OP_TCONV       = 2,  /* 3x3 stride-2 transposed convolution, pad 1, output_padding 1       */
  RCV_OP_WGRAD       = 3,  /* 3x3 filter gradient, split over pixels, partials to workspace      */
  RCV_OP_WGRAD_REDUCE= 4,  /* fixed-order sum of WGRAD partials -> parameter-layout gradient     */
  RCV_OP_PACK        = 5,  /* parameter tensors -> kernel filter layout, table driven            */
  RCV_OP_BN_FINALIZE = 6,  /* batch statistics -> scale/shift (+ running stats)                  */
  RCV_OP_BN_EVAL     = 7,  /* running statistics -> scale/shift; row 3 = bias folded through the BatchNorm: (p[X4] ? bias : 0)*scale + shift */
  RCV_OP_BN_BWD      = 8,  /* backward reductions -> (A,B,C) constants, dgamma, dbeta            */
  RCV_OP_COMBINE     = 9,  /* up = relu(t*s+h) + (r*s2+h2)          (decoder skip add)           */
  RCV_OP_CLS_FWD     = 10, /* 1x1 classifier, NHWC in (8 or 16 channels) -> NCHW logits, 1..8 classes (model.py:411; numClass = 5 - nb - ng - nr - nl, train.py:301) */
  RCV_OP_CLS_BWD     = 11, /* classifier backward (dgrad + wgrad + dbias) from NCHW dlogits, same shapes */
  RCV_OP_CE_FWD      = 12, /* weighted softmax cross-entropy forward (+argmax, +#correct)        */
  RCV_OP_CE_BWD      = 13, /* d loss / d logits                                                  */
  RCV_OP_POOL_FWD    = 14, /* 2x2/2 max-pool of (r*s+h)                                          */
  RCV_OP_POOL_BWD    = 15, /* max-pool backward (+skip gradient, + BN backward reductions)       */
  RCV_OP_ADAM_L1     = 16, /* fused Adam step with decay*sign(p) L1 gradient over a flat buffer  */
  RCV_OP_MEMSET      = 17, /* zero a float buffer                                                */
  RCV_OP_CONV1X1     = 18, /* generic 1x1 convolution NHWC -> NHWC/NCHW (LabelProp classifier)   */
  RCV_OP_ADD_SLICE   = 19, /* x[..., 0:Ca] += affine(a)   (LabelProp top skip, model.py:565)     */
  RCV_OP_MATERIALIZE = 20, /* out = load(in)  (a block's normalised output as a plain tensor)    */
  RCV_OP_BWD_STATS   = 21, /* out = g[.., aux0:aux0+cout] (g has cin channels per pixel, 0 = cout); partial rows of the BN-backward sums */
  RCV_OP_CONFUSION   = 22, /* counts[n][pred][label] += 1 per pixel (int32, accumulating)            */
  RCV_OP_DICE_FWD    = 23, /* weighted soft-Dice loss forward (+argmax, +#correct)   model.py:5-43 */
  RCV_OP_DICE_BWD    = 24, /* d loss / d logits of the Dice loss                                  */
  RCV_OP_NHWC_TO_NCHW= 25, /* out[n][c][p] = in[n][p][c] + bias[c], c < cout <= cin (3x3 classifier tail) */
  RCV_OP_NCHW_TO_NHWC= 26, /* out[n][p][c] = c < cin ? in[n][c][p] : 0, cout channels per pixel     */
  RCV_OP_SGD         = 27, /* torch.optim.SGD(momentum, weight_decay) over a flat buffer (trainer.py:176-178) */
  RCV_OP_NOP         = 28, /* nothing is launched (a slot of an op list whose work was folded into a later record)       */
  RCV_OP_WGRAD_REDUCE_BATCH = 29 /* RCV_OP_WGRAD_REDUCE of several layers in ONE launch: p[RCV_P_IN] -> rcv_reduce_job[i[RCV_I_COUNT]]
                                   * (device memory); same fixed summation order per layer as the single form                  */
};

/* how an operand is produced from memory while it is staged (rcv_op.i[RCV_I_INMODE] etc.) */
enum {
  RCV_LOAD_PLAIN    = 0,   /* v = x                                                              */
  RCV_LOAD_AFFINE   = 1,   /* v = x*c0[ch] + c1[ch]                (BatchNorm apply of producer) */
  RCV_LOAD_GRAD_ENC = 2,   /* v = aux>0 ? c0*x + c1 + c2*aux : 0   (BN-bwd then ReLU-bwd)        */
  RCV_LOAD_GRAD_DEC = 3,   /* v = c0*(aux*c3+c4>0 ? x : 0) + c1 + c2*aux   (ReLU-bwd then BN-bwd)*/
  RCV_LOAD_NCHW     = 4,   /* v = x, tensor is NCHW (network input image, <= 4 channels)           */
  RCV_LOAD_AFFINE_RELU = 5 /* v = max(x*c0+c1, 0)               (conv->BN->ReLU producer)        */
};

/* epilogue statistics written as per-workgroup partial rows [n_part][2][C] */
enum {
  RCV_STATS_NONE   = 0,
  RCV_STATS_FWD    = 1,    /* sum v, sum v*v                     (BatchNorm batch statistics)    */
  RCV_STATS_BWD_ENC= 2,    /* sum g, sum g*e      e = epi_aux    (BN backward, conv->ReLU->BN)   */
  RCV_STATS_BWD_DEC= 3     /* sum g*m, sum g*m*e  m = (e*c0+c1>0) (BN backward, convT->BN->ReLU) */
};

/* flag bits (rcv_op.flags) */
#define RCV_F_BIAS      1u    /* add bias[co]                                                     */
#define RCV_F_RELU      2u    /* clamp at zero after bias                                         */
#define RCV_F_RESID     4u    /* add resid[...] (same shape as the output) before stats/store     */
#define RCV_F_OUT_NCHW  8u    /* CONV1X1 / CLS: store NCHW                                        */
#define RCV_F_FLIP      16u   /* PACK: reverse the 3x3 taps                                       */
#define RCV_F_TRANSPOSED_SRC 32u /* PACK / WGRAD_REDUCE: parameter is [Cin][Cout][3][3] (convT)  */
#define RCV_F_ARGMAX    64u   /* CE_FWD: also write argmax mask and count correct pixels          */
#define RCV_F_TRAINING  128u  /* BN_FINALIZE: update running stats                                */
#define RCV_F_FUSED_UP   512u  /* CLS_FWD / CLS_BWD: the input is the decoder output relu(t*c0+c1) + f(r), formed on the fly from t (p[IN] resp.
                               * p[EPI_AUX]), its constants (p[IN_C] resp. p[EPI_C]), the skip tensor p[X3], its constants p[X4], i[AUX0] = its
                               * load mode -- RCV_OP_COMBINE is then not needed for this value                                             */
#define RCV_F_FUSED_CE   1024u /* with RCV_F_FUSED_UP: CLS_FWD also produces the weighted cross entropy (p[IN2] = int64 target, p[X0] = class weights or
                               * NULL, p[PART], p[X1] = float[4] loss_out as RCV_OP_CE_FWD, p[X2] = uint8 arg-max or NULL); CLS_BWD forms d loss / d logits
                               * itself (p[IN2] = target, p[X0] = class weights, p[BIAS], p[X5] = loss_out, p[IN2_AUX] = d loss scalar): the
                               * logits gradient tensor and RCV_OP_CE_BWD are not needed                                              */
#define RCV_F_SIDE_STREAM (1u << 16) /* rcv_run: enqueue this op on the handle's side stream (forked from / joined to the caller's
                                      * stream inside the call): ops off the critical path, e.g. the filter gradients of backward */
#define RCV_F_MFMA_FP32 (1u << 17) /* CONV / TCONV / WGRAD: contract on the fp32 matrix instructions (v_mfma_f32_16x16x4_f32) only.  Without it the
                                      wide layers form their fp32 products on the bf16 matrix pipe from operands split EXACTLY into three
                                      bf16 values (six partial products per multiply-add, fp32 accumulate; error <= the fp32 chain's against
                                      fp64, csrc/wgrad_bf3.hip): same results to fp32 rounding, 2-2.5 x the matrix rate.  Part of the plan key. */
#define RCV_F_CONCAT    256u  /* COMBINE: out[..,0:C] = relu(t*s+h), out[..,C:2C] = f(r)  (v2 skip concat, model.py:507) */
/* bits 20..22: profiling ablations of diagnostic builds (skip staging / skip the contraction); the shipped kernels of the wide
 * layers ignore them */
#define RCV_F_DBG_NOSTAGE (1u << 20)
#define RCV_F_DBG_NOSKIP  (1u << 22)
#define RCV_F_DBG_NOMFMA  (1u << 21)
#define RCV_F_DBG_NOEPI   (1u << 23)   /* filter gradient: skip the partial-filter stores (ablation timing only) */

/* integer slots */
enum {
  RCV_I_N = 0, RCV_I_H, RCV_I_W,          /* spatial dims of the gathered (input) tensor          */
  RCV_I_CIN, RCV_I_COUT,
  RCV_I_HO, RCV_I_WO,                     /* spatial dims of the output tensor                    */
  RCV_I_STRIDE, RCV_I_DIL,
  RCV_I_INMODE,                           /* RCV_LOAD_* of operand `in`                           */
  RCV_I_INMODE2,                          /* WGRAD: RCV_LOAD_* of the pointwise operand           */
  RCV_I_STATS,                            /* RCV_STATS_*                                          */
  RCV_I_NPART,                            /* rows of `part` (filled by rcv_op_workspace)          */
  RCV_I_NSPLIT,                           /* WGRAD: pixel splits (filled by rcv_op_workspace)     */
  RCV_I_COUNT,                            /* element / job count for table driven ops             */
  RCV_I_AUX0, RCV_I_AUX1,                 /* TCONV: AUX0 = 1 when the filter is packed in the merged-parity layout */
  RCV_I__N = 20
};

/* pointer slots */
enum {
  RCV_P_IN = 0,      /* gathered operand                                                          */
  RCV_P_IN_AUX,      /* second tensor of a GRAD_* load                                            */
  RCV_P_IN_C,        /* per-channel constants of the load: float[5][C] planar (c0..c4)            */
  RCV_P_W,           /* packed filter / parameter                                                 */
  RCV_P_BIAS,
  RCV_P_OUT,
  RCV_P_RESID,
  RCV_P_EPI_AUX,     /* tensor e of the BWD statistics                                            */
  RCV_P_EPI_C,       /* float[2][C]: (c0,c1) of RCV_STATS_BWD_DEC                                 */
  RCV_P_PART,        /* partial rows                                                              */
  RCV_P_IN2,         /* WGRAD: pointwise operand                                                  */
  RCV_P_IN2_AUX,
  RCV_P_IN2_C,
  RCV_P_X0, RCV_P_X1, RCV_P_X2, RCV_P_X3, RCV_P_X4, RCV_P_X5,   /* kind specific            */
  RCV_P__N = 20
};

typedef struct rcv_op {
  int32_t  kind;
  uint32_t flags;
  int32_t  i[RCV_I__N];
  float    f[8];
  void*    p[RCV_P__N];
} rcv_op;

/* Fills op->i[RCV_I_NPART] / [RCV_I_NSPLIT] for the tiling the library will use and returns
 * the bytes of the `part` workspace the op needs (0 if none).  This query (and rcv_op_kernel_label) REFUSES exactly the records a
 * launch would refuse for their shape -- channel counts, load modes, flag combinations; operand pointers and workspace row counts are
 * the only things checked at launch alone -- so a caller can validate a whole op list when it builds it (RCV_E_ARG + rcv_last_error). */
int rcv_op_workspace(const rcv_handle* h, rcv_op* op, size_t* part_bytes);

/* Enqueue ops[0..n) in order on `stream`.  The handle's device is made current for the duration of the call (and the caller's
 * current device restored): `stream` must belong to the handle's device. */
int rcv_run(rcv_handle* h, const rcv_op* ops, int n, void* stream);

/* rcv_run with options.  RCV_RUN_NO_JOIN: leave the side stream (RCV_F_SIDE_STREAM ops) un-joined when the call returns; the caller
 * joins it later with rcv_join_side (data-parallel training runs the backward list in slices and lets the communication stream,
 * not the compute stream, wait for the filter gradients of a slice). */
#define RCV_RUN_NO_JOIN 1u
int rcv_run_ex(rcv_handle* h, const rcv_op* ops, int n, void* stream, uint32_t run_flags);
/* Make `stream` wait for everything enqueued on the handle's side stream so far (no-op if none exists). */
int rcv_join_side(rcv_handle* h, void* stream);

/* Profiling aid (not for the training path: it creates HIP events and synchronises the stream):
 * runs ops[0..n) like rcv_run with a hipEvent pair around every op and writes the elapsed
 * milliseconds of op k to ms[k] (host memory). */
int rcv_run_timed(rcv_handle* h, const rcv_op* ops, int n, void* stream, float* ms);

/* Filter layout the library wants for an RCV_OP_CONV record before its filter is packed: 0 = [9 taps][Cin][Cout] (rcv_pack_job.merged
 * 0), 2 = Winograd F(2x2,3x3) transformed [16][Cin][Cout] (rcv_pack_job.merged 2), 3 / 4 / 5 = the split-bf16 layouts of
 * rcv_pack_job.merged (1.5 x the bytes of the plain layout); the record then carries the answer in i[RCV_I_AUX0].  RCV_OP_CONV: the wide
 * (Cin % 32 == 0, >= 64 channels) stride-1 layers whose grid covers the chip answer 3 (2 when the record carries RCV_F_MFMA_FP32), the
 * wide stride-2 layers (Cin % 16 == 0, >= 32; Cout >= 64) answer 5, the 16 / 32-channel layers whose fp32 form is matrix-pipe bound 3;
 * RCV_OP_TCONV records in the merged form (i[RCV_I_AUX0] = 1) answer 4 where the split-bf16 narrow kernel is the faster one;
 * `force` != 0 answers 2 for every shape the Winograd kernel can run. */
int rcv_op_filter_layout(const rcv_handle* h, const rcv_op* op, int force);

/* Label of the kernel (template instantiation / tiling) the library launches for `op`, e.g.
 * "conv_mfma<2,5,4,1,8>" -- written NUL-terminated into buf[0..size). */
int rcv_op_kernel_label(const rcv_handle* h, const rcv_op* op, char* buf, int size);

/* One row of the RCV_OP_PACK job table (device memory, p[RCV_P_IN] -> rcv_pack_job[count]). */
typedef struct rcv_pack_job {
  const float* src;   /* parameter tensor [D0][D1][3][3]                                          */
  float*       dst;   /* [9][rows_pad][cols_pad], zero filled pads                                */
  int32_t D0, D1;
  int32_t rows_from_d1;   /* 1: rows (contraction channel) = d1, cols = d0; 0: rows = d0, cols=d1 */
  int32_t flip;           /* 1: tap t reads source tap 8-t                                        */
  int32_t rows_pad, cols_pad;
  int32_t merged;         /* 1: transposed-conv "merged parity" layout [4 taps (dy,dx)][rows][4*cols] (see conv_mfma.hip);
                           * 2: Winograd layout [16][rows][cols] = G g G^T (see conv_wino.hip);
                           * 3: split-bf16 layout: every value as three bf16 (v = h + m + l exactly); with k = tap * rows_pad + row,
                           *    dst = [plane][k / 32][cols_pad][k % 32] bf16 (k-steps of 32, zero beyond the last tap; rows_pad a multiple of 8,
                           *    of 32 above 32; see conv_bf3.hip / convn_bf3.hip);
                           * 4: layout 1 (merged parity, 4 taps, 4 * cols virtual columns) split the same way (convn_bf3.hip);
                           * 5: split-bf16 in 16-row chunks for the stride-2 wide convs: k = tap * 16 + row % 16 inside chunk row / 16,
                           *    dst = [plane][chunk][5 k-steps][cols_pad][32] bf16 (rows_pad a multiple of 16; conv2_bf3_kernel)      */
  int32_t reserved;
  const float* scale; /* NULL, or one factor per OUTPUT channel (column) applied while packing: inference folds an eval-mode BatchNorm
                       * that follows the conv directly (relu(bn(conv(x))), model.py:175,190-194) into the filter, w'[co] = w[co] * scale[co] */
} rcv_pack_job;

/* One row of the RCV_OP_WGRAD_REDUCE_BATCH job table: the arguments of one RCV_OP_WGRAD_REDUCE record. `first_block` = number of
 * 64-element blocks of the jobs before this one (job j owns blocks [first_block_j, first_block_{j+1}); a block sums 64 consecutive
 * elements of the partial layout [9][CBP][CAP] (+ the bias row): blocks_j = ceil((9*CBP*CAP + (db ? CBP : 0)) / 64), with
 * CAP = CA <= 4 ? 4 : round16(CA), CBP = round16(CB)); i[RCV_I_NPART] of the record = total number of blocks.
 * nsplit == 0: a zero-fill job, db[0..CB) = 0 in ceil(CB / 256) blocks (the RCV_OP_MEMSET of a bias gradient, folded in). */
typedef struct rcv_reduce_job {
  const float* part;  /* [nsplit][9][CBP][CAP] then, if db, [nsplit][CBP]                          */
  float*       dw;    /* [CB][CA][3][3]                                                            */
  float*       db;    /* [CB] or NULL                                                              */
  int32_t nsplit, CB, CA, first_block;
} rcv_reduce_job;

/* ------------------------------------------------------------------------------------------ */
/* Named entry points (each = fill one record + rcv_run).  Reference call sites they replace:  */
/* ------------------------------------------------------------------------------------------ */

/* nn.Conv2d(k=3,pad=dil,stride,dilation) forward, fused bias / ReLU / BN statistics, optional
 * BatchNorm-apply of the producer folded into the load.   model.py:112,115-116; model.py:170,175
 * Also the data gradient of a stride-1 conv (flipped, transposed filter) and of the transposed
 * conv (aten::convolution_backward under train.py:57).                                          */
int rcv_conv3x3(rcv_handle* h, const rcv_op* op, void* stream);
/* nn.ConvTranspose2d(k=3,s=2,p=1,output_padding=1) forward (model.py:186-187,191) and the data
 * gradient of a stride-2 conv.                                                                  */
int rcv_convT3x3s2(rcv_handle* h, const rcv_op* op, void* stream);
/* filter gradients of both (aten::convolution_backward).                                        */
int rcv_wgrad3x3(rcv_handle* h, const rcv_op* op, void* stream);

/* nn.BatchNorm2d train-mode statistics -> normalisation constants (model.py:113,116,189,192).   *
 *   part [n_part][2][C] -> scale,shift (float[5][C] slot layout c0,c1), mean, istd, running.    */
int rcv_bn_finalize(rcv_handle* h, const float* part, int n_part, int C, double count,
                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                    float momentum, float eps, int training,
                    float* consts /*[5][C]*/, float* save_mean, float* save_istd, void* stream);
/* aten::native_batch_norm_backward reductions -> load constants for RCV_LOAD_GRAD_*.            */
int rcv_bn_backward(rcv_handle* h, const float* part, int n_part, int C, double count,
                    const float* gamma, const float* save_mean, const float* save_istd,
                    const float* fwd_consts, int decoder,
                    float* consts /*[5][C]*/, float* dgamma, float* dbeta, void* stream);

/* nn.MaxPool2d(2,2) of the normalised producer (model.py:97-100).                               */
int rcv_maxpool2x2_fwd(rcv_handle* h, const float* r, const float* consts, float* out,
                       int N, int H, int W, int C, void* stream);

/* CrossEntropyLoss2d (model.py:76-82) + torch.max(pred,1) / pixel accuracy (train.py:70-71).    *
 *   logits NCHW, target int64 [N][H][W]; loss_out[0]=loss, [1]=sum_w, [2]=#correct (as float).  */
int rcv_softmax_ce_argmax_fwd(rcv_handle* h, const float* logits, const int64_t* target,
                              const float* class_weight /*may be NULL*/, int N, int C, int H, int W,
                              float* part, int n_part, float* loss_out, uint8_t* argmax /*may be NULL*/,
                              void* stream);
int rcv_softmax_ce_bwd(rcv_handle* h, const float* logits, const int64_t* target,
                       const float* class_weight, const float* loss_out, const float* grad_out,
                       int N, int C, int H, int W, float* dlogits, void* stream);

/* DiceLoss (model.py:5-43, multi-class branch; train.py:315 --useDice) + arg-max / pixel accuracy.  class_weight is
 * the already rescaled weight vector (model.py:8).  out: float[4 + 16]: [0] = loss, [2] = #correct, [4..] = the
 * per-class coefficients rcv_dice_bwd consumes.                                                                   */
int rcv_dice_fwd(rcv_handle* h, const float* logits, const int64_t* target, const float* class_weight /*may be NULL*/,
                 float eps, int N, int C, int H, int W, float* part, int n_part, float* out,
                 uint8_t* argmax /*may be NULL*/, void* stream);
int rcv_dice_bwd(rcv_handle* h, const float* logits, const int64_t* target, const float* fwd_out, const float* grad_out,
                 int N, int C, int H, int W, float* dlogits, void* stream);

/* train.py:23-27,52-55 (decay * L1 -> gradient decay*sign(p)) + torch.optim.Adam.step            *
 * (train.py:67,357-363) over one flat fp32 buffer; lr is per element group via lr_scale[].      */
/* Per-image confusion matrices from the arg-max mask (uint8) and the labels (int64): replaces the Python
 * mask loops of valid() (train.py:136-153).  counts is int32 [N][C][C] indexed [n][pred][label]; the call ADDS.   */
int rcv_confusion(rcv_handle* h, const uint8_t* argmax, const int64_t* target, int N, int C, int H, int W,
                  int32_t* counts, void* stream);

/* torch.optim.SGD.step (trainer.py:176-178,221): g = grad*grad_scale + weight_decay*p; buf = step==1 ? g : momentum*buf + g;
 * p -= lr*buf.  lr_elem (may be NULL) gives a per-element learning rate (0 = parameter without a gradient: untouched). */
int rcv_sgd_step(rcv_handle* h, float* param, const float* grad, float* momentum_buf, const float* lr_elem /*may be NULL*/,
                 int64_t n, float lr, float momentum, float weight_decay, int step, float grad_scale, void* stream);

/* As an op record, p[RCV_P_IN_AUX] (may be NULL) is a device int32 holding the 1-based step number: it overrides `step` (the
 * bias corrections are then formed on the device), so that a captured graph of the whole training step can be replayed.
 * lr_elem (may be NULL): per-element learning rate; 0 = the element is not stepped at all (a parameter without a gradient).
 * As an op record, p[RCV_P_X5] (may be NULL) is the prune mask of train.py:59-65 (`param.grad[indices] = 0` after backward):
 * uint8 per element of the flat buffer, non-zero = the whole gradient of that element (L1 part included) is zero this step. */
int rcv_adam_l1_step(rcv_handle* h, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                     const float* lr_elem /*may be NULL*/, int64_t n, float lr, float beta1, float beta2,
                     float eps, float decay, int step, float grad_scale, void* stream);
/* the same with the prune mask (rcv_adam_l1_step passes NULL) */
int rcv_adam_l1_step_pruned(rcv_handle* h, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                            const float* lr_elem /*may be NULL*/, const uint8_t* prune_mask /*may be NULL*/, int64_t n, float lr,
                            float beta1, float beta2, float eps, float decay, int step, float grad_scale, void* stream);

/* The same step, which also books the iteration's metrics (train.py:52-53,69-73) in the same launch:
 * metrics[4] (double, device) += { loss_stats[0] + decay*sum|p|, decay*sum|p|, loss_stats[2], 1 } with sum|p| taken before
 * the update (the reference's l1reg(model)); loss_stats = the float row the loss op wrote ([0] loss, [2] #correct pixels).
 * workspace: rcv_op_workspace bytes of an RCV_OP_ADAM_L1 op with the same n (zero-filled once; the launch leaves its ticket
 * zero); workspace_rows = the op's i[RCV_I_NPART].  As an op: p[RCV_P_X3] = metrics, p[RCV_P_X4] = loss_stats, p[RCV_P_PART]. */
int rcv_adam_l1_step_metrics(rcv_handle* h, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                             const float* lr_elem /*may be NULL*/, int64_t n, float lr, float beta1, float beta2,
                             float eps, float decay, int step, float grad_scale, double* metrics, const float* loss_stats,
                             void* workspace, int workspace_rows, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RCV_H */

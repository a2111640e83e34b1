/*
 * yolo_hip.h -- C-ABI of libyolo_hip.so: the MI355X (gfx950) kernels behind the
 * tf2_YOLO drop-in (yolov{1_5,2,3,4}.Yolo / create_model / loss / metrics,
 * utils.tools.decode / nms / soft_nms).
 *
 * The reference (samson6460/tf2_YOLO) is 100 % Python on top of tf.keras and has no
 * FFI of its own, so every entry point below cites the
 * reference call site whose arithmetic it replaces (path:line under /root/reference).
 *
 * Conventions (all functions):
 *   - extern "C", plain pointers / ints / floats, no C++ or torch types.
 *   - every pointer is a DEVICE pointer unless the name ends in _host.
 *   - `stream` is a hipStream_t passed as void*; work is enqueued asynchronously, the
 *     library never synchronises the device and never allocates device memory: the
 *     caller owns all buffers including workspaces (sizes via the *_workspace_bytes
 *     queries or documented per function).
 *   - return value: 0 = YOLO_OK, negative = yolo_status; the message for the last
 *     failure on the calling thread is available from yolo_last_error().
 *   - activations are dense NHWC float32; filters are "KRSC" = [Cout][kh][kw][Cin]
 *     float32 (converted once from Keras HWIO by the host side).
 */
#ifndef YOLO_HIP_H_
#define YOLO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum yolo_status {
  YOLO_OK = 0,
  YOLO_ERR_INVALID_ARG = -1,   /* bad shape / null pointer / unsupported parameter */
  YOLO_ERR_LAUNCH = -2,        /* hipLaunch / hip runtime error                      */
  YOLO_ERR_WORKSPACE = -3      /* workspace too small                                */
} yolo_status;

/* activation codes shared by the bn_act / act kernels */
enum { YOLO_ACT_LINEAR = 0, YOLO_ACT_LEAKY = 1, YOLO_ACT_MISH = 2 };
/* head layouts */
enum { YOLO_HEAD_V3 = 3, YOLO_HEAD_V2 = 2, YOLO_HEAD_V1 = 1, YOLO_HEAD_V4 = 4 };
/* nms modes: utils/tools.py:687-786 (nms iou_mode=1, nms iou_mode=2 (DIoU), soft_nms) */
enum { YOLO_NMS_HARD = 1, YOLO_NMS_SOFT = 2, YOLO_NMS_DIOU = 3 };

const char* yolo_last_error(void);
/* library/ABI version, bumped when a signature changes */
int yolo_abi_version(void);
/* 1 if a HIP device is visible to the calling process, 0 otherwise (never throws) */
int yolo_device_available(void);
/* Yardstick for the roofline report (no reference counterpart): ONE launch of bare v_mfma_f32_32x32x16_f16 loops (`iters`
 * iterations of 6 MFMAs per wave, `workgroups` workgroups of 8 waves) on the caller's fp16 operands -- operands_f16 holds
 * workgroups * 512 * 32 halves (random data: the clock a chip holds depends on the switching activity), sink
 * workgroups * 512 floats. *flops_host (optional, HOST pointer) receives the MFMA FLOPs of the launch; the caller times it
 * with events on `stream`. */
int yolo_mfma_probe(const void* operands_f16, float* sink, int workgroups, int iters, double* flops_host, void* stream);

/* Run-time tuning / diagnostic switches of the library (no reference counterpart). key 0 = YOLO_OPT_CONV_WIN:
 * kernel used by yolo_conv2d_fwd_planes / _dgrad_planes for 3x3 stride-1 layers: 0 = per-tap streaming kernel,
 * 1 = input-window kernel with automatic tile choice, 2 / 4 = window kernel with 128x128 / 256x128 tiles.
 * Results are identical up to fp32 summation order. Defaults come from the environment (YOLO_CONV_WIN).
 * key 2 = YOLO_OPT_CONV_SK (YOLO_CONV_SK): what happens to launches whose tiles would leave most of the chip idle
 * (bs-1 inference): 0 = nothing, 1 = split-K (every tile computed by up to 32 workgroups, a reduce kernel adds the parts
 * in order and runs the epilogue), -1 = the stream-K form of the window kernel, > 1 = stream-K with that many workgroups.
 * key 5 = YOLO_OPT_CONV_PATCH (YOLO_CONV_PATCH): the window kernel on 2-D patches of 8 x 16 (16 x 16 for Cout <= 64) output
 * pixels for 3x3 stride-1 layers: 0 = off, 1 = automatic (rows longer than 64 pixels, Cout <= 64), 2 = wherever possible.
 * key 6 = YOLO_OPT_WGRAD_WIN (env YOLO_WGRAD_WIN, default 1): the 3x3 stride-1 filter gradient that streams the input once through an
 * LDS ring and takes all nine taps from it (conv_wgrad_win.hip): 0 = the per-tap kernel everywhere; wherever the shape allows: 1 = on
 * v_mfma_f32_32x32x16_f16, 3 = on v_mfma_f32_16x16x32_f16 (holds a higher clock on random data).
 * key -1 resets every option to its default. */
enum { YOLO_OPT_CONV_WIN = 0, YOLO_OPT_STAMPS = 1, YOLO_OPT_CONV_SK = 2, YOLO_OPT_CONV_PATCH = 5, YOLO_OPT_WGRAD_WIN = 6,
       YOLO_OPT_NMS_WALK = 7 /* (env YOLO_NMS_WALK, default 0) hard / DIoU NMS: 1 = the greedy walk kernel for every class instead of the
                                pair bit matrix + walk over the bits (same rows either way; tests) */,
       YOLO_OPT_AB = 8 /* (env YOLO_EXP, default 0) A/B bits of round 6, results identical either way: 8 = the loss kernel loads one
                          64-channel chunk ahead instead of one cell ahead, 16 = forward convolutions with BatchNorm statistics are
                          never split over several workgroups (as until round 5) */ };
int yolo_set_option(int key, int value);
/* Scratch for the split-K / stream-K forms of the planes convolutions (key 2 = YOLO_OPT_CONV_SK != 0): the accumulator
 * slabs of tiles computed by several workgroups (+ the stream-K form's tile tickets). The caller owns the memory
 * (yolo_conv_workspace_bytes() bytes, device); the call zeroes the ticket area on `stream`. ONE workspace per process:
 * every yolo_conv2d_fwd_planes(_epi) / _dgrad_planes call must be ordered after this call and after each other on one
 * stream (two models convolving on two streams at once would share the slabs; the kernels leave the tickets zero again).
 * p == NULL unregisters (one workgroup per tile everywhere). */
size_t yolo_conv_workspace_bytes(void);
int yolo_set_conv_workspace(void* p, size_t bytes, void* stream);
/* Diagnostic builds only (key 1 = YOLO_OPT_STAMPS != 0): device buffer that receives 8 x uint64 clock stamps per
 * workgroup of the stamped kernel variants; never read by any kernel. */
int yolo_set_debug_buffer(void* p, size_t bytes);

/* ------------------------------------------------------------------------------------
 * Convolution (replaces tf.keras Conv2D at yolov3/models/backbone.py:27-36,
 * yolov4/models/backbone.py:63-74, yolov{1_5,2}/models/backbone.py:9-18 and the head
 * convs yolov3/models/__init__.py:40-58).
 *
 * Implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32), NHWC, im2col-free:
 *   y[n,ho,wo,co] = bias[co] + sum_{r,s,ci} x[n, ho*sh + r - pad_t, wo*sw + s - pad_l, ci] * w[co,r,s,ci]
 * Zero padding is implicit (pad_t/pad_l, the bottom/right pad follows from Ho/Wo), which
 * covers Keras 'same' (asymmetric for stride 2), 'valid', and ZeroPadding2D((1,0),(1,0)).
 * ------------------------------------------------------------------------------------ */
typedef struct yolo_conv_desc {
  int N, H, W, Cin;      /* input  x: [N,H,W,Cin]                  */
  int Cout, kh, kw;      /* filter w: [Cout,kh,kw,Cin]              */
  int Ho, Wo;            /* output y: [N,Ho,Wo,Cout]                */
  int sh, sw;            /* strides                                 */
  int pad_t, pad_l;      /* implicit zero padding top / left        */
} yolo_conv_desc;

/* BatchNorm statistics buffers hold YOLO_BN_STAT_SLOTS replicas of [2*C] doubles (sum, sum of
 * squares); producers spread their atomics over the replicas, yolo_bn_finalize adds them up. */
#define YOLO_BN_STAT_SLOTS 64
/* The backward reduction buffer `red` of yolo_bn_act_bwd_* holds YOLO_BN_RED_SLOTS per-workgroup partial
 * results of [2*C] doubles (every workgroup stores its own slot: no atomics, no zeroing needed) followed by
 * the final [2*C] sums: (YOLO_BN_RED_SLOTS + 1) * 2 * C doubles. */
#define YOLO_BN_RED_SLOTS 512
#define YOLO_SPLIT_BATCH_UNITS 4   /* 16-row blocks per workgroup of yolo_split_planes_batch */

/* y = conv(x, w) (+ bias if bias != NULL). If stats != NULL (double[YOLO_BN_STAT_SLOTS][2*Cout],
 * zeroed by the caller) the epilogue also accumulates per-channel sum / sum-of-squares of y for
 * training-mode BatchNormalization (yolov3/models/backbone.py:54). */
int yolo_conv2d_fwd(const yolo_conv_desc* d, const float* x, const float* w,
                    const float* bias, float* y, double* stats, void* stream);
/* the same, and if absmax != NULL (uint32[Cout], zeroed by the caller; needs stats) the epilogue also leaves
 * the bit pattern of the per-channel max|y| there: the data-derived bound the "planes" scales want
 * (yolo_bn_finalize_bound) */
int yolo_conv2d_fwd_absmax(const yolo_conv_desc* d, const float* x, const float* w, const float* bias,
                           float* y, double* stats, unsigned* absmax, void* stream);

/* dx (+)= conv_transpose(dy, w).  wT is the filter re-laid as [Cin][kh][kw][Cout]
 * (yolo_filter_transpose).  accumulate != 0 adds into dx (fan-out of a tensor). */
int yolo_conv2d_dgrad(const yolo_conv_desc* d, const float* dy, const float* wT,
                      float* dx, int accumulate, void* stream);

/* dw += sum over pixels dy (x) x  (dw must be zeroed by the caller before the first
 * contribution; split-K partial sums are combined with fp32 atomics).
 * dbias (optional, [Cout]) += sum over pixels of dy. */
int yolo_conv2d_wgrad(const yolo_conv_desc* d, const float* x, const float* dy,
                      float* dw, float* dbias, void* stream);

/* dbias[co] += sum over the P = N*Ho*Wo pixels of dy[p][co] (the bias half of yolo_conv2d_wgrad) */
int yolo_conv2d_wgrad_bias(const float* dy, long long P, int Cout, float* dbias, void* stream);

/* wT[ci][r][s][co] = w[co][r][s][ci] */
int yolo_filter_transpose(const float* w, float* wT, int Cout, int taps, int Cin, void* stream);

/* "Planes" operands: a row-major fp32 matrix X[rows][C] (the pixels of an NHWC tensor, or the output
 * channels of a [Cout][kh*kw*Cin] filter), C % 16 == 0, stored as TWO fp16 planes h + l of s*X (s = one
 * power of two per tensor, kept in the buffer's trailing header; |s*x - h - l| <= 2^-24 |s*x|) and blocked 16
 * rows x 16 channels so that the conv kernels can bring MFMA operand fragments into LDS by DMA (format and
 * error analysis: tf2_yolo_amd/csrc/planes.hpp). A product is three fp16 MFMA passes (hh + hl + lh, each exact
 * in the fp32 accumulator): ~3 fp32 roundings per product, i.e. fp32-grade, at half the matrix work of the
 * exact bf16 x 6 scheme of yolo_conv2d_fwd. yolo_planes_bytes gives the buffer size (0 for unsupported
 * shapes); yolo_split_planes fills it from fp32 (max|x| pass + split pass). A tensor that feeds several
 * convolutions (forward, filter gradient) is split once. */
size_t yolo_planes_bytes(long long rows, int C);
int yolo_split_planes(const float* x, long long rows, int C, void* planes, void* stream);
/* Same for a dense [rows][C_src] source whose channel count is not a multiple of 16 (the 3*(5+classes) = 255
 * channels of a YOLOv3 head, yolov3/models/__init__.py: the gradient of the head conv's output): planes of
 * [rows][C], C = C_src rounded up to 16, the columns past C_src are zero. The conv kernels then run with a
 * descriptor of Cout = C and filters padded with zero rows. */
int yolo_split_planes_padded(const float* x, long long rows, int C_src, int C, void* planes, void* stream);

/* Batched forms for the filters of a whole network: ONE launch instead of one per layer. `jobs` is a device
 * array of njobs records of six int64: {source pointer, destination pointer, a, b, c, first_block};
 * first_block = number of 256-thread workgroups of all earlier jobs, total_blocks = their sum.
 * split job:     a = rows, b = C, c = 0, or the address of the HEADER (buffer + yolo_planes_bytes - 256) of an
 *                already split planes buffer holding the same values in another order (a filter and its
 *                transpose): its bound is reused and the max|x| pass skipped;
 *                workgroups = ceil((ceil(rows/16)+1) / YOLO_SPLIT_BATCH_UNITS) * ceil(C / 128)
 *                (a workgroup covers YOLO_SPLIT_BATCH_UNITS 16-row blocks x 128 channels)
 * transpose job: a = Cout, b = taps, c = Cin;  workgroups = ceil(Cin/32) * ceil(Cout/32) * taps */
int yolo_split_planes_batch(const void* jobs, int njobs, long long total_blocks, void* stream);
int yolo_filter_transpose_batch(const void* jobs, int njobs, long long total_blocks, void* stream);

/* yolo_conv2d_fwd / yolo_conv2d_dgrad on pre-split operands (same conv, same fused epilogue, same
 * fp32-accurate result): x_planes = planes of x viewed as [N*H*W][Cin], w_planes = planes of w viewed as
 * [Cout][kh*kw*Cin]; dy_planes = planes of dy [N*Ho*Wo][Cout], wT_planes = planes of wT [Cin][kh*kw*Cout].
 * Requires Cin % 16 == 0 and Cout >= 32 (dgrad: Cout % 16 == 0 and Cin >= 32). */
int yolo_conv2d_fwd_planes(const yolo_conv_desc* d, const void* x_planes, const void* w_planes,
                           const float* bias, float* y, double* stats, unsigned* absmax, void* stream);
int yolo_conv2d_dgrad_planes(const yolo_conv_desc* d, const void* dy_planes, const void* wT_planes,
                             float* dx, int accumulate, void* stream);
/* Inference form with the fused epilogue of SURVEY.md section 8b (the folded BatchNormalization + activation of
 * yolov3/models/backbone.py:39-55, yolov4/models/backbone.py:76-111 inside the convolution):
 *   y = act(scale[c] * (conv(x, w) + bias[c]) + shift[c]) (+ residual)        scale / shift from yolo_bn_fold_inference
 * epilogue = YOLO_EPI_NONE (plain conv, scale / shift ignored), _AFFINE (no activation), _AFFINE_LEAKY, _AFFINE_MISH;
 * residual (optional) is a tensor laid out like y (the Add of a residual block). absmax (optional, uint32[Cout], zeroed
 * by the caller) receives the bit patterns of the per-channel max|y| BEFORE the residual is added; yolo_split_planes_absmax
 * turns y into the planes the next convolution reads, taking its bound from that vector (+ *extra_bound, the bound of
 * the residual tensor) -- two launches per layer where the unfused path needs three (conv, bound, BN/activation).
 * The arithmetic is the unfused path's (same fp32 operations in the same order): the fp32 result y is bit-identical. */
enum { YOLO_EPI_NONE = 0, YOLO_EPI_AFFINE = 1, YOLO_EPI_AFFINE_LEAKY = 2, YOLO_EPI_AFFINE_MISH = 3 };
int yolo_conv2d_fwd_planes_epi(const yolo_conv_desc* d, const void* x_planes, const void* w_planes, const float* bias,
                               int epilogue, const float* scale, const float* shift, const float* residual, float* y,
                               unsigned* absmax, void* stream);
int yolo_split_planes_absmax(const float* x, long long rows, int C, const unsigned* absmax, const float* extra_bound,
                             void* planes, float* out_bound, void* stream);
/* The same unit (Conv2D + folded BatchNormalization + activation (+ Add), yolov3/models/backbone.py:39-55,
 * yolov4/models/backbone.py:76-111) with the planes of its result produced by the same launches wherever the library splits
 * the contraction (small batches: the bs-1 forward of Yolo.predict, every 3x3 / 1x1 layer from 104x104 down): the kernel
 * that adds the split-K parts writes y AND its planes. The planes' power-of-two scale must be known while the values are
 * written, so it comes from an a-priori bound instead of max|y|:
 *     |y| <= K * max|x| + D (+ max|residual|),   {K, D} = pred[0..1] from yolo_conv_pred_bound (per layer, from the
 *     filter's l1 norms and the folded scale / shift: computed once per set of weights),
 *     max|x|, max|residual| = the bounds the producers of those tensors recorded, each given as n words whose maximum is
 *     the bound: n = 1 (one float) or the n words another such unit left behind (below), n <= YOLO_INFER_BOUND_WORDS.
 * The result's own max|y| goes to out_words as ONE word per workgroup of the launch that finished the tiles (plain
 * stores -- a shared word would queue every workgroup's device-scope atomic on one address); *out_n_host (a HOST int) is
 * set to their number. Launches whose tiles fill the chip keep the two-pass form internally (yolo_conv2d_fwd_planes_epi +
 * yolo_split_planes_absmax: *out_bound = max_c absmax[c] + the residual's bound) and set *out_n_host = 0.
 * yolo_fold_bound turns words into one float for consumers that want one. y is bit-identical to
 * yolo_conv2d_fwd_planes_epi's; the planes differ from the two-pass ones only in their (looser) scale. */
#define YOLO_INFER_BOUND_WORDS 4096
int yolo_conv_pred_bound(const float* w, int Cout, int kdim, const float* scale, const float* shift, const float* bias,
                         float* pred2, void* stream);
int yolo_conv2d_fwd_infer_unit(const yolo_conv_desc* d, const void* x_planes, const void* w_planes, const float* bias,
                               int epilogue, const float* scale, const float* shift, const float* residual, float* y,
                               unsigned* absmax, const float* pred2, const void* in_bound, int in_n,
                               const void* residual_bound, int residual_n, void* out_planes, unsigned* out_words,
                               float* out_bound, int* out_n_host, void* stream);
int yolo_fold_bound(const void* words, int n, float* out_bound, void* stream);
/* The first unit of every Darknet body -- Conv2D(32, 3, padding "same") + BatchNormalization + LeakyReLU / Mish on the RGB
 * image (yolov3/models/backbone.py:60, yolov4/models/backbone.py:127, yolov2/models/backbone.py:44) -- as ONE inference launch:
 * the direct fp32 stem kernel applies the folded scale / shift and the activation to a pixel's 32 sums in registers and writes
 * the planes of the next convolution (scale from K * max|image| + D, pred2 of yolo_conv_pred_bound; max|image| as the words of
 * yolo_absmax_words: one per workgroup, *n_words_host of them, no atomics). wt = the filter prepared by yolo_stem_filter_prep
 * ([27 taps + bias row][32], once per set of weights); y (fp32, optional) only if somebody reads it. out_words / *out_n_host as
 * in yolo_conv2d_fwd_infer_unit. */
int yolo_absmax_words(const float* x, long long n, unsigned* words, int* n_words_host, void* stream);
int yolo_stem_filter_prep(const float* w, const float* bias, float* wt, void* stream);
int yolo_stem_fwd_infer_unit(const yolo_conv_desc* d, const float* x, const float* wt, int epilogue, const float* scale,
                             const float* shift, const float* pred2, const void* in_bound, int in_n, float* y,
                             void* out_planes, unsigned* out_words, int* out_n_host, void* stream);
/* Concatenate (keras.layers.Concatenate on channels: yolov3/models/darknet.py:88,93; the CSP / SPP / PAN concats of
 * yolov4/models/backbone.py:141,183, yolov4/models/darknet.py:97-127; the passthrough of yolov2/models/darknet.py:49) straight
 * into the planes of the result: up to four dense fp32 sources [rows][channels_host[i]] (multiples of 8; the sum a multiple
 * of 16), each with ONE device float bounds_host[i] >= max|source| (what its producer recorded), become planes [rows][sum]
 * in one pass; the bound of the result (the largest source bound) goes to *out_bound (optional). dst32 (optional) also
 * receives the fp32 concatenation. The three *_host arguments are HOST arrays of nsrc entries (device pointers inside). */
int yolo_split_planes_concat(const float* const* srcs_host, const int* channels_host, const float* const* bounds_host,
                             int nsrc, long long rows, void* planes, float* dst32, float* out_bound, void* stream);
/* The same with two things a bs-1 graph would otherwise pay a launch each for (round 6: no launch of a replayed graph costs
 * less than ~4.5 us): bound_words_host[i] (HOST ints, NULL = all 1) = how many words bounds_host[i] points to, their maximum
 * being the source's bound -- 1 = one float, more = the words a one-pass inference unit left (yolo_conv2d_fwd_infer_unit);
 * upsample_host[i] != 0 (HOST ints, NULL = none) = source i is [N][H/2][W/2][channels] and is read through
 * UpSampling2D(2) (nearest: pixel (y, x) of the result takes pixel (y/2, x/2); yolov3/models/darknet.py:87,92), the
 * result's rows being N x H x W pixels. */
int yolo_split_planes_concat_ex(const float* const* srcs_host, const int* channels_host, const float* const* bounds_host,
                                const int* bound_words_host, const int* upsample_host, int H, int W, int nsrc,
                                long long rows, void* planes, float* dst32, float* out_bound, void* stream);
/* The detection head of YOLOv2 / v3 / v4 as one call: t = Conv2D(A (5 + C), 1) of x (bias, no BatchNormalization;
 * yolov3/models/__init__.py:34-64, yolov4/models/__init__.py:35-65, yolov2/models/darknet.py:95-104) and y = the head's
 * activation of t (yolo_head_act_fwd). Operands as planes (yolo_conv2d_fwd_planes). Launches with few output pixels (bs-1
 * predict) and version v3 / v4 take ONE launch (csrc/conv_small.hip: the activation in the convolution's epilogue); anything
 * else is yolo_conv2d_fwd_planes + yolo_head_act_fwd behind this entry. t and y are both written. */
int yolo_conv2d_fwd_head_unit(const yolo_conv_desc* d, const void* x_planes, const void* w_planes, const float* bias,
                              int A, int C, int version, const float* anchors, float* t, float* y, void* stream);
/* yolo_conv2d_wgrad on pre-split operands (dw += ..., same contract; the bias gradient stays with
 * yolo_conv2d_wgrad_bias on the fp32 dy). Requires Cin % 16 == 0, Cout % 16 == 0, Cout >= 32, kh*kw*Cin >= 64.
 * Two kernels behind it, same results to fp32 summation order: 3x3 stride-1 'same' layers with Cout % 128 == 0 and
 * Cin % 32 == 0 (rows of 4 .. 215 pixels) stream x ONCE through a ring of pixel slots in LDS and take all nine taps from it
 * (csrc/conv_wgrad_win.hip; yolo_set_option key 6); everything else takes one tap per tile (csrc/conv_wgrad_planes.hip).
 * Reference op: TensorFlow's autodiff of Conv2D, yolov3/models/backbone.py:27-36. */
int yolo_conv2d_wgrad_planes(const yolo_conv_desc* d, const void* x_planes, const void* dy_planes,
                             float* dw, void* stream);

/* Reproducible filter / bias gradients (no reference counterpart; SURVEY.md section 5 asks for atomics-free reductions where
 * determinism matters). With a workspace registered, yolo_conv2d_wgrad_planes stores every workgroup's partial tile to a
 * slab and a second kernel adds the splits of the pixel contraction IN ORDER into dw; yolo_conv2d_wgrad_bias does the same
 * with its per-block partial sums: bit-identical results from run to run. Without one (p == NULL) both combine their
 * partials with fp32 atomics (last bits depend on the arrival order). The workspace is caller-owned, yolo_wgrad_workspace_bytes()
 * is the size that never shortens a launch's split; launches that use it must be ordered on ONE stream. YOLO_WGRAD_DETERMINISTIC=0
 * in the environment keeps the atomics even with a workspace (A/B timing). */
size_t yolo_wgrad_workspace_bytes(void);
int yolo_set_wgrad_workspace(void* p, size_t bytes);

/* ------------------------------------------------------------------------------------
 * BatchNormalization (training and inference) + activation (+ residual add)
 * (replaces BatchNormalization + LeakyReLU / Mish + Add at
 *  yolov3/models/backbone.py:39-71, yolov4/models/backbone.py:76-123).
 * eps = 1e-3, momentum = 0.99 are the Keras defaults used by the reference.
 * ------------------------------------------------------------------------------------ */

/* per-channel sum / sum-of-squares of x[P,C] spread over the replicas of stats
 * (double[YOLO_BN_STAT_SLOTS][2*C], caller-zeroed). Only needed when the producing conv did not
 * fuse the statistics. */
int yolo_bn_stats(const float* x, long long P, int C, double* stats, void* stream);

/* From stats: mean, biased var; writes scale = gamma/sqrt(var+eps), shift = beta - mean*scale,
 * saves mean / invstd for backward, updates the moving statistics
 *   moving = momentum*moving + (1-momentum)*batch   (variance fed: biased if
 *   unbiased_moving_var == 0, Bessel-corrected otherwise)
 * `stats` is read (all replicas summed), not modified. */
int yolo_bn_finalize(double* stats, long long P, int C, const float* gamma, const float* beta,
                     float eps, float momentum, int unbiased_moving_var,
                     float* moving_mean, float* moving_var,
                     float* scale, float* shift, float* save_mean, float* save_invstd,
                     void* stream);

/* inference: scale/shift from the moving statistics */
int yolo_bn_fold_inference(int C, const float* gamma, const float* beta,
                           const float* moving_mean, const float* moving_var, float eps,
                           float* scale, float* shift, void* stream);

/* out = act(scale*x + shift) (+ residual if residual != NULL); x, out: [P,C] */
int yolo_bn_act_fwd(const float* x, long long P, int C, const float* scale, const float* shift,
                    int act, const float* residual, float* out, void* stream);

/* Backward of out = act(BN_train(x)) given dout = dL/dout (the residual branch, if any,
 * receives dout unchanged and is handled by the caller).
 *   pass 1 (reduce): dgamma/dbeta partial sums -> red (double[(YOLO_BN_RED_SLOTS+1)*2*C]:
 *                    SLOTS atomic replicas followed by their sum)
 *   pass 2 (apply) : dx = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)); dgamma, dbeta
 *                    are ACCUMULATED (+=) into the flat gradient buffer.
 * dx may alias dout. */
int yolo_bn_act_bwd_reduce(const float* x, const float* dout, long long P, int C,
                           const float* scale, const float* shift,
                           const float* save_mean, const float* save_invstd,
                           int act, double* red, void* stream);
int yolo_bn_act_bwd_apply(const float* x, const float* dout, long long P, int C,
                          const float* gamma, const float* scale, const float* shift,
                          const float* save_mean, const float* save_invstd,
                          int act, double* red, float* dgamma, float* dbeta,
                          float* dx, void* stream);

/* yolo_bn_act_fwd / yolo_bn_act_bwd_apply that ALSO emit their result in the conv kernels' "planes"
 * operand format (yolo_planes_bytes(P, C) bytes, C % 16 == 0), so that the consumer convolutions need no
 * yolo_split_planes pass. planes == NULL: plain forms. The planes need an upper bound of max|result| for
 * their power-of-two scale; it comes from the statistics, not from a pass over the data:
 *  - forward: yolo_bn_finalize_bound leaves in *bound (one uint32, zeroed by the caller before) the bit
 *    pattern of max_c |scale_c| (max|x_c| + |mean_c|) + |beta_c| >= max|act(BN(x))| when the conv epilogue's
 *    per-channel absmax is given, else of max_c |gamma_c| sqrt(P var_c / (var_c + eps)) + |beta_c| (valid because
 *    (x_i - mean)^2 <= P var, but looser by up to sqrt(P)); yolo_bn_act_fwd_planes adds *residual_bound (bound
 *    of the residual tensor) and stores the sum in *out_bound (optional);
 *  - backward: yolo_bn_act_bwd_reduce_bound leaves in bound_aux[0..2] max|dz|, max_c|scale_c| and
 *    max_c |scale_c| (|mean(dz xhat)_c| sqrt(P) + |mean(dz)_c|), which bound dx; bound_aux holds 68 uint32,
 *    all zeroed by the caller ([3] unused, [4..67] are replica slots of max|dz|).
 * yolo_bn_act_fwd_planes: out may be NULL, yolo_bn_act_bwd_apply_planes: dx may be NULL, when only the planes
 * are wanted. */
int yolo_bn_finalize_bound(double* stats, long long P, int C, const float* gamma, const float* beta, float eps,
                           float momentum, int unbiased_moving_var, float* moving_mean, float* moving_var,
                           float* scale, float* shift, float* save_mean, float* save_invstd,
                           const unsigned* absmax, unsigned* bound, void* stream);
/* The same with `mean_offset` [C] (optional): the statistics in `stats` are those of y - mean_offset. Keras puts a bias on the
 * convolution in front of BatchNormalization in YOLOv1.5 / v2 (yolov2/models/backbone.py:11-18, Conv2D's use_bias default); in
 * training mode that bias cancels in (y - mean) exactly, so the executor leaves it out of the convolution -- the statistics
 * then carry no bias-sized mean (var = E[y^2] - mean^2 would lose mean^2 / var of its digits) -- and passes it here, where
 * only the moving mean needs it: moving_mean <- momentum * moving_mean + (1 - momentum) * (mean + mean_offset).
 * scale / shift / save_mean stay those of the offset-free tensor the convolution wrote. */
int yolo_bn_finalize_offset(double* stats, long long P, int C, const float* gamma, const float* beta, float eps,
                            float momentum, int unbiased_moving_var, float* moving_mean, float* moving_var, float* scale,
                            float* shift, float* save_mean, float* save_invstd, const unsigned* absmax, unsigned* bound,
                            const float* mean_offset, void* stream);
/* inference (folded scale / shift, no batch statistics): *bound = bit pattern of
 * max_c |scale_c| max|x_c| + |shift_c| from the conv epilogue's per-channel absmax */
int yolo_bn_infer_bound(int C, const float* scale, const float* shift, const unsigned* absmax,
                        unsigned* bound, void* stream);
int yolo_bn_act_fwd_planes(const float* x, long long P, int C, const float* scale, const float* shift,
                           int act, const float* residual, float* out, void* planes,
                           const unsigned* bn_bound, const float* residual_bound, float* out_bound,
                           void* stream);
/* BatchNormalization + activation + MaxPooling2D(2, 2) in ONE pass over the conv output y [N, 2 Ho, 2 Wo, C] (round 6; the
 * conv + BN + LeakyReLU + pool units of Darknet-19, yolov2/models/backbone.py:42-60, and of tiny-YOLOv3,
 * yolov3/models/darknet.py:107-135): out [N, Ho, Wo, C] fp32 (may be NULL when only the planes are wanted) and / or the
 * planes of the pooled tensor (C % 16 == 0; scaled from *bn_bound, the bound yolo_bn_finalize_bound left for the UNPOOLED
 * activation -- a maximum of four values cannot exceed it), argmax [N, Ho, Wo, C] = the winner's flat offset into the
 * activation tensor [N, 2 Ho, 2 Wo, C] (first maximum in row-major window order, as yolo_maxpool_fwd), which is never
 * written: BatchNormalization's backward reads y, yolo_maxpool2x2_bwd only argmax. C % 8 == 0. */
int yolo_bn_act_maxpool2x2_fwd(const float* y, int N, int Ho, int Wo, int C, const float* scale, const float* shift,
                               int act, float* out, int* argmax, void* planes, const unsigned* bn_bound,
                               float* out_bound, void* stream);
/* yolo_bn_act_fwd_planes with the residual given AS PLANES (the operand format its producer wrote for the convolutions:
 * h + l = the value to 22-23 bits, the bound in the planes header): the fp32 copy of a residual block's input then need
 * not exist. C % 16 == 0. */
int yolo_bn_act_fwd_res_planes(const float* x, long long P, int C, const float* scale, const float* shift, int act,
                               const void* residual_planes, float* out, void* planes, const unsigned* bn_bound,
                               float* out_bound, void* stream);
int yolo_bn_act_bwd_reduce_bound(const float* x, const float* dout, long long P, int C, const float* scale,
                                 const float* shift, const float* save_mean, const float* save_invstd,
                                 int act, double* red, unsigned* bound_aux, void* stream);
/* Backward of the stem unit -- Conv2D(32, 3x3, 'same') on the 3-channel image + BatchNormalization + activation
 * (yolov3/models/backbone.py:60, yolov4/models/backbone.py:127, yolov2/models/backbone.py:44) -- behind
 * yolo_bn_act_bwd_reduce(_bound): the backward apply of yolo_bn_act_bwd_apply and the filter gradient of
 * yolo_conv2d_wgrad in ONE pass over (y, dout); the 32-channel gradient tensor (709 MB at bs 32) is never written.
 * dw [32][27] and dgamma / dbeta are accumulated (+=); scratch: yolo_stem_bwd_scratch_bytes() bytes of device memory. */
size_t yolo_stem_bwd_scratch_bytes(void);
int yolo_stem_bn_bwd_wgrad(const yolo_conv_desc* d, const float* y, const float* dout, const float* image, const float* scale,
                           const float* shift, const float* save_mean, const float* save_invstd, int act, const double* red,
                           float* dgamma, float* dbeta, float* dw, void* scratch, size_t scratch_bytes, void* stream);
int yolo_bn_act_bwd_apply_planes(const float* x, const float* dout, long long P, int C, const float* gamma,
                                 const float* scale, const float* shift, const float* save_mean,
                                 const float* save_invstd, int act, double* red, float* dgamma, float* dbeta,
                                 float* dx, void* planes, const unsigned* bound_aux, void* stream);
/* The same two passes with dout given as a CHANNEL SLICE of a wider tensor: row p of dout starts at dout + p * ld_dout
 * (floats; ld_dout >= C, a multiple of 4, dout 16-byte aligned; the apply pass needs C % 8 == 0 for a pitch other than C).
 * This is how the gradient of a Concatenate (yolov3/models/darknet.py:88,93; the CSP / SPP / PAN concats of
 * yolov4/models/backbone.py:141,183 and yolov4/models/darknet.py:97-127) reaches the BatchNormalization backward of its
 * sources without being copied out slice by slice first (TensorFlow's autodiff of Concatenate is a split: views too). */
int yolo_bn_act_bwd_reduce_bound_ld(const float* x, const float* dout, long long ld_dout, long long P, int C,
                                    const float* scale, const float* shift, const float* save_mean,
                                    const float* save_invstd, int act, double* red, unsigned* bound_aux, void* stream);
/* yolo_bn_act_bwd_reduce_bound_ld in ONE launch (round 6): the workgroups that arrive last fold the per-workgroup slots
 * themselves, in slot order (bit-reproducible; two levels of ticket words, agent-scope relaxed atomics instead of cache-wide
 * fences: csrc/bn_act.hip), instead of a second launch doing it -- 72 launches fewer in a YOLOv3 backward pass
 * (yolov3/models/backbone.py:27-55: one BatchNormalization per DarknetConv2D_BN_Leaky). Measured neutral on the step
 * (DESIGN.md section 3.4): the executor keeps the two-launch form unless YOLO_BN_FOLD=1. tickets:
 * YOLO_BN_FOLD_TICKET_WORDS uint32, zeroed by the caller before the FIRST use (the kernel leaves them zero); NULL, or more
 * than 1024 channels: the two-launch form. */
#define YOLO_BN_FOLD_TICKET_WORDS 36
int yolo_bn_act_bwd_reduce_fold_ld(const float* x, const float* dout, long long ld_dout, long long P, int C,
                                   const float* scale, const float* shift, const float* save_mean,
                                   const float* save_invstd, int act, double* red, unsigned* bound_aux,
                                   unsigned* tickets, void* stream);
int yolo_bn_act_bwd_apply_planes_ld(const float* x, const float* dout, long long ld_dout, long long P, int C,
                                    const float* gamma, const float* scale, const float* shift, const float* save_mean,
                                    const float* save_invstd, int act, double* red, float* dgamma, float* dbeta,
                                    float* dx, void* planes, const unsigned* bound_aux, void* stream);

/* The reduction of yolo_bn_act_bwd_reduce_bound FUSED INTO THE DATA GRADIENT THAT COMPLETES dout (round 5). In a training
 * step the tensor dL/d(a), a = act(BatchNormalization(y)) (yolov3/models/backbone.py:52-55: DarknetConv2D_BN_Leaky;
 * yolov4/models/backbone.py:76-111), is finished by the data gradient of a's LAST consumer -- the 3x3 convolution behind a
 * residual block's 1x1 (plain form) or the next block's 1x1 adding into the gradient its Add already holds (accumulate
 * form, yolov3/models/backbone.py:58-71: resblock_body). yolo_conv2d_dgrad_planes_bnred is yolo_conv2d_dgrad_planes whose
 * epilogue, while it stores its tile of dx = dL/d(a), also reads the same tile of y (laid out like dx) and leaves the tile's
 * per-channel sums of dz = dx * act'(scale * y + shift) and dz * xhat in ITS OWN slot of `partials`
 * ([slot][2][Cin] floats; plain stores, fixed order: bit-reproducible like the standalone pass), and max|dz| in bound_aux.
 * The standalone pass reads dx and y again from HBM (8 B per element); the fused form reads y once (4 B) beside stores the
 * kernel makes anyway. *nslots receives the number of slots written (all of them are); slots_cap = room in `partials`
 * (yolo_bnred_slots_cap gives a sufficient value). Stride 1, or stride 2 with the parity classes in one launch.
 * yolo_bn_act_bwd_sum_partials then folds the slots in order into red's final sums (fp64) and completes bound_aux exactly
 * as yolo_bn_act_bwd_reduce_bound does, after which yolo_bn_act_bwd_apply_planes(_ld) runs unchanged. */
int yolo_bnred_slots_cap(const yolo_conv_desc* d);
int yolo_conv2d_dgrad_planes_bnred(const yolo_conv_desc* d, const void* dy_planes, const void* wT_planes, float* dx,
                                   int accumulate, const float* y, const float* scale, const float* shift,
                                   const float* save_mean, const float* save_invstd, int act, float* partials,
                                   int slots_cap, unsigned* bound_aux, int* nslots, void* stream);
int yolo_bn_act_bwd_sum_partials(const float* partials, int nslots, long long P, int C, const float* scale, double* red,
                                 unsigned* bound_aux, void* stream);

/* plain activation (no BN) forward / backward on [n] elements; used by conv(+bias)+act
 * units without BN, if any */
int yolo_act_fwd(const float* x, long long n, int act, float* out, void* stream);
int yolo_act_bwd(const float* x, const float* dout, long long n, int act, float* dx, void* stream);

/* ------------------------------------------------------------------------------------
 * Glue ops (UpSampling2D, Concatenate, MaxPooling2D, space_to_depth, Add):
 *   yolov3/models/darknet.py:83-93, yolov4/models/backbone.py:176-185,
 *   yolov2/models/backbone.py:44-65, yolov2/models/darknet.py:46-49.
 * ------------------------------------------------------------------------------------ */
/* dst[:, c_off:c_off+Csrc] = src   (P pixels; src dense [P,Csrc], dst dense [P,Cdst]) */
int yolo_copy_channels_in(const float* src, long long P, int Csrc, float* dst, int Cdst,
                          int c_off, void* stream);
/* dst (+)= src[:, c_off:c_off+Cdst]   (src dense [P,Csrc], dst dense [P,Cdst]) */
int yolo_copy_channels_out(const float* src, long long P, int Csrc, int c_off, float* dst,
                           int Cdst, int accumulate, void* stream);
/* nearest 2x: y[n,2h+a,2w+b,c] = x[n,h,w,c]; written into channel slice of y ([.,.,.,Cy] at c_off) */
int yolo_upsample2x_fwd(const float* x, int N, int H, int W, int C, float* y, int Cy, int c_off,
                        void* stream);
/* dx (+)= sum of the 4 children of dy's channel slice */
int yolo_upsample2x_bwd(const float* dy, int N, int H, int W, int C, int Cy, int c_off, float* dx,
                        int accumulate, void* stream);
/* a (+)= b over n elements */
int yolo_axpy(float* a, const float* b, long long n, void* stream);
/* max-pool, window k, stride s, -inf padding pad_t/pad_l (Keras 'same'/'valid' resolved by host).
 * argmax (int32 flat input offset per output element, -1 if window empty) is saved for backward;
 * y is written into channel slice [c_off, c_off+C) of a [.,.,.,Cy] tensor. */
int yolo_maxpool_fwd(const float* x, int N, int H, int W, int C, int k, int s, int pad_t, int pad_l,
                     int Ho, int Wo, float* y, int Cy, int c_off, int* argmax, void* stream);
/* dx[argmax] += dy (dx caller-initialised) */
int yolo_maxpool_bwd(const float* dy, int N, int Ho, int Wo, int C, int Cy, int c_off,
                     const int* argmax, float* dx, void* stream);
/* The same backward for stride-1 'same' pools (YOLOv4's SPP block: MaxPooling2D(5 / 9 / 13, strides 1, padding "same"),
 * yolov4/models/backbone.py:175-185) as a gather without atomics: input pixel (h, w) adds, in a fixed order, the dy of the
 * outputs whose window holds it and whose saved winner it is; dx += that sum (bit-reproducible, where the scatter form's
 * result depends on the order of its atomicAdds). Shapes the gather kernel does not cover fall back to yolo_maxpool_bwd. */
int yolo_maxpool_bwd_same(const float* dy, int N, int H, int W, int C, int Cy, int c_off, const int* argmax, int k,
                          int pad_t, int pad_l, float* dx, void* stream);
/* Backward of a 2x2 / stride-2 pool whose windows tile the input exactly (input [N, 2 Ho, 2 Wo, C], C % 4 == 0; the five
 * MaxPooling2D of Darknet-19, yolov2/models/backbone.py:42-60, and of tiny-YOLOv3, yolov3/models/darknet.py:107-135): every
 * input position is written (accumulate = 0: dx needs no zero fill) or added to (accumulate = 1) exactly once -- dy where the
 * saved winner of its window is, 0 elsewhere. No atomics, 16 bytes per lane. */
int yolo_maxpool2x2_bwd(const float* dy, int N, int Ho, int Wo, int C, const int* argmax, float* dx, int accumulate,
                        void* stream);
/* tf.nn.space_to_depth(x, 2): y[n,h,w,(dy*2+dx)*C+c] = x[n,2h+dy,2w+dx,c], into slice of Cy at c_off */
int yolo_space_to_depth2_fwd(const float* x, int N, int H, int W, int C, float* y, int Cy, int c_off,
                             void* stream);
int yolo_space_to_depth2_bwd(const float* dy, int N, int H, int W, int C, int Cy, int c_off, float* dx,
                             int accumulate, void* stream);

/* ------------------------------------------------------------------------------------
 * Detection head activations (yolov3/models/__init__.py:40-65, yolov4/models/__init__.py:41-66,
 * yolov2/models/darknet.py:79-102, yolov1_5/models/darknet.py:37-55).
 * t: raw head conv output [P, A*(5+C)] (v1: [P, 5B+C]); y: activated prediction, same shape.
 *   v3/v4: xy=sigmoid, wh=exp(t)*anchor, conf=sigmoid, class=sigmoid
 *   v2   : xy=sigmoid, wh=exp(t)*anchor, conf=sigmoid, class=softmax
 *   v1   : first 5B channels sigmoid, last C channels softmax
 * anchors: device float[A*2] (w,h) (ignored for v1).
 * ------------------------------------------------------------------------------------ */
int yolo_head_act_fwd(const float* t, long long P, int A, int C, int version,
                      const float* anchors, float* y, void* stream);
/* dt = dy * dy/dt ; danchors (optional, v4 trainable Anchor layer) += sum dy*exp(t) */
int yolo_head_act_bwd(const float* y, const float* dy, long long P, int A, int C, int version,
                      const float* anchors, float* dt, float* danchors, void* stream);

/* ------------------------------------------------------------------------------------
 * Anchor-grid losses, forward + backward in one pass over the cells
 * (wrap_yolo_loss: yolov3/losses/loss.py:40-164, yolov4/losses/loss.py:64-169,
 *  yolov2/losses/loss.py:40-137, yolov1_5/losses/loss.py:40-118).
 * y_true: [N,gh,gw,5+C]; y_pred: [N,gh,gw,A*(5+C)] (v1: [N,gh,gw,5B+C]) -- POST-activation,
 * exactly what the reference closure receives.
 * loss_out: device double[8]: [0]=total loss, [1..] = weighted-free parts
 *   v2/v3: xy, wh, conf_obj, conf_noobj, class, reg ; v4: box, conf_obj, conf_noobj, class, reg
 * dpred (optional): dL/dy_pred (same shape as y_pred), multiplied by grad_scale.
 * workspace: yolo_loss_workspace_bytes(cells) bytes, device.
 * ------------------------------------------------------------------------------------ */
typedef struct yolo_loss_cfg {
  int version;            /* 1,2,3,4                                                  */
  int N, gh, gw, A, C;    /* batch, grid, anchors (v1: boxes B), classes              */
  float anchors[32];      /* A pairs (w,h); ignored by v1                             */
  int   use_anchors;      /* 0 => panchors = 1 (yolov3/losses/loss.py:52-53)          */
  float binary_weight;
  float loss_weight[4];   /* v1-v3: xy,wh,conf,prob ; v4: box,conf,prob               */
  float ignore_thresh;
  int   use_focal_loss;   /* v3 only                                                  */
  float focal_gamma;      /* v3, v4                                                   */
  int   use_scale;        /* v3 (v2: always on)                                       */
  float wh_reg_weight;    /* v4 (v2/v3 fixed 0.01)                                    */
  float truth_thresh;     /* v4                                                       */
  float label_smooth;     /* v4                                                       */
} yolo_loss_cfg;

/* workspace: the fused kernel needs none (yolo_loss_workspace_bytes returns 0). OPTIONAL diagnostic output: a workspace of at
 * least 8 bytes per grid cell (N*gh*gw cells, A <= 16) receives two ints per cell -- [0] the responsible anchor (first maximal
 * IoU, tf.argmax), [1] bit b = "IoU of anchor b < ignore_thresh", bit 16 + b = "IoU of anchor b > truth_thresh" -- the
 * discrete decisions of yolov3/losses/loss.py:63-79 as THIS execution took them (parity tests hand them to the oracle: two
 * executions legitimately disagree about near-tied anchors). */
size_t yolo_loss_workspace_bytes(const yolo_loss_cfg* cfg);
int yolo_loss_fwd_bwd(const yolo_loss_cfg* cfg, const float* y_true, const float* y_pred,
                      double* loss_out, float* dpred, float grad_scale,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Metrics (yolov3/metrics/yolo_metrics.py:9-115, yolov1_5/metrics/yolo_metrics.py:9-107).
 * out: device double[8]:
 *   [0] sum over cells of binary_accuracy(c_true, max_b c_pred)   (obj_acc numerator; / cells)
 *   [1] sum(max_b iou * obj)      [2] sum(obj)
 *   [3] sum([argmax p_t == argmax p_p] * obj)  (over anchors; v1: per cell)
 *   [4] #[max_b(iou*same_class*obj) >= recall_thresh]
 *   [5] cells */
int yolo_metrics(const yolo_loss_cfg* cfg, const float* y_true, const float* y_pred,
                 float recall_thresh, double* out, void* stream);

/* ------------------------------------------------------------------------------------
 * Optimizer (Keras Adam defaults: README.md:241; beta1=.9 beta2=.999 eps=1e-7, bias-corrected)
 *   g <- g*grad_scale ; m,v update ; p -= lr_t * m/(sqrt(v)+eps) ; g <- 0 (if zero_grad)
 * ------------------------------------------------------------------------------------ */
int yolo_adam_step(float* p, float* g, float* m, float* v, long long n, float lr, float beta1,
                   float beta2, float eps, int step, float grad_scale, int zero_grad, void* stream);
/* The same update with its five scalars read from DEVICE memory: hyper = {lr_t, beta1, beta2, eps, grad_scale}, lr_t = the
 * bias-corrected rate of this step, yolo_adam_lr_t(lr, beta1, beta2, step) (computed in double exactly as yolo_adam_step does).
 * This is the form a captured hipGraph of the training step replays: the host refreshes `hyper` before every replay. */
float yolo_adam_lr_t(float lr, float beta1, float beta2, int step);
int yolo_adam_step_dev(float* p, float* g, float* m, float* v, long long n, const float* hyper, int zero_grad, void* stream);
int yolo_sgd_step(float* p, float* g, long long n, float lr, float grad_scale, int zero_grad,
                  void* stream);
int yolo_fill(float* p, long long n, float value, void* stream);

/* ------------------------------------------------------------------------------------
 * decode + NMS (utils/tools.py:370-438, 630-786), bit-exact index selection.
 * ------------------------------------------------------------------------------------ */
/* One level: pred [gh,gw,A*(5+C)] float32 (version 2/3/4) or [gh,gw,5B+C] (version 1).
 * Appends rows (x,y,w,h,conf,class,prob) as float64 to rows_out[7*row] starting at row *count
 * (device int, caller-zeroed before the first level), in the reference's C order
 * (y, x, box, class). Rows beyond max_rows are counted but not written. */
int yolo_decode_level(const float* pred, int gh, int gw, int A, int C, int version,
                      float threshold, double* rows_out, int max_rows, int* count,
                      void* workspace, size_t workspace_bytes, void* stream);
/* same, for float64 predictions (the reference's decode also runs on float64 label tensors,
 * utils/tools.py:441-470 vis_img on ground truth): product and threshold in float64 */
int yolo_decode_level_f64(const double* pred, int gh, int gw, int A, int C, int version,
                          double threshold, double* rows_out, int max_rows, int* count,
                          void* workspace, size_t workspace_bytes, void* stream);
size_t yolo_decode_workspace_bytes(int gh, int gw, int A, int C);

/* NMS over n decoded rows (device float64 [n,7]); keep_out: device uint8[n] (1 = kept).
 * The caller gathers kept rows class by class in ascending class id, original order inside
 * a class (utils/tools.py:730-732).
 * Round 5: hard / DIoU NMS make every pair test of a class first (a bit matrix, one 64-bit word per row and 64 columns, by
 * the whole chip) and then walk the bits, one workgroup per class; classes of more than 8192 rows walk the boxes directly.
 * The workspace therefore includes room for the matrices: yolo_nms_workspace_bytes(n, class_num) grows by
 * n * (min(n, 8192) / 64 + 1) * 8 bytes (156 MB for 151 186 rows; 1 MB for 4 425). Results are bit-identical to the walk
 * (yolo_set_option(YOLO_OPT_NMS_WALK, 1) selects it for every class). */
size_t yolo_nms_workspace_bytes(int n, int class_num);
int yolo_nms(const double* rows, int n, int class_num, int mode, double nms_threshold,
             double conf_threshold, double sigma, unsigned char* keep_out,
             void* workspace, size_t workspace_bytes, void* stream);
/* yolo_nms followed by the gather the reference does on the host (utils/tools.py:730-732): rows_out [<= n][7] receives the
 * kept rows, classes ascending and in their original order inside a class, *count_out (device int) their number; rows whose
 * class id is outside [0, class_num) are dropped, as the reference's per-class loop drops them. keep_out as yolo_nms. */
int yolo_nms_select(const double* rows, int n, int class_num, int mode, double nms_threshold, double conf_threshold,
                    double sigma, unsigned char* keep_out, double* rows_out, int* count_out, void* workspace,
                    size_t workspace_bytes, void* stream);

/* cal_iou as a stand-alone broadcasting kernel -- the reference's two public functions of that name:
 *   utils/tools.py:630-684       cal_iou(xywh_true, xywh_pred, mode): NumPy float64/float32, mode 1 IoU, 2 DIoU
 *   yolov3/losses/loss.py:9-37   cal_iou(xywh_true, xywh_pred, grid_shape): x / grid_w, y / grid_h first (same text
 *                                in yolov1_5 / yolov2); yolov4/losses/loss.py:10-61 return_ciou=True -> mode 3:
 *                                out = IoU, out2 = CIoU
 * The output has `ndim` dimensions of extents shape_host[]; operand element e's box starts at
 * sum_d index_d * strides_host[d] (strides in ELEMENTS, 0 on a broadcast dimension), its x, y, w, h are 4 consecutive
 * elements. is_f64: 1 = double operands / outputs, 0 = float. Same operation order as the reference, no fused
 * multiply-adds: the float64 results are bit-identical to utils.tools.cal_iou's (tests/golden/tools_golden.npz). */
int yolo_cal_iou(const void* xywh_true, const void* xywh_pred, void* out, void* out2, int is_f64, int mode, int ndim,
                 const long long* shape_host, const long long* true_strides_host, const long long* pred_strides_host,
                 double grid_w, double grid_h, void* stream);

/* ------------------------------------------------------------------------------------
 * Detection evaluation after decode / NMS (replaces the arithmetic of create_score_mat and
 * PRfunc, utils/measurement.py:16-150, 153-337). Rows are the (n,7) float64 arrays of
 * yolo_decode_level: x, y, w, h, conf, class, class prob. All fp64 / integer, bit-exact.
 * ------------------------------------------------------------------------------------ */

/* One image. For every detection: best_gt = index (inside the subset of ground truths of the detection's
 * class, np.argmax = first maximum) of the ground truth with the highest IoU, matched = that IoU >=
 * iou_threshold, best_iou (0 / 0 / 0 when the class has no ground truth: measurement.py:275-277).
 * class_counts[c*4 + {0,1,2,3}] += #detections, #ground truths, #matched detections, #distinct matched
 * ground truths of class c (accumulates over images; zero it first). gt_flags: scratch of ngt bytes. */
int yolo_match_detections(const double* gt_rows, int ngt, const double* det_rows, int ndet, int class_num,
                          double iou_threshold, int* best_gt, unsigned char* matched, double* best_iou,
                          unsigned char* gt_flags, long long* class_counts, void* stream);

/* rank[i] = number of rows of the same segment that sort before row i when keys are ordered descending
 * (np.argsort(key)[::-1]; equal keys: the later row first). segment may be NULL (one segment). */
int yolo_rank_desc(const double* key, const int* segment, int n, int* rank, void* stream);

/* Precision / recall curve of ONE class (measurement.py:299-323): detections (joint confidence, id of the
 * best ground truth, matched flag) are sorted by confidence; point i (0 <= i < n) is computed from the i+1
 * best detections: num_tpp = #matched, num_tp = #distinct matched ground truths, precision = num_tpp/(i+1)
 * (mode 0), num_tp/(num_tp + i+1 - num_tpp) (mode 1), num_tp/(i+1) (mode 2), recall = num_tp/num_gts;
 * point n = (0, last recall). precision and recall hold n+1 doubles. Needs n > 0 and num_gts > 0. */
size_t yolo_pr_curve_workspace_bytes(int n, int num_gts);
int yolo_pr_curve(const double* joint, const int* gt_id, const unsigned char* matched, int n, int num_gts,
                  int precision_mode, void* workspace, size_t workspace_bytes, double* precision,
                  double* recall, void* stream);

/* ------------------------------------------------------------------------------------
 * Gradient exchange (data parallelism; no reference counterpart). `yolo_allreduce_bucket` is the thin RCCL wrapper
 * SURVEY.md section 8b lists (round 5; rounds 1-4 left it out): in-place sum all-reduce of `count` fp32 values of the
 * caller's flat gradient buffer on the caller's stream with the CALLER'S communicator (`rccl_comm` = an ncclComm_t). The
 * library does not link RCCL: ncclAllReduce is resolved in the process at the first call (the RCCL the host already
 * loaded, else librccl.so from the loader path). It is for hosts that own their communicators; the Python host of this
 * repository exchanges gradients through torch.distributed (backend "nccl" = RCCL over xGMI; tf2_yolo_amd/dp.py), which
 * does not hand out its communicator -- same collective, same buffer slices, same stream ordering. The library guarantees
 * that the parameter gradients of a unit are complete on the stream(s) it was given when its backward calls have been
 * enqueued; yolo_adam_step applies the 1/world factor (grad_scale).
 * ------------------------------------------------------------------------------------ */
int yolo_allreduce_bucket(void* rccl_comm, float* grads, long long count, void* stream);

/* ------------------------------------------------------------------------------------
 * Label tensors (the step in front of the loss): box -> grid encoder of the reference's data sequences
 * (utils/tools.py:179-209: x = box_x % cell_w / cell_w, w = box_w / img_w, conf = 1, class bit; a later box of the
 * same cell overwrites x, y, w, h, class bits accumulate) and the 2x label pyramid utils/tools.py:342-367
 * (per 2x2 block that holds an object: the box with the largest w*h, first maximum; its centre re-expressed in the
 * coarser cell), as yolov3/__init__.py:41-53 applies it. float64 arithmetic in the reference's operation order:
 * bit-identical to its float64 arrays; label32 / out32 (optional) receive the float32 cast Keras feeds the loss.
 * boxes: [nb][4] = x1, y1, x2, y2 in pixels; cls: [nb] class ids; first: [N+1] box range of every image.
 * ------------------------------------------------------------------------------------ */
int yolo_encode_labels(const double* boxes, const int* cls, const int* first, int N, double img_h, double img_w,
                       int gh, int gw, int C, double* label64, float* label32, void* stream);
int yolo_down2xlabel(const double* label_in, int N, int gh, int gw, int ch, double* label_out, float* out32,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YOLO_HIP_H_ */

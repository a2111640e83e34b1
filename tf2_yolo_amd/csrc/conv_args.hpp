// Shared argument block of the implicit-GEMM "gather" convolution kernels (conv.hip: exact-fp32 MFMA;
// conv_split.hip: fp32 emulated by bf16 x 6 MFMA passes). See conv.hip for the formulation.
#pragma once
#include "common.hpp"

namespace yolo {

constexpr int MAX_TAPS = 49;

struct Tap {
  int oy, ox, woff;
};

// One launch for the stride x stride parity classes of a strided data gradient (conv.hip: dgrad_impl): class c computes the
// output pixels (y * osy + ooy, x * osx + oox) of its own Hg x Wg grid from its own taps [tap0, tap0 + ntaps)
struct ClassGeom {
  long long M;
  int Hg, Wg, ooy, oox, tap0, ntaps;
};

struct GatherConvArgs {
  const float* src;
  const float* wgt;
  const float* bias;
  float* dst;
  double* stats;  // optional [YOLO_BN_STAT_SLOTS][2*Cout]: per-channel sum / sum of squares of dst
  unsigned* absmax;  // optional (with stats) [Cout]: bit patterns of the per-channel max |dst|
  long long M;  // N*Hg*Wg
  int N, Hs, Ws, Cs;
  int Hg, Wg;
  int sy, sx;
  int Hd, Wd, Cd;
  int osy, osx, ooy, oox;
  int Cout, ldw;
  int ntaps, accumulate;
  int kw, pad_t, pad_l;  // FLAT mode: tap t = (r*kw+s), oy = r-pad_t, ox = s-pad_l
  int tiles_n;
  int nblocks;
  // planes kernels (conv_planes.hip): src / wgt point to fp16 planes (planes.hpp); byte sizes and all-zero block indices
  unsigned src_bytes, wgt_bytes;
  int zero_blk_src, zero_blk_wgt;
  int nt_store;  // planes kernels: non-temporal stores of the output (it is not re-read by this kernel)
  int vec_store; // planes kernels, 128x128 tiles: output through LDS as dwordx4 rows (YOLO_VEC_STORE, default 1)
  int kc;        // planes kernels: 16-channel blocks per chunk of the stage order
  int dbg;       // planes kernels: diagnostic knock-outs (YOLO_PLANES_DBG), 0 in production
  // fused inference epilogue of the planes kernels (planes_epilogue.hpp): dst = act(epi_scale[c] * y + epi_shift[c])
  // (+ epi_res, a tensor laid out like dst); epi_scale == nullptr: plain y. absmax then holds max|dst| before the residual
  const float* epi_scale;
  const float* epi_shift;
  const float* epi_res;
  int epi_act;
  // inference unit in one pass (yolo_conv2d_fwd_infer_unit, split-K launches: conv_split_reduce_kernel): the finished
  // values also go out as the planes of the consumer convolutions. Their scale cannot wait for max|dst| (it is needed
  // while the tile is written), so it comes from an a-priori bound: pl_pred = {K, D} of this layer (yolo_conv_pred_bound:
  // |dst| <= K * max|src| + D before the residual), pl_in_bound / pl_res_bound = the recorded bounds of the input and of
  // the residual tensor, each given as pl_*_n words whose maximum is the bound (1 = a plain float; more = the words another
  // such unit left behind). pl_out_words receives max|dst| for the consumers as ONE word per workgroup of the reduce launch
  // (plain stores; at most YOLO_INFER_BOUND_WORDS): one shared word would take an atomic from every workgroup on the same
  // address, and device-scope atomics on one address queue up (+4 us on a 6 us kernel; 64 shared slots: +1.7 us).
  unsigned char* out_planes;
  const float* pl_pred;
  const unsigned* pl_in_bound;
  const unsigned* pl_res_bound;
  int pl_in_n, pl_res_n;
  unsigned* pl_out_words;
  // conv_small.hip, detection-head form (yolo_conv2d_fwd_head_unit): the 1x1 convolution of a YOLOv2/v3/v4 head (bias, no
  // BatchNorm, Cout = A (5 + C), any Cout) and the head's activation in one launch: head_y = sigmoid(t) except channels 2, 3
  // of every anchor's group: exp(t) * anchor (yolov3/models/__init__.py:58-67); dst (the raw t) is written only if non-null
  float* head_y;
  const float* head_anchors;
  int head_A, head_C;
  // conv_win.hip, stream-K form: workgroups of the launch (0 = one tile per workgroup), part slabs, tile tickets
  int sk_grid;
  int tile_order;   // conv_win.hip: 0 = column tile fastest inside an XCD's run, 1 = row tile fastest
  float* sk_slabs;
  unsigned* sk_tickets;
  // split-K for launches that would leave most of the chip idle (bs-1 inference): split_parts > 1 = every tile is
  // computed by that many workgroups (equal runs of 16-channel blocks), each writing its accumulators to slab
  // (tile * split_parts + part) of sk_slabs; conv_split_reduce_kernel adds the parts in order and runs the epilogue
  int split_parts;
  // ncls > 1 (planes kernels, per-tap form): the tile index is (row tile, class, column tile); the classes of a pixel
  // block sit next to each other in an XCD's run of tiles, so the rows of `src` they share come from HBM once
  int ncls;
  ClassGeom cls[4];
  unsigned long long* stamps;  // diagnostic builds of conv_win.hip: 8 x u64 per workgroup (s_memtime / s_memrealtime)
  // BatchNorm-backward reduction fused into the data gradient that COMPLETES dL/d(dst) (planes_epilogue.hpp): dst is the
  // output a = act(scale * y + shift) of a conv + BatchNormalization unit whose pre-BN tensor is bwd_y (laid out like dst).
  // The epilogue forms dz = dst * act'(scale * y + shift) from the FINAL value it stores (the accumulate form included) and
  // leaves the tile's per-channel sums of dz and dz * xhat in its own slot of bwd_part ([slot][2][Cout] floats, plain
  // stores: no atomics, fixed order) -- the pass yolo_bn_act_bwd_reduce would make over dst and y (8 B per element) becomes a
  // 4 B per element read of y beside the tile's stores. bwd_aux: the 68 bound words of yolo_bn_act_bwd_reduce_bound.
  const float* bwd_y;
  const float* bwd_scale;
  const float* bwd_shift;
  const float* bwd_mean;
  const float* bwd_invstd;
  float* bwd_part;
  unsigned* bwd_aux;
  int bwd_act;
  int bwd_nslots;   // set by the launcher: slots of bwd_part this launch writes (every one of them)
  int bwd_cap;      // slots bwd_part has room for
  Tap taps[MAX_TAPS];
};

#define YOLO_BNRED_CHECK(a)                                                                                          \
  if ((a).bwd_y != nullptr && (a).bwd_nslots > (a).bwd_cap) {                                                        \
    set_error("conv_dgrad(bn reduce): %d partial slots needed, room for %d", (a).bwd_nslots, (a).bwd_cap);           \
    return YOLO_ERR_INVALID_ARG;                                                                                     \
  }

// Blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous run of
// logical tiles so tiles that share A rows / B columns hit the same L2. Bijective for any
// grid size (cdna_hip_programming.md section 5, "XCD swizzle must be bijective").
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int xcd = bid & 7, idx = bid >> 3;
  const int q = nblocks >> 3, r = nblocks & 7;
  const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}


struct WgradArgs {
  const float* src;
  const float* dy;
  float* dw;
  long long M;  // N*Hg*Wg pixels
  int N, Hs, Ws, Cs;
  int Hg, Wg;
  int sy, sx;
  int Cout, ldw;
  int ntaps;
  int kw, pad_t, pad_l;
  int tiles_co, tiles_j;
  int nblocks;      // planes kernel: workgroups of the (1-D, XCD-remapped) grid = tiles_co * tiles_j * splits
  long long chunk;  // pixels per split (multiple of 32)
  // planes kernel (conv_wgrad_planes.hip): src / dy point to fp16 planes
  unsigned src_bytes, dy_bytes;
  int zero_blk_src, zero_blk_dy;
  // planes kernel, reproducible form (yolo_set_wgrad_workspace): every workgroup stores its BM x BN partial to slab
  // `split * tiles + tile` (accumulator order) and wgrad_reduce_kernel adds the splits IN ORDER into dw; nullptr = fp32 atomics
  float* slabs;
  int splits;
};

// conv_split.hip
int launch_gather_split(GatherConvArgs& a, hipStream_t st);
bool gather_split_supported(const GatherConvArgs& a);
// conv_planes.hip
long long planes_bytes(long long rows, int C);
int launch_split_planes(const float* x, long long rows, int C, void* planes, hipStream_t st);
int launch_split_planes_absmax(const float* x, long long rows, int C, const unsigned* absmax, const float* extra_bound, int extra_n,
                               void* planes, float* out_bound, hipStream_t st);
int launch_split_planes_concat(const float* const* xs, const int* Cs, const float* const* bounds, int nsrc, long long rows,
                               void* planes, float* dst32, float* out_bound, hipStream_t st,
                               const int* bound_words = nullptr, const int* upsample = nullptr, int H = 0, int W = 0);
int launch_split_planes_padded(const float* x, long long rows, int Csrc, int C, void* planes, hipStream_t st);
int launch_split_planes_batch(const void* jobs, int njobs, long long total_blocks, hipStream_t st);
int launch_filter_transpose_batch(const void* jobs, int njobs, long long total_blocks, hipStream_t st);
int launch_gather_planes(GatherConvArgs& a, hipStream_t st);
// conv_small.hip: inference units with few output pixels in one launch
bool conv_small_supported(const GatherConvArgs& a);
bool conv_small_head_supported(const GatherConvArgs& a);
int launch_conv_small(GatherConvArgs& a, hipStream_t st, int* nwg);
bool gather_planes_supported(const GatherConvArgs& a);
// conv_win.hip (3x3 stride-1 forward / data gradient with the input window kept in LDS); returns 1 = not covered
int conv_split_parts(const GatherConvArgs& a, long long nb, int bm, int min_cb, int idle_div);
int launch_split_reduce(GatherConvArgs& a, int bm, hipStream_t st);
float* conv_split_slabs();
int launch_conv_win(GatherConvArgs& a, int variant, hipStream_t st);
bool conv_win_supported(const GatherConvArgs& a);
// run-time options (yolo_set_option; defaults from the environment): see runtime.hip
enum { OPT_CONV_WIN = 0, OPT_STAMPS = 1, OPT_CONV_SK = 2, OPT_DBG = 3, OPT_TILE_ORDER = 4, OPT_CONV_PATCH = 5, OPT_WGRAD_WIN = 6, OPT_NMS_WALK = 7, OPT_EXP = 8, OPT_COUNT = 16 };
// OPT_EXP (YOLO_EXP): A/B switches of round 6, all with CORRECT results -- bit 8 = the chunk-ahead loader of the loss kernel
// instead of the cell-ahead one (loss.hip; scripts/loss_bench.py), bit 16 = forward launches with BatchNorm statistics never
// split (as until round 5; conv_win.hip: conv_split_parts). (Bits 1 / 2 / 4 were timing knock-outs of the small launches:
// without bn_bwd_sum the gradients become NaN within a step, and a NaN-filled network runs 12 % FASTER than a real one --
// the matrix pipes draw less power on constant operands -- so those timings measured nothing and the bits are gone.)
int set_conv_workspace(void* p, size_t bytes, hipStream_t st);   // conv_win.hip
extern void* g_dbg_buf;        // yolo_set_debug_buffer
extern size_t g_dbg_bytes;
extern int g_opt[OPT_COUNT];
void init_options();
// conv_wgrad_planes.hip
int launch_wgrad_planes(WgradArgs& a, hipStream_t st);
bool wgrad_planes_supported(const WgradArgs& a);
// conv_wgrad_win.hip (3x3 stride-1 filter gradient with the input window kept in LDS)
int launch_wgrad_win(WgradArgs& a, hipStream_t st);
bool wgrad_win_supported(const WgradArgs& a);
// workspace of the reproducible (atomics-free) filter / bias gradient reductions: [colsum partials: 1 MiB][slabs]
constexpr size_t WGRAD_WS_COLSUM_BYTES = 1 << 20;
void* wgrad_workspace(size_t* bytes);
// stem.hip (direct fp32 kernel for the 3-channel 3x3 stem)
bool stem_fwd_supported(const yolo_conv_desc* d);
// the stem's inference unit in one pass (stem.hip): prepared filter [28][32] (launch_stem_filter_prep), epilogue arguments
struct StemEpiArgs {
  const float* scale;
  const float* shift;
  int act;
  void* planes;
  const float* pred;
  const unsigned* in_words;
  int in_n;
  unsigned* out_words;
};
bool stem_infer_supported(const yolo_conv_desc* d);
int launch_stem_filter_prep(const float* w, const float* bias, float* wt, hipStream_t st);
int launch_absmax_words(const float* x, long long n, unsigned* words, int* n_words, hipStream_t st);
int launch_stem_infer(const yolo_conv_desc* d, const float* x, const float* wt, float* y, const StemEpiArgs& e, int* out_n,
                      hipStream_t st);
size_t stem_bwd_scratch_bytes();
int launch_stem_bn_bwd_wgrad(const yolo_conv_desc* d, const float* y, const float* dout, const float* img, const float* scale,
                             const float* shift, const float* smean, const float* sinv, int act, const double* redsum,
                             float* dgamma, float* dbeta, float* dw, float* scratch, size_t scratch_bytes, hipStream_t st);
int launch_stem_fwd(const yolo_conv_desc* d, const float* x, const float* w, const float* bias, float* y, double* stats,
                    unsigned* absmax, hipStream_t st);
// conv_wgrad_split.hip
int launch_wgrad_split(WgradArgs& a, hipStream_t st);
bool wgrad_split_supported(const WgradArgs& a);

}  // namespace yolo

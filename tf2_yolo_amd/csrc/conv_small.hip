// Inference units with FEW output pixels (bs-1 predict: 13x13 .. 104x104 maps) in ONE launch: conv (1x1, or 3x3 with any
// stride) + folded BatchNormalization + activation (+ residual Add) + the planes of the result.
//
// Why its own kernel. At batch 1 a YOLOv3-416 layer has 169 .. 2704 output pixels: 2 .. 22 row tiles of 128. The training
// kernels (conv_win.hip, conv_planes.hip) then run split-K -- every tile by up to 32 workgroups, each a latency-bound stream
// of 8 KB stages through an LDS ring -- and a second launch adds the parts and runs the epilogue: 9.6-16 us + 6.9 us per
// layer, 69 such pairs = 1.4 of the 1.5 ms of Model.predict (profiles/r06_e_c5_bs1_kernel_stats.csv). Nothing in such a
// layer is re-used inside a tile often enough to be worth a trip through LDS, and no launch of the replayed graph costs
// less than ~4.5 us however little it does: one launch per unit instead of two. So here:
//   * tile = 32 pixels x 32 filters, ONE v_mfma_f32_32x32x16_f16 accumulator per wave (a 13x13 1024->512 layer is 96 tiles),
//     or 64 x 64 with four (a 52x52 3x3 128->256 layer is 172) -- the chip is filled without splitting K across workgroups;
//   * K is split across the EIGHT WAVES of the workgroup instead (16-channel steps s = wave, wave + 8, ...): every wave
//     accumulates its own 32 x 32 partial, the partials are added in wave order through LDS (32 KB) -- bitwise reproducible;
//   * operands go global -> registers: the planes format (planes.hpp) was laid out so that the 16-byte unit a lane needs for
//     its MFMA fragment (row r, 8 consecutive k) is contiguous, and 16 consecutive rows are 256 contiguous bytes: one
//     global_load_dwordx4 per lane and plane, whole cache lines per wave, no LDS, no barrier in the main loop. Four steps
//     (16 loads per lane) are requested before the previous four are multiplied: 8 waves x 64 lanes x 16 x 16 B = 128 KB in
//     flight per workgroup;
//   * 3x3 taps: lane r's source row for tap t is its own pixel shifted by the tap (or the all-zero block of the planes when
//     the tap falls outside the image) -- nine per-lane byte offsets in an LDS table, one ds_read_b32 per step;
//   * the epilogue is conv_split_reduce_kernel's (conv_win.hip): unscale, bias, folded BN, activation, residual, fp32 store,
//     planes scaled from the a-priori bound (GatherConvArgs::pl_pred), one word of max|dst| per workgroup.
// Used by yolo_conv2d_fwd_infer_unit and yolo_conv2d_fwd_head_unit for the launches small_tile / small_head_tile (below)
// pick: a workgroup streams ~50-65 GB/s whatever its tile and however many loads it keeps in flight, so a launch pays when it
// fits ONE round of at most 256 workgroups with few bytes each -- the 1x1 units, the heads, the 52x52 / 104x104 3x3 units.
#include "act.hpp"
#include "planes.hpp"
#include <cstdlib>

namespace yolo {

constexpr int SM_WAVES = 8;   // waves per workgroup = K split

template <int TM, int TN, int CH>
struct SmallBuf {
  u32x4 ah[CH][TM], al[CH][TM], bh[CH][TN], bl[CH][TN];
};

// NT taps (1 or 9); tile = 32 TM pixels x 32 TN filters (TM x TN accumulators per wave: an operand fragment a wave has
// loaded is used TN / TM times -- (1,1) asks L2 for 4 KB per 32 x 32 x 16 block product, (2,2) for 2 KB); CH steps per
// register buffer (two buffers)
// HEAD: the detection-head form (GatherConvArgs::head_y): any Cout (columns past it are masked), bias only, the head's
// activation instead of BatchNorm / activation / residual / planes
// Epilogue: ALL blocks' partials go to LDS at once (TM TN x 32 KB), then the 4 TM TN quarter-blocks (8 rows x 32 columns
// each) are shared out over the eight waves -- task t = wave, wave + 8: block t / 4, rows 8 (t % 4) + 4 (lane / 32) + e of
// column lane % 32, the mapping of a thread of conv_split_reduce_kernel -- two barriers per launch whatever the tile
// (block after block, finished by four of the waves: 2 TM TN barriers and 11.5 us for a 64 x 64 tile's launch instead of 10).
template <int TM, int TN, int NT>
struct SmallLds {
  static constexpr int NB = TM * TN;
  static constexpr int RED = 0;                                   // float [NB][8][16][64]
  static constexpr int T32 = RED + NB * SM_WAVES * 16 * 64 * 4;   // float [NB][32][33]
  static constexpr int AOFF = T32 + NB * 32 * 33 * 4;             // unsigned [NT][32 TM]
  static constexpr int MX = AOFF + NT * 32 * TM * 4;              // float [3][8]
  static constexpr int BYTES = MX + 3 * SM_WAVES * 4;
};

template <int NT, int TM, int TN, int CH, bool HEAD = false>
__global__ __launch_bounds__(64 * SM_WAVES) void conv_small_kernel(const GatherConvArgs a) {
  constexpr int BMT = 32 * TM, BNT = 32 * TN, NB = TM * TN;
  constexpr int NTASK = (4 * NB + SM_WAVES - 1) / SM_WAVES;      // quarter-blocks per wave
  using LDS = SmallLds<TM, TN, NT>;
  extern __shared__ unsigned char smem[];
  float (*red)[SM_WAVES][16][64] = reinterpret_cast<float (*)[SM_WAVES][16][64]>(smem + LDS::RED);
  float (*t32)[32][33] = reinterpret_cast<float (*)[32][33]>(smem + LDS::T32);
  unsigned (*s_aoff)[BMT] = reinterpret_cast<unsigned (*)[BMT]>(smem + LDS::AOFF);
  float (*s_mx)[SM_WAVES] = reinterpret_cast<float (*)[SM_WAVES]>(smem + LDS::MX);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hf = lane >> 5;
  const int tiles_m = (int)((a.M + BMT - 1) / BMT);
  // an XCD's workgroups (blockIdx % 8) take a contiguous run of tiles, row tile fastest: the workgroups that share a
  // column of the weights (the large operand of the 13x13 layers) sit on one XCD and read it from its L2
  const int logical = xcd_remap(blockIdx.x, a.nblocks);
  const int tile_n = logical / tiles_m, tile_m = logical - tile_n * tiles_m;
  const int KB = a.Cs >> 4;
  const int nsteps = NT * KB;
  const unsigned char* srcp = reinterpret_cast<const unsigned char*>(a.src);
  const unsigned char* wgtp = reinterpret_cast<const unsigned char*>(a.wgt);
  const unsigned lane_unit = (unsigned)(hf * 256);
  // byte offset of (source row of pixel mp under tap t, channel block 0) in the planes of src; a pixel past M, a tap
  // outside the image: the all-zero block
  auto src_row_offset = [&](const long long mp, const int t) -> unsigned {
    long long ms = (long long)a.zero_blk_src * 16 + (mp & 15);
    if (mp < a.M) {
      const int HgWg = a.Hg * a.Wg;
      const int n = (int)(mp / HgWg);
      const int rem = (int)(mp - (long long)n * HgWg);
      const int y = rem / a.Wg, x = rem - y * a.Wg;
      const int ys = y * a.sy + a.taps[t].oy, xs = x * a.sx + a.taps[t].ox;
      if (ys >= 0 && ys < a.Hs && xs >= 0 && xs < a.Ws) ms = ((long long)n * a.Hs + ys) * a.Ws + xs;
    }
    return (unsigned)((ms >> 4) * KB) * PL_RECORD + (unsigned)(ms & 15) * 16;
  };
  unsigned aoff1[TM];
  if constexpr (NT == 1) {
    if (a.sy == 1 && a.sx == 1 && a.Hg == a.Hs && a.Wg == a.Ws && a.taps[0].oy == 0 && a.taps[0].ox == 0) {
      // (pixel m is row m of src: no divisions)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const long long mp = (long long)tile_m * BMT + 32 * i + r;
        const long long ms = mp < a.M ? mp : (long long)a.zero_blk_src * 16 + (mp & 15);
        aoff1[i] = (unsigned)((ms >> 4) * KB) * PL_RECORD + (unsigned)(ms & 15) * 16 + lane_unit;
      }
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i) aoff1[i] = src_row_offset((long long)tile_m * BMT + 32 * i + r, 0) + lane_unit;
    }
  } else {
    for (int e = tid; e < NT * BMT; e += 64 * SM_WAVES) {
      const int t = e / BMT, rr = e - t * BMT;
      s_aoff[t][rr] = src_row_offset((long long)tile_m * BMT + rr, t);
    }
  }
  unsigned boff[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int crow = tile_n * BNT + 32 * j + r;
    boff[j] = (unsigned)((crow >> 4) * (a.ldw >> 4)) * PL_RECORD + (unsigned)(crow & 15) * 16 + lane_unit;
  }
  // steps past the end multiply by the all-zero block of the weights (no branch in the loop)
  const unsigned bzero = (unsigned)(a.zero_blk_wgt * (a.ldw >> 4)) * PL_RECORD + (unsigned)(r & 15) * 16 + lane_unit;

  // what the epilogue needs besides the sums is requested first, per quarter-block of this wave
  float ib = 0.f, rb = 0.f;
  const float unscale = reinterpret_cast<const float*>(srcp + a.src_bytes - PL_HEADER)[2] *
                        reinterpret_cast<const float*>(wgtp + a.wgt_bytes - PL_HEADER)[2];
  float bv[NTASK], esc[NTASK], esh[NTASK], rv[NTASK][4];
  if constexpr (!HEAD) {
    for (int w = tid; w < a.pl_in_n; w += 64 * SM_WAVES) ib = fmaxf(ib, __builtin_bit_cast(float, a.pl_in_bound[w]));
    if (a.pl_res_bound != nullptr)
      for (int w = tid; w < a.pl_res_n; w += 64 * SM_WAVES) rb = fmaxf(rb, __builtin_bit_cast(float, a.pl_res_bound[w]));
  }
#pragma unroll
  for (int k = 0; k < NTASK; ++k) {
    const int t = wave + SM_WAVES * k;
    bv[k] = 0.f; esc[k] = 1.f; esh[k] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) rv[k][e] = 0.f;
    if (t < 4 * NB) {
      const int b = t >> 2, q = t & 3, bi = b / TN, bj = b - bi * TN;
      const int col = tile_n * BNT + 32 * bj + r;
      if constexpr (HEAD) {
        bv[k] = (a.bias != nullptr && col < a.Cout) ? a.bias[col] : 0.f;
      } else {
        bv[k] = a.bias != nullptr ? a.bias[col] : 0.f;
        esc[k] = a.epi_scale[col];
        esh[k] = a.epi_shift[col];
        if (a.epi_res != nullptr) {
          const long long mrow = (long long)tile_m * BMT + 32 * bi + 8 * q + 4 * hf;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (mrow + e < a.M) rv[k][e] = a.epi_res[(mrow + e) * a.Cd + col];
        }
      }
    }
  }
  if constexpr (NT > 1) __syncthreads();   // s_aoff

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
  const int per_wave = (nsteps + SM_WAVES - 1) / SM_WAVES;       // steps of this wave: s = wave + 8 u
  const int nch = (per_wave + CH - 1) / CH;                      // chunks of CH steps
  // (t, kb) of the wave's next step, kept incrementally: the steps are requested in increasing order
  int nx_s = wave, nx_t = wave / KB, nx_kb = wave - (wave / KB) * KB;
  using Buf = SmallBuf<TM, TN, CH>;
  auto load = [&](Buf& b) {
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const bool live = nx_s < nsteps;                           // wave-uniform
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        unsigned ao;
        if constexpr (NT == 1) {
          ao = aoff1[i] + (live ? (unsigned)nx_s * PL_RECORD : 0u);
        } else {
          ao = s_aoff[live ? nx_t : 0][32 * i + r] + lane_unit + (live ? (unsigned)nx_kb * PL_RECORD : 0u);
        }
        b.ah[u][i] = *reinterpret_cast<const u32x4*>(srcp + ao);
        b.al[u][i] = *reinterpret_cast<const u32x4*>(srcp + ao + 512);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const unsigned bo = live ? boff[j] + (unsigned)nx_s * PL_RECORD : bzero;
        b.bh[u][j] = *reinterpret_cast<const u32x4*>(wgtp + bo);
        b.bl[u][j] = *reinterpret_cast<const u32x4*>(wgtp + bo + 512);
      }
      nx_s += SM_WAVES;
      nx_kb += SM_WAVES;
      while (nx_kb >= KB) {
        nx_kb -= KB;
        ++nx_t;
      }
    }
  };
  auto mma = [&](const Buf& b) {
#pragma unroll
    for (int u = 0; u < CH; ++u)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const f16x8 ah = __builtin_bit_cast(f16x8, b.ah[u][i]), al = __builtin_bit_cast(f16x8, b.al[u][i]);
          const f16x8 bh = __builtin_bit_cast(f16x8, b.bh[u][j]), bl = __builtin_bit_cast(f16x8, b.bl[u][j]);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][j], 0, 0, 0);
        }
  };
  Buf b0, b1;
  load(b0);
  for (int c = 0; c < nch; c += 2) {
    load(b1);             // (past the end: the weights' all-zero block)
    mma(b0);
    load(b0);
    mma(b1);
  }

  // every wave's partial of every block to LDS; the words of the a-priori bound (requested before the main loop) are
  // first looked at here
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) red[i * TN + j][wave][k][lane] = acc[i][j][k];
  if constexpr (!HEAD) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      ib = fmaxf(ib, __shfl_xor(ib, o, 64));
      rb = fmaxf(rb, __shfl_xor(rb, o, 64));
    }
    if (lane == 0) {
      s_mx[1][wave] = ib;
      s_mx[2][wave] = rb;
    }
  }
  __syncthreads();
  float psc = 1.f;
  if constexpr (!HEAD) {
    // scale of the outgoing planes from the a-priori bound (see conv_split_reduce_kernel)
    float in_b = 0.f, res_b = 0.f;
#pragma unroll
    for (int w = 0; w < SM_WAVES; ++w) {
      in_b = fmaxf(in_b, s_mx[1][w]);
      res_b = fmaxf(res_b, s_mx[2][w]);
    }
    const float bnd = (a.pl_pred[0] * in_b + a.pl_pred[1] + res_b) * 1.001f + 1e-30f;
    psc = planes_scale_from_bound(__builtin_bit_cast(unsigned, bnd));
    if (blockIdx.x == 0 && tid == 0) {
      unsigned* header = reinterpret_cast<unsigned*>(a.out_planes + planes_body_bytes(a.M, a.Cout));
      header[0] = __builtin_bit_cast(unsigned, bnd);
      reinterpret_cast<float*>(header)[1] = psc;
      reinterpret_cast<float*>(header)[2] = 1.f / psc;
    }
  }
  // this wave's quarter-blocks: the eight partials added in wave order, epilogue, fp32 store
  float mxf = 0.f;
  const int D = HEAD ? 5 + a.head_C : 1;
#pragma unroll
  for (int k = 0; k < NTASK; ++k) {
    const int t = wave + SM_WAVES * k;
    if (t < 4 * NB) {
      const int b = t >> 2, q = t & 3, bi = b / TN, bj = b - bi * TN;
      const int col = tile_n * BNT + 32 * bj + r;
      const long long mrow = (long long)tile_m * BMT + 32 * bi + 8 * q + 4 * hf;
      float anc = 0.f;
      int kk = 0;
      if constexpr (HEAD) {
        kk = col % D;
        if (col < a.Cout && (kk == 2 || kk == 3)) anc = a.head_anchors[(col / D) * 2 + (kk - 2)];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float sum = red[b][0][4 * q + e][lane];
#pragma unroll
        for (int w = 1; w < SM_WAVES; ++w) sum += red[b][w][4 * q + e][lane];
        if constexpr (HEAD) {
          if (mrow + e < a.M && col < a.Cout) {
            const float v = fmaf(sum, unscale, bv[k]);
            if (a.dst != nullptr) a.dst[(mrow + e) * a.Cd + col] = v;
            // (the expressions of head_fwd_pointwise_kernel, elementwise.hip)
            a.head_y[(mrow + e) * a.Cd + col] = (kk == 2 || kk == 3) ? expf(v) * anc : 1.f / (1.f + expf(-v));
          }
        } else {
          float v = 0.f;
          if (mrow + e < a.M) {
            v = fmaf(sum, unscale, bv[k]);
            v = act_fwd(fmaf(esc[k], v, esh[k]), a.epi_act);
            v += rv[k][e];
            a.dst[(mrow + e) * a.Cd + col] = v;
            mxf = fmaxf(mxf, fabsf(v));
          }
          t32[b][8 * q + 4 * hf + e][r] = v;
        }
      }
    }
  }
  if constexpr (HEAD) return;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mxf = fmaxf(mxf, __shfl_xor(mxf, o, 64));
  if (lane == 0) s_mx[0][wave] = mxf;
  __syncthreads();
  // the finished blocks as planes (through t32: one (row, 8-channel group) unit per thread and turn)
  for (int u = tid; u < NB * 128; u += 64 * SM_WAVES) {
    const int b = u >> 7, rr = (u >> 2) & 31, g = u & 3, bi = b / TN, bj = b - bi * TN;
    const long long mo = (long long)tile_m * BMT + 32 * bi + rr;
    const int c0 = tile_n * BNT + 32 * bj + g * 8;
    if (mo < a.M) {
      const f32x4 o0 = {t32[b][rr][g * 8 + 0], t32[b][rr][g * 8 + 1], t32[b][rr][g * 8 + 2], t32[b][rr][g * 8 + 3]};
      const f32x4 o1 = {t32[b][rr][g * 8 + 4], t32[b][rr][g * 8 + 5], t32[b][rr][g * 8 + 6], t32[b][rr][g * 8 + 7]};
      store_planes8(a.out_planes, mo, c0 >> 3, a.Cout, o0, o1, psc);
    }
  }
  // max|dst| of this workgroup's tile: ONE word per workgroup, a plain store
  if (tid == 0) {
    float m = 0.f;
#pragma unroll
    for (int w = 0; w < SM_WAVES; ++w) m = fmaxf(m, s_mx[0][w]);
    a.pl_out_words[blockIdx.x] = __builtin_bit_cast(unsigned, m);
  }
}

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// Tile of a launch, 0 = not for this kernel. Measured on YOLOv3-416 at bs 1 (profiles/r06_o_conv_small_tiles.txt, us per
// launch; before: the split-K pair of launches, or tile kernel + planes pass):
//   1x1  13x13 1024->512: (1,1) 96 workgroups 9.4 | (2,1) 12.4 | (2,2) 15.7         before 16.5
//   1x1  26x26 512->256:  (1,1) 176: 8.8 | (2,1) 10.9 | (2,2) 11.6                   before 16.5
//   1x1  52x52 256->128:  (1,1) 340: 14.1 | (2,1) 172: 11.2 | (2,2) 86: 11.5         before 16.5
//   1x1  104x104 128->64: (1,1) 676: 21.0 | (2,1) 338: 19.6 | (2,2) 169: 12.4        before 17.8
//   1x1  208x208 64->32:  (1,1) 1352: 40.8 | (2,1) 676: 29.6                          before 20.8
//   3x3  13x13 512->1024: (1,1) 192: 24.6 | (2,1) 96: 35.8 | (2,2) 48: 44.4          before 23.3
//   3x3  26x26 256->512:  (1,1) 352: 26.7 | (2,1) 176: 22.8 | (2,2) 88: 26.6         before 21.2
//   3x3  52x52 128->256:  (1,1) 680: 30.4 | (2,1) 344: 30.6 | (2,2) 172: 19.9        before 21-23
// One workgroup per CU at a time (8 waves of 160-246 registers) takes in ~50-65 GB/s whatever the tile: launches of more
// than 256 workgroups pay a second round, launches of few large tiles leave CUs without a stream. So: 1x1 units take the
// 32 x 32 tile while it gives at most 256 workgroups, the 64 x 64 tile while THAT gives at most 256, else the kernels of the
// training step; a 3x3 unit asks for 9x the bytes per output and is no faster than the split-K pair except at 52x52.
// YOLO_CONV_SMALL: 0 off, 1 (default) the policy below, 3 every 3x3 unit the grid limit allows too; YOLO_CONV_SMALL_TILE = 11 / 21 / 22 forces a tile and
// YOLO_CONV_SMALL_GRID the largest launch (experiments).
static int small_tile(const GatherConvArgs& a) {
  static const int on = env_int("YOLO_CONV_SMALL", 1);
  static const int tile_env = env_int("YOLO_CONV_SMALL_TILE", 0);
  static const int grid_env = env_int("YOLO_CONV_SMALL_GRID", 0);
  if (!on) return 0;
  if (a.ntaps != 1 && a.ntaps != 9) return 0;
  if ((a.Cs % 16) != 0 || (a.Cout % 32) != 0 || a.ldw != a.ntaps * a.Cs) return 0;
  if (a.stats != nullptr || a.accumulate || a.bwd_y != nullptr || a.ncls > 1) return 0;
  if (a.out_planes == nullptr || a.epi_scale == nullptr || a.pl_pred == nullptr || a.pl_out_words == nullptr) return 0;
  if (a.osy != 1 || a.osx != 1 || a.ooy != 0 || a.oox != 0 || a.Hd != a.Hg || a.Wd != a.Wg || a.Cd != a.Cout) return 0;
  for (int t = 0; t < a.ntaps; ++t)
    if (a.taps[t].woff != t * a.Cs) return 0;
  const long long g11 = ((a.M + 31) / 32) * (a.Cout / 32);
  const long long g22 = (a.Cout % 64) == 0 ? ((a.M + 63) / 64) * (a.Cout / 64) : (1LL << 40);
  if (tile_env) {   // forced tile: any launch of at most YOLO_CONV_SMALL_GRID (default 2048) 32 x 32 tiles
    const long long cap = grid_env > 0 ? grid_env : 2048;
    if (g11 > (cap < YOLO_INFER_BOUND_WORDS ? cap : YOLO_INFER_BOUND_WORDS)) return 0;
    return (tile_env == 22 && (a.Cout % 64) != 0) ? 21 : tile_env;
  }
  const long long cap = grid_env > 0 ? grid_env : 256;
  if (a.ntaps == 9 && on != 3) {
    // 3x3 units: only where the 64 x 64 tile fills at least half the chip in one round with short streams (at most 72
    // steps = 128 input channels: the 52x52 layers of YOLOv3-416, 19.9 against 23.9 us); elsewhere the split-K pair wins
    // (... and the 104x104 ones, 36 steps: 338 workgroups in two short rounds, 27.6 against 30 us)
    const int steps = a.ntaps * (a.Cs >> 4);
    static const int two_rounds = env_int("YOLO_CONV_SMALL_2R", 1);
    return (g22 >= 128 && ((g22 <= cap && steps <= 72) || (two_rounds && g22 <= 2 * cap && steps <= 36))) ? 22 : 0;
  }
  if (g11 <= cap) return 11;
  if (g22 <= cap) return 22;
  return 0;
}

bool conv_small_supported(const GatherConvArgs& a) { return small_tile(a) != 0; }

template <int NT, int TM, int TN, int CH, bool HEAD = false>
static int launch_small(GatherConvArgs& a, hipStream_t st, int* nwg) {
  a.nblocks = (int)(((a.M + 32 * TM - 1) / (32 * TM)) * ((a.Cout + 32 * TN - 1) / (32 * TN)));
  *nwg = a.nblocks;
  constexpr int lds = SmallLds<TM, TN, NT>::BYTES;   // (up to 150 KB: the 64 x 64 tile's four blocks at once)
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_small_kernel<NT, TM, TN, CH, HEAD>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_small_kernel<NT, TM, TN, CH, HEAD>), dim3((unsigned)a.nblocks), dim3(64 * SM_WAVES), lds, st, a);
  return check_launch("conv_small_kernel");
}

// the detection-head form: 1x1, stride 1, dense, bias, any Cout; 32 x 32 tiles while they give at most 256 workgroups, else
// 64 x 64 while those do
static int small_head_tile(const GatherConvArgs& a) {
  static const int on = env_int("YOLO_CONV_SMALL", 1);
  if (!on || a.head_y == nullptr || a.head_anchors == nullptr || a.head_A <= 0 || a.head_C <= 0) return 0;
  if (a.ntaps != 1 || (a.Cs % 16) != 0 || a.ldw != a.Cs || a.Cout != a.head_A * (5 + a.head_C)) return 0;
  if (a.sy != 1 || a.sx != 1 || a.Hg != a.Hs || a.Wg != a.Ws || a.taps[0].oy != 0 || a.taps[0].ox != 0) return 0;
  if (a.stats != nullptr || a.accumulate || a.bwd_y != nullptr || a.ncls > 1 || a.epi_scale != nullptr || a.epi_res != nullptr) return 0;
  if (a.osy != 1 || a.osx != 1 || a.ooy != 0 || a.oox != 0 || a.Hd != a.Hg || a.Wd != a.Wg || a.Cd != a.Cout) return 0;
  const long long g11 = ((a.M + 31) / 32) * ((a.Cout + 31) / 32), g22 = ((a.M + 63) / 64) * ((a.Cout + 63) / 64);
  if (g11 <= 256) return 11;
  if (g22 <= 256) return 22;
  return 0;
}
bool conv_small_head_supported(const GatherConvArgs& a) { return small_head_tile(a) != 0; }

// *nwg = workgroups of the launch = words of max|dst| written to a.pl_out_words
int launch_conv_small(GatherConvArgs& a, hipStream_t st, int* nwg) {
  const long long rowsA = (long long)a.N * a.Hs * a.Ws;
  const long long bytesA = planes_bytes(rowsA, a.Cs), bytesB = planes_bytes(a.Cout, a.ldw);
  if (bytesA >= (1LL << 32) || bytesB >= (1LL << 32)) {
    set_error("conv_small: operand planes exceed 4 GiB (%lld, %lld bytes)", bytesA, bytesB);
    return YOLO_ERR_INVALID_ARG;
  }
  a.src_bytes = (unsigned)bytesA;
  a.wgt_bytes = (unsigned)bytesB;
  a.zero_blk_src = (int)((rowsA + 15) / 16);
  a.zero_blk_wgt = (a.Cout + 15) / 16;
  a.split_parts = 1;
  if (a.head_y != nullptr) {
    const int ht = small_head_tile(a);
    if (ht == 11) return launch_small<1, 1, 1, 4, true>(a, st, nwg);
    if (ht == 22) return launch_small<1, 2, 2, 2, true>(a, st, nwg);
    set_error("conv_small(head): launch not supported");
    return YOLO_ERR_INVALID_ARG;
  }
  const int tile = small_tile(a);
  if (tile == 0) {
    set_error("conv_small: launch not supported");
    return YOLO_ERR_INVALID_ARG;
  }
  if (a.ntaps == 1) {
    if (tile == 22) return launch_small<1, 2, 2, 2>(a, st, nwg);
    if (tile == 21) return launch_small<1, 2, 1, 4>(a, st, nwg);
    return launch_small<1, 1, 1, 4>(a, st, nwg);
  }
  if (tile == 22) return launch_small<9, 2, 2, 2>(a, st, nwg);
  if (tile == 21) return launch_small<9, 2, 1, 4>(a, st, nwg);
  return launch_small<9, 1, 1, 4>(a, st, nwg);
}

}  // namespace yolo

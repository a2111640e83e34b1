// Convolution forward / dgrad on operands that were split into fp16 planes BEFORE the kernel runs ("planes"
// format, planes.hpp) and that reach LDS by LDS-DMA (buffer_load ... lds): the main loop has no VALU split
// arithmetic, no ds_write and no VGPR staging -- only DMA issue, fragment reads and the 3 fp16 MFMA passes
// (l*h, h*l, h*h) per 32x32 fragment pair; the epilogue undoes the two power-of-two operand scales.
#include "planes_epilogue.hpp"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace yolo {

// ---- producing planes from fp32 (tensors that no fused producer writes: filters, concat outputs, ...) ----
// pass 1: max |x| -> header[0] (bit pattern; non-negative floats order like unsigned integers)
__global__ __launch_bounds__(256) void planes_amax_kernel(const float* __restrict__ x, long long n4,
                                                         unsigned* __restrict__ header) {
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  // one atomic per workgroup, skipped when the header already holds a larger value (same-address atomics from all
  // eight XCDs serialise: one per WAVE made a 0.5 MB tensor take 20 us)
  __shared__ float wmax[4];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    if (__builtin_bit_cast(unsigned, m) > *reinterpret_cast<volatile unsigned*>(header))
      atomicMax(header, __builtin_bit_cast(unsigned, m));
  }
}

// pass 2: one thread = (row, 8-channel group); adjacent lanes read adjacent 32-byte pieces of a row.
// Also zero-fills the rows past the end + the zero block and completes the header (scale, 1/scale).
// Workgroup = 4 consecutive 16-row blocks x 16 channel groups (see the batched form below for why).
constexpr int SPLIT_BLOCKS_PER_WG = 4;
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, long long rows, int C,
                                                          unsigned char* __restrict__ out, long long rows_padded) {
  unsigned* header = reinterpret_cast<unsigned*>(out + planes_body_bytes(rows, C));
  const float sc = planes_scale_from_bound(header[0]);
  const int G = C >> 3;
  const int gbn = (G + 15) >> 4;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    reinterpret_cast<float*>(header)[1] = sc;
    reinterpret_cast<float*>(header)[2] = 1.f / sc;
  }
  const long long rb = blockIdx.x / gbn;
  const int g = (int)(blockIdx.x - rb * gbn) * 16 + (threadIdx.x >> 4);
  if (g >= G) return;
  const long long row0 = rb * (16 * SPLIT_BLOCKS_PER_WG) + (threadIdx.x & 15);
  f32x4 v[SPLIT_BLOCKS_PER_WG][2];
#pragma unroll
  for (int u = 0; u < SPLIT_BLOCKS_PER_WG; ++u) {
    const long long row = row0 + 16 * u;
    v[u][0] = v[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (row < rows) {
      const float* p = x + row * C + g * 8;
      v[u][0] = *reinterpret_cast<const f32x4*>(p);
      v[u][1] = *reinterpret_cast<const f32x4*>(p + 4);
    }
  }
#pragma unroll
  for (int u = 0; u < SPLIT_BLOCKS_PER_WG; ++u) {
    const long long row = row0 + 16 * u;
    if (row >= rows_padded) continue;
    const Planes8 s = split8(v[u][0], v[u][1], sc);
    unsigned char* o = out + planes_unit_offset(row, g, C);
    *reinterpret_cast<u32x4*>(o) = s.h;
    *reinterpret_cast<u32x4*>(o + 512) = s.l;
  }
}

// The same split with the bound taken from a per-channel max|x| vector (left by a conv epilogue) plus an optional
// extra term (the bound of a residual that was added afterwards): every workgroup folds the C values itself, so the
// inference path needs no separate bound kernel. Also leaves the bound in *out_bound (optional).
__global__ __launch_bounds__(256) void split_planes_absmax_kernel(const float* __restrict__ x, long long rows, int C,
                                                                 const unsigned* __restrict__ absmax,
                                                                 const float* __restrict__ extra, int extra_n,
                                                                 unsigned char* __restrict__ out, long long rows_padded,
                                                                 float* __restrict__ out_bound) {
  __shared__ float s_max[2][4];
  // (extra: the bound of the residual tensor as extra_n non-negative floats whose maximum it is)
  float ex = 0.f;
  if (extra != nullptr)
    for (int w = threadIdx.x; w < extra_n; w += 256) ex = fmaxf(ex, extra[w]);
  float m = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) m = fmaxf(m, __builtin_bit_cast(float, absmax[c]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    m = fmaxf(m, __shfl_xor(m, o, 64));
    ex = fmaxf(ex, __shfl_xor(ex, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    s_max[0][threadIdx.x >> 6] = m;
    s_max[1][threadIdx.x >> 6] = ex;
  }
  __syncthreads();
  const float bound = fmaxf(fmaxf(s_max[0][0], s_max[0][1]), fmaxf(s_max[0][2], s_max[0][3])) * 1.001f +
                      fmaxf(fmaxf(s_max[1][0], s_max[1][1]), fmaxf(s_max[1][2], s_max[1][3])) + 1e-30f;
  unsigned* header = reinterpret_cast<unsigned*>(out + planes_body_bytes(rows, C));
  const float sc = planes_scale_from_bound(__builtin_bit_cast(unsigned, bound));
  const int G = C >> 3;
  const int gbn = (G + 15) >> 4;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    header[0] = __builtin_bit_cast(unsigned, bound);
    reinterpret_cast<float*>(header)[1] = sc;
    reinterpret_cast<float*>(header)[2] = 1.f / sc;
    if (out_bound != nullptr) out_bound[0] = bound;
  }
  const long long rb = blockIdx.x / gbn;
  const int g = (int)(blockIdx.x - rb * gbn) * 16 + (threadIdx.x >> 4);
  if (g >= G) return;
  const long long row0 = rb * (16 * SPLIT_BLOCKS_PER_WG) + (threadIdx.x & 15);
  f32x4 v[SPLIT_BLOCKS_PER_WG][2];
#pragma unroll
  for (int u = 0; u < SPLIT_BLOCKS_PER_WG; ++u) {
    const long long row = row0 + 16 * u;
    v[u][0] = v[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (row < rows) {
      const float* p = x + row * C + g * 8;
      v[u][0] = *reinterpret_cast<const f32x4*>(p);
      v[u][1] = *reinterpret_cast<const f32x4*>(p + 4);
    }
  }
#pragma unroll
  for (int u = 0; u < SPLIT_BLOCKS_PER_WG; ++u) {
    const long long row = row0 + 16 * u;
    if (row >= rows_padded) continue;
    const Planes8 s = split8(v[u][0], v[u][1], sc);
    unsigned char* o = out + planes_unit_offset(row, g, C);
    *reinterpret_cast<u32x4*>(o) = s.h;
    *reinterpret_cast<u32x4*>(o + 512) = s.l;
  }
}

// Channel concatenation straight into planes: up to four dense fp32 sources [rows][C_s] (C_s % 8 == 0) become the planes of
// [rows][sum C_s] in ONE pass -- no fp32 copy of the concatenated tensor, no max pass: the bound is the largest of the
// sources' recorded bounds (each source left one float behind: BatchNorm's output bound, or its input's for pools /
// upsampling). Replaces copy_channels_in x nsrc + planes_amax + split_planes (20 B per element moved) by 8 B per element.
// Optionally (dst32 != nullptr) the fp32 concatenation is written as well, for consumers that are not planes convolutions.
struct ConcatSrcs {
  const float* x[4];
  const float* bound[4];
  int bound_n[4];   // words at bound[i] whose maximum is the source's bound (1 = one float; more = what a one-pass inference unit left)
  int up[4];        // 1 = source i has HALF the height and width of the result: UpSampling2D(2), nearest, read on the fly
  int g_end[4];     // exclusive prefix ends in 8-channel groups
  int C[4];
  int n;
  int H, W;         // the result's rows are N x H x W pixels (used when any up[i])
};
__global__ __launch_bounds__(256) void split_planes_concat_kernel(const ConcatSrcs cs, long long rows, int C,
                                                                 unsigned char* __restrict__ out, long long rows_padded,
                                                                 float* __restrict__ dst32, float* __restrict__ out_bound) {
  float b = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < cs.n) {
      if (cs.bound_n[i] <= 1) {
        b = fmaxf(b, cs.bound[i][0]);
      } else {
        // (the words of a one-pass unit: folded here instead of by a launch of their own -- 4.5 us in a bs-1 graph)
        __shared__ float sb[4][4];
        float m = 0.f;
        for (int w = threadIdx.x; w < cs.bound_n[i]; w += 256) m = fmaxf(m, cs.bound[i][w]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((threadIdx.x & 63) == 0) sb[i][threadIdx.x >> 6] = m;
        __syncthreads();
        b = fmaxf(b, fmaxf(fmaxf(sb[i][0], sb[i][1]), fmaxf(sb[i][2], sb[i][3])));
      }
    }
  const float bound = b * 1.001f + 1e-30f;
  unsigned* header = reinterpret_cast<unsigned*>(out + planes_body_bytes(rows, C));
  const float sc = planes_scale_from_bound(__builtin_bit_cast(unsigned, bound));
  const int G = C >> 3;
  const int gbn = (G + 15) >> 4;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    header[0] = __builtin_bit_cast(unsigned, bound);
    reinterpret_cast<float*>(header)[1] = sc;
    reinterpret_cast<float*>(header)[2] = 1.f / sc;
    if (out_bound != nullptr) out_bound[0] = bound;
  }
  const long long rb = blockIdx.x / gbn;
  const int g = (int)(blockIdx.x - rb * gbn) * 16 + (threadIdx.x >> 4);
  if (g >= G) return;
  // which source owns channel group g
  int si = 0, g0 = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (i + 1 < cs.n && g >= cs.g_end[i]) {
      si = i + 1;
      g0 = cs.g_end[i];
    }
  const float* src = cs.x[si];
  const int Cs = cs.C[si];
  const long long row0 = rb * (16 * SPLIT_BLOCKS_PER_WG) + (threadIdx.x & 15);
  f32x4 v[SPLIT_BLOCKS_PER_WG][2];
#pragma unroll
  for (int u = 0; u < SPLIT_BLOCKS_PER_WG; ++u) {
    const long long row = row0 + 16 * u;
    v[u][0] = v[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (row < rows) {
      long long srow = row;
      if (cs.up[si]) {   // pixel (y, x) of the result reads pixel (y / 2, x / 2) of the half-size source
        const int HW = cs.H * cs.W;
        const long long n = row / HW;
        const int rem = (int)(row - n * HW);
        const int y = rem / cs.W, x = rem - y * cs.W;
        srow = (n * (cs.H >> 1) + (y >> 1)) * (cs.W >> 1) + (x >> 1);
      }
      const float* p = src + srow * Cs + (g - g0) * 8;
      v[u][0] = *reinterpret_cast<const f32x4*>(p);
      v[u][1] = *reinterpret_cast<const f32x4*>(p + 4);
    }
  }
#pragma unroll
  for (int u = 0; u < SPLIT_BLOCKS_PER_WG; ++u) {
    const long long row = row0 + 16 * u;
    if (row >= rows_padded) continue;
    const Planes8 sp = split8(v[u][0], v[u][1], sc);
    unsigned char* o = out + planes_unit_offset(row, g, C);
    *reinterpret_cast<u32x4*>(o) = sp.h;
    *reinterpret_cast<u32x4*>(o + 512) = sp.l;
    if (dst32 != nullptr && row < rows) {
      float* q = dst32 + row * C + g * 8;
      *reinterpret_cast<f32x4*>(q) = v[u][0];
      *reinterpret_cast<f32x4*>(q + 4) = v[u][1];
    }
  }
}

// Dense [rows][Csrc] source with Csrc < C = Csrc rounded up to 16 (the 255-channel head gradients): scalar loads,
// columns >= Csrc of the planes are zero.
__global__ __launch_bounds__(256) void planes_amax_scalar_kernel(const float* __restrict__ x, long long n,
                                                                unsigned* __restrict__ header) {
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && __builtin_bit_cast(unsigned, m) > *reinterpret_cast<volatile unsigned*>(header))
    atomicMax(header, __builtin_bit_cast(unsigned, m));
}
__global__ __launch_bounds__(256) void split_planes_padded_kernel(const float* __restrict__ x, long long rows, int Csrc,
                                                                 int C, unsigned char* __restrict__ out,
                                                                 long long rows_padded) {
  unsigned* header = reinterpret_cast<unsigned*>(out + planes_body_bytes(rows, C));
  const float sc = planes_scale_from_bound(header[0]);
  const int G = C >> 3;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t == 0) {
    reinterpret_cast<float*>(header)[1] = sc;
    reinterpret_cast<float*>(header)[2] = 1.f / sc;
  }
  if (t >= rows_padded * G) return;
  const long long row = t / G;
  const int g = (int)(t - row * G);
  f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
  if (row < rows) {
    const float* p = x + row * Csrc;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (g * 8 + j < Csrc) v0[j] = p[g * 8 + j];
      if (g * 8 + 4 + j < Csrc) v1[j] = p[g * 8 + 4 + j];
    }
  }
  const Planes8 s = split8(v0, v1, sc);
  unsigned char* o = out + planes_unit_offset(row, g, C);
  *reinterpret_cast<u32x4*>(o) = s.h;
  *reinterpret_cast<u32x4*>(o + 512) = s.l;
}

// Workgroup tile BM x BN, NW = WGM*WGN waves; every wave is also the loader of ONE 32-row block of
// A (waves 0 .. BM/32-1) or B (the rest): two DMA instructions (planes h, l) per 16-k stage.
// Ring of 3 stage buffers in LDS + 2 fragment register sets, one barrier per stage. Iteration kt:
//   wait vmcnt(2)  -> my DMAs of stage kt+1 have landed (those of kt+2 may still fly)
//   barrier        -> everybody's have; everybody finished reading stage kt's buffer (last iteration)
//   read fragments of stage kt+1 into the other register set
//   3 MFMA passes per fragment pair on stage kt's registers, with the 2 DMAs of stage kt+3 (into stage
//   kt's buffer) issued between them
template <int BM, int BN, int WGM, int WGN, int DBG = 0>
__global__ __launch_bounds__(64 * WGM * WGN, (WGM * WGN >= 4 ? 2 : 1)) void gather_conv_planes_kernel(const GatherConvArgs a) {
  constexpr int NW = WGM * WGN;
  constexpr int TM = BM / WGM / 32;
  constexpr int TN = BN / WGN / 32;
  constexpr int RBA = BM / 32, RBB = BN / 32;
  // loader slots per wave: 1 (one 32-row block of A or of B per wave) or 2 (4-wave tiles); slots beyond the
  // RBA + RBB blocks repeat a B block -- identical bytes to the same LDS address -- so that every wave has
  // the same number of DMAs on its counter
  constexpr int LPW = (RBA + RBB + NW - 1) / NW;
  static_assert(LPW <= 4, "loader layout: slot s = wave + i*NW loads A block s, or B block (s - RBA) % RBB");
  constexpr int ND = PL_PLANES * LPW;   // DMA instructions per wave per stage
  constexpr int STAGE_BYTES = (RBA + RBB) * PL_PLANES * 1024;
  constexpr int NBUF = 3 + ((DBG >> 7) & 3);   // (bits 128 / 256: a deeper ring, NBUF - 1 stages in flight -- HBM-bound 1x1 layers)

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds_base = (unsigned)(size_t)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  // split-K (a.split_parts > 1, conv_win.hip: conv_split_reduce_kernel): grid = tiles x parts, part p of a tile
  // contracts the 16-channel blocks [p * cpt / P, (p + 1) * cpt / P) of every tap and stores its accumulators to a slab
  // (its own instantiation, DBG bit 32: inside the production kernel the slab path cost 44 registers = one wave per SIMD)
  constexpr bool SPLIT = (DBG & 32) != 0;
  const int SP = SPLIT ? a.split_parts : 1;
  const int part = SPLIT ? (int)blockIdx.x / a.nblocks : 0;
  const int tile = xcd_remap(SPLIT ? (int)blockIdx.x - part * a.nblocks : (int)blockIdx.x, a.nblocks);
  // multi-class launch (its own instantiation, DBG bit 64; a.ncls parity classes of a strided data gradient): tile =
  // (row tile, class, column tile); everything that depends on the output grid comes from the class
  constexpr bool MULTI = (DBG & 64) != 0;
  constexpr bool BNRED = (DBG & 512) != 0;   // the fused BatchNorm-backward reduction (planes_epilogue.hpp): its own kernels
  const int tile_n = tile % a.tiles_n;
  const int cls = MULTI ? (tile / a.tiles_n) % a.ncls : 0;
  const int tile_m = MULTI ? tile / (a.tiles_n * a.ncls) : tile / a.tiles_n;
  const long long m0 = (long long)tile_m * BM;
  const int n0 = tile_n * BN;
  const EpiGeom G = MULTI ? EpiGeom{a.cls[cls].M, a.cls[cls].Hg, a.cls[cls].Wg, a.cls[cls].ooy, a.cls[cls].oox, 0, 0, 0,
                                    tile_m * a.ncls + cls + 1}
                          : epi_geom_of(a);
  const int tap0 = MULTI ? a.cls[cls].tap0 : 0;
  const int ntaps = MULTI ? a.cls[cls].ntaps : a.ntaps;
  if (MULTI && m0 >= G.M) {   // (classes of an odd-sized image have different tile counts)
    if (a.bwd_y != nullptr)   // its slot of the fused BatchNorm-backward reduction still has to hold zeros
      for (int c = tid; c < BN; c += 64 * NW)
        if (n0 + c < a.Cout) {
          a.bwd_part[((long long)(G.slot1 - 1) * 2 + 0) * a.Cout + n0 + c] = 0.f;
          a.bwd_part[((long long)(G.slot1 - 1) * 2 + 1) * a.Cout + n0 + c] = 0.f;
        }
    return;
  }
  const int HgWg = G.Hg * G.Wg;

  // ---- loader role(s) ----
  const int r = lane & 31, hf = lane >> 5;
  const unsigned blkstrideA = (unsigned)((a.Cs >> 4) * PL_RECORD);
  const unsigned blkstrideB = (unsigned)((a.ldw >> 4) * PL_RECORD);
  const i32x4 rsrcA = planes_rsrc(a.src, a.src_bytes), rsrcB = planes_rsrc(a.wgt, a.wgt_bytes);
  const int cpt_all = a.Cs >> 4;  // 16-channel blocks per tap
  const int cb_lo = (part * cpt_all) / SP;
  const int cpt = ((part + 1) * cpt_all) / SP;   // one past my last channel block (the whole range without split-K)
  const int nk = ntaps * (cpt - cb_lo);

  bool isA[LPW];
  int nimg[LPW], ys0[LPW], xs0[LPW];
  unsigned rowbaseB[LPW], lds_mine[LPW], ld_voff[LPW], ld_soff[LPW];
#pragma unroll
  for (int i = 0; i < LPW; ++i) {
    const int slot = wave + i * NW;
    isA[i] = slot < RBA;
    const int rb = isA[i] ? slot : (slot - RBA) % RBB;
    nimg[i] = 0; ys0[i] = -(1 << 28); xs0[i] = 0; rowbaseB[i] = 0;
    if (isA[i]) {
      const long long m = m0 + rb * 32 + r;
      if (m < G.M) {
        nimg[i] = (int)(m / HgWg);
        const int rem = (int)(m - (long long)nimg[i] * HgWg);
        const int y = rem / G.Wg;
        ys0[i] = y * a.sy;
        xs0[i] = (rem - y * G.Wg) * a.sx;
      }
    } else {
      const int co = n0 + rb * 32 + r;
      const unsigned blk = co < a.Cout ? (unsigned)(co >> 4) : (unsigned)a.zero_blk_wgt;
      rowbaseB[i] = blk * blkstrideB + (co < a.Cout ? (co & 15) * 16 : 0) + hf * 256;
    }
    lds_mine[i] = lds_base + ((isA[i] ? 0 : RBA) + rb) * PL_PLANES * 1024;
    ld_voff[i] = 0; ld_soff[i] = 0;
  }

  // ---- loader state: the next stage to issue and its DMA offsets. Stage order: the 16-channel blocks are
  // walked in chunks of a.kc blocks; inside a chunk all taps, inside a tap the chunk's blocks. kc = Cs/16 is
  // plain tap-major order (every tap re-streams the whole input window: its lines have left L2 by the next tap
  // when 40+ workgroups share the 4 MB); a small kc makes the taps re-read the same few KB per pixel block
  // back to back while still streaming whole records. ----
  int ld_tap = 0, ld_kk = 0, ld_cb = cb_lo;   // tap, block inside the chunk, first block of the chunk
  auto loader_tap = [&]() {  // per-(chunk, tap) part of the source address (A: the shifted pixel, B: the filter tap)
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
      if (ld_cb >= cpt) {  // stages past the end (issued to keep the DMA count per stage uniform): zero block
        ld_voff[i] = isA[i] ? (unsigned)a.zero_blk_src * blkstrideA : (unsigned)a.zero_blk_wgt * blkstrideB;
        ld_soff[i] = 0;
      } else if (isA[i]) {
        const int ys = ys0[i] + a.taps[tap0 + ld_tap].oy, xs = xs0[i] + a.taps[tap0 + ld_tap].ox;
        const bool ok = ((unsigned)ys < (unsigned)a.Hs) && ((unsigned)xs < (unsigned)a.Ws);
        const int pix = (nimg[i] * a.Hs + ys) * a.Ws + xs;
        ld_voff[i] = (ok ? ((unsigned)pix >> 4) : (unsigned)a.zero_blk_src) * blkstrideA + (ok ? (pix & 15) * 16 : 0) + hf * 256;
        ld_soff[i] = (unsigned)ld_cb * PL_RECORD;
      } else {
        ld_voff[i] = rowbaseB[i];
        ld_soff[i] = (unsigned)((a.taps[tap0 + ld_tap].woff >> 4) + ld_cb) * PL_RECORD;
      }
    }
  };
  auto loader_next = [&]() {
#pragma unroll
    for (int i = 0; i < LPW; ++i) ld_soff[i] += PL_RECORD;
    const int kc_cur = (cpt - ld_cb < a.kc) ? (cpt - ld_cb) : a.kc;
    if (++ld_kk >= kc_cur) {
      ld_kk = 0;
      if (++ld_tap == ntaps) {
        ld_tap = 0;
        ld_cb += a.kc;
      }
      loader_tap();
    }
  };
  // DMA number d of a stage (0 .. ND-1): plane d % 2 of loader block d / 2
  auto issue_plane = [&](int d, int buf) {
    const int i = d / PL_PLANES, p = d - PL_PLANES * i;
    const unsigned so = __builtin_amdgcn_readfirstlane(ld_soff[i] + p * 512);
    const unsigned l = __builtin_amdgcn_readfirstlane(lds_mine[i] + buf * STAGE_BYTES + p * 1024);
    if (isA[i]) dma16(rsrcA, ld_voff[i], so, l);
    else dma16(rsrcB, ld_voff[i], so, l);
  };
  auto issue_stage = [&](int buf) {
#pragma unroll
    for (int d = 0; d < ND; ++d) issue_plane(d, buf);
    loader_next();
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  // two fragment register sets: the MFMAs of stage kt run on one while stage kt+1 is read into the other
  f16x8 fa[2][PL_PLANES][TM], fb[2][PL_PLANES][TN];
  auto read_frags = [&](int buf, auto SET) {
    constexpr int S = decltype(SET)::value;
    const unsigned char* sb = smem + buf * STAGE_BYTES + lane * 16;
#pragma unroll
    for (int p = 0; p < PL_PLANES; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[S][p][i] = *reinterpret_cast<const f16x8*>(sb + ((wm * TM + i) * PL_PLANES + p) * 1024);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[S][p][j] = *reinterpret_cast<const f16x8*>(sb + ((RBA + wn * TN + j) * PL_PLANES + p) * 1024);
    }
  };
  // three partial products, smallest first: l*h, h*l, h*h (planes: 0 = h, 1 = l). The two DMA instructions
  // of the stage being issued sit between the MFMAs: their issue slots are covered by the matrix pipe working
  // on the MFMAs already queued.
  auto mfma_stage = [&](auto SET, int wbuf) {
    constexpr int S = decltype(SET)::value;
    constexpr int NM = TM * TN * 3;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int pa = (q == 0) ? 1 : 0;
      const int pb = (q == 1) ? 1 : 0;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (!(DBG & 8)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[S][pa][i], fb[S][pb][j], acc[i][j], 0, 0, 0);
          const int idx = (q * TM + i) * TN + j;
#pragma unroll
          for (int d = 0; d < ND; ++d)
            if (idx == (((d + 1) * NM) / (ND + 1) > 0 ? ((d + 1) * NM) / (ND + 1) - 1 : 0)) {
              __builtin_amdgcn_sched_barrier(0);
              if constexpr (!(DBG & 1)) issue_plane(d, wbuf);
              __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    loader_next();
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  // prologue: stages 0..2 in flight (stages >= nk are dummies that read the zero block, so that every
  // wave always has exactly 2 DMAs per stage on its counter), stage 0's fragments in set 0
  loader_tap();
#pragma unroll
  for (int b = 0; b < NBUF; ++b) issue_stage(b);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 1) * ND) : "memory");   // the later stages may still fly
  __builtin_amdgcn_s_barrier();
  read_frags(0, S0{});

  // iteration kt: MFMAs of stage kt (registers) | fragment reads of stage kt+1 | DMA issue of stage kt+3
  // (into stage kt's buffer: everybody finished reading it before this iteration's barrier)
  auto step = [&](int rbuf, int wbuf, auto CUR, auto NXT) {
    // my pieces of stage kt+1 have landed (those of kt+2 may still fly)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * ND) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my reads of stage kt's buffer are done
    if constexpr (!(DBG & 4)) __builtin_amdgcn_s_barrier();
    if constexpr (!(DBG & 2)) read_frags(rbuf, NXT);
    __builtin_amdgcn_sched_barrier(0);
    mfma_stage(CUR, wbuf);
  };
  {
    int rbuf = 1, wbuf = 0;
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      step(rbuf, wbuf, S0{}, S1{});
      rbuf = (rbuf + 1 == NBUF) ? 0 : rbuf + 1;
      wbuf = (wbuf + 1 == NBUF) ? 0 : wbuf + 1;
      step(rbuf, wbuf, S1{}, S0{});
      rbuf = (rbuf + 1 == NBUF) ? 0 : rbuf + 1;
      wbuf = (wbuf + 1 == NBUF) ? 0 : wbuf + 1;
    }
    if (kt < nk) step(rbuf, wbuf, S0{}, S1{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dummy tail DMAs too
  __syncthreads();  // every wave is done with the stage buffers: the epilogue reuses them

  if constexpr (SPLIT)
    store_split_slab<TM, TN>(a, acc, BM * BN * 4, tile * SP + part, wave, lane);
  else if constexpr (MULTI)
    planes_epilogue<BM, BN, WGM, WGN, NBUF * STAGE_BYTES, 0, NoStamp, false, BNRED>(a, acc, smem, m0, n0, tile_m, wm, wn, lane, tid, NoStamp(), &G);
  else
    planes_epilogue<BM, BN, WGM, WGN, NBUF * STAGE_BYTES, (DBG & 31), NoStamp, false, BNRED>(a, acc, smem, m0, n0, tile_m, wm, wn, lane, tid);
}

template <int BM, int BN, int WGM, int WGN, int DBG = 0>
static int launch_planes(GatherConvArgs& a, hipStream_t st) {
  const long long tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.Cout + BN - 1) / BN;
  const long long nb = tiles_m * a.tiles_n;
  if (nb <= 0 || nb > 0x7fffffffLL) {
    set_error("conv(planes): bad grid %lld", nb);
    return YOLO_ERR_INVALID_ARG;
  }
  constexpr size_t lds = (3 + ((DBG >> 7) & 3)) * (BM / 32 + BN / 32) * PL_PLANES * 1024;
  if constexpr (DBG == 0) {
    if (a.ncls > 1) {   // the parity classes of a strided data gradient in one launch
      long long tm = 0;
      for (int c = 0; c < a.ncls; ++c) tm = std::max(tm, (a.cls[c].M + BM - 1) / BM);
      const long long nbm = tm * a.ncls * a.tiles_n;
      if (nbm <= 0 || nbm > 0x7fffffffLL) {
        set_error("conv(planes): bad grid %lld", nbm);
        return YOLO_ERR_INVALID_ARG;
      }
      a.nblocks = (int)nbm;
      a.split_parts = 1;
      a.bwd_nslots = (int)(tm * a.ncls);
      YOLO_BNRED_CHECK(a)
      static bool attr64 = false;
      if (!attr64) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_conv_planes_kernel<BM, BN, WGM, WGN, 64>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr64 = true;
      }
      if (a.bwd_y != nullptr) {
        static bool attr576 = false;
        if (!attr576) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_conv_planes_kernel<BM, BN, WGM, WGN, 64 | 512>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          attr576 = true;
        }
        hipLaunchKernelGGL((gather_conv_planes_kernel<BM, BN, WGM, WGN, 64 | 512>), dim3((unsigned)nbm), dim3(64 * WGM * WGN), lds, st, a);
        return check_launch("gather_conv_planes_kernel(classes, bn reduce)");
      }
      hipLaunchKernelGGL((gather_conv_planes_kernel<BM, BN, WGM, WGN, 64>), dim3((unsigned)nbm), dim3(64 * WGM * WGN), lds, st, a);
      return check_launch("gather_conv_planes_kernel(classes)");
    }
  }
  a.nblocks = (int)nb;
  a.bwd_nslots = (int)tiles_m;
  YOLO_BNRED_CHECK(a)
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_conv_planes_kernel<BM, BN, WGM, WGN, DBG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  a.split_parts = 1;
  if constexpr (BM == 128 && BN == 128 && WGM == 2 && WGN == 2 && DBG == 0) {
    // at least 8 stages per part (a.kc = the whole channel range in one chunk is the default stage order)
    const int min_cb = a.ntaps >= 8 ? 1 : (8 + a.ntaps - 1) / a.ntaps;
    a.split_parts = conv_split_parts(a, nb, BM, min_cb, 4);
    if (a.split_parts > 1) {
      a.sk_slabs = conv_split_slabs();
      static bool attr32 = false;
      if (!attr32) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_conv_planes_kernel<BM, BN, WGM, WGN, 32>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr32 = true;
      }
      hipLaunchKernelGGL((gather_conv_planes_kernel<BM, BN, WGM, WGN, 32>), dim3((unsigned)(nb * a.split_parts)),
                         dim3(64 * WGM * WGN), lds, st, a);
      if (int rc = check_launch("gather_conv_planes_kernel(split)")) return rc;
      return launch_split_reduce(a, BM, st);
    }
  }
  if constexpr ((DBG & ~(128 | 256)) == 0) {
    if (a.bwd_y != nullptr) {
      static bool attr_bn = false;
      if (!attr_bn) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_conv_planes_kernel<BM, BN, WGM, WGN, DBG | 512>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_bn = true;
      }
      hipLaunchKernelGGL((gather_conv_planes_kernel<BM, BN, WGM, WGN, DBG | 512>), dim3((unsigned)nb), dim3(64 * WGM * WGN), lds, st, a);
      return check_launch("gather_conv_planes_kernel(bn reduce)");
    }
  } else if (a.bwd_y != nullptr) {
    set_error("conv(planes): this diagnostic instantiation has no fused BatchNorm-backward reduction");
    return YOLO_ERR_INVALID_ARG;
  }
  hipLaunchKernelGGL((gather_conv_planes_kernel<BM, BN, WGM, WGN, DBG>), dim3((unsigned)nb), dim3(64 * WGM * WGN), lds, st, a);
  return check_launch("gather_conv_planes_kernel");
}

long long planes_bytes(long long rows, int C) { return planes_body_bytes(rows, C) + PL_HEADER; }

bool gather_planes_supported(const GatherConvArgs& a) { return (a.Cs % 16) == 0 && a.Cout >= 32; }

int launch_gather_planes(GatherConvArgs& a, hipStream_t st) {
  const long long rowsA = (long long)a.N * a.Hs * a.Ws;
  const long long bytesA = planes_bytes(rowsA, a.Cs), bytesB = planes_bytes(a.Cout, a.ldw);
  if (bytesA >= (1LL << 32) || bytesB >= (1LL << 32)) {
    set_error("conv(planes): operand planes exceed 4 GiB (%lld, %lld bytes)", bytesA, bytesB);
    return YOLO_ERR_INVALID_ARG;
  }
  a.src_bytes = (unsigned)bytesA;
  a.wgt_bytes = (unsigned)bytesB;
  a.zero_blk_src = (int)((rowsA + 15) / 16);
  a.zero_blk_wgt = (a.Cout + 15) / 16;
  static const int nt = [] { const char* e = getenv("YOLO_NT_STORE"); return e ? atoi(e) : 1; }();
  a.nt_store = nt;
  static const int vecst = [] { const char* e = getenv("YOLO_VEC_STORE"); return e ? atoi(e) : 1; }();
  // (the wave-private staging of planes_epilogue.hpp has no workgroup barriers: used at every size)
  a.vec_store = vecst;
  // diagnostic knock-outs of the main loop (wrong results): 1 no DMA, 2 no fragment reads, 4 no barrier, 8 no MFMA, 16 no output stores
  static const int dbg = [] { const char* e = getenv("YOLO_PLANES_DBG"); return e ? atoi(e) : 0; }();
  a.dbg = dbg;
  // channel-block chunk of the stage order (see the kernel): YOLO_PLANES_KC overrides
  static const int kc_env = [] { const char* e = getenv("YOLO_PLANES_KC"); return e ? atoi(e) : 0; }();
  a.kc = kc_env > 0 ? kc_env : (a.Cs >> 4);
  if (a.kc > (a.Cs >> 4)) a.kc = a.Cs >> 4;
  // 3x3 stride-1 forward / data gradient: the kernel that keeps the input window in LDS (conv_win.hip)
  init_options();
  if (g_opt[OPT_CONV_WIN] != 0 && a.ncls <= 1) {
    const int rc = launch_conv_win(a, g_opt[OPT_CONV_WIN], st);
    if (rc <= 0) return rc;
  }
  // 1x1 layers stream their operands once and, inside the training step, find them cold (behind kernels that left the
  // Infinity Cache full of dirty lines a 52x52 256->128 launch takes 50 us, on warm buffers 36): what bounds them is the
  // number of bytes in flight, so their DMA ring is deeper -- 5 stages (4 in flight, 80 KB, two workgroups per CU) under
  // the 128 x 128 tile, 4 under the narrower ones (a fifth costs them a resident workgroup). Same-box A/B of the whole
  // step: C3 30.81 -> 30.63 ms, C4 41.35 -> 40.84 ms. YOLO_PLANES_DEEP = 0 / 1 / 2 forces the number of extra stages.
  static const int deep_env = [] { const char* e = getenv("YOLO_PLANES_DEEP"); return e ? atoi(e) : -1; }();
  const int deep = deep_env >= 0 ? deep_env : (a.Cout <= 64 ? 1 : 2);
  if (deep && a.ntaps == 1 && a.ncls <= 1) {
    if (a.Cout <= 32) return deep == 1 ? launch_planes<128, 32, 4, 1, 128>(a, st) : launch_planes<128, 32, 4, 1, 256>(a, st);
    if (a.Cout <= 64) return deep == 1 ? launch_planes<128, 64, 4, 2, 128>(a, st) : launch_planes<128, 64, 4, 2, 256>(a, st);
  }
  if (a.Cout <= 32) return launch_planes<128, 32, 4, 1>(a, st);
  if (a.Cout <= 64) return launch_planes<128, 64, 4, 2>(a, st);
  // few row tiles (13x13 layers at bs 32: 43): 128x128 tiles leave CUs idle (172 tiles for a 512-channel data
  // gradient); the 128x64 tile doubles the workgroups. YOLO_PLANES_NARROW_BELOW = tile count under which it is used
  static const int narrow_below = [] { const char* e = getenv("YOLO_PLANES_NARROW_BELOW"); return e ? atoi(e) : 0; }();
  if (((a.M + 127) / 128) * ((a.Cout + 127) / 128) < narrow_below) return launch_planes<128, 64, 4, 2>(a, st);
  // 128x128 tile: 4 waves x (64x64) for the 3x3 launches this kernel still gets (stride 2, rows longer than 64 pixels),
  // 8 waves x (32x64) for 1x1 layers -- HBM-bound, they want loads in flight (24 waves per CU instead of 12), not MFMAs
  // per barrier. YOLO_PLANES_WAVES = 2 / 4 / 8 forces one form.
  static const int waves_env = [] { const char* e = getenv("YOLO_PLANES_WAVES"); return e ? atoi(e) : 0; }();
  const int waves = waves_env ? waves_env : (a.ntaps == 1 && a.ncls <= 1 ? 8 : 4);
#ifdef YOLO_PLANES_KNOCKOUTS   // diagnostic build (make KNOCKOUTS=1): compile-time knock-outs of the 128x128 4-wave kernel
  switch (a.dbg) {
    case 1: return launch_planes<128, 128, 2, 2, 1>(a, st);
    case 2: return launch_planes<128, 128, 2, 2, 2>(a, st);
    case 3: return launch_planes<128, 128, 2, 2, 3>(a, st);
    case 7: return launch_planes<128, 128, 2, 2, 7>(a, st);
    case 8: return launch_planes<128, 128, 2, 2, 8>(a, st);
    case 11: return launch_planes<128, 128, 2, 2, 11>(a, st);
    case 15: return launch_planes<128, 128, 2, 2, 15>(a, st);
    case 16: return launch_planes<128, 128, 2, 2, 16>(a, st);
    case 23: return launch_planes<128, 128, 2, 2, 23>(a, st);
    case 31: return launch_planes<128, 128, 2, 2, 31>(a, st);
    default: break;
  }
#endif
  if (waves == 2) return launch_planes<128, 128, 2, 1>(a, st);   // 2 waves x (64 x 128): one wave per SIMD, 512 registers
  if (waves == 4) {
    // (a.ncls > 1: the parity classes of a strided data gradient in one launch -- a.ntaps is class 0's there)
    if (deep && a.ntaps == 1 && a.ncls <= 1) return launch_planes<128, 128, 2, 2, 128>(a, st);
    return launch_planes<128, 128, 2, 2>(a, st);
  }
  {   // split-K (launches that leave the chip idle) lives in the 4-wave form
    const long long nb = ((a.M + 127) / 128) * ((a.Cout + 127) / 128);
    const int min_cb = a.ntaps >= 8 ? 1 : (8 + a.ntaps - 1) / a.ntaps;
    if (a.ncls <= 1 && conv_split_parts(a, nb, 128, min_cb, 4) > 1) return launch_planes<128, 128, 2, 2>(a, st);
  }
  if (deep == 1 && a.ntaps == 1 && a.ncls <= 1) return launch_planes<128, 128, 4, 2, 128>(a, st);
  if (deep == 2 && a.ntaps == 1 && a.ncls <= 1) return launch_planes<128, 128, 4, 2, 256>(a, st);
  return launch_planes<128, 128, 4, 2>(a, st);
}

int launch_split_planes_concat(const float* const* xs, const int* Cs, const float* const* bounds, int nsrc, long long rows,
                               void* planes, float* dst32, float* out_bound, hipStream_t st, const int* bound_words,
                               const int* upsample, int H, int W) {
  ConcatSrcs cs{};
  int C = 0;
  cs.H = H;
  cs.W = W;
  for (int i = 0; i < nsrc; ++i) {
    cs.x[i] = xs[i];
    cs.bound[i] = bounds[i];
    cs.bound_n[i] = bound_words != nullptr ? bound_words[i] : 1;
    cs.up[i] = upsample != nullptr ? upsample[i] : 0;
    cs.C[i] = Cs[i];
    C += Cs[i];
    cs.g_end[i] = C >> 3;
  }
  cs.n = nsrc;
  const long long rows_padded = ((rows + 15) / 16 + 1) * 16;   // data blocks + the all-zero block
  const int G = C >> 3;
  const long long blocks = ((rows_padded + 16 * SPLIT_BLOCKS_PER_WG - 1) / (16 * SPLIT_BLOCKS_PER_WG)) * ((G + 15) >> 4);
  if (blocks <= 0 || blocks > 0x7fffffffLL) {
    set_error("split_planes_concat: bad grid %lld", blocks);
    return YOLO_ERR_INVALID_ARG;
  }
  hipLaunchKernelGGL(split_planes_concat_kernel, dim3((unsigned)blocks), dim3(256), 0, st, cs, rows, C,
                     reinterpret_cast<unsigned char*>(planes), rows_padded, dst32, out_bound);
  return check_launch("split_planes_concat_kernel");
}

// ---- batched forms: one launch for all the filters of a network (150 launches of ~5 us otherwise) ----
// job = 6 x int64: {src pointer, dst pointer, a, b, c, first workgroup of the job in the launch}
struct BatchJob {
  const void* src;
  void* dst;
  long long a, b, c, first_block;
};
__device__ __forceinline__ int find_job(const BatchJob* jobs, int njobs, long long block) {
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {   // last job whose first_block <= block
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= block) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

// split job: a = rows, b = C, c = 0 or the header address of planes holding the same values (bound donor).
// Three launches over the same job table: clear the bound, max |x|, split.
__global__ void planes_header_clear_batch_kernel(const BatchJob* __restrict__ jobs, int njobs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= njobs) return;
  unsigned* header = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(jobs[i].dst) +
                                                 planes_body_bytes(jobs[i].a, (int)jobs[i].b));
  header[0] = 0u;
}
// Workgroup = YOLO_SPLIT_BATCH_UNITS consecutive 16-row blocks x 16 channel groups (128 channels): thread
// (r = tid & 15, gg = tid >> 4) reads 32 B of row r next to its neighbour groups' (128-B lines per row) and
// writes its two 16-B units next to the other 15 rows' (256-B sub-blocks): both sides move whole lines.
constexpr int BATCH_UNITS = YOLO_SPLIT_BATCH_UNITS;
struct BatchPos {
  long long row0;   // first row of this thread (+16 per unit)
  int g;            // 8-channel group
  bool live;
};
__device__ __forceinline__ BatchPos batch_pos(const BatchJob& j) {
  const int G = (int)j.b >> 3;
  const int gbn = (G + 15) >> 4;
  const long long lb = (long long)blockIdx.x - j.first_block;
  const long long rb = lb / gbn;
  const int gb = (int)(lb - rb * gbn);
  BatchPos p;
  p.g = gb * 16 + (threadIdx.x >> 4);
  p.row0 = rb * (16 * BATCH_UNITS) + (threadIdx.x & 15);
  p.live = p.g < G;
  return p;
}
__global__ __launch_bounds__(256) void planes_amax_batch_kernel(const BatchJob* __restrict__ jobs, int njobs) {
  const BatchJob j = jobs[find_job(jobs, njobs, blockIdx.x)];
  if (j.c != 0) return;   // the bound comes from another planes buffer of the same values (split kernel)
  const float* x = reinterpret_cast<const float*>(j.src);
  const BatchPos p = batch_pos(j);
  const int C = (int)j.b;
  f32x4 v[BATCH_UNITS][2];
#pragma unroll
  for (int u = 0; u < BATCH_UNITS; ++u) {
    const long long row = p.row0 + 16 * u;
    v[u][0] = v[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.live && row < j.a) {
      const float* q = x + row * C + p.g * 8;
      v[u][0] = *reinterpret_cast<const f32x4*>(q);
      v[u][1] = *reinterpret_cast<const f32x4*>(q + 4);
    }
  }
  float m = 0.f;
#pragma unroll
  for (int u = 0; u < BATCH_UNITS; ++u)
#pragma unroll
    for (int h = 0; h < 2; ++h)
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v[u][h][0]), fabsf(v[u][h][1])), fmaxf(fabsf(v[u][h][2]), fabsf(v[u][h][3]))));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f) {
    unsigned* header = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(j.dst) +
                                                   planes_body_bytes(j.a, (int)j.b));
    if (__builtin_bit_cast(unsigned, m) > *reinterpret_cast<volatile unsigned*>(header))
      atomicMax(header, __builtin_bit_cast(unsigned, m));
  }
}
__global__ __launch_bounds__(256) void split_planes_batch_kernel(const BatchJob* __restrict__ jobs, int njobs) {
  const BatchJob j = jobs[find_job(jobs, njobs, blockIdx.x)];
  const float* x = reinterpret_cast<const float*>(j.src);
  unsigned char* out = reinterpret_cast<unsigned char*>(j.dst);
  const long long rows = j.a;
  const int C = (int)j.b;
  unsigned* header = reinterpret_cast<unsigned*>(out + planes_body_bytes(rows, C));
  // c != 0: header of a planes buffer holding the same values in another order (a filter and its transpose)
  const unsigned bound = j.c != 0 ? *reinterpret_cast<const unsigned*>(j.c) : header[0];
  const float sc = planes_scale_from_bound(bound);
  const long long rows_padded = ((rows + 15) / 16 + 1) * 16;
  if ((long long)blockIdx.x == j.first_block && threadIdx.x == 0) {
    header[0] = bound;
    reinterpret_cast<float*>(header)[1] = sc;
    reinterpret_cast<float*>(header)[2] = 1.f / sc;
  }
  const BatchPos p = batch_pos(j);
  if (!p.live) return;
  f32x4 v[BATCH_UNITS][2];
#pragma unroll
  for (int u = 0; u < BATCH_UNITS; ++u) {
    const long long row = p.row0 + 16 * u;
    v[u][0] = v[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (row < rows) {
      const float* q = x + row * C + p.g * 8;
      v[u][0] = *reinterpret_cast<const f32x4*>(q);
      v[u][1] = *reinterpret_cast<const f32x4*>(q + 4);
    }
  }
#pragma unroll
  for (int u = 0; u < BATCH_UNITS; ++u) {
    const long long row = p.row0 + 16 * u;
    if (row >= rows_padded) continue;
    const Planes8 s8 = split8(v[u][0], v[u][1], sc);
    unsigned char* o = out + planes_unit_offset(row, p.g, C);
    *reinterpret_cast<u32x4*>(o) = s8.h;
    *reinterpret_cast<u32x4*>(o + 512) = s8.l;
  }
}

// transpose job: a = Cout, b = taps, c = Cin;  wT[ci][t][co] = w[co][t][ci]
__global__ __launch_bounds__(256) void filter_transpose_batch_kernel(const BatchJob* __restrict__ jobs, int njobs) {
  __shared__ float tile[32][33];
  const BatchJob j = jobs[find_job(jobs, njobs, blockIdx.x)];
  const float* w = reinterpret_cast<const float*>(j.src);
  float* wT = reinterpret_cast<float*>(j.dst);
  const int Cout = (int)j.a, taps = (int)j.b, Cin = (int)j.c;
  const int bx = (Cin + 31) / 32, by = (Cout + 31) / 32;
  const int lb = (int)((long long)blockIdx.x - j.first_block);
  const int t = lb / (bx * by);
  const int rem = lb - t * (bx * by);
  const int co0 = (rem / bx) * 32, ci0 = (rem % bx) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < Cout && ci < Cin) ? w[((long long)co * taps + t) * Cin + ci] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < Cin && co < Cout) wT[((long long)ci * taps + t) * Cout + co] = tile[tx][r];
  }
}

int launch_split_planes_batch(const void* jobs, int njobs, long long total_blocks, hipStream_t st) {
  if (njobs <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffffLL) {
    set_error("split_planes_batch: bad job table");
    return YOLO_ERR_INVALID_ARG;
  }
  const BatchJob* jb = reinterpret_cast<const BatchJob*>(jobs);
  hipLaunchKernelGGL(planes_header_clear_batch_kernel, dim3((njobs + 255) / 256), dim3(256), 0, st, jb, njobs);
  hipLaunchKernelGGL(planes_amax_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st, jb, njobs);
  hipLaunchKernelGGL(split_planes_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st, jb, njobs);
  return check_launch("split_planes_batch_kernel");
}

int launch_filter_transpose_batch(const void* jobs, int njobs, long long total_blocks, hipStream_t st) {
  if (njobs <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffffLL) {
    set_error("filter_transpose_batch: bad job table");
    return YOLO_ERR_INVALID_ARG;
  }
  hipLaunchKernelGGL(filter_transpose_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st,
                     reinterpret_cast<const BatchJob*>(jobs), njobs);
  return check_launch("filter_transpose_batch_kernel");
}

int launch_split_planes_padded(const float* x, long long rows, int Csrc, int C, void* planes, hipStream_t st) {
  if (C % 16 != 0 || rows <= 0 || Csrc <= 0 || Csrc > C || C - Csrc >= 16) {
    set_error("split_planes_padded: need 0 < Csrc <= C < Csrc + 16, C %% 16 == 0, rows > 0");
    return YOLO_ERR_INVALID_ARG;
  }
  const long long rows_padded = ((rows + 15) / 16 + 1) * 16;
  const long long blocks = (rows_padded * (C / 8) + 255) / 256;
  if (blocks > 0x7fffffffLL) {
    set_error("split_planes_padded: tensor too large");
    return YOLO_ERR_INVALID_ARG;
  }
  unsigned char* out = reinterpret_cast<unsigned char*>(planes);
  unsigned* header = reinterpret_cast<unsigned*>(out + planes_body_bytes(rows, C));
  if (hipMemsetAsync(header, 0, 16, st) != hipSuccess) {
    set_error("split_planes_padded: hipMemsetAsync failed");
    return YOLO_ERR_LAUNCH;
  }
  const long long n = rows * Csrc, n4 = n / 4;   // the dense source is read as float4 up to its last 0-3 values
  if (n4 > 0) hipLaunchKernelGGL(planes_amax_kernel, dim3(stream_grid(n4, 256)), dim3(256), 0, st, x, n4, header);
  if (n > 4 * n4)
    hipLaunchKernelGGL(planes_amax_scalar_kernel, dim3(1), dim3(256), 0, st, x + 4 * n4, n - 4 * n4, header);
  hipLaunchKernelGGL(split_planes_padded_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, rows, Csrc, C, out,
                     rows_padded);
  return check_launch("split_planes_padded_kernel");
}

int launch_split_planes_absmax(const float* x, long long rows, int C, const unsigned* absmax, const float* extra_bound, int extra_n,
                               void* planes, float* out_bound, hipStream_t st) {
  if (C % 16 != 0 || rows <= 0) {
    set_error("split_planes_absmax: C %% 16 != 0 or rows <= 0");
    return YOLO_ERR_INVALID_ARG;
  }
  const long long rows_padded = ((rows + 15) / 16 + 1) * 16;
  const long long blocks = ((rows_padded / 16 + SPLIT_BLOCKS_PER_WG - 1) / SPLIT_BLOCKS_PER_WG) * ((C / 8 + 15) / 16);
  if (blocks > 0x7fffffffLL) {
    set_error("split_planes_absmax: tensor too large");
    return YOLO_ERR_INVALID_ARG;
  }
  hipLaunchKernelGGL(split_planes_absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, rows, C, absmax, extra_bound,
                     extra_n, reinterpret_cast<unsigned char*>(planes), rows_padded, out_bound);
  return check_launch("split_planes_absmax_kernel");
}

int launch_split_planes(const float* x, long long rows, int C, void* planes, hipStream_t st) {
  if (C % 16 != 0 || rows <= 0) {
    set_error("split_planes: C %% 16 != 0 or rows <= 0");
    return YOLO_ERR_INVALID_ARG;
  }
  const long long rows_padded = ((rows + 15) / 16 + 1) * 16;
  const long long blocks = ((rows_padded / 16 + SPLIT_BLOCKS_PER_WG - 1) / SPLIT_BLOCKS_PER_WG) * ((C / 8 + 15) / 16);
  if (blocks > 0x7fffffffLL) {
    set_error("split_planes: tensor too large");
    return YOLO_ERR_INVALID_ARG;
  }
  unsigned char* out = reinterpret_cast<unsigned char*>(planes);
  unsigned* header = reinterpret_cast<unsigned*>(out + planes_body_bytes(rows, C));
  if (hipMemsetAsync(header, 0, 16, st) != hipSuccess) {
    set_error("split_planes: hipMemsetAsync failed");
    return YOLO_ERR_LAUNCH;
  }
  const long long n4 = rows * (C / 4);
  hipLaunchKernelGGL(planes_amax_kernel, dim3(stream_grid(n4, 256)), dim3(256), 0, st, x, n4, header);
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, rows, C, out, rows_padded);
  return check_launch("split_planes_kernel");
}

}  // namespace yolo

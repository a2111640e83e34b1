// Filter gradient of 3x3 stride-1 "same" convolutions with the INPUT WINDOW kept in LDS (round 4).
//   dW[co][t][ci] += sum_p dy[p][co] * x[p + dy_t * W + dx_t][ci]        (zero outside the image)
// wgrad_planes_kernel (conv_wgrad_planes.hip) gives a workgroup 128 filters x 128 columns of ONE tap and streams, for
// every one of the nine taps, the shifted pixels of x from L2 again: 341 B of operands per MFMA through L2 -> LDS (1.56 GB per
// launch on 52x52x128->256), and its own knock-outs show the DMA side as the largest single cost (159 -> 97 us without it).
// Here a workgroup owns 128 filters x (9 taps x 32 input channels). The contraction runs over pixels in linear order, so the
// x rows a tap needs are the SAME rows shifted by dy*W + dx pixels: x is streamed ONCE through a ring of RING pixel slots in
// LDS (32 new pixels per 32-pixel stage) and every tap's B fragment is a transposing read (ds_read_b64_tr_b16) at
// "my pixel + tap shift" in that ring; image borders are handled by pointing the lanes of an invalid (pixel, tap) pair at an
// all-zero slot. Per stage a workgroup moves 16 KiB of dy + 4 KiB of x for 216 MFMAs: 95 B per MFMA (3.6x less).
//
//   wave w: filters co0 + 32 w .. + 31, all nine taps of the 32 channels: nine 32x32 accumulators (144 registers),
//           two waves per SIMD, two workgroups per CU.
//   dy (A): staged exactly as in wgrad_planes_kernel (pieces of 16 pixels x 32 filters of one plane, lane-linear image
//           [pixel quad][sub-block][pixel], conflict-free transposed reads), two stage buffers.
//   x  (B): ring [plane][slot = pixel mod RING][64 B = 32 channels]; a DMA piece = 16 pixels of one plane; the four pixels of
//           a transposed read are consecutive slots = 256 contiguous bytes: conflict-free at any tap shift.
// One barrier per 32-pixel stage (54 MFMAs per wave), loads of stage s+1 issued at the head of stage s.
// The partial of every workgroup goes to its slab (accumulator order) and wgrad_win_reduce_kernel adds the splits in order:
// bit-reproducible like the per-tap kernel; without a registered workspace the epilogue falls back to fp32 atomics.
#include "planes.hpp"
#include <cstdlib>

namespace yolo {

typedef short ws16x4 __attribute__((ext_vector_type(4)));
typedef short ws16x8 __attribute__((ext_vector_type(8)));

constexpr int WW_CO = 128;       // filters per workgroup
constexpr int WW_CI = 32;        // input channels per workgroup (all nine taps)
constexpr int WW_STAGE_PX = 32;  // pixels per stage (two MFMA k-steps)
constexpr int WW_TILE_FLOATS = WW_CO * 9 * WW_CI;

// KO: diagnostic knock-outs (wrong results): 1 = no DMAs in the loop, 2 = no fragment reads, 4 = no MFMAs, 8 = no border masks
template <int RING, int KO = 0, int PF = 1>
__global__ __launch_bounds__(256, 2) void wgrad_win_kernel(const WgradArgs a) {
  constexpr int A_STAGE = 2 * 4 * PL_PLANES * 1024;   // [k-step][filter block][plane][1 KiB piece]
  constexpr int WIN_PLANE = RING * 64 + 64;           // ring + one all-zero slot
  constexpr int WIN_OFF = 2 * A_STAGE;
  constexpr unsigned RMASK = RING * 64 - 1;
  constexpr unsigned ZSLOT = RING * 64;               // byte offset of the zero slot inside a plane of the window

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds_base = (unsigned)(size_t)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int lid = xcd_remap(blockIdx.x, a.nblocks);
  const int tiles = a.tiles_co * a.tiles_j;
  const int split = lid / tiles;
  const int tile = lid - split * tiles;
  const int co0 = (tile % a.tiles_co) * WW_CO;
  const int ci0 = (tile / a.tiles_co) * WW_CI;
  const int p_begin = (int)((long long)split * a.chunk);   // multiple of 32
  int p_end = p_begin + (int)a.chunk;
  if (p_end > (int)a.M) p_end = (int)a.M;
  if (p_begin >= p_end) return;
  const int nst = (p_end - p_begin + WW_STAGE_PX - 1) / WW_STAGE_PX;
  const int W = a.Ws, H = a.Hs, M = (int)a.M;

  // (the window reads below address LDS from 0: launch_ww checks on the host that the kernel has no static LDS in front of it)
  // zero slots of the two window planes
  if (tid < 32) reinterpret_cast<unsigned*>(smem + WIN_OFF + (tid >> 4) * WIN_PLANE + ZSLOT)[tid & 15] = 0u;

  // ---- loader roles ----
  // dy: this wave's 32-filter block, pieces (k-step j, plane): lane -> pixel 4 (l >> 4) + (l & 3), sub-block (l >> 2) & 3
  const unsigned strideA = (unsigned)((a.Cout >> 4) * PL_RECORD);
  const unsigned strideB = (unsigned)((a.Cs >> 4) * PL_RECORD);
  const unsigned zeroA = (unsigned)a.zero_blk_dy * strideA, zeroB = (unsigned)a.zero_blk_src * strideB;
  const i32x4 rsrcA = planes_rsrc(a.dy, a.dy_bytes), rsrcB = planes_rsrc(a.src, a.src_bytes);
  const int lpix = 4 * (lane >> 4) + (lane & 3), lsb = (lane >> 2) & 3;
  const unsigned a_lane = (unsigned)(((co0 + wave * 32) >> 4) + (lsb >> 1)) * PL_RECORD + (lsb & 1) * 256 + lpix * 16;
  auto issue_A = [&](int stage, int buf) {   // 4 DMAs: the two 16-pixel blocks of the stage, both planes
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pb16 = p_begin + stage * WW_STAGE_PX + j * 16;
      const unsigned v = (pb16 < p_end) ? (unsigned)(pb16 >> 4) * strideA + a_lane : zeroA;
#pragma unroll
      for (int p = 0; p < PL_PLANES; ++p) {
        const unsigned l = __builtin_amdgcn_readfirstlane(lds_base + buf * A_STAGE + ((j * 4 + wave) * PL_PLANES + p) * 1024);
        dma16(rsrcA, v, (unsigned)(p * 512), l);
      }
    }
  };
  // x: one piece = 16 consecutive pixels of one plane of the 32 channels: lane -> pixel l >> 2, 8-channel chunk l & 3
  const int bpx = lane >> 2, bch = lane & 3;
  const unsigned b_lane = (unsigned)((ci0 >> 4) + (bch >> 1)) * PL_RECORD + (bch & 1) * 256;
  auto issue_B = [&](int P0, int plane) {   // P0: first pixel of the group (multiple of 16, may be < 0 or >= M)
    const int P = P0 + bpx;
    const unsigned v = ((unsigned)P < (unsigned)M) ? (unsigned)(P >> 4) * strideB + b_lane + (P & 15) * 16 : zeroB;
    const unsigned l = __builtin_amdgcn_readfirstlane(lds_base + WIN_OFF + plane * WIN_PLANE + (unsigned)((P0 & (RING - 1)) * 64));
    dma16(rsrcB, v, (unsigned)(plane * 512), l);
  };

  // ---- prologue: the window [lo, Lbase) and dy of stage 0 ----
  const int lo = (p_begin - W - 1) & ~15;
  const int Lbase = (p_begin + W + 33 + 15) & ~15;   // the loads issued in stage s (32 pixels from Lbase + 32 s) complete stage s + 1
  {
    const int npieces = ((Lbase - lo) >> 4) * PL_PLANES;
    for (int idx = wave; idx < npieces; idx += 4) issue_B(lo + (idx >> 1) * 16, idx & 1);
  }
  issue_A(0, 0);

  // ---- reader roles ----
  // transposed reads (as wgrad_planes_kernel): 16-lane group g -> channel half g & 1, pixel half g >> 1; lane (qq, pq) of the
  // group supplies pixel qq of the read, channels 4 pq .. 4 pq + 3 of the half
  const int grp = lane >> 4, i16 = lane & 15;
  const int qq = i16 >> 2, pq = i16 & 3;
  const int tr_off = ((2 * (grp >> 1)) * 16 + ((grp & 1) * 2 + (pq >> 1)) * 4 + qq) * 16 + (pq & 1) * 8;   // dy pieces
  const unsigned lc = (unsigned)((2 * (grp & 1) + (pq >> 1)) * 16 + (pq & 1) * 8);                        // inside a window slot
  typedef ws16x4 __attribute__((address_space(3))) * lds_p;
  // the lane's four pixels of a stage: k-step j, read r -> pixel p0 + 16 j + 8 (grp >> 1) + 4 r + qq
  unsigned pb[2][2];   // ring byte offset of the pixel's slot + lc
  int px[2][2], py[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int p = p_begin + 16 * j + 8 * (grp >> 1) + 4 * r + qq;
      pb[j][r] = (((unsigned)p & (RING - 1)) << 6) + lc;
      const int row = p / W;
      px[j][r] = p - row * W;
      py[j][r] = row % H;
    }
  const int adv_rows = WW_STAGE_PX / W, adv_rem = WW_STAGE_PX - adv_rows * W;   // 32 pixels = adv_rows rows + adv_rem pixels
  int shb[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) shb[t] = ((t / 3 - 1) * W + (t % 3 - 1)) * 64;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

  // (lgkmcnt too: the zero slot above is an LDS write, and a raw s_barrier carries no implicit wait for it)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // The stage body is scheduled by hand (sched_barrier(0) between the groups: hipcc otherwise issues a tap's four fragment
  // reads right in front of its MFMAs and waits for them): per tap-step i = (k-step j, tap t)
  //   MFMA 1 of i | the 4 fragment reads of i + 1 | MFMA 2 of i | the addresses of i + 2 (VALU) | MFMA 3 of i
  // two fragment register sets, two address sets; the dy fragments of k-step 1 are read during tap-step 6.
  // (dynamic LDS starts at address 0 in a kernel without static LDS: the window reads take WIN_OFF as an immediate offset)
  auto rd_pair = [&](unsigned a0, unsigned a1, int off) -> f16x8 {
    const ws16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(size_t)(a0 + off));
    const ws16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(size_t)(a1 + off));
    const ws16x8 v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    return __builtin_bit_cast(f16x8, v);
  };
  const unsigned zl = ZSLOT + lc;
  for (int s = 0; s < nst; ++s) {
    if constexpr (!(KO & 1)) {
      issue_A(s + 1, (s + 1) & 1);                                   // (past the end: the zero block)
      issue_B(Lbase + s * WW_STAGE_PX + (wave >> 1) * 16, wave & 1);
    }
    const unsigned abase = (unsigned)((s & 1) * A_STAGE + wave * PL_PLANES * 1024 + tr_off);
    bool okx0[2][2], okx1[2][2], oky0[2][2], oky1[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        okx0[j][r] = px[j][r] != 0;
        okx1[j][r] = px[j][r] != W - 1;
        oky0[j][r] = py[j][r] != 0;
        oky1[j][r] = py[j][r] != H - 1;
      }
    auto calc = [&](int i, unsigned (&ad)[2]) {   // ring offsets of tap-step i's two reads (the zero slot where the tap leaves the image)
      const int j = i / 9, t = i % 9, ty = t / 3, tx = t % 3;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        bool ok = true;
        if (ty == 0) ok = ok && oky0[j][r];
        if (ty == 2) ok = ok && oky1[j][r];
        if (tx == 0) ok = ok && okx0[j][r];
        if (tx == 2) ok = ok && okx1[j][r];
        const unsigned av = (pb[j][r] + (unsigned)shb[t]) & RMASK;
        ad[r] = ((KO & 8) || ok) ? av : zl;
      }
    };
    constexpr int NS = PF + 1;   // fragment / address register sets: the reads of tap-step i + PF are issued during tap-step i
    f16x8 ah[2], al[2], bh[NS], bl[NS];
    unsigned ad[NS][2];
#pragma unroll
    for (int k = 0; k <= PF; ++k) calc(k, ad[k]);
    if constexpr (!(KO & 2)) {
      ah[0] = rd_pair(abase, abase + 256, 0);
      al[0] = rd_pair(abase, abase + 256, 1024);
#pragma unroll
      for (int k = 0; k < PF; ++k) {
        bh[k] = rd_pair(ad[k][0], ad[k][1], WIN_OFF);
        bl[k] = rd_pair(ad[k][0], ad[k][1], WIN_OFF + WIN_PLANE);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      const int j = i / 9, t = i % 9, cur = i % NS, nxt = (i + PF) % NS;
      if constexpr (!(KO & 4)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[j], bh[cur], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(KO & 2)) {
        if (i + PF < 18) {
          bh[nxt] = rd_pair(ad[nxt][0], ad[nxt][1], WIN_OFF);
          bl[nxt] = rd_pair(ad[nxt][0], ad[nxt][1], WIN_OFF + WIN_PLANE);
        }
        if (i == 6 - PF) {   // dy fragments of the second k-step
          ah[1] = rd_pair(abase + 4 * PL_PLANES * 1024, abase + 4 * PL_PLANES * 1024 + 256, 0);
          al[1] = rd_pair(abase + 4 * PL_PLANES * 1024, abase + 4 * PL_PLANES * 1024 + 256, 1024);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(KO & 4)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], bl[cur], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (i + PF + 1 < 18) calc(i + PF + 1, ad[cur]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(KO & 4)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], bh[cur], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (i >= 16) {
        // the lane's pixels of the next stage: 32 pixels on = adv_rows rows + adv_rem pixels (k-step i - 16)
        const int jj = i - 16;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          pb[jj][r] = (pb[jj][r] + WW_STAGE_PX * 64) & RMASK;
          int x = px[jj][r] + adv_rem, y = py[jj][r] + adv_rows;
          if (x >= W) {
            x -= W;
            ++y;
          }
          if (y >= H) y -= H;
          px[jj][r] = x;
          py[jj][r] = y;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // my pieces of stage s + 1 have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my reads of stage s are done
    __builtin_amdgcn_s_barrier();
  }

  const float unscale =
      reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.dy) + a.dy_bytes - PL_HEADER)[2] *
      reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.src) + a.src_bytes - PL_HEADER)[2];
  if (a.slabs != nullptr) {
    // piece ((wave * 9 + t) * 4 + q4) of lane l at float (piece * 64 + l) * 4: one store instruction = 1 KiB contiguous
    float* mine = a.slabs + (size_t)lid * WW_TILE_FLOATS;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 f = {acc[t][4 * q4], acc[t][4 * q4 + 1], acc[t][4 * q4 + 2], acc[t][4 * q4 + 3]};
        *reinterpret_cast<f32x4*>(mine + (((wave * 9 + t) * 4 + q4) * 64 + lane) * 4) = f;
      }
    return;
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const long long cj = (long long)t * a.Cs + ci0 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      atomicAdd(&a.dw[(long long)co * a.ldw + cj], acc[t][r] * unscale);
    }
  }
}

// ---- the same kernel with TWO pixel halves per workgroup (round 6) ----
// Every workgroup of wgrad_win_kernel ends by storing its 128 x 288 fp32 partial (147 KB) to a slab, and
// wgrad_win_reduce_kernel reads all of them back: 75 MB written + 85 MB read per launch whatever the layer (512 workgroups x
// 147 KB), 32 of a launch's 166 us with the reduce (knock-outs, profiles/r04_a_wgrad_win_knockouts.log). Here a workgroup has
// EIGHT waves: waves 0-3 and waves 4-7 are two copies of the 4-wave kernel -- own dy stage buffers, own x ring, own half of the
// workgroup's pixel chunk -- that run in lockstep (one s_barrier per stage for all eight; both halves run the same number of
// stages, a half past its pixels multiplies zero blocks). At the end the upper half hands its accumulators over through LDS
// (147 KB of the CU's 160), the lower half adds them (fixed order) and stores ONE slab: half the workgroups (one per CU, the
// same two waves per SIMD), half the slab bytes, half the reduce. Bit-reproducible; the sums differ from the 4-wave kernel's
// in their last bits (pairs of splits are added first).
template <int RING>
__global__ __launch_bounds__(512, 1) void wgrad_win2_kernel(const WgradArgs a) {
  constexpr int KO = 0, PF = 1;
  constexpr int A_STAGE = 2 * 4 * PL_PLANES * 1024;   // [k-step][filter block][plane][1 KiB piece]
  constexpr int WIN_PLANE = RING * 64 + 64;           // ring + one all-zero slot
  constexpr int WIN_OFF = 2 * A_STAGE;
  constexpr int HALF_BYTES = 2 * A_STAGE + 2 * WIN_PLANE;   // one half's LDS: dy stages + the two planes of its x ring
  constexpr unsigned RMASK = RING * 64 - 1;
  constexpr unsigned ZSLOT = RING * 64;               // byte offset of the zero slot inside a plane of the window

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds_base = (unsigned)(size_t)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave8 >> 2, wave = wave8 & 3;         // (wave: the role inside the half, as in the 4-wave kernel)
  const unsigned hbase = (unsigned)(half * HALF_BYTES);   // this half's LDS region

  const int lid = xcd_remap(blockIdx.x, a.nblocks);
  const int tiles = a.tiles_co * a.tiles_j;
  const int split = lid / tiles;
  const int tile = lid - split * tiles;
  const int co0 = (tile % a.tiles_co) * WW_CO;
  const int ci0 = (tile / a.tiles_co) * WW_CI;
  // a.chunk = the WORKGROUP's pixels (a multiple of 64); half h takes [h * chunk / 2, (h + 1) * chunk / 2) of them
  const int hchunk = (int)(a.chunk >> 1);
  if ((long long)split * a.chunk >= a.M) return;         // (whole workgroup: never happens with launch_ww2's splits)
  const int p_begin = (int)((long long)split * a.chunk) + half * hchunk;   // multiple of 32
  int p_end = p_begin + hchunk;
  if (p_end > (int)a.M) p_end = (int)a.M;
  if (p_end < p_begin) p_end = p_begin;                  // an upper half past the last pixel: zero blocks only
  const int nst = hchunk / WW_STAGE_PX;                  // BOTH halves run the same stages (one barrier each)
  const int W = a.Ws, H = a.Hs, M = (int)a.M;

  // (the window reads below address LDS from 0: launch_ww checks on the host that the kernel has no static LDS in front of it)
  // zero slots of the two window planes
  if ((tid & 255) < 32)
    reinterpret_cast<unsigned*>(smem + hbase + WIN_OFF + ((tid & 255) >> 4) * WIN_PLANE + ZSLOT)[tid & 15] = 0u;

  // ---- loader roles ----
  // dy: this wave's 32-filter block, pieces (k-step j, plane): lane -> pixel 4 (l >> 4) + (l & 3), sub-block (l >> 2) & 3
  const unsigned strideA = (unsigned)((a.Cout >> 4) * PL_RECORD);
  const unsigned strideB = (unsigned)((a.Cs >> 4) * PL_RECORD);
  const unsigned zeroA = (unsigned)a.zero_blk_dy * strideA, zeroB = (unsigned)a.zero_blk_src * strideB;
  const i32x4 rsrcA = planes_rsrc(a.dy, a.dy_bytes), rsrcB = planes_rsrc(a.src, a.src_bytes);
  const int lpix = 4 * (lane >> 4) + (lane & 3), lsb = (lane >> 2) & 3;
  const unsigned a_lane = (unsigned)(((co0 + wave * 32) >> 4) + (lsb >> 1)) * PL_RECORD + (lsb & 1) * 256 + lpix * 16;
  auto issue_A = [&](int stage, int buf) {   // 4 DMAs: the two 16-pixel blocks of the stage, both planes
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pb16 = p_begin + stage * WW_STAGE_PX + j * 16;
      const unsigned v = (pb16 < p_end) ? (unsigned)(pb16 >> 4) * strideA + a_lane : zeroA;
#pragma unroll
      for (int p = 0; p < PL_PLANES; ++p) {
        const unsigned l = __builtin_amdgcn_readfirstlane(lds_base + hbase + buf * A_STAGE + ((j * 4 + wave) * PL_PLANES + p) * 1024);
        dma16(rsrcA, v, (unsigned)(p * 512), l);
      }
    }
  };
  // x: one piece = 16 consecutive pixels of one plane of the 32 channels: lane -> pixel l >> 2, 8-channel chunk l & 3
  const int bpx = lane >> 2, bch = lane & 3;
  const unsigned b_lane = (unsigned)((ci0 >> 4) + (bch >> 1)) * PL_RECORD + (bch & 1) * 256;
  auto issue_B = [&](int P0, int plane) {   // P0: first pixel of the group (multiple of 16, may be < 0 or >= M)
    const int P = P0 + bpx;
    const unsigned v = ((unsigned)P < (unsigned)M) ? (unsigned)(P >> 4) * strideB + b_lane + (P & 15) * 16 : zeroB;
    const unsigned l = __builtin_amdgcn_readfirstlane(lds_base + hbase + WIN_OFF + plane * WIN_PLANE + (unsigned)((P0 & (RING - 1)) * 64));
    dma16(rsrcB, v, (unsigned)(plane * 512), l);
  };

  // ---- prologue: the window [lo, Lbase) and dy of stage 0 ----
  const int lo = (p_begin - W - 1) & ~15;
  const int Lbase = (p_begin + W + 33 + 15) & ~15;   // the loads issued in stage s (32 pixels from Lbase + 32 s) complete stage s + 1
  {
    const int npieces = ((Lbase - lo) >> 4) * PL_PLANES;
    for (int idx = wave; idx < npieces; idx += 4) issue_B(lo + (idx >> 1) * 16, idx & 1);
  }
  issue_A(0, 0);

  // ---- reader roles ----
  // transposed reads (as wgrad_planes_kernel): 16-lane group g -> channel half g & 1, pixel half g >> 1; lane (qq, pq) of the
  // group supplies pixel qq of the read, channels 4 pq .. 4 pq + 3 of the half
  const int grp = lane >> 4, i16 = lane & 15;
  const int qq = i16 >> 2, pq = i16 & 3;
  const int tr_off = ((2 * (grp >> 1)) * 16 + ((grp & 1) * 2 + (pq >> 1)) * 4 + qq) * 16 + (pq & 1) * 8;   // dy pieces
  const unsigned lc = (unsigned)((2 * (grp & 1) + (pq >> 1)) * 16 + (pq & 1) * 8);                        // inside a window slot
  typedef ws16x4 __attribute__((address_space(3))) * lds_p;
  // the lane's four pixels of a stage: k-step j, read r -> pixel p0 + 16 j + 8 (grp >> 1) + 4 r + qq
  unsigned pb[2][2];   // ring byte offset of the pixel's slot + lc
  int px[2][2], py[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int p = p_begin + 16 * j + 8 * (grp >> 1) + 4 * r + qq;
      pb[j][r] = (((unsigned)p & (RING - 1)) << 6) + lc;
      const int row = p / W;
      px[j][r] = p - row * W;
      py[j][r] = row % H;
    }
  const int adv_rows = WW_STAGE_PX / W, adv_rem = WW_STAGE_PX - adv_rows * W;   // 32 pixels = adv_rows rows + adv_rem pixels
  int shb[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) shb[t] = ((t / 3 - 1) * W + (t % 3 - 1)) * 64;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

  // (lgkmcnt too: the zero slot above is an LDS write, and a raw s_barrier carries no implicit wait for it)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // The stage body is scheduled by hand (sched_barrier(0) between the groups: hipcc otherwise issues a tap's four fragment
  // reads right in front of its MFMAs and waits for them): per tap-step i = (k-step j, tap t)
  //   MFMA 1 of i | the 4 fragment reads of i + 1 | MFMA 2 of i | the addresses of i + 2 (VALU) | MFMA 3 of i
  // two fragment register sets, two address sets; the dy fragments of k-step 1 are read during tap-step 6.
  // (dynamic LDS starts at address 0 in a kernel without static LDS: the window reads take WIN_OFF as an immediate offset)
  auto rd_pair = [&](unsigned a0, unsigned a1, int off) -> f16x8 {
    const ws16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(size_t)(a0 + off));
    const ws16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(size_t)(a1 + off));
    const ws16x8 v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    return __builtin_bit_cast(f16x8, v);
  };
  const unsigned zl = ZSLOT + lc + hbase;
  for (int s = 0; s < nst; ++s) {
    if constexpr (!(KO & 1)) {
      issue_A(s + 1, (s + 1) & 1);                                   // (past the end: the zero block)
      issue_B(Lbase + s * WW_STAGE_PX + (wave >> 1) * 16, wave & 1);
    }
    const unsigned abase = hbase + (unsigned)((s & 1) * A_STAGE + wave * PL_PLANES * 1024 + tr_off);
    bool okx0[2][2], okx1[2][2], oky0[2][2], oky1[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        okx0[j][r] = px[j][r] != 0;
        okx1[j][r] = px[j][r] != W - 1;
        oky0[j][r] = py[j][r] != 0;
        oky1[j][r] = py[j][r] != H - 1;
      }
    auto calc = [&](int i, unsigned (&ad)[2]) {   // ring offsets of tap-step i's two reads (the zero slot where the tap leaves the image)
      const int j = i / 9, t = i % 9, ty = t / 3, tx = t % 3;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        bool ok = true;
        if (ty == 0) ok = ok && oky0[j][r];
        if (ty == 2) ok = ok && oky1[j][r];
        if (tx == 0) ok = ok && okx0[j][r];
        if (tx == 2) ok = ok && okx1[j][r];
        const unsigned av = ((pb[j][r] + (unsigned)shb[t]) & RMASK) + hbase;
        ad[r] = ((KO & 8) || ok) ? av : zl;
      }
    };
    constexpr int NS = PF + 1;   // fragment / address register sets: the reads of tap-step i + PF are issued during tap-step i
    f16x8 ah[2], al[2], bh[NS], bl[NS];
    unsigned ad[NS][2];
#pragma unroll
    for (int k = 0; k <= PF; ++k) calc(k, ad[k]);
    if constexpr (!(KO & 2)) {
      ah[0] = rd_pair(abase, abase + 256, 0);
      al[0] = rd_pair(abase, abase + 256, 1024);
#pragma unroll
      for (int k = 0; k < PF; ++k) {
        bh[k] = rd_pair(ad[k][0], ad[k][1], WIN_OFF);
        bl[k] = rd_pair(ad[k][0], ad[k][1], WIN_OFF + WIN_PLANE);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      const int j = i / 9, t = i % 9, cur = i % NS, nxt = (i + PF) % NS;
      if constexpr (!(KO & 4)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[j], bh[cur], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(KO & 2)) {
        if (i + PF < 18) {
          bh[nxt] = rd_pair(ad[nxt][0], ad[nxt][1], WIN_OFF);
          bl[nxt] = rd_pair(ad[nxt][0], ad[nxt][1], WIN_OFF + WIN_PLANE);
        }
        if (i == 6 - PF) {   // dy fragments of the second k-step
          ah[1] = rd_pair(abase + 4 * PL_PLANES * 1024, abase + 4 * PL_PLANES * 1024 + 256, 0);
          al[1] = rd_pair(abase + 4 * PL_PLANES * 1024, abase + 4 * PL_PLANES * 1024 + 256, 1024);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(KO & 4)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], bl[cur], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (i + PF + 1 < 18) calc(i + PF + 1, ad[cur]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(KO & 4)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], bh[cur], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (i >= 16) {
        // the lane's pixels of the next stage: 32 pixels on = adv_rows rows + adv_rem pixels (k-step i - 16)
        const int jj = i - 16;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          pb[jj][r] = (pb[jj][r] + WW_STAGE_PX * 64) & RMASK;
          int x = px[jj][r] + adv_rem, y = py[jj][r] + adv_rows;
          if (x >= W) {
            x -= W;
            ++y;
          }
          if (y >= H) y -= H;
          px[jj][r] = x;
          py[jj][r] = y;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // my pieces of stage s + 1 have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my reads of stage s are done
    __builtin_amdgcn_s_barrier();
  }

  // ---- the upper half's accumulators through LDS (piece ((wave * 9 + t) * 4 + q4) of lane l at (piece * 64 + l) * 16 bytes:
  // 147 456 bytes from 0 -- everybody is behind the loop's last barrier, the stage buffers and rings are dead) ----
  if (half == 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 f = {acc[t][4 * q4], acc[t][4 * q4 + 1], acc[t][4 * q4 + 2], acc[t][4 * q4 + 3]};
        *reinterpret_cast<f32x4*>(smem + ((((wave * 9 + t) * 4 + q4) * 64 + lane) * 16)) = f;
      }
  }
  __syncthreads();
  if (half == 1) return;
  if (a.slabs != nullptr) {
    float* mine = a.slabs + (size_t)lid * WW_TILE_FLOATS;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(smem + ((((wave * 9 + t) * 4 + q4) * 64 + lane) * 16));
        const f32x4 f = {acc[t][4 * q4] + u[0], acc[t][4 * q4 + 1] + u[1], acc[t][4 * q4 + 2] + u[2], acc[t][4 * q4 + 3] + u[3]};
        *reinterpret_cast<f32x4*>(mine + (((wave * 9 + t) * 4 + q4) * 64 + lane) * 4) = f;
      }
    return;
  }
  const float unscale =
      reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.dy) + a.dy_bytes - PL_HEADER)[2] *
      reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.src) + a.src_bytes - PL_HEADER)[2];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const long long cj = (long long)t * a.Cs + ci0 + (lane & 31);
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const f32x4 u = *reinterpret_cast<const f32x4*>(smem + ((((wave * 9 + t) * 4 + q4) * 64 + lane) * 16));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * q4 + e;
        const int co = co0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        atomicAdd(&a.dw[(long long)co * a.ldw + cj], (acc[t][r] + u[e]) * unscale);
      }
    }
  }
}

// dw += unscale * sum over splits (in split order) of the slabs' partials: one thread per 16-byte accumulator piece
__global__ __launch_bounds__(256) void wgrad_win_reduce_kernel(const WgradArgs a) {
  constexpr int PER_TILE = WW_TILE_FLOATS / 4;
  const int tiles = a.tiles_co * a.tiles_j;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const int tile = (int)(idx / PER_TILE);
  if (tile >= tiles) return;
  const int q = (int)(idx - (long long)tile * PER_TILE);
  const int lane = q & 63, piece = q >> 6;
  const int q4 = piece & 3, t = (piece >> 2) % 9, wave = piece / 36;
  const int co = (tile % a.tiles_co) * WW_CO + wave * 32 + 8 * q4 + 4 * (lane >> 5);
  const long long cj = (long long)t * a.Cs + (tile / a.tiles_co) * WW_CI + (lane & 31);
  const f32x4* sl = reinterpret_cast<const f32x4*>(a.slabs) + (size_t)tile * PER_TILE + q;
  const size_t sstride = (size_t)tiles * PER_TILE;
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 8 <= a.splits; s += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(sl + (size_t)(s + u) * sstride);
#pragma unroll
    for (int u = 0; u < 8; ++u) sum = sum + v[u];
  }
  for (; s < a.splits; ++s) sum = sum + __builtin_nontemporal_load(sl + (size_t)s * sstride);
  const float unscale =
      reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.dy) + a.dy_bytes - PL_HEADER)[2] *
      reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.src) + a.src_bytes - PL_HEADER)[2];
#pragma unroll
  for (int e = 0; e < 4; ++e) a.dw[(long long)(co + e) * a.ldw + cj] += sum[e] * unscale;
}


// ---- the same kernel on v_mfma_f32_16x16x32_f16 ----
// Why: the kernel runs at the clock the chip holds under it (zero-filled operands 107 us, random 142 us), and in the probe of
// its inner loop (scripts/hip_probe/wgrad_shape_probe.cpp: every fragment re-read from LDS by transposing reads, same bytes per
// FLOP) the 16x16x32 shape sustains 1645 raw TFLOP/s on random data where 32x32x16 sustains 1433 -- half the accumulator
// register traffic per FLOP. Same tile (128 filters x 9 taps x 32 channels per workgroup, 32 filters per wave), same ring,
// same DMA pieces; what changes is the k -> pixel map of a fragment (four 8-pixel octets, one per 16-lane group) and with it
// two LDS images, so that the two groups of a 32-lane half never share a bank:
//   dy pieces: sub-block S of pixel quad Q stored at S ^ 2 (Q >> 1) (the two 16-filter records swapped for pixels 8..15);
//   x ring:    8-channel chunk c of slot s stored at c ^ 2 ((s >> 3) & 1) (the two 16-channel halves swapped in every
//              other octet of slots) -- the swap lives in the DMA's per-lane SOURCE address and in the readers' addresses.
// Accumulators: acc[tap][channel half][filter block] of 16 x 16; slab piece (((wave * 9 + t) * 2 + hf) * 2 + m).
template <int RING, int KO = 0>
__global__ __launch_bounds__(256, 2) void wgrad_win16_kernel(const WgradArgs a) {
  constexpr int A_STAGE = 2 * 4 * PL_PLANES * 1024;
  constexpr int WIN_PLANE = RING * 64 + 64;
  constexpr int WIN_OFF = 2 * A_STAGE;
  constexpr unsigned RMASK = RING * 64 - 1;
  constexpr unsigned ZSLOT = RING * 64;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds_base = (unsigned)(size_t)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lid = xcd_remap(blockIdx.x, a.nblocks);
  const int tiles = a.tiles_co * a.tiles_j;
  const int split = lid / tiles;
  const int tile = lid - split * tiles;
  const int co0 = (tile % a.tiles_co) * WW_CO;
  const int ci0 = (tile / a.tiles_co) * WW_CI;
  const int p_begin = (int)((long long)split * a.chunk);
  int p_end = p_begin + (int)a.chunk;
  if (p_end > (int)a.M) p_end = (int)a.M;
  if (p_begin >= p_end) return;
  const int nst = (p_end - p_begin + WW_STAGE_PX - 1) / WW_STAGE_PX;
  const int W = a.Ws, H = a.Hs, M = (int)a.M;
  if (tid < 32) reinterpret_cast<unsigned*>(smem + WIN_OFF + (tid >> 4) * WIN_PLANE + ZSLOT)[tid & 15] = 0u;

  const unsigned strideA = (unsigned)((a.Cout >> 4) * PL_RECORD);
  const unsigned strideB = (unsigned)((a.Cs >> 4) * PL_RECORD);
  const unsigned zeroA = (unsigned)a.zero_blk_dy * strideA, zeroB = (unsigned)a.zero_blk_src * strideB;
  const i32x4 rsrcA = planes_rsrc(a.dy, a.dy_bytes), rsrcB = planes_rsrc(a.src, a.src_bytes);
  // dy piece: LDS position (quad Q = l >> 4, sub-block position (l >> 2) & 3, pixel l & 3) holds sub-block pos ^ 2 (Q >> 1)
  const int lq = lane >> 4, lpix = 4 * lq + (lane & 3), lsb = ((lane >> 2) & 3) ^ (2 * (lq >> 1));
  const unsigned a_lane = (unsigned)(((co0 + wave * 32) >> 4) + (lsb >> 1)) * PL_RECORD + (lsb & 1) * 256 + lpix * 16;
  auto issue_A = [&](int stage, int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pb16 = p_begin + stage * WW_STAGE_PX + j * 16;
      const unsigned v = (pb16 < p_end) ? (unsigned)(pb16 >> 4) * strideA + a_lane : zeroA;
#pragma unroll
      for (int p = 0; p < PL_PLANES; ++p) {
        const unsigned l = __builtin_amdgcn_readfirstlane(lds_base + buf * A_STAGE + ((j * 4 + wave) * PL_PLANES + p) * 1024);
        dma16(rsrcA, v, (unsigned)(p * 512), l);
      }
    }
  };
  // x piece: LDS position (pixel l >> 2, chunk position l & 3) holds chunk pos ^ 2 (pixel >> 3) (slot bit 3 = pixel bit 3: pieces
  // start at multiples of 16)
  const int bpx = lane >> 2, bch = (lane & 3) ^ (2 * (bpx >> 3));
  const unsigned b_lane = (unsigned)((ci0 >> 4) + (bch >> 1)) * PL_RECORD + (bch & 1) * 256;
  auto issue_B = [&](int P0, int plane) {
    const int P = P0 + bpx;
    const unsigned v = ((unsigned)P < (unsigned)M) ? (unsigned)(P >> 4) * strideB + b_lane + (P & 15) * 16 : zeroB;
    const unsigned l = __builtin_amdgcn_readfirstlane(lds_base + WIN_OFF + plane * WIN_PLANE + (unsigned)((P0 & (RING - 1)) * 64));
    dma16(rsrcB, v, (unsigned)(plane * 512), l);
  };
  const int lo = (p_begin - W - 1) & ~15;
  const int Lbase = (p_begin + W + 33 + 15) & ~15;
  {
    const int npieces = ((Lbase - lo) >> 4) * PL_PLANES;
    for (int idx = wave; idx < npieces; idx += 4) issue_B(lo + (idx >> 1) * 16, idx & 1);
  }
  issue_A(0, 0);

  // reader roles: 16-lane group g = pixel octet g of the 32-pixel stage; lane (qq, pq) of the group supplies pixel 8 g + 4 r + qq
  // of read r, channels 4 pq .. 4 pq + 3 of the 16-channel block
  const int grp = lane >> 4, i16 = lane & 15;
  const int qq = i16 >> 2, pq = i16 & 3;
  typedef ws16x4 __attribute__((address_space(3))) * lds_p;
  auto rd_pair = [&](unsigned a0, unsigned a1, int off) -> f16x8 {
    const ws16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(size_t)(a0 + off));
    const ws16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(size_t)(a1 + off));
    const ws16x8 v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    return __builtin_bit_cast(f16x8, v);
  };
  // dy fragment of filter block m: piece j = g >> 1, quad 2 (g & 1) + r, sub-block (2 m + (pq >> 1)) ^ 2 (g & 1)
  unsigned abase[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
    abase[m] = (unsigned)((grp >> 1) * 4 * PL_PLANES * 1024 + wave * PL_PLANES * 1024 + 512 * (grp & 1) +
                          64 * ((2 * m + (pq >> 1)) ^ (2 * (grp & 1))) + 16 * qq + 8 * (pq & 1));
  const unsigned lc = (unsigned)((pq >> 1) * 16 + (pq & 1) * 8);   // inside a 32-byte half of a window slot
  unsigned pb[2];
  int px[2], py[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int p = p_begin + 8 * grp + 4 * r + qq;
    pb[r] = (((unsigned)p & (RING - 1)) << 6) + lc;
    const int row = p / W;
    px[r] = p - row * W;
    py[r] = row % H;
  }
  const int adv_rows = WW_STAGE_PX / W, adv_rem = WW_STAGE_PX - adv_rows * W;
  int shb[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) shb[t] = ((t / 3 - 1) * W + (t % 3 - 1)) * 64;

  f32x4 acc[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[t][hf][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  // (lgkmcnt too: the zero slot above is an LDS write, and a raw s_barrier carries no implicit wait for it)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const unsigned zl = ZSLOT + lc;
  for (int s = 0; s < nst; ++s) {
    if constexpr (!(KO & 1)) {
      issue_A(s + 1, (s + 1) & 1);
      issue_B(Lbase + s * WW_STAGE_PX + (wave >> 1) * 16, wave & 1);
    }
    const unsigned aoff = (unsigned)((s & 1) * A_STAGE);
    bool okx0[2], okx1[2], oky0[2], oky1[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      okx0[r] = px[r] != 0;
      okx1[r] = px[r] != W - 1;
      oky0[r] = py[r] != 0;
      oky1[r] = py[r] != H - 1;
    }
    // addresses of tap t: [read r][channel half]; the half's 32 bytes swap with bit 3 of the slot (= bit 9 of the byte offset)
    auto calc = [&](int t, unsigned (&ad)[2][2]) {
      const int ty = t / 3, tx = t % 3;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        bool ok = true;
        if (ty == 0) ok = ok && oky0[r];
        if (ty == 2) ok = ok && oky1[r];
        if (tx == 0) ok = ok && okx0[r];
        if (tx == 2) ok = ok && okx1[r];
        unsigned b = (pb[r] + (unsigned)shb[t]) & RMASK;
        b = ((KO & 8) || ok) ? b : zl;
        ad[r][0] = b + (((b >> 9) & 1u) << 5);
        ad[r][1] = ad[r][0] ^ 32u;
      }
    };
    f16x8 ah[2], al[2], bh[2], bl[2];
    unsigned adt[2][2][2];   // [tap parity][read][half]
    calc(0, adt[0]);
    if constexpr (!(KO & 2)) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        ah[m] = rd_pair(abase[m] + aoff, abase[m] + aoff + 256, 0);
        al[m] = rd_pair(abase[m] + aoff, abase[m] + aoff + 256, 1024);
      }
      bh[0] = rd_pair(adt[0][0][0], adt[0][1][0], WIN_OFF);
      bl[0] = rd_pair(adt[0][0][0], adt[0][1][0], WIN_OFF + WIN_PLANE);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 18; ++i) {   // i = tap * 2 + channel half
      const int t = i >> 1, hf = i & 1, cur = i & 1, nxt = cur ^ 1;
      if constexpr (!(KO & 4)) {
        acc[t][hf][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[0], bh[cur], acc[t][hf][0], 0, 0, 0);
        acc[t][hf][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[1], bh[cur], acc[t][hf][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(KO & 2)) {
        if (i + 1 < 18) {   // the fragments of step i + 1 = (tap (i + 1) / 2, half (i + 1) & 1)
          const int tn = (i + 1) >> 1, hn = (i + 1) & 1;
          bh[nxt] = rd_pair(adt[tn & 1][0][hn], adt[tn & 1][1][hn], WIN_OFF);
          bl[nxt] = rd_pair(adt[tn & 1][0][hn], adt[tn & 1][1][hn], WIN_OFF + WIN_PLANE);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(KO & 4)) {
        acc[t][hf][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0], bl[cur], acc[t][hf][0], 0, 0, 0);
        acc[t][hf][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[1], bl[cur], acc[t][hf][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (hf == 0 && t + 1 < 9) calc(t + 1, adt[(t + 1) & 1]);   // (tap t + 1's addresses: read from step (t, 1) on)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(KO & 4)) {
        acc[t][hf][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0], bh[cur], acc[t][hf][0], 0, 0, 0);
        acc[t][hf][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[1], bh[cur], acc[t][hf][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (i == 17) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          pb[r] = (pb[r] + WW_STAGE_PX * 64) & RMASK;
          int x = px[r] + adv_rem, y = py[r] + adv_rows;
          if (x >= W) {
            x -= W;
            ++y;
          }
          if (y >= H) y -= H;
          px[r] = x;
          py[r] = y;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  const float unscale =
      reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.dy) + a.dy_bytes - PL_HEADER)[2] *
      reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.src) + a.src_bytes - PL_HEADER)[2];
  if (a.slabs != nullptr) {
    float* mine = a.slabs + (size_t)lid * WW_TILE_FLOATS;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int m = 0; m < 2; ++m)
          *reinterpret_cast<f32x4*>(mine + (((((wave * 9 + t) * 2 + hf) * 2 + m) * 64) + lane) * 4) = acc[t][hf][m];
    return;
  }
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const long long cj = (long long)t * a.Cs + ci0 + 16 * hf + (lane & 15);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int co = co0 + wave * 32 + 16 * m + 4 * (lane >> 4) + e;
          atomicAdd(&a.dw[(long long)co * a.ldw + cj], acc[t][hf][m][e] * unscale);
        }
    }
}

// the ordered reduce for the 16x16 accumulator order: piece (((wave * 9 + t) * 2 + hf) * 2 + m), lane l = filter rows
// 4 (l >> 4) .. + 3 of block m, channel 16 hf + (l & 15)
__global__ __launch_bounds__(256) void wgrad_win16_reduce_kernel(const WgradArgs a) {
  constexpr int PER_TILE = WW_TILE_FLOATS / 4;
  const int tiles = a.tiles_co * a.tiles_j;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const int tile = (int)(idx / PER_TILE);
  if (tile >= tiles) return;
  const int q = (int)(idx - (long long)tile * PER_TILE);
  const int lane = q & 63, piece = q >> 6;
  const int m = piece & 1, hf = (piece >> 1) & 1, t = (piece >> 2) % 9, wave = piece / 36;
  const int co = (tile % a.tiles_co) * WW_CO + wave * 32 + 16 * m + 4 * (lane >> 4);
  const long long cj = (long long)t * a.Cs + (tile / a.tiles_co) * WW_CI + 16 * hf + (lane & 15);
  const f32x4* sl = reinterpret_cast<const f32x4*>(a.slabs) + (size_t)tile * PER_TILE + q;
  const size_t sstride = (size_t)tiles * PER_TILE;
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 8 <= a.splits; s += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(sl + (size_t)(s + u) * sstride);
#pragma unroll
    for (int u = 0; u < 8; ++u) sum = sum + v[u];
  }
  for (; s < a.splits; ++s) sum = sum + __builtin_nontemporal_load(sl + (size_t)s * sstride);
  const float unscale =
      reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.dy) + a.dy_bytes - PL_HEADER)[2] *
      reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.src) + a.src_bytes - PL_HEADER)[2];
#pragma unroll
  for (int e = 0; e < 4; ++e) a.dw[(long long)(co + e) * a.ldw + cj] += sum[e] * unscale;
}

bool wgrad_win_supported(const WgradArgs& a) {
  if (a.ntaps != 9 || a.kw != 3 || a.sy != 1 || a.sx != 1 || a.pad_t != 1 || a.pad_l != 1) return false;
  if (a.Hg != a.Hs || a.Wg != a.Ws) return false;
  if ((a.Cout % WW_CO) != 0 || (a.Cs % WW_CI) != 0 || a.ldw != 9 * a.Cs) return false;
  // ring: the loads of stage s + 1 must not land on the slots stage s reads (RING >= 2 W + 81); the per-stage advance of a
  // lane's (x, y) is one conditional step each (32 / W + 2 <= H)
  if (a.Ws < 4 || 2 * a.Ws + 81 > 512 || 32 / a.Ws + 2 > a.Hs) return false;
  if (a.M < 64 || a.M >= (1LL << 31) - 4096) return false;
  return true;
}

template <int RING, int KO, int PF = 1, bool M16 = false>
static int launch_ww(WgradArgs& a, hipStream_t st) {
  const void* kfn = M16 ? reinterpret_cast<const void*>(&wgrad_win16_kernel<RING, KO>)
                        : reinterpret_cast<const void*>(&wgrad_win_kernel<RING, KO, PF>);
  a.tiles_co = a.Cout / WW_CO;
  a.tiles_j = a.Cs / WW_CI;
  const long long tiles = (long long)a.tiles_co * a.tiles_j;
  constexpr size_t lds = 2 * (2 * 4 * PL_PLANES * 1024) + 2 * (RING * 64 + 64);
  static int resident = 0;
  static bool lds_from_zero = true;
  if (resident == 0) {
    int per_cu = 0, dev = 0, cus = 0;
    // the kernels address their dynamic LDS from 0 (immediate offsets in the window reads): true only while no static
    // __shared__ sits in front of it. Checked here instead of trapping on the device; 1 = "not covered", the caller then
    // runs the per-tap kernel.
    hipFuncAttributes fa{};
    if (hipFuncGetAttributes(&fa, kfn) != hipSuccess || fa.sharedSizeBytes != 0) lds_from_zero = false;
    (void)hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 256, lds) ==
            hipSuccess &&
        hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
      resident = per_cu * cus;
    if (resident <= 0) resident = 512;
  }
  if (!lds_from_zero) return 1;
  // ONE workgroup per CU -- half of what the chip holds at once (256 on 256 CUs; round 6), at least 8 stages per workgroup.
  // Until round 5 the target was a full round of resident workgroups (two per CU). Alone the kernel prefers that; in the step
  // it is a LEAF on a stream with slack, and every workgroup it keeps resident takes 65 KB of a CU's LDS from the compute
  // stream's window data gradient -- the critical chain (the lesson of the two-half kernel below) -- while half the workgroups
  // are half the slab bytes (37 instead of 75 MB written, 43 instead of 85 MB read back, per launch). Same-box sweeps
  // (profiles/r06_j_wgrad_win_target_ab.log), 512 -> 256 / 160 / 128 workgroups: YOLOv3-416 28.80-28.90 -> 28.72-28.79 / 28.70 /
  // 28.69-28.77 ms; YOLOv2-416 7.22-7.24 -> 7.05-7.11 / 7.08 / 7.03-7.07; YOLOv4-608 36.98-37.09 -> 36.74-36.84 / 36.41 /
  // 36.31-36.39; 96: C3 29.01; 64: C3 30.98; 768 / 1024: slower than 512. 256 is the default: the headline is flat from 128 to
  // 256, and below 256 this kernel -- throttled on purpose -- becomes the one with the most device time in the step, so that
  // the step's roofline line would describe a launch that is not trying to fill the chip. YOLO_WGRAD_WIN_TARGET=128 is the
  // better setting for YOLOv4-608 (compute stream 98 % busy, filter-gradient stream with 20 % slack).
  static const long long target_env = [] { const char* e = getenv("YOLO_WGRAD_WIN_TARGET"); return e ? atoll(e) : 0LL; }();
  const long long target = target_env > 0 ? target_env : (resident >= 2 ? resident / 2 : resident);
  long long splits = target / tiles;
  const long long max_splits = (a.M + 255) / 256;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  a.slabs = nullptr;
  static const bool det_env = [] { const char* e = getenv("YOLO_WGRAD_DETERMINISTIC"); return !(e && atoi(e) == 0); }();
  size_t ws_bytes = 0;
  void* ws = wgrad_workspace(&ws_bytes);
  if (det_env && ws != nullptr && ws_bytes > WGRAD_WS_COLSUM_BYTES) {
    const long long cap = (long long)((ws_bytes - WGRAD_WS_COLSUM_BYTES) / ((size_t)WW_TILE_FLOATS * 4));
    if (cap >= tiles) {
      if (tiles * splits > cap) splits = cap / tiles;
      a.slabs = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(ws) + WGRAD_WS_COLSUM_BYTES);
    }
  }
  long long chunk = (a.M + splits - 1) / splits;
  chunk = (chunk + 31) / 32 * 32;
  splits = (a.M + chunk - 1) / chunk;
  a.chunk = chunk;
  a.splits = (int)splits;
  a.nblocks = (int)(tiles * splits);
  // (one split -- the small layers of a small batch -- still goes through its slab: measured round 6, a tile's owner adding
  // to dw itself with scalar read-modify-writes costs the kernel what the slab + reduce launch cost: 44 vs 25 + 17 us, and
  // the 128 x 256 per-tap tile 103 vs 22 + 27 us)
  if constexpr (M16) hipLaunchKernelGGL((wgrad_win16_kernel<RING, KO>), dim3((unsigned)a.nblocks), dim3(256), lds, st, a);
  else hipLaunchKernelGGL((wgrad_win_kernel<RING, KO, PF>), dim3((unsigned)a.nblocks), dim3(256), lds, st, a);
  if (int rc = check_launch("wgrad_win_kernel")) return rc;
  if (a.slabs != nullptr) {
    const long long pieces = tiles * (WW_TILE_FLOATS / 4);
    if constexpr (M16) hipLaunchKernelGGL(wgrad_win16_reduce_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(wgrad_win_reduce_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, a);
    return check_launch("wgrad_win_reduce_kernel");
  }
  return YOLO_OK;
}

// the two-half kernel (RING = 256 only: two rings of 512 slots do not fit a CU's LDS)
static int launch_ww2(WgradArgs& a, hipStream_t st) {
  constexpr int RING = 256;
  const void* kfn = reinterpret_cast<const void*>(&wgrad_win2_kernel<RING>);
  a.tiles_co = a.Cout / WW_CO;
  a.tiles_j = a.Cs / WW_CI;
  const long long tiles = (long long)a.tiles_co * a.tiles_j;
  constexpr size_t half_bytes = 2 * (2 * 4 * PL_PLANES * 1024) + 2 * (RING * 64 + 64);
  constexpr size_t handover = (size_t)WW_TILE_FLOATS * 4;
  constexpr size_t lds = 2 * half_bytes > handover ? 2 * half_bytes : handover;
  static int resident = 0;
  static bool ok = true;
  if (resident == 0) {
    int per_cu = 0, dev = 0, cus = 0;
    hipFuncAttributes fa{};
    if (hipFuncGetAttributes(&fa, kfn) != hipSuccess || fa.sharedSizeBytes != 0) ok = false;   // (dynamic LDS must start at 0)
    if (hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) ok = false;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 512, lds) == hipSuccess && per_cu >= 1 &&
        hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
      resident = per_cu * cus;
    else
      ok = false;
    if (resident <= 0) resident = 256;
  }
  if (!ok) return 1;
  size_t ws_bytes = 0;
  void* ws = wgrad_workspace(&ws_bytes);
  static const bool det_env = [] { const char* e = getenv("YOLO_WGRAD_DETERMINISTIC"); return !(e && atoi(e) == 0); }();
  if (!det_env || ws == nullptr || ws_bytes <= WGRAD_WS_COLSUM_BYTES) return 1;   // (the atomics fallback stays with the 4-wave kernel)
  long long splits = resident / tiles;
  const long long max_splits = (a.M + 511) / 512;   // at least 8 stages per half
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  const long long cap = (long long)((ws_bytes - WGRAD_WS_COLSUM_BYTES) / ((size_t)WW_TILE_FLOATS * 4));
  if (cap < tiles) return 1;
  if (tiles * splits > cap) splits = cap / tiles;
  a.slabs = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(ws) + WGRAD_WS_COLSUM_BYTES);
  long long chunk = (a.M + splits - 1) / splits;
  chunk = (chunk + 63) / 64 * 64;                   // two halves of whole 32-pixel stages
  splits = (a.M + chunk - 1) / chunk;
  a.chunk = chunk;
  a.splits = (int)splits;
  a.nblocks = (int)(tiles * splits);
  hipLaunchKernelGGL((wgrad_win2_kernel<RING>), dim3((unsigned)a.nblocks), dim3(512), lds, st, a);
  if (int rc = check_launch("wgrad_win2_kernel")) return rc;
  const long long pieces = tiles * (WW_TILE_FLOATS / 4);
  hipLaunchKernelGGL(wgrad_win_reduce_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, a);
  return check_launch("wgrad_win_reduce_kernel");
}

int launch_wgrad_win(WgradArgs& a, hipStream_t st) {
  const bool small = 2 * a.Ws + 81 <= 256;
  if (small && g_opt[OPT_WGRAD_WIN] == 4) {   // two pixel halves per workgroup: half the slabs (round 6)
    const int rc = launch_ww2(a, st);
    if (rc != 1) return rc;
  }
#ifdef YOLO_PLANES_KNOCKOUTS   // diagnostic build (make KNOCKOUTS=1)
  static const int ko = [] { const char* e = getenv("YOLO_WGRAD_KO"); return e ? atoi(e) : 0; }();
  if (small) switch (ko) {
      case 1: return launch_ww<256, 1>(a, st);
      case 2: return launch_ww<256, 2>(a, st);
      case 3: return launch_ww<256, 3>(a, st);
      case 4: return launch_ww<256, 4>(a, st);
      case 7: return launch_ww<256, 7>(a, st);
      case 8: return launch_ww<256, 8>(a, st);
      default: break;
    }
#endif
  if (small && g_opt[OPT_WGRAD_WIN] == 2) return launch_ww<256, 0, 2>(a, st);   // fragment reads two tap-steps ahead
  if (g_opt[OPT_WGRAD_WIN] == 3) return small ? launch_ww<256, 0, 1, true>(a, st) : launch_ww<512, 0, 1, true>(a, st);   // 16x16x32 MFMA
  return small ? launch_ww<256, 0>(a, st) : launch_ww<512, 0>(a, st);
}

}  // namespace yolo

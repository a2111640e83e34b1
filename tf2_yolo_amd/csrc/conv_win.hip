// 3x3 stride-1 convolution forward / data gradient on "planes" operands with the INPUT WINDOW KEPT IN LDS.
//
// conv_planes.hip streams, for every filter tap, the 128 shifted pixels of the tile from L2 into LDS again: nine
// times the same bytes (plus the halo), 341 B of operands per MFMA, and with 40+ workgroups per XCD the window has
// left the 4 MB L2 by the time the next tap wants it (254 MB beyond-L2 traffic per launch against ~70 MB of
// compulsory bytes). Here a workgroup loads, per 16-channel block, ONE window of the input -- every pixel any tap of
// any pixel of the tile touches -- and takes the nine shifted A fragments from it; only the filter tile (B) is
// streamed per tap.
//
// Window coordinates. Pixel (n, y, x) of the [N][H][W] source gets the "padded linear" index
//     u(n, y, x) = (n (H+1) + y + 1) (W+1) + x + 1
// i.e. rows of pitch W+1 whose column 0 is a zero column (the right neighbour of x = W-1 is column 0 of the next
// row, the left neighbour of x = 0 is column 0 of the same row) and one zero row between consecutive images (and
// above the first): tap (dy, dx) of a pixel is at u + dy (W+1) + dx for EVERY pixel, borders included. The tile's
// BM consecutive output pixels m0 .. m0+BM-1 need the contiguous range u(m0) - (W+1) - 1 .. u(m_last) + (W+1) + 1;
// it is loaded in chunks of 64 window pixels (one LDS-DMA instruction = 64 lanes x 16 B = one (plane, k-half) of
// 64 pixels; zero columns / rows / images past the end fetch the planes' all-zero block). LDS image of a window:
// [plane][k-half][NCH*64 pixels][16 B], so the A fragment of tap t for lane (r, hf) is ONE ds_read_b128 at
// base(r, hf) + tapoff_t * 16: 16 consecutive pixels = 16 distinct 16-byte slots, conflict-free up to the +1 jump
// where the tile crosses an image row.
//
// Schedule (per wave; NB = 2 filter DMAs per stage with 4 waves, 1 with 8): stages in (channel block c, tap t)
// order; B ring of 3 stage buffers (9 taps = 3 turns of the ring, so the buffer of a stage is a compile-time
// constant), two window buffers (block c reads one while block c+1's chunks land in the other, ONE chunk DMA per
// wave and stage during taps 0 .. NWS-1), two fragment register sets. Iteration (c, t):
//     wait vmcnt(NB + [a window DMA was issued in the previous stage])  -> my filter DMAs of the next stage landed
//     barrier
//     fragment reads of the next stage (A from the window at the next tap's offset, B from the ring)
//     12 MFMAs of this stage, with the window chunk DMA (t < NWS) and the filter DMAs of stage +3 between them
// DMAs are issued from inline asm and counted by hand, window chunk before filter pieces in every stage, so that
// "all but the youngest NB (+1)" is exactly "everything up to the next stage's filter pieces"; the window of block
// c+1 is complete two stages before its first read for the same reason (NWS <= 7).
//
// Persistent "stream-K" form (a.sk_grid > 0). With one workgroup per tile every workgroup of a round reaches its
// epilogue at the same time: the chip alternates between rounds of MFMA work and bursts of output stores in which
// the matrix pipes idle (measured on the 52x52x128->256 layer: 29 k of a tile's 80 k cycles in the epilogue,
// 6 k in the prologue). Here sk_grid resident workgroups each take an equal, contiguous share of the
// (tile, channel block) units; a workgroup's tile boundaries fall at a different point of its life than its
// neighbours', so the epilogues and prologues of some overlap the main loops of the others, and the last round is
// as full as the first. A tile whose channel blocks are shared by several workgroups is combined without any
// waiting: every part owner writes its accumulators to its slab in the workspace with write-through (sc1) stores,
// drains them, and draws a ticket (agent-scope atomic); whoever draws the last ticket acquires, adds the other
// parts IN PART ORDER (bitwise reproducible whatever the arrival order), resets the ticket and runs the epilogue.
// No workgroup ever waits for another one, so the result does not depend on residency, dispatch order or placement
// (cdna_hip_programming.md, Guideline 16 and "In-launch split-K reduction").
#include "planes_epilogue.hpp"
#include <cstdlib>
#include <type_traits>

namespace yolo {

// WGM x 2 waves of 64x64: tile (64*WGM) x 128. NCH = window chunks of 64 pixels (LDS: 2 x NCH x 4 KB + 24 KB).
// SPLIT: the split-K instantiation (every workgroup one part of one tile, accumulators to a slab, no epilogue). Its own
// kernel because the slab path inside the production instantiation tripled its scratch (152 -> 440-496 bytes per lane)
// and made every window launch of the training step 10-40 % slower.
// SK: the stream-K instantiation (a.sk_grid > 0: unit shares, slabs, tickets); the production kernel carries none of it.
//
// GEO = 1: PATCH geometry (round 3). The tile is a 2-D patch of BM / 16 rows x 16 columns of output pixels of ONE image
// and the window its (rows + 2) x 18 halo patch of the input -- 180 window pixels for a 128-pixel tile whatever the row
// length, where the linear window of a W-pixel row needs 2 (W + 1) + BM + 3 (549 for W = 208). Only the prologue differs
// (which source pixel each window slot fetches, where a lane's fragment starts, the tap offsets in the window) and the
// epilogue's row -> pixel map (planes_epilogue<..., PATCH>); the main loop is the same code. Used for rows longer than
// 64 pixels (104 / 208 in YOLOv3-416, 76 / 152 / 304 in YOLOv4-608).
// WGN = 1: 64 filter columns per tile (BN = 64) with WGM = 4 waves of 64 x 64 stacked in M (256 output pixels = a
// 16 x 16 patch): the Cout = 64 layers, which would waste half of a 128-column tile.
template <int WGM, int NCH, bool STAMPS = false, bool SPLIT = false, bool SK = false, int WGN = 2, int GEO = 0, bool BNRED = false>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void conv_win_kernel(const GatherConvArgs a) {
  constexpr int BM = 64 * WGM, BN = 64 * WGN;
  constexpr int NW = WGM * WGN;
  constexpr int TM = 2, TN = 2;
  constexpr int NB = 4 * WGN / NW;                 // filter DMAs per wave and stage (2 WGN blocks x 2 planes x 1 KB per stage)
  constexpr int NWS = (NCH * 4) / NW;              // window DMAs per wave and channel block, one per stage
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  static_assert(NB == 1 || NB == 2, "filter loader roles: one (block, plane) or one block with both planes per wave");
  static_assert((NCH * 4) % NW == 0 && NWS <= 7, "window chunks: NCH*4 DMAs spread evenly, at most 7 stages");
  static_assert(GEO == 0 || (!SPLIT && !SK), "patch geometry: one workgroup per tile");
  constexpr int WIN_BYTES = NCH * 4096;            // one window buffer: 2 planes x 2 halves x NCH*64 px x 16 B
  constexpr int PLANE_STRIDE = 2 * NCH * 1024;
  constexpr int BSTAGE = 4096 * WGN;               // 2 WGN blocks of 32 filters x 2 planes x 1 KB
  constexpr int BRING = 2 * WIN_BYTES;
  constexpr int LDS_TOTAL = 2 * WIN_BYTES + 3 * BSTAGE;
  // patch geometry: PH x 16 output pixels, window (PH + 2) x 18 slots
  constexpr int PH = BM / 16, PITCH = 18;
  static_assert(GEO == 0 || (PH + 2) * PITCH <= NCH * 64, "patch window does not fit the chunks");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds_base = (unsigned)(size_t)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  // diagnostic build: shader-clock stamps of wave 0 around the phases of every part (summed per phase), and the
  // constant 100 MHz clock at both ends
  [[maybe_unused]] unsigned long long stamp[4] = {0, 0, 0, 0}, phase_sum[3] = {0, 0, 0};
  auto take_stamp = [&](int i) {
    if constexpr (STAMPS) {
      __builtin_amdgcn_sched_barrier(0);
      stamp[i] = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  [[maybe_unused]] unsigned long long rt0 = 0, t_start = 0;
  [[maybe_unused]] int nparts_done = 0, epi_count = 0;
  [[maybe_unused]] unsigned long long epi_sum[6] = {0, 0, 0, 0, 0, 0};
  if constexpr (STAMPS) {
    rt0 = __builtin_amdgcn_s_memrealtime();
    t_start = __builtin_amdgcn_s_memtime();
  }

  const int H = a.Hs, W = a.Ws, P1 = GEO == 0 ? W + 1 : PITCH;
  const int HW = H * W;
  const int cpt = a.Cs >> 4;                       // 16-channel blocks

  const i32x4 rsrcA = planes_rsrc(a.src, a.src_bytes), rsrcB = planes_rsrc(a.wgt, a.wgt_bytes);
  const unsigned blkstrideA = (unsigned)(cpt * PL_RECORD);
  const unsigned blkstrideB = (unsigned)((a.ldw >> 4) * PL_RECORD);
  // tap offsets in the window (bytes): ((oy+1)(W+1) + ox+1) * 16
  int tapoff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) tapoff[t] = ((a.taps[t].oy + 1) * P1 + a.taps[t].ox + 1) * 16;

  // ---- my share of the (tile, channel block) units: [u_begin, u_end) ----
  // sk_grid == 0: one whole tile per workgroup. Otherwise the units are dealt out evenly to sk_grid workgroups,
  // logical workgroup wl = XCD-aware remap of blockIdx (neighbours in unit space share an L2).
  const int G = (SK || SPLIT) ? a.sk_grid : 0;
  const int wl = xcd_remap(blockIdx.x, G > 0 ? G : a.nblocks);
  const int Gd = G > 0 ? G : 1;
  const int sk_q = G > 0 ? (a.nblocks * cpt) / Gd : 0, sk_r = G > 0 ? (a.nblocks * cpt) % Gd : 0;
  auto first_unit = [&](int w) { return w * sk_q + (w < sk_r ? w : sk_r); };
  const int sk_qd = sk_q > 0 ? sk_q : 1;   // (sk_q is the constant 0 outside the stream-K instantiation)
  auto owner_of = [&](int u) { return u < sk_r * (sk_q + 1) ? u / (sk_q + 1) : sk_r + (u - sk_r * (sk_q + 1)) / sk_qd; };
  // split_parts > 1 (split-K with a reduce kernel behind it): logical workgroup wl = tile * P + part computes the
  // channel blocks [part * cpt / P, (part + 1) * cpt / P) of its tile and nothing else
  const int SP = SPLIT ? a.split_parts : 1;
  const int u_begin = SPLIT ? (wl / SP) * cpt + ((wl % SP) * cpt) / SP : G > 0 ? first_unit(wl) : wl * cpt;
  const int u_end = SPLIT ? (wl / SP) * cpt + ((wl % SP + 1) * cpt) / SP : G > 0 ? first_unit(wl + 1) : (wl + 1) * cpt;

  for (int u_cur = u_begin; u_cur < u_end;) {
  take_stamp(0);
  const int tile = u_cur / cpt;
  const int cb0 = u_cur - tile * cpt;                                   // first channel block of this part
  const int cb1 = (cpt - cb0 < u_end - u_cur) ? cpt : cb0 + (u_end - u_cur);   // one past the last
  u_cur += cb1 - cb0;
  // tile order inside an XCD's run of tiles: column tile fastest (neighbours share the input window) or, for layers
  // whose FILTER is the big operand (13x13: 18.9 MB, five L2s' worth), row tile fastest (neighbours stream the same
  // 128 filters; the windows are small) -- launch_win() decides
  const int tiles_m = a.nblocks / a.tiles_n;
  const int tile_n = a.tile_order ? tile / tiles_m : tile % a.tiles_n;
  const int tile_m = a.tile_order ? tile % tiles_m : tile / a.tiles_n;
  const long long m0 = (long long)tile_m * BM;
  const int n0 = tile_n * BN;
  // patch geometry: tile_m -> (image pn, patch row py0, patch column px0) of the output (= input: stride 1, same size)
  [[maybe_unused]] int pn = 0, py0 = 0, px0 = 0;
  if constexpr (GEO == 1) {
    const int tx = (W + 15) >> 4, ty = (H + PH - 1) / PH;
    pn = tile_m / (tx * ty);
    const int rem = tile_m - pn * (tx * ty);
    py0 = (rem / tx) * PH;
    px0 = (rem % tx) * 16;
  }

  // padded linear index of a pixel m
  auto u_of = [&](long long m) -> int {
    const int n = (int)(m / HW);
    const int rem = (int)(m - (long long)n * HW);
    const int y = rem / W;
    return (n * (H + 1) + y + 1) * P1 + (rem - y * W) + 1;
  };
  const int u_m0 = GEO == 0 ? __builtin_amdgcn_readfirstlane(u_of(m0)) : 0;
  const int u_lo = u_m0 - P1 - 1;

  // ---- window loader role: (plane, k-half) of this wave, NWS chunks; voffW = source unit of my pixel ----
  const int wplane = (wave >> 1) & 1, whf = wave & 1;
  unsigned voffW[NWS];
  int chunkW[NWS];
#pragma unroll
  for (int s = 0; s < NWS; ++s) {
    chunkW[s] = (NW == 4) ? s : 2 * s + (wave >> 2);
    const int u = u_lo + chunkW[s] * 64 + lane;
    unsigned v = (unsigned)a.zero_blk_src * blkstrideA;
    if constexpr (GEO == 1) {
      // window slot -> (row, column) of the halo patch -> source pixel (py0 - 1 + row, px0 - 1 + column) of image pn
      const int sl = chunkW[s] * 64 + lane;
      const int wy = sl / PITCH, wx = sl - wy * PITCH;
      const int sy = py0 - 1 + wy, sx = px0 - 1 + wx;
      if (wy < PH + 2 && (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) {
        const int pix = (pn * H + sy) * W + sx;
        v = ((unsigned)pix >> 4) * blkstrideA + (unsigned)(pix & 15) * 16;
      }
    } else
    if (u >= 0) {
      const int vrow = u / P1, xc = u - vrow * P1;
      const int n = vrow / (H + 1), vr = vrow - n * (H + 1);
      if (xc >= 1 && vr >= 1 && n < a.N) {
        const int pix = (n * H + vr - 1) * W + xc - 1;
        v = ((unsigned)pix >> 4) * blkstrideA + (unsigned)(pix & 15) * 16;
      }
    }
    voffW[s] = v;
  }
  const unsigned soffW0 = (unsigned)(wplane * 512 + whf * 256);
  const unsigned ldsW0 = lds_base + (unsigned)((wplane * 2 + whf) * NCH * 1024);

  // ---- filter loader role: 32-filter block rb, plane(s) ----
  const int r = lane & 31, hf = lane >> 5;
  const int rbB = (NB == 2) ? wave : (wave >> 1);
  unsigned voffB;
  {
    const int co = n0 + rbB * 32 + r;
    const unsigned blk = co < a.Cout ? (unsigned)(co >> 4) : (unsigned)a.zero_blk_wgt;
    voffB = blk * blkstrideB + (co < a.Cout ? (unsigned)(co & 15) * 16 : 0u) + (unsigned)hf * 256;
  }

  // ---- A fragment bases: lane (r, hf) of 32-row block (wm*TM + i) ----
  unsigned abase[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long long m = m0 + (wm * TM + i) * 32 + r;
    int wi;
    if constexpr (GEO == 1) {
      const int rr = (wm * TM + i) * 32 + r;        // tile row -> (patch row, column); its window origin is that slot
      wi = (rr >> 4) * PITCH + (rr & 15);
    } else {
      wi = (m < a.M) ? (u_of(m) - u_m0) : 0;
    }
    abase[i] = (unsigned)(hf * NCH * 1024 + wi * 16);
  }

  // window chunk DMA number s (0..NWS-1) of channel block cb into window buffer wb
  auto issue_window = [&](int s, int cb, int wb) {
    const int cbe = cb < cpt ? cb : cpt - 1;       // past the end: reload the last block (never read)
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)cbe * PL_RECORD + soffW0);
    const unsigned l = __builtin_amdgcn_readfirstlane(ldsW0 + (unsigned)(wb * WIN_BYTES + chunkW[s] * 1024));
    dma16(rsrcA, voffW[s], so, l);
  };
  // filter DMA number d (0..NB-1) of stage (cb, t) into ring buffer rb3
  auto issue_filter = [&](int d, int cb, int t, int rb3) {
    const int cbe = cb < cpt ? cb : cpt - 1;
    const int plane = (NB == 2) ? d : (wave & 1);
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(t * cpt + cbe) * PL_RECORD + (unsigned)plane * 512);
    const unsigned l = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(BRING + rb3 * BSTAGE + (rbB * 2 + plane) * 1024));
    dma16(rsrcB, voffB, so, l);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  f16x8 fa[2][PL_PLANES][TM], fb[2][PL_PLANES][TN];
  // fragments of stage (cb, t) into register set S
  auto read_frags = [&](int cb, auto T, auto SET) {
    constexpr int S = decltype(SET)::value;
    constexpr int t = decltype(T)::value;
    const unsigned so = (unsigned)((cb & 1) * WIN_BYTES + tapoff[t]);
    const unsigned char* sbB = smem + BRING + (t % 3) * BSTAGE + lane * 16;
#pragma unroll
    for (int p = 0; p < PL_PLANES; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[S][p][i] = *reinterpret_cast<const f16x8*>(smem + (abase[i] + so) + p * PLANE_STRIDE);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[S][p][j] = *reinterpret_cast<const f16x8*>(sbB + ((wn * TN + j) * PL_PLANES + p) * 1024);
    }
  };

  // the 12 MFMAs of a stage (l*h, h*l, h*h; planes 0 = h, 1 = l) on register set S, with this stage's DMA issue
  // (window chunk t of block c+1 first, then the filter pieces of stage +3) spread between them
  auto mfma_stage = [&](int c, auto T, auto SET) {
    constexpr int S = decltype(SET)::value;
    constexpr int t = decltype(T)::value;
    constexpr int ND = NB + (t < NWS ? 1 : 0);
    constexpr int NM = TM * TN * 3;
    constexpr int t3 = (t + 3) % 9;
    const int c3 = c + (t + 3) / 9;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int pa = (q == 0) ? 1 : 0;
      const int pb = (q == 1) ? 1 : 0;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[S][pa][i], fb[S][pb][j], acc[i][j], 0, 0, 0);
          const int idx = (q * TM + i) * TN + j;
#pragma unroll
          for (int d = 0; d < ND; ++d)
            if (idx == (((d + 1) * NM) / (ND + 1) > 0 ? ((d + 1) * NM) / (ND + 1) - 1 : 0)) {
              __builtin_amdgcn_sched_barrier(0);
              if (t < NWS && d == 0) issue_window(t, c + 1, (c + 1) & 1);
              else issue_filter(d - (t < NWS ? 1 : 0), c3, t3, t % 3);
              __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
  };

  // ---- prologue: window of the first block, filter stages (cb0,0) (cb0,1) (cb0,2); fragments of (cb0,0) in set 0 ----
#pragma unroll
  for (int s = 0; s < NWS; ++s) issue_window(s, cb0, cb0 & 1);
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int d = 0; d < NB; ++d) issue_filter(d, cb0, t, t);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NB) : "memory");   // filter stages 1 and 2 may still fly
  __builtin_amdgcn_s_barrier();
  read_frags(cb0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  take_stamp(1);

  // one stage: S = register set of this stage, the next stage's fragments go into the other one
  auto stage = [&](int c, auto T, auto SET) {
    constexpr int t = decltype(T)::value;
    constexpr int S = decltype(SET)::value;
    constexpr int wprev = (t >= 1 && t - 1 < NWS) ? 1 : 0;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB + wprev) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    constexpr int tn = (t + 1) % 9;
    read_frags(c + (t + 1) / 9, std::integral_constant<int, tn>{}, std::integral_constant<int, 1 - S>{});
    __builtin_amdgcn_sched_barrier(0);
    mfma_stage(c, T, SET);
  };
  auto block = [&](int c, auto P) {
    constexpr int p = decltype(P)::value;
    stage(c, std::integral_constant<int, 0>{}, std::integral_constant<int, (p + 0) & 1>{});
    stage(c, std::integral_constant<int, 1>{}, std::integral_constant<int, (p + 1) & 1>{});
    stage(c, std::integral_constant<int, 2>{}, std::integral_constant<int, (p + 2) & 1>{});
    stage(c, std::integral_constant<int, 3>{}, std::integral_constant<int, (p + 3) & 1>{});
    stage(c, std::integral_constant<int, 4>{}, std::integral_constant<int, (p + 4) & 1>{});
    stage(c, std::integral_constant<int, 5>{}, std::integral_constant<int, (p + 5) & 1>{});
    stage(c, std::integral_constant<int, 6>{}, std::integral_constant<int, (p + 6) & 1>{});
    stage(c, std::integral_constant<int, 7>{}, std::integral_constant<int, (p + 7) & 1>{});
    stage(c, std::integral_constant<int, 8>{}, std::integral_constant<int, (p + 8) & 1>{});
  };
  {
    int c = cb0;
    for (; c + 1 < cb1; c += 2) {
      block(c, std::integral_constant<int, 0>{});
      block(c + 1, std::integral_constant<int, 1>{});
    }
    if (c < cb1) block(c, std::integral_constant<int, 0>{});
  }
  take_stamp(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail DMAs too
  __syncthreads();  // every wave is done with the LDS: the epilogue reuses it

  bool finish = !SPLIT;
  if constexpr (SPLIT) store_split_slab<TM, TN>(a, acc, BM * BN * 4, wl, wave, lane);   // conv_split_reduce_kernel finishes the tile
  if constexpr (SK)
  if (cb1 - cb0 < cpt) {
    // ---- part of a tile: publish my accumulators, draw a ticket; the last arriver combines ----
    constexpr int SLAB_BYTES = BM * BN * 4;
    const int w_first = owner_of(tile * cpt), w_last = owner_of(tile * cpt + cpt - 1);
    const int nparts = w_last - w_first + 1;
    auto slab_of = [&](int w) {   // a workgroup's head part (its first tile) is slot 0, its tail part slot 1
      const int slot = (first_unit(w) / cpt == tile) ? 0 : 1;
      return reinterpret_cast<unsigned char*>(a.sk_slabs) + (size_t)(w * 2 + slot) * SLAB_BYTES;
    };
    {
      unsigned char* mine = slab_of(wl);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(mine, 0, SLAB_BYTES, 0x00020000);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            // (bit_cast of the whole vector: hipcc 7.2 miscompiles __builtin_bit_cast applied to an ext-vector ELEMENT)
            const f32x4 f = {acc[i][j][4 * q4], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]};
            const u32x4 v = __builtin_bit_cast(u32x4, f);
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, (((wave * TM + i) * TN + j) * 4 + q4) * 1024 + lane * 16, 0, 16 /* sc1 */);
          }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
    __syncthreads();
    unsigned* lds_word = reinterpret_cast<unsigned*>(smem);
    // release / acquire at agent scope around the ticket (the slabs cross XCDs, i.e. L2s): every wave's slab stores
    // have drained and the barrier has ordered them before thread 0's release (which writes this XCD's L2 back);
    // in the last arriver EVERY wave acquires (invalidates its view) before it reads the other parts
    if (tid == 0)
      *lds_word = __hip_atomic_fetch_add(&a.sk_tickets[tile], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned ticket = *lds_word;
    finish = (ticket == (unsigned)(nparts - 1));
    if (finish) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const int own = wl - w_first;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const size_t off = (size_t)((((wave * TM + i) * TN + j) * 4 + q4) * 1024 + lane * 16);
            f32x4 mine4 = {acc[i][j][4 * q4], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]};
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            for (int p = 0; p < nparts; ++p) {   // fixed order: part 0 + part 1 + ... (mine from registers)
              f32x4 v = mine4;
              if (a.dbg == 1 && p != own) { v = f32x4{0.f, 0.f, 0.f, 0.f}; }            // diagnostic: own part only
              else if (a.dbg == 2 && p == own) { v = f32x4{0.f, 0.f, 0.f, 0.f}; }       // diagnostic: the other parts only
              else if (p != own) v = *reinterpret_cast<const f32x4*>(slab_of(w_first + p) + off);   // (behind the acquire)
              t = (p == 0) ? v : t + v;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][4 * q4 + e] = t[e];
          }
      if (tid == 0) __hip_atomic_store(&a.sk_tickets[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();   // (lds_word is reused by the epilogue)
    }
  }
  if constexpr (STAMPS) {
    if (finish) {
      unsigned long long es[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      auto sf = [&](int i) {
        __builtin_amdgcn_sched_barrier(0);
        es[i] = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
      };
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long e_begin = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_waitcnt(0xC07F);
      planes_epilogue<BM, BN, WGM, WGN, LDS_TOTAL, 0>(a, acc, smem, m0, n0, tile_m, wm, wn, lane, tid, sf);
      // per-workgroup sums of the epilogue sections -> second record (offset 8 x gridDim.x)
      epi_sum[0] += es[0] - e_begin;   // row offsets + barrier
      epi_sum[1] += es[1] - es[0];     // scale loads
      epi_sum[2] += es[2] - es[1];     // LDS staging, half 0
      epi_sum[3] += es[3] - es[2];     // stores, half 0
      epi_sum[4] += es[4] - es[3];     // LDS staging, half 1
      epi_sum[5] += es[5] - es[4];     // stores, half 1
      ++epi_count;
    }
  } else if constexpr (GEO == 1) {
    const EpiGeom pg = EpiGeom{a.M, a.Hg, a.Wg, 0, 0, pn, py0, px0, 0};
    planes_epilogue<BM, BN, WGM, WGN, LDS_TOTAL, 0, NoStamp, true, BNRED>(a, acc, smem, m0, n0, tile_m, wm, wn, lane, tid, NoStamp(), &pg);
  } else if constexpr (!SPLIT) {
    if (finish) planes_epilogue<BM, BN, WGM, WGN, LDS_TOTAL, 0, NoStamp, false, BNRED>(a, acc, smem, m0, n0, tile_m, wm, wn, lane, tid);
  }
  __syncthreads();   // the next part's DMAs overwrite the LDS the epilogue used
  if constexpr (STAMPS) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the diagnostic build waits for its output stores)
    take_stamp(3);
    phase_sum[0] += stamp[1] - stamp[0];
    phase_sum[1] += stamp[2] - stamp[1];
    phase_sum[2] += stamp[3] - stamp[2];
    ++nparts_done;
  }
  }  // parts of this workgroup
  if constexpr (STAMPS) {
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0 && a.stamps != nullptr) {
      unsigned long long* o = a.stamps + (size_t)blockIdx.x * 8;
      o[0] = t_start;
      o[1] = phase_sum[0];   // prologues
      o[2] = phase_sum[1];   // main loops
      o[3] = phase_sum[2];   // drains, combines, epilogues
      o[4] = t_end;
      o[5] = rt0;
      o[6] = rt1;
      o[7] = (unsigned long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) |   // HW_REG_XCC_ID
             ((unsigned long long)nparts_done << 8) | ((unsigned long long)gridDim.x << 32);
      unsigned long long* o2 = a.stamps + ((size_t)gridDim.x + blockIdx.x) * 8;
      for (int i = 0; i < 6; ++i) o2[i] = epi_sum[i];
      o2[6] = (unsigned long long)epi_count;
      o2[7] = 0;
    }
  }
}

// largest window (in pixels) any tile of BM consecutive output pixels needs
static int win_pixels_needed(const GatherConvArgs& a, int BM) {
  const long long HW = (long long)a.Hs * a.Ws;
  const int P1 = a.Ws + 1;
  auto u_of = [&](long long m) -> long long {
    const long long n = m / HW, rem = m - n * HW;
    const long long y = rem / a.Ws;
    return (n * (a.Hs + 1) + y + 1) * P1 + (rem - y * a.Ws) + 1;
  };
  // (a walk over every row tile: 676 iterations for the 52 x 52 bs-32 layers. It is host time on the launch path, so the
  // answer is remembered per geometry; a handful of distinct layers exist per model)
  struct Key { long long M; int Hs, Ws, BM, need; };
  static thread_local Key cache[32];
  static thread_local int ncache = 0;
  for (int i = 0; i < ncache; ++i)
    if (cache[i].M == a.M && cache[i].Hs == a.Hs && cache[i].Ws == a.Ws && cache[i].BM == BM) return cache[i].need;
  long long worst = 0;
  for (long long m0 = 0; m0 < a.M; m0 += BM) {
    const long long ml = (m0 + BM - 1 < a.M) ? m0 + BM - 1 : a.M - 1;
    const long long need = u_of(ml) - u_of(m0) + 2 * P1 + 3;
    if (need > worst) worst = need;
  }
  if (ncache < 32) cache[ncache++] = Key{a.M, a.Hs, a.Ws, BM, (int)worst};
  return (int)worst;
}

bool conv_win_supported(const GatherConvArgs& a) {
  if (a.ntaps != 9 || a.sy != 1 || a.sx != 1 || a.Hg != a.Hs || a.Wg != a.Ws) return false;
  if ((a.Cs % 16) != 0 || a.Cout < 128 || a.ldw != 9 * a.Cs) return false;
  for (int t = 0; t < 9; ++t)
    if (a.taps[t].oy < -1 || a.taps[t].oy > 1 || a.taps[t].ox < -1 || a.taps[t].ox > 1 || a.taps[t].woff != t * a.Cs)
      return false;
  if ((long long)a.N * (a.Hs + 1) * (a.Ws + 1) + 2LL * a.Ws + 8 > 0x3fffffffLL) return false;
  return true;
}

// Workspace of the stream-K form (yolo_set_conv_workspace): [tickets: SK_TICKETS x u32, zero between launches]
// [slabs: 2 per workgroup]. Calls that use it must be ordered on one stream.
constexpr size_t SK_TICKETS = 1 << 18;
static void* g_sk_ws = nullptr;
static size_t g_sk_bytes = 0;
int set_conv_workspace(void* p, size_t bytes, hipStream_t st) {
  g_sk_ws = nullptr;
  g_sk_bytes = 0;
  if (p == nullptr || bytes <= SK_TICKETS * 4) return YOLO_OK;
  if (hipMemsetAsync(p, 0, SK_TICKETS * 4, st) != hipSuccess) {
    set_error("set_conv_workspace: hipMemsetAsync failed");
    return YOLO_ERR_LAUNCH;
  }
  g_sk_ws = p;
  g_sk_bytes = bytes;
  return YOLO_OK;
}


// ---- split-K for launches that leave most of the chip idle (bs-1 inference: 16 tiles of 288 stages on 256 CUs, every
// workgroup a latency-bound stream of 8 KB stages) ----
// The tiles are computed by `split_parts` workgroups each (equal runs of 16-channel blocks, conv_win_kernel and
// gather_conv_planes_kernel<128,128,2,2>), every part stores its accumulators to its own slab, and this kernel adds
// the parts of a tile IN PART ORDER (bitwise reproducible), then does what planes_epilogue does: unscale, bias,
// optional fused BatchNorm + activation (+ residual), optional accumulate, per-channel max|y|. One thread per 16-byte
// accumulator piece (4 rows x 1 column): the loads of all parts are independent, the stores of a wave cover 128-byte
// row pieces. (The stream-K form above lets the LAST ARRIVER of a tile read the other parts one after the other:
// fine for 4 parts, 190 us for 32.)
template <int BM>
__global__ __launch_bounds__(256) void conv_split_reduce_kernel(const GatherConvArgs a) {
  constexpr int BN = 128, WGN = 2, TM = 2, TN = 2;
  constexpr int QPT = BM * BN / 4;   // 16-byte pieces per tile
  constexpr int BPT = QPT / 256;     // workgroups per tile
  const int tile = blockIdx.x / BPT;
  const int qi = (blockIdx.x - tile * BPT) * 256 + threadIdx.x;
  const int lane = qi & 63, q4 = (qi >> 6) & 3, j = (qi >> 8) & 1, i = (qi >> 9) & 1, wave = qi >> 10;
  const int wm = wave / WGN, wn = wave % WGN;
  const int tile_n = tile % a.tiles_n, tile_m = tile / a.tiles_n;
  const int col = tile_n * BN + (wn * TN + j) * 32 + (lane & 31);
  const long long mrow = (long long)tile_m * BM + (wm * TM + i) * 32 + 8 * q4 + 4 * (lane >> 5);
  // (a workgroup = the four q4 pieces of ONE 32 x 32 accumulator block: 32 columns x 32 rows)
  // (one-pass inference units: the three scalars of the a-priori bound are fetched first, beside the slab loads)
  // (one-pass inference units: the words of the a-priori bound are fetched first, beside the slab loads; every wave folds
  // a quarter of them, the workgroup's maximum is formed behind the barrier further down)
  float ib = 0.f, rb = 0.f;
  if (a.out_planes != nullptr) {
    for (int w = threadIdx.x; w < a.pl_in_n; w += 256) ib = fmaxf(ib, __builtin_bit_cast(float, a.pl_in_bound[w]));
    if (a.pl_res_bound != nullptr)
      for (int w = threadIdx.x; w < a.pl_res_n; w += 256) rb = fmaxf(rb, __builtin_bit_cast(float, a.pl_res_bound[w]));
  }
  const bool active = col < a.Cout && mrow < a.M;
  float mx = 0.f, mxf = 0.f;              // max|.| before the residual (absmax) / of the values written
  float vout[4] = {0.f, 0.f, 0.f, 0.f};   // this thread's four rows of its column (0 where the row does not exist)
  float st1 = 0.f, st2 = 0.f;             // BatchNorm statistics of the thread's (existing) rows: sum, sum of squares
  if (active) {
  // everything this thread needs besides the slabs is fetched first (this kernel is a chain of memory latencies: the
  // per-column parameters and the residual values fly beside the parts instead of behind them)
  const float bv = a.bias != nullptr ? a.bias[col] : 0.f;
  const bool fused = a.epi_scale != nullptr;
  const float esc = fused ? a.epi_scale[col] : 1.f, esh = fused ? a.epi_shift[col] : 0.f;
  const float unscale =
      reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.src) + a.src_bytes - PL_HEADER)[2] *
      reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.wgt) + a.wgt_bytes - PL_HEADER)[2];
  const bool dense = a.osy == 1 && a.osx == 1 && a.ooy == 0 && a.oox == 0 && a.Hd == a.Hg && a.Wd == a.Wg;
  const int HgWg = a.Hg * a.Wg;
  long long offs[4];
  float rv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const long long m = mrow + e;
    offs[e] = -1;
    if (m < a.M) {
      if (dense) {
        offs[e] = m * a.Cd;
      } else {
        const int n = (int)(m / HgWg);
        const int rem = (int)(m - (long long)n * HgWg);
        const int y = rem / a.Wg;
        const int x = rem - y * a.Wg;
        offs[e] = (((long long)n * a.Hd + (y * a.osy + a.ooy)) * a.Wd + (x * a.osx + a.oox)) * a.Cd;
      }
      if (a.epi_res != nullptr) rv[e] = a.epi_res[offs[e] + col];
      if (a.accumulate) rv[e] += a.dst[offs[e] + col];
    }
  }
  const int P = a.split_parts;
  const f32x4* sl = reinterpret_cast<const f32x4*>(a.sk_slabs) + (size_t)tile * P * QPT + qi;
  // the loads of up to 16 parts fly together (each is a miss: the slabs were written a kernel ago by other CUs; four at a
  // time made this kernel a chain of eight round trips, 21 us); the additions stay in part order
  f32x4 t = {0.f, 0.f, 0.f, 0.f};
  int p = 0;
  for (; p + 16 <= P; p += 16) {
    f32x4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = sl[(size_t)(p + u) * QPT];
#pragma unroll
    for (int u = 0; u < 16; ++u) t = t + v[u];
  }
  for (; p + 4 <= P; p += 4) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = sl[(size_t)(p + u) * QPT];
#pragma unroll
    for (int u = 0; u < 4; ++u) t = t + v[u];
  }
  for (; p < P; ++p) t = t + sl[(size_t)p * QPT];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (offs[e] >= 0) {
      float v = fmaf(t[e], unscale, bv);
      if (fused) v = act_fwd(fmaf(esc, v, esh), a.epi_act);
      const float vstat = v;     // max|.| before the residual, as planes_epilogue
      st1 += vstat;
      st2 = fmaf(vstat, vstat, st2);
      v += rv[e];                // (residual and / or the accumulate form's old value; 0 otherwise)
      a.dst[offs[e] + col] = v;
      vout[e] = v;
      mx = fmaxf(mx, fabsf(fused ? vstat : v));
      mxf = fmaxf(mxf, fabsf(v));
    }
  }
  }
  if (a.stats != nullptr) {
    // BatchNorm statistics of the training-mode forward (round 6: split launches may carry them -- until then every
    // forward launch with statistics ran one workgroup per tile however few tiles it had: 120 us per 3x3 layer of YOLOv1.5
    // at bs 4, 16 tiles of 576 stages). The workgroup's 32 x 32 block: eight threads share a column (fixed order), one
    // fp64 atomic pair per column into the replica slot of the block's row block, as planes_epilogue does per tile.
    __shared__ float ss1[256], ss2[256];
    ss1[threadIdx.x] = st1;
    ss2[threadIdx.x] = st2;
    __syncthreads();
    if (threadIdx.x < 32 && col < a.Cout) {
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a1 += ss1[threadIdx.x + 32 * u];
        a2 += ss2[threadIdx.x + 32 * u];
      }
      double* slot = a.stats + (long long)((tile_m * (BM / 32) + wm * TM + i) & (YOLO_BN_STAT_SLOTS - 1)) * 2 * a.Cout;
      atomicAdd(&slot[col], (double)a1);
      atomicAdd(&slot[a.Cout + col], (double)a2);
    }
  }
  if (a.absmax != nullptr && a.out_planes == nullptr) {   // (with planes going out nobody reads the per-channel maxima)
    // one atomic per column and workgroup (256 same-address atomics from eight XCDs per column made this kernel 21 us)
    __shared__ float smx[256];
    smx[threadIdx.x] = mx;
    __syncthreads();
    if (threadIdx.x < 32) {
      float m8 = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) m8 = fmaxf(m8, smx[threadIdx.x + 32 * u]);
      if (col < a.Cout && __builtin_bit_cast(unsigned, m8) > a.absmax[col])
        atomicMax(&a.absmax[col], __builtin_bit_cast(unsigned, m8));
    }
  }
  if (a.out_planes != nullptr) {
    // The finished 32 x 32 block also goes out as planes (dense outputs: pixel m is row m of the tensor): through LDS,
    // 128 threads take one (row, 8-channel group) unit each. Scale from the a-priori bound (GatherConvArgs::pl_pred):
    // it is looser than max|dst| by the ratio of a filter's l1 norm to the dot products it actually produces (2^6..2^9
    // here), which moves the absolute error floor of the format from 2^-40 to ~2^-32 of the bound -- still far below
    // fp32's own rounding (planes.hpp)
    __shared__ float t32[32][33];
    __shared__ float s_mxf[3][4];
    const int rl = 8 * q4 + 4 * (lane >> 5);
#pragma unroll
    for (int e = 0; e < 4; ++e) t32[rl + e][lane & 31] = vout[e];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mxf = fmaxf(mxf, __shfl_xor(mxf, o, 64));
      ib = fmaxf(ib, __shfl_xor(ib, o, 64));
      rb = fmaxf(rb, __shfl_xor(rb, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      s_mxf[0][threadIdx.x >> 6] = mxf;
      s_mxf[1][threadIdx.x >> 6] = ib;
      s_mxf[2][threadIdx.x >> 6] = rb;
    }
    __syncthreads();
    const float in_b = fmaxf(fmaxf(s_mxf[1][0], s_mxf[1][1]), fmaxf(s_mxf[1][2], s_mxf[1][3]));
    const float res_b = fmaxf(fmaxf(s_mxf[2][0], s_mxf[2][1]), fmaxf(s_mxf[2][2], s_mxf[2][3]));
    const float bnd = (a.pl_pred[0] * in_b + a.pl_pred[1] + res_b) * 1.001f + 1e-30f;
    const float psc = planes_scale_from_bound(__builtin_bit_cast(unsigned, bnd));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      unsigned* header = reinterpret_cast<unsigned*>(a.out_planes + planes_body_bytes(a.M, a.Cout));
      header[0] = __builtin_bit_cast(unsigned, bnd);
      reinterpret_cast<float*>(header)[1] = psc;
      reinterpret_cast<float*>(header)[2] = 1.f / psc;
    }
    if (threadIdx.x < 128) {
      const int r = threadIdx.x >> 2, g = threadIdx.x & 3;
      const long long m = (long long)tile_m * BM + (wm * TM + i) * 32 + r;
      const int c0 = tile_n * BN + (wn * TN + j) * 32 + g * 8;
      if (m < a.M && c0 < a.Cout) {
        const f32x4 o0 = {t32[r][g * 8 + 0], t32[r][g * 8 + 1], t32[r][g * 8 + 2], t32[r][g * 8 + 3]};
        const f32x4 o1 = {t32[r][g * 8 + 4], t32[r][g * 8 + 5], t32[r][g * 8 + 6], t32[r][g * 8 + 7]};
        store_planes8(a.out_planes, m, c0 >> 3, a.Cout, o0, o1, psc);
      }
    }
    // max|dst| of this workgroup's block: ONE word per workgroup, a plain store (the consumers take the maximum of the
    // launch's words: GatherConvArgs::pl_out_words)
    if (threadIdx.x == 0)
      a.pl_out_words[blockIdx.x] = __builtin_bit_cast(unsigned, fmaxf(fmaxf(s_mxf[0][0], s_mxf[0][1]), fmaxf(s_mxf[0][2], s_mxf[0][3])));
  }
}

// Parts per tile for a launch of nb tiles of bm x 128 (1 = do not split). Only launches (since round 6 also those with
// BatchNorm statistics: conv_split_reduce_kernel makes them) whose tiles would fill less than 1/idle_div of the chip's workgroup slots (window kernel: half -- the 13x13 data
// gradients at bs 32, 172 tiles of 576 stages: 181 -> 169 us with two parts; per-tap kernel: a quarter -- the 13x13 1x1
// layers at bs 32 have 64 stages per tile and lose 8 % with two parts), each part at least min_cb channel blocks long, as many parts as
// it takes to give every CU two workgroups, at most 32; needs the workspace of yolo_set_conv_workspace.
int conv_split_parts(const GatherConvArgs& a, long long nb, int bm, int min_cb, int idle_div) {
  init_options();
  // (a.bwd_y: the fused BatchNorm-backward reduction lives in planes_epilogue only, not in conv_split_reduce_kernel)
  if (g_opt[OPT_CONV_SK] != 1 || a.bwd_y != nullptr || g_sk_ws == nullptr) return 1;
  if (a.stats != nullptr && (g_opt[OPT_EXP] & 16)) return 1;   // (A/B: forward launches with statistics unsplit, as until round 5)
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      cus = 256;
  }
  const long long slots = 2LL * cus;
  if (nb * idle_div > slots) return 1;
  long long P = (a.Cs >> 4) / min_cb;
  if (P > slots / nb) P = slots / nb;
  if (P > 32) P = 32;
  if (P < 2) return 1;
  if (SK_TICKETS * 4 + (size_t)(nb * P) * bm * 128 * 4 > g_sk_bytes) return 1;
  return (int)P;
}

int launch_split_reduce(GatherConvArgs& a, int bm, hipStream_t st) {
  if (bm == 128)
    hipLaunchKernelGGL(conv_split_reduce_kernel<128>, dim3((unsigned)a.nblocks * (128 * 128 / 4 / 256)), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(conv_split_reduce_kernel<256>, dim3((unsigned)a.nblocks * (256 * 128 / 4 / 256)), dim3(256), 0, st, a);
  return check_launch("conv_split_reduce_kernel");
}

float* conv_split_slabs() { return reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(g_sk_ws) + SK_TICKETS * 4); }

template <int WGM, int NCH>
static int launch_win(GatherConvArgs& a, hipStream_t st) {
  constexpr int BM = 64 * WGM;
  const long long tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.Cout + 127) / 128;
  const long long nb = tiles_m * a.tiles_n;
  if (nb <= 0 || nb * (a.Cs >> 4) > 0x7fffffffLL) {
    set_error("conv(window): bad grid %lld", nb);
    return YOLO_ERR_INVALID_ARG;
  }
  a.nblocks = (int)nb;
  a.bwd_nslots = (int)tiles_m;
  YOLO_BNRED_CHECK(a)
  constexpr size_t lds = 2 * NCH * 4096 + 3 * 8192;
  static bool attr_set = false;
  static int resident = 0;   // workgroups the chip holds at once (occupancy x CUs)
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_win_kernel<WGM, NCH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&conv_win_kernel<WGM, NCH>),
                                                     128 * WGM, lds) == hipSuccess &&
        hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
      resident = per_cu * cus;
  }
  // filters of 8 MB and more (the 13x13 3x3 layers): row tile fastest. Measured on the 13x13x1024->512 data gradient
  // (172 tiles, each streaming 4.7 MB of filters): 977 MB beyond-L2 per launch = 6 TB/s with the column tile fastest.
  {
    const long long wbytes = (long long)a.Cout * a.ldw * 4;
    a.tile_order = g_opt[OPT_TILE_ORDER] == 0 ? (wbytes >= (8LL << 20) ? 1 : 0) : g_opt[OPT_TILE_ORDER] - 1;
  }
  // stream-K: as many workgroups as the chip holds, each an equal share of the (tile, channel block) units
  a.sk_grid = 0;
  a.dbg = g_opt[OPT_DBG];
  unsigned grid = (unsigned)nb;
  // Policy (YOLO_CONV_SK / yolo_set_option key 2): 1 = automatic -- only launches that would leave most of the chip
  // idle (fewer tiles than half the resident workgroups AND at least 32 channel blocks per tile: the bs-1 inference
  // forward of the 13x13 layers, 16 tiles of 288 stages), split into at most 4 parts per tile (the last arriver reads the other parts'
  // slabs: 64 KB each); at training sizes one workgroup per tile measured faster (the slab round trips cost more
  // than the ragged last round). A value > 1 forces that many workgroups (benchmarks).
  // (measured at bs 1: 13x13x512->1024, 16 tiles of 288 stages: 66 us -> 41 us with 4 parts per tile, 54 us with 8,
  // 190 us with 32 -- the last arriver reads the other parts one after the other; layers with fewer than 32 channel
  // blocks (26x26: 41 us, 52x52: 24 us) do not gain at any split)
  // YOLO_CONV_SK=1 (default): split-K with the reduce kernel (conv_split_parts); -1: the stream-K policy above
  if (a.bwd_y != nullptr) {   // the fused BatchNorm-backward reduction: one workgroup per tile, its own instantiation
    static bool attr_bn = false;
    if (!attr_bn) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_win_kernel<WGM, NCH, false, false, false, 2, 0, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_bn = true;
    }
    a.split_parts = 1;
    hipLaunchKernelGGL((conv_win_kernel<WGM, NCH, false, false, false, 2, 0, true>), dim3(grid), dim3(128 * WGM), lds, st, a);
    return check_launch("conv_win_kernel(bn reduce)");
  }
  // (parts of at least 2 channel blocks = 18 stages; YOLO_WIN_SPLIT_MIN_CB = 1: twice the parts, half the stages)
  static const int split_min_cb = [] { const char* e = getenv("YOLO_WIN_SPLIT_MIN_CB"); return e ? atoi(e) : 2; }();
  a.split_parts = WGM == 2 ? conv_split_parts(a, nb, BM, split_min_cb, 2) : 1;
  if (a.split_parts > 1) {
    a.tile_order = 0;
    a.sk_grid = (int)(nb * a.split_parts);
    a.sk_slabs = conv_split_slabs();
    if constexpr (WGM == 2) {
      static bool attr_split = false;
      if (!attr_split) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_win_kernel<WGM, NCH, false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_split = true;
      }
      hipLaunchKernelGGL((conv_win_kernel<WGM, NCH, false, true>), dim3((unsigned)a.sk_grid), dim3(128 * WGM), lds, st, a);
    }
    if (int rc = check_launch("conv_win_kernel(split)")) return rc;
    return launch_split_reduce(a, BM, st);
  }
  const bool sk_auto = g_opt[OPT_CONV_SK] == -1 && nb * 2 < resident && (a.Cs >> 4) >= 32;
  if ((sk_auto || g_opt[OPT_CONV_SK] > 1) && resident > 0 && g_sk_ws != nullptr && nb <= (long long)SK_TICKETS) {
    long long G = g_opt[OPT_CONV_SK] > 1 ? g_opt[OPT_CONV_SK] : resident;
    const long long units = nb * (a.Cs >> 4);
    if (sk_auto && G > nb * 4) G = nb * 4;
    if (G > units) G = units;
    if (SK_TICKETS * 4 + (size_t)G * 2 * BM * 128 * 4 <= g_sk_bytes && G >= 1) {
      a.sk_grid = (int)G;
      a.sk_tickets = reinterpret_cast<unsigned*>(g_sk_ws);
      a.sk_slabs = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(g_sk_ws) + SK_TICKETS * 4);
      grid = (unsigned)G;
    }
  }
  if constexpr (WGM == 2 && NCH == 5) {   // the one stamped instantiation (diagnostics on the 52x52 layers)
    if (g_opt[OPT_STAMPS] != 0 && g_dbg_buf != nullptr && g_dbg_bytes >= (size_t)nb * 128) {
      static bool attr2 = false;
      if (!attr2) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_win_kernel<WGM, NCH, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr2 = true;
      }
      a.stamps = reinterpret_cast<unsigned long long*>(g_dbg_buf);
      hipLaunchKernelGGL((conv_win_kernel<WGM, NCH, true>), dim3(grid), dim3(128 * WGM), lds, st, a);
      return check_launch("conv_win_kernel(stamps)");
    }
  }
  if (a.sk_grid > 0) {
    static bool attr_sk = false;
    if (!attr_sk) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_win_kernel<WGM, NCH, false, false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_sk = true;
    }
    hipLaunchKernelGGL((conv_win_kernel<WGM, NCH, false, false, true>), dim3(grid), dim3(128 * WGM), lds, st, a);
    return check_launch("conv_win_kernel(stream-K)");
  }
  hipLaunchKernelGGL((conv_win_kernel<WGM, NCH>), dim3(grid), dim3(128 * WGM), lds, st, a);
  return check_launch("conv_win_kernel");
}

// ---- patch geometry (GEO = 1): rows longer than 64 pixels, and 64-column tiles for Cout <= 64 ----
static bool conv_patch_supported(const GatherConvArgs& a) {
  if (a.ntaps != 9 || a.sy != 1 || a.sx != 1 || a.Hg != a.Hs || a.Wg != a.Ws) return false;
  if (a.osy != 1 || a.osx != 1 || a.ooy != 0 || a.oox != 0 || a.Hd != a.Hg || a.Wd != a.Wg) return false;
  if ((a.Cs % 16) != 0 || a.Cout < 64 || a.ldw != 9 * a.Cs) return false;
  for (int t = 0; t < 9; ++t)
    if (a.taps[t].oy < -1 || a.taps[t].oy > 1 || a.taps[t].ox < -1 || a.taps[t].ox > 1 || a.taps[t].woff != t * a.Cs)
      return false;
  return true;
}

template <int WGM, int NCH, int WGN>
static int launch_patch(GatherConvArgs& a, hipStream_t st) {
  constexpr int BM = 64 * WGM, BN = 64 * WGN, PH = BM / 16;
  const long long tiles_m = (long long)a.N * ((a.Hg + PH - 1) / PH) * ((a.Wg + 15) / 16);
  a.tiles_n = (a.Cout + BN - 1) / BN;
  const long long nb = tiles_m * a.tiles_n;
  if (nb <= 0 || nb > 0x7fffffffLL) {
    set_error("conv(patch window): bad grid %lld", nb);
    return YOLO_ERR_INVALID_ARG;
  }
  a.nblocks = (int)nb;
  a.bwd_nslots = (int)tiles_m;
  YOLO_BNRED_CHECK(a)
  constexpr size_t lds = 2 * NCH * 4096 + 3 * 4096 * WGN;
  auto kern = &conv_win_kernel<WGM, NCH, false, false, false, WGN, 1>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  a.tile_order = 0;   // column tile fastest: the column tiles of a patch share its window in L2
  a.sk_grid = 0;
  a.split_parts = 1;
  a.dbg = g_opt[OPT_DBG];
  if (a.bwd_y != nullptr) {
    auto kern_bn = &conv_win_kernel<WGM, NCH, false, false, false, WGN, 1, true>;
    static bool attr_bn = false;
    if (!attr_bn) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern_bn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_bn = true;
    }
    hipLaunchKernelGGL(kern_bn, dim3((unsigned)nb), dim3(64 * WGM * WGN), lds, st, a);
    return check_launch("conv_win_kernel(patch, bn reduce)");
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(64 * WGM * WGN), lds, st, a);
  return check_launch("conv_win_kernel(patch)");
}

// Policy of the patch form (YOLO_CONV_PATCH / yolo_set_option key 5): 0 = off, 1 = automatic (rows longer than 64
// pixels; Cout <= 64 at any row length; launches of at least 256 tiles -- smaller ones keep the per-tap kernel and its
// split-K), 2 = wherever the shape allows (tests, benchmarks).
static int launch_conv_patch(GatherConvArgs& a, hipStream_t st) {
  const int mode = g_opt[OPT_CONV_PATCH];
  if (mode == 0 || !conv_patch_supported(a)) return 1;
  const bool narrow = a.Cout <= 64;
  if (mode == 1) {
    if (!narrow && a.Ws <= 64) return 1;   // the linear window covers these
    const long long tiles = narrow ? (long long)a.N * ((a.Hg + 15) / 16) * ((a.Wg + 15) / 16)
                                   : (long long)a.N * ((a.Hg + 7) / 8) * ((a.Wg + 15) / 16) * ((a.Cout + 127) / 128);
    if (tiles < 256) return 1;
  }
  return narrow ? launch_patch<4, 6, 1>(a, st) : launch_patch<2, 3, 2>(a, st);
}

// variant: 0 = automatic, 2 = 128x128 tiles (4 waves), 4 = 256x128 tiles (8 waves). Returns 1 when no window
// kernel covers the shape (the caller then uses the per-tap streaming kernel), 0 on success, < 0 on error.
int launch_conv_win(GatherConvArgs& a, int variant, hipStream_t st) {
  if (variant != 2 && variant != 4) {
    const int rc = launch_conv_patch(a, st);
    if (rc <= 0) return rc;
  }
  if (!conv_win_supported(a)) return 1;
  int wgm = variant;
  if (wgm != 2 && wgm != 4) {
    // automatic (measured on YOLOv3-416 bs 32, scripts/win_sweep.sh): rows of more than 64 pixels need windows that
    // leave one workgroup per CU -> the per-tap kernel. 128x128 tiles everywhere: the 256x128 tiles (8 waves) that round-2's
    // first builds used where 128x128 tiles fill between one and two workgroups per CU (338-344 tiles) are 0.4 % of the
    // step SLOWER since the production kernel lost its stream-K code (same-box A/B: 941 vs 945 images/s)
    if (a.Ws > 64) return 1;
    wgm = 2;
  }
  const int need = win_pixels_needed(a, 64 * wgm);
  const int nch = (need + 63) / 64;
  if (wgm == 2) {
    switch (nch) {
      case 1: case 2: case 3: return launch_win<2, 3>(a, st);
      case 4: return launch_win<2, 4>(a, st);
      case 5: return launch_win<2, 5>(a, st);
      case 6: return launch_win<2, 6>(a, st);
      case 7: return launch_win<2, 7>(a, st);
      default: return 1;
    }
  }
  switch ((nch + 1) / 2) {
    case 1: case 2: return launch_win<4, 4>(a, st);
    case 3: return launch_win<4, 6>(a, st);
    case 4: return launch_win<4, 8>(a, st);
    case 5: return launch_win<4, 10>(a, st);
    case 6: return launch_win<4, 12>(a, st);
    case 7: return launch_win<4, 14>(a, st);
    default: return 1;
  }
}

}  // namespace yolo

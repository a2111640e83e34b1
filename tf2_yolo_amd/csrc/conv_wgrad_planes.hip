// Filter gradient on pre-split operands ("planes" format, conv_planes.hip) staged by LDS-DMA:
//   dW[co][j] += sum_p dy[p][co] * x[p -> tap(j)][c(j)],  j = tap*Cin + c,
// rows = co, columns = flattened (tap, c), contraction over pixels in 16-pixel stages, split over
// grid.y and combined with fp32 atomics (as wgrad_split_kernel). Both operands are pixel-major, the
// MFMA wants 8 consecutive pixels per lane: the fragments come from the transposing LDS read
// ds_read_b64_tr_b16, which per 16-lane group turns a 4(pixel) x 16(channel) block into "4 pixels of MY
// channel" -- and a 256-byte sub-block of the planes format IS a 16(pixel) x 8(channel) matrix. Three fp16
// MFMA passes (l*h, h*l, h*h) per fragment pair; the epilogue undoes the two power-of-two operand scales.
//
// One DMA piece (64 lanes x 16 B = 1 KiB) = 16 pixels x 32 channels of one plane = 4 sub-blocks.
// Lane l fetches the unit (pixel 4*(l>>4) + (l&3), sub-block (l>>2)&3): quads of lanes read 64 contiguous
// bytes (4 pixels of a sub-block), and the lane-linear LDS image [pixel quad][sub-block][pixel] puts the
// 16 units a 32-lane transposed read touches (4 sub-blocks x 4 pixels) on 16 distinct 16-byte bank
// groups: conflict-free. Same wave roles, ring of 3 stage buffers, 2 fragment register sets and barrier /
// vmcnt discipline as gather_conv_planes_kernel.
#include "planes.hpp"
#include <type_traits>
#include <cstdlib>

namespace yolo {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// KO: diagnostic knock-outs (wrong results; YOLO_WGRAD_KO): 1 = no operand DMAs, 2 = no fragment reads, 4 = no MFMAs
template <int BM, int BN, int WGM, int WGN, int NB = 3, int KO = 0>
__global__ __launch_bounds__(64 * WGM * WGN, (NB * (BM / 32 + BN / 32) * PL_PLANES * 1024 > 80 * 1024 ? 1 : 2)) void wgrad_planes_kernel(const WgradArgs a) {
  constexpr int NW = WGM * WGN;
  constexpr int TM = BM / WGM / 32;
  constexpr int TN = BN / WGN / 32;
  constexpr int RBA = BM / 32, RBB = BN / 32;
  constexpr int STAGE_BYTES = (RBA + RBB) * PL_PLANES * 1024;
  constexpr int NBUF = NB;   // stages of the DMA ring (NB - 1 in flight); 1x1 layers, pure streams, take a deeper one

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds_base = (unsigned)(size_t)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  // 1-D grid, XCD-aware: blocks b, b+8, ... share an XCD (round-robin dispatch), so give every XCD a contiguous
  // run of (pixel chunk, tile) pairs with the tile index fastest: the tiles_co x tiles_j workgroups that contract
  // the SAME pixel chunk then sit on one XCD and fetch its x / dy rows once through that L2. (A 2-D grid dealt them
  // to all eight XCDs: 867 MB beyond-L2 traffic per launch on the 52x52x128->256 layer against 133 MB of operands,
  // i.e. HBM-bound at 4.3 TB/s; profiles/r02_a_conv_pmc.json.)
  const int lid = xcd_remap(blockIdx.x, a.nblocks);
  const int tiles = a.tiles_co * a.tiles_j;
  const int split = lid / tiles;
  const int tile = lid - split * tiles;
  const int tile_co = tile % a.tiles_co;
  const int j0 = (tile / a.tiles_co) * BN;
  const int co0 = tile_co * BM;
  const long long p_begin = (long long)split * a.chunk;   // multiple of 16
  long long p_end = p_begin + a.chunk;
  if (p_end > a.M) p_end = a.M;
  if (p_begin >= p_end) return;
  const int nk = (int)((p_end - p_begin + 15) / 16);
  const int Ktot = a.ntaps * a.Cs;

  // ---- loader role(s): lane -> (pixel of the stage, sub-block of the 32-channel block). An 8-wave
  // workgroup gives every wave ONE 32-channel block of dy (waves < RBA) or of x; a 4-wave workgroup gives every wave
  // RBA / 4 blocks of dy and RBB / 4 blocks of x (128x128: one of each; 128x256: one and two). ----
  constexpr bool BOTH = (RBA + RBB > NW);
  constexpr int NA = BOTH ? RBA / NW : 1, NQ = BOTH ? RBB / NW : 1;
  static_assert(!BOTH || (RBA % NW == 0 && RBB % NW == 0), "loader layout");
  constexpr int ND = PL_PLANES * (BOTH ? NA + NQ : 1);   // DMA instructions per wave per stage
  const bool hasA = BOTH || wave < RBA, hasB = BOTH || wave >= RBA;
  const int lpix = 4 * (lane >> 4) + (lane & 3);
  const int lsb = (lane >> 2) & 3;
  const unsigned strideA = (unsigned)((a.Cout >> 4) * PL_RECORD);   // bytes per 16-pixel block of dy planes
  const unsigned strideB = (unsigned)((a.Cs >> 4) * PL_RECORD);
  const unsigned zeroA = (unsigned)a.zero_blk_dy * strideA, zeroB = (unsigned)a.zero_blk_src * strideB;
  const i32x4 rsrcA = planes_rsrc(a.dy, a.dy_bytes), rsrcB = planes_rsrc(a.src, a.src_bytes);
  unsigned ldsA[NA], ldsB[NQ];

  // A (dy): unit (pixel block, 16-channel block co16, half, pixel); advances one pixel block per stage
  bool okA[NA], okB[NQ];
  unsigned voffA[NA], voffB[NQ];
  // B (x): column block -> (tap, 16-channel block), pixel decoded incrementally
  int b_oy[NQ], b_ox[NQ], b_chan[NQ];
  int pn = 0, py = 0, px = 0;
  long long pcur = p_begin + lpix;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int rb = BOTH ? wave + i * NW : wave;
    ldsA[i] = lds_base + rb * PL_PLANES * 1024;
    okA[i] = false;
    voffA[i] = zeroA;
    if (hasA) {
      const int co16 = ((co0 + rb * 32) >> 4) + (lsb >> 1);
      okA[i] = co16 * 16 < a.Cout;
      voffA[i] = okA[i] ? (unsigned)(p_begin >> 4) * strideA + (unsigned)co16 * PL_RECORD + (lsb & 1) * 256 + lpix * 16 : zeroA;
    }
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int rb = BOTH ? wave + i * NW : (wave - RBA) % RBB;
    ldsB[i] = lds_base + (RBA + rb) * PL_PLANES * 1024;
    okB[i] = false;
    voffB[i] = zeroB;
    b_oy[i] = b_ox[i] = b_chan[i] = 0;
    if (hasB) {
      const int j16 = ((j0 + rb * 32) >> 4) + (lsb >> 1);
      okB[i] = j16 * 16 < Ktot;
      const int cpt = a.Cs >> 4;
      const int t = okB[i] ? j16 / cpt : 0;
      const int r = t / a.kw;
      b_oy[i] = r - a.pad_t;
      b_ox[i] = (t - r * a.kw) - a.pad_l;
      b_chan[i] = (j16 - t * cpt) * PL_RECORD + (lsb & 1) * 256;
    }
  }
  if (hasB) {
    const int HgWg = a.Hg * a.Wg;
    const long long pp = pcur < a.M ? pcur : 0;
    pn = (int)(pp / HgWg);
    const int rem = (int)(pp - (long long)pn * HgWg);
    py = rem / a.Wg;
    px = rem - py * a.Wg;
  }
  int ld_stage = 0;  // stage the loader will issue next

  auto loader_addr = [&]() {   // x: source unit of this lane's pixel for the stage about to be issued
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int ys = py * a.sy + b_oy[i], xs = px * a.sx + b_ox[i];
      const bool ok = okB[i] && (ld_stage < nk) && (pcur < p_end) && ((unsigned)ys < (unsigned)a.Hs) &&
                      ((unsigned)xs < (unsigned)a.Ws);
      const int s = (pn * a.Hs + ys) * a.Ws + xs;
      voffB[i] = ok ? ((unsigned)s >> 4) * strideB + (s & 15) * 16 + b_chan[i] : zeroB;
    }
  };
  auto loader_next = [&]() {
    ++ld_stage;
    if (hasA) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        if (ld_stage >= nk) voffA[i] = zeroA;
        else if (okA[i]) voffA[i] += strideA;
      }
    }
    if (hasB) {
      pcur += 16;
      px += 16;
      while (px >= a.Wg) {
        px -= a.Wg;
        ++py;
      }
      while (py >= a.Hg) {
        py -= a.Hg;
        ++pn;
      }
      loader_addr();
    }
  };
  // DMA d of a stage: plane d & 1 of block d >> 1 of this wave (BOTH: its dy blocks first, then its x blocks)
  auto issue_plane = [&](int d, int buf) {
    if constexpr (KO & 1) return;
    const int p = d & 1, blk = d >> 1;
    const bool useA = BOTH ? (blk < NA) : hasA;
    const int bi = BOTH ? (blk < NA ? blk : blk - NA) : 0;
    const unsigned l = __builtin_amdgcn_readfirstlane((useA ? ldsA[BOTH && blk >= NA ? 0 : bi] : ldsB[bi]) + buf * STAGE_BYTES + p * 1024);
    if (useA) dma16(rsrcA, voffA[BOTH && blk >= NA ? 0 : bi], (unsigned)(p * 512), l);
    else dma16(rsrcB, voffB[bi], (unsigned)(p * 512), l);
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

  // transposed fragment reads: 16-lane group g -> channel half g&1, pixel half g>>1; lane (qq, pp) of the
  // group supplies pixel qq, channels 4pp..4pp+3 of the half (sub-block (g&1)*2 + (pp>>1))
  const int grp = lane >> 4, i16 = lane & 15;
  const int qq = i16 >> 2, pq = i16 & 3;
  const int tr_off = ((2 * (grp >> 1)) * 16 + ((grp & 1) * 2 + (pq >> 1)) * 4 + qq) * 16 + (pq & 1) * 8;
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  auto tr_frag = [&](const unsigned char* piece) -> f16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(piece + tr_off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(piece + tr_off + 256));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(f16x8, v);
  };

  f16x8 fa[2][PL_PLANES][TM], fb[2][PL_PLANES][TN];
  auto read_frags = [&](int buf, auto SET) {
    constexpr int S = decltype(SET)::value;
    if constexpr (KO & 2) return;
    const unsigned char* sb = smem + buf * STAGE_BYTES;
#pragma unroll
    for (int p = 0; p < PL_PLANES; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[S][p][i] = tr_frag(sb + ((wm * TM + i) * PL_PLANES + p) * 1024);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[S][p][j] = tr_frag(sb + ((RBA + wn * TN + j) * PL_PLANES + p) * 1024);
    }
  };
  auto mfma_stage = [&](auto SET, int wbuf) {
    constexpr int S = decltype(SET)::value;
    constexpr int NM = TM * TN * 3;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      // q: (A plane, B plane) = (l,h) (h,l) (h,h)
      const int pa = (q == 0) ? 1 : 0;
      const int pb = (q == 1) ? 1 : 0;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (!(KO & 4)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[S][pa][i], fb[S][pb][j], acc[i][j], 0, 0, 0);
          const int idx = (q * TM + i) * TN + j;
#pragma unroll
          for (int d = 0; d < ND; ++d)
            if (idx == (((d + 1) * NM) / (ND + 1) > 0 ? ((d + 1) * NM) / (ND + 1) - 1 : 0)) {
              __builtin_amdgcn_sched_barrier(0);
              issue_plane(d, wbuf);
              __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    loader_next();
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  if (hasB) loader_addr();
#pragma unroll
  for (int b = 0; b < NBUF; ++b) issue_stage(b);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 1) * ND) : "memory");
  __builtin_amdgcn_s_barrier();
  read_frags(0, S0{});

  auto step = [&](int rbuf, int wbuf, auto CUR, auto NXT) {
    // my pieces of the next stage have landed (those of the one after may still fly)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * ND) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my reads of the current stage's buffer are done
    __builtin_amdgcn_s_barrier();
    read_frags(rbuf, NXT);
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

  // 1 / (scale of dy * scale of x): both powers of two (planes headers)
  const float unscale =
      reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.dy) + a.dy_bytes - PL_HEADER)[2] *
      reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.src) + a.src_bytes - PL_HEADER)[2];
  if (a.slabs != nullptr) {
    // reproducible form: the partial goes to this workgroup's slab in accumulator order -- 16-byte piece
    // ((wave * TM + i) * TN + j) * 4 + q4 of lane l at float (piece * 64 + l) * 4 -- as plain dwordx4 stores (one
    // instruction = 1 KiB contiguous); wgrad_reduce_kernel adds the splits of a tile in split order
    float* mine = a.slabs + (size_t)lid * (BM * BN);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const f32x4 f = {acc[i][j][4 * q4], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]};
          *reinterpret_cast<f32x4*>(mine + ((((wave * TM + i) * TN + j) * 4 + q4) * 64 + lane) * 4) = f;
        }
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int cj = j0 + (wn * TN + j) * 32 + (lane & 31);
    const bool cok = cj < Ktot;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (cok && co < a.Cout) atomicAdd(&a.dw[(long long)co * a.ldw + cj], acc[i][j][r] * unscale);
      }
    }
  }
}

// dw[co][cj] += unscale * sum over splits (in split order) of the slabs' partials: one thread per 16-byte accumulator piece
// (4 rows x 1 column), up to 8 split loads in flight, the 32 lanes of a half-wave write 128 contiguous bytes of a dw row.
// No atomics anywhere: the filter gradient is bit-identical from run to run.
template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradArgs a) {
  constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
  constexpr int PER_TILE = BM * BN / 4;            // 16-byte pieces per tile
  const int tiles = a.tiles_co * a.tiles_j;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const int tile = (int)(idx / PER_TILE);
  if (tile >= tiles) return;
  const int q = (int)(idx - (long long)tile * PER_TILE);
  const int lane = q & 63, piece = q >> 6;
  const int q4 = piece & 3, j = (piece >> 2) % TN, i = (piece / (4 * TN)) % TM, wave = piece / (4 * TN * TM);
  const int wm = wave / WGN, wn = wave % WGN;
  const int co0 = (tile % a.tiles_co) * BM, j0 = (tile / a.tiles_co) * BN;
  const int cj = j0 + (wn * TN + j) * 32 + (lane & 31);
  const int co = co0 + (wm * TM + i) * 32 + 8 * q4 + 4 * (lane >> 5);
  if (cj >= a.ntaps * a.Cs || co >= a.Cout) return;
  const f32x4* sl = reinterpret_cast<const f32x4*>(a.slabs) + (size_t)tile * PER_TILE + q;
  const size_t sstride = (size_t)tiles * PER_TILE;
  f32x4 t = {0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 8 <= a.splits; s += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(sl + (size_t)(s + u) * sstride);
#pragma unroll
    for (int u = 0; u < 8; ++u) t = t + v[u];
  }
  for (; s < a.splits; ++s) t = t + __builtin_nontemporal_load(sl + (size_t)s * sstride);
  const float unscale =
      reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.dy) + a.dy_bytes - PL_HEADER)[2] *
      reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.src) + a.src_bytes - PL_HEADER)[2];
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (co + e < a.Cout) a.dw[(long long)(co + e) * a.ldw + cj] += t[e] * unscale;
}

static void* g_wgrad_ws = nullptr;
static size_t g_wgrad_ws_bytes = 0;
void* wgrad_workspace(size_t* bytes) {
  if (bytes != nullptr) *bytes = g_wgrad_ws_bytes;
  return g_wgrad_ws;
}

template <int BM, int BN, int WGM, int WGN, int NB = 3, int KO = 0>
static int launch_wp(WgradArgs& a, hipStream_t st) {
  a.tiles_co = (a.Cout + BM - 1) / BM;
  const int cols = a.ntaps * a.Cs;
  a.tiles_j = (cols + BN - 1) / BN;
  const long long tiles = (long long)a.tiles_co * a.tiles_j;
  // split the pixel contraction so that the grid is about `target` workgroups (2 workgroups x 256 CUs = one
  // round; every workgroup ends with BM x BN fp32 atomics, and those run at ~1.3 TB/s chip-wide)
  // Measured: 3x3 layers are flat from 1024 to 2048 workgroups, 1x1 layers (few pixels per split, the atomics
  // of 1024 workgroups cost as much as their MFMAs) run 25 % faster at 256.
  static const long long target_env = [] { const char* e = getenv("YOLO_WGRAD_TARGET"); return e ? atoll(e) : 0LL; }();
  // 3x3 layers: whole rounds of the workgroups the chip holds at once (occupancy x CUs): 1026 workgroups on 768
  // slots are two rounds, the second a third full (52x52x128->256: 167 us; 756 workgroups: 154 us)
  static int resident = 0;
  if (resident == 0) {
    int per_cu = 0, dev = 0, cus = 0;
    constexpr size_t lds_q = NB * (BM / 32 + BN / 32) * PL_PLANES * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_planes_kernel<BM, BN, WGM, WGN, NB, KO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&wgrad_planes_kernel<BM, BN, WGM, WGN, NB, KO>),
                                                     64 * WGM * WGN, lds_q) == hipSuccess &&
        hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
      resident = per_cu * cus;
    if (resident <= 0) resident = 512;
  }
  // (round 3, slabs instead of atomics, same finding: 512 / 768 workgroups for the 1x1 layers are 12-25 % slower than 256,
  // and a deeper DMA ring (NB = 4..6) changes nothing: these launches are not bound by bytes in flight)
  // (YOLO_WGRAD_TARGET_1X1: tuning knob for the 1x1 layers)
  static const long long target_1x1_env = [] { const char* e = getenv("YOLO_WGRAD_TARGET_1X1"); return e ? atoll(e) : 0LL; }();
  long long target = a.ntaps == 1 ? (target_1x1_env > 0 ? target_1x1_env : 256)
                                  : (target_env > 0 ? target_env : 1024);   // (YOLO_WGRAD_TARGET: 3x3 layers only)
  if (target_env <= 0 && a.ntaps > 1) {
    const long long s1 = resident / tiles, s2 = 2LL * resident / tiles;
    if (s1 >= 1 && tiles * s1 * 10 >= 7LL * resident) target = tiles * s1;         // one round, at least 70 % full
    else if (s2 >= 1 && tiles * s2 * 10 >= 17LL * resident) target = tiles * s2;   // two rounds
  }
  long long splits = target / tiles > 0 ? (target_env > 0 || a.ntaps == 1 ? (target + tiles - 1) / tiles : target / tiles) : 1;
  const long long max_splits = (a.M + 255) / 256;  // at least 16 stages per workgroup
  if (splits > max_splits) splits = max_splits;
  if (splits > 65535) splits = 65535;
  if (splits < 1) splits = 1;
  // reproducible form: the slabs of all workgroups must fit the registered workspace (fewer, longer splits otherwise)
  a.slabs = nullptr;
  static const bool det_env = [] { const char* e = getenv("YOLO_WGRAD_DETERMINISTIC"); return !(e && atoi(e) == 0); }();
  if (det_env && g_wgrad_ws != nullptr && g_wgrad_ws_bytes > WGRAD_WS_COLSUM_BYTES) {
    const long long cap = (long long)((g_wgrad_ws_bytes - WGRAD_WS_COLSUM_BYTES) / ((size_t)BM * BN * 4));
    if (cap >= tiles) {
      if (tiles * splits > cap) splits = cap / tiles;
      a.slabs = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(g_wgrad_ws) + WGRAD_WS_COLSUM_BYTES);
    }
  }
  long long chunk = (a.M + splits - 1) / splits;
  chunk = (chunk + 15) / 16 * 16;
  splits = (a.M + chunk - 1) / chunk;
  a.chunk = chunk;
  a.splits = (int)splits;
  if (tiles * splits > 0x7fffffffLL || splits > 65535) {
    set_error("wgrad(planes): bad grid %lld x %lld", tiles, splits);
    return YOLO_ERR_INVALID_ARG;
  }
  a.nblocks = (int)(tiles * splits);
  constexpr size_t lds = NB * (BM / 32 + BN / 32) * PL_PLANES * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_planes_kernel<BM, BN, WGM, WGN, NB, KO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((wgrad_planes_kernel<BM, BN, WGM, WGN, NB, KO>), dim3((unsigned)(tiles * splits)), dim3(64 * WGM * WGN), lds,
                     st, a);
  if (int rc = check_launch("wgrad_planes_kernel")) return rc;
  if (a.slabs != nullptr) {
    const long long pieces = tiles * (BM * BN / 4);
    hipLaunchKernelGGL((wgrad_reduce_kernel<BM, BN, WGM, WGN>), dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, a);
    return check_launch("wgrad_reduce_kernel");
  }
  return YOLO_OK;
}

bool wgrad_planes_supported(const WgradArgs& a) {
  return (a.Cout % 16) == 0 && (a.Cs % 16) == 0 && a.Cout >= 32 && a.ntaps * a.Cs >= 64;   // (Cout = 32: half of a 64-row tile)
}

int launch_wgrad_planes(WgradArgs& a, hipStream_t st) {
  const long long rowsX = (long long)a.N * a.Hs * a.Ws;
  const long long bytesX = planes_bytes(rowsX, a.Cs), bytesDy = planes_bytes(a.M, a.Cout);
  if (bytesX >= (1LL << 32) || bytesDy >= (1LL << 32)) {
    set_error("wgrad(planes): operand planes exceed 4 GiB (%lld, %lld bytes)", bytesX, bytesDy);
    return YOLO_ERR_INVALID_ARG;
  }
  a.src_bytes = (unsigned)bytesX;
  a.dy_bytes = (unsigned)bytesDy;
  a.zero_blk_src = (int)((rowsX + 15) / 16);
  a.zero_blk_dy = (int)((a.M + 15) / 16);
  const int cols = a.ntaps * a.Cs;
  init_options();
  if (g_opt[OPT_WGRAD_WIN] != 0 && wgrad_win_supported(a)) {
    const int rc = launch_wgrad_win(a, st);   // (1: its LDS layout assumption does not hold in this build -- per-tap kernel below)
    if (rc <= 0) return rc;
  }
  if (a.Cout <= 64 && cols <= 64) return launch_wp<64, 64, 2, 2>(a, st);
  if (a.Cout <= 64) return launch_wp<64, 128, 2, 4>(a, st);
  if (cols <= 64) return launch_wp<128, 64, 4, 2>(a, st);
  // 4 waves x (64 x 64): 12 MFMAs per wave between barriers instead of 6. Alone 160-178 us against 200-207 us on the
  // 3x3 layers; in the training step (beside the data-gradient stream) 33.25 against 33.58 ms. YOLO_WGRAD_WAVES=8: old form
  static const int waves = [] { const char* e = getenv("YOLO_WGRAD_WAVES"); return e ? atoi(e) : 4; }();
#ifdef YOLO_PLANES_KNOCKOUTS   // diagnostic build (make KNOCKOUTS=1)
  static const int ko = [] { const char* e = getenv("YOLO_WGRAD_KO"); return e ? atoi(e) : 0; }();
  switch (ko) {
    case 1: return launch_wp<128, 128, 2, 2, 3, 1>(a, st);
    case 2: return launch_wp<128, 128, 2, 2, 3, 2>(a, st);
    case 3: return launch_wp<128, 128, 2, 2, 3, 3>(a, st);
    case 4: return launch_wp<128, 128, 2, 2, 3, 4>(a, st);
    case 7: return launch_wp<128, 128, 2, 2, 3, 7>(a, st);
    default: break;
  }
#endif
  // 128 x 256 tile (4 waves x (64 x 128), 243 registers, two workgroups per CU): 25 % fewer LDS bytes per MFMA -- the
  // knock-out build (make KNOCKOUTS=1, YOLO_WGRAD_KO) shows this kernel's matrix work (89 us alone on 52x52x128->256) and
  // its LDS traffic (DMA + fragment reads: 97 us alone) hardly overlapping (159 us together): 128 B/clk of LDS port are as
  // busy as the matrix cores. Measured alone: 13x13x512->1024 188 -> 160 us, 26x26 +-0, 52x52 / 104x104 5-10 % slower (two
  // workgroups per CU instead of three), 1x1 layers 30 % slower: used for the 3x3 layers with few pixels (YOLO_WGRAD_WIDE:
  // 0 never, 1 wherever the shape allows, 2 = that policy)
  static const int wide = [] { const char* e = getenv("YOLO_WGRAD_WIDE"); return e ? atoi(e) : 2; }();
  if (cols >= 256 && (wide == 1 || (wide == 2 && a.ntaps > 1 && a.M <= 8192))) return launch_wp<128, 256, 2, 2>(a, st);
  if (waves == 4) return launch_wp<128, 128, 2, 2>(a, st);
  return launch_wp<128, 128, 4, 2>(a, st);
}

}  // namespace yolo

extern "C" size_t yolo_wgrad_workspace_bytes(void) {
  // 1 MiB of bias-gradient partials + slabs for two rounds of 768 resident workgroups of 128 x 128 fp32
  return yolo::WGRAD_WS_COLSUM_BYTES + (size_t)1536 * 128 * 128 * 4;
}

extern "C" int yolo_set_wgrad_workspace(void* p, size_t bytes) {
  yolo::g_wgrad_ws = (p != nullptr && bytes > yolo::WGRAD_WS_COLSUM_BYTES) ? p : nullptr;
  yolo::g_wgrad_ws_bytes = yolo::g_wgrad_ws != nullptr ? bytes : 0;
  return YOLO_OK;
}

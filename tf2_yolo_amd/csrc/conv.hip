// Implicit-GEMM convolution for gfx950 on the exact-fp32 matrix instruction
// v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD, 157 TFLOP/s chip peak).
//
// Replaces tf.keras Conv2D forward and the TF autodiff conv gradients used by
//   yolov3/models/backbone.py:27-36 (DarknetConv2D), yolov4/models/backbone.py:63-74,
//   yolov1_5/models/backbone.py:9-16, yolov2/models/backbone.py:11-18,
//   head convs yolov3/models/__init__.py:40-58.
//
// One "gather" kernel serves forward, stride-1 dgrad and stride-2 dgrad (the latter as
// s*s parity classes with sub-filters), because all three are
//     dst[n, y*osy+ooy, x*osx+oox, co] (+)= bias[co] +
//         sum_{t in taps} sum_{c} src[n, y*sy+oy_t, x*sx+ox_t, c] * W[co][woff_t + c]
// over an output grid (y, x) in [0,Hg) x [0,Wg): no im2col buffer, no padded copy, the
// zero padding is a bounds test on the source pixel.
//
// GEMM view: M = N*Hg*Wg output pixels (rows, A operand), N = Cout (columns, B operand),
// K = ntaps*Cs. A block owns a BM x BN tile; K is walked in BK = 32 slices that never
// straddle a tap (Cs % 32 == 0; otherwise the FLAT variant decodes k -> (tap, c) per
// element, used for the Cin = 3 stem and the 255-channel head gradient).
//
// Data path per K slice: global (16 B/lane, rows of 128 B) -> registers -> LDS
// [rows][32+4] (pad 4 floats: the ds_read_b128 fragment reads of 16 distinct rows hit 16
// distinct 4-bank groups, conflict-free) -> one ds_read_b128 per 32-row fragment per 8 k
// -> 4 MFMAs per fragment pair. The k order inside a slice is permuted identically for A
// and B (lane half h reads k = 8q+4h..8q+4h+3), which the contraction does not care about.
// Double-buffered LDS, one barrier per slice, next slice's global loads in flight during
// the 64 MFMAs (4096 cycles) of the current one.
#include "conv_args.hpp"
#include <cstdlib>
#include <type_traits>

namespace yolo {

template <int BM, int BN, int WGM, int WGN, bool FLAT>
__global__ __launch_bounds__(256) void gather_conv_kernel(const GatherConvArgs a) {
  constexpr int BK = 32;
  constexpr int LD = BK + 4;
  constexpr int TM = BM / WGM / 32;
  constexpr int TN = BN / WGN / 32;
  constexpr int AR = BM / 32;  // A rows staged per thread
  constexpr int BR = BN / 32;  // B rows staged per thread
  static_assert(WGM * WGN == 4, "4 waves per block");
  static_assert(TM >= 1 && TN >= 1, "wave tile");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BUF = (BM + BN) * LD;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;

  const int tile = xcd_remap(blockIdx.x, a.nblocks);
  const int tile_n = tile % a.tiles_n;
  const int tile_m = tile / a.tiles_n;
  const long long m0 = (long long)tile_m * BM;
  const int n0 = tile_n * BN;

  const int lrow = tid >> 3;
  const int kcol = (tid & 7) * 4;

  // Buffer resources: an out-of-range offset makes the hardware return 0, which IS the zero padding
  // (and the M / Cout tails) -- no select on the loaded value, so nothing forces the loads to be
  // waited for before the MFMAs of the current slice have been issued.
  constexpr unsigned OOB = 0xFFFFFFF0u;  // byte offset beyond any buffer (tensors are < 2^31 elements... < 4 GB)

  // Per-thread descriptors of the A rows it stages (fixed for the whole K loop).
  int rowel[AR];  // element offset of (n, y*sy, x*sx, 0) in src
  int ys0[AR], xs0[AR];
  const int HgWg = a.Hg * a.Wg;
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const long long m = m0 + lrow + 32 * i;
    if (m < a.M) {
      const int n = (int)(m / HgWg);
      const int rem = (int)(m - (long long)n * HgWg);
      const int y = rem / a.Wg;
      const int x = rem - y * a.Wg;
      ys0[i] = y * a.sy;
      xs0[i] = x * a.sx;
      rowel[i] = ((n * a.Hs + ys0[i]) * a.Ws + xs0[i]) * a.Cs;
    } else {
      rowel[i] = 0;
      ys0[i] = -(1 << 28);  // fails every bounds test
      xs0[i] = 0;
    }
  }
  int browel[BR];  // element offset of weight row co (or -1)
#pragma unroll
  for (int j = 0; j < BR; ++j) {
    const int co = n0 + lrow + 32 * j;
    browel[j] = (co < a.Cout) ? co * a.ldw : -1;
  }

  const int Ktot = a.ntaps * a.Cs;
  const int cpt = FLAT ? 1 : a.Cs / BK;                  // K slices per tap
  const int nk = FLAT ? (Ktot + BK - 1) / BK : a.ntaps * cpt;

  // two staging register sets: while set S is being written to LDS (slice kt+1), the other one
  // receives the global loads of slice kt+2 -- two slices of HBM/L2 latency cover, no extra LDS
  f32x4 ra[2][AR], rb[2][BR];

  auto load_slice = [&](int kt, auto SET) {
    constexpr int S = decltype(SET)::value;
    // descriptors rebuilt from kernel arguments here so that they stay in SGPRs (a descriptor the
    // compiler cannot prove wave-uniform is wrapped in a waterfall loop per load)
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.src), 0, (unsigned)((long long)a.N * a.Hs * a.Ws * a.Cs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.wgt), 0, (unsigned)((long long)a.Cout * a.ldw * 4), 0x00020000);
    if constexpr (!FLAT) {
      const int tap = kt / cpt;
      const int c0 = (kt - tap * cpt) * BK;
      const int oy = a.taps[tap].oy, ox = a.taps[tap].ox, woff = a.taps[tap].woff;
      const int tapel = (oy * a.Ws + ox) * a.Cs + c0 + kcol;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const int ys = ys0[i] + oy, xs = xs0[i] + ox;
        const bool ok = ((unsigned)ys < (unsigned)a.Hs) && ((unsigned)xs < (unsigned)a.Ws);
        const unsigned off = ok ? (unsigned)(rowel[i] + tapel) * 4u : OOB;
        ra[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcA, off, 0, 0));
      }
      const int wel = woff + c0 + kcol;
#pragma unroll
      for (int j = 0; j < BR; ++j) {
        const unsigned off = (browel[j] >= 0) ? (unsigned)(browel[j] + wel) * 4u : OOB;
        rb[S][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcB, off, 0, 0));
      }
    } else {
      const int kbase = kt * BK + kcol;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = kbase + e;
        const bool kok = k < Ktot;
        const int tap = kok ? k / a.Cs : 0;
        const int c = k - tap * a.Cs;
        const int r = tap / a.kw;
        const int s = tap - r * a.kw;
        const int oy = r - a.pad_t, ox = s - a.pad_l;
        const int tapel = (oy * a.Ws + ox) * a.Cs + c;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
          const int ys = ys0[i] + oy, xs = xs0[i] + ox;
          const bool ok = kok && ((unsigned)ys < (unsigned)a.Hs) && ((unsigned)xs < (unsigned)a.Ws);
          const unsigned off = ok ? (unsigned)(rowel[i] + tapel) * 4u : OOB;
          ra[S][i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrcA, off, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) {
          const unsigned off = (kok && browel[j] >= 0) ? (unsigned)(browel[j] + k) * 4u : OOB;
          rb[S][j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrcB, off, 0, 0));
        }
      }
    }
  };

  auto store_a = [&](int buf, auto SET) {
    constexpr int S = decltype(SET)::value;
    float* sA = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < AR; ++i)
      *reinterpret_cast<f32x4*>(&sA[(lrow + 32 * i) * LD + kcol]) = ra[S][i];
  };
  auto store_b = [&](int buf, auto SET) {
    constexpr int S = decltype(SET)::value;
    float* sB = smem + buf * BUF + BM * LD;
#pragma unroll
    for (int j = 0; j < BR; ++j)
      *reinterpret_cast<f32x4*>(&sB[(lrow + 32 * j) * LD + kcol]) = rb[S][j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frag_row = lane & 31;
  const int frag_k = (lane >> 5) * 4;

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  // one K slice: MFMAs on LDS buffer `buf`; interleaved with them, slice kt+2 starts loading into
  // register set SL and slice kt+1 (register set SS, loaded one iteration ago) is written to the other
  // LDS buffer, which every wave finished reading before the previous barrier.
  auto slice = [&](int kt, auto SL, auto SS) {
    const int buf = kt & 1;
    if (kt + 2 < nk) load_slice(kt + 2, SL);
    const float* sA = smem + buf * BUF + (wm * TM * 32 + frag_row) * LD + frag_k;
    const float* sB = smem + buf * BUF + BM * LD + (wn * TN * 32 + frag_row) * LD + frag_k;
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(sA + i * 32 * LD + q * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(sB + j * 32 * LD + q * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
      if (kt + 1 < nk) {
        if (q == 1) store_a(buf ^ 1, SS);
        if (q == 2) store_b(buf ^ 1, SS);
      }
    }
    __syncthreads();
  };

  if constexpr (FLAT) {
    // scalar-gather variant (Cin = 3 stem, 255-wide head gradient): short K, register-hungry loads ->
    // classic one-set schedule (load slice kt+1 at the top, write it to LDS after the MFMAs)
    load_slice(0, S0{});
    store_a(0, S0{});
    store_b(0, S0{});
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nk) load_slice(kt + 1, S0{});
      const float* sA = smem + buf * BUF + (wm * TM * 32 + frag_row) * LD + frag_k;
      const float* sB = smem + buf * BUF + BM * LD + (wn * TN * 32 + frag_row) * LD + frag_k;
#pragma unroll
      for (int q = 0; q < BK / 8; ++q) {
        f32x4 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(sA + i * 32 * LD + q * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(sB + j * 32 * LD + q * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
      }
      if (kt + 1 < nk) {
        store_a(buf ^ 1, S0{});
        store_b(buf ^ 1, S0{});
      }
      __syncthreads();
    }
  } else {
    load_slice(0, S0{});
    store_a(0, S0{});
    store_b(0, S0{});
    if (nk > 1) load_slice(1, S1{});
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      slice(kt, S0{}, S1{});      // loads kt+2 -> set 0, stores set 1 (slice kt+1)
      slice(kt + 1, S1{}, S0{});  // loads kt+3 -> set 1, stores set 0 (slice kt+2)
    }
    if (kt < nk) slice(kt, S0{}, S1{});
  }

  // Epilogue. Row -> destination offset table through LDS (the K loop is done with it).
  long long* rowoff = reinterpret_cast<long long*>(smem);
  for (int r = tid; r < BM; r += 256) {
    const long long m = m0 + r;
    long long off = -1;
    if (m < a.M) {
      const int n = (int)(m / HgWg);
      const int rem = (int)(m - (long long)n * HgWg);
      const int y = rem / a.Wg;
      const int x = rem - y * a.Wg;
      off = (((long long)n * a.Hd + (y * a.osy + a.ooy)) * a.Wd + (x * a.osx + a.oox)) * a.Cd;
    }
    rowoff[r] = off;
  }
  __syncthreads();

  // BatchNorm statistics fused into the epilogue: per-column partial sums over this block's rows
  // (fp32 over <= 128 rows), combined across the waves that share the columns through LDS, then ONE
  // fp64 atomic per column per block into one of YOLO_BN_STAT_SLOTS replicas (spreads contention).
  float* sred = smem + 2 * BM;  // after the row-offset table (BM long longs)
  float csum[TN], csq[TN], cmx[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
    const bool cok = col < a.Cout;
    const float bv = (a.bias != nullptr && cok) ? a.bias[col] : 0.f;
    float s1 = 0.f, s2 = 0.f, mx = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const long long off = rowoff[row];
        if (cok && off >= 0) {
          float v = acc[i][j][r] + bv;
          if (a.accumulate) v += a.dst[off + col];
          a.dst[off + col] = v;
          s1 += v;
          s2 += v * v;
          mx = fmaxf(mx, fabsf(v));
        }
      }
    }
    csum[j] = s1;
    csq[j] = s2;
    cmx[j] = mx;
  }
  if (a.stats != nullptr || a.absmax != nullptr) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float s1 = csum[j] + __shfl_xor(csum[j], 32, 64);
      const float s2 = csq[j] + __shfl_xor(csq[j], 32, 64);
      const float mx = fmaxf(cmx[j], __shfl_xor(cmx[j], 32, 64));
      if (lane < 32) {
        const int c = (wn * TN + j) * 32 + lane;  // column within the block tile
        sred[(wm * BN + c) * 3 + 0] = s1;
        sred[(wm * BN + c) * 3 + 1] = s2;
        sred[(wm * BN + c) * 3 + 2] = mx;
      }
    }
    __syncthreads();
    for (int c = tid; c < BN; c += 256) {
      const int col = n0 + c;
      if (col < a.Cout) {
        float s1 = 0.f, s2 = 0.f, mx = 0.f;
#pragma unroll
        for (int w = 0; w < WGM; ++w) {
          s1 += sred[(w * BN + c) * 3 + 0];
          s2 += sred[(w * BN + c) * 3 + 1];
          mx = fmaxf(mx, sred[(w * BN + c) * 3 + 2]);
        }
        if (a.stats != nullptr) {
          double* slot = a.stats + (long long)(tile_m & (YOLO_BN_STAT_SLOTS - 1)) * 2 * a.Cout;
          atomicAdd(&slot[col], (double)s1);
          atomicAdd(&slot[a.Cout + col], (double)s2);
        }
        // per-channel max|y| (bit patterns of non-negative floats order like integers); most tiles skip the atomic
        if (a.absmax != nullptr && __builtin_bit_cast(unsigned, mx) > a.absmax[col])
          atomicMax(&a.absmax[col], __builtin_bit_cast(unsigned, mx));
      }
    }
  }
}

template <int BM, int BN, int WGM, int WGN, bool FLAT>
static int launch_gather(GatherConvArgs& a, hipStream_t st) {
  const long long tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.Cout + BN - 1) / BN;
  const long long nb = tiles_m * a.tiles_n;
  if (nb <= 0 || nb > 0x7fffffffLL) {
    set_error("conv: bad grid %lld", nb);
    return YOLO_ERR_INVALID_ARG;
  }
  a.nblocks = (int)nb;
  constexpr size_t lds = 2 * (BM + BN) * 36 * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_conv_kernel<BM, BN, WGM, WGN, FLAT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((gather_conv_kernel<BM, BN, WGM, WGN, FLAT>), dim3((unsigned)nb), dim3(256), lds, st, a);
  return check_launch("gather_conv_kernel");
}

// 0 = exact-fp32 MFMA kernels (this file), 1 = fp32 emulated with 6 bf16 MFMA passes (conv_split.hip)
static int g_conv_mode = [] {
  const char* e = getenv("YOLO_CONV_MODE");
  if (e && (e[0] == 'f' || e[0] == '0')) return 0;   // "fp32": exact-fp32 MFMA everywhere
  return 1;                                          // default: "split" (bf16 x 6, fp32-accurate)
}();

static int dispatch_gather(GatherConvArgs& a, bool flat, hipStream_t st) {
  if (!flat && g_conv_mode == 1 && gather_split_supported(a)) return launch_gather_split(a, st);
  // every configuration keeps LDS <= 80 KB so that two workgroups share a CU
  if (flat) {
    if (a.Cout <= 32) return launch_gather<128, 32, 4, 1, true>(a, st);
    if (a.Cout <= 64) return launch_gather<128, 64, 2, 2, true>(a, st);
    return launch_gather<128, 128, 2, 2, true>(a, st);
  }
  if (a.Cout <= 32) return launch_gather<128, 32, 4, 1, false>(a, st);
  if (a.Cout <= 64) return launch_gather<128, 64, 2, 2, false>(a, st);
  // Wide outputs: 128x128 tiles unless the grid would not even fill the 256 CUs x 2 workgroups once;
  // then 64-row tiles (twice the blocks) balance the chip: 13x13 / 26x26 layers gain 10-15 %.
  static const int force = [] { const char* e = getenv("YOLO_CONV_TILE_M"); return e ? atoi(e) : 0; }();
  const long long blocks128 = ((a.M + 127) / 128) * ((a.Cout + 127) / 128);
  bool use64 = false;
  if (force == 64) use64 = true;
  else if (force == 128) use64 = false;
  else use64 = blocks128 <= 512;  // measured: below one full round of 2 workgroups per CU, halve the tiles
  if (use64) return launch_gather<64, 128, 1, 4, false>(a, st);
  return launch_gather<128, 128, 2, 2, false>(a, st);
}

static int validate_desc(const yolo_conv_desc* d) {
  YOLO_REQUIRE(d != nullptr, "conv: null descriptor");
  YOLO_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "conv: non-positive dims");
  YOLO_REQUIRE(d->kh > 0 && d->kw > 0 && d->kh * d->kw <= MAX_TAPS, "conv: kernel %dx%d unsupported", d->kh, d->kw);
  YOLO_REQUIRE(d->sh >= 1 && d->sw >= 1 && d->sh <= 2 && d->sw <= 2, "conv: stride %dx%d unsupported", d->sh, d->sw);
  YOLO_REQUIRE(d->Ho > 0 && d->Wo > 0, "conv: bad output size");
  YOLO_REQUIRE(d->pad_t >= 0 && d->pad_l >= 0 && d->pad_t < d->kh && d->pad_l < d->kw, "conv: bad padding");
  // every output pixel's window must start inside the (virtually) padded input
  YOLO_REQUIRE((long long)(d->Ho - 1) * d->sh - d->pad_t < d->H && (long long)(d->Wo - 1) * d->sw - d->pad_l < d->W,
               "conv: output %dx%d too large for input %dx%d", d->Ho, d->Wo, d->H, d->W);
  // 32-bit byte offsets in the buffer loads: every tensor must stay below 4 GB
  YOLO_REQUIRE((long long)d->N * d->H * d->W * d->Cin < (1LL << 30) &&
               (long long)d->N * d->Ho * d->Wo * d->Cout < (1LL << 30) &&
               (long long)d->Cout * d->kh * d->kw * d->Cin < (1LL << 30), "conv: tensor too large (>= 4 GB)");
  return YOLO_OK;
}

// ---------------------------------------------------------------------------------------
// wgrad: dW[co][j] += sum_p dy[p][co] * src[p -> tap(j)][c(j)],   j = tap*Cin + c  (KRSC row of a filter)
// GEMM view: rows = co (A operand, from dy), cols = j over the FLATTENED (tap, ci) axis (B operand,
// gathered from src), contraction over pixels p, split over blocks (grid.y) and combined with fp32
// atomics (each lane-half writes one 128-B segment: the full-rate shape).
// Flattening the columns means a block with BN = 128 covers 4 taps of a Cin = 32 layer: dy is read once
// per column tile instead of once per tap, which is what the big-M / narrow-channel early layers need.
// A thread's column (hence its tap offset) is fixed for the whole pixel loop.
// Both operands arrive pixel-major, so the LDS images are [32 pixels][BM|BN] and the fragments are
// ds_read_b32 with consecutive lanes on consecutive banks; fragments of k-step s+1 are fetched before
// the MFMAs of step s are issued.
// ---------------------------------------------------------------------------------------
template <int BM, int BN, int WGM, int WGN, bool ASCALAR, bool BSCALAR>
__global__ __launch_bounds__(64 * WGM * WGN) void wgrad_kernel(const WgradArgs a) {
  constexpr int NT = 64 * WGM * WGN;
  constexpr int BK = 32;
  constexpr int TM = BM / WGM / 32;
  constexpr int TN = BN / WGN / 32;
  constexpr int AF4 = BM / 4;  // float4 per pixel row of A
  constexpr int ARP = (NT / AF4) < BK ? (NT / AF4) : BK;  // pixel rows per pass
  constexpr int AP = BK / ARP;
  constexpr int BF4 = BN / 4;
  constexpr int BRP = (NT / BF4) < BK ? (NT / BF4) : BK;
  constexpr int BP = BK / BRP;
  static_assert(TM >= 1 && TN >= 1, "wave tile");
  static_assert(AF4 <= NT && BF4 <= NT, "tile too wide for the block");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BUF = BK * (BM + BN);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;

  const int tile_co = blockIdx.x % a.tiles_co;
  const int j0 = (blockIdx.x / a.tiles_co) * BN;
  const int co0 = tile_co * BM;
  const long long p_begin = (long long)blockIdx.y * a.chunk;
  long long p_end = p_begin + a.chunk;
  if (p_end > a.M) p_end = a.M;
  if (p_begin >= p_end) return;
  const int nk = (int)((p_end - p_begin + BK - 1) / BK);

  const int a_row = tid / AF4, a_col = (tid % AF4) * 4;
  const int b_row = tid / BF4, b_col = (tid % BF4) * 4;
  const bool a_act = a_row < ARP;  // threads beyond the tile width idle in the staging passes
  const bool b_act = b_row < BRP;
  const int HgWg = a.Hg * a.Wg;
  const int Ktot = a.ntaps * a.Cs;

  constexpr unsigned OOB = 0xFFFFFFF0u;

  // this thread's B column(s): flattened j -> (tap, c); fixed for the whole pixel loop
  int boy[BSCALAR ? 4 : 1], box[BSCALAR ? 4 : 1], bc[BSCALAR ? 4 : 1];
  bool bok[BSCALAR ? 4 : 1];
#pragma unroll
  for (int e = 0; e < (BSCALAR ? 4 : 1); ++e) {
    const int j = j0 + b_col + e;
    bok[e] = j < Ktot;
    const int t = bok[e] ? j / a.Cs : 0;
    const int r = t / a.kw;
    boy[e] = r - a.pad_t;
    box[e] = (t - r * a.kw) - a.pad_l;
    bc[e] = j - t * a.Cs;
  }
  const bool aok_col = co0 + a_col < a.Cout;

  // pixel coordinates of this thread's B rows, advanced by BK per staged slice (slices are staged in
  // order), so the per-slice integer divisions disappear from the loop
  int pn[BP], py[BP], px[BP];
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    const long long p = p_begin + b_row + i * BRP;
    const long long pp = p < a.M ? p : 0;
    pn[i] = (int)(pp / HgWg);
    const int rem = (int)(pp - (long long)pn[i] * HgWg);
    py[i] = rem / a.Wg;
    px[i] = rem - py[i] * a.Wg;
  }

  f32x4 ra[2][AP], rb[2][BP];

  auto load_slice = [&](int kt, auto SET) {
    constexpr int S = decltype(SET)::value;
    const __amdgpu_buffer_rsrc_t rsrcA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (unsigned)(a.M * a.Cout * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.src), 0, (unsigned)((long long)a.N * a.Hs * a.Ws * a.Cs * 4), 0x00020000);
    const long long pbase = p_begin + (long long)kt * BK;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const long long p = pbase + a_row + i * ARP;
      const bool pok = a_act && (p < p_end);
      if constexpr (!ASCALAR) {
        const unsigned off = (pok && aok_col) ? (unsigned)(p * a.Cout + co0 + a_col) * 4u : OOB;
        ra[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcA, off, 0, 0));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = pok && (co0 + a_col + e < a.Cout);
          const unsigned off = ok ? (unsigned)(p * a.Cout + co0 + a_col + e) * 4u : OOB;
          ra[S][i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrcA, off, 0, 0));
        }
      }
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const long long p = pbase + b_row + i * BRP;
      const bool pok = b_act && (p < p_end);
      const int n = pn[i], y = py[i], x = px[i];
      const int ybase = y * a.sy, xbase = x * a.sx;
      if constexpr (!BSCALAR) {
        const int ys = ybase + boy[0], xs = xbase + box[0];
        const bool ok = pok && bok[0] && ((unsigned)ys < (unsigned)a.Hs) && ((unsigned)xs < (unsigned)a.Ws);
        const unsigned off = ok ? (unsigned)(((n * a.Hs + ys) * a.Ws + xs) * a.Cs + bc[0]) * 4u : OOB;
        rb[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcB, off, 0, 0));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ys = ybase + boy[e], xs = xbase + box[e];
          const bool ok = pok && bok[e] && ((unsigned)ys < (unsigned)a.Hs) && ((unsigned)xs < (unsigned)a.Ws);
          const unsigned off = ok ? (unsigned)(((n * a.Hs + ys) * a.Ws + xs) * a.Cs + bc[e]) * 4u : OOB;
          rb[S][i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrcB, off, 0, 0));
        }
      }
      // advance this row's pixel by BK for the next staged slice
      int nx = x + BK, ny = y, nn = n;
      while (nx >= a.Wg) {
        nx -= a.Wg;
        ++ny;
      }
      while (ny >= a.Hg) {
        ny -= a.Hg;
        ++nn;
      }
      px[i] = nx;
      py[i] = ny;
      pn[i] = nn;
    }
  };

  auto store_a = [&](int buf, auto SET) {
    constexpr int S = decltype(SET)::value;
    float* sA = smem + buf * BUF;
    if (a_act) {
#pragma unroll
      for (int i = 0; i < AP; ++i) *reinterpret_cast<f32x4*>(&sA[(a_row + i * ARP) * BM + a_col]) = ra[S][i];
    }
  };
  auto store_b = [&](int buf, auto SET) {
    constexpr int S = decltype(SET)::value;
    float* sB = smem + buf * BUF + BK * BM;
    if (b_act) {
#pragma unroll
      for (int i = 0; i < BP; ++i) *reinterpret_cast<f32x4*>(&sB[(b_row + i * BRP) * BN + b_col]) = rb[S][i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  // same two-set schedule as gather_conv_kernel: slice kt+2 is loading into set SL while slice kt+1
  // (set SS) is written to the other LDS buffer in the middle of slice kt's MFMAs
  // the scalar-gather variants (Cin = 3 stem, 255-wide heads) are register-hungry and short: they keep
  // the classic one-set schedule (load slice kt+1 at the top, write it after the MFMAs)
  constexpr bool DEEP = !(ASCALAR || BSCALAR);
  auto slice = [&](int kt, auto SL, auto SS) {
    const int buf = kt & 1;
    if constexpr (DEEP) {
      if (kt + 2 < nk) load_slice(kt + 2, SL);
    } else {
      if (kt + 1 < nk) load_slice(kt + 1, SS);
    }
    const float* sA = smem + buf * BUF + fh * BM + wm * TM * 32 + fr;
    const float* sB = smem + buf * BUF + BK * BM + fh * BN + wn * TN * 32 + fr;
    float af[2][TM], bf[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = sA[i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[0][j] = sB[j * 32];
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      if (s + 1 < BK / 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[nxt][i] = sA[2 * (s + 1) * BM + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[nxt][j] = sB[2 * (s + 1) * BN + j * 32];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
      if (kt + 1 < nk) {
        if (s == (DEEP ? 5 : BK / 2 - 1)) store_a(buf ^ 1, SS);
        if (s == (DEEP ? 10 : BK / 2 - 1)) store_b(buf ^ 1, SS);
      }
    }
    __syncthreads();
  };

  load_slice(0, S0{});
  store_a(0, S0{});
  store_b(0, S0{});
  if constexpr (DEEP) {
    if (nk > 1) load_slice(1, S1{});
  }
  __syncthreads();
  if constexpr (DEEP) {
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      slice(kt, S0{}, S1{});
      slice(kt + 1, S1{}, S0{});
    }
    if (kt < nk) slice(kt, S0{}, S1{});
  } else {
    for (int kt = 0; kt < nk; ++kt) slice(kt, S0{}, S0{});
  }

  if (a.slabs != nullptr) {
    // reproducible form (yolo_set_wgrad_workspace): the tile's partial to slab (split, tile) in [BM][BN] order, every
    // element (the slab is BM x BN whatever the tensor's edge); wgrad_exact_reduce_kernel adds the splits in order
    float* slab = a.slabs + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (BM * BN);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          slab[((wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * BN + (wn * TN + j) * 32 + (lane & 31)] =
              acc[i][j][r];
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
        if (cok && co < a.Cout) atomicAdd(&a.dw[(long long)co * a.ldw + cj], acc[i][j][r]);
      }
    }
  }
}

// dw += the slabs of wgrad_kernel's reproducible form, splits added in order: one thread per filter element
__global__ __launch_bounds__(256) void wgrad_exact_reduce_kernel(const WgradArgs a, const int BM, const int BN) {
  const int cols = a.ntaps * a.Cs;
  const long long n = (long long)a.Cout * cols;
  const long long tiles = (long long)a.tiles_co * a.tiles_j;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int co = (int)(e / cols), cj = (int)(e - (long long)co * cols);
    const long long tile = (long long)(cj / BN) * a.tiles_co + co / BM;   // (= blockIdx.x of wgrad_kernel: column tile * tiles_co + row tile)
    const float* p = a.slabs + (size_t)tile * (BM * BN) + (size_t)(co % BM) * BN + (cj % BN);
    float s = 0.f;
    for (int sp = 0; sp < a.splits; ++sp) s += p[(size_t)sp * tiles * (BM * BN)];
    a.dw[(long long)co * a.ldw + cj] += s;
  }
}

template <int BM, int BN, int WGM, int WGN, bool ASCALAR, bool BSCALAR>
static int launch_wgrad(WgradArgs& a, hipStream_t st) {
  a.tiles_co = (a.Cout + BM - 1) / BM;
  const int cols = a.ntaps * a.Cs;
  a.tiles_j = (cols + BN - 1) / BN;
  const long long tiles = (long long)a.tiles_co * a.tiles_j;
  // split the pixel contraction so that the grid has ~4 workgroups of 256 threads per CU
  constexpr int NT = 64 * WGM * WGN;
  long long splits = (1024LL * (256 / NT) + tiles - 1) / tiles;
  const long long max_splits = (a.M + 255) / 256;  // at least 8 K-slices per block
  if (splits > max_splits) splits = max_splits;
  if (splits > 65535) splits = 65535;
  if (splits < 1) splits = 1;
  long long chunk = (a.M + splits - 1) / splits;
  chunk = (chunk + 31) / 32 * 32;
  splits = (a.M + chunk - 1) / chunk;
  // reproducible form: the slabs of all workgroups must fit the registered workspace (fewer, longer splits otherwise);
  // without a workspace (or YOLO_WGRAD_DETERMINISTIC=0) the splits meet in fp32 atomics, as until round 6 -- the 7x7 RGB
  // stem of YOLOv1.5 and every layer whose channel counts keep it off the planes kernels took this path, and its
  // gradients differed in the last bit from run to run (scripts/step_repro.py c1)
  a.slabs = nullptr;
  static const bool det_env = [] { const char* e = getenv("YOLO_WGRAD_DETERMINISTIC"); return !(e && atoi(e) == 0); }();
  size_t ws_bytes = 0;
  unsigned char* ws = reinterpret_cast<unsigned char*>(wgrad_workspace(&ws_bytes));
  if (det_env && ws != nullptr && ws_bytes > WGRAD_WS_COLSUM_BYTES) {
    const long long cap = (long long)((ws_bytes - WGRAD_WS_COLSUM_BYTES) / ((size_t)BM * BN * 4));
    if (cap >= tiles) {
      if (tiles * splits > cap) {
        splits = cap / tiles;
        chunk = (a.M + splits - 1) / splits;
        chunk = (chunk + 31) / 32 * 32;
        splits = (a.M + chunk - 1) / chunk;
      }
      a.slabs = reinterpret_cast<float*>(ws + WGRAD_WS_COLSUM_BYTES);
    }
  }
  a.chunk = chunk;
  a.splits = (int)splits;
  if (tiles > 0x7fffffffLL || splits > 65535) {
    set_error("wgrad: bad grid %lld x %lld", tiles, splits);
    return YOLO_ERR_INVALID_ARG;
  }
  constexpr size_t lds = 2 * 32 * (BM + BN) * sizeof(float);
  hipLaunchKernelGGL((wgrad_kernel<BM, BN, WGM, WGN, ASCALAR, BSCALAR>), dim3((unsigned)tiles, (unsigned)splits),
                     dim3(NT), lds, st, a);
  if (int rc = check_launch("wgrad_kernel")) return rc;
  if (a.slabs != nullptr) {
    const long long n = (long long)a.Cout * cols;
    hipLaunchKernelGGL(wgrad_exact_reduce_kernel, dim3((unsigned)stream_grid(n, 256)), dim3(256), 0, st, a, BM, BN);
    return check_launch("wgrad_exact_reduce_kernel");
  }
  return YOLO_OK;
}

template <bool ASCALAR, bool BSCALAR>
static int dispatch_wgrad(WgradArgs& a, hipStream_t st) {
  const int cols = a.ntaps * a.Cs;
  if (a.Cout <= 32 && cols <= 32) return launch_wgrad<32, 32, 1, 1, ASCALAR, BSCALAR>(a, st);   // Cin = 3 stem
  if (a.Cout <= 32) return launch_wgrad<32, 128, 1, 4, ASCALAR, BSCALAR>(a, st);
  if (cols <= 32) return launch_wgrad<128, 32, 4, 1, ASCALAR, BSCALAR>(a, st);
  if (a.Cout <= 64 && cols <= 64) return launch_wgrad<64, 64, 2, 2, ASCALAR, BSCALAR>(a, st);
  if (a.Cout <= 64) return launch_wgrad<64, 128, 2, 2, ASCALAR, BSCALAR>(a, st);
  if (cols <= 64) return launch_wgrad<128, 64, 2, 2, ASCALAR, BSCALAR>(a, st);
  return launch_wgrad<128, 128, 2, 2, ASCALAR, BSCALAR>(a, st);
}

// wT[ci][t][co] = w[co][t][ci]
__global__ void filter_transpose_kernel(const float* __restrict__ w, float* __restrict__ wT, int Cout, int taps,
                                        int Cin) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
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

// column sums of dy[P][C] into out[C]; used for conv bias gradients. part == nullptr: fp32 atomics straight into out.
// Otherwise (reproducible form, yolo_set_wgrad_workspace): every (block, pixel-row lane) stores its partial to
// part[(blockIdx.x * blockDim.y + threadIdx.y) * C + c] and colsum_finish_kernel adds them in that order.
__global__ void colsum_kernel(const float* __restrict__ x, long long P, int C, float* __restrict__ out,
                              float* __restrict__ part) {
  // block: 256 threads = (256/cw) pixel rows x cw channel lanes, cw = min(C,256) rounded to pow2<=256
  const int c = blockIdx.y * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (long long p = (long long)blockIdx.x * blockDim.y + threadIdx.y; p < P; p += (long long)gridDim.x * blockDim.y)
    s += x[p * C + c];
  if (part != nullptr) part[((long long)blockIdx.x * blockDim.y + threadIdx.y) * C + c] = s;
  else atomicAdd(&out[c], s);
}
// 64 channels x 16 part lanes per block: lane j adds parts j, j + 16, ... (independent chains, four loads in flight each),
// the 16 lane sums are added in lane order: a fixed order, whatever the number of parts
__global__ void colsum_finish_kernel(const float* __restrict__ part, int nparts, int C, float* __restrict__ out) {
  __shared__ float lanes[16][64];
  const int c = blockIdx.x * 64 + threadIdx.x, j = threadIdx.y;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    int i = j;
    for (; i + 48 < nparts; i += 64) {
      s0 += part[(long long)i * C + c];
      s1 += part[(long long)(i + 16) * C + c];
      s2 += part[(long long)(i + 32) * C + c];
      s3 += part[(long long)(i + 48) * C + c];
    }
    for (; i < nparts; i += 16) s0 += part[(long long)i * C + c];
  }
  lanes[j][threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (j == 0 && c < C) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += lanes[k][threadIdx.x];
    out[c] += s;
  }
}

// out[0] = max of n non-negative floats (the words a one-pass inference unit left: GatherConvArgs::pl_out_words)
__global__ __launch_bounds__(256) void fold_bound_kernel(const unsigned* __restrict__ words, int n, float* __restrict__ out) {
  __shared__ float s_v[4];
  float v = 0.f;
  for (int w = threadIdx.x; w < n; w += 256) v = fmaxf(v, __builtin_bit_cast(float, words[w]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0) s_v[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = fmaxf(fmaxf(s_v[0], s_v[1]), fmaxf(s_v[2], s_v[3]));
}

// K = max_c |scale_c| * sum_j |w[c][j]|, D = max_c |scale_c * bias_c + shift_c| of one conv-BN unit (inference): then
// |act(scale * (conv(x) + bias) + shift)| <= K * max|x| + D for LeakyReLU, Mish and the identity (|act(z)| <= |z|).
// One workgroup per output channel, the row sum in fp64, out2 = {bits(K), bits(D)} by atomicMax (zeroed by the caller).
__global__ __launch_bounds__(256) void conv_pred_bound_kernel(const float* __restrict__ w, int kdim,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              const float* __restrict__ bias, unsigned* __restrict__ out2) {
  __shared__ double part[256];
  const int c = blockIdx.x;
  double s = 0.0;
  for (int j = threadIdx.x; j < kdim; j += 256) s += (double)fabsf(w[(long long)c * kdim + j]);
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float sc = scale != nullptr ? scale[c] : 1.f, sh = shift != nullptr ? shift[c] : 0.f;
    const float b = bias != nullptr ? bias[c] : 0.f;
    const float k = (float)(fabs((double)sc) * part[0] * (1.0 + 1e-6));          // (rounded up: the bound must hold)
    const float d = (float)(fabs((double)sc * (double)b + (double)sh) * (1.0 + 1e-6));
    atomicMax(&out2[0], __builtin_bit_cast(unsigned, k));
    atomicMax(&out2[1], __builtin_bit_cast(unsigned, d));
  }
}

}  // namespace yolo

using namespace yolo;

static void fill_fwd_args(const yolo_conv_desc* d, GatherConvArgs& a) {
  a.N = d->N; a.Hs = d->H; a.Ws = d->W; a.Cs = d->Cin;
  a.Hg = d->Ho; a.Wg = d->Wo;
  a.sy = d->sh; a.sx = d->sw;
  a.Hd = d->Ho; a.Wd = d->Wo; a.Cd = d->Cout;
  a.osy = 1; a.osx = 1; a.ooy = 0; a.oox = 0;
  a.Cout = d->Cout;
  a.ldw = d->kh * d->kw * d->Cin;
  a.ntaps = d->kh * d->kw;
  a.accumulate = 0;
  a.kw = d->kw; a.pad_t = d->pad_t; a.pad_l = d->pad_l;
  a.M = (long long)d->N * d->Ho * d->Wo;
  for (int r = 0; r < d->kh; ++r)
    for (int s = 0; s < d->kw; ++s) a.taps[r * d->kw + s] = Tap{r - d->pad_t, s - d->pad_l, (r * d->kw + s) * d->Cin};
}

extern "C" int yolo_conv2d_fwd_absmax(const yolo_conv_desc* d, const float* x, const float* w, const float* bias,
                                      float* y, double* stats, unsigned* absmax, void* stream) {
  if (int rc = validate_desc(d)) return rc;
  YOLO_REQUIRE(x && w && y, "conv_fwd: null pointer");
  if (g_conv_mode == 1 && stem_fwd_supported(d)) return launch_stem_fwd(d, x, w, bias, y, stats, absmax, as_stream(stream));
  GatherConvArgs a{};
  a.src = x;
  a.wgt = w;
  a.bias = bias;
  a.dst = y;
  fill_fwd_args(d, a);
  a.stats = stats;
  a.absmax = absmax;
  const bool flat = (d->Cin % 32) != 0;
  return dispatch_gather(a, flat, as_stream(stream));
}

// Backward of the stem unit (Conv2D(32, 3x3, same) on the 3-channel image + BatchNormalization + activation) behind
// yolo_bn_act_bwd_reduce: the BN / activation backward apply and the filter gradient in one pass over (y, dout); the
// 32-channel gradient tensor is never written. dw (+=), dgamma / dbeta (+=) as the unfused entry points.
extern "C" size_t yolo_stem_bwd_scratch_bytes(void) { return stem_bwd_scratch_bytes(); }
extern "C" int yolo_stem_bn_bwd_wgrad(const yolo_conv_desc* d, const float* y, const float* dout, const float* image,
                                      const float* scale, const float* shift, const float* save_mean,
                                      const float* save_invstd, int act, const double* red, float* dgamma, float* dbeta,
                                      float* dw, void* scratch, size_t scratch_bytes, void* stream) {
  if (int rc = validate_desc(d)) return rc;
  YOLO_REQUIRE(y && dout && image && scale && shift && save_mean && save_invstd && red && dw, "stem_bn_bwd_wgrad: null pointer");
  YOLO_REQUIRE(d->Cin == 3 && d->Cout == 32 && d->kh == 3 && d->kw == 3 && d->sh == 1 && d->sw == 1 && d->Ho == d->H &&
                   d->Wo == d->W && d->W >= 16,
               "stem_bn_bwd_wgrad: only Conv2D(32, 3x3, stride 1, same) on 3 channels, rows of 16 pixels or more");
  YOLO_REQUIRE(act >= 0 && act <= 2, "stem_bn_bwd_wgrad: bad activation %d", act);
  return launch_stem_bn_bwd_wgrad(d, y, dout, image, scale, shift, save_mean, save_invstd, act,
                                  red + (long long)YOLO_BN_RED_SLOTS * 2 * d->Cout, dgamma, dbeta, dw,
                                  reinterpret_cast<float*>(scratch), scratch_bytes, as_stream(stream));
}

extern "C" int yolo_conv2d_fwd(const yolo_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                               double* stats, void* stream) {
  return yolo_conv2d_fwd_absmax(d, x, w, bias, y, stats, nullptr, stream);
}

extern "C" int yolo_conv2d_fwd_planes(const yolo_conv_desc* d, const void* x_planes, const void* w_planes,
                                      const float* bias, float* y, double* stats, unsigned* absmax, void* stream) {
  if (int rc = validate_desc(d)) return rc;
  YOLO_REQUIRE(x_planes && w_planes && y, "conv_fwd_planes: null pointer");
  GatherConvArgs a{};
  a.src = reinterpret_cast<const float*>(x_planes);
  a.wgt = reinterpret_cast<const float*>(w_planes);
  a.bias = bias;
  a.dst = y;
  fill_fwd_args(d, a);
  a.stats = stats;
  a.absmax = absmax;
  YOLO_REQUIRE(gather_planes_supported(a), "conv_fwd_planes: needs Cin %% 16 == 0 and Cout >= 32");
  return launch_gather_planes(a, as_stream(stream));
}

// inference: y = act(scale[c] * (conv(x, w) + bias) + shift[c]) (+ residual); absmax = per-channel max of |y| before the
// residual is added (the caller adds the residual tensor's bound)
extern "C" int yolo_conv2d_fwd_planes_epi(const yolo_conv_desc* d, const void* x_planes, const void* w_planes,
                                          const float* bias, int epilogue, const float* scale, const float* shift,
                                          const float* residual, float* y, unsigned* absmax, void* stream) {
  if (int rc = validate_desc(d)) return rc;
  YOLO_REQUIRE(x_planes && w_planes && y, "conv_fwd_planes_epi: null pointer");
  YOLO_REQUIRE(epilogue == YOLO_EPI_NONE || epilogue == YOLO_EPI_AFFINE_LEAKY || epilogue == YOLO_EPI_AFFINE_MISH ||
               epilogue == YOLO_EPI_AFFINE, "conv_fwd_planes_epi: bad epilogue %d", epilogue);
  YOLO_REQUIRE(epilogue == YOLO_EPI_NONE || (scale && shift), "conv_fwd_planes_epi: affine epilogue without scale / shift");
  GatherConvArgs a{};
  a.src = reinterpret_cast<const float*>(x_planes);
  a.wgt = reinterpret_cast<const float*>(w_planes);
  a.bias = bias;
  a.dst = y;
  fill_fwd_args(d, a);
  a.stats = nullptr;
  a.absmax = absmax;
  if (epilogue != YOLO_EPI_NONE) {
    a.epi_scale = scale;
    a.epi_shift = shift;
    a.epi_act = epilogue == YOLO_EPI_AFFINE_LEAKY ? YOLO_ACT_LEAKY : epilogue == YOLO_EPI_AFFINE_MISH ? YOLO_ACT_MISH : YOLO_ACT_LINEAR;
  }
  a.epi_res = residual;
  YOLO_REQUIRE(gather_planes_supported(a), "conv_fwd_planes_epi: needs Cin %% 16 == 0 and Cout >= 32");
  return launch_gather_planes(a, as_stream(stream));
}

extern "C" int yolo_conv_pred_bound(const float* w, int Cout, int kdim, const float* scale, const float* shift,
                                    const float* bias, float* out2, void* stream) {
  YOLO_REQUIRE(w && out2 && Cout > 0 && kdim > 0, "conv_pred_bound: bad args");
  hipLaunchKernelGGL(conv_pred_bound_kernel, dim3((unsigned)Cout), dim3(256), 0, as_stream(stream), w, kdim, scale, shift,
                     bias, reinterpret_cast<unsigned*>(out2));
  return check_launch("conv_pred_bound_kernel");
}

extern "C" int yolo_conv2d_fwd_infer_unit(const yolo_conv_desc* d, const void* x_planes, const void* w_planes,
                                          const float* bias, int epilogue, const float* scale, const float* shift,
                                          const float* residual, float* y, unsigned* absmax, const float* pred,
                                          const void* in_bound, int in_n, const void* residual_bound, int residual_n,
                                          void* out_planes, unsigned* out_words, float* out_bound, int* out_n_host,
                                          void* stream) {
  if (int rc = validate_desc(d)) return rc;
  YOLO_REQUIRE(x_planes && w_planes && y && absmax && pred && in_bound && out_planes && out_words && out_bound && out_n_host,
               "conv_fwd_infer_unit: null pointer");
  YOLO_REQUIRE(epilogue == YOLO_EPI_AFFINE_LEAKY || epilogue == YOLO_EPI_AFFINE_MISH || epilogue == YOLO_EPI_AFFINE,
               "conv_fwd_infer_unit: bad epilogue %d", epilogue);
  YOLO_REQUIRE(scale && shift, "conv_fwd_infer_unit: affine epilogue without scale / shift");
  YOLO_REQUIRE(d->Cout % 16 == 0, "conv_fwd_infer_unit: planes output needs Cout %% 16 == 0 (Cout=%d)", d->Cout);
  YOLO_REQUIRE(residual == nullptr || residual_bound != nullptr, "conv_fwd_infer_unit: a residual needs its bound");
  YOLO_REQUIRE(in_n >= 1 && in_n <= YOLO_INFER_BOUND_WORDS &&
               (residual == nullptr || (residual_n >= 1 && residual_n <= YOLO_INFER_BOUND_WORDS)),
               "conv_fwd_infer_unit: a bound is 1..%d words", YOLO_INFER_BOUND_WORDS);
  GatherConvArgs a{};
  a.src = reinterpret_cast<const float*>(x_planes);
  a.wgt = reinterpret_cast<const float*>(w_planes);
  a.bias = bias;
  a.dst = y;
  fill_fwd_args(d, a);
  a.stats = nullptr;
  a.absmax = absmax;
  a.epi_scale = scale;
  a.epi_shift = shift;
  a.epi_act = epilogue == YOLO_EPI_AFFINE_LEAKY ? YOLO_ACT_LEAKY : epilogue == YOLO_EPI_AFFINE_MISH ? YOLO_ACT_MISH : YOLO_ACT_LINEAR;
  a.epi_res = residual;
  a.out_planes = reinterpret_cast<unsigned char*>(out_planes);
  a.pl_pred = pred;
  a.pl_in_bound = reinterpret_cast<const unsigned*>(in_bound);
  a.pl_in_n = in_n;
  a.pl_res_bound = residual != nullptr ? reinterpret_cast<const unsigned*>(residual_bound) : nullptr;
  a.pl_res_n = residual != nullptr ? residual_n : 0;
  a.pl_out_words = out_words;
  YOLO_REQUIRE(gather_planes_supported(a), "conv_fwd_infer_unit: needs Cin %% 16 == 0 and Cout >= 32");
  // few output pixels (bs-1 predict): the whole unit in ONE launch, K split across the waves of a workgroup (conv_small.hip)
  if (conv_small_supported(a)) return launch_conv_small(a, as_stream(stream), out_n_host);
  if (int rc = launch_gather_planes(a, as_stream(stream))) return rc;
  // split-K launch: conv_split_reduce_kernel wrote the planes and one word of max|y| per workgroup. Otherwise (the tiles
  // filled the chip): the separate pass, its bound from the epilogue's per-channel maxima (+ the residual's bound)
  if (a.split_parts > 1) {
    *out_n_host = a.nblocks * (128 * 128 / 4 / 256);   // workgroups of conv_split_reduce_kernel<128>
    return YOLO_OK;
  }
  *out_n_host = 0;
  return launch_split_planes_absmax(y, (long long)d->N * d->Ho * d->Wo, d->Cout, absmax,
                                    reinterpret_cast<const float*>(a.pl_res_bound), a.pl_res_n, out_planes, out_bound,
                                    as_stream(stream));
}

extern "C" int yolo_absmax_words(const float* x, long long n, unsigned* words, int* n_words_host, void* stream) {
  YOLO_REQUIRE(x && words && n_words_host && n > 0, "absmax_words: bad args");
  return launch_absmax_words(x, n, words, n_words_host, as_stream(stream));
}

extern "C" int yolo_stem_filter_prep(const float* w, const float* bias, float* wt, void* stream) {
  YOLO_REQUIRE(w && wt, "stem_filter_prep: null pointer");
  return launch_stem_filter_prep(w, bias, wt, as_stream(stream));
}

extern "C" int yolo_stem_fwd_infer_unit(const yolo_conv_desc* d, const float* x, const float* wt, int epilogue,
                                        const float* scale, const float* shift, const float* pred, const void* in_bound,
                                        int in_n, float* y, void* out_planes, unsigned* out_words, int* out_n_host,
                                        void* stream) {
  if (int rc = validate_desc(d)) return rc;
  YOLO_REQUIRE(x && wt && scale && shift && pred && in_bound && out_planes && out_words && out_n_host,
               "stem_fwd_infer_unit: null pointer");
  YOLO_REQUIRE(stem_infer_supported(d), "stem_fwd_infer_unit: Conv2D(32, 3, strides 1, padding same) on 3 channels only");
  YOLO_REQUIRE(epilogue == YOLO_EPI_AFFINE_LEAKY || epilogue == YOLO_EPI_AFFINE_MISH || epilogue == YOLO_EPI_AFFINE,
               "stem_fwd_infer_unit: bad epilogue %d", epilogue);
  YOLO_REQUIRE(in_n >= 1 && in_n <= YOLO_INFER_BOUND_WORDS, "stem_fwd_infer_unit: a bound is 1..%d words", YOLO_INFER_BOUND_WORDS);
  StemEpiArgs e;
  e.scale = scale; e.shift = shift;
  e.act = epilogue == YOLO_EPI_AFFINE_LEAKY ? YOLO_ACT_LEAKY : epilogue == YOLO_EPI_AFFINE_MISH ? YOLO_ACT_MISH : YOLO_ACT_LINEAR;
  e.planes = out_planes; e.pred = pred;
  e.in_words = reinterpret_cast<const unsigned*>(in_bound); e.in_n = in_n; e.out_words = out_words;
  return launch_stem_infer(d, x, wt, y, e, out_n_host, as_stream(stream));
}

extern "C" int yolo_fold_bound(const void* words, int n, float* out_bound, void* stream) {
  YOLO_REQUIRE(words && out_bound && n >= 1 && n <= YOLO_INFER_BOUND_WORDS, "fold_bound: 1..%d words", YOLO_INFER_BOUND_WORDS);
  hipLaunchKernelGGL(fold_bound_kernel, dim3(1), dim3(256), 0, as_stream(stream), reinterpret_cast<const unsigned*>(words), n,
                     out_bound);
  return check_launch("fold_bound_kernel");
}

extern "C" int yolo_split_planes_absmax(const float* x, long long rows, int C, const unsigned* absmax,
                                        const float* extra_bound, void* planes, float* out_bound, void* stream) {
  YOLO_REQUIRE(x && planes && absmax, "split_planes_absmax: null pointer");
  return launch_split_planes_absmax(x, rows, C, absmax, extra_bound, 1, planes, out_bound, as_stream(stream));
}

extern "C" int yolo_split_planes_concat(const float* const* srcs_host, const int* channels_host,
                                        const float* const* bounds_host, int nsrc, long long rows, void* planes, float* dst32,
                                        float* out_bound, void* stream) {
  YOLO_REQUIRE(srcs_host && channels_host && bounds_host && planes && nsrc >= 1 && nsrc <= 4 && rows > 0,
               "split_planes_concat: 1..4 sources, non-null tables");
  int C = 0;
  for (int i = 0; i < nsrc; ++i) {
    YOLO_REQUIRE(srcs_host[i] && bounds_host[i] && channels_host[i] > 0 && (channels_host[i] % 8) == 0,
                 "split_planes_concat: every source needs a pointer, a bound and a channel count that is a multiple of 8");
    C += channels_host[i];
  }
  YOLO_REQUIRE((C % 16) == 0, "split_planes_concat: the concatenated channel count must be a multiple of 16");
  return launch_split_planes_concat(srcs_host, channels_host, bounds_host, nsrc, rows, planes, dst32, out_bound,
                                    as_stream(stream));
}

extern "C" int yolo_split_planes_concat_ex(const float* const* srcs_host, const int* channels_host,
                                           const float* const* bounds_host, const int* bound_words_host,
                                           const int* upsample_host, int H, int W, int nsrc, long long rows, void* planes,
                                           float* dst32, float* out_bound, void* stream) {
  YOLO_REQUIRE(srcs_host && channels_host && bounds_host && planes && nsrc >= 1 && nsrc <= 4 && rows > 0,
               "split_planes_concat: 1..4 sources, non-null tables");
  int C = 0;
  bool any_up = false;
  for (int i = 0; i < nsrc; ++i) {
    YOLO_REQUIRE(srcs_host[i] && bounds_host[i] && channels_host[i] > 0 && (channels_host[i] % 8) == 0,
                 "split_planes_concat: every source needs a pointer, a bound and a channel count that is a multiple of 8");
    YOLO_REQUIRE(bound_words_host == nullptr || (bound_words_host[i] >= 1 && bound_words_host[i] <= YOLO_INFER_BOUND_WORDS),
                 "split_planes_concat: a bound is 1..%d words", YOLO_INFER_BOUND_WORDS);
    any_up = any_up || (upsample_host != nullptr && upsample_host[i] != 0);
    C += channels_host[i];
  }
  YOLO_REQUIRE((C % 16) == 0, "split_planes_concat: the concatenated channel count must be a multiple of 16");
  YOLO_REQUIRE(!any_up || (H > 0 && W > 0 && (H % 2) == 0 && (W % 2) == 0 && rows % ((long long)H * W) == 0),
               "split_planes_concat: an upsampled source needs even H, W with rows = N * H * W (H=%d W=%d rows=%lld)", H, W, rows);
  return launch_split_planes_concat(srcs_host, channels_host, bounds_host, nsrc, rows, planes, dst32, out_bound,
                                    as_stream(stream), bound_words_host, upsample_host, H, W);
}

extern "C" int yolo_conv2d_fwd_head_unit(const yolo_conv_desc* d, const void* x_planes, const void* w_planes,
                                         const float* bias, int A, int C, int version, const float* anchors, float* t,
                                         float* y, void* stream) {
  if (int rc = validate_desc(d)) return rc;
  YOLO_REQUIRE(x_planes && w_planes && t && y && A > 0 && C > 0, "conv_fwd_head_unit: bad args");
  YOLO_REQUIRE(d->Cout == A * (5 + C), "conv_fwd_head_unit: Cout = %d is not A (5 + C) = %d", d->Cout, A * (5 + C));
  GatherConvArgs a{};
  a.src = reinterpret_cast<const float*>(x_planes);
  a.wgt = reinterpret_cast<const float*>(w_planes);
  a.bias = bias;
  a.dst = t;
  fill_fwd_args(d, a);
  YOLO_REQUIRE(gather_planes_supported(a), "conv_fwd_head_unit: needs Cin %% 16 == 0 and Cout >= 32");
  if ((version == YOLO_HEAD_V3 || version == YOLO_HEAD_V4) && anchors != nullptr) {
    a.head_y = y;
    a.head_anchors = anchors;
    a.head_A = A;
    a.head_C = C;
    if (conv_small_head_supported(a)) {
      int nwg = 0;
      return launch_conv_small(a, as_stream(stream), &nwg);
    }
    a.head_y = nullptr;
  }
  if (int rc = launch_gather_planes(a, as_stream(stream))) return rc;
  return yolo_head_act_fwd(t, (long long)d->N * d->Ho * d->Wo, A, C, version, anchors, y, stream);
}

extern "C" size_t yolo_planes_bytes(long long rows, int C) {
  if (rows <= 0 || C <= 0 || (C % 16) != 0) return 0;
  return (size_t)planes_bytes(rows, C);
}

extern "C" int yolo_split_planes_batch(const void* jobs, int njobs, long long total_blocks, void* stream) {
  YOLO_REQUIRE(jobs, "split_planes_batch: null job table");
  return launch_split_planes_batch(jobs, njobs, total_blocks, as_stream(stream));
}

extern "C" int yolo_filter_transpose_batch(const void* jobs, int njobs, long long total_blocks, void* stream) {
  YOLO_REQUIRE(jobs, "filter_transpose_batch: null job table");
  return launch_filter_transpose_batch(jobs, njobs, total_blocks, as_stream(stream));
}

extern "C" int yolo_split_planes(const float* x, long long rows, int C, void* planes, void* stream) {
  YOLO_REQUIRE(x && planes, "split_planes: null pointer");
  return launch_split_planes(x, rows, C, planes, as_stream(stream));
}

extern "C" int yolo_split_planes_padded(const float* x, long long rows, int C_src, int C, void* planes, void* stream) {
  YOLO_REQUIRE(x && planes, "split_planes_padded: null pointer");
  return launch_split_planes_padded(x, rows, C_src, C, planes, as_stream(stream));
}

// the fused BatchNorm-backward reduction of yolo_conv2d_dgrad_planes_bnred (GatherConvArgs::bwd_*)
struct BnRedArgs {
  const float* y;
  const float* scale;
  const float* shift;
  const float* mean;
  const float* invstd;
  int act;
  float* part;
  int cap;
  unsigned* aux;
  int nslots;   // out
};
static void set_bnred(GatherConvArgs& a, const BnRedArgs* b) {
  if (b == nullptr) return;
  a.bwd_y = b->y; a.bwd_scale = b->scale; a.bwd_shift = b->shift; a.bwd_mean = b->mean; a.bwd_invstd = b->invstd;
  a.bwd_act = b->act; a.bwd_part = b->part; a.bwd_cap = b->cap; a.bwd_aux = b->aux;
}

static int dgrad_impl(const yolo_conv_desc* d, const float* dy, const float* wT, float* dx, int accumulate,
                      void* stream, bool planes, BnRedArgs* bnred = nullptr) {
  if (int rc = validate_desc(d)) return rc;
  YOLO_REQUIRE(dy && wT && dx, "conv_dgrad: null pointer");
  // hi = ho*sh + r - pad_t  =>  for input-row parity class py (hi = y*sh + py) the taps with
  // (py + pad_t - r) % sh == 0 contribute from ho = y + (py + pad_t - r)/sh.
  const bool flat = (d->Cout % 32) != 0;
  YOLO_REQUIRE(!flat || (d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1),
               "conv_dgrad: Cout %% 32 != 0 only supported for 1x1 stride-1 (head) convs");
  // planes kernels, stride 2: ONE launch for the four parity classes (each class alone re-streams all of dy: 354 MB four
  // times for the 416 -> 208 layer at bs 32). YOLO_DGRAD_CLASSES=0: one launch per class, as the fp32 kernels do.
  static const bool fuse_classes = [] { const char* e = getenv("YOLO_DGRAD_CLASSES"); return !(e && atoi(e) == 0); }();
  if (planes && fuse_classes && d->sh * d->sw > 1 && d->sh * d->sw <= 4) {
    GatherConvArgs a{};
    a.src = dy;
    a.wgt = wT;
    a.dst = dx;
    a.N = d->N; a.Hs = d->Ho; a.Ws = d->Wo; a.Cs = d->Cout;
    a.sy = 1; a.sx = 1;
    a.Hd = d->H; a.Wd = d->W; a.Cd = d->Cin;
    a.osy = d->sh; a.osx = d->sw;
    a.Cout = d->Cin;
    a.ldw = d->kh * d->kw * d->Cout;
    a.accumulate = accumulate;
    a.kw = 1;
    int nt = 0, nc = 0;
    bool ok = true;
    for (int py = 0; py < d->sh && ok; ++py)
      for (int px = 0; px < d->sw && ok; ++px) {
        ClassGeom& c = a.cls[nc++];
        c.Hg = (d->H - py + d->sh - 1) / d->sh;
        c.Wg = (d->W - px + d->sw - 1) / d->sw;
        c.M = (long long)d->N * c.Hg * c.Wg;
        c.ooy = py; c.oox = px; c.tap0 = nt; c.ntaps = 0;
        for (int r = 0; r < d->kh; ++r) {
          const int ty = py + d->pad_t - r;
          if (((ty % d->sh) + d->sh) % d->sh != 0) continue;
          for (int s = 0; s < d->kw; ++s) {
            const int tx = px + d->pad_l - s;
            if (((tx % d->sw) + d->sw) % d->sw != 0) continue;
            a.taps[nt++] = Tap{ty / d->sh, tx / d->sw, (r * d->kw + s) * d->Cout};
            ++c.ntaps;
          }
        }
        if (c.ntaps == 0 || c.Hg <= 0 || c.Wg <= 0) ok = false;   // (a class without taps: the per-class path below)
      }
    if (ok) {
      a.ncls = nc;
      a.M = a.cls[0].M; a.Hg = a.cls[0].Hg; a.Wg = a.cls[0].Wg; a.ooy = 0; a.oox = 0; a.ntaps = a.cls[0].ntaps;
      YOLO_REQUIRE(gather_planes_supported(a), "conv_dgrad_planes: needs Cout %% 16 == 0 and Cin >= 32");
      set_bnred(a, bnred);
      const int rc = launch_gather_planes(a, as_stream(stream));
      if (bnred != nullptr) bnred->nslots = a.bwd_nslots;
      return rc;
    }
  }
  YOLO_REQUIRE(bnred == nullptr || d->sh * d->sw == 1,
               "conv_dgrad_planes_bnred: a strided data gradient needs its parity classes in one launch");
  for (int py = 0; py < d->sh; ++py) {
    for (int px = 0; px < d->sw; ++px) {
      GatherConvArgs a{};
      a.src = dy;
      a.wgt = wT;
      a.bias = nullptr;
      a.dst = dx;
      a.N = d->N; a.Hs = d->Ho; a.Ws = d->Wo; a.Cs = d->Cout;
      a.Hg = (d->H - py + d->sh - 1) / d->sh;
      a.Wg = (d->W - px + d->sw - 1) / d->sw;
      if (a.Hg <= 0 || a.Wg <= 0) continue;
      a.sy = 1; a.sx = 1;
      a.Hd = d->H; a.Wd = d->W; a.Cd = d->Cin;
      a.osy = d->sh; a.osx = d->sw; a.ooy = py; a.oox = px;
      a.Cout = d->Cin;
      a.ldw = d->kh * d->kw * d->Cout;
      a.accumulate = accumulate;
      a.kw = 1; a.pad_t = 0; a.pad_l = 0;
      a.M = (long long)d->N * a.Hg * a.Wg;
      int nt = 0;
      for (int r = 0; r < d->kh; ++r) {
        const int ty = py + d->pad_t - r;
        if (((ty % d->sh) + d->sh) % d->sh != 0) continue;
        for (int s = 0; s < d->kw; ++s) {
          const int tx = px + d->pad_l - s;
          if (((tx % d->sw) + d->sw) % d->sw != 0) continue;
          // exact division (ty, tx are multiples of the stride, possibly negative)
          a.taps[nt++] = Tap{ty / d->sh, tx / d->sw, (r * d->kw + s) * d->Cout};
        }
      }
      a.ntaps = nt;
      if (nt == 0) {
        // this parity class receives no gradient: write zeros unless accumulating
        if (!accumulate) {
          a.ntaps = 1;
          a.taps[0] = Tap{-(1 << 20), -(1 << 20), 0};  // always out of bounds => zeros
        } else {
          continue;
        }
      }
      if (planes) {
        YOLO_REQUIRE(gather_planes_supported(a), "conv_dgrad_planes: needs Cout %% 16 == 0 and Cin >= 32");
        set_bnred(a, bnred);
        if (int rc = launch_gather_planes(a, as_stream(stream))) return rc;
        if (bnred != nullptr) bnred->nslots = a.bwd_nslots;
      } else if (int rc = dispatch_gather(a, flat, as_stream(stream))) {
        return rc;
      }
    }
  }
  return YOLO_OK;
}

extern "C" int yolo_conv2d_dgrad(const yolo_conv_desc* d, const float* dy, const float* wT, float* dx, int accumulate,
                                 void* stream) {
  return dgrad_impl(d, dy, wT, dx, accumulate, stream, false);
}

extern "C" int yolo_conv2d_dgrad_planes(const yolo_conv_desc* d, const void* dy_planes, const void* wT_planes, float* dx,
                                        int accumulate, void* stream) {
  return dgrad_impl(d, reinterpret_cast<const float*>(dy_planes), reinterpret_cast<const float*>(wT_planes), dx,
                    accumulate, stream, true);
}

extern "C" int yolo_bnred_slots_cap(const yolo_conv_desc* d) {
  // row tiles are at least 128 output pixels of the data gradient (= input pixels of the convolution); 2-D patch tiles
  // and parity classes round up per image row / per class: 64 pixels per slot plus slack covers every launcher
  if (d == nullptr) return 0;
  const long long px = (long long)d->N * d->H * d->W;
  const long long cap = px / 64 + 2LL * d->N + 64;
  return cap > 0x7fffffffLL ? 0 : (int)cap;
}

extern "C" int yolo_conv2d_dgrad_planes_bnred(const yolo_conv_desc* d, const void* dy_planes, const void* wT_planes, float* dx,
                                              int accumulate, const float* y, const float* scale, const float* shift,
                                              const float* save_mean, const float* save_invstd, int act, float* partials,
                                              int slots_cap, unsigned* bound_aux, int* nslots, void* stream) {
  YOLO_REQUIRE(d && y && scale && shift && save_mean && save_invstd && partials && nslots && slots_cap > 0,
               "conv_dgrad_planes_bnred: bad args");
  YOLO_REQUIRE(d->Cin % 4 == 0, "conv_dgrad_planes_bnred: Cin=%d must be a multiple of 4", d->Cin);
  YOLO_REQUIRE(act >= 0 && act <= 2, "conv_dgrad_planes_bnred: bad activation %d", act);
  BnRedArgs b{y, scale, shift, save_mean, save_invstd, act, partials, slots_cap, bound_aux, 0};
  const int rc = dgrad_impl(d, reinterpret_cast<const float*>(dy_planes), reinterpret_cast<const float*>(wT_planes), dx,
                            accumulate, stream, true, &b);
  *nslots = b.nslots;
  return rc;
}

static void fill_wgrad_args(const yolo_conv_desc* d, WgradArgs& a) {
  a.N = d->N; a.Hs = d->H; a.Ws = d->W; a.Cs = d->Cin;
  a.Hg = d->Ho; a.Wg = d->Wo;
  a.sy = d->sh; a.sx = d->sw;
  a.Cout = d->Cout;
  a.ldw = d->kh * d->kw * d->Cin;
  a.ntaps = d->kh * d->kw;
  a.kw = d->kw; a.pad_t = d->pad_t; a.pad_l = d->pad_l;
  a.M = (long long)d->N * d->Ho * d->Wo;
}

extern "C" int yolo_conv2d_wgrad_planes(const yolo_conv_desc* d, const void* x_planes, const void* dy_planes, float* dw,
                                        void* stream) {
  if (int rc = validate_desc(d)) return rc;
  YOLO_REQUIRE(x_planes && dy_planes && dw, "conv_wgrad_planes: null pointer");
  WgradArgs a{};
  a.src = reinterpret_cast<const float*>(x_planes);
  a.dy = reinterpret_cast<const float*>(dy_planes);
  a.dw = dw;
  fill_wgrad_args(d, a);
  YOLO_REQUIRE(wgrad_planes_supported(a),
               "conv_wgrad_planes: needs Cin %% 16 == 0, Cout %% 16 == 0, Cout >= 32 and kh*kw*Cin >= 64");
  return launch_wgrad_planes(a, as_stream(stream));
}

extern "C" int yolo_conv2d_wgrad(const yolo_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                                 void* stream) {
  if (int rc = validate_desc(d)) return rc;
  YOLO_REQUIRE(x && dy && dw, "conv_wgrad: null pointer");
  WgradArgs a{};
  a.src = x;
  a.dy = dy;
  a.dw = dw;
  fill_wgrad_args(d, a);
  const bool ascalar = (d->Cout % 4) != 0;
  const bool bflat = (d->Cin % 4) != 0;
  int rc;
  hipStream_t st = as_stream(stream);
  if (g_conv_mode == 1 && wgrad_split_supported(a)) rc = launch_wgrad_split(a, st);
  else if (ascalar && bflat) rc = dispatch_wgrad<true, true>(a, st);
  else if (ascalar) rc = dispatch_wgrad<true, false>(a, st);
  else if (bflat) rc = dispatch_wgrad<false, true>(a, st);
  else rc = dispatch_wgrad<false, false>(a, st);
  if (rc) return rc;
  if (dbias != nullptr) return yolo_conv2d_wgrad_bias(dy, a.M, d->Cout, dbias, stream);
  return YOLO_OK;
}

extern "C" int yolo_conv2d_wgrad_bias(const float* dy, long long P, int Cout, float* dbias, void* stream) {
  YOLO_REQUIRE(dy && dbias && P > 0 && Cout > 0, "conv_wgrad_bias: bad args");
  int cw = 32;
  while (cw < Cout && cw < 256) cw <<= 1;
  dim3 block(cw, 256 / cw);
  const int gy = (Cout + cw - 1) / cw;
  long long gx = (P + block.y * 64 - 1) / (block.y * 64);
  if (gx > 1024) gx = 1024;
  if (gx < 1) gx = 1;
  // reproducible form: partials in the first MiB of the wgrad workspace, summed in order (fewer blocks if they do not fit)
  size_t ws_bytes = 0;
  float* part = reinterpret_cast<float*>(wgrad_workspace(&ws_bytes));
  if (part != nullptr) {
    const long long cap = (long long)(WGRAD_WS_COLSUM_BYTES / 4) / ((long long)block.y * Cout);
    if (cap < 1) part = nullptr;
    else if (gx > cap) gx = cap;
  }
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)gx, gy), block, 0, as_stream(stream), dy, P, Cout, dbias, part);
  if (int rc = check_launch("colsum_kernel")) return rc;
  if (part != nullptr) {
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((Cout + 63) / 64), dim3(64, 16), 0, as_stream(stream), part,
                       (int)(gx * block.y), Cout, dbias);
    return check_launch("colsum_finish_kernel");
  }
  return YOLO_OK;
}

extern "C" int yolo_filter_transpose(const float* w, float* wT, int Cout, int taps, int Cin, void* stream) {
  YOLO_REQUIRE(w && wT && Cout > 0 && taps > 0 && Cin > 0, "filter_transpose: bad args");
  dim3 grid((Cin + 31) / 32, (Cout + 31) / 32, taps);
  hipLaunchKernelGGL(filter_transpose_kernel, grid, dim3(256), 0, as_stream(stream), w, wT, Cout, taps, Cin);
  return check_launch("filter_transpose_kernel");
}

// fp32 convolution on the bf16 matrix cores: every fp32 operand is split EXACTLY into three bf16 pieces
// (a = h + m + l, 8 significant bits each = the full 24-bit significand) and the product a*b is formed
// from the six partial products whose weight is >= 2^-16 (hh, hm, mh, hl, lh, mm), each exact in the
// MFMA's fp32 accumulator; the three dropped terms are below 2^-24 of |a*b|, i.e. below one fp32 ulp.
// Result: fp32-accurate convolution (same error as an fp32 FMA chain, verified by the 1e-4 parity tests
// against the float64 oracle) at bf16-MFMA speed / 6:
//     v_mfma_f32_32x32x16_bf16: 1024 FLOP/clk/SIMD  vs  v_mfma_f32_32x32x2_f32: 64 FLOP/clk/SIMD
// so six bf16 passes cost 6/16 of one fp32-MFMA pass -- a 2.67x higher ceiling (417 vs 157 TFLOP/s of
// fp32-equivalent work on MI355X). This is the CDNA4-specific lever: the 16:1 bf16:fp32 matrix-rate ratio.
//
// Same implicit-GEMM formulation, tap table, buffer-load zero padding, XCD tile order, fused BN
// statistics and epilogue as gather_conv_kernel (conv.hip). Differences:
//   * K stage = 16 (one MFMA k-step); per stage a wave issues TM*TN*6 = 24 MFMAs (768 cycles);
//   * staging: fp32 global -> registers -> split by truncation (3 AND + 2 SUB per element, no rounding
//     needed because the pieces are exact) -> three bf16 planes in LDS, rows of 16 bf16 padded to 48 B
//     (the ds_read_b128 fragment reads of 16 distinct rows hit 16 distinct 4-bank groups);
//   * fragments: lane (r = lane&31, h = lane>>5) reads 8 consecutive k of row r at k = 8h (one b128).
#include "conv_args.hpp"
#include <cstdlib>
#include <type_traits>

namespace yolo {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int SPLIT_BK = 16;
constexpr int SPLIT_ROW = 24;  // bf16 elements per LDS row (16 data + 8 pad = 48 B)

struct Planes {
  u32x2 h, m, l;  // 4 bf16 each
};

// exact 3-way split of 4 floats into bf16 planes (truncation: every piece has the sign of the input).
// Written with whole-vector operations: the per-element array form was mis-compiled by hipcc 7.2 (every
// element received element 0's leading piece; scripts/hip_probe/split_probe.cpp is the regression probe).
__device__ __forceinline__ Planes split4(const f32x4 v) {
  const u32x4 mask = {0xFFFF0000u, 0xFFFF0000u, 0xFFFF0000u, 0xFFFF0000u};
  const u32x4 hb = __builtin_bit_cast(u32x4, v) & mask;
  const f32x4 r1 = v - __builtin_bit_cast(f32x4, hb);
  const u32x4 mb = __builtin_bit_cast(u32x4, r1) & mask;
  const f32x4 r2 = r1 - __builtin_bit_cast(f32x4, mb);
  const u32x4 lb = __builtin_bit_cast(u32x4, r2) & mask;
  Planes p;
  p.h = u32x2{(hb[0] >> 16) | hb[1], (hb[2] >> 16) | hb[3]};
  p.m = u32x2{(mb[0] >> 16) | mb[1], (mb[2] >> 16) | mb[3]};
  p.l = u32x2{(lb[0] >> 16) | lb[1], (lb[2] >> 16) | lb[3]};
  return p;
}

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void gather_conv_split_kernel(const GatherConvArgs a) {
  constexpr int NT = 64 * WGM * WGN;
  constexpr int BK = SPLIT_BK;
  constexpr int ROW = SPLIT_ROW;
  constexpr int TM = BM / WGM / 32;
  constexpr int TN = BN / WGN / 32;
  constexpr int RPP = NT / 4;   // rows staged per pass (4 float4 per 16-wide row)
  constexpr int AR = (BM + RPP - 1) / RPP;
  constexpr int BR = (BN + RPP - 1) / RPP;
  static_assert(TM >= 1 && TN >= 1 && AR >= 1 && BR >= 1, "tile config");

  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
  constexpr int PLANE_A = BM * ROW;              // bf16 elements
  constexpr int PLANE_B = BN * ROW;
  constexpr int BUF = 3 * (PLANE_A + PLANE_B);   // bf16 elements per stage buffer

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;

  const int tile = xcd_remap(blockIdx.x, a.nblocks);
  const int tile_n = tile % a.tiles_n;
  const int tile_m = tile / a.tiles_n;
  const long long m0 = (long long)tile_m * BM;
  const int n0 = tile_n * BN;

  const int lrow = tid >> 2;
  const int kcol = (tid & 3) * 4;
  constexpr unsigned OOB = 0xFFFFFFF0u;

  int rowel[AR], ys0[AR], xs0[AR];
  const int HgWg = a.Hg * a.Wg;
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const long long m = m0 + lrow + RPP * i;
    if (m < a.M && lrow + RPP * i < BM) {
      const int n = (int)(m / HgWg);
      const int rem = (int)(m - (long long)n * HgWg);
      const int y = rem / a.Wg;
      const int x = rem - y * a.Wg;
      ys0[i] = y * a.sy;
      xs0[i] = x * a.sx;
      rowel[i] = ((n * a.Hs + ys0[i]) * a.Ws + xs0[i]) * a.Cs;
    } else {
      rowel[i] = 0;
      ys0[i] = -(1 << 28);
      xs0[i] = 0;
    }
  }
  int browel[BR];
#pragma unroll
  for (int j = 0; j < BR; ++j) {
    const int co = n0 + lrow + RPP * j;
    browel[j] = (co < a.Cout && lrow + RPP * j < BN) ? co * a.ldw : -1;
  }

  const int cpt = a.Cs / BK;  // stages per tap
  const int nk = a.ntaps * cpt;

  f32x4 ra[2][AR], rb[2][BR];

  auto load_stage = [&](int kt, auto SET) {
    constexpr int S = decltype(SET)::value;
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.src), 0, (unsigned)((long long)a.N * a.Hs * a.Ws * a.Cs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.wgt), 0, (unsigned)((long long)a.Cout * a.ldw * 4), 0x00020000);
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
  };

  // split + write one operand's staged rows into the three planes of LDS buffer `buf`
  auto store_a = [&](int buf, auto SET) {
    constexpr int S = decltype(SET)::value;
    unsigned short* base = smem16 + buf * BUF;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      if (lrow + RPP * i < BM) {
        const Planes p = split4(ra[S][i]);
        const int o = (lrow + RPP * i) * ROW + kcol;
        *reinterpret_cast<u32x2*>(base + o) = p.h;
        *reinterpret_cast<u32x2*>(base + PLANE_A + o) = p.m;
        *reinterpret_cast<u32x2*>(base + 2 * PLANE_A + o) = p.l;
      }
    }
  };
  auto store_b = [&](int buf, auto SET) {
    constexpr int S = decltype(SET)::value;
    unsigned short* base = smem16 + buf * BUF + 3 * PLANE_A;
#pragma unroll
    for (int j = 0; j < BR; ++j) {
      if (lrow + RPP * j < BN) {
        const Planes p = split4(rb[S][j]);
        const int o = (lrow + RPP * j) * ROW + kcol;
        *reinterpret_cast<u32x2*>(base + o) = p.h;
        *reinterpret_cast<u32x2*>(base + PLANE_B + o) = p.m;
        *reinterpret_cast<u32x2*>(base + 2 * PLANE_B + o) = p.l;
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frag_off = (lane & 31) * ROW + (lane >> 5) * 8;  // row r, k = 8h (bf16 elements)

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  auto stage = [&](int kt, auto SL, auto SS) {
    const int buf = kt & 1;
    if (kt + 2 < nk) load_stage(kt + 2, SL);
    const unsigned short* pa = smem16 + buf * BUF + (wm * TM * 32) * ROW + frag_off;
    const unsigned short* pb = smem16 + buf * BUF + 3 * PLANE_A + (wn * TN * 32) * ROW + frag_off;
    bf16x8 af[3][TM], bf[3][TN];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[p][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(pa + p * PLANE_A + i * 32 * ROW));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bf[p][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(pb + p * PLANE_B + j * 32 * ROW));
    }
    // six partial products, smallest first: l*h, h*l, m*m, m*h, h*m, h*h  (planes: 0 = h, 1 = m, 2 = l)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        f32x16 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bf[0][j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[2][j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], c, 0, 0, 0);
        acc[i][j] = c;
      }
    if (kt + 1 < nk) {
      store_a(buf ^ 1, SS);
      store_b(buf ^ 1, SS);
    }
    __syncthreads();
  };

  load_stage(0, S0{});
  store_a(0, S0{});
  store_b(0, S0{});
  if (nk > 1) load_stage(1, S1{});
  __syncthreads();
  {
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      stage(kt, S0{}, S1{});
      stage(kt + 1, S1{}, S0{});
    }
    if (kt < nk) stage(kt, S0{}, S1{});
  }

  // ---- epilogue: identical to gather_conv_kernel (C/D layout is dtype-independent) ----
  float* smem = reinterpret_cast<float*>(smem16);
  long long* rowoff = reinterpret_cast<long long*>(smem);
  for (int r = tid; r < BM; r += NT) {
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

  float* sred = smem + 2 * BM;
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
        const int c = (wn * TN + j) * 32 + lane;
        sred[(wm * BN + c) * 3 + 0] = s1;
        sred[(wm * BN + c) * 3 + 1] = s2;
        sred[(wm * BN + c) * 3 + 2] = mx;
      }
    }
    __syncthreads();
    for (int c = tid; c < BN; c += NT) {
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

template <int BM, int BN, int WGM, int WGN>
static int launch_split(GatherConvArgs& a, hipStream_t st) {
  const long long tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.Cout + BN - 1) / BN;
  const long long nb = tiles_m * a.tiles_n;
  if (nb <= 0 || nb > 0x7fffffffLL) {
    set_error("conv(split): bad grid %lld", nb);
    return YOLO_ERR_INVALID_ARG;
  }
  a.nblocks = (int)nb;
  constexpr size_t lds = 2 * 3 * (BM + BN) * SPLIT_ROW * sizeof(unsigned short);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_conv_split_kernel<BM, BN, WGM, WGN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((gather_conv_split_kernel<BM, BN, WGM, WGN>), dim3((unsigned)nb), dim3(64 * WGM * WGN), lds, st, a);
  return check_launch("gather_conv_split_kernel");
}

bool gather_split_supported(const GatherConvArgs& a) { return (a.Cs % SPLIT_BK) == 0 && a.Cout > 32; }

int launch_gather_split(GatherConvArgs& a, hipStream_t st) {
  // 8-wave workgroups (4 waves per SIMD with two workgroups per CU) cover the per-stage barrier and the
  // split VALU better than 4-wave ones: +7 % on the 128x128 tile (YOLO_SPLIT_WAVES=4 restores 4 waves)
  static const int waves = [] { const char* e = getenv("YOLO_SPLIT_WAVES"); return e ? atoi(e) : 8; }();
  if (a.Cout <= 64) return launch_split<128, 64, 2, 2>(a, st);
  const long long blocks128 = ((a.M + 127) / 128) * ((a.Cout + 127) / 128);
  if (blocks128 <= 512) return launch_split<64, 128, 1, 4>(a, st);   // (8 waves measured slower here)
  if (waves == 8) return launch_split<128, 128, 4, 2>(a, st);
  return launch_split<128, 128, 2, 2>(a, st);
}

}  // namespace yolo

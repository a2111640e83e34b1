// gather_conv_planes_kernel (conv_planes.hip) rebuilt on the 16x16x32 MFMA shape.
//
// Why it exists: the conv kernels are bound by what the chip can draw, not by their schedule (zero-filled
// operands run the same binary ~1.5x faster), and bare MFMA loops on random operands deliver 1950 TFLOP/s with
// v_mfma_f32_16x16x32_f16 against 1253 with v_mfma_f32_32x32x16_f16 (scripts/hip_probe/mfma_shape_probe.cpp).
// Result: inside the full kernel (DMA + LDS reads + epilogue) the two shapes measure EQUAL within 3 % on every
// layer -- the matrix shape is not where the power goes -- so the 32x32x16 build stays the default and this
// one is selected with YOLO_PLANES_MFMA=16 (same results to rounding, covered by the same tests).
//
// Same operands (planes.hpp), same loaders, same LDS pieces (32 rows x 16 k per DMA instruction). One MFMA
// now spans 32 k = TWO 16-k stages: lane (r = l&15, kq = l>>4) takes the unit (row r, half kq&1) of stage
// kq>>1, i.e. it reads its 16 bytes from one of two adjacent stage buffers (per-lane address; the 16 lanes of
// a ds_read_b128 group still hit 16 distinct 16-byte slots). Per iteration = stage pair:
//   wait vmcnt(0)   my DMAs of pair i+1 have landed (pair i+2 is not issued yet)
//   barrier         everybody's have; everybody finished reading pair i's buffers
//   A fragments of pair i+1 -> the other A register set
//   for each 16-column block j: 6 MFMAs (2 row blocks x 3 passes l*h, h*l, h*h) on pair i, then B_j of pair
//   i+1 is read into the registers just freed; the 4 DMAs of pair i+2 (into pair i's buffers) in between
// LDS ring of 4 stage buffers (64 KB -> 2 workgroups/CU); registers: A 2 x 16, B 32, accumulators 32.
#include "planes.hpp"
#include <cstdlib>
#include <type_traits>

namespace yolo {

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void gather_conv_planes16_kernel(const GatherConvArgs a) {
  constexpr int NW = WGM * WGN;
  constexpr int NT = 64 * NW;
  constexpr int TM = BM / WGM / 16;   // 16-row blocks per wave
  constexpr int TN = BN / WGN / 16;   // 16-column blocks per wave
  constexpr int RBA = BM / 32, RBB = BN / 32;
  static_assert(RBA + RBB <= NW, "at least one loader wave per 32-row block");
  static_assert(TM % 2 == 0 && TN % 2 == 0, "wave tile in 32-row / 32-column pieces");
  constexpr int STAGE_BYTES = (RBA + RBB) * PL_PLANES * 1024;
  constexpr int NBUF = 4;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds_base = (unsigned)(size_t)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  const int tile = xcd_remap(blockIdx.x, a.nblocks);
  const int tile_n = tile % a.tiles_n;
  const int tile_m = tile / a.tiles_n;
  const long long m0 = (long long)tile_m * BM;
  const int n0 = tile_n * BN;
  const int HgWg = a.Hg * a.Wg;

  // ---- loader role (as gather_conv_planes_kernel) ----
  const bool loadA = wave < RBA;
  const int rb = loadA ? wave : (wave - RBA) % RBB;
  const int r = lane & 31, hf = lane >> 5;
  int nimg = 0, ys0 = -(1 << 28), xs0 = 0;
  unsigned rowbaseB = 0;
  if (loadA) {
    const long long m = m0 + rb * 32 + r;
    if (m < a.M) {
      nimg = (int)(m / HgWg);
      const int rem = (int)(m - (long long)nimg * HgWg);
      const int y = rem / a.Wg;
      ys0 = y * a.sy;
      xs0 = (rem - y * a.Wg) * a.sx;
    }
  } else {
    const int co = n0 + rb * 32 + r;
    const unsigned blk = co < a.Cout ? (unsigned)(co >> 4) : (unsigned)a.zero_blk_wgt;
    rowbaseB = blk * (unsigned)((a.ldw >> 4) * PL_RECORD) + (co < a.Cout ? (co & 15) * 16 : 0) + hf * 256;
  }
  const unsigned blkstrideA = (unsigned)((a.Cs >> 4) * PL_RECORD);
  const i32x4 rsrc = planes_rsrc(loadA ? (const void*)a.src : (const void*)a.wgt, loadA ? a.src_bytes : a.wgt_bytes);
  const unsigned lds_mine = lds_base + ((loadA ? 0 : RBA) + rb) * PL_PLANES * 1024;

  const int cpt = a.Cs >> 4;  // stages per tap
  const int nk = a.ntaps * cpt;
  const int npairs = (nk + 1) >> 1;   // an odd last stage is paired with a dummy stage that reads the zero block

  int ld_tap = 0, ld_kb = 0;
  unsigned ld_voff = 0, ld_soff = 0;
  auto loader_tap = [&]() {
    if (ld_tap >= a.ntaps) {
      ld_voff = (loadA ? (unsigned)a.zero_blk_src * blkstrideA : (unsigned)a.zero_blk_wgt * (unsigned)((a.ldw >> 4) * PL_RECORD));
      ld_soff = 0;
      return;
    }
    if (loadA) {
      const int ys = ys0 + a.taps[ld_tap].oy, xs = xs0 + a.taps[ld_tap].ox;
      const bool ok = ((unsigned)ys < (unsigned)a.Hs) && ((unsigned)xs < (unsigned)a.Ws);
      const int pix = (nimg * a.Hs + ys) * a.Ws + xs;
      ld_voff = (ok ? ((unsigned)pix >> 4) : (unsigned)a.zero_blk_src) * blkstrideA + (ok ? (pix & 15) * 16 : 0) + hf * 256;
      ld_soff = 0;
    } else {
      ld_voff = rowbaseB;
      ld_soff = (unsigned)(a.taps[ld_tap].woff >> 4) * PL_RECORD;
    }
  };
  auto loader_next = [&]() {
    ld_soff += PL_RECORD;
    if (++ld_kb == cpt) {
      ld_kb = 0;
      ++ld_tap;
      loader_tap();
    }
  };
  // DMA d (0..3) of a stage pair: stage d>>1 of the pair, plane d&1
  auto issue_dma = [&](int d, int pair_buf) {
    const int p = d & 1;
    const unsigned so = __builtin_amdgcn_readfirstlane(ld_soff + p * 512);
    const unsigned l = __builtin_amdgcn_readfirstlane(lds_mine + (pair_buf + (d >> 1)) * STAGE_BYTES + p * 1024);
    dma16(rsrc, ld_voff, so, l);
    if (p == 1) loader_next();
  };
  auto issue_pair = [&](int pair_buf) {
#pragma unroll
    for (int d = 0; d < 4; ++d) issue_dma(d, pair_buf);
  };

  f32x4v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

  // fragment addressing: lane (r16, kq) -> stage kq>>1 of the pair, unit (row r16 of the 16-row sub-block, half kq&1)
  const int r16 = lane & 15, kq = lane >> 4;
  const unsigned lane_off = (unsigned)((kq >> 1) * STAGE_BYTES + (r16 + 32 * (kq & 1)) * 16);
  f16x8 fa[2][PL_PLANES][TM], fb[PL_PLANES][TN];
  auto read_a = [&](int pair_buf, auto SET) {
    constexpr int S = decltype(SET)::value;
    const unsigned char* sb = smem + pair_buf * STAGE_BYTES + lane_off;
#pragma unroll
    for (int p = 0; p < PL_PLANES; ++p)
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[S][p][i] = *reinterpret_cast<const f16x8*>(sb + ((wm * (TM / 2) + (i >> 1)) * PL_PLANES + p) * 1024 + (i & 1) * 256);
  };
  auto read_b = [&](int pair_buf, int j) {
    const unsigned char* sb = smem + pair_buf * STAGE_BYTES + lane_off;
#pragma unroll
    for (int p = 0; p < PL_PLANES; ++p)
      fb[p][j] = *reinterpret_cast<const f16x8*>(sb + ((RBA + wn * (TN / 2) + (j >> 1)) * PL_PLANES + p) * 1024 + (j & 1) * 256);
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  // prologue: pairs 0 and 1 in flight (4 DMAs each per wave), pair 0's fragments in registers
  loader_tap();
  issue_pair(0);
  issue_pair(2);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_a(0, S0{});
#pragma unroll
  for (int j = 0; j < TN; ++j) read_b(0, j);

  // iteration i: MFMAs of pair i | fragment reads of pair i+1 (buffers nb) | DMA issue of pair i+2 (into pair
  // i's buffers cb: everybody finished reading them before this iteration's barrier)
  auto step = [&](int cb, int nb, bool more, auto CUR, auto NXT) {
    constexpr int S = decltype(CUR)::value;
    if (more) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of pair i+1 have landed
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my reads of pair i's buffers are done
      __builtin_amdgcn_s_barrier();
      read_a(nb, NXT);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int pa = (q == 0) ? 1 : 0;   // (A plane, B plane) = (l,h) (h,l) (h,h)
        const int pb = (q == 1) ? 1 : 0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[S][pa][i], fb[pb][j], acc[i][j], 0, 0, 0);
      }
      if (more) {
        __builtin_amdgcn_sched_barrier(0);
        if (j < 4) issue_dma(j, cb);       // the 4 DMAs of pair i+2 ride behind the first 4 column blocks
        read_b(nb, j);                      // B_j of pair i+1 into the registers this column block just freed
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (more && TN < 4) {
#pragma unroll
      for (int d = TN; d < 4; ++d) issue_dma(d, cb);
    }
  };
  {
    int cb = 0, nb = 2;
    int i = 0;
    for (; i + 1 < npairs; i += 2) {
      step(cb, nb, true, S0{}, S1{});
      step(nb, cb, i + 2 < npairs, S1{}, S0{});
    }
    if (i < npairs) step(cb, nb, false, S0{}, S1{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // every wave is done with the stage buffers: the epilogue reuses them

  // ---- epilogue: 16x16 C/D layout: lane l holds column l&15, rows 4*(l>>4) + e, e = 0..3 ----
  float* smf = reinterpret_cast<float*>(smem);
  long long* rowoff = reinterpret_cast<long long*>(smf);
  for (int rr = tid; rr < BM; rr += NT) {
    const long long m = m0 + rr;
    long long off = -1;
    if (m < a.M) {
      const int n = (int)(m / HgWg);
      const int rem = (int)(m - (long long)n * HgWg);
      const int y = rem / a.Wg;
      const int x = rem - y * a.Wg;
      off = (((long long)n * a.Hd + (y * a.osy + a.ooy)) * a.Wd + (x * a.osx + a.oox)) * a.Cd;
    }
    rowoff[rr] = off;
  }
  __syncthreads();

  const float unscale =
      reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.src) + a.src_bytes - PL_HEADER)[2] *
      reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.wgt) + a.wgt_bytes - PL_HEADER)[2];
  float* sred = smf + 2 * BM;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int cl = (wn * TN + j) * 16 + r16;   // column inside the workgroup tile
    const int col = n0 + cl;
    const bool cok = col < a.Cout;
    const float bv = (a.bias != nullptr && cok) ? a.bias[col] : 0.f;
    float s1 = 0.f, s2 = 0.f, mx = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = (wm * TM + i) * 16 + 4 * kq + e;
        const long long off = rowoff[row];
        if (cok && off >= 0) {
          float v = fmaf(acc[i][j][e], unscale, bv);
          if (a.accumulate) v += a.dst[off + col];
          a.dst[off + col] = v;
          s1 += v;
          s2 += v * v;
          mx = fmaxf(mx, fabsf(v));
        }
      }
    }
    if (a.stats != nullptr || a.absmax != nullptr) {
      s1 += __shfl_xor(s1, 16, 64);
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 16, 64);
      s2 += __shfl_xor(s2, 32, 64);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      if (lane < 16) {
        sred[(wm * BN + cl) * 3 + 0] = s1;
        sred[(wm * BN + cl) * 3 + 1] = s2;
        sred[(wm * BN + cl) * 3 + 2] = mx;
      }
    }
  }
  if (a.stats != nullptr || a.absmax != nullptr) {
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
        if (a.absmax != nullptr && __builtin_bit_cast(unsigned, mx) > a.absmax[col])
          atomicMax(&a.absmax[col], __builtin_bit_cast(unsigned, mx));
      }
    }
  }
}

template <int BM, int BN, int WGM, int WGN>
static int launch_planes16(GatherConvArgs& a, hipStream_t st) {
  const long long tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.Cout + BN - 1) / BN;
  const long long nb = tiles_m * a.tiles_n;
  if (nb <= 0 || nb > 0x7fffffffLL) {
    set_error("conv(planes16): bad grid %lld", nb);
    return YOLO_ERR_INVALID_ARG;
  }
  a.nblocks = (int)nb;
  constexpr size_t lds = 4 * (BM / 32 + BN / 32) * PL_PLANES * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_conv_planes16_kernel<BM, BN, WGM, WGN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((gather_conv_planes16_kernel<BM, BN, WGM, WGN>), dim3((unsigned)nb), dim3(64 * WGM * WGN), lds, st, a);
  return check_launch("gather_conv_planes16_kernel");
}

// called by launch_gather_planes (conv_planes.hip) once the operand sizes / zero blocks are filled in
int launch_gather_planes16(GatherConvArgs& a, hipStream_t st) {
  if (a.Cout <= 64) return launch_planes16<128, 64, 4, 2>(a, st);
  return launch_planes16<128, 128, 4, 2>(a, st);
}

}  // namespace yolo

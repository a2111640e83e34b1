// Filter gradient on the bf16 matrix cores with the exact 3-way split (see conv_split.hip):
//   dW[co][j] += sum_p dy[p][co] * src[p -> tap(j)][c(j)],  j = tap*Cin + c.
// Both operands arrive PIXEL-major (the contraction index is the slow one), while the MFMA wants 8
// consecutive k per lane. Instead of transposing in the staging pass, the LDS images stay
// [16 pixels][channels] (plain 8-byte stores of 4 split channels) and the fragments are fetched with
// ds_read_b64_tr_b16: per 16-lane group it reads a 4(k) x 16(channel) block and hands every lane the 4
// k-values of ITS channel -- the transpose is free. Two such reads give the 8 k of one MFMA operand.
// Row stride = channels/2 + 16 dwords (== 16 mod 64), so the 4 rows a 32-lane half touches fall into 4
// disjoint 16-bank windows: conflict-free.
// Same split-K over blocks + fp32 atomics, same 2-set software pipeline, same incremental pixel decode
// as wgrad_kernel (conv.hip).
#include "conv_args.hpp"
#include <type_traits>

namespace yolo {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct WPlanes {
  u32x2 h, m, l;
};
__device__ __forceinline__ WPlanes wsplit4(const f32x4 v) {
  const u32x4 mask = {0xFFFF0000u, 0xFFFF0000u, 0xFFFF0000u, 0xFFFF0000u};
  const u32x4 hb = __builtin_bit_cast(u32x4, v) & mask;
  const f32x4 r1 = v - __builtin_bit_cast(f32x4, hb);
  const u32x4 mb = __builtin_bit_cast(u32x4, r1) & mask;
  const f32x4 r2 = r1 - __builtin_bit_cast(f32x4, mb);
  const u32x4 lb = __builtin_bit_cast(u32x4, r2) & mask;
  WPlanes p;
  p.h = u32x2{(hb[0] >> 16) | hb[1], (hb[2] >> 16) | hb[3]};
  p.m = u32x2{(mb[0] >> 16) | mb[1], (mb[2] >> 16) | mb[3]};
  p.l = u32x2{(lb[0] >> 16) | lb[1], (lb[2] >> 16) | lb[3]};
  return p;
}

constexpr int row_stride_bf16(int channels) {
  // dwords per row = channels/2, padded so that (dwords % 64) is 16 or 48
  const int dw = channels / 2;
  const int m = dw % 64;
  return (m == 16 || m == 48) ? channels : channels + 32;
}

// 8 consecutive k (pixels kq*8 .. kq*8+7 of the stage) of channel `ch0 + (lane&15)` for this lane's
// 16-lane group; `img` points at the plane, rs = row stride in bf16 elements
__device__ __forceinline__ bf16x8 tr_frag(const unsigned short* img, int rs, int k0, int ch_base, int lane) {
  const int i16 = lane & 15;
  const int q = i16 >> 2, pp = i16 & 3;
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  const unsigned short* p0 = img + (k0 + q) * rs + ch_base + 4 * pp;
  const unsigned short* p1 = p0 + 4 * rs;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p1));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void wgrad_split_kernel(const WgradArgs a) {
  constexpr int BK = 16;
  constexpr int TM = BM / WGM / 32;
  constexpr int TN = BN / WGN / 32;
  constexpr int RSA = row_stride_bf16(BM);
  constexpr int RSB = row_stride_bf16(BN);
  constexpr int AF4 = BM / 4, BF4 = BN / 4;
  constexpr int ARP = (256 / AF4) < BK ? (256 / AF4) : BK;
  constexpr int AP = BK / ARP;
  constexpr int BRP = (256 / BF4) < BK ? (256 / BF4) : BK;
  constexpr int BP = BK / BRP;
  static_assert(WGM * WGN == 4 && TM >= 1 && TN >= 1, "tile config");
  constexpr int PLANE_A = BK * RSA, PLANE_B = BK * RSB;
  constexpr int BUF = 3 * (PLANE_A + PLANE_B);

  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];

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
  const bool a_act = a_row < ARP, b_act = b_row < BRP;
  const int HgWg = a.Hg * a.Wg;
  const int Ktot = a.ntaps * a.Cs;
  constexpr unsigned OOB = 0xFFFFFFF0u;

  // this thread's B column group: flattened j -> (tap, c), fixed for the whole pixel loop
  const int jcol = j0 + b_col;
  const bool bok = jcol < Ktot;
  const int bt = bok ? jcol / a.Cs : 0;
  const int br = bt / a.kw;
  const int boy = br - a.pad_t, box = (bt - br * a.kw) - a.pad_l, bc = jcol - bt * a.Cs;
  const bool aok_col = co0 + a_col < a.Cout;

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

  auto load_stage = [&](int kt, auto SET) {
    constexpr int S = decltype(SET)::value;
    const __amdgpu_buffer_rsrc_t rsrcA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (unsigned)(a.M * a.Cout * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.src), 0, (unsigned)((long long)a.N * a.Hs * a.Ws * a.Cs * 4), 0x00020000);
    const long long pbase = p_begin + (long long)kt * BK;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const long long p = pbase + a_row + i * ARP;
      const bool ok = a_act && aok_col && (p < p_end);
      const unsigned off = ok ? (unsigned)(p * a.Cout + co0 + a_col) * 4u : OOB;
      ra[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcA, off, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const long long p = pbase + b_row + i * BRP;
      const int n = pn[i], y = py[i], x = px[i];
      const int ys = y * a.sy + boy, xs = x * a.sx + box;
      const bool ok = b_act && bok && (p < p_end) && ((unsigned)ys < (unsigned)a.Hs) && ((unsigned)xs < (unsigned)a.Ws);
      const unsigned off = ok ? (unsigned)(((n * a.Hs + ys) * a.Ws + xs) * a.Cs + bc) * 4u : OOB;
      rb[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcB, off, 0, 0));
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

  auto store_stage = [&](int buf, auto SET) {
    constexpr int S = decltype(SET)::value;
    unsigned short* sa = smem16 + buf * BUF;
    unsigned short* sb = sa + 3 * PLANE_A;
    if (a_act) {
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const WPlanes p = wsplit4(ra[S][i]);
        const int o = (a_row + i * ARP) * RSA + a_col;
        *reinterpret_cast<u32x2*>(sa + o) = p.h;
        *reinterpret_cast<u32x2*>(sa + PLANE_A + o) = p.m;
        *reinterpret_cast<u32x2*>(sa + 2 * PLANE_A + o) = p.l;
      }
    }
    if (b_act) {
#pragma unroll
      for (int i = 0; i < BP; ++i) {
        const WPlanes p = wsplit4(rb[S][i]);
        const int o = (b_row + i * BRP) * RSB + b_col;
        *reinterpret_cast<u32x2*>(sb + o) = p.h;
        *reinterpret_cast<u32x2*>(sb + PLANE_B + o) = p.m;
        *reinterpret_cast<u32x2*>(sb + 2 * PLANE_B + o) = p.l;
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

  // lane geometry of the transposed reads: 16-lane group g covers channels (g&1)*16.. of a 32-wide block
  // and the k half (g>>1)
  const int grp = lane >> 4;
  const int ch_in_blk = (grp & 1) * 16;
  const int k0 = (grp >> 1) * 8;

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  auto stage = [&](int kt, auto SL, auto SS) {
    const int buf = kt & 1;
    if (kt + 2 < nk) load_stage(kt + 2, SL);
    const unsigned short* sa = smem16 + buf * BUF;
    const unsigned short* sb = sa + 3 * PLANE_A;
    bf16x8 af[3][TM], bf[3][TN];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[p][i] = tr_frag(sa + p * PLANE_A, RSA, k0, (wm * TM + i) * 32 + ch_in_blk, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bf[p][j] = tr_frag(sb + p * PLANE_B, RSB, k0, (wn * TN + j) * 32 + ch_in_blk, lane);
    }
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
    if (kt + 1 < nk) store_stage(buf ^ 1, SS);
    __syncthreads();
  };

  load_stage(0, S0{});
  store_stage(0, S0{});
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

template <int BM, int BN, int WGM, int WGN>
static int launch_ws(WgradArgs& a, hipStream_t st) {
  a.tiles_co = (a.Cout + BM - 1) / BM;
  const int cols = a.ntaps * a.Cs;
  a.tiles_j = (cols + BN - 1) / BN;
  const long long tiles = (long long)a.tiles_co * a.tiles_j;
  long long splits = (1024 + tiles - 1) / tiles;
  const long long max_splits = (a.M + 255) / 256;
  if (splits > max_splits) splits = max_splits;
  if (splits > 65535) splits = 65535;
  if (splits < 1) splits = 1;
  long long chunk = (a.M + splits - 1) / splits;
  chunk = (chunk + 15) / 16 * 16;
  splits = (a.M + chunk - 1) / chunk;
  a.chunk = chunk;
  if (tiles > 0x7fffffffLL || splits > 65535) {
    set_error("wgrad(split): bad grid %lld x %lld", tiles, splits);
    return YOLO_ERR_INVALID_ARG;
  }
  constexpr size_t lds = 2 * 3 * 16 * (row_stride_bf16(BM) + row_stride_bf16(BN)) * sizeof(unsigned short);
  hipLaunchKernelGGL((wgrad_split_kernel<BM, BN, WGM, WGN>), dim3((unsigned)tiles, (unsigned)splits), dim3(256), lds, st,
                     a);
  return check_launch("wgrad_split_kernel");
}

bool wgrad_split_supported(const WgradArgs& a) {
  return (a.Cout % 4) == 0 && (a.Cs % 4) == 0 && a.Cout >= 64 && a.ntaps * a.Cs >= 64;
}

int launch_wgrad_split(WgradArgs& a, hipStream_t st) {
  const int cols = a.ntaps * a.Cs;
  if (a.Cout <= 64 && cols <= 64) return launch_ws<64, 64, 2, 2>(a, st);
  if (a.Cout <= 64) return launch_ws<64, 128, 2, 2>(a, st);
  if (cols <= 64) return launch_ws<128, 64, 2, 2>(a, st);
  return launch_ws<128, 128, 2, 2>(a, st);
}

}  // namespace yolo

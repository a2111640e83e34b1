// Shared pieces of the "planes" kernels (conv_planes.hip, conv_wgrad_planes.hip, bn_act.hip): the operand
// format and the LDS-DMA primitive.
//
// PLANES FORMAT (f16 x 2, scaled). A row-major fp32 matrix X[rows][C] (rows = pixels of an NHWC tensor or
// the output channels of a KRSC filter; C % 16 == 0) is stored as two fp16 planes h, l with
//     s * x  =  h + l + e,   h = fp16(s*x),  l = fp16(s*x - h),   |e| <= max(2^-22 |s*x|, 2^-25)
// (round to nearest: two 11-bit significands = 22-23 significant bits, fp32 has 24; the second term is fp16's
// subnormal spacing, reached by l when |s*x| < 1/8; s is ONE power of two per tensor, chosen from an upper
// bound B >= max|x| so that s*B <= 2^15: no overflow). In units of x: relative error <= 2^-22 (typically 2^-23,
// i.e. 2-4 fp32 roundings) for elements within 2^-18 of the bound, absolute error 2^-40 B below that. A
// product x*y is formed from THREE fp16 MFMA passes h*h + h*l + l*h, each exact in the fp32 accumulator; the
// dropped l*l term is below 2^-22 |x*y|. A K-term dot product therefore carries ~sqrt(K) 2^-22 relative error
// plus K * 2^-40 * B_x * B_y: measured 2e-6 against the exact kernels at K = 9216 -- a few times the
// rounding noise of an fp32 FMA chain and fifty times inside the 1e-4 parity bar -- at HALF the matrix work
// of the exact bf16 x 6 split (conv_split.hip, still selectable: YOLO_CONV_PLANES=0), and the conv kernels
// are bound by exactly that work (measured: 1.86x faster with half the passes).
//
// Layout: 16-row blocks; inside a block one 1024-byte record per 16-channel block kb; inside a record four
// 256-byte sub-blocks (plane p in {h, l}) x (k-half hf in {0, 1}), each 16 rows x 8 fp16:
//     byte(row, c, p) = ((row>>4) * C/16 + c/16) * 1024 + (2*p + (c%16)/8) * 256 + (row&15) * 16 + (c%8)*2
// then ONE all-zero block (target of padding taps and of rows / columns past the edge), then a 256-byte
// header: [0] bit pattern of the bound B (float), [1] s, [2] 1/s. 4 bytes per element, like the fp32 tensor.
// Why 16 x 8 sub-blocks: an MFMA 32x32x16 operand is "lane (r, hf) holds 8 consecutive k of row r" and an
// LDS-DMA instruction writes its 64 lanes' 16 bytes lane-linearly; with lane (r, hf) fetching unit (row r,
// half hf) the LDS image of a 32-row block IS the fragment (linear, conflict-free ds_read_b128) while
// adjacent lanes read adjacent 16-byte units, i.e. whole cache lines.
#pragma once
#include "conv_args.hpp"

namespace yolo {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int PL_PLANES = 2;
constexpr int PL_RECORD = 1024;  // bytes per (16-row block, 16-channel block): 2 planes x 2 halves x 256 B
constexpr int PL_HEADER = 256;   // trailing header: bound bits, scale, 1/scale

__host__ __device__ inline long long planes_body_bytes(long long rows, int C) {
  return ((rows + 15) / 16 + 1) * (long long)(C / 16) * PL_RECORD;
}

// byte offset of the 16-byte unit (row, 8-channel group g8) of plane h; plane l is +512
__device__ __forceinline__ long long planes_unit_offset(long long row, int g8, int C) {
  return ((row >> 4) * (C >> 4) + (g8 >> 1)) * PL_RECORD + (g8 & 1) * 256 + (row & 15) * 16;
}

// power-of-two scale from an upper bound of max|x| (bit pattern of a non-negative float): s * bound <= 2^15
__device__ __forceinline__ float planes_scale_from_bound(unsigned bound_bits) {
  const float b = __builtin_bit_cast(float, bound_bits);
  if (!(b > 0.f) || !(b < 3.0e38f)) return 1.f;         // zero tensor, inf or nan: nothing sensible to do
  int e;
  (void)frexpf(b, &e);                                   // b = m * 2^e, m in [0.5, 1)  =>  b <= 2^e
  int k = 15 - e;
  if (k > 100) k = 100;
  if (k < -100) k = -100;
  return ldexpf(1.f, k);
}

struct Planes8 {
  u32x4 h, l;  // 8 fp16 each
};

// 8 floats (already multiplied by nothing: the scale is applied here) -> the two fp16 planes
__device__ __forceinline__ Planes8 split8(const f32x4 v0, const f32x4 v1, const float s) {
  const f32x4 t0 = v0 * s, t1 = v1 * s;
  const f16x4 h0 = __builtin_convertvector(t0, f16x4), h1 = __builtin_convertvector(t1, f16x4);
  const f32x4 r0 = t0 - __builtin_convertvector(h0, f32x4), r1 = t1 - __builtin_convertvector(h1, f32x4);
  const f16x4 l0 = __builtin_convertvector(r0, f16x4), l1 = __builtin_convertvector(r1, f16x4);
  const f16x8 h = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
  const f16x8 l = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
  Planes8 p;
  p.h = __builtin_bit_cast(u32x4, h);
  p.l = __builtin_bit_cast(u32x4, l);
  return p;
}

__device__ __forceinline__ void store_planes8(unsigned char* planes, long long row, int g8, int C, const f32x4 v0,
                                              const f32x4 v1, const float sc) {
  const Planes8 s = split8(v0, v1, sc);
  unsigned char* o = planes + planes_unit_offset(row, g8, C);
  *reinterpret_cast<u32x4*>(o) = s.h;
  *reinterpret_cast<u32x4*>(o + 512) = s.l;
}

// raw buffer descriptor in SGPRs: base, stride 0, num_records = bytes, raw bounds-checked addressing
__device__ __forceinline__ i32x4 planes_rsrc(const void* base, unsigned bytes) {
  const unsigned long long pb = (unsigned long long)(size_t)base;
  return i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)pb),
               __builtin_amdgcn_readfirstlane((int)(unsigned)(pb >> 32) & 0xFFFF),
               __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
}

// LDS-DMA: 64 lanes x 16 bytes from buffer offset (voff + soff) to LDS bytes [lds, lds + 1024), lane-linear.
// Invisible to hipcc's s_waitcnt bookkeeping: completion is counted by hand (vmcnt) in the kernels.
// M0 = LDS destination; nothing else in these kernels uses M0, so it is not saved.
__device__ __forceinline__ void dma16(const i32x4 rsrc, const unsigned voff, const unsigned soff, const unsigned lds) {
  asm volatile(
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %0, %1, %2 offen lds"
      :
      : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds)
      : "memory");
}

}  // namespace yolo

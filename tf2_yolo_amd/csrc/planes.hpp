// Shared pieces of the "planes" kernels (conv_planes.hip, conv_wgrad_planes.hip): the operand format
// constants and the LDS-DMA primitive. Format: see the header of conv_planes.hip.
#pragma once
#include "conv_args.hpp"

namespace yolo {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int PL_RECORD = 1536;  // bytes per (16-row block, 16-channel block): 3 planes x 2 halves x 256 B

// byte offset of the 16-byte unit (row, 8-channel group g8) of plane 0 ("h"); planes m / l are +512 / +1024
__device__ __forceinline__ long long planes_unit_offset(long long row, int g8, int C) {
  return ((row >> 4) * (C >> 4) + (g8 >> 1)) * PL_RECORD + (g8 & 1) * 256 + (row & 15) * 16;
}

struct Planes8 {
  u32x4 h, m, l;  // 8 bf16 each
};

// exact 3-way truncation split (see conv_split.hip: split4) of 8 floats
__device__ __forceinline__ Planes8 split8(const f32x4 v0, const f32x4 v1) {
  const u32x4 mask = {0xFFFF0000u, 0xFFFF0000u, 0xFFFF0000u, 0xFFFF0000u};
  Planes8 p;
  {
    const u32x4 hb = __builtin_bit_cast(u32x4, v0) & mask;
    const f32x4 r1 = v0 - __builtin_bit_cast(f32x4, hb);
    const u32x4 mb = __builtin_bit_cast(u32x4, r1) & mask;
    const f32x4 r2 = r1 - __builtin_bit_cast(f32x4, mb);
    const u32x4 lb = __builtin_bit_cast(u32x4, r2) & mask;
    p.h[0] = (hb[0] >> 16) | hb[1]; p.h[1] = (hb[2] >> 16) | hb[3];
    p.m[0] = (mb[0] >> 16) | mb[1]; p.m[1] = (mb[2] >> 16) | mb[3];
    p.l[0] = (lb[0] >> 16) | lb[1]; p.l[1] = (lb[2] >> 16) | lb[3];
  }
  {
    const u32x4 hb = __builtin_bit_cast(u32x4, v1) & mask;
    const f32x4 r1 = v1 - __builtin_bit_cast(f32x4, hb);
    const u32x4 mb = __builtin_bit_cast(u32x4, r1) & mask;
    const f32x4 r2 = r1 - __builtin_bit_cast(f32x4, mb);
    const u32x4 lb = __builtin_bit_cast(u32x4, r2) & mask;
    p.h[2] = (hb[0] >> 16) | hb[1]; p.h[3] = (hb[2] >> 16) | hb[3];
    p.m[2] = (mb[0] >> 16) | mb[1]; p.m[3] = (mb[2] >> 16) | mb[3];
    p.l[2] = (lb[0] >> 16) | lb[1]; p.l[3] = (lb[2] >> 16) | lb[3];
  }
  return p;
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

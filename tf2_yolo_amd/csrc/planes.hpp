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

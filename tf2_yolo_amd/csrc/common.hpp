// Shared helpers for the gfx950 kernels of libyolo_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/yolo_hip.h"

namespace yolo {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return YOLO_ERR_LAUNCH;
  }
  return YOLO_OK;
}

#define YOLO_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      ::yolo::set_error(__VA_ARGS__);      \
      return YOLO_ERR_INVALID_ARG;         \
    }                                      \
  } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// grid size for bandwidth-bound grid-stride kernels: enough blocks to fill 256 CUs x 8
inline int stream_grid(long long work_items, int block) {
  long long g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return (int)g;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ double wave_reduce_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace yolo

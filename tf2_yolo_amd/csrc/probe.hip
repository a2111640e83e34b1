// Yardstick kernels for the roofline report (no reference counterpart): what the matrix pipe of THIS chip sustains under its
// power management, measured by the caller with events around one launch.
//   yolo_mfma_probe: bare v_mfma_f32_32x32x16_f16 loops on caller-supplied (random) fp16 register operands, no memory traffic in
//   the loop, two 8-wave workgroups per CU. On random data an MI355X holds about half of its 2.5 PFLOP/s dense fp16 peak
//   (MI355X_MICROARCH.md "DVFS give-back"); a conv kernel's rate divided by this figure (and by the 3 passes per product)
//   says how close it is to what ANY fp16 MFMA code can reach at the clock the chip holds.
#include "common.hpp"

namespace yolo {

typedef _Float16 h8v __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void mfma_probe_kernel(const h8v* __restrict__ in, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const h8v a0 = in[t * 4 + 0], a1 = in[t * 4 + 1], b0 = in[t * 4 + 2], b1 = in[t * 4 + 3];
  f32x16 c0 = {}, c1 = {};
  for (int i = 0; i < iters; ++i) {   // 6 MFMAs of 32x32x16 = 6 * 32768 FLOP per wave and iteration
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c1, 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  out[t] = s;
}

}  // namespace yolo

extern "C" int yolo_mfma_probe(const void* operands_f16, float* sink, int workgroups, int iters, double* flops_host,
                               void* stream) {
  using namespace yolo;
  YOLO_REQUIRE(operands_f16 && sink && workgroups > 0 && workgroups <= 65536 && iters > 0, "yolo_mfma_probe: bad arguments");
  hipLaunchKernelGGL(mfma_probe_kernel, dim3((unsigned)workgroups), dim3(512), 0, as_stream(stream),
                     reinterpret_cast<const h8v*>(operands_f16), sink, iters);
  if (flops_host != nullptr) *flops_host = (double)workgroups * 8.0 * (double)iters * 6.0 * 32768.0;
  return check_launch("mfma_probe_kernel");
}

// Which fp16 MFMA shape sustains more FLOP/s under the chip's power management? Bare MFMA loops on random
// register operands (no memory traffic), same FLOPs per wave, 2 waves per SIMD on every CU.
// build: hipcc -O3 --offload-arch=gfx950 mfma_shape_probe.cpp -o mfma_shape_probe ; run: ./mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k32(const h8* __restrict__ in, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  h8 a0 = in[t * 4 + 0], a1 = in[t * 4 + 1], b0 = in[t * 4 + 2], b1 = in[t * 4 + 3];
  f16v c0 = {}, c1 = {};
  for (int i = 0; i < iters; ++i) {   // 6 MFMAs of 32x32x16 = 6 * 32768 FLOP per wave
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c1, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  out[t] = s;
}
__global__ __launch_bounds__(512) void k16(const h8* __restrict__ in, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  h8 a0 = in[t * 4 + 0], a1 = in[t * 4 + 1], b0 = in[t * 4 + 2], b1 = in[t * 4 + 3];
  f4v c[8] = {};
  for (int i = 0; i < iters; ++i) {   // 12 MFMAs of 16x16x32 = 12 * 16384 FLOP per wave (same as above)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      c[2 * j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(j & 1 ? a1 : a0, j & 2 ? b1 : b0, c[2 * j], 0, 0, 0);
      c[2 * j + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(j & 1 ? a0 : a1, j & 2 ? b0 : b1, c[2 * j + 1], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, j & 1 ? b1 : b0, c[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < 8; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
  out[t] = s;
}

int main() {
  const int blocks = 512, threads = 512, n = blocks * threads;   // 2 workgroups of 8 waves per CU
  std::vector<_Float16> h((size_t)n * 32);
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
  h8* din; float* dout;
  hipMalloc(&din, h.size() * 2); hipMalloc(&dout, n * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int zero = 0; zero < 2; ++zero) {
    if (zero) hipMemset(din, 0, h.size() * 2); else hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int shape = 0; shape < 2; ++shape) {
      const int iters = 20000;
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (shape == 0) hipLaunchKernelGGL(k32, dim3(blocks), dim3(threads), 0, 0, din, dout, iters);
        else hipLaunchKernelGGL(k16, dim3(blocks), dim3(threads), 0, 0, din, dout, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      const double flops = (double)n / 64 * iters * 6 * 32768.0;
      printf("%s operands, %s: %.2f ms  %.0f TFLOP/s\n", zero ? "zero" : "random", shape ? "16x16x32" : "32x32x16", best,
             flops / best / 1e9);
    }
  }
  return 0;
}

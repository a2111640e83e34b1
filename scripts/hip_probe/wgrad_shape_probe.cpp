// Which MFMA shape should the x-window filter gradient (csrc/conv_wgrad_win.hip) be built on? The kernel is at the clock the
// chip holds under it (zero-filled operands: 107 us, random: 142 us), so the question is power, not cycles. This probe runs
// the kernel's inner loop in isolation -- every operand fragment re-read from LDS by ds_read_b64_tr_b16 at the kernel's own
// ratio (40 transposing reads per 27 MFMAs of 32x32x16 = per 54 MFMAs of 16x16x32), nine accumulator chains of three
// passes, two waves per SIMD, 64 KiB of LDS per workgroup, no global traffic -- once per shape, on random and on zero data.
// build: hipcc -O3 --offload-arch=gfx950 wgrad_shape_probe.cpp -o wgrad_shape_probe.bin ; run: ./wgrad_shape_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef s4 __attribute__((address_space(3))) * lds_p;

constexpr int LDS_BYTES = 65536;

__device__ __forceinline__ h8 rd(unsigned a) {   // one fragment = two transposing reads (512 B per wave-instruction, conflict-free)
  const s4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(size_t)a);
  const s4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(size_t)(a + 512));
  const s8 v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
  return __builtin_bit_cast(h8, v);
}

__device__ __forceinline__ void fill_lds(unsigned char* smem, const uint4* in) {
  for (int i = threadIdx.x; i < LDS_BYTES / 16; i += blockDim.x) reinterpret_cast<uint4*>(smem)[i] = in[i];
  __syncthreads();
}

// 32x32x16: per 16-pixel k-step one dy fragment pair (h, l) and nine taps x (h, l) fragments, 27 MFMAs
__global__ __launch_bounds__(256, 2) void k32(const uint4* __restrict__ in, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  fill_lds(smem, in);
  const unsigned lane8 = (threadIdx.x & 63) * 8, wv = (threadIdx.x >> 6) * 1024;
  f16v acc[9] = {};
  unsigned base = wv;
  for (int it = 0; it < iters; ++it) {
    const h8 ah = rd((base + lane8) & 0x7FF8u), al = rd((base + 2048 + lane8) & 0x7FF8u);
    h8 bh[2], bl[2];
    bh[0] = rd((base + 4096 + lane8) & 0x7FF8u);
    bl[0] = rd((base + 6144 + lane8) & 0x7FF8u);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int cur = t & 1, nxt = cur ^ 1;
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[cur], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < 9) {
        const unsigned a = (base + 8192 + t * 4096 + lane8) & 0x7FF8u;
        bh[nxt] = rd(a);
        bl[nxt] = rd(a + 2048);
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[cur], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[cur], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    base = (base + 4096 + 64) & (LDS_BYTES - 1);
    base &= ~63u;
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) s += acc[t][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 16x16x32: per 32 pixels two dy fragment pairs (two 16-filter blocks) and nine taps x two 16-channel halves x (h, l), 108 MFMAs
__global__ __launch_bounds__(256, 2) void k16(const uint4* __restrict__ in, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  fill_lds(smem, in);
  const unsigned lane8 = (threadIdx.x & 63) * 8, wv = (threadIdx.x >> 6) * 1024;
  f4v acc[9][2][2] = {};
  unsigned base = wv;
  for (int it = 0; it < iters; ++it) {
    h8 ah[2], al[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      ah[m] = rd((base + m * 4096 + lane8) & 0x7FF8u);
      al[m] = rd((base + m * 4096 + 2048 + lane8) & 0x7FF8u);
    }
    h8 bh[2], bl[2];   // one (tap, channel half) at a time, two register sets
    bh[0] = rd((base + 8192 + lane8) & 0x7FF8u);
    bl[0] = rd((base + 10240 + lane8) & 0x7FF8u);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 18; ++u) {   // u = tap * 2 + channel half
      const int t = u >> 1, hf = u & 1, cur = u & 1, nxt = cur ^ 1;
      acc[t][hf][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[0], bh[cur], acc[t][hf][0], 0, 0, 0);
      acc[t][hf][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[1], bh[cur], acc[t][hf][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (u + 1 < 18) {
        const unsigned a = (base + 12288 + u * 4096 + lane8) & 0x7FF8u;
        bh[nxt] = rd(a);
        bl[nxt] = rd(a + 2048);
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[t][hf][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0], bl[cur], acc[t][hf][0], 0, 0, 0);
      acc[t][hf][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[1], bl[cur], acc[t][hf][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc[t][hf][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0], bh[cur], acc[t][hf][0], 0, 0, 0);
      acc[t][hf][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[1], bh[cur], acc[t][hf][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    base = (base + 4096 + 64) & (LDS_BYTES - 1);
    base &= ~63u;
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int m = 0; m < 2; ++m) s += acc[t][hf][m][0] + acc[t][hf][m][1] + acc[t][hf][m][2] + acc[t][hf][m][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  const int blocks = 512, threads = 256, n = blocks * threads;   // 2 workgroups of 4 waves per CU: 2 waves per SIMD
  std::vector<_Float16> h(LDS_BYTES / 2);
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
  uint4* din; float* dout;
  hipMalloc(&din, LDS_BYTES); hipMalloc(&dout, n * 4);
  hipFuncSetAttribute((const void*)k32, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipFuncSetAttribute((const void*)k16, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int zero = 0; zero < 2; ++zero) {
    if (zero) hipMemset(din, 0, LDS_BYTES); else hipMemcpy(din, h.data(), LDS_BYTES, hipMemcpyHostToDevice);
    for (int shape = 0; shape < 2; ++shape) {
      // equal FLOPs per launch: one k32 iteration = 27 MFMAs of 32768 FLOP, one k16 iteration = 108 MFMAs of 16384 FLOP
      const int iters = shape == 0 ? 8000 : 4000;
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (shape == 0) hipLaunchKernelGGL(k32, dim3(blocks), dim3(threads), LDS_BYTES, 0, din, dout, iters);
        else hipLaunchKernelGGL(k16, dim3(blocks), dim3(threads), LDS_BYTES, 0, din, dout, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      const double flops = (double)n / 64 * 8000.0 * 27 * 32768.0;
      printf("%s LDS data, %s + transposing fragment reads: %.2f ms  %.0f raw fp16 TFLOP/s (%.0f algorithmic at 3 passes)\n",
             zero ? "zero" : "random", shape ? "16x16x32" : "32x32x16", best, flops / best / 1e9, flops / best / 1e9 / 3);
    }
  }
  return 0;
}

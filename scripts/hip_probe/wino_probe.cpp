// are fewer MFMAs available through Winograd F(2x2, 3x3) on the fp16 x 2-plane operand format?
// One 3x3 stride-1 output needs 9 products directly and 16 / 4 = 4 in the transformed domain: 2.25x fewer matrix
// instructions. But the operands of those instructions are TRANSFORMED inputs, V = B^T d B, and in this format every
// transformed value has to be re-split on the VALU into its two fp16 planes (scale, convert, subtract, convert) after being
// rebuilt from the planes of d (two converts and an add) -- and it feeds only (output columns of the workgroup) x 3 MAC
// passes. This probe runs the two inner loops side by side on the same scaffolding (LDS-resident random data, no global
// traffic, 2 workgroups of 4 waves per CU like the production window kernel):
//   direct:   a wave owns a 64 x 64 tile; per (tap, 16 channels): 8 fragment reads (ds_read_b128), 12 MFMAs 32x32x16
//             -- conv_win_kernel's stage: 4096 outputs x 16 channels x 9 taps per 108 MFMAs
//   winograd: a workgroup owns 32 tiles (128 output pixels) x 64 output columns, a wave 4 of the 16 positions; per 16
//             channels and wave: 24 fragment reads + 24 MFMAs (32 tiles x 64 columns x 4 positions x 3 passes), and its share
//             of the input transform: per 4 steps 32 window reads (16 pixels x 2 planes x 8 channels), the arithmetic
//             on 8 channels (rebuild 16 inputs, 32 adds of B^T d B, x 1/4, split 16 values into planes: REAL arithmetic,
//             the probe keeps the results), 32 writes of the transformed planes
//             -- 8192 outputs x 16 channels per workgroup and step = 2048 per wave per 24 MFMAs
// Reported: raw fp16 MFMA TFLOP/s of each loop and OUTPUT-CHANNEL products per second (outputs x input channels), the figure
// that decides: winograd must deliver >= 1.35x the direct loop's.
// Numerical side of the question: scripts/wino_error.py (NumPy emulation of the same arithmetic against float64).
// build: hipcc -O3 --offload-arch=gfx950 wino_probe.cpp -o wino_probe.bin ; run: ./wino_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int LDS_BYTES = 65536;

__device__ __forceinline__ h8 rd(const unsigned char* smem, unsigned a) {
  return *reinterpret_cast<const h8*>(smem + (a & (LDS_BYTES - 16)));
}
__device__ __forceinline__ void fill_lds(unsigned char* smem, const uint4* in) {
  for (int i = threadIdx.x; i < LDS_BYTES / 16; i += blockDim.x) reinterpret_cast<uint4*>(smem)[i] = in[i];
  __syncthreads();
}

// ---- direct: conv_win_kernel's stage ----
__global__ __launch_bounds__(256, 2) void k_direct(const uint4* __restrict__ in, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  fill_lds(smem, in);
  const unsigned l16 = (threadIdx.x & 63) * 16, wv = (threadIdx.x >> 6) * 4096;
  f16v acc[2][2] = {};
  unsigned base = wv;
  h8 fa[2][2][2], fb[2][2][2];   // [set][plane][block]
  auto read = [&](int s, unsigned b) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[s][p][i] = rd(smem, b + (p * 2 + i) * 1024 + l16);
        fb[s][p][i] = rd(smem, b + 8192 + (p * 2 + i) * 1024 + l16);
      }
  };
  read(0, base);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      base = (base + 1040) & (LDS_BYTES - 16);
      read(1 - s, base);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int pa = q == 0 ? 1 : 0, pb = q == 1 ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s][pa][i], fb[s][pb][j], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float sum = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int q = 0; q < 16; ++q) sum += acc[i][j][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

// ---- winograd: 4 positions per MFMA wave; the input transform on waves of its own (TRANSFORM: 8-wave workgroups, waves
// 4..7 transform while waves 0..3 multiply -- one MFMA wave and one transform wave per SIMD, one barrier per 4 steps) ----
template <bool TRANSFORM>
__global__ __launch_bounds__(TRANSFORM ? 512 : 256, TRANSFORM ? 1 : 2) void k_wino(const uint4* __restrict__ in, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  fill_lds(smem, in);
  const int wave = threadIdx.x >> 6;
  const unsigned l16 = (threadIdx.x & 63) * 16, wv = (wave & 3) * 4096;
  unsigned base = wv;
  if (TRANSFORM && wave >= 4) {
    float keep = 0.f;
    for (int it = 0; it < iters; ++it) {
      // one 4x4 patch x 8 channels per thread and 4 steps: 16 pixels x 2 planes in (ds_read_b128), rebuild, B^T d B,
      // x 1/4, split, 16 positions x 2 planes out (ds_write_b128)
      float d[16][8];
#pragma unroll
      for (int px = 0; px < 16; ++px) {
        const h8 h = rd(smem, base + 16384 + px * 1040 + l16), l = rd(smem, base + 32768 + px * 1040 + l16);
#pragma unroll
        for (int c = 0; c < 8; ++c) d[px][c] = (float)h[c] + (float)l[c];          // rebuild (exact)
      }
      float t[16][8];
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int x = 0; x < 4; ++x) {   // B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
          t[0 * 4 + x][c] = d[0 * 4 + x][c] - d[2 * 4 + x][c];
          t[1 * 4 + x][c] = d[1 * 4 + x][c] + d[2 * 4 + x][c];
          t[2 * 4 + x][c] = d[2 * 4 + x][c] - d[1 * 4 + x][c];
          t[3 * 4 + x][c] = d[1 * 4 + x][c] - d[3 * 4 + x][c];
        }
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        float v[4][8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          v[0][c] = t[y * 4 + 0][c] - t[y * 4 + 2][c];
          v[1][c] = t[y * 4 + 1][c] + t[y * 4 + 2][c];
          v[2][c] = t[y * 4 + 2][c] - t[y * 4 + 1][c];
          v[3][c] = t[y * 4 + 1][c] - t[y * 4 + 3][c];
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          h8 hh, ll;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const float sc = v[x][c] * 0.25f;
            hh[c] = (_Float16)sc;
            ll[c] = (_Float16)(sc - (float)hh[c]);
          }
          const int p = y * 4 + x;
          *reinterpret_cast<h8*>(smem + ((base + 49152 + p * 1024 + l16) & (LDS_BYTES - 16))) = hh;
          *reinterpret_cast<h8*>(smem + ((base + 49152 + 512 + p * 1024 + l16) & (LDS_BYTES - 16))) = ll;
          keep += (float)ll[0];
        }
      }
      base = (base + 4160) & (LDS_BYTES - 16);
      __syncthreads();
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = keep;
    return;
  }
  f16v acc[4][2] = {};   // [position][column block]
  for (int it = 0; it < iters; ++it) {   // one iteration = 4 steps of 16 channels
#pragma unroll
    for (int step = 0; step < 4; ++step) {
      base = (base + 1040) & (LDS_BYTES - 16);
#pragma unroll
      for (int pos = 0; pos < 4; ++pos) {
        h8 a[2], b[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          a[p] = rd(smem, base + (pos * 6 + p) * 1024 + l16);
          b[p][0] = rd(smem, base + (pos * 6 + 2 + p * 2) * 1024 + l16);
          b[p][1] = rd(smem, base + (pos * 6 + 3 + p * 2) * 1024 + l16);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int pa = q == 0 ? 1 : 0, pb = q == 1 ? 1 : 0;
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[pos][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[pa], b[pb][j], acc[pos][j], 0, 0, 0);
        }
      }
    }
    if (TRANSFORM) __syncthreads();   // (the transformed planes of the next 4 steps are complete: one barrier per 4 steps)
  }
  float sum = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 2; ++j)
      for (int q = 0; q < 16; ++q) sum += acc[i][j][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

int main() {
  const int blocks = 512, threads = 256, n = blocks * 512;   // 2 workgroups of 4 waves per CU (winograd + transform: 1 of 8)
  std::vector<_Float16> h(LDS_BYTES / 2);
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
  uint4* din; float* dout;
  hipMalloc(&din, LDS_BYTES); hipMalloc(&dout, n * 4);
  hipMemcpy(din, h.data(), LDS_BYTES, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k_direct, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipFuncSetAttribute((const void*)k_wino<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipFuncSetAttribute((const void*)k_wino<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double prod_direct = 0;
  for (int kind = 0; kind < 3; ++kind) {
    const int iters = kind == 0 ? 6000 : 3000;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL(k_direct, dim3(blocks), dim3(threads), LDS_BYTES, 0, din, dout, iters);
      else if (kind == 1) hipLaunchKernelGGL(k_wino<false>, dim3(blocks), dim3(threads), LDS_BYTES, 0, din, dout, iters);
      else hipLaunchKernelGGL(k_wino<true>, dim3(blocks / 2), dim3(512), LDS_BYTES, 0, din, dout, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    const double waves = kind == 2 ? (double)(blocks / 2) * 4 : (double)blocks * 4;   // MFMA waves
    // MFMAs per wave and iteration: direct 2 stages x 12; winograd 4 steps x 24
    const double mfmas = waves * iters * (kind == 0 ? 24.0 : 96.0);
    const double raw = mfmas * 32768.0 / (best * 1e-3) / 1e12;
    // output-channel products (outputs x input channels): direct: a stage = 4096 outputs x 16 channels x (1/9 of the taps);
    // winograd: a wave-step = 2048 outputs x 16 channels, complete
    const double prod = waves * iters * (kind == 0 ? 2.0 * 4096 * 16 / 9.0 : 4.0 * 2048 * 16) / (best * 1e-3);
    if (kind == 0) prod_direct = prod;
    printf("%-44s %8.2f ms  %6.0f raw fp16 TFLOP/s  %.3e output x channel products / s  (%.2fx direct)\n",
           kind == 0 ? "direct (conv_win stage: 8 reads / 12 MFMAs)" : kind == 1 ? "winograd MFMA side only (24 reads / 24 MFMAs)"
                                                                                  : "winograd with its input transform on the VALU",
           best, raw, prod, prod / prod_direct);
  }
  return 0;
}

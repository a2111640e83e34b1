// One-hot B (identity) with dense mixed-magnitude integer A through v_mfma_f32_32x32x16_bf16:
// D[i][j] must equal A[i][j] for j < 16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
__device__ u16 f2bf(float f) { return (u16)(__builtin_bit_cast(unsigned, f) >> 16); }
__global__ void k(const float* A /*32x16*/, const float* B /*16x32*/, float* D /*32x32*/, int chain) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  u16 a[8], b[8];
  for (int j = 0; j < 8; ++j) { a[j] = f2bf(A[r * 16 + 8 * h + j]); b[j] = f2bf(B[(8 * h + j) * 32 + r]); }
  bf16x8 av, bv, zv;
  __builtin_memcpy(&av, a, 16); __builtin_memcpy(&bv, b, 16);
  u16 z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  __builtin_memcpy(&zv, z, 16);
  f32x16 c = {};
  if (chain) {  // the 5 zero passes of the split scheme first
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zv, bv, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, zv, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zv, zv, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zv, bv, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, zv, c, 0, 0, 0);
  }
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) { const int row = (i & 3) + 8 * (i >> 2) + 4 * h; D[row * 32 + r] = c[i]; }
}
int main() {
  float hA[512], hB[512], hD[1024], *dA, *dB, *dD;
  srand(5);
  for (int i = 0; i < 512; ++i) { hA[i] = (float)(rand() % 511 - 255); hB[i] = 0.f; }
  for (int kk = 0; kk < 16; ++kk) hB[kk * 32 + kk] = 1.f;   // B[k][j] = delta(k, j)
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 4096);
  hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
  for (int chain = 0; chain < 2; ++chain) {
    k<<<1, 64>>>(dA, dB, dD, chain); hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    int wrong = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 16; ++j) if (hD[i * 32 + j] != hA[i * 16 + j]) { if (wrong < 5) printf("  A=%g seen %g\n", hA[i * 16 + j], hD[i * 32 + j]); ++wrong; }
    printf("chain %d: wrong %d / 512\n", chain, wrong);
  }
  return 0;
}

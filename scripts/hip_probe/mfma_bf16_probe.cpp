// Probe the lane->element map and exactness of v_mfma_f32_32x32x16_bf16 with integer data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
__device__ u16 f2bf(float f) { return (u16)(__builtin_bit_cast(unsigned, f) >> 16); }
__global__ void k(const float* A /*32x16*/, const float* B /*16x32*/, float* D /*32x32*/) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  u16 a[8], b[8];
  for (int j = 0; j < 8; ++j) { a[j] = f2bf(A[r * 16 + 8 * h + j]); b[j] = f2bf(B[(8 * h + j) * 32 + r]); }
  bf16x8 av, bv;
  __builtin_memcpy(&av, a, 16); __builtin_memcpy(&bv, b, 16);
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) { const int row = (i & 3) + 8 * (i >> 2) + 4 * h; D[row * 32 + r] = c[i]; }
}
int main() {
  float hA[512], hB[512], hD[1024], *dA, *dB, *dD;
  srand(1);
  auto rbf = [](){ float f = (float)rand() / RAND_MAX * 4.f - 2.f; unsigned u; __builtin_memcpy(&u, &f, 4); u &= 0xFFFF0000u; __builtin_memcpy(&f, &u, 4); return f; };
  const bool rnd = getenv("PROBE_RANDOM") != nullptr;
  for (int i = 0; i < 512; ++i) { hA[i] = rnd ? rbf() : (float)((i * 7) % 13 - 6); hB[i] = rnd ? rbf() : (float)((i * 5) % 11 - 5) * 0.5f; }
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 4096);
  hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double s = 0; for (int kk = 0; kk < 16; ++kk) s += (double)hA[i * 16 + kk] * hB[kk * 32 + j];
    maxerr = fmax(maxerr, fabs(s - hD[i * 32 + j]));
  }
  printf("max abs err %g (D[0][0]=%g D[5][7]=%g)\n", maxerr, hD[0], hD[5 * 32 + 7]);
  return 0;
}

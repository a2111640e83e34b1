// Dense integer (<=255) data through 2 chained v_mfma_f32_32x32x16_bf16: exactness check.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
__device__ u16 f2bf(float f) { return (u16)(__builtin_bit_cast(unsigned, f) >> 16); }
__global__ void k(const float* A /*32x32*/, const float* B /*32x32*/, float* D /*32x32*/) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 c = {};
  for (int s = 0; s < 2; ++s) {
    u16 a[8], b[8];
    for (int j = 0; j < 8; ++j) { a[j] = f2bf(A[r * 32 + 16 * s + 8 * h + j]); b[j] = f2bf(B[(16 * s + 8 * h + j) * 32 + r]); }
    bf16x8 av, bv;
    __builtin_memcpy(&av, a, 16); __builtin_memcpy(&bv, b, 16);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) { const int row = (i & 3) + 8 * (i >> 2) + 4 * h; D[row * 32 + r] = c[i]; }
}
int main(int argc, char** argv) {
  const int top = argc > 1 ? atoi(argv[1]) : 255;
  float hA[1024], hB[1024], hD[1024], *dA, *dB, *dD;
  srand(3);
  for (int i = 0; i < 1024; ++i) { hA[i] = (float)(rand() % (2 * top + 1) - top); hB[i] = (float)(rand() % (2 * top + 1) - top); }
  hipMalloc(&dA, 4096); hipMalloc(&dB, 4096); hipMalloc(&dD, 4096);
  hipMemcpy(dA, hA, 4096, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 4096, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
  double maxerr = 0; int wrong = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double s = 0; for (int kk = 0; kk < 32; ++kk) s += (double)hA[i * 32 + kk] * hB[kk * 32 + j];
    const double e = fabs(s - hD[i * 32 + j]); if (e > 0) ++wrong; maxerr = fmax(maxerr, e);
  }
  printf("top %d: max abs err %g, wrong %d / 1024\n", top, maxerr, wrong);
  return 0;
}

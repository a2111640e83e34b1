// Check the exact 3-way bf16 split used by conv_split.hip on the device.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
struct Planes { u32x2 h, m, l; };
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Planes split4(const f32x4 v) {
  const u32x4 mask = {0xFFFF0000u, 0xFFFF0000u, 0xFFFF0000u, 0xFFFF0000u};
  const u32x4 hb = __builtin_bit_cast(u32x4, v) & mask;
  const f32x4 r1 = v - __builtin_bit_cast(f32x4, hb);
  const u32x4 mb = __builtin_bit_cast(u32x4, r1) & mask;
  const f32x4 r2 = r1 - __builtin_bit_cast(f32x4, mb);
  const u32x4 lb = __builtin_bit_cast(u32x4, r2) & mask;
  Planes p;
  p.h = u32x2{(hb[0] >> 16) | hb[1], (hb[2] >> 16) | hb[3]};
  p.m = u32x2{(mb[0] >> 16) | mb[1], (mb[2] >> 16) | mb[3]};
  p.l = u32x2{(lb[0] >> 16) | lb[1], (lb[2] >> 16) | lb[3]};
  return p;
}
__global__ void k(const f32x4* in, u32x2* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Planes p = split4(in[i]);
  out[3 * i + 0] = p.h; out[3 * i + 1] = p.m; out[3 * i + 2] = p.l;
}
static float bf(unsigned short u) { unsigned x = (unsigned)u << 16; float f; __builtin_memcpy(&f, &x, 4); return f; }
int main() {
  const int n = 1 << 16;
  float* h = (float*)malloc(n * 16); unsigned* o = (unsigned*)malloc(n * 24);
  srand(7);
  for (int i = 0; i < 4 * n; ++i) h[i] = (i & 1) ? (float)(rand() % 511 - 255) : ((float)rand() / RAND_MAX - 0.5f) * 8.f;
  f32x4* d; u32x2* dout; hipMalloc(&d, n * 16); hipMalloc(&dout, n * 24);
  hipMemcpy(d, h, n * 16, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(d, dout, n); hipMemcpy(o, dout, n * 24, hipMemcpyDeviceToHost);
  double maxrel = 0; int bad_int = 0;
  for (int i = 0; i < n; ++i) for (int e = 0; e < 4; ++e) {
    const unsigned short* p = (const unsigned short*)(o + 6 * i);
    const double s = (double)bf(p[e]) + bf(p[4 + e]) + bf(p[8 + e]);
    const double v = h[4 * i + e];
    const double err = fabs(s - v);
    if (v != 0) maxrel = fmax(maxrel, err / fabs(v));
    if ((e & 1) && (bf(p[e]) != v || p[4 + e] != 0)) ++bad_int;
  }
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 4; ++e) {
    const unsigned short* p = (const unsigned short*)(o + 6 * i);
    printf("v=%g  h=%g m=%g l=%g  (raw %04x %04x %04x)\n", h[4 * i + e], bf(p[e]), bf(p[4 + e]), bf(p[8 + e]), p[e], p[4 + e], p[8 + e]);
  }
  printf("max rel residual %g ; integer inputs not captured by the h plane alone: %d\n", maxrel, bad_int);
  return 0;
}

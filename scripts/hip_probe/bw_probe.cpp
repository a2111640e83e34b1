// HBM bandwidth the streaming kernels can hope for on this box: read-only, write-only, copy and read+2 writes (the shape of
// bn_act_fwd8_kernel<true>: 4 B in, 4 + 4 B out) over buffers far larger than the caches, several grid sizes, plain and
// non-temporal accesses. build: hipcc -O3 --offload-arch=gfx950 scripts/hip_probe/bw_probe.cpp -o scripts/hip_probe/bw_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool NT> __global__ __launch_bounds__(256) void k_read(const f4* __restrict__ a, long long n, float* out) {
  f4 s = {0, 0, 0, 0};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    s += NT ? __builtin_nontemporal_load(&a[i]) : a[i];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = 1.f;
}
template <bool NT> __global__ __launch_bounds__(256) void k_write(f4* __restrict__ a, long long n) {
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store(v, &a[i]); else a[i] = v;
  }
}
template <bool NT, int W> __global__ __launch_bounds__(256) void k_copy(const f4* __restrict__ a, f4* __restrict__ b, f4* __restrict__ c, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const f4 v = NT ? __builtin_nontemporal_load(&a[i]) : a[i];
    if (NT) __builtin_nontemporal_store(v, &b[i]); else b[i] = v;
    if (W == 2) { if (NT) __builtin_nontemporal_store(v * 2.f, &c[i]); else c[i] = v * 2.f; }
  }
}
template <class F> static double timeit(F f, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters * 1e-3;
}
int main() {
  const long long bytes = 708837376LL;   // the 416x416x32 x bs 32 tensor
  const long long n = bytes / 16;
  f4 *a, *b, *c; float* out;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&out, 4));
  CK(hipMemset(a, 0, bytes));
  const int grids[] = {1024, 2048, 4096, 8192, 16384, 65536};
  for (int g : grids) {
    double t;
    t = timeit([&] { hipLaunchKernelGGL(k_read<false>, dim3(g), dim3(256), 0, 0, a, n, out); }, 10);
    printf("grid %6d read        %7.1f us %6.2f TB/s\n", g, t * 1e6, bytes / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL(k_read<true>, dim3(g), dim3(256), 0, 0, a, n, out); }, 10);
    printf("grid %6d read  nt    %7.1f us %6.2f TB/s\n", g, t * 1e6, bytes / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL(k_write<false>, dim3(g), dim3(256), 0, 0, b, n); }, 10);
    printf("grid %6d write       %7.1f us %6.2f TB/s\n", g, t * 1e6, bytes / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL(k_write<true>, dim3(g), dim3(256), 0, 0, b, n); }, 10);
    printf("grid %6d write nt    %7.1f us %6.2f TB/s\n", g, t * 1e6, bytes / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL((k_copy<false, 1>), dim3(g), dim3(256), 0, 0, a, b, c, n); }, 10);
    printf("grid %6d copy        %7.1f us %6.2f TB/s\n", g, t * 1e6, 2 * bytes / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL((k_copy<true, 1>), dim3(g), dim3(256), 0, 0, a, b, c, n); }, 10);
    printf("grid %6d copy  nt    %7.1f us %6.2f TB/s\n", g, t * 1e6, 2 * bytes / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL((k_copy<false, 2>), dim3(g), dim3(256), 0, 0, a, b, c, n); }, 10);
    printf("grid %6d 1r2w        %7.1f us %6.2f TB/s\n", g, t * 1e6, 3 * bytes / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL((k_copy<true, 2>), dim3(g), dim3(256), 0, 0, a, b, c, n); }, 10);
    printf("grid %6d 1r2w  nt    %7.1f us %6.2f TB/s\n", g, t * 1e6, 3 * bytes / t / 1e12);
  }
  return 0;
}

// What does a cross-stream dependency cost the stream that PRODUCES it? The training step's backward pass hands every
// layer's dy planes from the compute stream to the filter-gradient stream: hipEventRecord(main) between two kernels of the
// main stream, hipStreamWaitEvent(side). This probe times N iterations of [A on main | B on side, after A | C on main] with
//   0: no dependency at all (B never waits)                       -- the floor
//   1: hipEventRecord(ev, main) behind A, hipStreamWaitEvent(side, ev)       -- what the step does (torch.cuda.Event)
//   2: A launched with hipExtLaunchKernelGGL(..., stopEvent = ev), hipStreamWaitEvent(side, ev)  -- no marker packet on main
//   3: mode 1 plus the reverse edge: main waits for an event the side stream recorded two iterations ago (buffer reuse)
//   4: mode 2 plus the reverse edge
// build: hipcc -O2 --offload-arch=gfx950 event_gap_probe.cpp -o event_gap_probe.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void spin(float* p, int iters) {
  float v = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.0001f;
  p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 200;
  const int iters = argc > 2 ? atoi(argv[2]) : 4000;   // ~20 us per kernel
  float* buf; CK(hipMalloc(&buf, 1 << 24));
  hipStream_t m, s; CK(hipStreamCreate(&m)); CK(hipStreamCreate(&s));
  std::vector<hipEvent_t> ev(N), rev(N);
  for (int i = 0; i < N; ++i) { CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&rev[i], hipEventDisableTiming)); }
  const dim3 g(256), b(256);
  for (int mode = 0; mode <= 4; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
      auto h0 = std::chrono::steady_clock::now();
      CK(hipEventRecord(t0, m));
      for (int i = 0; i < N; ++i) {
        if ((mode == 3 || mode == 4) && i >= 2) CK(hipStreamWaitEvent(m, rev[i - 2], 0));
        if (mode == 2 || mode == 4) hipExtLaunchKernelGGL(spin, g, b, 0, m, nullptr, ev[i], 0, buf, iters);
        else hipLaunchKernelGGL(spin, g, b, 0, m, buf, iters);
        if (mode == 1 || mode == 3) CK(hipEventRecord(ev[i], m));
        if (mode != 0) CK(hipStreamWaitEvent(s, ev[i], 0));
        hipLaunchKernelGGL(spin, g, b, 0, s, buf + (1 << 20), iters);
        if (mode == 3 || mode == 4) CK(hipEventRecord(rev[i], s));
        hipLaunchKernelGGL(spin, g, b, 0, m, buf + (2 << 20), iters);
      }
      CK(hipEventRecord(t1, m));
      auto h1 = std::chrono::steady_clock::now();
      CK(hipStreamSynchronize(m)); CK(hipStreamSynchronize(s));
      float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
      if (rep == 2)
        printf("mode %d: main stream %.3f ms for %d iterations = %.2f us per iteration (2 kernels); host enqueue %.2f us per iteration\n", mode, ms, N,
               ms * 1e3 / N, std::chrono::duration<double, std::micro>(h1 - h0).count() / N);
      CK(hipEventDestroy(t0)); CK(hipEventDestroy(t1));
    }
  }
  return 0;
}

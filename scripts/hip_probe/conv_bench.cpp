// Standalone A/B benchmark of the planes convolution kernels through the C-ABI (no Python, no torch: starts in
// milliseconds on a gpurun box). Variants are values of a library option (yolo_set_option), timed in interleaved
// rounds inside ONE process on the same random data; every variant's output is checked against variant 0's.
//
// build: hipcc -O2 -std=c++17 scripts/hip_probe/conv_bench.cpp -Iinclude -Ltf2_yolo_amd -lyolo_hip \
//            -Wl,-rpath,'$ORIGIN/../../tf2_yolo_amd' -o scripts/hip_probe/conv_bench.bin
// usage: conv_bench.bin <mode fwd|dgrad|wgrad> <optkey> <v0,v1,...> <iters> <rounds> <H,Cin,Cout,k,s,N> [more layers...]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "yolo_hip.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)
#define YK(x)                                                                      \
  do {                                                                             \
    int r_ = (x);                                                                  \
    if (r_ != 0) {                                                                 \
      fprintf(stderr, "%s:%d yolo error %d: %s\n", __FILE__, __LINE__, r_, yolo_last_error()); \
      exit(3);                                                                     \
    }                                                                              \
  } while (0)

static std::vector<int> ints(const char* s) {
  std::vector<int> v;
  std::string t(s);
  size_t p = 0;
  while (p <= t.size()) {
    size_t q = t.find(',', p);
    if (q == std::string::npos) q = t.size();
    v.push_back(atoi(t.substr(p, q - p).c_str()));
    p = q + 1;
  }
  return v;
}

static std::vector<float> g_host[4];
static float* dev_random(size_t n, float scale, unsigned seed, bool zeros) {
  std::vector<float>& h = g_host[seed];
  h.assign(n, 0.f);
  if (!zeros) {
    std::mt19937 g(seed);
    std::normal_distribution<float> d(0.f, 1.f);
    for (size_t i = 0; i < n; ++i) h[i] = d(g) * scale;
  }
  float* p;
  CK(hipMalloc(&p, n * sizeof(float)));
  CK(hipMemcpy(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return p;
}

int main(int argc, char** argv) {
  if (argc < 7) {
    fprintf(stderr, "usage: %s <fwd|dgrad|wgrad> <optkey> <v0,v1,..> <iters> <rounds> <H,Cin,Cout,k,s,N> ...\n", argv[0]);
    return 1;
  }
  const std::string mode = argv[1];
  const int key = atoi(argv[2]);
  const std::vector<int> variants = ints(argv[3]);
  const int iters = atoi(argv[4]), rounds = atoi(argv[5]);
  const bool zeros = getenv("CONV_BENCH_ZEROS") != nullptr;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  // key 99 = composite variant v: option 0 (window kernel) = v % 10, option 2 (stream-K form) = v / 10
#define YOLO_SET(k, v) YK(yolo_set_option(k, v))
  auto set_variant = [&](int v) {
    if (key == 99) { YK(yolo_set_option(0, v % 10)); YK(yolo_set_option(2, v / 10)); }
    else if (key == 97) { YOLO_SET(0, 1); YOLO_SET(4, v); }   // tile order of the window kernel: 0 auto, 1 column-fastest, 2 row-fastest
    else if (key == 98) { YK(yolo_set_option(0, 2)); YK(yolo_set_option(2, v / 10)); YK(yolo_set_option(3, v % 10)); }
    else YK(yolo_set_option(key, v));
  };
  void* ws;
  CK(hipMalloc(&ws, yolo_conv_workspace_bytes()));
  YK(yolo_set_conv_workspace(ws, yolo_conv_workspace_bytes(), st));
  if (getenv("CONV_BENCH_WGRAD_WS") != nullptr) {   // the reproducible filter-gradient form (slabs + ordered reduce), as Network runs it
    void* wws;
    CK(hipMalloc(&wws, yolo_wgrad_workspace_bytes()));
    YK(yolo_set_wgrad_workspace(wws, yolo_wgrad_workspace_bytes()));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int li = 6; li < argc; ++li) {
    const std::vector<int> L = ints(argv[li]);
    if (L.size() != 6) { fprintf(stderr, "bad layer spec %s\n", argv[li]); return 1; }
    const int H = L[0], Cin = L[1], Cout = L[2], k = L[3], s = L[4], N = L[5];
    yolo_conv_desc d{};
    d.N = N; d.H = H; d.W = H; d.Cin = Cin; d.Cout = Cout; d.kh = k; d.kw = k; d.sh = s; d.sw = s;
    if (s == 1) { d.Ho = H; d.Wo = H; d.pad_t = d.pad_l = (k - 1) / 2; }
    else { d.Ho = H / 2; d.Wo = H / 2; d.pad_t = d.pad_l = (k == 3) ? 1 : 0; }   // darknet: pad top/left 1, valid
    const long long Pin = (long long)N * H * H, Pout = (long long)N * d.Ho * d.Wo;
    const int taps = k * k;
    float* x = dev_random((size_t)Pin * Cin, 1.f, 1, zeros);
    float* w = dev_random((size_t)Cout * taps * Cin, 0.05f, 2, zeros);
    float* dy = dev_random((size_t)Pout * Cout, 1.f, 3, zeros);
    float *wT, *y, *dx, *dw;
    CK(hipMalloc(&wT, (size_t)Cout * taps * Cin * 4));
    CK(hipMalloc(&y, (size_t)Pout * Cout * 4));
    CK(hipMalloc(&dx, (size_t)Pin * Cin * 4));
    CK(hipMalloc(&dw, (size_t)Cout * taps * Cin * 4));
    YK(yolo_filter_transpose(w, wT, Cout, taps, Cin, st));
    void *xp, *wp, *dyp, *wTp;
    CK(hipMalloc(&xp, yolo_planes_bytes(Pin, Cin)));
    CK(hipMalloc(&wp, yolo_planes_bytes(Cout, taps * Cin)));
    CK(hipMalloc(&dyp, yolo_planes_bytes(Pout, Cout)));
    CK(hipMalloc(&wTp, yolo_planes_bytes(Cin, taps * Cout)));
    YK(yolo_split_planes(x, Pin, Cin, xp, st));
    YK(yolo_split_planes(w, Cout, taps * Cin, wp, st));
    YK(yolo_split_planes(dy, Pout, Cout, dyp, st));
    YK(yolo_split_planes(wT, Cin, taps * Cout, wTp, st));
    CK(hipStreamSynchronize(st));
    const size_t out_n = mode == "fwd" ? (size_t)Pout * Cout : mode == "dgrad" ? (size_t)Pin * Cin : (size_t)Cout * taps * Cin;
    float* out = mode == "fwd" ? y : mode == "dgrad" ? dx : dw;
    auto run = [&]() {
      if (mode == "fwd") YK(yolo_conv2d_fwd_planes(&d, xp, wp, nullptr, y, nullptr, nullptr, st));
      else if (mode == "dgrad") YK(yolo_conv2d_dgrad_planes(&d, dyp, wTp, dx, 0, st));
      else YK(yolo_conv2d_wgrad_planes(&d, xp, dyp, dw, st));
    };
    const double flops = 2.0 * Pout * Cout * taps * Cin;
    std::vector<float> ref(out_n), got(out_n);
    std::vector<std::vector<float>> times(variants.size());
    std::vector<double> err(variants.size(), 0.0);
    for (size_t vi = 0; vi < variants.size(); ++vi) {
      set_variant(variants[vi]);
      CK(hipMemsetAsync(out, 0, out_n * 4, st));
      run();
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(vi == 0 ? ref.data() : got.data(), out, out_n * 4, hipMemcpyDeviceToHost));
      if (vi > 0) {
        double mx = 0, rm = 0;
        for (size_t i = 0; i < out_n; ++i) {
          mx = std::max(mx, (double)fabsf(got[i] - ref[i]));
          rm = std::max(rm, (double)fabsf(ref[i]));
        }
        err[vi] = mx / (rm > 0 ? rm : 1);
        if (err[vi] > 1e-4) {   // where: count bad 128-row x 128-column tiles (output viewed as [rows][C])
          const size_t C = mode == "fwd" ? (size_t)Cout : mode == "dgrad" ? (size_t)Cin : (size_t)taps * Cin;
          const size_t rows = out_n / C;
          size_t bad = 0, first = (size_t)-1, bad_tiles = 0;
          std::vector<char> tb(((rows + 127) / 128) * ((C + 127) / 128), 0);
          for (size_t i = 0; i < out_n; ++i)
            if (fabsf(got[i] - ref[i]) > 1e-4 * rm) {
              ++bad;
              if (first == (size_t)-1) first = i;
              tb[(i / C / 128) * ((C + 127) / 128) + (i % C) / 128] = 1;
            }
          for (char c : tb) bad_tiles += c;
          if (mode == "fwd" && k == 3 && s == 1) {   // which 16-channel blocks does the bad value contain?
            const size_t row = first / C, col = first % C;
            const int n = (int)(row / ((size_t)H * H)), yy = (int)(row % ((size_t)H * H)) / H, xx = (int)(row % H);
            printf("  per-channel-block contributions to (row %zu, col %zu):", row, col);
            double tot = 0;
            for (int cb = 0; cb < Cin / 16; ++cb) {
              double sblk = 0;
              for (int r = 0; r < 3; ++r)
                for (int q = 0; q < 3; ++q) {
                  const int ys = yy + r - 1, xs = xx + q - 1;
                  if (ys < 0 || ys >= H || xs < 0 || xs >= H) continue;
                  for (int c = cb * 16; c < cb * 16 + 16; ++c)
                    sblk += (double)g_host[1][(((size_t)n * H + ys) * H + xs) * Cin + c] *
                            (double)g_host[2][(((size_t)col * 3 + r) * 3 + q) * Cin + c];
                }
              printf(" %.5f", sblk);
              tot += sblk;
            }
            printf("  | total %.5f  got %.5f\n", tot, got[first]);
          }
          printf("  variant %d: %zu bad elements in %zu of %zu tiles(128x128); first at row %zu col %zu (got %g ref %g)\n",
                 variants[vi], bad, bad_tiles, tb.size(), first / C, first % C, got[first], ref[first]);
          for (size_t t = 0, shown = 0; t < tb.size() && shown < 24; ++t)
            if (tb[t]) { printf("    bad tile m=%zu n=%zu\n", t / ((C + 127) / 128), t % ((C + 127) / 128)); ++shown; }
        }
      }
      run();  // warm
    }
    for (int r = 0; r < rounds; ++r)
      for (size_t vi = 0; vi < variants.size(); ++vi) {
        set_variant(variants[vi]);
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) run();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        times[vi].push_back(ms / iters);
      }
    printf("%s H=%d Cin=%d Cout=%d k=%d s=%d N=%d  (%.2f GFLOP)%s\n", mode.c_str(), H, Cin, Cout, k, s, N, flops / 1e9,
           zeros ? " ZEROS" : "");
    for (size_t vi = 0; vi < variants.size(); ++vi) {
      std::sort(times[vi].begin(), times[vi].end());
      const float mn = times[vi].front(), md = times[vi][times[vi].size() / 2];
      printf("  opt%d=%-3d  min %8.1f us  median %8.1f us  %7.1f TF/s (median)  maxdiff/max|ref| %.2e\n", key, variants[vi],
             mn * 1e3, md * 1e3, flops / md / 1e9, err[vi]);
    }
    if (getenv("CONV_BENCH_SLABS") != nullptr && mode == "fwd") {
      // after a stream-K run with G = 2 * tiles (every tile = two parts of cpt/2 blocks, all slot 0): element (0,0) of
      // tile 0 is float 0 of the slabs of workgroups 0 and 1; compare with the host's half sums
      const int G = atoi(getenv("CONV_BENCH_SLABS"));
      YK(yolo_set_option(0, 2)); YK(yolo_set_option(2, G)); YK(yolo_set_option(3, 0));
      run();
      CK(hipStreamSynchronize(st));
      std::vector<float> h0(4), h1(4);
      const size_t slab0 = (size_t)(1 << 20);
      CK(hipMemcpy(h0.data(), (char*)ws + slab0, 16, hipMemcpyDeviceToHost));
      CK(hipMemcpy(h1.data(), (char*)ws + slab0 + 2 * 65536, 16, hipMemcpyDeviceToHost));
      double half[2] = {0, 0};
      for (int cb = 0; cb < Cin / 16; ++cb)
        for (int r = 0; r < 3; ++r)
          for (int q = 0; q < 3; ++q) {
            const int ys = r - 1, xs = q - 1;
            if (ys < 0 || xs < 0) continue;
            for (int c = cb * 16; c < cb * 16 + 16; ++c)
              half[cb >= Cin / 32] += (double)g_host[1][((size_t)ys * H + xs) * Cin + c] * (double)g_host[2][(((size_t)0 * 3 + r) * 3 + q) * Cin + c];
          }
      std::vector<unsigned> tk(8);
      CK(hipMemcpy(tk.data(), ws, 32, hipMemcpyDeviceToHost));
      printf("  slabs: wg0[0..3] = %g %g %g %g  wg1[0..3] = %g %g %g %g | host halves (row 0, col 0): %g %g | ratios %g %g | out %g | tickets %u %u %u %u\n",
             h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3], half[0], half[1], h0[0] / half[0], h1[0] / half[1], 0.0, tk[0], tk[1], tk[2], tk[3]);
    }
    if (getenv("CONV_BENCH_STAMPS") != nullptr) {
      // diagnostic build of the window kernel (option 1): per-workgroup clock stamps
      //   [0] start [1] sum of prologues [2] of main loops [3] of drains/combines/epilogues [4] end (shader clock);
      //   [5],[6] 100 MHz clock at both ends; [7] XCC id | parts << 8
      const size_t nwg_max = 1 << 16;
      unsigned long long* sb;
      CK(hipMalloc(&sb, nwg_max * 64));
      CK(hipMemset(sb, 0, nwg_max * 64));
      YK(yolo_set_debug_buffer(sb, nwg_max * 64));
      YK(yolo_set_option(1, 1));
      set_variant(atoi(getenv("CONV_BENCH_STAMPS")));
      for (int i = 0; i < 3; ++i) run();
      CK(hipStreamSynchronize(st));
      CK(hipMemset(sb, 0, nwg_max * 64));
      run();
      CK(hipStreamSynchronize(st));
      YK(yolo_set_option(1, 0));
      std::vector<unsigned long long> h(nwg_max * 8);
      CK(hipMemcpy(h.data(), sb, nwg_max * 64, hipMemcpyDeviceToHost));
      std::vector<double> ph[5], clk;
      unsigned long long rmin = ~0ull, rmax = 0;
      std::vector<unsigned long long> starts, ends;
      size_t nwg = 0;
      int xcc_count[16] = {0};
      size_t grid = 0;
      grid = (size_t)(h[7] >> 32);   // the kernel leaves gridDim.x in the upper half of word 7
      {
        double es[6] = {0, 0, 0, 0, 0, 0}, cnt = 0;
        for (size_t i = 0; i < grid; ++i) {
          const unsigned long long* o2 = &h[(grid + i) * 8];
          for (int k2 = 0; k2 < 6; ++k2) es[k2] += (double)o2[k2];
          cnt += (double)o2[6];
        }
        if (cnt > 0)
          printf("  epilogue sections, cycles per epilogue (%d workgroups, %.0f epilogues): rowoff %.0f  scales %.0f  stage0 %.0f  store0 %.0f  stage1 %.0f  store1 %.0f\n",
                 (int)grid, cnt, es[0] / cnt, es[1] / cnt, es[2] / cnt, es[3] / cnt, es[4] / cnt, es[5] / cnt);
      }
      for (size_t i = 0; i < grid; ++i) {
        const unsigned long long* o = &h[i * 8];
        if (o[0] == 0) continue;
        ++nwg;
        for (int k2 = 0; k2 < 3; ++k2) ph[k2].push_back((double)o[k2 + 1]);
        ph[3].push_back((double)((o[7] >> 8) & 0xffffff));
        ph[4].push_back((double)(o[4] - o[0]));
        clk.push_back((double)(o[4] - o[0]) / (double)(o[6] - o[5]) * 100.0);
        rmin = std::min(rmin, o[5]); rmax = std::max(rmax, o[6]);
        starts.push_back(o[5]); ends.push_back(o[6]);
        xcc_count[o[7] & 15]++;
      }
      auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
      auto mean = [](std::vector<double>& v) { double a = 0; for (double x : v) a += x; return v.empty() ? 0.0 : a / v.size(); };
      printf("  stamps: %zu workgroups, span %.1f us; cycles per workgroup (mean/median): prologues %.0f/%.0f  loops %.0f/%.0f  epilogues+combines %.0f/%.0f  parts %.1f/%.0f  total %.0f/%.0f; clock MHz mean %.0f median %.0f\n",
             nwg, (double)(rmax - rmin) / 100.0, mean(ph[0]), med(ph[0]), mean(ph[1]), med(ph[1]), mean(ph[2]), med(ph[2]),
             mean(ph[3]), med(ph[3]), mean(ph[4]), med(ph[4]), mean(clk), med(clk));
      if (nwg == 0) { printf("  (no stamped instantiation for this shape)\n"); CK(hipFree(sb)); goto stamps_done; }
      std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end());
      printf("  start times (us after first) deciles:");
      for (int q = 0; q <= 10; ++q) printf(" %.1f", (double)(starts[std::min(nwg - 1, nwg * q / 10)] - rmin) / 100.0);
      printf("\n  end times deciles:");
      for (int q = 0; q <= 10; ++q) printf(" %.1f", (double)(ends[std::min(nwg - 1, nwg * q / 10)] - rmin) / 100.0);
      printf("\n  workgroups per XCC:");
      for (int q = 0; q < 8; ++q) printf(" %d", xcc_count[q]);
      printf("\n");
      CK(hipFree(sb));
    }
  stamps_done:
    fflush(stdout);
    for (void* p : {(void*)x, (void*)w, (void*)dy, (void*)wT, (void*)y, (void*)dx, (void*)dw, xp, wp, dyp, wTp}) CK(hipFree(p));
  }
  return 0;
}

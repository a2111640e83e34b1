// HBM-bound glue kernels: channel-slice copies (Concatenate), nearest 2x up-sampling,
// max-pooling (2x2 s2, SPP 5/9/13 s1 'same', 2x2 s1 'same'), space_to_depth(2), Add,
// detection-head activations and the optimizer step.
//
// Replaces UpSampling2D / Concatenate / Add (yolov3/models/darknet.py:83-93,
// yolov3/models/backbone.py:71), MaxPooling2D (yolov2/models/backbone.py:44-65,
// yolov4/models/backbone.py:176-185), tf.nn.space_to_depth (yolov2/models/darknet.py:46-49),
// the head activations (yolov3/models/__init__.py:40-65 and siblings) and Keras Adam
// (README.md:241).
#include "common.hpp"
#include "planes.hpp"
#include <cfloat>

namespace yolo {

// ---- channel-slice copies ----------------------------------------------------------------
__global__ void copy_in_kernel(const float* __restrict__ src, long long P, int Cs, float* __restrict__ dst, int Cd,
                               int c_off, int vec) {
  if (vec) {
    const int Cs4 = Cs >> 2;
    const long long n = P * Cs4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
      const long long p = i / Cs4;
      const int c = (int)(i - p * Cs4) * 4;
      *reinterpret_cast<f32x4*>(dst + p * Cd + c_off + c) = *reinterpret_cast<const f32x4*>(src + p * Cs + c);
    }
  } else {
    const long long n = P * Cs;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
      const long long p = i / Cs;
      const int c = (int)(i - p * Cs);
      dst[p * Cd + c_off + c] = src[i];
    }
  }
}

__global__ void copy_out_kernel(const float* __restrict__ src, long long P, int Cs, int c_off,
                                float* __restrict__ dst, int Cd, int accumulate, int vec) {
  if (vec) {
    const int Cd4 = Cd >> 2;
    const long long n = P * Cd4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
      const long long p = i / Cd4;
      const int c = (int)(i - p * Cd4) * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(src + p * Cs + c_off + c);
      if (accumulate) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(dst + p * Cd + c);
        v += o;
      }
      *reinterpret_cast<f32x4*>(dst + p * Cd + c) = v;
    }
  } else {
    const long long n = P * Cd;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
      const long long p = i / Cd;
      const int c = (int)(i - p * Cd);
      float v = src[p * Cs + c_off + c];
      if (accumulate) v += dst[i];
      dst[i] = v;
    }
  }
}

// ---- nearest 2x up-sampling --------------------------------------------------------------
__global__ void upsample_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C4, float* __restrict__ y,
                                    int Cy, int c_off) {
  // one thread per OUTPUT float4
  const int Ho = 2 * H, Wo = 2 * W;
  const long long n = (long long)N * Ho * Wo * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long long p = i / C4;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((long long)b * H + (ho >> 1)) * W + (wo >> 1)) * (C4 * 4) + c * 4);
    *reinterpret_cast<f32x4*>(y + (((long long)b * Ho + ho) * Wo + wo) * Cy + c_off + c * 4) = v;
  }
}

__global__ void upsample_bwd_kernel(const float* __restrict__ dy, int N, int H, int W, int C4, int Cy, int c_off,
                                    float* __restrict__ dx, int accumulate) {
  const int Ho = 2 * H, Wo = 2 * W;
  const long long n = (long long)N * H * W * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long long p = i / C4;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const int b = (int)(p / H);
    const float* base = dy + (((long long)b * Ho + 2 * h) * Wo + 2 * w) * Cy + c_off + c * 4;
    f32x4 s = *reinterpret_cast<const f32x4*>(base);
    s += *reinterpret_cast<const f32x4*>(base + Cy);
    s += *reinterpret_cast<const f32x4*>(base + (long long)Wo * Cy);
    s += *reinterpret_cast<const f32x4*>(base + (long long)Wo * Cy + Cy);
    f32x4* o = reinterpret_cast<f32x4*>(dx) + i;
    if (accumulate) s += *o;
    *o = s;
  }
}

__global__ void axpy_kernel(float* __restrict__ a, const float* __restrict__ b, long long n) {
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 v = reinterpret_cast<f32x4*>(a)[i];
    v += reinterpret_cast<const f32x4*>(b)[i];
    reinterpret_cast<f32x4*>(a)[i] = v;
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    a[i] += b[i];
}

__global__ void fill_kernel(float* __restrict__ p, long long n, float v) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    p[i] = v;
}

// ---- max pooling ---------------------------------------------------------------------------
// tf max-pool semantics: padding acts as -inf; the gradient goes to the first maximal element
// in window scan order (row-major), which is what argmax records.
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C, int k, int s, int pad_t,
                                   int pad_l, int Ho, int Wo, float* __restrict__ y, int Cy, int c_off,
                                   int* __restrict__ argmax) {
  const long long n = (long long)N * Ho * Wo * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long long p = i / C;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho);
    const int b = (int)(p / Ho);
    float best = -FLT_MAX;
    int arg = -1;
    for (int r = 0; r < k; ++r) {
      const int h = ho * s + r - pad_t;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int q = 0; q < k; ++q) {
        const int w = wo * s + q - pad_l;
        if ((unsigned)w >= (unsigned)W) continue;
        const long long off = (((long long)b * H + h) * W + w) * C + c;
        const float v = x[off];
        if (arg < 0 || v > best) {
          best = v;
          arg = (int)off;
        }
      }
    }
    y[(((long long)b * Ho + ho) * Wo + wo) * Cy + c_off + c] = best;
    if (argmax) argmax[i] = arg;
  }
}

__global__ void maxpool_bwd_kernel(const float* __restrict__ dy, long long n, int C, int Cy, int c_off,
                                   const int* __restrict__ argmax, float* __restrict__ dx) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long p = i / C;
    const int arg = argmax[i];
    if (arg >= 0) atomicAdd(&dx[arg], dy[p * Cy + c_off + c]);
  }
}

// ---- 2x2 / stride-2 pools whose windows tile the input exactly (H = 2 Ho, W = 2 Wo: the five MaxPooling2D of Darknet-19,
// yolov2/models/backbone.py:42-60, and of tiny-YOLOv3, yolov3/models/darknet.py:107-135) ----
// The general kernels above take one float per thread, three divisions per element and, backward, an atomicAdd into a
// zero-filled tensor (YOLOv2-416 at bs 16: 5 + 5 launches, 0.46 + 0.60 ms, plus 0.24 ms of zero fills). Here a thread owns four
// channels of one OUTPUT pixel: four 16-byte loads forward; backward it writes all four input positions of its window -- the
// gradient where the recorded winner is, zero elsewhere -- so the tensor needs no zero fill and no atomics. Same values, same
// winner (first maximum in row-major window order, v > best).
__global__ __launch_bounds__(256) void maxpool2x2_fwd_kernel(const float* __restrict__ x, long long n4, int Ho, int Wo, int C,
                                                             float* __restrict__ y, int* __restrict__ argmax) {
  const int C4 = C >> 2, W = 2 * Wo;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const long long p = i / C4;             // output pixel (b, ho, wo), row-major
    const int wo = (int)(p % Wo);
    const long long bh = p / Wo;            // b * Ho + ho; the input rows of the window are 2 bh and 2 bh + 1 (H = 2 Ho)
    const long long o00 = ((2 * bh) * W + 2 * wo) * C + c4 * 4;
    f32x4 best = *reinterpret_cast<const f32x4*>(x + o00);
    i32x4 arg = {(int)o00, (int)o00 + 1, (int)o00 + 2, (int)o00 + 3};
#pragma unroll
    for (int t = 1; t < 4; ++t) {
      const long long o = o00 + ((long long)(t >> 1) * W + (t & 1)) * C;
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + o);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (v[e] > best[e]) {
          best[e] = v[e];
          arg[e] = (int)o + e;
        }
    }
    *reinterpret_cast<f32x4*>(y + p * C + c4 * 4) = best;
    if (argmax) *reinterpret_cast<i32x4*>(argmax + p * C + c4 * 4) = arg;
  }
}

template <bool ACCUM>
__global__ __launch_bounds__(256) void maxpool2x2_bwd_kernel(const float* __restrict__ dy, long long n4, int Ho, int Wo, int C,
                                                             const int* __restrict__ argmax, float* __restrict__ dx) {
  const int C4 = C >> 2, W = 2 * Wo;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const long long p = i / C4;
    const int wo = (int)(p % Wo);
    const long long bh = p / Wo;
    const long long o00 = ((2 * bh) * W + 2 * wo) * C + c4 * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + p * C + c4 * 4);
    const i32x4 arg = *reinterpret_cast<const i32x4*>(argmax + p * C + c4 * 4);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const long long o = o00 + ((long long)(t >> 1) * W + (t & 1)) * C;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (arg[e] == (int)o + e) ? g[e] : 0.f;
      f32x4* d = reinterpret_cast<f32x4*>(dx + o);
      if (ACCUM) *d = *d + v;
      else *d = v;
    }
  }
}

// ---- stride-1 'same' pools with large windows (the 5 / 9 / 13 pools of YOLOv4's SPP block on a 19x19 or 13x13 map) ----
// The kernel above reads k^2 values per output one after the other (169 for k = 13: 131 us for a 12 MB tensor). Here a
// workgroup holds the H x W plane of 8 channels of one image in LDS and pools it in two passes -- along the rows (value +
// column of the first maximum), then down the columns of those -- 2k LDS reads per output. Same result bit for bit, same
// winner: the first maximum in row-major window order is the first row holding the maximum and, in it, its first column.
constexpr int POOL_CG = 8;
__global__ __launch_bounds__(256) void maxpool_plane_fwd_kernel(const float* __restrict__ x, int H, int W, int C, int k,
                                                                int pad_t, int pad_l, float* __restrict__ y, int Cy,
                                                                int c_off, int* __restrict__ argmax) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pool_sm[];
  const int HW = H * W, n = HW * POOL_CG;
  float* val = reinterpret_cast<float*>(pool_sm);
  float* rval = val + n;
  unsigned short* rarg = reinterpret_cast<unsigned short*>(rval + n);
  const int groups = C / POOL_CG;
  const int b = blockIdx.x / groups, c0 = (blockIdx.x - b * groups) * POOL_CG;
  const long long base = (long long)b * HW;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int p = e / POOL_CG, c = e - p * POOL_CG;
    val[e] = x[(base + p) * C + c0 + c];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < n; e += 256) {
    const int p = e / POOL_CG, c = e - p * POOL_CG;
    const int h = p / W, wo = p - h * W;
    float best = -FLT_MAX;
    int arg = -1;
    for (int q = 0; q < k; ++q) {
      const int w = wo + q - pad_l;
      if ((unsigned)w >= (unsigned)W) continue;
      const float v = val[(h * W + w) * POOL_CG + c];
      if (arg < 0 || v > best) {
        best = v;
        arg = w;
      }
    }
    rval[e] = best;
    rarg[e] = (unsigned short)(arg < 0 ? 0xFFFF : arg);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < n; e += 256) {
    const int p = e / POOL_CG, c = e - p * POOL_CG;
    const int ho = p / W, wo = p - ho * W;
    float best = -FLT_MAX;
    int arg = -1;
    for (int r = 0; r < k; ++r) {
      const int h = ho + r - pad_t;
      if ((unsigned)h >= (unsigned)H) continue;
      const int idx = (h * W + wo) * POOL_CG + c;
      const unsigned short ra = rarg[idx];
      if (ra == 0xFFFF) continue;
      const float v = rval[idx];
      if (arg < 0 || v > best) {
        best = v;
        arg = h * W + ra;
      }
    }
    y[(base + p) * Cy + c_off + c0 + c] = best;
    if (argmax) argmax[(base + p) * C + c0 + c] = arg < 0 ? -1 : (int)((base + arg) * C + c0 + c);
  }
}

// backward of the same pools as a GATHER: input pixel (h, w) adds, in a fixed order, the gradients of the outputs whose
// window holds it and whose saved winner it is -- no atomics (the scatter form's fp32 atomicAdds made the step's result
// depend on their order, and 169 windows can share one winner), dx += the sum
__global__ __launch_bounds__(256) void maxpool_plane_bwd_kernel(const float* __restrict__ dy, int H, int W, int C, int Cy,
                                                                int c_off, const int* __restrict__ argmax, int k, int pad_t,
                                                                int pad_l, float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pool_sm[];
  const int HW = H * W, n = HW * POOL_CG;
  float* dyv = reinterpret_cast<float*>(pool_sm);
  unsigned short* win = reinterpret_cast<unsigned short*>(dyv + n);
  const int groups = C / POOL_CG;
  const int b = blockIdx.x / groups, c0 = (blockIdx.x - b * groups) * POOL_CG;
  const long long base = (long long)b * HW;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int p = e / POOL_CG, c = e - p * POOL_CG;
    dyv[e] = dy[(base + p) * Cy + c_off + c0 + c];
    const int a = argmax[(base + p) * C + c0 + c];
    win[e] = (unsigned short)(a < 0 ? 0xFFFF : (int)((long long)a / C - base));   // the winner's pixel inside this plane
  }
  __syncthreads();
  for (int e = threadIdx.x; e < n; e += 256) {
    const int p = e / POOL_CG, c = e - p * POOL_CG;
    const int h = p / W, w = p - h * W;
    // output (ho, wo) covers input (h, w) iff ho - pad_t <= h <= ho - pad_t + k - 1
    const int ho_lo = h + pad_t - (k - 1) < 0 ? 0 : h + pad_t - (k - 1), ho_hi = h + pad_t >= H ? H - 1 : h + pad_t;
    const int wo_lo = w + pad_l - (k - 1) < 0 ? 0 : w + pad_l - (k - 1), wo_hi = w + pad_l >= W ? W - 1 : w + pad_l;
    float sum = 0.f;
    for (int ho = ho_lo; ho <= ho_hi; ++ho)
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        const int idx = (ho * W + wo) * POOL_CG + c;
        if (win[idx] == (unsigned short)p) sum += dyv[idx];
      }
    dx[(base + p) * C + c0 + c] += sum;
  }
}

static bool pool_plane_ok(int H, int W, int C, int k, int s, int Ho, int Wo) {
  static const bool on = [] { const char* e = getenv("YOLO_POOL_PLANE"); return !(e && atoi(e) == 0); }();
  return on && s == 1 && Ho == H && Wo == W && k >= 4 && (C % POOL_CG) == 0 && H * W <= 1024;
}

// ---- space_to_depth(2) ----------------------------------------------------------------------
__global__ void s2d_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C, float* __restrict__ y, int Cy,
                               int c_off) {
  // x: [N,H,W,C] -> y slice [N,H/2,W/2,4C], channel (dy*2+dx)*C + c
  const long long n = (long long)N * H * W * C;
  const int Ho = H >> 1, Wo = W >> 1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long long p = i / C;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const int b = (int)(p / H);
    y[(((long long)b * Ho + (h >> 1)) * Wo + (w >> 1)) * Cy + c_off + ((h & 1) * 2 + (w & 1)) * C + c] = x[i];
  }
}
__global__ void s2d_bwd_kernel(const float* __restrict__ dy, int N, int H, int W, int C, int Cy, int c_off,
                               float* __restrict__ dx, int accumulate) {
  const long long n = (long long)N * H * W * C;
  const int Ho = H >> 1, Wo = W >> 1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long long p = i / C;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const int b = (int)(p / H);
    float v = dy[(((long long)b * Ho + (h >> 1)) * Wo + (w >> 1)) * Cy + c_off + ((h & 1) * 2 + (w & 1)) * C + c];
    if (accumulate) v += dx[i];
    dx[i] = v;
  }
}

// ---- detection head -------------------------------------------------------------------------
// v2/v3/v4: one wave-lane per (pixel, anchor, channel) element for the pointwise part; the
// softmax variants (v2 classes, v1 classes) use one thread per (pixel, anchor) group.
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void head_fwd_pointwise_kernel(const float* __restrict__ t, long long P, int A, int C,
                                          const float* __restrict__ anchors, float* __restrict__ y) {
  const int D = 5 + C;
  const long long n = P * A * D;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % D);
    const int a = (int)((i / D) % A);
    const float v = t[i];
    float o;
    if (k == 2 || k == 3) o = expf(v) * anchors[a * 2 + (k - 2)];
    else o = sigmoid_f(v);
    y[i] = o;
  }
}

// softmax over `C` trailing channels of each group; groups of stride D, offset `off`
__global__ void head_softmax_fwd_kernel(const float* __restrict__ t, long long groups, int D, int off, int C,
                                        float* __restrict__ y) {
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < groups;
       g += (long long)gridDim.x * blockDim.x) {
    const float* tp = t + g * D + off;
    float* yp = y + g * D + off;
    float m = -FLT_MAX;
    for (int c = 0; c < C; ++c) m = fmaxf(m, tp[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(tp[c] - m);
    const float inv = 1.f / s;
    for (int c = 0; c < C; ++c) yp[c] = expf(tp[c] - m) * inv;
  }
}

__global__ void head_sigmoid_fwd_kernel(const float* __restrict__ t, long long P, int D, int nsig,
                                        float* __restrict__ y) {
  const long long n = P * nsig;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long p = i / nsig;
    const int k = (int)(i - p * nsig);
    y[p * D + k] = sigmoid_f(t[p * D + k]);
  }
}

__global__ void head_bwd_pointwise_kernel(const float* __restrict__ y, const float* __restrict__ dy, long long P,
                                          int A, int C, int sigmoid_classes, float* __restrict__ dt) {
  const int D = 5 + C;
  const long long n = P * A * D;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % D);
    if (k >= 5 && !sigmoid_classes) continue;
    const float yv = y[i], g = dy[i];
    dt[i] = (k == 2 || k == 3) ? g * yv : g * yv * (1.f - yv);
  }
}

__global__ void head_softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, long long groups,
                                        int D, int off, int C, float* __restrict__ dt) {
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < groups;
       g += (long long)gridDim.x * blockDim.x) {
    const float* yp = y + g * D + off;
    const float* gp = dy + g * D + off;
    float* dp = dt + g * D + off;
    float dot = 0.f;
    for (int c = 0; c < C; ++c) dot += gp[c] * yp[c];
    for (int c = 0; c < C; ++c) dp[c] = yp[c] * (gp[c] - dot);
  }
}

__global__ void head_sigmoid_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, long long P, int D,
                                        int nsig, float* __restrict__ dt) {
  const long long n = P * nsig;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long p = i / nsig;
    const int k = (int)(i - p * nsig);
    const float yv = y[p * D + k];
    dt[p * D + k] = dy[p * D + k] * yv * (1.f - yv);
  }
}

// d anchors (v4 trainable Anchor layer, yolov4/models/backbone.py:40-60): y_wh = exp(t)*anchor
// => dL/danchor = sum dy * exp(t) = sum dy * y / anchor
__global__ void head_danchor_kernel(const float* __restrict__ y, const float* __restrict__ dy, long long P, int A,
                                    int C, const float* __restrict__ anchors, float* __restrict__ danchors) {
  const int D = 5 + C;
  const int slot = blockIdx.y;  // a*2 + j
  const int a = slot >> 1, j = slot & 1;
  double s = 0.0;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
    const long long i = (p * A + a) * D + 2 + j;
    s += (double)dy[i] * (double)y[i];
  }
  s = wave_reduce_sum(s);
  if ((threadIdx.x & 63) == 0) atomicAdd(&danchors[slot], (float)(s / (double)anchors[slot]));
}

// ---- optimizers ------------------------------------------------------------------------------
// hyper != nullptr: lr_t, beta1, beta2, eps, grad_scale come from device memory (5 floats) instead of the kernel
// arguments -- the form a captured hipGraph replays with a new bias-corrected learning rate every step
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n,
                                                   float lr_t, float b1, float b2, float eps, float gs, int zero,
                                                   const float* __restrict__ hyper) {
  if (hyper != nullptr) {
    lr_t = hyper[0];
    b1 = hyper[1];
    b2 = hyper[2];
    eps = hyper[3];
    gs = hyper[4];
  }
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    f32x4 gv = reinterpret_cast<f32x4*>(g)[i];
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = gv[e] * gs;
      mv[e] = b1 * mv[e] + (1.f - b1) * gg;
      vv[e] = b2 * vv[e] + (1.f - b2) * gg * gg;
      pv[e] -= lr_t * mv[e] / (sqrtf(vv[e]) + eps);
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    if (zero) reinterpret_cast<f32x4*>(g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float gg = g[i] * gs;
    const float mm = b1 * m[i] + (1.f - b1) * gg;
    const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
    m[i] = mm;
    v[i] = vv;
    p[i] -= lr_t * mm / (sqrtf(vv) + eps);
    if (zero) g[i] = 0.f;
  }
}

__global__ void sgd_kernel(float* __restrict__ p, float* __restrict__ g, long long n, float lr, float gs, int zero) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    p[i] -= lr * gs * g[i];
    if (zero) g[i] = 0.f;
  }
}

}  // namespace yolo

using namespace yolo;

extern "C" int yolo_copy_channels_in(const float* src, long long P, int Csrc, float* dst, int Cdst, int c_off,
                                     void* stream) {
  YOLO_REQUIRE(src && dst && P > 0 && Csrc > 0 && c_off >= 0 && c_off + Csrc <= Cdst, "copy_channels_in: bad args");
  const int vec = (Csrc % 4 == 0 && Cdst % 4 == 0 && c_off % 4 == 0) ? 1 : 0;
  const long long n = vec ? P * (Csrc / 4) : P * Csrc;
  hipLaunchKernelGGL(copy_in_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), src, P, Csrc, dst,
                     Cdst, c_off, vec);
  return check_launch("copy_in_kernel");
}

extern "C" int yolo_copy_channels_out(const float* src, long long P, int Csrc, int c_off, float* dst, int Cdst,
                                      int accumulate, void* stream) {
  YOLO_REQUIRE(src && dst && P > 0 && Cdst > 0 && c_off >= 0 && c_off + Cdst <= Csrc, "copy_channels_out: bad args");
  const int vec = (Csrc % 4 == 0 && Cdst % 4 == 0 && c_off % 4 == 0) ? 1 : 0;
  const long long n = vec ? P * (Cdst / 4) : P * Cdst;
  hipLaunchKernelGGL(copy_out_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), src, P, Csrc, c_off,
                     dst, Cdst, accumulate, vec);
  return check_launch("copy_out_kernel");
}

extern "C" int yolo_upsample2x_fwd(const float* x, int N, int H, int W, int C, float* y, int Cy, int c_off,
                                   void* stream) {
  YOLO_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && c_off >= 0 && c_off + C <= Cy, "upsample_fwd: bad args");
  YOLO_REQUIRE(C % 4 == 0 && Cy % 4 == 0 && c_off % 4 == 0, "upsample_fwd: channels must be multiples of 4");
  const long long n = (long long)N * 4 * H * W * (C / 4);
  hipLaunchKernelGGL(upsample_fwd_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), x, N, H, W,
                     C / 4, y, Cy, c_off);
  return check_launch("upsample_fwd_kernel");
}

extern "C" int yolo_upsample2x_bwd(const float* dy, int N, int H, int W, int C, int Cy, int c_off, float* dx,
                                   int accumulate, void* stream) {
  YOLO_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && c_off >= 0 && c_off + C <= Cy, "upsample_bwd: bad args");
  YOLO_REQUIRE(C % 4 == 0 && Cy % 4 == 0 && c_off % 4 == 0, "upsample_bwd: channels must be multiples of 4");
  const long long n = (long long)N * H * W * (C / 4);
  hipLaunchKernelGGL(upsample_bwd_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), dy, N, H, W,
                     C / 4, Cy, c_off, dx, accumulate);
  return check_launch("upsample_bwd_kernel");
}

extern "C" int yolo_axpy(float* a, const float* b, long long n, void* stream) {
  YOLO_REQUIRE(a && b && n > 0, "axpy: bad args");
  hipLaunchKernelGGL(axpy_kernel, dim3(stream_grid(n / 4 + 1, 256)), dim3(256), 0, as_stream(stream), a, b, n);
  return check_launch("axpy_kernel");
}

extern "C" int yolo_fill(float* p, long long n, float value, void* stream) {
  YOLO_REQUIRE(p && n > 0, "fill: bad args");
  hipLaunchKernelGGL(fill_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), p, n, value);
  return check_launch("fill_kernel");
}

extern "C" int yolo_maxpool_fwd(const float* x, int N, int H, int W, int C, int k, int s, int pad_t, int pad_l,
                                int Ho, int Wo, float* y, int Cy, int c_off, int* argmax, void* stream) {
  YOLO_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && k > 0 && s > 0 && Ho > 0 && Wo > 0, "maxpool_fwd: bad args");
  YOLO_REQUIRE(c_off >= 0 && c_off + C <= Cy, "maxpool_fwd: bad channel slice");
  YOLO_REQUIRE((long long)N * H * W * C < (1LL << 31), "maxpool_fwd: tensor too large for int32 argmax");
  if (pool_plane_ok(H, W, C, k, s, Ho, Wo)) {
    const size_t lds = (size_t)H * W * POOL_CG * 10;
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&maxpool_plane_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 1024 * POOL_CG * 10);
      attr = true;
    }
    hipLaunchKernelGGL(maxpool_plane_fwd_kernel, dim3((unsigned)(N * (C / POOL_CG))), dim3(256), lds, as_stream(stream), x, H, W,
                       C, k, pad_t, pad_l, y, Cy, c_off, argmax);
    return check_launch("maxpool_plane_fwd_kernel");
  }
  const long long n = (long long)N * Ho * Wo * C;
  if (k == 2 && s == 2 && pad_t == 0 && pad_l == 0 && H == 2 * Ho && W == 2 * Wo && (C & 3) == 0 && Cy == C && c_off == 0 &&
      (reinterpret_cast<size_t>(x) & 15) == 0 && (reinterpret_cast<size_t>(y) & 15) == 0 &&
      (argmax == nullptr || (reinterpret_cast<size_t>(argmax) & 15) == 0)) {
    hipLaunchKernelGGL(maxpool2x2_fwd_kernel, dim3(stream_grid(n / 4, 256)), dim3(256), 0, as_stream(stream), x, n / 4, Ho, Wo, C,
                       y, argmax);
    return check_launch("maxpool2x2_fwd_kernel");
  }
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), x, N, H, W, C, k,
                     s, pad_t, pad_l, Ho, Wo, y, Cy, c_off, argmax);
  return check_launch("maxpool_fwd_kernel");
}

extern "C" int yolo_maxpool2x2_bwd(const float* dy, int N, int Ho, int Wo, int C, const int* argmax, float* dx, int accumulate,
                                   void* stream) {
  YOLO_REQUIRE(dy && argmax && dx && N > 0 && Ho > 0 && Wo > 0 && C > 0 && (C & 3) == 0, "maxpool2x2_bwd: bad args (C %% 4 == 0)");
  YOLO_REQUIRE(((reinterpret_cast<size_t>(dy) | reinterpret_cast<size_t>(argmax) | reinterpret_cast<size_t>(dx)) & 15) == 0,
               "maxpool2x2_bwd: 16-byte aligned tensors");
  YOLO_REQUIRE((long long)N * 4 * Ho * Wo * C < (1LL << 31), "maxpool2x2_bwd: tensor too large for int32 argmax");
  const long long n4 = (long long)N * Ho * Wo * (C / 4);
  if (accumulate)
    hipLaunchKernelGGL(maxpool2x2_bwd_kernel<true>, dim3(stream_grid(n4, 256)), dim3(256), 0, as_stream(stream), dy, n4, Ho, Wo, C,
                       argmax, dx);
  else
    hipLaunchKernelGGL(maxpool2x2_bwd_kernel<false>, dim3(stream_grid(n4, 256)), dim3(256), 0, as_stream(stream), dy, n4, Ho, Wo, C,
                       argmax, dx);
  return check_launch("maxpool2x2_bwd_kernel");
}

extern "C" int yolo_maxpool_bwd_same(const float* dy, int N, int H, int W, int C, int Cy, int c_off, const int* argmax, int k,
                                     int pad_t, int pad_l, float* dx, void* stream) {
  YOLO_REQUIRE(dy && argmax && dx && N > 0 && H > 0 && W > 0 && C > 0 && k > 0, "maxpool_bwd_same: bad args");
  YOLO_REQUIRE(c_off >= 0 && c_off + C <= Cy, "maxpool_bwd_same: bad channel slice");
  if (pool_plane_ok(H, W, C, k, 1, H, W)) {
    const size_t lds = (size_t)H * W * POOL_CG * 6;
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&maxpool_plane_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 1024 * POOL_CG * 6);
      attr = true;
    }
    hipLaunchKernelGGL(maxpool_plane_bwd_kernel, dim3((unsigned)(N * (C / POOL_CG))), dim3(256), lds, as_stream(stream), dy, H, W,
                       C, Cy, c_off, argmax, k, pad_t, pad_l, dx);
    return check_launch("maxpool_plane_bwd_kernel");
  }
  return yolo_maxpool_bwd(dy, N, H, W, C, Cy, c_off, argmax, dx, stream);
}

extern "C" int yolo_maxpool_bwd(const float* dy, int N, int Ho, int Wo, int C, int Cy, int c_off, const int* argmax,
                                float* dx, void* stream) {
  YOLO_REQUIRE(dy && argmax && dx && N > 0 && Ho > 0 && Wo > 0 && C > 0, "maxpool_bwd: bad args");
  const long long n = (long long)N * Ho * Wo * C;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), dy, n, C, Cy,
                     c_off, argmax, dx);
  return check_launch("maxpool_bwd_kernel");
}

extern "C" int yolo_space_to_depth2_fwd(const float* x, int N, int H, int W, int C, float* y, int Cy, int c_off,
                                        void* stream) {
  YOLO_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && (H % 2 == 0) && (W % 2 == 0), "space_to_depth: bad args");
  YOLO_REQUIRE(c_off >= 0 && c_off + 4 * C <= Cy, "space_to_depth: bad channel slice");
  const long long n = (long long)N * H * W * C;
  hipLaunchKernelGGL(s2d_fwd_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), x, N, H, W, C, y, Cy,
                     c_off);
  return check_launch("s2d_fwd_kernel");
}

extern "C" int yolo_space_to_depth2_bwd(const float* dy, int N, int H, int W, int C, int Cy, int c_off, float* dx,
                                        int accumulate, void* stream) {
  YOLO_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && (H % 2 == 0) && (W % 2 == 0), "space_to_depth_bwd: bad args");
  YOLO_REQUIRE(c_off >= 0 && c_off + 4 * C <= Cy, "space_to_depth_bwd: bad channel slice");
  const long long n = (long long)N * H * W * C;
  hipLaunchKernelGGL(s2d_bwd_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), dy, N, H, W, C, Cy,
                     c_off, dx, accumulate);
  return check_launch("s2d_bwd_kernel");
}

extern "C" int yolo_head_act_fwd(const float* t, long long P, int A, int C, int version, const float* anchors,
                                 float* y, void* stream) {
  YOLO_REQUIRE(t && y && P > 0 && A > 0 && C > 0, "head_act_fwd: bad args");
  hipStream_t st = as_stream(stream);
  if (version == YOLO_HEAD_V3 || version == YOLO_HEAD_V4 || version == YOLO_HEAD_V2) {
    YOLO_REQUIRE(anchors != nullptr, "head_act_fwd: anchors required");
    const long long n = P * A * (5 + C);
    hipLaunchKernelGGL(head_fwd_pointwise_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, st, t, P, A, C, anchors, y);
    if (int rc = check_launch("head_fwd_pointwise_kernel")) return rc;
    if (version == YOLO_HEAD_V2) {
      hipLaunchKernelGGL(head_softmax_fwd_kernel, dim3(stream_grid(P * A, 256)), dim3(256), 0, st, t, P * A, 5 + C, 5, C,
                         y);
      return check_launch("head_softmax_fwd_kernel");
    }
    return YOLO_OK;
  }
  if (version == YOLO_HEAD_V1) {
    const int D = 5 * A + C;
    hipLaunchKernelGGL(head_sigmoid_fwd_kernel, dim3(stream_grid(P * 5 * A, 256)), dim3(256), 0, st, t, P, D, 5 * A, y);
    if (int rc = check_launch("head_sigmoid_fwd_kernel")) return rc;
    hipLaunchKernelGGL(head_softmax_fwd_kernel, dim3(stream_grid(P, 256)), dim3(256), 0, st, t, P, D, 5 * A, C, y);
    return check_launch("head_softmax_fwd_kernel");
  }
  set_error("head_act_fwd: bad version %d", version);
  return YOLO_ERR_INVALID_ARG;
}

extern "C" int yolo_head_act_bwd(const float* y, const float* dy, long long P, int A, int C, int version,
                                 const float* anchors, float* dt, float* danchors, void* stream) {
  YOLO_REQUIRE(y && dy && dt && P > 0 && A > 0 && C > 0, "head_act_bwd: bad args");
  hipStream_t st = as_stream(stream);
  if (version == YOLO_HEAD_V3 || version == YOLO_HEAD_V4 || version == YOLO_HEAD_V2) {
    const long long n = P * A * (5 + C);
    hipLaunchKernelGGL(head_bwd_pointwise_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, st, y, dy, P, A, C,
                       version == YOLO_HEAD_V2 ? 0 : 1, dt);
    if (int rc = check_launch("head_bwd_pointwise_kernel")) return rc;
    if (version == YOLO_HEAD_V2) {
      hipLaunchKernelGGL(head_softmax_bwd_kernel, dim3(stream_grid(P * A, 256)), dim3(256), 0, st, y, dy, P * A, 5 + C, 5,
                         C, dt);
      if (int rc = check_launch("head_softmax_bwd_kernel")) return rc;
    }
    if (danchors != nullptr) {
      YOLO_REQUIRE(anchors != nullptr, "head_act_bwd: anchors required for danchors");
      long long gx = (P + 255) / 256;
      if (gx > 256) gx = 256;
      hipLaunchKernelGGL(head_danchor_kernel, dim3((unsigned)gx, 2 * A), dim3(256), 0, st, y, dy, P, A, C, anchors,
                         danchors);
      return check_launch("head_danchor_kernel");
    }
    return YOLO_OK;
  }
  if (version == YOLO_HEAD_V1) {
    const int D = 5 * A + C;
    hipLaunchKernelGGL(head_sigmoid_bwd_kernel, dim3(stream_grid(P * 5 * A, 256)), dim3(256), 0, st, y, dy, P, D, 5 * A,
                       dt);
    if (int rc = check_launch("head_sigmoid_bwd_kernel")) return rc;
    hipLaunchKernelGGL(head_softmax_bwd_kernel, dim3(stream_grid(P, 256)), dim3(256), 0, st, y, dy, P, D, 5 * A, C, dt);
    return check_launch("head_softmax_bwd_kernel");
  }
  set_error("head_act_bwd: bad version %d", version);
  return YOLO_ERR_INVALID_ARG;
}

extern "C" int yolo_adam_step(float* p, float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                              float eps, int step, float grad_scale, int zero_grad, void* stream) {
  YOLO_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam_step: bad args");
  // Keras Adam: lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t); p -= lr_t * m / (sqrt(v) + eps)
  const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step));
  hipLaunchKernelGGL(adam_kernel, dim3(stream_grid(n / 4 + 1, 256)), dim3(256), 0, as_stream(stream), p, g, m, v, n,
                     (float)lr_t, beta1, beta2, eps, grad_scale, zero_grad, (const float*)nullptr);
  return check_launch("adam_kernel");
}

extern "C" float yolo_adam_lr_t(float lr, float beta1, float beta2, int step) {
  return (float)((double)lr * sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step)));
}

extern "C" int yolo_adam_step_dev(float* p, float* g, float* m, float* v, long long n, const float* hyper, int zero_grad,
                                  void* stream) {
  YOLO_REQUIRE(p && g && m && v && hyper && n > 0, "adam_step_dev: bad args");
  hipLaunchKernelGGL(adam_kernel, dim3(stream_grid(n / 4 + 1, 256)), dim3(256), 0, as_stream(stream), p, g, m, v, n, 0.f,
                     0.f, 0.f, 0.f, 0.f, zero_grad, hyper);
  return check_launch("adam_kernel");
}

extern "C" int yolo_sgd_step(float* p, float* g, long long n, float lr, float grad_scale, int zero_grad,
                             void* stream) {
  YOLO_REQUIRE(p && g && n > 0, "sgd_step: bad args");
  hipLaunchKernelGGL(sgd_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), p, g, n, lr, grad_scale,
                     zero_grad);
  return check_launch("sgd_kernel");
}

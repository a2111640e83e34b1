// Label tensors on the device: the box -> grid encoder of the reference's data
// sequences (utils/tools.py:179-209, identical in read_file_to_dataset) and the 2x label pyramid
// (utils/tools.py:342-367, used by yolov3/__init__.py:41-53). Everything is computed in float64 exactly as the
// reference's NumPy / Python-float code does (same operations, same order), so the results are bit-identical to
// its float64 arrays; a float32 copy (what Keras feeds the loss) is written beside it.
#include "common.hpp"

namespace yolo {

// CPython / NumPy float divmod (floatobject.c float_divmod, npy_divmod): returns x // y, *mod = x % y.
// (floor(x / y) and fmod alone differ from it in the last bit in rare cases; the cell index and the in-cell offset
// must come out exactly as the reference's `//` and `%` give them.)
__device__ __forceinline__ double py_divmod(double vx, double wx, double* mod_out) {
  double mod = fmod(vx, wx);
  double div = (vx - mod) / wx;
  if (mod != 0.0) {
    if ((wx < 0) != (mod < 0)) {
      mod += wx;
      div -= 1.0;
    }
  } else {
    mod = copysign(0.0, wx);
  }
  double floordiv;
  if (div != 0.0) {
    floordiv = floor(div);
    if (div - floordiv > 0.5) floordiv += 1.0;
  } else {
    floordiv = copysign(0.0, vx / wx);
  }
  *mod_out = mod;
  return floordiv;
}

// One workgroup per image: clear the image's label block, then ONE thread walks the image's boxes in order --
// "last writer wins" for x, y, w, h (the reference overwrites), class bits accumulate (it never clears them).
__global__ __launch_bounds__(256) void encode_labels_kernel(const double* __restrict__ boxes, const int* __restrict__ cls,
                                                           const int* __restrict__ first, double img_h, double img_w,
                                                           int gh, int gw, int C, double* __restrict__ label64,
                                                           float* __restrict__ label32) {
  const int n = blockIdx.x;
  const int ch = 5 + C;
  const long long cells = (long long)gh * gw * ch;
  double* L = label64 + (long long)n * cells;
  for (long long i = threadIdx.x; i < cells; i += blockDim.x) L[i] = 0.0;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double cell_h = img_h / gh, cell_w = img_w / gw;     // grid_height, grid_width (:183-184)
    for (int b = first[n]; b < first[n + 1]; ++b) {
      const double x1 = boxes[4 * b], y1 = boxes[4 * b + 1], x2 = boxes[4 * b + 2], y2 = boxes[4 * b + 3];
      const double bx = x1 + (x2 - x1) / 2, by = y1 + (y2 - y1) / 2, bw = x2 - x1, bh = y2 - y1;   // :189-192
      double mx, my;
      const double fx = py_divmod(bx, cell_w, &mx), fy = py_divmod(by, cell_h, &my);               // :194-195
      long long xi = (long long)fx, yi = (long long)fy;
      if (xi < gw && yi < gh) {                                                                     // :197
        // a negative index wraps like NumPy's (an augmented box can start left of / above the image);
        // below -g the reference raises IndexError: skipped here
        if (xi < 0) xi += gw;
        if (yi < 0) yi += gh;
        if (xi < 0 || yi < 0) continue;
        double* c = L + ((long long)yi * gw + xi) * ch;
        c[0] = mx / cell_w;                                                                          // :198-199
        c[1] = my / cell_h;                                                                          // :200-201
        c[2] = bw / img_w;                                                                           // :202-203
        c[3] = bh / img_h;                                                                           // :204-205
        c[4] = 1.0;                                                                                  // :206
        const int k = cls[b];
        if (k >= 0 && k < C) c[5 + k] = 1.0;                                                         // :207
      }
    }
  }
  if (label32 != nullptr) {
    __syncthreads();
    float* F = label32 + (long long)n * cells;
    for (long long i = threadIdx.x; i < cells; i += blockDim.x) F[i] = (float)L[i];
  }
}

// one thread per (output cell, channel quad); the 2x2 block is examined by every thread of the cell
__global__ __launch_bounds__(256) void down2xlabel_kernel(const double* __restrict__ in, int N, int gh, int gw, int ch,
                                                         double* __restrict__ out, float* __restrict__ out32) {
  const int h2 = gh / 2, w2 = gw / 2;
  const long long total = (long long)N * h2 * w2 * ch;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(t % ch);
    long long cell = t / ch;
    const int j = (int)(cell % w2);
    cell /= w2;
    const int i = (int)(cell % h2);
    const int n = (int)(cell / h2);
    const double* base = in + (((long long)n * gh + 2 * i) * gw + 2 * j) * ch;
    // crop[..., 4].max() == 1 and (crop[..., 2] * crop[..., 3]).argmax(): first maximum in (dy, dx) order (:358-359)
    double cmax = -1.0, best = 0.0;
    int id = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double* p = base + ((long long)(q >> 1) * gw + (q & 1)) * ch;
      cmax = fmax(cmax, p[4]);
      const double area = p[2] * p[3];
      if (q == 0 || area > best) {
        best = area;
        id = q;
      }
    }
    double v = 0.0;
    if (cmax == 1.0) {
      const double* p = base + ((long long)(id >> 1) * gw + (id & 1)) * ch;
      v = p[c];
      if (c == 0) v = (v + (double)(id & 1)) / 2;          // (crop_xy + [max_id % 2, max_id // 2]) / 2   (:361-362)
      else if (c == 1) v = (v + (double)(id >> 1)) / 2;
    }
    out[t] = v;
    if (out32 != nullptr) out32[t] = (float)v;
  }
}

}  // namespace yolo

using namespace yolo;

extern "C" int yolo_encode_labels(const double* boxes, const int* cls, const int* first, int N, double img_h, double img_w,
                                  int gh, int gw, int C, double* label64, float* label32, void* stream) {
  YOLO_REQUIRE(boxes && cls && first && label64, "encode_labels: null pointer");
  YOLO_REQUIRE(N > 0 && gh > 0 && gw > 0 && C >= 0 && img_h > 0 && img_w > 0, "encode_labels: bad sizes");
  hipLaunchKernelGGL(encode_labels_kernel, dim3((unsigned)N), dim3(256), 0, as_stream(stream), boxes, cls, first, img_h,
                     img_w, gh, gw, C, label64, label32);
  return check_launch("encode_labels_kernel");
}

extern "C" int yolo_down2xlabel(const double* label_in, int N, int gh, int gw, int ch, double* label_out, float* out32,
                                void* stream) {
  YOLO_REQUIRE(label_in && label_out, "down2xlabel: null pointer");
  YOLO_REQUIRE(N > 0 && gh >= 2 && gw >= 2 && ch >= 5, "down2xlabel: bad sizes");
  const long long total = (long long)N * (gh / 2) * (gw / 2) * ch;
  hipLaunchKernelGGL(down2xlabel_kernel, dim3((unsigned)stream_grid(total, 256)), dim3(256), 0, as_stream(stream), label_in,
                     N, gh, gw, ch, label_out, out32);
  return check_launch("down2xlabel_kernel");
}

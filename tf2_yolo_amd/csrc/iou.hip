// cal_iou as a stand-alone broadcasting kernel (the two public `cal_iou` functions of the reference):
//   utils/tools.py:630-684            cal_iou(xywh_true, xywh_pred, mode): NumPy, IoU / DIoU, no grid divisor
//   yolov3/losses/loss.py:9-37        cal_iou(xywh_true, xywh_pred, grid_shape): xy / grid, IoU (same text in v1.5 / v2)
//   yolov4/losses/loss.py:10-61       ... return_ciou=True: (IoU, CIoU)
// One thread per output element; the operands are addressed through per-dimension element strides (0 on a
// broadcast dimension), so `boxes[:, None]` against `boxes[None, :]` costs no materialised copy. The arithmetic
// keeps the reference's operation order in the operand type T and is compiled WITHOUT fused multiply-adds
// (NumPy / TF round every product): the float64 results equal utils.tools.cal_iou's bit for bit
// (tests/golden/tools_golden.npz: iou_mat, diou_mat). Bound: HBM (32 B read + 8 B written per pair at most).
#include "common.hpp"

namespace yolo {

struct IouGeom {
  int ndim;
  long long total;
  long long shape[8];
  long long sa[8], sb[8];   // element strides of the box start, 0 = broadcast
  double gw, gh;            // divisors of x / y (1 = none)
  int mode;                 // 1 IoU, 2 DIoU, 3 IoU + CIoU
};

template <typename T>
__device__ __forceinline__ T atan_t(T x);
template <>
__device__ __forceinline__ float atan_t<float>(float x) { return atanf(x); }
template <>
__device__ __forceinline__ double atan_t<double>(double x) { return atan(x); }

template <typename T>
__global__ __launch_bounds__(256) void cal_iou_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out,
                                                      T* __restrict__ out2, IouGeom g) {
#pragma clang fp contract(off)
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < g.total; e += (long long)gridDim.x * blockDim.x) {
    long long r = e, oa = 0, ob = 0;
    for (int d = g.ndim - 1; d >= 0; --d) {
      const long long i = r % g.shape[d];
      r /= g.shape[d];
      oa += i * g.sa[d];
      ob += i * g.sb[d];
    }
    const T* t = a + oa;   // "true"
    const T* p = b + ob;   // "pred"
    const T gw = (T)g.gw, gh = (T)g.gh;
    const T tx = t[0] / gw, ty = t[1] / gh, tw = t[2], th = t[3];
    const T px = p[0] / gw, py = p[1] / gh, pw = p[2], ph = p[3];
    const T thx = tw / (T)2, thy = th / (T)2, phx = pw / (T)2, phy = ph / (T)2;
    const T tminx = tx - thx, tmaxx = tx + thx, tminy = ty - thy, tmaxy = ty + thy;
    const T pminx = px - phx, pmaxx = px + phx, pminy = py - phy, pmaxy = py + phy;
    // np.maximum / tf.maximum(pred, true): NaNs aside the order does not matter
    const T iw = fmax(fmin(pmaxx, tmaxx) - fmax(pminx, tminx), (T)0);
    const T ih = fmax(fmin(pmaxy, tmaxy) - fmax(pminy, tminy), (T)0);
    const T inter = iw * ih;
    const T ta = tw * th, pa = pw * ph;
    const T uni = pa + ta - inter;
    const T iou = inter / (uni + (T)1e-07);
    if (g.mode == 1) {
      out[e] = iou;
      continue;
    }
    const T ewx = fmax(pmaxx, tmaxx) - fmin(pminx, tminx);
    const T ewy = fmax(pmaxy, tmaxy) - fmin(pminy, tminy);
    const T c2 = ewx * ewx + ewy * ewy;
    const T dx = tx - px, dy = ty - py;
    const T rho2 = dx * dx + dy * dy;
    if (g.mode == 2) {
      out[e] = iou - rho2 / c2;
      continue;
    }
    const T at = atan_t<T>(tw / (th + (T)1e-07));
    const T ap = atan_t<T>(pw / (ph + (T)1e-07));
    const T d = at - ap;
    const T v = (T)(4.0 / (3.141592653589793 * 3.141592653589793)) * (d * d);
    const T alpha = v / ((T)1 - iou + v);
    out[e] = iou;
    out2[e] = iou - rho2 / c2 - alpha * v;
  }
}

}  // namespace yolo

extern "C" int yolo_cal_iou(const void* xywh_true, const void* xywh_pred, void* out, void* out2, int is_f64, int mode, int ndim,
                            const long long* shape_host, const long long* true_strides_host,
                            const long long* pred_strides_host, double grid_w, double grid_h, void* stream) {
  using namespace yolo;
  YOLO_REQUIRE(xywh_true && xywh_pred && out, "yolo_cal_iou: null pointer");
  YOLO_REQUIRE(mode >= 1 && mode <= 3, "yolo_cal_iou: mode %d (1 IoU, 2 DIoU, 3 IoU + CIoU)", mode);
  YOLO_REQUIRE(mode != 3 || out2, "yolo_cal_iou: mode 3 needs the second output");
  YOLO_REQUIRE(ndim >= 0 && ndim <= 8, "yolo_cal_iou: %d dimensions (at most 8)", ndim);
  YOLO_REQUIRE(grid_w != 0 && grid_h != 0, "yolo_cal_iou: zero grid divisor");
  IouGeom g;
  g.ndim = ndim;
  g.total = 1;
  for (int d = 0; d < 8; ++d) {
    g.shape[d] = d < ndim ? shape_host[d] : 1;
    g.sa[d] = d < ndim ? true_strides_host[d] : 0;
    g.sb[d] = d < ndim ? pred_strides_host[d] : 0;
    YOLO_REQUIRE(g.shape[d] >= 0, "yolo_cal_iou: negative extent");
    g.total *= g.shape[d];
  }
  g.gw = grid_w;
  g.gh = grid_h;
  g.mode = mode;
  if (g.total == 0) return YOLO_OK;
  const int grid = stream_grid(g.total, 256);
  if (is_f64)
    cal_iou_kernel<double><<<grid, 256, 0, as_stream(stream)>>>((const double*)xywh_true, (const double*)xywh_pred, (double*)out,
                                                               (double*)out2, g);
  else
    cal_iou_kernel<float><<<grid, 256, 0, as_stream(stream)>>>((const float*)xywh_true, (const float*)xywh_pred, (float*)out,
                                                              (float*)out2, g);
  return check_launch("yolo_cal_iou");
}

// Activations shared by the BN/activation kernels (bn_act.hip) and the fused inference epilogue of the planes
// convolutions (planes_epilogue.hpp): LeakyReLU(0.1) and Mish, forward and derivative.
#pragma once
#include "common.hpp"

namespace yolo {

// Mish(z) = z * tanh(softplus(z)) (yolov4/models/backbone.py:22-37). With e = exp(-|z|), exactly:
//   z > 0:  tanh(softplus(z)) = (1 + 2e) / (1 + 2e + 2e^2),        1 - tanh = 2e^2 / (1 + 2e + 2e^2)
//   z <= 0: tanh(softplus(z)) = (e^2 + 2e) / (e^2 + 2e + 2),       1 - tanh = 2 / (e^2 + 2e + 2)
// (from tanh(log(1 + u)) = ((1+u)^2 - 1) / ((1+u)^2 + 1), u = e^z): one exp and one division instead of
// log1p + exp + tanh, all terms positive (no cancellation for large |z|, where 1 - tanh^2 would lose every bit).
struct MishParts {
  float t, omt, e;   // tanh(softplus(z)), 1 - t, exp(-|z|)
};
// The exponential and the reciprocals are the hardware ones (v_exp_f32 on x log2(e), v_rcp_f32: 1 ulp each): with libm's
// expf and two IEEE divisions the Mish passes of YOLOv4 were VALU-bound (~45 VALU operations per element where a
// streaming pass has room for ~25: bn_bwd_reduce 52 us per launch where LeakyReLU needs 30). e = exp(-|z|) <= 1 enters
// only sums of positive terms, so a relative error of a few ulp in it stays a few ulp in t and 1 - t.
__device__ __forceinline__ float fast_exp_neg(float x) {   // exp(x) for x <= 0
  return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
}
__device__ __forceinline__ MishParts mish_parts(float z) {
  const float e = fast_exp_neg(-fabsf(z));
  const bool pos = z > 0.f;
  const float e2 = e * e;
  const float a = pos ? fmaf(2.f, e, 1.f) : fmaf(2.f, e, e2);   // numerator of t
  const float b = pos ? 2.f * e2 : 2.f;                          // numerator of 1 - t
  const float r = __builtin_amdgcn_rcpf(a + b);
  return MishParts{a * r, b * r, e};
}
__device__ __forceinline__ float act_fwd(float z, int act) {
  if (act == YOLO_ACT_LEAKY) return z > 0.f ? z : 0.1f * z;
  if (act == YOLO_ACT_MISH) return z * mish_parts(z).t;
  return z;
}
__device__ __forceinline__ float act_grad(float z, int act) {
  if (act == YOLO_ACT_LEAKY) return z > 0.f ? 1.f : 0.1f;
  if (act == YOLO_ACT_MISH) {
    // d/dz [z t] = t + z (1 - t)(1 + t) sigmoid(z) as ONE rational function of e = exp(-|z|) (t = N / D as in mish_parts):
    //   z > 0:  N = 1 + 2e,    D = N + 2e^2,  (1 - t)(1 + t) sigmoid = 4 e^2 (1 + e) / D^2
    //   z <= 0: N = e^2 + 2e,  D = N + 2,     (1 - t)(1 + t) sigmoid = 4 e   (1 + e) / D^2
    //   mish'(z) = (N D + 4 z w) / D^2,  w = e^2 (1 + e) or e (1 + e)
    // -- one exponential, one reciprocal, 15 multiply-adds and selects (the form t + z omt (1 + t) sg needed two
    // reciprocals and ~22; these passes are VALU-bound with Mish). The cancellation near the zero of mish' (z = -1.19) is
    // the function's own: both forms subtract two terms of size ~0.3 there.
    const float e = fast_exp_neg(-fabsf(z));
    const bool pos = z > 0.f;
    const float e2 = e * e;
    const float n = fmaf(2.f, e, pos ? 1.f : e2);
    const float d = n + (pos ? 2.f * e2 : 2.f);
    const float w = (pos ? e2 : e) * (1.f + e);
    const float r = __builtin_amdgcn_rcpf(d);
    return fmaf(4.f * z, w, n * d) * r * r;
  }
  return 1.f;
}

}  // namespace yolo

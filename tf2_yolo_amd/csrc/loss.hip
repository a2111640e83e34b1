// Anchor-grid YOLO losses (v1.5, v2, v3, v4): forward and dL/dy_pred in ONE pass.
//
// Follows, op for op in fp32, the reference closures
//   yolov3/losses/loss.py:9-37 (cal_iou), :51-162 (yolo_loss)
//   yolov2/losses/loss.py:47-135
//   yolov4/losses/loss.py:10-61 (cal_iou + CIoU), :75-167
//   yolov1_5/losses/loss.py:46-116
// with the TensorFlow autodiff conventions of SURVEY.md Appendix B (maximum/minimum ties send
// the gradient to the first operand, clip_by_value passes gradient inside [lo,hi], masks are
// constants). The per-cell arithmetic is entirely cell-local, so the kernel
// is one wavefront per cell: lanes 0..A-1 evaluate the per-anchor IoU (and, for v1.5/v4, its
// forward-mode partials w.r.t. the four box coordinates), a shuffle broadcast picks the
// responsible anchor, then the 64 lanes sweep the cell's A*(5+C) prediction channels with
// coalesced loads/stores, accumulate the loss parts in fp64 and write the gradient row.
// HBM-bound: reads y_true + y_pred once, writes the gradient once.
#include "common.hpp"
#include "conv_args.hpp"
#include <cfloat>
#include <cstdlib>

namespace yolo {

constexpr float LEPS = 1e-7f;
constexpr float ONE_M_EPS = (float)(1.0 - 1e-7);

// value + partial derivatives w.r.t. the raw prediction (px, py, pw, ph)
struct D4 {
  float v;
  float d[4];
};
__device__ __forceinline__ D4 dconst(float v) { return D4{v, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ D4 dvar(float v, int i) {
  D4 r = dconst(v);
  r.d[i] = 1.f;
  return r;
}
__device__ __forceinline__ D4 operator+(const D4& a, const D4& b) {
  D4 r{a.v + b.v, {}};
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] + b.d[i];
  return r;
}
__device__ __forceinline__ D4 operator-(const D4& a, const D4& b) {
  D4 r{a.v - b.v, {}};
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] - b.d[i];
  return r;
}
__device__ __forceinline__ D4 operator*(const D4& a, const D4& b) {
  D4 r{a.v * b.v, {}};
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
__device__ __forceinline__ D4 operator/(const D4& a, const D4& b) {
  D4 r{a.v / b.v, {}};
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] / b.v - (a.v / b.v) / b.v * b.d[i];
  return r;
}
__device__ __forceinline__ D4 dmax(const D4& x, const D4& y) { return (x.v >= y.v) ? x : y; }  // tie -> x
__device__ __forceinline__ D4 dmin(const D4& x, const D4& y) { return (x.v <= y.v) ? x : y; }  // tie -> x
__device__ __forceinline__ D4 dsq(const D4& a) {
  D4 r{a.v * a.v, {}};
  for (int i = 0; i < 4; ++i) r.d[i] = 2.f * a.v * a.d[i];
  return r;
}
__device__ __forceinline__ D4 datan(const D4& a) {
  D4 r{atanf(a.v), {}};
  const float g = 1.f / (1.f + a.v * a.v);
  for (int i = 0; i < 4; ++i) r.d[i] = g * a.d[i];
  return r;
}

struct IouOut {
  D4 iou;
  D4 ciou;
};

// cal_iou with pred box P (raw px,py in cell units, pw,ph normalised) vs the cell's truth T.
template <bool CIOU>
__device__ __forceinline__ IouOut iou_dual(const float* T, const float* P, float gw, float gh) {
  const D4 px = dvar(P[0], 0), py = dvar(P[1], 1), pw = dvar(P[2], 2), ph = dvar(P[3], 3);
  const D4 cx = px / dconst(gw), cy = py / dconst(gh);
  const float tcx = T[0] / gw, tcy = T[1] / gh, tw = T[2], th = T[3];
  const float thw = tw / 2.f, thh = th / 2.f;
  const float tminx = tcx - thw, tmaxx = tcx + thw, tminy = tcy - thh, tmaxy = tcy + thh;
  const D4 hw = pw / dconst(2.f), hh = ph / dconst(2.f);
  const D4 minx = cx - hw, maxx = cx + hw, miny = cy - hh, maxy = cy + hh;
  const D4 iminx = dmax(minx, dconst(tminx)), iminy = dmax(miny, dconst(tminy));
  const D4 imaxx = dmin(maxx, dconst(tmaxx)), imaxy = dmin(maxy, dconst(tmaxy));
  const D4 iw = dmax(imaxx - iminx, dconst(0.f)), ih = dmax(imaxy - iminy, dconst(0.f));
  const D4 inter = iw * ih;
  const float tarea = tw * th;
  const D4 parea = pw * ph;
  const D4 uni = parea + dconst(tarea) - inter;
  IouOut o;
  o.iou = inter / (uni + dconst(LEPS));
  o.ciou = o.iou;
  if constexpr (CIOU) {
    const D4 eminx = dmin(minx, dconst(tminx)), eminy = dmin(miny, dconst(tminy));
    const D4 emaxx = dmax(maxx, dconst(tmaxx)), emaxy = dmax(maxy, dconst(tmaxy));
    const D4 ewx = emaxx - eminx, ewy = emaxy - eminy;
    const D4 c2 = dsq(ewx) + dsq(ewy);
    const D4 rho2 = dsq(dconst(tcx) - cx) + dsq(dconst(tcy) - cy);
    const float atan_t = atanf(tw / (th + LEPS));
    const D4 atan_p = datan(pw / (ph + dconst(LEPS)));
    const D4 v = dconst((float)(4.0 / (M_PI * M_PI))) * dsq(dconst(atan_t) - atan_p);
    const D4 alpha = v / (dconst(1.f) - o.iou + v);
    o.ciou = o.iou - rho2 / c2 - alpha * v;
  }
  return o;
}

struct LossParams {
  yolo_loss_cfg c;
  long long cells;
  float inv_n;  // grad_scale / N
  int* dec;     // optional [cells][2]: the discrete decisions of every cell (see yolo_loss_fwd_bwd in yolo_hip.h)
};

// One prediction channel of one cell: the loss parts it adds and dL/dp (before the 1/N factor). b = anchor, k = channel inside
// the anchor (k >= 5: class k - 5), p = the prediction, tk = the class target of a class channel, Tb = the cell's true box,
// t4 = its object flag; iou_b / d0..d3 = the IoU of anchor b (and, v1.5 / v4, its partials), an = the anchor extent of a w / h
// channel, cb = anchor b's confidence (v1.5), r_b = 1 for the responsible anchor. Shared by the two loaders of loss_kernel.
template <int VER>
__device__ __forceinline__ float loss_channel(const yolo_loss_cfg& c, const int b, const int k, const float p, const float tk_cur,
                                              const float t4, const float (&Tb)[4], const float iou_b, const float d0,
                                              const float d1, const float d2, const float d3, const float an, const float cb,
                                              const float r_b, double (&part)[7]) {
  (void)b;
  const float tbk = (k == 0) ? Tb[0] : (k == 1) ? Tb[1] : (k == 2) ? Tb[2] : Tb[3];
  float obj, noobj;
  if (VER == 1) {
    obj = t4 * r_b;
    noobj = 1.f - obj;
  } else {
    obj = t4 * r_b;
    if (VER == 4 && c.truth_thresh < 1.f) obj = obj + ((iou_b > c.truth_thresh) ? 1.f : 0.f) * (1.f - obj);
    noobj = (1.f - obj) * ((iou_b < c.ignore_thresh) ? 1.f : 0.f);
  }
  float g = 0.f;  // dL/dp before the 1/N factor

  if (k < 4) {
    const float dIk = (k == 0) ? d0 : (k == 1) ? d1 : (k == 2) ? d2 : d3;
    if (VER == 2 || VER == 3) {
      const float s = (VER == 2 || c.use_scale) ? (2.f - Tb[2] * Tb[3]) : 1.f;
      if (k < 2) {
        const float diff = tbk - p;
        const float l = obj * s * (diff * diff);
        part[1] += (double)l;
        part[0] += (double)c.loss_weight[0] * (double)l;
        g = -2.f * c.loss_weight[0] * obj * s * diff;
      } else {
        const float tl = logf(fmaxf(tbk / an, LEPS));
        const float pl = logf(p / an);
        const float diff = tl - pl;
        const float l = obj * s * (diff * diff);
        const float reg = pl * pl;
        part[2] += (double)l;
        part[6] += (double)reg;
        part[0] += (double)c.loss_weight[1] * (double)l + 0.01 * (double)reg;
        g = (-2.f * c.loss_weight[1] * obj * s * diff + 0.02f * pl) / p;
      }
    } else if (VER == 4) {
      // box loss gradient through CIoU; wh regulariser on the log-ratio
      g = -c.loss_weight[0] * obj * dIk;
      if (k >= 2) {
        const float pl = logf(p / an);
        const float reg = pl * pl;
        part[6] += (double)reg;
        part[0] += (double)c.wh_reg_weight * (double)reg;
        g += 2.f * c.wh_reg_weight * pl / p;
      }
    } else {  // VER == 1
      // confidence target is the IoU itself and is differentiated through
      g = 2.f * c.loss_weight[2] * obj * (iou_b - cb) * dIk;
      if (k < 2) {
        const float diff = tbk - p;
        const float l = obj * (diff * diff);
        part[1] += (double)l;
        part[0] += (double)c.loss_weight[0] * (double)l;
        g += -2.f * c.loss_weight[0] * obj * diff;
      } else {
        const float tq = sqrtf(fmaxf(tbk, LEPS));
        const float pq = sqrtf(fmaxf(p, LEPS));
        const float diff = tq - pq;
        const float l = obj * (diff * diff);
        part[2] += (double)l;
        part[0] += (double)c.loss_weight[1] * (double)l;
        if (p >= LEPS) g += -2.f * c.loss_weight[1] * obj * diff * (0.5f / pq);
      }
    }
  } else if (k == 4) {
    if (VER == 1) {
      const float e = iou_b - p;
      const float lo = obj * (e * e);
      const float ln = noobj * (p * p);
      part[3] += (double)lo;
      part[4] += (double)ln;
      part[0] += (double)c.loss_weight[2] * ((double)lo + (double)c.binary_weight * (double)ln);
      g = c.loss_weight[2] * (-2.f * obj * e + 2.f * c.binary_weight * noobj * p);
    } else if (VER == 2 || (VER == 3 && !c.use_focal_loss)) {
      const float lo = obj * ((1.f - p) * (1.f - p));
      const float ln = noobj * (p * p);
      part[3] += (double)lo;
      part[4] += (double)ln;
      part[0] += (double)c.loss_weight[2] * ((double)lo + (double)c.binary_weight * (double)ln);
      g = c.loss_weight[2] * (-2.f * obj * (1.f - p) + 2.f * c.binary_weight * noobj * p);
    } else if (VER == 3) {  // focal
      const float cc = fminf(fmaxf(p, LEPS), ONE_M_EPS);
      const float pass = (p >= LEPS && p <= ONE_M_EPS) ? 1.f : 0.f;
      const float gm = c.focal_gamma;
      // (gamma = 2, the reference's default: x^2 and x^1 without libm's powf -- the ONE confidence lane of a 64-channel
      // chunk ran it four times while 63 lanes waited)
      const bool g2 = gm == 2.f;
      const float a1 = g2 ? (1.f - cc) * (1.f - cc) : powf(1.f - cc, gm), l1 = logf(cc);
      const float a0 = g2 ? cc * cc : powf(cc, gm), l0 = logf(1.f - cc);
      const float lo = -obj * a1 * l1;
      const float ln = -noobj * a0 * l0;
      part[3] += (double)lo;
      part[4] += (double)ln;
      part[0] += (double)c.loss_weight[2] * ((double)lo + (double)c.binary_weight * (double)ln);
      const float d1 = -obj * (-gm * (g2 ? (1.f - cc) : powf(1.f - cc, gm - 1.f)) * l1 + a1 / cc);
      const float d0 = -noobj * (gm * (g2 ? cc : powf(cc, gm - 1.f)) * l0 - a0 / (1.f - cc));
      g = c.loss_weight[2] * (d1 + c.binary_weight * d0) * pass;
    } else {  // VER == 4
      const float cc = fminf(fmaxf(p, LEPS), ONE_M_EPS);
      const float pass = (p >= LEPS && p <= ONE_M_EPS) ? 1.f : 0.f;
      const float gm = c.focal_gamma;
      float eo, en, deo, den;  // errors and d(error)/dc
      if (c.label_smooth > 0.f) {
        const float uo = 1.f - c.label_smooth - cc;
        const float un = c.label_smooth - cc;
        eo = fabsf(uo);
        en = fabsf(un);
        deo = (uo > 0.f) ? -1.f : (uo < 0.f ? 1.f : 0.f);
        den = (un > 0.f) ? -1.f : (un < 0.f ? 1.f : 0.f);
      } else {
        eo = 1.f - cc;
        en = cc;
        deo = -1.f;
        den = 1.f;
      }
      const bool g2 = gm == 2.f;   // (see the focal branch above)
      const float ao = g2 ? eo * eo : powf(eo, gm), lo_ = logf(1.f - eo);
      const float an_ = g2 ? en * en : powf(en, gm), ln_ = logf(1.f - en);
      const float lo = -obj * ao * lo_;
      const float ln = -noobj * an_ * ln_;
      part[3] += (double)lo;
      part[4] += (double)ln;
      part[0] += (double)c.loss_weight[1] * ((double)lo + (double)c.binary_weight * (double)ln);
      const float dfo = gm * (g2 ? eo : powf(eo, gm - 1.f)) * lo_ - ao / (1.f - eo);  // d/de [e^g log(1-e)]
      const float dfn = gm * (g2 ? en : powf(en, gm - 1.f)) * ln_ - an_ / (1.f - en);
      g = c.loss_weight[1] * (-obj * dfo * deo - c.binary_weight * noobj * dfn * den) * pass;
    }
  } else {  // class channel
    const float tk = tk_cur;
    const float pc = fminf(fmaxf(p, LEPS), ONE_M_EPS);
    const float pass = (p >= LEPS && p <= ONE_M_EPS) ? 1.f : 0.f;
    const float wcls = (VER == 4) ? c.loss_weight[2] : c.loss_weight[3];
    if (VER != 1 && obj == 0.f) {
      // every class term carries the factor obj: loss part and gradient are (signed) zeros -- 99.7 % of the slots of a
      // 52x52 level -- so the two logarithms and two divisions per element are skipped; adding -0.0 to the sums and
      // storing -0.0 instead of +0.0 would change nothing either
      g = 0.f;
    } else if (VER == 3 || VER == 4) {
      const float l = -obj * (tk * logf(pc) + (1.f - tk) * logf(1.f - pc));
      part[5] += (double)l;
      part[0] += (double)wcls * (double)l;
      g = -wcls * obj * (tk / pc - (1.f - tk) / (1.f - pc)) * pass;
    } else if (VER == 2) {
      const float l = -obj * (tk * logf(pc));
      part[5] += (double)l;
      part[0] += (double)wcls * (double)l;
      g = -wcls * obj * (tk / pc) * pass;
    } else {  // VER == 1: per cell, mask is the cell's objectness only
      const float l = -t4 * tk * logf(pc);
      part[5] += (double)l;
      part[0] += (double)wcls * (double)l;
      g = -wcls * t4 * (tk / pc) * pass;
    }
  }
  return g;
}

// parts: [0] total (weighted), then per-version diagnostics (see yolo_hip.h)
// PREF (round 6; v2 / v3 / v4 with at most 256 prediction channels per cell): EVERYTHING a cell needs -- its four 64-channel
// chunks of predictions, the class targets beside them, the true box and the anchors' predicted boxes, 13 loads per lane --
// is requested ONE CELL AHEAD, while the current cell is worked on. One wave per cell with the next chunk's two loads in
// flight (round 4) kept ~0.5 KB per wave in the air: 8 waves per SIMD x 1024 SIMDs x 0.5 KB over a ~2 us round trip is the
// 1.1-1.5 TB/s the kernel sat at, whatever its arithmetic did. A cell ahead it is 1.4 KB per wave and no load is waited for
// at the head of a cell (the box loads that start a cell's dependency chain come out of registers).
template <int VER, bool PREF>
__global__ __launch_bounds__(256) void loss_kernel(const LossParams lp, const float* __restrict__ y_true,
                                                   const float* __restrict__ y_pred, float* __restrict__ dpred,
                                                   double* __restrict__ out) {
  const yolo_loss_cfg& c = lp.c;
  const int A = c.A, C = c.C;
  const int TD = 5 + C;
  const int PD = (VER == 1) ? 5 * A + C : A * (5 + C);
  const int lane = threadIdx.x & 63;
  const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  const float gw = (float)c.gw, gh = (float)c.gh;
  const double invN = 1.0 / (double)c.N;

  double part[7] = {0, 0, 0, 0, 0, 0, 0};

  // lane j keeps anchors[j] (kernarg reads with compile-time indices), fetched by shuffle below
  float my_anchor = 1.f;
#pragma unroll
  for (int j = 0; j < 32; ++j)
    if (lane == j) my_anchor = c.use_anchors ? c.anchors[j] : 1.f;

  auto chan_of = [&](int idx, int& b, int& k) {   // anchor, channel within anchor (k >= 5: class k-5)
    if (VER == 1) {
      if (idx < 5 * A) {
        b = idx / 5;
        k = idx - b * 5;
      } else {
        b = 0;
        k = 5 + (idx - 5 * A);
      }
    } else {
      b = idx / TD;
      k = idx - b * TD;
    }
  };
  // the per-anchor IoU on lanes 0..A-1, the responsible anchor, the decisions, the v4 box term: shared by both loaders
  auto cell_head = [&](long long cell, const float (&Tb)[4], const float t4, const float (&Pb)[4], float& iou, float (&dI)[4],
                       int& resp) {
    iou = -1.f;
    float ciou = 0.f;
    dI[0] = dI[1] = dI[2] = dI[3] = 0.f;   // partials of the differentiated score (iou for v1, ciou for v4)
    if (lane < A) {
      const IouOut o = iou_dual<VER == 4>(Tb, Pb, gw, gh);
      iou = o.iou.v;
      ciou = o.ciou.v;
      const D4& src = (VER == 4) ? o.ciou : o.iou;
      for (int i = 0; i < 4; ++i) dI[i] = src.d[i];
    }
    // responsible anchor: first maximal IoU (tf.argmax)
    resp = 0;
    float best = __shfl(iou, 0, 64);
    for (int b = 1; b < A; ++b) {
      const float ib = __shfl(iou, b, 64);
      if (ib > best) {
        best = ib;
        resp = b;
      }
    }
    if (lp.dec != nullptr) {
      // the decisions a second execution can only reproduce by being told: [0] responsible anchor, [1] bit b = "anchor b is
      // below ignore_thresh", bit 16 + b = "anchor b is above truth_thresh" (tests force the oracle's decisions to these)
      const unsigned long long ign = __ballot(lane < A && iou < c.ignore_thresh);
      const unsigned long long tru = __ballot(lane < A && iou > c.truth_thresh);
      if (lane == 0) {
        lp.dec[cell * 2] = resp;
        lp.dec[cell * 2 + 1] = (int)((unsigned)(ign & 0xffffu) | ((unsigned)(tru & 0xffffu) << 16));
      }
    }
    // v4 box term lives on the anchor lanes
    if (VER == 4 && lane < A) {
      float obj = t4 * (lane == resp ? 1.f : 0.f);
      if (c.truth_thresh < 1.f) obj = obj + ((iou > c.truth_thresh) ? 1.f : 0.f) * (1.f - obj);
      const double box = (double)(obj * (1.f - ciou));
      part[1] += box;
      part[0] += (double)c.loss_weight[0] * box;
    }
  };

  if constexpr (PREF) {
    constexpr int NCK = 4;   // 64-channel chunks of a cell (PD <= 256, checked by the launcher)
    float pn[NCK], tkn[NCK], tbn, pbn[4];        // the NEXT cell: predictions, class targets, T[0..4] on lanes 0..4, my anchor's box
    auto prefetch = [&](long long cell) {
      const float* T = y_true + cell * TD;
      const float* Pc = y_pred + cell * PD;
#pragma unroll
      for (int q = 0; q < NCK; ++q) {
        int idx = q * 64 + lane;
        idx = idx < PD ? idx : PD - 1;
        int b, k;
        chan_of(idx, b, k);
        pn[q] = Pc[idx];
        tkn[q] = T[k >= 5 ? k : 4];
      }
      tbn = T[lane < 5 ? lane : 4];
      const float* P = Pc + (lane < A ? lane : 0) * TD;
#pragma unroll
      for (int j = 0; j < 4; ++j) pbn[j] = P[j];
    };
    long long cell = wave0;
    if (cell < lp.cells) prefetch(cell);
    for (; cell < lp.cells; cell += nwaves) {
      float pc[NCK], tkc[NCK], Pb[4];
#pragma unroll
      for (int q = 0; q < NCK; ++q) {
        pc[q] = pn[q];
        tkc[q] = tkn[q];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) Pb[j] = pbn[j];
      const float tbc = tbn;
      if (cell + nwaves < lp.cells) prefetch(cell + nwaves);
      float* G = dpred ? dpred + cell * PD : nullptr;
      const float Tb[4] = {__shfl(tbc, 0, 64), __shfl(tbc, 1, 64), __shfl(tbc, 2, 64), __shfl(tbc, 3, 64)};
      const float t4 = __shfl(tbc, 4, 64);
      float iou, dI[4];
      int resp;
      cell_head(cell, Tb, t4, Pb, iou, dI, resp);
#pragma unroll
      for (int q = 0; q < NCK; ++q) {
        const int base = q * 64;
        if (base < PD) {   // (wave-uniform)
          const bool active = (base + lane) < PD;
          const int idx = active ? (base + lane) : (PD - 1);
          int b, k;
          chan_of(idx, b, k);
          // all cross-lane traffic happens here, with every lane participating
          const float iou_b = __shfl(iou, b, 64);
          const float d0 = __shfl(dI[0], b, 64), d1 = __shfl(dI[1], b, 64);
          const float d2 = __shfl(dI[2], b, 64), d3 = __shfl(dI[3], b, 64);
          const float an = __shfl(my_anchor, (b * 2 + ((k == 3) ? 1 : 0)) & 31, 64);
          const float r_b = (b == resp) ? 1.f : 0.f;
          if (active) {
            const float g = loss_channel<VER>(c, b, k, pc[q], tkc[q], t4, Tb, iou_b, d0, d1, d2, d3, an, 0.f, r_b, part);
            if (G) G[idx] = g * lp.inv_n;
          }
        }
      }
    }
  } else {
  for (long long cell = wave0; cell < lp.cells; cell += nwaves) {
    const float* T = y_true + cell * TD;
    const float* Pc = y_pred + cell * PD;
    float* G = dpred ? dpred + cell * PD : nullptr;

    // ---- per-anchor IoU on lanes 0..A-1 ----
    const float t4 = T[4];
    const float Tb[4] = {T[0], T[1], T[2], T[3]};
    float Pb[4] = {0.f, 0.f, 0.f, 0.f};
    if (lane < A) {
      const float* P = Pc + ((VER == 1) ? lane * 5 : lane * (5 + C));
      for (int j = 0; j < 4; ++j) Pb[j] = P[j];
    }
    float iou, dI[4];
    int resp;
    cell_head(cell, Tb, t4, Pb, iou, dI, resp);

    // ---- sweep the prediction channels ----
    // (the loads of chunk i + 1 -- the prediction and, for a class channel, its target -- are issued before chunk i is worked on:
    // one wave per cell had ONE load in flight per 64 channels; 52x52 level 227 -> 179 us for 206 MB)
    float p_nxt, tk_nxt;
    {
      const int idx0 = lane < PD ? lane : (PD - 1);
      int b0, k0;
      chan_of(idx0, b0, k0);
      p_nxt = Pc[idx0];
      tk_nxt = T[k0 >= 5 ? k0 : 4];
    }
    for (int base = 0; base < PD; base += 64) {
      const bool active = (base + lane) < PD;
      const int idx = active ? (base + lane) : (PD - 1);
      int b, k;
      chan_of(idx, b, k);
      const float p = p_nxt, tk_cur = tk_nxt;
      if (base + 64 < PD) {
        const int idn = (base + 64 + lane) < PD ? (base + 64 + lane) : (PD - 1);
        int bn, kn;
        chan_of(idn, bn, kn);
        p_nxt = Pc[idn];
        tk_nxt = T[kn >= 5 ? kn : 4];
      }
      // all cross-lane traffic happens here, with every lane participating
      const float iou_b = __shfl(iou, b, 64);
      const float d0 = __shfl(dI[0], b, 64), d1 = __shfl(dI[1], b, 64);
      const float d2 = __shfl(dI[2], b, 64), d3 = __shfl(dI[3], b, 64);
      const float an = __shfl(my_anchor, (b * 2 + ((k == 3) ? 1 : 0)) & 31, 64);
      const float cb = (VER == 1) ? Pc[b * 5 + 4] : 0.f;
      const float r_b = (b == resp) ? 1.f : 0.f;
      if (!active) continue;  // no cross-lane operations below this line
      const float g = loss_channel<VER>(c, b, k, p, tk_cur, t4, Tb, iou_b, d0, d1, d2, d3, an, cb, r_b, part);
      if (G) G[idx] = g * lp.inv_n;
    }
  }
  }

  // block reduction of the 7 parts, one fp64 atomic per part per block
  __shared__ double red[7][4];
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const double s = wave_reduce_sum(part[q]);
    if (lane == 0) red[q][wave] = s;
  }
  __syncthreads();
  if (threadIdx.x < 7) {
    const double s = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    atomicAdd(&out[threadIdx.x], s * invN);
  }
}

// ---- metrics (yolov3/metrics/yolo_metrics.py, yolov1_5/metrics/yolo_metrics.py) -----------
template <int VER>
__global__ __launch_bounds__(256) void metrics_kernel(const LossParams lp, const float* __restrict__ y_true,
                                                      const float* __restrict__ y_pred, float recall_thresh,
                                                      double* __restrict__ out) {
  const yolo_loss_cfg& c = lp.c;
  const int A = c.A, C = c.C;
  const int TD = 5 + C;
  const int PD = (VER == 1) ? 5 * A + C : A * (5 + C);
  const int lane = threadIdx.x & 63;
  const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  const float gw = (float)c.gw, gh = (float)c.gh;
  double part[5] = {0, 0, 0, 0, 0};

  for (long long cell = wave0; cell < lp.cells; cell += nwaves) {
    const float* T = y_true + cell * TD;
    const float* Pc = y_pred + cell * PD;
    const float t4 = T[4];
    float iou = -FLT_MAX, conf = -FLT_MAX;
    if (lane < A) {
      const float* P = Pc + ((VER == 1) ? lane * 5 : lane * (5 + C));
      const float Tb[4] = {T[0], T[1], T[2], T[3]};
      const float Pb[4] = {P[0], P[1], P[2], P[3]};
      iou = iou_dual<false>(Tb, Pb, gw, gh).iou.v;
      conf = P[4];
    }
    // argmax over classes: truth (lanes cooperate), first maximal index
    auto wave_argmax = [&](const float* v) -> int {
      float bv = -FLT_MAX;
      int bi = 0x7fffffff;
      for (int k = lane; k < C; k += 64) {
        const float x = v[k];
        if (x > bv) {
          bv = x;
          bi = k;
        }
      }
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) {
          bv = ov;
          bi = oi;
        }
      }
      return bi;
    };
    const int pi_true = wave_argmax(T + 5);
    float max_conf = -FLT_MAX, max_iou = -FLT_MAX, max_rec = -FLT_MAX;
    float eq_sum = 0.f;
    int pi_pred_cell = 0;
    if (VER == 1) pi_pred_cell = wave_argmax(Pc + 5 * A);
    for (int b = 0; b < A; ++b) {
      const float ib = __shfl(iou, b, 64);
      const float cb = __shfl(conf, b, 64);
      const int pi_pred = (VER == 1) ? pi_pred_cell : wave_argmax(Pc + b * TD + 5);
      const float eq = ((pi_pred == pi_true) ? 1.f : 0.f) * t4;
      max_conf = fmaxf(max_conf, cb);
      max_iou = fmaxf(max_iou, ib);
      max_rec = fmaxf(max_rec, ib * eq);
      eq_sum += eq;
    }
    if (lane == 0) {
      const float pred_bin = (max_conf > 0.5f) ? 1.f : 0.f;
      part[0] += (t4 == pred_bin) ? 1.0 : 0.0;
      part[1] += (double)(max_iou * t4);
      part[2] += (double)t4;
      part[3] += (VER == 1) ? (double)(((pi_pred_cell == pi_true) ? 1.f : 0.f) * t4) : (double)eq_sum;
      part[4] += (max_rec >= recall_thresh) ? 1.0 : 0.0;
    }
  }
  __shared__ double red[5][4];
  const int wave = threadIdx.x >> 6;
  if (lane == 0)
    for (int q = 0; q < 5; ++q) red[q][wave] = part[q];
  __syncthreads();
  if (threadIdx.x < 5) {
    const double s = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    atomicAdd(&out[threadIdx.x], s);
  }
  if (blockIdx.x == 0 && threadIdx.x == 5) out[5] = (double)lp.cells;
}

static int validate_cfg(const yolo_loss_cfg* cfg) {
  YOLO_REQUIRE(cfg != nullptr, "loss: null cfg");
  YOLO_REQUIRE(cfg->version >= 1 && cfg->version <= 4, "loss: bad version %d", cfg->version);
  YOLO_REQUIRE(cfg->N > 0 && cfg->gh > 0 && cfg->gw > 0 && cfg->C > 0, "loss: bad shape");
  YOLO_REQUIRE(cfg->A > 0 && cfg->A <= 16, "loss: anchors per cell %d out of range [1,16]", cfg->A);
  return YOLO_OK;
}

}  // namespace yolo

using namespace yolo;

extern "C" size_t yolo_loss_workspace_bytes(const yolo_loss_cfg* cfg) {
  (void)cfg;
  return 0;  // the fused kernel needs no scratch; kept in the ABI for forward compatibility
}

extern "C" int yolo_loss_fwd_bwd(const yolo_loss_cfg* cfg, const float* y_true, const float* y_pred, double* loss_out,
                                 float* dpred, float grad_scale, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  if (int rc = validate_cfg(cfg)) return rc;
  YOLO_REQUIRE(y_true && y_pred && loss_out, "loss: null pointer");
  LossParams lp;
  lp.c = *cfg;
  lp.cells = (long long)cfg->N * cfg->gh * cfg->gw;
  lp.inv_n = grad_scale / (float)cfg->N;
  // the kernel needs no scratch; a workspace of at least 8 bytes per cell receives the cells' discrete decisions
  lp.dec = (workspace != nullptr && workspace_bytes >= (size_t)lp.cells * 8 && cfg->A <= 16) ? reinterpret_cast<int*>(workspace)
                                                                                              : nullptr;
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(loss_out, 0, 8 * sizeof(double), st) != hipSuccess) {
    set_error("loss: memset failed");
    return YOLO_ERR_LAUNCH;
  }
  long long blocks = (lp.cells + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  dim3 grid((unsigned)blocks), block(256);
  // the loader that requests a whole cell one cell ahead (loss_kernel<VER, true>): v2 / v3 / v4, at most 256 prediction
  // channels per cell; YOLO_LOSS_PREFETCH=0 keeps the chunk-ahead loader (A/B, tests: the two must agree bit for bit)
  static const bool pref_env = [] { const char* e = getenv("YOLO_LOSS_PREFETCH"); return !(e && atoi(e) == 0); }();
  const bool pref = pref_env && !(g_opt[OPT_EXP] & 8) && cfg->version != 1 && cfg->A * (5 + cfg->C) <= 256;
  switch (cfg->version) {
    case 1: hipLaunchKernelGGL((loss_kernel<1, false>), grid, block, 0, st, lp, y_true, y_pred, dpred, loss_out); break;
    case 2:
      if (pref) hipLaunchKernelGGL((loss_kernel<2, true>), grid, block, 0, st, lp, y_true, y_pred, dpred, loss_out);
      else hipLaunchKernelGGL((loss_kernel<2, false>), grid, block, 0, st, lp, y_true, y_pred, dpred, loss_out);
      break;
    case 3:
      if (pref) hipLaunchKernelGGL((loss_kernel<3, true>), grid, block, 0, st, lp, y_true, y_pred, dpred, loss_out);
      else hipLaunchKernelGGL((loss_kernel<3, false>), grid, block, 0, st, lp, y_true, y_pred, dpred, loss_out);
      break;
    default:
      if (pref) hipLaunchKernelGGL((loss_kernel<4, true>), grid, block, 0, st, lp, y_true, y_pred, dpred, loss_out);
      else hipLaunchKernelGGL((loss_kernel<4, false>), grid, block, 0, st, lp, y_true, y_pred, dpred, loss_out);
      break;
  }
  return check_launch("loss_kernel");
}

extern "C" int yolo_metrics(const yolo_loss_cfg* cfg, const float* y_true, const float* y_pred, float recall_thresh,
                            double* out, void* stream) {
  if (int rc = validate_cfg(cfg)) return rc;
  YOLO_REQUIRE(y_true && y_pred && out, "metrics: null pointer");
  LossParams lp;
  lp.c = *cfg;
  lp.cells = (long long)cfg->N * cfg->gh * cfg->gw;
  lp.inv_n = 0.f;
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(out, 0, 8 * sizeof(double), st) != hipSuccess) {
    set_error("metrics: memset failed");
    return YOLO_ERR_LAUNCH;
  }
  long long blocks = (lp.cells + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  dim3 grid((unsigned)blocks), block(256);
  switch (cfg->version) {
    case 1: hipLaunchKernelGGL(metrics_kernel<1>, grid, block, 0, st, lp, y_true, y_pred, recall_thresh, out); break;
    case 2: hipLaunchKernelGGL(metrics_kernel<2>, grid, block, 0, st, lp, y_true, y_pred, recall_thresh, out); break;
    case 3: hipLaunchKernelGGL(metrics_kernel<3>, grid, block, 0, st, lp, y_true, y_pred, recall_thresh, out); break;
    default: hipLaunchKernelGGL(metrics_kernel<4>, grid, block, 0, st, lp, y_true, y_pred, recall_thresh, out); break;
  }
  return check_launch("metrics_kernel");
}

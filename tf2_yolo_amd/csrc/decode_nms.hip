// Score-filter decode and the NMS family, bit-exact with the reference's NumPy code:
//   decode   utils/tools.py:370-438   (threshold on the fp32 product conf*prob, rows in C order
//                                       (y, x, box, class), coordinates in fp64)
//   cal_iou  utils/tools.py:630-684   (IoU / DIoU in fp64, +1e-7 in the denominator)
//   nms      utils/tools.py:687-733   (per class, greedy in descending conf*prob, test `>=`)
//   soft_nms utils/tools.py:736-786   (static order, Gaussian decay, no "kept" guard)
//
// decode is an ordered stream compaction (count -> scan -> write). NMS ranks every row inside
// its class by brute force (O(n^2) pair tests run at tens of Gpairs/s on 256 CUs, so a radix
// sort is not worth its code), then resolves each class in one workgroup: hard/DIoU walk the
// sorted list with the suppression test fanned out over 1024 lanes; soft-NMS has no serial
// dependency at all (a row's fate depends only on the rows ranked before it) and is one
// thread per row. Ties in conf*prob are ordered "higher original index first" (the
// reference's argsort is unstable there).
#include "common.hpp"

namespace yolo {

constexpr int DEC_BLOCK = 256;
constexpr int DEC_ITEMS = 8;  // candidates per thread

struct DecodeGeom {
  int gh, gw, A, C, version;
  long long total;  // gh*gw*A*C candidate slots
};

template <typename T>
__device__ __forceinline__ bool decode_flag(const T* pred, const DecodeGeom& g, long long e, int& yi, int& xi, int& bi,
                                            int& ci, long long& box_off, long long& prob_off) {
  ci = (int)(e % g.C);
  long long r = e / g.C;
  bi = (int)(r % g.A);
  r /= g.A;
  xi = (int)(r % g.gw);
  yi = (int)(r / g.gw);
  const long long cell = (long long)yi * g.gw + xi;
  if (g.version == 1) {
    const int D = 5 * g.A + g.C;
    box_off = cell * D + bi * 5;
    prob_off = cell * D + 5 * g.A + ci;
  } else {
    const int D = g.A * (5 + g.C);
    box_off = cell * D + (long long)bi * (5 + g.C);
    prob_off = box_off + 5 + ci;
  }
  return true;
}

template <typename T>
__global__ __launch_bounds__(DEC_BLOCK) void decode_count_kernel(const T* __restrict__ pred, DecodeGeom g, T thr,
                                                                 int* __restrict__ block_counts) {
  __shared__ int wsum[DEC_BLOCK / 64];
  const long long base = ((long long)blockIdx.x * DEC_BLOCK + threadIdx.x) * DEC_ITEMS;
  int cnt = 0;
  for (int it = 0; it < DEC_ITEMS; ++it) {
    const long long e = base + it;
    if (e < g.total) {
      int yi, xi, bi, ci;
      long long bo, po;
      decode_flag(pred, g, e, yi, xi, bi, ci, bo, po);
      const T joint = pred[bo + 4] * pred[po];
      cnt += (joint >= thr) ? 1 : 0;
    }
  }
  int s = cnt;
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < DEC_BLOCK / 64; ++w) t += wsum[w];
    block_counts[blockIdx.x] = t;
  }
}

// single block: exclusive scan of block_counts[nb] in place, offset by *count; *count += total
__global__ void decode_scan_kernel(int* __restrict__ block_counts, int nb, int* __restrict__ count) {
  __shared__ int carry;
  __shared__ int buf[1024];
  if (threadIdx.x == 0) carry = *count;
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = (i < nb) ? block_counts[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int t = (threadIdx.x >= o) ? buf[threadIdx.x - o] : 0;
      __syncthreads();
      buf[threadIdx.x] += t;
      __syncthreads();
    }
    const int incl = buf[threadIdx.x];
    if (i < nb) block_counts[i] = carry + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *count = carry;
}

template <typename T>
__global__ __launch_bounds__(DEC_BLOCK) void decode_write_kernel(const T* __restrict__ pred, DecodeGeom g, T thr,
                                                                 const int* __restrict__ block_offsets,
                                                                 double* __restrict__ rows, int max_rows) {
  __shared__ int wsum[DEC_BLOCK / 64];
  const long long base = ((long long)blockIdx.x * DEC_BLOCK + threadIdx.x) * DEC_ITEMS;
  bool flag[DEC_ITEMS];
  int cnt = 0;
  for (int it = 0; it < DEC_ITEMS; ++it) {
    const long long e = base + it;
    flag[it] = false;
    if (e < g.total) {
      int yi, xi, bi, ci;
      long long bo, po;
      decode_flag(pred, g, e, yi, xi, bi, ci, bo, po);
      const T joint = pred[bo + 4] * pred[po];
      flag[it] = joint >= thr;
    }
    cnt += flag[it] ? 1 : 0;
  }
  // exclusive prefix over the block's threads (thread order == candidate order)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = cnt;
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < wave; ++w) woff += wsum[w];
  int pos = block_offsets[blockIdx.x] + woff + incl - cnt;
  for (int it = 0; it < DEC_ITEMS; ++it) {
    if (!flag[it]) continue;
    if (pos < max_rows) {
      const long long e = base + it;
      int yi, xi, bi, ci;
      long long bo, po;
      decode_flag(pred, g, e, yi, xi, bi, ci, bo, po);
      double* r = rows + (long long)pos * 7;
      r[0] = ((double)xi + (double)pred[bo + 0]) / (double)g.gw;
      r[1] = ((double)yi + (double)pred[bo + 1]) / (double)g.gh;
      r[2] = (double)pred[bo + 2];
      r[3] = (double)pred[bo + 3];
      r[4] = (double)pred[bo + 4];
      r[5] = (double)ci;
      r[6] = (double)pred[po];
    }
    ++pos;
  }
}

// -------------------------------------- NMS --------------------------------------------------
struct NmsWs {
  double* score;     // [n]
  double* box;       // [n][4] in sorted order
  double* sscore;    // [n] score in sorted order
  int* cls;          // [n]
  int* sorted_idx;   // [n] sorted position -> original row
  int* pos_of;       // [n] original row -> sorted position (or -1)
  int* class_cnt;    // [class_num + 1]
  int* class_off;    // [class_num + 1]
  unsigned char* removed;  // [n] by sorted position
  int* fill;         // [class_num + 1] rows placed so far per class (bucket kernel)
  int* bidx;         // [n] rows grouped by class (any order inside a class)
  double* bscore;    // [n] their scores
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static size_t nms_carve(int n, int class_num, void* base, NmsWs* ws) {
  size_t off = 0;
  char* b = reinterpret_cast<char*>(base);
  auto take = [&](size_t bytes) {
    void* p = b ? b + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  NmsWs w;
  w.score = (double*)take(sizeof(double) * n);
  w.box = (double*)take(sizeof(double) * 4 * n);
  w.sscore = (double*)take(sizeof(double) * n);
  w.cls = (int*)take(sizeof(int) * n);
  w.sorted_idx = (int*)take(sizeof(int) * n);
  w.pos_of = (int*)take(sizeof(int) * n);
  w.class_cnt = (int*)take(sizeof(int) * (class_num + 1));
  w.class_off = (int*)take(sizeof(int) * (class_num + 1));
  w.removed = (unsigned char*)take(n);
  w.fill = (int*)take(sizeof(int) * (class_num + 1));
  w.bidx = (int*)take(sizeof(int) * n);
  w.bscore = (double*)take(sizeof(double) * n);
  if (ws) *ws = w;
  return off;
}

__global__ void nms_prepare_kernel(const double* __restrict__ rows, int n, int class_num, NmsWs ws) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* r = rows + (long long)i * 7;
  int c = (int)r[5];  // astype("int") truncates toward zero
  if (c < 0 || c >= class_num) c = -1;
  ws.cls[i] = c;
  ws.score[i] = r[4] * r[6];
  ws.pos_of[i] = -1;
  ws.removed[i] = 0;
  if (c >= 0) atomicAdd(&ws.class_cnt[c], 1);
}

__global__ void nms_class_scan_kernel(int class_num, NmsWs ws) {
  // single thread: class_num is small (<= a few thousand)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int acc = 0;
    for (int c = 0; c < class_num; ++c) {
      ws.class_off[c] = acc;
      acc += ws.class_cnt[c];
    }
    ws.class_off[class_num] = acc;
  }
}

// rows grouped by class: slot = class_off[c] + (arrival order inside the class). The order inside a class is whatever the
// atomics give -- the rank below does not depend on it.
__global__ void nms_bucket_kernel(int n, NmsWs ws) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = ws.cls[i];
  if (c < 0) return;
  const int slot = ws.class_off[c] + atomicAdd(&ws.fill[c], 1);
  ws.bidx[slot] = i;
  ws.bscore[slot] = ws.score[i];
}

// rank of every row inside its class by (score descending, ties: higher original index first -- np.argsort leaves ties
// undefined, utils/tools.py:717) = the number of rows of the class that come before it: one thread per bucket slot,
// walking ITS CLASS's slots only (sum n_c^2 comparisons instead of n^2: 80 classes of 1641 rows = 2.2e8 instead of
// 1.7e10); the threads of a wave share a class almost always, so the loads are broadcasts
__global__ __launch_bounds__(256) void nms_rank_kernel(const double* __restrict__ rows, int n, int class_num, NmsWs ws) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= ws.class_off[class_num]) return;
  const int i = ws.bidx[e];
  const double si = ws.bscore[e];
  const int ci = ws.cls[i];
  const int beg = ws.class_off[ci], end = ws.class_off[ci + 1];
  int rank = 0;
  for (int j = beg; j < end; ++j) {
    const double sj = ws.bscore[j];
    const int jj = ws.bidx[j];
    rank += ((sj > si) || (sj == si && jj > i)) ? 1 : 0;
  }
  const int pos = beg + rank;
  ws.sorted_idx[pos] = i;
  ws.pos_of[i] = pos;
  const double* r = rows + (long long)i * 7;
  ws.box[(long long)pos * 4 + 0] = r[0];
  ws.box[(long long)pos * 4 + 1] = r[1];
  ws.box[(long long)pos * 4 + 2] = r[2];
  ws.box[(long long)pos * 4 + 3] = r[3];
  ws.sscore[pos] = si;
}

// utils/tools.py:630-684 in fp64; a = "true" (visited box), b = "pred"
__device__ __forceinline__ double pair_score(const double* a, const double* b, bool diou) {
#pragma clang fp contract(off)   // NumPy rounds every product: no fused multiply-adds in the threshold comparisons
  const double ahx = a[2] / 2., ahy = a[3] / 2., bhx = b[2] / 2., bhy = b[3] / 2.;
  const double aminx = a[0] - ahx, amaxx = a[0] + ahx, aminy = a[1] - ahy, amaxy = a[1] + ahy;
  const double bminx = b[0] - bhx, bmaxx = b[0] + bhx, bminy = b[1] - bhy, bmaxy = b[1] + bhy;
  const double iw = fmax(fmin(bmaxx, amaxx) - fmax(bminx, aminx), 0.);
  const double ih = fmax(fmin(bmaxy, amaxy) - fmax(bminy, aminy), 0.);
  const double inter = iw * ih;
  const double ta = a[2] * a[3], pa = b[2] * b[3];
  const double uni = pa + ta - inter;
  const double iou = inter / (uni + 1e-07);
  if (!diou) return iou;
  const double ewx = fmax(bmaxx, amaxx) - fmin(bminx, aminx);
  const double ewy = fmax(bmaxy, amaxy) - fmin(bminy, aminy);
  const double c2 = ewx * ewx + ewy * ewy;
  const double dx = a[0] - b[0], dy = a[1] - b[1];
  const double rho2 = dx * dx + dy * dy;
  return iou - rho2 / c2;
}

// one workgroup per class: greedy walk over the sorted segment. Classes of up to NMS_LDS_BOXES rows keep their boxes and
// "removed" flags in LDS (every step of the walk is then one barrier and a few LDS reads, ~0.1 us, instead of a round
// trip to memory: 1641 steps per class on BASELINE.md's 131 304-row input); larger classes use the global arrays.
constexpr int NMS_LDS_BOXES = 4096;
__global__ __launch_bounds__(1024) void nms_hard_kernel(NmsWs ws, double thr, int diou) {
  extern __shared__ __attribute__((aligned(16))) unsigned char nms_smem[];
  const int c = blockIdx.x;
  const int beg = ws.class_off[c], end = ws.class_off[c + 1];
  const int nc = end - beg;
  if (nc <= NMS_LDS_BOXES) {
    double* sbox = reinterpret_cast<double*>(nms_smem);                    // [nc][4]
    unsigned char* srem = nms_smem + (size_t)NMS_LDS_BOXES * 32;            // [nc]
    for (int t = threadIdx.x; t < nc * 4; t += 1024) sbox[t] = ws.box[(long long)beg * 4 + t];
    for (int t = threadIdx.x; t < nc; t += 1024) srem[t] = 0;
    __syncthreads();
    for (int p = 0; p < nc; ++p) {
      if (!srem[p]) {    // (last written before the barrier that ended the previous iteration)
        const double a[4] = {sbox[p * 4], sbox[p * 4 + 1], sbox[p * 4 + 2], sbox[p * 4 + 3]};
        for (int q = p + 1 + threadIdx.x; q < nc; q += 1024) {
          if (srem[q]) continue;
          const double b[4] = {sbox[q * 4], sbox[q * 4 + 1], sbox[q * 4 + 2], sbox[q * 4 + 3]};
          if (pair_score(a, b, diou != 0) >= thr) srem[q] = 1;
        }
      }
      __syncthreads();
    }
    for (int t = threadIdx.x; t < nc; t += 1024) ws.removed[beg + t] = srem[t];
    return;
  }
  for (int p = beg; p < end; ++p) {
    // removed[p] was last written before the barrier that ended the previous iteration
    if (!ws.removed[p]) {
      const double a[4] = {ws.box[(long long)p * 4], ws.box[(long long)p * 4 + 1], ws.box[(long long)p * 4 + 2],
                           ws.box[(long long)p * 4 + 3]};
      for (int q = p + 1 + threadIdx.x; q < end; q += 1024) {
        if (ws.removed[q]) continue;
        const double b[4] = {ws.box[(long long)q * 4], ws.box[(long long)q * 4 + 1], ws.box[(long long)q * 4 + 2],
                             ws.box[(long long)q * 4 + 3]};
        if (pair_score(a, b, diou != 0) >= thr) ws.removed[q] = 1;
      }
    }
    __syncthreads();
  }
}

// soft-NMS: row q is deleted iff its score, decayed in rank order by every earlier row of its
// class with IoU >= thr, falls below conf_threshold after some decay.
__global__ void nms_soft_kernel(int n, int class_num, NmsWs ws, double thr, double conf_thr, double sigma) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n || q >= ws.class_off[class_num]) return;  // rows with an out-of-range class are not ranked
  const int i = ws.sorted_idx[q];
  const int c = ws.cls[i];
  const int beg = ws.class_off[c];
  const double b[4] = {ws.box[(long long)q * 4], ws.box[(long long)q * 4 + 1], ws.box[(long long)q * 4 + 2],
                       ws.box[(long long)q * 4 + 3]};
  double conf = ws.sscore[q];
  bool del = false;
  for (int p = beg; p < q; ++p) {
    const double a[4] = {ws.box[(long long)p * 4], ws.box[(long long)p * 4 + 1], ws.box[(long long)p * 4 + 2],
                         ws.box[(long long)p * 4 + 3]};
    const double iou = pair_score(a, b, false);
    if (iou >= thr) {
      conf *= exp(-1. * (iou * iou) / sigma);
      if (conf < conf_thr) del = true;
    }
  }
  ws.removed[q] = del ? 1 : 0;
}

__global__ void nms_finish_kernel(int n, NmsWs ws, unsigned char* __restrict__ keep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int pos = ws.pos_of[i];
  keep[i] = (pos >= 0 && !ws.removed[pos]) ? 1 : 0;
}

// ---- kept rows in the reference's output order: classes ascending, original order inside a class (utils/tools.py:730-732) ----
__global__ void nms_kept_count_kernel(int n, NmsWs ws, const unsigned char* __restrict__ keep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !keep[i]) return;
  atomicAdd(&ws.fill[ws.cls[i]], 1);     // (fill was zeroed again after the bucket kernel used it)
}
__global__ void nms_kept_scan_kernel(int class_num, NmsWs ws, int* __restrict__ count) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int acc = 0;
    for (int c = 0; c < class_num; ++c) {
      const int k = ws.fill[c];
      ws.class_cnt[c] = acc;             // (class_cnt is free by now: first output position of the class)
      acc += k;
    }
    *count = acc;
  }
}
// one thread per bucket slot: a kept row's place inside its class = the kept rows of the class with a smaller index
__global__ __launch_bounds__(256) void nms_gather_kernel(const double* __restrict__ rows, int class_num, NmsWs ws,
                                                         const unsigned char* __restrict__ keep, double* __restrict__ out) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= ws.class_off[class_num]) return;
  const int i = ws.bidx[e];
  if (!keep[i]) return;
  const int c = ws.cls[i];
  const int beg = ws.class_off[c], end = ws.class_off[c + 1];
  int r = 0;
  for (int j = beg; j < end; ++j) {
    const int jj = ws.bidx[j];
    r += (jj < i && keep[jj]) ? 1 : 0;
  }
  const double* src = rows + (long long)i * 7;
  double* dst = out + (long long)(ws.class_cnt[c] + r) * 7;
#pragma unroll
  for (int k = 0; k < 7; ++k) dst[k] = src[k];
}

}  // namespace yolo

using namespace yolo;

extern "C" size_t yolo_decode_workspace_bytes(int gh, int gw, int A, int C) {
  const long long total = (long long)gh * gw * A * C;
  const long long nb = (total + DEC_BLOCK * DEC_ITEMS - 1) / (DEC_BLOCK * DEC_ITEMS);
  return (size_t)(nb + 1) * sizeof(int);
}

template <typename T>
static int decode_impl(const T* pred, int gh, int gw, int A, int C, int version, T threshold, double* rows_out,
                       int max_rows, int* count, void* workspace, size_t workspace_bytes, void* stream) {
  YOLO_REQUIRE(pred && rows_out && count && workspace, "decode: null pointer");
  YOLO_REQUIRE(gh > 0 && gw > 0 && A > 0 && C > 0 && max_rows >= 0, "decode: bad shape");
  YOLO_REQUIRE(version >= 1 && version <= 4, "decode: Invalid version: %d", version);
  DecodeGeom g{gh, gw, A, C, version, (long long)gh * gw * A * C};
  const long long nb = (g.total + DEC_BLOCK * DEC_ITEMS - 1) / (DEC_BLOCK * DEC_ITEMS);
  if (workspace_bytes < (size_t)(nb + 1) * sizeof(int)) {
    set_error("decode: workspace %zu < %zu", workspace_bytes, (size_t)(nb + 1) * sizeof(int));
    return YOLO_ERR_WORKSPACE;
  }
  int* bc = reinterpret_cast<int*>(workspace);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL((decode_count_kernel<T>), dim3((unsigned)nb), dim3(DEC_BLOCK), 0, st, pred, g, threshold, bc);
  hipLaunchKernelGGL(decode_scan_kernel, dim3(1), dim3(1024), 0, st, bc, (int)nb, count);
  hipLaunchKernelGGL((decode_write_kernel<T>), dim3((unsigned)nb), dim3(DEC_BLOCK), 0, st, pred, g, threshold, bc,
                     rows_out, max_rows);
  return check_launch("decode kernels");
}

extern "C" int yolo_decode_level(const float* pred, int gh, int gw, int A, int C, int version, float threshold,
                                 double* rows_out, int max_rows, int* count, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  return decode_impl<float>(pred, gh, gw, A, C, version, threshold, rows_out, max_rows, count, workspace,
                            workspace_bytes, stream);
}

extern "C" int yolo_decode_level_f64(const double* pred, int gh, int gw, int A, int C, int version, double threshold,
                                     double* rows_out, int max_rows, int* count, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  return decode_impl<double>(pred, gh, gw, A, C, version, threshold, rows_out, max_rows, count, workspace,
                             workspace_bytes, stream);
}

extern "C" size_t yolo_nms_workspace_bytes(int n, int class_num) {
  if (n <= 0 || class_num <= 0) return 256;
  return nms_carve(n, class_num, nullptr, nullptr);
}

extern "C" int yolo_nms(const double* rows, int n, int class_num, int mode, double nms_threshold,
                        double conf_threshold, double sigma, unsigned char* keep_out, void* workspace,
                        size_t workspace_bytes, void* stream) {
  YOLO_REQUIRE(n >= 0 && class_num > 0, "nms: bad sizes");
  if (n == 0) return YOLO_OK;
  YOLO_REQUIRE(rows && keep_out && workspace, "nms: null pointer");
  YOLO_REQUIRE(mode == YOLO_NMS_HARD || mode == YOLO_NMS_SOFT || mode == YOLO_NMS_DIOU, "nms: bad mode %d", mode);
  NmsWs ws;
  const size_t need = nms_carve(n, class_num, workspace, &ws);
  if (workspace_bytes < need) {
    set_error("nms: workspace %zu < %zu", workspace_bytes, need);
    return YOLO_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(ws.class_cnt, 0, sizeof(int) * (class_num + 1), st) != hipSuccess) {
    set_error("nms: memset failed");
    return YOLO_ERR_LAUNCH;
  }
  if (hipMemsetAsync(ws.fill, 0, sizeof(int) * (class_num + 1), st) != hipSuccess) {
    set_error("nms: memset failed");
    return YOLO_ERR_LAUNCH;
  }
  const int nb = (n + 255) / 256;
  hipLaunchKernelGGL(nms_prepare_kernel, dim3(nb), dim3(256), 0, st, rows, n, class_num, ws);
  hipLaunchKernelGGL(nms_class_scan_kernel, dim3(1), dim3(64), 0, st, class_num, ws);
  hipLaunchKernelGGL(nms_bucket_kernel, dim3(nb), dim3(256), 0, st, n, ws);
  hipLaunchKernelGGL(nms_rank_kernel, dim3(nb), dim3(256), 0, st, rows, n, class_num, ws);
  if (mode == YOLO_NMS_SOFT) {
    hipLaunchKernelGGL(nms_soft_kernel, dim3(nb), dim3(256), 0, st, n, class_num, ws, nms_threshold, conf_threshold, sigma);
  } else {
    constexpr size_t lds = (size_t)NMS_LDS_BOXES * 33;
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nms_hard_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
    hipLaunchKernelGGL(nms_hard_kernel, dim3(class_num), dim3(1024), lds, st, ws, nms_threshold,
                       mode == YOLO_NMS_DIOU ? 1 : 0);
  }
  hipLaunchKernelGGL(nms_finish_kernel, dim3(nb), dim3(256), 0, st, n, ws, keep_out);
  return check_launch("nms kernels");
}

extern "C" int yolo_nms_select(const double* rows, int n, int class_num, int mode, double nms_threshold,
                               double conf_threshold, double sigma, unsigned char* keep_out, double* rows_out,
                               int* count_out, void* workspace, size_t workspace_bytes, void* stream) {
  YOLO_REQUIRE(count_out != nullptr, "nms_select: null pointer");
  hipStream_t st = as_stream(stream);
  if (n == 0) {
    if (hipMemsetAsync(count_out, 0, sizeof(int), st) != hipSuccess) return YOLO_ERR_LAUNCH;
    return YOLO_OK;
  }
  YOLO_REQUIRE(rows_out != nullptr, "nms_select: null pointer");
  if (int rc = yolo_nms(rows, n, class_num, mode, nms_threshold, conf_threshold, sigma, keep_out, workspace, workspace_bytes,
                        stream))
    return rc;
  NmsWs ws;
  nms_carve(n, class_num, workspace, &ws);
  if (hipMemsetAsync(ws.fill, 0, sizeof(int) * (class_num + 1), st) != hipSuccess) {
    set_error("nms_select: memset failed");
    return YOLO_ERR_LAUNCH;
  }
  const int nb = (n + 255) / 256;
  hipLaunchKernelGGL(nms_kept_count_kernel, dim3(nb), dim3(256), 0, st, n, ws, keep_out);
  hipLaunchKernelGGL(nms_kept_scan_kernel, dim3(1), dim3(64), 0, st, class_num, ws, count_out);
  hipLaunchKernelGGL(nms_gather_kernel, dim3(nb), dim3(256), 0, st, rows, class_num, ws, keep_out, rows_out);
  return check_launch("nms gather kernels");
}

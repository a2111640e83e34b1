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
#include "conv_args.hpp"   // (run-time options: g_opt, init_options)

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
// Round 5 layout of the hard / DIoU form. The greedy walk of a class is a chain of decisions, but the pair tests it needs
// are not: a class of n_c rows has n_c (n_c - 1) / 2 of them and none depends on another. So they are all made first, by
// the whole chip (nms_mask_kernel: 64 x 64 tiles of the upper triangle of every class, one wave per tile, bit (p, q) =
// "p suppresses q" -> one 64-bit word per row and tile), and the walk itself becomes bit arithmetic on that matrix
// (nms_scan_kernel: one workgroup per class, 64 rows at a time through LDS: the 64 x 64 diagonal block is resolved with
// scalar instructions, the surviving rows' other words are OR-ed into the class's "removed" bit set). Before: one
// workgroup per class made every test of its class itself, one barrier per visited row -- 3.9 ms for the 151 186
// candidates of a random-weight YOLOv3-416 prediction (3 557 rows in the largest class, nothing suppressed), 80 of 256 CUs
// busy. Classes of more than NMS_MASK_MAX rows (their matrix would not fit the workspace bound) keep a walk kernel.
constexpr int NMS_MASK_MAX = 8192;
constexpr int NMS_MASK_WORDS = NMS_MASK_MAX / 64;   // words per row at most

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
  long long* mask_off;          // [class_num + 1] first word of the class's bit matrix (-1: too large, walk kernel)
  int* tile_off;                // [class_num + 1] first tile of the class in nms_mask_kernel's tile list
  unsigned long long* mask;     // the matrices: row p of class c = words [mask_off[c] + p * T_c, + T_c), T_c = ceil(n_c / 64)
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
  w.mask_off = (long long*)take(sizeof(long long) * (class_num + 1));
  w.tile_off = (int*)take(sizeof(int) * (class_num + 1));
  // sum over the classes of n_c * ceil(n_c / 64) words with every n_c <= NMS_MASK_MAX and sum n_c <= n
  const size_t per_row = (size_t)((n < NMS_MASK_MAX ? n : NMS_MASK_MAX) / 64 + 1);
  w.mask = (unsigned long long*)take(sizeof(unsigned long long) * per_row * (size_t)n);
  if (ws) *ws = w;
  return off;
}

// per-class counters of a workgroup in LDS, one global atomic per (workgroup, class present): 151 186 rows adding to 80
// addresses one by one took 108 us per pass (device-scope atomics on one address queue up)
constexpr int NMS_LDS_CLASSES = 2048;

__global__ __launch_bounds__(256) void nms_prepare_kernel(const double* __restrict__ rows, int n, int class_num, NmsWs ws) {
  __shared__ int s_cnt[NMS_LDS_CLASSES];
  const bool lds = class_num <= NMS_LDS_CLASSES;
  if (lds)
    for (int c = threadIdx.x; c < class_num; c += 256) s_cnt[c] = 0;
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const double* r = rows + (long long)i * 7;
    int c = (int)r[5];  // astype("int") truncates toward zero
    if (c < 0 || c >= class_num) c = -1;
    ws.cls[i] = c;
    ws.score[i] = r[4] * r[6];
    ws.pos_of[i] = -1;
    ws.removed[i] = 0;
    if (c >= 0) {
      if (lds) atomicAdd(&s_cnt[c], 1);
      else atomicAdd(&ws.class_cnt[c], 1);
    }
  }
  if (lds) {
    __syncthreads();
    for (int c = threadIdx.x; c < class_num; c += 256)
      if (s_cnt[c] != 0) atomicAdd(&ws.class_cnt[c], s_cnt[c]);
  }
}

__global__ void nms_class_scan_kernel(int class_num, NmsWs ws, int want_mask) {
  // single thread: class_num is small (<= a few thousand)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int acc = 0, tiles = 0;
    long long words = 0;
    for (int c = 0; c < class_num; ++c) {
      const int nc = ws.class_cnt[c];
      ws.class_off[c] = acc;
      acc += nc;
      ws.tile_off[c] = tiles;
      if (want_mask && nc > 0 && nc <= NMS_MASK_MAX) {
        const int T = (nc + 63) >> 6;
        ws.mask_off[c] = words;
        words += (long long)nc * T;
        tiles += T * (T + 1) / 2;
      } else {
        ws.mask_off[c] = -1;
      }
    }
    ws.class_off[class_num] = acc;
    ws.tile_off[class_num] = tiles;
    ws.mask_off[class_num] = words;
  }
}

// rows grouped by class: slot = class_off[c] + (arrival order inside the class). The order inside a class is whatever the
// atomics give -- the rank below does not depend on it. A workgroup reserves its rows' slots of a class with ONE atomic.
__global__ __launch_bounds__(256) void nms_bucket_kernel(int n, int class_num, NmsWs ws) {
  __shared__ int s_cnt[NMS_LDS_CLASSES];
  const bool lds = class_num <= NMS_LDS_CLASSES;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int c = i < n ? ws.cls[i] : -1;
  int slot = -1;
  if (lds) {
    for (int k = threadIdx.x; k < class_num; k += 256) s_cnt[k] = 0;
    __syncthreads();
    const int local = c >= 0 ? atomicAdd(&s_cnt[c], 1) : 0;
    __syncthreads();
    for (int k = threadIdx.x; k < class_num; k += 256) {
      const int m = s_cnt[k];
      if (m != 0) s_cnt[k] = atomicAdd(&ws.fill[k], m);   // (now: the first of this workgroup's slots of class k)
    }
    __syncthreads();
    if (c >= 0) slot = ws.class_off[c] + s_cnt[c] + local;
  } else if (c >= 0) {
    slot = ws.class_off[c] + atomicAdd(&ws.fill[c], 1);
  }
  if (slot >= 0) {
    ws.bidx[slot] = i;
    ws.bscore[slot] = ws.score[i];
  }
}

// The order needs a TOTAL order on the scores: a NaN score compares false both ways, and a comparison sort that is handed
// such keys can leave its (-inf, -1) padding entries among the real rows (an out-of-bounds row index), a rank by counting
// colliding positions. np.argsort puts NaN LAST, so the reference's `argsort()[::-1]` (utils/tools.py:717) visits NaN rows
// FIRST: the key of a NaN score is +inf (ties with other NaN / +inf rows by index like every other tie). The score that is
// stored and reported stays the row's own.
__device__ __forceinline__ double nms_order_key(double s) { return s != s ? __builtin_huge_val() : s; }

// rank of every row inside its class by (score descending, ties: higher original index first -- np.argsort leaves ties
// undefined, utils/tools.py:717) = the number of rows of the class that come before it: one thread per bucket slot, the
// class's slots in tiles of 256 through LDS (a workgroup = 256 consecutive slots = one class, or the few small classes
// that meet in it: the tiles cover their union and a thread counts the entries of its own class). Sum n_c^2 comparisons
// instead of n^2; before the tiles every comparison was two dependent global loads: 677 us on 151 186 rows.
__global__ __launch_bounds__(256) void nms_rank_kernel(const double* __restrict__ rows, int n, int class_num, NmsWs ws) {
  __shared__ double s_sc[256];
  __shared__ int s_ix[256];
  __shared__ int s_lo, s_hi;
  const int total = ws.class_off[class_num];
  const int e0 = blockIdx.x * 256;
  if (e0 >= total) return;
  const int e = e0 + threadIdx.x;
  const bool live = e < total;
  const int i = live ? ws.bidx[e] : 0;
  const double si_raw = live ? ws.bscore[e] : 0.;
  const double si = nms_order_key(si_raw);
  const int ci = live ? ws.cls[i] : 0;
  const int beg = live ? ws.class_off[ci] : 0, end = live ? ws.class_off[ci + 1] : 0;
  if (threadIdx.x == 0) s_lo = beg;
  const int e_last = (e0 + 255 < total ? e0 + 255 : total - 1);
  if (e == e_last) s_hi = end;
  // (classes of up to NMS_MASK_MAX rows were sorted in LDS by nms_sort_kernel: only the larger ones are ranked here)
  const bool counted = live && (end - beg > NMS_MASK_MAX);
  if (!__syncthreads_or(counted ? 1 : 0)) return;
  const int lo = s_lo, hi = s_hi;
  int rank = 0;
  for (int j0 = lo; j0 < hi; j0 += 256) {
    __syncthreads();
    const int j = j0 + threadIdx.x;
    if (j < hi) {
      s_sc[threadIdx.x] = nms_order_key(ws.bscore[j]);
      s_ix[threadIdx.x] = ws.bidx[j];
    }
    __syncthreads();
    const int k_lo = beg > j0 ? beg - j0 : 0;
    const int k_hi = end < j0 + 256 ? end - j0 : 256;
    if (counted)
      for (int k = k_lo; k < k_hi; ++k) {
        const double sj = s_sc[k];      // (already a key: no NaN)
        const int jj = s_ix[k];
        rank += ((sj > si) || (sj == si && jj > i)) ? 1 : 0;
      }
  }
  if (!counted) return;
  const int pos = beg + rank;
  ws.sorted_idx[pos] = i;
  ws.pos_of[i] = pos;
  const double* r = rows + (long long)i * 7;
  ws.box[(long long)pos * 4 + 0] = r[0];
  ws.box[(long long)pos * 4 + 1] = r[1];
  ws.box[(long long)pos * 4 + 2] = r[2];
  ws.box[(long long)pos * 4 + 3] = r[3];
  ws.sscore[pos] = si_raw;
}

// The same order for classes of up to NMS_MASK_MAX rows: one workgroup per class, a bitonic sort of (score, row) in LDS
// (8192 keys: 91 passes of 4 compare-exchanges per thread -- the rank by counting made n_c comparisons per row: 644 us for
// classes of 3 557 rows).
__device__ __forceinline__ bool nms_before(double sa, int ia, double sb, int ib) {
  return (sa > sb) || (sa == sb && ia > ib);
}
__global__ __launch_bounds__(1024) void nms_sort_kernel(const double* __restrict__ rows, NmsWs ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char nms_smem[];
  double* s_sc = reinterpret_cast<double*>(nms_smem);                       // [N2]
  int* s_ix = reinterpret_cast<int*>(nms_smem + (size_t)NMS_MASK_MAX * 8);  // [N2]
  const int c = blockIdx.x;
  const int beg = ws.class_off[c], nc = ws.class_off[c + 1] - beg;
  if (nc <= 0 || nc > NMS_MASK_MAX) return;
  int N2 = 64;
  while (N2 < nc) N2 <<= 1;
  for (int t = threadIdx.x; t < N2; t += 1024) {
    s_sc[t] = t < nc ? nms_order_key(ws.bscore[beg + t]) : -__builtin_huge_val();   // padding sorts behind every row
    s_ix[t] = t < nc ? ws.bidx[beg + t] : -1;
  }
  __syncthreads();
  for (int k = 2; k <= N2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (N2 >> 1); t += 1024) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // bit log2(j) clear
        const int hi = lo | j;
        const double sa = s_sc[lo], sb = s_sc[hi];
        const int ia = s_ix[lo], ib = s_ix[hi];
        const bool fwd = (lo & k) == 0;               // this run is sorted "best first", the next one the other way
        const bool swap = fwd ? nms_before(sb, ib, sa, ia) : nms_before(sa, ia, sb, ib);
        if (swap) {
          s_sc[lo] = sb; s_sc[hi] = sa;
          s_ix[lo] = ib; s_ix[hi] = ia;
        }
      }
      __syncthreads();
    }
  for (int r = threadIdx.x; r < nc; r += 1024) {
    const int i = s_ix[r];
    if (i < 0) continue;      // (cannot happen with a total order: a padding entry among the first nc; never index with it)
    const int pos = beg + r;
    ws.sorted_idx[pos] = i;
    ws.pos_of[i] = pos;
    const double* src = rows + (long long)i * 7;
    ws.box[(long long)pos * 4 + 0] = src[0];
    ws.box[(long long)pos * 4 + 1] = src[1];
    ws.box[(long long)pos * 4 + 2] = src[2];
    ws.box[(long long)pos * 4 + 3] = src[3];
    ws.sscore[pos] = ws.score[i];   // the row's own score (s_sc holds the order key)
  }
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

// the same test with the box corners of `a` precomputed and the pairs that cannot pass a POSITIVE threshold rejected
// before their divisions: no overlap -> inter = 0 -> iou = +0 (the denominator is positive), and DIoU subtracts a
// non-negative (or NaN) term from it: `>= thr` is false either way. thr <= 0 takes the full arithmetic (thr_pos false).
struct BoxA {
  double x, y, w, h, minx, maxx, miny, maxy, area;
};
__device__ __forceinline__ BoxA box_a(const double* a) {
#pragma clang fp contract(off)
  const double hx = a[2] / 2., hy = a[3] / 2.;
  return BoxA{a[0], a[1], a[2], a[3], a[0] - hx, a[0] + hx, a[1] - hy, a[1] + hy, a[2] * a[3]};
}
__device__ __forceinline__ bool pair_suppresses(const BoxA& A, const double* b, bool diou, double thr, bool thr_pos) {
#pragma clang fp contract(off)
  const double bhx = b[2] / 2., bhy = b[3] / 2.;
  const double bminx = b[0] - bhx, bmaxx = b[0] + bhx, bminy = b[1] - bhy, bmaxy = b[1] + bhy;
  const double iw = fmax(fmin(bmaxx, A.maxx) - fmax(bminx, A.minx), 0.);
  const double ih = fmax(fmin(bmaxy, A.maxy) - fmax(bminy, A.miny), 0.);
  const double inter = iw * ih;
  if (thr_pos && inter == 0.) return false;
  const double pa = b[2] * b[3];
  const double uni = pa + A.area - inter;
  const double iou = inter / (uni + 1e-07);
  if (!diou) return iou >= thr;
  const double ewx = fmax(bmaxx, A.maxx) - fmin(bminx, A.minx);
  const double ewy = fmax(bmaxy, A.maxy) - fmin(bminy, A.miny);
  const double c2 = ewx * ewx + ewy * ewy;
  const double dx = A.x - b[0], dy = A.y - b[1];
  const double rho2 = dx * dx + dy * dy;
  return (iou - rho2 / c2) >= thr;
}

// One wave per 64 x 64 tile (tr <= tc) of a class's pair matrix; lane = row p = 64 tr + lane of the tile, the tile's 64
// column boxes sit in the wave's own 2 KB of LDS. Tiles are numbered class after class (NmsWs::tile_off), rows of tiles
// inside a class; the waves of a fixed grid stride over the list (the host does not know its length).
// TRANS (soft-NMS): the transposed matrix -- lane = column q of the tile, bits over the tile's rows p < q, word [q][tr] --
// so that a row of the matrix lists, in rank order, the earlier rows that decay q. IoU is symmetric in its operands bit for
// bit (min / max / the commutative sum of the two areas), so the roles of the two boxes may be swapped.
template <bool TRANS>
__global__ __launch_bounds__(256) void nms_mask_kernel(int class_num, NmsWs ws, double thr, int diou) {
  __shared__ double s_box[4][64 * 4];
  __shared__ int s_toff[NMS_LDS_CLASSES + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool lds_tab = class_num <= NMS_LDS_CLASSES;
  if (lds_tab)
    for (int c = threadIdx.x; c <= class_num; c += 256) s_toff[c] = ws.tile_off[c];
  __syncthreads();
  const int* toff = lds_tab ? s_toff : ws.tile_off;
  const int total = toff[class_num];
  const bool thr_pos = thr > 0.;
  double* sb = s_box[wave];
  for (int t = blockIdx.x * 4 + wave; t < total; t += gridDim.x * 4) {
    int lo = 0, hi = class_num - 1;        // the class whose tile range holds t (empty classes have empty ranges)
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (toff[mid] <= t) lo = mid; else hi = mid - 1;
    }
    const int c = lo;
    const int beg = ws.class_off[c], nc = ws.class_off[c + 1] - beg;
    const int T = (nc + 63) >> 6;
    const int u = t - toff[c];
    // u = tr * T - tr (tr - 1) / 2 + (tc - tr): tile row from the quadratic, corrected for the rounding of sqrt
    int tr = (int)(((double)(2 * T + 1) - sqrt((double)(2 * T + 1) * (double)(2 * T + 1) - 8.0 * (double)u)) * 0.5);
    if (tr < 0) tr = 0;
    if (tr > T - 1) tr = T - 1;
    while (tr > 0 && tr * T - tr * (tr - 1) / 2 > u) --tr;
    while (tr + 1 < T && (tr + 1) * T - (tr + 1) * tr / 2 <= u) ++tr;
    const int tc = tr + (u - (tr * T - tr * (tr - 1) / 2));
    const int t_mine = TRANS ? tc : tr, t_other = TRANS ? tr : tc;   // the tile index of the lanes' rows / of the LDS boxes
    const int o_mine = t_other * 64 + lane;
#pragma unroll
    for (int k = 0; k < 4; ++k) sb[lane * 4 + k] = o_mine < nc ? ws.box[(long long)(beg + o_mine) * 4 + k] : 0.;
    const int m = t_mine * 64 + lane;
    unsigned long long bits = 0;
    if (m < nc) {
      const double a4[4] = {ws.box[(long long)(beg + m) * 4], ws.box[(long long)(beg + m) * 4 + 1],
                            ws.box[(long long)(beg + m) * 4 + 2], ws.box[(long long)(beg + m) * 4 + 3]};
      const BoxA A = box_a(a4);
      const int jn = nc - t_other * 64 < 64 ? nc - t_other * 64 : 64;
      for (int j = 0; j < jn; ++j) {
        const double b4[4] = {sb[j * 4], sb[j * 4 + 1], sb[j * 4 + 2], sb[j * 4 + 3]};
        const int o = t_other * 64 + j;
        if ((TRANS ? o < m : o > m) && pair_suppresses(A, b4, diou != 0, thr, thr_pos)) bits |= 1ull << j;
      }
      ws.mask[ws.mask_off[c] + (long long)m * T + t_other] = bits;
    }
  }
}

// The walk of one class over its bit matrix. `rem` (bit q = row q is suppressed) lives in wave 0's registers: lane l holds
// words l and l + 64. Tile row tr (64 rows) is staged in LDS -- words tr .. T - 1 of each row -- by waves 1..3 while wave 0
// works on the tile row before it. Wave 0, per tile row: (1) the 64 x 64 diagonal block decides which of the 64 rows survive
// (they can suppress each other): lane r holds row r's diagonal word, the walk is 64 scalar steps of readlane / or;
// (2) every surviving row's remaining words are OR-ed into rem (independent LDS reads).
constexpr int NMS_SCAN_WAVES = 8;
__global__ __launch_bounds__(64 * NMS_SCAN_WAVES) void nms_scan_kernel(NmsWs ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char nms_smem[];
  unsigned long long* buf = reinterpret_cast<unsigned long long*>(nms_smem);   // [2][64][NMS_MASK_WORDS]
  const int c = blockIdx.x;
  const long long moff = ws.mask_off[c];
  if (moff < 0) return;
  const int beg = ws.class_off[c], nc = ws.class_off[c + 1] - beg;
  const int T = (nc + 63) >> 6;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long* M = ws.mask + moff;
  // words [tr, T) of rows 64 tr .. 64 tr + 63 -> buf[tr & 1]; a wave takes every nwaves-th row and has ALL its loads in
  // flight before it stores the first (a row is at most two words per lane)
  auto stage = [&](int tr, int first_wave, int nwaves) {
    unsigned long long* dst = buf + (size_t)(tr & 1) * 64 * NMS_MASK_WORDS;
    constexpr int RMAX = 64;   // (nwaves >= 1)
    unsigned long long v0[10], v1[10];
    for (int r0 = wave - first_wave; r0 < RMAX; r0 += nwaves * 10) {
#pragma unroll
      for (int u = 0; u < 10; ++u) {
        const int r = r0 + u * nwaves;
        const int p = tr * 64 + r;
        const int w0 = tr + lane, w1 = tr + lane + 64;
        const bool ok = r < 64 && p < nc;
        v0[u] = (ok && w0 < T) ? M[(long long)p * T + w0] : 0ull;
        v1[u] = (ok && w1 < T) ? M[(long long)p * T + w1] : 0ull;
      }
#pragma unroll
      for (int u = 0; u < 10; ++u) {
        const int r = r0 + u * nwaves;
        if (r < 64) {
          dst[r * NMS_MASK_WORDS + lane] = v0[u];
          if (tr + lane + 64 < T) dst[r * NMS_MASK_WORDS + lane + 64] = v1[u];
        }
      }
    }
  };
  stage(0, 0, NMS_SCAN_WAVES);
  __syncthreads();
  // the loader waves (1 .. 7, rows wave - 1 + 7 u) run ONE tile row further ahead in registers: the loads of tile row tr + 2
  // are issued before the barrier that ends iteration tr and land while wave 0 works on tr + 1 (a stage's latency otherwise
  // sat between two barriers: 5.6 us per tile row where wave 0 needs 1.7)
  static_assert((NMS_SCAN_WAVES - 1) * 10 >= 64, "one batch of 10 rows per loader wave covers a tile row");
  unsigned long long pv0[10], pv1[10];
  auto fetch = [&](int tr) {
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      const int r = wave - 1 + u * (NMS_SCAN_WAVES - 1);
      const int p = tr * 64 + r;
      const int w0 = tr + lane, w1 = tr + lane + 64;
      const bool ok = tr < T && r < 64 && p < nc;
      pv0[u] = (ok && w0 < T) ? M[(long long)p * T + w0] : 0ull;
      pv1[u] = (ok && w1 < T) ? M[(long long)p * T + w1] : 0ull;
    }
  };
  auto put = [&](int tr) {
    unsigned long long* dst = buf + (size_t)(tr & 1) * 64 * NMS_MASK_WORDS;
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      const int r = wave - 1 + u * (NMS_SCAN_WAVES - 1);
      if (r < 64) {
        dst[r * NMS_MASK_WORDS + lane] = pv0[u];
        if (tr + lane + 64 < T) dst[r * NMS_MASK_WORDS + lane + 64] = pv1[u];
      }
    }
  };
  if (wave != 0) fetch(1);
  unsigned long long rem0 = 0, rem1 = 0;
  for (int tr = 0; tr < T; ++tr) {
    if (wave != 0) {
      if (tr + 1 < T) {
        put(tr + 1);
        fetch(tr + 2);
      }
    } else {
      const unsigned long long* src = buf + (size_t)(tr & 1) * 64 * NMS_MASK_WORDS;
      const int rows_here = nc - tr * 64 < 64 ? nc - tr * 64 : 64;
      const unsigned long long diag = lane < rows_here ? src[lane * NMS_MASK_WORDS] : 0ull;
      // the removed word of this tile row so far (owner: lane tr & 63, register tr >> 6)
      const unsigned long long mine = (tr < 64) ? rem0 : rem1;
      const int owner = tr & 63;
      unsigned long long cur = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(mine >> 32), owner) << 32) |
                               (unsigned)__builtin_amdgcn_readlane((int)(mine & 0xffffffffull), owner);
      unsigned long long kept = 0;
      for (int r = 0; r < rows_here; ++r) {
        if (!((cur >> r) & 1ull)) {
          kept |= 1ull << r;
          cur |= ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(diag >> 32), r) << 32) |
                 (unsigned)__builtin_amdgcn_readlane((int)(diag & 0xffffffffull), r);
        }
      }
      // the surviving rows' words right of the diagonal: lane l takes words tr + l (l >= 1) and tr + l + 64 of EVERY row of
      // the tile row -- unconditional, independent LDS reads (staged rows beyond the class hold zeros) -- and keeps those
      // of the survivors
      const bool use0 = lane > 0 && tr + lane < T, use1 = tr + lane + 64 < T;
      unsigned long long acc0 = 0, acc1 = 0;
      if (use0 || use1) {
#pragma unroll 16
        for (int r = 0; r < 64; ++r) {
          const unsigned long long sel = 0ull - ((kept >> r) & 1ull);
          acc0 |= sel & src[r * NMS_MASK_WORDS + lane];
          if (use1) acc1 |= sel & src[r * NMS_MASK_WORDS + lane + 64];
        }
      }
      // word tr + lane of rem lives in lane (tr + lane) & 63, register (tr + lane) >> 6: rotate the sums to their owners
      // through LDS-free lane indexing -- every lane instead fetches the sum of the lane that holds ITS words
      {
        const int src_lane0 = (lane - tr) & 63;            // the lane whose acc covers word `lane` (if lane > tr) ...
        const unsigned long long a0 = __shfl(use0 ? acc0 : 0ull, src_lane0, 64);
        const unsigned long long a1 = __shfl(acc1, src_lane0, 64);
        // word w = lane (register 0): covered by acc0 of lane w - tr if 0 < w - tr < 64, by acc1 of lane w - tr - 64 ... (w < 64 <= tr + 64 always false)
        if (lane > tr) rem0 |= a0;
        // word w = lane + 64 (register 1): w - tr in (0, 64) -> acc0 of lane w - tr = lane + 64 - tr; w - tr >= 64 -> acc1 of lane w - tr - 64
        const int d = lane + 64 - tr;                       // distance of word lane + 64 from the diagonal word
        const unsigned long long b0 = __shfl(use0 ? acc0 : 0ull, d & 63, 64);
        if (d > 0 && d < 64) rem1 |= b0;
        else if (d >= 64) rem1 |= a1;
        (void)a1;
      }
      if (lane == owner) {
        if (tr < 64) rem0 = cur; else rem1 = cur;
      }
    }
    __syncthreads();
  }
  // bits -> the per-row flags the other kernels read
  unsigned long long* words = buf;
  if (wave == 0) {
    words[lane] = rem0;
    words[lane + 64] = rem1;
  }
  __syncthreads();
  for (int p = threadIdx.x; p < nc; p += 64 * NMS_SCAN_WAVES)
    ws.removed[beg + p] = (unsigned char)((words[p >> 6] >> (p & 63)) & 1ull);
}

// classes too large for the bit matrix (more than NMS_MASK_MAX rows): the greedy walk itself, one workgroup per class,
// flags in global memory; a barrier only behind a row that was visited (a suppressed row changes nothing), the next
// surviving row found 64 flags at a time
__global__ __launch_bounds__(1024) void nms_walk_kernel(NmsWs ws, double thr, int diou, int force) {
  const int c = blockIdx.x;
  if (ws.mask_off[c] >= 0 && !force) return;
  const int beg = ws.class_off[c], end = ws.class_off[c + 1];
  const int lane = threadIdx.x & 63;
  const bool thr_pos = thr > 0.;
  int p = beg;
  while (p < end) {
    {
      const double a4[4] = {ws.box[(long long)p * 4], ws.box[(long long)p * 4 + 1], ws.box[(long long)p * 4 + 2],
                            ws.box[(long long)p * 4 + 3]};
      const BoxA A = box_a(a4);
      for (int q = p + 1 + threadIdx.x; q < end; q += 1024) {
        if (ws.removed[q]) continue;
        const double b[4] = {ws.box[(long long)q * 4], ws.box[(long long)q * 4 + 1], ws.box[(long long)q * 4 + 2],
                             ws.box[(long long)q * 4 + 3]};
        if (pair_suppresses(A, b, diou != 0, thr, thr_pos)) ws.removed[q] = 1;
      }
    }
    __syncthreads();   // (workgroup-scope: the flags written above are visible to every wave of this workgroup)
    int nxt = p + 1;
    while (nxt < end) {   // every wave finds the same row: flags at or before it no longer change
      const int idx = nxt + lane;
      const unsigned long long m = __ballot(idx < end && !ws.removed[idx]);
      if (m) {
        nxt += __builtin_ctzll(m);
        break;
      }
      nxt += 64;
    }
    p = nxt;
    __syncthreads();   // nobody starts writing flags of the next step while a wave still scans
  }
}

// soft-NMS: row q is deleted iff its score, decayed in rank order by every earlier row of its
// class with IoU >= thr, falls below conf_threshold after some decay. One thread per row; the earlier rows' boxes in
// tiles of 256 through LDS (union of the classes that meet in the workgroup, as in nms_rank_kernel).
// Classes of up to NMS_MASK_MAX rows (round 5): the pairs with IoU >= thr come from the transposed bit matrix
// (nms_mask_kernel<true>), so a row only recomputes the IoU of the few earlier rows that actually decay it -- in rank order,
// with the arithmetic of the loop below (the product of the decays is rounded in that order).
__global__ __launch_bounds__(256) void nms_soft_mask_kernel(int class_num, NmsWs ws, double thr, double conf_thr, double sigma) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= ws.class_off[class_num]) return;
  const int c = ws.cls[ws.sorted_idx[q]];
  const long long moff = ws.mask_off[c];
  if (moff < 0) return;     // (nms_soft_kernel below takes the larger classes)
  const int beg = ws.class_off[c], nc = ws.class_off[c + 1] - beg;
  const int T = (nc + 63) >> 6;
  const int ql = q - beg;
  const double b[4] = {ws.box[(long long)q * 4], ws.box[(long long)q * 4 + 1], ws.box[(long long)q * 4 + 2],
                       ws.box[(long long)q * 4 + 3]};
  double conf = ws.sscore[q];
  bool del = false;
  const unsigned long long* row = ws.mask + moff + (long long)ql * T;
  for (int tr = 0; tr <= (ql >> 6); ++tr) {
    unsigned long long w = row[tr];
    while (w) {
      const int j = __builtin_ctzll(w);
      w &= w - 1;
      const long long p = beg + tr * 64 + j;
      const double a[4] = {ws.box[p * 4], ws.box[p * 4 + 1], ws.box[p * 4 + 2], ws.box[p * 4 + 3]};
      const double iou = pair_score(a, b, false);
      if (iou >= thr) {     // (what the bit says; kept so that the two kernels are the same text)
        conf *= exp(-1. * (iou * iou) / sigma);
        if (conf < conf_thr) del = true;
      }
    }
  }
  ws.removed[q] = del ? 1 : 0;
}

__global__ __launch_bounds__(256) void nms_soft_kernel(int n, int class_num, NmsWs ws, double thr, double conf_thr, double sigma) {
  __shared__ double s_box[256 * 4];
  __shared__ int s_lo;
  const int total = ws.class_off[class_num];
  const int q0 = blockIdx.x * 256;
  if (q0 >= total) return;   // rows with an out-of-range class are not ranked
  const int q = q0 + threadIdx.x;
  const bool live = q < total;
  const int i = live ? ws.sorted_idx[q] : 0;
  const int beg = live ? ws.class_off[ws.cls[i]] : 0;
  if (threadIdx.x == 0) s_lo = beg;
  // (rows of classes that have a bit matrix were decided by nms_soft_mask_kernel)
  const bool mine = live && ws.mask_off[ws.cls[i]] < 0;
  if (!__syncthreads_or(mine ? 1 : 0)) return;
  const int lo = s_lo;
  const int hi = (q0 + 255 < total ? q0 + 255 : total - 1);   // the last row of the workgroup needs rows < hi
  double b[4] = {0., 0., 0., 0.};
  if (live)
    for (int k = 0; k < 4; ++k) b[k] = ws.box[(long long)q * 4 + k];
  double conf = live ? ws.sscore[q] : 0.;
  bool del = false;
  const bool thr_pos = thr > 0.;
  for (int p0 = lo; p0 < hi; p0 += 256) {
    __syncthreads();
    const int pp = p0 + threadIdx.x;
    if (pp < hi)
      for (int k = 0; k < 4; ++k) s_box[threadIdx.x * 4 + k] = ws.box[(long long)pp * 4 + k];
    __syncthreads();
    const int k_lo = beg > p0 ? beg - p0 : 0;
    const int k_hi = q < p0 + 256 ? q - p0 : 256;
    if (mine)
      for (int k = k_lo; k < k_hi; ++k) {
        const double a[4] = {s_box[k * 4], s_box[k * 4 + 1], s_box[k * 4 + 2], s_box[k * 4 + 3]};
        // (boxes that do not overlap have IoU +0: below any positive threshold, before the divisions)
        if (thr_pos && !(fmin(b[0] + b[2] / 2., a[0] + a[2] / 2.) > fmax(b[0] - b[2] / 2., a[0] - a[2] / 2.) &&
                         fmin(b[1] + b[3] / 2., a[1] + a[3] / 2.) > fmax(b[1] - b[3] / 2., a[1] - a[3] / 2.)))
          continue;
        const double iou = pair_score(a, b, false);
        if (iou >= thr) {
          conf *= exp(-1. * (iou * iou) / sigma);
          if (conf < conf_thr) del = true;
        }
      }
  }
  if (mine) ws.removed[q] = del ? 1 : 0;
}

__global__ void nms_finish_kernel(int n, NmsWs ws, unsigned char* __restrict__ keep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int pos = ws.pos_of[i];
  keep[i] = (pos >= 0 && !ws.removed[pos]) ? 1 : 0;
}

// ---- kept rows in the reference's output order: classes ascending, original order inside a class (utils/tools.py:730-732) ----
__global__ __launch_bounds__(256) void nms_kept_count_kernel(int n, int class_num, NmsWs ws, const unsigned char* __restrict__ keep) {
  __shared__ int s_cnt[NMS_LDS_CLASSES];
  const bool lds = class_num <= NMS_LDS_CLASSES;
  if (lds)
    for (int c = threadIdx.x; c < class_num; c += 256) s_cnt[c] = 0;
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n && keep[i]) {     // (fill was zeroed again after the bucket kernel used it)
    if (lds) atomicAdd(&s_cnt[ws.cls[i]], 1);
    else atomicAdd(&ws.fill[ws.cls[i]], 1);
  }
  if (lds) {
    __syncthreads();
    for (int c = threadIdx.x; c < class_num; c += 256)
      if (s_cnt[c] != 0) atomicAdd(&ws.fill[c], s_cnt[c]);
  }
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
// (tiles of 256 slots through LDS, as nms_rank_kernel)
__global__ __launch_bounds__(256) void nms_gather_kernel(const double* __restrict__ rows, int class_num, NmsWs ws,
                                                         const unsigned char* __restrict__ keep, double* __restrict__ out) {
  __shared__ int s_ix[256];      // original index of a kept row, -1 otherwise
  __shared__ int s_lo, s_hi;
  const int total = ws.class_off[class_num];
  const int e0 = blockIdx.x * 256;
  if (e0 >= total) return;
  const int e = e0 + threadIdx.x;
  const bool live = e < total;
  const int i = live ? ws.bidx[e] : 0;
  const int c = live ? ws.cls[i] : 0;
  const int beg = live ? ws.class_off[c] : 0, end = live ? ws.class_off[c + 1] : 0;
  const bool mine = live && keep[i];
  if (threadIdx.x == 0) s_lo = beg;
  const int e_last = (e0 + 255 < total ? e0 + 255 : total - 1);
  if (e == e_last) s_hi = end;
  __syncthreads();
  const int lo = s_lo, hi = s_hi;
  int r = 0;
  for (int j0 = lo; j0 < hi; j0 += 256) {
    __syncthreads();
    const int j = j0 + threadIdx.x;
    if (j < hi) {
      const int jj = ws.bidx[j];
      s_ix[threadIdx.x] = keep[jj] ? jj : -1;
    }
    __syncthreads();
    const int k_lo = beg > j0 ? beg - j0 : 0;
    const int k_hi = end < j0 + 256 ? end - j0 : 256;
    if (mine)
      for (int k = k_lo; k < k_hi; ++k) {
        const int jj = s_ix[k];
        r += (jj >= 0 && jj < i) ? 1 : 0;
      }
  }
  if (!mine) return;
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
  init_options();
  const int force_walk = g_opt[OPT_NMS_WALK];   // yolo_set_option key 7 (tests): no bit matrices -- the walk / tiled kernels for every class
  hipLaunchKernelGGL(nms_prepare_kernel, dim3(nb), dim3(256), 0, st, rows, n, class_num, ws);
  hipLaunchKernelGGL(nms_class_scan_kernel, dim3(1), dim3(64), 0, st, class_num, ws, force_walk ? 0 : 1);
  // workgroups of the pair-matrix kernel: tiles <= sum over classes of T (T + 1) / 2 <= (n / 64 + class_num) (T_max + 1) / 2
  long long mask_grid = 1;
  {
    static int cus = 0;
    if (cus == 0) {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    }
    const long long t_max = (n < NMS_MASK_MAX ? n : NMS_MASK_MAX) / 64 + 1;
    const long long tiles_bound = ((long long)n / 64 + class_num) * (t_max + 1) / 2;
    mask_grid = (tiles_bound + 3) / 4;
    if (mask_grid > 8LL * cus) mask_grid = 8LL * cus;
    if (mask_grid < 1) mask_grid = 1;
  }
  hipLaunchKernelGGL(nms_bucket_kernel, dim3(nb), dim3(256), 0, st, n, class_num, ws);
  {   // order inside the classes: LDS sort (classes of up to NMS_MASK_MAX rows), rank by counting (the larger ones)
    constexpr size_t lds_sort = (size_t)NMS_MASK_MAX * 12;
    static bool attr_sort = false;
    if (!attr_sort) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nms_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sort);
      attr_sort = true;
    }
    hipLaunchKernelGGL(nms_sort_kernel, dim3(class_num), dim3(1024), lds_sort, st, rows, ws);
    hipLaunchKernelGGL(nms_rank_kernel, dim3(nb), dim3(256), 0, st, rows, n, class_num, ws);
  }
  if (mode == YOLO_NMS_SOFT) {
    if (!force_walk) {
      hipLaunchKernelGGL(nms_mask_kernel<true>, dim3((unsigned)mask_grid), dim3(256), 0, st, class_num, ws, nms_threshold, 0);
      hipLaunchKernelGGL(nms_soft_mask_kernel, dim3(nb), dim3(256), 0, st, class_num, ws, nms_threshold, conf_threshold, sigma);
    }
    hipLaunchKernelGGL(nms_soft_kernel, dim3(nb), dim3(256), 0, st, n, class_num, ws, nms_threshold, conf_threshold, sigma);
  } else {
    if (!force_walk) {
      // pair tests of every class by the whole chip, then one workgroup per class walks its bit matrix
      hipLaunchKernelGGL(nms_mask_kernel<false>, dim3((unsigned)mask_grid), dim3(256), 0, st, class_num, ws, nms_threshold,
                         mode == YOLO_NMS_DIOU ? 1 : 0);
      constexpr size_t lds = (size_t)2 * 64 * NMS_MASK_WORDS * 8;
      static bool attr_set = false;
      if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nms_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
      }
      hipLaunchKernelGGL(nms_scan_kernel, dim3(class_num), dim3(64 * NMS_SCAN_WAVES), lds, st, ws);
    }
    hipLaunchKernelGGL(nms_walk_kernel, dim3(class_num), dim3(1024), 0, st, ws, nms_threshold, mode == YOLO_NMS_DIOU ? 1 : 0,
                       force_walk);
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
  hipLaunchKernelGGL(nms_kept_count_kernel, dim3(nb), dim3(256), 0, st, n, class_num, ws, keep_out);
  hipLaunchKernelGGL(nms_kept_scan_kernel, dim3(1), dim3(64), 0, st, class_num, ws, count_out);
  hipLaunchKernelGGL(nms_gather_kernel, dim3(nb), dim3(256), 0, st, rows, class_num, ws, keep_out, rows_out);
  return check_launch("nms gather kernels");
}

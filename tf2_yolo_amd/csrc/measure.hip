// Detection evaluation on the device: the arithmetic of the reference's utils/measurement.py
// (create_score_mat :16-150, PRfunc :153-337) after decode / NMS -- matching detections to ground truth
// by IoU, counting true positives, and the cumulative precision / recall curve. Everything is fp64 and
// integer work, bit-exact against the reference's own outputs (tests/golden/measurement_golden.npz).
//
//   yolo_match_detections  per image: for every detection the best-IoU ground-truth box of ITS class
//                          (np.argmax: first maximum, index inside the class subset) and IoU >= threshold;
//                          per class: #detections, #ground truths, #matched detections (TPP), #distinct
//                          matched ground truths (TP)                               (measurement.py:104-138)
//   yolo_rank_desc         rank of every row inside its segment by key, descending (np.argsort(...)[::-1];
//                          ties, which numpy leaves to its unstable sort: the later row first)
//   yolo_pr_curve          one class: sort detections by joint confidence, then for every prefix the number
//                          of matched detections and of DISTINCT matched ground truths -> precision (3 modes)
//                          and recall, plus the closing (0, last recall) point      (measurement.py:299-323)
#include "common.hpp"

namespace yolo {

// utils/tools.py:630-684 (mode 1) in fp64; a = ground truth, b = prediction
__device__ __forceinline__ double iou_xywh(const double* a, const double* b) {
  const double ahx = a[2] / 2., ahy = a[3] / 2., bhx = b[2] / 2., bhy = b[3] / 2.;
  const double aminx = a[0] - ahx, amaxx = a[0] + ahx, aminy = a[1] - ahy, amaxy = a[1] + ahy;
  const double bminx = b[0] - bhx, bmaxx = b[0] + bhx, bminy = b[1] - bhy, bmaxy = b[1] + bhy;
  const double iw = fmax(fmin(bmaxx, amaxx) - fmax(bminx, aminx), 0.);
  const double ih = fmax(fmin(bmaxy, amaxy) - fmax(bminy, aminy), 0.);
  const double inter = iw * ih;
  const double uni = b[2] * b[3] + a[2] * a[3] - inter;
  return inter / (uni + 1e-07);
}

// one thread per detection, the ground truths of an image are few
__global__ void match_kernel(const double* __restrict__ gt, int ngt, const double* __restrict__ det, int ndet,
                             double thr, int* __restrict__ best_gt, unsigned char* __restrict__ matched,
                             double* __restrict__ best_iou) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ndet) return;
  const double* d = det + (long long)i * 7;
  const int c = (int)d[5];
  double best = -1.0;
  int arg = 0, k = 0;
  for (int j = 0; j < ngt; ++j) {
    const double* g = gt + (long long)j * 7;
    if ((int)g[5] != c) continue;
    const double v = iou_xywh(g, d);
    if (v > best) {   // strict: the first maximum wins, like np.argmax
      best = v;
      arg = k;
    }
    ++k;
  }
  best_gt[i] = (k > 0) ? arg : 0;
  matched[i] = (k > 0 && best >= thr) ? 1 : 0;
  best_iou[i] = (k > 0) ? best : 0.0;
}

// counts[c] = {detections, ground truths, matched detections, distinct matched ground truths} of class c
// in this image; one workgroup, flags = scratch of ngt bytes (ground truth j of class c <-> its index in
// the class subset)
__global__ __launch_bounds__(256) void match_counts_kernel(const double* __restrict__ gt, int ngt,
                                                         const double* __restrict__ det, int ndet,
                                                         const int* __restrict__ best_gt,
                                                         const unsigned char* __restrict__ matched, int class_num,
                                                         unsigned char* __restrict__ flags,
                                                         long long* __restrict__ counts) {
  for (int j = threadIdx.x; j < ngt; j += 256) flags[j] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < ndet; i += 256) {
    const int c = (int)det[(long long)i * 7 + 5];
    if (c < 0 || c >= class_num) continue;
    atomicAdd((unsigned long long*)&counts[c * 4 + 0], 1ULL);
    if (matched[i]) {
      atomicAdd((unsigned long long*)&counts[c * 4 + 2], 1ULL);
      // locate the best_gt[i]-th ground truth of class c
      int k = 0;
      for (int j = 0; j < ngt; ++j) {
        if ((int)gt[(long long)j * 7 + 5] != c) continue;
        if (k == best_gt[i]) {
          flags[j] = 1;
          break;
        }
        ++k;
      }
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < ngt; j += 256) {
    const int c = (int)gt[(long long)j * 7 + 5];
    if (c < 0 || c >= class_num) continue;
    atomicAdd((unsigned long long*)&counts[c * 4 + 1], 1ULL);
    if (flags[j]) atomicAdd((unsigned long long*)&counts[c * 4 + 3], 1ULL);
  }
}

__global__ __launch_bounds__(256) void rank_desc_kernel(const double* __restrict__ key, const int* __restrict__ seg,
                                                      int n, int* __restrict__ rank) {
  __shared__ double s_key[256];
  __shared__ int s_seg[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool valid = i < n;
  const double ki = valid ? key[i] : 0.0;
  const int si = valid ? (seg ? seg[i] : 0) : -2;
  int r = 0;
  for (int base = 0; base < n; base += 256) {
    const int j = base + threadIdx.x;
    s_key[threadIdx.x] = (j < n) ? key[j] : 0.0;
    s_seg[threadIdx.x] = (j < n) ? (seg ? seg[j] : 0) : -3;
    __syncthreads();
    const int lim = (n - base < 256) ? (n - base) : 256;
    for (int t = 0; t < lim; ++t) {
      const int j2 = base + t;
      const bool before = (s_key[t] > ki) || (s_key[t] == ki && j2 > i);
      r += (s_seg[t] == si && before) ? 1 : 0;
    }
    __syncthreads();
  }
  if (valid) rank[i] = r;
}

// first_rank[g] = smallest rank among the MATCHED detections whose ground truth is g
__global__ void pr_first_kernel(const int* __restrict__ rank, const int* __restrict__ gid,
                                const unsigned char* __restrict__ matched, int n, int num_gts,
                                int* __restrict__ first_rank, unsigned char* __restrict__ sorted_flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (matched[i] && gid[i] >= 0 && gid[i] < num_gts) atomicMin(&first_rank[gid[i]], rank[i]);
  sorted_flags[rank[i]] = matched[i] ? 1 : 0;   // bit 0: matched (bit 1 = first of its ground truth, set below)
}
__global__ void pr_mark_kernel(const int* __restrict__ rank, const int* __restrict__ gid,
                               const unsigned char* __restrict__ matched, int n, int num_gts,
                               const int* __restrict__ first_rank, unsigned char* __restrict__ sorted_flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (matched[i] && gid[i] >= 0 && gid[i] < num_gts && first_rank[gid[i]] == rank[i]) sorted_flags[rank[i]] |= 2;
}

// one workgroup walks the sorted flags in chunks of 1024 with a running (tpp, tp) pair
__global__ __launch_bounds__(1024) void pr_scan_kernel(const unsigned char* __restrict__ sorted_flags, int n,
                                                      int num_gts, int mode, double* __restrict__ precision,
                                                      double* __restrict__ recall) {
  __shared__ int s_a[1024], s_b[1024];
  __shared__ int run_a, run_b;
  if (threadIdx.x == 0) {
    run_a = 0;
    run_b = 0;
  }
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    const unsigned char f = (i < n) ? sorted_flags[i] : 0;
    s_a[threadIdx.x] = f & 1;
    s_b[threadIdx.x] = (f >> 1) & 1;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {   // Hillis-Steele inclusive scan
      const int va = (threadIdx.x >= o) ? s_a[threadIdx.x - o] : 0;
      const int vb = (threadIdx.x >= o) ? s_b[threadIdx.x - o] : 0;
      __syncthreads();
      s_a[threadIdx.x] += va;
      s_b[threadIdx.x] += vb;
      __syncthreads();
    }
    if (i < n) {
      const long long tpp = run_a + s_a[threadIdx.x], tp = run_b + s_b[threadIdx.x];
      const long long dets = i + 1, fp = dets - tpp;
      double p;
      if (mode == 0) p = (double)tpp / (double)dets;
      else if (mode == 1) p = (double)tp / (double)(tp + fp);
      else p = (double)tp / (double)dets;
      precision[i] = p;
      recall[i] = (double)tp / (double)num_gts;
      if (i == n - 1) {
        precision[n] = 0.0;
        recall[n] = (double)tp / (double)num_gts;
      }
    }
    __syncthreads();
    if (threadIdx.x == 1023) {
      run_a += s_a[1023];
      run_b += s_b[1023];
    }
    __syncthreads();
  }
}

__global__ void fill_int_kernel(int* p, int n, int v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace yolo

using namespace yolo;

extern "C" int yolo_match_detections(const double* gt_rows, int ngt, const double* det_rows, int ndet, int class_num,
                                     double iou_threshold, int* best_gt, unsigned char* matched, double* best_iou,
                                     unsigned char* gt_flags, long long* class_counts, void* stream) {
  YOLO_REQUIRE(ngt >= 0 && ndet >= 0 && class_num > 0 && class_counts, "match_detections: bad args");
  YOLO_REQUIRE(ndet == 0 || (det_rows && best_gt && matched && best_iou), "match_detections: null detection buffers");
  YOLO_REQUIRE(ngt == 0 || (gt_rows && gt_flags), "match_detections: null ground-truth buffers");
  hipStream_t st = as_stream(stream);
  if (ndet > 0) {
    hipLaunchKernelGGL(match_kernel, dim3((ndet + 255) / 256), dim3(256), 0, st, gt_rows, ngt, det_rows, ndet,
                       iou_threshold, best_gt, matched, best_iou);
    if (int rc = check_launch("match_kernel")) return rc;
  }
  hipLaunchKernelGGL(match_counts_kernel, dim3(1), dim3(256), 0, st, gt_rows, ngt, det_rows, ndet, best_gt, matched,
                     class_num, gt_flags, class_counts);
  return check_launch("match_counts_kernel");
}

extern "C" int yolo_rank_desc(const double* key, const int* segment, int n, int* rank, void* stream) {
  YOLO_REQUIRE(n >= 0, "rank_desc: bad args");
  if (n == 0) return YOLO_OK;
  YOLO_REQUIRE(key && rank, "rank_desc: null pointer");
  hipLaunchKernelGGL(rank_desc_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), key, segment, n, rank);
  return check_launch("rank_desc_kernel");
}

extern "C" size_t yolo_pr_curve_workspace_bytes(int n, int num_gts) {
  if (n < 0) n = 0;
  if (num_gts < 0) num_gts = 0;
  return (size_t)(n + num_gts + 2) * sizeof(int) + (size_t)n + 256;
}

extern "C" int yolo_pr_curve(const double* joint, const int* gt_id, const unsigned char* matched, int n, int num_gts,
                             int precision_mode, void* workspace, size_t workspace_bytes, double* precision,
                             double* recall, void* stream) {
  YOLO_REQUIRE(n > 0 && num_gts > 0, "pr_curve: needs at least one detection and one ground truth");
  YOLO_REQUIRE(joint && gt_id && matched && workspace && precision && recall, "pr_curve: null pointer");
  YOLO_REQUIRE(precision_mode >= 0 && precision_mode <= 2, "pr_curve: bad precision mode %d", precision_mode);
  YOLO_REQUIRE(workspace_bytes >= yolo_pr_curve_workspace_bytes(n, num_gts), "pr_curve: workspace too small");
  hipStream_t st = as_stream(stream);
  int* rank = reinterpret_cast<int*>(workspace);
  int* first_rank = rank + n;
  unsigned char* flags = reinterpret_cast<unsigned char*>(first_rank + num_gts + 2);
  hipLaunchKernelGGL(rank_desc_kernel, dim3((n + 255) / 256), dim3(256), 0, st, joint, (const int*)nullptr, n, rank);
  hipLaunchKernelGGL(fill_int_kernel, dim3((num_gts + 255) / 256), dim3(256), 0, st, first_rank, num_gts, 0x7fffffff);
  hipLaunchKernelGGL(pr_first_kernel, dim3((n + 255) / 256), dim3(256), 0, st, rank, gt_id, matched, n, num_gts,
                     first_rank, flags);
  hipLaunchKernelGGL(pr_mark_kernel, dim3((n + 255) / 256), dim3(256), 0, st, rank, gt_id, matched, n, num_gts,
                     first_rank, flags);
  hipLaunchKernelGGL(pr_scan_kernel, dim3(1), dim3(1024), 0, st, flags, n, num_gts, precision_mode, precision, recall);
  return check_launch("pr_curve");
}

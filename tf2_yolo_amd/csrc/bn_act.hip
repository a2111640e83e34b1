// Training / inference BatchNormalization fused with the activation (LeakyReLU 0.1 / Mish)
// and the residual Add. HBM-bound streaming kernels: 16 B per lane, grid-stride, per-channel
// statistics accumulated in fp64 so that the 1e-4 parity bar holds through 72 layers.
//
// Replaces BatchNormalization + LeakyReLU / Mish (+ Add) of
//   yolov3/models/backbone.py:39-71, yolov4/models/backbone.py:22-37,76-123,
//   yolov{1_5,2}/models/backbone.py:9-18.
#include "planes.hpp"
#include "act.hpp"
#include <type_traits>

namespace yolo {

// All per-channel reductions spread their fp64 atomics over YOLO_BN_STAT_SLOTS replicas of the [NQ][C]
// result (replica = blockIdx.x mod SLOTS): 1024 blocks hammering the same 2C addresses ran at the
// contended-atomic rate (MI355X_MICROARCH.md: 14x slower) and cost more than the streaming pass itself.
// Column layout shared by the per-channel reductions: a block covers `cw` float4 columns
// (cw = min(C/4, 256)) x (256/cw) pixel rows per pass; grid.y walks column chunks.
struct ColGeom {
  int cw, rpp;
};
static inline ColGeom col_geom(int C4) {
  ColGeom g;
  g.cw = C4 < 256 ? C4 : 256;
  g.rpp = 256 / g.cw;
  return g;
}

// inference: bound of act(scale*y + shift) from the conv epilogue's per-channel max|y| (one workgroup)
__global__ __launch_bounds__(256) void bn_infer_bound_kernel(int C, const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             const unsigned* __restrict__ absmax,
                                                             unsigned* __restrict__ bound) {
  __shared__ float s_max[4];
  float b = 0.f;
  for (int c = threadIdx.x; c < C; c += 256)
    b = fmaxf(b, fabsf(scale[c]) * __builtin_bit_cast(float, absmax[c]) * 1.001f + fabsf(shift[c]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) b = fmaxf(b, __shfl_xor(b, o, 64));
  if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = b;
  __syncthreads();
  if (threadIdx.x == 0)
    bound[0] = __builtin_bit_cast(unsigned, fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3])) + 1e-30f);
}

template <int NQ, bool EXCLUSIVE = false>
__device__ __forceinline__ void block_col_reduce(double (&v)[NQ][4], int cw, int rpp, int row_lane, int col,
                                                 bool active, double* smem /* [NQ*4][256] */,
                                                 double* out /* [NQ][C] */, int C, int c4) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) smem[(q * 4 + e) * 256 + tid] = active ? v[q][e] : 0.0;
  __syncthreads();
  if (active && row_lane == 0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double s = 0.0;
        for (int r = 0; r < rpp; ++r) s += smem[(q * 4 + e) * 256 + r * cw + col];
        if (EXCLUSIVE)   // slot blockIdx.x belongs to this workgroup alone (gridDim.x <= YOLO_BN_RED_SLOTS)
          // (agent-scope store = written through to the memory side: the in-launch fold of bn_bwd_reduce_kernel<true> reads
          // the slots from workgroups on other XCDs without any cache-wide fence)
          __hip_atomic_store(&out[(long long)blockIdx.x * NQ * C + (long long)q * C + c4 * 4 + e], s, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        else
          atomicAdd(&out[(long long)(blockIdx.x & (YOLO_BN_STAT_SLOTS - 1)) * NQ * C + (long long)q * C + c4 * 4 + e], s);
      }
  }
}

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, long long P, int C, int cw,
                                                       int rpp, double* __restrict__ stats) {
  __shared__ double smem[8 * 256];
  const int tid = threadIdx.x;
  const int row_lane = tid / cw, col = tid - row_lane * cw;
  const int C4 = C >> 2;
  const int c4 = blockIdx.y * cw + col;
  const bool active = (row_lane < rpp) && (c4 < C4);
  double v[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  if (active) {
    // each block sweeps ONE contiguous range of rows front to back (sequential DRAM pages)
    const long long per_block = (P + gridDim.x - 1) / gridDim.x;
    const long long p_lo = (long long)blockIdx.x * per_block;
    const long long p_hi = (p_lo + per_block < P) ? p_lo + per_block : P;
    const long long stride = rpp;
    for (long long p = p_lo + row_lane; p < p_hi; p += 4 * stride) {
      f32x4 t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long pu = p + u * stride;
        t[u] = (pu < p_hi) ? *reinterpret_cast<const f32x4*>(x + pu * C + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const double d = (double)t[u][e];
          v[0][e] += d;
          v[1][e] += d * d;
        }
    }
  }
  block_col_reduce<2>(v, cw, rpp, row_lane, col, active, smem, stats, C, c4);
}

// One workgroup = 16 channels x the 64 replica slots (1024 threads: thread t reads slot t >> 4 of channel t & 15, so every
// load instruction of a wave covers four whole 128-byte lines; two LDS levels fold the slots in a fixed order), then 16
// threads finish their channels. (Until round 6 a WAVE owned a channel and lane r read slot r: 64 lanes = 64 different cache
// lines per load, 16x the traffic -- 12.5 us per launch on the 1024-channel layers of YOLOv1.5 at bs 4, where the whole
// training step is 3 ms; 5.6 us in the YOLOv3-416 step.)
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const double* __restrict__ stats, long long P, int C,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float momentum, int unbiased, float* __restrict__ mmean, float* __restrict__ mvar,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ smean,
                                   float* __restrict__ sinv, const unsigned* __restrict__ absmax,
                                   unsigned* __restrict__ bound, const float* __restrict__ mean_offset) {
  __shared__ double sh1[YOLO_BN_STAT_SLOTS][17], sh2[YOLO_BN_STAT_SLOTS][17];
  __shared__ double sp1[16][17], sp2[16][17];
  const int ch = threadIdx.x & 15, slot = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + ch;
  double v1 = 0.0, v2 = 0.0;
  if (c < C) {
    v1 = stats[(long long)slot * 2 * C + c];
    v2 = stats[(long long)slot * 2 * C + C + c];
  }
  sh1[slot][ch] = v1;
  sh2[slot][ch] = v2;
  __syncthreads();
  if (threadIdx.x < 256) {   // part = t >> 4 adds its four slots in slot order
    const int part = threadIdx.x >> 4;
    double a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a1 += sh1[part * 4 + u][ch];
      a2 += sh2[part * 4 + u][ch];
    }
    sp1[part][ch] = a1;
    sp2[part][ch] = a2;
  }
  __syncthreads();
  float bnd = 0.f;
  if (threadIdx.x < 16 && c < C) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      s1 += sp1[u][ch];
      s2 += sp2[u][ch];
    }
    const double mean = s1 / (double)P;
    double var = s2 / (double)P - mean * mean;
    if (var < 0.0) var = 0.0;
    const double inv = 1.0 / sqrt(var + (double)eps);
    const float sc = (float)((double)gamma[c] * inv);
    scale[c] = sc;
    shift[c] = (float)((double)beta[c] - mean * (double)gamma[c] * inv);
    smean[c] = (float)mean;
    sinv[c] = (float)inv;
    if (bound != nullptr) {
      // |act(scale*y + shift)| <= |gamma| * |y - mean| * inv + |beta| and (y_i - mean)^2 <= sum_j (y_j - mean)^2 = P*var:
      // an upper bound of the layer's output that needs no pass over the data (planes.hpp: any B >= max|x| will do)
      // (with the conv epilogue's per-channel max|y| the bound is tight: |y - mean| <= max|y| + |mean|)
      const double dev = absmax != nullptr ? (double)__builtin_bit_cast(float, absmax[c]) + fabs(mean)
                                           : sqrt((double)P * var);
      bnd = (float)(fabs((double)gamma[c]) * inv * dev * 1.001 + fabs((double)beta[c]) + 1e-30);
    }
    if (mmean != nullptr) {
      double fed = var;
      if (unbiased && P > 1) fed = var * (double)P / (double)(P - 1);
      // mean_offset: the statistics are those of y - offset (a conv bias left out of the convolution: it cancels in
      // training-mode BatchNormalization, so the conv + BN unit never adds it; only the moving mean has to know about it)
      const double mean_full = mean + (mean_offset != nullptr ? (double)mean_offset[c] : 0.0);
      mmean[c] = (float)((double)momentum * mmean[c] + (1.0 - (double)momentum) * mean_full);
      mvar[c] = (float)((double)momentum * mvar[c] + (1.0 - (double)momentum) * fed);
    }
  }
  // ONE atomic per workgroup (16 channels) for the tensor's bound: one per channel -- 1024 device-scope atomics on one
  // address for the 13x13 layers -- made this kernel 14.6 us where the 32-channel layers take 4.7 (atomics on one address queue up)
  if (bound != nullptr && threadIdx.x < 64) {   // (lanes 0..15 of wave 0 hold the 16 channel bounds)
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) bnd = fmaxf(bnd, __shfl_xor(bnd, o, 64));
    if (threadIdx.x == 0 && bnd > 0.f) atomicMax(bound, __builtin_bit_cast(unsigned, bnd));
  }
}

__global__ void bn_fold_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mmean, const float* __restrict__ mvar, float eps,
                               float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double inv = 1.0 / sqrt((double)mvar[c] + (double)eps);
  scale[c] = (float)((double)gamma[c] * inv);
  shift[c] = (float)((double)beta[c] - (double)mmean[c] * (double)gamma[c] * inv);
}

__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ x, long long n4, int C4,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int act,
                                                         const float* __restrict__ res, float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
    const f32x4 sc = reinterpret_cast<const f32x4*>(scale)[c4];
    const f32x4 sh = reinterpret_cast<const f32x4*>(shift)[c4];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = act_fwd(fmaf(sc[e], xv[e], sh[e]), act);
    if (res != nullptr) {
      const f32x4 r = reinterpret_cast<const f32x4*>(res)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] += r[e];
    }
    reinterpret_cast<f32x4*>(out)[i] = o;
  }
}

// FOLD (round 6): the kernel also FINISHES the reduction -- what bn_bwd_sum_kernel did in a launch of its own (7.7 us + its
// place in the compute queue, 72 times per step on the critical chain of backward). Two levels of "last workgroup to arrive":
// the workgroups of a launch form groups of 16 consecutive blockIdx.x; the last of a group to arrive (ticket word of the
// group) adds the group's 16 slots IN SLOT ORDER into the group's first slot; the last GROUP to finish (ticket word of the
// column chunk) adds the group sums in group order into the final sums, makes the bound words and sets the tickets back to
// zero. Who arrives last changes nothing in the arithmetic: bit-reproducible like the two-launch form (whose summation
// order -- 16 strided partial sums -- it does not reproduce: the fp64 sums differ in their last bits). The slots cross
// XCDs: release fence before a ticket, acquire fence behind it (as bn_bwd_sum_part_kernel / the stream-K window kernel).
// tickets: 33 words per column chunk (blockIdx.y): [0] groups done, [1 + g] workgroups of group g done.
constexpr int BN_FOLD_GROUP = 16;
constexpr int BN_FOLD_TICKETS = 1 + YOLO_BN_RED_SLOTS / BN_FOLD_GROUP;   // per column chunk
template <bool FOLD>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ dout, long long ldd, long long P, int C,
                                                            int cw, int rpp, const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ smean,
                                                            const float* __restrict__ sinv, int act,
                                                            double* red, unsigned* aux, unsigned* tickets) {
  __shared__ double smem[8 * 256];
  float mdz = 0.f;
  const int tid = threadIdx.x;
  const int row_lane = tid / cw, col = tid - row_lane * cw;
  const int C4 = C >> 2;
  const int c4 = blockIdx.y * cw + col;
  const bool active = (row_lane < rpp) && (c4 < C4);
  double v[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  if (active) {
    const f32x4 sc = reinterpret_cast<const f32x4*>(scale)[c4];
    const f32x4 sh = reinterpret_cast<const f32x4*>(shift)[c4];
    const f32x4 mu = reinterpret_cast<const f32x4*>(smean)[c4];
    const f32x4 iv = reinterpret_cast<const f32x4*>(sinv)[c4];
    const long long per_block = (P + gridDim.x - 1) / gridDim.x;
    const long long p_lo = (long long)blockIdx.x * per_block;
    const long long p_hi = (p_lo + per_block < P) ? p_lo + per_block : P;
    const long long stride = rpp;
    // the activation is a compile-time constant inside the loop (a run-time `act` leaves three scalar branches per
    // ELEMENT in it): 1-5 % on the BatchNorm passes in same-box A/Bs. (The Mish reduce stays VALU-bound either way --
    // 230-245 us for a tensor that LeakyReLU streams in 130 -- and its time differs by 40 % from box to box.)
    auto sweep = [&](auto ACT_) {
      constexpr int A = decltype(ACT_)::value;
      // software-pipelined: the eight loads of the NEXT four rows are in flight while these four rows are worked on (with
      // Mish that is 700 instructions; at two waves per SIMD the kernel otherwise alternates between waiting for memory
      // and computing: 19 % VALU-busy at 3 TB/s, profiles: rocprofv3 --pmc SQ_ACTIVE_INST_VALU)
      f32x4 xv[4], dv[4], xn[4], dn[4];
      auto fetch = [&](long long p, f32x4 (&xo)[4], f32x4 (&dO)[4]) {
        // all 8 loads are issued unconditionally (rows past the end re-read row p and are zeroed afterwards):
        // a per-load "in range ? load : 0" makes hipcc branch around every load and wait for each in turn
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long long pu = p + u * stride;
          const long long pc = pu < p_hi ? pu : p;
          xo[u] = *reinterpret_cast<const f32x4*>(x + pc * C + c4 * 4);
          dO[u] = *reinterpret_cast<const f32x4*>(dout + pc * ldd + c4 * 4);   // (ldd: row pitch of dout, C when dense)
        }
      };
      long long p = p_lo + row_lane;
      if (p < p_hi) fetch(p, xv, dv);
      while (p < p_hi) {
        const long long pn = p + 4 * stride;
        if (pn < p_hi) fetch(pn, xn, dn);
#pragma unroll
        for (int u = 1; u < 4; ++u)
          if (p + u * stride >= p_hi) dv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float z = fmaf(sc[e], xv[u][e], sh[e]);
            const float dz = dv[u][e] * act_grad(z, A);   // dv = 0 for out-of-range rows
            mdz = fmaxf(mdz, fabsf(dz));
            const float xh = (xv[u][e] - mu[e]) * iv[e];
            // (round 6: adding the four rows of a batch in fp32 and only the batch sums in fp64 -- a quarter of the fp64
            // operations -- measured +-0 on YOLOv4-608 and YOLOv3-416: the Mish pass is not bound by them)
            v[0][e] += (double)dz;
            v[1][e] += (double)dz * (double)xh;
          }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          xv[u] = xn[u];
          dv[u] = dn[u];
        }
        p = pn;
      }
    };
    if (act == YOLO_ACT_MISH) sweep(std::integral_constant<int, YOLO_ACT_MISH>{});
    else if (act == YOLO_ACT_LEAKY) sweep(std::integral_constant<int, YOLO_ACT_LEAKY>{});
    else sweep(std::integral_constant<int, YOLO_ACT_LINEAR>{});
  }
  block_col_reduce<2, true>(v, cw, rpp, row_lane, col, active, smem, red, C, c4);
  if (aux != nullptr) {   // max |dz| of the tensor (bit patterns of non-negative floats order like integers):
    // one atomic per workgroup, spread over 64 replica slots aux[4..67] (bn_bwd_sum_kernel folds them)
    __shared__ float s_max[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mdz = fmaxf(mdz, __shfl_xor(mdz, o, 64));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = mdz;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float m = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
      if (m > 0.f) atomicMax(&aux[4 + ((blockIdx.x + blockIdx.y) & 63)], __builtin_bit_cast(unsigned, m));
    }
  }
  if constexpr (FOLD) {
    // No cache-wide fences here: a release / acquire fence at agent scope writes back / invalidates the whole L2 of the XCD,
    // and in the two-stream step that L2 is full of the OTHER stream's filter-gradient operands and slabs -- the first version
    // of this fold (fence + ticket per workgroup) cost 6 ms per step. Instead every value that crosses workgroups is moved
    // with agent-scope RELAXED atomics (sc1 loads / stores: through to the memory side, past the XCD's L2) and ordered by
    // hand: my stores are complete (vmcnt(0)) before my workgroup's barrier, the barrier comes before the ticket.
    __shared__ unsigned s_tk;
    auto ld = [](const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto st = [](double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    const int gx = (int)gridDim.x;
    const int grp = (int)blockIdx.x / BN_FOLD_GROUP, ngrp = (gx + BN_FOLD_GROUP - 1) / BN_FOLD_GROUP;
    const int gsz = (gx - grp * BN_FOLD_GROUP < BN_FOLD_GROUP) ? gx - grp * BN_FOLD_GROUP : BN_FOLD_GROUP;
    unsigned* tk = tickets + (size_t)blockIdx.y * BN_FOLD_TICKETS;
    const int c_lo = (int)blockIdx.y * cw * 4;
    const int c_n = (C - c_lo < cw * 4) ? C - c_lo : cw * 4;      // this column chunk's channels
    // ---- level 1: my slot (and my max|dz| atomic) is out; am I the last of my group? ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_tk = __hip_atomic_fetch_add(&tk[1 + grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_tk != (unsigned)(gsz - 1)) return;
    {
      const long long s0 = (long long)grp * BN_FOLD_GROUP;
      for (int cc = tid; cc < c_n; cc += 256) {
        const int c = c_lo + cc;
        double v0[BN_FOLD_GROUP], v1[BN_FOLD_GROUP];
#pragma unroll
        for (int r = 0; r < BN_FOLD_GROUP; ++r) {   // all loads in flight together; slots past the group read slot s0 again
          const long long sl = s0 + (r < gsz ? r : 0);
          v0[r] = ld(&red[sl * 2 * C + c]);
          v1[r] = ld(&red[sl * 2 * C + C + c]);
        }
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int r = 0; r < BN_FOLD_GROUP; ++r) {   // slot order
          a0 += r < gsz ? v0[r] : 0.0;
          a1 += r < gsz ? v1[r] : 0.0;
        }
        st(&red[s0 * 2 * C + c], a0);               // (the group's own first slot: nobody else reads the group's slots)
        st(&red[s0 * 2 * C + C + c], a1);
      }
    }
    if (tid == 0) __hip_atomic_store(&tk[1 + grp], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
    // ---- level 2: the group's sum is out; is mine the last group? ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_tk = __hip_atomic_fetch_add(&tk[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_tk != (unsigned)(ngrp - 1)) return;
    for (int cc = tid; cc < ((c_n + 63) & ~63); cc += 256) {
      float u1 = 0.f, u2 = 0.f;
      if (cc < c_n) {
        const int c = c_lo + cc;
        double a0 = 0.0, a1 = 0.0;
        int g = 0;
        for (; g + 8 <= ngrp; g += 8) {             // eight groups (sixteen loads) in flight, added in group order
          double v0[8], v1[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            v0[u] = ld(&red[(long long)(g + u) * BN_FOLD_GROUP * 2 * C + c]);
            v1[u] = ld(&red[(long long)(g + u) * BN_FOLD_GROUP * 2 * C + C + c]);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            a0 += v0[u];
            a1 += v1[u];
          }
        }
        for (; g < ngrp; ++g) {
          a0 += ld(&red[(long long)g * BN_FOLD_GROUP * 2 * C + c]);
          a1 += ld(&red[(long long)g * BN_FOLD_GROUP * 2 * C + C + c]);
        }
        red[(long long)YOLO_BN_RED_SLOTS * 2 * C + c] = a0;      // (read by the next launch: plain stores)
        red[(long long)YOLO_BN_RED_SLOTS * 2 * C + C + c] = a1;
        const double asc = fabs((double)scale[c]);
        u1 = (float)asc;
        u2 = (float)(asc * (fabs(a1 / (double)P) * sqrt((double)P) + fabs(a0 / (double)P)) * 1.001);
      }
      if (aux != nullptr) {   // per-channel parts of the bound of dx (bn_bwd_apply8_kernel)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          u1 = fmaxf(u1, __shfl_xor(u1, o, 64));
          u2 = fmaxf(u2, __shfl_xor(u2, o, 64));
        }
        if ((tid & 63) == 0) {
          atomicMax(&aux[1], __builtin_bit_cast(unsigned, u1));
          atomicMax(&aux[2], __builtin_bit_cast(unsigned, u2));
        }
      }
    }
    if (aux != nullptr && tid < 64) {   // the 64 replica words of max|dz| (with several column chunks every chunk's last
      // workgroup folds what is there: the maximum over all of them is the tensor's)
      unsigned m = __hip_atomic_load(&aux[4 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned v = __shfl_xor(m, o, 64);
        m = v > m ? v : m;
      }
      if (tid == 0 && m != 0u) atomicMax(&aux[0], m);
    }
    if (tid == 0) __hip_atomic_store(&tk[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ dout, long long n4, int C4,
                                                           double invP, const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ smean,
                                                           const float* __restrict__ sinv, int act,
                                                           const double* __restrict__ red, float* __restrict__ dx) {
  const int C = C4 * 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
    const f32x4 dv = reinterpret_cast<const f32x4*>(dout)[i];
    const f32x4 sc = reinterpret_cast<const f32x4*>(scale)[c4];
    const f32x4 sh = reinterpret_cast<const f32x4*>(shift)[c4];
    const f32x4 mu = reinterpret_cast<const f32x4*>(smean)[c4];
    const f32x4 iv = reinterpret_cast<const f32x4*>(sinv)[c4];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float z = fmaf(sc[e], xv[e], sh[e]);
      const float dz = dv[e] * act_grad(z, act);
      const float xh = (xv[e] - mu[e]) * iv[e];
      const float mdz = (float)(red[c4 * 4 + e] * invP);
      const float mdzx = (float)(red[C + c4 * 4 + e] * invP);
      o[e] = sc[e] * (dz - mdz - xh * mdzx);
    }
    reinterpret_cast<f32x4*>(dx)[i] = o;
  }
}

// ---- column-fixed forms (C % 8 == 0): a thread owns one 8-channel group for a run of rows, so the
// per-channel coefficients live in registers (the generic kernels above re-load them and take a 64-bit
// modulo per element: 128 B of parameter loads per 16 B of data). Optionally the result is also written
// in the "planes" operand format of the conv kernels (conv_planes.hip) -- exact 3-way bf16 split, one
// 16-byte unit per plane per thread -- so the consumer convolutions need no separate split pass. Rows in
// [P, rows_padded) of the planes (block tail + the all-zero block) are zero-filled here. ----
// Thread mapping: a wave owns one 32-channel quad (4 groups of 8 channels) and walks 16-row blocks; lane =
// row-in-block + 16 * group-in-quad. Every fp32 load/store instruction of the wave then touches 16 rows x
// 128 contiguous bytes, and every planes store writes 4 whole 256-byte sub-blocks (16 rows x 16 B).
// C % 32 != 0 (C = 16 * odd) leaves the last quad half empty.
struct RowGeom {
  int wpr;            // waves side by side along the channels in one workgroup (1, 2 or 4)
  unsigned gx, gy;
  long long blocks_per_wg;  // 16-row blocks each wave row-slot walks
};
static inline RowGeom row_geom(long long rows, int C) {
  RowGeom g;
  const int quads = (C + 31) / 32;
  g.wpr = quads >= 4 ? 4 : quads >= 2 ? 2 : 1;
  g.gy = (unsigned)((quads + g.wpr - 1) / g.wpr);
  const long long rblocks = (rows + 15) / 16;
  const int slots = 4 / g.wpr;                       // row-block slots per workgroup pass
  long long per = (rblocks + 2047) / 2048;           // <= ~2048 workgroups along the rows
  if (per < 2 * slots) per = 2 * slots;
  per = (per + slots - 1) / slots * slots;
  g.blocks_per_wg = per;
  g.gx = (unsigned)((rblocks + per - 1) / per);
  return g;
}

// every thread derives the tensor's scale from the bound; one thread completes the planes header
__device__ __forceinline__ float planes_begin(unsigned char* planes, long long P, int C, float bound) {
  const unsigned bits = __builtin_bit_cast(unsigned, bound);
  const float sc = planes_scale_from_bound(bits);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    unsigned* header = reinterpret_cast<unsigned*>(planes + planes_body_bytes(P, C));
    header[0] = bits;
    reinterpret_cast<float*>(header)[1] = sc;
    reinterpret_cast<float*>(header)[2] = 1.f / sc;
  }
  return sc;
}

template <bool PLANES>
__global__ __launch_bounds__(256) void bn_act_fwd8_kernel(const float* __restrict__ x, long long P, int C, int wpr,
                                                          long long blocks_per_wg, long long rows_total,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int act,
                                                          const float* __restrict__ res, float* __restrict__ out,
                                                          unsigned char* __restrict__ planes,
                                                          const unsigned* __restrict__ bn_bound,
                                                          const float* __restrict__ res_bound,
                                                          float* __restrict__ out_bound,
                                                          const unsigned char* __restrict__ res_planes) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the residual may come as planes (the operand format its producer wrote for the convolutions: h + l = the value
  // to 22-23 bits; then its fp32 copy need not exist at all): header = {bound, scale, 1 / scale}
  const float* rhead = res_planes != nullptr ? reinterpret_cast<const float*>(res_planes + planes_body_bytes(P, C)) : nullptr;
  const float rinv = rhead != nullptr ? rhead[2] : 0.f;
  // bound of the output = bound of the BN/activation part (bn_finalize) + bound of the residual
  float psc = 1.f;
  if (PLANES || out_bound != nullptr) {
    const float b = (bn_bound ? __builtin_bit_cast(float, bn_bound[0]) : 0.f) +
                    (rhead != nullptr ? rhead[0] : (res != nullptr && res_bound != nullptr) ? res_bound[0] : 0.f);
    if (out_bound != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) out_bound[0] = b;
    if (PLANES) psc = planes_begin(planes, P, C, b);
  }
  const int g8 = (blockIdx.y * wpr + wave % wpr) * 4 + (lane >> 4);
  if (g8 >= (C >> 3)) return;
  const int slots = 4 / wpr;
  const f32x4 sc0 = reinterpret_cast<const f32x4*>(scale)[2 * g8], sc1 = reinterpret_cast<const f32x4*>(scale)[2 * g8 + 1];
  const f32x4 sh0 = reinterpret_cast<const f32x4*>(shift)[2 * g8], sh1 = reinterpret_cast<const f32x4*>(shift)[2 * g8 + 1];
  const long long rb_lo = (long long)blockIdx.x * blocks_per_wg;
  long long p_hi = (rb_lo + blocks_per_wg) * 16;
  if (p_hi > rows_total) p_hi = rows_total;
  // (LeakyReLU as a compile-time constant inside the loop, see bn_bwd_reduce_kernel: 1-5 % on this pass. Mish keeps the
  // run-time value: specialised, its arithmetic differs in the last bit from one instantiation to the next, and the
  // fused inference epilogue / the planes and plain forms of these kernels are tested to be bit-identical)
  auto sweep = [&](auto ACT_) {
    const int A = ACT_;
#pragma unroll 2
  for (long long p = (rb_lo + wave / wpr) * 16 + (lane & 15); p < p_hi; p += 16 * slots) {
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    if (p < P) {
      const long long e = p * C + g8 * 8;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(x + e), x1 = *reinterpret_cast<const f32x4*>(x + e + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        o0[k] = act_fwd(fmaf(sc0[k], x0[k], sh0[k]), A);
        o1[k] = act_fwd(fmaf(sc1[k], x1[k], sh1[k]), A);
      }
      if (res_planes != nullptr) {
        const unsigned char* ru = res_planes + planes_unit_offset(p, g8, C);
        const f16x8 rh = *reinterpret_cast<const f16x8*>(ru), rl = *reinterpret_cast<const f16x8*>(ru + 512);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          o0[k] += ((float)rh[k] + (float)rl[k]) * rinv;           // (h + l exact in fp32, 1 / scale a power of two)
          o1[k] += ((float)rh[4 + k] + (float)rl[4 + k]) * rinv;
        }
      } else if (res != nullptr) {
        o0 += *reinterpret_cast<const f32x4*>(res + e);
        o1 += *reinterpret_cast<const f32x4*>(res + e + 4);
      }
      if (out != nullptr) {   // (nullptr: every consumer reads the planes, the fp32 copy is not needed)
        *reinterpret_cast<f32x4*>(out + e) = o0;
        *reinterpret_cast<f32x4*>(out + e + 4) = o1;
      }
    }
    if (PLANES) store_planes8(planes, p, g8, C, o0, o1, psc);
  }
  };
  if (act == YOLO_ACT_LEAKY) sweep(std::integral_constant<int, YOLO_ACT_LEAKY>{});
  else sweep(act);
}

// dx = scale * (dz - mean(dz) - xhat * mean(dz * xhat)),  dz = dout * act'(scale*x + shift)
//    = scale*dz + B*(x - mean) + K  with per-channel B = -scale*invstd*mean(dz*xhat), K = -scale*mean(dz)
// (x - mean is formed first, as in the generic kernel: no cancellation between B*x and B*mean)
template <bool PLANES, bool WRITE_DX>
__global__ __launch_bounds__(256) void bn_bwd_apply8_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ dout, long long ldd, long long P, int C,
                                                            int wpr, long long blocks_per_wg, long long rows_total,
                                                            double invP, const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ smean,
                                                            const float* __restrict__ sinv, int act,
                                                            const double* __restrict__ red, float* __restrict__ dx,
                                                            unsigned char* __restrict__ planes,
                                                            const unsigned* __restrict__ aux,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // |dx| <= max_c|scale_c| * max|dz| + max_c |scale_c| (|mean(dz xhat)| sqrt(P) + |mean(dz)|): aux = {max|dz|,
  // max|scale|, max of the second term} from the reduce / sum kernels
  float psc = 1.f;
  if (PLANES) {
    const float b = __builtin_bit_cast(float, aux[0]) * __builtin_bit_cast(float, aux[1]) * 1.001f +
                    __builtin_bit_cast(float, aux[2]) + 1e-30f;
    psc = planes_begin(planes, P, C, b);
  }
  const int g8 = (blockIdx.y * wpr + wave % wpr) * 4 + (lane >> 4);
  if (g8 >= (C >> 3)) return;
  const int slots = 4 / wpr;
  float sc[8], sh[8], mu[8], cb[8], ck[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = g8 * 8 + k;
    sc[k] = scale[c];
    sh[k] = shift[c];
    mu[k] = smean[c];
    const float mdz = (float)(red[c] * invP);
    const float mdzx = (float)(red[C + c] * invP);
    cb[k] = -sc[k] * sinv[c] * mdzx;
    ck[k] = -sc[k] * mdz;
    // the parameter gradients are the channel sums themselves (dbeta = sum dz, dgamma = sum dz xhat): the first
    // row-slot of the first workgroup along the rows adds them (one lane per 8-channel group) - no extra launch
    if (blockIdx.x == 0 && wave / wpr == 0 && (lane & 15) == 0) {
      if (dbeta != nullptr) dbeta[c] += (float)red[c];
      if (dgamma != nullptr) dgamma[c] += (float)red[C + c];
    }
  }
  const long long rb_lo = (long long)blockIdx.x * blocks_per_wg;
  long long p_hi = (rb_lo + blocks_per_wg) * 16;
  if (p_hi > rows_total) p_hi = rows_total;
  // (LeakyReLU as a compile-time constant inside the loop, see bn_bwd_reduce_kernel: 1-5 % on this pass. Mish keeps the
  // run-time value: specialised, its arithmetic differs in the last bit from one instantiation to the next, and the
  // fused inference epilogue / the planes and plain forms of these kernels are tested to be bit-identical)
  auto sweep = [&](auto ACT_) {
    const int A = ACT_;
#pragma unroll 2
  for (long long p = (rb_lo + wave / wpr) * 16 + (lane & 15); p < p_hi; p += 16 * slots) {
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    if (p < P) {
      const long long e = p * C + g8 * 8;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(x + e), x1 = *reinterpret_cast<const f32x4*>(x + e + 4);
      const long long ed = p * ldd + g8 * 8;   // (ldd: row pitch of dout, C when dense)
      const f32x4 d0 = *reinterpret_cast<const f32x4*>(dout + ed), d1 = *reinterpret_cast<const f32x4*>(dout + ed + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dz0 = d0[k] * act_grad(fmaf(sc[k], x0[k], sh[k]), A);
        const float dz1 = d1[k] * act_grad(fmaf(sc[4 + k], x1[k], sh[4 + k]), A);
        o0[k] = fmaf(sc[k], dz0, fmaf(cb[k], x0[k] - mu[k], ck[k]));
        o1[k] = fmaf(sc[4 + k], dz1, fmaf(cb[4 + k], x1[k] - mu[4 + k], ck[4 + k]));
      }
      if (WRITE_DX) {
        *reinterpret_cast<f32x4*>(dx + e) = o0;
        *reinterpret_cast<f32x4*>(dx + e + 4) = o1;
      }
    }
    if (PLANES) store_planes8(planes, p, g8, C, o0, o1, psc);
  }
  };
  if (act == YOLO_ACT_LEAKY) sweep(std::integral_constant<int, YOLO_ACT_LEAKY>{});
  else sweep(act);
}

// red layout: [SLOTS replicas][2][C] followed by the final [2][C] sums
// workgroup = 16 channels x 16 slot groups: a thread sums every 16th per-workgroup partial of its channel (4
// independent accumulators keep the loads in flight), LDS folds the 16 groups
__global__ __launch_bounds__(256) void bn_bwd_sum_kernel(int C, double* __restrict__ red, int nslots, long long P,
                                                         const float* __restrict__ scale,
                                                         unsigned* __restrict__ aux) {
  __shared__ double s_part[2][16][16];
  const int ch = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + ch;
  double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
  if (c < C) {
    int r = grp;
    // eight slots (sixteen loads) in flight per round trip: with two, the 512 slots of a large layer were sixteen
    // dependent round trips = 8 us of a kernel that moves a few hundred KB
    for (; r + 112 < nslots; r += 128) {
      double v0[8], v1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v0[u] = red[(long long)(r + 16 * u) * 2 * C + c];
        v1[u] = red[(long long)(r + 16 * u) * 2 * C + C + c];
      }
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        a0 += v0[u];
        a1 += v1[u];
        b0 += v0[u + 1];
        b1 += v1[u + 1];
      }
    }
    for (; r + 16 < nslots; r += 32) {
      a0 += red[(long long)r * 2 * C + c];
      a1 += red[(long long)r * 2 * C + C + c];
      b0 += red[(long long)(r + 16) * 2 * C + c];
      b1 += red[(long long)(r + 16) * 2 * C + C + c];
    }
    if (r < nslots) {
      a0 += red[(long long)r * 2 * C + c];
      a1 += red[(long long)r * 2 * C + C + c];
    }
  }
  s_part[0][grp][ch] = a0 + b0;
  s_part[1][grp][ch] = a1 + b1;
  __syncthreads();
  float t1 = 0.f, t2 = 0.f;
  if (grp == 0 && c < C) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      s0 += s_part[0][g][ch];
      s1 += s_part[1][g][ch];
    }
    red[(long long)YOLO_BN_RED_SLOTS * 2 * C + c] = s0;
    red[(long long)YOLO_BN_RED_SLOTS * 2 * C + C + c] = s1;
    if (aux != nullptr) {   // per-channel parts of the bound of dx (bn_bwd_apply8_kernel): |x - mean| invstd <= sqrt(P)
      const double asc = fabs((double)scale[c]);
      t1 = (float)asc;
      t2 = (float)(asc * (fabs(s1 / (double)P) * sqrt((double)P) + fabs(s0 / (double)P)) * 1.001);
    }
  }
  if (aux != nullptr && threadIdx.x < 64) {   // wave 0 holds the 16 channel results in lanes 0..15
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      t1 = fmaxf(t1, __shfl_xor(t1, o, 64));
      t2 = fmaxf(t2, __shfl_xor(t2, o, 64));
    }
    if (threadIdx.x == 0) {
      if (t1 > __builtin_bit_cast(float, aux[1])) atomicMax(&aux[1], __builtin_bit_cast(unsigned, t1));
      if (t2 > __builtin_bit_cast(float, aux[2])) atomicMax(&aux[2], __builtin_bit_cast(unsigned, t2));
    }
  }
  if (aux != nullptr && blockIdx.x == 0 && threadIdx.x >= 64 && threadIdx.x < 128) {   // fold the 64 slots of max|dz|
    unsigned m = aux[4 + (threadIdx.x & 63)];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned v = __shfl_xor(m, o, 64);
      m = v > m ? v : m;
    }
    if ((threadIdx.x & 63) == 0) aux[0] = m;
  }
}

// The slots of a reduction FUSED into a data gradient (planes_epilogue.hpp: GatherConvArgs::bwd_part, [nslots][2][C]
// floats, one slot per row tile of that launch -- 43 for a 13x13 layer at batch 32, 10 816 for a 208x208 one) folded in
// slot order into the final fp64 sums of `red`, plus the bound words, as bn_bwd_sum_kernel does for the standalone pass.
// grid (C / 16, chunks): a workgroup = 16 channels x 16 slot groups over `chunk` slots. One chunk: the workgroup writes
// the final sums. Several: it writes its chunk's sums to slot blockIdx.y of `red`, and the LAST workgroup to arrive
// (ticket aux[3]; the chunk sums cross XCDs: fence before the ticket, acquire behind it) adds the chunks IN ORDER for
// every channel -- who arrives last changes nothing in the arithmetic.
__global__ __launch_bounds__(256) void bn_bwd_sum_part_kernel(int C, const float* __restrict__ part, int nslots, int chunk,
                                                              double* red, long long P, const float* __restrict__ scale,
                                                              unsigned* aux) {
  __shared__ double s_part[2][16][16];
  __shared__ unsigned s_ticket;
  const int ch = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + ch;
  const int r_lo = blockIdx.y * chunk;
  const int r_hi = (r_lo + chunk < nslots) ? r_lo + chunk : nslots;
  const bool single = gridDim.y == 1;
  double a0 = 0.0, a1 = 0.0;
  if (c < C) {
    int r = r_lo + grp;
    for (; r + 112 < r_hi; r += 128) {   // eight slots (sixteen loads) in flight per round trip
      float v0[8], v1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v0[u] = part[((long long)(r + 16 * u) * 2 + 0) * C + c];
        v1[u] = part[((long long)(r + 16 * u) * 2 + 1) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a0 += (double)v0[u];
        a1 += (double)v1[u];
      }
    }
    for (; r < r_hi; r += 16) {
      a0 += (double)part[((long long)r * 2 + 0) * C + c];
      a1 += (double)part[((long long)r * 2 + 1) * C + c];
    }
  }
  s_part[0][grp][ch] = a0;
  s_part[1][grp][ch] = a1;
  __syncthreads();
  float t1 = 0.f, t2 = 0.f;
  if (grp == 0 && c < C) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      s0 += s_part[0][g][ch];
      s1 += s_part[1][g][ch];
    }
    const long long dst = single ? (long long)YOLO_BN_RED_SLOTS : (long long)blockIdx.y;
    red[dst * 2 * C + c] = s0;
    red[dst * 2 * C + C + c] = s1;
    if (single && aux != nullptr) {
      const double asc = fabs((double)scale[c]);
      t1 = (float)asc;
      t2 = (float)(asc * (fabs(s1 / (double)P) * sqrt((double)P) + fabs(s0 / (double)P)) * 1.001);
    }
  }
  if (single) {
    if (aux != nullptr && threadIdx.x < 64) {   // wave 0 holds the 16 channel results in lanes 0..15
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        t1 = fmaxf(t1, __shfl_xor(t1, o, 64));
        t2 = fmaxf(t2, __shfl_xor(t2, o, 64));
      }
      if (threadIdx.x == 0) {
        if (t1 > __builtin_bit_cast(float, aux[1])) atomicMax(&aux[1], __builtin_bit_cast(unsigned, t1));
        if (t2 > __builtin_bit_cast(float, aux[2])) atomicMax(&aux[2], __builtin_bit_cast(unsigned, t2));
      }
    }
    if (aux != nullptr && blockIdx.x == 0 && threadIdx.x >= 64 && threadIdx.x < 128) {   // fold the 64 slots of max|dz|
      unsigned m = aux[4 + (threadIdx.x & 63)];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned v = __shfl_xor(m, o, 64);
        m = v > m ? v : m;
      }
      if ((threadIdx.x & 63) == 0) aux[0] = m;
    }
    return;
  }
  // ---- several chunks: ticket, the last arriver finishes every channel ----
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0)
    s_ticket = __hip_atomic_fetch_add(&aux[3], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (s_ticket != gridDim.x * gridDim.y - 1) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const int nch = (int)gridDim.y;
  for (int cc = threadIdx.x; cc < ((C + 63) & ~63); cc += 256) {
    float u1 = 0.f, u2 = 0.f;
    if (cc < C) {
      double s0 = 0.0, s1 = 0.0;
      for (int k = 0; k < nch; ++k) {   // chunk order
        s0 += red[(long long)k * 2 * C + cc];
        s1 += red[(long long)k * 2 * C + C + cc];
      }
      red[(long long)YOLO_BN_RED_SLOTS * 2 * C + cc] = s0;
      red[(long long)YOLO_BN_RED_SLOTS * 2 * C + C + cc] = s1;
      const double asc = fabs((double)scale[cc]);
      u1 = (float)asc;
      u2 = (float)(asc * (fabs(s1 / (double)P) * sqrt((double)P) + fabs(s0 / (double)P)) * 1.001);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      u1 = fmaxf(u1, __shfl_xor(u1, o, 64));
      u2 = fmaxf(u2, __shfl_xor(u2, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMax(&aux[1], __builtin_bit_cast(unsigned, u1));
      atomicMax(&aux[2], __builtin_bit_cast(unsigned, u2));
    }
  }
  if (threadIdx.x < 64) {
    unsigned m = aux[4 + threadIdx.x];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned v = __shfl_xor(m, o, 64);
      m = v > m ? v : m;
    }
    if (threadIdx.x == 0) {
      aux[0] = m;
      __hip_atomic_store(&aux[3], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the ticket word, for the next use
    }
  }
}

__global__ void bn_bwd_params_kernel(int C, const double* __restrict__ redsum, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (dbeta) dbeta[c] += (float)redsum[c];
  if (dgamma) dgamma[c] += (float)redsum[C + c];
}

__global__ void act_fwd_kernel(const float* __restrict__ x, long long n, int act, float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = act_fwd(x[i], act);
}
__global__ void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dout, long long n, int act,
                               float* __restrict__ dx) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dx[i] = dout[i] * act_grad(x[i], act);
}

static int reduce_grid_x(long long P, int rpp) {
  long long g = (P + (long long)rpp * 16 - 1) / ((long long)rpp * 16);  // >= 16 rows per thread
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace yolo

using namespace yolo;

extern "C" int yolo_bn_stats(const float* x, long long P, int C, double* stats, void* stream) {
  YOLO_REQUIRE(x && stats && P > 0 && C > 0, "bn_stats: bad args");
  YOLO_REQUIRE(C % 4 == 0, "bn_stats: C=%d must be a multiple of 4", C);
  const ColGeom g = col_geom(C / 4);
  dim3 grid(reduce_grid_x(P, g.rpp), (C / 4 + g.cw - 1) / g.cw);
  hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(256), 0, as_stream(stream), x, P, C, g.cw, g.rpp, stats);
  return check_launch("bn_stats_kernel");
}

extern "C" int yolo_bn_finalize_offset(double* stats, long long P, int C, const float* gamma, const float* beta,
                                       float eps, float momentum, int unbiased_moving_var, float* moving_mean,
                                       float* moving_var, float* scale, float* shift, float* save_mean,
                                       float* save_invstd, const unsigned* absmax, unsigned* bound,
                                       const float* mean_offset, void* stream) {
  YOLO_REQUIRE(stats && gamma && beta && scale && shift && save_mean && save_invstd && P > 0 && C > 0,
               "bn_finalize: bad args");
  YOLO_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), "bn_finalize: moving stats must come in pairs");
  static_assert(YOLO_BN_STAT_SLOTS == 64, "bn_finalize_kernel: one lane per replica slot");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, as_stream(stream), stats, P, C, gamma,
                     beta, eps, momentum, unbiased_moving_var, moving_mean, moving_var, scale, shift, save_mean,
                     save_invstd, absmax, bound, mean_offset);
  return check_launch("bn_finalize_kernel");
}

extern "C" int yolo_bn_finalize_bound(double* stats, long long P, int C, const float* gamma, const float* beta,
                                      float eps, float momentum, int unbiased_moving_var, float* moving_mean,
                                      float* moving_var, float* scale, float* shift, float* save_mean,
                                      float* save_invstd, const unsigned* absmax, unsigned* bound, void* stream) {
  return yolo_bn_finalize_offset(stats, P, C, gamma, beta, eps, momentum, unbiased_moving_var, moving_mean, moving_var,
                                 scale, shift, save_mean, save_invstd, absmax, bound, nullptr, stream);
}

extern "C" int yolo_bn_finalize(double* stats, long long P, int C, const float* gamma, const float* beta, float eps,
                                float momentum, int unbiased_moving_var, float* moving_mean, float* moving_var,
                                float* scale, float* shift, float* save_mean, float* save_invstd, void* stream) {
  return yolo_bn_finalize_bound(stats, P, C, gamma, beta, eps, momentum, unbiased_moving_var, moving_mean, moving_var,
                                scale, shift, save_mean, save_invstd, nullptr, nullptr, stream);
}

extern "C" int yolo_bn_fold_inference(int C, const float* gamma, const float* beta, const float* moving_mean,
                                      const float* moving_var, float eps, float* scale, float* shift, void* stream) {
  YOLO_REQUIRE(C > 0 && gamma && beta && moving_mean && moving_var && scale && shift, "bn_fold: bad args");
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), C, gamma, beta,
                     moving_mean, moving_var, eps, scale, shift);
  return check_launch("bn_fold_kernel");
}

extern "C" int yolo_bn_infer_bound(int C, const float* scale, const float* shift, const unsigned* absmax,
                                   unsigned* bound, void* stream) {
  YOLO_REQUIRE(C > 0 && scale && shift && absmax && bound, "bn_infer_bound: bad args");
  hipLaunchKernelGGL(bn_infer_bound_kernel, dim3(1), dim3(256), 0, as_stream(stream), C, scale, shift, absmax, bound);
  return check_launch("bn_infer_bound_kernel");
}

static int bn_act_fwd_impl(const float* x, long long P, int C, const float* scale, const float* shift, int act,
                           const float* residual, float* out, void* planes, const unsigned* bn_bound,
                           const float* residual_bound, float* out_bound, const void* residual_planes, void* stream);

extern "C" int yolo_bn_act_fwd_planes(const float* x, long long P, int C, const float* scale, const float* shift,
                                      int act, const float* residual, float* out, void* planes,
                                      const unsigned* bn_bound, const float* residual_bound, float* out_bound,
                                      void* stream) {
  return bn_act_fwd_impl(x, P, C, scale, shift, act, residual, out, planes, bn_bound, residual_bound, out_bound, nullptr,
                         stream);
}

// the residual given as PLANES (what its producer wrote for the convolutions; its bound comes from their header): the
// fp32 copy of a residual block's input then need not be written at all. C % 16 == 0.
extern "C" int yolo_bn_act_fwd_res_planes(const float* x, long long P, int C, const float* scale, const float* shift,
                                          int act, const void* residual_planes, float* out, void* planes,
                                          const unsigned* bn_bound, float* out_bound, void* stream) {
  YOLO_REQUIRE(residual_planes != nullptr && C % 16 == 0, "bn_act_fwd_res_planes: needs residual planes and C %% 16 == 0");
  return bn_act_fwd_impl(x, P, C, scale, shift, act, nullptr, out, planes, bn_bound, nullptr, out_bound, residual_planes,
                         stream);
}

static int bn_act_fwd_impl(const float* x, long long P, int C, const float* scale, const float* shift, int act,
                           const float* residual, float* out, void* planes, const unsigned* bn_bound,
                           const float* residual_bound, float* out_bound, const void* residual_planes, void* stream) {
  const unsigned char* rpl = reinterpret_cast<const unsigned char*>(residual_planes);
  YOLO_REQUIRE(x && scale && shift && (out || (planes && C % 8 == 0)) && P > 0 && C > 0, "bn_act_fwd: bad args");
  YOLO_REQUIRE(C % 4 == 0, "bn_act_fwd: C=%d must be a multiple of 4", C);
  YOLO_REQUIRE(act >= 0 && act <= 2, "bn_act_fwd: bad activation %d", act);
  YOLO_REQUIRE(planes == nullptr || C % 16 == 0, "bn_act_fwd: planes output needs C %% 16 == 0 (C=%d)", C);
  YOLO_REQUIRE(planes == nullptr || bn_bound != nullptr, "bn_act_fwd: planes output needs the bound from bn_finalize");
  YOLO_REQUIRE(planes == nullptr || residual == nullptr || residual_bound != nullptr,
               "bn_act_fwd: planes output with a residual needs the residual's bound");
  if (C % 8 == 0) {
    const long long rows = planes ? ((P + 15) / 16 + 1) * 16 : P;
    const RowGeom g = row_geom(rows, C);
    if (planes)
      hipLaunchKernelGGL(bn_act_fwd8_kernel<true>, dim3(g.gx, g.gy), dim3(256), 0, as_stream(stream), x, P, C, g.wpr,
                         g.blocks_per_wg, rows, scale, shift, act, residual, out,
                         reinterpret_cast<unsigned char*>(planes), bn_bound, residual_bound, out_bound, rpl);
    else
      hipLaunchKernelGGL(bn_act_fwd8_kernel<false>, dim3(g.gx, g.gy), dim3(256), 0, as_stream(stream), x, P, C, g.wpr,
                         g.blocks_per_wg, rows, scale, shift, act, residual, out, (unsigned char*)nullptr, bn_bound,
                         residual_bound, out_bound, rpl);
    return check_launch("bn_act_fwd8_kernel");
  }
  const long long n4 = P * (C / 4);
  hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(stream_grid(n4, 256)), dim3(256), 0, as_stream(stream), x, n4, C / 4,
                     scale, shift, act, residual, out);
  return check_launch("bn_act_fwd_kernel");
}

// ---- BatchNormalization + activation + MaxPooling2D(2, 2) in one pass (round 6) ----
// Darknet-19 / tiny-YOLOv3 put a 2x2 / stride-2 pool behind five of their conv + BN + LeakyReLU units
// (yolov2/models/backbone.py:42-60, yolov3/models/darknet.py:107-135). As three launches -- BN apply writing the fp32
// activation (the pool reads fp32), the pool, then an absmax + split pass that makes the planes of the pooled tensor for the
// next convolution -- the big pre-pool activation is written and read once each (709 MB for the first pool of YOLOv2-416 at
// bs 16) although nothing ever reads it again: BatchNorm's backward needs y, the pool's backward only the recorded winner.
// Here a thread owns 8 channels of one POOLED pixel: it normalises and activates the four window positions from y, takes the
// maximum (first maximum in row-major window order, v > best, as yolo_maxpool_fwd), records the winner as the flat offset
// into the (never written) activation tensor, and writes the pooled value as fp32 (optional) and as planes, scaled from the
// BatchNorm output's bound (a maximum of four values cannot exceed it). Rows [P, rows_total) of the planes are zero-filled.
__global__ __launch_bounds__(256) void bn_act_pool2_fwd_kernel(const float* __restrict__ y, long long P, long long rows_total,
                                                               int Ho, int Wo, int C, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, int act,
                                                               float* __restrict__ out, int* __restrict__ argmax,
                                                               unsigned char* __restrict__ planes,
                                                               const unsigned* __restrict__ bn_bound,
                                                               float* __restrict__ out_bound) {
  float psc = 1.f;
  const float b = bn_bound ? __builtin_bit_cast(float, bn_bound[0]) : 0.f;
  if (out_bound != nullptr && blockIdx.x == 0 && threadIdx.x == 0) out_bound[0] = b;
  if (planes != nullptr) psc = planes_begin(planes, P, C, b);
  const int C8 = C >> 3, W = 2 * Wo;
  const long long n = rows_total * C8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    // (16 consecutive threads = the 16 rows of one planes sub-block: whole 256-byte pieces per store; fp32 rows are read /
    // written 32 bytes per thread)
    const long long blk = i / (16 * C8);
    const int r16 = (int)(i & 15);
    const int g8 = (int)((i >> 4) % C8);
    const long long p = blk * 16 + r16;          // pooled pixel (b, ho, wo), row-major
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    if (p < P) {
      const f32x4 sc0 = reinterpret_cast<const f32x4*>(scale)[2 * g8], sc1 = reinterpret_cast<const f32x4*>(scale)[2 * g8 + 1];
      const f32x4 sh0 = reinterpret_cast<const f32x4*>(shift)[2 * g8], sh1 = reinterpret_cast<const f32x4*>(shift)[2 * g8 + 1];
      const int wo = (int)(p % Wo);
      const long long bh = p / Wo;               // b * Ho + ho: the window's input rows are 2 bh and 2 bh + 1
      const long long o00 = ((2 * bh) * W + 2 * wo) * C + g8 * 8;
      i32x4 a0, a1;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const long long o = o00 + ((long long)(t >> 1) * W + (t & 1)) * C;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(y + o), x1 = *reinterpret_cast<const f32x4*>(y + o + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float v0 = act_fwd(fmaf(sc0[k], x0[k], sh0[k]), act);
          const float v1 = act_fwd(fmaf(sc1[k], x1[k], sh1[k]), act);
          if (t == 0 || v0 > o0[k]) {
            o0[k] = v0;
            a0[k] = (int)o + k;
          }
          if (t == 0 || v1 > o1[k]) {
            o1[k] = v1;
            a1[k] = (int)o + 4 + k;
          }
        }
      }
      const long long e = p * C + g8 * 8;
      if (out != nullptr) {
        *reinterpret_cast<f32x4*>(out + e) = o0;
        *reinterpret_cast<f32x4*>(out + e + 4) = o1;
      }
      *reinterpret_cast<i32x4*>(argmax + e) = a0;
      *reinterpret_cast<i32x4*>(argmax + e + 4) = a1;
    }
    if (planes != nullptr) store_planes8(planes, p, g8, C, o0, o1, psc);
  }
}

extern "C" int yolo_bn_act_maxpool2x2_fwd(const float* y, int N, int Ho, int Wo, int C, const float* scale, const float* shift,
                                          int act, float* out, int* argmax, void* planes, const unsigned* bn_bound,
                                          float* out_bound, void* stream) {
  YOLO_REQUIRE(y && scale && shift && argmax && (out || planes) && N > 0 && Ho > 0 && Wo > 0, "bn_act_maxpool2x2_fwd: bad args");
  YOLO_REQUIRE(C > 0 && C % 8 == 0 && (planes == nullptr || C % 16 == 0),
               "bn_act_maxpool2x2_fwd: C=%d must be a multiple of 8 (16 with planes)", C);
  YOLO_REQUIRE(act >= 0 && act <= 2, "bn_act_maxpool2x2_fwd: bad activation %d", act);
  YOLO_REQUIRE(planes == nullptr || bn_bound != nullptr, "bn_act_maxpool2x2_fwd: planes output needs the bound from bn_finalize");
  YOLO_REQUIRE((long long)N * 4 * Ho * Wo * C < (1LL << 31), "bn_act_maxpool2x2_fwd: tensor too large for int32 argmax");
  const long long P = (long long)N * Ho * Wo;
  const long long rows = planes ? ((P + 15) / 16 + 1) * 16 : (P + 15) / 16 * 16;
  const long long n = rows * (C / 8);
  hipLaunchKernelGGL(bn_act_pool2_fwd_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), y, P, rows, Ho, Wo, C,
                     scale, shift, act, out, argmax, reinterpret_cast<unsigned char*>(planes), bn_bound, out_bound);
  return check_launch("bn_act_pool2_fwd_kernel");
}

extern "C" int yolo_bn_act_fwd(const float* x, long long P, int C, const float* scale, const float* shift, int act,
                               const float* residual, float* out, void* stream) {
  return yolo_bn_act_fwd_planes(x, P, C, scale, shift, act, residual, out, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int yolo_bn_act_bwd_reduce_fold_ld(const float* x, const float* dout, long long ld_dout, long long P, int C,
                                              const float* scale, const float* shift, const float* save_mean,
                                              const float* save_invstd, int act, double* red, unsigned* bound_aux,
                                              unsigned* tickets, void* stream) {
  YOLO_REQUIRE(x && dout && scale && shift && save_mean && save_invstd && red && P > 0 && C > 0,
               "bn_act_bwd_reduce: bad args");
  YOLO_REQUIRE(C % 4 == 0, "bn_act_bwd_reduce: C=%d must be a multiple of 4", C);
  YOLO_REQUIRE(ld_dout >= C && ld_dout % 4 == 0 && (reinterpret_cast<size_t>(dout) & 15) == 0,
               "bn_act_bwd_reduce: dout row pitch %lld must be >= C, a multiple of 4, dout 16-byte aligned", ld_dout);
  const long long ldd = ld_dout;
  const ColGeom g = col_geom(C / 4);
  int gx = reduce_grid_x(P, g.rpp);
  if (gx > YOLO_BN_RED_SLOTS) gx = YOLO_BN_RED_SLOTS;   // every workgroup owns one slot of `red`
  dim3 grid(gx, (C / 4 + g.cw - 1) / g.cw);
  if (tickets != nullptr && (int)grid.y * BN_FOLD_TICKETS <= YOLO_BN_FOLD_TICKET_WORDS) {
    // one launch: the last workgroups to arrive fold the slots (see the kernel)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<true>, grid, dim3(256), 0, as_stream(stream), x, dout, ldd, P, C, g.cw, g.rpp,
                       scale, shift, save_mean, save_invstd, act, red, bound_aux, tickets);
    return check_launch("bn_bwd_reduce_kernel<fold>");
  }
  hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, grid, dim3(256), 0, as_stream(stream), x, dout, ldd, P, C, g.cw, g.rpp,
                     scale, shift, save_mean, save_invstd, act, red, bound_aux, (unsigned*)nullptr);
  hipLaunchKernelGGL(bn_bwd_sum_kernel, dim3((C + 15) / 16), dim3(256), 0, as_stream(stream), C, red, gx, P, scale,
                     bound_aux);
  return check_launch("bn_bwd_reduce_kernel");
}

extern "C" int yolo_bn_act_bwd_reduce_bound_ld(const float* x, const float* dout, long long ld_dout, long long P, int C,
                                               const float* scale, const float* shift, const float* save_mean,
                                               const float* save_invstd, int act, double* red, unsigned* bound_aux,
                                               void* stream) {
  return yolo_bn_act_bwd_reduce_fold_ld(x, dout, ld_dout, P, C, scale, shift, save_mean, save_invstd, act, red, bound_aux,
                                        nullptr, stream);
}

extern "C" int yolo_bn_act_bwd_sum_partials(const float* partials, int nslots, long long P, int C, const float* scale,
                                            double* red, unsigned* bound_aux, void* stream) {
  YOLO_REQUIRE(partials && red && scale && nslots > 0 && P > 0 && C > 0, "bn_act_bwd_sum_partials: bad args");
  int chunk = 128;
  while ((nslots + chunk - 1) / chunk > YOLO_BN_RED_SLOTS) chunk *= 2;
  int nchunks = (nslots + chunk - 1) / chunk;
  if (nslots <= 256) { chunk = nslots; nchunks = 1; }
  YOLO_REQUIRE(nchunks == 1 || bound_aux != nullptr, "bn_act_bwd_sum_partials: %d slots need the ticket word of bound_aux", nslots);
  hipLaunchKernelGGL(bn_bwd_sum_part_kernel, dim3((C + 15) / 16, nchunks), dim3(256), 0, as_stream(stream), C, partials,
                     nslots, chunk, red, P, scale, bound_aux);
  return check_launch("bn_bwd_sum_part_kernel");
}

extern "C" int yolo_bn_act_bwd_reduce_bound(const float* x, const float* dout, long long P, int C, const float* scale,
                                            const float* shift, const float* save_mean, const float* save_invstd,
                                            int act, double* red, unsigned* bound_aux, void* stream) {
  return yolo_bn_act_bwd_reduce_bound_ld(x, dout, C, P, C, scale, shift, save_mean, save_invstd, act, red, bound_aux, stream);
}

extern "C" int yolo_bn_act_bwd_reduce(const float* x, const float* dout, long long P, int C, const float* scale,
                                      const float* shift, const float* save_mean, const float* save_invstd, int act,
                                      double* red, void* stream) {
  return yolo_bn_act_bwd_reduce_bound(x, dout, P, C, scale, shift, save_mean, save_invstd, act, red, nullptr, stream);
}

extern "C" int yolo_bn_act_bwd_apply_planes(const float* x, const float* dout, long long P, int C, const float* gamma,
                                            const float* scale, const float* shift, const float* save_mean,
                                            const float* save_invstd, int act, double* red, float* dgamma,
                                            float* dbeta, float* dx, void* planes, const unsigned* bound_aux,
                                            void* stream) {
  return yolo_bn_act_bwd_apply_planes_ld(x, dout, C, P, C, gamma, scale, shift, save_mean, save_invstd, act, red, dgamma, dbeta,
                                         dx, planes, bound_aux, stream);
}

extern "C" int yolo_bn_act_bwd_apply_planes_ld(const float* x, const float* dout, long long ld_dout, long long P, int C,
                                               const float* gamma, const float* scale, const float* shift,
                                               const float* save_mean, const float* save_invstd, int act, double* red,
                                               float* dgamma, float* dbeta, float* dx, void* planes,
                                               const unsigned* bound_aux, void* stream) {
  (void)gamma;
  YOLO_REQUIRE(ld_dout == C || (C % 8 == 0 && ld_dout > C && ld_dout % 4 == 0 && (reinterpret_cast<size_t>(dout) & 15) == 0),
               "bn_act_bwd_apply: a dout row pitch (%lld) other than C needs C %% 8 == 0, pitch %% 4 == 0, dout 16-byte aligned",
               ld_dout);
  const long long ldd = ld_dout;
  YOLO_REQUIRE(x && dout && scale && shift && save_mean && save_invstd && red && (dx || planes) && P > 0 && C > 0,
               "bn_act_bwd_apply: bad args");
  YOLO_REQUIRE(C % 4 == 0, "bn_act_bwd_apply: C=%d must be a multiple of 4", C);
  YOLO_REQUIRE(planes == nullptr || C % 16 == 0, "bn_act_bwd_apply: planes output needs C %% 16 == 0 (C=%d)", C);
  YOLO_REQUIRE(planes == nullptr || bound_aux != nullptr,
               "bn_act_bwd_apply: planes output needs the bound words of yolo_bn_act_bwd_reduce_bound");
  const long long n4 = P * (C / 4);
  hipStream_t st = as_stream(stream);
  const double* redsum = red + (long long)YOLO_BN_RED_SLOTS * 2 * C;
  if (C % 8 == 0) {
    const long long rows = planes ? ((P + 15) / 16 + 1) * 16 : P;
    const RowGeom g = row_geom(rows, C);
    unsigned char* pl = reinterpret_cast<unsigned char*>(planes);
#define YOLO_BWD8(PL, DX)                                                                                            \
  hipLaunchKernelGGL((bn_bwd_apply8_kernel<PL, DX>), dim3(g.gx, g.gy), dim3(256), 0, st, x, dout, ldd, P, C, g.wpr,  \
                     g.blocks_per_wg, rows, 1.0 / (double)P, scale, shift, save_mean, save_invstd, act, redsum, dx, pl,  \
                     bound_aux, dgamma, dbeta)
    if (planes && dx) YOLO_BWD8(true, true);
    else if (planes) YOLO_BWD8(true, false);
    else YOLO_BWD8(false, true);
#undef YOLO_BWD8
    return check_launch("bn_bwd_apply8_kernel");   // (dgamma / dbeta are added by the kernel itself)
  } else {
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_grid(n4, 256)), dim3(256), 0, st, x, dout, n4, C / 4,
                       1.0 / (double)P, scale, shift, save_mean, save_invstd, act, redsum, dx);
  }
  if (int rc = check_launch("bn_bwd_apply_kernel")) return rc;
  if (dgamma || dbeta) {
    hipLaunchKernelGGL(bn_bwd_params_kernel, dim3((C + 255) / 256), dim3(256), 0, st, C, redsum, dgamma, dbeta);
    return check_launch("bn_bwd_params_kernel");
  }
  return YOLO_OK;
}

extern "C" int yolo_bn_act_bwd_apply(const float* x, const float* dout, long long P, int C, const float* gamma,
                                     const float* scale, const float* shift, const float* save_mean,
                                     const float* save_invstd, int act, double* red, float* dgamma, float* dbeta,
                                     float* dx, void* stream) {
  YOLO_REQUIRE(dx != nullptr, "bn_act_bwd_apply: bad args");
  return yolo_bn_act_bwd_apply_planes(x, dout, P, C, gamma, scale, shift, save_mean, save_invstd, act, red, dgamma,
                                      dbeta, dx, nullptr, nullptr, stream);
}

extern "C" int yolo_act_fwd(const float* x, long long n, int act, float* out, void* stream) {
  YOLO_REQUIRE(x && out && n > 0, "act_fwd: bad args");
  hipLaunchKernelGGL(act_fwd_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), x, n, act, out);
  return check_launch("act_fwd_kernel");
}
extern "C" int yolo_act_bwd(const float* x, const float* dout, long long n, int act, float* dx, void* stream) {
  YOLO_REQUIRE(x && dout && dx && n > 0, "act_bwd: bad args");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), x, dout, n, act, dx);
  return check_launch("act_bwd_kernel");
}

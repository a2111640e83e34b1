// The RGB stem: Conv2D(32, 3, padding="same") on the 3-channel image, the first layer of every Darknet body
// (yolov3/models/backbone.py:60, yolov4/models/backbone.py:127, yolov2/models/backbone.py:44).
// K = 27 and a 32-channel output at full resolution make it a WRITE-bound layer (bs 32 at 416x416: 66 MB in,
// 709 MB out, 9.6 GFLOP): the implicit-GEMM kernels spend their time on 27-wide gathers, so the stem has kernels of its
// own. Default (round 3): stem_mfma_kernel -- the 28 x 32 product (27 taps + the bias row) on the fp32 matrix cores, 14 x
// v_mfma_f32_32x32x2_f32 per 32 pixels, BatchNorm statistics per lane (a lane ends up with ONE channel of 16 pixels), and
// with an epilogue flag the whole inference unit (folded BatchNormalization + activation + planes of the next layer).
// stem_conv3x3_kernel (rounds 1-2, YOLO_STEM_MFMA=0) is the direct FMA form: one pixel per lane, the filter through the
// scalar cache, exact fp32 FMA chains. The backward (stem_bn_bwd_wgrad_kernel) fuses the BatchNorm backward apply with the
// filter gradient.
#include "common.hpp"
#include "conv_args.hpp"
#include "act.hpp"
#include "planes.hpp"

namespace yolo {

typedef float f32x16s __attribute__((ext_vector_type(16)));
constexpr int STEM_CO = 32;
constexpr int STEM_K = 27;

__device__ float g_stem_wt[8 * (STEM_K + 1) * STEM_CO];   // ring of prepared stem filters (launch_stem_fwd)

// the filter as [j = (r*3+s)*3+ci][co] + a bias row, for the kernel that takes it through the SCALAR cache
__global__ __launch_bounds__(256) void stem_filter_prep_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                              float* __restrict__ wt) {
  for (int i = threadIdx.x; i < STEM_K * STEM_CO; i += 256) wt[(i % STEM_K) * STEM_CO + i / STEM_K] = w[i];   // w: [co][j]
  if (threadIdx.x < STEM_CO) wt[STEM_K * STEM_CO + threadIdx.x] = bias != nullptr ? bias[threadIdx.x] : 0.f;
}

// SREG: the filter comes from `w` = the prepared [28][32] copy by scalar loads (uniform addresses: s_load into SGPRs, the
// FMAs take it as their scalar operand) instead of 216 broadcast ds_read_b128 per pixel, which had the CU's LDS port
// as the bottleneck (216 x 8 clocks per 64 pixels and wave = 280 us for the bs-32 tensor; the FMAs need 70 us)
template <bool STATS, bool SREG>
__global__ __launch_bounds__(256) void stem_conv3x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          double* __restrict__ stats, unsigned* __restrict__ absmax,
                                                          int H, int W, int pad_t, int pad_l, int M) {
  __shared__ __attribute__((aligned(16))) float w_s[SREG ? 1 : STEM_K + 1][STEM_CO];   // [j = (r*3+s)*3+ci][co]; row 27 = bias
  if constexpr (!SREG) {
    for (int i = threadIdx.x; i < STEM_K * STEM_CO; i += 256) w_s[i % STEM_K][i / STEM_K] = w[i];   // w: [co][j]
    if (threadIdx.x < STEM_CO) w_s[STEM_K][threadIdx.x] = bias != nullptr ? bias[threadIdx.x] : 0.f;
  }
  __shared__ __attribute__((aligned(16))) float t_s[4][64 * STEM_CO];   // per-wave output staging (8 KB each)
  __syncthreads();
  float s1[STATS ? STEM_CO : 1], s2[STATS ? STEM_CO : 1], mx[STATS ? STEM_CO : 1];
  if (STATS) {
#pragma unroll
    for (int c = 0; c < STEM_CO; ++c) s1[c] = s2[c] = mx[c] = 0.f;
  }
  const int HW = H * W;
  const int stride = gridDim.x * 256;
  // whole waves iterate together (the output staging is per wave): lanes past the end compute pixel M-1 again
  // and are masked at the store / statistics
  for (int pb = blockIdx.x * 256 + (threadIdx.x & ~63); pb < M; pb += stride) {
    const int p_raw = pb + (threadIdx.x & 63);
    const bool live = p_raw < M;
    const int p = p_raw;
    const int pc = live ? p_raw : M - 1;
    const int n = pc / HW;
    const int rem = pc - n * HW;
    const int yy = rem / W;
    const int xx = rem - yy * W;
    float patch[STEM_K];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int iy = yy + r - pad_t, ix = xx + s - pad_l;
        const bool ok = ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);
        const float* q = x + ((long long)(n * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float v = q[c];
          patch[(r * 3 + s) * 3 + c] = ok ? v : 0.f;
        }
      }
    // (the filter is re-read from LDS for every pixel: without the clobber hipcc hoists all 864 values out of
    // the pixel loop and spills them)
    asm volatile("" ::: "memory");
    // packed fp32 FMAs (v_pk_fma_f32: two channels per lane and instruction): the 864 FMAs per pixel are the
    // kernel's critical resource
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc2[STEM_CO / 2];
    if constexpr (SREG) {
#pragma unroll
      for (int c2 = 0; c2 < STEM_CO / 2; ++c2) acc2[c2] = f32x2{w[STEM_K * STEM_CO + 2 * c2], w[STEM_K * STEM_CO + 2 * c2 + 1]};
#pragma unroll
      for (int j = 0; j < STEM_K; ++j) {
        const f32x2 pj = {patch[j], patch[j]};
#pragma unroll
        for (int c2 = 0; c2 < STEM_CO / 2; ++c2)   // (uniform addresses with constant offsets: scalar loads)
          acc2[c2] = __builtin_elementwise_fma(pj, f32x2{w[j * STEM_CO + 2 * c2], w[j * STEM_CO + 2 * c2 + 1]}, acc2[c2]);
      }
    } else {
#pragma unroll
    for (int c4 = 0; c4 < STEM_CO / 4; ++c4) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(&w_s[STEM_K][c4 * 4]);
      acc2[2 * c4] = f32x2{bv[0], bv[1]};
      acc2[2 * c4 + 1] = f32x2{bv[2], bv[3]};
    }
#pragma unroll
    for (int j = 0; j < STEM_K; ++j) {
      const f32x2 pj = {patch[j], patch[j]};
#pragma unroll
      for (int c4 = 0; c4 < STEM_CO / 4; ++c4) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(&w_s[j][c4 * 4]);   // same address in every lane: broadcast
        acc2[2 * c4] = __builtin_elementwise_fma(pj, f32x2{wv[0], wv[1]}, acc2[2 * c4]);
        acc2[2 * c4 + 1] = __builtin_elementwise_fma(pj, f32x2{wv[2], wv[3]}, acc2[2 * c4 + 1]);
      }
    }
    }
    float acc[STEM_CO];
#pragma unroll
    for (int c = 0; c < STEM_CO; ++c) acc[c] = acc2[c >> 1][c & 1];
    // transpose through LDS so that one store instruction writes 1 KB of consecutive output (8 pixels x 128 B):
    // a lane writing its own pixel's 128 B would issue 64 partial lines per instruction
    {
      float* t = &t_s[threadIdx.x >> 6][0];
      const int lane = threadIdx.x & 63;
#pragma unroll
      for (int c4 = 0; c4 < STEM_CO / 4; ++c4) {
        const f32x4 v = {acc[c4 * 4], acc[c4 * 4 + 1], acc[c4 * 4 + 2], acc[c4 * 4 + 3]};
        // pixel-major rows of 32 floats, 16-byte units rotated by the pixel so that the 64 lanes' writes spread over the banks
        *reinterpret_cast<f32x4*>(t + lane * STEM_CO + (((c4 + lane) & 7) << 2)) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (one wave: its own LDS writes are visible to itself)
      const int p0 = p - lane;                    // first pixel of this wave's 64 (consecutive: p = base + lane)
      const int valid = M - p0 < 64 ? M - p0 : 64;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int px = i * 8 + (lane >> 3), c4 = lane & 7;
        const f32x4 v = *reinterpret_cast<const f32x4*>(t + px * STEM_CO + (((c4 + px) & 7) << 2));
        if (px < valid)
          __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(y + (long long)(p0 + px) * STEM_CO + c4 * 4));
      }
    }
    if (STATS && live) {
#pragma unroll
      for (int c = 0; c < STEM_CO; ++c) {
        s1[c] += acc[c];
        s2[c] = fmaf(acc[c], acc[c], s2[c]);
        mx[c] = fmaxf(mx[c], fabsf(acc[c]));
      }
    }
  }
  if (STATS) {
    // per-wave totals of this wave's pixels (~40 per lane), then one fp64 atomic per channel and wave into the
    // replica slot of the workgroup (YOLO_BN_STAT_SLOTS, as the implicit-GEMM epilogues do)
    const int lane = threadIdx.x & 63;
    double* slot = stats != nullptr ? stats + (long long)(blockIdx.x & (YOLO_BN_STAT_SLOTS - 1)) * 2 * STEM_CO : nullptr;
#pragma unroll
    for (int c = 0; c < STEM_CO; ++c) {
      float a1 = s1[c], a2 = s2[c], am = mx[c];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        a1 += __shfl_xor(a1, o, 64);
        a2 += __shfl_xor(a2, o, 64);
        am = fmaxf(am, __shfl_xor(am, o, 64));
      }
      if (lane == 0) {
        if (slot != nullptr) {
          atomicAdd(&slot[c], (double)a1);
          atomicAdd(&slot[STEM_CO + c], (double)a2);
        }
        if (absmax != nullptr && __builtin_bit_cast(unsigned, am) > absmax[c])
          atomicMax(&absmax[c], __builtin_bit_cast(unsigned, am));
      }
    }
  }
}


// ---- backward of the stem unit: BatchNorm + activation backward APPLY fused with the filter gradient ----
// The unfused path writes d(conv out) of the 32-channel full-resolution tensor (709 MB at bs 32) only for the stem's
// filter gradient to read it again (0.40 + 0.35 ms). Here every wave computes d(conv out) of 16 pixels x 32 channels
// exactly as bn_bwd_apply8_kernel does, passes it through a 2 KB strip of LDS into the A operand of the fp32 MFMA
// (v_mfma_f32_32x32x2_f32: channels x pixel pairs), gathers the 27 taps of those pixels from the image as the B
// operand, and keeps the 32 x 27 filter gradient in 16 accumulator registers; nothing of size P x 32 is written.
// Workgroup partials go to `partial` ([gridDim.x][1024] in accumulator order), stem_wgrad_sum_kernel adds them to dW.
__global__ __launch_bounds__(256) void stem_bn_bwd_wgrad_kernel(
    const float* __restrict__ y, const float* __restrict__ dout, const float* __restrict__ img, long long P, int H, int W,
    int pad_t, int pad_l, double invP, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ smean, const float* __restrict__ sinv, int act, const double* __restrict__ redsum,
    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ partial) {
  constexpr int C = STEM_CO, TLD = 36;   // strip rows of 36 floats: 16-byte aligned, conflict-free column reads
  __shared__ __attribute__((aligned(16))) float strip[4][16 * TLD];
  __shared__ float wsum[4][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;   // apply role: pixel r of the block, channels 8g .. 8g+7
  float sc[8], sh[8], mu[8], cb[8], ck[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = g * 8 + k;
    sc[k] = scale[c];
    sh[k] = shift[c];
    mu[k] = smean[c];
    const float mdz = (float)(redsum[c] * invP);
    const float mdzx = (float)(redsum[C + c] * invP);
    cb[k] = -sc[k] * sinv[c] * mdzx;
    ck[k] = -sc[k] * mdz;
    if (blockIdx.x == 0 && wave == 0 && r == 0) {   // dbeta = sum dz, dgamma = sum dz xhat (as bn_bwd_apply8_kernel)
      if (dbeta != nullptr) dbeta[c] += (float)redsum[c];
      if (dgamma != nullptr) dgamma[c] += (float)redsum[C + c];
    }
  }
  // MFMA role: k-half kh = lane / 32 (pixel 2i + kh of the block), column n = lane % 32 = (r3 * 3 + s3) * 3 + ci
  const int kh = lane >> 5, n = lane & 31;
  const bool ncol = n < STEM_K;
  const int tap = n / 3, ci = n - tap * 3;
  const int dyo = tap / 3 - pad_t, dxo = tap % 3 - pad_l;
  const int HW = H * W;
  f32x16s acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  float* st = &strip[wave][0];
  const long long nblk = (P + 15) / 16;
  for (long long blk = (long long)blockIdx.x * 4 + wave; blk < nblk; blk += (long long)gridDim.x * 4) {
    const long long p0 = blk * 16;
    // ---- d(conv out) of my pixel / my 8 channels ----
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    if (p0 + r < P) {
      const long long e = (p0 + r) * C + g * 8;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(y + e), x1 = *reinterpret_cast<const f32x4*>(y + e + 4);
      const f32x4 d0 = *reinterpret_cast<const f32x4*>(dout + e), d1 = *reinterpret_cast<const f32x4*>(dout + e + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dz0 = d0[k] * act_grad(fmaf(sc[k], x0[k], sh[k]), act);
        const float dz1 = d1[k] * act_grad(fmaf(sc[4 + k], x1[k], sh[4 + k]), act);
        o0[k] = fmaf(sc[k], dz0, fmaf(cb[k], x0[k] - mu[k], ck[k]));
        o1[k] = fmaf(sc[4 + k], dz1, fmaf(cb[4 + k], x1[k] - mu[4 + k], ck[4 + k]));
      }
    }
    *reinterpret_cast<f32x4*>(st + r * TLD + g * 8) = o0;
    *reinterpret_cast<f32x4*>(st + r * TLD + g * 8 + 4) = o1;
    // ---- the taps of the block's pixels: pixel p0 + 2i + kh, column n ----
    const int nimg0 = (int)(p0 / HW);
    const int rem0 = (int)(p0 - (long long)nimg0 * HW);
    const int y0 = rem0 / W, x0p = rem0 - y0 * W;
    float bv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = 2 * i + kh;
      int xx = x0p + k, yy = y0, nn = nimg0;
      while (xx >= W) { xx -= W; ++yy; }
      while (yy >= H) { yy -= H; ++nn; }
      const int iy = yy + dyo, ix = xx + dxo;
      const bool ok = ncol && (p0 + k < P) && ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);
      // (!ok covers pixels past P in the last block, whose nn would be N: every index is clamped, the load stays
      // inside the image tensor for any N * H * W)
      const float* q = img + ((long long)((ok ? nn : 0) * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * 3 + (ncol ? ci : 0);
      const float v = *q;
      bv[i] = ok ? v : 0.f;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (one wave: its own strip writes are visible to itself)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float av = st[(2 * i + kh) * TLD + n];   // A[m = channel n][k = kh]
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[i], acc, 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // strip reads done before the next block's writes
  }
  // ---- workgroup partial: the four waves' accumulators added in wave order ----
#pragma unroll
  for (int q = 0; q < 16; ++q) wsum[wave][q * 64 + lane] = acc[q];
  __syncthreads();
  for (int e = threadIdx.x; e < 1024; e += 256)
    partial[(long long)blockIdx.x * 1024 + e] = (wsum[0][e] + wsum[1][e]) + (wsum[2][e] + wsum[3][e]);
}

// dW[c][j] += sum over the workgroup partials, in partial order (no atomics: reproducible). Accumulator element
// (q, lane): channel m = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5), column j = lane & 31 (j < 27 exist).
__global__ __launch_bounds__(1024) void stem_wgrad_sum_kernel(const float* __restrict__ partial, int nparts,
                                                             float* __restrict__ dw) {
  __shared__ float red[16][64];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;   // 16 groups of partials
  float s = 0.f;
  for (int pidx = grp; pidx < nparts; pidx += 16) s += partial[(long long)pidx * 1024 + e];
  red[grp][threadIdx.x & 63] = s;
  __syncthreads();
  if (grp == 0) {
    float t = 0.f;
#pragma unroll
    for (int gI = 0; gI < 16; ++gI) t += red[gI][threadIdx.x];
    const int q = e >> 6, lane = e & 63;
    const int m = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5), j = lane & 31;
    if (j < STEM_K) dw[m * STEM_K + j] += t;
  }
}

// ---- the stem's inference unit in one launch (yolo_stem_fwd_infer_unit) ----
// At inference sizes (bs 1: 173 k pixels) the training kernel above -- one pixel and 864 dependent FMAs per lane -- is a
// latency chain (63 us). Here a wave owns ONE 8-channel group of 64 pixels (its 27 x 8 filter slice and folded scale / shift
// arrive by scalar loads), a workgroup the four groups of those pixels: a quarter of the chain, four times the waves. The
// folded BatchNormalization and the activation are applied in registers and the result leaves as the planes of the next
// convolution, scaled by the a-priori bound K max|image| + D (as conv_split_reduce_kernel does); y (fp32) only if somebody
// reads it; one word of max|result| per workgroup goes to out_words.
struct StemEpi {
  const float* scale;
  const float* shift;
  int act;
  unsigned char* planes;
  const float* pred;
  const unsigned* in_words;
  int in_n;
  unsigned* out_words;
};
template <bool STATS, bool EPI>
__global__ __launch_bounds__(256) void stem_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                       float* __restrict__ y, double* __restrict__ stats,
                                                       unsigned* __restrict__ absmax, int H, int W, int pad_t, int pad_l,
                                                       int M, const StemEpi ep) {
  // The 28 x 32 product (27 taps + a constant 1 for the bias row) on the fp32 matrix cores: v_mfma_f32_32x32x2_f32, rows =
  // 32 pixels, columns = the 32 channels, 14 steps of two taps. Lane (m = lane & 31, kh = lane >> 5) supplies pixel m's taps
  // 2i + kh, holds filter rows 2i + kh of channel m in 14 registers for the whole launch, and ends with channel m of 16
  // pixels in its accumulator: the folded scale / shift are per-lane scalars, a store instruction writes two whole 128-byte
  // pixel rows, and nothing waits for scalar loads (the FMA form with the filter in SGPRs took 47 us at bs 1).
  constexpr int TLD = 36;                          // strip rows of 36 floats (16-byte aligned, conflict-free 8-float reads)
  __shared__ __attribute__((aligned(16))) float strip[4][32 * TLD];
  __shared__ float s_max[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 31, kh = lane >> 5;
  [[maybe_unused]] float psc = 1.f, sc = 1.f, sh = 0.f;
  if constexpr (EPI) {
    float ib = 0.f;
    for (int i = lane; i < ep.in_n; i += 64) ib = fmaxf(ib, __builtin_bit_cast(float, ep.in_words[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ib = fmaxf(ib, __shfl_xor(ib, o, 64));
    const float bnd = (ep.pred[0] * ib + ep.pred[1]) * 1.001f + 1e-30f;
    psc = planes_scale_from_bound(__builtin_bit_cast(unsigned, bnd));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      unsigned* header = reinterpret_cast<unsigned*>(ep.planes + planes_body_bytes(M, STEM_CO));
      header[0] = __builtin_bit_cast(unsigned, bnd);
      reinterpret_cast<float*>(header)[1] = psc;
      reinterpret_cast<float*>(header)[2] = 1.f / psc;
    }
    sc = ep.scale[m];
    sh = ep.shift[m];
  }
  float bw[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) bw[i] = wt[(2 * i + kh) * STEM_CO + m];   // (row 27 = bias, met by the constant-1 tap)
  // per-lane constants of tap j = 2i + kh = (r * 3 + s) * 3 + ci (both candidates are compile-time constants): row / column
  // offset for the bounds test, float offset from the pixel's own first channel
  int toff[14];
  short tdr[14], tds[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    const int j0 = 2 * i, j1 = 2 * i + 1;
    const int r = kh ? j1 / 9 : j0 / 9, s3 = kh ? (j1 / 3) % 3 : (j0 / 3) % 3, ci = kh ? j1 % 3 : j0 % 3;
    tdr[i] = (short)((2 * i + kh) < STEM_K ? r - pad_t : -20000);   // (tap 27: never in range)
    tds[i] = (short)(s3 - pad_l);
    toff[i] = ((r - pad_t) * W + (s3 - pad_l)) * 3 + ci;
  }
  // training form: channel m's sum / sum of squares / max|.| of this lane's pixels (fp32 inside a tile, fp64 across tiles)
  [[maybe_unused]] double d1 = 0.0, d2 = 0.0;
  [[maybe_unused]] float mxs = 0.f;
  const int HW = H * W;
  float vmax = 0.f;
  float* st = &strip[wave][0];
  const int ntiles = (M + 31) / 32;
  for (int t = blockIdx.x * 4 + wave; t < ntiles; t += gridDim.x * 4) {
    const int p0 = t * 32;
    const int pm = p0 + m < M ? p0 + m : M - 1;
    const int n = pm / HW;
    const int rem = pm - n * HW;
    const int yy = rem / W, xx = rem - yy * W;
    // pixel (n, yy, xx) IS linear pixel pm: tap (dr, ds, ci) of it sits at float pm * 3 + toff -- one add per tap (the first
    // version rebuilt a 64-bit ((n * H + iy) * W + ix) * 3 + ci per tap: 2000 cycles of integer multiplies per tile where
    // the MFMAs need 900)
    float av[14];
    const int base = pm * 3;
#pragma unroll
    for (int i = 0; i < 14; ++i) {
      const bool ok = ((unsigned)(yy + tdr[i]) < (unsigned)H) && ((unsigned)(xx + tds[i]) < (unsigned)W);
      const float v = x[ok ? base + toff[i] : 0];
      av[i] = ok ? v : 0.f;
    }
    if (kh) av[13] = 1.f;   // tap 27 does not exist: the constant that meets the bias row of the prepared filter
    f32x16s acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
    for (int i = 0; i < 14; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bw[i], acc, 0, 0, 0);
    // acc[q] = channel m of pixel p0 + (q & 3) + 8 (q >> 2) + 4 kh
    if constexpr (EPI) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * kh;
        const float v = act_fwd(fmaf(sc, acc[q], sh), ep.act);
        st[row * TLD + m] = v;
        const bool live = p0 + row < M;
        vmax = fmaxf(vmax, live ? fabsf(v) : 0.f);
        if (y != nullptr && live) y[(long long)(p0 + row) * STEM_CO + m] = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (one wave: its own strip writes are visible to itself)
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int u = lane + 64 * it, px = u >> 2, g8 = u & 3;
        const f32x4 o0 = *reinterpret_cast<const f32x4*>(st + px * TLD + g8 * 8);
        const f32x4 o1 = *reinterpret_cast<const f32x4*>(st + px * TLD + g8 * 8 + 4);
        if (p0 + px < M) store_planes8(ep.planes, p0 + px, g8, STEM_CO, o0, o1, psc);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // strip reads done before the next tile's writes
    } else {
      // (branch-free: a per-row "if" turns into sixteen exec-mask blocks; rows past the end count as zeros)
      const bool full = p0 + 32 <= M;
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * kh;
        const float v = (full || p0 + row < M) ? acc[q] : 0.f;
        if (STATS) {
          t1 += v;
          t2 = fmaf(v, v, t2);
          mxs = fmaxf(mxs, fabsf(v));
        }
      }
      // through the wave's strip so that one store instruction writes 1 KB of consecutive output (8 pixels x 128 B) instead
      // of 64 dwords in two 128-byte pieces
#pragma unroll
      for (int q = 0; q < 16; ++q) st[((q & 3) + 8 * (q >> 2) + 4 * kh) * TLD + m] = acc[q];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (one wave: its own strip writes are visible to itself)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int px = it * 8 + (lane >> 3), c4 = lane & 7;
        const f32x4 v = *reinterpret_cast<const f32x4*>(st + px * TLD + c4 * 4);
        if (full || p0 + px < M) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(y + (long long)(p0 + px) * STEM_CO + c4 * 4));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // strip reads done before the next tile's writes
      if (STATS) {
        d1 += (double)t1;
        d2 += (double)t2;
      }
    }
  }
  if constexpr (STATS) {
    // lanes m and m + 32 hold the same channel: one fp64 atomic per channel and wave into the workgroup's replica slot
    d1 += __shfl_xor(d1, 32, 64);
    d2 += __shfl_xor(d2, 32, 64);
    mxs = fmaxf(mxs, __shfl_xor(mxs, 32, 64));
    if (kh == 0) {
      if (stats != nullptr) {
        double* slot = stats + (long long)(blockIdx.x & (YOLO_BN_STAT_SLOTS - 1)) * 2 * STEM_CO;
        atomicAdd(&slot[m], d1);
        atomicAdd(&slot[STEM_CO + m], d2);
      }
      if (absmax != nullptr && __builtin_bit_cast(unsigned, mxs) > absmax[m]) atomicMax(&absmax[m], __builtin_bit_cast(unsigned, mxs));
    }
  }
  if constexpr (!EPI) return;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
  if (lane == 0) s_max[wave] = vmax;
  __syncthreads();
  if (threadIdx.x == 0) ep.out_words[blockIdx.x] = __builtin_bit_cast(unsigned, fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3])));
}

static int stem_mfma_per_cu() {
  // the fp32 matrix-core form (stem_mfma_kernel) is the default: 8 workgroups per CU; YOLO_STEM_MFMA=0 selects the FMA kernel
  // (same-box A/B of the whole step: 29.98 -> 29.87 ms)
  static const int v = [] { const char* e = getenv("YOLO_STEM_MFMA"); return e ? atoi(e) : 8; }();
  return v;
}

bool stem_fwd_supported(const yolo_conv_desc* d) {
  static const bool on = [] { const char* e = getenv("YOLO_STEM_DIRECT"); return !(e && atoi(e) == 0); }();
  // (the FMA kernel below ~1 M pixels - bs 1 inference - gives a lane a single pixel and is bound by the latency of its
  // filter loads: the implicit-GEMM kernel is faster there; the matrix-core form has no such floor)
  const long long M = (long long)d->N * d->H * d->W;
  return on && d->Cin == 3 && d->Cout == STEM_CO && d->kh == 3 && d->kw == 3 && d->sh == 1 && d->sw == 1 &&
         d->Ho == d->H && d->Wo == d->W && (stem_mfma_per_cu() > 0 || M >= (1LL << 20)) && M < (1LL << 31) - (1 << 20);
}

int launch_stem_fwd(const yolo_conv_desc* d, const float* x, const float* w, const float* bias, float* y, double* stats,
                    unsigned* absmax, hipStream_t st) {
  const int M = d->N * d->H * d->W;
  const bool st_on = stats != nullptr || absmax != nullptr;
  // one resident round: every lane walks M / (grid * 256) pixels (a second, partial round would idle most CUs)
  // YOLO_STEM_SREG=0: the filter broadcast from LDS (round 1's kernel) instead of the scalar cache
  static const bool sreg = [] { const char* e = getenv("YOLO_STEM_SREG"); return !(e && atoi(e) == 0); }();
  static float* wt_ring = nullptr;   // 8 prepared filters (3.5 KB each), used round robin: launches in flight never share one
  static unsigned wt_next = 0;        // (a __device__ array of the library: nothing is allocated at run time)
  if (sreg && wt_ring == nullptr &&
      hipGetSymbolAddress(reinterpret_cast<void**>(&wt_ring), HIP_SYMBOL(g_stem_wt)) != hipSuccess) {
    set_error("stem: hipGetSymbolAddress of the filter scratch failed");
    return YOLO_ERR_LAUNCH;
  }
  static int per_cu[2] = {0, 0};
  if (per_cu[st_on] == 0) {
    int n = 0;
    const void* fn = sreg ? (st_on ? reinterpret_cast<const void*>(&stem_conv3x3_kernel<true, true>)
                                   : reinterpret_cast<const void*>(&stem_conv3x3_kernel<false, true>))
                          : (st_on ? reinterpret_cast<const void*>(&stem_conv3x3_kernel<true, false>)
                                   : reinterpret_cast<const void*>(&stem_conv3x3_kernel<false, false>));
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 256, 0) != hipSuccess || n < 1) n = 2;
    per_cu[st_on] = n;
  }
  static int cus = 0;   // (all devices of a node are the same part; queried once: the call is slow)
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
           prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  int grid = (M + 255) / 256;
  if (grid > per_cu[st_on] * cus) grid = per_cu[st_on] * cus;
  const int mfma = stem_mfma_per_cu();
  if (sreg && mfma) {
    float* wt = wt_ring + (size_t)(wt_next++ & 7) * (STEM_K + 1) * STEM_CO;
    hipLaunchKernelGGL(stem_filter_prep_kernel, dim3(1), dim3(256), 0, st, w, bias, wt);
    int g2 = ((M + 31) / 32 + 3) / 4;
    if (g2 > mfma * cus) g2 = mfma * cus;   // (YOLO_STEM_MFMA = workgroups per CU)
    if (st_on)
      hipLaunchKernelGGL((stem_mfma_kernel<true, false>), dim3(g2), dim3(256), 0, st, x, wt, y, stats, absmax, d->H, d->W,
                         d->pad_t, d->pad_l, M, StemEpi{});
    else
      hipLaunchKernelGGL((stem_mfma_kernel<false, false>), dim3(g2), dim3(256), 0, st, x, wt, y, stats, absmax, d->H, d->W,
                         d->pad_t, d->pad_l, M, StemEpi{});
    return check_launch("stem_mfma_kernel");
  }
  if (sreg) {
    float* wt = wt_ring + (size_t)(wt_next++ & 7) * (STEM_K + 1) * STEM_CO;
    hipLaunchKernelGGL(stem_filter_prep_kernel, dim3(1), dim3(256), 0, st, w, bias, wt);
    if (st_on)
      hipLaunchKernelGGL((stem_conv3x3_kernel<true, true>), dim3(grid), dim3(256), 0, st, x, wt, bias, y, stats, absmax, d->H,
                         d->W, d->pad_t, d->pad_l, M);
    else
      hipLaunchKernelGGL((stem_conv3x3_kernel<false, true>), dim3(grid), dim3(256), 0, st, x, wt, bias, y, stats, absmax,
                         d->H, d->W, d->pad_t, d->pad_l, M);
  } else if (st_on)
    hipLaunchKernelGGL((stem_conv3x3_kernel<true, false>), dim3(grid), dim3(256), 0, st, x, w, bias, y, stats, absmax, d->H, d->W,
                       d->pad_t, d->pad_l, M);
  else
    hipLaunchKernelGGL((stem_conv3x3_kernel<false, false>), dim3(grid), dim3(256), 0, st, x, w, bias, y, stats, absmax, d->H,
                       d->W, d->pad_t, d->pad_l, M);
  return check_launch("stem_conv3x3_kernel");
}

// one word of max|x| per workgroup (no atomics, nothing to zero): the bound of a tensor nobody recorded one for -- the image
__global__ __launch_bounds__(256) void absmax_words_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ words) {
  __shared__ float wmax[4];
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) words[blockIdx.x] = __builtin_bit_cast(unsigned, fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])));
}

// workgroups of the fused stem backward (= partials of 4 KB each in the caller's scratch): two resident rounds' worth
constexpr int STEM_BWD_GRID = 1024;
size_t stem_bwd_scratch_bytes() { return (size_t)STEM_BWD_GRID * 1024 * sizeof(float); }

int launch_stem_bn_bwd_wgrad(const yolo_conv_desc* d, const float* y, const float* dout, const float* img,
                             const float* scale, const float* shift, const float* smean, const float* sinv, int act,
                             const double* redsum, float* dgamma, float* dbeta, float* dw, float* scratch,
                             size_t scratch_bytes, hipStream_t st) {
  const long long P = (long long)d->N * d->H * d->W;
  if (scratch == nullptr || scratch_bytes < stem_bwd_scratch_bytes()) {
    set_error("stem backward: scratch of %zu bytes needed", stem_bwd_scratch_bytes());
    return YOLO_ERR_INVALID_ARG;
  }
  long long grid = (P / 16 + 3) / 4;
  if (grid > STEM_BWD_GRID) grid = STEM_BWD_GRID;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(stem_bn_bwd_wgrad_kernel, dim3((unsigned)grid), dim3(256), 0, st, y, dout, img, P, d->H, d->W, d->pad_t,
                     d->pad_l, 1.0 / (double)P, scale, shift, smean, sinv, act, redsum, dgamma, dbeta, scratch);
  hipLaunchKernelGGL(stem_wgrad_sum_kernel, dim3(16), dim3(1024), 0, st, scratch, (int)grid, dw);
  return check_launch("stem_bn_bwd_wgrad_kernel");
}

bool stem_infer_supported(const yolo_conv_desc* d) {
  const long long M = (long long)d->N * d->H * d->W;
  return d->Cin == 3 && d->Cout == STEM_CO && d->kh == 3 && d->kw == 3 && d->sh == 1 && d->sw == 1 && d->Ho == d->H &&
         d->Wo == d->W && M < (1LL << 31) - (1 << 20);
}

int launch_stem_filter_prep(const float* w, const float* bias, float* wt, hipStream_t st) {
  hipLaunchKernelGGL(stem_filter_prep_kernel, dim3(1), dim3(256), 0, st, w, bias, wt);
  return check_launch("stem_filter_prep_kernel");
}

int launch_absmax_words(const float* x, long long n, unsigned* words, int* n_words, hipStream_t st) {
  long long g = (n + 256 * 16 - 1) / (256 * 16);   // ~16 values per thread
  if (g > 256) g = 256;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(absmax_words_kernel, dim3((unsigned)g), dim3(256), 0, st, x, n, words);
  *n_words = (int)g;
  return check_launch("absmax_words_kernel");
}

int launch_stem_infer(const yolo_conv_desc* d, const float* x, const float* wt, float* y, const StemEpiArgs& e, int* out_n,
                      hipStream_t st) {
  const int M = d->N * d->H * d->W;
  int grid = ((M + 31) / 32 + 3) / 4;            // one 32-pixel tile per wave
  if (grid > YOLO_INFER_BOUND_WORDS) grid = YOLO_INFER_BOUND_WORDS;   // (one word of max|result| per workgroup)
  StemEpi ep;
  ep.scale = e.scale; ep.shift = e.shift; ep.act = e.act;
  ep.planes = reinterpret_cast<unsigned char*>(e.planes);
  ep.pred = e.pred; ep.in_words = e.in_words; ep.in_n = e.in_n; ep.out_words = e.out_words;
  hipLaunchKernelGGL((stem_mfma_kernel<false, true>), dim3(grid), dim3(256), 0, st, x, wt, y, (double*)nullptr,
                     (unsigned*)nullptr, d->H, d->W, d->pad_t, d->pad_l, M, ep);
  *out_n = grid;
  return check_launch("stem_mfma_kernel(infer)");
}

}  // namespace yolo

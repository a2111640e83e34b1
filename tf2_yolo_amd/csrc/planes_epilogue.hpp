// Epilogue shared by the planes convolution kernels (conv_planes.hip, conv_win.hip): undo the two power-of-two
// operand scales, add the bias, optionally accumulate into dst, store (128-row-multiple tiles: through LDS as
// dwordx4 rows), and reduce the BatchNorm statistics / per-channel max|y| of the tile.
// Must be entered after every wave is done with the LDS region it reuses (vmcnt(0) + barrier by the caller);
// LDS_BYTES = bytes of `smem` the epilogue may use.
#pragma once
#include "planes.hpp"

namespace yolo {

struct NoStamp {
  __device__ __forceinline__ void operator()(int) const {}
};

template <int BM, int BN, int WGM, int WGN, int LDS_BYTES, int DBG = 0, class STAMP = NoStamp>
__device__ __forceinline__ void planes_epilogue(const GatherConvArgs& a, f32x16 (&acc)[BM / WGM / 32][BN / WGN / 32],
                                                unsigned char* smem, const long long m0, const int n0, const int tile_m,
                                                const int wm, const int wn, const int lane, const int tid,
                                                const STAMP& stampf = STAMP()) {
  constexpr int NT = 64 * WGM * WGN;
  constexpr int TM = BM / WGM / 32;
  constexpr int TN = BN / WGN / 32;
  const int HgWg = a.Hg * a.Wg;
  float* smf = reinterpret_cast<float*>(smem);
  long long* rowoff = reinterpret_cast<long long*>(smf);
  for (int rr = tid; rr < BM; rr += NT) {
    const long long m = m0 + rr;
    long long off = -1;
    if (m < a.M) {
      const int n = (int)(m / HgWg);
      const int rem = (int)(m - (long long)n * HgWg);
      const int y = rem / a.Wg;
      const int x = rem - y * a.Wg;
      off = (((long long)n * a.Hd + (y * a.osy + a.ooy)) * a.Wd + (x * a.osx + a.oox)) * a.Cd;
    }
    rowoff[rr] = off;
  }
  __syncthreads();
  stampf(0);

  // 1 / (scale of A * scale of B): both powers of two (planes headers)
  const float unscale =
      reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.src) + a.src_bytes - PL_HEADER)[2] *
      reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.wgt) + a.wgt_bytes - PL_HEADER)[2];
  float* sred = smf + 2 * BM;
  float csum[TN], csq[TN], cmx[TN];
  asm volatile("" ::"v"(unscale));
  stampf(1);
  // rows of this tile that exist (the last row tile of a tensor is partial): row r of the tile is real iff r < rows_valid
  const int rows_valid = (a.M - m0 < (long long)BM) ? (int)(a.M - m0) : BM;
  // 128-row-multiple tiles store through LDS: the C/D layout gives a lane one column and 16 scattered rows (64 dword
  // stores per lane, two 128-B row pieces per instruction); transposed in 64-row passes, every store instruction
  // writes 1 KB = two full 512-B rows of the tile as dwordx4 (the store tail is issue-bound, not bandwidth-bound).
  // The statistics are accumulated with selects, no LDS reads and no branches inside the 64-value loops (the first
  // version re-read the row offset from LDS and branched per value: 10 k cycles per pass, 21 k of a 128x128 tile's 80 k).
  constexpr bool VEC_TILE = ((BM % 128) == 0 && (BN % 32) == 0);
  const bool vec = VEC_TILE && a.vec_store && (a.Cout & 3) == 0 && (a.Cd & 3) == 0 && !(a.accumulate && a.stats != nullptr) &&
                   (DBG & 16) == 0;
  if (vec) {
    constexpr int TLD = BN + 4;                    // floats per staged row (16-B aligned, rows 4 banks apart)
    constexpr int C4 = BN / 4;                     // dwordx4 pieces per row
    float* tile = smf + 2 * BM + WGM * BN * 3;     // after rowoff and sred; 64 x (BN+4) x 4 B
    static_assert((2 * BM + WGM * BN * 3 + 64 * TLD) * 4 <= LDS_BYTES, "epilogue staging exceeds the LDS it may reuse");
    static_assert((64 * C4) % NT == 0, "copy-out: whole iterations");
    float bvj[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
      bvj[j] = (a.bias != nullptr && col < a.Cout) ? a.bias[col] : 0.f;
      csum[j] = csq[j] = cmx[j] = 0.f;
    }
#pragma unroll
    for (int h = 0; h < BM / 64; ++h) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int rbk = wm * TM + i;               // 32-row block of the tile: wave-uniform
        if ((rbk >> 1) != h) continue;
        const int rl0 = (rbk & 1) * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int cl = (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int rl = rl0 + (q & 3) + 8 * (q >> 2);
            const float v = fmaf(acc[i][j][q], unscale, bvj[j]);
            tile[rl * TLD + cl] = v;
            const float vm = (h * 64 + rl < rows_valid) ? v : 0.f;
            csum[j] += vm;
            csq[j] = fmaf(vm, vm, csq[j]);
            cmx[j] = fmaxf(cmx[j], fabsf(vm));
          }
        }
      }
      __syncthreads();
      stampf(2 + 2 * h);
      {
        constexpr int ITER = 64 * C4 / NT;
        long long offs[ITER];
        f32x4 vv[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
          const int idx = tid + it * NT;
          const int rl = idx / C4, c4 = idx - rl * C4;
          offs[it] = rowoff[h * 64 + rl];
          vv[it] = *reinterpret_cast<const f32x4*>(tile + rl * TLD + c4 * 4);
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
          const int idx = tid + it * NT;
          const int c4 = idx % C4;
          const int col = n0 + c4 * 4;
          if (offs[it] >= 0 && col < a.Cout) {
            f32x4* p = reinterpret_cast<f32x4*>(a.dst + offs[it] + col);
            f32x4 v = vv[it];
            if (a.accumulate) v += *p;
            if (a.nt_store) __builtin_nontemporal_store(v, p); else *p = v;
          }
        }
      }
      __syncthreads();
      stampf(3 + 2 * h);
    }
  } else {
#pragma unroll
    for (int j = 0; j < TN; ++j) csum[j] = csq[j] = cmx[j] = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      long long offs[16];   // the 16 rows of this lane in the 32-row block: one batch of LDS reads
#pragma unroll
      for (int q = 0; q < 16; ++q) offs[q] = rowoff[(wm * TM + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
        const bool cok = col < a.Cout;
        const float bv = (a.bias != nullptr && cok) ? a.bias[col] : 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const bool ok = cok && offs[q] >= 0;
          float v = fmaf(acc[i][j][q], unscale, bv);
          if (ok) {
            if (a.accumulate) v += a.dst[offs[q] + col];
            if constexpr ((DBG & 16) != 0) { if (v == 1234.5678f) a.dst[offs[q] + col] = v; }
            else if (a.nt_store) __builtin_nontemporal_store(v, &a.dst[offs[q] + col]); else a.dst[offs[q] + col] = v;
          }
          const float vm = ok ? v : 0.f;
          csum[j] += vm;
          csq[j] = fmaf(vm, vm, csq[j]);
          cmx[j] = fmaxf(cmx[j], fabsf(vm));
        }
      }
    }
  }
  if (a.stats != nullptr || a.absmax != nullptr) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float s1 = csum[j] + __shfl_xor(csum[j], 32, 64);
      const float s2 = csq[j] + __shfl_xor(csq[j], 32, 64);
      const float mx = fmaxf(cmx[j], __shfl_xor(cmx[j], 32, 64));
      if (lane < 32) {
        const int c = (wn * TN + j) * 32 + lane;
        sred[(wm * BN + c) * 3 + 0] = s1;
        sred[(wm * BN + c) * 3 + 1] = s2;
        sred[(wm * BN + c) * 3 + 2] = mx;
      }
    }
    __syncthreads();
    for (int c = tid; c < BN; c += NT) {
      const int col = n0 + c;
      if (col < a.Cout) {
        float s1 = 0.f, s2 = 0.f, mx = 0.f;
#pragma unroll
        for (int w = 0; w < WGM; ++w) {
          s1 += sred[(w * BN + c) * 3 + 0];
          s2 += sred[(w * BN + c) * 3 + 1];
          mx = fmaxf(mx, sred[(w * BN + c) * 3 + 2]);
        }
        if (a.stats != nullptr) {
          double* slot = a.stats + (long long)(tile_m & (YOLO_BN_STAT_SLOTS - 1)) * 2 * a.Cout;
          atomicAdd(&slot[col], (double)s1);
          atomicAdd(&slot[a.Cout + col], (double)s2);
        }
        // per-channel max|y| (bit patterns of non-negative floats order like integers); most tiles skip the atomic
        if (a.absmax != nullptr && __builtin_bit_cast(unsigned, mx) > a.absmax[col])
          atomicMax(&a.absmax[col], __builtin_bit_cast(unsigned, mx));
      }
    }
  }
}

}  // namespace yolo

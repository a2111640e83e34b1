// Epilogue shared by the planes convolution kernels (conv_planes.hip, conv_win.hip): undo the two power-of-two
// operand scales, add the bias, optionally accumulate into dst, store (128-row-multiple tiles: through LDS as
// dwordx4 rows), and reduce the BatchNorm statistics / per-channel max|y| of the tile.
// Must be entered after every wave is done with the LDS region it reuses (vmcnt(0) + barrier by the caller);
// LDS_BYTES = bytes of `smem` the epilogue may use.
#pragma once
#include "planes.hpp"
#include "act.hpp"

namespace yolo {

struct NoStamp {
  __device__ __forceinline__ void operator()(int) const {}
};

// output geometry of the tile: the launch's (GatherConvArgs) or, in a multi-class launch, the class's
struct EpiGeom {
  long long M;
  int Hg, Wg, ooy, oox;
  // PATCH tiles (conv_win.hip, 2-D patch geometry): tile row rr is output pixel (pn, py0 + rr / 16, px0 + rr % 16)
  int pn, py0, px0;
  // slot of GatherConvArgs::bwd_part this tile owns, plus one (0 = the tile's row-tile index): multi-class launches
  int slot1;
};
__device__ __forceinline__ EpiGeom epi_geom_of(const GatherConvArgs& a) {
  return EpiGeom{a.M, a.Hg, a.Wg, a.ooy, a.oox, 0, 0, 0, 0};
}

// split-K (GatherConvArgs::split_parts > 1): a part's accumulators go to its slab in accumulator order --
// 16-byte piece ((wave * TM + i) * TN + j) * 4 + q4 of lane l at byte (piece * 64 + l) * 16 -- for
// conv_split_reduce_kernel (conv_win.hip), which adds the parts of a tile in part order and runs the epilogue
template <int TM, int TN>
__device__ __forceinline__ void store_split_slab(const GatherConvArgs& a, f32x16 (&acc)[TM][TN], const int slab_bytes,
                                                 const int slab, const int wave, const int lane) {
  unsigned char* mine = reinterpret_cast<unsigned char*>(a.sk_slabs) + (size_t)slab * slab_bytes;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(mine, 0, slab_bytes, 0x00020000);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        // (bit_cast of the whole vector: hipcc 7.2 miscompiles __builtin_bit_cast applied to an ext-vector ELEMENT)
        const f32x4 f = {acc[i][j][4 * q4], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f), rs,
                                               (((wave * TM + i) * TN + j) * 4 + q4) * 1024 + lane * 16, 0, 0);
      }
}

// PATCH: the tile's rows are a 2-D patch of output pixels, BM / 16 rows of 16 (EpiGeom::pn / py0 / px0), not BM consecutive
// pixels: rows outside the image do not exist, and "which rows count for the statistics" is a per-lane bit mask
// instead of a prefix of the tile.
// BNRED: the instantiation that can make the fused BatchNorm-backward reduction (GatherConvArgs::bwd_y). Its own template
// parameter, i.e. its own kernels: the reduction holds a tile of y in registers beside the accumulators, and inside the
// forward kernels that cost registers (the 8-wave 1x1 kernel fell from 3 to 2 waves per SIMD) for a path they never take.
template <int BM, int BN, int WGM, int WGN, int LDS_BYTES, int DBG = 0, class STAMP = NoStamp, bool PATCH = false, bool BNRED = false>
__device__ __forceinline__ void planes_epilogue(const GatherConvArgs& a, f32x16 (&acc)[BM / WGM / 32][BN / WGN / 32],
                                                unsigned char* smem, const long long m0, const int n0, const int tile_m,
                                                const int wm, const int wn, const int lane, const int tid,
                                                const STAMP& stampf = STAMP(), const EpiGeom* geom = nullptr) {
  const EpiGeom G = geom != nullptr ? *geom : epi_geom_of(a);
  constexpr int NT = 64 * WGM * WGN;
  constexpr int NW = WGM * WGN;
  constexpr int TM = BM / WGM / 32;
  constexpr int TN = BN / WGN / 32;
  const int wave = wm * WGN + wn;
  float* smf = reinterpret_cast<float*>(smem);
  long long* rowoff = reinterpret_cast<long long*>(smf);
  // offset (in floats) of every tile row in dst (-1: the row does not exist). Dense outputs (the forward pass and
  // stride-1 data gradients: output pixel m IS row m of dst) need no divisions.
  const bool dense = a.osy == 1 && a.osx == 1 && G.ooy == 0 && G.oox == 0 && a.Hd == G.Hg && a.Wd == G.Wg;
  {
    const int HgWg = G.Hg * G.Wg;
    for (int rr = tid; rr < BM; rr += NT) {
      const long long m = m0 + rr;
      long long off = -1;
      if constexpr (PATCH) {
        const int y = G.py0 + (rr >> 4), x = G.px0 + (rr & 15);
        if (y < G.Hg && x < G.Wg) off = (((long long)G.pn * a.Hd + y) * a.Wd + x) * a.Cd;
      } else if (m < G.M) {
        if (dense) {
          off = m * a.Cd;
        } else {
          const int n = (int)(m / HgWg);
          const int rem = (int)(m - (long long)n * HgWg);
          const int y = rem / G.Wg;
          const int x = rem - y * G.Wg;
          off = (((long long)n * a.Hd + (y * a.osy + G.ooy)) * a.Wd + (x * a.osx + G.oox)) * a.Cd;
        }
      }
      rowoff[rr] = off;
    }
  }
  __syncthreads();
  stampf(0);

  // 1 / (scale of A * scale of B): both powers of two (planes headers)
  const float unscale =
      reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.src) + a.src_bytes - PL_HEADER)[2] *
      reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.wgt) + a.wgt_bytes - PL_HEADER)[2];
  float* sred = smf + 2 * BM;
  float csum[TN], csq[TN], cmx[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) csum[j] = csq[j] = cmx[j] = 0.f;
  [[maybe_unused]] float bs1[TN], bs2[TN];   // scalar path of the fused BatchNorm-backward reduction (bnred below)
#pragma unroll
  for (int j = 0; j < TN; ++j) bs1[j] = bs2[j] = 0.f;
  asm volatile("" ::"v"(unscale));
  stampf(1);
  // rows of this tile that exist (the last row tile of a tensor is partial): row r of the tile is real iff r < rows_valid
  const int rows_valid = PATCH ? BM : ((G.M - m0 < (long long)BM) ? (int)(G.M - m0) : BM);
  const bool want_stats = a.stats != nullptr || a.absmax != nullptr;   // (data gradients, inference: none)
  // PATCH: bit (i * 16 + q) = "row (wm * TM + i) * 32 + (q & 3) + 8 (q >> 2) + 4 (lane >> 5) exists" (this lane's 16 rows
  // of each of its 32-row blocks), read once from the row offsets
  [[maybe_unused]] unsigned long long rowmask = ~0ull;
  if constexpr (PATCH) {
    static_assert(BM / WGM / 32 <= 4, "row mask: at most 4 row blocks per wave");
    rowmask = 0;
    if (want_stats) {
#pragma unroll
      for (int i = 0; i < BM / WGM / 32; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (rowoff[(wm * (BM / WGM / 32) + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)] >= 0)
            rowmask |= 1ull << (i * 16 + q);
    }
  }
  // BatchNorm-backward reduction of the tensor this launch completes (GatherConvArgs::bwd_y): per-channel sums of
  // dz = v * act'(scale * y + shift) and dz * xhat over the tile's rows, from the values as they are stored
  const bool bnred = BNRED && a.bwd_y != nullptr && a.stats == nullptr;
  float bn_mx = 0.f;   // max |dz| seen by this lane
  // Vector path. The C/D layout of the 32x32 MFMA leaves a lane with ONE column and 16 scattered rows of its
  // sub-tile (64 dword stores per lane, 128-B row pieces); every wave instead transposes its own sub-tile through a
  // PRIVATE strip of LDS, RP rows at a time, and stores dwordx4 (whole 128..512-B row pieces per lane group): no
  // workgroup barrier, all waves at once, and the stores of one pass fly while the next pass is staged. The
  // statistics are accumulated with selects -- no LDS reads, no branches inside the value loops. (History: the first
  // version staged 64-row halves of the whole tile between workgroup barriers, two waves at a time, and re-read the
  // row offset from LDS with a branch per value: 21 k of a 128x128 tile's 80 k cycles.)
  constexpr int WCOLS = TN * 32;                   // columns of a wave's sub-tile
  constexpr int TLD = WCOLS + 4;                   // floats per staged row (16-B aligned, rows 4 banks apart)
  constexpr int FIXED = 2 * BM + WGM * BN * 3;     // floats: row offsets + statistics scratch
  constexpr int RP = ((FIXED + NW * 32 * TLD) * 4 <= LDS_BYTES) ? 32 : 16;   // rows per pass
  constexpr bool VEC_OK = (FIXED + NW * RP * TLD) * 4 <= LDS_BYTES && (RP * WCOLS / 4) % 64 == 0;
  const bool vec = VEC_OK && a.vec_store && (a.Cout & 3) == 0 && (a.Cd & 3) == 0 && !(a.accumulate && a.stats != nullptr) &&
                   (DBG & 16) == 0;
  if constexpr (VEC_OK) {
  if (vec) {
    float* strip = smf + FIXED + wave * (RP * TLD);
    constexpr int C4 = WCOLS / 4;                  // dwordx4 pieces per row of the sub-tile
    constexpr int ITER = RP * C4 / 64;
    // (64 % C4 == 0: a lane stores the SAME four columns in every iteration of every pass, so the reduction's per-channel
    // coefficients and sums live in registers)
    static_assert(64 % C4 == 0, "store loop: a lane keeps its four columns");
    const int bcol = n0 + wn * WCOLS + (lane % C4) * 4;
    f32x4 b_sc = {0.f, 0.f, 0.f, 0.f}, b_sh = b_sc, b_mu = b_sc, b_iv = b_sc, b_s1 = b_sc, b_s2 = b_sc;
    // ALL the y values (and, for small sub-tiles, the old values of the accumulate form) this lane will need are requested
    // HERE, before the first staging pass: one memory latency per tile instead of one per batch of row pieces. (Measured,
    // the window data gradient still pays 13-21 us per launch for its y tile, the bandwidth-bound 1x1
    // accumulate form 34 us -- about what the standalone reduction costs beside the filter-gradient stream: the option is off.)
    constexpr int NPASS = TM * (32 / RP);
    constexpr bool PRE_OLD = TM * TN <= 2;
    [[maybe_unused]] f32x4 ypre[BNRED ? NPASS : 1][BNRED ? ITER : 1];
    [[maybe_unused]] f32x4 opre[BNRED && PRE_OLD ? NPASS : 1][BNRED && PRE_OLD ? ITER : 1];
    if constexpr (BNRED)
    if (bnred) {
      const bool cok = bcol < a.Cout;
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
          const int rl = (lane + it * 64) / C4;
          const long long o = rowoff[(wm * TM) * 32 + ps * RP + rl];   // (pass ps = (i, hp): rows (wm TM + i) 32 + hp RP ..)
          const bool ok = cok && o >= 0;
          ypre[ps][it] = ok ? *reinterpret_cast<const f32x4*>(a.bwd_y + o + bcol) : f32x4{0.f, 0.f, 0.f, 0.f};
          if constexpr (PRE_OLD)
            opre[ps][it] = (ok && a.accumulate) ? *reinterpret_cast<const f32x4*>(a.dst + o + bcol) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      if (cok) {
        b_sc = *reinterpret_cast<const f32x4*>(a.bwd_scale + bcol);
        b_sh = *reinterpret_cast<const f32x4*>(a.bwd_shift + bcol);
        b_mu = *reinterpret_cast<const f32x4*>(a.bwd_mean + bcol);
        b_iv = *reinterpret_cast<const f32x4*>(a.bwd_invstd + bcol);
      }
    }
    float bvj[TN], escj[TN], eshj[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
      bvj[j] = (a.bias != nullptr && col < a.Cout) ? a.bias[col] : 0.f;
      escj[j] = (a.epi_scale != nullptr && col < a.Cout) ? a.epi_scale[col] : 1.f;
      eshj[j] = (a.epi_scale != nullptr && col < a.Cout) ? a.epi_shift[col] : 0.f;
    }
    const bool fused = a.epi_scale != nullptr;
    // The loop-invariant choices are made ONCE, outside the 64-value loops (hipcc leaves a scalar branch per value and
    // choice in them otherwise): the fused inference epilogue rewrites the accumulators in place first (the same
    // operations in the same order; the common path then sees v = fma(acc, 1, 0) = acc), and the statistics have their
    // own copy of the staging loop.
    float us2 = unscale;
    if (fused) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q)
            acc[i][j][q] = act_fwd(fmaf(escj[j], fmaf(acc[i][j][q], unscale, bvj[j]), eshj[j]), a.epi_act);
      us2 = 1.f;
#pragma unroll
      for (int j = 0; j < TN; ++j) bvj[j] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int hp = 0; hp < 32 / RP; ++hp) {       // RP = 16: registers 0..7 (rows 0..15), then 8..15 (rows 16..31)
        const int row0 = (wm * TM + i) * 32 + hp * RP;          // first tile row of this pass (wave-uniform)
        if (want_stats) {
#pragma unroll
          for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int q = hp * (RP / 2); q < (hp + 1) * (RP / 2); ++q) {
              const int rl = (q & 3) + 8 * ((q >> 2) - hp * (RP / 8)) + 4 * (lane >> 5);   // row inside the pass
              const float v = fmaf(acc[i][j][q], us2, bvj[j]);
              strip[rl * TLD + j * 32 + (lane & 31)] = v;
              const float vm = (PATCH ? ((rowmask >> (i * 16 + q)) & 1) != 0 : (row0 + rl < rows_valid)) ? v : 0.f;
              csum[j] += vm;
              csq[j] = fmaf(vm, vm, csq[j]);
              cmx[j] = fmaxf(cmx[j], fabsf(vm));
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int q = hp * (RP / 2); q < (hp + 1) * (RP / 2); ++q) {
              const int rl = (q & 3) + 8 * ((q >> 2) - hp * (RP / 8)) + 4 * (lane >> 5);
              strip[rl * TLD + j * 32 + (lane & 31)] = fmaf(acc[i][j][q], us2, bvj[j]);
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // my strip is written (wave-private: no barrier)
        bool bn_here = false;
        if constexpr (BNRED) bn_here = bnred;
        if (bn_here) {
          if constexpr (BNRED) {
          // (its own form of the store loop, four row pieces at a time: their y values and, in the accumulate form, the old
          // values of dst fly together; short batches keep the epilogue inside the main loop's register budget)
          constexpr int SUB = ITER < 4 ? ITER : 4;
          const int ps = i * (32 / RP) + hp;
#pragma unroll
          for (int h0 = 0; h0 < ITER; h0 += SUB) {
            long long o4[SUB];
            f32x4 v4[SUB], p4[SUB];
#pragma unroll
            for (int k = 0; k < SUB; ++k) {
              const int idx = lane + (h0 + k) * 64;
              const int rl = idx / C4, c4 = idx - rl * C4;
              o4[k] = rowoff[row0 + rl];
              v4[k] = *reinterpret_cast<const f32x4*>(strip + rl * TLD + c4 * 4);
              if constexpr (PRE_OLD) p4[k] = opre[ps][h0 + k];
              else
                p4[k] = (o4[k] >= 0 && bcol < a.Cout && a.accumulate) ? *reinterpret_cast<const f32x4*>(a.dst + o4[k] + bcol)
                                                                     : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int k = 0; k < SUB; ++k) {
              if (o4[k] >= 0 && bcol < a.Cout) {
                const f32x4 v = v4[k] + p4[k];
                *reinterpret_cast<f32x4*>(a.dst + o4[k] + bcol) = v;   // (plain store: bn_act_bwd_apply reads this tensor next)
                const f32x4 yv = ypre[ps][h0 + k];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float dz = v[e] * act_grad(fmaf(b_sc[e], yv[e], b_sh[e]), a.bwd_act);
                  b_s1[e] += dz;
                  b_s2[e] = fmaf(dz, (yv[e] - b_mu[e]) * b_iv[e], b_s2[e]);
                  bn_mx = fmaxf(bn_mx, fabsf(dz));
                }
              }
            }
          }
          }
        } else {
        long long offs[ITER];
        f32x4 vv[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
          const int idx = lane + it * 64;
          const int rl = idx / C4, c4 = idx - rl * C4;
          offs[it] = rowoff[row0 + rl];
          vv[it] = *reinterpret_cast<const f32x4*>(strip + rl * TLD + c4 * 4);
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
          const int idx = lane + it * 64;
          const int c4 = idx % C4;
          const int col = n0 + wn * WCOLS + c4 * 4;
          if (offs[it] >= 0 && col < a.Cout) {
            f32x4* p = reinterpret_cast<f32x4*>(a.dst + offs[it] + col);
            f32x4 v = vv[it];
            if (a.epi_res != nullptr) v += *reinterpret_cast<const f32x4*>(a.epi_res + offs[it] + col);
            if (a.accumulate) v += *p;
            if (a.nt_store) __builtin_nontemporal_store(v, p); else *p = v;
          }
        }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the strip is read before the next pass rewrites it
      }
    }
    if (bnred) {   // lanes l, l + C4, l + 2 C4, ... hold the same four columns: fixed-order butterfly, then one row of sred
#pragma unroll
      for (int o = C4; o < 64; o <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          b_s1[e] += __shfl_xor(b_s1[e], o, 64);
          b_s2[e] += __shfl_xor(b_s2[e], o, 64);
        }
      if (lane < C4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sred[(wm * BN + wn * WCOLS + lane * 4 + e) * 3 + 0] = b_s1[e];
          sred[(wm * BN + wn * WCOLS + lane * 4 + e) * 3 + 1] = b_s2[e];
        }
      }
    }
    stampf(2);
  }
  }
  if (!vec) {
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
        const float esc = (a.epi_scale != nullptr && cok) ? a.epi_scale[col] : 1.f;
        const float esh = (a.epi_scale != nullptr && cok) ? a.epi_shift[col] : 0.f;
        const float rsc = (bnred && cok) ? a.bwd_scale[col] : 0.f, rsh = (bnred && cok) ? a.bwd_shift[col] : 0.f;
        const float rmu = (bnred && cok) ? a.bwd_mean[col] : 0.f, riv = (bnred && cok) ? a.bwd_invstd[col] : 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const bool ok = cok && offs[q] >= 0;
          float v = fmaf(acc[i][j][q], unscale, bv);
          if (a.epi_scale != nullptr) v = act_fwd(fmaf(esc, v, esh), a.epi_act);
          const float vstat = v;           // statistics / max|.| before the residual (its bound is added by the caller)
          if (ok) {
            if (a.epi_res != nullptr) v += a.epi_res[offs[q] + col];
            if (a.accumulate) v += a.dst[offs[q] + col];
            if constexpr ((DBG & 16) != 0) { if (v == 1234.5678f) a.dst[offs[q] + col] = v; }
            else if (a.nt_store && !bnred) __builtin_nontemporal_store(v, &a.dst[offs[q] + col]); else a.dst[offs[q] + col] = v;
            if (bnred) {
              const float yv = a.bwd_y[offs[q] + col];
              const float dz = v * act_grad(fmaf(rsc, yv, rsh), a.bwd_act);
              bs1[j] += dz;
              bs2[j] = fmaf(dz, (yv - rmu) * riv, bs2[j]);
              bn_mx = fmaxf(bn_mx, fabsf(dz));
            }
          }
          const float vm = ok ? (a.epi_scale != nullptr ? vstat : v) : 0.f;
          csum[j] += vm;
          csq[j] = fmaf(vm, vm, csq[j]);
          cmx[j] = fmaxf(cmx[j], fabsf(vm));
        }
      }
    }
  }
  if (bnred) {
    if (!vec) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const float s1 = bs1[j] + __shfl_xor(bs1[j], 32, 64);
        const float s2 = bs2[j] + __shfl_xor(bs2[j], 32, 64);
        if (lane < 32) {
          const int c = (wn * TN + j) * 32 + lane;
          sred[(wm * BN + c) * 3 + 0] = s1;
          sred[(wm * BN + c) * 3 + 1] = s2;
        }
      }
    }
    __syncthreads();
    const long long slot = G.slot1 > 0 ? G.slot1 - 1 : tile_m;
    for (int c = tid; c < BN; c += NT) {
      const int col = n0 + c;
      if (col < a.Cout) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < WGM; ++w) {   // fixed order
          s1 += sred[(w * BN + c) * 3 + 0];
          s2 += sred[(w * BN + c) * 3 + 1];
        }
        a.bwd_part[(slot * 2 + 0) * a.Cout + col] = s1;
        a.bwd_part[(slot * 2 + 1) * a.Cout + col] = s2;
      }
    }
    if (a.bwd_aux != nullptr) {   // max |dz| (bit patterns of non-negative floats order like integers): 64 replica words
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) bn_mx = fmaxf(bn_mx, __shfl_xor(bn_mx, o, 64));
      unsigned* w = &a.bwd_aux[4 + ((int)(slot + n0 / BN) & 63)];
      if (lane == 0 && __builtin_bit_cast(unsigned, bn_mx) > *w) atomicMax(w, __builtin_bit_cast(unsigned, bn_mx));
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

// Shared epilogue of the LDS-ring GEMM kernels (gemm_dma.hip, gemm_ring64.hip): one 32-row slab of a wave's accumulators has been
// staged in LDS as fp32 ([wave][32][SLD], written by every wave, then __syncthreads()); this writes it out as 16-byte row segments.
//
// Modes (GemmArgs::epi, a compile-time parameter of the kernels; the fused ones exist for plain-row A operands only):
//   EPI_PLAIN      C = alpha*acc (+bias) (+rowbias) (+R) (+C), or a raw fp32 split-K slab
//   EPI_GEGLU_TAN  tangent of GEGLU fused into the FF-in product: the weight rows are interleaved in blocks of 64 (tape.py), so the
//                  wave pair (wx = 0, 1) of a 128-column tile holds da | dg of the SAME 64 hidden units; with the primal factors
//                  G1 = gelu(g), G2 = a*gelu'(g) (left in p.hprim by the primal GEGLU kernel, elementwise.hip) the pair writes
//                  dy = da*G1 + dg*G2 [M][F] -- dh [M][2F] never exists in HBM, and no transcendental runs in the epilogue.
//   EPI_GEGLU_ADJ  adjoint of GEGLU fused into the adjoint of the FF-out product: the tile holds gy for 128 hidden units; writes
//                  ga = gy*G1 and gg = gy*G2 at their interleaved positions of gh [M][2F].
#pragma once
#include "kernels.h"

namespace dpb {

template <int FL, int WN, int SLD, int EPI>
__device__ __forceinline__ void epilogue_slab(const GemmArgs& p, bf16* C, const bf16* R, const float* smem_f, int wave, int lane, int mrow0, int n0,
                                     long slab_idx) {
  constexpr int CPR = WN / 8;
  const int wx = wave & 1;
  const float* stage = smem_f + wave * 32 * SLD;
  if constexpr (EPI == EPI_GEGLU_TAN) {
    if constexpr (WN == 64) {
      const float* sa = smem_f + (wave & ~1) * 32 * SLD;      // a-half staged by wave wx = 0, g-half by its sibling wx = 1
      const float* sg = sa + 32 * SLD;
      const int F2 = p.N;                                      // 2F interleaved columns
#pragma unroll
      for (int it = 0; it < 2; ++it) {                         // the pair shares the 32 rows: 16 each
        const int item = it * 64 + lane;
        const int row = wx * 16 + item / CPR, c8 = item % CPR;
        const int m = mrow0 + row;
        const int n = n0 + c8 * 8;                             // interleaved column of the a-values (tile-aligned: n0 % 128 == 0)
        if (m >= p.M || n >= p.N) continue;
        float da[8], dg[8], ap[8], gp[8], o[8];
        Vec<float>::load(sa + row * SLD + c8 * 8, da); Vec<float>::load(sa + row * SLD + c8 * 8 + 4, da + 4);
        Vec<float>::load(sg + row * SLD + c8 * 8, dg); Vec<float>::load(sg + row * SLD + c8 * 8 + 4, dg + 4);
        const int smp = m / p.rows_per_sample, l = m - smp * p.rows_per_sample;
        const bf16* hp = (const bf16*)p.hprim + ((long)(smp / p.epi_kps) * p.rows_per_sample + l) * F2 + n;
        H16<FL>::load8(hp, ap);
        H16<FL>::load8(hp + 64, gp);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = p.alpha * (da[e] * ap[e] + dg[e] * gp[e]);      // ap = G1 = gelu(g), gp = G2 = a gelu'(g)
        H16<FL>::store8(C + (long)m * p.ldc + (n0 >> 1) + c8 * 8, o);
      }
    } else if constexpr (WN == 128) {                          // 256x256 tile: the wave's own 128 columns are a | g of 64 hidden units
      const int F2 = p.N;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int item = it * 64 + lane;
        const int row = item >> 3, c8 = item & 7;
        const int m = mrow0 + row;
        const int n = n0 + wx * WN + c8 * 8;                   // interleaved column of the a-values (n0 % 256 == 0)
        if (m >= p.M || n >= p.N) continue;
        float da[8], dg[8], ap[8], gp[8], o[8];
        Vec<float>::load(stage + row * SLD + c8 * 8, da); Vec<float>::load(stage + row * SLD + c8 * 8 + 4, da + 4);
        Vec<float>::load(stage + row * SLD + 64 + c8 * 8, dg); Vec<float>::load(stage + row * SLD + 64 + c8 * 8 + 4, dg + 4);
        const int smp = m / p.rows_per_sample, l = m - smp * p.rows_per_sample;
        const bf16* hp = (const bf16*)p.hprim + ((long)(smp / p.epi_kps) * p.rows_per_sample + l) * F2 + n;
        H16<FL>::load8(hp, ap);
        H16<FL>::load8(hp + 64, gp);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = p.alpha * (da[e] * ap[e] + dg[e] * gp[e]);
        H16<FL>::store8(C + (long)m * p.ldc + ((n0 + wx * WN) >> 1) + c8 * 8, o);
      }
    }
    return;
  }
  constexpr int ITEMS = 32 * CPR / 64;
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int item = it * 64 + lane;
    const int row = item / CPR, c8 = item % CPR;
    const int m = mrow0 + row;
    const int n = n0 + wx * WN + c8 * 8;
    if (m >= p.M || n >= p.N) continue;
    float v[8];
    Vec<float>::load(stage + row * SLD + c8 * 8, v);
    Vec<float>::load(stage + row * SLD + c8 * 8 + 4, v + 4);
    if (EPI == EPI_PLAIN && p.splitk > 1) {       // split-K partial: raw fp32 slab, reduced by splitk_reduce_kernel
      float* sp = p.slab + slab_idx * (long)p.M * p.N + (long)m * p.N + n;
      if (n + 8 <= p.N && !(p.N & 3)) {
        Vec<float>::store(sp, v);
        Vec<float>::store(sp + 4, v + 4);
      } else {
        for (int e = 0; e < 8 && n + e < p.N; ++e) sp[e] = v[e];
      }
      continue;
    }
    if constexpr (EPI == EPI_GEGLU_ADJ) {                 // v = gy of hidden units n..n+7 (n % 8 == 0, N = F % 64 == 0)
      const int F2 = 2 * p.N;
      const int ni = ((n >> 6) << 7) + (n & 63);   // interleaved column of a; g sits 64 further
      const int smp = m / p.rows_per_sample, l = m - smp * p.rows_per_sample;
      const bf16* hp = (const bf16*)p.hprim + ((long)(smp / p.epi_kps) * p.rows_per_sample + l) * F2 + ni;
      float ap[8], gp[8], oa[8], og[8];
      H16<FL>::load8(hp, ap);
      H16<FL>::load8(hp + 64, gp);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float gy = p.alpha * v[e];
        oa[e] = gy * ap[e];                        // ap = G1 = gelu(g), gp = G2 = a gelu'(g)
        og[e] = gy * gp[e];
      }
      bf16* cp = C + (long)m * p.ldc + ni;
      if (p.accumulate) {
        float t1[8], t2[8];
        H16<FL>::load8(cp, t1);
        H16<FL>::load8(cp + 64, t2);
#pragma unroll
        for (int e = 0; e < 8; ++e) { oa[e] += t1[e]; og[e] += t2[e]; }
      }
      H16<FL>::store8(cp, oa);
      H16<FL>::store8(cp + 64, og);
      continue;
    }
    int smp = 0;
    if (p.rowbias) smp = (m / p.rows_per_sample) / p.rowbias_div;
    bf16* cp = C + (long)m * p.ldc + n;
    if (p.vec_ok && n + 8 <= p.N) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
      float b8[8];
      if (p.bias) {
        Vec<float>::load(p.bias + n, b8);
        Vec<float>::load(p.bias + n + 4, b8 + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += b8[e];
      }
      if (p.rowbias) {
        H16<FL>::load8((const bf16*)p.rowbias + (long)smp * p.N + n, b8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += b8[e];
      }
      if (R) {
        H16<FL>::load8(R + (long)m * p.ldr + n, b8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += b8[e];
      }
      if (p.accumulate) {
        H16<FL>::load8(cp, b8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += b8[e];
      }
      H16<FL>::store8(cp, v);
    } else {
      for (int e = 0; e < 8 && n + e < p.N; ++e) {
        float x = p.alpha * v[e];
        if (p.bias) x += p.bias[n + e];
        if (p.rowbias) x += ld16<FL>((const bf16*)p.rowbias + (long)smp * p.N + n + e);
        if (R) x += ld16<FL>(R + (long)m * p.ldr + n + e);
        if (p.accumulate) x += ld16<FL>(cp + e);
        st16<FL>(cp + e, x);
      }
    }
  }
}

}  // namespace dpb

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
//   EPI_GEGLU_FWD  forward pass only (dpb_forward): y = a*gelu(g) from the FF-in product, bitwise the unfused product + GEGLU kernel
//   EPI_LN_TAN     (row-complete tile: BN = N, epilogue_ln below) h = acc (+R) -> C, and the LayerNorm tangent of h at the primal row -> C2
//   EPI_LN_ADJ     (row-complete tile) the LayerNorm adjoint of acc (= cotangent of the LayerNorm output) (+)-> C
#pragma once
#include "kernels.h"

#ifndef DPB_SLAB_FULL_LINES
#define DPB_SLAB_FULL_LINES 1   // split-K slabs stored one 16-byte chunk per lane, whole lines per instruction (0: the 8-floats-per-lane form of rounds 1-5; A/B builds)
#endif
#ifndef DPB_SLAB_STORE
#define DPB_SLAB_STORE 0        // store flavour of the fp32 slabs (common.h store_out16): 0 plain, 1 sc1, 2 nt (A/B builds)
#endif

namespace dpb {

// Row-complete tiles (the block holds whole rows of the product: BN = N, two waves side by side).  64 rows of fp32 accumulators are staged as
// [wave][32][SLD] (waves 2 wy, 2 wy + 1 hold the two halves of the same 32 rows); each wave then owns 16 of the 64 rows, EIGHT rows at a time:
// 8 lanes per row, lane l of a row holds the 16-byte chunks l, l + 8, ... (N / 64 of them), so the row reductions are 3-step shuffles inside
// 8-lane groups (one row per wave with 6-step ds_bpermute reductions, 4 dependent ones per row, cost 35 us per launch: latency-bound with one
// wave per SIMD).  LayerNorm algebra as in norm.hip (ln_rows_kernel):
//   tangent:  z = gamma o rstd (v - mean(v) - xhat mean(xhat v)),            v = h = acc + R (rounded to 16 bit like the stored h)
//   adjoint:  g = rstd (w - mean(w) - xhat mean(xhat w)),  w = gamma o gz,   gz = acc (rounded to 16 bit like a stored cotangent)
__device__ inline float seg8_sum(float v) {
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 1, 64);
  return v;
}
template <int WN> struct LnPre { uint4 xr[2][2 * WN / 64], rres[2][2 * WN / 64]; };   // primal rows and residual / accumulated rows of one 64-row step
// issued at kernel start (the addresses do not depend on the product): the loads land under the K loop instead of in front of the epilogue
template <int FL, int WN, int EPI>
__device__ __forceinline__ void ln_prefetch(const GemmArgs& p, const bf16* C, const bf16* R, int wave, int lane, int mrow0_block, LnPre<WN>& q) {
  constexpr int NI = 2 * WN / 64;
  const int wy = wave >> 1, rg = lane >> 3, l = lane & 7;
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int m = min(mrow0_block + wy * 32 + (wave & 1) * 16 + ps * 8 + rg, p.M - 1);
    const int smp = m / p.rows_per_sample, lr = m - smp * p.rows_per_sample;
    const long prow = (long)(smp / p.epi_kps) * p.rows_per_sample + lr;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int n = (l + 8 * i) * 8;
      q.xr[ps][i] = *reinterpret_cast<const uint4*>((const bf16*)p.ln_x + prow * p.N + n);
      q.rres[ps][i] = make_uint4(0, 0, 0, 0);
      if (EPI == EPI_LN_TAN && R) q.rres[ps][i] = *reinterpret_cast<const uint4*>(R + (long)m * p.ldr + n);
      if (EPI == EPI_LN_ADJ && p.accumulate) q.rres[ps][i] = *reinterpret_cast<const uint4*>(C + (long)m * p.ldc + n);   // the cotangent accumulated so far
    }
  }
}

template <int FL, int WN, int SLD, int EPI>
__device__ __forceinline__ void epilogue_ln(const GemmArgs& p, bf16* C, const bf16* R, const float* smem_f, int wave, int lane, int mrow0_block, const LnPre<WN>& q) {
  constexpr int NI = 2 * WN / 64;                                // chunks per lane: N / 8 chunks over 8 lanes
  static_assert(2 * WN % 64 == 0, "N = 2 WN must be a multiple of 64");
  const OutBuf ob2 = out_buf(EPI == EPI_LN_TAN ? p.C2 : (void*)C, (long)p.M * p.ldc * 2);
  const int wy = wave >> 1, rg = lane >> 3, l = lane & 7;
  const float inv_c = 1.f / (float)p.N;
  float gam[NI][8];
#pragma unroll
  for (int i = 0; i < NI; ++i) { Vec<float>::load(p.ln_gamma + (l + 8 * i) * 8, gam[i]); Vec<float>::load(p.ln_gamma + (l + 8 * i) * 8 + 4, gam[i] + 4); }
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int row = (wave & 1) * 16 + ps * 8 + rg;               // of the 32 rows staged by the wave pair (2 wy, 2 wy + 1)
    const int m = mrow0_block + wy * 32 + row;
    const bool live = m < p.M;                                   // whole 8-lane groups agree: the shuffles stay inside the group
    float v[NI][8], x[NI][8];
    float sx = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int n = (l + 8 * i) * 8, half = n / WN, cn = n - half * WN;
      const float* sp = smem_f + ((wy * 2 + half) * 32 + row) * SLD + cn;
      Vec<float>::load(sp, v[i]); Vec<float>::load(sp + 4, v[i] + 4);
      H16<FL>::load8(reinterpret_cast<const bf16*>(&q.xr[ps][i]), x[i]);
      if (EPI == EPI_LN_TAN && R) {
        float r8[8];
        H16<FL>::load8(reinterpret_cast<const bf16*>(&q.rres[ps][i]), r8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] += r8[e];
      }
      const bf16x8 pk = H16<FL>::pack8(v[i]);                    // the 16-bit value a separate product would have stored
      if (EPI == EPI_LN_TAN && live) *reinterpret_cast<bf16x8*>(C + (long)m * p.ldc + n) = pk;
      H16<FL>::load8(reinterpret_cast<const bf16*>(&pk), v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) sx += x[i][e];
    }
    const float mean = seg8_sum(sx) * inv_c;
    float vs = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) { x[i][e] -= mean; vs += x[i][e] * x[i][e]; }
    const float rstd = rsqrtf(seg8_sum(vs) * inv_c + p.ln_eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (EPI == EPI_LN_ADJ) v[i][e] *= gam[i][e];
        x[i][e] *= rstd;                                         // xhat
        s1 += v[i][e];
        s2 += x[i][e] * v[i][e];
      }
    const float m1 = seg8_sum(s1) * inv_c, m2 = seg8_sum(s2) * inv_c;
    if (!live) continue;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int n = (l + 8 * i) * 8;
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float w = rstd * (v[i][e] - m1 - x[i][e] * m2);
        o[e] = EPI == EPI_LN_TAN ? w * gam[i][e] : w;
      }
      if (EPI == EPI_LN_ADJ && p.accumulate) {
        float old[8];
        H16<FL>::load8(reinterpret_cast<const bf16*>(&q.rres[ps][i]), old);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += old[e];
      }
      store8_at<FL>(ob2, (EPI == EPI_LN_TAN ? (bf16*)p.C2 : C) + (long)m * p.ldc + n, o);
    }
  }
}

// (Round 4 built GroupNorm statistics of the output into this epilogue -- per-lane column sums, a fixed xor tree, one float4 per wave strip and
// column -- so that the consumer GroupNorm needed no statistics launch: 335 -> 321 launches, 0.4-0.7 % SLOWER; the tangent / adjoint statistics
// need the primal tensor next to the output, an extra read that lands in the one HBM-bound phase of the product.  profiles/r04_gn_epi_stats.txt.)
template <int FL, int WN, int SLD, int EPI>
__device__ __forceinline__ void epilogue_slab(const GemmArgs& p, bf16* C, const bf16* R, const float* smem_f, int wave, int lane, int mrow0, int n0,
                                     long slab_idx) {
  constexpr int CPR = WN / 8;
  const int wx = wave & 1;
  const float* stage = smem_f + wave * 32 * SLD;
  const OutBuf cb = out_buf(C, (long)p.M * p.ldc * 2);           // write-through stores through the output's buffer descriptor (common.h)
  if constexpr (EPI == EPI_GEGLU_FWD) {
    // forward pass (dpb_forward): y = a * gelu(g) straight from the FF-in product -- h [M][2F] is never written.  a and g are rounded to 16 bit first
    // and the expression is the primal GEGLU kernel's (elementwise.hip), so the result is bitwise what product + geglu_kernel give.
    const int F2 = p.N;
    constexpr int NIT = WN == 64 ? 2 : 4;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int item = it * 64 + lane;
      int row, c8, n, ncol;
      const float *pa, *pg;
      if constexpr (WN == 64) {                              // the wave pair (wx = 0, 1) holds a | g of the same 64 hidden units: 16 of the 32 rows each
        const float* sa = smem_f + (wave & ~1) * 32 * SLD;
        row = wx * 16 + item / CPR; c8 = item % CPR;
        n = n0 + c8 * 8; ncol = (n0 >> 1) + c8 * 8;
        pa = sa + row * SLD + c8 * 8; pg = sa + 32 * SLD + row * SLD + c8 * 8;
      } else {                                               // 256 x 256 tile: the wave's own 128 columns are a | g of 64 hidden units
        row = item >> 3; c8 = item & 7;
        n = n0 + wx * WN + c8 * 8; ncol = ((n0 + wx * WN) >> 1) + c8 * 8;
        pa = stage + row * SLD + c8 * 8; pg = pa + 64;
      }
      const int m = mrow0 + row;
      if (m >= p.M || n >= p.N) continue;
      float a8[8], g8[8], o[8];
      Vec<float>::load(pa, a8); Vec<float>::load(pa + 4, a8 + 4);
      Vec<float>::load(pg, g8); Vec<float>::load(pg + 4, g8 + 4);
      if (p.bias) {
        float b8[8];
        Vec<float>::load(p.bias + n, b8); Vec<float>::load(p.bias + n + 4, b8 + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) a8[e] = p.alpha * a8[e] + b8[e];
        Vec<float>::load(p.bias + n + 64, b8); Vec<float>::load(p.bias + n + 68, b8 + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) g8[e] = p.alpha * g8[e] + b8[e];
      }
      const bf16x8 ar = H16<FL>::pack8(a8), gr = H16<FL>::pack8(g8);     // the 16-bit h the unfused path stores and reloads
      H16<FL>::load8(reinterpret_cast<const bf16*>(&ar), a8);
      H16<FL>::load8(reinterpret_cast<const bf16*>(&gr), g8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float er = erff(g8[e] * 0.70710678118654752f);
        const float g1 = 0.5f * g8[e] * (1.f + er);
        o[e] = a8[e] * g1;
      }
      store8_at<FL>(cb, C + (long)m * p.ldc + ncol, o);
    }
    (void)F2;
    return;
  }
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
        store8_at<FL>(cb, C + (long)m * p.ldc + (n0 >> 1) + c8 * 8, o);
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
        store8_at<FL>(cb, C + (long)m * p.ldc + ((n0 + wx * WN) >> 1) + c8 * 8, o);
      }
    }
    return;
  }
  // item it of a lane = row it * RPI + lane / CPR of the staged slab, 16-byte column lane % CPR: the column is the same for every item of a lane
  constexpr int ITEMS = 32 * CPR / 64, RPI = 64 / CPR;
  static_assert(64 % CPR == 0, "a lane keeps its column across items");
  const int c8 = lane % CPR, r0 = lane / CPR;
  const int n = n0 + wx * WN + c8 * 8;
  if (DPB_SLAB_FULL_LINES && EPI == EPI_PLAIN && p.splitk > 1 && !(p.N & 3)) {   // split-K partial, N a multiple of 4 (else the 8-floats-per-lane form below)
    // One 16-byte chunk per lane and instruction, consecutive lanes on consecutive chunks of a row: every store instruction writes whole 128-byte lines
    // (the 8-floats-per-lane form wrote the two halves of each 32 bytes in two instructions: half lines, which a write-through store turns into a
    // read-modify-write at the memory side).  Same values at the same addresses.
    constexpr int CPR4 = WN / 4;
    if constexpr (64 % CPR4 == 0) {
      constexpr int RPI4 = 64 / CPR4;
      const int c4 = lane % CPR4, r4 = lane / CPR4, n4 = n0 + wx * WN + c4 * 4;
      {
#pragma unroll
        for (int it = 0; it < 32 / RPI4; ++it) {
          const int row = it * RPI4 + r4, m = mrow0 + row;
          if (m >= p.M || n4 >= p.N) continue;
          float v[4];
          Vec<float>::load(stage + row * SLD + c4 * 4, v);
          const u32x4_ bits = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
          store_out16<DPB_SLAB_STORE>(p.slab + slab_idx * (long)p.M * p.N + (long)m * p.N + n4, bits);
        }
        return;
      }
    }
  }
  if (n >= p.N) return;
  if (EPI == EPI_PLAIN && p.splitk > 1) {         // split-K partial: raw fp32 slab, reduced by splitk_reduce_kernel (or by the consumer: SlabSrc)
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int row = it * RPI + r0, m = mrow0 + row;
      if (m >= p.M) continue;
      float v[8];
      Vec<float>::load(stage + row * SLD + c8 * 8, v);
      Vec<float>::load(stage + row * SLD + c8 * 8 + 4, v + 4);
      float* sp = p.slab + slab_idx * (long)p.M * p.N + (long)m * p.N + n;
      if (n + 8 <= p.N && !(p.N & 3)) {
        Vec<float>::store(sp, v);
        Vec<float>::store(sp + 4, v + 4);
      } else {
        for (int e = 0; e < 8 && n + e < p.N; ++e) sp[e] = v[e];
      }
    }
    return;
  }
  if constexpr (EPI == EPI_GEGLU_ADJ) {           // v = gy of hidden units n..n+7 (n % 8 == 0, N = F % 64 == 0)
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int row = it * RPI + r0, m = mrow0 + row;
      if (m >= p.M) continue;
      float v[8];
      Vec<float>::load(stage + row * SLD + c8 * 8, v);
      Vec<float>::load(stage + row * SLD + c8 * 8 + 4, v + 4);
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
      store8_at<FL>(cb, cp, oa);
      store8_at<FL>(cb, cp + 64, og);
    }
    return;
  }
  if (p.vec_ok && n + 8 <= p.N) {
    // The operands of the epilogue (row bias, residual, the value accumulated so far) do not depend on the product: the loads of up to four items
    // are issued together, at clamped rows, before anything waits -- one memory round trip per batch instead of one per operand and item (the
    // item-by-item form cost a 64x64-level product with a residual ~5 us of dependent L2 round trips).  Same additions in the same order.
    float b8[8];
    if (p.bias) {
      Vec<float>::load(p.bias + n, b8);
      Vec<float>::load(p.bias + n + 4, b8 + 4);
    }
    constexpr int BT = ITEMS > 4 ? 4 : ITEMS;
#pragma unroll
    for (int it0 = 0; it0 < ITEMS; it0 += BT) {
      uint4 rr[BT], ro[BT], rb[BT];
#pragma unroll
      for (int u = 0; u < BT; ++u) {
        const int mc = min(mrow0 + (it0 + u) * RPI + r0, p.M - 1);
        if (R) rr[u] = *reinterpret_cast<const uint4*>(R + (long)mc * p.ldr + n);
        if (p.accumulate) ro[u] = *reinterpret_cast<const uint4*>(C + (long)mc * p.ldc + n);
        if (p.rowbias) rb[u] = *reinterpret_cast<const uint4*>((const bf16*)p.rowbias + (long)((mc / p.rows_per_sample) / p.rowbias_div) * p.N + n);
      }
#pragma unroll
      for (int u = 0; u < BT; ++u) {
        const int row = (it0 + u) * RPI + r0, m = mrow0 + row;
        if (m >= p.M) continue;
        float v[8], t8[8];
        Vec<float>::load(stage + row * SLD + c8 * 8, v);
        Vec<float>::load(stage + row * SLD + c8 * 8 + 4, v + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
        if (p.bias) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += b8[e];
        }
        if (p.rowbias) {
          H16<FL>::load8(reinterpret_cast<const bf16*>(&rb[u]), t8);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += t8[e];
        }
        if (R) {
          H16<FL>::load8(reinterpret_cast<const bf16*>(&rr[u]), t8);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += t8[e];
        }
        if (p.accumulate) {
          H16<FL>::load8(reinterpret_cast<const bf16*>(&ro[u]), t8);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += t8[e];
        }
        store8_at<FL>(cb, C + (long)m * p.ldc + n, v);
      }
    }
  } else {
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int row = it * RPI + r0, m = mrow0 + row;
      if (m >= p.M) continue;
      float v[8];
      Vec<float>::load(stage + row * SLD + c8 * 8, v);
      Vec<float>::load(stage + row * SLD + c8 * 8 + 4, v + 4);
      const int smp = p.rowbias ? (m / p.rows_per_sample) / p.rowbias_div : 0;
      bf16* cp = C + (long)m * p.ldc + n;
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

// GroupNorm(+SiLU) and LayerNorm for NHWC activations: primal, tangent (JVP) and adjoint (VJP wrt input).
//
// HBM-bound kernels.  Every thread owns a fixed 16-byte channel chunk (coalesced rows, per-channel
// constants loaded once) and walks pixels.  GroupNorm: pass 1 accumulates per-(sample,group) sums, pass 2 applies.
// Statistics of the two-pass kernels are bitwise reproducible by default (GNArgs::det = 1): the statistics launch reduces a block's
// per-channel partials through LDS in a fixed order and stores ONE partial per (block, group) with plain stores; the APPLY launch (a kernel
// boundary later, so the partials are visible without fences, tickets or atomics) has every block add the partials of its sample / tangent
// in block order (fp64, 4 fixed segments + a fixed 4-term tail) before it starts -- 32 L2-resident loads per thread.  No run-time-ordered
// floating-point addition anywhere.  det = 0 (dpb_debug_set("gn_deterministic", 0), A/B only) is the round-1 path: LDS float atomics + one
// fp64 global atomic per group per block, whose order varies from run to run (sigma_1 = 459.527 / 459.501 / 459.533 across identical runs).
// Feature maps whose per-sample group window fits a block's registers use the ONE-launch kernel below (deterministic by construction).
// The tangent and adjoint share one algebraic form:
//     out = rstd * (v - mean(v) - xhat * mean(xhat * v))
// with v = dx (tangent; gamma and SiLU' applied after) or v = gamma * SiLU'(y) * gz (adjoint; before).
// Primal statistics are computed once per x_t and reused by all k tangents / cotangents.
#include "kernels.h"

namespace dpb {

static int g_gn_det = getenv("DPB_GN_DETERMINISTIC") ? atoi(getenv("DPB_GN_DETERMINISTIC")) : 1;
void gn_debug_deterministic(int on) { g_gn_det = on; }
int gn_deterministic() { return g_gn_det; }


// d[row][c0 .. c0 + CH) of a deferred split-K reduction (SlabSrc): slabs added in slab order, + residual, rounded to T like the reduce kernel's
// store (the caller continues with exactly the values the separate reduce + load would have given); optional store for other readers.
template <typename T>
__device__ inline uint4 slab_chunk(const SlabSrc& s, long row, int c0) {
  constexpr int CH = TT<T>::CH;
  float v[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) v[e] = 0.f;
  const float* sp = s.slab + row * s.N + c0;
  auto ld = [&](int k, float* t) {
    Vec<float>::load(sp + (long)k * s.MN, t);
    if constexpr (CH == 8) Vec<float>::load(sp + (long)k * s.MN + 4, t + 4);
  };
  int k = 0;
  for (; k + 8 <= s.splitk; k += 8) {               // eight, then four slabs in flight; always added in slab order
    float t[8][CH];
#pragma unroll
    for (int u = 0; u < 8; ++u) ld(k + u, t[u]);
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int e = 0; e < CH; ++e) v[e] += t[u][e];
  }
  if (k + 4 <= s.splitk) {
    float t[4][CH];
#pragma unroll
    for (int u = 0; u < 4; ++u) ld(k + u, t[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < CH; ++e) v[e] += t[u][e];
    k += 4;
  }
  for (; k < s.splitk; ++k) {
    float t[CH];
    ld(k, t);
#pragma unroll
    for (int e = 0; e < CH; ++e) v[e] += t[e];
  }
  if (s.R) {
    float r[CH];
    Vec<T>::load((const T*)s.R + row * s.ldr + c0, r);
#pragma unroll
    for (int e = 0; e < CH; ++e) v[e] += r[e];
  }
  const uint4 packed = Raw<T>::pack(v);
  if (s.store) *reinterpret_cast<uint4*>((T*)s.store + row * s.N + c0) = packed;
  return packed;
}

template <typename T, int MODE, bool STATS>
__global__ __launch_bounds__(256) void gn_kernel(GNArgs a, int ppb) {
  constexpr int CH = TT<T>::CH;
  const OutBuf yb = out_buf(a.y, (long)max(a.NT, a.Bp) * a.HW * a.C * (long)sizeof(T));   // write-through output stores (common.h)
  extern __shared__ float lch[];    // STATS, deterministic path: per-channel partial sums [rpi][C][2] (dynamic: rpi * C * 8 bytes)
  __shared__ float lsum[2 * 256];   // STATS, atomic path: [G][2]; apply pass, deterministic path: the reduced statistics [G][2]
  __shared__ double lseg[4][2 * 256];   // apply pass, deterministic path: 4 block-range segment sums per statistic
  const int tid = threadIdx.x;
  const int j = blockIdx.y;                       // sample (primal) or tangent index
  const int b = (MODE == MODE_PRIMAL) ? j : j / a.kps;
  const int cols = a.C / CH;
  const int cw = cols < 256 ? cols : 256;
  const int rpi = 256 / cw;
  const int r = tid / cw, c0 = tid % cw;
  const int ncp = (cols + cw - 1) / cw;
  const int cpg = a.C / a.G;
  const int p0 = blockIdx.x * ppb;
  const int p1 = min(p0 + ppb, a.HW);
  const double inv_n = 1.0 / ((double)a.HW * cpg);
  if (STATS && !a.det) {
    for (int i = tid; i < 2 * a.G; i += 256) lsum[i] = 0.f;
    __syncthreads();
  }
  if (!STATS && a.red) {
    // fixed-order reduction of the statistics launch's per-block partials of sample / tangent j: segment q adds blocks [q nb/4, (q+1) nb/4) in
    // block order, then the 4 segment sums are added in order -- the same bits in every block of every run
    const int nblk = gridDim.x, n2 = 2 * a.G;
    const float* pj = a.part + (long)j * nblk * n2;
    for (int t = tid; t < 4 * n2; t += 256) {
      const int q = t / n2, i = t - q * n2;
      const int b0 = (int)((long)nblk * q / 4), b1 = (int)((long)nblk * (q + 1) / 4);
      double acc = 0.0;
      for (int bb = b0; bb < b1; bb += 32) {            // 32 independent L2-resident loads in flight, then added in block order
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = pj[(long)min(bb + u, b1 - 1) * n2 + i];      // (clamped, not predicated: predicated loads compile to
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = bb + u < b1 ? v[u] : 0.f;                     //  a branch and a full s_waitcnt per load)
#pragma unroll
        for (int u = 0; u < 32; ++u) acc += (double)v[u];
      }
      lseg[q][i] = acc;
    }
    __syncthreads();
    for (int i = tid; i < n2; i += 256) {
      const double acc = ((lseg[0][i] + lseg[1][i]) + lseg[2][i]) + lseg[3][i];
      if (MODE == MODE_PRIMAL) lseg[0][i] = acc;            // raw (sum, sum of squares), finalised below
      else lsum[i] = (float)(acc * inv_n);
    }
    __syncthreads();
    if (MODE == MODE_PRIMAL) {
      for (int g = tid; g < a.G; g += 256) {
        const double m = lseg[0][2 * g] * inv_n;
        double v = lseg[0][2 * g + 1] * inv_n - m * m;
        if (v < 0) v = 0;
        const double rs = 1.0 / sqrt(v + (double)a.eps);
        lsum[2 * g] = (float)m; lsum[2 * g + 1] = (float)rs;
        if (blockIdx.x == 0) { a.pstats[((long)j * a.G + g) * 2] = m; a.pstats[((long)j * a.G + g) * 2 + 1] = rs; }   // kept for the tangent / adjoint passes
      }
      __syncthreads();
    }
  }
  if (r < rpi) {
    for (int q = 0; q < ncp; ++q) {
      const int col = c0 + q * cw;
      if (col >= cols) break;
      const int ch0 = col * CH;
      float mean[CH], rstd[CH], gam[CH], bet[CH], m1[CH], m2[CH];
      int grp[CH];
      Vec<float>::load(a.gamma + ch0, gam);
      Vec<float>::load(a.beta + ch0, bet);
      if constexpr (CH == 8) { Vec<float>::load(a.gamma + ch0 + 4, gam + 4); Vec<float>::load(a.beta + ch0 + 4, bet + 4); }
      if (cpg >= CH) {
        // a 16-byte chunk spans at most two groups: fetch their statistics once instead of once per channel
        // (the per-channel form cost 48 dependent scalar loads per thread, as much as the thread's whole pixel walk)
        const int g0 = ch0 / cpg, g1 = min(g0 + 1, a.G - 1);
        float me[2] = {0.f, 0.f}, rs[2] = {0.f, 0.f}, t1[2] = {0.f, 0.f}, t2[2] = {0.f, 0.f};
        if (MODE == MODE_PRIMAL && !STATS && a.red) {
          me[0] = lsum[2 * g0]; rs[0] = lsum[2 * g0 + 1]; me[1] = lsum[2 * g1]; rs[1] = lsum[2 * g1 + 1];
        } else if (MODE != MODE_PRIMAL || !STATS) {
          me[0] = (float)a.pstats[((long)b * a.G + g0) * 2]; rs[0] = (float)a.pstats[((long)b * a.G + g0) * 2 + 1];
          me[1] = (float)a.pstats[((long)b * a.G + g1) * 2]; rs[1] = (float)a.pstats[((long)b * a.G + g1) * 2 + 1];
        }
        if (MODE != MODE_PRIMAL && !STATS && a.red) {
          t1[0] = lsum[2 * g0]; t2[0] = lsum[2 * g0 + 1]; t1[1] = lsum[2 * g1]; t2[1] = lsum[2 * g1 + 1];
        } else if (MODE != MODE_PRIMAL && !STATS) {
          t1[0] = (float)(a.tstats[((long)j * a.G + g0) * 2] * inv_n); t2[0] = (float)(a.tstats[((long)j * a.G + g0) * 2 + 1] * inv_n);
          t1[1] = (float)(a.tstats[((long)j * a.G + g1) * 2] * inv_n); t2[1] = (float)(a.tstats[((long)j * a.G + g1) * 2 + 1] * inv_n);
        }
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          const int hi = (ch0 + e) >= (g0 + 1) * cpg ? 1 : 0;
          grp[e] = g0 + hi;
          mean[e] = me[hi]; rstd[e] = rs[hi]; m1[e] = t1[hi]; m2[e] = t2[hi];
        }
      } else {
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          grp[e] = (ch0 + e) / cpg;
          mean[e] = rstd[e] = m1[e] = m2[e] = 0.f;
          if (MODE == MODE_PRIMAL && !STATS && a.red) {
            mean[e] = lsum[2 * grp[e]]; rstd[e] = lsum[2 * grp[e] + 1];
          } else if (MODE != MODE_PRIMAL || !STATS) {
            mean[e] = (float)a.pstats[((long)b * a.G + grp[e]) * 2];
            rstd[e] = (float)a.pstats[((long)b * a.G + grp[e]) * 2 + 1];
          }
          if (MODE != MODE_PRIMAL && !STATS && a.red) {
            m1[e] = lsum[2 * grp[e]]; m2[e] = lsum[2 * grp[e] + 1];
          } else if (MODE != MODE_PRIMAL && !STATS) {
            m1[e] = (float)(a.tstats[((long)j * a.G + grp[e]) * 2] * inv_n);
            m2[e] = (float)(a.tstats[((long)j * a.G + grp[e]) * 2 + 1] * inv_n);
          }
        }
      }
      float s1[CH], s2[CH];
#pragma unroll
      for (int e = 0; e < CH; ++e) s1[e] = s2[e] = 0.f;
      for (int p = p0 + r; p < p1; p += rpi) {
        float x[CH], d[CH], o[CH];
        Vec<T>::load((const T*)a.x + ((long)b * a.HW + p) * a.C + ch0, x);
        if (MODE != MODE_PRIMAL) Vec<T>::load((const T*)a.d + ((long)j * a.HW + p) * a.C + ch0, d);
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          if (MODE == MODE_PRIMAL) {
            if (STATS) {
              s1[e] += x[e];
              s2[e] += x[e] * x[e];
            } else {
              float y = (x[e] - mean[e]) * rstd[e] * gam[e] + bet[e];
              o[e] = a.silu ? silu_(y) : y;
            }
          } else {
            float xh = (x[e] - mean[e]) * rstd[e];
            float y = gam[e] * xh + bet[e];
            float act = a.silu ? dsilu_(y) : 1.f;
            float v = (MODE == MODE_TANGENT) ? d[e] : gam[e] * act * d[e];
            if (STATS) {
              s1[e] += v;
              s2[e] += xh * v;
            } else {
              float w = rstd[e] * (v - m1[e] - xh * m2[e]);
              o[e] = (MODE == MODE_TANGENT) ? gam[e] * act * w : w;
            }
          }
        }
        if (!STATS) {
          T* yp = (T*)a.y + ((long)j * a.HW + p) * a.C + ch0;
          if (a.accumulate) {
            float old[CH];
            Vec<T>::load(yp, old);
#pragma unroll
            for (int e = 0; e < CH; ++e) o[e] += old[e];
          }
          vec_store_at<T>(yb, yp, o);
        }
      }
      if (STATS && a.det && cpg >= CH) {
        // a 16-byte chunk spans at most two groups: two (sum, sum) pairs per thread, [row][chunk column][slot] in LDS
        float p0 = 0.f, p1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          if (grp[e] == grp[0]) { p0 += s1[e]; p1 += s2[e]; } else { q0 += s1[e]; q1 += s2[e]; }
        }
        float4* lp = reinterpret_cast<float4*>(lch);
        lp[(long)r * cols + col] = make_float4(p0, p1, q0, q1);
      } else if (STATS && a.det) {
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          lch[((long)r * a.C + ch0 + e) * 2] = s1[e];
          lch[((long)r * a.C + ch0 + e) * 2 + 1] = s2[e];
        }
      } else if (STATS) {
        // a 16-byte chunk spans at most a few groups: combine equal-group channels in registers first, so a thread
        // issues one pair of LDS atomics per group it touches instead of one pair per channel
        float g1 = s1[0], g2 = s2[0];
#pragma unroll
        for (int e = 1; e < CH; ++e) {
          if (grp[e] != grp[e - 1]) {
            atomicAdd(&lsum[grp[e - 1] * 2], g1);
            atomicAdd(&lsum[grp[e - 1] * 2 + 1], g2);
            g1 = 0.f; g2 = 0.f;
          }
          g1 += s1[e]; g2 += s2[e];
        }
        atomicAdd(&lsum[grp[CH - 1] * 2], g1);
        atomicAdd(&lsum[grp[CH - 1] * 2 + 1], g2);
      }
    }
  }
  if (STATS && !a.det) {
    __syncthreads();
    double* dst = (MODE == MODE_PRIMAL) ? a.pstats : a.tstats;
    for (int i = tid; i < 2 * a.G; i += 256) atomicAdd(&dst[(long)j * a.G * 2 + i], (double)lsum[i]);
  }
  if (STATS && a.det) {
    __syncthreads();
    // ordered in-block reduction: thread i < 2G adds the rpi x cpg per-channel partials of its group, always in the same order, and stores the
    // block's partial; the apply launch adds the blocks' partials in block order (no ticket, no fence: the kernel boundary orders them)
    float* part = a.part + ((long)j * gridDim.x + blockIdx.x) * 2 * a.G;
    for (int i = tid; i < 2 * a.G; i += 256) {
      const int g = i >> 1, w = i & 1;
      float acc = 0.f;
      if (cpg >= CH) {                                  // chunk columns touching group g, rows in order; slot 0 = the chunk's first group
        const int c_lo = g * cpg / CH, c_hi = (g * cpg + cpg - 1) / CH;
        for (int c = c_lo; c <= c_hi; ++c) {
          const int slot = (c * CH / cpg == g) ? 0 : 2;
          for (int rr = 0; rr < rpi; ++rr) acc += lch[((long)rr * cols + c) * 4 + slot + w];
        }
      } else {
        for (int rr = 0; rr < rpi; ++rr)
          for (int c = 0; c < cpg; ++c) acc += lch[((long)rr * a.C + g * cpg + c) * 2 + w];
      }
      part[i] = acc;
    }
  }
}

// ---------------------------------------------------------------- GroupNorm in ONE launch (small feature maps)
// A block owns GC whole groups -- a contiguous window of CW = GC * C/G channels -- of one sample / tangent and keeps its x (and v)
// chunks in registers between the statistics phase and the apply phase: one read of every input instead of two, one launch instead of
// two, and a fixed-order block reduction.  Used when the window of one sample fits the block's registers (8x8 ... 32x32 levels of the
// U-Nets); larger maps take the two-pass kernels above.
template <typename T, int MODE, int MAXC>
__global__ __launch_bounds__(512) void gn_fused_kernel(GNArgs a, int GC) {
  constexpr int CH = TT<T>::CH, NT = 512;
  const OutBuf yb = out_buf(a.y, (long)max(a.NT, a.Bp) * a.HW * a.C * (long)sizeof(T));
  __shared__ float red[4][NT];          // per-thread partials: [slot * 2 + stat][thread]
  __shared__ float seg[4 * 64 * 8];     // per (statistic, chunk column): 8 row-segment sums
  __shared__ double colsum[4][64];      // per chunk column of the window
  __shared__ float gst[2][64];          // per group of the window: the two statistics (means)
  __shared__ double gsum[2][32];        // primal: raw fp64 group sums awaiting (mean, rstd)
  const int tid = threadIdx.x;
  const int j = blockIdx.y, b = (MODE == MODE_PRIMAL) ? j : j / a.kps;
  const int cpg = a.C / a.G, CW = GC * cpg, CPC = CW / CH;
  const int gbase = blockIdx.x * GC, ch_base = gbase * cpg;
  const int ppi = NT / CPC, NTa = ppi * CPC;                 // pixels per sweep, active threads
  const bool active = tid < NTa;
  const int c = tid % CPC, pr = tid / CPC;
  const int ch0 = ch_base + c * CH;
  const int g0 = ch0 / cpg;                                    // the chunk touches groups g0 and (maybe) g0 + 1   (cpg >= CH)
  const int split = (g0 + 1) * cpg - ch0;                      // elements e >= split belong to g0 + 1
  const double inv_n = 1.0 / ((double)a.HW * cpg);
  float gam[CH], bet[CH];
  float mean[2] = {0.f, 0.f}, rstd[2] = {0.f, 0.f};
  if (active) {
    Vec<float>::load(a.gamma + ch0, gam);
    Vec<float>::load(a.beta + ch0, bet);
    if constexpr (CH == 8) { Vec<float>::load(a.gamma + ch0 + 4, gam + 4); Vec<float>::load(a.beta + ch0 + 4, bet + 4); }
    if (MODE != MODE_PRIMAL) {
      const int g1 = min(g0 + 1, a.G - 1);
      mean[0] = (float)a.pstats[((long)b * a.G + g0) * 2]; rstd[0] = (float)a.pstats[((long)b * a.G + g0) * 2 + 1];
      mean[1] = (float)a.pstats[((long)b * a.G + g1) * 2]; rstd[1] = (float)a.pstats[((long)b * a.G + g1) * 2 + 1];
    }
  }
  uint4 xr[MAXC], dr[MAXC];
  float s[4] = {0.f, 0.f, 0.f, 0.f};                           // [slot][stat]
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int p = pr + i * ppi;
    if (active && p < a.HW) {
      xr[i] = *reinterpret_cast<const uint4*>((const T*)a.x + ((long)b * a.HW + p) * a.C + ch0);
      if (MODE != MODE_PRIMAL) dr[i] = a.src.slab ? slab_chunk<T>(a.src, (long)j * a.HW + p, ch0) : *reinterpret_cast<const uint4*>((const T*)a.d + ((long)j * a.HW + p) * a.C + ch0);
    }
  }
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int p = pr + i * ppi;
    if (active && p < a.HW) {
      float x[CH], d[CH];
      Raw<T>::unpack(xr[i], x);
      if (MODE != MODE_PRIMAL) Raw<T>::unpack(dr[i], d);
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        const int hi = e >= split ? 1 : 0;
        if (MODE == MODE_PRIMAL) {
          s[hi * 2] += x[e];
          s[hi * 2 + 1] += x[e] * x[e];
        } else {
          const float xh = (x[e] - mean[hi]) * rstd[hi];
          const float y = gam[e] * xh + bet[e];
          const float act = a.silu ? dsilu_(y) : 1.f;
          const float v = (MODE == MODE_TANGENT) ? d[e] : gam[e] * act * d[e];
          s[hi * 2] += v;
          s[hi * 2 + 1] += xh * v;
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) red[q][tid] = active ? s[q] : 0.f;
  __syncthreads();
  // column sums over the block's pixel rows in two fixed-order stages (8 segments of rows, then the 8 segment sums): a single serial
  // sweep of up to 102 dependent LDS reads per column cost more than the rest of the kernel
  {
    const int nseg = 8, seg_len = (ppi + nseg - 1) / nseg;
    if (tid < 4 * CPC * nseg) {
      const int sg = tid % nseg, qc = tid / nseg, q = qc / CPC, col = qc % CPC;
      float acc = 0.f;
      const int r0 = sg * seg_len, r1 = min(r0 + seg_len, ppi);
      for (int r = r0; r < r1; ++r) acc += red[q][col + CPC * r];
      seg[qc * nseg + sg] = acc;
    }
    __syncthreads();
    if (tid < 4 * CPC) {
      const int q = tid / CPC, col = tid % CPC;
      double acc = 0.0;
#pragma unroll
      for (int sg = 0; sg < nseg; ++sg) acc += (double)seg[tid * nseg + sg];
      colsum[q][col] = acc;
    }
  }
  __syncthreads();
  if (tid < 2 * GC) {                                         // group sums over the chunk columns, in column order
    const int gl = tid >> 1, w = tid & 1;
    double acc = 0.0;
    for (int col = 0; col < CPC; ++col) {
      const int cg = (ch_base + col * CH) / cpg - gbase;
      if (cg == gl) acc += colsum[w][col];
      if (cg + 1 == gl) acc += colsum[2 + w][col];
    }
    if (MODE == MODE_PRIMAL) {
      gsum[w][gl] = acc;                                      // finalised below (needs both sums)
    } else {
      gst[w][gl] = (float)(acc * inv_n);
    }
  }
  __syncthreads();
  if (MODE == MODE_PRIMAL) {
    if (tid < GC) {
      const double m = gsum[0][tid] * inv_n;
      double v = gsum[1][tid] * inv_n - m * m;
      if (v < 0) v = 0;
      const double rs = 1.0 / sqrt(v + (double)a.eps);
      a.pstats[((long)j * a.G + gbase + tid) * 2] = m;
      a.pstats[((long)j * a.G + gbase + tid) * 2 + 1] = rs;
      gst[0][tid] = (float)m;
      gst[1][tid] = (float)rs;
    }
    __syncthreads();
  }
  if (!active) return;
  const int l0 = g0 - gbase, l1 = min(l0 + 1, GC - 1);
  const float t1[2] = {gst[0][l0], gst[0][l1]}, t2[2] = {gst[1][l0], gst[1][l1]};   // primal: (mean, rstd); else (mean v, mean xhat v)
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int p = pr + i * ppi;
    if (p < a.HW) {
      float x[CH], d[CH], o[CH];
      Raw<T>::unpack(xr[i], x);
      if (MODE != MODE_PRIMAL) Raw<T>::unpack(dr[i], d);
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        const int hi = e >= split ? 1 : 0;
        if (MODE == MODE_PRIMAL) {
          const float y = (x[e] - t1[hi]) * t2[hi] * gam[e] + bet[e];
          o[e] = a.silu ? silu_(y) : y;
        } else {
          const float xh = (x[e] - mean[hi]) * rstd[hi];
          const float y = gam[e] * xh + bet[e];
          const float act = a.silu ? dsilu_(y) : 1.f;
          const float v = (MODE == MODE_TANGENT) ? d[e] : gam[e] * act * d[e];
          const float w = rstd[hi] * (v - t1[hi] - xh * t2[hi]);
          o[e] = (MODE == MODE_TANGENT) ? gam[e] * act * w : w;
        }
      }
      T* yp = (T*)a.y + ((long)j * a.HW + p) * a.C + ch0;
      if (a.accumulate) {
        float old[CH];
        Vec<T>::load(yp, old);
#pragma unroll
        for (int e = 0; e < CH; ++e) o[e] += old[e];
      }
      {
        const uint4 pk = Raw<T>::pack(o);
        if constexpr (std::is_same<T, float>::value) *reinterpret_cast<uint4*>(yp) = pk;
        else store_out16_at(yb, yp, u32x4_{pk.x, pk.y, pk.z, pk.w});
      }
    }
  }
}

// window of GC groups for the one-launch kernel: the narrowest 16-byte aligned window of >= 64 bytes whose pixels fit 4 chunks per thread
// (measured, profiles/r02_*: 8x8 / 16x16 maps 6-10 us against 2 x 7 us for the two passes; at 32x32 only 40-80 blocks carry 16 chunks per
// thread and the launch takes 24-32 us against 18 us -- those maps stay on the two-pass kernels)
static int gn_fused_groups(int C, int G, int HW, int CH, int es) {
  static const int on = getenv("DPB_GN_FUSED") ? atoi(getenv("DPB_GN_FUSED")) : 1;   // tuning switch (0: always two passes)
  const int cpg = C / G;
  if (!on || cpg < CH) return 0;
  for (int gc = 1; gc <= G && gc <= 32; gc <<= 1) {
    const int cw = gc * cpg;
    if (G % gc || cw % CH || cw * es < 64 || cw / CH > 16) continue;      // (4 statistics x chunk columns x 8 row segments <= 512 threads)
    const int cpc = cw / CH, ppi = 512 / cpc;
    if (ppi < 1) return 0;
    return (HW + ppi - 1) / ppi <= 4 ? gc : 0;
  }
  return 0;
}

// Large maps (more than GN_RED_MAX statistics blocks per sample: DDPM 64x64 and up, the image autoencoder): one small launch adds the per-block
// partials in the same fixed order into the fp64 statistics (and finalises the primal ones), so the apply blocks need not each walk them.
constexpr int GN_RED_MAX = 256;
template <int MODE>
__global__ __launch_bounds__(1024) void gn_reduce_kernel(GNArgs a, int nblk) {
  // one block per sample / tangent: SEG = 1024 / 2G block-range segments per statistic, each added in block order with 32 clamped loads in flight
  // (a plain loop waits out one L2 round trip per partial: 60 us per launch on the 256 x 256 maps), then the segment sums in segment order
  __shared__ double lseg[1024];
  const int tid = threadIdx.x, j = blockIdx.x, n2 = 2 * a.G;
  const int SEG = max(1, 1024 / n2);
  const double inv_n = 1.0 / ((double)a.HW * (a.C / a.G));
  const float* pj = a.part + (long)j * nblk * n2;
  if (tid < SEG * n2) {
    const int q = tid / n2, i = tid - q * n2;
    const int b0 = (int)((long)nblk * q / SEG), b1 = (int)((long)nblk * (q + 1) / SEG);
    double acc = 0.0;
    for (int bb = b0; bb < b1; bb += 32) {
      float v[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) v[u] = pj[(long)min(bb + u, b1 - 1) * n2 + i];
#pragma unroll
      for (int u = 0; u < 32; ++u) acc += bb + u < b1 ? (double)v[u] : 0.0;
    }
    lseg[q * n2 + i] = acc;
  }
  __syncthreads();
  auto total = [&](int i) { double t = 0.0; for (int q = 0; q < SEG; ++q) t += lseg[q * n2 + i]; return t; };
  double* dst = (MODE == MODE_PRIMAL) ? a.pstats : a.tstats;
  if (MODE != MODE_PRIMAL) {
    for (int i = tid; i < n2; i += 1024) dst[(long)j * n2 + i] = total(i);
  } else {
    for (int g = tid; g < a.G; g += 1024) {
      const double s1 = total(2 * g), s2 = total(2 * g + 1);
      const double m = s1 * inv_n;
      double v = s2 * inv_n - m * m;
      if (v < 0) v = 0;
      dst[((long)j * a.G + g) * 2] = m;
      dst[((long)j * a.G + g) * 2 + 1] = 1.0 / sqrt(v + (double)a.eps);
    }
  }
}

__global__ void gn_finalize(double* st, int n_groups, double inv_n, double eps) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_groups) return;
  double m = st[2 * i] * inv_n;
  double v = st[2 * i + 1] * inv_n - m * m;
  if (v < 0) v = 0;
  st[2 * i] = m;
  st[2 * i + 1] = 1.0 / sqrt(v + eps);
}

// pixels per block and statistics blocks per sample / tangent of the two-pass kernels: ONE rule for gn_launch and for groupnorm_launches
// (the engine's launch count and its "is this a one-launch GroupNorm" test), so the two cannot drift
static int gn_two_pass_ppb(int HW, int n) {
  static const long gn_blocks = getenv("DPB_GN_BLOCKS") ? atol(getenv("DPB_GN_BLOCKS")) : 512;   // tuning override
  int ppb = 64;
  while (ppb > 8 && (long)((HW + ppb - 1) / ppb) * n < gn_blocks) ppb >>= 1;
  return ppb;
}
// gn_kernel's static LDS (lsum 2 KB + lseg 16 KB) + up to 48 KB dynamic = 66 KB: gfx950 (160 KB) only, like the rest of this library

template <typename T, int MODE>
static int gn_launch(const GNArgs& a, hipStream_t st) {
  constexpr int CH = TT<T>::CH;
  if (a.C % CH || a.C % a.G || a.G > 256) { set_error("groupnorm: C=%d G=%d unsupported", a.C, a.G); return -1; }
  const int n = (MODE == MODE_PRIMAL) ? a.Bp : a.NT;
  if (const int gc = gn_fused_groups(a.C, a.G, a.HW, CH, (int)sizeof(T))) {
    const int cpc = gc * (a.C / a.G) / CH, ppi = 512 / cpc, sweeps = (a.HW + ppi - 1) / ppi;
    dim3 grid(a.G / gc, n);
    if (sweeps > 4) { set_error("groupnorm: %d sweeps exceed the one-launch kernel's register window", sweeps); return -1; }
    hipLaunchKernelGGL((gn_fused_kernel<T, MODE, 4>), grid, dim3(512), 0, st, a, gc);
    DPB_CHECK(hipGetLastError());
    return 0;
  }
  if (a.src.slab) { set_error("groupnorm: split-K slab input is taken by the one-launch kernel only"); return -1; }
  const int ppb = gn_two_pass_ppb(a.HW, n);
  dim3 grid((a.HW + ppb - 1) / ppb, n);
  size_t lds = 0;
  if (a.det) {
    const int cols = a.C / CH, cw = cols < 256 ? cols : 256, rpi = 256 / cw;
    lds = (size_t)rpi * a.C * 2 * sizeof(float);
    if (!a.part || (size_t)grid.x * n * 2 * a.G * sizeof(float) > a.part_bytes) {
      set_error("groupnorm: statistics scratch missing or too small (%u blocks x %d x %d groups)", grid.x, n, a.G);
      return -1;
    }
    if (lds > 48 * 1024) { set_error("groupnorm: C=%d needs %zu bytes of LDS for the ordered reduction", a.C, lds); return -1; }
  }           // (atomic path: the caller has zeroed pstats / tstats -- one memset per pass in the engine)
  GNArgs b = a;
  b.red = a.det && (int)grid.x <= GN_RED_MAX;     // the apply blocks add the partials themselves (32 ... 256 L2-resident loads per thread quarter)
  hipLaunchKernelGGL((gn_kernel<T, MODE, true>), grid, dim3(256), lds, st, b, ppb);
  if (a.det && !b.red) hipLaunchKernelGGL((gn_reduce_kernel<MODE>), dim3(n), dim3(1024), 0, st, b, (int)grid.x);
  if (MODE == MODE_PRIMAL && !a.det) {
    int ng = a.Bp * a.G;
    hipLaunchKernelGGL(gn_finalize, dim3((ng + 255) / 256), dim3(256), 0, st, a.pstats, ng, 1.0 / ((double)a.HW * (a.C / a.G)), (double)a.eps);
  }
  hipLaunchKernelGGL((gn_kernel<T, MODE, false>), grid, dim3(256), 0, st, b, ppb);
  DPB_CHECK(hipGetLastError());
  return 0;
}

int groupnorm_launches(int dtype, int mode, const GNArgs& a) {   // kernels launch_groupnorm issues (engine statistics)
  if (gn_fused_groups(a.C, a.G, a.HW, dt_chunk(dtype), dtype == DT_F32 ? 4 : 2)) return 1;
  if (!a.det) return mode == MODE_PRIMAL ? 3 : 2;
  const int n = mode == MODE_PRIMAL ? a.Bp : a.NT;
  const int ppb = gn_two_pass_ppb(a.HW, n);
  return (a.HW + ppb - 1) / ppb <= GN_RED_MAX ? 2 : 3;
}

int launch_groupnorm(int dtype, int mode, const GNArgs& a, hipStream_t st) {
  if (mode == MODE_PRIMAL) return DPB_DISPATCH_T(dtype, T, (gn_launch<T, MODE_PRIMAL>(a, st)));
  if (mode == MODE_TANGENT) return DPB_DISPATCH_T(dtype, T, (gn_launch<T, MODE_TANGENT>(a, st)));
  return DPB_DISPATCH_T(dtype, T, (gn_launch<T, MODE_ADJOINT>(a, st)));
}

// ---------------------------------------------------------------- LayerNorm: one wave per token row
template <typename T, int MODE, int MAXI>           // MAXI*64 >= chunks per row (C <= 1280 f32 / 2560 bf16 at MAXI = 5): sized per launch so
__global__ __launch_bounds__(256) void ln_kernel(LNArgs a, long nrows) {   // that the 320- / 640-channel rows do not carry 80 idle registers
  constexpr int CH = TT<T>::CH;
  const OutBuf yb = out_buf(a.y, nrows * a.C * (long)sizeof(T));
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int nch = a.C / CH;
  long prow = row;
  if (MODE != MODE_PRIMAL) {
    long j = row / a.rows_per_sample, l = row - j * a.rows_per_sample;
    prow = (j / a.kps) * a.rows_per_sample + l;
  }
  const T* xp = (const T*)a.x + prow * a.C;
  float x[MAXI][CH];
  float v[MAXI][CH];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    int c = lane + i * 64;
    if (c < nch) {
      Vec<T>::load(xp + c * CH, x[i]);
      if (MODE != MODE_PRIMAL) {                                                          // issued with x: both in flight across the reductions
        if (a.src.slab) Raw<T>::unpack(slab_chunk<T>(a.src, row, c * CH), v[i]);
        else Vec<T>::load((const T*)a.d + row * a.C + c * CH, v[i]);
      }
#pragma unroll
      for (int e = 0; e < CH; ++e) s += x[i][e];
    }
  }
  const float inv_c = 1.f / a.C;
  const float mean = wave_sum(s) * inv_c;
  float vs = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    int c = lane + i * 64;
    if (c < nch) {
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        x[i][e] -= mean;
        vs += x[i][e] * x[i][e];
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(vs) * inv_c + a.eps);
  if (MODE == MODE_PRIMAL) {
    T* yp = (T*)a.y + row * a.C;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      int c = lane + i * 64;
      if (c < nch) {
        float o[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) o[e] = x[i][e] * rstd * a.gamma[c * CH + e] + a.beta[c * CH + e];
        vec_store_at<T>(yb, yp + c * CH, o);
      }
    }
    return;
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    int c = lane + i * 64;
    if (c < nch) {
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        if (MODE == MODE_ADJOINT) v[i][e] *= a.gamma[c * CH + e];
        x[i][e] *= rstd;                          // xhat
        s1 += v[i][e];
        s2 += x[i][e] * v[i][e];
      }
    }
  }
  const float m1 = wave_sum(s1) * inv_c, m2 = wave_sum(s2) * inv_c;
  T* yp = (T*)a.y + row * a.C;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    int c = lane + i * 64;
    if (c < nch) {
      float o[CH];
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        float w = rstd * (v[i][e] - m1 - x[i][e] * m2);
        o[e] = (MODE == MODE_TANGENT) ? w * a.gamma[c * CH + e] : w;
      }
      if (a.accumulate) {
        float old[CH];
        Vec<T>::load(yp + c * CH, old);
#pragma unroll
        for (int e = 0; e < CH; ++e) o[e] += old[e];
      }
      vec_store_at<T>(yb, yp + c * CH, o);
    }
  }
}

// The same map with LPR lanes per row (64 / LPR rows per wave), for row lengths that are LPR x 3..5 chunks: with one wave per row the
// 320- / 640-channel rows of the 64 x 64 and 32 x 32 levels keep only 40 of 64 (80 of 128) lanes busy.  Lane l of a row group owns chunks
// l, l + LPR, ...: the LPR lanes of one load instruction still cover LPR x 16 contiguous bytes of the row.
template <int LPR>
__device__ inline float seg_sum(float v) {
#pragma unroll
  for (int o = LPR >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <typename T, int MODE, int LPR, int NI>
__global__ __launch_bounds__(256) void ln_rows_kernel(LNArgs a, long nrows) {
  constexpr int CH = TT<T>::CH, RPW = 64 / LPR;
  const OutBuf yb = out_buf(a.y, nrows * a.C * (long)sizeof(T));
  const int lane = threadIdx.x & 63, l = lane % LPR;
  const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
  const bool live = row < nrows;                 // whole row groups: every lane of a group agrees, the shuffles stay inside the group
  const long r = live ? row : nrows - 1;
  long prow = r;
  if (MODE != MODE_PRIMAL) {
    long j = r / a.rows_per_sample, ll = r - j * a.rows_per_sample;
    prow = (j / a.kps) * a.rows_per_sample + ll;
  }
  const T* xp = (const T*)a.x + prow * a.C;
  float x[NI][CH];
  float v[NI][CH];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = l + i * LPR;
    Vec<T>::load(xp + c * CH, x[i]);
    if (MODE != MODE_PRIMAL) {
      if (a.src.slab) { if (live) Raw<T>::unpack(slab_chunk<T>(a.src, r, c * CH), v[i]); else for (int e = 0; e < CH; ++e) v[i][e] = 0.f; }
      else Vec<T>::load((const T*)a.d + r * a.C + c * CH, v[i]);
    }
#pragma unroll
    for (int e = 0; e < CH; ++e) s += x[i][e];
  }
  const float inv_c = 1.f / a.C;
  const float mean = seg_sum<LPR>(s) * inv_c;
  float vs = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      x[i][e] -= mean;
      vs += x[i][e] * x[i][e];
    }
  const float rstd = rsqrtf(seg_sum<LPR>(vs) * inv_c + a.eps);
  T* yp = (T*)a.y + r * a.C;
  if (MODE == MODE_PRIMAL) {
    if (!live) return;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = l + i * LPR;
      float o[CH];
#pragma unroll
      for (int e = 0; e < CH; ++e) o[e] = x[i][e] * rstd * a.gamma[c * CH + e] + a.beta[c * CH + e];
      vec_store_at<T>(yb, yp + c * CH, o);
    }
    return;
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = l + i * LPR;
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      if (MODE == MODE_ADJOINT) v[i][e] *= a.gamma[c * CH + e];
      x[i][e] *= rstd;                            // xhat
      s1 += v[i][e];
      s2 += x[i][e] * v[i][e];
    }
  }
  const float m1 = seg_sum<LPR>(s1) * inv_c, m2 = seg_sum<LPR>(s2) * inv_c;
  if (!live) return;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = l + i * LPR;
    float o[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      float w = rstd * (v[i][e] - m1 - x[i][e] * m2);
      o[e] = (MODE == MODE_TANGENT) ? w * a.gamma[c * CH + e] : w;
    }
    if (a.accumulate) {
      float old[CH];
      Vec<T>::load(yp + c * CH, old);
#pragma unroll
      for (int e = 0; e < CH; ++e) o[e] += old[e];
    }
    vec_store_at<T>(yb, yp + c * CH, o);
  }
}

template <typename T, int MODE, int LPR>
static bool ln_rows_launch(const LNArgs& a, long nrows, int ni, hipStream_t st) {
  const dim3 grid((unsigned)((nrows + 4 * (64 / LPR) - 1) / (4 * (64 / LPR))));
  if (ni == 3) hipLaunchKernelGGL((ln_rows_kernel<T, MODE, LPR, 3>), grid, dim3(256), 0, st, a, nrows);
  else if (ni == 5) hipLaunchKernelGGL((ln_rows_kernel<T, MODE, LPR, 5>), grid, dim3(256), 0, st, a, nrows);
  else return false;
  return true;
}

template <typename T, int MODE>
static int ln_launch(const LNArgs& a, hipStream_t st) {
  constexpr int CH = TT<T>::CH;
  if (a.C % CH || a.C / CH > 5 * 64) { set_error("layernorm: C=%d unsupported", a.C); return -1; }
  long nrows = (long)((MODE == MODE_PRIMAL) ? a.Bp : a.NT) * a.rows_per_sample;
  {  // several rows per wave when the row is 8 / 16 / 32 lanes x 3 or 5 chunks (C = 320, 640, 1280; 768: CLIP) -- every lane busy
    static const int rows_env = getenv("DPB_LN_ROWS") ? atoi(getenv("DPB_LN_ROWS")) : 1;   // tuning switch
    const int nch = a.C / CH;
    bool done = false;
    if (rows_env && nrows > 0) {
      for (int lpr = 8; lpr <= 32 && !done; lpr *= 2) {
        if (nch % lpr || (nch / lpr != 3 && nch / lpr != 5)) continue;
        if (lpr == 8) done = ln_rows_launch<T, MODE, 8>(a, nrows, nch / lpr, st);
        else if (lpr == 16) done = ln_rows_launch<T, MODE, 16>(a, nrows, nch / lpr, st);
        else done = ln_rows_launch<T, MODE, 32>(a, nrows, nch / lpr, st);
      }
    }
    if (done) { DPB_CHECK(hipGetLastError()); return 0; }
  }
  const dim3 grid((unsigned)((nrows + 3) / 4));
  const int need = (a.C / CH + 63) / 64;
  if (need <= 1) hipLaunchKernelGGL((ln_kernel<T, MODE, 1>), grid, dim3(256), 0, st, a, nrows);
  else if (need <= 2) hipLaunchKernelGGL((ln_kernel<T, MODE, 2>), grid, dim3(256), 0, st, a, nrows);
  else if (need <= 3) hipLaunchKernelGGL((ln_kernel<T, MODE, 3>), grid, dim3(256), 0, st, a, nrows);
  else hipLaunchKernelGGL((ln_kernel<T, MODE, 5>), grid, dim3(256), 0, st, a, nrows);
  DPB_CHECK(hipGetLastError());
  return 0;
}

int launch_layernorm(int dtype, int mode, const LNArgs& a, hipStream_t st) {
  if (mode == MODE_PRIMAL) return DPB_DISPATCH_T(dtype, T, (ln_launch<T, MODE_PRIMAL>(a, st)));
  if (mode == MODE_TANGENT) return DPB_DISPATCH_T(dtype, T, (ln_launch<T, MODE_TANGENT>(a, st)));
  return DPB_DISPATCH_T(dtype, T, (ln_launch<T, MODE_ADJOINT>(a, st)));
}

}  // namespace dpb

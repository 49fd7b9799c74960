// GroupNorm(+SiLU) and LayerNorm for NHWC activations: primal, tangent (JVP) and adjoint (VJP wrt input).
//
// HBM-bound kernels.  Every thread owns a fixed 16-byte channel chunk (coalesced rows, per-channel
// constants loaded once) and walks pixels.  GroupNorm: pass 1 accumulates per-(sample,group) sums
// (LDS float atomics per block -> one fp64 global atomic per group per block), pass 2 applies.
// The tangent and adjoint share one algebraic form:
//     out = rstd * (v - mean(v) - xhat * mean(xhat * v))
// with v = dx (tangent; gamma and SiLU' applied after) or v = gamma * SiLU'(y) * gz (adjoint; before).
// Primal statistics are computed once per x_t and reused by all k tangents / cotangents.
#include "kernels.h"

namespace dpb {

template <typename T, int MODE, bool STATS>
__global__ __launch_bounds__(256) void gn_kernel(GNArgs a, int ppb) {
  constexpr int CH = TT<T>::CH;
  __shared__ float lsum[2 * 256];   // up to 256 groups... G <= 128 used: [G][2]
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
  if (STATS) {
    for (int i = tid; i < 2 * a.G; i += 256) lsum[i] = 0.f;
    __syncthreads();
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
        if (MODE != MODE_PRIMAL || !STATS) {
          me[0] = (float)a.pstats[((long)b * a.G + g0) * 2]; rs[0] = (float)a.pstats[((long)b * a.G + g0) * 2 + 1];
          me[1] = (float)a.pstats[((long)b * a.G + g1) * 2]; rs[1] = (float)a.pstats[((long)b * a.G + g1) * 2 + 1];
        }
        if (MODE != MODE_PRIMAL && !STATS) {
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
          if (MODE != MODE_PRIMAL || !STATS) {
            mean[e] = (float)a.pstats[((long)b * a.G + grp[e]) * 2];
            rstd[e] = (float)a.pstats[((long)b * a.G + grp[e]) * 2 + 1];
          }
          if (MODE != MODE_PRIMAL && !STATS) {
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
          Vec<T>::store(yp, o);
        }
      }
      if (STATS) {
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
  if (STATS) {
    __syncthreads();
    double* dst = (MODE == MODE_PRIMAL) ? a.pstats : a.tstats;
    for (int i = tid; i < 2 * a.G; i += 256) atomicAdd(&dst[(long)j * a.G * 2 + i], (double)lsum[i]);
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

template <typename T, int MODE>
static int gn_launch(const GNArgs& a, hipStream_t st) {
  constexpr int CH = TT<T>::CH;
  if (a.C % CH || a.C % a.G || a.G > 256) { set_error("groupnorm: C=%d G=%d unsupported", a.C, a.G); return -1; }
  const int n = (MODE == MODE_PRIMAL) ? a.Bp : a.NT;
  int ppb = 64;
  static const long gn_blocks = getenv("DPB_GN_BLOCKS") ? atol(getenv("DPB_GN_BLOCKS")) : 512;   // tuning override
  while (ppb > 8 && (long)((a.HW + ppb - 1) / ppb) * n < gn_blocks) ppb >>= 1;
  dim3 grid((a.HW + ppb - 1) / ppb, n);
  hipLaunchKernelGGL((gn_kernel<T, MODE, true>), grid, dim3(256), 0, st, a, ppb);
  if (MODE == MODE_PRIMAL) {
    int ng = a.Bp * a.G;
    hipLaunchKernelGGL(gn_finalize, dim3((ng + 255) / 256), dim3(256), 0, st, a.pstats, ng,
                       1.0 / ((double)a.HW * (a.C / a.G)), (double)a.eps);
  }
  hipLaunchKernelGGL((gn_kernel<T, MODE, false>), grid, dim3(256), 0, st, a, ppb);
  DPB_CHECK(hipGetLastError());
  return 0;
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
      if (MODE != MODE_PRIMAL) Vec<T>::load((const T*)a.d + row * a.C + c * CH, v[i]);   // issued with x: both in flight across the reductions
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
        Vec<T>::store(yp + c * CH, o);
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
      Vec<T>::store(yp + c * CH, o);
    }
  }
}

template <typename T, int MODE>
static int ln_launch(const LNArgs& a, hipStream_t st) {
  constexpr int CH = TT<T>::CH;
  if (a.C % CH || a.C / CH > 5 * 64) { set_error("layernorm: C=%d unsupported", a.C); return -1; }
  long nrows = (long)((MODE == MODE_PRIMAL) ? a.Bp : a.NT) * a.rows_per_sample;
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

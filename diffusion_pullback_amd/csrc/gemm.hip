// MFMA GEMM / implicit-GEMM convolution for gfx950 (wave64).
//
//   C[z][m][n] = alpha * ( sum_k A[z][m][k] B[z][n][k]  +  sum_k A2[z][m][k] B2[z][n][k] )  (+bias) (+rowbias) (+R) (+C)
//
// One kernel serves every dense contraction of the pullback path: 3x3 / strided /
// transposed / upsampling convolutions (A rows gathered from NHWC pixels), 1x1
// convolutions and Linear layers (one tap), and the batched Q K^T / P V products of
// attention (two-level batch z = (tangent, head)).  Linear maps have identical
// primal, tangent and (with the pre-transposed weight) adjoint kernels, so the
// JVP batch and the VJP batch of the power iteration are both just larger M.
// The optional second operand pair extends the K loop (dS = dQ K^T + Q dK^T and
// dO = dP V + P dV are ONE launch each: no read-modify-write pass over the scores).
//
// Tiling: 256 threads = 4 waves (2x2); each wave owns (BM/2)x(BN/2) as 32x32 MFMA
// tiles.  bf16: v_mfma_f32_32x32x16_bf16, LDS tiles row-major [rows][32+8] so a
// fragment is one ds_read_b128; f32: v_mfma_f32_32x32x2_f32, LDS tiles k-major
// [16][rows+4] so a fragment is one conflict-free ds_read_b32.  K advances 4
// 16-byte chunks per step; global->register prefetch of step t+1 overlaps the MFMAs
// of step t, LDS is double buffered (one barrier per step).
// Epilogue: accumulators are staged through LDS (fp32) and leave as full 16/32-byte
// row segments with bias / time-embedding row bias / residual / accumulate fused.
// Small-M problems (8x8 feature maps: M = 64k rows) split K over blockIdx.z into
// fp32 slabs that a second kernel reduces (weights stream from HBM exactly once).
// Workgroup ids are remapped so that one XCD (= one L2) walks neighbouring tiles.
#include <cstdio>
#include <vector>

#include "kernels.h"

namespace dpb {


template <typename T> struct V8;   // 8 consecutive elements <-> float[8]
template <> struct V8<float> {
  __device__ static inline void load(const float* p, float* o) { Vec<float>::load(p, o); Vec<float>::load(p + 4, o + 4); }
  __device__ static inline void store(float* p, const float* o) { Vec<float>::store(p, o); Vec<float>::store(p + 4, o + 4); }
};
template <> struct V8<bf16> {
  __device__ static inline void load(const bf16* p, float* o) { Vec<bf16>::load(p, o); }
  __device__ static inline void store(bf16* p, const float* o) { Vec<bf16>::store(p, o); }
};
template <> struct V8<f16> {
  __device__ static inline void load(const f16* p, float* o) { Vec<f16>::load(p, o); }
  __device__ static inline void store(f16* p, const float* o) { Vec<f16>::store(p, o); }
};
template <typename T> struct FlavourOf { static constexpr int FL = 0; };
template <> struct FlavourOf<f16> { static constexpr int FL = 1; };

template <typename T, int BM, int BN, int KCH, int GATHER>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
  constexpr int CH = TT<T>::CH;
  constexpr int BK = KCH * CH;           // K elements per step (KCH 16-byte chunks per row)
  constexpr int RPP = 256 / KCH;          // tile rows covered by one pass of the 256 threads
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int LDA_S = F32 ? (BM + 4) : (BK + 8);
  constexpr int LDB_S = F32 ? (BN + 4) : (BK + 8);
  constexpr int A_ELEMS = F32 ? BK * LDA_S : BM * LDA_S;
  constexpr int B_ELEMS = F32 ? BK * LDB_S : BN * LDB_S;
  constexpr int NA = BM / RPP, NB = BN / RPP;
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr int SLD = WN + 4;                                   // stage row stride (floats)
  constexpr int MAIN_BYTES = 2 * (A_ELEMS + B_ELEMS) * (int)sizeof(T);
  constexpr int STAGE_BYTES = 4 * 32 * SLD * 4;
  constexpr int SMEM_BYTES = MAIN_BYTES > STAGE_BYTES ? MAIN_BYTES : STAGE_BYTES;
  __shared__ __attribute__((aligned(16))) char smem_raw[SMEM_BYTES];
  T* As = reinterpret_cast<T*>(smem_raw);
  T* Bs = As + 2 * A_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tilesN = (p.N + BN - 1) / BN;
  // XCD-aware remap (bijective): hardware places block b on XCD b % 8; give each XCD a contiguous tile range
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm_i = bid / tilesN, tn_i = bid % tilesN;
  const int m0 = tm_i * BM, n0 = tn_i * BN;
  const int z1 = blockIdx.y / p.Z2, z2 = blockIdx.y % p.Z2;

  const T* A = (const T*)p.A + (long)(z1 / p.divA) * p.sA1 + (long)z2 * p.sA2;
  const T* B = (const T*)p.B + (long)(z1 / p.divB) * p.sB1 + (long)z2 * p.sB2;
  T* C = (T*)p.C + (long)z1 * p.sC1 + (long)z2 * p.sC2;
  const T* R = p.R ? (const T*)p.R + (long)z1 * p.sR1 + (long)z2 * p.sR2 : nullptr;

  const int kq = tid % KCH, r0 = tid / KCH;
  // ---- per-thread A rows.  GATHER is a compile-time mode; the gathered pixel is resolved once per filter tap
  // (a_cur[i] = channel vector of the source pixel, nullptr = padding), so a K-step costs one add per chunk.
  const int K = p.K, Cin = p.Cin, Wd = p.W, Hd = p.H, lda = p.lda, strd = p.stride, pad = p.pad, KS = p.KS;
  const T* a_base[NA];
  const T* a_cur[NA];
  int a_oy[NA], a_ox[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    int row = r0 + i * RPP;
    int m = m0 + row;
    a_oy[i] = a_ox[i] = 0;
    if constexpr (GATHER == GATHER_NONE) {
      a_base[i] = m < p.M ? A + (long)m * lda : nullptr;
    } else {
      int hw = p.Ho * p.Wo;
      int smp = m / hw, rem = m - smp * hw;
      a_oy[i] = rem / p.Wo;
      a_ox[i] = rem - a_oy[i] * p.Wo;
      a_base[i] = m < p.M ? A + (long)smp * Hd * Wd * lda : nullptr;
    }
  }
  const T* b_base[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    int n = n0 + r0 + i * RPP;
    b_base[i] = n < p.N ? B + (long)n * p.ldb : nullptr;
  }
  // K range of this block (split-K over blockIdx.z; steps of BK)
  const int nk1 = (K + BK - 1) / BK;
  const int nk2 = p.A2 ? (p.K2 + BK - 1) / BK : 0;
  const int nk_all = nk1 + nk2;
  int kt0 = 0, kt1 = nk_all;
  if (p.splitk > 1) {
    const int per = (nk_all + p.splitk - 1) / p.splitk;
    kt0 = blockIdx.z * per;
    kt1 = min(nk_all, kt0 + per);
  }
  // running (k, tap, channel) of this thread's chunk column (all of a thread's chunks share the column)
  int kl = kt0;                       // next step to load
  int klim = K;
  int kc = kt0 * BK + kq * CH, tap = 0, cc = kc;
  if constexpr (GATHER != GATHER_NONE) {
    tap = kc / Cin;
    cc = kc - tap * Cin;
  }
  auto retap = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if constexpr (GATHER == GATHER_NONE) {
        a_cur[i] = a_base[i];
      } else {
        int ky = 0, kx = 0;
        if (KS == 3) { ky = (tap * 11) >> 5; kx = tap - ky * 3; }
        int iy, ix;
        bool ok = a_base[i] != nullptr;
        if constexpr (GATHER == GATHER_CONV) {
          iy = a_oy[i] * strd + ky - pad;
          ix = a_ox[i] * strd + kx - pad;
          ok = ok && iy >= 0 && iy < Hd && ix >= 0 && ix < Wd;
        } else if constexpr (GATHER == GATHER_CONVT) {
          int ty = a_oy[i] + pad - ky, tx = a_ox[i] + pad - kx;
          ok = ok && ty >= 0 && tx >= 0;
          if (strd == 2) { ok = ok && !((ty | tx) & 1); iy = ty >> 1; ix = tx >> 1; } else { iy = ty; ix = tx; }
          ok = ok && iy < Hd && ix < Wd;
        } else {   // GATHER_UPCONV: nearest x2 then 3x3 pad 1
          int uy = a_oy[i] + ky - 1, ux = a_ox[i] + kx - 1;
          ok = ok && uy >= 0 && ux >= 0 && uy < 2 * Hd && ux < 2 * Wd;
          iy = uy >> 1; ix = ux >> 1;
        }
        a_cur[i] = ok ? a_base[i] + ((long)iy * Wd + ix) * lda : nullptr;
      }
    }
  };
  retap();

  uint4 ra[NA], rb[NB];
  auto gload = [&]() {
    if (nk2 && kl == nk1) {           // switch to the second operand pair (plain rows only)
      const T* A2 = (const T*)p.A2 + (long)(z1 / p.divA2) * p.sA21 + (long)z2 * p.sA22;
      const T* B2 = (const T*)p.B2 + (long)(z1 / p.divB2) * p.sB21 + (long)z2 * p.sB22;
#pragma unroll
      for (int i = 0; i < NA; ++i) a_cur[i] = a_base[i] ? A2 + (long)(m0 + r0 + i * RPP) * p.lda2 : nullptr;
#pragma unroll
      for (int i = 0; i < NB; ++i) b_base[i] = b_base[i] ? B2 + (long)(n0 + r0 + i * RPP) * p.ldb2 : nullptr;
      kc = kq * CH;
      cc = kc;
      klim = p.K2;
    }
    ++kl;
    const bool kok = kc < klim;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      ra[i] = (kok && a_cur[i]) ? *reinterpret_cast<const uint4*>(a_cur[i] + cc) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NB; ++i)
      rb[i] = (kok && b_base[i]) ? *reinterpret_cast<const uint4*>(b_base[i] + kc) : make_uint4(0, 0, 0, 0);
    // advance to the next K step
    kc += BK;
    cc += BK;
    if constexpr (GATHER != GATHER_NONE) {
      if (cc >= Cin) {                  // next filter tap (every Cin/BK steps)
        do { cc -= Cin; ++tap; } while (cc >= Cin);
        retap();
      }
    }
  };
  auto sstore = [&](int buf) {
    T* as = As + buf * A_ELEMS;
    T* bs = Bs + buf * B_ELEMS;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int row = r0 + i * RPP;
      if constexpr (F32) {
        const float* f = reinterpret_cast<const float*>(&ra[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) as[(kq * 4 + j) * LDA_S + row] = f[j];
      } else {
        *reinterpret_cast<uint4*>(as + row * LDA_S + kq * 8) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int row = r0 + i * RPP;
      if constexpr (F32) {
        const float* f = reinterpret_cast<const float*>(&rb[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) bs[(kq * 4 + j) * LDB_S + row] = f[j];
      } else {
        *reinterpret_cast<uint4*>(bs + row * LDB_S + kq * 8) = rb[i];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wy = wave >> 1, wx = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;

  if (kt0 < kt1) {
    gload();
    sstore(0);
  }
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    if (kt + 1 < kt1) gload();
    const T* as = As + buf * A_ELEMS;
    const T* bs = Bs + buf * B_ELEMS;
    if constexpr (F32) {
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = as[(kk * 2 + lhi) * LDA_S + wy * WM + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = bs[(kk * 2 + lhi) * LDB_S + wx * WN + j * 32 + l31];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        bf16x8 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a[i] = *reinterpret_cast<const bf16x8*>(as + (wy * WM + i * 32 + l31) * LDA_S + kk * 16 + lhi * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          b[j] = *reinterpret_cast<const bf16x8*>(bs + (wx * WN + j * 32 + l31) * LDB_S + kk * 16 + lhi * 8);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = H16<FlavourOf<T>::FL>::mfma(a[i], b[j], acc[i][j]);
      }
    }
    if (kt + 1 < kt1) sstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue through LDS.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* stage = reinterpret_cast<float*>(smem_raw) + wave * 32 * SLD;
  constexpr int CPR = WN / 8;                    // 8-column chunks per staged row
  constexpr int ITEMS = 32 * CPR / 64;           // chunks per lane per 32-row slab
  float* slab = p.splitk > 1 ? p.slab + ((long)blockIdx.z * gridDim.y + blockIdx.y) * (long)p.M * p.N : nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * lhi) * SLD + j * 32 + l31] = acc[i][j][r];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int item = it * 64 + lane;
      const int row = item / CPR, c8 = item % CPR;
      const int m = m0 + wy * WM + i * 32 + row;
      const int n = n0 + wx * WN + c8 * 8;
      if (m >= p.M || n >= p.N) continue;
      float v[8];
      Vec<float>::load(stage + row * SLD + c8 * 8, v);
      Vec<float>::load(stage + row * SLD + c8 * 8 + 4, v + 4);
      if (slab) {                                   // split-K partial: raw fp32, reduced by splitk_reduce_kernel
        float* sp = slab + (long)m * p.N + n;
        if (n + 8 <= p.N && !(p.N & 3)) {
          Vec<float>::store(sp, v);
          Vec<float>::store(sp + 4, v + 4);
        } else {
          for (int e = 0; e < 8 && n + e < p.N; ++e) sp[e] = v[e];
        }
        continue;
      }
      const bool full = p.vec_ok && n + 8 <= p.N;
      int smp = 0;
      if (p.rowbias) smp = (m / p.rows_per_sample) / p.rowbias_div;
      T* cp = C + (long)m * p.ldc + n;
      if (full) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
        if (p.bias) {
          float b8[8];
          V8<float>::load(p.bias + n, b8);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += b8[e];
        }
        if (p.rowbias) {
          float b8[8];
          V8<T>::load((const T*)p.rowbias + (long)smp * p.N + n, b8);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += b8[e];
        }
        if (R) {
          float b8[8];
          V8<T>::load(R + (long)m * p.ldr + n, b8);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += b8[e];
        }
        if (p.accumulate) {
          float b8[8];
          V8<T>::load(cp, b8);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += b8[e];
        }
        V8<T>::store(cp, v);
      } else {
        for (int e = 0; e < 8 && n + e < p.N; ++e) {
          float x = p.alpha * v[e];
          if (p.bias) x += p.bias[n + e];
          if (p.rowbias) x += TT<T>::ld((const T*)p.rowbias + (long)smp * p.N + n + e);
          if (R) x += TT<T>::ld(R + (long)m * p.ldr + n + e);
          if (p.accumulate) x += TT<T>::ld(cp + e);
          TT<T>::st(cp + e, x);
        }
      }
    }
    __syncthreads();
  }
}

// sums the split-K fp32 slabs and applies the fused epilogue; 8 consecutive columns per thread when aligned
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs p) {
  const long MN = (long)p.M * p.N;
  const int Z = p.Z1 * p.Z2;
  if (p.vec_ok && !(p.N & 7)) {
    const int n8 = p.N >> 3;
    const long total = (long)p.M * n8 * Z;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
      const int c = (int)(idx % n8);
      const long mz = idx / n8;
      const int m = (int)(mz % p.M), z = (int)(mz / p.M);
      const int n = c * 8;
      const long mn = (long)m * p.N + n;
      const int z1 = z / p.Z2, z2 = z % p.Z2;
      T* cp = (T*)p.C + (long)z1 * p.sC1 + (long)z2 * p.sC2 + (long)m * p.ldc + n;
      // the epilogue operands are independent of the slabs: fetch them first, so that their latency lies under the slab reads instead of
      // behind the last sum
      float bias8[8], rb8[8], r8[8], old8[8];
      if (p.bias) V8<float>::load(p.bias + n, bias8);
      if (p.rowbias) V8<T>::load((const T*)p.rowbias + (long)((m / p.rows_per_sample) / p.rowbias_div) * p.N + n, rb8);
      if (p.R) V8<T>::load((const T*)p.R + (long)z1 * p.sR1 + (long)z2 * p.sR2 + (long)m * p.ldr + n, r8);
      if (p.accumulate) V8<T>::load(cp, old8);
      float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      int s = 0;
      for (; s + 8 <= p.splitk; s += 8) {          // eight, then four slabs in flight per thread; summed in slab order (the order is part of
        float t[8][8];                              // the bitwise kernel-equivalence contract)
#pragma unroll
        for (int u = 0; u < 8; ++u) V8<float>::load(p.slab + ((long)(s + u) * Z + z) * MN + mn, t[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += t[u][e];
      }
      if (s + 4 <= p.splitk) {
        float t[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) V8<float>::load(p.slab + ((long)(s + u) * Z + z) * MN + mn, t[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += t[u][e];
        s += 4;
      }
      for (; s < p.splitk; ++s) {
        float t[8];
        V8<float>::load(p.slab + ((long)s * Z + z) * MN + mn, t);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += t[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
      if (p.bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bias8[e];
      }
      if (p.rowbias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rb8[e];
      }
      if (p.R) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += r8[e];
      }
      if (p.accumulate) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += old8[e];
      }
      V8<T>::store(cp, v);
    }
    return;
  }
  const long total = MN * Z;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int z = (int)(idx / MN);
    const long mn = idx - (long)z * MN;
    const int m = (int)(mn / p.N), n = (int)(mn - (long)m * p.N);
    float acc = 0.f;
    for (int s = 0; s < p.splitk; ++s) acc += p.slab[((long)s * Z + z) * MN + mn];
    const int z1 = z / p.Z2, z2 = z % p.Z2;
    float x = p.alpha * acc;
    if (p.bias) x += p.bias[n];
    if (p.rowbias) x += TT<T>::ld((const T*)p.rowbias + (long)((m / p.rows_per_sample) / p.rowbias_div) * p.N + n);
    if (p.R) x += TT<T>::ld((const T*)p.R + (long)z1 * p.sR1 + (long)z2 * p.sR2 + (long)m * p.ldr + n);
    T* cp = (T*)p.C + (long)z1 * p.sC1 + (long)z2 * p.sC2 + (long)m * p.ldc + n;
    if (p.accumulate) x += TT<T>::ld(cp);
    TT<T>::st(cp, x);
  }
}

// tuning overrides (dpb_debug_set): 0 = heuristic
static int g_force_tile = 0, g_force_splitk = 0, g_kch = 0, g_dma_auto = 1, g_force_order = -1;
void gemm_debug_order(int o) { g_force_order = o; }
void gemm_debug_set(int tile, int splitk, int kch) { g_force_tile = tile; g_force_splitk = splitk; g_kch = kch; }
void gemm_debug_dma_auto(int on) { g_dma_auto = on; }
static int g_p8 = 1;
void gemm_debug_p8(int on) { g_p8 = on; }
static int g_wres = 1;
void gemm_debug_wres(int on) { g_wres = on; }
static thread_local int t_reduce_launched = 0;   // set by launch_t when a splitk_reduce_kernel launch followed the product
static thread_local GemmArgs* t_pending = nullptr;   // launch_gemm(..., pending): where a deferrable reduction is parked instead of launched

int gemm_uses_big_tile(int dtype, const GemmArgs& a) {
  // Register-staged kernel, measured on MI355X (tools/gpu_gemm_bench.py, profiles/r01_gemm_microbench.txt):
  // bf16 -- the 64x64 tile (7 blocks/CU in flight) beats the 128x128 tile, which is latency-bound at <= 4 blocks/CU
  //         (large bf16 problems go to the asynchronous ring kernels instead);
  // fp32 -- the slow fp32 MFMA (64 cycles) hides the load latency: with >= 4 tiles per CU the 128x128 tile wins
  //         (119 vs 93 TF/s on the 256x256-resolution DDPM convolutions), below that the 64x64 tile.
  if (g_force_tile) return g_force_tile == 128;
  if (dtype != DT_F32 || a.A2) return 0;
  const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.Z1 * a.Z2;
  return t128 >= 1024;
}

// Per-shape tuning overrides, for in-pipeline kernel selection experiments (tools/gpu_gemm_override.py):
//   DPB_GEMM_OVERRIDE="MxNxK:gather=code/split,..."   code as dpb_debug_set("gemm_tile"), split 0 = heuristic
struct ShapeOverride { int M, N, K, gather, code, split; };
static const std::vector<ShapeOverride>& shape_overrides() {
  static const std::vector<ShapeOverride> v = [] {
    std::vector<ShapeOverride> o;
    const char* e = getenv("DPB_GEMM_OVERRIDE");
    while (e && *e) {
      ShapeOverride r{};
      int used = 0;
      if (sscanf(e, "%dx%dx%d:%d=%d/%d%n", &r.M, &r.N, &r.K, &r.gather, &r.code, &r.split, &used) == 6) o.push_back(r);
      else break;
      e += used;
      if (*e == ',') ++e;
    }
    return o;
  }();
  return v;
}
static const ShapeOverride* find_override(const GemmArgs& a) {
  for (const auto& r : shape_overrides())
    if (r.M == a.M && r.N == a.N && r.K == a.K && r.gather == a.gather && a.Z1 * a.Z2 == 1) return &r;
  return nullptr;
}

// The weights-resident streaming kernel (gemm_wres.hip, tile code 540): plain products with K = 320 and N a multiple of 320 (the 320-channel linear layers
// of the 64 x 64 level, tangent / adjoint passes) from the row count at which streaming beats the tile kernels.  Measured against the round-5 dispatch per
// shape (profiles/r06_wres_shapes.txt, event brackets, same session): N = 320 -- 20480 rows 15.0 vs 11.8 us (the 200 KB weight preload per CU and three
// serial tiles are not amortised: the rings keep the 5-tangent pass), 40960 rows equal, 81920 rows 30 vs 35 us, 327680 rows 93 vs 156 us (4.5 TB/s of
// operand traffic, 6.7 with a row operand); N = 960 (three column slices re-stream A) -- ahead only at 327680 rows (273 vs 359 us).
static int wres_wants(int dtype, const GemmArgs& a) {
  static const int wres_env = getenv("DPB_WRES") ? atoi(getenv("DPB_WRES")) : 1;              // tuning switch (0: ring / 8-phase tiles as in round 5)
  static const int min_m = getenv("DPB_WRES_MIN_M") ? atoi(getenv("DPB_WRES_MIN_M")) : 49152; // tuning switch: rows from which the N = 320 products take it
  if (!wres_env || !g_wres || !gemm_wres_supported(dtype, a)) return 0;
  return a.M >= (a.N == 320 ? (long)min_m : 3L * min_m);
}

// The 8-phase 256 x 256 tile (gemm_p8.hip) takes a product when its tiles fill the chip (one 8-wave block per CU): >= 160 tiles; at most 25 % of the
// tile area padding (N = 320 is 37.5 %: stays on the 128-column rings / the halo kernel -- except plain rows with >= 2048 tiles, where the tile still
// wins by 3-10 %); the last round of 256 tiles at least 58 % occupied (or >= 4 rounds); K >= 640 (a K = 320 product is five K tiles: prologue and
// epilogue dominate -- 62 vs 56 us on 20480 x 2560 x 320 -- until >= 4096 tiles amortise them: 850 vs 922 us on 327680 x 2560 x 320).
// Measured per shape against the round-4 dispatch at 5 / 20 / 80 tangents: profiles/r05_p8_shapes.txt (plain rows +5-25 %; 3x3 convolutions against the
// halo-tile kernel: N = 1280 +16-19 %, N = 640 +10 % at 20 tangents and equal at 80).
static int p8_wants(int dtype, const GemmArgs& a) {
  static const int p8_env = getenv("DPB_P8") ? atoi(getenv("DPB_P8")) : 1;      // tuning switch (0: rings / halo kernel as in round 4)
  if (!p8_env || !g_p8 || dtype == DT_F32 || a.A2 || !a.zeros || a.Z1 * a.Z2 != 1 || a.K % 8 || !gemm_p8_fits32(a)) return 0;
  if (a.gather != GATHER_NONE && (a.Cin % 64 || a.epi != EPI_PLAIN)) return 0;
  if (a.epi == EPI_LN_TAN || a.epi == EPI_LN_ADJ || (a.epi != EPI_PLAIN && a.N % 256)) return 0;
  const long t256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256), rounds = (t256 + 255) / 256;
  if (t256 < 160 || (a.K < 640 && !(a.K >= 320 && t256 >= 4096))) return 0;
  const double fill = (double)a.M * a.N / ((double)t256 * 65536.0);
  if (fill < ((t256 >= 2048 && a.gather == GATHER_NONE) ? 0.6 : 0.75)) return 0;
  // N = 640 convolutions (17 % padding): ahead of the halo-tile kernel at 20 tangents (157 vs 173 us), behind it at 80 -- inside the configs[3] pass
  // 686 / 736 us against 644 / 692 (profiles/r05_roofline_sd15_mid_k10x8_bf16.md, first run): the halo kernel keeps them from 512 tiles on
  if (a.gather != GATHER_NONE && fill < 0.9 && t256 > 512) return 0;
  return t256 >= 1024 || (double)t256 >= 0.58 * 256.0 * (double)rounds;
}

// the halo-tile 3x3 convolution (gemm_halo.hip): every supported shape of the path measured faster than the implicit-GEMM rings
int gemm_uses_halo(int dtype, const GemmArgs& a) {
  static const int halo_env = getenv("DPB_CONV_HALO") ? atoi(getenv("DPB_CONV_HALO")) : 1;   // tuning switch (0: implicit-GEMM rings)
  const ShapeOverride* ov = g_force_tile ? nullptr : find_override(a);
  const int force = ov ? ov->code : g_force_tile;
  const bool want = force == 600 || (halo_env && force == 0 && g_dma_auto && !p8_wants(dtype, a));
  // (8x8 images, four per tile, are supported but measure no better than the split-K ring: forced only)
  return want && dtype != DT_F32 && conv_halo_supported(a) && (a.H * a.W >= 256 || force == 600);
}

// the asynchronous LDS-ring kernel (gemm_dma.hip): bf16, one operand pair, enough 128x128 tiles to fill the chip
int gemm_uses_dma(int dtype, const GemmArgs& a) {
  const ShapeOverride* ov = g_force_tile ? nullptr : find_override(a);
  const int force = ov ? ov->code : g_force_tile;
  if (dtype == DT_F32 || a.A2 || !a.zeros || force == 64 || force == 128) return 0;
  if (force == 129) return 128;
  if (force == 131) return 130;
  if (force == 133) return 132;
  if (force == 257) return 256;
  if (force == 65) return 64;
  if (force == 67) return 66;
  if (force >= 512 && force <= 517) return force;
  if (force == 521 && a.epi == EPI_PLAIN && a.gather != GATHER_UPCONV) return 521;     // (also as an implicit-GEMM convolution)
  if (force >= 521 && force <= 523) return (a.gather == GATHER_NONE && a.epi == EPI_PLAIN) ? force : 515;
  if (force == 518) return a.gather == GATHER_NONE ? 518 : 515;
  if (force == 530) return ((a.epi == EPI_PLAIN || a.gather == GATHER_NONE) && (a.gather == GATHER_NONE || a.Cin % 64 == 0)) ? 530 : 515;
  if (force == 540) return gemm_wres_supported(dtype, a) ? 540 : 515;
  const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.Z1 * a.Z2;
  const long t64 = (long)((a.M + 63) / 64) * ((a.N + 63) / 64) * a.Z1 * a.Z2;
  // measured on the path's layer shapes (profiles/r01_gemm_microbench.txt): the 128x128 ring wins once every CU holds
  // >= ~2 tiles, and for long-K under-filled problems when combined with split-K; short-K mid-size problems go to the
  // 64x64 ring; everything else (tiny problems, fp32, dual-operand products) to the register-staged kernel.
  if (!g_dma_auto || a.K < 256) return 0;
  if (wres_wants(dtype, a)) return 540;
  if (p8_wants(dtype, a)) return 530;
  // 256x256 8-wave tile (half the L2->LDS bytes per flop): plain-row products that give >= 160 such tiles with < 7 % padding and K >= 640
  // -- 12-26 % ahead of the 128x128 ring there, behind it below (profiles/r02_gemm_big_microbench.txt, r02_gemm_split_microbench.txt)
  static const int big_env = getenv("DPB_TILE256") ? atoi(getenv("DPB_TILE256")) : 1;   // tuning switch
  if (big_env && a.gather == GATHER_NONE && a.K >= 640 && a.Z1 * a.Z2 == 1) {
    const long t256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    if (t256 >= 160 && (double)t256 * 65536.0 <= 1.07 * (double)a.M * a.N) return 518;
  }
  // Half tiles for launches that give every CU at most ONE 128x128 block (128 <= tiles < 256: the plain-row products of the 32x32 level at k = 5,
  // 5120x640xK = 200 tiles): a lone 4-wave block has no co-resident partner to cover its barrier / fragment latency (43 us for 5120x640x2560
  // against 8.6 us of MFMA time per tile); 64x128 tiles with a 3-stage ring (72 KiB) put two blocks on most CUs -- inside the pass 13.4 -> 11.2 us
  // (K = 640, twenty per iteration), 30.8 -> 23.4 (K = 1920), 40.4 -> 32.8 (K = 2560); K = 5120 keeps the two-fold split of the 128x128 tile
  // (46.5 vs 56.3 us).  128x64 tiles measure the same, the 2-stage 64x128 ring (three blocks per CU) less (profiles/r04_gemm_override_half_tiles.txt).
  static const int half_env = getenv("DPB_HALF_TILE") ? atoi(getenv("DPB_HALF_TILE")) : 3;   // tuning switch (bit 0 / bit 1: the two rules below)
  static const int half_kmin = getenv("DPB_HALF_KMIN") ? atoi(getenv("DPB_HALF_KMIN")) : 256;   // tuning switch (256: the K = 320 products of a batch-2 forward at 64x64 too: forward -0.3 %, iteration neutral)
  if ((half_env & 1) && a.gather == GATHER_NONE && a.epi == EPI_PLAIN && a.Z1 * a.Z2 == 1 && a.K >= half_kmin && a.K <= 4096 && t128 >= 128 && t128 < 256) return 521;
  // ... and for long-K products whose last 128-row tile is at most half full (M = 320 = 64 k rows of the 8x8 level at k = 5: 2.5 tiles, 17 % padded
  // MFMAs): 64-row tiles cover M exactly, give 50 instead of 30 tiles, and the split-K plan needs 9 instead of 15 fp32 slabs for its ~450 blocks --
  // the consumers (one-launch GroupNorm, LayerNorm) gather 40 % fewer slab bytes: 8.577 -> 8.52 ms per iteration inside the pass on the seventeen
  // 8x8-level convolutions (22.4 -> 21.7 us each; 15 or 6 splits on the same tile: 25-27 us)
  if ((half_env & 2) && a.epi == EPI_PLAIN && a.gather != GATHER_UPCONV && a.Z1 * a.Z2 == 1 && a.K >= 2048 && t128 < 128 && (a.M % 128) && (a.M % 128) <= 64) return 521;
  // BK = 64 ring (gemm_ring64.hip: whole-line DMA + in-wave fragment prefetch, 128x128 tile, 2 stages -> 2 blocks/CU):
  // ahead of the BK = 32 rings by 10-35 % from ~8 stages of K on, with split-K when the tiles leave CUs idle
  if (a.K >= 512 && (t128 >= 200 || a.K >= 2048)) return 515;
  // short K (320 on the 64x64 level) with the chip filled by 128x128 tiles: the BK=64 ring again -- 11.5 vs 12.9 us on 20480x320x320,
  // 27.9 vs 30.4 us at N = 960, equal at N >= 1280 (profiles/r02_gemm_shortk_microbench.txt); the BK=32 ring (130) remains for K % 64 != 0
  // -- up to ~1300 tiles; beyond (several samples advanced together: 81920 rows) the BK=32 ring's three blocks per CU hide the residual /
  // row-bias loads of the epilogue better (43 vs 45 us on 81920x320x320 inside the pass)
  if (t128 >= 400) return (a.K % 64 == 0 && t128 <= 1280) ? 515 : 130;
  if (a.K >= 2048 && t128 >= 64) return 256;    // long K, under-filled: 256x128 ring + split-K (fewest operand re-reads)
  // 64x64 ring, unsplit: also for ~100-250 tiles at K >= 1024 -- inside the pass 11.6 vs 13.8 us (320x1280x1280, ten per iteration) and 11.0 vs
  // 17.2 us (1280x640x1280) against the register-staged kernel with split-K 5 + reduce (profiles/r02_gemm_override_in_pipeline.txt)
  if (t64 >= 256 || (t64 >= 96 && a.K >= 1024)) return 64;
  return 0;
}

// fused epilogues (epilogue.h) live in the ring kernels with 128-column tiles and need the whole K range in one block
int gemm_epi_supported(int dtype, const GemmArgs& a) {
  if (a.epi == EPI_LN_TAN || a.epi == EPI_LN_ADJ)                // row-complete 128 x 320 tile (gemm_ring64.hip, tile code 520)
    return dtype != DT_F32 && a.Z1 * a.Z2 == 1 && a.gather == GATHER_NONE && a.N == 320 && a.ldc == 320 && a.M >= 128 && a.K % 64 == 0 && a.zeros &&
           a.ln_x && a.ln_gamma && !a.A2 && !a.bias && !a.rowbias && a.alpha == 1.f && (a.epi == EPI_LN_ADJ || (a.C2 && !a.accumulate)) && (!a.R || a.ldr % 8 == 0);
  if (dtype == DT_F32 || a.Z1 * a.Z2 != 1 || a.gather != GATHER_NONE || a.N % 128 || a.M <= 0) return 0;
  if (a.epi == EPI_GEGLU_ADJ && a.N % 64) return 0;
  const int dt = gemm_uses_dma(dtype, a);
  if (dt == 518 || dt >= 530) return a.N % 256 == 0;
  return dt == 128 || dt == 130 || dt == 132 || dt == 256 || (dt >= 512 && dt <= 517);
}

// split-K for the ring kernels: long-K problems that leave CUs idle (weights then stream from HBM once, in parallel)
int gemm_pick_splitk_dma(const GemmArgs& a, int tile) {
  if (!a.slab) return 1;
  const int T = (tile == 518 || tile >= 530) ? 256 : (tile == 128 || tile == 130 || tile == 132 || tile == 256 || (tile >= 512 && tile <= 517) || tile == 521 || tile == 522) ? 128 : 64;
  const int TMm = (tile == 256 || tile == 513 || tile == 516 || tile == 517 || tile == 518 || tile >= 530) ? 256 : tile == 523 ? 128 : (tile == 521 || tile == 522) ? 64 : T;
  const long tiles = (long)((a.M + TMm - 1) / TMm) * ((a.N + T - 1) / T) * a.Z1 * a.Z2;
  const int nk = (a.K + 31) / 32;
  long s;
  const ShapeOverride* ov = g_force_splitk ? nullptr : find_override(a);
  if (g_force_splitk) s = g_force_splitk;
  else if (ov && ov->split) s = ov->split;
  else if (tile >= 512) {                     // 2 resident blocks per CU: aim at ~450 blocks, >= 8 stages of 64 each
    // (512 until the deferred reductions moved the slab gathers into the consumers: 448 / 384 measure 0.5 % ahead of 512 there, 320 / 256 behind;
    // 768: +4 % -- profiles/r03_ab_sessions.txt)
    static const long target = getenv("DPB_SPLITK_TARGET") ? atol(getenv("DPB_SPLITK_TARGET")) : 448;   // tuning switch
    // >= 192 tiles (3/4 of the CUs hold a block): splitting only pays for K >= 4096 and only two-fold -- measured per shape in
    // profiles/r02_gemm_split_microbench.txt (5120x640: K 1920 / 2560 24 / 32 us unsplit vs 34 / 41 us three-fold, K 5120 52 us two-fold vs
    // 58 unsplit; 1280x3840x1280 25 vs 36 us; 320x10240x1280 18 vs 25 us)
    if (tile == 518 || tile == 530 || tile == 540) return 1;  // one block per CU by construction: never split
    if (tiles >= 192) s = (tiles < 256 && nk >= 128) ? 2 : 1;
    else {
      s = std::max<long>(1, (target + tiles / 2) / tiles);
      s = std::min<long>(s, std::max(1, nk / 16));
    }
  } else {
    if (tiles >= 256 || nk < 64) return 1;      // only when CUs would idle and K is long enough to amortise the slabs
    s = (1024 + tiles - 1) / tiles;
    s = std::min<long>(s, nk / 12);             // keep >= 12 K steps per block (ring depth 3)
    s = std::min<long>(s, 32);
  }
  s = std::min<long>(s, nk);
  const long per = (long)a.M * a.N * a.Z1 * a.Z2 * 4;
  if (s * per > (long)a.slab_bytes) s = (long)a.slab_bytes / per;
  return (int)std::max<long>(s, 1);
}

int gemm_kch(const GemmArgs& a) {
  if (g_kch) return g_kch;
  return 4;
}

// number of K splits for under-filled launches (1 = none)
int gemm_pick_splitk(int dtype, const GemmArgs& a) {
  if (a.A2 || !a.slab) return 1;
  const int BK = (dtype == DT_F32 ? 4 : 8) * gemm_kch(a);
  const int T = gemm_uses_big_tile(dtype, a) ? 128 : 64;
  const long tiles = (long)((a.M + T - 1) / T) * ((a.N + T - 1) / T) * a.Z1 * a.Z2;
  const int nk = (a.K + BK - 1) / BK;
  long s;
  const ShapeOverride* ov = g_force_splitk ? nullptr : find_override(a);
  if (g_force_splitk) {
    s = g_force_splitk;
  } else if (ov && ov->split) {
    s = ov->split;
  } else {
    if (T == 128 || tiles >= 768 || nk < 32) return 1;
    s = (1024 + tiles - 1) / tiles;             // aim at ~4 blocks per CU
    s = std::min<long>(s, nk / 8);              // keep >= 8 K steps per block
    s = std::min<long>(s, 32);
  }
  s = std::min<long>(s, nk);
  const long per = (long)a.M * a.N * a.Z1 * a.Z2 * 4;
  if (s * per > (long)a.slab_bytes) s = (long)a.slab_bytes / per;
  return (int)std::max<long>(s, 1);
}

// ---- the launch plan of one product: which kernel, which tile, how many K splits.  A pure host function (dpb_debug_gemm_plan exposes it, so
// the dispatch rules are testable without a GPU); EVERY path's split count passes the slab-capacity clamp here, in one place.
enum { PLAN_REG64 = 0, PLAN_REG128 = 1, PLAN_RING = 2, PLAN_HALO = 3 };
GemmPlan gemm_plan(int dtype, const GemmArgs& a) {
  GemmPlan pl{PLAN_REG64, 0, 1};
  long s = 1;
  if (gemm_uses_halo(dtype, a)) {
    // 3x3 stride-1 convolutions: halo-tile kernel (gemm_halo.hip), one 256x128 tile per block, K split over 64-channel chunks
    pl.kind = PLAN_HALO; pl.tile = 600;
    const long tiles = (long)((a.M + 255) / 256) * ((a.N + 127) / 128);
    const int nch = a.Cin / 64;
    const ShapeOverride* ov = g_force_splitk ? nullptr : find_override(a);
    if (g_force_splitk) s = g_force_splitk;
    else if (ov && ov->split) s = ov->split;
    else if (a.slab) {
      // one resident block per CU: time ~ rounds x (chunks per block + ~2 chunks of prologue / epilogue); measured optimum on
      // the path's layers (profiles/r01_gemm_microbench.txt): 64^2 -> 1, 32^2 -> 2, 16^2 -> 5 splits
      long best = 1L << 60;
      for (long c = 1; c <= nch; ++c) {
        const long cost = ((tiles * c + 255) / 256) * ((nch + c - 1) / c + 2);
        if (cost < best) { best = cost; s = c; }
      }
    }
    s = std::min<long>(s, nch);
  } else {
    if (a.epi != EPI_PLAIN && !gemm_epi_supported(dtype, a)) {
      set_error("gemm: fused epilogue %d requested for a launch no ring kernel with 128-column tiles takes (M=%d N=%d K=%d)", a.epi, a.M, a.N, a.K);
      pl.kind = -1;
      return pl;
    }
    if (a.epi == EPI_LN_TAN || a.epi == EPI_LN_ADJ) {
      pl.kind = PLAN_RING; pl.tile = 520; s = 1;
    } else if (const int dt = gemm_uses_dma(dtype, a)) {
      pl.kind = PLAN_RING; pl.tile = dt;
      s = (a.epi != EPI_PLAIN || dt == 540) ? 1 : gemm_pick_splitk_dma(a, dt);      // (the weights-resident kernel never splits K, whatever a debug override asks)
    } else {
      pl.kind = gemm_uses_big_tile(dtype, a) ? PLAN_REG128 : PLAN_REG64;
      pl.tile = pl.kind == PLAN_REG128 ? 128 : 64;
      s = gemm_pick_splitk(dtype, a);
    }
  }
  const long per = (long)a.M * a.N * a.Z1 * a.Z2 * 4;          // one fp32 slab
  if (s > 1 && (!a.slab || s * per > (long)a.slab_bytes)) s = a.slab ? (long)a.slab_bytes / per : 1;
  pl.splitk = (int)std::max<long>(s, 1);
  return pl;
}

template <typename T, int BM, int BN, int KCH>
static void launch_reg_t(const GemmArgs& a, dim3 grid, hipStream_t st) {
  switch (a.gather) {
    case GATHER_NONE: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KCH, GATHER_NONE>), grid, dim3(256), 0, st, a); break;
    case GATHER_CONV: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KCH, GATHER_CONV>), grid, dim3(256), 0, st, a); break;
    case GATHER_CONVT: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KCH, GATHER_CONVT>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, KCH, GATHER_UPCONV>), grid, dim3(256), 0, st, a); break;
  }
}

template <typename T>
static int launch_t(int dtype, GemmArgs a, hipStream_t st) {
  constexpr int CH = TT<T>::CH;
  if (a.K % CH || a.lda % CH || a.ldb % CH || (a.gather != GATHER_NONE && a.Cin % CH)) {
    set_error("gemm: K=%d lda=%d ldb=%d Cin=%d must be multiples of %d", a.K, a.lda, a.ldb, a.Cin, CH);
    return -1;
  }
  if (a.A2 && (a.K2 % CH || a.lda2 % CH || a.ldb2 % CH || a.gather != GATHER_NONE)) {
    set_error("gemm: second operand pair misaligned or combined with a gather");
    return -1;
  }
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) { set_error("gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K); return -1; }
  a.vec_ok = !(a.ldc & 7) && !(a.sC1 & 7) && !(a.sC2 & 7) && (!a.R || (!(a.ldr & 7) && !(a.sR1 & 7) && !(a.sR2 & 7))) &&
             (!a.rowbias || !(a.N & 7)) && !((uintptr_t)a.C & 15) && !((uintptr_t)a.R & 15) && !((uintptr_t)a.bias & 15);
  const int Z = a.Z1 * a.Z2;
  {  // unique operand bytes per batch entry: the larger operand should be the one each XCD sees only 1/8 of
    const double ua = (double)a.M * (a.gather == GATHER_NONE ? a.K : a.Cin), ub = (double)a.N * a.K;
    static const int env_order = getenv("DPB_GEMM_ORDER") ? atoi(getenv("DPB_GEMM_ORDER")) : -1;   // tuning override
    const int force = g_force_order >= 0 ? g_force_order : env_order;
    a.order = force >= 0 ? force : (ub > ua ? 1 : 0);
  }
  const GemmPlan pl = gemm_plan(dtype, a);
  if (pl.kind < 0) return -1;
  a.splitk = pl.splitk;
  if (pl.kind == PLAN_HALO) {
    if (int r = launch_conv_halo(a, st)) return r;
  } else if (pl.kind == PLAN_RING) {
    if (int r = pl.tile == 540 ? launch_gemm_wres(a, st) : pl.tile >= 530 ? launch_gemm_p8(a, pl.tile, st) : pl.tile >= 512 ? launch_gemm_ring64(a, pl.tile, st) : launch_gemm_dma(a, pl.tile, st)) return r;
  } else if (pl.kind == PLAN_REG128) {
    dim3 grid(((a.M + 127) / 128) * ((a.N + 127) / 128), Z, a.splitk);
    launch_reg_t<T, 128, 128, 4>(a, grid, st);
  } else {
    dim3 grid(((a.M + 63) / 64) * ((a.N + 63) / 64), Z, a.splitk);
    if (gemm_kch(a) == 8) launch_reg_t<T, 64, 64, 8>(a, grid, st);
    else launch_reg_t<T, 64, 64, 4>(a, grid, st);
  }
  if (a.splitk > 1) {
    const bool plain = Z == 1 && a.alpha == 1.f && !a.bias && !a.rowbias && !a.accumulate && a.ldc == a.N && a.vec_ok && !(a.N & 7) && a.epi == EPI_PLAIN;
    if (t_pending && plain) {
      *t_pending = a;                               // the consumer (or launch_gemm_reduce) adds the slabs
    } else {
      long total = (long)a.M * a.N * Z / 4;
      unsigned g = (unsigned)std::max<long>(1, std::min<long>((total + 255) / 256, 4096));
      hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(g), dim3(256), 0, st, a);
      t_reduce_launched = 1;
    }
  }
  DPB_CHECK(hipGetLastError());
  return 0;
}

int launch_gemm_reduce(int dtype, const GemmArgs& a, hipStream_t st) {
  if (a.splitk <= 1) return 0;
  const long total = (long)a.M * a.N * a.Z1 * a.Z2 / 4;
  const unsigned g = (unsigned)std::max<long>(1, std::min<long>((total + 255) / 256, 4096));
  DPB_DISPATCH_STMT(dtype, T, hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(g), dim3(256), 0, st, a));
  DPB_CHECK(hipGetLastError());
  return 0;
}

int launch_gemm(int dtype, const GemmArgs& a, hipStream_t st, int* launches, GemmArgs* pending) {
  GemmArgs b = a;
  b.fl = dtype == DT_F16;           // 16-bit flavour of the specialised kernels (H16<fl>)
  t_reduce_launched = 0;
  if (pending) pending->splitk = 1;
  t_pending = pending;
  static const int trace = getenv("DPB_GEMM_TRACE") ? atoi(getenv("DPB_GEMM_TRACE")) : 0;   // debugging: print every product, synchronise after it
  if (trace) {
    fprintf(stderr, "gemm M=%d N=%d K=%d Z=%d gather=%d epi=%d lda=%d ldb=%d ldc=%d ldr=%d acc=%d R=%d bias=%d rowbias=%d kind=%d\n", b.M, b.N, b.K, b.Z1 * b.Z2, b.gather,
            b.epi, b.lda, b.ldb, b.ldc, b.ldr, b.accumulate, b.R != nullptr, b.bias != nullptr, b.rowbias != nullptr, gemm_uses_dma(dtype, b));
    fflush(stderr);
  }
  const int r = DPB_DISPATCH_T(dtype, T, launch_t<T>(dtype, b, st));
  if (trace && hipStreamSynchronize(st) != hipSuccess) fprintf(stderr, "gemm: the launch above failed\n");
  t_pending = nullptr;
  if (launches) *launches = 1 + t_reduce_launched;
  return r;
}

}  // namespace dpb

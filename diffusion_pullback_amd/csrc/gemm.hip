// MFMA GEMM / implicit-GEMM convolution for gfx950 (wave64).
//
//   C[z][m][n] = alpha * sum_k A[z][m][k] * B[z][n][k]  (+bias) (+rowbias) (+R) (+C)
//
// One kernel serves every dense contraction of the pullback path: 3x3 / strided /
// transposed / upsampling convolutions (A rows gathered from NHWC pixels), 1x1
// convolutions and Linear layers (one tap), and the batched Q K^T / P V products of
// attention (two-level batch z = (tangent, head)).  Linear maps have identical
// primal, tangent and (with the pre-transposed weight) adjoint kernels, so the
// JVP batch and the VJP batch of the power iteration are both just larger M.
//
// Tiling: 256 threads = 4 waves (2x2); each wave owns (BM/2)x(BN/2) as 32x32 MFMA
// tiles.  bf16: v_mfma_f32_32x32x16_bf16, LDS tiles row-major [rows][32+8] so a
// fragment is one ds_read_b128; f32: v_mfma_f32_32x32x2_f32, LDS tiles k-major
// [16][rows+4] so a fragment is one conflict-free ds_read_b32.  K advances 4
// 16-byte chunks per step; global->register prefetch of step t+1 overlaps the MFMAs
// of step t, LDS is double buffered (one barrier per step).
#include "kernels.h"

namespace dpb {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
  constexpr int CH = TT<T>::CH;
  constexpr int BK = 4 * CH;
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int LDA_S = F32 ? (BM + 4) : (BK + 8);
  constexpr int LDB_S = F32 ? (BN + 4) : (BK + 8);
  constexpr int A_ELEMS = F32 ? BK * LDA_S : BM * LDA_S;
  constexpr int B_ELEMS = F32 ? BK * LDB_S : BN * LDB_S;
  constexpr int NA = BM / 64, NB = BN / 64;
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  __shared__ __attribute__((aligned(16))) T smem[2 * (A_ELEMS + B_ELEMS)];
  T* As = smem;
  T* Bs = smem + 2 * A_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tilesN = (p.N + BN - 1) / BN;
  const int tm_i = blockIdx.x / tilesN, tn_i = blockIdx.x % tilesN;
  const int m0 = tm_i * BM, n0 = tn_i * BN;
  const int z1 = blockIdx.y / p.Z2, z2 = blockIdx.y % p.Z2;

  const T* A = (const T*)p.A + (long)(z1 / p.divA) * p.sA1 + (long)z2 * p.sA2;
  const T* B = (const T*)p.B + (long)(z1 / p.divB) * p.sB1 + (long)z2 * p.sB2;
  T* C = (T*)p.C + (long)z1 * p.sC1 + (long)z2 * p.sC2;
  const T* R = p.R ? (const T*)p.R + (long)z1 * p.sR1 + (long)z2 * p.sR2 : nullptr;

  const int kq = tid & 3;
  // ---- per-thread A rows
  const T* a_base[NA];
  int a_oy[NA], a_ox[NA];
  bool a_ok[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    int row = (tid + i * 256) >> 2;
    int m = m0 + row;
    a_ok[i] = m < p.M;
    a_oy[i] = a_ox[i] = 0;
    if (p.gather == GATHER_NONE) {
      a_base[i] = A + (long)m * p.lda;
    } else {
      int hw = p.Ho * p.Wo;
      int smp = m / hw, rem = m - smp * hw;
      a_oy[i] = rem / p.Wo;
      a_ox[i] = rem - a_oy[i] * p.Wo;
      a_base[i] = A + (long)smp * p.H * p.W * p.lda;
    }
  }
  const T* b_base[NB];
  bool b_ok[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    int n = n0 + ((tid + i * 256) >> 2);
    b_ok[i] = n < p.N;
    b_base[i] = B + (long)n * p.ldb;
  }
  // running (tap, channel) of this thread's chunk column
  int kc = kq * CH, tap = 0, cc = kc;
  if (p.gather != GATHER_NONE) {
    tap = kc / p.Cin;
    cc = kc - tap * p.Cin;
  }

  uint4 ra[NA], rb[NB];
  auto gload = [&]() {
    const bool kok = kc < p.K;
    int ky = 0, kx = 0;
    if (p.gather != GATHER_NONE && p.KS == 3) {
      ky = (tap * 11) >> 5;
      kx = tap - ky * 3;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const T* ptr = nullptr;
      bool ok = a_ok[i] && kok;
      if (p.gather == GATHER_NONE) {
        ptr = a_base[i] + kc;
      } else {
        int iy, ix;
        if (p.gather == GATHER_CONV) {
          iy = a_oy[i] * p.stride + ky - p.pad;
          ix = a_ox[i] * p.stride + kx - p.pad;
          ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        } else if (p.gather == GATHER_CONVT) {
          int ty = a_oy[i] + p.pad - ky, tx = a_ox[i] + p.pad - kx;
          ok = ok && ty >= 0 && tx >= 0;
          if (p.stride == 2) {
            ok = ok && !((ty | tx) & 1);
            iy = ty >> 1; ix = tx >> 1;
          } else {
            iy = ty; ix = tx;
          }
          ok = ok && iy < p.H && ix < p.W;
        } else {   // GATHER_UPCONV: nearest x2 then 3x3 pad 1
          int uy = a_oy[i] + ky - 1, ux = a_ox[i] + kx - 1;
          ok = ok && uy >= 0 && ux >= 0 && uy < 2 * p.H && ux < 2 * p.W;
          iy = uy >> 1; ix = ux >> 1;
        }
        ptr = a_base[i] + ((long)iy * p.W + ix) * p.lda + cc;
      }
      ra[i] = ok ? *reinterpret_cast<const uint4*>(ptr) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      bool ok = b_ok[i] && kok;
      rb[i] = ok ? *reinterpret_cast<const uint4*>(b_base[i] + kc) : make_uint4(0, 0, 0, 0);
    }
    // advance to the next K step
    kc += BK;
    if (p.gather != GATHER_NONE) {
      cc += BK;
      while (cc >= p.Cin) { cc -= p.Cin; ++tap; }
    }
  };
  auto sstore = [&](int buf) {
    T* as = As + buf * A_ELEMS;
    T* bs = Bs + buf * B_ELEMS;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int row = (tid + i * 256) >> 2;
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
      int row = (tid + i * 256) >> 2;
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
  const int nk = (p.K + BK - 1) / BK;

  gload();
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload();
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
      for (int kk = 0; kk < 2; ++kk) {
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
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wx * WN + j * 32 + l31;
      if (n >= p.N) continue;
      const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wy * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (m >= p.M) continue;
        float v = p.alpha * acc[i][j][r] + bv;
        if (p.rowbias) {
          int smp = (m / p.rows_per_sample) / p.rowbias_div;
          v += TT<T>::ld((const T*)p.rowbias + (long)smp * p.N + n);
        }
        if (R) v += TT<T>::ld(R + (long)m * p.ldr + n);
        T* cp = C + (long)m * p.ldc + n;
        if (p.accumulate) v += TT<T>::ld(cp);
        TT<T>::st(cp, v);
      }
    }
  }
}

int gemm_uses_big_tile(const GemmArgs& a) {
  long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.Z1 * a.Z2;
  return t128 >= 192 && a.N > 64;
}

template <typename T>
static int launch_t(const GemmArgs& a, hipStream_t st) {
  constexpr int CH = TT<T>::CH;
  if (a.K % CH || a.lda % CH || a.ldb % CH || (a.gather != GATHER_NONE && a.Cin % CH)) {
    set_error("gemm: K=%d lda=%d ldb=%d Cin=%d must be multiples of %d", a.K, a.lda, a.ldb, a.Cin, CH);
    return -1;
  }
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) { set_error("gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K); return -1; }
  const int Z = a.Z1 * a.Z2;
  if (gemm_uses_big_tile(a)) {
    dim3 grid(((a.M + 127) / 128) * ((a.N + 127) / 128), Z);
    hipLaunchKernelGGL((gemm_kernel<T, 128, 128>), grid, dim3(256), 0, st, a);
  } else {
    dim3 grid(((a.M + 63) / 64) * ((a.N + 63) / 64), Z);
    hipLaunchKernelGGL((gemm_kernel<T, 64, 64>), grid, dim3(256), 0, st, a);
  }
  DPB_CHECK(hipGetLastError());
  return 0;
}

int launch_gemm(int dtype, const GemmArgs& a, hipStream_t st) {
  return dtype == DT_F32 ? launch_t<float>(a, st) : launch_t<bf16>(a, st);
}

}  // namespace dpb

// Softmax pieces of the materialised attention path (primal / tangent / adjoint) and the small batched transpose.
//
// The contractions (Q K^T, P V, and their tangent / adjoint forms) run on the MFMA GEMM kernel
// (gemm.hip); these kernels are the HBM-bound row passes between them.  One wave owns one score
// row, held in registers (16-byte chunks, lane-strided => coalesced).
//   forward : P = softmax(S)                         (scale folded into the GEMM's alpha)
//   tangent : dP = P o (dS - <P, dS>)                (also the adjoint: gS = P o (gP - <P, gP>))
//   adjT    : gS^T[j][i] = P^T[j][i] * (gP^T[j][i] - D[i])   with D = <P_i, gP_i> from the row pass
#include "kernels.h"

namespace dpb {

template <typename T, int NCH>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(T* S, long nrows, int Lk_all, int ld, int causal_Lq) {
  constexpr int CH = TT<T>::CH;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  // causal (text-encoder) attention: query i = row % Lq sees keys 0..i; the masked probabilities are stored as exact zeros
  const int Lk = causal_Lq ? min(Lk_all, (int)(row % causal_Lq) + 1) : Lk_all;
  T* sp = S + row * ld;
  const int nch = ld / CH;
  float x[NCH][CH];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = lane + i * 64;
    if (c < nch) {
      Vec<T>::load(sp + c * CH, x[i]);
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        if (c * CH + e >= Lk) x[i][e] = -INFINITY;
        m = fmaxf(m, x[i][e]);
      }
    }
  }
  m = wave_max(m);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = lane + i * 64;
    if (c < nch) {
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        x[i][e] = __expf(x[i][e] - m);
        s += x[i][e];
      }
    }
  }
  const float inv = 1.f / wave_sum(s);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = lane + i * 64;
    if (c < nch) {
#pragma unroll
      for (int e = 0; e < CH; ++e) x[i][e] *= inv;
      Vec<T>::store(sp + c * CH, x[i]);
    }
  }
}

template <typename T, int NCH>
__global__ __launch_bounds__(256) void softmax_jvp_kernel(const T* P, T* dS, float* D, long nrows, int Z2, int kps, int Lq,
                                                          int Lk, int ld) {
  constexpr int CH = TT<T>::CH;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const long z = row / Lq;
  const int i_q = (int)(row - z * Lq);
  const long zp = ((z / Z2) / kps) * Z2 + z % Z2;
  const T* pp = P + (zp * Lq + i_q) * ld;
  T* dp = dS + row * ld;
  const int nch = ld / CH;
  float p[NCH][CH], d[NCH][CH];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = lane + i * 64;
    if (c < nch) {
      Vec<T>::load(pp + c * CH, p[i]);
      Vec<T>::load(dp + c * CH, d[i]);
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        if (c * CH + e >= Lk) { p[i][e] = 0.f; d[i][e] = 0.f; }   // padded columns may hold stale scratch
        dot += p[i][e] * d[i][e];
      }
    }
  }
  dot = wave_sum(dot);
  if (D && lane == 0) D[row] = dot;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = lane + i * 64;
    if (c < nch) {
#pragma unroll
      for (int e = 0; e < CH; ++e) d[i][e] = p[i][e] * (d[i][e] - dot);
      Vec<T>::store(dp + c * CH, d[i]);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_adjT_kernel(const T* PT, T* gPT, const float* D, long Z, int Z2, int kps, int Lk,
                                                           int Lq, int ld) {
  constexpr int CH = TT<T>::CH;
  const int nch = ld / CH;
  const long total = Z * Lk * nch;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % nch);
    const long rowj = idx / nch;           // z * Lk + j
    const long z = rowj / Lk;
    const int j = (int)(rowj - z * Lk);
    const long zp = ((z / Z2) / kps) * Z2 + z % Z2;
    float p[CH], g[CH];
    Vec<T>::load(PT + (zp * Lk + j) * ld + c * CH, p);
    T* gp = gPT + rowj * ld + c * CH;
    Vec<T>::load(gp, g);
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      int i = c * CH + e;
      g[e] = i < Lq ? p[e] * (g[e] - D[z * Lq + i]) : 0.f;
    }
    Vec<T>::store(gp, g);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* in, T* out, int Z2, long s1, long s2, int R, int Ccols, int ldin,
                                                        int ldout, long outZ) {
  __shared__ float tile[32][33];
  const int z = blockIdx.z, z1 = z / Z2, z2 = z % Z2;
  const T* ip = in + z1 * s1 + z2 * s2;
  T* op = out + (long)z * outZ;
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    int r = r0 + k, c = c0 + tx;
    tile[k][tx] = (r < R && c < Ccols) ? TT<T>::ld(ip + (long)r * ldin + c) : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    int c = c0 + k, r = r0 + tx;
    if (c < Ccols && r < ldout) TT<T>::st(op + (long)c * ldout + r, r < R ? tile[tx][k] : 0.f);   // zero the K padding
  }
}

template <typename T>
static int pick_fwd(T* S, long nrows, int Lk, int ld, int causal_Lq, hipStream_t st) {
  constexpr int CH = TT<T>::CH;
  int need = (ld / CH + 63) / 64;
  dim3 grid((unsigned)((nrows + 3) / 4)), blk(256);
  if (need <= 1) hipLaunchKernelGGL((softmax_fwd_kernel<T, 1>), grid, blk, 0, st, S, nrows, Lk, ld, causal_Lq);
  else if (need <= 2) hipLaunchKernelGGL((softmax_fwd_kernel<T, 2>), grid, blk, 0, st, S, nrows, Lk, ld, causal_Lq);
  else if (need <= 4) hipLaunchKernelGGL((softmax_fwd_kernel<T, 4>), grid, blk, 0, st, S, nrows, Lk, ld, causal_Lq);
  else if (need <= 8) hipLaunchKernelGGL((softmax_fwd_kernel<T, 8>), grid, blk, 0, st, S, nrows, Lk, ld, causal_Lq);
  else if (need <= 16) hipLaunchKernelGGL((softmax_fwd_kernel<T, 16>), grid, blk, 0, st, S, nrows, Lk, ld, causal_Lq);
  else { set_error("softmax: row length %d too long", ld); return -1; }
  DPB_CHECK(hipGetLastError());
  return 0;
}

int launch_softmax_fwd(int dtype, void* S, long Z, int Lq, int Lk, int ld, int causal, hipStream_t st) {
  if (ld % dt_chunk(dtype)) { set_error("softmax: ld=%d not chunk aligned", ld); return -1; }
  const int cq = causal ? Lq : 0;
  return DPB_DISPATCH_T(dtype, T, pick_fwd<T>((T*)S, Z * Lq, Lk, ld, cq, st));
}

template <typename T>
static int pick_jvp(const T* P, T* dS, float* D, long nrows, int Z2, int kps, int Lq, int Lk, int ld, hipStream_t st) {
  constexpr int CH = TT<T>::CH;
  int need = (ld / CH + 63) / 64;
  dim3 grid((unsigned)((nrows + 3) / 4)), blk(256);
  if (need <= 1) hipLaunchKernelGGL((softmax_jvp_kernel<T, 1>), grid, blk, 0, st, P, dS, D, nrows, Z2, kps, Lq, Lk, ld);
  else if (need <= 2) hipLaunchKernelGGL((softmax_jvp_kernel<T, 2>), grid, blk, 0, st, P, dS, D, nrows, Z2, kps, Lq, Lk, ld);
  else if (need <= 4) hipLaunchKernelGGL((softmax_jvp_kernel<T, 4>), grid, blk, 0, st, P, dS, D, nrows, Z2, kps, Lq, Lk, ld);
  else if (need <= 8) hipLaunchKernelGGL((softmax_jvp_kernel<T, 8>), grid, blk, 0, st, P, dS, D, nrows, Z2, kps, Lq, Lk, ld);
  else if (need <= 16) hipLaunchKernelGGL((softmax_jvp_kernel<T, 16>), grid, blk, 0, st, P, dS, D, nrows, Z2, kps, Lq, Lk, ld);
  else { set_error("softmax: row length %d too long", ld); return -1; }
  DPB_CHECK(hipGetLastError());
  return 0;
}

int launch_softmax_jvp(int dtype, const void* P, void* dS, float* D, long Z, int Z2, int kps, int Lq, int Lk, int ld,
                       hipStream_t st) {
  if (ld % dt_chunk(dtype)) { set_error("softmax: ld=%d not chunk aligned", ld); return -1; }
  return DPB_DISPATCH_T(dtype, T, pick_jvp<T>((const T*)P, (T*)dS, D, Z * Lq, Z2, kps, Lq, Lk, ld, st));
}

int launch_softmax_adjT(int dtype, const void* PT, void* gPT, const float* D, long Z, int Z2, int kps, int Lk, int Lq, int ld,
                        hipStream_t st) {
  long total = Z * Lk * (ld / dt_chunk(dtype));
  unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  DPB_DISPATCH_STMT(dtype, T, hipLaunchKernelGGL((softmax_adjT_kernel<T>), dim3(grid), dim3(256), 0, st, (const T*)PT, (T*)gPT, D, Z, Z2, kps, Lk, Lq, ld));
  DPB_CHECK(hipGetLastError());
  return 0;
}

int launch_transpose(int dtype, const void* in, void* out, int Z1, int Z2, long s1, long s2, int R, int Ccols, int ldin, int ldout,
                     long outZstride, hipStream_t st) {
  dim3 grid((ldout + 31) / 32, (Ccols + 31) / 32, Z1 * Z2);
  if (dtype == DT_F32)   // pure data movement: the two 16-bit types share the bf16 instantiation
    hipLaunchKernelGGL((transpose_kernel<float>), grid, dim3(256), 0, st, (const float*)in, (float*)out, Z2, s1, s2, R, Ccols, ldin, ldout, outZstride);
  else
    hipLaunchKernelGGL((transpose_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)in, (bf16*)out, Z2, s1, s2, R, Ccols, ldin, ldout, outZstride);
  DPB_CHECK(hipGetLastError());
  return 0;
}

}  // namespace dpb

// Re-orthonormalisation step of the subspace iteration: thin SVD of the k x N matrix W = J^T J V_prev
// (k <= 56 in 16-wide column tiles, N up to 196 608), fp32 in/out with fp64 Gram and eigen-solve.
//
// Replaces torch.linalg.svd(v_, full_matrices=False) at reference src/utils/utils.py:799 (and :233):
//   W = U S V^T  ->  rows of V^T (descending S) and s = sqrt(S).
// Method: G = W W^T (k x k, fp64) -> cyclic Jacobi eigen-decomposition G = Q L Q^T on one thread
// -> V^T = L^-1/2 Q^T W.  Every reduction across blocks goes through per-block partials added in block order (no atomics: bitwise
// reproducible).  One streaming pass over W for G, one for V^T: HBM/L2-bound, ~2 reads + 1
// write of k*N floats.  LAPACK leaves the sign of each singular vector arbitrary; here each row is
// signed to have non-negative overlap with the previous iterate (needs W V_prev^T, accumulated in the
// same pass), which makes the reference's stop rule allclose(V_prev, V) well defined.
// Also emits ||V - V_prev||_2 and max(|V - V_prev| - 1e-5 |V|) so the stop test (utils.py:803-806)
// needs a single 8-byte read-back.
#include "kernels.h"

namespace dpb {

constexpr int KMAX_ALL = 56;          // largest supported rank (the reference's default pca_rank is 50)

// grid (nblk, k, ceil(k/16)): block (b, i, jt) accumulates G[i][jt*16..] and X[i][jt*16..] = W_i . Vprev_j over its slice of N and stores them as
// partial b: Gp[b][0 | 1][k][k] (plain stores; eig_kernel adds the partials in block order -- no atomics, bitwise reproducible)
__global__ __launch_bounds__(256) void gram_kernel(const float* W, const float* Vp, double* Gp, int k, long N) {
  constexpr int KMAX = 16;
  const int i = blockIdx.y, j0 = blockIdx.z * KMAX;
  double g[KMAX], x[KMAX];   // fp64 accumulation: small singular values survive the squaring in the Gram matrix
#pragma unroll
  for (int j = 0; j < KMAX; ++j) g[j] = x[j] = 0.0;
  for (long n = (long)blockIdx.x * 256 + threadIdx.x; n < N; n += (long)gridDim.x * 256) {
    const double wi = W[(long)i * N + n];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (j0 + j < k) {
        g[j] += wi * (double)W[(long)(j0 + j) * N + n];
        x[j] += wi * (double)Vp[(long)(j0 + j) * N + n];
      }
    }
  }
  __shared__ double red[2][KMAX][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    if (j0 + j < k) {
      double a = g[j], b = x[j];
      for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
      if (lane == 0) { red[0][j][wave] = a; red[1][j][wave] = b; }
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * KMAX) {
    int w = threadIdx.x / KMAX, j = threadIdx.x % KMAX;
    if (j0 + j < k) {
      double s = red[w][j][0] + red[w][j][1] + red[w][j][2] + red[w][j][3];
      Gp[(((long)blockIdx.x * 2 + w) * k + i) * k + j0 + j] = s;
    }
  }
}

// one wave: parallel cyclic Jacobi eigen-solve of the symmetric k x k Gram matrix (thread r owns row/col r),
// then the mixing matrix Cm with V_i = sum_j Cm[i][j] W_j.
template <int KMAX>
__global__ __launch_bounds__(64) void eig_kernel(const double* Gp, int nbg, double* Cm, float* s_out, int k) {
  __shared__ double A[KMAX][KMAX + 1], Q[KMAX][KMAX + 1], X[KMAX * KMAX];
  __shared__ int order[KMAX];
  // LDS budget: KMAX = 56 -> A + Q + X = 76 KB of static LDS: fine on gfx950 (160 KB per workgroup), over the 64 KB of gfx90a / gfx942.
  // This library is gfx950-only by design (Makefile ARCH, no dual paths); the assert documents the ceiling instead of a launch failure elsewhere.
  static_assert(sizeof(double) * (2 * KMAX * (KMAX + 1) + KMAX * KMAX) + sizeof(int) * KMAX <= 160 * 1024, "eig_kernel exceeds the gfx950 LDS");
  const int r = threadIdx.x;
  for (int e = r; e < k * k; e += 64) {          // Gram matrix and overlaps: the blocks' partials added in block order
    double g = 0.0, x = 0.0;
    for (int b0 = 0; b0 < nbg; b0 += 32) {       // 32 partials of each in flight (clamped loads), then added in block order: two dependent rounds
      double tg[32], tx[32];                      // for the 64 Gram blocks of a 4x64x64 latent instead of eight (one L2 round trip each on the critical path)
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        const long b = min(b0 + u, nbg - 1);
        tg[u] = Gp[(b * 2) * k * k + e];
        tx[u] = Gp[(b * 2 + 1) * k * k + e];
      }
#pragma unroll
      for (int u = 0; u < 32; ++u)
        if (b0 + u < nbg) { g += tg[u]; x += tx[u]; }
    }
    Q[e / k][e % k] = g;
    X[e] = x;
  }
  __syncthreads();
  if (r < k)
    for (int j = 0; j < k; ++j) A[r][j] = 0.5 * (Q[r][j] + Q[j][r]);
  __syncthreads();
  if (r < k)
    for (int j = 0; j < k; ++j) Q[r][j] = r == j ? 1.0 : 0.0;
  __syncthreads();
  for (int sweep = 0; sweep < 14; ++sweep) {
    // converged when the off-diagonal mass is negligible (checked by every thread on the same data: uniform)
    double off = 0, diag = 0;
    for (int i = 0; i < k; ++i) { diag += A[i][i] * A[i][i]; for (int j = i + 1; j < k; ++j) off += A[i][j] * A[i][j]; }
    __syncthreads();
    if (off <= 1e-28 * diag) break;
    for (int p = 0; p < k - 1; ++p)
      for (int q = p + 1; q < k; ++q) {
        const double apq = A[p][q], app = A[p][p], aqq = A[q][q];
        __syncthreads();
        if (fabs(apq) <= 1e-300 || fabs(apq) <= 1e-18 * sqrt(fabs(app * aqq))) continue;   // uniform
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        if (r < k) {                       // A <- A J   (row r)
          double arp = A[r][p], arq = A[r][q];
          A[r][p] = c * arp - s * arq;
          A[r][q] = s * arp + c * arq;
          double qrp = Q[r][p], qrq = Q[r][q];
          Q[r][p] = c * qrp - s * qrq;
          Q[r][q] = s * qrp + c * qrq;
        }
        __syncthreads();
        if (r < k) {                       // A <- J^T A (column r)
          double apr = A[p][r], aqr = A[q][r];
          A[p][r] = c * apr - s * aqr;
          A[q][r] = s * apr + c * aqr;
        }
        __syncthreads();
      }
  }
  if (r == 0) {
    for (int i = 0; i < k; ++i) order[i] = i;
    for (int i = 0; i < k; ++i)          // selection sort, descending eigenvalue
      for (int j = i + 1; j < k; ++j)
        if (A[order[j]][order[j]] > A[order[i]][order[i]]) { int t = order[i]; order[i] = order[j]; order[j] = t; }
  }
  __syncthreads();
  if (r < k) {
    const int i = r, e = order[i];
    double lam = A[e][e] > 0 ? A[e][e] : 0.0;
    double sig = sqrt(lam);                       // singular value of W
    s_out[i] = (float)sqrt(sig);                  // reference returns s.sqrt() (utils.py:810)
    double inv = sig > 1e-150 ? 1.0 / sig : 0.0;
    double dot = 0;
    for (int j = 0; j < k; ++j) dot += Q[j][e] * inv * X[j * k + i];
    double sgn = dot < 0 ? -1.0 : 1.0;
    for (int j = 0; j < k; ++j) Cm[i * k + j] = sgn * Q[j][e] * inv;
  }
}

template <int KMAX>
__global__ __launch_bounds__(256) void orth_apply_kernel(const float* W, const float* Vp, float* V, const double* Cm, double* acc, int k,
                                                         long N) {
  __shared__ double cm[KMAX * KMAX];
  for (int i = threadIdx.x; i < k * k; i += 256) cm[i] = Cm[i];
  __syncthreads();
  double d2 = 0;
  float viol = 0.f;
  for (long n = (long)blockIdx.x * 256 + threadIdx.x; n < N; n += (long)gridDim.x * 256) {
    float w[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) w[j] = j < k ? W[(long)j * N + n] : 0.f;
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
      if (i < k) {
        double v = 0;
#pragma unroll
        for (int j = 0; j < KMAX; ++j)
          if (j < k) v += cm[i * k + j] * (double)w[j];
        float vf = (float)v;
        float dlt = Vp[(long)i * N + n] - vf;
        V[(long)i * N + n] = vf;
        d2 += (double)dlt * dlt;
        viol = fmaxf(viol, fabsf(dlt) - 1e-5f * fabsf(vf));
      }
    }
  }
  // block reduce
  __shared__ double rd[4];
  __shared__ float rv[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int o = 32; o > 0; o >>= 1) {
    d2 += __shfl_xor(d2, o, 64);
    viol = fmaxf(viol, __shfl_xor(viol, o, 64));
  }
  if (lane == 0) { rd[wave] = d2; rv[wave] = viol; }
  __syncthreads();
  if (threadIdx.x == 0) {                        // this block's partial (finish kernel: block order)
    acc[2 * blockIdx.x] = ((rd[0] + rd[1]) + rd[2]) + rd[3];
    acc[2 * blockIdx.x + 1] = (double)fmaxf(fmaxf(fmaxf(rv[0], rv[1]), fmaxf(rv[2], rv[3])), 0.f);
  }
}

__global__ void orth_finish_kernel(const double* acc, int nb, float* conv) {
  double d2 = 0.0, m = 0.0;
  for (int b0 = 0; b0 < nb; b0 += 16) {          // sixteen partial pairs in flight, added in block order
    double t[16], v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int b = min(b0 + u, nb - 1); t[u] = acc[2 * b]; v[u] = acc[2 * b + 1]; }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (b0 + u < nb) { d2 += t[u]; m = fmax(m, v[u]); }
  }
  conv[0] = (float)sqrt(d2);
  conv[1] = (float)m;
}

// blocks of the streaming passes: one 256-column slice per block up to 256 blocks (a 4 x 64 x 64 latent: 64 blocks, one load round per thread -- with 2048
// columns per block the eight dependent rounds of a thread made each pass 22 us of pure latency), beyond that a grid-stride loop
static inline unsigned orth_nb(long N) { unsigned nb = (unsigned)((N + 255) / 256); return nb > 256 ? 256 : (nb < 1 ? 1 : nb); }
static inline unsigned orth_nbg(long N) { const unsigned nb = orth_nb(N); return nb > 64 ? 64 : nb; }
size_t orth_scratch_bytes(int k, long N) {     // Cm [k][k] | Gram partials [nbg][2][k][k] | (dist^2, violation) partials [nb][2]
  return sizeof(double) * ((size_t)k * k * (1 + 2 * orth_nbg(N)) + 2 * (size_t)orth_nb(N));
}

int launch_orth(const OrthArgs& a, hipStream_t st) {
  if (a.k < 1 || a.k > KMAX_ALL) { set_error("orth: pca_rank k=%d outside [1,%d]", a.k, KMAX_ALL); return -1; }
  const int k = a.k;
  if (a.scratch_bytes < orth_scratch_bytes(k, a.N)) { set_error("orth: scratch of %zu bytes, dpb_orth_scratch_bytes(k=%d, N=%ld) = %zu needed", a.scratch_bytes, k, a.N, orth_scratch_bytes(k, a.N)); return -1; }
  const unsigned nb = orth_nb(a.N), nbg = orth_nbg(a.N);
  double* Cm = a.scratch;
  double* Gp = Cm + k * k;
  double* acc = Gp + (size_t)2 * nbg * k * k;
  hipLaunchKernelGGL(gram_kernel, dim3(nbg, k, (k + 15) / 16), dim3(256), 0, st, a.W, a.Vprev, Gp, k, a.N);
  if (k <= 16) {
    hipLaunchKernelGGL((eig_kernel<16>), dim3(1), dim3(64), 0, st, Gp, (int)nbg, Cm, a.s, k);
    hipLaunchKernelGGL((orth_apply_kernel<16>), dim3(nb), dim3(256), 0, st, a.W, a.Vprev, a.V, Cm, acc, k, a.N);
  } else {
    hipLaunchKernelGGL((eig_kernel<KMAX_ALL>), dim3(1), dim3(64), 0, st, Gp, (int)nbg, Cm, a.s, k);
    hipLaunchKernelGGL((orth_apply_kernel<KMAX_ALL>), dim3(nb), dim3(256), 0, st, a.W, a.Vprev, a.V, Cm, acc, k, a.N);
  }
  hipLaunchKernelGGL(orth_finish_kernel, dim3(1), dim3(1), 0, st, acc, (int)nb, a.conv);
  DPB_CHECK(hipGetLastError());
  return 0;
}

}  // namespace dpb

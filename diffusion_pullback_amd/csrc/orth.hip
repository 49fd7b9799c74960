// Re-orthonormalisation step of the subspace iteration: thin SVD of the k x N matrix W = J^T J V_prev
// (k <= 128 in 16-wide column tiles, N up to 196 608), fp32 in/out with fp64 Gram and eigen-solve.
//
// Replaces torch.linalg.svd(v_, full_matrices=False) at reference src/utils/utils.py:799 (and :233):
//   W = U S V^T  ->  rows of V^T (descending S) and s = sqrt(S).
// Method: G = W W^T (k x k, fp64) -> Jacobi eigen-decomposition G = Q L Q^T in one block (k <= 16: cyclic order on one wave, the headline's
// kernel since round 1; k > 16: round-robin order, k/2 disjoint rotations per step on four waves, round 6) -> V^T = L^-1/2 Q^T W.  Every reduction across blocks goes through per-block partials added in block order (no atomics: bitwise
// reproducible).  One streaming pass over W for G, one for V^T: HBM/L2-bound, ~2 reads + 1
// write of k*N floats.  LAPACK leaves the sign of each singular vector arbitrary; here each row is
// signed to have non-negative overlap with the previous iterate (needs W V_prev^T, accumulated in the
// same pass), which makes the reference's stop rule allclose(V_prev, V) well defined.
// Also emits ||V - V_prev||_2 and max(|V - V_prev| - 1e-5 |V|) so the stop test (utils.py:803-806)
// needs a single 8-byte read-back.
#include "kernels.h"

namespace dpb {

// samples of a batch advanced together (dpb_pullback_iterate): one launch per kernel for all of them, the sample on a grid axis; strides in elements
struct OrthBatch { long w, v, s, conv, scratch; };

constexpr int KMAX_REG = 56;          // largest rank of the register-tiled apply kernel (the reference's default pca_rank is 50)
constexpr int KMAX_ALL = ORTH_MAX_RANK;   // largest supported rank, 128 (A [k][k+1] fp64 of the eigen-solve has to fit the 160 KB of LDS)

// grid (nblk, k, ceil(k/16)): block (b, i, jt) accumulates G[i][jt*16..] and X[i][jt*16..] = W_i . Vprev_j over its slice of N and stores them as
// partial b: Gp[b][0 | 1][k][k] (plain stores; eig_kernel adds the partials in block order -- no atomics, bitwise reproducible)
__global__ __launch_bounds__(256) void gram_kernel(const float* W, const float* Vp, double* Gp, int k, long N, OrthBatch ob) {
  constexpr int KMAX = 16;
  const int i = blockIdx.y % k, j0 = blockIdx.z * KMAX, smp = blockIdx.y / k;
  W += smp * ob.w; Vp += smp * ob.v; Gp += smp * ob.scratch;
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
__global__ __launch_bounds__(64) void eig_kernel(const double* Gp, int nbg, double* Cm, float* s_out, int k, OrthBatch ob) {
  Gp += blockIdx.x * ob.scratch; Cm += blockIdx.x * ob.scratch; s_out += blockIdx.x * ob.s;
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

// One block of 1024 threads: two-sided Jacobi in the round-robin (tournament) order.  A step rotates floor(k/2) DISJOINT index pairs at once: the angle of
// a pair (p, q) depends on A[p][p], A[q][q], A[p][q] only, which no other pair of the step touches, so a step is the product of its rotations in any
// order -- A <- A J for every row (and Q <- Q J), barrier, A <- J^T A for every column, barrier.  k - 1 (k even) or k steps make a sweep that visits every
// pair once.  Against the one-wave cyclic kernel below (3 barriers per ROTATION): k = 50 is 49 steps of 25 rotations per sweep instead of 1225 dependent
// rotations -- 3.9 ms -> see profiles/r06_eig_parallel.txt.  fp64 throughout; same thresholds, same Gram partial order, same output rules as eig_kernel.
// QGLOBAL: the eigenvector matrix lives in global memory (the slot of Gram partial 0, free once the partials are summed) when A and Q do not both fit the LDS.
template <int KMAX, bool QGLOBAL, int NT>
__global__ __launch_bounds__(NT) void eig_par_kernel(double* Gp, int nbg, double* Cm, float* s_out, int k, OrthBatch ob) {
  constexpr int NW = NT / 64;
  Gp += blockIdx.x * ob.scratch; Cm += blockIdx.x * ob.scratch; s_out += blockIdx.x * ob.s;
  constexpr int QR = QGLOBAL ? 1 : KMAX, QC = QGLOBAL ? 1 : KMAX + 1;
  __shared__ double A[KMAX][KMAX + 1], Ql[QR][QC];
  __shared__ double rc[KMAX / 2], rs[KMAX / 2], lam[KMAX], red[2][NW], sgn[KMAX];
  __shared__ int rp[KMAX / 2], rq[KMAX / 2], order[KMAX];
  static_assert(sizeof(double) * (KMAX * (KMAX + 1) + QR * QC + 3 * KMAX + 2 * NW) + sizeof(int) * 2 * KMAX <= 160 * 1024, "eig_par_kernel exceeds the gfx950 LDS");
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  double* const Xg = Gp + (long)k * k;           // overlaps W . Vprev^T: summed into partial 0's own slot
  double* const Qg = Gp;                         // QGLOBAL: eigenvectors in partial 0's Gram slot, TRANSPOSED (the lanes of a wave walk the rows of Q: coalesced)
  auto Qat = [&](int i, int j) -> double& { if constexpr (QGLOBAL) return Qg[(long)j * k + i]; else return Ql[i][j]; };
  for (int e = t; e < k * k; e += NT) {          // Gram matrix and overlaps: the blocks' partials added in block order (as eig_kernel)
    double g = 0.0, x = 0.0;
    for (int b0 = 0; b0 < nbg; b0 += 32) {
      double tg[32], tx[32];
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
    A[e / k][e % k] = g;                         // raw sums; symmetrised below
    Xg[e] = x;                                   // element e of partial 0 was read by this thread only: no other reader to race with
  }
  __syncthreads();
  for (int e = t; e < k * k; e += NT) {          // A <- (G + G^T) / 2: the pair (i, j), (j, i) belongs to the thread that holds e = (i, j), i < j
    const int i = e / k, j = e % k;
    if (i < j) { const double v = 0.5 * (A[i][j] + A[j][i]); A[i][j] = v; A[j][i] = v; }
    Qat(i, j) = i == j ? 1.0 : 0.0;              // (QGLOBAL: partial 0's Gram slot, free since the barrier above)
  }
  __syncthreads();
  const int m = (k + 1) & ~1, half = m / 2, steps = m - 1;      // players 0..m-1 (player k is a bye when k is odd)
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0, diag = 0;                    // converged when the off-diagonal mass is negligible: fixed-order block reduction, uniform result
    for (int i = wave; i < k; i += NW)
      for (int j = lane; j < k; j += 64) {
        const double v = A[i][j] * A[i][j];
        if (i == j) diag += v; else if (i < j) off += v;
      }
    for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o, 64); diag += __shfl_xor(diag, o, 64); }
    if (lane == 0) { red[0][wave] = off; red[1][wave] = diag; }
    __syncthreads();
    off = 0; diag = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { off += red[0][w]; diag += red[1][w]; }
    __syncthreads();
    if (off <= 1e-28 * diag) break;
    for (int st = 0; st < steps; ++st) {
      if (t < half) {                            // the step's pairs and their angles
        int p = t == 0 ? m - 1 : (st + t) % (m - 1);
        int q = t == 0 ? st % (m - 1) : (st - t + (m - 1)) % (m - 1);
        if (p > q) { const int w = p; p = q; q = w; }
        double c = 1.0, s = 0.0;
        if (q < k) {
          const double apq = A[p][q], app = A[p][p], aqq = A[q][q];
          if (!(fabs(apq) <= 1e-300 || fabs(apq) <= 1e-18 * sqrt(fabs(app * aqq)))) {
            const double theta = (aqq - app) / (2.0 * apq);
            const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            c = 1.0 / sqrt(tt * tt + 1.0); s = tt * c;
          }
        }
        rp[t] = p; rq[t] = q < k ? q : p; rc[t] = c; rs[t] = s;       // (bye or negligible element: s = 0, skipped below)
      }
      __syncthreads();
      for (int i = wave; i < half; i += NW) {    // A <- A J, Q <- Q J: the wave takes pair i, its lanes the rows
        const int p = rp[i], q = rq[i];
        const double c = rc[i], s = rs[i];
        if (s == 0.0) continue;                  // wave-uniform
        for (int r = lane; r < k; r += 64) {
          const double arp = A[r][p], arq = A[r][q];
          A[r][p] = c * arp - s * arq;
          A[r][q] = s * arp + c * arq;
          const double qrp = Qat(r, p), qrq = Qat(r, q);
          Qat(r, p) = c * qrp - s * qrq;
          Qat(r, q) = s * qrp + c * qrq;
        }
      }
      __syncthreads();
      for (int i = wave; i < half; i += NW) {    // A <- J^T A: the wave takes pair i, its lanes the columns
        const int p = rp[i], q = rq[i];
        const double c = rc[i], s = rs[i];
        if (s == 0.0) continue;
        for (int r = lane; r < k; r += 64) {
          const double apr = A[p][r], aqr = A[q][r];
          A[p][r] = c * apr - s * aqr;
          A[q][r] = s * apr + c * aqr;
        }
      }
      __syncthreads();
    }
  }
  if (t < k) lam[t] = A[t][t];
  __syncthreads();
  if (t < k) {                                   // descending eigenvalue; ties by index (rank counting instead of a serial sort)
    int rank = 0;
    const double mine = lam[t];
    for (int j = 0; j < k; ++j) rank += (lam[j] > mine || (lam[j] == mine && j < t)) ? 1 : 0;
    order[rank] = t;
  }
  __syncthreads();
  if (t < k) {
    const int i = t, e = order[i];
    const double l = lam[e] > 0 ? lam[e] : 0.0;
    const double sig = sqrt(l);                   // singular value of W
    s_out[i] = (float)sqrt(sig);                  // reference returns s.sqrt() (utils.py:810)
    const double inv = sig > 1e-150 ? 1.0 / sig : 0.0;
    double dot = 0;
    for (int j = 0; j < k; ++j) dot += Qat(j, e) * inv * Xg[j * k + i];
    sgn[i] = (dot < 0 ? -1.0 : 1.0) * inv;
  }
  __syncthreads();
  for (int id = t; id < k * k; id += NT) {
    const int i = id / k, j = id % k;
    Cm[id] = sgn[i] * Qat(j, order[i]);
  }
}

template <int KMAX>
__global__ __launch_bounds__(256) void orth_apply_kernel(const float* W, const float* Vp, float* V, const double* Cm, double* acc, int k,
                                                         long N, OrthBatch ob) {
  W += blockIdx.y * ob.w; Vp += blockIdx.y * ob.v; V += blockIdx.y * ob.v; Cm += blockIdx.y * ob.scratch; acc += blockIdx.y * ob.scratch;
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

// k > 56: the same product V = Cm W in row tiles of 16 (the register-tiled kernel above holds a whole column of W and unrolls k x k products: 73 KB of code
// at 56).  One block per 256-column slice as above (same partial layout: scratch contract unchanged); W is re-read once per row tile (L2-resident).
// In-place safe like the kernel above: a thread reads Vp[i][n] before it writes V[i][n], nobody else touches that element.
__global__ __launch_bounds__(256) void orth_apply_tiled_kernel(const float* W, const float* Vp, float* V, const double* Cm, double* acc, int k, long N,
                                                               OrthBatch ob) {
  W += blockIdx.y * ob.w; Vp += blockIdx.y * ob.v; V += blockIdx.y * ob.v; Cm += blockIdx.y * ob.scratch; acc += blockIdx.y * ob.scratch;
  constexpr int IT = 16;
  extern __shared__ double cmt[];                // [IT][k]
  double d2 = 0;
  float viol = 0.f;
  for (int i0 = 0; i0 < k; i0 += IT) {
    __syncthreads();
    for (int e = threadIdx.x; e < IT * k; e += 256) cmt[e] = i0 + e / k < k ? Cm[(long)i0 * k + e] : 0.0;
    __syncthreads();
    for (long n = (long)blockIdx.x * 256 + threadIdx.x; n < N; n += (long)gridDim.x * 256) {
      double v[IT];
#pragma unroll
      for (int i = 0; i < IT; ++i) v[i] = 0.0;
      for (int j = 0; j < k; ++j) {
        const double w = (double)W[(long)j * N + n];
#pragma unroll
        for (int i = 0; i < IT; ++i) v[i] += cmt[i * k + j] * w;
      }
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        if (i0 + i < k) {
          const float vf = (float)v[i];
          const float dlt = Vp[(long)(i0 + i) * N + n] - vf;
          V[(long)(i0 + i) * N + n] = vf;
          d2 += (double)dlt * dlt;
          viol = fmaxf(viol, fabsf(dlt) - 1e-5f * fabsf(vf));
        }
      }
    }
  }
  __shared__ double rd[4];
  __shared__ float rv[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int o = 32; o > 0; o >>= 1) {
    d2 += __shfl_xor(d2, o, 64);
    viol = fmaxf(viol, __shfl_xor(viol, o, 64));
  }
  if (lane == 0) { rd[wave] = d2; rv[wave] = viol; }
  __syncthreads();
  if (threadIdx.x == 0) {
    acc[2 * blockIdx.x] = ((rd[0] + rd[1]) + rd[2]) + rd[3];
    acc[2 * blockIdx.x + 1] = (double)fmaxf(fmaxf(fmaxf(rv[0], rv[1]), fmaxf(rv[2], rv[3])), 0.f);
  }
}

__global__ void orth_finish_kernel(const double* acc, int nb, float* conv, OrthBatch ob) {
  acc += blockIdx.x * ob.scratch; conv += blockIdx.x * ob.conv;
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
  const unsigned B = a.batch < 1 ? 1 : (unsigned)a.batch;
  if (B > 1 && (a.scratch_stride < orth_scratch_bytes(k, a.N) || a.scratch_stride % sizeof(double) || (long)B * k > 65535)) {
    set_error("orth: batch of %u samples with a scratch stride of %zu bytes (k=%d)", B, a.scratch_stride, k); return -1;
  }
  const OrthBatch ob{a.stride_w, a.stride_v, a.stride_s, a.stride_conv, (long)(a.scratch_stride / sizeof(double))};
  const unsigned nb = orth_nb(a.N), nbg = orth_nbg(a.N);
  double* Cm = a.scratch;
  double* Gp = Cm + k * k;
  double* acc = Gp + (size_t)2 * nbg * k * k;
  hipLaunchKernelGGL(gram_kernel, dim3(nbg, k * B, (k + 15) / 16), dim3(256), 0, st, a.W, a.Vprev, Gp, k, a.N, ob);
  // Eigen-solve: k <= 5 the one-wave cyclic kernel of rounds 1-5 (the headline's bits are unchanged; at k = 5 the two take the same time), above that
  // the round-robin kernel -- four waves up to k = 16 (8: 127 -> 78 us per dpb_orth, 10: 223 -> 134, 16: 528 -> 215), sixteen beyond
  // (50: 6.3 -> 1.2 ms; profiles/r06_eig_parallel.txt).  DPB_EIG_PAR=0: the cyclic kernel wherever it exists (k <= 56), 2: round-robin for every k.
  static const int par = getenv("DPB_EIG_PAR") ? atoi(getenv("DPB_EIG_PAR")) : 1;
  const bool cyclic = par == 0 ? k <= KMAX_REG : (par == 2 ? false : k <= 5);
  if (cyclic && k <= 16) {
    hipLaunchKernelGGL((eig_kernel<16>), dim3(B), dim3(64), 0, st, Gp, (int)nbg, Cm, a.s, k, ob);
  } else if (cyclic) {
    hipLaunchKernelGGL((eig_kernel<KMAX_REG>), dim3(B), dim3(64), 0, st, Gp, (int)nbg, Cm, a.s, k, ob);
  } else if (k <= 16) {
    hipLaunchKernelGGL((eig_par_kernel<16, false, 256>), dim3(B), dim3(256), 0, st, Gp, (int)nbg, Cm, a.s, k, ob);
  } else if (k <= 32) {
    hipLaunchKernelGGL((eig_par_kernel<32, false, 1024>), dim3(B), dim3(1024), 0, st, Gp, (int)nbg, Cm, a.s, k, ob);
  } else if (k <= 64) {
    hipLaunchKernelGGL((eig_par_kernel<64, false, 1024>), dim3(B), dim3(1024), 0, st, Gp, (int)nbg, Cm, a.s, k, ob);
  } else if (k <= 96) {
    hipLaunchKernelGGL((eig_par_kernel<96, false, 1024>), dim3(B), dim3(1024), 0, st, Gp, (int)nbg, Cm, a.s, k, ob);
  } else {
    hipLaunchKernelGGL((eig_par_kernel<KMAX_ALL, true, 1024>), dim3(B), dim3(1024), 0, st, Gp, (int)nbg, Cm, a.s, k, ob);
  }
  if (k <= 16) {
    hipLaunchKernelGGL((orth_apply_kernel<16>), dim3(nb, B), dim3(256), 0, st, a.W, a.Vprev, a.V, Cm, acc, k, a.N, ob);
  } else if (k <= KMAX_REG) {
    hipLaunchKernelGGL((orth_apply_kernel<KMAX_REG>), dim3(nb, B), dim3(256), 0, st, a.W, a.Vprev, a.V, Cm, acc, k, a.N, ob);
  } else {
    hipLaunchKernelGGL(orth_apply_tiled_kernel, dim3(nb, B), dim3(256), sizeof(double) * 16 * k, st, a.W, a.Vprev, a.V, Cm, acc, k, a.N, ob);
  }
  hipLaunchKernelGGL(orth_finish_kernel, dim3(B), dim3(1), 0, st, acc, (int)nb, a.conv, ob);
  DPB_CHECK(hipGetLastError());
  return 0;
}

}  // namespace dpb

// HBM-bound elementwise kernels: GEGLU (primal/tangent/adjoint), SiLU, accumulate, channel-window copies,
// NCHW<->NHWC boundary transposes, 2x2 sum pooling (adjoint of nearest upsampling), DDIM step.
// All use 16-byte chunks per lane and grid-stride loops.
#include "kernels.h"

namespace dpb {

static inline unsigned grid_for(long n) {
  long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void geglu_kernel(GegluArgs a, long nrows) {
  constexpr int CH = TT<T>::CH;
  const int fch = a.F / CH;
  const long total = nrows * fch;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long row = idx / fch;
    const int c = (int)(idx - row * fch) * CH;
    long prow = row;
    if (MODE != MODE_PRIMAL) {
      long j = row / a.rows_per_sample, l = row - j * a.rows_per_sample;
      prow = (j / a.kps) * a.rows_per_sample + l;
    }
    // column of a-unit c inside a [..][2F] row, and the distance to its gate: plain layout (a | g halves) or 64-blocks interleaved
    const int ca = a.il ? ((c >> 6) << 7) + (c & 63) : c, dg_ = a.il ? 64 : a.F;
    const T* hp = (const T*)a.h + prow * 2 * a.F;
    float av[CH], gv[CH], o[CH];
    Vec<T>::load(hp + ca, av);
    Vec<T>::load(hp + ca + dg_, gv);
    // The primal pass REPLACES (a, g) in h by the two factors every tangent / adjoint pass needs,
    //   G1 = gelu(g)  (in a's slot)   and   G2 = a * gelu'(g)  (in g's slot),
    // so that  dy = da*G1 + dg*G2  and  (ga, gg) = gy * (G1, G2)  cost two FMAs per element -- cheap enough to live in GEMM epilogues
    // (epilogue.h); h is read by nothing else after this kernel.
    if (MODE == MODE_PRIMAL) {
      float g1[CH], g2[CH];
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        const float er = erff(gv[e] * 0.70710678118654752f);
        g1[e] = 0.5f * gv[e] * (1.f + er);
        g2[e] = av[e] * (0.5f * (1.f + er) + gv[e] * 0.39894228040143268f * __expf(-0.5f * gv[e] * gv[e]));
        o[e] = av[e] * g1[e];
      }
      Vec<T>::store((T*)a.y + row * a.F + c, o);
      if (a.stash) {
        T* hw = (T*)a.h + prow * 2 * a.F;
        Vec<T>::store(hw + ca, g1);
        Vec<T>::store(hw + ca + dg_, g2);
      }
    } else if (MODE == MODE_TANGENT) {
      const T* dp = (const T*)a.d + row * 2 * a.F;
      float da[CH], dg[CH];
      Vec<T>::load(dp + ca, da);
      Vec<T>::load(dp + ca + dg_, dg);
#pragma unroll
      for (int e = 0; e < CH; ++e) o[e] = da[e] * av[e] + dg[e] * gv[e];          // av = G1, gv = G2
      Vec<T>::store((T*)a.y + row * a.F + c, o);
    } else {
      float gy[CH], o2[CH];
      Vec<T>::load((const T*)a.d + row * a.F + c, gy);
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        o[e] = gy[e] * av[e];
        o2[e] = gy[e] * gv[e];
      }
      T* yp = (T*)a.y + row * 2 * a.F;
      if (a.accumulate) {
        float t1[CH], t2[CH];
        Vec<T>::load(yp + ca, t1);
        Vec<T>::load(yp + ca + dg_, t2);
#pragma unroll
        for (int e = 0; e < CH; ++e) { o[e] += t1[e]; o2[e] += t2[e]; }
      }
      Vec<T>::store(yp + ca, o);
      Vec<T>::store(yp + ca + dg_, o2);
    }
  }
}

template <typename T>
static int geglu_t(int mode, const GegluArgs& a, hipStream_t st) {
  if (a.F % TT<T>::CH) { set_error("geglu: F=%d not chunk aligned", a.F); return -1; }
  if (a.il && (a.il != 64 || a.F % 64)) { set_error("geglu: interleave %d with F=%d unsupported (blocks of 64)", a.il, a.F); return -1; }
  long nrows = (long)(mode == MODE_PRIMAL ? a.Bp : a.NT) * a.rows_per_sample;
  unsigned g = grid_for(nrows * (a.F / TT<T>::CH));
  if (mode == MODE_PRIMAL) hipLaunchKernelGGL((geglu_kernel<T, MODE_PRIMAL>), dim3(g), dim3(256), 0, st, a, nrows);
  else if (mode == MODE_TANGENT) hipLaunchKernelGGL((geglu_kernel<T, MODE_TANGENT>), dim3(g), dim3(256), 0, st, a, nrows);
  else hipLaunchKernelGGL((geglu_kernel<T, MODE_ADJOINT>), dim3(g), dim3(256), 0, st, a, nrows);
  DPB_CHECK(hipGetLastError());
  return 0;
}
int launch_geglu(int dtype, int mode, const GegluArgs& a, hipStream_t st) {
  return DPB_DISPATCH_T(dtype, T, geglu_t<T>(mode, a, st));
}

template <typename T, int OP>   // 0: silu  1: y = x  2: y += x  3: quick_gelu  4: exact (erf) gelu
__global__ __launch_bounds__(256) void unary_kernel(const T* x, T* y, long nchunks) {
  constexpr int CH = TT<T>::CH;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (long)gridDim.x * 256) {
    float v[CH];
    Vec<T>::load(x + i * CH, v);
    if (OP == 0) {
#pragma unroll
      for (int e = 0; e < CH; ++e) v[e] = silu_(v[e]);
    } else if (OP == 3) {
#pragma unroll
      for (int e = 0; e < CH; ++e) v[e] = v[e] / (1.f + __expf(-1.702f * v[e]));
    } else if (OP == 4) {
#pragma unroll
      for (int e = 0; e < CH; ++e) v[e] = gelu_(v[e]);
    } else if (OP == 2) {
      float o[CH];
      Vec<T>::load(y + i * CH, o);
#pragma unroll
      for (int e = 0; e < CH; ++e) v[e] += o[e];
    }
    Vec<T>::store(y + i * CH, v);
  }
}

template <typename T>
static int unary_t(int op, const void* x, void* y, long n, hipStream_t st) {
  if (n % TT<T>::CH) { set_error("elementwise: n=%ld not chunk aligned", n); return -1; }
  long nc = n / TT<T>::CH;
  unsigned g = grid_for(nc);
  if (op == 0) hipLaunchKernelGGL((unary_kernel<T, 0>), dim3(g), dim3(256), 0, st, (const T*)x, (T*)y, nc);
  else if (op == 1) hipLaunchKernelGGL((unary_kernel<T, 1>), dim3(g), dim3(256), 0, st, (const T*)x, (T*)y, nc);
  else if (op == 3) hipLaunchKernelGGL((unary_kernel<T, 3>), dim3(g), dim3(256), 0, st, (const T*)x, (T*)y, nc);
  else if (op == 4) hipLaunchKernelGGL((unary_kernel<T, 4>), dim3(g), dim3(256), 0, st, (const T*)x, (T*)y, nc);
  else hipLaunchKernelGGL((unary_kernel<T, 2>), dim3(g), dim3(256), 0, st, (const T*)x, (T*)y, nc);
  DPB_CHECK(hipGetLastError());
  return 0;
}
int launch_silu(int dtype, const void* x, void* y, long n, hipStream_t st) {
  return DPB_DISPATCH_T(dtype, T, unary_t<T>(0, x, y, n, st));
}
int launch_quick_gelu(int dtype, const void* x, void* y, long n, hipStream_t st) {
  return DPB_DISPATCH_T(dtype, T, unary_t<T>(3, x, y, n, st));
}
int launch_gelu(int dtype, const void* x, void* y, long n, hipStream_t st) {
  return DPB_DISPATCH_T(dtype, T, unary_t<T>(4, x, y, n, st));
}

// token + position embedding lookup of the text encoder, written in the engine's fp32 [b][c][t] boundary layout
template <typename T>
__global__ __launch_bounds__(256) void embed_tokens_kernel(const int* ids, const T* tok, const T* pos, float* out, int L, int C, int vocab) {
  const int b = blockIdx.y, t = blockIdx.x;
  int id = ids[b * L + t];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  for (int c = threadIdx.x; c < C; c += 256)
    out[((long)b * C + c) * L + t] = TT<T>::ld(tok + (long)id * C + c) + TT<T>::ld(pos + (long)t * C + c);
}
int launch_embed_tokens(int dtype, const int* ids, const void* tok, const void* pos, float* out, int batch, int L, int C, int vocab, hipStream_t st) {
  if (batch <= 0 || L <= 0 || C <= 0 || vocab <= 0) { set_error("embed_tokens: empty problem"); return -1; }
  DPB_DISPATCH_STMT(dtype, T, hipLaunchKernelGGL((embed_tokens_kernel<T>), dim3(L, batch), dim3(256), 0, st, ids, (const T*)tok, (const T*)pos, out, L, C, vocab));
  DPB_CHECK(hipGetLastError());
  return 0;
}

int launch_axpy(int dtype, const void* x, void* y, long n, int accumulate, hipStream_t st) {
  int op = accumulate ? 2 : 1;
  return DPB_DISPATCH_T(dtype, T, unary_t<T>(op, x, y, n, st));
}

template <typename T>
__global__ __launch_bounds__(256) void copy_cols_kernel(const T* src, int lds_, int cs0, T* dst, int ldd, int cd0, long rows,
                                                        int ncols, int accumulate) {
  constexpr int CH = TT<T>::CH;
  const int nch = ncols / CH;
  const long total = rows * nch;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long r = idx / nch;
    const int c = (int)(idx - r * nch) * CH;
    float v[CH];
    Vec<T>::load(src + r * lds_ + cs0 + c, v);
    T* dp = dst + r * ldd + cd0 + c;
    if (accumulate) {
      float o[CH];
      Vec<T>::load(dp, o);
#pragma unroll
      for (int e = 0; e < CH; ++e) v[e] += o[e];
    }
    Vec<T>::store(dp, v);
  }
}
int launch_copy_cols(int dtype, const void* src, int lds_, int cs0, void* dst, int ldd, int cd0, long rows, int ncols, int accumulate,
                     hipStream_t st) {
  int CH = dt_chunk(dtype);
  if (ncols % CH || cs0 % CH || cd0 % CH || lds_ % CH || ldd % CH) { set_error("copy_cols: misaligned window"); return -1; }
  unsigned g = grid_for(rows * (ncols / CH));
  DPB_DISPATCH_STMT(dtype, T, hipLaunchKernelGGL((copy_cols_kernel<T>), dim3(g), dim3(256), 0, st, (const T*)src, lds_, cs0, (T*)dst, ldd, cd0, rows, ncols, accumulate));
  DPB_CHECK(hipGetLastError());
  return 0;
}

// fp32 NCHW -> T NHWC (channel pad to Cpad with zeros).  Small tensors (boundary only): tile transpose through LDS.
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* src, T* dst, int C, int HW, int Cpad) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {   // k: channel, tx: pixel (contiguous in src)
    int c = c0 + k, p = p0 + tx;
    tile[k][tx] = (c < C && p < HW) ? src[((long)n * C + c) * HW + p] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {   // k: pixel, tx: channel (contiguous in dst)
    int p = p0 + k, c = c0 + tx;
    if (p < HW && c < Cpad) TT<T>::st(dst + ((long)n * HW + p) * Cpad + c, tile[tx][k]);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* src, float* dst, int C, int HW, int Cpad) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {   // k: pixel, tx: channel
    int p = p0 + k, c = c0 + tx;
    tile[k][tx] = (p < HW && c < C) ? TT<T>::ld(src + ((long)n * HW + p) * Cpad + c) : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {   // k: channel, tx: pixel
    int c = c0 + k, p = p0 + tx;
    if (c < C && p < HW) dst[((long)n * C + c) * HW + p] = tile[tx][k];
  }
}
int launch_nchw_to_nhwc(int dtype, const float* src, void* dst, int n, int C, int HW, int Cpad, hipStream_t st) {
  dim3 grid((HW + 31) / 32, (Cpad + 31) / 32, n);
  DPB_DISPATCH_STMT(dtype, T, hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), grid, dim3(256), 0, st, src, (T*)dst, C, HW, Cpad));
  DPB_CHECK(hipGetLastError());
  return 0;
}
int launch_nhwc_to_nchw(int dtype, const void* src, float* dst, int n, int C, int HW, int Cpad, hipStream_t st) {
  dim3 grid((HW + 31) / 32, (C + 31) / 32, n);
  DPB_DISPATCH_STMT(dtype, T, hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), grid, dim3(256), 0, st, (const T*)src, dst, C, HW, Cpad));
  DPB_CHECK(hipGetLastError());
  return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void pool2x2_kernel(const T* in, T* out, int n, int H, int W, int C, int accumulate) {
  constexpr int CH = TT<T>::CH;
  const int cch = C / CH;
  const long total = (long)n * H * W * cch;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % cch) * CH;
    long pix = idx / cch;
    const int x = (int)(pix % W);
    pix /= W;
    const int y = (int)(pix % H);
    const long s = pix / H;
    float acc[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) acc[e] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        float v[CH];
        Vec<T>::load(in + ((s * 2 * H + 2 * y + dy) * 2 * W + 2 * x + dx) * C + c, v);
#pragma unroll
        for (int e = 0; e < CH; ++e) acc[e] += v[e];
      }
    T* op = out + ((s * H + y) * W + x) * C + c;
    if (accumulate) {
      float o[CH];
      Vec<T>::load(op, o);
#pragma unroll
      for (int e = 0; e < CH; ++e) acc[e] += o[e];
    }
    Vec<T>::store(op, acc);
  }
}
int launch_pool2x2_sum(int dtype, const void* in, void* out, int n, int H, int W, int C, int accumulate, hipStream_t st) {
  int CH = dt_chunk(dtype);
  unsigned g = grid_for((long)n * H * W * (C / CH));
  DPB_DISPATCH_STMT(dtype, T, hipLaunchKernelGGL((pool2x2_kernel<T>), dim3(g), dim3(256), 0, st, (const T*)in, (T*)out, n, H, W, C, accumulate));
  DPB_CHECK(hipGetLastError());
  return 0;
}

__global__ __launch_bounds__(256) void ddim_kernel(const float* x, const float* e, float* out, float* x0, long n, float sa, float s1a,
                                                   float san, float s1an) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float p = (x[i] - e[i] * s1a) / sa;
    if (x0) x0[i] = p;
    out[i] = san * p + s1an * e[i];
  }
}
int launch_ddim_step(const float* x, const float* e, float* out, float* x0, long n, float a_t, float a_next, hipStream_t st) {
  // same operation order as the reference (utils.py:301-306): sqrt in fp32 of the fp32 alphas
  hipLaunchKernelGGL(ddim_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, e, out, x0, n, sqrtf(a_t), sqrtf(1.f - a_t), sqrtf(a_next),
                     sqrtf(1.f - a_next));
  DPB_CHECK(hipGetLastError());
  return 0;
}

__global__ __launch_bounds__(256) void lincomb_kernel(const float* x, const float* y, const float* z, float* out, long n, float a, float b,
                                                      float c) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v = a * x[i] + b * y[i];
    if (z) v += c * z[i];
    out[i] = v;
  }
}
int launch_lincomb(const float* x, const float* y, const float* z, float* out, long n, float a, float b, float c, hipStream_t st) {
  hipLaunchKernelGGL(lincomb_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, y, z, out, n, a, b, c);
  DPB_CHECK(hipGetLastError());
  return 0;
}

}  // namespace dpb

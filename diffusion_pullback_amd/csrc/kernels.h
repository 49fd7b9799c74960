// Host-side launch interface of the gfx950 kernels (internal; the public C-ABI is include/dpb.h).
#pragma once
#include <algorithm>

#include "common.h"

namespace dpb {

// ---------------------------------------------------------------- GEMM / implicit-GEMM convolution
// C[z][m][n] = alpha * sum_k A[z][m][k] * B[z][n][k]  (+bias[n]) (+rowbias[sample(m)][n]) (+R[z][m][n]) (+C if accumulate)
// A: plain rows (lda) or gathered NHWC pixels (conv).  B is always [N][K], K contiguous.
enum { GATHER_NONE = 0, GATHER_CONV = 1, GATHER_CONVT = 2, GATHER_UPCONV = 3 };
enum { EPI_PLAIN = 0, EPI_GEGLU_TAN = 1, EPI_GEGLU_ADJ = 2, EPI_LN_TAN = 3, EPI_LN_ADJ = 4, EPI_GEGLU_FWD = 5 };   // fused epilogues of the ring GEMMs (epilogue.h)
struct GemmArgs {
  const void* A = nullptr; const void* B = nullptr; void* C = nullptr; const void* R = nullptr;
  const float* bias = nullptr;
  const void* rowbias = nullptr;      // [samples][N] in T; sample(m) = (m / rows_per_sample) / rowbias_div
  int rows_per_sample = 1, rowbias_div = 1;
  int M = 0, N = 0, K = 0;
  int lda = 0, ldb = 0, ldc = 0, ldr = 0;
  // two-level batch z = z1 * Z2 + z2 ; per-operand offset = (z1 / div) * s1 + z2 * s2  (elements)
  int Z1 = 1, Z2 = 1;
  long sA1 = 0, sA2 = 0, sB1 = 0, sB2 = 0, sC1 = 0, sC2 = 0, sR1 = 0, sR2 = 0;
  int divA = 1, divB = 1;
  float alpha = 1.f;
  int accumulate = 0;
  // gather description (A is [samples][H*W][Cin] NHWC, output pixels Ho x Wo, KS x KS taps)
  int gather = GATHER_NONE;
  int H = 0, W = 0, Cin = 0, Ho = 0, Wo = 0, KS = 1, stride = 1, pad = 0;
  // optional second operand pair appended to the K loop (plain rows only)
  const void* A2 = nullptr; const void* B2 = nullptr;
  int K2 = 0, lda2 = 0, ldb2 = 0, divA2 = 1, divB2 = 1;
  long sA21 = 0, sA22 = 0, sB21 = 0, sB22 = 0;
  // split-K scratch (fp32 slabs); splitk / vec_ok are filled in by launch_gemm
  float* slab = nullptr; size_t slab_bytes = 0;
  const void* zeros = nullptr;        // >= 16 zero bytes in device memory (DMA kernel reads it for padding / out-of-range rows)
  int splitk = 1, vec_ok = 0;
  // fused GEGLU epilogues (ring kernels with 128-column tiles, no split-K): primal FF-in output [prows][2F], columns interleaved in
  // blocks of 64 (a | g); tangent row m belongs to primal sample (m / rows_per_sample) / epi_kps
  int epi = EPI_PLAIN, epi_kps = 1;
  const void* hprim = nullptr;
  // fused LayerNorm epilogues (row-complete 128 x 320 tile, gemm_ring64.hip): EPI_LN_TAN writes h = acc (+R) to C AND its LayerNorm tangent to C2;
  // EPI_LN_ADJ adds the LayerNorm adjoint of the product (the cotangent of the LayerNorm OUTPUT) to C.  ln_x: primal LayerNorm input [prows][N]
  // (row m belongs to primal row ((m / rows_per_sample) / epi_kps) * rows_per_sample + m % rows_per_sample), ln_gamma fp32 [N]
  const void* ln_x = nullptr; const float* ln_gamma = nullptr; float ln_eps = 1e-5f; void* C2 = nullptr;
  int fl = 0;                         // 16-bit flavour of the specialised kernels: 0 bf16, 1 f16 (filled in by launch_gemm)
  int order = 0;                      // block processing order per XCD: 0 A-major, 1 B-major (weight-heavy); filled in by launch_gemm
};
// *launches: kernels enqueued (1, or 2 with splitk_reduce_kernel).  `pending` != nullptr: if the launch is split over K and its epilogue is plain
// (one batch entry, alpha 1, no bias / row bias / accumulate, dense rows), the reduce kernel is NOT launched -- *pending receives the prepared
// arguments (pending->splitk > 1) and the caller either hands the slabs to a consumer that reduces them itself (SlabSrc, norm.hip) or calls
// launch_gemm_reduce; otherwise pending->splitk is set to 1.
struct GemmPlan { int kind, tile, splitk; };   // kind: 0 / 1 register-staged 64x64 / 128x128, 2 LDS ring (tile = its code), 3 halo-tile convolution; -1 error
int launch_gemm(int dtype, const GemmArgs& a, hipStream_t st, int* launches = nullptr, GemmArgs* pending = nullptr);
int launch_gemm_reduce(int dtype, const GemmArgs& pending, hipStream_t st);
GemmPlan gemm_plan(int dtype, const GemmArgs& a);   // host-only: the kernel launch_gemm picks, with its K split (clamped to the slab scratch)
int gemm_uses_big_tile(int dtype, const GemmArgs& a);
void gemm_debug_set(int tile, int splitk, int kch);
int gemm_kch(const GemmArgs& a);
int launch_gemm_dma(const GemmArgs& a, int tile, hipStream_t st);   // bf16, single operand pair, no split-K (gemm_dma.hip); tile 128 | 64 | 66 (64 with a 6-stage ring)
int launch_gemm_ring64(const GemmArgs& a, int tile, hipStream_t st);   // BK = 64 ring (gemm_ring64.hip); tile 512 | 513 | 514 | 515
bool gemm_p8_fits32(const GemmArgs& a);                                // its 32-bit element offsets reach every operand row
int launch_gemm_p8(const GemmArgs& a, int tile, hipStream_t st);      // 8-phase ping-pong loop (gemm_p8.hip); tile 530 = 256x256x64, 8 waves
int conv_halo_supported(const GemmArgs& a);                        // 3x3 stride-1 convolution in halo-tile form (gemm_halo.hip)
int launch_conv_halo(const GemmArgs& a, hipStream_t st);
int gemm_uses_halo(int dtype, const GemmArgs& a);
int gemm_epi_supported(int dtype, const GemmArgs& a);   // can this launch take GemmArgs::epi != EPI_PLAIN?
int gemm_uses_dma(int dtype, const GemmArgs& a);   // 0 = register-staged kernel, else the tile code for launch_gemm_dma
void gemm_debug_dma_auto(int on);
void gn_debug_deterministic(int on);
int gn_deterministic();
bool gemm_wres_supported(int dtype, const GemmArgs& a);                // weights-resident streaming kernel (gemm_wres.hip, tile code 540): K = 320, N % 320 == 0, plain epilogue
int launch_gemm_wres(const GemmArgs& a, hipStream_t st);
void gemm_debug_wres(int on);   // 1 (default): the dispatch may pick the weights-resident kernel; 0: round-5 dispatch, bitwise A/B
void gemm_debug_p8(int on);     // 1 (default): the dispatch may pick the 8-phase tile; 0: round-4 dispatch (rings / halo kernel), bitwise A/B
void gemm_debug_order(int o);   // -1 heuristic, 0 A-major, 1 B-major
int gemm_pick_splitk_dma(const GemmArgs& a, int tile);   // tuning overrides for micro-benchmarks (0 = heuristic)   // 1: 128x128 tile instantiation, 0: 64x64

// ---------------------------------------------------------------- normalisation
enum { MODE_PRIMAL = 0, MODE_TANGENT = 1, MODE_ADJOINT = 2 };
// A deferred split-K reduction handed to the normalisation kernel that consumes the product: its tangent / cotangent input d[row][c..] is
// sum_s slab[s][row][c..] (+ R[row][c..]), added in slab order and rounded to the engine dtype exactly as splitk_reduce_kernel would have
// stored it (bitwise the same values), optionally written to `store` when another op reads the buffer too.
struct SlabSrc {
  const float* slab = nullptr; int splitk = 0; long MN = 0; int N = 0;
  const void* R = nullptr; int ldr = 0;
  void* store = nullptr;
};
struct GNArgs {
  const void* x = nullptr;        // primal input  [Bp][HW][C]
  const void* d = nullptr;        // tangent dx or cotangent gz [NT][HW][C] (modes 1,2)
  void* y = nullptr;              // output (primal y / tangent dz / cotangent gx)
  const float* gamma = nullptr; const float* beta = nullptr;
  double* pstats = nullptr;       // [Bp][G][2]  primal (sum, sumsq) -> finalised to (mean, rstd) in place
  double* tstats = nullptr;       // [NT][G][2]  tangent / adjoint sums
  float* part = nullptr;          // per-block partial sums [n][blocks][G][2] (scratch shared by every GroupNorm launch of the stream)
  size_t part_bytes = 0;
  int* ticket = nullptr;          // [n] arrival counters, zero between launches
  int det = 1;                    // 1 (default): bitwise reproducible statistics (fixed-order reductions); 0: atomics (round-1 path, A/B only)
  int red = 0;                    // set by launch_groupnorm: the apply pass adds the statistics launch's per-block partials itself
  int Bp = 1, NT = 0, kps = 1;    // tangent j belongs to primal sample j / kps
  int HW = 0, C = 0, G = 32;
  float eps = 1e-5f;
  int silu = 0, accumulate = 0;
  SlabSrc src;                    // tangent / adjoint, one-launch kernel only: d comes from split-K slabs
};
int launch_groupnorm(int dtype, int mode, const GNArgs& a, hipStream_t st);
int groupnorm_launches(int dtype, int mode, const GNArgs& a);

struct LNArgs {
  const void* x = nullptr; const void* d = nullptr; void* y = nullptr;
  const float* gamma = nullptr; const float* beta = nullptr;
  int rows_per_sample = 0, Bp = 1, NT = 0, kps = 1, C = 0;
  float eps = 1e-5f;
  int accumulate = 0;
  SlabSrc src;                    // tangent / adjoint: d comes from split-K slabs
};
int launch_layernorm(int dtype, int mode, const LNArgs& a, hipStream_t st);

// ---------------------------------------------------------------- attention pieces
// rows: Z * Lq rows of length ld (valid columns < Lk, the rest are written as 0)
int launch_softmax_fwd(int dtype, void* S, long Z, int Lq, int Lk, int ld, int causal, hipStream_t st);   // causal: query i sees keys <= i
// dP = P o (dS - rowsum(P o dS)), in place on dS; P row index uses z_p = (z / Z2 / kps) * Z2 + z % Z2; optional D out
int launch_softmax_jvp(int dtype, const void* P, void* dS, float* D, long Z, int Z2, int kps, int Lq, int Lk, int ld,
                       hipStream_t st);
// gST[z][j][i] = PT[zp][j][i] * (gPT[z][j][i] - D[z][i])   (in place on gPT)
int launch_softmax_adjT(int dtype, const void* PT, void* gPT, const float* D, long Z, int Z2, int kps, int Lk, int Lq,
                        int ld, hipStream_t st);
// out[z][c][r] = in[z][r][c] for r < R, c < Ccols ; in row stride ldin, in batch strides (s1 over z1, s2 over z2)
int launch_transpose(int dtype, const void* in, void* out, int Z1, int Z2, long s1, long s2, int R, int Ccols, int ldin,
                     int ldout, long outZstride, hipStream_t st);

// fused (flash-style) tangent / adjoint self-attention, bf16, head dim 40 or 80 (attn_fused.hip)
struct FusedAttnArgs {
  const void *Q = nullptr, *K = nullptr, *V = nullptr, *O = nullptr, *KT = nullptr, *VT = nullptr, *QT = nullptr;
  const float* stats = nullptr;
  float* Drow = nullptr;                             // scratch [nt][H][L] floats for the shared-P key-major adjoint (row dots gO . O)
  const void *dQ = nullptr, *dK = nullptr, *dV = nullptr, *dVT = nullptr; void* dO = nullptr;
  const void *gO = nullptr, *gOT = nullptr; void *gQ = nullptr, *gK = nullptr, *gV = nullptr;
  int accQ = 0, accK = 0, accV = 0;
  int fl = 0;                                        // 16-bit flavour: 0 bf16, 1 f16
  int L = 0, C = 0, Co = 0, H = 0, d = 0, kps = 1;   // C: row stride of q/k/v (and their tangents / cotangents), Co: of o
  float scale = 1.f;
};
int fused_attention_supported(int dtype, int d, int L, int kv_const);
void attn_debug_shared(int bits);   // bit 1: shared-P key-major adjoint of the d = 40 layers (default 2; 0 = per-cotangent kernel)
// constant-K/V (cross) attention tangent / adjoint in one launch (attn_fused.hip): Y = c_out [P o (c_in X A^T - delta)] B
struct CrossAttnArgs {
  const void *Q = nullptr, *K = nullptr, *V = nullptr;   // primal q [B][L][C]; k, v [B][Lk][Ck] (column windows allowed)
  const void* BT = nullptr;                              // per-head transpose [B][H][d][Lkp] of V (tangent) or K (adjoint)
  const void* X = nullptr; void* Y = nullptr;            // dQ -> dO (tangent) or gO -> gQ (adjoint)
  int L = 0, Lk = 0, Lkp = 0, C = 0, Ck = 0, Cx = 0, Cy = 0, H = 0, d = 0, kps = 1, adjoint = 0, accumulate = 0, fl = 0;
  int primal = 0;                                        // 1: the forward pass itself, Y = softmax(scale Q K^T) V (X unused; BT may be null: built in LDS)
  float scale = 1.f;
};
int cross_attention_supported(int dtype, int d, int Lq, int Lk, int kv_const);
int launch_attn_cross(const CrossAttnArgs& f, int nt, hipStream_t st);
int launch_row_stats(int fl, const void* S, float* stats, long nrows, int Lk, int ld, hipStream_t st);
int attn_adj_route_bits(int d, int L, int kps, int nt);   // bit 0: multi-cotangent query-major kernel, bit 1: shared-probability key-major kernel
int attn_adj_launches(int d, int L, int kps, int nt);   // kernels launch_attn_adj_fused enqueues for such a layer (2 or 3)
int launch_attn_fwd_fused(const FusedAttnArgs& f, int batch, void* O, float* stats, hipStream_t st);   // primal O + row statistics
int launch_attn_jvp_fused(const FusedAttnArgs& f, int nt, hipStream_t st);
int launch_attn_adj_fused(const FusedAttnArgs& f, int nt, hipStream_t st);

// ---------------------------------------------------------------- elementwise
struct GegluArgs {
  const void* h = nullptr;     // primal [Bp*rows][2F]
  const void* d = nullptr;     // tangent dh [NT*rows][2F] or cotangent gy [NT*rows][F]
  void* y = nullptr;           // primal y [.. F] / tangent dy [.. F] / cotangent gh [.. 2F]
  int rows_per_sample = 0, Bp = 1, NT = 0, kps = 1, F = 0;
  int accumulate = 0;
  int stash = 1;               // primal: 1 = overwrite (a, g) in h by the factors (G1, G2) the tangent / adjoint passes read; 0 = forward only, h untouched
  int il = 0;                  // 0: h = [a | g] halves; 64: a / g interleaved in blocks of 64 columns (FF-in weight rows repacked, tape.py)
};
int launch_geglu(int dtype, int mode, const GegluArgs& a, hipStream_t st);
int launch_silu(int dtype, const void* x, void* y, long n, hipStream_t st);
int launch_quick_gelu(int dtype, const void* x, void* y, long n, hipStream_t st);   // x * sigmoid(1.702 x) (CLIP ViT-L text encoder MLP)
int launch_gelu(int dtype, const void* x, void* y, long n, hipStream_t st);         // exact erf GELU (OpenCLIP-H text encoder MLP of SD-2.x)
// out[b][c][t] = tok[ids[b][t]][c] + pos[t][c]  (fp32, the engine's NCHW boundary layout with H*W = L tokens); tables in `dtype`
int launch_embed_tokens(int dtype, const int* ids, const void* tok, const void* pos, float* out, int batch, int L, int C, int vocab, hipStream_t st);
int launch_axpy(int dtype, const void* x, void* y, long n, int accumulate, hipStream_t st);   // y (+)= x
// channel concat / split on [rows][C] tensors: copy src[rows][Cs] <-> dst[rows][Cd] column window at c0
int launch_copy_cols(int dtype, const void* src, int lds, int cs0, void* dst, int ldd, int cd0, long rows, int ncols,
                     int accumulate, hipStream_t st);
// fp32 NCHW [n][C][HW]  <->  T NHWC [n][HW][Cpad]
int launch_nchw_to_nhwc(int dtype, const float* src, void* dst, int n, int C, int HW, int Cpad, hipStream_t st);
int launch_nhwc_to_nchw(int dtype, const void* src, float* dst, int n, int C, int HW, int Cpad, hipStream_t st);
// 2x2 sum pooling of a cotangent (adjoint of nearest x2 upsampling): in [n][2H*2W][C] -> out [n][H*W][C]
int launch_pool2x2_sum(int dtype, const void* in, void* out, int n, int H, int W, int C, int accumulate, hipStream_t st);

// ---------------------------------------------------------------- re-orthonormalisation (fp32 in/out, fp64 Gram)
struct OrthArgs {
  const float* W = nullptr;      // [k][N]  = J^T J V_prev
  const float* Vprev = nullptr;  // [k][N]
  float* V = nullptr;            // [k][N]  right singular vectors of W, rows, descending
  float* s = nullptr;            // [k]     sqrt(singular values of W)   (reference: s.sqrt())
  float* conv = nullptr;         // [2]     {||V - Vprev||_2, max(|V-Vprev| - 1e-5|V|)}
  double* scratch = nullptr;     // >= orth_scratch_bytes(k, N)
  size_t scratch_bytes = 0;
  int k = 0; long N = 0;
  // samples of a batch in ONE set of launches (dpb_pullback_iterate): sample b at W + b*stride_w, Vprev / V + b*stride_v, s + b*stride_s,
  // conv + b*stride_conv (elements), scratch + b*scratch_stride (bytes, a multiple of 8, >= orth_scratch_bytes)
  int batch = 1; long stride_w = 0, stride_v = 0, stride_s = 0, stride_conv = 0; size_t scratch_stride = 0;
};
constexpr int ORTH_MAX_RANK = 128;   // largest pca_rank of the re-orthonormalisation (orth.hip: the k x k fp64 matrix of the eigen-solve lives in LDS)
int launch_orth(const OrthArgs& a, hipStream_t st);
size_t orth_scratch_bytes(int k, long N);

// ---------------------------------------------------------------- DDIM
// x_next = sqrt(a_next) * (x - e*sqrt(1-a_t))/sqrt(a_t) + sqrt(1-a_next) * e      (fp32, elementwise)
int launch_ddim_step(const float* x, const float* e, float* out, float* x0, long n, float a_t, float a_next, hipStream_t st);
// out = a*x + b*y + c*z (z may be null)
int launch_lincomb(const float* x, const float* y, const float* z, float* out, long n, float a, float b, float c, hipStream_t st);

}  // namespace dpb

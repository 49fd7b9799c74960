/* dpb.h -- C ABI of the MI355X-native pullback engine (libdpb.so, gfx950).
 *
 * Drop-in boundary for ONE path of enkeejunior1/Diffusion-Pullback: the power-iteration
 * low-rank SVD of the U-Net latent->feature Jacobian and the U-Net forwards of the DDIM loop.
 * The reference has no FFI (it is pure Python over diffusers); its boundary is Python method
 * injection onto the U-Net object (reference src/utils/utils.py:103-104, :326-337).  The host
 * shim diffusion_pullback_amd/ re-creates those methods on top of the entry points below;
 * INTEGRATION.md shows the binding a maintainer would add.  Each entry point cites the
 * reference code it replaces.
 *
 * Conventions: plain pointers and sizes, no torch types.  All tensor pointers are DEVICE
 * pointers unless marked host.  Every call enqueues work on the engine's stream
 * (dpb_engine_set_stream; default stream 0) and returns without synchronising unless stated.
 * Return value: 0 = success, nonzero = failure (message via dpb_last_error()).  One engine per
 * GPU per process; calls on one engine must not overlap (thread-compatible, not thread-safe).
 * The engine never allocates device memory: the caller sizes (dpb_engine_workspace_bytes) and
 * provides (dpb_engine_set_workspace) one zero-initialisable workspace, and owns the weights.
 */
#ifndef DPB_H
#define DPB_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPB_ABI_VERSION 1

enum { DPB_F32 = 0, DPB_BF16 = 1, DPB_F16 = 2 };   /* storage + MFMA input type; accumulation is always fp32 */

/* ---- network description: a tape of NHWC ops over numbered activation buffers ------------- */
enum {
  DPB_OP_CONV = 1,      /* conv KSxKS / 1x1 / Linear: out = W*in (+bias) (+rowbias[temb]) (+res)      */
  DPB_OP_GROUPNORM = 2, /* GroupNorm(G, eps) (+SiLU)                                                   */
  DPB_OP_LAYERNORM = 3, /* LayerNorm over channels                                                     */
  DPB_OP_ATTENTION = 4, /* multi-head softmax(q k^T d^-1/2) v ; in0=q in1=k in2=v                      */
  DPB_OP_GEGLU = 5,     /* [rows][2F] -> [rows][F] : a * gelu_erf(g).  dpb_primal OVERWRITES the input buffer
                           by the factors (gelu(g), a*gelu'(g)) its tangent / adjoint passes read: the input must
                           have no other consumer (checked at create) and is not a meaningful tap afterwards;
                           dpb_forward leaves it untouched                                              */
  DPB_OP_SILU = 6,      /* elementwise x*sigmoid(x); ip[0] = 1: quick-GELU x*sigmoid(1.702x), 2: erf GELU (primal only) */
  DPB_OP_CONCAT = 7     /* channel concat of in0, in1                                                  */
};
enum { DPB_GATHER_NONE = 0, DPB_GATHER_CONV = 1, DPB_GATHER_UPCONV = 3 };
enum { DPB_BUF_ACT = 0,     /* per-sample activation [rows][channels]                                   */
       DPB_BUF_SHARED = 1   /* one copy shared by the batch, independent of x (time-embedding path)     */ };

typedef struct dpb_buffer_desc {
  int32_t rows;       /* H*W or token count, per sample */
  int32_t channels;   /* stored channel count (multiple of 8) */
  int32_t kind;       /* DPB_BUF_* */
  int32_t valid_channels; /* un-padded channel count seen at the fp32 NCHW boundary; 0 = channels */
} dpb_buffer_desc;

typedef struct dpb_op_desc {
  int32_t kind;              /* DPB_OP_* */
  int32_t in0, in1, in2;     /* input buffer ids, -1 = none */
  int32_t out;               /* output buffer id */
  int32_t res;               /* CONV: buffer added to the output (residual / shortcut), -1 = none */
  int32_t rowbias;           /* CONV: DPB_BUF_SHARED buffer [1][Cout] added to every row (temb projection), -1 */
  int32_t ip[12];            /* CONV: H W Cin Ho Wo Cout KS stride pad gather rowbias_col (first column of this op's Cout-wide window in the rowbias buffer) ; GROUPNORM: G silu ;
                                ATTENTION: heads oq ok ov causal ; GEGLU: F interleave(0|64)                                               */
  float fp[4];               /* GROUPNORM/LAYERNORM: eps */
  const void* w[4];          /* CONV: w[0]=W [Cout][KS*KS*Cin] (engine dtype), w[1]=W^T [Cin][KS*KS*Cout]
                                (engine dtype, for the adjoint; may be NULL for ops never differentiated),
                                w[2]=bias fp32 [Cout] or NULL ; norms: w[0]=gamma fp32, w[1]=beta fp32   */
} dpb_op_desc;

typedef struct dpb_net_desc {
  int32_t dtype;             /* DPB_F32 | DPB_BF16 | DPB_F16 */
  int32_t max_batch;         /* max primal samples per call */
  int32_t max_tangents;      /* max total tangents/cotangents per call (k * samples) */
  int32_t n_buffers, n_ops;
  const dpb_buffer_desc* buffers;
  const dpb_op_desc* ops;
  int32_t x_buf;             /* input buffer (NHWC, channels padded); x_channels true channels */
  int32_t x_channels;
  int32_t temb_buf;          /* DPB_BUF_SHARED [1][temb_dim] sinusoid input, -1 = none */
  int32_t temb_dim;
  int32_t temb_flip_sin_to_cos;   /* 1: [cos|sin] (diffusers SD), 0: [sin|cos] (DDPM) */
  int32_t temb_half_minus_one;    /* 1: exponent denominator half_dim-1 (DDPM), 0: half_dim (SD) */
  int32_t ctx_buf;           /* per-sample conditioning buffer [ctx_len][ctx_dim], -1 = none */
} dpb_net_desc;

typedef struct dpb_engine dpb_engine;

const char* dpb_last_error(void);
int dpb_abi_version(void);

/* Build the executor for a network.  Weight pointers in `net` must stay valid for the engine's life.
 * Replaces: the diffusers U-Net module tree the reference walks in get_h (utils.py:438-527, :114-163). */
int dpb_engine_create(const dpb_net_desc* net, dpb_engine** out);
void dpb_engine_destroy(dpb_engine* e);
int dpb_engine_set_stream(dpb_engine* e, void* hip_stream);
size_t dpb_engine_workspace_bytes(const dpb_engine* e);
/* `ws` must be 256-byte aligned device memory of at least workspace_bytes; it is zero-filled here. */
int dpb_engine_set_workspace(dpb_engine* e, void* ws, size_t bytes);

/* Primal pass: run ops up to (and including) the producer of `upto_buf`, keeping every activation
 * and normalisation statistic resident for the tangent/adjoint passes.
 * x: fp32 NCHW [batch][x_channels][rows(x_buf)], t: timestep (host float, shared by the batch),
 * ctx: fp32 [batch][ctx_len][ctx_dim] or NULL.
 * Replaces: unet.get_h(...) / unet(x, t, encoder_hidden_states) (utils.py:438-527; edit.py:454-458). */
int dpb_primal(dpb_engine* e, const float* x, int batch, float t, const float* ctx, int upto_buf);
/* Forward only (the U-Net calls of the DDIM / guidance loop, edit.py:454-458, :484-502): the same pass without the tangent / adjoint
 * stash (no K^T / Q^T / P^T copies, GEGLU inputs left untouched), result copied to `out` as fp32 NCHW [batch][channels][rows(upto_buf)].
 * Invalidates the engine's primal state: dpb_jvp / dpb_vjp / dpb_pullback_iterate fail until the next dpb_primal. */
int dpb_forward(dpb_engine* e, const float* x, int batch, float t, const float* ctx, int upto_buf, int channels, float* out);
/* Copy a primal activation out as fp32 NCHW [batch][channels][rows] (first `channels` channels). */
int dpb_read_buffer(dpb_engine* e, int buf, int channels, float* out);

/* U = J V : forward-mode pass of `nt` tangents through the same kernels (nt = batch * tangents per
 * sample, tangent j belongs to sample j / (nt/batch)).  V fp32 NCHW [nt][x_channels][rows(x)],
 * U fp32 NCHW [nt][channels(tap)][rows(tap)].   Replaces: torch.func.jacfwd block, utils.py:766-775. */
int dpb_jvp(dpb_engine* e, int tap_buf, const float* V, int nt, float* U);
/* W = J^T U : adjoint pass wrt the input only.  Replaces: autograd.functional.jacobian, utils.py:790-797. */
int dpb_vjp(dpb_engine* e, int tap_buf, const float* U, int nt, float* W);

/* Thin SVD of W [k][N] (fp32, k <= 128): V rows = right singular vectors (descending), s = sqrt(singular values),
 * conv[0] = ||V - Vprev||_2, conv[1] = max(|V - Vprev| - 1e-5|V|).  scratch: device memory of >= dpb_orth_scratch_bytes(k, N) bytes (Gram
 * matrix and distance partials per block, added in block order: the result is bitwise reproducible).  V may alias Vprev (in place:
 * every element of Vprev is read by the thread that overwrites it, after the Gram pass has finished with it).
 * Replaces: torch.linalg.svd + dist/allclose inputs, utils.py:799-806.  Engine-independent. */
int dpb_orth(const float* W, const float* Vprev, float* V, float* s, float* conv, void* scratch, int k, int64_t N,
             void* hip_stream);
size_t dpb_orth_scratch_bytes(int k, int64_t N);   /* 0 for k outside [1, 128] */
/* The same with the caller's scratch size stated and validated.  CONTRACT CHANGE of round 3, called out here because dpb_orth cannot check it: the
 * scratch grew from 8 (3 k^2 + 2) bytes to dpb_orth_scratch_bytes(k, N) (per-block partials of the fixed-order reductions, up to ~129 k^2 + 512
 * doubles); a caller that still sizes it by the old rule gets out-of-bounds device writes from dpb_orth.  New callers bind THIS entry point
 * (the in-repo shim does): it fails with a message instead. */
int dpb_orth_checked(const float* W, const float* Vprev, float* V, float* s, float* conv, void* scratch, size_t scratch_bytes, int k, int64_t N,
                     void* hip_stream);

/* n_iters full power iterations with no host synchronisation: V <- orth(J^T J V), U = J V_prev, for all B samples
 * of the last dpb_primal together (independent bases, one shared weight stream; B*k <= max_tangents).
 * V [B][k][N_in] in/out, U [B][k][N_h] out, s [B][k] out, conv [B][2] out (of the last iteration).
 * Replaces: the loop body utils.py:756-808 (k <= 128), once per sample of the batch. */
int dpb_pullback_iterate(dpb_engine* e, int tap_buf, float* V, float* U, float* s, float* conv, int k, int n_iters);

/* DDIM update (utils.py:301-306 / :1220-1225, eta = 0) and the x-space-guidance axpy (edit.py:490, :501). */
int dpb_ddim_step(const float* x, const float* eps, float* out, float* x0, int64_t n, float alpha_t, float alpha_next,
                  void* hip_stream);
int dpb_lincomb(const float* x, const float* y, const float* z, float* out, int64_t n, float a, float b, float c,
                void* hip_stream);

/* Token + position embedding lookup of the CLIP text encoder behind pipe._encode_prompt (src/modules/edit.py:505-522):
 * out[b][c][t] = tok_table[ids[b][t]][c] + pos_table[t][c], fp32 in the engine's NCHW boundary layout (H*W = tokens),
 * ready for dpb_primal of a text-encoder tape.  Tables are [vocab][channels] / [tokens][channels] in `dtype`. */
int dpb_embed_tokens(const int32_t* ids, const void* tok_table, const void* pos_table, int dtype, float* out, int batch,
                     int tokens, int channels, int vocab, void* hip_stream);

/* Introspection used by tests / bench: number of kernel launches and algorithmic GEMM flops of the last pass. */
int dpb_engine_stats(const dpb_engine* e, int64_t* launches, double* gemm_flops, double* gemm_bytes);
/* Measurement aid (bench.py roofline leg, never on in a timed region): bracket every GEMM launch with HIP
 * events on the engine's stream; _read synchronises and sums the launches of one GEMM kernel kind: count, total
 * milliseconds, algorithmic flops.  kind: 0 register-staged 64x64, 1 register-staged 128x128, 2 BK=32 ring 128x128 /
 * 256x128, 3 BK=32 ring 64x64, 4 BK=64 ring with 128-column tiles (gemm_ring64.hip), 5 halo-tile 3x3 convolution (gemm_halo.hip), 6 BK=64 ring
 * with the 256x256 tile, 11 the 8-phase 256x256 tile (gemm_p8.hip: two wave groups alternating between load and matrix segments, counted vmcnt), 12 the
 * weights-resident streaming kernel of the K = 320 linear layers (gemm_wres.hip: weight slice in registers, activation rows streamed through an LDS ring);
 * attention (algorithmic flops = 2 / 5 / 7 L x L x d products per head and sample / tangent / cotangent): 7 flash forward,
 * 8 fused self-attention tangent, 9 fused self-attention adjoint (its query-major and key-major launches in ONE bracket; the CSV's `gather`
 * column holds the route: bit 0 multi-cotangent query-major kernel, bit 1 shared-probability key-major kernel), 10 one-launch cross-attention
 * tangent (gather 0) / adjoint (gather 1): 2 L x 77 x d products.  kind + 1000 returns the RAW bracket times of that kind (without the empty-bracket
 * correction of dpb_engine_profile_overhead, which clamps a bracket shorter than the correction to 0); any other kind (neither 0..12 nor 1000..1012) fails.
 * _dump writes one CSV line per recorded bracket. */
int dpb_engine_profile(dpb_engine* e, int enable);
int dpb_engine_profile_read(dpb_engine* e, int kind, int64_t* count, double* total_ms, double* flops);
int dpb_engine_profile_dump(dpb_engine* e, const char* csv_path);
/* The per-launch times of _read / _dump are event-bracket times minus a calibrated empty-bracket time (measured when profiling is switched on);
 * this returns that correction so that the raw bracket times can be reconstructed (raw = reported + overhead per launch). */
int dpb_engine_profile_overhead(const dpb_engine* e, double* bracket_overhead_ms);
/* Tuning overrides for micro-benchmarks and the bitwise kernel-equivalence tests (0 / -1 = heuristic): "gemm_tile"
 * (64, 128: register-staged; 129, 131, 133, 257, 65, 67: BK=32 rings; 512..518: BK=64 rings, 518 = 256x256 tile for plain-row operands, 530 = the 8-phase 256x256 tile (gemm_p8.hip), 540 = the weights-resident streaming kernel (gemm_wres.hip; products it does not take fall back to 515), 521 / 522 / 523 =
 * half tiles 64x128 (3 / 2 stages) and 128x64 for plain-row operands;
 * 600: halo-tile 3x3 convolution), "gemm_splitk" (n), "gemm_kch", "p8" (1, default: the dispatch may pick the 8-phase tile; 0 = the ring / halo dispatch of round 4), "wres" (1, default: K = 320 / N % 320 == 0 plain products of >= 8192 rows go to the weights-resident kernel; 0 = the round-5 dispatch; bitwise equal), "gemm_dma_auto" (0|1), "gemm_order" (-1 | 0 A-major | 1 B-major block
 * order per XCD), "gn_deterministic" (1, default: GroupNorm statistics of the two-pass kernels reduced in a fixed order -> bitwise
 * reproducible runs; 0 = the round-1 atomic statistics, A/B only), "graph_iterate" (0|1: dpb_pullback_iterate replays a captured hipGraph on a non-default stream;
 * measured equal to eager launches, default 0), "attn_shared" (2, default: shared-probability key-major adjoint of the head-dim-40
 * self-attention layers; 0 = the per-cotangent kernel of round 2; 6 = also route head dim 64 (SD-2.x) through the shared-probability adjoint kernels, measured
 * no faster there), "lazy_reduce" (1, default: a split-K product consumed by a one-launch GroupNorm or a
 * LayerNorm leaves its fp32 slabs to that kernel instead of running splitk_reduce_kernel; 0 = always reduce; bitwise the same results),
 * "ln_fuse" (0, default since round 6: separate launches; 1 = LayerNorm in the epilogue of the 320-wide products), "cross_primal" (1, default: the forward of a text-conditioned attention layer is
 * ONE launch; 0 = GEMM + softmax + transpose + GEMM), "geglu_fwd" (1, default: dpb_forward applies GEGLU in the epilogue of the unsplit FF-in products), "iter_alias" (1, default: inside dpb_pullback_iterate the tap's
 * tangent passes from the tangent to the adjoint pass on the device, U is written by the last iteration only; 0 = fp32 round trip through U every iteration; bitwise equal).
 * Environment, read once per process (tuning / ablation only): DPB_GEMM_OVERRIDE="MxNxK:gather=code/split,..." forces
 * kernel and split count per product shape; DPB_P8 (0: no 8-phase tile), DPB_WRES (0: no weights-resident kernel), DPB_WRES_MIN_M, DPB_TILE256, DPB_CONV_HALO, DPB_SPLITK_TARGET, DPB_GEMM_ORDER, DPB_GN_FUSED, DPB_GN_BLOCKS,
 * DPB_GN_DETERMINISTIC, DPB_LN_ROWS, DPB_LN_FUSE, DPB_LAZY_REDUCE, DPB_ATTN_WAVES, DPB_ATTN_MULTI, DPB_ATTN_SHARED, DPB_ATTN_XCD, DPB_FUSED_ATTN_MIN_L,
 * DPB_NO_FUSED_ATTN, DPB_NO_CROSS_ATTN, DPB_NO_GEGLU_FUSE switch individual kernels / fusions off or pick their variants; DPB_EIG_PAR (0: the one-wave cyclic
 * eigen-solve of rounds 1-5 for every k <= 56, 2: the round-robin one for every k), DPB_ORTH_BATCH (0: the samples of a batch re-orthonormalised one by one); DPB_GEMM_TRACE=1 prints every
 * product and synchronises after it (debugging). */
int dpb_debug_set(const char* key, int value);

/* Host-only (no GPU work): the launch plan the GEMM dispatch picks for a product C[M][N] = A[M][K] B[N][K]^T -- plain rows (conv_hw = 0) or
 * a 3x3 / stride 1 / pad 1 convolution on conv_hw x conv_hw images of conv_cin channels (K = 9 conv_cin) -- with `slab_bytes` of split-K
 * scratch.  kind: 0 / 1 register-staged 64x64 / 128x128 tile, 2 asynchronous LDS ring (tile = its code, see dpb_debug_set), 3 halo-tile
 * convolution; epilogue 0 plain, 1 / 2 fused GEGLU tangent / adjoint.  splitk x M x N x 4 bytes never exceeds slab_bytes.  Lets the
 * dispatch rules be tested on a machine without a GPU (tests/test_host_logic.py). */
int dpb_debug_gemm_plan(int dtype, int M, int N, int K, int conv_hw, int conv_cin, int epilogue, int64_t slab_bytes, int* kind, int* tile,
                        int* splitk);

#ifdef __cplusplus
}
#endif
#endif /* DPB_H */

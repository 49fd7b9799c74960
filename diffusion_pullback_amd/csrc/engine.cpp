// Tape executor + C ABI (include/dpb.h) of the MI355X pullback engine.
//
// A network is a tape of NHWC ops over numbered activation buffers.  Three passes run over a
// prefix of the tape:
//   primal  : batch B, keeps EVERY activation, GroupNorm statistic and attention matrix resident
//             in HBM (288 GB/GPU makes recomputation pointless).  x_t and t are fixed during a
//             power iteration, so this pass runs once per sample, not once per JVP/VJP as the
//             reference's autodiff does (src/utils/utils.py:766-797).
//   tangent : nt = B*k tangents pushed forward through the same kernels (linear ops: identical
//             GEMM with M scaled by k; nonlinear ops: closed-form tangent using the primal stash).
//   adjoint : nt cotangents pulled back in reverse tape order, input gradients only (no weight
//             gradients), fan-in handled by first-write/accumulate flags per buffer.
// Ops whose inputs do not depend on x (time-embedding MLP, text-conditioning K/V projections)
// are primal-only.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/dpb.h"
#include "kernels.h"

namespace dpb {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* last_error() { return g_err; }

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
static inline int round8(int v) { return (v + 7) / 8 * 8; }

struct Buf {
  int rows = 0, C = 0, kind = 0, Cv = 0;   // Cv: valid (un-padded) channels
  bool is_const = true;      // independent of x
  size_t p_off = 0, t_off = 0, g_off = 0, g_off0 = 0;   // g_off0: planned cotangent storage (g_off may be swapped within an adjoint pass)
};

struct AttnPlan {            // per attention op, persisted from the primal pass
  int heads = 0, d = 0, Lq = 0, Lk = 0, Lqp = 0, Lkp = 0;
  bool causal = false;                             // text-encoder attention: query i sees keys <= i (materialised path only)
  bool kv_const = false, fused = false, cross = false;   // cross: constant K/V, one-launch tangent / adjoint (attn_cross_kernel)
  int oq = 0, ok = 0, ov = 0;                      // column offsets of q / k / v inside their buffers (fused QKV projection)
  size_t P = 0, PT = 0, KT = 0, VT = 0, QT = 0, stats = 0;   // offsets
};

struct Op {
  dpb_op_desc d;
  bool is_const = true;
  int attn = -1;             // index into plans
  size_t pstats = 0, tstats = 0;   // GroupNorm stat slots (offsets)
  int geglu_next = -1;       // CONV (FF-in): the GEGLU op that is the only consumer of its output (interleaved layout) -> fused tangent epilogue
  int geglu_prev = -1;       // CONV (FF-out): the GEGLU op that produces its input                                   -> fused adjoint epilogue
  int ln_next = -1;          // CONV: the LayerNorm op that reads its 320-wide output  -> tangent: product + LayerNorm tangent in one launch (EPI_LN_TAN)
  int ln_prev = -1;          // CONV: the LayerNorm op whose output is its only input -> adjoint: product + LayerNorm adjoint in one launch (EPI_LN_ADJ)
};

}  // namespace dpb

using namespace dpb;

struct dpb_engine {
  int dtype = DT_F32, es = 4;
  int maxB = 1, maxT = 1;
  std::vector<Buf> bufs;
  std::vector<Op> ops;
  std::vector<AttnPlan> plans;
  std::vector<int> producer;        // buffer -> op index producing it (-1 for inputs)
  int x_buf = -1, x_channels = 0, temb_buf = -1, temb_dim = 0, temb_flip = 0, temb_hm1 = 0, ctx_buf = -1;
  hipStream_t stream = 0;
  char* ws = nullptr;
  size_t ws_bytes = 0;
  // arena offsets
  size_t pstats_off = 0, pstats_bytes = 0, tstats_off = 0, tstats_bytes = 0, gnpart = 0, gnpart_bytes = 0, gnticket = 0;
  size_t S1 = 0, S2 = 0, T1 = 0, Dv = 0, convtmp = 0, io_in = 0, io_out = 0, orth = 0, slab = 0, slab_bytes = 64u << 20, zeros = 0;
  size_t orth_stride = 0;                  // re-orthonormalisation scratch per sample of the batch
  size_t pbW = 0;                          // pullback loop fp32 staging of W = J^T J V
  size_t temb_host_stage = 0;
  int cur_batch = 0;
  std::vector<int> uses;            // buffer -> number of ops reading it (in0 / in1 / in2 / res)
  int cur_tap = -1;                 // tap buffer of the pass being run
  struct { bool on = false; GemmArgs a; } pend;   // a split-K product whose reduction is deferred to the normalisation op that consumes it
  bool fwd_only = false;            // dpb_forward: primal pass that keeps no tangent / adjoint stash (DDIM loop)
  std::vector<char> ginit;
  std::vector<char> skip;           // ops whose work a fused epilogue of another op has done in the current pass
  long n_launch = 0;
  double flops = 0, gbytes = 0;
  // captured power iteration (dpb_debug_set("graph_iterate", 1)), valid for one (tap, k, batch, buffer set)
  struct GraphKey {
    int tap, k, B; const void *V, *U, *s, *conv;
    bool operator==(const GraphKey& o) const { return tap == o.tap && k == o.k && B == o.B && V == o.V && U == o.U && s == o.s && conv == o.conv; }
  };
  hipGraphExec_t gexec = nullptr;
  GraphKey gkey{};
  long g_launches = 0; double g_flops = 0, g_bytes = 0;
  // optional per-launch timing of the GEMM kernel (bench.py roofline leg); off in the timed region
  bool profiling = false;
  struct Prof { hipEvent_t a, b; double flops; int big; int M, N, K, Z, gather; };
  std::vector<Prof> prof;
  float prof_overhead_ms = 0.f;     // elapsed time of an EMPTY event bracket on this stream (calibrated in dpb_engine_profile), subtracted per launch

  char* P(int b) const { return ws + bufs[b].p_off; }
  char* T(int b) const { return ws + bufs[b].t_off; }
  char* G(int b) const { return ws + bufs[b].g_off; }
};

namespace {

int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return -1;
}

void gemm_prep(dpb_engine* e, GemmArgs& a) {
  a.slab = (float*)(e->ws + e->slab);
  a.zeros = e->ws + e->zeros;
  a.slab_bytes = e->slab_bytes;
}

// split-K products whose consumer is a GroupNorm (one-launch kernel) / LayerNorm leave their slabs to it (A/B switch: DPB_LAZY_REDUCE=0,
// dpb_debug_set("lazy_reduce", 0): every split-K product runs its own reduce kernel; the results are bitwise the same)
int g_lazy_reduce = getenv("DPB_LAZY_REDUCE") ? atoi(getenv("DPB_LAZY_REDUCE")) : 1;
// LayerNorm tangent / adjoint in the epilogue of the neighbouring 320-wide product (A/B switch: DPB_LN_FUSE=0, dpb_debug_set("ln_fuse", 0))
// Default OFF since round 6: with write-through output stores the separate product + ln_rows launches are 0.5 % ahead on the headline (118.4 / 118.5 vs 117.7 /
// 117.9 it/s, same session, profiles/r06_switch_ab.txt) and equal on the 80-tangent pass -- the row-complete 128 x 320 tile is one block per CU, and what it saved was
// mostly the dirty write-back of the intermediate at the kernel boundary.  The fused tile stays in the library (tests/test_gpu_edge.py) behind the switch.
int g_ln_fuse = getenv("DPB_LN_FUSE") ? atoi(getenv("DPB_LN_FUSE")) : 0;
int g_ln_kmax = getenv("DPB_LN_FUSE_KMAX") ? atoi(getenv("DPB_LN_FUSE_KMAX")) : 1024;   // adjoint: longest K that still takes the row-complete LayerNorm tile (tuning switch)

int flush_pending(dpb_engine* e) {                 // the designated consumer did not come next: reduce the parked product the ordinary way
  if (!e->pend.on) return 0;
  e->pend.on = false;
  e->n_launch++;
  return launch_gemm_reduce(e->dtype, e->pend.a, e->stream);
}

// ---- measurement brackets (dpb_engine_profile): an event pair around one launch (or one fused group of launches) with its algorithmic flops.
// kind: 0..6 the GEMM kernel kinds (include/dpb.h), 7 flash attention forward, 8 fused attention tangent, 9 fused attention adjoint (query-major +
// key-major launches together; `gather` holds the route bits of attn_adj_route_bits), 10 one-launch cross-attention tangent / adjoint
int prof_open(dpb_engine* e, double flops, int kind, int M, int N, int K, int Z, int gather) {
  if (!e->profiling) return -1;
  dpb_engine::Prof p;
  p.flops = flops; p.big = kind; p.M = M; p.N = N; p.K = K; p.Z = Z; p.gather = gather;
  if (hipEventCreate(&p.a) != hipSuccess) return -1;
  if (hipEventCreate(&p.b) != hipSuccess) { (void)hipEventDestroy(p.a); return -1; }
  (void)hipEventRecord(p.a, e->stream);
  e->prof.push_back(p);
  return (int)e->prof.size() - 1;
}
void prof_close(dpb_engine* e, int idx) {
  if (idx >= 0) (void)hipEventRecord(e->prof[idx].b, e->stream);
}

// GEGLU in the epilogue of the forward pass's unsplit FF-in products (A/B switch: DPB_GEGLU_FWD=0, dpb_debug_set("geglu_fwd", 0))
int g_geglu_fwd = getenv("DPB_GEGLU_FWD") ? atoi(getenv("DPB_GEGLU_FWD")) : 1;
// one-launch forward of the text-conditioned attention layers (A/B switch: DPB_CROSS_PRIMAL=0, dpb_debug_set("cross_primal", 0): the materialised path)
int g_cross_primal = getenv("DPB_CROSS_PRIMAL") ? atoi(getenv("DPB_CROSS_PRIMAL")) : 1;
int gemm(dpb_engine* e, GemmArgs a, bool can_defer = false) {
  // a product parked by an EARLIER gemm() of the same op (run_op has flushed everything older) must be reduced before this one reuses the slabs
  if (e->pend.on)
    if (int r = flush_pending(e)) return r;
  gemm_prep(e, a);
  GemmArgs* pend = (can_defer && g_lazy_reduce) ? &e->pend.a : nullptr;
  const double kk = (double)a.K + (a.A2 ? a.K2 : 0);
  e->flops += 2.0 * a.M * (double)a.N * kk * a.Z1 * a.Z2;
  e->gbytes += ((double)a.M * a.K + (double)a.N * a.K + (double)a.M * a.N) * a.Z1 * a.Z2 * e->es;
  int nl = 1;                                   // kernels enqueued: the product itself (+ splitk_reduce_kernel for split-K launches)
  if (!e->profiling) { const int r = launch_gemm(e->dtype, a, e->stream, &nl, pend); e->n_launch += nl; e->pend.on = pend && pend->splitk > 1; return r; }
  int kind;
  { GemmArgs az = a; az.zeros = e->ws + e->zeros; const int dm = gemm_uses_dma(e->dtype, a); kind = gemm_uses_halo(e->dtype, az) ? 5 : dm == 540 ? 12 : dm == 530 ? 11 : dm == 518 ? 6 : dm >= 512 ? 4 : (dm == 128 || dm == 130 || dm == 132 || dm == 256) ? 2 : dm ? 3 : gemm_uses_big_tile(e->dtype, a); }   // 0: 64x64 register-staged, 2: 128x128 ring, 3: 64x64 ring, 4: BK=64 ring (128x128 tile), 5: halo-tile 3x3 convolution, 6: BK=64 ring, 256x256 tile, 11: 8-phase 256x256 tile (gemm_p8.hip)
  // the same bracket helpers as the attention launches (an event that cannot be created or recorded costs the bracket, never leaks its partner)
  const int pi = prof_open(e, 2.0 * a.M * (double)a.N * kk * a.Z1 * a.Z2, kind, a.M, a.N, a.K, a.Z1 * a.Z2, a.gather);
  int r = launch_gemm(e->dtype, a, e->stream, &nl, pend);
  e->n_launch += nl;
  e->pend.on = pend && pend->splitk > 1;
  prof_close(e, pi);
  return r;
}

// ------------------------------------------------------------------ CONV
// mode 0 primal (n=B), 1 tangent (n=nt)
int conv_fwd(dpb_engine* e, const Op& op, int mode, int n) {
  const dpb_op_desc& d = op.d;
  const int H = d.ip[0], W = d.ip[1], Cin = d.ip[2], Ho = d.ip[3], Wo = d.ip[4], Cout = d.ip[5], KS = d.ip[6];
  const Buf& bi = e->bufs[d.in0];
  const Buf& bo = e->bufs[d.out];
  if (mode == 1 && bi.is_const) {     // only the residual carries a tangent
    e->n_launch++;
    return launch_axpy(e->dtype, e->T(d.res), e->T(d.out), (long)n * bo.rows * bo.C, 0, e->stream);
  }
  GemmArgs g;
  const bool shared_out = bo.kind == DPB_BUF_SHARED;
  const int ns = shared_out ? 1 : n;
  g.A = mode == 0 ? e->P(d.in0) : e->T(d.in0);
  g.B = d.w[0];
  g.C = mode == 0 ? e->P(d.out) : e->T(d.out);
  g.N = Cout;
  g.K = KS * KS * Cin;
  g.ldb = g.K;
  g.ldc = bo.C;
  g.lda = bi.C;
  g.gather = d.ip[9];
  if (g.gather == GATHER_NONE) {
    g.M = ns * bi.rows;
  } else {
    g.M = ns * Ho * Wo;
    g.H = H; g.W = W; g.Cin = Cin; g.Ho = Ho; g.Wo = Wo; g.KS = KS; g.stride = d.ip[7]; g.pad = d.ip[8];
  }
  if (mode == 0) {
    g.bias = (const float*)d.w[2];
    if (d.rowbias >= 0) {
      g.rowbias = e->P(d.rowbias) + (size_t)d.ip[10] * e->es;      // ip[10]: column window of a net-wide fused projection (tape.py shared_begin)
      g.rows_per_sample = bo.rows;
      g.rowbias_div = 1 << 30;
    }
  }
  if (d.res >= 0 && (mode == 0 || !e->bufs[d.res].is_const)) {
    g.R = mode == 0 ? e->P(d.res) : e->T(d.res);
    g.ldr = e->bufs[d.res].C;
  }
  if (mode == 1 && op.ln_next >= 0 && g_ln_fuse) {   // the LayerNorm that reads this product's output: its tangent leaves the same launch
    const dpb_op_desc& ld = e->ops[op.ln_next].d;
    GemmArgs f = g;
    f.epi = EPI_LN_TAN; f.C2 = e->T(ld.out); f.ln_x = e->P(ld.in0); f.ln_gamma = (const float*)ld.w[0]; f.ln_eps = ld.fp[0];
    f.rows_per_sample = bo.rows; f.epi_kps = n / e->cur_batch;
    gemm_prep(e, f);
    if (gemm_epi_supported(e->dtype, f)) {
      e->skip[op.ln_next] = 1;
      return gemm(e, f);
    }
  }
  if (mode == 1 && op.geglu_next >= 0) {   // FF-in tangent: GEGLU's tangent in the epilogue, dh [rows][2F] is never written
    const dpb_op_desc& gd = e->ops[op.geglu_next].d;
    GemmArgs f = g;
    f.epi = EPI_GEGLU_TAN; f.hprim = e->P(d.out); f.epi_kps = n / e->cur_batch; f.rows_per_sample = bo.rows;
    f.C = e->T(gd.out); f.ldc = e->bufs[gd.out].C;
    gemm_prep(e, f);
    if (gemm_epi_supported(e->dtype, f)) {
      e->skip[op.geglu_next] = 1;
      return gemm(e, f);
    }
  }
  if (mode == 0 && e->fwd_only && op.geglu_next >= 0 && g_geglu_fwd && !shared_out && d.out != e->cur_tap) {   // (a pass that stops AT h needs h written)
    // forward only (dpb_forward): an FF-in product that runs unsplit anyway applies GEGLU in its epilogue -- h [rows][2F] is neither written nor
    // re-read (84 MB per 64x64-level layer at batch 2), one launch less; bitwise the separate product + GEGLU kernel
    const dpb_op_desc& gd = e->ops[op.geglu_next].d;
    GemmArgs f = g;
    f.epi = EPI_GEGLU_FWD; f.C = e->P(gd.out); f.ldc = e->bufs[gd.out].C; f.rowbias = nullptr; f.R = nullptr;
    gemm_prep(e, f);
    GemmArgs probe = g; gemm_prep(e, probe);
    if (!g.rowbias && !g.R && gemm_epi_supported(e->dtype, f) && gemm_plan(e->dtype, probe).splitk == 1) {
      e->skip[op.geglu_next] = 1;
      return gemm(e, f);
    }
  }
  // tangent: a following one-launch GroupNorm / LayerNorm may add the split-K slabs itself.  (Primal / forward-only products too were tried in
  // round 3 -- bias and time-embedding row added by the consumer: 599 -> 573 launches per B = 2 forward but 7.47 -> 7.55 ms: the slabs are 16x
  // the bytes of the 16-bit tensor and the primal consumers have nothing to hide them under.)
  return gemm(e, g, mode == 1 && !shared_out);
}

int conv_adj(dpb_engine* e, const Op& op, int n) {
  const dpb_op_desc& d = op.d;
  const int H = d.ip[0], W = d.ip[1], Cin = d.ip[2], Ho = d.ip[3], Wo = d.ip[4], KS = d.ip[6];
  const Buf& bi = e->bufs[d.in0];
  const Buf& bo = e->bufs[d.out];
  const int Cout = bo.C;              // padded channel count of the cotangent (w[1] is [Cin][KS*KS*CoutPadded])
  if (!bi.is_const) {
    if (!d.w[1]) return fail("conv op has no transposed weight (w[1]) but its adjoint is needed");
    GemmArgs g;
    g.A = e->G(d.out);
    g.lda = bo.C;
    g.B = d.w[1];
    g.N = Cin;
    g.K = KS * KS * Cout;
    g.ldb = g.K;
    g.ldc = bi.C;
    const int gather = d.ip[9];
    // the LayerNorm that wrote this product's input: its adjoint in the epilogue (K <= 1024: beyond -- the FF-in adjoint, K = 8 C -- the
    // row-complete tile's one block per CU loses more in the K loop than the fusion saves: g_ln_fuse bit 1 forces it for A/Bs)
    if (gather == GATHER_NONE && op.ln_prev >= 0 && g_ln_fuse && e->uses[d.in0] == 1 && (g.K <= g_ln_kmax || (g_ln_fuse & 2))) {
      const dpb_op_desc& ld = e->ops[op.ln_prev].d;
      GemmArgs f = g;
      f.M = n * bi.rows;
      f.epi = EPI_LN_ADJ; f.C = e->G(ld.in0); f.ldc = e->bufs[ld.in0].C; f.accumulate = e->ginit[ld.in0];
      f.ln_x = e->P(ld.in0); f.ln_gamma = (const float*)ld.w[0]; f.ln_eps = ld.fp[0];
      f.rows_per_sample = bi.rows; f.epi_kps = n / e->cur_batch;
      gemm_prep(e, f);
      if (gemm_epi_supported(e->dtype, f)) {
        if (int r = gemm(e, f)) return r;
        e->ginit[ld.in0] = 1;               // G(d.in0) stays unwritten: the LayerNorm op sees ginit == 0 and is skipped
        goto residual;
      }
    }
    if (gather == GATHER_NONE && op.geglu_prev >= 0) {   // FF-out adjoint: GEGLU's adjoint in the epilogue, gy [rows][F] is never written
      const dpb_op_desc& gd = e->ops[op.geglu_prev].d;
      GemmArgs f = g;
      f.M = n * bi.rows;
      f.epi = EPI_GEGLU_ADJ; f.hprim = e->P(gd.in0); f.epi_kps = n / e->cur_batch; f.rows_per_sample = bi.rows;
      f.C = e->G(gd.in0); f.ldc = e->bufs[gd.in0].C; f.accumulate = e->ginit[gd.in0];
      gemm_prep(e, f);
      if (gemm_epi_supported(e->dtype, f)) {
        if (int r = gemm(e, f)) return r;
        e->ginit[gd.in0] = 1;               // G(d.in0) stays unwritten: the GEGLU op sees ginit == 0 and is skipped
        goto residual;
      }
    }
    const bool lone = e->uses[d.in0] == 1 && d.in0 != e->x_buf;   // the cotangent has this one contribution: its producer's adjoint may add the slabs
    if (gather == GATHER_NONE) {
      g.M = n * bi.rows;
      g.C = e->G(d.in0);
      g.accumulate = e->ginit[d.in0];
      if (int r = gemm(e, g, lone)) return r;
    } else if (gather == GATHER_CONV) {
      g.gather = GATHER_CONVT;
      g.M = n * H * W;
      g.H = Ho; g.W = Wo; g.Cin = Cout; g.Ho = H; g.Wo = W; g.KS = KS; g.stride = d.ip[7]; g.pad = d.ip[8];
      g.C = e->G(d.in0);
      g.accumulate = e->ginit[d.in0];
      if (int r = gemm(e, g, lone)) return r;
    } else {   // UPCONV: adjoint conv at the upsampled resolution, then 2x2 sum pooling
      g.gather = GATHER_CONVT;
      g.M = n * Ho * Wo;
      g.H = Ho; g.W = Wo; g.Cin = Cout; g.Ho = Ho; g.Wo = Wo; g.KS = KS; g.stride = 1; g.pad = 1;
      g.C = e->ws + e->convtmp;
      if (int r = gemm(e, g)) return r;
      e->n_launch++;
      if (int r = launch_pool2x2_sum(e->dtype, e->ws + e->convtmp, e->G(d.in0), n, H, W, bi.C, e->ginit[d.in0], e->stream)) return r;
    }
    e->ginit[d.in0] = 1;
  }
residual:
  if (d.res >= 0 && !e->bufs[d.res].is_const) {
    Buf& br = e->bufs[d.res];
    if (!e->ginit[d.res] && br.rows == bo.rows && br.C == bo.C && br.kind == bo.kind) {
      // first cotangent of the residual stream: G(out) is dead once its producer (this op) has run, so hand its storage
      // over instead of copying it (38 copies of up to 13 MB per adjoint pass on SD-1.5); dpb_vjp restores the plan
      std::swap(br.g_off, e->bufs[d.out].g_off);
    } else {
      e->n_launch++;
      if (int r = launch_axpy(e->dtype, e->G(d.out), e->G(d.res), (long)n * bo.rows * bo.C, e->ginit[d.res], e->stream)) return r;
    }
    e->ginit[d.res] = 1;
  }
  return 0;
}

// ------------------------------------------------------------------ norms / elementwise
// The normalisation op about to run reads `d`: if that is the output of the parked split-K product, hand it the slabs (run_op has already made
// sure that nothing else is parked).  `store`: the reduced tensor has other readers (residual stream, tap) and must be written as well.
void take_pending(dpb_engine* e, SlabSrc& src, const void* d, bool store) {
  if (!e->pend.on || e->pend.a.C != d) return;
  const GemmArgs& g = e->pend.a;
  src.slab = g.slab; src.splitk = g.splitk; src.MN = (long)g.M * g.N; src.N = g.N;
  src.R = g.R; src.ldr = g.ldr;
  src.store = store ? g.C : nullptr;
  e->pend.on = false;
}

// does `op`, about to run in `mode`, consume the parked product itself?  (GroupNorm: only the one-launch kernel takes slabs.)
bool consumes_pending(dpb_engine* e, const Op& op, int mode) {
  if (mode == MODE_PRIMAL || (op.d.kind != DPB_OP_GROUPNORM && op.d.kind != DPB_OP_LAYERNORM)) return false;
  const void* d = mode == MODE_TANGENT ? (const void*)e->T(op.d.in0) : (const void*)e->G(op.d.out);
  if (d != e->pend.a.C) return false;
  const Buf& bi = e->bufs[op.d.in0];
  if (e->pend.a.N != bi.C) return false;
  if (op.d.kind == DPB_OP_GROUPNORM) {
    GNArgs a; a.HW = bi.rows; a.C = bi.C; a.G = op.d.ip[0]; a.NT = 1;
    return groupnorm_launches(e->dtype, mode, a) == 1;
  }
  return true;
}

int gn_run(dpb_engine* e, const Op& op, int mode, int n) {
  const dpb_op_desc& d = op.d;
  const Buf& bi = e->bufs[d.in0];
  GNArgs a;
  a.x = e->P(d.in0);
  a.gamma = (const float*)d.w[0];
  a.beta = (const float*)d.w[1];
  a.pstats = (double*)(e->ws + op.pstats);
  a.tstats = (double*)(e->ws + op.tstats);
  a.part = (float*)(e->ws + e->gnpart); a.part_bytes = e->gnpart_bytes; a.ticket = (int*)(e->ws + e->gnticket);
  a.det = gn_deterministic();
  a.HW = bi.rows; a.C = bi.C; a.G = d.ip[0]; a.silu = d.ip[1]; a.eps = d.fp[0];
  a.Bp = e->cur_batch;
  if (mode == MODE_PRIMAL) {
    a.Bp = n;
    a.y = e->P(d.out);
  } else {
    a.NT = n;
    a.kps = n / e->cur_batch;
    if (mode == MODE_TANGENT) {
      a.d = e->T(d.in0);
      a.y = e->T(d.out);
    } else {
      a.d = e->G(d.out);
      a.y = e->G(d.in0);
      a.accumulate = e->ginit[d.in0];
      e->ginit[d.in0] = 1;
    }
    take_pending(e, a.src, a.d, mode == MODE_TANGENT && (e->uses[d.in0] > 1 || d.in0 == e->cur_tap));
  }
  e->n_launch += groupnorm_launches(e->dtype, mode, a);
  return launch_groupnorm(e->dtype, mode, a, e->stream);
}

int ln_run(dpb_engine* e, const Op& op, int mode, int n) {
  const dpb_op_desc& d = op.d;
  const Buf& bi = e->bufs[d.in0];
  LNArgs a;
  a.x = e->P(d.in0);
  a.gamma = (const float*)d.w[0];
  a.beta = (const float*)d.w[1];
  a.rows_per_sample = bi.rows; a.C = bi.C; a.eps = d.fp[0];
  a.Bp = e->cur_batch;
  if (mode == MODE_PRIMAL) {
    a.Bp = n;
    a.y = e->P(d.out);
  } else {
    a.NT = n;
    a.kps = n / e->cur_batch;
    if (mode == MODE_TANGENT) {
      a.d = e->T(d.in0);
      a.y = e->T(d.out);
    } else {
      a.d = e->G(d.out);
      a.y = e->G(d.in0);
      a.accumulate = e->ginit[d.in0];
      e->ginit[d.in0] = 1;
    }
    take_pending(e, a.src, a.d, mode == MODE_TANGENT && (e->uses[d.in0] > 1 || d.in0 == e->cur_tap));
  }
  e->n_launch++;
  return launch_layernorm(e->dtype, mode, a, e->stream);
}

int geglu_run(dpb_engine* e, const Op& op, int mode, int n) {
  const dpb_op_desc& d = op.d;
  const Buf& bi = e->bufs[d.in0];
  GegluArgs a;
  a.h = e->P(d.in0);
  a.rows_per_sample = bi.rows; a.F = bi.C / 2; a.il = d.ip[1];
  a.Bp = e->cur_batch;
  if (mode == MODE_PRIMAL) {
    a.Bp = n;
    a.y = e->P(d.out);
    a.stash = !e->fwd_only;
  } else {
    a.NT = n;
    a.kps = n / e->cur_batch;
    if (mode == MODE_TANGENT) {
      a.d = e->T(d.in0);
      a.y = e->T(d.out);
    } else {
      a.d = e->G(d.out);
      a.y = e->G(d.in0);
      a.accumulate = e->ginit[d.in0];
      e->ginit[d.in0] = 1;
    }
  }
  e->n_launch++;
  return launch_geglu(e->dtype, mode, a, e->stream);
}

int concat_run(dpb_engine* e, const Op& op, int mode, int n) {
  const dpb_op_desc& d = op.d;
  const Buf& b0 = e->bufs[d.in0];
  const Buf& b1 = e->bufs[d.in1];
  const Buf& bo = e->bufs[d.out];
  const long rows = (long)n * bo.rows;
  e->n_launch += 2;
  if (mode == MODE_ADJOINT) {
    if (!b0.is_const) {
      if (int r = launch_copy_cols(e->dtype, e->G(d.out), bo.C, 0, e->G(d.in0), b0.C, 0, rows, b0.C, e->ginit[d.in0], e->stream)) return r;
      e->ginit[d.in0] = 1;
    }
    if (!b1.is_const) {
      if (int r = launch_copy_cols(e->dtype, e->G(d.out), bo.C, b0.C, e->G(d.in1), b1.C, 0, rows, b1.C, e->ginit[d.in1], e->stream)) return r;
      e->ginit[d.in1] = 1;
    }
    return 0;
  }
  if (mode == MODE_TANGENT && (b0.is_const || b1.is_const)) return fail("concat of x-independent and x-dependent buffers is unsupported");
  char* o = mode == MODE_PRIMAL ? e->P(d.out) : e->T(d.out);
  const char* i0 = mode == MODE_PRIMAL ? e->P(d.in0) : e->T(d.in0);
  const char* i1 = mode == MODE_PRIMAL ? e->P(d.in1) : e->T(d.in1);
  if (int r = launch_copy_cols(e->dtype, i0, b0.C, 0, o, bo.C, 0, rows, b0.C, 0, e->stream)) return r;
  return launch_copy_cols(e->dtype, i1, b1.C, 0, o, bo.C, b0.C, rows, b1.C, 0, e->stream);
}

// ------------------------------------------------------------------ attention (materialised scores)
// Operand addressing: q / k / v may be column windows of wider buffers (fused QKV projection: one [rows][3C] buffer),
// so every operand has its own row stride (ld*) and column offset (o*); the output buffer is [rows][C].
struct AttnPtrs {
  const char *Q, *K, *V; char* O;
  int ldq, ldk, ldv, ldo;
};
static AttnPtrs attn_ptrs(dpb_engine* e, const dpb_op_desc& d, const AttnPlan& p, int which /*0 primal, 1 tangent, 2 cotangent*/) {
  AttnPtrs a;
  auto base = [&](int b) { return which == 0 ? e->P(b) : which == 1 ? e->T(b) : e->G(b); };
  a.ldq = e->bufs[d.in0].C; a.ldk = e->bufs[d.in1].C; a.ldv = e->bufs[d.in2].C; a.ldo = e->bufs[d.out].C;
  a.Q = base(d.in0) + (size_t)p.oq * e->es;
  a.K = (which == 0 || !p.kv_const) ? base(d.in1) + (size_t)p.ok * e->es : nullptr;
  a.V = (which == 0 || !p.kv_const) ? base(d.in2) + (size_t)p.ov * e->es : nullptr;
  a.O = base(d.out);
  return a;
}

static void fill_fused(dpb_engine* e, const AttnPlan& p, const AttnPtrs& x, FusedAttnArgs& f, int kps, float scale);

int attn_primal(dpb_engine* e, const Op& op, int B) {
  const dpb_op_desc& d = op.d;
  const AttnPlan& p = e->plans[op.attn];
  const int H = p.heads;
  const float scale = 1.f / sqrtf((float)p.d);
  char* ws = e->ws;
  const AttnPtrs x = attn_ptrs(e, d, p, 0);
  if (p.fused) {   // flash forward: O and the row statistics, no L x L object; the kernels build transposed operand fragments with
                   // LDS transpose reads from the row tiles, so no per-head transposed copies are kept either
    e->n_launch += 1;
    FusedAttnArgs f;
    fill_fused(e, p, x, f, 1, scale);
    const double fl = 2.0 * p.Lq * (double)p.Lk * p.d * 2 * B * H;
    e->flops += fl;
    const int pi = prof_open(e, fl, 7, p.Lq, p.Lk, p.d, B * H, 0);
    const int r = launch_attn_fwd_fused(f, B, x.O, (float*)(ws + p.stats), e->stream);
    prof_close(e, pi);
    return r;
  }
  if (p.cross && g_cross_primal) {
    // text-conditioned layers: O = softmax(scale Q K^T) V in ONE launch (the 77 keys fit one masked tile; V^T is built in LDS), instead of
    // GEMM + softmax + transpose + GEMM; the tangent / adjoint kernels of the pullback passes still read the per-head V^T / K^T copies
    CrossAttnArgs f;
    f.Q = x.Q; f.K = x.K; f.V = x.V; f.BT = nullptr; f.X = nullptr; f.Y = x.O;
    f.L = p.Lq; f.Lk = p.Lk; f.Lkp = p.Lkp; f.C = x.ldq; f.Ck = x.ldk; f.Cx = x.ldq; f.Cy = x.ldo;
    f.H = H; f.d = p.d; f.kps = 1; f.primal = 1; f.scale = scale; f.fl = e->dtype == DT_F16;
    e->n_launch++;
    const double fl = 2.0 * p.Lq * (double)p.Lk * p.d * 2 * B * H;
    e->flops += fl;
    const int pi = prof_open(e, fl, 10, p.Lq, p.Lk, p.d, B * H, 2);
    const int r = launch_attn_cross(f, B, e->stream);
    prof_close(e, pi);
    if (r || e->fwd_only) return r;
    e->n_launch += 2;
    if (int r2 = launch_transpose(e->dtype, x.V, ws + p.VT, B, H, (long)p.Lk * x.ldv, p.d, p.Lk, p.d, x.ldv, p.Lkp, (long)p.d * p.Lkp, e->stream)) return r2;
    return launch_transpose(e->dtype, x.K, ws + p.KT, B, H, (long)p.Lk * x.ldk, p.d, p.Lk, p.d, x.ldk, p.Lkp, (long)p.d * p.Lkp, e->stream);
  }
  GemmArgs g;   // S = scale * Q K^T
  g.A = x.Q; g.lda = x.ldq; g.sA1 = (long)p.Lq * x.ldq; g.sA2 = p.d;
  g.B = x.K; g.ldb = x.ldk; g.sB1 = (long)p.Lk * x.ldk; g.sB2 = p.d;
  g.C = ws + p.P; g.ldc = p.Lkp; g.sC1 = (long)H * p.Lq * p.Lkp; g.sC2 = (long)p.Lq * p.Lkp;
  g.M = p.Lq; g.N = p.Lk; g.K = p.d; g.Z1 = B; g.Z2 = H; g.alpha = scale;
  if (int r = gemm(e, g)) return r;
  e->n_launch += 3;
  if (p.fused) {
    e->n_launch++;
    if (int r = launch_row_stats(e->dtype == DT_F16, ws + p.P, (float*)(ws + p.stats), (long)B * H * p.Lq, p.Lk, p.Lkp, e->stream)) return r;
  }
  if (int r = launch_softmax_fwd(e->dtype, ws + p.P, (long)B * H, p.Lq, p.Lk, p.Lkp, p.causal, e->stream)) return r;
  // V^T, K^T per head ([d][Lkp], zero padded)
  if (int r = launch_transpose(e->dtype, x.V, ws + p.VT, B, H, (long)p.Lk * x.ldv, p.d, p.Lk, p.d, x.ldv, p.Lkp, (long)p.d * p.Lkp, e->stream)) return r;
  if (e->fwd_only) e->n_launch--;   // K^T serves the adjoint only
  else if (int r = launch_transpose(e->dtype, x.K, ws + p.KT, B, H, (long)p.Lk * x.ldk, p.d, p.Lk, p.d, x.ldk, p.Lkp, (long)p.d * p.Lkp, e->stream)) return r;
  GemmArgs o;   // O = P V
  o.A = ws + p.P; o.lda = p.Lkp; o.sA1 = (long)H * p.Lq * p.Lkp; o.sA2 = (long)p.Lq * p.Lkp;
  o.B = ws + p.VT; o.ldb = p.Lkp; o.sB1 = (long)H * p.d * p.Lkp; o.sB2 = (long)p.d * p.Lkp;
  o.C = x.O; o.ldc = x.ldo; o.sC1 = (long)p.Lq * x.ldo; o.sC2 = p.d;
  o.M = p.Lq; o.N = p.d; o.K = p.Lkp; o.Z1 = B; o.Z2 = H;
  if (int r = gemm(e, o)) return r;
  if (!p.kv_const && !e->fwd_only) {
    e->n_launch += 2;
    if (!p.fused)
      if (int r = launch_transpose(e->dtype, ws + p.P, ws + p.PT, B * H, 1, (long)p.Lq * p.Lkp, 0, p.Lq, p.Lk, p.Lkp, p.Lqp, (long)p.Lk * p.Lqp, e->stream)) return r;
    if (int r = launch_transpose(e->dtype, x.Q, ws + p.QT, B, H, (long)p.Lq * x.ldq, p.d, p.Lq, p.d, x.ldq, p.Lqp, (long)p.d * p.Lqp, e->stream)) return r;
  }
  return 0;
}

static void fill_fused(dpb_engine* e, const AttnPlan& p, const AttnPtrs& x, FusedAttnArgs& f, int kps, float scale) {
  char* ws = e->ws;
  f.Q = x.Q; f.K = x.K; f.V = x.V; f.O = x.O; f.VT = ws + p.VT; f.KT = ws + p.KT; f.QT = ws + p.QT;
  f.stats = (const float*)(ws + p.stats);
  f.L = p.Lq; f.C = x.ldq; f.Co = x.ldo; f.H = p.heads; f.d = p.d; f.kps = kps; f.scale = scale; f.fl = e->dtype == DT_F16;
}

int attn_tangent(dpb_engine* e, const Op& op, int nt) {
  const dpb_op_desc& d = op.d;
  const AttnPlan& p = e->plans[op.attn];
  const int H = p.heads, kps = nt / e->cur_batch;
  const float scale = 1.f / sqrtf((float)p.d);
  char* ws = e->ws;
  char* S1 = ws + e->S1;
  const AttnPtrs x = attn_ptrs(e, d, p, 0), t = attn_ptrs(e, d, p, 1);
  if (p.fused) {
    e->n_launch += 1;
    FusedAttnArgs f;
    fill_fused(e, p, x, f, kps, scale);
    f.dQ = t.Q; f.dK = t.K; f.dV = t.V; f.dO = t.O;
    const double fl = 2.0 * p.Lq * (double)p.Lk * p.d * 5 * nt * H;     // dS (2 products), dP V, P dV + the recomputed scores: 5 L x L x d products
    e->flops += fl;
    const int pi = prof_open(e, fl, 8, p.Lq, p.Lk, p.d, nt * H, 0);
    const int r = launch_attn_jvp_fused(f, nt, e->stream);
    prof_close(e, pi);
    return r;
  }
  if (p.cross) {   // constant K/V: dO = [P o (scale dQ K^T - delta)] V in one launch
    CrossAttnArgs f;
    f.Q = x.Q; f.K = x.K; f.V = x.V; f.BT = ws + p.VT; f.X = t.Q; f.Y = t.O;
    f.L = p.Lq; f.Lk = p.Lk; f.Lkp = p.Lkp; f.C = x.ldq; f.Ck = x.ldk; f.Cx = t.ldq; f.Cy = t.ldo;
    f.H = H; f.d = p.d; f.kps = kps; f.adjoint = 0; f.scale = scale; f.fl = e->dtype == DT_F16;
    e->n_launch++;
    const double fl = 2.0 * p.Lq * (double)p.Lk * p.d * 2 * nt * H;
    e->flops += fl;
    const int pi = prof_open(e, fl, 10, p.Lq, p.Lk, p.d, nt * H, 0);
    const int r = launch_attn_cross(f, nt, e->stream);
    prof_close(e, pi);
    return r;
  }
  GemmArgs g;   // dS = scale * dQ K^T
  g.A = t.Q; g.lda = t.ldq; g.sA1 = (long)p.Lq * t.ldq; g.sA2 = p.d;
  g.B = x.K; g.ldb = x.ldk; g.sB1 = (long)p.Lk * x.ldk; g.sB2 = p.d; g.divB = kps;
  g.C = S1; g.ldc = p.Lkp; g.sC1 = (long)H * p.Lq * p.Lkp; g.sC2 = (long)p.Lq * p.Lkp;
  g.M = p.Lq; g.N = p.Lk; g.K = p.d; g.Z1 = nt; g.Z2 = H; g.alpha = scale;
  if (!p.kv_const) {   // + scale * Q dK^T in the same launch (second operand pair)
    g.A2 = x.Q; g.lda2 = x.ldq; g.sA21 = (long)p.Lq * x.ldq; g.sA22 = p.d; g.divA2 = kps;
    g.B2 = t.K; g.ldb2 = t.ldk; g.sB21 = (long)p.Lk * t.ldk; g.sB22 = p.d; g.divB2 = 1;
    g.K2 = p.d;
  }
  if (int r = gemm(e, g)) return r;
  e->n_launch++;
  if (int r = launch_softmax_jvp(e->dtype, ws + p.P, S1, nullptr, (long)nt * H, H, kps, p.Lq, p.Lk, p.Lkp, e->stream)) return r;
  GemmArgs o;   // dO = dP V
  o.A = S1; o.lda = p.Lkp; o.sA1 = (long)H * p.Lq * p.Lkp; o.sA2 = (long)p.Lq * p.Lkp;
  o.B = ws + p.VT; o.ldb = p.Lkp; o.sB1 = (long)H * p.d * p.Lkp; o.sB2 = (long)p.d * p.Lkp; o.divB = kps;
  o.C = t.O; o.ldc = t.ldo; o.sC1 = (long)p.Lq * t.ldo; o.sC2 = p.d;
  o.M = p.Lq; o.N = p.d; o.K = p.Lkp; o.Z1 = nt; o.Z2 = H;
  if (!p.kv_const) {   // + P dV in the same launch
    char* T1 = ws + e->T1;
    e->n_launch++;
    if (int r = launch_transpose(e->dtype, t.V, T1, nt, H, (long)p.Lk * t.ldv, p.d, p.Lk, p.d, t.ldv, p.Lkp, (long)p.d * p.Lkp, e->stream)) return r;
    o.A2 = ws + p.P; o.lda2 = p.Lkp; o.sA21 = (long)H * p.Lq * p.Lkp; o.sA22 = (long)p.Lq * p.Lkp; o.divA2 = kps;
    o.B2 = T1; o.ldb2 = p.Lkp; o.sB21 = (long)H * p.d * p.Lkp; o.sB22 = (long)p.d * p.Lkp; o.divB2 = 1;
    o.K2 = p.Lkp;
  }
  return gemm(e, o);
}

int attn_adjoint(dpb_engine* e, const Op& op, int nt) {
  const dpb_op_desc& d = op.d;
  const AttnPlan& p = e->plans[op.attn];
  const int H = p.heads, kps = nt / e->cur_batch;
  const float scale = 1.f / sqrtf((float)p.d);
  char* ws = e->ws;
  char* S1 = ws + e->S1;
  float* Dv = (float*)(ws + e->Dv);
  const AttnPtrs x = attn_ptrs(e, d, p, 0), c = attn_ptrs(e, d, p, 2);
  const char* gO = c.O;
  // first-write / accumulate flags, read up front: q, k, v may be windows of ONE buffer (fused QKV)
  const int accQ = e->ginit[d.in0], accK = p.kv_const ? 0 : e->ginit[d.in1], accV = p.kv_const ? 0 : e->ginit[d.in2];
  if (p.fused) {
    e->n_launch += attn_adj_launches(p.d, p.Lq, kps, nt);
    FusedAttnArgs f;
    fill_fused(e, p, x, f, kps, scale);
    f.gO = gO; f.gQ = (void*)c.Q; f.gK = (void*)c.K; f.gV = (void*)c.V; f.Drow = Dv;
    f.accQ = accQ; f.accK = accK; f.accV = accV;
    const double fl = 2.0 * p.Lq * (double)p.Lk * p.d * 7 * nt * H;     // query-major: scores, gP, gQ (3); key-major: scores^T, gP^T, gV, gK (4)
    e->flops += fl;
    const int pi = prof_open(e, fl, 9, p.Lq, p.Lk, p.d, nt * H, attn_adj_route_bits(p.d, p.Lq, kps, nt));
    const int r = launch_attn_adj_fused(f, nt, e->stream);
    prof_close(e, pi);
    if (r) return r;
    e->ginit[d.in0] = e->ginit[d.in1] = e->ginit[d.in2] = 1;
    return 0;
  }
  if (p.cross) {   // constant K/V: gQ (+)= scale [P o (gO V^T - delta)] K in one launch
    CrossAttnArgs f;
    f.Q = x.Q; f.K = x.K; f.V = x.V; f.BT = ws + p.KT; f.X = gO; f.Y = (void*)c.Q;
    f.L = p.Lq; f.Lk = p.Lk; f.Lkp = p.Lkp; f.C = x.ldq; f.Ck = x.ldk; f.Cx = c.ldo; f.Cy = c.ldq;
    f.H = H; f.d = p.d; f.kps = kps; f.adjoint = 1; f.accumulate = accQ; f.scale = scale; f.fl = e->dtype == DT_F16;
    e->n_launch++;
    const double fl = 2.0 * p.Lq * (double)p.Lk * p.d * 2 * nt * H;
    e->flops += fl;
    const int pi = prof_open(e, fl, 10, p.Lq, p.Lk, p.d, nt * H, 1);
    const int r = launch_attn_cross(f, nt, e->stream);
    prof_close(e, pi);
    if (r) return r;
    e->ginit[d.in0] = 1;
    return 0;
  }
  GemmArgs g;   // gP = gO V^T
  g.A = gO; g.lda = c.ldo; g.sA1 = (long)p.Lq * c.ldo; g.sA2 = p.d;
  g.B = x.V; g.ldb = x.ldv; g.sB1 = (long)p.Lk * x.ldv; g.sB2 = p.d; g.divB = kps;
  g.C = S1; g.ldc = p.Lkp; g.sC1 = (long)H * p.Lq * p.Lkp; g.sC2 = (long)p.Lq * p.Lkp;
  g.M = p.Lq; g.N = p.Lk; g.K = p.d; g.Z1 = nt; g.Z2 = H;
  if (int r = gemm(e, g)) return r;
  e->n_launch++;
  if (int r = launch_softmax_jvp(e->dtype, ws + p.P, S1, p.kv_const ? nullptr : Dv, (long)nt * H, H, kps, p.Lq, p.Lk, p.Lkp, e->stream)) return r;
  GemmArgs q;   // gQ (+)= scale * gS K
  q.A = S1; q.lda = p.Lkp; q.sA1 = (long)H * p.Lq * p.Lkp; q.sA2 = (long)p.Lq * p.Lkp;
  q.B = ws + p.KT; q.ldb = p.Lkp; q.sB1 = (long)H * p.d * p.Lkp; q.sB2 = (long)p.d * p.Lkp; q.divB = kps;
  q.C = (void*)c.Q; q.ldc = c.ldq; q.sC1 = (long)p.Lq * c.ldq; q.sC2 = p.d;
  q.M = p.Lq; q.N = p.d; q.K = p.Lkp; q.Z1 = nt; q.Z2 = H; q.alpha = scale;
  q.accumulate = accQ;
  if (int r = gemm(e, q)) return r;
  e->ginit[d.in0] = 1;
  if (p.kv_const) return 0;
  char* T1 = ws + e->T1;
  char* S2 = ws + e->S2;
  e->n_launch += 2;
  // gO^T per head [d][Lqp]
  if (int r = launch_transpose(e->dtype, gO, T1, nt, H, (long)p.Lq * c.ldo, p.d, p.Lq, p.d, c.ldo, p.Lqp, (long)p.d * p.Lqp, e->stream)) return r;
  GemmArgs v;   // gV (+)= P^T gO
  v.A = ws + p.PT; v.lda = p.Lqp; v.sA1 = (long)H * p.Lk * p.Lqp; v.sA2 = (long)p.Lk * p.Lqp; v.divA = kps;
  v.B = T1; v.ldb = p.Lqp; v.sB1 = (long)H * p.d * p.Lqp; v.sB2 = (long)p.d * p.Lqp;
  v.C = (void*)c.V; v.ldc = c.ldv; v.sC1 = (long)p.Lk * c.ldv; v.sC2 = p.d;
  v.M = p.Lk; v.N = p.d; v.K = p.Lqp; v.Z1 = nt; v.Z2 = H;
  v.accumulate = accV;
  if (int r = gemm(e, v)) return r;
  GemmArgs t;   // gP^T = V gO^T
  t.A = x.V; t.lda = x.ldv; t.sA1 = (long)p.Lk * x.ldv; t.sA2 = p.d; t.divA = kps;
  t.B = gO; t.ldb = c.ldo; t.sB1 = (long)p.Lq * c.ldo; t.sB2 = p.d;
  t.C = S2; t.ldc = p.Lqp; t.sC1 = (long)H * p.Lk * p.Lqp; t.sC2 = (long)p.Lk * p.Lqp;
  t.M = p.Lk; t.N = p.Lq; t.K = p.d; t.Z1 = nt; t.Z2 = H;
  if (int r = gemm(e, t)) return r;
  if (int r = launch_softmax_adjT(e->dtype, ws + p.PT, S2, Dv, (long)nt * H, H, kps, p.Lk, p.Lq, p.Lqp, e->stream)) return r;
  GemmArgs k;   // gK (+)= scale * gS^T Q
  k.A = S2; k.lda = p.Lqp; k.sA1 = (long)H * p.Lk * p.Lqp; k.sA2 = (long)p.Lk * p.Lqp;
  k.B = ws + p.QT; k.ldb = p.Lqp; k.sB1 = (long)H * p.d * p.Lqp; k.sB2 = (long)p.d * p.Lqp; k.divB = kps;
  k.C = (void*)c.K; k.ldc = c.ldk; k.sC1 = (long)p.Lk * c.ldk; k.sC2 = p.d;
  k.M = p.Lk; k.N = p.d; k.K = p.Lqp; k.Z1 = nt; k.Z2 = H; k.alpha = scale;
  k.accumulate = accK;
  if (int r = gemm(e, k)) return r;
  e->ginit[d.in1] = e->ginit[d.in2] = 1;
  return 0;
}

int run_op(dpb_engine* e, const Op& op, int mode, int n) {
  if (e->pend.on && !consumes_pending(e, op, mode))
    if (int r = flush_pending(e)) return r;
  switch (op.d.kind) {
    case DPB_OP_CONV: return mode == MODE_ADJOINT ? conv_adj(e, op, n) : conv_fwd(e, op, mode, n);
    case DPB_OP_GROUPNORM: return gn_run(e, op, mode, n);
    case DPB_OP_LAYERNORM: return ln_run(e, op, mode, n);
    case DPB_OP_GEGLU: return geglu_run(e, op, mode, n);
    case DPB_OP_CONCAT: return concat_run(e, op, mode, n);
    case DPB_OP_ATTENTION:
      return mode == MODE_PRIMAL ? attn_primal(e, op, n) : mode == MODE_TANGENT ? attn_tangent(e, op, n) : attn_adjoint(e, op, n);
    case DPB_OP_SILU: {
      if (mode != MODE_PRIMAL) return fail("SILU / quick-GELU ops are primal only (time-embedding path, text encoder)");
      const Buf& b = e->bufs[op.d.in0];
      e->n_launch++;
      if (op.d.ip[0] == 2)
        return launch_gelu(e->dtype, e->P(op.d.in0), e->P(op.d.out), (long)(b.kind == DPB_BUF_SHARED ? 1 : n) * b.rows * b.C, e->stream);
      if (op.d.ip[0] == 1)
        return launch_quick_gelu(e->dtype, e->P(op.d.in0), e->P(op.d.out), (long)(b.kind == DPB_BUF_SHARED ? 1 : n) * b.rows * b.C, e->stream);
      return launch_silu(e->dtype, e->P(op.d.in0), e->P(op.d.out), (long)(b.kind == DPB_BUF_SHARED ? 1 : n) * b.rows * b.C, e->stream);
    }
  }
  return fail("unknown op kind %d", op.d.kind);
}

int check_tap(dpb_engine* e, int tap, int nt) {
  if (!e->ws) return fail("workspace not set (dpb_engine_set_workspace)");
  if (tap < 0 || tap >= (int)e->bufs.size() || e->producer[tap] < 0) return fail("invalid tap buffer %d", tap);
  if (e->bufs[tap].is_const) return fail("tap buffer %d does not depend on x", tap);
  if (e->cur_batch <= 0) return fail("dpb_primal must run before jvp/vjp");
  if (nt <= 0 || nt > e->maxT || nt % e->cur_batch) return fail("nt=%d must be a positive multiple of batch=%d and <= max_tangents=%d", nt, e->cur_batch, e->maxT);
  return 0;
}

}  // namespace

// =================================================================== C ABI
extern "C" {

const char* dpb_last_error(void) { return g_err; }
int dpb_abi_version(void) { return DPB_ABI_VERSION; }

int dpb_engine_create(const dpb_net_desc* net, dpb_engine** out) {
  if (!net || !out) return fail("null argument");
  if (net->dtype != DPB_F32 && net->dtype != DPB_BF16 && net->dtype != DPB_F16) return fail("bad dtype %d", net->dtype);
  if (net->max_batch < 1 || net->max_tangents < 1) return fail("max_batch/max_tangents must be >= 1");
  dpb_engine* e = new dpb_engine();
  e->dtype = net->dtype;
  e->es = net->dtype == DPB_F32 ? 4 : 2;
  e->maxB = net->max_batch;
  e->maxT = net->max_tangents;
  e->x_buf = net->x_buf; e->x_channels = net->x_channels; e->temb_buf = net->temb_buf; e->temb_dim = net->temb_dim;
  e->temb_flip = net->temb_flip_sin_to_cos; e->temb_hm1 = net->temb_half_minus_one; e->ctx_buf = net->ctx_buf;
  const int nb = net->n_buffers;
  e->bufs.resize(nb);
  e->producer.assign(nb, -1);
  auto bad = [&](const char* m, int i) { fail("op %d: %s", i, m); delete e; return -1; };
  for (int i = 0; i < nb; ++i) {
    e->bufs[i].rows = net->buffers[i].rows;
    e->bufs[i].C = net->buffers[i].channels;
    e->bufs[i].kind = net->buffers[i].kind;
    e->bufs[i].Cv = net->buffers[i].valid_channels > 0 ? net->buffers[i].valid_channels : net->buffers[i].channels;
    if (e->bufs[i].rows < 1 || e->bufs[i].C < 8 || e->bufs[i].C % 8) { fail("buffer %d: rows=%d channels=%d (channels must be a multiple of 8)", i, e->bufs[i].rows, e->bufs[i].C); delete e; return -1; }
  }
  if (net->x_buf < 0 || net->x_buf >= nb) { fail("bad x_buf"); delete e; return -1; }
  e->bufs[net->x_buf].is_const = false;
  auto okb = [&](int b) { return b >= 0 && b < nb; };
  for (int i = 0; i < net->n_ops; ++i) {
    Op op;
    op.d = net->ops[i];
    const dpb_op_desc& d = op.d;
    if (!okb(d.in0) || !okb(d.out)) return bad("bad buffer id", i);
    bool c = e->bufs[d.in0].is_const;
    if (d.kind == DPB_OP_ATTENTION || d.kind == DPB_OP_CONCAT) {
      if (!okb(d.in1)) return bad("bad in1", i);
      c = c && e->bufs[d.in1].is_const;
    }
    if (d.kind == DPB_OP_ATTENTION) {
      if (!okb(d.in2)) return bad("bad in2", i);
      c = c && e->bufs[d.in2].is_const;
      AttnPlan p;
      p.heads = d.ip[0];
      p.oq = d.ip[1]; p.ok = d.ip[2]; p.ov = d.ip[3];
      p.causal = d.ip[4] != 0;
      const int Cattn = e->bufs[d.out].C;
      if (p.heads < 1 || Cattn % p.heads) return bad("channels not divisible by heads", i);
      p.d = Cattn / p.heads;
      if (p.oq % 8 || p.ok % 8 || p.ov % 8 || p.oq + Cattn > e->bufs[d.in0].C || p.ok + Cattn > e->bufs[d.in1].C || p.ov + Cattn > e->bufs[d.in2].C)
        return bad("bad q/k/v column window", i);
      if (p.d % 8) return bad("head dim must be a multiple of 8", i);
      p.Lq = e->bufs[d.in0].rows; p.Lk = e->bufs[d.in1].rows;
      p.Lqp = round8(p.Lq); p.Lkp = round8(p.Lk);
      if (e->bufs[d.in1].is_const != e->bufs[d.in2].is_const) return bad("k and v must both depend on x or both not", i);
      if (e->bufs[d.in0].is_const) return bad("attention query independent of x is unsupported", i);
      p.kv_const = e->bufs[d.in1].is_const;
      op.attn = (int)e->plans.size();
      e->plans.push_back(p);
    }
    if (d.kind == DPB_OP_CONV) {
      if (d.res >= 0) { if (!okb(d.res)) return bad("bad res", i); c = c && e->bufs[d.res].is_const; }
      if (d.rowbias >= 0 && (!okb(d.rowbias) || e->bufs[d.rowbias].kind != DPB_BUF_SHARED)) return bad("rowbias must be a SHARED buffer", i);
      if (d.rowbias >= 0 && (d.ip[10] < 0 || d.ip[10] % 8 || d.ip[10] + round8(d.ip[5]) > e->bufs[d.rowbias].C)) return bad("bad rowbias column window", i);
      if (!d.w[0]) return bad("missing weight", i);
      if (d.ip[2] != e->bufs[d.in0].C || e->bufs[d.out].C != round8(d.ip[5])) return bad("conv channel mismatch", i);
    }
    op.is_const = c;
    e->bufs[d.out].is_const = c;
    if (e->producer[d.out] >= 0) return bad("buffer written twice (tape must be SSA)", i);
    e->producer[d.out] = i;
    e->ops.push_back(op);
  }
  // ---------------- GEGLU fusion pairs: FF-in conv -> GEGLU (interleaved layout, sole consumer) -> FF-out conv (sole consumer)
  {
    std::vector<int> uses(nb, 0), user(nb, -1);
    for (size_t i = 0; i < e->ops.size(); ++i) {
      const dpb_op_desc& d = e->ops[i].d;
      for (int b : {d.in0, d.in1, d.in2, d.res}) if (b >= 0 && b < nb) { uses[b]++; user[b] = (int)i; }
    }
    e->uses = uses;
    // LayerNorm fused into the neighbouring product's epilogue (row-complete 128 x 320 ring tile, 16-bit engines): tangent -- the product that
    // WRITES the LayerNorm input also writes the LayerNorm tangent; adjoint -- the adjoint of the product that READS the LayerNorm output
    // applies the LayerNorm adjoint to its result.  (The 64 x 64 level of SD: C = 320.)
    for (size_t j = 0; j < e->ops.size(); ++j) {
      const dpb_op_desc& d = e->ops[j].d;
      if (d.kind != DPB_OP_LAYERNORM || e->ops[j].is_const || e->dtype == DT_F32 || e->bufs[d.in0].C != 320 || e->bufs[d.in0].kind != DPB_BUF_ACT) continue;
      const int pi = e->producer[d.in0];
      if (pi >= 0 && e->ops[pi].d.kind == DPB_OP_CONV && e->ops[pi].d.ip[9] == DPB_GATHER_NONE && !e->ops[pi].is_const) e->ops[pi].ln_next = (int)j;
      if (uses[d.out] == 1) {
        const int ci = user[d.out];
        const dpb_op_desc& cd = e->ops[ci].d;
        if (cd.kind == DPB_OP_CONV && cd.ip[9] == DPB_GATHER_NONE && cd.in0 == d.out && cd.w[1]) e->ops[ci].ln_prev = (int)j;
      }
    }
    for (size_t j = 0; j < e->ops.size(); ++j) {
      const dpb_op_desc& d = e->ops[j].d;
      // the primal GEGLU overwrites its input by the factors (G1, G2) (elementwise.hip): nothing else may read that buffer
      if (d.kind == DPB_OP_GEGLU && uses[d.in0] != 1) return bad("the input buffer of a GEGLU op must have no other consumer (the primal pass overwrites it in place)", (int)j);
      if (d.kind != DPB_OP_GEGLU || d.ip[1] != 64 || e->ops[j].is_const || getenv("DPB_NO_GEGLU_FUSE")) continue;   // (env: A/B tuning switch)
      const int pi = e->producer[d.in0];
      if (pi >= 0 && uses[d.in0] == 1) {
        const dpb_op_desc& pd = e->ops[pi].d;
        if (pd.kind == DPB_OP_CONV && pd.ip[9] == DPB_GATHER_NONE && pd.res < 0 && e->bufs[d.in0].kind == DPB_BUF_ACT) e->ops[pi].geglu_next = (int)j;
      }
      if (uses[d.out] == 1) {
        const int ci = user[d.out];
        const dpb_op_desc& cd = e->ops[ci].d;
        if (cd.kind == DPB_OP_CONV && cd.ip[9] == DPB_GATHER_NONE && cd.in0 == d.out) e->ops[ci].geglu_prev = (int)j;
      }
    }
    e->skip.assign(e->ops.size(), 0);
  }
  // ---------------- memory plan
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
  const size_t es = e->es;
  for (auto& b : e->bufs) {
    size_t n = b.kind == DPB_BUF_SHARED ? 1 : e->maxB;
    b.p_off = take(n * b.rows * (size_t)b.C * es);
  }
  for (auto& b : e->bufs)
    if (!b.is_const) b.t_off = take((size_t)e->maxT * b.rows * b.C * es);
  for (auto& b : e->bufs)
    if (!b.is_const) b.g_off = b.g_off0 = take((size_t)e->maxT * b.rows * b.C * es);
  size_t s1 = 0, s2 = 0, t1 = 0, dv = 0, ctmp = 0, maxrc = 0;
  for (auto& op : e->ops) {
    const dpb_op_desc& d = op.d;
    if (d.kind == DPB_OP_ATTENTION) {
      AttnPlan& p = e->plans[op.attn];
      const size_t H = p.heads;
      // long bf16 self-attention layers run the flash-style kernels (attn_fused.hip): no L x L object is ever stored
      p.fused = !p.causal && fused_attention_supported(e->dtype, p.d, p.Lq, p.kv_const) && !getenv("DPB_NO_FUSED_ATTN") &&
                p.Lq >= (getenv("DPB_FUSED_ATTN_MIN_L") ? atoi(getenv("DPB_FUSED_ATTN_MIN_L")) : 64) &&       // tuning override (256: the 8x8 level back on the materialised path)
                e->bufs[d.in0].C == e->bufs[d.in1].C && e->bufs[d.in0].C == e->bufs[d.in2].C;
      p.cross = !p.causal && cross_attention_supported(e->dtype, p.d, p.Lq, p.Lk, p.kv_const) && !getenv("DPB_NO_CROSS_ATTN") &&
                e->bufs[d.in1].C == e->bufs[d.in2].C;
      if (!p.fused) p.P = take((size_t)e->maxB * H * p.Lq * p.Lkp * es);
      p.VT = take((size_t)e->maxB * H * p.d * p.Lkp * es);
      p.KT = take((size_t)e->maxB * H * p.d * p.Lkp * es);
      if (!p.fused) s1 = std::max(s1, (size_t)e->maxT * H * p.Lq * p.Lkp * es);
      size_t lmax = std::max(p.Lqp, p.Lkp);
      t1 = std::max(t1, (size_t)e->maxT * H * p.d * lmax * es);
      if (p.fused) p.stats = take((size_t)e->maxB * H * p.Lq * 2 * sizeof(float));
      if (!p.kv_const) {
        if (!p.fused) p.PT = take((size_t)e->maxB * H * p.Lk * p.Lqp * es);
        p.QT = take((size_t)e->maxB * H * p.d * p.Lqp * es);
        if (!p.fused) s2 = std::max(s2, (size_t)e->maxT * H * p.Lk * p.Lqp * es);
        dv = std::max(dv, (size_t)e->maxT * H * p.Lq * sizeof(float));
      }
    } else if (d.kind == DPB_OP_GROUPNORM) {
      op.pstats = e->pstats_bytes;
      e->pstats_bytes += (size_t)e->maxB * d.ip[0] * 2 * sizeof(double);
      if (!op.is_const) {
        op.tstats = e->tstats_bytes;
        e->tstats_bytes += (size_t)e->maxT * d.ip[0] * 2 * sizeof(double);
      }
    } else if (d.kind == DPB_OP_CONV && d.ip[9] == DPB_GATHER_UPCONV && !op.is_const) {
      ctmp = std::max(ctmp, (size_t)e->maxT * d.ip[3] * d.ip[4] * e->bufs[d.in0].C * es);
    }
  }
  for (auto& b : e->bufs) maxrc = std::max(maxrc, (size_t)b.rows * b.C);
  e->pstats_off = take(e->pstats_bytes);
  e->tstats_off = take(e->tstats_bytes);
  {  // GroupNorm statistics scratch: one partial per (sample or tangent, block, group); a launch uses at most max(512, HW / 8) blocks in all
    size_t need = 0;
    for (auto& op : e->ops)
      if (op.d.kind == DPB_OP_GROUPNORM) {
        const size_t hw = e->bufs[op.d.in0].rows, nmax = (size_t)std::max(e->maxB, e->maxT);
        const size_t blocks = std::max<size_t>(nmax * ((hw + 7) / 8), 1);          // ppb >= 8
        need = std::max(need, std::min<size_t>(blocks, std::max<size_t>(1024, nmax * ((hw + 63) / 64))) * 2 * op.d.ip[0] * sizeof(float));
      }
    e->gnpart_bytes = need;
    e->gnpart = take(need);
    e->gnticket = take(sizeof(int) * (size_t)std::max(e->maxB, e->maxT));
  }
  for (auto& op : e->ops)
    if (op.d.kind == DPB_OP_GROUPNORM) { op.pstats += e->pstats_off; op.tstats += e->tstats_off; }
  e->S1 = take(s1); e->S2 = take(s2); e->T1 = take(t1); e->Dv = take(dv); e->convtmp = take(ctmp);
  const size_t nio = (size_t)std::max(e->maxB, e->maxT) * maxrc * sizeof(float);
  e->io_in = take(nio);
  e->io_out = take(nio);
  e->orth_stride = align_up(orth_scratch_bytes(std::min(ORTH_MAX_RANK, std::max(56, e->maxT)), (long)e->bufs[e->x_buf].rows * e->x_channels));   // k <= max_tangents
  e->orth = take(e->orth_stride * (size_t)e->maxB);
  e->slab = take(e->slab_bytes);
  e->zeros = take(256);
  const size_t nx = (size_t)e->bufs[e->x_buf].rows * e->x_channels;
  e->pbW = take((size_t)e->maxT * nx * sizeof(float));
  e->ws_bytes = off;
  e->ginit.assign(nb, 0);
  *out = e;
  return 0;
}

void dpb_engine_destroy(dpb_engine* e) {
  if (e && e->gexec) (void)hipGraphExecDestroy(e->gexec);
  delete e;
}

int dpb_engine_set_stream(dpb_engine* e, void* s) {
  if (!e) return fail("null engine");
  e->stream = (hipStream_t)s;
  return 0;
}

size_t dpb_engine_workspace_bytes(const dpb_engine* e) { return e ? e->ws_bytes : 0; }

int dpb_engine_set_workspace(dpb_engine* e, void* ws, size_t bytes) {
  if (!e || !ws) return fail("null argument");
  if (bytes < e->ws_bytes) return fail("workspace too small: %zu < %zu", bytes, e->ws_bytes);
  if ((uintptr_t)ws % 256) return fail("workspace must be 256-byte aligned");
  e->ws = (char*)ws;
  e->cur_batch = 0;
  DPB_CHECK(hipMemsetAsync(ws, 0, e->ws_bytes, e->stream));
  return 0;
}

static int primal_pass(dpb_engine* e, const float* x, int batch, float t, const float* ctx, int upto_buf) {
  if (!e || !x) return fail("null argument");
  if (!e->ws) return fail("workspace not set (dpb_engine_set_workspace)");
  if (batch < 1 || batch > e->maxB) return fail("batch=%d outside [1,%d]", batch, e->maxB);
  if (upto_buf < 0 || upto_buf >= (int)e->bufs.size() || e->producer[upto_buf] < 0) return fail("invalid upto buffer %d", upto_buf);
  e->n_launch = 0; e->flops = 0; e->gbytes = 0;
  const Buf& bx = e->bufs[e->x_buf];
  e->n_launch++;
  if (int r = launch_nchw_to_nhwc(e->dtype, x, e->P(e->x_buf), batch, e->x_channels, bx.rows, bx.C, e->stream)) return r;
  if (e->ctx_buf >= 0) {
    if (!ctx) return fail("this network needs ctx (encoder_hidden_states)");
    const Buf& bc = e->bufs[e->ctx_buf];
    // ctx is already [batch][rows][Cv] channel-last (Cv = un-padded width): cast (and zero-pad to C) via the nchw kernel with HW=1
    e->n_launch++;
    if (int r = launch_nchw_to_nhwc(e->dtype, ctx, e->P(e->ctx_buf), batch * bc.rows, bc.Cv, 1, bc.C, e->stream)) return r;
  }
  if (e->temb_buf >= 0) {
    // sinusoidal timestep embedding, computed on the host in fp32 exactly as the reference frameworks do
    // (diffusion.py:783-804 / diffusers Timesteps), uploaded through the io staging area.
    const int dim = e->temb_dim, half = dim / 2;
    std::vector<float> emb(e->bufs[e->temb_buf].C, 0.f);
    for (int i = 0; i < half; ++i) {
      float f;
      if (e->temb_hm1) f = expf((float)i * (float)(-(log(10000.0) / (double)(half - 1))));   // diffusion.py:797-798
      else f = expf(((float)(-log(10000.0)) * (float)i) / (float)half);                      // diffusers get_timestep_embedding
      float ang = t * f;
      float s = sinf(ang), c = cosf(ang);
      if (e->temb_flip) { emb[i] = c; emb[half + i] = s; } else { emb[i] = s; emb[half + i] = c; }
    }
    float* stage = (float*)(e->ws + e->io_in);
    DPB_CHECK(hipMemcpyAsync(stage, emb.data(), emb.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
    DPB_CHECK(hipStreamSynchronize(e->stream));   // emb is a host temporary
    e->n_launch++;
    if (int r = launch_nchw_to_nhwc(e->dtype, stage, e->P(e->temb_buf), 1, (int)emb.size(), 1, (int)emb.size(), e->stream)) return r;
  }
  if (e->pstats_bytes && !gn_deterministic()) DPB_CHECK(hipMemsetAsync(e->ws + e->pstats_off, 0, e->pstats_bytes, e->stream));   // atomic statistics path accumulates
  e->cur_batch = batch;
  e->cur_tap = upto_buf; e->pend.on = false;
  const int last = e->producer[upto_buf];
  std::fill(e->skip.begin(), e->skip.end(), 0);
  for (int i = 0; i <= last; ++i) {
    if (e->skip[i]) continue;                      // (forward only: a GEGLU applied by the epilogue of the FF-in product)
    if (int r = run_op(e, e->ops[i], MODE_PRIMAL, batch)) return r;
  }
  return 0;
}

int dpb_primal(dpb_engine* e, const float* x, int batch, float t, const float* ctx, int upto_buf) {
  if (e) e->fwd_only = false;
  return primal_pass(e, x, batch, t, ctx, upto_buf);
}

int dpb_forward(dpb_engine* e, const float* x, int batch, float t, const float* ctx, int upto_buf, int channels, float* out) {
  if (!e || !out) return fail("null argument");
  e->fwd_only = true;
  int r = primal_pass(e, x, batch, t, ctx, upto_buf);
  e->fwd_only = false;
  if (!r) r = dpb_read_buffer(e, upto_buf, channels, out);
  e->cur_batch = 0;                                // no stash was kept: dpb_jvp / dpb_vjp / dpb_pullback_iterate refuse until the next dpb_primal
  return r;
}

int dpb_read_buffer(dpb_engine* e, int buf, int channels, float* out) {
  if (!e || !out) return fail("null argument");
  if (buf < 0 || buf >= (int)e->bufs.size()) return fail("bad buffer %d", buf);
  const Buf& b = e->bufs[buf];
  if (channels < 1 || channels > b.C) return fail("bad channel count");
  int n = b.kind == DPB_BUF_SHARED ? 1 : e->cur_batch;
  if (n < 1) return fail("no primal state to read (dpb_forward invalidates it; run dpb_primal first)");
  return launch_nhwc_to_nchw(e->dtype, e->P(buf), out, n, channels, b.rows, b.C, e->stream);
}

// U == nullptr (dpb_pullback_iterate, every iteration but the last): the tangent of the tap stays in T(tap); the adjoint pass takes it from there
static int jvp_pass(dpb_engine* e, int tap, const float* V, int nt, float* U) {
  if (!e || !V) return fail("null argument");
  if (int r = check_tap(e, tap, nt)) return r;
  e->n_launch = 0; e->flops = 0; e->gbytes = 0;
  const Buf& bx = e->bufs[e->x_buf];
  e->n_launch++;
  if (int r = launch_nchw_to_nhwc(e->dtype, V, e->T(e->x_buf), nt, e->x_channels, bx.rows, bx.C, e->stream)) return r;
  if (e->tstats_bytes && !gn_deterministic()) DPB_CHECK(hipMemsetAsync(e->ws + e->tstats_off, 0, e->tstats_bytes, e->stream));   // atomic statistics accumulate
  const int last = e->producer[tap];
  std::fill(e->skip.begin(), e->skip.end(), 0);
  e->cur_tap = tap; e->pend.on = false;
  for (int i = 0; i <= last; ++i) {
    if (e->ops[i].is_const || e->skip[i]) continue;
    if (int r = run_op(e, e->ops[i], MODE_TANGENT, nt)) return r;
  }
  if (int r = flush_pending(e)) return r;
  if (!U) return 0;
  const Buf& bt = e->bufs[tap];
  e->n_launch++;
  return launch_nhwc_to_nchw(e->dtype, e->T(tap), U, nt, bt.Cv, bt.rows, bt.C, e->stream);
}

int dpb_jvp(dpb_engine* e, int tap, const float* V, int nt, float* U) {
  if (!U) return fail("null argument");
  return jvp_pass(e, tap, V, nt, U);
}

// U == nullptr: the cotangent seed IS the tangent the last jvp_pass left in T(tap) -- its storage is handed to G(tap) for this pass (the fp32 NCHW round
// trip through U is the identity on 16-bit and fp32 values alike, so the results are bitwise those of the two conversion kernels it replaces)
static int vjp_pass(dpb_engine* e, int tap, const float* U, int nt, float* W) {
  if (!e || !W) return fail("null argument");
  if (int r = check_tap(e, tap, nt)) return r;
  e->n_launch = 0; e->flops = 0; e->gbytes = 0;
  const Buf& bt = e->bufs[tap];
  std::fill(e->ginit.begin(), e->ginit.end(), 0);
  for (auto& b : e->bufs) b.g_off = b.g_off0;
  if (U) {
    e->n_launch++;
    if (int r = launch_nchw_to_nhwc(e->dtype, U, e->G(tap), nt, bt.Cv, bt.rows, bt.C, e->stream)) return r;
  } else {
    e->bufs[tap].g_off = e->bufs[tap].t_off;       // restored from g_off0 at the start of the next adjoint pass
  }
  e->ginit[tap] = 1;
  if (e->tstats_bytes && !gn_deterministic()) DPB_CHECK(hipMemsetAsync(e->ws + e->tstats_off, 0, e->tstats_bytes, e->stream));
  e->cur_tap = tap; e->pend.on = false;
  for (int i = e->producer[tap]; i >= 0; --i) {
    const Op& op = e->ops[i];
    if (op.is_const || !e->ginit[op.d.out]) continue;
    if (int r = run_op(e, op, MODE_ADJOINT, nt)) return r;
  }
  if (int r = flush_pending(e)) return r;
  if (!e->ginit[e->x_buf]) return fail("tap buffer %d is not connected to x", tap);
  const Buf& bx = e->bufs[e->x_buf];
  e->n_launch++;
  const int r = launch_nhwc_to_nchw(e->dtype, e->G(e->x_buf), W, nt, e->x_channels, bx.rows, bx.C, e->stream);
  // Invariant of Buf::g_off / t_off: between passes every buffer's cotangent storage is its own (g_off == g_off0).  Inside the pass the residual adjoint
  // swaps g_off between buffers and U == nullptr lends the tap's TANGENT storage to its cotangent; undo both here so that nothing that reads G() or
  // T(tap) after the pass (a debug read, a later feature) sees aliased data.  (The launch above is already enqueued with the pointer it needs.)
  for (auto& b : e->bufs) b.g_off = b.g_off0;
  return r;
}

int dpb_vjp(dpb_engine* e, int tap, const float* U, int nt, float* W) {
  if (!U) return fail("null argument");
  return vjp_pass(e, tap, U, nt, W);
}

int dpb_orth(const float* W, const float* Vprev, float* V, float* s, float* conv, void* scratch, int k, int64_t N, void* stream) {
  if (!W || !Vprev || !V || !s || !conv || !scratch) return fail("null argument");
  OrthArgs a;
  a.W = W; a.Vprev = Vprev; a.V = V; a.s = s; a.conv = conv; a.scratch = (double*)scratch; a.k = k; a.N = N;
  a.scratch_bytes = orth_scratch_bytes(k, N);      // the caller's contract (include/dpb.h)
  return launch_orth(a, (hipStream_t)stream);
}

int dpb_orth_checked(const float* W, const float* Vprev, float* V, float* s, float* conv, void* scratch, size_t scratch_bytes, int k, int64_t N,
                     void* stream) {
  if (!W || !Vprev || !V || !s || !conv || !scratch) return fail("null argument");
  if (k < 1 || k > ORTH_MAX_RANK || N < 1) return fail("dpb_orth: k=%d outside [1,%d] or N=%lld < 1", k, ORTH_MAX_RANK, (long long)N);
  const size_t need = orth_scratch_bytes(k, N);
  if (scratch_bytes < need) return fail("dpb_orth: scratch of %zu bytes, dpb_orth_scratch_bytes(%d, %lld) = %zu", scratch_bytes, k, (long long)N, need);
  OrthArgs a;
  a.W = W; a.Vprev = Vprev; a.V = V; a.s = s; a.conv = conv; a.scratch = (double*)scratch; a.k = k; a.N = N;
  a.scratch_bytes = scratch_bytes;
  return launch_orth(a, (hipStream_t)stream);
}

size_t dpb_orth_scratch_bytes(int k, int64_t N) { return (k < 1 || k > ORTH_MAX_RANK || N < 1) ? 0 : orth_scratch_bytes(k, N); }

static int g_iter_alias = getenv("DPB_ITER_ALIAS") ? atoi(getenv("DPB_ITER_ALIAS")) : 1;   // A/B switch: 0 = convert U out and back in every iteration
static int g_orth_batch = getenv("DPB_ORTH_BATCH") ? atoi(getenv("DPB_ORTH_BATCH")) : 1;   // A/B switch: 0 = re-orthonormalise the samples of a batch one by one
static int g_graph_iterate = 0;      // dpb_debug_set("graph_iterate", 1): replay the power iteration as a captured hipGraph (measurement option)

int dpb_pullback_iterate(dpb_engine* e, int tap, float* V, float* U, float* s, float* conv, int k, int n_iters) {
  if (!e || !V || !U || !s || !conv) return fail("null argument");
  if (k < 1 || k > ORTH_MAX_RANK) return fail("pca_rank k=%d outside [1,%d]", k, ORTH_MAX_RANK);
  const int B = e->cur_batch;                       // samples advanced together: one weight stream for all of them
  if (int r = check_tap(e, tap, k * (B > 0 ? B : 1))) return r;
  const int nt = k * B;
  const long N = (long)e->bufs[e->x_buf].rows * e->x_channels;
  float* Wm = (float*)(e->ws + e->pbW);
  long launches = 0; double fl = 0, gb = 0;
  const bool alias_ok = e->bufs[tap].Cv == e->bufs[tap].C && g_iter_alias;   // (padded tap channels: the conversion kernels zero them, an alias would not)
  auto body = [&](bool want_u) -> int {             // one power iteration: k JVPs, k VJPs, re-orthonormalisation, V <- V_new; no host sync
    // U = J V_prev is an OUTPUT of the last iteration only (utils.py:810): before that the tap's tangent goes straight from T(tap) into the adjoint
    // pass -- no nhwc -> fp32 nchw -> nhwc round trip (two launches per iteration, bitwise the same values)
    const bool keep = alias_ok && !want_u;
    if (int r = jvp_pass(e, tap, V, nt, keep ? nullptr : U)) return r;
    launches += e->n_launch; fl += e->flops; gb += e->gbytes;
    if (int r = vjp_pass(e, tap, keep ? nullptr : U, nt, Wm)) return r;
    launches += e->n_launch; fl += e->flops; gb += e->gbytes;
    {                                               // independent k x N re-orthonormalisation per sample, all samples in one set of four launches
      OrthArgs a;                                   // (in place, V is Vprev: see dpb.h)
      a.W = Wm; a.Vprev = V; a.V = V; a.s = s; a.conv = conv; a.scratch = (double*)(e->ws + e->orth); a.k = k; a.N = N;
      a.scratch_bytes = orth_scratch_bytes(k, N);
      a.batch = B; a.stride_w = (long)k * N; a.stride_v = (long)k * N; a.stride_s = k; a.stride_conv = 2; a.scratch_stride = e->orth_stride;
      if (g_orth_batch) {
        if (int r = launch_orth(a, e->stream)) return r;
      } else {                                      // A/B switch DPB_ORTH_BATCH=0: four launches per sample, as rounds 1-5 (same bits)
        a.batch = 1;
        for (int b = 0; b < B; ++b) {
          if (int r = launch_orth(a, e->stream)) return r;
          a.W += a.stride_w; a.Vprev += a.stride_v; a.V += a.stride_v; a.s += k; a.conv += 2; a.scratch += e->orth_stride / sizeof(double);
        }
      }
    }
    launches += g_orth_batch ? 4 : 4 * B;
    return 0;
  };
  int it = 0;
  if (g_graph_iterate && !e->profiling && e->stream != 0) {
    // The launch sequence of an iteration is fixed for fixed (tap, k, batch, buffers): capture it once (after one eager iteration, so that
    // every code object is loaded) and replay it.  Measured on MI355X: no gain -- the stream never runs dry.
    const dpb_engine::GraphKey key{tap, k, B, V, U, s, conv};
    if (!e->gexec || !(key == e->gkey)) {
      if (int r = body(true)) return r;
      ++it;
      if (e->gexec) { (void)hipGraphExecDestroy(e->gexec); e->gexec = nullptr; }
      hipGraph_t g = nullptr;
      const long l0 = launches; const double f0 = fl, b0 = gb;
      DPB_CHECK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
      const int r = body(true);                     // the captured iteration always writes U (it may be the last one replayed)
      const hipError_t ce = hipStreamEndCapture(e->stream, &g);
      e->g_launches = launches - l0; e->g_flops = fl - f0; e->g_bytes = gb - b0;
      launches = l0; fl = f0; gb = b0;               // captured, not executed
      if (r) { if (g) (void)hipGraphDestroy(g); return r; }
      DPB_CHECK(ce);
      DPB_CHECK(hipGraphInstantiate(&e->gexec, g, nullptr, nullptr, 0));
      (void)hipGraphDestroy(g);
      e->gkey = key;
    }
    for (; it < n_iters; ++it) {
      DPB_CHECK(hipGraphLaunch(e->gexec, e->stream));
      launches += e->g_launches; fl += e->g_flops; gb += e->g_bytes;
    }
  }
  for (; it < n_iters; ++it)
    if (int r = body(it == n_iters - 1)) return r;
  e->n_launch = launches; e->flops = fl; e->gbytes = gb;
  return 0;
}

int dpb_ddim_step(const float* x, const float* eps, float* out, float* x0, int64_t n, float a_t, float a_next, void* stream) {
  if (!x || !eps || !out) return fail("null argument");
  return launch_ddim_step(x, eps, out, x0, n, a_t, a_next, (hipStream_t)stream);
}

int dpb_embed_tokens(const int32_t* ids, const void* tok_table, const void* pos_table, int dtype, float* out, int batch, int tokens,
                     int channels, int vocab, void* stream) {
  if (!ids || !tok_table || !pos_table || !out) return fail("null argument");
  if (dtype != DPB_F32 && dtype != DPB_BF16 && dtype != DPB_F16) return fail("bad dtype %d", dtype);
  return launch_embed_tokens(dtype, ids, tok_table, pos_table, out, batch, tokens, channels, vocab, (hipStream_t)stream);
}

int dpb_lincomb(const float* x, const float* y, const float* z, float* out, int64_t n, float a, float b, float c, void* stream) {
  if (!x || !y || !out) return fail("null argument");
  return launch_lincomb(x, y, z, out, n, a, b, c, (hipStream_t)stream);
}

int dpb_engine_profile(dpb_engine* e, int enable) {
  if (!e) return fail("null engine");
  for (auto& p : e->prof) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  e->prof.clear();
  e->profiling = enable != 0;
  if (e->profiling) {
    // an event pair brackets more than the kernel's execution (event processing, dispatch ramp of the launch it encloses): measure the
    // bracket around a 64-element copy -- a kernel whose own execution is ~1.5 us in rocprofv3's kernel trace -- and subtract the
    // excess from every launch, so the per-launch durations agree with the kernel trace (cross-checked in profiles/)
    hipEvent_t a, b;
    DPB_CHECK(hipEventCreate(&a));
    DPB_CHECK(hipEventCreate(&b));
    float best = 1e9f;
    for (int i = 0; i < 24; ++i) {
      DPB_CHECK(hipEventRecord(a, e->stream));
      if (int r = launch_axpy(e->dtype, e->ws + e->zeros, e->ws + e->zeros, 64, 0, e->stream)) return r;
      DPB_CHECK(hipEventRecord(b, e->stream));
      DPB_CHECK(hipEventSynchronize(b));
      float ms = 0;
      DPB_CHECK(hipEventElapsedTime(&ms, a, b));
      if (i >= 4) best = ms < best ? ms : best;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    best -= 1.5e-3f;
    if (best < 0.f) best = 0.f;
    e->prof_overhead_ms = best < 1e8f ? best : 0.f;
  }
  return 0;
}

int dpb_engine_profile_dump(dpb_engine* e, const char* path) {
  if (!e || !path) return fail("null argument");
  DPB_CHECK(hipStreamSynchronize(e->stream));
  FILE* f = fopen(path, "w");
  if (!f) return fail("cannot open %s", path);
  fprintf(f, "idx,big,gather,M,N,K,Z,us,tflops,raw_us,bracket_overhead_us\n");   // us = raw_us - bracket_overhead_us (calibrated empty-bracket time)
  int i = 0;
  for (auto& p : e->prof) {
    float raw = 0;
    (void)hipEventElapsedTime(&raw, p.a, p.b);
    const float ms = raw > e->prof_overhead_ms ? raw - e->prof_overhead_ms : 0.f;
    fprintf(f, "%d,%d,%d,%d,%d,%d,%d,%.2f,%.2f,%.2f,%.2f\n", i++, p.big, p.gather, p.M, p.N, p.K, p.Z, ms * 1e3, ms > 0 ? p.flops / (ms * 1e-3) / 1e12 : 0.0,
            raw * 1e3, e->prof_overhead_ms * 1e3);
  }
  fclose(f);
  return 0;
}

int dpb_engine_profile_read(dpb_engine* e, int big_tile, int64_t* count, double* total_ms, double* flops) {
  if (!e || !count || !total_ms || !flops) return fail("null argument");
  DPB_CHECK(hipStreamSynchronize(e->stream));
  *count = 0; *total_ms = 0; *flops = 0;
  const bool raw = big_tile >= 1000;                 // kind + 1000: the RAW bracket times (no empty-bracket correction, nothing clamped)
  if (raw) big_tile -= 1000;
  if (big_tile < 0 || big_tile > 12) return fail("dpb_engine_profile_read: kind must be 0..12 or 1000..1012");
  for (auto& p : e->prof) {
    if (p.big != big_tile) continue;
    float ms = 0;
    DPB_CHECK(hipEventElapsedTime(&ms, p.a, p.b));
    if (!raw) ms = ms > e->prof_overhead_ms ? ms - e->prof_overhead_ms : 0.f;
    *count += 1; *total_ms += ms; *flops += p.flops;
  }
  return 0;
}

int dpb_engine_profile_overhead(const dpb_engine* e, double* bracket_overhead_ms) {
  if (!e || !bracket_overhead_ms) return fail("null argument");
  *bracket_overhead_ms = e->prof_overhead_ms;
  return 0;
}

int dpb_debug_set(const char* key, int value) {
  static int tile = 0, splitk = 0, kch = 0;
  if (!key) return fail("null key");
  if (!strcmp(key, "gemm_tile")) tile = value;
  else if (!strcmp(key, "gemm_splitk")) splitk = value;
  else if (!strcmp(key, "gemm_kch")) kch = value;
  else if (!strcmp(key, "gemm_dma_auto")) { gemm_debug_dma_auto(value); return 0; }
  else if (!strcmp(key, "gemm_order")) { gemm_debug_order(value); return 0; }
  else if (!strcmp(key, "p8")) { gemm_debug_p8(value); return 0; }
  else if (!strcmp(key, "wres")) { gemm_debug_wres(value); return 0; }
  else if (!strcmp(key, "gn_deterministic")) { gn_debug_deterministic(value); return 0; }
  else if (!strcmp(key, "graph_iterate")) { g_graph_iterate = value; return 0; }
  else if (!strcmp(key, "attn_shared")) { attn_debug_shared(value); return 0; }
  else if (!strcmp(key, "lazy_reduce")) { g_lazy_reduce = value; return 0; }
  else if (!strcmp(key, "ln_fuse")) { g_ln_fuse = value; return 0; }
  else if (!strcmp(key, "cross_primal")) { g_cross_primal = value; return 0; }
  else if (!strcmp(key, "geglu_fwd")) { g_geglu_fwd = value; return 0; }
  else if (!strcmp(key, "iter_alias")) { g_iter_alias = value; return 0; }
  else return fail("unknown debug key %s", key);
  gemm_debug_set(tile, splitk, kch);
  return 0;
}

int dpb_debug_gemm_plan(int dtype, int M, int N, int K, int conv_hw, int conv_cin, int epilogue, int64_t slab_bytes, int* kind, int* tile,
                        int* splitk) {
  if (!kind || !tile || !splitk) return fail("null argument");
  if (dtype != DPB_F32 && dtype != DPB_BF16 && dtype != DPB_F16) return fail("bad dtype %d", dtype);
  static char dummy[16];                       // the plan only looks at which pointers are set, never through them
  GemmArgs a;
  a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = K; a.ldc = N;
  a.zeros = dummy;
  if (slab_bytes > 0) { a.slab = (float*)dummy; a.slab_bytes = (size_t)slab_bytes; }
  a.epi = epilogue;
  if (conv_hw > 0) {                           // 3x3, stride 1, pad 1 on conv_hw x conv_hw images of conv_cin channels
    if (conv_cin <= 0 || K != 9 * conv_cin || M % (conv_hw * conv_hw)) return fail("conv plan: K must be 9*cin and M whole images");
    a.gather = GATHER_CONV; a.H = a.W = a.Ho = a.Wo = conv_hw; a.Cin = conv_cin; a.KS = 3; a.stride = 1; a.pad = 1; a.lda = conv_cin;
  }
  const GemmPlan pl = gemm_plan(dtype, a);
  if (pl.kind < 0) return -1;
  *kind = pl.kind; *tile = pl.tile; *splitk = pl.splitk;
  return 0;
}

int dpb_engine_stats(const dpb_engine* e, int64_t* launches, double* gemm_flops, double* gemm_bytes) {
  if (!e) return fail("null engine");
  if (launches) *launches = e->n_launch;
  if (gemm_flops) *gemm_flops = e->flops;
  if (gemm_bytes) *gemm_bytes = e->gbytes;
  return 0;
}

}  // extern "C"

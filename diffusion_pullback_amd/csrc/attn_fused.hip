// Fused (flash-style) tangent and adjoint self-attention for the long-sequence layers (bf16, head dim 40 / 80).
//
// The materialised path writes and re-reads k*heads*L^2 scores several times per pass (1.3 GB each at L = 4096,
// k = 5); these kernels recompute the probabilities tile by tile from Q, K and the primal row statistics
// (m_i = row max, l_i = row sum of exp, kept from the primal pass) and never touch HBM with an L x L object.
//
//   tangent :  dO_i = sum_j P_ij dS_ij V_j + sum_j P_ij dV_j - delta_i O_i ,  dS = scale (dQ K^T + Q dK^T),
//              delta_i = sum_j P_ij dS_ij                                      [attn_jvp_kernel]
//   adjoint :  gP_ij = gO_i . V_j ,  D_i = gO_i . O_i ,  gS = P o (gP - D)
//              gQ_i = scale sum_j gS_ij K_j                                     [attn_adj_q_kernel, query-major]
//              gK_j = scale sum_i gS_ij Q_i ,  gV_j = sum_i P_ij gO_i           [attn_adj_kv_kernel, key-major]
//
// Structure (all three): 4 waves per block, each wave owns 32 rows of the "outer" index (queries, or keys for the
// kv kernel) held as MFMA B operands in registers; 128-row tiles of the "inner" index stream through LDS.  Score
// products are computed transposed (inner index = MFMA row, outer = lane) so that each lane owns ONE outer row:
// the softmax algebra is register-local and the probabilities re-enter the second MFMA as B fragments by a plain
// fp32->bf16 pack, with no cross-lane movement.  The second-stage A operands (V^T, K^T, Q^T, dV^T, gO^T: [d][inner row]
// fragments) are read from the SAME row tiles with gfx950's LDS transpose read (ds_read_b64_tr_b16, lds_tr_frag below):
// no transposed copy of any operand exists in HBM or LDS.  (Only the 77-key cross-attention kernel still takes a
// pre-transposed K^T / V^T of the constant prompt projections, made once per sample.)
// v_mfma_f32_32x32x16_bf16 throughout; head dim padded to 48/80 for the score products and 64/96 for the outputs.
#include "kernels.h"

namespace dpb {

#ifndef DPB_ATT_ABL
#define DPB_ATT_ABL 0     // measurement-only builds (make ablate_attn, tools/gpu_attn_ablate.sh; WRONG results): bit 0 drops a quarter of the head-dim-side MFMAs of the
#endif                    // d = 40 kernels (what 16x16x32 tiles would save), bit 1 one of the six dS MFMAs of the tangent kernel (a stacked [K | dK] product), bit 2 the exp
#ifndef DPB_OUT_STORE_ATT
#define DPB_OUT_STORE_ATT 0   // flavour of the 8-byte output stores of the fused attention kernels (common.h store_out8): 0 plain, 1 sc1, 2 nt (A/B builds)
#endif
#define DPB_ATT_PAD 8     // row padding (bf16 elements) of the LDS [row][d] tiles (16 / 24 measured in round 3: 9.02 / 9.28 vs 8.99 ms per iteration)

constexpr int att_waves(int d) { return d > 80 ? 4 : 8; }   // waves per block: 8 x 32 = 256 outer rows share every streamed tile
                                                            // (head dim 160: 4 waves, the fragments need the 512-register budget)

template <int D, int W = att_waves(D)> struct FA {
  static constexpr int WAVES = W, NT = WAVES * 64;
  static constexpr int NS = (D + 15) / 16;      // k-steps of the score products
  static constexpr int DP = NS * 16;            // padded head dim (score products)
  static constexpr int ND = (D + 31) / 32;      // 32-wide output tiles over the head dim
  static constexpr int DO = ND * 32;
  static constexpr int BI = D <= 80 ? 128 : 64;  // inner rows per LDS stage: at 128 the one-stage-ahead register prefetch has
                                                // twice the MFMA time to land (key-major adjoint -8 %); head dim 160 would exceed the LDS
  static constexpr int LDR = (DP > DO ? DP : DO) + DPB_ATT_PAD;   // LDS stride of [row][d] tiles (bf16 elements): 36 / 52 / 84 dwords = 4 x odd, so
                                                // ds_read_b128 fragment reads are conflict-free; >= DO columns so that the transpose
                                                // reads of lds_tr_frag stay inside the row (columns DP.. are never consumed)
  static constexpr int LDT = BI + 4;            // LDS stride of [d][row] tiles: 68 elements = 34 dwords = 2*odd, so the 32 rows of a
                                                // ds_read_b64 fragment read hit 32 distinct even banks (72 gave a 2-way conflict)
  static constexpr int ROW_ELEMS = BI * LDR;
  static constexpr int T_ELEMS = DO * LDT;
};

// Register-staged tile loads: fetch() issues the global loads of the NEXT stage before the MFMAs of the current one,
// commit() writes them to LDS after the barrier, so HBM/L2 latency overlaps the compute.
// Row tile: [BI rows][D cols] sub-matrix (row stride gs) -> LDS [BI][LDR], columns D..DP zero filled.
template <int D, int NTH = FA<D>::NT> struct RowRegs { uint4 v[(FA<D>::BI * (FA<D>::DP / 8) + NTH - 1) / NTH]; };
// ONE != 0: column D of the tile (the first padding column, when the score padding DP > D provides one) holds the 16-bit constant ONE (1.0) instead
// of 0 -- as an A-operand row of the output products it makes the MFMA return the plain row sum of the B operand (attn_jvp_kernel's delta)
template <int D, int NTH = FA<D>::NT, unsigned ONE = 0>
__device__ inline void fetch_row(const bf16* __restrict__ src, long gs, RowRegs<D, NTH>& rg, int tid) {
  using F = FA<D>;
  constexpr int CPR = F::DP / 8, N = (F::BI * CPR + NTH - 1) / NTH;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int c = tid + i * NTH;
    const int r = c / CPR, cc = (c % CPR) * 8;
    rg.v[i] = make_uint4(ONE != 0 && cc == D ? ONE : 0u, 0, 0, 0);
    if (c < F::BI * CPR && cc < D) rg.v[i] = *reinterpret_cast<const uint4*>(src + (long)r * gs + cc);
  }
}
template <int D, int NTH = FA<D>::NT>
__device__ inline void commit_row(const RowRegs<D, NTH>& rg, bf16* lds, int tid) {
  using F = FA<D>;
  constexpr int CPR = F::DP / 8, N = (F::BI * CPR + NTH - 1) / NTH;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int c = tid + i * NTH;
    if (c < F::BI * CPR) *reinterpret_cast<uint4*>(lds + (c / CPR) * F::LDR + (c % CPR) * 8) = rg.v[i];
  }
}
// T tile: [D rows][BI cols] sub-matrix of a transposed copy (row stride gs) -> LDS [DO][LDT], rows D..DO zero filled.
template <int D, int NTH = FA<D>::NT> struct TRegs { uint4 v[(FA<D>::DO * (FA<D>::BI / 8) + NTH - 1) / NTH]; };
template <int D, int NTH = FA<D>::NT>
__device__ inline void fetch_t(const bf16* __restrict__ src, long gs, TRegs<D, NTH>& rg, int tid) {
  using F = FA<D>;
  constexpr int CPR = F::BI / 8, N = (F::DO * CPR + NTH - 1) / NTH;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int c = tid + i * NTH;
    const int r = c / CPR, cc = (c % CPR) * 8;
    rg.v[i] = make_uint4(0, 0, 0, 0);
    if (c < F::DO * CPR && r < D) rg.v[i] = *reinterpret_cast<const uint4*>(src + (long)r * gs + cc);
  }
}
template <int D, int NTH = FA<D>::NT>
__device__ inline void commit_t(const TRegs<D, NTH>& rg, bf16* lds, int tid) {
  using F = FA<D>;
  constexpr int CPR = F::BI / 8, N = (F::DO * CPR + NTH - 1) / NTH;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int c = tid + i * NTH;
    if (c < F::DO * CPR) {                      // rows are 8-byte (not 16-byte) aligned: two ds_write_b64
      bf16* dst = lds + (c / CPR) * F::LDT + (c % CPR) * 8;
      *reinterpret_cast<uint2*>(dst) = make_uint2(rg.v[i].x, rg.v[i].y);
      *reinterpret_cast<uint2*>(dst + 4) = make_uint2(rg.v[i].z, rg.v[i].w);
    }
  }
}
// B-operand fragments of 32 outer rows (row = lane&31), zero beyond D
template <int D>
__device__ inline void load_outer_frags(const bf16* __restrict__ rowptr, bf16x8* f, int lhi) {
  using F = FA<D>;
#pragma unroll
  for (int st = 0; st < F::NS; ++st) {
    const int col = st * 16 + lhi * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (col < D) v = *reinterpret_cast<const uint4*>(rowptr + col);
    f[st] = *reinterpret_cast<bf16x8*>(&v);
  }
}
__device__ inline bf16x8 lds_a_frag(const bf16* tile, int row, int ld, int col) {
  return *reinterpret_cast<const bf16x8*>(tile + row * ld + col);
}
// A fragment of a [d][row] tile for one 16-wide k-step: element j <-> inner row base + 4*lhi + (j&3) + 8*(j>>2)
__device__ inline bf16x8 lds_t_frag(const bf16* tile, int drow, int ld, int base, int lhi) {
  uint2 lo = *reinterpret_cast<const uint2*>(tile + drow * ld + base + 4 * lhi);
  uint2 hi = *reinterpret_cast<const uint2*>(tile + drow * ld + base + 8 + 4 * lhi);
  uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);
  return *reinterpret_cast<bf16x8*>(&v);
}
// The same A fragment ([32 d rows][16 inner rows], element j <-> inner row base + 4*lhi + (j&3) + 8*(j>>2)) read from a
// ROW tile [inner row][d] with gfx950's LDS transpose read: a 16-lane group hands ds_read_b64_tr_b16 the [4][16] block
// "4 inner rows x 16 d columns" (lane i: row i/4, columns 4(i%4)..+3) and lane i receives column i of it.  No per-head
// transposed copy of the operand is needed (neither in HBM nor as a second LDS tile).
typedef __attribute__((ext_vector_type(4))) short short4_;
__device__ inline bf16x8 lds_tr_frag(const bf16* tile, int ld, int base, int d0, int lane) {
  const int i16 = lane & 15, g = lane >> 4;
  const bf16* p = tile + (base + 4 * (g >> 1) + (i16 >> 2)) * ld + d0 + (g & 1) * 16 + (i16 & 3) * 4;
  short4_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_*)p);
  short4_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_*)(p + 8 * ld));
  uint2 l2 = *reinterpret_cast<uint2*>(&lo), h2 = *reinterpret_cast<uint2*>(&hi);
  uint4 v = make_uint4(l2.x, l2.y, h2.x, h2.y);
  return *reinterpret_cast<bf16x8*>(&v);
}
template <int FL>
__device__ inline void pack_b(const float* x, bf16x8* out) {   // 16 fp32 (acc register order) -> two B fragments
  out[0] = H16<FL>::pack8(x);                                   // v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32
  out[1] = H16<FL>::pack8(x + 8);
}
// XCD-aware block mapping.  Workgroups are dealt round-robin to the 8 XCDs in dispatch order (x fastest); the blocks of
// one (tangent, head) group stream the SAME inner tensors, so they should share an L2: with the natural order the 16
// blocks of a group land on all 8 XCDs and every L2 sees every group (measured 360-410 MB of HBM traffic per launch
// for ~80 MB of unique inputs).  Remap so that XCD x processes groups x, x+8, x+16, ... with all blocks of a group.
struct BlockXY { int x, y; };
__device__ inline BlockXY xcd_group_blocks(int on) {
  const int nx = gridDim.x, ny = gridDim.y;
  if (!on || (ny & 7)) return {(int)blockIdx.x, (int)blockIdx.y};
  const int lin = blockIdx.y * nx + blockIdx.x, xcd = lin & 7, slot = lin >> 3;
  return {slot % nx, (slot / nx) * 8 + xcd};
}

// shared-P key-major adjoint of the d = 40 layers: bit 1 (A/B switch: DPB_ATTN_SHARED, dpb_debug_set("attn_shared"); bit 0 was the tangent variants)
static int g_attn_shared = getenv("DPB_ATTN_SHARED") ? atoi(getenv("DPB_ATTN_SHARED")) : 2;
void attn_debug_shared(int bits) { g_attn_shared = bits; }

#define MFMA(a, b, c) H16<FL>::mfma(a, b, c)   // FL: the enclosing kernel's 16-bit flavour (0 bf16, 1 f16)

struct FusedArgs {
  const bf16 *Q, *K, *V, *O;          // primal [B][L][C]
  const bf16 *KT, *VT, *QT;           // primal per-head transposes [B][H][d][L]
  const float* stats;                 // primal row statistics [B][H][L][2] = (m, 1/l) of the scaled scores
  float* Drow;                        // scratch [nt][H][L]: D = gO . O per (cotangent, head, query) (shared-P key-major adjoint)
  float* Dout;                        // attn_adj_q_multi_kernel: where to leave its D_t (= Drow when the key-major shared kernel runs next), or nullptr
  const bf16 *dQ, *dK, *dV, *dVT;     // tangent inputs [nt][L][C], dV^T [nt][H][d][L]
  bf16* dO;                           // tangent output [nt][L][C]
  const bf16 *gO, *gOT;               // cotangent of the output [nt][L][C], gO^T [nt][H][d][L]
  bf16 *gQ, *gK, *gV;                 // cotangent outputs [nt][L][C]
  int accQ, accK, accV;               // accumulate into existing cotangents
  int xcd;                            // XCD-aware block order (off by default: measured neutral, MALL absorbs the re-reads; DPB_ATTN_XCD=1 turns it on)
  int L, C, Co, H, kps;               // C: row stride of Q/K/V-like tensors, Co: row stride of O-like tensors
  float scale;
};

// ------------------------------------------------------------------------------------------------ primal (flash forward)
// O = softmax(scale Q K^T) V with online softmax; also emits the row statistics (m, 1/l) the tangent / adjoint
// kernels need, so the L x L probabilities are never materialised for the fused layers.  Same tiling as below:
// lane <-> query, so the running max / sum and the rescale of the accumulator are register-local.
template <int D, int FL, int W = att_waves(D)>   // W waves (32 query rows each) per block
__global__ __launch_bounds__((FA<D, W>::NT)) void attn_fwd_kernel(FusedArgs a, bf16* O, float* stats_out) {
  using F = FA<D, W>;
  __shared__ __attribute__((aligned(16))) bf16 sm[2 * F::ROW_ELEMS];
  bf16* sK = sm; bf16* sV = sK + F::ROW_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const BlockXY blk = xcd_group_blocks(a.xcd);
  const int b = blk.y / a.H, h = blk.y % a.H;
  const int q_ = blk.x * (F::WAVES * 32) + wave * 32 + l31;
  const bool live = q_ < a.L;                 // short sequences (L < the block's rows): whole waves past the end compute on a clamped row and store nothing
  const int q = live ? q_ : a.L - 1;
  const long LC = (long)a.L * a.C;
  const bf16* Kp = a.K + b * LC + h * D;
  const bf16* Vp = a.V + b * LC + h * D;
  bf16x8 qf[F::NS];
  load_outer_frags<D>(a.Q + b * LC + (long)q * a.C + h * D, qf, lhi);
  const float c2 = a.scale * 1.44269504088896f;          // scores in log2 units (c2 > 0: the max commutes with the scaling)
  // Round 6 (the DDIM / guidance loop's kernel: 19 % of a batch-20 forward).  The loop was VALU-bound 2-3x over its MFMAs (per 128 keys: 28 MFMAs = 896
  // cycles against 250 VALU + 78 packed + 68 exp): (i) the running max is taken over the RAW scores and scaled once; (ii) DEFERRED rescale: the accumulators
  // keep a stale max until some query of the wave has outgrown it by more than 2^FWD_THR (wave-uniform branch; P = exp2(s - m_stale) <= 2^FWD_THR, exact in
  // exact arithmetic because l carries the same factor) -- after the first tiles the 32 multiplies per 32 keys are not executed at all; (iii) d = 40: the row
  // sum l comes out of the P V product itself (a ones column in V's first padding column, as attn_jvp_kernel's delta), not from 16 adds per 32 keys.
  // The emitted statistics are (m_stale, 1 / l): any consistent pair reproduces P = exp(s - m) / l in the tangent / adjoint kernels.
  constexpr float FWD_THR = 4.f;
  constexpr bool LSUM = F::DP > D;
  constexpr unsigned ONE16 = FL ? 0x3C00u : 0x3F80u, VONE = LSUM ? ONE16 : 0u;
  f32x16 acc[F::ND];
#pragma unroll
  for (int d = 0; d < F::ND; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  RowRegs<D, F::NT> rK, rV;
  fetch_row<D, F::NT>(Kp, a.C, rK, tid); fetch_row<D, F::NT, VONE>(Vp, a.C, rV, tid);
  for (int k0 = 0; k0 < a.L; k0 += F::BI) {
    __syncthreads();
    commit_row<D, F::NT>(rK, sK, tid); commit_row<D, F::NT>(rV, sV, tid);
    __syncthreads();
    if (k0 + F::BI < a.L) {
      const int k1 = k0 + F::BI;
      fetch_row<D, F::NT>(Kp + (long)k1 * a.C, a.C, rK, tid); fetch_row<D, F::NT, VONE>(Vp + (long)k1 * a.C, a.C, rV, tid);
    }
#pragma unroll
    for (int kb = 0; kb < F::BI / 32; ++kb) {
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int stp = 0; stp < F::NS; ++stp) s = MFMA(lds_a_frag(sK, kb * 32 + l31, F::LDR, stp * 16 + lhi * 8), qf[stp], s);
      float mx = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c2;       // the other 16 keys of this query live in lane ^ 32
      if (!__all(mx - m <= FWD_THR)) {                   // (first tile: m = -inf; a NaN score lands here too)
        const float mn = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
#pragma unroll
        for (int d = 0; d < F::ND; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[d][r] *= alpha;
        if (!LSUM) l *= alpha;
        m = mn;
      }
      float p[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c2, -m));
      if (!LSUM) {
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) ps += p[r];
        ps += __shfl_xor(ps, 32, 64);
        l += ps;
      }
      bf16x8 pb[2];
      pack_b<FL>(p, pb);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int d = 0; d < F::ND; ++d)
          acc[d] = MFMA(lds_tr_frag(sV, F::LDR, kb * 32 + ks * 16, d * 32, lane), pb[ks], acc[d]);
    }
  }
  if (LSUM) {                                            // column D of the output tile: register (D % 32 / 8) * 4 of tile D / 32 in the lanes with lhi == 0
    static_assert(!LSUM || (D % 8 == 0 && (D % 32) / 8 * 4 < 16), "ones column of V outside the accumulator layout");
    const float lv = acc[D / 32][(D % 32) / 8 * 4];
    const float lo = __shfl_xor(lv, 32, 64);             // by ALL lanes: a shuffle under `lhi != 0` would read its (inactive) source lanes as 0
    l = lhi == 0 ? lv : lo;
  }
  const float il = 1.f / l;
  if (lhi == 0 && live) {
    float* st = stats_out + (((long)b * a.H + h) * a.L + q) * 2;
    st[0] = m * 0.69314718055994531f;                     // natural-log units: the (stale) max the probabilities are taken against
    st[1] = il;
  }
  bf16* Op = O + b * (long)a.L * a.Co + (long)q * a.Co + h * D;
#pragma unroll
  for (int d = 0; d < F::ND; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = d * 32 + 8 * g + 4 * lhi;
      if (col < D && live) {
        store_out8<DPB_OUT_STORE_ATT>(Op + col, H16<FL>::pack2(acc[d][g * 4] * il, acc[d][g * 4 + 1] * il),
                                                         H16<FL>::pack2(acc[d][g * 4 + 2] * il, acc[d][g * 4 + 3] * il));
      }
    }
}

// ------------------------------------------------------------------------------------------------ tangent
template <int D, int FL, int W = att_waves(D)>   // W waves (32 outer rows each) per block
__global__ __launch_bounds__((FA<D, W>::NT), (W == 4 && D <= 40 ? 2 : 1)) void attn_jvp_kernel(FusedArgs a) {
  using F = FA<D, W>;
  __shared__ __attribute__((aligned(16))) bf16 sm[4 * F::ROW_ELEMS];
  bf16* sK = sm; bf16* sdK = sK + F::ROW_ELEMS; bf16* sV = sdK + F::ROW_ELEMS; bf16* sdV = sV + F::ROW_ELEMS;   // V^T / dV^T fragments: LDS transpose reads
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const BlockXY blk = xcd_group_blocks(a.xcd);
  const int j = blk.y / a.H, h = blk.y % a.H, b = j / a.kps;
  const int q_ = blk.x * (F::WAVES * 32) + wave * 32 + l31;
  const bool live = q_ < a.L;                 // short sequences (L < the block's rows): whole waves past the end compute on a clamped row and store nothing
  const int q = live ? q_ : a.L - 1;
  const long LC = (long)a.L * a.C;
  const bf16* Qp = a.Q + b * LC + h * D;
  const bf16* Kp = a.K + b * LC + h * D;
  const bf16* dQp = a.dQ + j * LC + h * D;
  const bf16* dKp = a.dK + j * LC + h * D;
  const bf16* Vp = a.V + b * LC + h * D;
  const bf16* dVp = a.dV + j * LC + h * D;
  bf16x8 qf[F::NS], dqf[F::NS];
  load_outer_frags<D>(Qp + (long)q * a.C, qf, lhi);
  load_outer_frags<D>(dQp + (long)q * a.C, dqf, lhi);
  const float* st = a.stats + (((long)b * a.H + h) * a.L + q) * 2;
  const float c2 = a.scale * 1.44269504088896f;
  const float m2 = st[0] * 1.44269504088896f - __builtin_log2f(st[1]);   // 1 / l folded into the exponent: p = exp2(c2 s - m2), no multiply per probability
  // delta_i = sum_j x_ij as a row of the output MFMAs: column D of the V tile (a padding column of the 32-wide output tiles) holds 1.0, so element
  // D of the accumulator row IS the sum of the (16-bit) x operand -- 16 dependent v_add per 32 keys less in a loop whose VALU time equals its MFMA time
  constexpr bool SUMROW = F::DO > D && ((D % 32) & 4) == 0;
  constexpr unsigned ONE16 = FL ? 0x3C00u : 0x3F80u, VONE = (SUMROW && F::DP > D) ? ONE16 : 0u;
  if constexpr (SUMROW && F::DP == D) {                 // no padding chunk in the committed rows: the column is written once
    for (int r = tid; r < F::BI; r += F::NT) {
      reinterpret_cast<unsigned short*>(sV)[r * F::LDR + D] = (unsigned short)ONE16;
      reinterpret_cast<unsigned short*>(sdV)[r * F::LDR + D] = 0;
    }
  }
  f32x16 acc[F::ND];
#pragma unroll
  for (int d = 0; d < F::ND; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  float delta = 0.f;
  RowRegs<D, F::NT> rK, rdK, rV, rdV;
  fetch_row<D, F::NT>(Kp, a.C, rK, tid); fetch_row<D, F::NT>(dKp, a.C, rdK, tid);
  fetch_row<D, F::NT, VONE>(Vp, a.C, rV, tid); fetch_row<D, F::NT>(dVp, a.C, rdV, tid);
  for (int k0 = 0; k0 < a.L; k0 += F::BI) {
    __syncthreads();                      // previous stage fully consumed
    commit_row<D, F::NT>(rK, sK, tid); commit_row<D, F::NT>(rdK, sdK, tid); commit_row<D, F::NT>(rV, sV, tid); commit_row<D, F::NT>(rdV, sdV, tid);
    __syncthreads();
    if (k0 + F::BI < a.L) {               // prefetch the next stage under this stage's MFMAs
      const int k1 = k0 + F::BI;
      fetch_row<D, F::NT>(Kp + (long)k1 * a.C, a.C, rK, tid); fetch_row<D, F::NT>(dKp + (long)k1 * a.C, a.C, rdK, tid);
      fetch_row<D, F::NT, VONE>(Vp + (long)k1 * a.C, a.C, rV, tid); fetch_row<D, F::NT>(dVp + (long)k1 * a.C, a.C, rdV, tid);
    }
    // Fragment reads run one step ahead of their MFMAs (explicit software pipeline; the compiler otherwise sinks every
    // ds_read to just before its use and each MFMA group waits out the LDS latency): the second-stage V^T / dV^T
    // fragments of a 32-key block are issued before its score MFMAs, the next block's K / dK fragments before the softmax.
    constexpr bool PIPE = D <= 40;                  // register budget: 56 more VGPRs at D = 40
    bf16x8 kf[F::NS], dkf[F::NS];
    auto load1 = [&](int kb) {
#pragma unroll
      for (int stp = 0; stp < F::NS; ++stp) {
        kf[stp] = lds_a_frag(sK, kb * 32 + l31, F::LDR, stp * 16 + lhi * 8);
        dkf[stp] = lds_a_frag(sdK, kb * 32 + l31, F::LDR, stp * 16 + lhi * 8);
      }
    };
    if constexpr (PIPE) load1(0);
#pragma unroll
    for (int kb = 0; kb < F::BI / 32; ++kb) {
      bf16x8 vf[2][F::ND], dvf[2][F::ND];
      if constexpr (PIPE) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int d = 0; d < F::ND; ++d) {
            vf[ks][d] = lds_tr_frag(sV, F::LDR, kb * 32 + ks * 16, d * 32, lane);
            dvf[ks][d] = lds_tr_frag(sdV, F::LDR, kb * 32 + ks * 16, d * 32, lane);
          }
        __builtin_amdgcn_sched_barrier(0);
      } else {
        load1(kb);
      }
      f32x16 s, ds;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = ds[r] = 0.f;
#pragma unroll
      for (int stp = 0; stp < F::NS; ++stp) {
        s = MFMA(kf[stp], qf[stp], s);
        ds = MFMA(kf[stp], dqf[stp], ds);
        if (!((DPB_ATT_ABL & 2) && D == 40 && stp == F::NS - 1)) ds = MFMA(dkf[stp], qf[stp], ds);
      }
      if constexpr (PIPE) {
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 1 < F::BI / 32) load1(kb + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      float p[16], x[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[r] = (DPB_ATT_ABL & 4) ? c2 * s[r] - m2 : __builtin_amdgcn_exp2f(c2 * s[r] - m2);
        x[r] = p[r] * (a.scale * ds[r]);
        if constexpr (!SUMROW) delta += x[r];
      }
      bf16x8 pb[2], xb[2];
      pack_b<FL>(p, pb);
      pack_b<FL>(x, xb);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int d = 0; d < F::ND; ++d) {
          if constexpr (!PIPE) {
            vf[ks][d] = lds_tr_frag(sV, F::LDR, kb * 32 + ks * 16, d * 32, lane);
            dvf[ks][d] = lds_tr_frag(sdV, F::LDR, kb * 32 + ks * 16, d * 32, lane);
          }
          if ((DPB_ATT_ABL & 1) && D == 40 && ks == 1 && d == 1) continue;
          acc[d] = MFMA(vf[ks][d], xb[ks], acc[d]);
          acc[d] = MFMA(dvf[ks][d], pb[ks], acc[d]);
        }
    }
  }
  if constexpr (SUMROW) delta = __shfl(acc[D / 32][((D % 32) & 3) + 4 * ((D % 32) >> 3)], l31, 64);   // output column D lives in the lhi = 0 lanes
  else delta += __shfl_xor(delta, 32, 64);
  // dO[q][dcol] = acc - delta * O[q][dcol];  lane owns query q, register r <-> dcol = d*32 + (r&3) + 8*(r>>2) + 4*lhi
  const long LCo = (long)a.L * a.Co;
  const bf16* Op = a.O + b * LCo + (long)q * a.Co + h * D;
  bf16* dOp = a.dO + j * LCo + (long)q * a.Co + h * D;
#pragma unroll
  for (int d = 0; d < F::ND; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = d * 32 + 8 * g + 4 * lhi;
      if (col < D && live) {
        uint2 ov = *reinterpret_cast<const uint2*>(Op + col);
        unsigned ow[2] = {ov.x, ov.y};
        unsigned w[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float o0 = H16<FL>::lo(ow[i]), o1 = H16<FL>::hi(ow[i]);
          float v0 = acc[d][g * 4 + 2 * i] - delta * o0, v1 = acc[d][g * 4 + 2 * i + 1] - delta * o1;
          w[i] = H16<FL>::pack2(v0, v1);
        }
        store_out8<DPB_OUT_STORE_ATT>(dOp + col, w[0], w[1]);
      }
    }
}

// ------------------------------------------------------------------------------------------------ adjoint, query-major (gQ)
template <int D, int FL>
__global__ __launch_bounds__(FA<D>::NT) void attn_adj_q_kernel(FusedArgs a) {
  using F = FA<D>;
  __shared__ __attribute__((aligned(16))) bf16 sm[2 * F::ROW_ELEMS];
  bf16* sK = sm; bf16* sV = sK + F::ROW_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const BlockXY blk = xcd_group_blocks(a.xcd);
  const int j = blk.y / a.H, h = blk.y % a.H, b = j / a.kps;
  const int q_ = blk.x * (F::WAVES * 32) + wave * 32 + l31;
  const bool live = q_ < a.L;                 // short sequences (L < the block's rows): whole waves past the end compute on a clamped row and store nothing
  const int q = live ? q_ : a.L - 1;
  const long LC = (long)a.L * a.C;
  const bf16* Kp = a.K + b * LC + h * D;
  const bf16* Vp = a.V + b * LC + h * D;
  bf16x8 qf[F::NS], gof[F::NS], of[F::NS];
  load_outer_frags<D>(a.Q + b * LC + (long)q * a.C + h * D, qf, lhi);
  const long LCo = (long)a.L * a.Co;
  load_outer_frags<D>(a.gO + j * LCo + (long)q * a.Co + h * D, gof, lhi);
  load_outer_frags<D>(a.O + b * LCo + (long)q * a.Co + h * D, of, lhi);
  float Dq = 0.f;   // D_q = gO_q . O_q  (each lane holds half of the columns)
#pragma unroll
  for (int stp = 0; stp < F::NS; ++stp) {
    const unsigned short* g16 = reinterpret_cast<const unsigned short*>(&gof[stp]);
    const unsigned short* o16 = reinterpret_cast<const unsigned short*>(&of[stp]);
#pragma unroll
    for (int e = 0; e < 8; ++e) Dq += H16<FL>::up(g16[e]) * H16<FL>::up(o16[e]);
  }
  Dq += __shfl_xor(Dq, 32, 64);
  const float* st = a.stats + (((long)b * a.H + h) * a.L + q) * 2;
  const float c2 = a.scale * 1.44269504088896f;
  const float m2 = st[0] * 1.44269504088896f - __builtin_log2f(st[1]);   // 1 / l folded into the exponent
  f32x16 acc[F::ND];
#pragma unroll
  for (int d = 0; d < F::ND; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  RowRegs<D> rK, rV;
  fetch_row<D>(Kp, a.C, rK, tid); fetch_row<D>(Vp, a.C, rV, tid);
  for (int k0 = 0; k0 < a.L; k0 += F::BI) {
    __syncthreads();
    commit_row<D>(rK, sK, tid); commit_row<D>(rV, sV, tid);
    __syncthreads();
    if (k0 + F::BI < a.L) {
      const int k1 = k0 + F::BI;
      fetch_row<D>(Kp + (long)k1 * a.C, a.C, rK, tid); fetch_row<D>(Vp + (long)k1 * a.C, a.C, rV, tid);
    }
#pragma unroll
    for (int kb = 0; kb < F::BI / 32; ++kb) {
      f32x16 s, gp;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = gp[r] = 0.f;
#pragma unroll
      for (int stp = 0; stp < F::NS; ++stp) {
        s = MFMA(lds_a_frag(sK, kb * 32 + l31, F::LDR, stp * 16 + lhi * 8), qf[stp], s);
        gp = MFMA(lds_a_frag(sV, kb * 32 + l31, F::LDR, stp * 16 + lhi * 8), gof[stp], gp);
      }
      float gs[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) gs[r] = __builtin_amdgcn_exp2f(c2 * s[r] - m2) * (gp[r] - Dq);
      bf16x8 gsb[2];
      pack_b<FL>(gs, gsb);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int d = 0; d < F::ND; ++d)
          acc[d] = MFMA(lds_tr_frag(sK, F::LDR, kb * 32 + ks * 16, d * 32, lane), gsb[ks], acc[d]);
    }
  }
  bf16* gQp = a.gQ + j * LC + (long)q * a.C + h * D;
#pragma unroll
  for (int d = 0; d < F::ND; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = d * 32 + 8 * g + 4 * lhi;
      if (col < D && live) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = a.scale * acc[d][g * 4 + i];
        if (a.accQ) {
          uint2 ov = *reinterpret_cast<const uint2*>(gQp + col);
          v[0] += H16<FL>::lo(ov.x); v[1] += H16<FL>::hi(ov.x);
          v[2] += H16<FL>::lo(ov.y); v[3] += H16<FL>::hi(ov.y);
        }
        store_out8<DPB_OUT_STORE_ATT>(gQp + col, H16<FL>::pack2(v[0], v[1]), H16<FL>::pack2(v[2], v[3]));
      }
    }
}

// ------------------------------------------------------------------------------------------------ adjoint, query-major, all cotangents of a sample in one block
// The probabilities P depend on the primal sample only, yet every (cotangent, head) block of the kernel above recomputes
// S = Q K^T and exp() for them.  Here one block owns 128 queries of one (sample, head) and carries up to TJ cotangents:
// S and P once per tile, then per cotangent gP_t = gO_t V^T, gS_t = P o (gP_t - D_t), gQ_t += gS_t K.  Per 64-key tile and
// wave: 6 + 14*TJ MFMAs instead of 20*TJ, 10 LDS fragment reads instead of 20*TJ, 32 exp instead of 32*TJ; the TJ
// independent cotangent streams give the scheduler MFMA work to overlap with each other's softmax arithmetic.
// (The same construction for the tangent kernel and for the key-major adjoint needs a dK / dV^T resp. gO / gO^T tile per
// tangent in every stage; with one 4-wave block per CU its 49 KB-per-32-keys stream is latency-bound -- measured 760-870 us
// against 430 us for attn_jvp_kernel -- so only the query-major adjoint, whose streamed tiles are all shared, uses it.)
// 4 waves (one per SIMD, up to 512 registers each: TJ*ND accumulators + TJ*NS cotangent fragments stay in registers).
template <int D, int TJ, int FL>
__global__ __launch_bounds__(256) void attn_adj_q_multi_kernel(FusedArgs a) {
  using F = FA<D>;
  constexpr int NTH = 256;
  __shared__ __attribute__((aligned(16))) bf16 sm[2 * F::ROW_ELEMS];
  bf16* sK = sm; bf16* sV = sK + F::ROW_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int ngrp = (a.kps + TJ - 1) / TJ;
  const int grp = blockIdx.y % ngrp, bh = blockIdx.y / ngrp, b = bh / a.H, h = bh % a.H;
  const int j0 = b * a.kps + grp * TJ, nj = min(TJ, a.kps - grp * TJ);
  const int q = blockIdx.x * 128 + wave * 32 + l31;
  const long LC = (long)a.L * a.C, LCo = (long)a.L * a.Co;
  const bf16* Kp = a.K + b * LC + h * D;
  const bf16* Vp = a.V + b * LC + h * D;
  // (Folding D_t into the gP product -- 1.0 in the padding columns D, D + 1 of the V tile, -D_t split into two 16-bit halves in the cotangent fragments,
  // so that the MFMA returns gP_t - D_t -- removes 16 subtracts per cotangent and 32 keys and measured neutral: 8.742 / 8.736 vs 8.737 ms per iteration.)
  bf16x8 qf[F::NS], gof[TJ][F::NS];
  load_outer_frags<D>(a.Q + b * LC + (long)q * a.C + h * D, qf, lhi);
  float Dq[TJ];
  {
    bf16x8 of[F::NS];
    load_outer_frags<D>(a.O + b * LCo + (long)q * a.Co + h * D, of, lhi);
#pragma unroll
    for (int t = 0; t < TJ; ++t) {
      Dq[t] = 0.f;
      if (t < nj) {
        load_outer_frags<D>(a.gO + (long)(j0 + t) * LCo + (long)q * a.Co + h * D, gof[t], lhi);
#pragma unroll
        for (int stp = 0; stp < F::NS; ++stp) {
          const unsigned short* g16 = reinterpret_cast<const unsigned short*>(&gof[t][stp]);
          const unsigned short* o16 = reinterpret_cast<const unsigned short*>(&of[stp]);
#pragma unroll
          for (int e = 0; e < 8; ++e) Dq[t] += H16<FL>::up(g16[e]) * H16<FL>::up(o16[e]);
        }
        Dq[t] += __shfl_xor(Dq[t], 32, 64);
        if (a.Dout && lhi == 0) a.Dout[((long)(j0 + t) * a.H + h) * a.L + q] = Dq[t];   // D_t for the key-major kernel launched next (no row-dot pre-pass)
      } else {
#pragma unroll
        for (int stp = 0; stp < F::NS; ++stp) gof[t][stp] = bf16x8{};
      }
    }
  }
  const float* st = a.stats + (((long)b * a.H + h) * a.L + q) * 2;
  const float c2 = a.scale * 1.44269504088896f;
  const float m2 = st[0] * 1.44269504088896f - __builtin_log2f(st[1]);   // 1 / l folded into the exponent
  f32x16 acc[TJ][F::ND];
#pragma unroll
  for (int t = 0; t < TJ; ++t)
#pragma unroll
    for (int d = 0; d < F::ND; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][d][r] = 0.f;
  RowRegs<D, NTH> rK, rV;
  fetch_row<D, NTH>(Kp, a.C, rK, tid); fetch_row<D, NTH>(Vp, a.C, rV, tid);
  for (int k0 = 0; k0 < a.L; k0 += F::BI) {
    __syncthreads();
    commit_row<D, NTH>(rK, sK, tid); commit_row<D, NTH>(rV, sV, tid);
    __syncthreads();
    if (k0 + F::BI < a.L) {
      const int k1 = k0 + F::BI;
      fetch_row<D, NTH>(Kp + (long)k1 * a.C, a.C, rK, tid); fetch_row<D, NTH>(Vp + (long)k1 * a.C, a.C, rV, tid);
    }
#pragma unroll
    for (int kb = 0; kb < F::BI / 32; ++kb) {
      bf16x8 vf[F::NS], ktf[2][F::ND];
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int stp = 0; stp < F::NS; ++stp) {
        s = MFMA(lds_a_frag(sK, kb * 32 + l31, F::LDR, stp * 16 + lhi * 8), qf[stp], s);
        vf[stp] = lds_a_frag(sV, kb * 32 + l31, F::LDR, stp * 16 + lhi * 8);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int d = 0; d < F::ND; ++d) ktf[ks][d] = lds_tr_frag(sK, F::LDR, kb * 32 + ks * 16, d * 32, lane);   // K^T from the K row tile
      float p[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(c2 * s[r] - m2);
#pragma unroll
      for (int t = 0; t < TJ; ++t) {
        if (t < nj) {
          f32x16 gp;
#pragma unroll
          for (int r = 0; r < 16; ++r) gp[r] = 0.f;
#pragma unroll
          for (int stp = 0; stp < F::NS; ++stp) gp = MFMA(vf[stp], gof[t][stp], gp);
          float gs[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) gs[r] = p[r] * (gp[r] - Dq[t]);
          bf16x8 gsb[2];
          pack_b<FL>(gs, gsb);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int d = 0; d < F::ND; ++d) {
              if ((DPB_ATT_ABL & 1) && D == 40 && ks == 1 && d == 1) continue;
              acc[t][d] = MFMA(ktf[ks][d], gsb[ks], acc[t][d]);
            }
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < TJ; ++t) {
    if (t >= nj) continue;
    bf16* gQp = a.gQ + (long)(j0 + t) * LC + (long)q * a.C + h * D;
#pragma unroll
    for (int d = 0; d < F::ND; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = d * 32 + 8 * g + 4 * lhi;
        if (col < D) {
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = a.scale * acc[t][d][g * 4 + i];
          if (a.accQ) {
            uint2 ov = *reinterpret_cast<const uint2*>(gQp + col);
            v[0] += H16<FL>::lo(ov.x); v[1] += H16<FL>::hi(ov.x);
            v[2] += H16<FL>::lo(ov.y); v[3] += H16<FL>::hi(ov.y);
          }
          store_out8<DPB_OUT_STORE_ATT>(gQp + col, H16<FL>::pack2(v[0], v[1]), H16<FL>::pack2(v[2], v[3]));
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------ adjoint, key-major (gK, gV)
template <int D, int FL, int W = att_waves(D)>
__global__ __launch_bounds__((FA<D, W>::NT), (D <= 40 ? 2 : 1)) void attn_adj_kv_kernel(FusedArgs a) {
  using F = FA<D, W>;
  __shared__ __attribute__((aligned(16))) bf16 sm[2 * F::ROW_ELEMS];
  __shared__ float sstat[2][F::BI];          // m*log2e - log2(1/l), D per query of the stage
  bf16* sQ = sm; bf16* sgO = sQ + F::ROW_ELEMS;     // Q^T / gO^T fragments come from these row tiles by LDS transpose reads
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const BlockXY blk = xcd_group_blocks(a.xcd);
  const int j = blk.y / a.H, h = blk.y % a.H, b = j / a.kps;
  const int key_ = blk.x * (F::WAVES * 32) + wave * 32 + l31;
  const bool live = key_ < a.L;               // short sequences: whole waves past the end compute on a clamped row and store nothing
  const int key = live ? key_ : a.L - 1;
  const long LC = (long)a.L * a.C;
  const bf16* Qp = a.Q + b * LC + h * D;
  const long LCo = (long)a.L * a.Co;
  const bf16* Op = a.O + b * LCo + h * D;
  const bf16* gOp = a.gO + j * LCo + h * D;
  const float* stp_ = a.stats + ((long)b * a.H + h) * a.L * 2;
  bf16x8 kf[F::NS], vf[F::NS];
  load_outer_frags<D>(a.K + b * LC + (long)key * a.C + h * D, kf, lhi);
  load_outer_frags<D>(a.V + b * LC + (long)key * a.C + h * D, vf, lhi);
  const float c2 = a.scale * 1.44269504088896f;
  f32x16 accK[F::ND], accV[F::ND];
#pragma unroll
  for (int d = 0; d < F::ND; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) accK[d][r] = accV[d][r] = 0.f;
  RowRegs<D, F::NT> rQ, rgO;
  float st0 = 0.f, st2 = 0.f;
  auto fetch_stats = [&](int q0) {       // per-query statistics of a stage, D_q = gO_q . O_q
    if (tid < F::BI) {
      const int qq = q0 + tid;
      st0 = stp_[2 * qq] * 1.44269504088896f - __builtin_log2f(stp_[2 * qq + 1]);   // 1 / l folded into the exponent
      float dq = 0.f;
      for (int c = 0; c < D; c += 8) {
        float g8[8], o8[8];
        H16<FL>::load8(gOp + (long)qq * a.Co + c, g8);
        H16<FL>::load8(Op + (long)qq * a.Co + c, o8);
#pragma unroll
        for (int e = 0; e < 8; ++e) dq += g8[e] * o8[e];
      }
      st2 = dq;
    }
  };
  fetch_row<D, F::NT>(Qp, a.C, rQ, tid); fetch_row<D, F::NT>(gOp, a.Co, rgO, tid);
  fetch_stats(0);
  for (int q0 = 0; q0 < a.L; q0 += F::BI) {
    __syncthreads();
    commit_row<D, F::NT>(rQ, sQ, tid); commit_row<D, F::NT>(rgO, sgO, tid);
    if (tid < F::BI) { sstat[0][tid] = st0; sstat[1][tid] = st2; }
    __syncthreads();
    if (q0 + F::BI < a.L) {
      const int q1 = q0 + F::BI;
      fetch_row<D, F::NT>(Qp + (long)q1 * a.C, a.C, rQ, tid); fetch_row<D, F::NT>(gOp + (long)q1 * a.Co, a.Co, rgO, tid);
      fetch_stats(q1);
    }
#pragma unroll
    for (int qb = 0; qb < F::BI / 32; ++qb) {
      f32x16 s, gp;     // [query = register row][key = lane]
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = gp[r] = 0.f;
#pragma unroll
      for (int stp = 0; stp < F::NS; ++stp) {
        s = MFMA(lds_a_frag(sQ, qb * 32 + l31, F::LDR, stp * 16 + lhi * 8), kf[stp], s);
        gp = MFMA(lds_a_frag(sgO, qb * 32 + l31, F::LDR, stp * 16 + lhi * 8), vf[stp], gp);
      }
      float p[16], gs[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        p[r] = __builtin_amdgcn_exp2f(c2 * s[r] - sstat[0][qi]);
        gs[r] = p[r] * (gp[r] - sstat[1][qi]);
      }
      bf16x8 pb[2], gsb[2];
      pack_b<FL>(p, pb);
      pack_b<FL>(gs, gsb);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int d = 0; d < F::ND; ++d) {
          accV[d] = MFMA(lds_tr_frag(sgO, F::LDR, qb * 32 + ks * 16, d * 32, lane), pb[ks], accV[d]);
          accK[d] = MFMA(lds_tr_frag(sQ, F::LDR, qb * 32 + ks * 16, d * 32, lane), gsb[ks], accK[d]);
        }
    }
  }
  bf16* gKp = a.gK + j * LC + (long)key * a.C + h * D;
  bf16* gVp = a.gV + j * LC + (long)key * a.C + h * D;
#pragma unroll
  for (int d = 0; d < F::ND; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = d * 32 + 8 * g + 4 * lhi;
      if (col < D && live) {
        float vk[4], vv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { vk[i] = a.scale * accK[d][g * 4 + i]; vv[i] = accV[d][g * 4 + i]; }
        if (a.accK) {
          uint2 ov = *reinterpret_cast<const uint2*>(gKp + col);
          vk[0] += H16<FL>::lo(ov.x); vk[1] += H16<FL>::hi(ov.x);
          vk[2] += H16<FL>::lo(ov.y); vk[3] += H16<FL>::hi(ov.y);
        }
        if (a.accV) {
          uint2 ov = *reinterpret_cast<const uint2*>(gVp + col);
          vv[0] += H16<FL>::lo(ov.x); vv[1] += H16<FL>::hi(ov.x);
          vv[2] += H16<FL>::lo(ov.y); vv[3] += H16<FL>::hi(ov.y);
        }
        store_out8<DPB_OUT_STORE_ATT>(gKp + col, H16<FL>::pack2(vk[0], vk[1]), H16<FL>::pack2(vk[2], vk[3]));
        store_out8<DPB_OUT_STORE_ATT>(gVp + col, H16<FL>::pack2(vv[0], vv[1]), H16<FL>::pack2(vv[2], vv[3]));
      }
    }
}

// D[j][h][q] = gO_j[q][h] . O_b[q][h]  (one thread per (cotangent, query, head); 2 x D/8 16-byte loads)
template <int D, int FL>
__global__ __launch_bounds__(256) void attn_rowdot_kernel(FusedArgs a, int nt) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)nt * a.L * a.H) return;
  const int h = (int)(idx % a.H);
  const long jq = idx / a.H;
  const int q = (int)(jq % a.L), j = (int)(jq / a.L), b = j / a.kps;
  const bf16* gp_ = a.gO + ((long)j * a.L + q) * a.Co + h * D;
  const bf16* op_ = a.O + ((long)b * a.L + q) * a.Co + h * D;
  float dq = 0.f;
#pragma unroll
  for (int c = 0; c < D; c += 8) {
    float g8[8], o8[8];
    H16<FL>::load8(gp_ + c, g8);
    H16<FL>::load8(op_ + c, o8);
#pragma unroll
    for (int e = 0; e < 8; ++e) dq += g8[e] * o8[e];
  }
  a.Drow[((long)j * a.H + h) * a.L + q] = dq;
}

// ------------------------------------------------------------------------------------------------ adjoint, key-major, probabilities shared by the cotangents of a sample
// The probabilities P depend on the primal sample only, yet every (cotangent, head) block of attn_adj_kv_kernel recomputes S = Q K^T and its exp().
// Here one block owns 64 KEYS of one (sample, head) and ALL TJ cotangents of the sample: 2 PRODUCER waves (one per 32-key group, K fragments in
// registers) compute P^T tile by tile -- Q fragments and the row statistics read straight from global / L2 in MFMA operand layout, one stage ahead --
// and publish it as packed 16-bit B fragments in LDS; 2 x TJ CONSUMER waves (key group g, cotangent t: V fragments in registers) read it back (same
// lane <-> key layout, 2 ds_read_b128), form gS_t = P o (gO_t V^T - D_t) and accumulate gV_t += P^T gO_t, gK_t += scale gS_t^T Q.  Per 32 queries
// and cotangent: 11 MFMAs + ~230 VALU cycles instead of 14 + 640; the 64 x 64-level grid at k = 5 is 512 blocks = exactly two rounds of the 256 CUs.
// Streams 1 + TJ row tiles of 64 queries per stage (Q, gO_t) through a double-buffered LDS ring (the LDS writes of stage s+1 run under the arithmetic
// of stage s; one barrier per stage) and the TJ x 64 row dots D_t = gO_t . O (attn_rowdot_kernel).  Measured 421 -> 326 us per launch (k = 5).
// Each role runs its OWN copy of the stage loop (same barrier sequence): the producers' look-ahead registers and the consumers' accumulators never
// share an allocation (one loop for both: 303 spilled VGPRs).
// (Sharing P in the TANGENT kernel was built twice in round 3 and removed: (i) this producer / consumer construction with 2 + 2 TJ tiles per stage
// (K, V, dK_t, dV_t): 420 us single-buffered, 435 us double-buffered against 398 us for attn_jvp_kernel -- the per-tangent dK_t / dV_t fragment
// reads keep the LDS pipe as busy as before while 12-wave barriers add waits (SQ_WAIT_ANY 13 % -> 44 %); (ii) all TJ tangents in one wave, one
// 4-wave block per CU like attn_adj_q_multi_kernel (3 + 14 TJ MFMAs per 32 keys, dK_t fragments read one tangent ahead): 256 + 256 registers with
// 10 spilled, ~530 us -- with one wave per SIMD nothing hides the 7 per-tangent fragment reads per 14 MFMAs.  LDS row paddings of 16 / 24 elements
// instead of 8: no gain either (9.02 / 9.28 vs 8.99 ms per iteration).)
// (D_t carried by the gP product -- -D_t as two 16-bit halves in the padding columns of the gO_t tiles, 1.0 in the V fragments -- instead of the sD table
// and a subtract per probability: 2-12 spilled VGPRs at the 168-register budget of 12 waves, 8.858 vs 8.737 ms per iteration; removed.)
template <int D, int TJ> struct SHK {
  static constexpr int QG = 2, NP = QG, NW = NP + QG * TJ, NT = NW * 64;
  static constexpr int BI = 64;                                     // queries per LDS stage (two 32-query blocks); the stage ring is double-buffered
  static constexpr int NS = (D + 15) / 16, DP = NS * 16, ND = (D + 31) / 32, DO = ND * 32;
  static constexpr int LDR = (DP > DO ? DP : DO) + DPB_ATT_PAD, ROW_ELEMS = BI * LDR;
  static constexpr int NTILE = 1 + TJ, CPR = DP / 8;                // tile 0 Q, 1 + t gO_t
  static constexpr int CPT = BI * CPR;                              // 16-byte chunks per tile
  static constexpr int NLD = NTILE * CPT / NT;                      // loads per thread and stage (chunk c = tid + NT i)
  static_assert(NTILE * CPT % NT == 0, "whole loads per thread");
};

template <int D, int TJ, int FL>
__global__ __launch_bounds__((SHK<D, TJ>::NT)) void attn_adj_kv_shared_kernel(FusedArgs a) {
  using S = SHK<D, TJ>;
  __shared__ __attribute__((aligned(16))) bf16 sm[2 * S::NTILE * S::ROW_ELEMS];    // [stage parity][tile][query][d]
  __shared__ __attribute__((aligned(16))) uint4 sP[2][2][S::QG][2][64];            // [stage parity][32-query block][key group][fragment][lane]
  __shared__ float sD[2][TJ][S::BI];                                                // [stage parity] D_t of the stage's queries
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int ngrp = (a.kps + TJ - 1) / TJ;
  const int grp = blockIdx.y % ngrp, bh = blockIdx.y / ngrp, b = bh / a.H, h = bh % a.H;
  const int j0 = b * a.kps + grp * TJ, nj = min(TJ, a.kps - grp * TJ);
  const long LC = (long)a.L * a.C, LCo = (long)a.L * a.Co;
  const bf16* Qp = a.Q + b * LC + h * D;
  const float* stp_ = a.stats + ((long)b * a.H + h) * a.L * 2;
  // ---- tile loader: chunk c = tid + NT i  ->  (tile, row, 16-byte column), fixed per thread; tile 0 = Q (row stride C), tile 1 + t = gO_t (row stride Co)
  const bf16* lsrc[S::NLD];
  long lstr[S::NLD];
  int ldst[S::NLD];
#pragma unroll
  for (int i = 0; i < S::NLD; ++i) {
    const int c = tid + i * S::NT, tile = c / S::CPT, lrow = (c % S::CPT) / S::CPR, lcc = (c % S::CPR) * 8;
    const bf16* base = nullptr;
    lstr[i] = tile == 0 ? a.C : a.Co;
    if (tile == 0) base = Qp;
    else if (tile - 1 < nj) base = a.gO + (long)(j0 + tile - 1) * LCo + h * D;
    lsrc[i] = (base && lcc < D) ? base + (long)lrow * lstr[i] + lcc : nullptr;
    ldst[i] = tile * S::ROW_ELEMS + lrow * S::LDR + lcc;
  }
  uint4 rg[S::NLD];
  float dreg = 0.f;
  const int dt_ = tid / S::BI, dq_ = tid % S::BI;                                 // row-dot duty: thread < TJ * BI forwards D_{dt_}(q0 + dq_)
  auto fetch = [&](int q0) {
#pragma unroll
    for (int i = 0; i < S::NLD; ++i) rg[i] = lsrc[i] ? *reinterpret_cast<const uint4*>(lsrc[i] + (long)q0 * lstr[i]) : make_uint4(0, 0, 0, 0);
    if (dt_ < nj) dreg = a.Drow[((long)(j0 + dt_) * a.H + h) * a.L + q0 + dq_];   // D_t = gO_t . O per query (attn_rowdot_kernel)
  };
  auto commit = [&](int par) {
#pragma unroll
    for (int i = 0; i < S::NLD; ++i) *reinterpret_cast<uint4*>(sm + par * S::NTILE * S::ROW_ELEMS + ldst[i]) = rg[i];
    if (dt_ < TJ) sD[par][dt_][dq_] = dreg;
  };
  // ---- roles (own copy of the stage loop each, same barrier sequence)
  const bool producer = wave < S::NP;
  const int g = producer ? wave : (wave - S::NP) % S::QG;                         // 32-key group
  const int t = producer ? 0 : (wave - S::NP) / S::QG;                            // cotangent slot of a consumer wave
  const int key = blockIdx.x * (S::QG * 32) + g * 32 + l31;
  const float c2 = a.scale * 1.44269504088896f;
  const int nst = a.L / S::BI;
  bf16x8 of[S::NS];                                                                // producer: K fragments; consumer: V fragments (outer rows = keys)
  load_outer_frags<D>((producer ? a.K : a.V) + b * LC + (long)key * a.C + h * D, of, lhi);
  fetch(0);
  commit(0);
  if (nst > 1) fetch(S::BI);
  if (producer) {
    bf16x8 qfn[2][S::NS];                                                          // Q fragments (A operand) of the stage computed next
    float4 stn[2][8];                                                              // ... and its row statistics: (m, 1/l) of 16 queries per 32-query block
    auto load_q = [&](int q0) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
        for (int st_ = 0; st_ < S::NS; ++st_) {
          const int col = st_ * 16 + lhi * 8;
          uint4 v = make_uint4(0, 0, 0, 0);
          if (col < D) v = *reinterpret_cast<const uint4*>(Qp + (long)(q0 + qb * 32 + l31) * a.C + col);
          qfn[qb][st_] = *reinterpret_cast<bf16x8*>(&v);
        }
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {                                           // register rows 4gg..4gg+3 <-> queries q0 + qb*32 + 8gg + 4lhi + 0..3
          const float* sp = stp_ + 2 * (long)(q0 + qb * 32 + 8 * gg + 4 * lhi);
          stn[qb][2 * gg] = *reinterpret_cast<const float4*>(sp);
          stn[qb][2 * gg + 1] = *reinterpret_cast<const float4*>(sp + 4);
        }
      }
    };
    auto produce = [&](int par) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
        for (int st_ = 0; st_ < S::NS; ++st_) sc = MFMA(qfn[qb][st_], of[st_], sc);   // [query = register row][key = lane]
        float p[16];
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
          const float4 s0 = stn[qb][2 * gg], s1 = stn[qb][2 * gg + 1];
          p[4 * gg] = __builtin_amdgcn_exp2f(c2 * sc[4 * gg] - s0.x * 1.44269504088896f) * s0.y;
          p[4 * gg + 1] = __builtin_amdgcn_exp2f(c2 * sc[4 * gg + 1] - s0.z * 1.44269504088896f) * s0.w;
          p[4 * gg + 2] = __builtin_amdgcn_exp2f(c2 * sc[4 * gg + 2] - s1.x * 1.44269504088896f) * s1.y;
          p[4 * gg + 3] = __builtin_amdgcn_exp2f(c2 * sc[4 * gg + 3] - s1.z * 1.44269504088896f) * s1.w;
        }
        bf16x8 pb[2];
        pack_b<FL>(p, pb);
        sP[par][qb][g][0][lane] = *reinterpret_cast<uint4*>(&pb[0]);
        sP[par][qb][g][1][lane] = *reinterpret_cast<uint4*>(&pb[1]);
      }
    };
    load_q(0);
    produce(0);
    if (nst > 1) load_q(S::BI);
    __syncthreads();
    for (int s = 0; s < nst; ++s) {
      if (s + 1 < nst) {
        commit((s + 1) & 1);
        if (s + 2 < nst) fetch((s + 2) * S::BI);
        produce((s + 1) & 1);
        if (s + 2 < nst) load_q((s + 2) * S::BI);
      }
      __syncthreads();
    }
    return;
  }
  f32x16 accK[S::ND], accV[S::ND];
#pragma unroll
  for (int d = 0; d < S::ND; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) accK[d][r] = accV[d][r] = 0.f;
  __syncthreads();
  for (int s = 0; s < nst; ++s) {
    if (s + 1 < nst) {
      commit((s + 1) & 1);
      if (s + 2 < nst) fetch((s + 2) * S::BI);
    }
    if (t < nj) {
      const bf16* sQ = sm + (s & 1) * S::NTILE * S::ROW_ELEMS;
      const bf16* sgO = sQ + (1 + t) * S::ROW_ELEMS;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        f32x16 gp;
#pragma unroll
        for (int r = 0; r < 16; ++r) gp[r] = 0.f;
#pragma unroll
        for (int st_ = 0; st_ < S::NS; ++st_) gp = MFMA(lds_a_frag(sgO, qb * 32 + l31, S::LDR, st_ * 16 + lhi * 8), of[st_], gp);
        uint4 pw[2] = {sP[s & 1][qb][g][0][lane], sP[s & 1][qb][g][1][lane]};
        const unsigned w[8] = {pw[0].x, pw[0].y, pw[0].z, pw[0].w, pw[1].x, pw[1].y, pw[1].z, pw[1].w};
        float gs[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int r0 = 2 * i, r1 = 2 * i + 1;
          const int q0i = qb * 32 + (r0 & 3) + 8 * (r0 >> 2) + 4 * lhi;
          gs[r0] = H16<FL>::lo(w[i]) * (gp[r0] - sD[s & 1][t][q0i]);
          gs[r1] = H16<FL>::hi(w[i]) * (gp[r1] - sD[s & 1][t][q0i + 1]);
        }
        bf16x8 gsb[2];
        pack_b<FL>(gs, gsb);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int d = 0; d < S::ND; ++d) {
            if ((DPB_ATT_ABL & 1) && D == 40 && ks == 1 && d == 1) continue;
            accV[d] = MFMA(lds_tr_frag(sgO, S::LDR, qb * 32 + ks * 16, d * 32, lane), *reinterpret_cast<bf16x8*>(&pw[ks]), accV[d]);
            accK[d] = MFMA(lds_tr_frag(sQ, S::LDR, qb * 32 + ks * 16, d * 32, lane), gsb[ks], accK[d]);
          }
      }
    }
    __syncthreads();
  }
  if (t >= nj) return;
  bf16* gKp = a.gK + (long)(j0 + t) * LC + (long)key * a.C + h * D;
  bf16* gVp = a.gV + (long)(j0 + t) * LC + (long)key * a.C + h * D;
#pragma unroll
  for (int d = 0; d < S::ND; ++d)
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      const int col = d * 32 + 8 * gg + 4 * lhi;
      if (col < D) {
        float vk[4], vv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { vk[i] = a.scale * accK[d][gg * 4 + i]; vv[i] = accV[d][gg * 4 + i]; }
        if (a.accK) {
          uint2 ov = *reinterpret_cast<const uint2*>(gKp + col);
          vk[0] += H16<FL>::lo(ov.x); vk[1] += H16<FL>::hi(ov.x);
          vk[2] += H16<FL>::lo(ov.y); vk[3] += H16<FL>::hi(ov.y);
        }
        if (a.accV) {
          uint2 ov = *reinterpret_cast<const uint2*>(gVp + col);
          vv[0] += H16<FL>::lo(ov.x); vv[1] += H16<FL>::hi(ov.x);
          vv[2] += H16<FL>::lo(ov.y); vv[3] += H16<FL>::hi(ov.y);
        }
        store_out8<DPB_OUT_STORE_ATT>(gKp + col, H16<FL>::pack2(vk[0], vk[1]), H16<FL>::pack2(vk[2], vk[3]));
        store_out8<DPB_OUT_STORE_ATT>(gVp + col, H16<FL>::pack2(vv[0], vv[1]), H16<FL>::pack2(vv[2], vv[3]));
      }
    }
}

// ------------------------------------------------------------------------------------------------ constant-K/V (cross) attention
// Text-conditioned layers: K and V are projections of the prompt embedding, which does not depend on x_t, so only the
// query side carries a tangent / cotangent and both passes are the SAME row-local map
//     Y_i = c_out * sum_j W_ij B_j ,   W = P o (c_in X A^T - delta) ,   delta_i = sum_j P_ij (c_in X A^T)_ij ,
// tangent: (X, A, B, c_in, c_out) = (dQ, K, V, scale, 1) -> dO;   adjoint: (gO, V, K, 1, scale) -> gQ.
// The 77 keys fit one tile (padded to 96 = three 32-key MFMA blocks, masked), so P is recomputed in registers from Q
// and K (fp32, exact softmax statistics) instead of being re-read from the stored bf16 probabilities; one launch
// replaces GEMM + softmax-Jacobian + GEMM and their [nt][H][L][80] round trips.  Same lane <-> query layout as above.
struct CrossArgs {
  const bf16 *Q, *K, *V;              // primal q [B][L][C], k / v [B][Lk][Ck]
  const bf16* BT;                     // per-head transpose of the second-stage operand B: [B][H][d][Lkp]
  const bf16* X; bf16* Y;             // [nt][L][Cx] in, [nt][L][Cy] out
  int L, Lk, Lkp, C, Ck, Cx, Cy, H, kps, a_is_v, accumulate, xcd;
  int primal;                         // 1: Y = P V (the forward pass itself: no X, no delta); BT may be null -- B^T is then built in LDS from the row tile
  float scale, c_in, c_out;
};
constexpr int XKEYS = 96, XLDT = XKEYS + 4;

template <int D, int FL>
__global__ __launch_bounds__(256) void attn_cross_kernel(CrossArgs a) {
  using F = FA<D>;
  __shared__ __attribute__((aligned(16))) bf16 sm[2 * XKEYS * F::LDR + F::DO * XLDT];
  bf16* sK = sm; bf16* sV = sK + XKEYS * F::LDR; bf16* sBT = sV + XKEYS * F::LDR;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const BlockXY blk = xcd_group_blocks(a.xcd);
  const int j = blk.y / a.H, h = blk.y % a.H, b = j / a.kps;
  const int q = blk.x * (nthr >> 1) + wave * 32 + l31;
  {   // K, V rows (zero beyond Lk / D) and B^T (zero beyond D / Lkp): one small tile each, loaded once per block
    constexpr int CPR = F::DP / 8;
    const bf16* Kp = a.K + (long)b * a.Lk * a.Ck + h * D;
    const bf16* Vp = a.V + (long)b * a.Lk * a.Ck + h * D;
    for (int c = tid; c < XKEYS * CPR; c += nthr) {
      const int r = c / CPR, cc = (c % CPR) * 8;
      uint4 kv = make_uint4(0, 0, 0, 0), vv = kv;
      if (r < a.Lk && cc < D) {
        kv = *reinterpret_cast<const uint4*>(Kp + (long)r * a.Ck + cc);
        vv = *reinterpret_cast<const uint4*>(Vp + (long)r * a.Ck + cc);
      }
      *reinterpret_cast<uint4*>(sK + r * F::LDR + cc) = kv;
      *reinterpret_cast<uint4*>(sV + r * F::LDR + cc) = vv;
    }
    constexpr int CPT = XKEYS / 8;
    if (a.BT) {
      const bf16* Bp = a.BT + ((long)b * a.H + h) * D * a.Lkp;
      for (int c = tid; c < F::DO * CPT; c += nthr) {
        const int r = c / CPT, cc = (c % CPT) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < D && cc < a.Lkp) v = *reinterpret_cast<const uint4*>(Bp + (long)r * a.Lkp + cc);
        bf16* dst = sBT + r * XLDT + cc;
        *reinterpret_cast<uint2*>(dst) = make_uint2(v.x, v.y);
        *reinterpret_cast<uint2*>(dst + 4) = make_uint2(v.z, v.w);
      }
    } else {
      // no transposed copy in HBM (forward pass): B^T [DO][96] straight from the 77 global rows of B, 8 head-dim values per 16-byte load
      const bf16* Bp = (a.a_is_v ? a.K : a.V) + (long)b * a.Lk * a.Ck + h * D;
      for (int c = tid; c < XKEYS * (F::DO / 8); c += nthr) {
        const int r = c % XKEYS, cc = (c / XKEYS) * 8;            // consecutive lanes: consecutive keys -> conflict-free 2-byte LDS stores
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < a.Lk && cc < D) v = *reinterpret_cast<const uint4*>(Bp + (long)r * a.Ck + cc);
        const unsigned short* e = reinterpret_cast<const unsigned short*>(&v);
#pragma unroll
        for (int i = 0; i < 8; ++i) sBT[(cc + i) * XLDT + r].v = e[i];
      }
    }
  }
  bf16x8 qf[F::NS], xf[F::NS];
  load_outer_frags<D>(a.Q + (long)b * a.L * a.C + (long)q * a.C + h * D, qf, lhi);
  if (!a.primal) load_outer_frags<D>(a.X + (long)j * a.L * a.Cx + (long)q * a.Cx + h * D, xf, lhi);
  __syncthreads();
  const bf16* sA = a.a_is_v ? sV : sK;
  f32x16 s[3], t[3];
#pragma unroll
  for (int kb = 0; kb < 3; ++kb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[kb][r] = t[kb][r] = 0.f;
#pragma unroll
    for (int stp = 0; stp < F::NS; ++stp) {
      s[kb] = MFMA(lds_a_frag(sK, kb * 32 + l31, F::LDR, stp * 16 + lhi * 8), qf[stp], s[kb]);
      if (!a.primal) t[kb] = MFMA(lds_a_frag(sA, kb * 32 + l31, F::LDR, stp * 16 + lhi * 8), xf[stp], t[kb]);
    }
  }
  const float c2 = a.scale * 1.44269504088896f;
  float mx = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < 3; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      s[kb][r] = key < a.Lk ? s[kb][r] * c2 : -INFINITY;
      mx = fmaxf(mx, s[kb][r]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float l = 0.f;
#pragma unroll
  for (int kb = 0; kb < 3; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[kb][r] = __builtin_amdgcn_exp2f(s[kb][r] - mx); l += s[kb][r]; }
  l += __shfl_xor(l, 32, 64);
  const float il = 1.f / l;
  float delta = 0.f;
#pragma unroll
  for (int kb = 0; kb < 3; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[kb][r] *= il;                                   // P
      t[kb][r] *= a.c_in * s[kb][r];                    // P o (c_in X A^T)
      delta += t[kb][r];
    }
  delta += __shfl_xor(delta, 32, 64);
  f32x16 acc[F::ND];
#pragma unroll
  for (int d = 0; d < F::ND; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
#pragma unroll
  for (int kb = 0; kb < 3; ++kb) {
    float w[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) w[r] = a.primal ? s[kb][r] : t[kb][r] - delta * s[kb][r];
    bf16x8 wb[2];
    pack_b<FL>(w, wb);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int d = 0; d < F::ND; ++d)
        acc[d] = MFMA(lds_t_frag(sBT, d * 32 + l31, XLDT, kb * 32 + ks * 16, lhi), wb[ks], acc[d]);
  }
  bf16* Yp = a.Y + (long)j * a.L * a.Cy + (long)q * a.Cy + h * D;
#pragma unroll
  for (int d = 0; d < F::ND; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = d * 32 + 8 * g + 4 * lhi;
      if (col < D) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = a.c_out * acc[d][g * 4 + i];
        if (a.accumulate) {
          uint2 ov = *reinterpret_cast<const uint2*>(Yp + col);
          v[0] += H16<FL>::lo(ov.x); v[1] += H16<FL>::hi(ov.x);
          v[2] += H16<FL>::lo(ov.y); v[3] += H16<FL>::hi(ov.y);
        }
        *reinterpret_cast<uint2*>(Yp + col) =
            make_uint2(H16<FL>::pack2(v[0], v[1]), H16<FL>::pack2(v[2], v[3]));
      }
    }
}

// head dims with a kernel instantiation: 40 / 80 / 160 (SD-1.x), 64 (SD-2.x: every level has 64-wide heads)
static bool head_dim_ok(int d) { return d == 40 || d == 64 || d == 80 || d == 160; }
// expands `stmt` with D = the head dim and FL = the 16-bit flavour as compile-time constants
#define DPB_ATT_DISPATCH(d, fl, ...)                                                        \
  do {                                                                                       \
    if ((fl) == 0) { constexpr int FL = 0; DPB_ATT_D(d, __VA_ARGS__) } else { constexpr int FL = 1; DPB_ATT_D(d, __VA_ARGS__) } \
  } while (0)
#define DPB_ATT_D(d, ...)                                          \
  if ((d) == 40) { constexpr int D = 40; __VA_ARGS__; }            \
  else if ((d) == 64) { constexpr int D = 64; __VA_ARGS__; }       \
  else if ((d) == 80) { constexpr int D = 80; __VA_ARGS__; }       \
  else { constexpr int D = 160; __VA_ARGS__; }

int cross_attention_supported(int dtype, int d, int Lq, int Lk, int kv_const) {
  return dtype != DT_F32 && kv_const && head_dim_ok(d) && Lk <= XKEYS && Lq % 32 == 0 && (Lq >= 128 ? Lq % 128 == 0 : true);
}

int launch_attn_cross(const CrossAttnArgs& f, int nt, hipStream_t st) {
  CrossArgs a;
  a.Q = (const bf16*)f.Q; a.K = (const bf16*)f.K; a.V = (const bf16*)f.V; a.BT = (const bf16*)f.BT;
  a.X = (const bf16*)f.X; a.Y = (bf16*)f.Y;
  a.L = f.L; a.Lk = f.Lk; a.Lkp = f.Lkp; a.C = f.C; a.Ck = f.Ck; a.Cx = f.Cx; a.Cy = f.Cy; a.H = f.H; a.kps = f.kps;
  a.a_is_v = f.adjoint; a.accumulate = f.accumulate; a.scale = f.scale; a.xcd = 0;   // every block loads its own small K/V tile: nothing to share
  a.c_in = f.adjoint ? 1.f : f.scale; a.c_out = f.adjoint ? f.scale : 1.f;
  a.primal = f.primal;
  if (f.primal) { a.a_is_v = 0; a.c_in = 1.f; a.c_out = 1.f; a.accumulate = 0; }
  const int waves = f.L >= 128 ? 4 : f.L / 32;
  dim3 grid(f.L / (waves * 32), nt * f.H);
  if (!head_dim_ok(f.d)) { set_error("cross attention: head dim %d unsupported", f.d); return -1; }
  DPB_ATT_DISPATCH(f.d, f.fl, hipLaunchKernelGGL((attn_cross_kernel<D, FL>), grid, dim3(waves * 64), 0, st, a));
  DPB_CHECK(hipGetLastError());
  return 0;
}

// row statistics of the primal probabilities from the materialised scaled scores S (before softmax):
// one wave per row; stats[row] = (max, 1 / sum exp(S - max))
template <int FL>
__global__ __launch_bounds__(256) void row_stats_kernel(const bf16* S, float* stats, long nrows, int Lk, int ld) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const bf16* sp = S + row * ld;
  float m = -INFINITY;
  for (int c = lane * 8; c < Lk; c += 512) {
    float v[8];
    H16<FL>::load8(sp + c, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) if (c + e < Lk) m = fmaxf(m, v[e]);
  }
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane * 8; c < Lk; c += 512) {
    float v[8];
    H16<FL>::load8(sp + c, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) if (c + e < Lk) s += __expf(v[e] - m);
  }
  s = wave_sum(s);
  if (lane == 0) { stats[2 * row] = m; stats[2 * row + 1] = 1.f / s; }
}

int fused_attention_supported(int dtype, int d, int L, int kv_const) {
  // L >= 256 in whole blocks; or the one-stage case L = 64 at head dim 160 (the 8x8 level of SD-1.x: one 64-key stage, half of the block's waves idle) --
  // three launches per iteration instead of the materialised path's ~19 tiny ones
  return dtype != DT_F32 && !kv_const && head_dim_ok(d) && ((L >= 256 && L % (att_waves(d) * 32) == 0) || (d == 160 && L == 64));
}

int launch_row_stats(int fl, const void* S, float* stats, long nrows, int Lk, int ld, hipStream_t st) {
  if (fl) hipLaunchKernelGGL((row_stats_kernel<1>), dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, st, (const bf16*)S, stats, nrows, Lk, ld);
  else hipLaunchKernelGGL((row_stats_kernel<0>), dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, st, (const bf16*)S, stats, nrows, Lk, ld);
  DPB_CHECK(hipGetLastError());
  return 0;
}

static int attn_xcd_on() { static int on = getenv("DPB_ATTN_XCD") ? atoi(getenv("DPB_ATTN_XCD")) : 0; return on; }

static FusedArgs to_args(const FusedAttnArgs& f) {
  FusedArgs a;
  a.xcd = attn_xcd_on();
  a.Q = (const bf16*)f.Q; a.K = (const bf16*)f.K; a.V = (const bf16*)f.V; a.O = (const bf16*)f.O;
  a.KT = (const bf16*)f.KT; a.VT = (const bf16*)f.VT; a.QT = (const bf16*)f.QT; a.stats = f.stats; a.Drow = f.Drow; a.Dout = nullptr;
  a.dQ = (const bf16*)f.dQ; a.dK = (const bf16*)f.dK; a.dV = (const bf16*)f.dV; a.dVT = (const bf16*)f.dVT; a.dO = (bf16*)f.dO;
  a.gO = (const bf16*)f.gO; a.gOT = (const bf16*)f.gOT; a.gQ = (bf16*)f.gQ; a.gK = (bf16*)f.gK; a.gV = (bf16*)f.gV;
  a.accQ = f.accQ; a.accK = f.accK; a.accV = f.accV;
  a.L = f.L; a.C = f.C; a.Co = f.Co; a.H = f.H; a.kps = f.kps; a.scale = f.scale;
  return a;
}

int launch_attn_fwd_fused(const FusedAttnArgs& f, int batch, void* O, float* stats, hipStream_t st) {
  FusedArgs a = to_args(f);
  dim3 grid((f.L + att_waves(f.d) * 32 - 1) / (att_waves(f.d) * 32), batch * f.H);
  if (!head_dim_ok(f.d)) { set_error("fused attention: head dim %d unsupported", f.d); return -1; }
  // one sample at the 64x64 / 32x32 levels: 8-wave blocks of 256 queries leave half (or more) of the 256 CUs without a block (16 x 8 heads = 128
  // blocks at L = 4096) -- 4-wave blocks of 128 queries then (the DDIM inversion / forward-to-edit_t steps and the primal pass of a pullback run at batch 1)
  static const int fwd4 = getenv("DPB_ATTN_FWD4") ? atoi(getenv("DPB_ATTN_FWD4")) : 1;   // tuning switch
  if (fwd4 && att_waves(f.d) == 8 && f.L % 128 == 0 && f.L >= 256 && (long)grid.x * grid.y < 256) {
    dim3 g4(f.L / 128, batch * f.H);
    if (f.d == 40) { if (f.fl) hipLaunchKernelGGL((attn_fwd_kernel<40, 1, 4>), g4, dim3(256), 0, st, a, (bf16*)O, stats); else hipLaunchKernelGGL((attn_fwd_kernel<40, 0, 4>), g4, dim3(256), 0, st, a, (bf16*)O, stats); }
    else if (f.d == 64) { if (f.fl) hipLaunchKernelGGL((attn_fwd_kernel<64, 1, 4>), g4, dim3(256), 0, st, a, (bf16*)O, stats); else hipLaunchKernelGGL((attn_fwd_kernel<64, 0, 4>), g4, dim3(256), 0, st, a, (bf16*)O, stats); }
    else { if (f.fl) hipLaunchKernelGGL((attn_fwd_kernel<80, 1, 4>), g4, dim3(256), 0, st, a, (bf16*)O, stats); else hipLaunchKernelGGL((attn_fwd_kernel<80, 0, 4>), g4, dim3(256), 0, st, a, (bf16*)O, stats); }
    DPB_CHECK(hipGetLastError());
    return 0;
  }
  DPB_ATT_DISPATCH(f.d, f.fl, hipLaunchKernelGGL((attn_fwd_kernel<D, FL>), grid, dim3(att_waves(D) * 64), 0, st, a, (bf16*)O, stats));
  DPB_CHECK(hipGetLastError());
  return 0;
}

// Block size of the d = 40 tangent / key-major adjoint kernels.  With 8 waves (256 outer rows) a block streams every inner tile once for 256
// rows, but the grid of the 64x64 level at k = 5 -- 16 x 40 blocks -- is 2.5 rounds of the 256 CUs: the third round runs half empty.  With 4 waves,
// two blocks per CU, the 1280 blocks are two full rounds plus exactly one block per CU: measured +0.6 % end to end in one session (109.7 vs 109.1
// iterations/s).  Grids that are whole rounds of 8-wave blocks (k = 10, several samples) keep 8 waves: forcing 4 there costs 3.7 % (83.8 -> 80.7).
static int att_block_waves(int d, int L, int pairs) {
  static const int force = getenv("DPB_ATTN_WAVES") ? atoi(getenv("DPB_ATTN_WAVES")) : 0;     // tuning switch (4 | 8)
  if (d != 40 || L % 256) return att_waves(d);
  if (force == 4 || force == 8) return force;
  return ((long)(L / 256) * pairs) % 256 == 0 ? 8 : 4;
}

int launch_attn_jvp_fused(const FusedAttnArgs& f, int nt, hipStream_t st) {
  FusedArgs a = to_args(f);
  if (!head_dim_ok(f.d)) { set_error("fused attention: head dim %d unsupported", f.d); return -1; }
  if (att_block_waves(f.d, f.L, nt * f.H) == 4 && f.d == 40) {
    dim3 g4(f.L / 128, nt * f.H);
    if (f.fl) hipLaunchKernelGGL((attn_jvp_kernel<40, 1, 4>), g4, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((attn_jvp_kernel<40, 0, 4>), g4, dim3(256), 0, st, a);
    DPB_CHECK(hipGetLastError());
    return 0;
  }
  dim3 grid((f.L + att_waves(f.d) * 32 - 1) / (att_waves(f.d) * 32), nt * f.H);
  DPB_ATT_DISPATCH(f.d, f.fl, hipLaunchKernelGGL((attn_jvp_kernel<D, FL>), grid, dim3(att_waves(D) * 64), 0, st, a));
  DPB_CHECK(hipGetLastError());
  return 0;
}

// (Running the query-major kernel (gQ) on a side stream next to the key-major kernel (gK, gV) -- they write disjoint column windows -- was
// measured in round 2: both kernels hold a CU's LDS alone, so they time-slice instead of overlapping: 587 us for the pair vs 585 us back to back.)
// which kernels the adjoint of one fused layer takes: bit 0 multi-cotangent query-major kernel, bit 1 shared-probability key-major kernel
static int attn_adj_route(int d, int L, int kps, int nt, bool have_drow) {
  static const int multi = getenv("DPB_ATTN_MULTI") ? atoi(getenv("DPB_ATTN_MULTI")) : 1;   // shared-P multi-cotangent kernel (tuning switch)
  // The shared-probability adjoint kernels serve head dim 40 (the SD-1.x 64x64 level).  They are instantiated for d = 64 too (SD-2.x: every level), but
  // there they measured no gain -- SD-2.1-base mid, k = 5, fp16: 8.41 ms per iteration on the per-cotangent kernels, 8.43 with the query-major one,
  // 8.54 with both (5 heads at L = 4096 give the key-major kernel 320 blocks = 1.25 rounds of one block per CU) -- so d = 64 takes them only when
  // bit 2 of the attn_shared switch asks (the parity test does).
  const int shared = g_attn_shared;
  const bool dsh = d == 40 || (d == 64 && (shared & 4));
  int r = 0;
  if (dsh && (multi & 1) && L % 128 == 0 && nt % kps == 0) r |= 1;
  if (dsh && (shared & 2) && L % 64 == 0 && kps >= 4 && nt % kps == 0 && have_drow) r |= 2;
  return r;
}
// launches of launch_attn_adj_fused: query-major + key-major (+ the row-dot pre-pass when the key-major shared kernel cannot take D_t from the multi-cotangent one)
int attn_adj_route_bits(int d, int L, int kps, int nt) { return attn_adj_route(d, L, kps, nt, true); }
int attn_adj_launches(int d, int L, int kps, int nt) { const int r = attn_adj_route(d, L, kps, nt, true); return 2 + (r == 2); }

int launch_attn_adj_fused(const FusedAttnArgs& f, int nt, hipStream_t st) {
  FusedArgs a = to_args(f);
  dim3 grid((f.L + att_waves(f.d) * 32 - 1) / (att_waves(f.d) * 32), nt * f.H);
  if (!head_dim_ok(f.d)) { set_error("fused attention: head dim %d unsupported", f.d); return -1; }
  const int route = attn_adj_route(f.d, f.L, f.kps, nt, f.Drow != nullptr);
  if (route & 1) {
    constexpr int TJ = 5;
    const int ngrp = (f.kps + TJ - 1) / TJ;
    const dim3 gq(f.L / 128, (nt / f.kps) * f.H * ngrp);
    if (route & 2) a.Dout = a.Drow;               // the row dots D_t = gO_t . O it computes anyway, left for the key-major kernel
#define DPB_ADJQ(DV, FLV) hipLaunchKernelGGL((attn_adj_q_multi_kernel<DV, TJ, FLV>), gq, dim3(256), 0, st, a)
    if (f.d == 40) { if (f.fl) DPB_ADJQ(40, 1); else DPB_ADJQ(40, 0); }
    else { if (f.fl) DPB_ADJQ(64, 1); else DPB_ADJQ(64, 0); }
#undef DPB_ADJQ
  } else {
    DPB_ATT_DISPATCH(f.d, f.fl, hipLaunchKernelGGL((attn_adj_q_kernel<D, FL>), grid, dim3(att_waves(D) * 64), 0, st, a));
  }
  if (route & 2) {
    constexpr int TJ = 5;
    const int ngrp = (f.kps + TJ - 1) / TJ;
    const dim3 gs(f.L / 64, (nt / f.kps) * f.H * ngrp);
    const unsigned nrd = (unsigned)(((long)nt * f.L * f.H + 255) / 256);
#define DPB_ADJKV(DV, FLV)                                                                                                          \
    do {                                                                                                                              \
      if (!(route & 1)) hipLaunchKernelGGL((attn_rowdot_kernel<DV, FLV>), dim3(nrd), dim3(256), 0, st, a, nt);                       \
      hipLaunchKernelGGL((attn_adj_kv_shared_kernel<DV, TJ, FLV>), gs, dim3(SHK<DV, TJ>::NT), 0, st, a);                             \
    } while (0)
    if (f.d == 40) { if (f.fl) DPB_ADJKV(40, 1); else DPB_ADJKV(40, 0); }
    else { if (f.fl) DPB_ADJKV(64, 1); else DPB_ADJKV(64, 0); }
#undef DPB_ADJKV
  } else if (att_block_waves(f.d, f.L, nt * f.H) == 4 && f.d == 40) {
    dim3 g4(f.L / 128, nt * f.H);
    if (f.fl) hipLaunchKernelGGL((attn_adj_kv_kernel<40, 1, 4>), g4, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((attn_adj_kv_kernel<40, 0, 4>), g4, dim3(256), 0, st, a);
  } else {
    DPB_ATT_DISPATCH(f.d, f.fl, hipLaunchKernelGGL((attn_adj_kv_kernel<D, FL>), grid, dim3(att_waves(D) * 64), 0, st, a));
  }
  DPB_CHECK(hipGetLastError());
  return 0;
}

}  // namespace dpb

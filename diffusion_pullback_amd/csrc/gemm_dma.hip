// bf16 MFMA GEMM / implicit-GEMM convolution with an asynchronous global->LDS ring (gfx950 LDS-DMA).
//
// Same contract as gemm_kernel (gemm.hip) for the single-operand-pair case, 128x128 tile, 4 waves (64x64 each).
// What differs is how operands reach LDS: every wave issues `global_load_lds_dwordx4` (1 KiB per wave instruction,
// no VGPR round trip, no ds_write pass) into a ring of S stages and keeps P = S-1 K-steps in flight; a stage is
// consumed after a COUNTED `s_waitcnt vmcnt(4*(stages still in flight))` + one raw `s_barrier` per K-step.  The
// register-staged kernel has exactly one K-step in flight per block and is latency-bound (measured 135-270 TF/s on
// the path's shapes); this one hides the L2/HBM latency inside the block.
//
// LDS image of a stage: A tile [128 rows][4 x 16 B] then B tile [128][4 x 16 B], rows 64 B apart, NO padding (the DMA
// writes lane-linear: wave-uniform base + lane*16).  Bank conflicts of the ds_read_b128 fragment reads are removed
// by an XOR swizzle applied on the SOURCE side: LDS chunk (row, c) holds logical K-chunk c ^ ((row >> 2) & 3), and the
// fragment reads apply the same involution.  Out-of-range rows / taps / K tails read a 16-byte zero page instead of
// branching, so every wave issues exactly 4 DMA instructions per stage and the vmcnt arithmetic is static.
// LDS fragment reads are inline asm (`ds_read_b128` + `s_waitcnt lgkmcnt(0)` + sched_barrier): hipcc would otherwise
// put `s_waitcnt vmcnt(0)` in front of every compiler-visible LDS read while a DMA is pending and drain the ring.
#include "epilogue.h"

namespace dpb {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__device__ inline bf16x8 lds_read_b128(unsigned addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

template <int BM, int BN, int S, int GATHER, int FL = 0, int EPI = EPI_PLAIN>   // FL: 16-bit flavour (H16<FL>): 0 bf16, 1 f16; EPI: epilogue.h
__global__ __launch_bounds__(256) void gemm_dma_kernel(GemmArgs p) {
  constexpr int BK = 32, CH = 8;
  constexpr int NIA = BM / 64, NIB = BN / 64;                          // DMA wave-instructions per stage per wave (A, B)
  constexpr int A_BYTES = BM * BK * 2, STAGE = (BM + BN) * BK * 2;     // 8 KiB + 8 KiB
  constexpr int P = S - 1;                                             // K-steps in flight
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32, SLD = WN + 4;
  constexpr int SMEM_BYTES = S * STAGE > 4 * 32 * SLD * 4 ? S * STAGE : 4 * 32 * SLD * 4;   // ring, reused by the epilogue staging
  __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tilesN = (p.N + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / tilesN) * BM, n0 = (bid % tilesN) * BN;
  const int z1 = blockIdx.y / p.Z2, z2 = blockIdx.y % p.Z2;
  const bf16* A = (const bf16*)p.A + (long)(z1 / p.divA) * p.sA1 + (long)z2 * p.sA2;
  const bf16* B = (const bf16*)p.B + (long)(z1 / p.divB) * p.sB1 + (long)z2 * p.sB2;
  bf16* C = (bf16*)p.C + (long)z1 * p.sC1 + (long)z2 * p.sC2;
  const bf16* R = p.R ? (const bf16*)p.R + (long)z1 * p.sR1 + (long)z2 * p.sR2 : nullptr;
  const bf16* zero = (const bf16*)p.zeros;

  // K range of this block (split-K over blockIdx.z, in steps of BK)
  const int nk_all = (p.K + BK - 1) / BK;
  int kt_begin = 0, nk = nk_all;
  if (p.splitk > 1) {
    const int per = (nk_all + p.splitk - 1) / p.splitk;
    kt_begin = blockIdx.z * per;
    nk = max(0, min(nk_all, kt_begin + per) - kt_begin);
  }
  // ---- DMA slots of this lane: wave instruction i covers LDS chunks (wave*NI+i)*64 + lane of the A (resp. B) tile.
  // A slot = one 16-byte chunk (row, physical chunk column); its logical K chunk is swizzled on the source side.
  // The gather (GATHER is a compile-time mode) is resolved once per filter tap, not once per K-step: a_cur[i] is the
  // source pixel's channel vector for the current tap (nullptr = padding), so the steady state is one add per slot.
  const int K = p.K, Cin = p.Cin, Wd = p.W, Hd = p.H, lda = p.lda, strd = p.stride, pad = p.pad, KS = p.KS;
  const bf16* a_base[NIA];
  const bf16* a_cur[NIA];
  int a_oy[NIA], a_ox[NIA], kca[NIA], tap[NIA], cc[NIA];
  auto retap = [&](int i) {
    if constexpr (GATHER == GATHER_NONE) {
      a_cur[i] = a_base[i];
    } else {
      int ky = 0, kx = 0;
      if (KS == 3) { ky = (tap[i] * 11) >> 5; kx = tap[i] - ky * 3; }
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
      } else {
        int uy = a_oy[i] + ky - 1, ux = a_ox[i] + kx - 1;
        ok = ok && uy >= 0 && ux >= 0 && uy < 2 * Hd && ux < 2 * Wd;
        iy = uy >> 1; ix = ux >> 1;
      }
      a_cur[i] = ok ? a_base[i] + ((long)iy * Wd + ix) * lda : nullptr;
    }
  };
#pragma unroll
  for (int i = 0; i < NIA; ++i) {
    const int pos = (wave * NIA + i) * 64 + lane, row = pos >> 2, phys = pos & 3;
    const int kq = phys ^ ((row >> 2) & 3);
    kca[i] = kt_begin * BK + kq * CH;
    const int m = m0 + row;
    a_oy[i] = a_ox[i] = 0;
    tap[i] = 0;
    cc[i] = kca[i];
    if constexpr (GATHER == GATHER_NONE) {
      a_base[i] = m < p.M ? A + (long)m * lda : nullptr;
    } else {
      const int hw = p.Ho * p.Wo, smp = m / hw, rem = m - smp * hw;
      a_oy[i] = rem / p.Wo;
      a_ox[i] = rem - a_oy[i] * p.Wo;
      a_base[i] = m < p.M ? A + (long)smp * Hd * Wd * lda : nullptr;
      tap[i] = kca[i] / Cin;
      cc[i] = kca[i] - tap[i] * Cin;
    }
    retap(i);
  }
  const bf16* b_base[NIB];
  int kcb[NIB];
#pragma unroll
  for (int i = 0; i < NIB; ++i) {
    const int pos = (wave * NIB + i) * 64 + lane, row = pos >> 2, phys = pos & 3;
    kcb[i] = kt_begin * BK + (phys ^ ((row >> 2) & 3)) * CH;
    const int n = n0 + row;
    b_base[i] = n < p.N ? B + (long)n * p.ldb : nullptr;
  }
  auto issue = [&](int slot) {
    char* st = smem + slot * STAGE;
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      const bf16* src = (a_cur[i] && kca[i] < K) ? a_cur[i] + cc[i] : zero;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(st + (wave * NIA + i) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      const bf16* src = (b_base[i] && kcb[i] < K) ? b_base[i] + kcb[i] : zero;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(st + A_BYTES + (wave * NIB + i) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      kca[i] += BK;
      cc[i] += BK;
      if constexpr (GATHER != GATHER_NONE) {
        if (cc[i] >= Cin) {                        // next filter tap (every Cin/32 steps; uniform when Cin % 32 == 0)
          do { cc[i] -= Cin; ++tap[i]; } while (cc[i] >= Cin);
          retap(i);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NIB; ++i) kcb[i] += BK;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wy = wave >> 1, wx = wave & 1, l31 = lane & 31, lhi = lane >> 5;
  // fragment read addresses (bytes, inside a stage): row*64 + ((kk*2+lhi) ^ ((row>>2)&3))*16
  unsigned fa[TM][2], fb[TN][2];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = wy * WM + i * 32 + l31, sw = (row >> 2) & 3;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fa[i][kk] = row * 64 + (((kk * 2 + lhi) ^ sw) << 4);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wx * WN + j * 32 + l31, sw = (row >> 2) & 3;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fb[j][kk] = A_BYTES + row * 64 + (((kk * 2 + lhi) ^ sw) << 4);
  }

#pragma unroll
  for (int s = 0; s < P; ++s)
    if (s < nk) issue(s);
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt landed for this wave: everything issued after it may still be in flight (4 DMA instructions per stage)
    const int later = min(P - 1, nk - 1 - kt);     // stages issued after stage kt that may still be in flight
    constexpr int U = NIA + NIB;                   // DMA instructions per stage per wave
    static_assert(P - 1 <= 3, "add vmcnt cases for a deeper ring");
    if constexpr (U == 2) {
      if (later >= 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (later == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (later == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if constexpr (U == 4) {
      if (later >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (later == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      static_assert(U == 6, "unsupported tile shape");
      if (later >= 3) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
      else if (later == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (later == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();          // ... and for every other wave; also: stage kt-1 is fully consumed
    if (kt + P < nk) issue((kt + P) % S);  // refill the slot stage kt-1 occupied
    const unsigned sb = lds0 + (kt % S) * STAGE;
    bf16x8 a[TM][2], b[TN][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i][kk] = lds_read_b128(sb + fa[i][kk]);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j][kk] = lds_read_b128(sb + fb[j][kk]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = H16<FL>::mfma(a[i][kk], b[j][kk], acc[i][j]);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- epilogue through LDS (same scheme as gemm_kernel)
  float* stage = reinterpret_cast<float*>(smem) + wave * 32 * SLD;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * lhi) * SLD + j * 32 + l31] = acc[i][j][r];
    __syncthreads();
    epilogue_slab<FL, WN, SLD, EPI>(p, C, R, reinterpret_cast<const float*>(smem), wave, lane, m0 + wy * WM + i * 32, n0,
                               (long)blockIdx.z * gridDim.y + blockIdx.y);
    __syncthreads();
  }
}

template <int BM, int BN, int S, int FL>
static void launch_dma_f(const GemmArgs& a, dim3 grid, hipStream_t st) {
  switch (a.gather) {
    case GATHER_NONE:
      if (a.epi == EPI_GEGLU_TAN) hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, S, GATHER_NONE, FL, EPI_GEGLU_TAN>), grid, dim3(256), 0, st, a);
      else if (a.epi == EPI_GEGLU_ADJ) hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, S, GATHER_NONE, FL, EPI_GEGLU_ADJ>), grid, dim3(256), 0, st, a);
      else if (a.epi == EPI_GEGLU_FWD) hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, S, GATHER_NONE, FL, EPI_GEGLU_FWD>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, S, GATHER_NONE, FL>), grid, dim3(256), 0, st, a);
      break;
    case GATHER_CONV: hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, S, GATHER_CONV, FL>), grid, dim3(256), 0, st, a); break;
    case GATHER_CONVT: hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, S, GATHER_CONVT, FL>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, S, GATHER_UPCONV, FL>), grid, dim3(256), 0, st, a); break;
  }
}
template <int BM, int BN, int S>
static void launch_dma_t(const GemmArgs& a, dim3 grid, hipStream_t st) {
  if (a.fl) launch_dma_f<BM, BN, S, 1>(a, grid, st);
  else launch_dma_f<BM, BN, S, 0>(a, grid, st);
}

int launch_gemm_dma(const GemmArgs& a, int tile, hipStream_t st) {
  const int sk = a.splitk > 1 ? a.splitk : 1;
  const int Z = a.Z1 * a.Z2;
  auto tiles = [&](int bm, int bn) { return dim3(((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn), Z, sk); };
  if (tile == 128) launch_dma_t<128, 128, 4>(a, tiles(128, 128), st);
  else if (tile == 130) launch_dma_t<128, 128, 3>(a, tiles(128, 128), st);        // 48 KiB ring -> 3 blocks/CU (default)
  else if (tile == 256) launch_dma_t<256, 128, 3>(a, tiles(256, 128), st);        // wave tile 128x64, 72 KiB ring
  else if (tile == 132) launch_dma_t<128, 128, 2>(a, tiles(128, 128), st);
  else launch_dma_t<64, 64, 4>(a, tiles(64, 64), st);
  DPB_CHECK(hipGetLastError());
  return 0;
}

}  // namespace dpb

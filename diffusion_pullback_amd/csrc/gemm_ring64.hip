// bf16 MFMA GEMM / implicit-GEMM convolution, asynchronous global->LDS ring with 128-byte row segments (BK = 64) and an
// in-wave software pipeline.  Second generation of gemm_dma.hip, written after ablating that kernel on MI355X
// (tools/gpu_gemm_ablate.py, profiles/r01_gemm_ablation.txt):
//   * with BK = 32 a DMA row segment is 64 B = HALF a 128-byte cache line, so every line is requested twice (by
//     consecutive K-steps); the DMA stream alone then costs 340 us of the 450 us a 20480x2560x2560 product takes.  With
//     one full line per lane-octet the same bytes move ~1.7x faster  ->  BK = 64, rows 128 B apart in LDS.
//   * a wave used to read all fragments of a K-step, wait, then issue its MFMAs (LDS + MFMA alone: 196 us vs 107 us of
//     pure MFMA time).  Here the fragments of K16-substep kk+1 are in flight while the MFMAs of substep kk issue; the
//     first substep of the next stage is fetched behind the last MFMAs of the current one.
// One block = 4 waves, wave tile (BM/2) x (BN/2), ring of S stages of (BM + BN) x 128 B: 128x128 -> 32 KiB / stage,
// 256x128 -> 48 KiB / stage (S = 3: 144 of the CU's 160 KiB, one resident block whose 4 waves each own a SIMD).
//
// LDS image of a stage: A rows then B rows, 128 B per row = 8 chunks of 16 B, no padding (the DMA writes lane-linear,
// 8 rows per wave instruction).  Chunk (row, c) holds logical K-chunk c ^ ((row >> 1) & 7): with the ds_read_b128 lane
// groups of gfx950 ({0-3,12-15,20-27}, ...) the 16 lanes of a group then cover 16 distinct 16-byte bank groups.
// The swizzle is applied on the SOURCE side (a lane octet permutes the chunks of ONE global line, so requests stay
// whole lines).  Padding rows / taps / K tails read a 16-byte zero page, so the vmcnt arithmetic stays static.
#include "epilogue.h"

#ifndef DPB_ABLATE
#define DPB_ABLATE 0   // micro-benchmark builds only (tools/gpu_gemm_ablate.py): 1 no MFMA, 2 no DMA refills, 4 no LDS fragment reads,
                       // 8 no epilogue global stores (and no bias / residual loads), 16 no epilogue at all
#endif

namespace dpb {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__device__ inline bf16x8 lds_read_frag(unsigned addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

// s_waitcnt lgkmcnt(N) that the compiler must keep between the fragment reads and their consumers: the fragments are
// threaded through it as read-write operands
template <int N, int NA, int NB>
__device__ inline void wait_frags(bf16x8 (&a)[NA], bf16x8 (&b)[NB]) {
  static_assert((NA == 1 || NA == 2 || NA == 4 || NA == 5) && (NB == 1 || NB == 2 || (NA == 2 && (NB == 4 || NB == 5))), "fragment counts of the supported wave tiles");
  if constexpr (NA == 1 && NB == 2) {           // 64 x 128 block tile: 32 x 64 wave tiles
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a[0]), "+v"(b[0]), "+v"(b[1]) : "n"(N));
  } else if constexpr (NA == 1) {
    static_assert(NB == 1, "64x64 tile: one fragment each");
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a[0]), "+v"(b[0]) : "n"(N));
  } else if constexpr (NA == 2 && NB == 5) {
    asm volatile("s_waitcnt lgkmcnt(%7)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]) : "n"(N));
  } else if constexpr (NA == 2 && NB == 4) {
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N));
  } else if constexpr (NA == 2 && NB == 2) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]) : "n"(N));
  } else if constexpr (NA == 2 && NB == 1) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]) : "n"(N));
  } else if constexpr (NA == 4 && NB == 2) {
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]) : "n"(N));
  } else if constexpr (NA == 4 && NB == 1) {
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]) : "n"(N));
  } else if constexpr (NA == 5 && NB == 2) {
    asm volatile("s_waitcnt lgkmcnt(%7)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(b[0]), "+v"(b[1]) : "n"(N));
  } else {
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(b[0]) : "n"(N));
  }
}

template <int U>
__device__ inline void wait_dma(int stages_in_flight) {   // s_waitcnt vmcnt(stages_in_flight * U), U DMA instructions per stage
  static_assert(3 * U <= 63, "vmcnt is a 6-bit counter");
  if (stages_in_flight >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * U) : "memory");
  else if (stages_in_flight == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * U) : "memory");
  else if (stages_in_flight == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int BM, int BN, int S, int GATHER, int WAVES = 4, int FL = 0, int EPI = EPI_PLAIN>   // FL: 16-bit flavour (H16<FL>): 0 bf16, 1 f16; EPI: epilogue.h
__global__ __launch_bounds__(WAVES * 64) void gemm_ring64_kernel(GemmArgs p) {
  constexpr int BK = 64, CH = 8, KK = BK / 16;
  constexpr int NIA = BM / (8 * WAVES), NIB = BN / (8 * WAVES), U = NIA + NIB;   // DMA wave-instructions (8 rows each) per stage per wave
  static_assert(NIA >= 1 && NIB >= 1 && BM % (8 * WAVES) == 0 && BN % (8 * WAVES) == 0, "tile too small for the wave count");
  constexpr int A_BYTES = BM * BK * 2, STAGE = (BM + BN) * BK * 2;
  constexpr int WM = BM / (WAVES / 2), WN = BN / 2, TM = WM / 32, TN = WN / 32, SLD = WN + 4;   // wave grid (WAVES/2) x 2
  constexpr int NR = TM + TN;                                          // fragment reads per K16 substep
  constexpr int SMEM_BYTES = S * STAGE > WAVES * 32 * SLD * 4 ? S * STAGE : WAVES * 32 * SLD * 4;
  static_assert(S >= 2 && S <= 5, "ring depth");
  __shared__ __attribute__((aligned(128))) char smem[SMEM_BYTES];
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- block -> (tile, batch, K split).  Workgroups are dealt round-robin to the 8 XCDs in dispatch order; give each XCD
  // (= each L2) a CONTIGUOUS run of a processing order in which neighbours share an operand.  order 0 ("A-major", big
  // activations / small weights): the N tiles of one A row panel are adjacent.  order 1 ("B-major", weight-heavy
  // layers): all M tiles of one (N tile, K split) weight block are adjacent, so the block is fetched from HBM once
  // instead of once per M tile and per XCD (measured 270 MB of reads for 36 MB of operands before this).
  const int tilesN = (p.N + BN - 1) / BN, tilesM = (p.M + BM - 1) / BM;
  int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  {
    const int nwg = gridDim.x * gridDim.y * gridDim.z, q = nwg >> 3, r = nwg & 7, xcd = lin & 7, idx = lin >> 3;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tm, tn;
  if (p.order == 0) { tn = lin % tilesN; lin /= tilesN; tm = lin % tilesM; lin /= tilesM; }
  else { tm = lin % tilesM; lin /= tilesM; tn = lin % tilesN; lin /= tilesN; }
  const int ksplit = lin % (int)gridDim.z, zb = lin / (int)gridDim.z;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z1 = zb / p.Z2, z2 = zb % p.Z2;
  const bf16* A = (const bf16*)p.A + (long)(z1 / p.divA) * p.sA1 + (long)z2 * p.sA2;
  const bf16* B = (const bf16*)p.B + (long)(z1 / p.divB) * p.sB1 + (long)z2 * p.sB2;
  bf16* C = (bf16*)p.C + (long)z1 * p.sC1 + (long)z2 * p.sC2;
  const bf16* R = p.R ? (const bf16*)p.R + (long)z1 * p.sR1 + (long)z2 * p.sR2 : nullptr;
  const bf16* zero = (const bf16*)p.zeros;

  const int nk_all = (p.K + BK - 1) / BK;
  int kt_begin = 0, nk = nk_all;
  if (p.splitk > 1) {
    const int per = (nk_all + p.splitk - 1) / p.splitk;
    kt_begin = ksplit * per;
    nk = max(0, min(nk_all, kt_begin + per) - kt_begin);
  }

  // ---- DMA slots: wave instruction i of the A (B) tile covers LDS chunks (wave*NI + i)*64 + lane = rows 8(wave*NI+i)..+7.
  // The gather is resolved once per filter tap (a_cur = the source pixel's channel vector, nullptr = padding).
  const int K = p.K, Cin = p.Cin, Wd = p.W, Hd = p.H, lda = p.lda, strd = p.stride, pad = p.pad, KS = p.KS;
  const bf16* a_base[NIA];
  const bf16* a_cur[NIA];
  int a_oyx[NIA], kca[NIA], tap[NIA], cc[NIA];
  auto retap = [&](int i) {
    if constexpr (GATHER == GATHER_NONE) {
      a_cur[i] = a_base[i];
    } else {
      int ky = 0, kx = 0;
      if (KS == 3) { ky = (tap[i] * 11) >> 5; kx = tap[i] - ky * 3; }
      const int oy = a_oyx[i] >> 16, ox = a_oyx[i] & 0xffff;
      int iy, ix;
      bool ok = a_base[i] != nullptr;
      if constexpr (GATHER == GATHER_CONV) {
        iy = oy * strd + ky - pad;
        ix = ox * strd + kx - pad;
        ok = ok && iy >= 0 && iy < Hd && ix >= 0 && ix < Wd;
      } else if constexpr (GATHER == GATHER_CONVT) {
        int ty = oy + pad - ky, tx = ox + pad - kx;
        ok = ok && ty >= 0 && tx >= 0;
        if (strd == 2) { ok = ok && !((ty | tx) & 1); iy = ty >> 1; ix = tx >> 1; } else { iy = ty; ix = tx; }
        ok = ok && iy < Hd && ix < Wd;
      } else {
        int uy = oy + ky - 1, ux = ox + kx - 1;
        ok = ok && uy >= 0 && ux >= 0 && uy < 2 * Hd && ux < 2 * Wd;
        iy = uy >> 1; ix = ux >> 1;
      }
      a_cur[i] = ok ? a_base[i] + ((long)iy * Wd + ix) * lda : nullptr;
    }
  };
#pragma unroll
  for (int i = 0; i < NIA; ++i) {
    const int pos = (wave * NIA + i) * 64 + lane, row = pos >> 3, phys = pos & 7;
    kca[i] = kt_begin * BK + (phys ^ ((row >> 1) & 7)) * CH;
    const int m = m0 + row;
    a_oyx[i] = 0;
    tap[i] = 0;
    cc[i] = kca[i];
    if constexpr (GATHER == GATHER_NONE) {
      a_base[i] = m < p.M ? A + (long)m * lda : nullptr;
    } else {
      const int hw = p.Ho * p.Wo, smp = m / hw, rem = m - smp * hw, oy = rem / p.Wo;
      a_oyx[i] = (oy << 16) | (rem - oy * p.Wo);
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
    const int pos = (wave * NIB + i) * 64 + lane, row = pos >> 3, phys = pos & 7;
    kcb[i] = kt_begin * BK + (phys ^ ((row >> 1) & 7)) * CH;
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
        if (cc[i] >= Cin) {                        // next filter tap (uniform across the wave when Cin % 64 == 0)
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
  LnPre<WN> lnq[(EPI == EPI_LN_TAN || EPI == EPI_LN_ADJ) ? TM : 1];      // LayerNorm epilogues: their row operands, in flight under the K loop
  if constexpr (EPI == EPI_LN_TAN || EPI == EPI_LN_ADJ) {
#pragma unroll
    for (int i = 0; i < TM; ++i) ln_prefetch<FL, WN, EPI>(p, C, R, wave, lane, m0 + i * 32 + (wave >> 1) * (WM - 32), lnq[i]);
  }

  const int wy = wave >> 1, wx = wave & 1, l31 = lane & 31, lhi = lane >> 5;
  // fragment read addresses inside a stage: row*128 + (((kk*2 + lhi) ^ ((row>>1)&7)) << 4) = fa0 ^ (kk << 5)
  unsigned fa0[TM], fb0[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = wy * WM + i * 32 + l31;
    fa0[i] = row * 128 + ((lhi ^ ((row >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wx * WN + j * 32 + l31;
    fb0[j] = A_BYTES + row * 128 + ((lhi ^ ((row >> 1) & 7)) << 4);
  }
  bf16x8 fa[2][TM], fb[2][TN];                      // double-buffered fragments
  auto read_frags = [&](unsigned sb, int kk, int buf) {
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[buf][i] = (DPB_ABLATE & 4) ? bf16x8{} : lds_read_frag(sb + (fa0[i] ^ (kk << 5)));
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[buf][j] = (DPB_ABLATE & 4) ? bf16x8{} : lds_read_frag(sb + (fb0[j] ^ (kk << 5)));
  };

  if (nk > 0) {
#pragma unroll
    for (int s = 0; s < S; ++s)
      if (s < nk) issue(s);
    wait_dma<U>(min(S - 1, nk - 1));                // stage 0 landed for this wave ...
    __builtin_amdgcn_s_barrier();                   // ... and for every other wave
    read_frags(lds0, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned sb = lds0 + (kt % S) * STAGE;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk + 1 < KK) {
          read_frags(sb, kk + 1, nxt);
          if constexpr (!(DPB_ABLATE & 4)) wait_frags<NR>(fa[cur], fb[cur]);      // substep kk landed, kk+1 in flight
        } else {
          if constexpr (!(DPB_ABLATE & 4)) wait_frags<0>(fa[cur], fb[cur]);       // every read of stage kt by this wave is done
          if (kt + 1 < nk) {
            wait_dma<U>(min(S - 2, nk - 2 - kt));   // stage kt+1 landed for this wave
            __builtin_amdgcn_s_barrier();           // stage kt+1 visible to all; stage kt consumed by all
            if (kt + S < nk && !(DPB_ABLATE & 2)) issue(kt % S);
            read_frags(lds0 + ((kt + 1) % S) * STAGE, 0, nxt);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if constexpr (DPB_ABLATE & 1) {
              if constexpr (!(DPB_ABLATE & 4)) asm volatile("" ::"v"(fa[cur][i]), "v"(fb[cur][j]));
            } else {
              acc[i][j] = H16<FL>::mfma(fa[cur][i], fb[cur][j], acc[i][j]);
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- epilogue through LDS (32 accumulator rows per wave at a time -> 16-byte row-contiguous stores)
  if constexpr ((DPB_ABLATE & 16) != 0) {
    if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(p.C)[0] = 1.f;   // keep the accumulators alive
    return;
  }
  float* stage = reinterpret_cast<float*>(smem) + wave * 32 * SLD;
  static_for<0, TM>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * lhi) * SLD + j * 32 + l31] = acc[i][j][r];
    __syncthreads();
    if constexpr ((DPB_ABLATE & 8) != 0) {
      if (stage[lane] == 123.456f) reinterpret_cast<float*>(p.C)[0] = stage[lane + 1];
    } else if constexpr (EPI == EPI_LN_TAN || EPI == EPI_LN_ADJ) {
      static_assert(WAVES == 4 && TM == 2, "row-complete LayerNorm epilogue: 2 x 2 waves, 64 staged rows per step");
      // staged rows: wave pair wy holds tile rows wy*WM + i*32 .. +31; epilogue_ln's row index wy*32 + row maps to tile row wy*WM + i*32 + row
      epilogue_ln<FL, WN, SLD, EPI>(p, C, R, reinterpret_cast<const float*>(smem), wave, lane, m0 + i * 32 + (wave >> 1) * (WM - 32), lnq[i]);
    } else {
      epilogue_slab<FL, WN, SLD, EPI>(p, C, R, reinterpret_cast<const float*>(smem), wave, lane, m0 + wy * WM + i * 32, n0, (long)ksplit * gridDim.y + zb);
    }
    __syncthreads();
  });
}

template <int BM, int BN, int S, int WAVES, int FL>
static void launch_ring64_f(const GemmArgs& a, dim3 grid, hipStream_t st) {
  switch (a.gather) {
    case GATHER_NONE:
      if constexpr (BN == 128) {                  // the fused GEGLU epilogues pair the two 64-column waves of a 128-column tile
        if (a.epi == EPI_GEGLU_TAN) { hipLaunchKernelGGL((gemm_ring64_kernel<BM, BN, S, GATHER_NONE, WAVES, FL, EPI_GEGLU_TAN>), grid, dim3(WAVES * 64), 0, st, a); break; }
        if (a.epi == EPI_GEGLU_ADJ) { hipLaunchKernelGGL((gemm_ring64_kernel<BM, BN, S, GATHER_NONE, WAVES, FL, EPI_GEGLU_ADJ>), grid, dim3(WAVES * 64), 0, st, a); break; }
        if (a.epi == EPI_GEGLU_FWD) { hipLaunchKernelGGL((gemm_ring64_kernel<BM, BN, S, GATHER_NONE, WAVES, FL, EPI_GEGLU_FWD>), grid, dim3(WAVES * 64), 0, st, a); break; }
      }
      hipLaunchKernelGGL((gemm_ring64_kernel<BM, BN, S, GATHER_NONE, WAVES, FL>), grid, dim3(WAVES * 64), 0, st, a);
      break;
    case GATHER_CONV: hipLaunchKernelGGL((gemm_ring64_kernel<BM, BN, S, GATHER_CONV, WAVES, FL>), grid, dim3(WAVES * 64), 0, st, a); break;
    case GATHER_CONVT: hipLaunchKernelGGL((gemm_ring64_kernel<BM, BN, S, GATHER_CONVT, WAVES, FL>), grid, dim3(WAVES * 64), 0, st, a); break;
    default: hipLaunchKernelGGL((gemm_ring64_kernel<BM, BN, S, GATHER_UPCONV, WAVES, FL>), grid, dim3(WAVES * 64), 0, st, a); break;
  }
}
template <int BM, int BN, int S, int WAVES = 4>
static void launch_ring64_t(const GemmArgs& a, dim3 grid, hipStream_t st) {
  if (a.fl) launch_ring64_f<BM, BN, S, WAVES, 1>(a, grid, st);
  else launch_ring64_f<BM, BN, S, WAVES, 0>(a, grid, st);
}

// tile codes: 512 = 128x128 S3 (96 KiB, 1 block/CU), 513 = 256x128 S3 (144 KiB), 514 = 128x128 S4, 515 = 128x128 S2 (2 blocks/CU),
// 516 = 256x128 S3 with 8 waves (64x64 wave tiles, one shared B tile), 517 = the same with S2 (96 KiB),
// 518 = 256x256 S2 with 8 waves (64x128 wave tiles, 128 KiB ring: half the L2->LDS bytes per flop of 128x128; plain rows only).
// (64x64 tiles on this ring for the small 1280^3 / 320x1280x1280 products: measured equal to the BK=32 64x64 ring within 0.3 % end to end --
// those launches are latency-bound, not DMA-bound; removed.)
// (320x128 / 320x64 tiles -- the M = 64 k rows of the 8x8-level layers at k = 5 in ONE tile, weight panel streamed once -- were built
// and measured in round 2: one 4-wave block per CU is latency-bound, 47-52 us against 41 us for 128x128 S2 with split-K 8; removed.)
int launch_gemm_ring64(const GemmArgs& a, int tile, hipStream_t st) {
  const int sk = a.splitk > 1 ? a.splitk : 1;
  const int Z = a.Z1 * a.Z2;
  auto tiles = [&](int bm, int bn) { return dim3(((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn), Z, sk); };
  if (tile == 513) launch_ring64_t<256, 128, 3>(a, tiles(256, 128), st);
  else if (tile == 516) launch_ring64_t<256, 128, 3, 8>(a, tiles(256, 128), st);
  else if (tile == 517) launch_ring64_t<256, 128, 2, 8>(a, tiles(256, 128), st);
  else if (tile == 518) {
    if (a.gather != GATHER_NONE) { set_error("gemm: the 256x256 ring tile takes plain-row operands only"); return -1; }
    const dim3 g = tiles(256, 256);
#define DPB_RING518(FLV, EPIV) hipLaunchKernelGGL((gemm_ring64_kernel<256, 256, 2, GATHER_NONE, 8, FLV, EPIV>), g, dim3(512), 0, st, a)
    if (a.epi == EPI_GEGLU_TAN) { if (a.fl) DPB_RING518(1, EPI_GEGLU_TAN); else DPB_RING518(0, EPI_GEGLU_TAN); }
    else if (a.epi == EPI_GEGLU_ADJ) { if (a.fl) DPB_RING518(1, EPI_GEGLU_ADJ); else DPB_RING518(0, EPI_GEGLU_ADJ); }
    else if (a.epi == EPI_GEGLU_FWD) { if (a.fl) DPB_RING518(1, EPI_GEGLU_FWD); else DPB_RING518(0, EPI_GEGLU_FWD); }
    else { if (a.fl) DPB_RING518(1, EPI_PLAIN); else DPB_RING518(0, EPI_PLAIN); }
#undef DPB_RING518
  }
  else if (tile == 520) {                       // row-complete 128 x 320 tile with a fused LayerNorm epilogue (plain rows, N = 320, no split)
    if (a.gather != GATHER_NONE || a.N != 320 || sk != 1 || (a.epi != EPI_LN_TAN && a.epi != EPI_LN_ADJ)) { set_error("gemm: tile 520 is the N = 320 LayerNorm-epilogue tile"); return -1; }
    const dim3 g = tiles(128, 320);
#define DPB_RING520(FLV, EPIV) hipLaunchKernelGGL((gemm_ring64_kernel<128, 320, 2, GATHER_NONE, 4, FLV, EPIV>), g, dim3(256), 0, st, a)
    if (a.epi == EPI_LN_TAN) { if (a.fl) DPB_RING520(1, EPI_LN_TAN); else DPB_RING520(0, EPI_LN_TAN); }
    else { if (a.fl) DPB_RING520(1, EPI_LN_ADJ); else DPB_RING520(0, EPI_LN_ADJ); }
#undef DPB_RING520
  }
  else if (tile == 521 && a.epi == EPI_PLAIN && (a.gather == GATHER_CONV || a.gather == GATHER_CONVT)) {   // the same half tile as an implicit-GEMM convolution
    const dim3 g = tiles(64, 128);
    if (a.gather == GATHER_CONV) { if (a.fl) hipLaunchKernelGGL((gemm_ring64_kernel<64, 128, 3, GATHER_CONV, 4, 1, EPI_PLAIN>), g, dim3(256), 0, st, a); else hipLaunchKernelGGL((gemm_ring64_kernel<64, 128, 3, GATHER_CONV, 4, 0, EPI_PLAIN>), g, dim3(256), 0, st, a); }
    else { if (a.fl) hipLaunchKernelGGL((gemm_ring64_kernel<64, 128, 3, GATHER_CONVT, 4, 1, EPI_PLAIN>), g, dim3(256), 0, st, a); else hipLaunchKernelGGL((gemm_ring64_kernel<64, 128, 3, GATHER_CONVT, 4, 0, EPI_PLAIN>), g, dim3(256), 0, st, a); }
  }
  else if (tile >= 521 && tile <= 523) {        // half tiles for the <= 256-tile launches of the 32x32 level (plain rows, plain epilogue): twice the blocks, 2-3 per CU
    if (a.gather != GATHER_NONE || a.epi != EPI_PLAIN) { set_error("gemm: tile %d takes plain-row operands and the plain epilogue only", tile); return -1; }
#define DPB_RINGH(BMV, BNV, SV) do { const dim3 g = tiles(BMV, BNV); \
      if (a.fl) hipLaunchKernelGGL((gemm_ring64_kernel<BMV, BNV, SV, GATHER_NONE, 4, 1, EPI_PLAIN>), g, dim3(256), 0, st, a); \
      else hipLaunchKernelGGL((gemm_ring64_kernel<BMV, BNV, SV, GATHER_NONE, 4, 0, EPI_PLAIN>), g, dim3(256), 0, st, a); } while (0)
    if (tile == 521) DPB_RINGH(64, 128, 3);      // 72 KiB ring: 2 blocks per CU
    else if (tile == 522) DPB_RINGH(64, 128, 2); // 48 KiB ring: 3 blocks per CU
    else DPB_RINGH(128, 64, 3);
#undef DPB_RINGH
  }
  else if (tile == 514) launch_ring64_t<128, 128, 4>(a, tiles(128, 128), st);
  else if (tile == 515) launch_ring64_t<128, 128, 2>(a, tiles(128, 128), st);
  else launch_ring64_t<128, 128, 3>(a, tiles(128, 128), st);
  DPB_CHECK(hipGetLastError());
  return 0;
}

}  // namespace dpb

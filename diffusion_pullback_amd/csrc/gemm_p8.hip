// bf16 / f16 MFMA GEMM and implicit-GEMM convolution, 256 x 256 x 64 block tile, EIGHT waves in two groups that alternate between a
// "load" segment (LDS fragment reads + LDS-DMA issue) and a "matrix" segment (8 MFMAs) -- the deep-pipelined main loop that replaces the
// one-barrier-per-K-step rings (gemm_ring64.hip) on the large products of the path (the conv / linear stack diffusers' U-Net runs under
// /root/reference/src/utils/utils.py:466-499).
//
// Why another loop structure (round-5 finding): the ring kernels wait `vmcnt` + barrier once per K step with every wave in
// lockstep, so DMA latency, fragment latency and MFMA issue serialise inside each step; they stop at 0.17-0.30 of the MFMA peak whatever the tile.
// Here
//   * waves (wr, wc) = (wave >> 2, wave & 3) form a 2 x 4 grid, wave tile 128 x 64 (0.75 KB of LDS reads per 32x32x16 MFMA instead of 1 KB);
//   * a K tile is cut into FOUR phases, one 64 x 32 quadrant pair of the wave tile each: {read fragments, issue one 16 KiB half-tile DMA,
//     s_barrier, lgkmcnt(0), 8 MFMAs, s_barrier};
//   * the two wave groups (wr = 0 | 1: one wave of each on every SIMD) run ONE BARRIER APART, so on each SIMD one wave issues MFMAs while its partner
//     reads LDS / issues DMA;
//   * DMA is waited for once per K tile with a COUNTED vmcnt (three half tiles stay in flight across the barriers), never drained.
//
// LDS (128 KiB, one array): operand area (A | B, 64 KiB each) x half (h: 32 KiB) x K-tile parity (d: 16 KiB) = [128 rows][64 k] 16-bit, 128-byte rows,
// 16-byte chunk (row, c) holding logical K chunk c ^ ((row >> 1) & 7) -- the conflict-free image of gemm_ring64.hip, swizzled on the SOURCE side
// of the DMA.  Half h of A holds block rows (r >> 6) * 128 + h * 64 + (r & 63), half h of B block columns (r >> 5) * 64 + h * 32 + (r & 31)
// (r = row inside the half): every wave reads 64 rows of EACH A half and 32 rows of EACH B half, and its output is one contiguous 128 x 64 tile.
//
// Schedule of K tile s (parity d = s & 1); "read" = ds_read_b128 into registers, "stage X(t)" = 2 LDS-DMA instructions per lane for half tile X of
// K tile t; fragment reads are balanced 8 / 4 / 8 / 4 over the phases (B0 of the NEXT K tile is read in phase 4):
//   phase 1: read A0[d] (8)     | stage A1(s+1) -> A1[d^1]               | barrier | lgkmcnt(0) | MFMA a0 x b0 | barrier
//   phase 2: read B1[d] (4)     | stage B0(s+2) -> B0[d]                 | barrier | lgkmcnt(0) | MFMA a0 x b1 | barrier
//   phase 3: read A1[d] (8)     | stage A0(s+2) -> A0[d]   | vmcnt(10)   | barrier | lgkmcnt(0) | MFMA a1 x b0 | barrier
//   phase 4: read B0[d^1] (4)   | stage B1(s+2) -> B1[d]   | vmcnt(6)    | barrier |              MFMA a1 x b1 | lgkmcnt(0) | barrier
// Ordering rules this satisfies (group G0 runs one barrier ahead of G1, so G0's load segment of phase q+1 overlaps G1's matrix segment of phase q):
//   RAW  a half tile is read one phase AFTER the phase whose FIRST barrier follows every wave's covering vmcnt (nothing else orders a ds_read behind
//        another wave's LDS-DMA): phase 3's vmcnt(10) retires B0(s+1), issued five phases earlier -> read in phase 4; phase 4's vmcnt(6) retires
//        A0 / B1 / A1 of K tile s+1 (B0 / A0 / B1 of s+2 stay in flight) -> read in phases 1-3 of tile s+1.
//   WAR  a buffer is restaged two phases after its last read, whose lgkmcnt(0) precedes the reading phase's SECOND barrier (A0: read 1 -> staged 3,
//        B1: 2 -> 4, A1: 3 -> 1 of the next tile, B0[d]: read in phase 4 of tile s-1 -> staged in phase 2 of tile s).
// Rows / columns beyond M / N, K tails and the two K tiles "after the end" still issue their DMA (from a clamped row, the block's last K tile or a
// 16-byte zero page), so the DMA count per phase -- and with it every counted wait -- is static.
//
// Measured (profiles/r05_p8_variants.txt): 1.22 PF/s on 65536 x 2560 x 2560 (ring tiles: 1.00), bitwise equal to the ring kernel.  Variants tried and
// dropped there: the guide's 12 / 4 / 8 / 0 read schedule (equal), DMA issued inside the matrix segment with the waves taking turns (equal), one M0
// value per half tile (equal).  Loop ablation: the load path alone takes longer than the matrix path alone (L2 -> LDS at ~40 B/clk/CU).
#include "epilogue.h"

namespace dpb {

typedef __attribute__((address_space(3))) void p8_lds_t;
typedef __attribute__((address_space(1))) const void p8_gbl_t;

template <int OFF>
__device__ __forceinline__ bf16x8 p8_read(unsigned addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// s_waitcnt lgkmcnt(0) the compiler must keep between the fragment reads and the MFMAs that consume them (the fragments are threaded through it)
__device__ __forceinline__ void p8_wait4(bf16x8 (&b)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}
__device__ __forceinline__ void p8_wait8(bf16x8 (&a)[2][4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]));
}
__device__ __forceinline__ void p8_wait12(bf16x8 (&a)[2][4], bf16x8 (&b)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]), "+v"(b[0]), "+v"(b[1]),
                 "+v"(b[2]), "+v"(b[3]));
}

#ifndef DPB_P8_ABLATE
#define DPB_P8_ABLATE 0   // micro-benchmark builds only (make ablate_p8, tools/gpu_p8_ablate.py): 1 no DMA in the K loop, 2 no fragment reads, 4 no MFMA
#endif

// FAST (plain rows, K % 64 == 0): the DMA source is a wave-uniform 64-bit base (SGPR pair, advanced by SALU) + a loop-invariant lane offset -- one
// vector instruction per DMA instead of four (+4-5 %).  Rows / columns beyond M / N read the last valid row (their products are never stored), K tiles
// behind the block's range re-read its last tile (they are never multiplied).
template <int GATHER, int FL, int EPI, int FAST>
__global__ __launch_bounds__(512) void gemm_p8_kernel(GemmArgs p) {
  static_assert(!FAST || GATHER == GATHER_NONE, "uniform-base addressing is for plain rows");
  constexpr int ABL = DPB_P8_ABLATE;
  constexpr int BM = 256, BN = 256, BK = 64, CH = 8;
  constexpr int AREA = 65536, HALF = 32768, PAR = 16384;       // LDS bytes: operand area, half, K-tile parity
  constexpr int WN = 64, SLD = WN + 4;
  static_assert(8 * 32 * SLD * 4 <= 2 * AREA, "epilogue staging fits the ring");
  // The counted waits of the K loop.  vmcnt counts this wave's VMEM instructions, oldest retired first: every stage_a / stage_b call issues exactly
  // DMA_PER_STAGE LDS-DMA instructions per lane (16 KiB half tile = 8 waves x DMA_PER_STAGE x 1 KiB), nothing else is in the vector-memory queue inside
  // the loop, and the waits below are "all but the youngest n stages".  Change the pieces per stage (tile size, 8-byte pieces, ...) and the waits follow.
  constexpr int DMA_PER_STAGE = 2;
  constexpr int VM_PH3 = 5 * DMA_PER_STAGE;   // phase 3: B0(s+1) landed; A0 B1 A1 of s+1 and B0 A0 of s+2 (five stages) stay in flight
  constexpr int VM_PH4 = 3 * DMA_PER_STAGE;   // phase 4 / prologue: K tile s+1 landed; B0 A0 B1 of s+2 (three stages) stay in flight
  static_assert(PAR == 8 * DMA_PER_STAGE * 1024, "a half tile of one parity = 8 waves x DMA_PER_STAGE pieces of 1 KiB");
  static_assert(VM_PH3 == 10 && VM_PH4 == 6 && VM_PH3 < 64, "the schedule in the header comment (vmcnt(10) / vmcnt(6)) assumes two pieces per stage");
  __shared__ __attribute__((aligned(1024))) char smem[2 * AREA];
  const unsigned lds0 = (unsigned)(uintptr_t)(p8_lds_t*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  // ---- block -> (tile, batch, K split): each XCD (= each L2) gets a contiguous run of the processing order (gemm_ring64.hip)
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
  const int kend = min(p.K, (kt_begin + nk) * BK);            // K tiles past this block's range (the two dummy tiles behind the last one) read zeros

  // ---- DMA slots.  Lane's chunk inside a 1 KiB wave instruction: LDS row 8 (j*8 + wave) + (lane >> 3) of the half, physical chunk lane & 7;
  // the swizzle key (row >> 1) & 7 = ((wave & 1) << 2) | ((lane >> 4) & 3) is the same for every slot of the lane, so is its K offset kl.
  const int kl = ((lane & 7) ^ (((wave & 1) << 2) | ((lane >> 4) & 3))) * CH;
  const int Cin = p.Cin, Wd = p.W, Hd = p.H, lda = p.lda, strd = p.stride, pad = p.pad, KS = p.KS;
  // Row addresses are 32-bit ELEMENT offsets from the wave-uniform A / B (P8_NOROW = no such row: the DMA reads the zero page); the launcher routes
  // operands of 2^32 elements or more to the ring kernels.  (As 64-bit pointers the gather variants spilled two VGPRs, reloaded -- behind a full
  // vmcnt(0) drain of the DMA queue -- at every filter-tap change inside the K loop: tests/test_host_logic.py reads the ISA for scratch use.)
  constexpr unsigned P8_NOROW = 0xffffffffu;
  unsigned a_base[4];                // slot sa = h*2 + j
  unsigned a_cur[4];
  int a_oyx[4];
  int kca[2], tapa[2], cca[2];       // per half: the two slots of a half advance together
  auto retap = [&](int sa, int tap) {
    if constexpr (GATHER == GATHER_NONE) {
      a_cur[sa] = a_base[sa];
    } else {
      int ky = 0, kx = 0;
      if (KS == 3) { ky = (tap * 11) >> 5; kx = tap - ky * 3; }
      const int oy = a_oyx[sa] >> 16, ox = a_oyx[sa] & 0xffff;
      int iy, ix;
      bool ok = a_base[sa] != P8_NOROW;
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
      a_cur[sa] = ok ? a_base[sa] + (unsigned)(iy * Wd + ix) * (unsigned)lda : P8_NOROW;
    }
  };
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    kca[h] = kt_begin * BK + kl;
    tapa[h] = 0;
    cca[h] = kca[h];
    if constexpr (GATHER != GATHER_NONE) {
      tapa[h] = kca[h] / Cin;
      cca[h] = kca[h] - tapa[h] * Cin;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int sa = h * 2 + j, r = (j * 8 + wave) * 8 + (lane >> 3);
      const int m = m0 + (r >> 6) * 128 + h * 64 + (r & 63);
      a_oyx[sa] = 0;
      if constexpr (GATHER == GATHER_NONE) {
        a_base[sa] = m < p.M ? (unsigned)m * (unsigned)lda : P8_NOROW;
      } else {
        const int hw = p.Ho * p.Wo, smp = m / hw, rem = m - smp * hw, oy = rem / p.Wo;
        a_oyx[sa] = (oy << 16) | (rem - oy * p.Wo);
        a_base[sa] = m < p.M ? (unsigned)(smp * Hd * Wd) * (unsigned)lda : P8_NOROW;
      }
      retap(sa, tapa[h]);
    }
  }
  unsigned b_base[4];
  int kcb[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    kcb[h] = kt_begin * BK + kl;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = (j * 8 + wave) * 8 + (lane >> 3);
      const int n = n0 + (r >> 5) * 64 + h * 32 + (r & 31);
      b_base[h * 2 + j] = n < p.N ? (unsigned)n * (unsigned)p.ldb : P8_NOROW;
    }
  }
  // FAST: byte offsets of the lane's rows from the block's first row / column (clamped into the matrix), K position as a uniform byte offset
  unsigned aoff[4], boff[4];
  const char* Ab = (const char*)(A + (long)m0 * lda);
  const char* Bb = (const char*)(B + (long)n0 * p.ldb);
  int kua[2], kub[2];
  const int kumax = (kt_begin + max(nk, 1) - 1) * (BK * 2);
  if constexpr (FAST) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      kua[h] = kub[h] = kt_begin * (BK * 2);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = (j * 8 + wave) * 8 + (lane >> 3);
        const int m = min(m0 + (r >> 6) * 128 + h * 64 + (r & 63), p.M - 1), n = min(n0 + (r >> 5) * 64 + h * 32 + (r & 31), p.N - 1);
        aoff[h * 2 + j] = (unsigned)((m - m0) * lda + kl) * 2u;
        boff[h * 2 + j] = (unsigned)((n - n0) * p.ldb + kl) * 2u;
      }
    }
  }
  // stage the next K tile of half h of A / B into parity d (2 DMA instructions)
  bool in_loop = false;
  auto stage_a = [&](auto hc, auto dc) {
    constexpr int h = decltype(hc)::value, d = decltype(dc)::value;
    if ((ABL & 1) && in_loop) return;
    char* dst = smem + h * HALF + d * PAR + wave * 1024;
    if constexpr (FAST) {
      const char* sb = Ab + min(kua[h], kumax);
#pragma unroll
      for (int j = 0; j < DMA_PER_STAGE; ++j) __builtin_amdgcn_global_load_lds((p8_gbl_t*)(sb + (unsigned long)aoff[h * 2 + j]), (p8_lds_t*)(dst + j * 8192), 16, 0, 0);
      kua[h] += BK * 2;
    } else {
#pragma unroll
      for (int j = 0; j < DMA_PER_STAGE; ++j) {
        const bf16* src = (a_cur[h * 2 + j] != P8_NOROW && kca[h] < kend) ? A + (unsigned long)(a_cur[h * 2 + j] + (unsigned)cca[h]) : zero;
        __builtin_amdgcn_global_load_lds((p8_gbl_t*)src, (p8_lds_t*)(dst + j * 8192), 16, 0, 0);
      }
      kca[h] += BK;
      cca[h] += BK;
      if constexpr (GATHER != GATHER_NONE) {
        if (cca[h] >= Cin) {                         // next filter tap (uniform across the wave: Cin % 64 == 0)
          do { cca[h] -= Cin; ++tapa[h]; } while (cca[h] >= Cin);
          retap(h * 2, tapa[h]);
          retap(h * 2 + 1, tapa[h]);
        }
      }
    }
  };
  auto stage_b = [&](auto hc, auto dc) {
    constexpr int h = decltype(hc)::value, d = decltype(dc)::value;
    if ((ABL & 1) && in_loop) return;
    char* dst = smem + AREA + h * HALF + d * PAR + wave * 1024;
    if constexpr (FAST) {
      const char* sb = Bb + min(kub[h], kumax);
#pragma unroll
      for (int j = 0; j < DMA_PER_STAGE; ++j) __builtin_amdgcn_global_load_lds((p8_gbl_t*)(sb + (unsigned long)boff[h * 2 + j]), (p8_lds_t*)(dst + j * 8192), 16, 0, 0);
      kub[h] += BK * 2;
    } else {
#pragma unroll
      for (int j = 0; j < DMA_PER_STAGE; ++j) {
        const bf16* src = (b_base[h * 2 + j] != P8_NOROW && kcb[h] < kend) ? B + (unsigned long)(b_base[h * 2 + j] + (unsigned)kcb[h]) : zero;
        __builtin_amdgcn_global_load_lds((p8_gbl_t*)src, (p8_lds_t*)(dst + j * 8192), 16, 0, 0);
      }
      kcb[h] += BK;
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // ---- fragment read addresses: row (wr*64 | wc*32) + l31 of a half, K16 substep kk: chunk (2 kk + lhi) ^ ((l31 >> 1) & 7); half / parity / the
  // second 32-row fragment are immediate offsets
  const int l31 = lane & 31, lhi = lane >> 5;
  unsigned adA[4], adB[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const unsigned ko = (unsigned)(((2 * kk + lhi) ^ ((l31 >> 1) & 7)) << 4);
    adA[kk] = lds0 + (wr * 64 + l31) * 128 + ko;
    adB[kk] = lds0 + AREA + (wc * 32 + l31) * 128 + ko;
  }

  f32x16 acc[2][2][2];               // [A half][32-row fragment][B half]
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][i][y][r] = 0.f;
  bf16x8 fa[2][4], fb0[4], fb1[4];
  if constexpr (ABL != 0) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { fa[0][kk] = fa[1][kk] = fb0[kk] = fb1[kk] = bf16x8{}; }
  }

  auto read_a = [&](auto hc, auto dc) {
    constexpr int o = decltype(hc)::value * HALF + decltype(dc)::value * PAR;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if constexpr (ABL & 2) { asm volatile("" : "+v"(fa[0][kk]), "+v"(fa[1][kk])); continue; }
      fa[0][kk] = p8_read<o>(adA[kk]);
      fa[1][kk] = p8_read<o + 4096>(adA[kk]);
    }
  };
  auto read_b = [&](auto hc, auto dc, bf16x8 (&fb)[4]) {
    constexpr int o = decltype(hc)::value * HALF + decltype(dc)::value * PAR;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if constexpr (ABL & 2) { asm volatile("" : "+v"(fb[kk])); continue; }
      fb[kk] = p8_read<o>(adB[kk]);
    }
  };
  auto mma = [&](auto xc, auto yc, bf16x8 (&fb)[4]) {
    constexpr int x = decltype(xc)::value, y = decltype(yc)::value;
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if constexpr (ABL & 4) { asm volatile("" ::"v"(fa[0][kk]), "v"(fa[1][kk]), "v"(fb[kk])); continue; }
      acc[x][0][y] = H16<FL>::mfma(fa[0][kk], fb[kk], acc[x][0][y]);
      acc[x][1][y] = H16<FL>::mfma(fa[1][kk], fb[kk], acc[x][1][y]);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  auto ktile = [&](auto dc) {
    using D = decltype(dc);
    using DX = std::integral_constant<int, D::value ^ 1>;
    // phase 1: a0 x b0 (b0 was read in phase 4 of the previous K tile / the prologue)
    read_a(I0{}, D{});
    stage_a(I1{}, DX{});
    bar();
    p8_wait12(fa, fb0);
    mma(I0{}, I0{}, fb0);
    bar();
    // phase 2: a0 x b1
    read_b(I1{}, D{}, fb1);
    stage_b(I0{}, D{});
    bar();
    p8_wait4(fb1);
    mma(I0{}, I1{}, fb1);
    bar();
    // phase 3: a1 x b0
    read_a(I1{}, D{});
    stage_a(I0{}, D{});
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_PH3) : "memory");    // B0 of K tile s+1 (issued five phases ago) has landed: read it in phase 4
    bar();
    p8_wait8(fa);
    mma(I1{}, I0{}, fb0);
    bar();
    // phase 4: a1 x b1, and b0 of the next K tile
    read_b(I0{}, DX{}, fb0);
    stage_b(I1{}, D{});
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_PH4) : "memory");     // the rest of K tile s+1 has landed (this wave's share); B0 / A0 / B1 of s+2 stay in flight
    bar();
    mma(I1{}, I1{}, fb1);
    p8_wait4(fb0);                                        // retired before this phase's second barrier: B0[d^1] may be restaged in phase 2 of the next tile
    bar();
  };

  if (nk > 0) {
    stage_b(I0{}, I0{}); stage_a(I0{}, I0{}); stage_b(I1{}, I0{}); stage_a(I1{}, I0{});      // K tile 0
    stage_b(I0{}, I1{}); stage_a(I0{}, I1{}); stage_b(I1{}, I1{});                            // K tile 1 without its A1 half (phase 1 of tile 0)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(VM_PH4) : "memory");
    bar();
    read_b(I0{}, I0{}, fb0);
    p8_wait4(fb0);                                   // retired before the first barrier of the loop: B0[0] is restaged in phase 2 of K tile 0
    if (wr == 1) bar();                              // group 1 runs one barrier behind group 0 from here on
    in_loop = true;
    for (int s = 0; s < nk; s += 2) {
      ktile(I0{});
      if (s + 1 >= nk) break;
      ktile(I1{});
    }
    if (wr == 0) bar();
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- epilogue through LDS: 32 accumulator rows per wave at a time -> 16-byte row-contiguous stores (epilogue.h).  epilogue_slab places a wave
  // at columns n0 + (wave & 1) * 64 of its n0: hand it the 128-column pair base of this wave.
  float* stg = reinterpret_cast<float*>(smem) + wave * 32 * SLD;
  const int n0w = n0 + (wc >> 1) * 128;
  auto stage_acc = [&](auto ic) {
    constexpr int x = decltype(ic)::value >> 1, i = decltype(ic)::value & 1;
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lhi) * SLD + y * 32 + l31] = acc[x][i][y][r];
  };
  if constexpr (EPI == EPI_PLAIN) {
    if (p.splitk <= 1 && p.vec_ok && !(p.N & 7)) {
      // The common epilogue (alpha, bias, row bias, residual, accumulate; 16-byte stores), with one resident block per CU: a wave stages and re-reads
      // only ITS OWN 32 x 64 slab, so the four slab rounds need no block barrier (LDS executes a wave's instructions in order), and the operands of
      // round r+1 -- which do not depend on the product -- are loaded while round r is staged and stored: one exposed memory round trip per wave
      // instead of four (K = 640 products spent a third of their time here).  Same additions in the same order as epilogue_slab: bitwise equal.
      const int c8 = lane & 7, r0 = lane >> 3, n = n0w + (wave & 1) * WN + c8 * 8;
      const bool act = n < p.N;                       // (the lane still STAGES its accumulators: other lanes read them)
      const OutBuf cb8 = out_buf(C, (long)p.M * p.ldc * 2);          // write-through stores through the output's buffer descriptor (common.h)
      {
        float b8[8];
        if (p.bias && act) { Vec<float>::load(p.bias + n, b8); Vec<float>::load(p.bias + n + 4, b8 + 4); }
        struct Ops { uint4 rr[4], ro[4], rb[4]; } ops[2];
        auto load_ops = [&](int round, Ops& o) {
          const int mrow0 = m0 + wr * 128 + (round >> 1) * 64 + (round & 1) * 32;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int mc = min(mrow0 + u * 8 + r0, p.M - 1);
            if (!act) continue;
            if (R) o.rr[u] = *reinterpret_cast<const uint4*>(R + (long)mc * p.ldr + n);
            if (p.accumulate) o.ro[u] = *reinterpret_cast<const uint4*>(C + (long)mc * p.ldc + n);
            if (p.rowbias) o.rb[u] = *reinterpret_cast<const uint4*>((const bf16*)p.rowbias + (long)((mc / p.rows_per_sample) / p.rowbias_div) * p.N + n);
          }
        };
        load_ops(0, ops[0]);
        static_for<0, 4>([&](auto ic) {
          constexpr int round = decltype(ic)::value;
          if constexpr (round + 1 < 4) load_ops(round + 1, ops[(round + 1) & 1]);
          stage_acc(ic);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const Ops& o = ops[round & 1];
          const int mrow0 = m0 + wr * 128 + (round >> 1) * 64 + (round & 1) * 32;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int row = u * 8 + r0, m = mrow0 + row;
            float v[8], t8[8];
            Vec<float>::load(stg + row * SLD + c8 * 8, v);
            Vec<float>::load(stg + row * SLD + c8 * 8 + 4, v + 4);
            if (m >= p.M || !act) continue;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
            if (p.bias) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += b8[e];
            }
            if (p.rowbias) {
              H16<FL>::load8(reinterpret_cast<const bf16*>(&o.rb[u]), t8);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += t8[e];
            }
            if (R) {
              H16<FL>::load8(reinterpret_cast<const bf16*>(&o.rr[u]), t8);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += t8[e];
            }
            if (p.accumulate) {
              H16<FL>::load8(reinterpret_cast<const bf16*>(&o.ro[u]), t8);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += t8[e];
            }
            store8_at<FL>(cb8, C + (long)m * p.ldc + n, v);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the slab is re-staged by the next round: keep its reads ahead of those writes
          __builtin_amdgcn_wave_barrier();
        });
      }
      return;
    }
  }
  static_for<0, 4>([&](auto ic) {
    constexpr int x = decltype(ic)::value >> 1, i = decltype(ic)::value & 1;
    stage_acc(ic);
    __syncthreads();
    epilogue_slab<FL, WN, SLD, EPI>(p, C, R, reinterpret_cast<const float*>(smem), wave, lane, m0 + wr * 128 + x * 64 + i * 32, n0w, (long)ksplit * gridDim.y + zb);
    __syncthreads();
  });
}

template <int FL>
static void launch_p8_f(const GemmArgs& a, dim3 grid, hipStream_t st) {
  const bool fast = a.gather == GATHER_NONE && a.K % 64 == 0;
#define DPB_P8(G, E) do { if (fast) hipLaunchKernelGGL((gemm_p8_kernel<G, FL, E, (G == GATHER_NONE)>), grid, dim3(512), 0, st, a); \
                          else hipLaunchKernelGGL((gemm_p8_kernel<G, FL, E, 0>), grid, dim3(512), 0, st, a); } while (0)
  switch (a.gather) {
    case GATHER_NONE:
      if (a.epi == EPI_GEGLU_TAN) DPB_P8(GATHER_NONE, EPI_GEGLU_TAN);
      else if (a.epi == EPI_GEGLU_ADJ) DPB_P8(GATHER_NONE, EPI_GEGLU_ADJ);
      else if (a.epi == EPI_GEGLU_FWD) DPB_P8(GATHER_NONE, EPI_GEGLU_FWD);
      else DPB_P8(GATHER_NONE, EPI_PLAIN);
      break;
    case GATHER_CONV: DPB_P8(GATHER_CONV, EPI_PLAIN); break;
    case GATHER_CONVT: DPB_P8(GATHER_CONVT, EPI_PLAIN); break;
    default: DPB_P8(GATHER_UPCONV, EPI_PLAIN); break;
  }
#undef DPB_P8
}

// the kernel addresses operand rows by 32-bit element offsets from A / B (per batch entry)
bool gemm_p8_fits32(const GemmArgs& a) {
  double ea = (double)a.M * a.lda;
  if (a.gather != GATHER_NONE && a.Ho > 0 && a.Wo > 0) ea = ((double)((a.M + a.Ho * a.Wo - 1) / (a.Ho * a.Wo)) + 1.0) * a.H * a.W * a.lda;
  return ea + a.K < 4294967295.0 && (double)a.N * a.ldb + a.K < 4294967295.0;
}

// tile code 530: 256 x 256 x 64, 8 waves, 4 phases per K tile
int launch_gemm_p8(const GemmArgs& a, int tile, hipStream_t st) {
  if (tile != 530) { set_error("gemm: unknown 8-phase tile code %d", tile); return -1; }
  if (!gemm_p8_fits32(a)) { set_error("gemm: the 8-phase tile addresses operands by 32-bit element offsets (M = %d, lda = %d, N = %d, ldb = %d)", a.M, a.lda, a.N, a.ldb); return -1; }
  if (a.epi == EPI_LN_TAN || a.epi == EPI_LN_ADJ || (a.epi != EPI_PLAIN && a.gather != GATHER_NONE)) { set_error("gemm: the 8-phase tile has the plain and GEGLU epilogues only"); return -1; }
  if (a.gather != GATHER_NONE && a.Cin % 64) { set_error("gemm: the 8-phase tile gathers whole 64-channel K tiles (Cin = %d)", a.Cin); return -1; }
  const int sk = a.splitk > 1 ? a.splitk : 1;
  const dim3 grid(((a.M + 255) / 256) * ((a.N + 255) / 256), a.Z1 * a.Z2, sk);
  if (a.fl) launch_p8_f<1>(a, grid, st);
  else launch_p8_f<0>(a, grid, st);
  DPB_CHECK(hipGetLastError());
  return 0;
}

}  // namespace dpb

// Weights-resident streaming product for the short, narrow linear layers of the 64 x 64 level: C[M][N] = A[M][320] W[N][320]^T, N a multiple of 320, 16-bit
// operands (the proj_in / attention out-proj / cross-attention q, out / proj_out products of diffusers' Transformer2DModel at 320 channels, tangent and
// adjoint passes: the linear layers under /root/reference/src/utils/utils.py:466-499 with M = tangents x 4096 rows).
//
// Why its own kernel (round 6): these products move 26 bytes per row pair for 4 GFLOP per 20480 rows -- arithmetic intensity ~100 flop/B, bound by their own
// operand traffic -- yet a tile kernel spends its time in per-tile prologues: with K = 320 a 128 x 128 tile is five K steps between a cold first DMA and an
// epilogue, and the ring kernels reach 1.6-2.0 TB/s on them (16.7 us for 20480 rows, 204 us for 327680; profiles/r05_roofline_report.md).  Here the WEIGHTS
// stay put and the activations stream:
//   * one persistent block of FIVE waves per CU; wave w owns output columns 64 w .. 64 w + 63 of the block's 320-column slice and keeps its 64 x 320 weight
//     slice in REGISTERS for the whole launch, as the 2 x 20 MFMA B fragments it is consumed as (160 VGPRs; loaded once, 200 KB per CU through L2);
//   * the activation rows stream through a 4-deep LDS ring of 32-row tiles (32 x 320 = 20 KiB each, LDS-DMA, three tiles in flight per CU), all five waves
//     read every tile (ds_read_b128, the 128-byte-row source-swizzled image of gemm_ring64.hip: five [32][64] K blocks per tile);
//   * the epilogue is wave-local: accumulators -> the wave's fp32 staging slab -> 16-byte row segments (one full 128-byte line per row and wave); its one
//     row operand (residual, or the cotangent accumulated so far) arrives by LDS-DMA too, so the loop holds NO register-destination load and every wait is a
//     hand-counted vmcnt (the compiler inserts none: all LDS accesses in the loop are inline asm, cf. cdna_hip_programming.md "three .s-level traps").
// Same MFMA sequence per output element as the BK = 64 ring (k ascending in 16-wide steps, v_mfma_f32_32x32x16) and the same epilogue arithmetic: results
// are bitwise those of gemm_ring64_kernel (tests/test_gpu_parity.py).
//
// vmcnt bookkeeping (per wave; loads retire in order, stores may not: a wait for "all but the youngest n LOADS" is safe whatever the stores do):
//   per iteration: [XP operand pieces] [4 tile pieces] ... [4 stores]; tile `it` was issued three iterations ago -> younger loads 2 x (4 + XP) -> vmcnt(8 | 16);
//   the operand of tile `it` is followed by the 4 pieces of tile it + 3 only -> vmcnt(4) in front of the epilogue.
#include "epilogue.h"

namespace dpb {

typedef __attribute__((address_space(3))) void wr_lds_t;
typedef __attribute__((address_space(1))) const void wr_gbl_t;

template <int OFF>
__device__ __forceinline__ bf16x8 wr_read(unsigned addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ void wr_read8f(unsigned addr, float (&o)[8]) {     // two 16-byte halves of eight staged fp32 values
  typedef __attribute__((ext_vector_type(4))) float f4;
  f4 a, b;
  asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(a), "=&v"(b) : "v"(addr), "n"(OFF), "n"(OFF + 16));
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b));
#pragma unroll
  for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[4 + e] = b[e]; }
}
template <int OFF>
__device__ __forceinline__ void wr_write(unsigned addr, float v) {
  asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ void wr_wait4(bf16x8 (&f)[4]) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])); }
__device__ __forceinline__ void wr_wait4_keep4(bf16x8 (&f)[4]) { asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])); }

// HASX: one 16-bit row operand X[M][ldx] is added to the scaled product (the residual R, or C itself when accumulating)
template <int FL, int HASX>
__global__ __launch_bounds__(320) void gemm_wres_kernel(GemmArgs p) {
  constexpr int K = 320, KB = K / 64, KS = K / 16, BM = 32, WAVES = 5, D = 4;
  constexpr int TILE = BM * K * 2;                         // 20 KiB: [K block][32 rows][128 B]
  constexpr int SLD = 68, STG = 32 * SLD * 4;              // per-wave fp32 staging slab [32][68]
  constexpr int XB = 32 * 64 * 2;                          // per-wave operand slab [32 rows][64 columns] 16-bit, lane-linear (the lane that fetched a chunk reads it)
  constexpr int XP = HASX ? 4 : 0;                         // LDS-DMA pieces per wave, tile and operand
  constexpr int VM_TILE = (D - 2) * (4 + XP);              // tile `it` landed: the two younger tiles (+ their operands) may still be in flight
  constexpr int VM_X = 4;                                  // operand of tile `it` landed: only the 4 pieces of tile it + D - 1 are younger
  static_assert(VM_TILE < 64, "vmcnt is a 6-bit counter");
  static_assert(D * TILE + WAVES * (STG + XB) <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(1024))) char smem[D * TILE + WAVES * (STG + XB)];
  const unsigned lds0 = (unsigned)(uintptr_t)(wr_lds_t*)smem;

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ncol0 = blockIdx.y * 320 + wave * 64;          // first output column of the wave
  const bf16* A = (const bf16*)p.A;
  const bf16* W = (const bf16*)p.B;
  bf16* C = (bf16*)p.C;
  const bf16* X = HASX ? (p.R ? (const bf16*)p.R : (const bf16*)p.C) : nullptr;
  const int ldx = p.R ? p.ldr : p.ldc;
  const int M = p.M, lda = p.lda, ldc = p.ldc;
  const int ntiles = (M + BM - 1) / BM, G = gridDim.x, first = blockIdx.x;
  if (first >= ntiles) return;                             // (uniform per block)
  const int n_my = (ntiles - first + G - 1) / G;

  // ---- tile loader: wave w moves K block w of a tile, 8 rows per 1 KiB piece; lane -> row 8 i + (lane >> 3), physical 16-byte chunk lane & 7 holding logical
  // chunk (lane & 7) ^ ((row >> 1) & 7) (source-side swizzle, gemm_ring64.hip).  Rows past M re-read row M - 1 (their products are never stored).
  unsigned a_col[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * i + (lane >> 3);
    a_col[i] = (unsigned)(64 * wave + (((lane & 7) ^ ((row >> 1) & 7)) << 3));
  }
  auto issue_tile = [&](int it, int slot) {
    const int m0 = (first + it * G) * BM;
    char* dst = smem + slot * TILE + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = min(m0 + 8 * i + (lane >> 3), M - 1);
      __builtin_amdgcn_global_load_lds((wr_gbl_t*)(A + (unsigned long)((unsigned)m * (unsigned)lda + a_col[i])), (wr_lds_t*)(dst + i * 1024), 16, 0, 0);
    }
  };
  char* const xs = smem + D * TILE + WAVES * STG + wave * XB;
  auto issue_x = [&](int it) {
    if constexpr (HASX) {
      const int m0 = (first + it * G) * BM;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = min(m0 + 8 * i + (lane >> 3), M - 1);
        __builtin_amdgcn_global_load_lds((wr_gbl_t*)(X + (unsigned long)((unsigned)m * (unsigned)ldx + (unsigned)(ncol0 + (lane & 7) * 8))), (wr_lds_t*)(xs + i * 1024), 16,
                                         0, 0);
      }
    }
  };

  // ---- prologue: three tiles in flight, then the weight slice (B fragments of v_mfma_f32_32x32x16: lane -> column 32 j + l31, k = 16 s + 8 lhi .. + 7).
  // Four consecutive k steps of a row share one 128-byte line: issued back to back they cost one L2 request each.
#pragma unroll
  for (int it = 0; it < D - 1; ++it) issue_tile(it, it);
  bf16x8 wf[2][KS];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const bf16* wr = W + (long)(ncol0 + 32 * j + l31) * p.ldb + 8 * lhi;
#pragma unroll
    for (int s = 0; s < KS; ++s) wf[j][s] = *reinterpret_cast<const bf16x8*>(wr + 16 * s);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // the compiler cannot see that the asm above waited for the weight loads: without this it parks its own vmcnt(0) in front of the first MFMA of
  // iteration 0 -- behind the operand and tile pieces that iteration has just issued (a full drain of the freshly started pipeline)
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(wf[j][s]));
  __builtin_amdgcn_s_barrier();

  // fragment read addresses inside a tile: K block kb (immediate offset kb * 4096), row l31, chunk (2 kk + lhi) ^ ((l31 >> 1) & 7)
  unsigned fo[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fo[kk] = lds0 + l31 * 128 + (((2 * kk + lhi) ^ ((l31 >> 1) & 7)) << 4);
  const unsigned stg = lds0 + D * TILE + wave * STG;
  const unsigned stg_w = stg + (4 * lhi * SLD + l31) * 4;                    // accumulator register r -> row (r & 3) + 8 (r >> 2) + 4 lhi, column 32 j + l31
  const unsigned stg_r = stg + ((lane >> 3) * SLD + (lane & 7) * 8) * 4;     // item u -> row 8 u + (lane >> 3), columns 8 (lane & 7) .. + 7
  const unsigned xs_r = lds0 + D * TILE + WAVES * STG + wave * XB + lane * 16;
  const float alpha = p.alpha;
  const OutBuf cbw = out_buf(C, (long)M * ldc * 2);          // write-through stores through the output's buffer descriptor (common.h)

  int slot = 0;
  for (int it = 0; it < n_my; ++it) {
    if (it > 0) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_TILE) : "memory");         // this wave's pieces of tile `it` have landed ...
      __builtin_amdgcn_s_barrier();                                           // ... and everybody's; tile it - 1 has been read by every wave
    }
    issue_x(it);
    {
      int ns = slot + D - 1; if (ns >= D) ns -= D;
      issue_tile(it + D - 1, ns);                                             // (past the block's last tile: clamped rows, never read -- the piece count stays static)
    }
    // ---- 32 x 320 x 320: 20 k steps, the fragments of K block kb + 1 in flight under the MFMAs of block kb
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const unsigned tb = (unsigned)(slot * TILE);
    bf16x8 fa[2][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fa[0][kk] = wr_read<0>(tb + fo[kk]);
    static_for<0, KB>([&](auto kbc) {
      constexpr int kb = decltype(kbc)::value, cur = kb & 1, nxt = cur ^ 1;
      if constexpr (kb + 1 < KB) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fa[nxt][kk] = wr_read<(kb + 1) * 4096>(tb + fo[kk]);
        wr_wait4_keep4(fa[cur]);
      } else {
        wr_wait4(fa[cur]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        acc[0] = H16<FL>::mfma(fa[cur][kk], wf[0][kb * 4 + kk], acc[0]);
        acc[1] = H16<FL>::mfma(fa[cur][kk], wf[1][kb * 4 + kk], acc[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // ---- wave-local epilogue: stage, then 4 items per lane (row 8 u + (lane >> 3), 8 columns): alpha, + operand, 16-byte store.
    // The staging writes are inline asm: hipcc's hazard recogniser does not pad an asm statement, and an LDS instruction that reads a VGPR an MFMA is
    // still writing gets the OLD value (the first two writes after the last MFMA did: rows 0, 1, 4, 5 of every tile, run-to-run different) -- the wait
    // states of the longest XDL write -> LDS-data read hazard are spent here by hand (cdna_hip_programming.md 5.7).
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]));
    static_for<0, 2>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      static_for<0, 16>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        wr_write<(((r & 3) + 8 * (r >> 2)) * SLD + 32 * j) * 4>(stg_w, acc[j][r]);
      });
    });
    if constexpr (HASX) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(VM_X) : "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int m0 = (first + it * G) * BM + (lane >> 3);
    static_for<0, 4>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      float v[8];
      wr_read8f<u * 8 * SLD * 4>(stg_r, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= alpha;
      if constexpr (HASX) {
        bf16x8 xr = wr_read<u * 1024>(xs_r);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xr));
        float t8[8];
        H16<FL>::load8(reinterpret_cast<const bf16*>(&xr), t8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += t8[e];
      }
      const int m = m0 + 8 * u;
      if (m < M) store8_at<FL>(cbw, C + (unsigned long)((unsigned)m * (unsigned)ldc + (unsigned)(ncol0 + (lane & 7) * 8)), v);
    });
    if (++slot == D) slot = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the dummy pieces behind the last tile must not outlive the block's LDS allocation
}

// the product shapes the kernel takes (host side; the dispatch in gemm.hip asks before it plans a ring tile)
bool gemm_wres_supported(int dtype, const GemmArgs& a) {
  if (dtype == DT_F32 || a.gather != GATHER_NONE || a.epi != EPI_PLAIN || a.A2 || a.Z1 * a.Z2 != 1) return false;
  if (a.K != 320 || a.N % 320 || a.M < 32) return false;
  if (a.bias || a.rowbias || (a.R && a.accumulate)) return false;                      // one 16-bit row operand at most (tangent / adjoint products)
  if ((a.lda & 7) || (a.ldb & 7) || (a.ldc & 7) || (a.R && (a.ldr & 7))) return false;
  if (((uintptr_t)a.A | (uintptr_t)a.B | (uintptr_t)a.C | (uintptr_t)a.R) & 15) return false;
  const double lim = 4294967295.0;                                                        // rows are addressed by 32-bit element offsets
  return (double)a.M * a.lda < lim && (double)a.M * a.ldc < lim && (!a.R || (double)a.M * a.ldr < lim);
}

int launch_gemm_wres(const GemmArgs& a, hipStream_t st) {
  const int dtype = a.fl ? DT_F16 : DT_BF16;
  if (!gemm_wres_supported(dtype, a)) { set_error("gemm: the weights-resident kernel takes plain 16-bit products with K = 320, N %% 320 == 0 and at most one row operand"); return -1; }
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) pr.multiProcessorCount = 256;
    cus = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
  }
  const int ntiles = (a.M + 31) / 32;
  const dim3 grid((unsigned)std::min(cus, ntiles), (unsigned)(a.N / 320));
  const bool hasx = a.R || a.accumulate;
  if (a.fl) { if (hasx) hipLaunchKernelGGL((gemm_wres_kernel<1, 1>), grid, dim3(320), 0, st, a); else hipLaunchKernelGGL((gemm_wres_kernel<1, 0>), grid, dim3(320), 0, st, a); }
  else { if (hasx) hipLaunchKernelGGL((gemm_wres_kernel<0, 1>), grid, dim3(320), 0, st, a); else hipLaunchKernelGGL((gemm_wres_kernel<0, 0>), grid, dim3(320), 0, st, a); }
  DPB_CHECK(hipGetLastError());
  return 0;
}

}  // namespace dpb

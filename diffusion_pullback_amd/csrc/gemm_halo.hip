// bf16 3x3 stride-1 convolution (forward gather and its adjoint), halo-tile form of the LDS-ring MFMA GEMM.
//
// gemm_ring64.hip runs a 3x3 convolution as an implicit GEMM: every K-step DMAs a fresh [pixels][64 channels] A tile, so
// each input pixel travels L2 -> LDS nine times (once per filter tap), and the measured limiter of that kernel is exactly
// this DMA stream (profiles/r01_gemm_ablation.txt).  Here a block owns 256 output pixels = R whole image rows of one
// sample and keeps the (R+2) x (W+2) pixel HALO of one 64-channel chunk in LDS; the nine taps of that chunk read their A
// fragments from it at shifted pixel offsets, and only the [128 x 64] weight tile streams per K-step:
//   bytes through the DMA per 256x128 output tile and 64-channel chunk:  ~50 KB halo + 9 x 16 KB weights = 194 KB
//   versus 9 x (32 + 16) KB = 432 KB for the implicit-GEMM form of the same tile (0.45x; 0.34x of two 128x128 tiles).
// 8 waves (4 x 2, 64x64 wave tiles) so that one resident block still gives every SIMD two waves; LDS = 2 halo buffers
// (the next chunk's halo lands while the current one is consumed) + a 3-stage weight ring = 152 KB.
//
// K order is chunk-major (chunk, then tap), so results differ from the tap-major kernels in fp32 association only.
// Halo pixel p of a buffer lives at p*128 B, chunk c at ((c ^ ((p >> 1) & 7)) * 16 B (same source-side swizzle as the
// ring: conflict-free ds_read_b128 for runs of consecutive pixels); out-of-image pixels read the zero page.
// The weight-ring / barrier / in-wave fragment-prefetch structure and the epilogue are those of gemm_ring64.hip.
#include "epilogue.h"

namespace dpb {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__device__ inline bf16x8 halo_lds_read(unsigned addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
__device__ inline void halo_wait_frags4(bf16x8 (&a)[2], bf16x8 (&b)[2]) {
  asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
}
__device__ inline void halo_wait_frags0(bf16x8 (&a)[2], bf16x8 (&b)[2]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
}
__device__ inline void halo_wait_vm(int n) {       // n in {0, 2, 4, 7, 9, 11, 13}: DMA instructions allowed to stay in flight
  if (n >= 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
  else if (n == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
  else if (n == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if (n == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

constexpr int HALO_BM = 256, HALO_BN = 128, HALO_NH = 7;               // NH: halo DMA instructions per wave (8 x 7 x 64 >= 396 pixels x 8 chunks)
constexpr int HALO_MAXPIX = 400;
constexpr int HALO_BYTES = ((HALO_MAXPIX * 8 + 63) / 64) * 1024;        // whole wave instructions: 51,200 B
constexpr int HALO_BSTAGE = HALO_BN * 128, HALO_S = 3;

template <int GATHER, int FL = 0>   // FL: 16-bit flavour (H16<FL>): 0 bf16, 1 f16
__global__ __launch_bounds__(512) void conv_halo_kernel(GemmArgs p) {
  constexpr int BM = HALO_BM, BN = HALO_BN, WAVES = 8, NH = HALO_NH, S = HALO_S, KK = 4;
  constexpr int NIB = BN / (8 * WAVES);                                  // 2 weight DMA instructions per wave and stage
  constexpr int WN = BN / 2, SLD = WN + 4;
  constexpr int B0 = 2 * HALO_BYTES, DUMMY = B0 + S * HALO_BSTAGE;       // LDS map: halo 0 | halo 1 | weight ring | 1 KiB sink for unused DMA slots
  constexpr int SMEM_BYTES = DUMMY + 1024;
  static_assert(SMEM_BYTES >= WAVES * 32 * SLD * 4 && SMEM_BYTES <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(128))) char smem[SMEM_BYTES];
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tilesN = (p.N + BN - 1) / BN, tilesM = (p.M + BM - 1) / BM;
  int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  {
    const int nwg = gridDim.x * gridDim.y * gridDim.z, q = nwg >> 3, r = nwg & 7, xcd = lin & 7, idx = lin >> 3;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tm, tn;
  if (p.order == 0) { tn = lin % tilesN; lin /= tilesN; tm = lin % tilesM; lin /= tilesM; }
  else { tm = lin % tilesM; lin /= tilesM; tn = lin % tilesN; lin /= tilesN; }
  const int ksplit = lin;                                                 // Z1 = Z2 = 1 (checked on the host)
  const int m0 = tm * BM, n0 = tn * BN;
  const bf16* A = (const bf16*)p.A;
  const bf16* B = (const bf16*)p.B;
  bf16* C = (bf16*)p.C;
  const bf16* R = (const bf16*)p.R;
  const bf16* zero = (const bf16*)p.zeros;

  // geometry: the tile = image rows y0 .. y0+RT-1 of one sample (H*W a multiple of 256), or SPT whole samples (H*W < 256,
  // e.g. the four 8x8 images of a 256-pixel tile), each with its own zero-padded halo
  const int W = p.W, H = p.H, Cin = p.Cin, lda = p.lda, HWp = W + 2;
  const int hw = H * W, SPT = hw >= BM ? 1 : BM / hw, RT = hw >= BM ? BM / W : H;
  const int smp = m0 / hw, y0 = hw >= BM ? (m0 - smp * hw) / W : 0, nsmp = p.M / hw;
  const int hpix = (RT + 2) * HWp, npix = SPT * hpix, tpix = RT * W;      // halo pixels per sample, per tile; output pixels per sample
  // K range in chunks of 64 input channels
  const int nch_all = Cin / 64;
  int c_begin = 0, nch = nch_all;
  if (p.splitk > 1) {
    const int per = (nch_all + p.splitk - 1) / p.splitk;
    c_begin = ksplit * per;
    nch = max(0, min(nch_all, c_begin + per) - c_begin);
  }
  const int ns = nch * 9;                                                 // stages: (chunk, tap)

  // ---- halo DMA slots: wave instruction i covers LDS chunks (wave*NH + i)*64 + lane = pixel slot>>3, physical chunk slot&7
  const bf16* h_src[NH];                                                  // source pixel's channel vector + logical chunk offset (nullptr = zero page)
  int h_dst[NH];                                                          // byte offset inside a halo buffer, or -1: sink
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const int inst = wave * NH + i, slot = inst * 64 + lane, pix = slot >> 3, phys = slot & 7;
    h_dst[i] = inst * 1024 < HALO_BYTES ? inst * 1024 : -1;
    h_src[i] = nullptr;
    if (pix < npix) {
      const int sl = pix / hpix, rp = pix - sl * hpix, hy = rp / HWp, hx = rp - hy * HWp, iy = y0 - 1 + hy, ix = hx - 1;
      if (smp + sl < nsmp && iy >= 0 && iy < H && ix >= 0 && ix < W)
        h_src[i] = A + ((long)((smp + sl) * H + iy) * W + ix) * lda + ((phys ^ ((pix >> 1) & 7)) << 3);
    }
  }
  const bf16* b_src[NIB];
#pragma unroll
  for (int i = 0; i < NIB; ++i) {
    const int pos = (wave * NIB + i) * 64 + lane, row = pos >> 3, phys = pos & 7, n = n0 + row;
    b_src[i] = n < p.N ? B + (long)n * p.ldb + ((phys ^ ((row >> 1) & 7)) << 3) : nullptr;
  }
  auto issue_halo = [&](int c) {                                          // chunk c (absolute) -> halo buffer c & 1
    char* hb = smem + (c & 1) * HALO_BYTES;
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const bf16* src = h_src[i] ? h_src[i] + c * 64 : zero;
      char* dst = h_dst[i] >= 0 ? hb + h_dst[i] : smem + DUMMY;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)dst, 16, 0, 0);
    }
  };
  auto issue_b = [&](int s) {                                             // stage s = (chunk c_begin + s/9, tap s%9) -> ring slot s % S
    const int c = c_begin + s / 9, tap = s - (s / 9) * 9;
    char* st = smem + B0 + (s % S) * HALO_BSTAGE;
    const int koff = tap * Cin + c * 64;
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      const bf16* src = b_src[i] ? b_src[i] + koff : zero;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(st + (wave * NIB + i) * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wy = wave >> 1, wx = wave & 1, l31 = lane & 31, lhi = lane >> 5;
  int pixm[2];                                                            // halo pixel of the CENTRE tap for this lane's two A-fragment rows
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = wy * 64 + i * 32 + l31, sl = m / tpix, r = m - sl * tpix, oy = r / W, ox = r - oy * W;
    pixm[i] = sl * hpix + (oy + 1) * HWp + ox + 1;
  }
  unsigned fb0[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = wx * WN + j * 32 + l31;
    fb0[j] = B0 + row * 128 + ((lhi ^ ((row >> 1) & 7)) << 4);
  }
  bf16x8 fa[2][2], fb[2][2];
  unsigned abase[2], asw[2];                                              // per stage: byte address of the tap's pixel, its swizzle
  auto set_stage = [&](int s) {
    const int c = c_begin + s / 9, tap = s - (s / 9) * 9, ky = (tap * 11) >> 5, kx = tap - ky * 3;
    const int toff = GATHER == GATHER_CONV ? (ky - 1) * HWp + (kx - 1) : (1 - ky) * HWp + (1 - kx);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pix = pixm[i] + toff;
      abase[i] = lds0 + (c & 1) * HALO_BYTES + pix * 128;
      asw[i] = (pix >> 1) & 7;
    }
  };
  auto read_frags = [&](int s, int kk, int buf) {
    const unsigned sbB = lds0 + (s % S) * HALO_BSTAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[buf][i] = halo_lds_read(abase[i] + ((((kk << 1) | lhi) ^ asw[i]) << 4));
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[buf][j] = halo_lds_read(sbB + (fb0[j] ^ (kk << 5)));
  };

  if (ns > 0) {
    // prologue: halo(c_begin), B(0), B(1), then what the boundary before stage 0 issues: halo(c_begin + 1), B(2)
    issue_halo(c_begin);
    issue_b(0);
    if (ns > 1) issue_b(1);
    if (nch > 1) issue_halo(c_begin + 1);
    if (ns > 2) issue_b(2);
    halo_wait_vm((ns > 1 ? 2 : 0) + (nch > 1 ? NH : 0) + (ns > 2 ? 2 : 0));      // everything up to B(0) has landed
    __builtin_amdgcn_s_barrier();
    set_stage(0);
    read_frags(0, 0, 0);
    for (int s = 0; s < ns; ++s) {
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk + 1 < KK) {
          read_frags(s, kk + 1, nxt);
          halo_wait_frags4(fa[cur], fb[cur]);
        } else {
          halo_wait_frags0(fa[cur], fb[cur]);                            // every LDS read of stage s by this wave is done
          if (s + 1 < ns) {
            const int tap_s = s - (s / 9) * 9, ch_s = s / 9;
            // B(s+1) landed: later in the queue are B(s+2) and, if stage s opened a chunk, the halo issued before it
            halo_wait_vm((s + 2 < ns ? 2 : 0) + ((tap_s == 0 && ch_s + 1 < nch) ? NH : 0));
            __builtin_amdgcn_s_barrier();                                 // stage s+1 (and its halo) visible to all; stage s consumed by all
            const int tap_n = tap_s == 8 ? 0 : tap_s + 1, ch_n = tap_s == 8 ? ch_s + 1 : ch_s;
            if (tap_n == 0 && ch_n + 1 < nch) issue_halo(c_begin + ch_n + 1);   // buffer of chunk ch_n - 1: all its stages are consumed
            if (s + 3 < ns) issue_b(s + 3);
            set_stage(s + 1);
            read_frags(s + 1, 0, nxt);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = H16<FL>::mfma(fa[cur][i], fb[cur][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- epilogue through LDS: the staging layout and the slab writer of the ring kernels (epilogue.h)
  float* stage = reinterpret_cast<float*>(smem) + wave * 32 * SLD;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * lhi) * SLD + j * 32 + l31] = acc[i][j][r];
    __syncthreads();
    epilogue_slab<FL, WN, SLD, EPI_PLAIN>(p, C, R, reinterpret_cast<const float*>(smem), wave, lane, m0 + wy * 64 + i * 32, n0, (long)ksplit);
    __syncthreads();
  }
}

// 3x3, stride 1, pad 1, forward gather or its adjoint; whole image rows (or whole small images) per tile; 64-channel chunks
int conv_halo_supported(const GemmArgs& a) {
  if (a.gather != GATHER_CONV && a.gather != GATHER_CONVT) return 0;
  if (a.KS != 3 || a.stride != 1 || a.pad != 1 || a.Z1 * a.Z2 != 1 || a.A2) return 0;
  if (a.H != a.Ho || a.W != a.Wo || a.Cin % 64 || a.K != 9 * a.Cin) return 0;
  const int hw = a.H * a.W;
  if (a.M % hw) return 0;
  if (hw >= HALO_BM) {                              // whole image rows of one sample per tile
    if (HALO_BM % a.W || hw % HALO_BM) return 0;
    if ((HALO_BM / a.W + 2) * (a.W + 2) > HALO_MAXPIX) return 0;
  } else {                                          // whole samples per tile (8x8: four)
    if (HALO_BM % hw || a.W < 8) return 0;
    if ((HALO_BM / hw) * (a.H + 2) * (a.W + 2) > HALO_MAXPIX) return 0;
  }
  if (a.lda % 8 || a.ldb % 8 || !a.zeros) return 0;
  return 1;
}

int launch_conv_halo(const GemmArgs& a, hipStream_t st) {
  const int sk = a.splitk > 1 ? a.splitk : 1;
  dim3 grid(((a.M + HALO_BM - 1) / HALO_BM) * ((a.N + HALO_BN - 1) / HALO_BN), 1, sk);
  if (a.gather == GATHER_CONV) {
    if (a.fl) hipLaunchKernelGGL((conv_halo_kernel<GATHER_CONV, 1>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv_halo_kernel<GATHER_CONV, 0>), grid, dim3(512), 0, st, a);
  } else {
    if (a.fl) hipLaunchKernelGGL((conv_halo_kernel<GATHER_CONVT, 1>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv_halo_kernel<GATHER_CONVT, 0>), grid, dim3(512), 0, st, a);
  }
  DPB_CHECK(hipGetLastError());
  return 0;
}

}  // namespace dpb

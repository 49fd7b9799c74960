// Probe of gfx950's LDS transpose read (ds_read_b64_tr_b16): which element does lane l / register j receive?
// hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o diffusion_pullback_amd/csrc/build/tr_probe && (on the GPU box) ./.../tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short short4_;
constexpr int LD = 72;
__global__ void k(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short sm[64 * LD];
  const int tid = threadIdx.x;
  for (int i = tid; i < 64 * LD; i += 64) sm[i] = (unsigned short)((i / LD) * 100 + (i % LD));   // value = row*100 + col
  __syncthreads();
  const int i16 = tid & 15, g = tid >> 4;
  const unsigned short* p = sm + (i16 >> 2) * LD + g * 16 + (i16 & 3) * 4;   // lane i: row i/4, cols g*16 + (i%4)*4 ..+3
  short4_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_*)p);
  for (int j = 0; j < 4; ++j) out[tid * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  return 0;
}

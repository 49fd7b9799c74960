// Shared device helpers for the gfx950 (CDNA4, wave64) pullback kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dpb {

enum { DT_F32 = 0, DT_BF16 = 1 };

struct bf16 {
  unsigned short v;
};

__host__ __device__ inline float bf2f(unsigned short v) {
  union { unsigned u; float f; } c;
  c.u = ((unsigned)v) << 16;
  return c.f;
}
__host__ __device__ inline unsigned short f2bf(float f) {   // round to nearest even
#if defined(__HIP_DEVICE_COMPILE__)
  __bf16 b = (__bf16)f;                   // v_cvt_pk_bf16_f32 on gfx950
  return *reinterpret_cast<unsigned short*>(&b);
#else
  union { unsigned u; float f; } c;
  c.f = f;
  unsigned u = c.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
#endif
}

template <typename T> struct TT;
template <> struct TT<float> {
  static constexpr int CH = 4;                 // elements per 16-byte chunk
  __device__ static inline float ld(const float* p) { return *p; }
  __device__ static inline void st(float* p, float v) { *p = v; }
};
template <> struct TT<bf16> {
  static constexpr int CH = 8;
  __device__ static inline float ld(const bf16* p) { return bf2f(p->v); }
  __device__ static inline void st(bf16* p, float v) { p->v = f2bf(v); }
};

// 16-byte vector load/store <-> float[CH]
template <typename T> struct Vec;
template <> struct Vec<float> {
  static constexpr int N = 4;
  __device__ static inline void load(const float* p, float* o) {
    float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  __device__ static inline void store(float* p, const float* o) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  }
};
template <> struct Vec<bf16> {
  static constexpr int N = 8;
  __device__ static inline void load(const bf16* p, float* o) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = __uint_as_float(w[i] << 16);
      o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ static inline void store(bf16* p, const float* o) {
    typedef __attribute__((ext_vector_type(8))) __bf16 v8;
    v8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (__bf16)o[i];          // 4 x v_cvt_pk_bf16_f32
    *reinterpret_cast<v8*>(p) = r;
  }
};

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ inline float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ inline float silu_(float x) { return x * sigmoidf_(x); }
__device__ inline float dsilu_(float x) {   // d/dx x*sigmoid(x)
  float s = sigmoidf_(x);
  return s * (1.f + x * (1.f - s));
}
__device__ inline float gelu_(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ inline float dgelu_(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}

#define DPB_CHECK(expr)                                                                  \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      dpb::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
      return -1;                                                                         \
    }                                                                                    \
  } while (0)

void set_error(const char* fmt, ...);
const char* last_error();

}  // namespace dpb

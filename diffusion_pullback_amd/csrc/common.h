// Shared device helpers for the gfx950 (CDNA4, wave64) pullback kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace dpb {

enum { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };

struct bf16 {   // bfloat16 storage; the specialised 16-bit kernels also use it as RAW 16-bit storage (flavour FL, H16<FL> below)
  unsigned short v;
};
struct f16 {    // IEEE half storage
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

__device__ inline float h2f(unsigned short v) {
  _Float16 h = *reinterpret_cast<_Float16*>(&v);
  return (float)h;
}
__device__ inline unsigned short f2h(float f) {   // round to nearest even (v_cvt_f16_f32); overflow -> inf
  _Float16 h = (_Float16)f;
  return *reinterpret_cast<unsigned short*>(&h);
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

template <> struct TT<f16> {
  static constexpr int CH = 8;
  __device__ static inline float ld(const f16* p) { return h2f(p->v); }
  __device__ static inline void st(f16* p, float v) { p->v = f2h(v); }
};

// How the 16-byte OUTPUT stores of the kernels leave the CU (round 6).  A plain store leaves its line dirty in the XCD's L2 and the kernel boundary writes
// every dirty line back before the dependent launch starts (MI355X_MICROARCH.md price list, row "boundary": + B / 6 TB/s for B dirty bytes -- 2 us behind a
// 13 MB activation of the 64 x 64 level, at ~335 boundaries per power iteration).  `sc1` stores are written through while the kernel runs, so the boundary has
// nothing left to flush: same-session A/Bs on the headline +0.8 ... +5.0 % depending on the box (`nt` less), profiles/r06_out_store_ab.txt.  Values are
// unchanged (a store flavour, not an arithmetic change).  DPB_OUT_STORE: 1 sc1 (default) | 0 plain | 2 nt | 3 sc0 sc1 -- the A/B builds of tools/build_variant.sh.
// Inline asm because no builtin carries cache bits on a flat global store; the trailing s_nop 1 keeps hipcc's next instruction from overwriting the data
// registers before the store has read them (cdna_hip_programming.md 5.7 item 1).  The data always comes from a VALU conversion, never straight from an MFMA.
#ifndef DPB_OUT_STORE
#define DPB_OUT_STORE 1
#endif
#ifndef DPB_OUT_STORE_F32
#define DPB_OUT_STORE_F32 0       // the same for 16-byte fp32 stores (split-K slabs, fp32 engines): A/B only
#endif
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
template <int MODE>
__device__ __forceinline__ void store_out16(void* p, u32x4_ v) {
  if constexpr (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");   // (A/B only: without the pad)
  else if constexpr (MODE == 5) *(volatile __attribute__((address_space(1))) u32x4_*)(p) = v;                                            // hipcc's own `sc0 sc1` store (hazards padded by the compiler)
  else *reinterpret_cast<u32x4_*>(p) = v;
}
template <int MODE>
__device__ __forceinline__ void store_out8(void* p, unsigned a, unsigned b) {     // 8-byte form (attention outputs: row-per-lane fragments)
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_;
  u32x2_ v = {a, b};
  if constexpr (MODE == 1) asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (MODE == 2) asm volatile("global_store_dwordx2 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else *reinterpret_cast<u32x2_*>(p) = v;
}

// 16-byte vector load/store <-> float[CH]
template <typename T> struct Vec;
template <> struct Vec<float> {
  static constexpr int N = 4;
  __device__ static inline void load(const float* p, float* o) {
    float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  __device__ static inline void store(float* p, const float* o) {
    if constexpr (DPB_OUT_STORE_F32 != 0) {
      u32x4_ v = {__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])};
      store_out16<DPB_OUT_STORE_F32>(p, v);
    } else {
      *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    }
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
    store_out16<DPB_OUT_STORE>(p, *reinterpret_cast<u32x4_*>(&r));
  }
};

template <> struct Vec<f16> {
  static constexpr int N = 8;
  typedef __attribute__((ext_vector_type(8))) _Float16 v8;
  __device__ static inline void load(const f16* p, float* o) {
    v8 r = *reinterpret_cast<const v8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)r[i];            // v_cvt_f32_f16
  }
  __device__ static inline void store(f16* p, const float* o) {
    v8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (_Float16)o[i];          // 4 x v_cvt_pk_f16_f32
    store_out16<DPB_OUT_STORE>(p, *reinterpret_cast<u32x4_*>(&r));
  }
};

// a 16-byte chunk held in registers between two phases of a kernel: raw bits <-> float[CH]
template <typename T> struct Raw;
template <> struct Raw<float> {
  __device__ static inline void unpack(const uint4& r, float* o) { o[0] = __uint_as_float(r.x); o[1] = __uint_as_float(r.y); o[2] = __uint_as_float(r.z); o[3] = __uint_as_float(r.w); }
  __device__ static inline uint4 pack(const float* o) { return make_uint4(__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])); }
};
template <> struct Raw<bf16> {
  __device__ static inline void unpack(const uint4& r, float* o) {
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
  __device__ static inline uint4 pack(const float* o) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (unsigned)f2bf(o[2 * i]) | ((unsigned)f2bf(o[2 * i + 1]) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
};
template <> struct Raw<f16> {
  __device__ static inline void unpack(const uint4& r, float* o) {
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[2 * i] = h2f((unsigned short)(w[i] & 0xffffu)); o[2 * i + 1] = h2f((unsigned short)(w[i] >> 16)); }
  }
  __device__ static inline uint4 pack(const float* o) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (unsigned)f2h(o[2 * i]) | ((unsigned)f2h(o[2 * i + 1]) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
};

// 16-bit flavour of the specialised MFMA kernels (LDS-ring GEMMs, halo convolution, fused attention).  Those kernels only move
// 16-bit words except at three places -- the MFMA opcode, fp32 -> 16-bit packing, 16-bit -> fp32 unpacking -- so they keep ONE
// raw storage type (struct bf16) and take the flavour as a template parameter: FL = 0 bfloat16, FL = 1 IEEE half.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int FL> struct H16;
template <> struct H16<0> {
  typedef bf16 T;
  __device__ static inline float up(unsigned short v) { return bf2f(v); }
  __device__ static inline unsigned short dn(float f) { return f2bf(f); }
  __device__ static inline float lo(unsigned w) { return __uint_as_float(w << 16); }            // halves of a packed pair
  __device__ static inline float hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
  __device__ static inline unsigned pack2(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }
  __device__ static inline f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  __device__ static inline void load8(const bf16* p, float* o) { Vec<bf16>::load(p, o); }
  __device__ static inline void store8(bf16* p, const float* o) { Vec<bf16>::store(p, o); }
  __device__ static inline bf16x8 pack8(const float* x) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (__bf16)x[i];           // v_cvt_pk_bf16_f32
    return r;
  }
};
template <> struct H16<1> {
  typedef f16 T;
  __device__ static inline float up(unsigned short v) { return h2f(v); }
  __device__ static inline unsigned short dn(float f) { return f2h(f); }
  __device__ static inline float lo(unsigned w) { return h2f((unsigned short)(w & 0xffffu)); }
  __device__ static inline float hi(unsigned w) { return h2f((unsigned short)(w >> 16)); }
  __device__ static inline unsigned pack2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    h2 r = {(_Float16)a, (_Float16)b};                        // v_cvt_pk_f16_f32
    return *reinterpret_cast<unsigned*>(&r);
  }
  __device__ static inline f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<f16x8*>(&a), *reinterpret_cast<f16x8*>(&b), c, 0, 0, 0);
  }
  __device__ static inline void load8(const bf16* p, float* o) { Vec<f16>::load(reinterpret_cast<const f16*>(p), o); }
  __device__ static inline void store8(bf16* p, const float* o) { Vec<f16>::store(reinterpret_cast<f16*>(p), o); }
  __device__ static inline bf16x8 pack8(const float* x) {
    f16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (_Float16)x[i];         // v_cvt_pk_f16_f32
    return *reinterpret_cast<bf16x8*>(&r);
  }
};
// Write-through stores WITHOUT the hand-placed pad: where the output tensor's base is wave-uniform (a kernel argument plus block-level offsets) the store can be
// a buffer store with the sc1 cache bit through hipcc's own builtin -- the compiler then schedules it and pads its hazards itself (the asm form above always
// pays `s_nop 1`: ~1 % of the iteration, profiles/r06_out_store_ab.txt).  Offsets are 32-bit BYTE offsets from the base: `ok` is false for tensors of 4 GiB or
// more, which fall back to the asm form.  DPB_OUT_BUF=0 builds use the asm form everywhere (A/B).
#ifndef DPB_OUT_BUF
#define DPB_OUT_BUF 1
#endif
struct OutBuf {
  __amdgpu_buffer_rsrc_t rs;
  const char* base;
  bool ok;
};
__device__ __forceinline__ OutBuf out_buf(void* base, long bytes) {
  OutBuf b;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)base), hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)base >> 32));
  b.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long)hi << 32) | lo), 0, -1, 0x00020000);     // raw buffer, 2^32 - 1 bytes: the caller's `ok` bounds it
  b.base = (const char*)base;
  b.ok = DPB_OUT_BUF != 0 && DPB_OUT_STORE == 1 && bytes > 0 && bytes < 4294967295L;
  return b;
}
__device__ __forceinline__ void store_out16_at(const OutBuf& ob, void* p, u32x4_ v) {
  if (ob.ok) __builtin_amdgcn_raw_buffer_store_b128(v, ob.rs, (unsigned)((const char*)p - ob.base), 0, 16 /* sc1 */);
  else store_out16<DPB_OUT_STORE>(p, v);
}
template <int FL>
__device__ __forceinline__ void store8_at(const OutBuf& ob, bf16* p, const float* o) {      // H16<FL>::store8 through the tensor's descriptor
  const bf16x8 r = H16<FL>::pack8(o);
  store_out16_at(ob, p, *reinterpret_cast<const u32x4_*>(&r));
}
template <typename T>
__device__ __forceinline__ void vec_store_at(const OutBuf& ob, T* p, const float* o) {       // Vec<T>::store through the tensor's descriptor (fp32: plain)
  if constexpr (std::is_same<T, float>::value) {
    Vec<float>::store(p, o);
  } else {
    const bf16x8 r = H16<std::is_same<T, f16>::value ? 1 : 0>::pack8(o);      // the conversion Vec<T>::store does
    store_out16_at(ob, p, *reinterpret_cast<const u32x4_*>(&r));
  }
}

template <int FL> __device__ inline float ld16(const bf16* p) { return H16<FL>::up(p->v); }
template <int FL> __device__ inline void st16(bf16* p, float v) { p->v = H16<FL>::dn(v); }

// run `call` with T = float | bf16 | f16 according to the engine dtype
#define DPB_DISPATCH_T(dtype, T, call) \
  ((dtype) == dpb::DT_F32 ? [&] { using T = float; return call; }() : (dtype) == dpb::DT_BF16 ? [&] { using T = dpb::bf16; return call; }() : [&] { using T = dpb::f16; return call; }())
// statement form (kernel launches): `...` is compiled with T = float | bf16 | f16
#define DPB_DISPATCH_STMT(dtype, T, ...)                                              \
  do {                                                                                \
    if ((dtype) == dpb::DT_F32) { using T = float; __VA_ARGS__; }                     \
    else if ((dtype) == dpb::DT_BF16) { using T = dpb::bf16; __VA_ARGS__; }           \
    else { using T = dpb::f16; __VA_ARGS__; }                                         \
  } while (0)
inline int dt_chunk(int dtype) { return dtype == DT_F32 ? 4 : 8; }   // elements per 16-byte chunk

// compile-time loop: f(std::integral_constant<int, I>) for I = 0..N-1 (keeps indices into register arrays static where `#pragma unroll`
// is only a hint: a dynamically indexed accumulator array is demoted to scratch memory)
template <int I, int N, typename F>
__device__ inline void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

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

// Shared device helpers for the gfx950 kernels (wave64, MFMA 16x16x32 f16/bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lip2speech_hip.h"

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

union frag16 {       // one MFMA A/B operand: 8 x 16-bit = 4 VGPRs
  uint4 u;
  f16x8_t h;
  bf16x8_t b;
  uint16_t s[8];
};

struct ElemF16 {
  static constexpr int kDtype = L2S_F16;
  static __device__ __forceinline__ float to_f32(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
  static __device__ __forceinline__ uint16_t from_f32(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {   // one v_cvt_pk_f16_f32, round to nearest even
    typedef float f2_t __attribute__((ext_vector_type(2)));
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f2_t){lo, hi}, h2_t));
  }
  static __device__ __forceinline__ f32x4_t mfma(const frag16& a, const frag16& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, b.h, c, 0, 0, 0);
  }
};

struct ElemBF16 {
  static constexpr int kDtype = L2S_BF16;
  static __device__ __forceinline__ float to_f32(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
  static __device__ __forceinline__ uint16_t from_f32(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return (uint32_t)from_f32(lo) | ((uint32_t)from_f32(hi) << 16); }
  static __device__ __forceinline__ f32x4_t mfma(const frag16& a, const frag16& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b, b.b, c, 0, 0, 0);
  }
};

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the 16-bit rounding of every consumer): one
// v_exp + one v_rcp + 6 FMAs instead of the ~45-instruction libm erff, which made GELU epilogues VALU-bound.
__device__ __forceinline__ float l2s_erf(float x) {
  const float a = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
  float pl = fmaf(1.061405429f, t, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  const float e = __expf(-a * a);
  const float r = fmaf(-pl * t, e, 1.0f);
  return copysignf(r, x);
}
__device__ __forceinline__ float l2s_gelu(float x) { return 0.5f * x * (1.0f + l2s_erf(x * 0.70710678118654752f)); }
// GELU for the hot 16-bit epilogues (FC1 of every encoder layer, 64 K outputs per tile): two values per call on the packed-fp32
// VALU forms (v_pk_mul / v_pk_fma), no transcendental.  erf(x / sqrt 2) = xc * Q(xc^2) with xc = clamp(x, +-4.75) and Q a
// degree-9 minimax polynomial (fitted on the GELU error 0.5 |x| |erf error|, overshooting by 3e-5 towards the clamp so that
// the clamp of the result to [-1, 1] leaves no tail error).  |error| <= 1.6e-5 for |x| <= 3 and <= 3.8e-5 overall (fp32
// Horner included) - below the fp16 / bf16 rounding step of every output above 0.06 in magnitude, against l2s_gelu's 4e-7;
// measured 9 VALU issue slots per value instead of ~22 (the erf form costs a v_rcp and a v_exp at quarter rate).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t l2s_gelu2(f32x2_t x) {
  f32x2_t xc;
  xc.x = __builtin_amdgcn_fmed3f(x.x, -4.75f, 4.75f);
  xc.y = __builtin_amdgcn_fmed3f(x.y, -4.75f, 4.75f);
  const f32x2_t u = xc * xc;
  f32x2_t q = u * -1.822768196e-12f + 2.421760752e-10f;
  q = q * u + -1.429140140e-08f;
  q = q * u + 4.968618750e-07f;
  q = q * u + -1.141144065e-05f;
  q = q * u + 1.845128930e-04f;
  q = q * u + -2.188001532e-03f;
  q = q * u + 1.950200040e-02f;
  q = q * u + -1.324454832e-01f;
  q = q * u + 7.976594608e-01f;
  f32x2_t er = xc * q;
  er.x = __builtin_amdgcn_fmed3f(er.x, -1.0f, 1.0f);
  er.y = __builtin_amdgcn_fmed3f(er.y, -1.0f, 1.0f);
  const f32x2_t hx = x * 0.5f;
  return hx * er + hx;
}
__device__ __forceinline__ float l2s_swish(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Opt a kernel instantiation in to > 64 KB of dynamic LDS, once per DEVICE: the attribute lives on the device's copy of the
// function, so a process that drives a second GPU has to set it there too.  `state` is the launcher's function-local static
// (one bit per device ordinal); concurrent host threads race benignly - at worst both set the attribute.
struct L2sSmemOptIn {
  unsigned long long done = 0;
};
template <typename KernT>
inline int l2s_smem_opt_in(KernT kern, int bytes, L2sSmemOptIn& state) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (__atomic_load_n(&state.done, __ATOMIC_ACQUIRE) & bit) return 0;
  e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  __atomic_fetch_or(&state.done, bit, __ATOMIC_RELEASE);
  return 0;
}

#define L2S_CHECK_LAUNCH()                      \
  do {                                          \
    hipError_t e__ = hipGetLastError();         \
    if (e__ != hipSuccess) return (int)e__;     \
  } while (0)

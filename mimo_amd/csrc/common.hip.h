// common.hip.h — shared device helpers for the gfx950 kernels of libmimo_hip.so.
// CDNA4 only: wave = 64 lanes, MFMA 16x16x32 / 32x32x16 (f16|bf16 in, f32 accumulate).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mimo_hip.h"

// cache-policy experiment (tools/_ab builds): aux bits of the streaming buffer loads / stores (2 = nt)
#ifndef MIMO_LD_AUX
#define MIMO_LD_AUX 0
#endif
#ifndef MIMO_ST_AUX
#define MIMO_ST_AUX 0
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 16-bit storage is carried as raw uint16_t / uint4 (8 halfs); HT<DT> gives it meaning.
template <int DT>
struct HT;

template <>
struct HT<MIMO_F16> {
  static __device__ __forceinline__ float to_f(uint16_t b) {
    return (float)__builtin_bit_cast(_Float16, b);
  }
  static __device__ __forceinline__ uint16_t from_f(float f) {
    return __builtin_bit_cast(uint16_t, (_Float16)f);
  }
  static __device__ __forceinline__ f32x4 mfma16(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mfma32(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

template <>
struct HT<MIMO_BF16> {
  static __device__ __forceinline__ float to_f(uint16_t b) {
    return __builtin_bit_cast(float, ((uint32_t)b) << 16);
  }
  static __device__ __forceinline__ uint16_t from_f(float f) {
    // round-to-nearest-even, NaN preserved
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
  }
  static __device__ __forceinline__ f32x4 mfma16(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mfma32(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};

// two floats -> one packed 16-bit pair, round-to-nearest-even, ONE instruction on gfx950
// (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32) instead of two converts + shift + or
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
template <int DT>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  if (DT == MIMO_F16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

template <int DT>
__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = HT<DT>::to_f((uint16_t)(w[i] & 0xffffu));
    f[2 * i + 1] = HT<DT>::to_f((uint16_t)(w[i] >> 16));
  }
}

template <int DT>
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  uint4 v;
  v.x = pack2<DT>(f[0], f[1]);
  v.y = pack2<DT>(f[2], f[3]);
  v.z = pack2<DT>(f[4], f[5]);
  v.w = pack2<DT>(f[6], f[7]);
  return v;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// Exact-form (erf) GELU, as torch F.gelu's default: x Phi(x) = max(x, 0) - |x| Q(|x|) with the Gaussian tail
// Q(u) = erfc(u / sqrt 2) / 2 = 2^P(u).  P = log2 Q is smooth; a degree-7 polynomial fitted for the absolute error of
// u Q(u) on [0, 6] (beyond 6 the tail is below 1e-9: u is clamped) leaves |gelu error| <= 2.9e-7 evaluated in fp32 — tighter
// than the Abramowitz-Stegun 7.1.26 erf it replaces (4.7e-7) — for ONE transcendental (v_exp) instead of two (v_rcp + v_exp)
// and 7 FMAs that pair up as v_pk_fma_f32.  The GEGLU epilogues are VALU-bound on exactly this.
constexpr float GELU_P[8] = {-0.9999997019767761f, -1.1511194705963135f, -0.4590896666049957f, -0.05285593867301941f,
                             0.007583224214613438f, -0.0005337183247320354f, -2.2708369215251878e-05f, 5.1834608711942565e-06f};
__device__ __forceinline__ f32x2_t gelu_erf_2(f32x2_t x) {
  const f32x2_t u = {__builtin_amdgcn_fmed3f(__builtin_fabsf(x.x), 0.f, 6.0f), __builtin_amdgcn_fmed3f(__builtin_fabsf(x.y), 0.f, 6.0f)};
  f32x2_t p = {GELU_P[7], GELU_P[7]};
#pragma unroll
  for (int k = 6; k >= 0; --k) p = __builtin_elementwise_fma(p, u, (f32x2_t){GELU_P[k], GELU_P[k]});
  const f32x2_t q = {__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)};
  const f32x2_t r = {__builtin_fmaxf(x.x, 0.f), __builtin_fmaxf(x.y, 0.f)};
  return __builtin_elementwise_fma(-u, q, r);
}
__device__ __forceinline__ f32x4 gelu_erf_4(f32x4 x) {
  const f32x2_t a = gelu_erf_2((f32x2_t){x[0], x[1]}), b = gelu_erf_2((f32x2_t){x[2], x[3]});
  return (f32x4){a.x, a.y, b.x, b.y};
}
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float u = __builtin_amdgcn_fmed3f(__builtin_fabsf(x), 0.f, 6.0f);
  float p = GELU_P[7];
#pragma unroll
  for (int k = 6; k >= 0; --k) p = fmaf(p, u, GELU_P[k]);
  return fmaf(-u, __builtin_amdgcn_exp2f(p), fmaxf(x, 0.f));
}

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

// XCD-aware bijective remap of a 1-D block id: blocks b, b+8, b+16 ... (which the dispatcher
// places on the same XCD) get consecutive logical ids, so tiles that share an operand panel
// share one L2 (cdna_hip_programming.md T1, bijective form).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u, slot = bid >> 3;
  const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

// Tuning knobs exist only in the -DMIMO_TUNE build (libmimo_hip_tune.so, tools/microbench.py): the shipped library
// reads no environment and has no mutable global state; every knob folds to its default at compile time.
#ifdef MIMO_TUNE
#include <stdlib.h>
inline int tune_env(const char* key, int dflt) {
  const char* v = getenv(key);
  return v ? atoi(v) : dflt;
}
#else
constexpr int tune_env(const char*, int dflt) { return dflt; }
#endif

#define MIMO_LAUNCH_CHECK()                    \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    if (e__ != hipSuccess) return (int)e__;    \
  } while (0)

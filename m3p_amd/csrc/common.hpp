// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the M3P hot path.
// wave = 64 lanes; bf16 is clang's native __bf16 (v_cvt_pk_bf16_f32 on gfx950, RNE).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define M3P_WAVE 64
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

#include "../../include/m3p_hip.h"   // error codes + the C ABI being implemented

#define M3P_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return (int)e__;                  \
  } while (0)

// ---------------------------------------------------------------------------
// counter-based dropout RNG: one 32-bit hash per element, layout independent, so the
// forward kernel, the backward kernel and the NumPy twin in tests/ (m3p_amd/rng.py)
// all regenerate the same keep mask from (seed, linear element index).
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t m3p_hash32(uint32_t idx, uint32_t seed) {
  uint32_t h = idx * 0x9E3779B1u + seed;
  h ^= h >> 16; h *= 0x21f0aaadu;
  h ^= h >> 15; h *= 0x735a2d97u;
  h ^= h >> 15;
  return h;
}
// keep iff top 24 bits >= thresh24, thresh24 = round(p * 2^24)  ->  P(keep) = 1 - p
__host__ __device__ __forceinline__ bool m3p_keep(uint32_t idx, uint32_t seed, uint32_t thresh24) {
  return (m3p_hash32(idx, seed) >> 8) >= thresh24;
}

// ---------------------------------------------------------------------------
// math
// ---------------------------------------------------------------------------
__device__ __forceinline__ float gelu_erf_f(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
// d/dx [0.5 x (1 + erf(x/sqrt2))] = 0.5 (1 + erf(x/sqrt2)) + x * exp(-x^2/2) / sqrt(2 pi)
__device__ __forceinline__ float gelu_erf_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
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

// activations may be bf16 or fp32 in HBM; all arithmetic is fp32 in registers
template <typename T> struct Vec4;
template <> struct Vec4<float> {
  static __device__ __forceinline__ f32x4 load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ void store(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Vec4<bf16> {
  static __device__ __forceinline__ f32x4 load(const bf16* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  }
  static __device__ __forceinline__ void store(bf16* p, f32x4 v) {
    *reinterpret_cast<bf16x4*>(p) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  }
};

__device__ __forceinline__ f32x4 round_bf16(f32x4 v) {
  return f32x4{(float)(bf16)v[0], (float)(bf16)v[1], (float)(bf16)v[2], (float)(bf16)v[3]};
}

// XCD-aware remap of a 1-D block id: each of the 8 XCDs (private L2) gets a contiguous
// run of logical ids; bijective for any grid size (cdna guide T1).
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7;
  const int xcd = bid & 7, pos = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + pos;
}

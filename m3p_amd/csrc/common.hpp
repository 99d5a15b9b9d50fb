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
// counter-based dropout RNG, layout independent, so the forward kernel, the backward kernel and the NumPy twin in tests/
// (m3p_amd/rng.py) all regenerate the same keep mask from (seed, linear element index).  ONE 32-bit hash serves TWO
// elements (round 4): element idx takes the low (idx even) or high (idx odd) 16 bits of hash32(idx >> 1) and is kept iff
// they are >= thresh16 = thresh24 >> 8 (p = 0.1: 6553 / 65536, i.e. P(drop) = 0.09999; the scale stays 1 / (1 - p)).
// Wherever four or eight consecutive elements sit in one lane (every dropout site) half as many hashes as elements suffice.
// The hash (round 6) is three rounds of { h ^= h >> 16 ; h += (h & 0xFFFFFF) * K } behind h = idx + seed, and a closing
// h ^= h >> 16: on gfx950 that is v_add, 3 x (an SDWA xor with src1_sel:WORD_1 - or shift + xor -, v_mad_u32_u24) and one more
// xor-shift - full-rate VALU instructions only.  Rounds 1-5 used a lowbias32-style finaliser behind idx * 0x9E3779B1 + seed:
// three v_mul_lo_u32 - quarter rate, four issue slots each - and seven other instructions, 19 slots per hash with the matrix
// pipe idle at every dropout site (attention probabilities, two GEMM epilogues per layer, LayerNorm backward, the embedding).
// The 24-bit multiply-add keeps the top byte in the addend, so no state is lost to the 24-bit operand.  Measured like the old
// one (tools/hash_eval.py: avalanche over the 28 index bits and the 32 seed bits within sampling noise - max |p - 1/2| 0.005
// over 200 k samples for both; keep rate 0.90002 at p = 0.1; serial correlation of the keep decisions <= 3e-4 at lags 1 .. S^2;
// cross-seed correlation 1.5e-3; chi-square of the output bytes 268 / 292 on 255 degrees of freedom).
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t m3p_hash32(uint32_t idx, uint32_t seed) {
  uint32_t h = idx + seed;
  h ^= h >> 16; h += (h & 0xFFFFFFu) * 0x9E3779u;
  h ^= h >> 16; h += (h & 0xFFFFFFu) * 0x85EBCBu;
  h ^= h >> 16; h += (h & 0xFFFFFFu) * 0xC2B2AFu;
  h ^= h >> 16;
  return h;
}
// thresh24 = round(p * 2^24) is what the C ABI carries; the decision uses its top 16 bits
__host__ __device__ __forceinline__ bool m3p_keep(uint32_t idx, uint32_t seed, uint32_t thresh24) {
  const uint32_t h = m3p_hash32(idx >> 1, seed);
  return ((idx & 1u) ? (h >> 16) : (h & 0xFFFFu)) >= (thresh24 >> 8);
}
// keep decisions of NE consecutive elements starting at an EVEN index: NE / 2 hashes
template <int NE>
__host__ __device__ __forceinline__ void m3p_keep_even(uint32_t base, uint32_t seed, uint32_t thresh24, bool (&k)[NE]) {
  static_assert((NE & 1) == 0, "pairs");
  const uint32_t t16 = thresh24 >> 8, pb = base >> 1;
#pragma unroll
  for (int q = 0; q < NE / 2; ++q) {
    const uint32_t h = m3p_hash32(pb + (uint32_t)q, seed);
    k[2 * q] = (h & 0xFFFFu) >= t16;
    k[2 * q + 1] = (h >> 16) >= t16;
  }
}
// ... starting anywhere: NE / 2 + 1 hashes (the odd start shifts the pairs by one element)
template <int NE>
__host__ __device__ __forceinline__ void m3p_keep_run(uint32_t base, uint32_t seed, uint32_t thresh24, bool (&k)[NE]) {
  bool w[NE + 2];
  m3p_keep_even<NE + 2>(base & ~1u, seed, thresh24, w);
  const bool odd = (base & 1u) != 0;
#pragma unroll
  for (int j = 0; j < NE; ++j) k[j] = odd ? w[j + 1] : w[j];
}

// ---------------------------------------------------------------------------
// math
// ---------------------------------------------------------------------------
// erf-based GELU (M3P/src/model/transformer.py:48-56) in ~16 VALU ops instead of libm's erff:
// Abramowitz-Stegun 7.1.26, |erf error| <= 1.5e-7 (two orders below bf16/fp32-accumulate
// noise of the surrounding GEMM).  exp(-z^2), z = x/sqrt(2), is shared with the Gaussian pdf
// of the derivative.  v_exp_f32 / v_rcp_f32 are the only transcendentals.
struct GeluParts { float cdf; float pdf; };   // Phi(x) and phi(x)
__device__ __forceinline__ GeluParts gelu_parts(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);   // v_rcp_f32 (1 ulp); __frcp_rn expands to a 10-op IEEE divide
  const float e = __expf(-z * z);   // = exp(-x^2/2)
  float poly = 1.061405429f;
  poly = poly * t - 1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t - 0.284496736f;
  poly = poly * t + 0.254829592f;
  const float erf_abs = 1.0f - poly * t * e;            // erf(|x|/sqrt2)
  const float cdf_abs = 0.5f + 0.5f * erf_abs;          // Phi(|x|)
  GeluParts r;
  r.cdf = (x >= 0.f) ? cdf_abs : 1.0f - cdf_abs;
  r.pdf = 0.39894228040143267794f * e;
  return r;
}
__device__ __forceinline__ float gelu_erf_f(float x) { return x * gelu_parts(x).cdf; }
// d/dx [x Phi(x)] = Phi(x) + x phi(x)
__device__ __forceinline__ float gelu_erf_grad_f(float x) {
  const GeluParts g = gelu_parts(x);
  return g.cdf + x * g.pdf;
}

// gelu_erf'(u) in ONE byte for the FFN data gradient (M3P_EPI_MULQ): gelu' lies in [-0.1290, 1.1290], the code is the
// nearest level of the grid g = (code - 27) / 201 over [-0.1343, 1.1343] (step 4.98e-3: |error| <= 2.5e-3, rms 1.4e-3 - the size
// of the bf16 rounding of the gradient it multiplies), decoded with one fma.  The grid CONTAINS 0 and 1 (codes 27 and 228; round 5,
// ADVICE r4): dead and saturated units - most of an FFN's activations - decode to 0 and 1 within one fp32 rounding of the fma
// instead of carrying a systematic -0.0015 / +0.0015 bias.  It is symmetric about 1/2 like gelu' itself, and since round 6 the
// centre 1/2 sits BETWEEN two levels (127 | 128):  gelu'(-u) = 1 - gelu'(u)  <->  code(-u) = 255 - code(u) = code(u) ^ 0xFF,
// so an epilogue that looks the code of |u| up in a table fixes the sign with one XOR on four packed codes (gemm.hip:
// epilogue_piece_geluq_lut).
constexpr float GQ_STEP = 1.0f / 201.0f, GQ_OFF = 27.0f / 201.0f, GQ_INV = 201.0f;
__device__ __forceinline__ uint32_t gelu_grad_code(float g) {
  return (uint32_t)__builtin_rintf(fminf(fmaxf((g + GQ_OFF) * GQ_INV, 0.f), 255.f));
}
// Fragment order of the code bytes of an [M, N] activation (M, N multiples of 256): the bytes of a 256 x 256 tile follow the
// eight-wave GEMM's accumulator layout, so that its epilogue fetches a 16-row block of a wave's share with ONE coalesced
// 16-byte load per lane and no trip through LDS:   tile (tm, tn) -> wave w = 4 (row half) + (64-column quarter)
//   -> row block i (16 rows) -> lane l = 16 fg + fr -> 16 bytes (j, r): element (row 128 wm + 16 i + fr, col 64 wn + 16 j + 4 fg + r)
__device__ __forceinline__ size_t gq_block_offset(int tm, int tn, int tiles_n, int w, int i) {
  return ((((size_t)tm * tiles_n + tn) * 8 + w) * 8 + i) * 1024;
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

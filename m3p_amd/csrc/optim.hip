// Optimizer path of Trainer.optimize (M3P/src/xtrainer.py:205-243) on flat fp32 arenas:
//   * global gradient L2 norm (clip_grad_norm_, xtrainer.py:225) as one streaming reduction
//   * Adam.step (M3P/src/optim.py:45-86) fused with the clip scaling, the refresh of the
//     bf16 working copy the GEMMs read, and zero_grad — one pass over p, g, m, v.
//   * bf16 transposes of the weight copies used by the data-gradient GEMMs.
// All HBM-bound: 16-byte accesses, grid-stride.
#include "common.hpp"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, size_t n4, double* __restrict__ out) {
  __shared__ double red[4];
  float acc = 0.f;
  // four 16-byte loads in flight per thread: the bucket-sized launches of the sharded data-parallel step (15 of ~75 MB) ran
  // at 2 TB/s with one
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + 4 * (i + u * stride)));
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += (v[u][0] * v[u][0] + v[u][1] * v[u][1]) + (v[u][2] * v[u][2] + v[u][3] * v[u][3]);
  }
  for (; i < n4; i += stride) {
    const f32x4 v = Vec4<float>::load(g + 4 * i);
    acc += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = (double)acc;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}

// the same over up to SUMSQ_MAX_RANGES ranges of one buffer in ONE launch (the sharded data-parallel step owns a piece of every
// bucket: fifteen launches of ~30 us for what one launch does in the time the bytes take).  Blocks are dealt to the ranges
// in proportion to their sizes (blk0[r] = first block of range r).
constexpr int SUMSQ_MAX_RANGES = 32;
struct SumsqRanges {
  const float* g[SUMSQ_MAX_RANGES];
  unsigned long long n4[SUMSQ_MAX_RANGES];
  int blk0[SUMSQ_MAX_RANGES + 1];
  int n;
};
__global__ __launch_bounds__(256) void sumsq_ranges_kernel(SumsqRanges a, double* __restrict__ out) {
  __shared__ double red[4];
  int r = 0;
  while (r + 1 < a.n && (int)blockIdx.x >= a.blk0[r + 1]) ++r;
  const float* __restrict__ g = a.g[r];
  const size_t n4 = a.n4[r];
  const size_t stride = (size_t)(a.blk0[r + 1] - a.blk0[r]) * blockDim.x;
  size_t i = (size_t)((int)blockIdx.x - a.blk0[r]) * blockDim.x + threadIdx.x;
  float acc = 0.f;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + 4 * (i + u * stride)));
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += (v[u][0] * v[u][0] + v[u][1] * v[u][1]) + (v[u][2] * v[u][2] + v[u][3] * v[u][3]);
  }
  for (; i < n4; i += stride) {
    const f32x4 v = Vec4<float>::load(g + 4 * i);
    acc += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = (double)acc;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}

#ifndef M3P_ADAM_STREAM
#define M3P_ADAM_STREAM 1
#endif
struct AdamArgs {
  float* p; float* g; float* m; float* v; bf16* w16;
  size_t n4;
  float lr, beta1, beta2, eps, weight_decay, step_size;
  const double* gnorm_sq;   // device scalar: sum of squares of ALL gradients (or NULL)
  float max_norm;           // <= 0: no clipping
  float grad_scale;         // extra multiplier on g (1/world for DP averaging, loss-scale inverse)
  int zero_grad;
};

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  float coef = a.grad_scale;
  if (a.gnorm_sq && a.max_norm > 0.f) {
    // torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1
    const float norm = (float)sqrt(*a.gnorm_sq) * a.grad_scale;
    const float c = a.max_norm / (norm + 1e-6f);
    coef *= (c < 1.f) ? c : 1.f;
  }
  const float ob1 = 1.f - a.beta1, ob2 = 1.f - a.beta2, wdl = a.weight_decay * a.lr;
#if M3P_ADAM_STREAM
  // Nine streams (4 read, 5 written) of data touched once per step: non-temporal accesses keep them out of the way
  // of the bf16 weights / activations the next forward re-reads, and two 16-byte quads per thread and iteration put
  // eight loads in flight before the first use.
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < a.n4; i0 += 2 * stride) {
    const size_t i1 = i0 + stride;
    const bool two = i1 < a.n4;
    f32x4 p[2], g[2], m[2], v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const size_t i = u ? i1 : i0;
      if (u == 0 || two) {
        p[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.p + 4 * i));
        g[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.g + 4 * i));
        m[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.m + 4 * i));
        v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.v + 4 * i));
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const size_t i = u ? i1 : i0;
      if (u == 0 || two) {
        const f32x4 gc = g[u] * coef;
        const f32x4 mn = m[u] * a.beta1 + gc * ob1;
        const f32x4 vn = v[u] * a.beta2 + gc * gc * ob2;
        f32x4 pn = p[u];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float denom = sqrtf(vn[j]) + a.eps;
          if (wdl != 0.f) pn[j] -= wdl * pn[j];
          pn[j] -= a.step_size * (mn[j] / denom);
        }
        __builtin_nontemporal_store(pn, reinterpret_cast<f32x4*>(a.p + 4 * i));
        __builtin_nontemporal_store(mn, reinterpret_cast<f32x4*>(a.m + 4 * i));
        __builtin_nontemporal_store(vn, reinterpret_cast<f32x4*>(a.v + 4 * i));
        if (a.w16) Vec4<bf16>::store(a.w16 + 4 * i, pn);       // (re-read by the very next forward: default policy)
        if (a.zero_grad) __builtin_nontemporal_store(f32x4{0.f, 0.f, 0.f, 0.f}, reinterpret_cast<f32x4*>(a.g + 4 * i));
      }
    }
  }
#else
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 p = Vec4<float>::load(a.p + 4 * i);
    const f32x4 g = Vec4<float>::load(a.g + 4 * i) * coef;
    f32x4 m = Vec4<float>::load(a.m + 4 * i) * a.beta1 + g * ob1;
    f32x4 v = Vec4<float>::load(a.v + 4 * i) * a.beta2 + g * g * ob2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float denom = sqrtf(v[j]) + a.eps;
      if (wdl != 0.f) p[j] -= wdl * p[j];
      p[j] -= a.step_size * (m[j] / denom);
    }
    Vec4<float>::store(a.p + 4 * i, p);
    Vec4<float>::store(a.m + 4 * i, m);
    Vec4<float>::store(a.v + 4 * i, v);
    if (a.w16) Vec4<bf16>::store(a.w16 + 4 * i, p);
    if (a.zero_grad) Vec4<float>::store(a.g + 4 * i, f32x4{0.f, 0.f, 0.f, 0.f});
  }
#endif
}

// bf16 -> fp8 (e4m3) / bf8 (e5m2) with a per-tensor scale, saturating, + running max |x| for the next step's scale
// (delayed scaling, m3p_amd/fp8.py).  8 elements (16 B in, 8 B out) per thread and iteration.
template <bool BF8>
__global__ __launch_bounds__(256) void quant_fp8_kernel(const bf16* __restrict__ src, int ld_src, uint8_t* __restrict__ dst, int ld_dst,
                                                        int rows, int cols8, const float* __restrict__ scale, float* __restrict__ amax) {
  const float s = scale ? *scale : 1.f;
  constexpr float kMax = BF8 ? 57344.f : 448.f;
  float mx = 0.f;
  const size_t total = (size_t)rows * cols8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols8), c = (int)(i - (size_t)r * cols8) * 8;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + (size_t)r * ld_src + c);
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = (float)v[e];
      mx = fmaxf(mx, fabsf(x));
      f[e] = fminf(fmaxf(x * s, -kMax), kMax);
    }
    int w0 = 0, w1 = 0;
    if (BF8) {
      w0 = __builtin_amdgcn_cvt_pk_bf8_f32(f[0], f[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_bf8_f32(f[2], f[3], w0, true);
      w1 = __builtin_amdgcn_cvt_pk_bf8_f32(f[4], f[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_bf8_f32(f[6], f[7], w1, true);
    } else {
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
    }
    *reinterpret_cast<int2*>(dst + (size_t)r * ld_dst + c) = int2{w0, w1};
  }
  if (amax) {
    // One candidate per workgroup, and only if it beats the value already there: thousands of same-address atomics
    // from every wave cost ~200 us per launch (they serialise at the memory side), the filtered ones a few.
    __shared__ float wmax[4];
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
      unsigned int* slot = reinterpret_cast<unsigned int*>(amax);
      // non-negative floats order like their bit patterns: an integer atomic max is a float max.  The plain read may be
      // stale (another XCD's L2) - then it is too small and the atomic merely runs without effect.
      if (__float_as_uint(mx) > __builtin_nontemporal_load(slot)) atomicMax(slot, __float_as_uint(mx));
    }
  }
}

// Batched form for the weights (fp8 path: ~120 matrices per optimizer step, 8-10 us of launch each on their own): one
// grid.y slice per matrix.  desc[i] = {src, dst, scale ptr, amax ptr or 0, elements / 8} (int64 each); e4m3 only.
__global__ __launch_bounds__(256) void quant_fp8_batch_kernel(const long long* __restrict__ desc) {
  const long long* dsc = desc + 5 * blockIdx.y;
  const bf16* src = reinterpret_cast<const bf16*>(dsc[0]);
  uint8_t* dst = reinterpret_cast<uint8_t*>(dsc[1]);
  const float s = *reinterpret_cast<const float*>(dsc[2]);
  float* amax = reinterpret_cast<float*>(dsc[3]);
  const size_t total = (size_t)dsc[4];
  float mx = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + 8 * i);
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = (float)v[e];
      mx = fmaxf(mx, fabsf(x));
      f[e] = fminf(fmaxf(x * s, -448.f), 448.f);
    }
    int w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
    *reinterpret_cast<int2*>(dst + 8 * i) = int2{w0, w1};
  }
  if (amax) {
    __shared__ float wmax[4];
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
      unsigned int* slot = reinterpret_cast<unsigned int*>(amax);
      if (__float_as_uint(mx) > __builtin_nontemporal_load(slot)) atomicMax(slot, __float_as_uint(mx));
    }
  }
}

// dst[c][r] = src[r][c], 64x64 tiles through LDS; ld_dst >= rows (pad columns untouched)
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst,
                                                             int rows, int cols, int ld_src, int ld_dst) {
  __shared__ bf16 tile[64][66];
  const int tr = blockIdx.y * 64, tc = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = tr + i, c = tc + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[(size_t)r * ld_src + c] : (bf16)0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = tc + i, r = tr + tx;
    if (c < cols && r < rows) dst[(size_t)c * ld_dst + r] = tile[tx][i];
  }
}

// batched variant: desc[i] = {src, dst, rows, cols, ld_src, ld_dst} (int64 each), one grid.y slice per matrix
__global__ __launch_bounds__(256) void transpose_batch_kernel(const long long* __restrict__ desc) {
  __shared__ bf16 tile[64][72];       // 144-byte rows: 16-byte aligned for the row-wise stores, 4-bank skew for the column reads
  const long long* dsc = desc + 6 * blockIdx.y;
  const bf16* src = reinterpret_cast<const bf16*>(dsc[0]);
  bf16* dst = reinterpret_cast<bf16*>(dsc[1]);
  const int rows = (int)dsc[2], cols = (int)dsc[3], ld_src = (int)dsc[4], ld_dst = (int)dsc[5];
  const int tiles_c = (cols + 63) / 64, tiles_r = (rows + 63) / 64;
  if ((int)blockIdx.x >= tiles_c * tiles_r) return;
  const int tr = (blockIdx.x / tiles_c) * 64, tc = (blockIdx.x % tiles_c) * 64;
  // whole 64 x 64 tiles of 16-byte-aligned matrices (every layer weight): 16-byte accesses on both sides - a lane reads 8
  // consecutive columns of a row and later writes 8 consecutive rows of a column (2-byte accesses moved 2.2 TB/s: r02)
  const bool fast = tr + 64 <= rows && tc + 64 <= cols && (ld_src & 7) == 0 && (ld_dst & 7) == 0 &&
                    (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
  if (fast) {
    const int q = threadIdx.x & 7, rr = threadIdx.x >> 3;          // 8 lanes per 128-byte row piece, 32 rows per pass
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int r = rr + 32 * p;
      *reinterpret_cast<bf16x8*>(&tile[r][8 * q]) = *reinterpret_cast<const bf16x8*>(src + (size_t)(tr + r) * ld_src + tc + 8 * q);
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int c = rr + 32 * p;                                   // output row = source column
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = tile[8 * q + e][c];
      *reinterpret_cast<bf16x8*>(dst + (size_t)(tc + c) * ld_dst + tr + 8 * q) = o;
    }
    return;
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = tr + i, c = tc + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[(size_t)r * ld_src + c] : (bf16)0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = tc + i, r = tr + tx;
    if (c < cols && r < rows) dst[(size_t)c * ld_dst + r] = tile[tx][i];
  }
}

// h = gelu_erf(u), bf16 in/out, 16-byte accesses (the FFN activation as its own streaming
// pass: 129 M elements at HBM speed cost less than the same VALU work serialised behind the
// MFMA stream of the persistent GEMM — see DESIGN.md)
template <bool GRAD>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16* u, bf16* __restrict__ h, bf16* dh, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(u + 8 * i);
    bf16x8 o, d;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = (float)v[j];
      const GeluParts g = gelu_parts(x);        // Phi(x), phi(x) share one exp
      o[j] = (bf16)(x * g.cdf);
      if (GRAD) d[j] = (bf16)(g.cdf + x * g.pdf);
    }
    *reinterpret_cast<bf16x8*>(h + 8 * i) = o;
    if (GRAD) *reinterpret_cast<bf16x8*>(dh + 8 * i) = d;     // may overwrite u: each thread has read its 8 values
  }
}

// The same pass leaving gelu'(u) as one byte per element in the eight-wave GEMM's fragment order (common.hpp): a workgroup
// takes 16 rows x 256 columns = one row block of the four column quarters of a tile (4 KB of codes).  Phase 1 walks the
// rows in 16-byte pieces (coalesced u reads and h writes) and drops the codes into LDS in row order; phase 2 reads them in
// accumulator order (lane = 16 fg + fr holds columns 16 j + 4 fg .. + 3 of row fr) and writes 1 KB per wave, coalesced.
__global__ __launch_bounds__(256) void gelu_fwd_gq_kernel(const bf16* __restrict__ u, bf16* __restrict__ h, uint8_t* __restrict__ gq,
                                                          int N, int tiles_n, int nblocks) {
  __shared__ __attribute__((aligned(16))) uint8_t codes[16][256 + 16];
  const int t = threadIdx.x;
  for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const int rb = b / tiles_n, tn = b - rb * tiles_n;        // 16-row block, column tile
    const size_t row0 = (size_t)rb * 16;
    const int col0 = tn * 256;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int r = (t >> 5) + 8 * half, c = (t & 31) * 8;
      const size_t off = (row0 + r) * (size_t)N + col0 + c;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(u + off);
      bf16x8 o;
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = (float)v[j];
        const GeluParts g = gelu_parts(x);
        o[j] = (bf16)(x * g.cdf);
        const uint32_t q = gelu_grad_code(g.cdf + x * g.pdf);
        if (j < 4) lo |= q << (8 * j); else hi |= q << (8 * (j - 4));
      }
      *reinterpret_cast<bf16x8*>(h + off) = o;
      *reinterpret_cast<uint2*>(&codes[r][c]) = uint2{lo, hi};
    }
    __syncthreads();
    {
      const int wn = t >> 6, l = t & 63, fr = l & 15, fg = l >> 4;
      uint4 q;
      q.x = *reinterpret_cast<const uint32_t*>(&codes[fr][wn * 64 + 0 + fg * 4]);
      q.y = *reinterpret_cast<const uint32_t*>(&codes[fr][wn * 64 + 16 + fg * 4]);
      q.z = *reinterpret_cast<const uint32_t*>(&codes[fr][wn * 64 + 32 + fg * 4]);
      q.w = *reinterpret_cast<const uint32_t*>(&codes[fr][wn * 64 + 48 + fg * 4]);
      const int tm = rb >> 4, ib = rb & 15, wm = ib >> 3, i = ib & 7;
      *reinterpret_cast<uint4*>(gq + gq_block_offset(tm, tn, tiles_n, wm * 4 + wn, i) + l * 16) = q;
    }
    __syncthreads();
  }
}

// The same pass also emitting the e4m3 copy of h the fp8 lin2 product reads (fp8 path: a separate quantisation pass would
// read the 280-MB activation again) + the running max |h| for the next step's scale.  The 8-bit value is the
// quantisation of the ROUNDED bf16 h, i.e. exactly what m3p_quant_fp8 would produce from h.
__global__ __launch_bounds__(256) void gelu_fwd_q8_kernel(const bf16* __restrict__ u, bf16* __restrict__ h, uint8_t* __restrict__ h8,
                                                          size_t n8, const float* __restrict__ scale, float* __restrict__ amax) {
  const float s = *scale;
  float mx = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(u + 8 * i);
    bf16x8 o;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = (float)v[j];
      const GeluParts g = gelu_parts(x);
      o[j] = (bf16)(x * g.cdf);
      const float hr = (float)o[j];
      mx = fmaxf(mx, fabsf(hr));
      f[j] = fminf(fmaxf(hr * s, -448.f), 448.f);
    }
    *reinterpret_cast<bf16x8*>(h + 8 * i) = o;
    int w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
    *reinterpret_cast<int2*>(h8 + 8 * i) = int2{w0, w1};
  }
  if (amax) {
    __shared__ float wmax[4];
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
      unsigned int* slot = reinterpret_cast<unsigned int*>(amax);
      if (__float_as_uint(mx) > __builtin_nontemporal_load(slot)) atomicMax(slot, __float_as_uint(mx));
    }
  }
}

// du = dy * gelu_erf'(u): backward of a GELU that sits between a Linear and a LayerNorm
// (BertPredictionHeadTransform, transformer.py:595-606), bf16 in/out, 16-byte accesses
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ u,
                                                       bf16* __restrict__ du, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(dy + 8 * i);
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(u + 8 * i);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)g[j] * gelu_erf_grad_f((float)v[j]));
    *reinterpret_cast<bf16x8*>(du + 8 * i) = o;
  }
}

// Masked-region feature regression loss (xtrainer.py:2346): one wave per row,
// row_sq[i] = sum_j (pred[i][j] - tgt[i][j])^2 and dpred = 2 (pred - tgt) * gscale in bf16.
__global__ __launch_bounds__(256) void mse_fwd_bwd_kernel(const bf16* __restrict__ pred, int ld_pred, const float* __restrict__ tgt,
                                                          int ld_tgt, bf16* __restrict__ dpred, float* __restrict__ row_sq,
                                                          int rows, int cols, float gscale) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= rows) return;
  const bf16* pr = pred + (size_t)row * ld_pred;
  const float* tr = tgt + (size_t)row * ld_tgt;
  bf16* dr = dpred + (size_t)row * ld_pred;
  float acc = 0.f;
  for (int c = lane * 4; c < cols; c += 256) {
    const f32x4 p = Vec4<bf16>::load(pr + c);
    const f32x4 t = *reinterpret_cast<const f32x4*>(tr + c);
    const f32x4 e = p - t;
    acc += (e[0] * e[0] + e[1] * e[1]) + (e[2] * e[2] + e[3] * e[3]);
    Vec4<bf16>::store(dr + c, e * (2.f * gscale));
  }
  acc = wave_sum(acc);
  if (lane == 0) row_sq[row] = acc;
}

}  // namespace

extern "C" {

int m3p_gelu_bwd(const void* dy, const void* u, void* du, long long n, void* stream) {
  if (n <= 0 || (n % 8) != 0 || ((uintptr_t)dy & 15) || ((uintptr_t)u & 15) || ((uintptr_t)du & 15)) return M3P_EINVAL;
  const size_t n8 = (size_t)n / 8;
  const int blocks = (int)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)dy, (const bf16*)u, (bf16*)du, n8);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_mse_fwd_bwd(const void* pred, int ld_pred, const float* tgt, int ld_tgt, void* dpred, float* row_sq, int rows,
                    int cols, float grad_scale, void* stream) {
  if (rows <= 0 || cols <= 0 || (cols % 4) != 0 || (ld_pred % 4) != 0 || (ld_tgt % 4) != 0) return M3P_EINVAL;
  if (((uintptr_t)pred & 7) || ((uintptr_t)dpred & 7) || ((uintptr_t)tgt & 15)) return M3P_EINVAL;
  hipLaunchKernelGGL(mse_fwd_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)pred, ld_pred, tgt,
                     ld_tgt, (bf16*)dpred, row_sq, rows, cols, grad_scale);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_gelu_fwd(const void* u, void* h, void* dh, long long n, void* stream) {
  if (n <= 0 || (n % 8) != 0 || ((uintptr_t)u & 15) || ((uintptr_t)h & 15) || ((uintptr_t)dh & 15)) return M3P_EINVAL;
  const size_t n8 = (size_t)n / 8;
  const int blocks = (int)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);
  if (dh)
    hipLaunchKernelGGL(gelu_fwd_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)u, (bf16*)h, (bf16*)dh, n8);
  else
    hipLaunchKernelGGL(gelu_fwd_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)u, (bf16*)h, (bf16*)nullptr, n8);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_gelu_fwd_gq(const void* u, void* h, void* gq, int M, int N, void* stream) {
  if (M <= 0 || N <= 0 || (M % 256) || (N % 256) || !u || !h || !gq) return M3P_EINVAL;
  if (((uintptr_t)u & 15) || ((uintptr_t)h & 15) || ((uintptr_t)gq & 15)) return M3P_EINVAL;
  const long long nb = (long long)(M / 16) * (N / 256);
  if (nb > 0x7fffffffLL) return M3P_EINVAL;
  const int blocks = (int)(nb < 16384 ? nb : 16384);
  hipLaunchKernelGGL(gelu_fwd_gq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)u, (bf16*)h, (uint8_t*)gq, N,
                     N / 256, (int)nb);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_gelu_fwd_q8(const void* u, void* h, void* h8, long long n, const float* scale, float* amax, void* stream) {
  if (n <= 0 || (n % 8) != 0 || ((uintptr_t)u & 15) || ((uintptr_t)h & 15) || ((uintptr_t)h8 & 7) || !scale) return M3P_EINVAL;
  const size_t n8 = (size_t)n / 8;
  const int blocks = (int)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);
  hipLaunchKernelGGL(gelu_fwd_q8_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)u, (bf16*)h, (uint8_t*)h8, n8,
                     scale, amax);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_transpose_batch_bf16(const long long* desc, int n_desc, int max_tiles, void* stream) {
  if (n_desc <= 0 || max_tiles <= 0) return M3P_EINVAL;
  hipLaunchKernelGGL(transpose_batch_kernel, dim3(max_tiles, n_desc), dim3(256), 0, (hipStream_t)stream, desc);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}


int m3p_quant_fp8(const void* src, int ld_src, void* dst, int ld_dst, int rows, int cols, const float* scale, float* amax,
                  int bf8, void* stream) {
  if (rows <= 0 || cols <= 0 || (cols % 8) != 0 || (ld_src % 8) != 0 || (ld_dst % 8) != 0 || ld_src < cols || ld_dst < cols)
    return M3P_EINVAL;
  if (((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return M3P_EINVAL;
  const size_t total = (size_t)rows * (cols / 8);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (bf8)
    hipLaunchKernelGGL(quant_fp8_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, ld_src,
                       (uint8_t*)dst, ld_dst, rows, cols / 8, scale, amax);
  else
    hipLaunchKernelGGL(quant_fp8_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, ld_src,
                       (uint8_t*)dst, ld_dst, rows, cols / 8, scale, amax);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_quant_fp8_batch(const long long* desc, int n_desc, int blocks_per_matrix, void* stream) {
  if (!desc || n_desc <= 0 || blocks_per_matrix <= 0 || blocks_per_matrix > 4096) return M3P_EINVAL;
  hipLaunchKernelGGL(quant_fp8_batch_kernel, dim3(blocks_per_matrix, n_desc), dim3(256), 0, (hipStream_t)stream, desc);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_sumsq_f32(const float* g, long long n, double* out, void* stream) {
  if (n <= 0 || (n % 4) != 0 || ((uintptr_t)g & 15)) return M3P_EINVAL;
  const size_t n4 = (size_t)n / 4;
  const size_t want = (n4 + 1023) / 1024;      // ~four quads per thread
  const int blocks = (int)(want < 1 ? 1 : (want < 2048 ? want : 2048));
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, n4, out);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_sumsq_ranges_f32(const float* base, const long long* starts, const long long* counts, int n_ranges, double* out, void* stream) {
  if (!base || !starts || !counts || !out || n_ranges <= 0) return M3P_EINVAL;
  int done = 0;
  while (done < n_ranges) {            // (more ranges than one launch's descriptor holds: several launches)
    SumsqRanges a;
    a.n = 0;
    long long total4 = 0;
    for (int r = done; r < n_ranges && a.n < SUMSQ_MAX_RANGES; ++r) {
      if (counts[r] < 0 || (counts[r] % 4) != 0 || starts[r] < 0 || (((uintptr_t)(base + starts[r])) & 15)) return M3P_EINVAL;
      if (counts[r] == 0) continue;
      a.g[a.n] = base + starts[r];
      a.n4[a.n] = (unsigned long long)counts[r] / 4;
      total4 += counts[r] / 4;
      ++a.n;
    }
    int taken = 0;
    for (int r = done, k = 0; r < n_ranges && k < SUMSQ_MAX_RANGES; ++r) { if (counts[r] != 0) ++k; ++taken; }
    done += taken;
    if (a.n == 0) continue;
    // ~four quads per thread, at most 2048 blocks, at least one block per range
    long long want = (total4 + 1023) / 1024;
    if (want > 2048) want = 2048;
    if (want < a.n) want = a.n;
    int b = 0;
    for (int r = 0; r < a.n; ++r) {
      a.blk0[r] = b;
      long long share = (long long)((double)a.n4[r] / (double)total4 * (double)want);
      if (share < 1) share = 1;
      b += (int)share;
    }
    a.blk0[a.n] = b;
    hipLaunchKernelGGL(sumsq_ranges_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, a, out);
    M3P_CHECK_LAUNCH();
  }
  return M3P_OK;
}

// The same update over several pieces of the flat arenas in ONE launch (round 5): blocks are dealt to the pieces in proportion
// to their sizes like m3p_sumsq_ranges_f32.  A sharded data-parallel rank steps fifteen bucket shards per optimizer step, the
// single-GPU step two pieces (the lazily zeroed vocabulary range and the rest): per piece a launch ramps up and drains on its
// own - fifteen launches streamed the one-rank wrapped step's 9.5 GB in 2.00 ms where two take 1.61 (profiles/r05_dp_prof.txt).
constexpr int ADAM_MAX_RANGES = 32;
struct AdamRanges {
  AdamArgs c;                                  // common fields; p / g / m / v / w16 = the arenas' bases, n4 / step_size / zero_grad unused
  unsigned long long start4[ADAM_MAX_RANGES];  // first quad of the piece
  unsigned long long n4[ADAM_MAX_RANGES];
  float step_size[ADAM_MAX_RANGES];
  int zero_grad[ADAM_MAX_RANGES];
  int blk0[ADAM_MAX_RANGES + 1];
  int n;
};
#ifndef M3P_ADAM_Q
#define M3P_ADAM_Q 2          // 16-byte quads per thread and trip (x 4 read streams = loads in flight before the first use)
#endif
#ifndef M3P_ADAM_MAXBLK
#define M3P_ADAM_MAXBLK 4096
#endif
__global__ __launch_bounds__(256) void adam_ranges_kernel(AdamRanges a) {
  int r = 0;
  while (r + 1 < a.n && (int)blockIdx.x >= a.blk0[r + 1]) ++r;
  float coef = a.c.grad_scale;
  if (a.c.gnorm_sq && a.c.max_norm > 0.f) {
    const float norm = (float)sqrt(*a.c.gnorm_sq) * a.c.grad_scale;
    const float cc = a.c.max_norm / (norm + 1e-6f);
    coef *= (cc < 1.f) ? cc : 1.f;
  }
  const float ob1 = 1.f - a.c.beta1, ob2 = 1.f - a.c.beta2, wdl = a.c.weight_decay * a.c.lr;
  const float step_size = a.step_size[r];
  const bool zero = a.zero_grad[r] != 0;
  const size_t base = (size_t)a.start4[r], n4 = (size_t)a.n4[r];
  float* const P = a.c.p + 4 * base; float* const G = a.c.g + 4 * base; float* const M = a.c.m + 4 * base; float* const V = a.c.v + 4 * base;
  bf16* const W = a.c.w16 ? a.c.w16 + 4 * base : nullptr;
  const size_t stride = (size_t)(a.blk0[r + 1] - a.blk0[r]) * blockDim.x;
  constexpr int Q = M3P_ADAM_Q;
  for (size_t i0 = (size_t)((int)blockIdx.x - a.blk0[r]) * blockDim.x + threadIdx.x; i0 < n4; i0 += Q * stride) {
    f32x4 p[Q], g[Q], m[Q], v[Q];
#pragma unroll
    for (int u = 0; u < Q; ++u) {
      const size_t i = i0 + u * stride;
      if (u == 0 || i < n4) {
        p[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P + 4 * i));
        g[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(G + 4 * i));
        m[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(M + 4 * i));
        v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(V + 4 * i));
      }
    }
#pragma unroll
    for (int u = 0; u < Q; ++u) {
      const size_t i = i0 + u * stride;
      if (u == 0 || i < n4) {
        const f32x4 gc = g[u] * coef;
        const f32x4 mn = m[u] * a.c.beta1 + gc * ob1;
        const f32x4 vn = v[u] * a.c.beta2 + gc * gc * ob2;
        f32x4 pn = p[u];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float denom = sqrtf(vn[j]) + a.c.eps;
          if (wdl != 0.f) pn[j] -= wdl * pn[j];
          pn[j] -= step_size * (mn[j] / denom);
        }
        __builtin_nontemporal_store(pn, reinterpret_cast<f32x4*>(P + 4 * i));
        __builtin_nontemporal_store(mn, reinterpret_cast<f32x4*>(M + 4 * i));
        __builtin_nontemporal_store(vn, reinterpret_cast<f32x4*>(V + 4 * i));
        if (W) Vec4<bf16>::store(W + 4 * i, pn);
        if (zero) __builtin_nontemporal_store(f32x4{0.f, 0.f, 0.f, 0.f}, reinterpret_cast<f32x4*>(G + 4 * i));
      }
    }
  }
}

int m3p_adam_step_ranges(float* p, float* g, float* m, float* v, void* w16, const long long* starts, const long long* counts,
                         const float* step_sizes, const int* zero_grad, int n_ranges, float lr, float beta1, float beta2, float eps,
                         float weight_decay, const double* gnorm_sq, float max_norm, float grad_scale, void* stream) {
  if (!p || !g || !m || !v || !starts || !counts || !step_sizes || !zero_grad || n_ranges <= 0) return M3P_EINVAL;
  if (((uintptr_t)p & 15) || ((uintptr_t)g & 15) || ((uintptr_t)m & 15) || ((uintptr_t)v & 15) || ((uintptr_t)w16 & 7)) return M3P_EINVAL;
  for (int r = 0; r < n_ranges; ++r)
    if (counts[r] < 0 || (counts[r] % 4) != 0 || starts[r] < 0 || (starts[r] % 4) != 0) return M3P_EINVAL;
  int done = 0;
  while (done < n_ranges) {
    AdamRanges a;
    a.c = AdamArgs{p, g, m, v, (bf16*)w16, 0, lr, beta1, beta2, eps, weight_decay, 0.f, gnorm_sq, max_norm,
                   grad_scale == 0.f ? 1.f : grad_scale, 0};
    a.n = 0;
    long long total4 = 0;
    for (; done < n_ranges && a.n < ADAM_MAX_RANGES; ++done) {
      if (counts[done] == 0) continue;
      a.start4[a.n] = (unsigned long long)starts[done] / 4;
      a.n4[a.n] = (unsigned long long)counts[done] / 4;
      a.step_size[a.n] = step_sizes[done];
      a.zero_grad[a.n] = zero_grad[done];
      total4 += counts[done] / 4;
      ++a.n;
    }
    if (a.n == 0) continue;
    long long want = (total4 + 255) / 256;      // one quad per thread and trip like m3p_adam_step, at most 4096 blocks, at least one per piece
    if (want > M3P_ADAM_MAXBLK) want = M3P_ADAM_MAXBLK;
    if (want < a.n) want = a.n;
    int b = 0;
    for (int r = 0; r < a.n; ++r) {
      a.blk0[r] = b;
      long long share = (long long)((double)a.n4[r] / (double)total4 * (double)want);
      if (share < 1) share = 1;
      b += (int)share;
    }
    a.blk0[a.n] = b;
    hipLaunchKernelGGL(adam_ranges_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, a);
    M3P_CHECK_LAUNCH();
  }
  return M3P_OK;
}

int m3p_adam_step(float* p, float* g, float* m, float* v, void* w16, long long n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, float step_size, const double* gnorm_sq, float max_norm,
                  float grad_scale, int zero_grad, void* stream) {
  if (n <= 0 || (n % 4) != 0) return M3P_EINVAL;
  if (((uintptr_t)p & 15) || ((uintptr_t)g & 15) || ((uintptr_t)m & 15) || ((uintptr_t)v & 15) || ((uintptr_t)w16 & 7))
    return M3P_EINVAL;
  AdamArgs a = {p, g, m, v, (bf16*)w16, (size_t)n / 4, lr, beta1, beta2, eps, weight_decay, step_size, gnorm_sq,
                max_norm, grad_scale == 0.f ? 1.f : grad_scale, zero_grad};
  const int blocks = (int)((a.n4 + 255) / 256 < 4096 ? (a.n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_transpose_bf16(const void* src, void* dst, int rows, int cols, int ld_src, int ld_dst, void* stream) {
  if (rows <= 0 || cols <= 0 || ld_src < cols || ld_dst < rows) return M3P_EINVAL;
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)src, (bf16*)dst, rows, cols, ld_src, ld_dst);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

}  // extern "C"

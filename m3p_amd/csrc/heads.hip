// Prediction heads of TransformerModel.predict (M3P/src/model/transformer.py:1183-1214):
//   * row gather / scatter for the masked-LM positions (:1208 boolean-mask gather)
//   * cross-entropy over the vocabulary logits (PredLayer.forward :104-117,
//     F.cross_entropy(mean)) with the gradient written in place of the logits.
// The vocabulary projection itself is m3p_gemm_nt_bf16 against the tied embedding.
#include "common.hpp"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

// dst[i,:] = src[idx[i],:]   (rows of d bf16, 8-byte chunks)
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16* __restrict__ src, const int32_t* __restrict__ idx,
                                                          bf16* __restrict__ dst, int n, int d) {
  const int nchunk = d >> 2;
  const size_t total = (size_t)n * nchunk;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / nchunk), c = (int)(i - (size_t)r * nchunk);
    *reinterpret_cast<bf16x4*>(dst + (size_t)r * d + 4 * c) =
        *reinterpret_cast<const bf16x4*>(src + (size_t)idx[r] * d + 4 * c);
  }
}
// dst[idx[i],:] += src[i,:]  (idx unique -> plain read-modify-write)
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const bf16* __restrict__ src, const int32_t* __restrict__ idx,
                                                               bf16* __restrict__ dst, int n, int d) {
  const int nchunk = d >> 2;
  const size_t total = (size_t)n * nchunk;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / nchunk), c = (int)(i - (size_t)r * nchunk);
    bf16* p = dst + (size_t)idx[r] * d + 4 * c;
    Vec4<bf16>::store(p, Vec4<bf16>::load(p) + Vec4<bf16>::load(src + (size_t)r * d + 4 * c));
  }
}

// dst[ids[i],:] += rows[i,:] in fp32 (ids repeat: atomics; rows of pad_index skipped, nn.Embedding(padding_idx)).
// One wave per row, one atomic instruction = 64 consecutive floats (full 128-B lines), like the fused
// scatter of embed_bwd_rows_kernel.  Applies the token rows all-gathered from the other data-parallel ranks.
__global__ __launch_bounds__(256) void scatter_add_token_rows_kernel(const bf16* __restrict__ rows, int ld_rows, const int64_t* __restrict__ ids,
                                                                     float* __restrict__ dst, int n, int d, int pad_index) {
  const int lane = threadIdx.x & 63;
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwave = (int)((gridDim.x * blockDim.x) >> 6);
  for (int r = wave; r < n; r += nwave) {
    const int64_t id = ids[r];
    if (id == pad_index) continue;
    const bf16* src = rows + (size_t)r * ld_rows;
    float* out = dst + (size_t)id * d;
    for (int e = lane; e < d; e += 64) {
      const float v = (float)src[e];
      if (v != 0.f) unsafeAtomicAdd(out + e, v);
    }
  }
}

// One 1024-thread block per row (a 500-KB row of V = 250 002 bf16 logits: 16 waves keep enough
// 16-byte loads in flight to stream it at HBM speed, and the second pass re-reads it while it is
// still in the last-level cache).  Pass 1: per-thread running (max, sum-exp) updated once per 8
// logits (one rescale per group instead of one exp-or-branch per element), combined across the
// block.  Pass 2: dlogits = (softmax - onehot) * gscale in place, padding columns [V, ld) zeroed
// so the gradient tensor can be contracted over the padded pitch.
__global__ __launch_bounds__(1024) void ce_fwd_bwd_kernel(bf16* __restrict__ logits, int ld, int V,
                                                          const int64_t* __restrict__ target, float* __restrict__ row_loss,
                                                          float* __restrict__ loss_sum, float loss_scale, float gscale) {
  __shared__ float s_m[16], s_s[16];
  constexpr float kLog2e = 1.4426950408889634f;
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
  bf16* lr = logits + (size_t)row * ld;
  const int nchunk = ld >> 3;                  // 8 logits = 16 bytes (ld % 8 == 0 checked by the launcher)
  float m = -INFINITY, s = 0.f;               // running max and sum of exp(x - m)
  for (int c = tid; c < nchunk; c += 1024) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(lr + 8 * c);
    float x[8];
    float gm = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      x[j] = (8 * c + j < V) ? (float)v[j] : -INFINITY;
      gm = fmaxf(gm, x[j]);
    }
    if (gm > m) {                              // rare after the first few groups
      s *= __builtin_amdgcn_exp2f((m - gm) * kLog2e);      // m = -inf: exp2(-inf) = 0, s is 0 anyway
      m = gm;
    }
    const float mb = m * kLog2e;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += __builtin_amdgcn_exp2f(__builtin_fmaf(x[j], kLog2e, -mb));
  }
  // combine (m, s) pairs across the wave, then across the 16 waves
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    const float mm = fmaxf(m, m2);
    s = (m == -INFINITY ? 0.f : s * __expf(m - mm)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mm));
    m = mm;
  }
  if (lane == 0) { s_m[wib] = m; s_s[wib] = s; }
  __syncthreads();
  float M = -INFINITY;
#pragma unroll
  for (int w = 0; w < 16; ++w) M = fmaxf(M, s_m[w]);
  float Ssum = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) Ssum += (s_m[w] == -INFINITY) ? 0.f : s_s[w] * __expf(s_m[w] - M);
  const float lse = M + __logf(Ssum);
  const int64_t tgt = target[row];
  if (tid == 0) {
    const float l = lse - (float)lr[tgt];
    row_loss[row] = l;
    // (one atomic per row on a single address serialises ~5k adds; callers that want the
    //  mean reduce row_loss themselves and pass loss_sum = NULL)
    if (loss_sum) unsafeAtomicAdd(loss_sum, l * loss_scale);
  }
  __syncthreads();   // row_loss read of lr[tgt] happens before the in-place overwrite below
  const float lb = lse * kLog2e;
  for (int c = tid; c < nchunk; c += 1024) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(lr + 8 * c);
    bf16x8 g;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = 8 * c + j;
      float p = (col < V) ? __builtin_amdgcn_exp2f(__builtin_fmaf((float)v[j], kLog2e, -lb)) : 0.f;
      if (col == tgt) p -= 1.f;
      g[j] = (bf16)(p * gscale);
    }
    *reinterpret_cast<bf16x8*>(lr + 8 * c) = g;
  }
}

// ---- cross-entropy with the vocabulary-bias gradient folded in (the MLM head at V = 250 002: the separate column-sum
// pass over the 2.4-GB gradient costs 0.42 ms).  Three launches:
//   ce_stats_kernel      one 1024-thread block per row: log-sum-exp + loss (pass 1 of ce_fwd_bwd_kernel)
//   ce_grad_tile_kernel  one 256-thread block per (64 rows x 2048 columns): the gradient in place AND the column sums of
//                        its 64 rows in registers -> part[row group][column] (fp32)
//   ce_colsum_reduce     colsum[c] = sum over row groups
__global__ __launch_bounds__(1024) void ce_stats_kernel(const bf16* __restrict__ logits, int ld, int V, const int64_t* __restrict__ target,
                                                        float* __restrict__ row_loss, float* __restrict__ row_lse) {
  __shared__ float s_m[16], s_s[16];
  constexpr float kLog2e = 1.4426950408889634f;
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
  const bf16* lr = logits + (size_t)row * ld;
  const int nchunk = ld >> 3;
  float m = -INFINITY, s = 0.f;
  for (int c = tid; c < nchunk; c += 1024) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(lr + 8 * c);
    float x[8];
    float gm = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      x[j] = (8 * c + j < V) ? (float)v[j] : -INFINITY;
      gm = fmaxf(gm, x[j]);
    }
    if (gm > m) {
      s *= __builtin_amdgcn_exp2f((m - gm) * kLog2e);
      m = gm;
    }
    const float mb = m * kLog2e;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += __builtin_amdgcn_exp2f(__builtin_fmaf(x[j], kLog2e, -mb));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    const float mm = fmaxf(m, m2);
    s = (m == -INFINITY ? 0.f : s * __expf(m - mm)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mm));
    m = mm;
  }
  if (lane == 0) { s_m[wib] = m; s_s[wib] = s; }
  __syncthreads();
  if (tid == 0) {
    float M = -INFINITY;
    for (int w = 0; w < 16; ++w) M = fmaxf(M, s_m[w]);
    float Ssum = 0.f;
    for (int w = 0; w < 16; ++w) Ssum += (s_m[w] == -INFINITY) ? 0.f : s_s[w] * __expf(s_m[w] - M);
    const float lse = M + __logf(Ssum);
    row_lse[row] = lse;
    row_loss[row] = lse - (float)lr[target[row]];
  }
}

constexpr int CE_RB = 64, CE_CB = 2048;      // rows / columns of a gradient tile
// log-sum-exp of the rows from the (maximum, sum) pairs of their 64-column blocks (the vocabulary projection's
// M3P_EPI_BIAS_LSE epilogue, stats [n_blocks][n_rows]).  Stage 1: a workgroup owns 64 rows and one of CE_LSE_SPLIT ranges of
// blocks; its four waves take every fourth block of the range (a wave reads 64 consecutive rows of a block: 512 bytes) and
// fold through LDS -> part [split][row].  Stage 2: a thread per row folds the splits and reads the target's logit.
constexpr int CE_LSE_SPLIT = 32;
__device__ __forceinline__ void lse_fold(float& m, float& s, float m2, float s2) {
  const float mm = fmaxf(m, m2);
  const float a = (m == -INFINITY) ? 0.f : s * __expf(m - mm);
  const float b = (m2 == -INFINITY) ? 0.f : s2 * __expf(m2 - mm);
  m = mm;
  s = a + b;
}
__global__ __launch_bounds__(256) void ce_lse_partial_kernel(const float2* __restrict__ stats, int n_blocks, int n_rows,
                                                             float2* __restrict__ part) {
  __shared__ float2 sh[4][64];
  const int r = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int row = blockIdx.x * 64 + r;
  const int per = (n_blocks + CE_LSE_SPLIT - 1) / CE_LSE_SPLIT;
  const int b0 = blockIdx.y * per, b1 = min(n_blocks, b0 + per);
  float m = -INFINITY, s = 0.f;
  int b = b0 + q;
  for (; b + 12 < b1; b += 16) {           // four independent loads in flight per thread: the fold is a dependent chain
    float2 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = stats[(size_t)(b + 4 * k) * n_rows + row];
#pragma unroll
    for (int k = 0; k < 4; ++k) lse_fold(m, s, v[k].x, v[k].y);
  }
  for (; b < b1; b += 4) {
    const float2 v = stats[(size_t)b * n_rows + row];
    lse_fold(m, s, v.x, v.y);
  }
  sh[q][r] = float2{m, s};
  __syncthreads();
  if (q == 0) {
#pragma unroll
    for (int k = 1; k < 4; ++k) lse_fold(m, s, sh[k][r].x, sh[k][r].y);
    part[(size_t)blockIdx.y * n_rows + row] = float2{m, s};
  }
}
__global__ __launch_bounds__(256) void ce_lse_final_kernel(const float2* __restrict__ part, int n_rows, const bf16* __restrict__ logits,
                                                           int ld, const int64_t* __restrict__ target, float* __restrict__ row_loss,
                                                           float* __restrict__ row_lse) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= n_rows) return;
  float m = -INFINITY, s = 0.f;
  for (int k = 0; k < CE_LSE_SPLIT; ++k) {
    const float2 v = part[(size_t)k * n_rows + row];
    lse_fold(m, s, v.x, v.y);
  }
  const float lse = m + __logf(s);
  row_lse[row] = lse;
  row_loss[row] = lse - (float)logits[(size_t)row * ld + target[row]];
}

__global__ __launch_bounds__(256) void ce_grad_tile_kernel(bf16* __restrict__ logits, int ld, int n_rows, int V,
                                                           const int64_t* __restrict__ target, const float* __restrict__ row_lse,
                                                           float gscale, float* __restrict__ part) {
  constexpr float kLog2e = 1.4426950408889634f;
  const int col0 = blockIdx.x * CE_CB + threadIdx.x * 8;
  if (col0 >= ld) return;
  const int r0 = blockIdx.y * CE_RB, r1 = min(n_rows, r0 + CE_RB);
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // four rows' loads in flight before the first use: the gradient is written over the logits, so left to itself the compiler
  // keeps every load behind the previous row's store (one 16-byte load in flight per thread: 5.0 TB/s on 4.9 GB)
  constexpr int U = 4;
  for (int rb = r0; rb < r1; rb += U) {
    bf16x8 v[U];
    float lb[U];
    int tc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = min(rb + u, r1 - 1);
      v[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(logits + (size_t)r * ld + col0));
      lb[u] = row_lse[r] * kLog2e;
      tc[u] = (int)(target[r] - col0);               // the target's position inside this thread's eight columns (or outside)
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (rb + u < r1) {
        bf16x8 g;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float q = (col0 + j < V) ? __builtin_amdgcn_exp2f(__builtin_fmaf((float)v[u][j], kLog2e, -lb[u])) : 0.f;
          if (j == tc[u]) q -= 1.f;
          g[j] = (bf16)(q * gscale);
          cs[j] += (float)g[j];                      // the sum of the ROUNDED gradient: what summing the bf16 tensor gives
        }
        *reinterpret_cast<bf16x8*>(logits + (size_t)(rb + u) * ld + col0) = g;
      }
    }
  }
  float* out = part + (size_t)blockIdx.y * ld + col0;
  *reinterpret_cast<f32x4*>(out) = f32x4{cs[0], cs[1], cs[2], cs[3]};
  *reinterpret_cast<f32x4*>(out + 4) = f32x4{cs[4], cs[5], cs[6], cs[7]};
}

// (row groups split over gridDim.y, folded with atomics into the zeroed output: 245 column blocks alone leave most of
//  the chip idle and every thread 152 dependent loads deep - 95 us for 152 MB)
__global__ __launch_bounds__(256) void ce_colsum_reduce_kernel(const float* __restrict__ part, int ld, int n_groups, float* __restrict__ out) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c >= ld) return;
  const int per = (n_groups + gridDim.y - 1) / gridDim.y;
  const int g0 = blockIdx.y * per, g1 = min(n_groups, g0 + per);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int g = g0; g < g1; ++g) acc += *reinterpret_cast<const f32x4*>(part + (size_t)g * ld + c);
#pragma unroll
  for (int j = 0; j < 4; ++j) unsafeAtomicAdd(out + c + j, acc[j]);
}

// out[c] += scale * sum_r x[r, c]   (x bf16 [n, ld], c < ncols); rows split over gridDim.y
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ x, int ld, int n, int ncols,
                                                     float* __restrict__ out, const float* __restrict__ scale_ptr) {
  const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
  if (4 * c4 >= ncols) return;
  const int rows_per = (n + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per, r1 = min(n, r0 + rows_per);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int r = r0; r < r1; ++r) acc += Vec4<bf16>::load(x + (size_t)r * ld + 4 * c4);
  const float sc = scale_ptr ? *scale_ptr : 1.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (4 * c4 + j < ncols) unsafeAtomicAdd(out + 4 * c4 + j, acc[j] * sc);
}

}  // namespace

extern "C" {

size_t m3p_ce_colsum_workspace_bytes(int ld, int n_rows) {
  return (size_t)((n_rows + CE_RB - 1) / CE_RB) * (size_t)ld * sizeof(float);
}

int m3p_ce_fwd_bwd_colsum(void* logits, int ld, int n_rows, int V, const int64_t* target, float* row_loss, float* row_lse,
                          float grad_scale, float* colsum, void* workspace, size_t workspace_bytes, void* stream) {
  if (n_rows <= 0 || V <= 0 || ld < V || (ld % 8) != 0 || ((uintptr_t)logits & 15) || ((uintptr_t)colsum & 15) ||
      ((uintptr_t)workspace & 15))
    return M3P_EINVAL;
  if (workspace_bytes < m3p_ce_colsum_workspace_bytes(ld, n_rows)) return M3P_EINVAL;
  hipLaunchKernelGGL(ce_stats_kernel, dim3(n_rows), dim3(1024), 0, (hipStream_t)stream, (const bf16*)logits, ld, V, target, row_loss, row_lse);
  return m3p_ce_bwd_colsum(logits, ld, n_rows, V, target, row_lse, grad_scale, colsum, workspace, workspace_bytes, stream);
}

int m3p_ce_lse_from_blocks(const void* stats, int n_blocks, int n_rows, const void* logits, int ld, const int64_t* target,
                           float* row_loss, float* row_lse, void* scratch, void* stream) {
  if (n_blocks <= 0 || n_rows <= 0 || (n_rows % 64) != 0 || !stats || !logits || !target || !row_loss || !row_lse || !scratch ||
      ((uintptr_t)stats & 7) || ((uintptr_t)scratch & 7))
    return M3P_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_lse_partial_kernel, dim3(n_rows / 64, CE_LSE_SPLIT), dim3(256), 0, st, (const float2*)stats, n_blocks, n_rows,
                     (float2*)scratch);
  hipLaunchKernelGGL(ce_lse_final_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, st, (const float2*)scratch, n_rows,
                     (const bf16*)logits, ld, target, row_loss, row_lse);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_ce_bwd_colsum(void* logits, int ld, int n_rows, int V, const int64_t* target, const float* row_lse,
                      float grad_scale, float* colsum, void* workspace, size_t workspace_bytes, void* stream) {
  if (n_rows <= 0 || V <= 0 || ld < V || (ld % 8) != 0 || ((uintptr_t)logits & 15) || ((uintptr_t)colsum & 15) ||
      ((uintptr_t)workspace & 15) || !row_lse)
    return M3P_EINVAL;
  if (workspace_bytes < m3p_ce_colsum_workspace_bytes(ld, n_rows)) return M3P_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int groups = (n_rows + CE_RB - 1) / CE_RB;
  hipLaunchKernelGGL(ce_grad_tile_kernel, dim3((ld + CE_CB - 1) / CE_CB, groups), dim3(256), 0, st, (bf16*)logits, ld, n_rows, V,
                     target, (const float*)row_lse, grad_scale, (float*)workspace);
  if (hipMemsetAsync(colsum, 0, (size_t)ld * sizeof(float), st) != hipSuccess) return M3P_EINVAL;
  hipLaunchKernelGGL(ce_colsum_reduce_kernel, dim3((ld / 4 + 255) / 256, groups >= 16 ? 8 : 1), dim3(256), 0, st,
                     (const float*)workspace, ld, groups, colsum);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_colsum_bf16(const void* x, int ld, int n, int ncols, float* out, const float* scale_ptr, void* stream) {
  if (n <= 0 || ncols <= 0 || (ld % 4) != 0 || ld < ((ncols + 3) / 4) * 4) return M3P_EINVAL;
  const int c4 = (ncols + 3) / 4;
  int ysplit = 1;
  while (ysplit < 64 && ((c4 + 255) / 256) * ysplit < 1024 && n / (ysplit * 2) >= 16) ysplit *= 2;
  hipLaunchKernelGGL(colsum_kernel, dim3((c4 + 255) / 256, ysplit), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ld,
                     n, ncols, out, scale_ptr);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_gather_rows(const void* src, const int32_t* idx, void* dst, int n, int d, void* stream) {
  if (n <= 0) return M3P_OK;
  if ((d % 4) != 0) return M3P_EINVAL;
  const size_t total = (size_t)n * (d / 4);
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, idx, (bf16*)dst, n, d);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_scatter_add_rows(const void* src, const int32_t* idx, void* dst, int n, int d, void* stream) {
  if (n <= 0) return M3P_OK;
  if ((d % 4) != 0) return M3P_EINVAL;
  const size_t total = (size_t)n * (d / 4);
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, idx, (bf16*)dst, n, d);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_scatter_add_token_rows(const void* rows, int ld_rows, const int64_t* ids, float* dst, int n, int d, int pad_index, void* stream) {
  if (n <= 0) return M3P_OK;
  if (d <= 0 || ld_rows < d) return M3P_EINVAL;
  const int blocks = (n + 3) / 4 < 4096 ? (n + 3) / 4 : 4096;
  hipLaunchKernelGGL(scatter_add_token_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)rows, ld_rows, ids,
                     dst, n, d, pad_index);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_ce_fwd_bwd(void* logits, int ld, int n_rows, int V, const int64_t* target, float* row_loss, float* loss_sum,
                   float loss_scale, float grad_scale, void* stream) {
  if (n_rows <= 0 || V <= 0 || ld < V || (ld % 8) != 0 || ((uintptr_t)logits & 15)) return M3P_EINVAL;
  hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3(n_rows), dim3(1024), 0, (hipStream_t)stream, (bf16*)logits, ld, V, target,
                     row_loss, loss_sum, loss_scale, grad_scale);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

}  // extern "C"

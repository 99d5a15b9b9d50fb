// Small index / mask / loss kernels that replace chains of framework elementwise launches around the hot path
// (profiles/r02_bench_kernel_stats.csv: ~100 `at::native` launches per training step; each dependent launch costs a
// ~1.5-us boundary on top of its own few microseconds, and under data parallelism they are what a step's host thread
// spends its time enqueueing).  All of them are latency-bound single-pass kernels: nothing here is shaped for MFMA.
//   m3p_seq_masks          prefix validity of jointfwd (M3P/src/model/transformer.py:59-78 get_masks, :917-919)
//   m3p_mask_to_rows       row numbers of the True entries of pred_mask in (t, b) order (:1208 boolean gather)
//   m3p_cast_rows_f32_bf16 the collate's (n, R, 2048) region features read through their (R, n) transposed view (:897-898)
//   m3p_scale_*            upstream-gradient scaling of the MLM head's operands by a DEVICE scalar (no host read)
//   m3p_itm_loss_fwd_bwd   xtrainer.py:2357-2372: CE over groups of sample_n + BCE against the one-hot labels
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256)
void seq_masks_kernel(const long long* __restrict__ len_a, const long long* __restrict__ len_b, int B, int S,
                      int* __restrict__ totlen, uint8_t* __restrict__ rowmask) {
  const int b = blockIdx.x;
  const int n = (int)(len_a[b] + (len_b ? len_b[b] : 0));
  if (threadIdx.x == 0) totlen[b] = n;
  for (int s = threadIdx.x; s < S; s += blockDim.x) rowmask[(size_t)b * S + s] = (s < n) ? 1 : 0;
}

// one workgroup: thread t owns the contiguous slice [t * per, (t + 1) * per) of the mask, block-wide exclusive scan of the
// slice counts through LDS, ordered write.  n_mask <= 2^20 (per <= 1024).
__global__ __launch_bounds__(1024)
void mask_to_rows_kernel(const uint8_t* __restrict__ mask, int n_mask, int inner, long long s0, long long s1,
                         long long soff, int d, int* __restrict__ rows, int n_rows) {
  __shared__ int wsum[16];
  __shared__ int total_sh;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int per = (n_mask + 1023) / 1024;
  const int lo = min(tid * per, n_mask), hi = min(lo + per, n_mask);
  int cnt = 0;
  for (int i = lo; i < hi; ++i) cnt += mask[i] ? 1 : 0;
  // inclusive scan inside the wave
  int inc = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(inc, o, 64);
    if (lane >= o) inc += v;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < 16; ++i) { const int v = wsum[i]; wsum[i] = run; run += v; }
    total_sh = run;
  }
  __syncthreads();
  int pos = wsum[w] + inc - cnt;
  for (int i = lo; i < hi; ++i) {
    if (mask[i]) {
      if (pos < n_rows) {
        const int t = i / inner, b = i - t * inner;
        rows[pos] = (int)((soff + t * s0 + b * s1) / d);
      }
      ++pos;
    }
  }
  // fewer True entries than the caller counted on the host: the tail points at row 0 (never the case for a consistent batch)
  for (int k = total_sh + tid; k < n_rows; k += 1024) rows[k] = 0;
}

__global__ __launch_bounds__(256)
void cast_rows_kernel(const float* __restrict__ in, long long s0, long long s1, int n1, int cols,
                      bf16* __restrict__ out, long long nrows) {
  // one wave per row piece of 256 floats: 16 B loads, 8 B stores
  const int cpr = cols >> 2;                                  // float4 per row
  const long long total = nrows * cpr;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / cpr;
    const int c = (int)(i - row * cpr) * 4;
    const long long i0 = row / n1, i1 = row - i0 * n1;
    const f32x4 v = *reinterpret_cast<const f32x4*>(in + i0 * s0 + i1 * s1 + c);
    *reinterpret_cast<bf16x4*>(out + row * cols + c) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  }
}

template <typename TI>
__global__ __launch_bounds__(256)
void scale_to_bf16_kernel(const TI* __restrict__ in, const float* __restrict__ g, bf16* __restrict__ out, long long n4) {
  const float s = g[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 v = Vec4<TI>::load(in + 4 * i);
    v *= s;
    Vec4<bf16>::store(out + 4 * i, v);
  }
}

__global__ __launch_bounds__(256)
void axpy_dev_kernel(float* __restrict__ dst, const float* __restrict__ src, const float* __restrict__ g, long long n) {
  const float s = g[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dst[i] += s * src[i];
}

__device__ __forceinline__ float softplus_f(float x) {      // log(1 + exp(x)), stable
  return fmaxf(x, 0.f) + log1pf(__expf(-fabsf(x)));
}

// one workgroup.  scores [G * n] fp32 (group g = scores[g n .. g n + n)), pos [G] int64.
//   ce  = mean_g ( logsumexp(scores_g) - scores_g[pos_g] )               (F.cross_entropy(view(-1, n), pos))
//   bce = mean_i ( softplus(s_i) - y_i s_i ),  y = one_hot(pos)          (F.binary_cross_entropy_with_logits)
// loss = w_ce ce + w_bce bce;  dscores = d loss / d scores.
__global__ __launch_bounds__(256)
void itm_loss_kernel(const float* __restrict__ scores, const long long* __restrict__ pos, int G, int n, float w_ce,
                     float w_bce, float* __restrict__ loss, float* __restrict__ dscores) {
  __shared__ float red[4];
  float acc = 0.f;
  const float inv_g = 1.f / (float)G, inv_all = 1.f / ((float)G * (float)n);
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    const float* s = scores + (size_t)g * n;
    const long long p64 = pos[g];
    // a label outside [0, n) is a caller bug (F.one_hot / cross_entropy raise on it): no out-of-bounds read, and the loss
    // comes back NaN so that it cannot go unnoticed
    const bool bad = p64 < 0 || p64 >= n;
    const int p = bad ? 0 : (int)p64;
    float lse = 0.f;
    if (w_ce != 0.f) {
      float mx = -INFINITY;
      for (int j = 0; j < n; ++j) mx = fmaxf(mx, s[j]);
      float se = 0.f;
      for (int j = 0; j < n; ++j) se += __expf(s[j] - mx);
      lse = mx + __logf(se);
    }
    float l = bad ? __builtin_nanf("") : 0.f;
    if (w_ce != 0.f) l += w_ce * inv_g * (lse - s[p]);
    for (int j = 0; j < n; ++j) {
      const float y = (j == p) ? 1.f : 0.f;
      float d = 0.f;
      if (w_ce != 0.f) d += w_ce * inv_g * (__expf(s[j] - lse) - y);
      if (w_bce != 0.f) {
        l += w_bce * inv_all * (softplus_f(s[j]) - y * s[j]);
        d += w_bce * inv_all * (1.f / (1.f + __expf(-s[j])) - y);
      }
      dscores[(size_t)g * n + j] = d;
    }
    acc += l;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = red[0] + red[1] + red[2] + red[3];
}

int grid_for(long long work, int per_block) {
  long long g = (work + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" {

int m3p_seq_masks(const int64_t* lengths, const int64_t* lengths_b, int B, int S, int32_t* totlen, uint8_t* rowmask,
                  void* stream) {
  if (B <= 0 || S <= 0 || !lengths || !totlen || !rowmask) return M3P_EINVAL;
  hipLaunchKernelGGL(seq_masks_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const long long*)lengths,
                     (const long long*)lengths_b, B, S, totlen, rowmask);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_mask_to_rows(const uint8_t* mask, int n_mask, int inner, long long s0, long long s1, long long soff, int d,
                     int32_t* rows, int n_rows, void* stream) {
  if (n_mask <= 0 || inner <= 0 || d <= 0 || n_rows < 0 || !mask || (n_rows && !rows)) return M3P_EINVAL;
  if (n_rows == 0) return M3P_OK;
  hipLaunchKernelGGL(mask_to_rows_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, mask, n_mask, inner, s0, s1, soff, d,
                     rows, n_rows);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_cast_rows_f32_bf16(const float* in, long long s0, long long s1, int n0, int n1, int cols, void* out, void* stream) {
  if (n0 <= 0 || n1 <= 0 || cols <= 0 || (cols & 3) || (s0 & 3) || (s1 & 3) || ((uintptr_t)in & 15) || ((uintptr_t)out & 7))
    return M3P_EINVAL;
  const long long nrows = (long long)n0 * n1;
  hipLaunchKernelGGL(cast_rows_kernel, dim3(grid_for(nrows * (cols >> 2), 1024)), dim3(256), 0, (hipStream_t)stream, in, s0, s1,
                     n1, cols, (bf16*)out, nrows);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_scale_bf16_dev(const void* in, int in_is_f32, const float* g, void* out, long long n, void* stream) {
  if (n <= 0 || (n & 3) || !g || ((uintptr_t)in & (in_is_f32 ? 15 : 7)) || ((uintptr_t)out & 7)) return M3P_EINVAL;
  const int grid = grid_for(n >> 2, 1024);
  if (in_is_f32)
    hipLaunchKernelGGL(scale_to_bf16_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)in, g, (bf16*)out, n >> 2);
  else
    hipLaunchKernelGGL(scale_to_bf16_kernel<bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)in, g, (bf16*)out, n >> 2);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_axpy_dev_f32(float* dst, const float* src, const float* g, long long n, void* stream) {
  if (n <= 0 || !dst || !src || !g) return M3P_EINVAL;
  hipLaunchKernelGGL(axpy_dev_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, dst, src, g, n);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_itm_loss_fwd_bwd(const float* scores, const int64_t* pos, int n_groups, int sample_n, float w_ce, float w_bce,
                         float* loss, float* dscores, void* stream) {
  if (n_groups <= 0 || sample_n <= 0 || !scores || !pos || !loss || !dscores) return M3P_EINVAL;
  hipLaunchKernelGGL(itm_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scores, (const long long*)pos, n_groups,
                     sample_n, w_ce, w_bce, loss, dscores);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

}  // extern "C"

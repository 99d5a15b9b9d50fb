// Image-text-matching head of M3P (M3P/src/model/transformer.py:546-558 BertPooler,
// :1194-1197 predict(is_relation=True)):  score[b] = w2 . tanh(W1 h[b, 0, :] + b1) + b2
// where position 0 of the joint sequence is image region 0.  B x d x d multiply-adds: tiny,
// fp32 on master weights, one launch forward and two backward (the d x d weight gradient goes
// through the regular weight-gradient GEMM on the bf16 copies this backward leaves).
#include "common.hpp"

namespace {

// grid = B, block = 256.  One wave per output feature j (lanes over k, coalesced W1 rows).
__global__ __launch_bounds__(256) void itm_head_fwd_kernel(const bf16* __restrict__ h, int ld_h, const float* __restrict__ W1,
                                                           const float* __restrict__ b1, const float* __restrict__ w2,
                                                           const float* __restrict__ b2, float* __restrict__ pooled,
                                                           float* __restrict__ scores, int d) {
  extern __shared__ float sm[];          // [d] input row | [d] pooled row | [4] partials
  float* hs = sm;
  float* ps = sm + d;
  float* red = sm + 2 * d;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const bf16* hr = h + (size_t)b * ld_h;
  for (int k = tid; k < d; k += 256) hs[k] = (float)hr[k];
  __syncthreads();
  for (int j = wid; j < d; j += 4) {
    const float* wr = W1 + (size_t)j * d;
    float acc = 0.f;
    for (int k = lane; k < d; k += 64) acc += wr[k] * hs[k];
    acc = wave_sum(acc);
    if (lane == 0) {
      const float p = tanhf(acc + b1[j]);
      ps[j] = p;
      pooled[(size_t)b * d + j] = p;
    }
  }
  __syncthreads();
  float s = 0.f;
  for (int j = tid; j < d; j += 256) s += w2[j] * ps[j];
  s = wave_sum(s);
  if (lane == 0) red[wid] = s;
  __syncthreads();
  if (tid == 0) scores[b] = red[0] + red[1] + red[2] + red[3] + b2[0];
}

// grid = B, block = 256: dpre = dscore * w2 * (1 - pooled^2); dh[k] = sum_j dpre[j] W1[j][k];
// also leaves bf16 copies of dpre and of the input row (operands of the W1 weight-gradient GEMM).
__global__ __launch_bounds__(256) void itm_head_bwd_rows_kernel(const float* __restrict__ dscores, const bf16* __restrict__ h,
                                                                int ld_h, const float* __restrict__ pooled,
                                                                const float* __restrict__ W1, const float* __restrict__ w2,
                                                                bf16* __restrict__ dh, bf16* __restrict__ dpre16,
                                                                bf16* __restrict__ h16, int d) {
  extern __shared__ float sm[];          // [d] dpre
  const int b = blockIdx.x, tid = threadIdx.x;
  const float ds = dscores[b];
  const bf16* hr = h + (size_t)b * ld_h;
  for (int j = tid; j < d; j += 256) {
    const float p = pooled[(size_t)b * d + j];
    const float g = ds * w2[j] * (1.f - p * p);
    sm[j] = g;
    dpre16[(size_t)b * d + j] = (bf16)g;
    h16[(size_t)b * d + j] = hr[j];
  }
  __syncthreads();
  for (int k = tid; k < d; k += 256) {
    float acc = 0.f;
    for (int j = 0; j < d; ++j) acc += sm[j] * W1[(size_t)j * d + k];
    dh[(size_t)b * d + k] = (bf16)acc;
  }
}

// grid = ceil(d / 256): column reductions over the batch, accumulated into the gradient arena
__global__ __launch_bounds__(256) void itm_head_bwd_cols_kernel(const float* __restrict__ dscores, const float* __restrict__ pooled,
                                                                const float* __restrict__ w2, float* __restrict__ db1,
                                                                float* __restrict__ dw2, float* __restrict__ db2, int B, int d) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < d) {
    const float w = w2[j];
    float a1 = 0.f, a2 = 0.f;
    for (int b = 0; b < B; ++b) {
      const float ds = dscores[b], p = pooled[(size_t)b * d + j];
      a1 += ds * w * (1.f - p * p);
      a2 += ds * p;
    }
    db1[j] += a1;
    dw2[j] += a2;
  }
  if (j == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dscores[b];
    db2[0] += s;
  }
}

}  // namespace

extern "C" {

int m3p_itm_head_fwd(const void* h, int ld_h, const float* W1, const float* b1, const float* w2, const float* b2,
                     float* pooled, float* scores, int B, int d, void* stream) {
  if (B <= 0 || d <= 0 || d > 4096 || ld_h < d) return M3P_EINVAL;
  const size_t lds = (size_t)(2 * d + 4) * sizeof(float);
  hipLaunchKernelGGL(itm_head_fwd_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, (const bf16*)h, ld_h, W1, b1, w2, b2,
                     pooled, scores, d);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_itm_head_bwd(const float* dscores, const void* h, int ld_h, const float* pooled, const float* W1, const float* w2,
                     void* dh, void* dpre16, void* h16, float* db1, float* dw2, float* db2, int B, int d, void* stream) {
  if (B <= 0 || d <= 0 || d > 4096 || ld_h < d) return M3P_EINVAL;
  hipLaunchKernelGGL(itm_head_bwd_rows_kernel, dim3(B), dim3(256), (size_t)d * sizeof(float), (hipStream_t)stream, dscores,
                     (const bf16*)h, ld_h, pooled, W1, w2, (bf16*)dh, (bf16*)dpre16, (bf16*)h16, d);
  M3P_CHECK_LAUNCH();
  hipLaunchKernelGGL(itm_head_bwd_cols_kernel, dim3((d + 255) / 256), dim3(256), 0, (hipStream_t)stream, dscores, pooled, w2,
                     db1, dw2, db2, B, d);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

}  // extern "C"

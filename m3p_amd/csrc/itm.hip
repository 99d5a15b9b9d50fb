// Image-text-matching head of M3P (M3P/src/model/transformer.py:546-558 BertPooler,
// :1194-1197 predict(is_relation=True)):  score[b] = w2 . tanh(W1 h[b, 0, :] + b1) + b2
// where position 0 of the joint sequence is image region 0.
// The two d x d products (W1 h forward, W1^T dpre and dpre^T h backward) run on the regular
// bf16 MFMA GEMMs (m3p_gemm_nt_bf16 with the bias epilogue, m3p_gemm_wgrad_bf16); the kernels
// here are the element-wise / reduction glue: tanh + score, the tanh derivative, and the
// bias / w2 gradient sums.  (A first version did the d x d products in fp32 with a wave per
// output feature: 935 us for B = 256, d = 768 - twenty attention kernels' worth.)
#include "common.hpp"

namespace {

// grid = B, block = 256: pooled = tanh(pre), score = w2 . pooled + b2
__global__ __launch_bounds__(256) void itm_score_fwd_kernel(const bf16* __restrict__ pre, const float* __restrict__ w2,
                                                            const float* __restrict__ b2, float* __restrict__ pooled,
                                                            float* __restrict__ scores, int d) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float s = 0.f;
  for (int j = tid; j < d; j += 256) {
    const float p = tanhf((float)pre[(size_t)b * d + j]);
    pooled[(size_t)b * d + j] = p;
    s += w2[j] * p;
  }
  s = wave_sum(s);
  if (lane == 0) red[wid] = s;
  __syncthreads();
  if (tid == 0) scores[b] = red[0] + red[1] + red[2] + red[3] + b2[0];
}

// grid = B, block = 256: dpre = dscore * w2 * (1 - pooled^2) as bf16, row-major [B, d] (operand of
// dW1 += dpre^T h) and transposed [d, ldt] (operand of dh = dpre W1 through the weight-gradient GEMM)
__global__ __launch_bounds__(256) void itm_score_bwd_rows_kernel(const float* __restrict__ dscores, const float* __restrict__ pooled,
                                                                 const float* __restrict__ w2, bf16* __restrict__ dpre16,
                                                                 bf16* __restrict__ dpreT16, int ldt, int d) {
  const int b = blockIdx.x;
  const float ds = dscores[b];
  for (int j = threadIdx.x; j < d; j += 256) {
    const float p = pooled[(size_t)b * d + j];
    const bf16 g = (bf16)(ds * w2[j] * (1.f - p * p));
    dpre16[(size_t)b * d + j] = g;
    dpreT16[(size_t)j * ldt + b] = g;
  }
}

// grid = ceil(d / 64), block = 256 = 4 batch groups x 64 features: column sums over the batch,
// accumulated into the gradient arena
__global__ __launch_bounds__(256) void itm_score_bwd_cols_kernel(const float* __restrict__ dscores, const float* __restrict__ pooled,
                                                                 const float* __restrict__ w2, float* __restrict__ db1,
                                                                 float* __restrict__ dw2, float* __restrict__ db2, int B, int d) {
  __shared__ float s1[4][64], s2[4][64], s3[4];
  const int jl = threadIdx.x & 63, bg = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + jl;
  float a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (j < d) {
    const float w = w2[j];
    for (int b = bg; b < B; b += 4) {
      const float ds = dscores[b], p = pooled[(size_t)b * d + j];
      a1 += ds * w * (1.f - p * p);
      a2 += ds * p;
      if (jl == 0) a3 += ds;
    }
  }
  s1[bg][jl] = a1; s2[bg][jl] = a2;
  if (jl == 0) s3[bg] = a3;
  __syncthreads();
  if (bg == 0 && j < d) {
    db1[j] += (s1[0][jl] + s1[1][jl]) + (s1[2][jl] + s1[3][jl]);
    dw2[j] += (s2[0][jl] + s2[1][jl]) + (s2[2][jl] + s2[3][jl]);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) db2[0] += (s3[0] + s3[1]) + (s3[2] + s3[3]);
}

}  // namespace

extern "C" {

int m3p_itm_score_fwd(const void* pre, const float* w2, const float* b2, float* pooled, float* scores, int B, int d,
                      void* stream) {
  if (B <= 0 || d <= 0) return M3P_EINVAL;
  hipLaunchKernelGGL(itm_score_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const bf16*)pre, w2, b2, pooled,
                     scores, d);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_itm_score_bwd(const float* dscores, const float* pooled, const float* w2, void* dpre16, void* dpreT16, int ldt,
                      float* db1, float* dw2, float* db2, int B, int d, void* stream) {
  if (B <= 0 || d <= 0 || ldt < B) return M3P_EINVAL;
  hipLaunchKernelGGL(itm_score_bwd_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dscores, pooled, w2,
                     (bf16*)dpre16, (bf16*)dpreT16, ldt, d);
  M3P_CHECK_LAUNCH();
  hipLaunchKernelGGL(itm_score_bwd_cols_kernel, dim3((d + 63) / 64), dim3(256), 0, (hipStream_t)stream, dscores, pooled, w2,
                     db1, dw2, db2, B, d);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

}  // extern "C"

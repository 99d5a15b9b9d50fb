// Attention of the causal / cross-attention decoder at inference time: transformer.py:149-210 with a key /
// value cache (:187-195) - every query row attends a prefix of a cached key / value sequence: its own position and
// everything before it (causal self-attention over the cache the step just extended) or the valid part of the encoder
// output (cross-attention).  Decoding is HBM / latency work - one new token per sequence and step, klen <= a few hundred
// keys of 128 bytes - so there are no MFMAs here: one wave per (sequence, head, query row), coalesced 16-byte loads, fp32
// softmax (the reference's is fp32 too, :202), bf16 context out.
#include "common.hpp"

namespace {

constexpr int QA_MAX_KEYS = 1024;     // keys per (sequence, head) a wave can score (LDS: 4 KB of scores per wave)

__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// q    bf16 [B*Tq, ld_q]   (already scaled by 1/sqrt(dh) and biased: the projection's epilogue does both)
// kv   bf16: key j of sequence b, head h at kv + b*kv_bstride + j*ld_kv + h*DH; its value H*DH elements further
// klen int32 [B] (nullable: every sequence has Lk keys); query t of a causal call sees keys 0 .. pos0 + t
// ctx  bf16 [B*Tq, H*DH]
// lse  fp32 [B, H, Tq] (nullable): log-sum-exp of the scores, kept for the backward of the training path
// thresh24 != 0: dropout on the probabilities (transformer.py:203), stream (seed) indexed ((b*H + h)*Tq + t)*Lk + key
template <int DH>
__global__ __launch_bounds__(256) void attn_query_fwd_kernel(const bf16* __restrict__ q, int ld_q, const bf16* __restrict__ kv,
                                                            long long kv_bstride, int ld_kv, const int32_t* __restrict__ klen,
                                                            bf16* __restrict__ ctx, float* __restrict__ lse, int B, int Tq, int H,
                                                            int Lk, int causal, int pos0, uint32_t seed, uint32_t thresh24,
                                                            float inv_keep) {
  __shared__ float sc[4][QA_MAX_KEYS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  long long item = (long long)blockIdx.x * 4 + wv;                // (b, t, h), h fastest
  const bool live = item < (long long)B * Tq * H;                 // (idle waves of the last block walk item 0 without storing)
  if (!live) item = 0;
  const int h = (int)(item % H);
  const int t = (int)((item / H) % Tq);
  const int b = (int)(item / ((long long)H * Tq));
  int nk = klen ? min(klen[b], Lk) : Lk;
  if (causal) nk = min(nk, pos0 + t + 1);
  const int d = H * DH;
  bf16* out = ctx + ((size_t)b * Tq + t) * d + h * DH;
  // (nk == 0 - a sequence of length 0: the reference would produce NaN; such rows are masked out by the caller - gives 0)
  // the query row in registers (the same 2 * DH bytes in every lane: broadcast loads)
  float qf[DH];
  const bf16* qp = q + ((size_t)b * Tq + t) * ld_q + h * DH;
#pragma unroll
  for (int c = 0; c < DH / 8; ++c) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(qp + c * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[c * 8 + e] = (float)v[e];
  }
  const bf16* kb = kv + (size_t)b * kv_bstride + h * DH;
  // scores: lane l takes keys l, l + 64, ...
  float mx = -INFINITY;
  for (int j = lane; j < nk; j += 64) {
    const bf16* kp = kb + (size_t)j * ld_kv;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < DH / 8; ++c) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(kp + c * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += qf[c * 8 + e] * (float)v[e];
    }
    sc[wv][j] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max_f(mx);
  float sum = 0.f;
  const uint32_t rbase = (uint32_t)(((size_t)(b * H + h) * Tq + t) * Lk);
  for (int j = lane; j < nk; j += 64) {
    const float p = __expf(sc[wv][j] - mx);
    sum += p;
    // (the normaliser is the sum of ALL probabilities; dropout zeroes / rescales the ones that enter the context)
    sc[wv][j] = (thresh24 == 0 || m3p_keep(rbase + (uint32_t)j, seed, thresh24)) ? p * inv_keep : 0.f;
  }
  sum = wave_sum_f(sum);
  if (lse && live && lane == 0) lse[((size_t)b * H + h) * Tq + t] = nk > 0 ? mx + __logf(sum) : 0.f;
  __syncthreads();                  // the probabilities other lanes wrote are read below
  // context: lane l owns output feature l (lanes >= DH idle); a key's value row is one coalesced 2 * DH-byte read
  float acc = 0.f;
  if (lane < DH) {
    const bf16* vb = kb + d + lane;
    int j = 0;
    for (; j + 4 <= nk; j += 4) {
      const float v0 = (float)vb[(size_t)j * ld_kv], v1 = (float)vb[(size_t)(j + 1) * ld_kv];
      const float v2 = (float)vb[(size_t)(j + 2) * ld_kv], v3 = (float)vb[(size_t)(j + 3) * ld_kv];
      acc += sc[wv][j] * v0 + sc[wv][j + 1] * v1 + sc[wv][j + 2] * v2 + sc[wv][j + 3] * v3;
    }
    for (; j < nk; ++j) acc += sc[wv][j] * (float)vb[(size_t)j * ld_kv];
    if (live) out[lane] = (bf16)(nk > 0 ? acc / sum : 0.f);
  }
}

// Backward of the rows attention (training of the causal / cross-attention sub-layers: short target sequences, so a wave
// per (sequence, head, query row) like the forward).  Recomputes p_j = exp(q k_j - lse), dP_j = dO v_j (through the
// dropout), D = sum_j p_j dP_j, dS_j = p_j (dP_j - D);  dq = qscale * sum_j dS_j k_j  (the gradient of the UNSCALED
// projection, as m3p_attn_bwd returns it), and adds dS_j q / drop(p_j) dO into the fp32 key / value gradients
// dkv [B, Lk, 2 H DH] with one 64-float atomic per (key, head) and wave.
template <int DH>
__global__ __launch_bounds__(256) void attn_rows_bwd_kernel(const bf16* __restrict__ q, int ld_q, const bf16* __restrict__ kv,
                                                           long long kv_bstride, int ld_kv, const int32_t* __restrict__ klen,
                                                           const bf16* __restrict__ dctx, const float* __restrict__ lse,
                                                           bf16* __restrict__ dq, int ld_dq, float* __restrict__ dkv, int B, int Tq,
                                                           int H, int Lk, int causal, int pos0, float qscale, uint32_t seed,
                                                           uint32_t thresh24, float inv_keep) {
  __shared__ float s_ds[4][QA_MAX_KEYS];      // dP_j, then dS_j
  __shared__ float s_p[4][QA_MAX_KEYS];       // p_j
  __shared__ float s_pd[4][QA_MAX_KEYS];      // drop(p_j)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  long long item = (long long)blockIdx.x * 4 + wv;
  const bool live = item < (long long)B * Tq * H;
  if (!live) item = 0;
  const int h = (int)(item % H);
  const int t = (int)((item / H) % Tq);
  const int b = (int)(item / ((long long)H * Tq));
  int nk = klen ? min(klen[b], Lk) : Lk;
  if (causal) nk = min(nk, pos0 + t + 1);
  const int d = H * DH;
  float qf[DH], gf[DH];
  const bf16* qp = q + ((size_t)b * Tq + t) * ld_q + h * DH;
  const bf16* gp = dctx + ((size_t)b * Tq + t) * d + h * DH;
#pragma unroll
  for (int c = 0; c < DH / 8; ++c) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(qp + c * 8);
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(gp + c * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) { qf[c * 8 + e] = (float)v[e]; gf[c * 8 + e] = (float)g[e]; }
  }
  const bf16* kb = kv + (size_t)b * kv_bstride + h * DH;
  const float l = lse[((size_t)b * H + h) * Tq + t];
  const uint32_t rbase = (uint32_t)(((size_t)(b * H + h) * Tq + t) * Lk);
  float dsum = 0.f;
  for (int j = lane; j < nk; j += 64) {
    const bf16* kp = kb + (size_t)j * ld_kv;
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int c = 0; c < DH / 8; ++c) {
      const bf16x8 kk = *reinterpret_cast<const bf16x8*>(kp + c * 8);
      const bf16x8 vv = *reinterpret_cast<const bf16x8*>(kp + d + c * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s += qf[c * 8 + e] * (float)kk[e]; dp += gf[c * 8 + e] * (float)vv[e]; }
    }
    const float p = __expf(s - l);
    const float keepf = (thresh24 == 0 || m3p_keep(rbase + (uint32_t)j, seed, thresh24)) ? inv_keep : 0.f;
    dp *= keepf;                       // gradient wrt p_j through the dropout
    s_p[wv][j] = p;
    s_pd[wv][j] = p * keepf;
    s_ds[wv][j] = dp;
    dsum += p * dp;
  }
  dsum = wave_sum_f(dsum);
  for (int j = lane; j < nk; j += 64) s_ds[wv][j] = s_p[wv][j] * (s_ds[wv][j] - dsum);
  __syncthreads();
  if (lane < DH && live) {
    const float ql = (float)qp[lane], gl = (float)gp[lane];
    float acc = 0.f;
    float* dkp = dkv + ((size_t)b * Lk) * (2 * d) + h * DH + lane;
    for (int j = 0; j < nk; ++j) {
      const float ds = s_ds[wv][j];
      acc += ds * (float)kb[(size_t)j * ld_kv + lane];
      unsafeAtomicAdd(dkp + (size_t)j * (2 * d), ds * ql);
      unsafeAtomicAdd(dkp + (size_t)j * (2 * d) + d, s_pd[wv][j] * gl);
    }
    dq[((size_t)b * Tq + t) * ld_dq + h * DH + lane] = (bf16)(acc * qscale);
  }
}

}  // namespace

extern "C" {

int m3p_attn_rows_fwd(const void* q, int ld_q, const void* kv, long long kv_bstride, int ld_kv, const int32_t* klen, void* ctx,
                      float* lse, int B, int Tq, int H, int dh, int Lk, int causal, int pos0, uint32_t seed, uint32_t thresh24,
                      float inv_keep, void* stream) {
  if (B <= 0 || Tq <= 0 || H <= 0 || Lk <= 0 || Lk > QA_MAX_KEYS || (dh != 32 && dh != 64)) return M3P_EINVAL;
  if ((ld_q % 8) != 0 || (ld_kv % 8) != 0 || (kv_bstride % 8) != 0 || ((uintptr_t)q & 15) || ((uintptr_t)kv & 15)) return M3P_EINVAL;
  if (ld_kv < 2 * H * dh || ld_q < H * dh) return M3P_EINVAL;
  if ((unsigned long long)B * H * Tq * Lk >= (1ull << 32)) return M3P_EINVAL;       // 32-bit dropout stream index
  const long long items = (long long)B * Tq * H;
  const dim3 grid((unsigned)((items + 3) / 4)), block(256);
  if (dh == 64)
    hipLaunchKernelGGL(attn_query_fwd_kernel<64>, grid, block, 0, (hipStream_t)stream, (const bf16*)q, ld_q, (const bf16*)kv,
                       kv_bstride, ld_kv, klen, (bf16*)ctx, lse, B, Tq, H, Lk, causal, pos0, seed, thresh24, inv_keep);
  else
    hipLaunchKernelGGL(attn_query_fwd_kernel<32>, grid, block, 0, (hipStream_t)stream, (const bf16*)q, ld_q, (const bf16*)kv,
                       kv_bstride, ld_kv, klen, (bf16*)ctx, lse, B, Tq, H, Lk, causal, pos0, seed, thresh24, inv_keep);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_attn_query_fwd(const void* q, int ld_q, const void* kv, long long kv_bstride, int ld_kv, const int32_t* klen, void* ctx,
                       int B, int Tq, int H, int dh, int Lk, int causal, int pos0, void* stream) {
  return m3p_attn_rows_fwd(q, ld_q, kv, kv_bstride, ld_kv, klen, ctx, nullptr, B, Tq, H, dh, Lk, causal, pos0, 0, 0, 1.f, stream);
}

int m3p_attn_rows_bwd(const void* q, int ld_q, const void* kv, long long kv_bstride, int ld_kv, const int32_t* klen,
                      const void* dctx, const float* lse, void* dq, int ld_dq, float* dkv, int B, int Tq, int H, int dh, int Lk,
                      int causal, int pos0, float qscale, uint32_t seed, uint32_t thresh24, float inv_keep, void* stream) {
  if (B <= 0 || Tq <= 0 || H <= 0 || Lk <= 0 || Lk > QA_MAX_KEYS || (dh != 32 && dh != 64) || !lse || !dkv) return M3P_EINVAL;
  if ((ld_q % 8) != 0 || (ld_kv % 8) != 0 || (kv_bstride % 8) != 0 || ((uintptr_t)q & 15) || ((uintptr_t)kv & 15) ||
      ((uintptr_t)dctx & 15))
    return M3P_EINVAL;
  if (ld_kv < 2 * H * dh || ld_q < H * dh || ld_dq < H * dh) return M3P_EINVAL;
  const long long items = (long long)B * Tq * H;
  const dim3 grid((unsigned)((items + 3) / 4)), block(256);
  if (dh == 64)
    hipLaunchKernelGGL(attn_rows_bwd_kernel<64>, grid, block, 0, (hipStream_t)stream, (const bf16*)q, ld_q, (const bf16*)kv,
                       kv_bstride, ld_kv, klen, (const bf16*)dctx, lse, (bf16*)dq, ld_dq, dkv, B, Tq, H, Lk, causal, pos0, qscale,
                       seed, thresh24, inv_keep);
  else
    hipLaunchKernelGGL(attn_rows_bwd_kernel<32>, grid, block, 0, (hipStream_t)stream, (const bf16*)q, ld_q, (const bf16*)kv,
                       kv_bstride, ld_kv, klen, (const bf16*)dctx, lse, (bf16*)dq, ld_dq, dkv, B, Tq, H, Lk, causal, pos0, qscale,
                       seed, thresh24, inv_keep);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

}  // extern "C"

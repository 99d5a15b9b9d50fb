// LayerNorm forward / backward for [rows, d] bf16 activations on gfx950.
// HBM-bound: one wave per row, 8-byte (4 x bf16) coalesced accesses, the whole row lives
// in registers between the two reduction passes (mean, then centred variance — the same
// two-pass arithmetic as ATen's CPU kernel), wave64 xor-shuffle reductions, fp32 stats.
#include "common.hpp"
#include "../../include/m3p_hip.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#ifndef M3P_LN_BWD_BLOCKS
#define M3P_LN_BWD_BLOCKS 512
#endif

namespace {

// NI = ceil(d / 256): 4-element chunks per lane
template <int NI>
__global__ __launch_bounds__(256)
void ln_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                   const uint8_t* __restrict__ rowmask, bf16* __restrict__ y, float* __restrict__ mean,
                   float* __restrict__ rstd, int rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int nchunk = d >> 2;
  const float inv_d = 1.0f / (float)d;

  f32x4 g[NI], b[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) { g[i] = Vec4<float>::load(gamma + 4 * c); b[i] = Vec4<float>::load(beta + 4 * c); }
  }
  for (int r = wave; r < rows; r += nwaves) {
    const bf16* xr = x + (size_t)r * d;
    f32x4 v[NI];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane + 64 * i;
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < nchunk) v[i] = Vec4<bf16>::load(xr + 4 * c);
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mu = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        const f32x4 t = v[i] - mu;
        q += (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]);
      }
    }
    const float rs = 1.0f / sqrtf(wave_sum(q) * inv_d + eps);
    const float mk = rowmask ? (rowmask[r] ? 1.f : 0.f) : 1.f;
    bf16* yr = y + (size_t)r * d;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        f32x4 o = ((v[i] - mu) * rs * g[i] + b[i]) * mk;
        Vec4<bf16>::store(yr + 4 * c, o);
      }
    }
    if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
  }
}

template <int NI>
__global__ __launch_bounds__(256)
void ln_bwd_kernel(const bf16* __restrict__ dy_a, const bf16* __restrict__ dy_b, const bf16* __restrict__ x,
                   const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
                   const uint8_t* __restrict__ rowmask, bf16* __restrict__ dx, bf16* __restrict__ dx_drop,
                   float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias_drop,
                   int rows, int d, uint32_t seed, uint32_t thresh24, float inv_keep) {
  __shared__ float red[4][NI * 256];   // [wave][column], reused per quantity
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int nchunk = d >> 2;
  const float inv_d = 1.0f / (float)d;

  f32x4 g[NI], acc_g[NI], acc_b[NI], acc_d[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = lane + 64 * i;
    g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < nchunk) g[i] = Vec4<float>::load(gamma + 4 * c);
    acc_g[i] = acc_b[i] = acc_d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int r = wave; r < rows; r += nwaves) {
    const size_t ro = (size_t)r * d;
    const float mu = mean[r], rs = rstd[r];
    const float mk = rowmask ? (rowmask[r] ? 1.f : 0.f) : 1.f;
    f32x4 dyv[NI], xh[NI];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane + 64 * i;
      dyv[i] = xh[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < nchunk) {
        dyv[i] = Vec4<bf16>::load(dy_a + ro + 4 * c);
        if (dy_b) dyv[i] += Vec4<bf16>::load(dy_b + ro + 4 * c);
        dyv[i] *= mk;
        xh[i] = (Vec4<bf16>::load(x + ro + 4 * c) - mu) * rs;
        const f32x4 gd = dyv[i] * g[i];
        s1 += (gd[0] + gd[1]) + (gd[2] + gd[3]);
        const f32x4 gx = gd * xh[i];
        s2 += (gx[0] + gx[1]) + (gx[2] + gx[3]);
        acc_g[i] += dyv[i] * xh[i];
        acc_b[i] += dyv[i];
      }
    }
    const float c1 = wave_sum(s1) * inv_d, c2 = wave_sum(s2) * inv_d;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        const f32x4 o = (dyv[i] * g[i] - c1 - xh[i] * c2) * rs;
        Vec4<bf16>::store(dx + ro + 4 * c, o);
        if (dx_drop || dbias_drop) {
          // round to bf16 first: the wgrad/dgrad GEMMs consume the bf16 tensor, so the
          // bias gradient is the column sum of exactly those values
          bf16x4 ob = bf16x4{(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
          f32x4 od;
          const uint32_t base = (uint32_t)r * (uint32_t)d + 4u * (uint32_t)c;      // (d % 4 == 0: even)
          bool kp[4] = {true, true, true, true};
          if (thresh24) m3p_keep_even<4>(base, seed, thresh24, kp);
#pragma unroll
          for (int e = 0; e < 4; ++e) od[e] = kp[e] ? (float)ob[e] * (thresh24 ? inv_keep : 1.f) : 0.f;
          bf16x4 odb = bf16x4{(bf16)od[0], (bf16)od[1], (bf16)od[2], (bf16)od[3]};
          if (dx_drop) *reinterpret_cast<bf16x4*>(dx_drop + ro + 4 * c) = odb;
          acc_d[i] += f32x4{(float)odb[0], (float)odb[1], (float)odb[2], (float)odb[3]};
        }
      }
    }
  }
  // block reduction over the 4 waves (one quantity at a time through LDS), then one
  // atomic per column per block
  float* const outs[3] = {dgamma, dbeta, dbias_drop};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    if (outs[q] == nullptr) continue;   // block-uniform
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int col = 4 * (lane + 64 * i);
      const f32x4 a = (q == 0) ? acc_g[i] : (q == 1) ? acc_b[i] : acc_d[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wib][col + e] = a[e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += blockDim.x)
      unsafeAtomicAdd(outs[q] + c, (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
    __syncthreads();
  }
}

// Backward, wide variant for d = 256 * NC (768, 1024): HALF a wave per row with 16-byte
// accesses (8 bf16 per lane per chunk), so a wave keeps two rows and twice the bytes in flight;
// row reductions are xor-shuffles over 32 lanes.  Same arithmetic as ln_bwd_kernel.
template <int NC>
__global__ __launch_bounds__(256)
void ln_bwd_hw_kernel(const bf16* __restrict__ dy_a, const bf16* __restrict__ dy_b, const bf16* __restrict__ x,
                      const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
                      const uint8_t* __restrict__ rowmask, bf16* __restrict__ dx, bf16* __restrict__ dx_drop,
                      float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias_drop,
                      int rows, uint32_t seed, uint32_t thresh24, float inv_keep) {
  constexpr int D = 256 * NC;
  __shared__ float red[8][D];
  const int sub = threadIdx.x & 31, hw = threadIdx.x >> 5;   // 8 half-waves per block
  const int ghw = blockIdx.x * 8 + hw, nhw = gridDim.x * 8;
  const float inv_d = 1.0f / (float)D;
  float g[NC][8], acc_g[NC][8], acc_b[NC][8], acc_d[NC][8];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = (sub + 32 * i) * 8;
    const f32x4 g0 = Vec4<float>::load(gamma + c), g1 = Vec4<float>::load(gamma + c + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { g[i][e] = g0[e]; g[i][4 + e] = g1[e]; }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc_g[i][e] = acc_b[i][e] = acc_d[i][e] = 0.f;
  }
  auto hsum = [](float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  for (int r = ghw; r < rows; r += nhw) {
    const size_t ro = (size_t)r * D;
    const float mu = mean[r], rs = rstd[r];
    const float mk = rowmask ? (rowmask[r] ? 1.f : 0.f) : 1.f;
    float dyv[NC][8], xh[NC][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = (sub + 32 * i) * 8;
      bf16x8 a = *reinterpret_cast<const bf16x8*>(dy_a + ro + c);
      const bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + ro + c);
      bf16x8 b2;
      if (dy_b) b2 = *reinterpret_cast<const bf16x8*>(dy_b + ro + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float dv = (float)a[e];
        if (dy_b) dv += (float)b2[e];
        dv *= mk;
        const float xe = ((float)xv[e] - mu) * rs;
        dyv[i][e] = dv; xh[i][e] = xe;
        const float gd = dv * g[i][e];
        s1 += gd; s2 += gd * xe;
        acc_g[i][e] += dv * xe;
        acc_b[i][e] += dv;
      }
    }
    const float c1 = hsum(s1) * inv_d, c2 = hsum(s2) * inv_d;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = (sub + 32 * i) * 8;
      bf16x8 ob, odb;
#pragma unroll
      for (int e = 0; e < 8; ++e) ob[e] = (bf16)((dyv[i][e] * g[i][e] - c1 - xh[i][e] * c2) * rs);
      *reinterpret_cast<bf16x8*>(dx + ro + c) = ob;
      if (dx_drop || dbias_drop) {
        const uint32_t base = (uint32_t)r * (uint32_t)D + (uint32_t)c;
        bool kp[8] = {true, true, true, true, true, true, true, true};
        if (thresh24) m3p_keep_even<8>(base, seed, thresh24, kp);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          odb[e] = (bf16)(kp[e] ? (float)ob[e] * (thresh24 ? inv_keep : 1.f) : 0.f);
          acc_d[i][e] += (float)odb[e];
        }
        if (dx_drop) *reinterpret_cast<bf16x8*>(dx_drop + ro + c) = odb;
      }
    }
  }
  float* const outs[3] = {dgamma, dbeta, dbias_drop};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    if (outs[q] == nullptr) continue;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = (sub + 32 * i) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) red[hw][c + e] = (q == 0) ? acc_g[i][e] : (q == 1) ? acc_b[i][e] : acc_d[i][e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += red[w][c];
      unsafeAtomicAdd(outs[q] + c, t);
    }
    __syncthreads();
  }
}

inline int ln_ni(int d) { return (d + 255) / 256; }

}  // namespace

extern "C" {

int m3p_layernorm_fwd(const void* x, const float* gamma, const float* beta, const uint8_t* rowmask, void* y,
                      float* mean, float* rstd, int rows, int d, float eps, void* stream) {
  if (rows <= 0 || d <= 0 || (d % 4) != 0 || d > 2048) return M3P_EINVAL;
  if (((uintptr_t)x & 7) || ((uintptr_t)y & 7) || ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15)) return M3P_EINVAL;
  const int blocks = min((rows + 3) / 4, 4096);
  hipStream_t st = (hipStream_t)stream;
#define M3P_LN_FWD(NI)                                                                                  \
  hipLaunchKernelGGL(ln_fwd_kernel<NI>, dim3(blocks), dim3(256), 0, st, (const bf16*)x, gamma, beta,    \
                     rowmask, (bf16*)y, mean, rstd, rows, d, eps)
  switch (ln_ni(d)) {
    case 1: M3P_LN_FWD(1); break;
    case 2: M3P_LN_FWD(2); break;
    case 3: M3P_LN_FWD(3); break;
    case 4: M3P_LN_FWD(4); break;
    default: M3P_LN_FWD(8); break;
  }
#undef M3P_LN_FWD
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_layernorm_bwd(const void* dy_a, const void* dy_b, const void* x, const float* gamma, const float* mean,
                      const float* rstd, const uint8_t* rowmask, void* dx, void* dx_drop, float* dgamma,
                      float* dbeta, float* dbias_drop, int rows, int d, uint32_t seed, uint32_t thresh24,
                      float inv_keep, void* stream) {
  if (rows <= 0 || d <= 0 || (d % 4) != 0 || d > 2048) return M3P_EINVAL;
  if (!dy_a || !x || !dx || !dgamma || !dbeta) return M3P_EINVAL;
  if (((uintptr_t)x & 7) || ((uintptr_t)dy_a & 7) || ((uintptr_t)dx & 7) || ((uintptr_t)gamma & 15)) return M3P_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if ((d == 768 || d == 1024) && !(((uintptr_t)x | (uintptr_t)dy_a | (uintptr_t)dx | (uintptr_t)dy_b | (uintptr_t)dx_drop) & 15)) {
    const int blocks_hw = min((rows + 7) / 8, M3P_LN_BWD_BLOCKS);   // fewer blocks: the 3*d end-of-block atomics hit the same 2304 addresses
    if (d == 768)
      hipLaunchKernelGGL(ln_bwd_hw_kernel<3>, dim3(blocks_hw), dim3(256), 0, st, (const bf16*)dy_a, (const bf16*)dy_b,
                         (const bf16*)x, gamma, mean, rstd, rowmask, (bf16*)dx, (bf16*)dx_drop, dgamma, dbeta, dbias_drop,
                         rows, seed, thresh24, inv_keep);
    else
      hipLaunchKernelGGL(ln_bwd_hw_kernel<4>, dim3(blocks_hw), dim3(256), 0, st, (const bf16*)dy_a, (const bf16*)dy_b,
                         (const bf16*)x, gamma, mean, rstd, rowmask, (bf16*)dx, (bf16*)dx_drop, dgamma, dbeta, dbias_drop,
                         rows, seed, thresh24, inv_keep);
    M3P_CHECK_LAUNCH();
    return M3P_OK;
  }
  const int blocks = min((rows + 3) / 4, 512);
#define M3P_LN_BWD(NI)                                                                                     \
  hipLaunchKernelGGL(ln_bwd_kernel<NI>, dim3(blocks), dim3(256), 0, st, (const bf16*)dy_a,                 \
                     (const bf16*)dy_b, (const bf16*)x, gamma, mean, rstd, rowmask, (bf16*)dx,              \
                     (bf16*)dx_drop, dgamma, dbeta, dbias_drop, rows, d, seed, thresh24, inv_keep)
  switch (ln_ni(d)) {
    case 1: M3P_LN_BWD(1); break;
    case 2: M3P_LN_BWD(2); break;
    case 3: M3P_LN_BWD(3); break;
    case 4: M3P_LN_BWD(4); break;
    default: M3P_LN_BWD(8); break;
  }
#undef M3P_LN_BWD
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

}  // extern "C"

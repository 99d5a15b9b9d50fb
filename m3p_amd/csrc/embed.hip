// Input assembly of TransformerModel.jointfwd (M3P/src/model/transformer.py:901-943) and
// its backward, as row kernels (one wave64 per [d]-row, 8-byte bf16 accesses, fp32 math):
//
//   image rows  (s <  R): e = W_img x + b   (GEMM, done by the caller)  + W_loc loc + b_loc
//                          i = dropout(LN_img(e))                         (:257-268)
//                          z = (i + Pos[s]) * mask                        (:929-940)
//   token rows  (s >= R): z = (Emb[x] + Pos[s]) * mask                    (:913, :936-940)
//   all rows            : h = dropout(LN_emb(z))                          (:942-943)
//
// Layout: internal activations are batch-major rows m = b*S + s; the image projection
// and its gradient keep the caller's sequence-major row order r*B + b (the order of x_img),
// so no transpose of the 2048-d region features is ever materialised.
#include "common.hpp"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#ifndef M3P_EMB_BWD_BLOCKS
#define M3P_EMB_BWD_BLOCKS 512
#endif

namespace {

constexpr float kEps = 1e-12f;

template <int NI>
__device__ __forceinline__ void row_ln(const f32x4 (&v)[NI], int nchunk, int lane, float inv_d, float& mu, float& rs) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);   // lanes beyond nchunk hold zeros
  mu = wave_sum(s) * inv_d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    if (lane + 64 * i < nchunk) {
      const f32x4 t = v[i] - mu;
      q += (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]);
    }
  }
  rs = 1.0f / sqrtf(wave_sum(q) * inv_d + kEps);
}

struct EmbedFwdArgs {
  const int64_t* tok;       // (T,B) token ids, sequence-major
  const bf16* emb;          // [V,d] bf16 copy of embeddings.weight
  const float* pos;         // [>=S,d] position_embeddings.weight (fp32 master)
  const bf16* img_proj;     // [R*B,d] rows r*B+b: W_img x + b_img
  const float* loc;         // (R,B,5)
  const float* w_loc;       // [d,5]
  const float* b_loc;       // [d]
  const float* g_img; const float* be_img;   // image LayerNorm
  const float* g_emb; const float* be_emb;   // layer_norm_emb
  const int32_t* totlen;    // [B] valid prefix length (len_img + len_txt)
  bf16* h;                  // [B*S,d] encoder input
  bf16* z;                  // [B*S,d] LN_emb input (saved)
  float* mean_emb; float* rstd_emb;   // [B*S]
  bf16* e;                  // [R*B,d] LN_img input (saved)
  float* mean_img; float* rstd_img;   // [R*B]
  int B, T, R, d;
  uint32_t seed_img, seed_emb, thresh24; float inv_keep;
  // AoA refiner hook (jointfwd's refine_image=True, transformer.py:905-906): the image rows leave after their
  // LayerNorm + dropout, go through the refiner, and come back in place of the computed ones.
  //   img_mode 0: none.  1: image rows only (T = 0): write them to img_rows [B*R, d] (row b*R + r) and stop.
  //   2: take the image rows from img_rows instead of computing them (e / mean_img / rstd_img untouched).
  bf16* img_rows; int img_mode;
};

template <int NI>
__global__ __launch_bounds__(256) void embed_fwd_kernel(EmbedFwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int S = a.R + a.T, d = a.d, nchunk = d >> 2;
  const float inv_d = 1.0f / (float)d;
  for (int m = wave; m < a.B * S; m += nwaves) {
    const int b = m / S, s = m - b * S;
    const float mk = (a.img_mode == 1 || s < a.totlen[b]) ? 1.f : 0.f;
    f32x4 v[NI];
    if (s < a.R && a.img_mode == 2) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < nchunk) v[i] = Vec4<bf16>::load(a.img_rows + ((size_t)b * a.R + s) * d + 4 * c);
      }
    } else if (s < a.R) {
      const int ri = s * a.B + b;
      float lc[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) lc[k] = a.loc[(size_t)ri * 5 + k];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < nchunk) {
          f32x4 t = Vec4<bf16>::load(a.img_proj + (size_t)ri * d + 4 * c) + Vec4<float>::load(a.b_loc + 4 * c);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float* w = a.w_loc + (size_t)(4 * c + j) * 5;
            t[j] += w[0] * lc[0] + w[1] * lc[1] + w[2] * lc[2] + w[3] * lc[3] + w[4] * lc[4];
          }
          // keep what backward will re-read: the bf16-rounded pre-LN value
          t = round_bf16(t);
          Vec4<bf16>::store(a.e + (size_t)ri * d + 4 * c, t);
          v[i] = t;
        }
      }
      float mu, rs;
      row_ln<NI>(v, nchunk, lane, inv_d, mu, rs);
      if (lane == 0) { a.mean_img[ri] = mu; a.rstd_img[ri] = rs; }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
          f32x4 o = (v[i] - mu) * rs * Vec4<float>::load(a.g_img + 4 * c) + Vec4<float>::load(a.be_img + 4 * c);
          if (a.thresh24) {
            const uint32_t base = (uint32_t)ri * (uint32_t)d + 4u * c;
            { bool kp[4]; m3p_keep_even<4>(base, a.seed_img, a.thresh24, kp);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = kp[j] ? o[j] * a.inv_keep : 0.f; }
          }
          v[i] = o;
          if (a.img_mode == 1) Vec4<bf16>::store(a.img_rows + ((size_t)b * a.R + s) * d + 4 * c, o);
        }
      }
      if (a.img_mode == 1) continue;
    } else {
      const int64_t id = a.tok[(size_t)(s - a.R) * a.B + b];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < nchunk) v[i] = Vec4<bf16>::load(a.emb + (size_t)id * d + 4 * c);
      }
    }
    // + position, * mask, store z (bf16), LN_emb on the stored values
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        const f32x4 t = round_bf16((v[i] + Vec4<float>::load(a.pos + (size_t)s * d + 4 * c)) * mk);
        Vec4<bf16>::store(a.z + (size_t)m * d + 4 * c, t);
        v[i] = t;
      }
    }
    float mu, rs;
    row_ln<NI>(v, nchunk, lane, inv_d, mu, rs);
    if (lane == 0) { a.mean_emb[m] = mu; a.rstd_emb[m] = rs; }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        f32x4 o = (v[i] - mu) * rs * Vec4<float>::load(a.g_emb + 4 * c) + Vec4<float>::load(a.be_emb + 4 * c);
        if (a.thresh24) {
          const uint32_t base = (uint32_t)m * (uint32_t)d + 4u * c;
          { bool kp[4]; m3p_keep_even<4>(base, a.seed_emb, a.thresh24, kp);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = kp[j] ? o[j] * a.inv_keep : 0.f; }
        }
        Vec4<bf16>::store(a.h + (size_t)m * d + 4 * c, o);
      }
    }
  }
}

// ------------------------------ backward ------------------------------
struct EmbedBwdArgs {
  const bf16* dh;           // [B*S,d] gradient wrt encoder input
  const bf16* z; const float* mean_emb; const float* rstd_emb; const float* g_emb;
  const bf16* e; const float* mean_img; const float* rstd_img; const float* g_img;
  const int64_t* tok; const int32_t* totlen; const float* loc;
  bf16* dz;                 // [B*S,d] scratch: gradient wrt z (post-mask)
  bf16* de;                 // [R*B,d] gradient wrt the image projection (rows r*B+b)
  float* d_g_emb; float* d_be_emb; float* d_pos;   // accumulated (atomics)
  float* d_emb;             // [V,d] embeddings.weight gradient (scatter-add); unused when d_tok_rows is set
  bf16* d_tok_rows;         // optional [T*B,d]: the token rows' gradients (row t*B+b; zero rows for pad / masked
                            // tokens) INSTEAD of the scatter-add - the data-parallel sparse exchange applies them
  float* d_g_img; float* d_be_img; float* d_b_img; float* d_b_loc; float* d_w_loc;   // [d],[d],[d],[d],[d,5]
  int B, T, R, d, pad_index, bsplit;
  uint32_t seed_img, seed_emb, thresh24; float inv_keep;
};

template <int NI>
__device__ __forceinline__ void block_reduce_atomic(float (*red)[NI * 256], const f32x4 (&acc)[NI], float* out, int d,
                                                    int lane, int wib) {
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wib][4 * (lane + 64 * i) + j] = acc[i][j];
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += blockDim.x)
    unsafeAtomicAdd(out + c, (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
  __syncthreads();
}

// A: all rows.  grid = S * bsplit blocks; block (s, part) walks its share of the batch.
template <int NI>
__global__ __launch_bounds__(256) void embed_bwd_rows_kernel(EmbedBwdArgs a) {
  __shared__ float red[4][NI * 256];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int S = a.R + a.T, d = a.d, nchunk = d >> 2;
  const float inv_d = 1.0f / (float)d;
  const int s = blockIdx.x / a.bsplit, part = blockIdx.x - s * a.bsplit;
  f32x4 g[NI], acc_g[NI], acc_b[NI], acc_p[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = lane + 64 * i;
    g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < nchunk) g[i] = Vec4<float>::load(a.g_emb + 4 * c);
    acc_g[i] = acc_b[i] = acc_p[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int b = part * 4 + wib; b < a.B; b += 4 * a.bsplit) {
    const size_t m = (size_t)b * S + s;
    const float mu = a.mean_emb[m], rs = a.rstd_emb[m];
    const float mk = (s < a.totlen[b]) ? 1.f : 0.f;
    f32x4 dy[NI], xh[NI];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane + 64 * i;
      dy[i] = xh[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < nchunk) {
        dy[i] = Vec4<bf16>::load(a.dh + m * d + 4 * c);
        if (a.thresh24) {
          const uint32_t base = (uint32_t)m * (uint32_t)d + 4u * c;
          { bool kp[4]; m3p_keep_even<4>(base, a.seed_emb, a.thresh24, kp);
#pragma unroll
          for (int j = 0; j < 4; ++j) dy[i][j] = kp[j] ? dy[i][j] * a.inv_keep : 0.f; }
        }
        xh[i] = (Vec4<bf16>::load(a.z + m * d + 4 * c) - mu) * rs;
        const f32x4 gd = dy[i] * g[i];
        s1 += (gd[0] + gd[1]) + (gd[2] + gd[3]);
        const f32x4 gx = gd * xh[i];
        s2 += (gx[0] + gx[1]) + (gx[2] + gx[3]);
        acc_g[i] += dy[i] * xh[i];
        acc_b[i] += dy[i];
      }
    }
    const float c1 = wave_sum(s1) * inv_d, c2 = wave_sum(s2) * inv_d;
    int64_t id = -1;
    if (s >= a.R) id = a.tok[(size_t)(s - a.R) * a.B + b];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        const f32x4 o = (dy[i] * g[i] - c1 - xh[i] * c2) * rs * mk;
        acc_p[i] += o;
        if (s < a.R) {
          Vec4<bf16>::store(a.dz + m * d + 4 * c, o);
        } else {
          *reinterpret_cast<f32x4*>(&red[wib][4 * c]) = o;       // this wave's row, re-read below dword-contiguous
        }
      }
    }
    if (s >= a.R && a.d_tok_rows) {
      const bool live = id != a.pad_index && mk != 0.f;
      bf16* dr = a.d_tok_rows + ((size_t)(s - a.R) * a.B + b) * d;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
          f32x4 o = *reinterpret_cast<const f32x4*>(&red[wib][4 * c]);
          if (!live) o = f32x4{0.f, 0.f, 0.f, 0.f};
          Vec4<bf16>::store(dr + 4 * c, o);
        }
      }
    } else if (s >= a.R && id != a.pad_index && mk != 0.f) {
      // scatter-add into the embedding gradient: one atomic instruction = 64 CONSECUTIVE floats (two
      // full 128-B lines) instead of 64 float4-strided ones (eight partial lines)
      float* de = a.d_emb + (size_t)id * d;
#pragma unroll
      for (int k = 0; k < 4 * NI; ++k) {
        const int e = lane + 64 * k;
        if (e < d) unsafeAtomicAdd(de + e, red[wib][e]);
      }
    }
  }
  __syncthreads();     // the row buffers double as the block-reduction scratch below
  block_reduce_atomic<NI>(red, acc_g, a.d_g_emb, d, lane, wib);
  block_reduce_atomic<NI>(red, acc_b, a.d_be_emb, d, lane, wib);
  block_reduce_atomic<NI>(red, acc_p, a.d_pos + (size_t)s * d, d, lane, wib);
}

// C: image rows.  grid = R * bsplit; needs dz of the image rows from kernel A.
template <int NI>
__global__ __launch_bounds__(256) void embed_bwd_img_kernel(EmbedBwdArgs a) {
  __shared__ float red[4][NI * 256];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int S = a.R + a.T, d = a.d, nchunk = d >> 2;
  const float inv_d = 1.0f / (float)d;
  const int r = blockIdx.x / a.bsplit, part = blockIdx.x - r * a.bsplit;
  f32x4 g[NI], acc_g[NI], acc_b[NI], acc_e[NI], acc_w[5][NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = lane + 64 * i;
    g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < nchunk) g[i] = Vec4<float>::load(a.g_img + 4 * c);
    acc_g[i] = acc_b[i] = acc_e[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 5; ++k) acc_w[k][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int b = part * 4 + wib; b < a.B; b += 4 * a.bsplit) {
    const size_t m = (size_t)b * S + r;
    const size_t ri = (size_t)r * a.B + b;
    const float mu = a.mean_img[ri], rs = a.rstd_img[ri];
    float lc[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) lc[k] = a.loc[ri * 5 + k];
    f32x4 dy[NI], xh[NI];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane + 64 * i;
      dy[i] = xh[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < nchunk) {
        dy[i] = Vec4<bf16>::load(a.dz + m * d + 4 * c);
        if (a.thresh24) {
          const uint32_t base = (uint32_t)ri * (uint32_t)d + 4u * c;
          { bool kp[4]; m3p_keep_even<4>(base, a.seed_img, a.thresh24, kp);
#pragma unroll
          for (int j = 0; j < 4; ++j) dy[i][j] = kp[j] ? dy[i][j] * a.inv_keep : 0.f; }
        }
        xh[i] = (Vec4<bf16>::load(a.e + ri * d + 4 * c) - mu) * rs;
        const f32x4 gd = dy[i] * g[i];
        s1 += (gd[0] + gd[1]) + (gd[2] + gd[3]);
        const f32x4 gx = gd * xh[i];
        s2 += (gx[0] + gx[1]) + (gx[2] + gx[3]);
        acc_g[i] += dy[i] * xh[i];
        acc_b[i] += dy[i];
      }
    }
    const float c1 = wave_sum(s1) * inv_d, c2 = wave_sum(s2) * inv_d;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        const f32x4 ob = round_bf16((dy[i] * g[i] - c1 - xh[i] * c2) * rs);   // what the wgrad GEMM will see
        Vec4<bf16>::store(a.de + ri * d + 4 * c, ob);
        acc_e[i] += ob;
#pragma unroll
        for (int k = 0; k < 5; ++k) acc_w[k][i] += ob * lc[k];
      }
    }
  }
  block_reduce_atomic<NI>(red, acc_g, a.d_g_img, d, lane, wib);
  block_reduce_atomic<NI>(red, acc_b, a.d_be_img, d, lane, wib);
  // the image-projection bias and the location bias see the same gradient: colsum(de)
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wib][4 * (lane + 64 * i) + j] = acc_e[i][j];
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    const float v = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    unsafeAtomicAdd(a.d_b_img + c, v);
    unsafeAtomicAdd(a.d_b_loc + c, v);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 5; ++k) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[wib][4 * (lane + 64 * i) + j] = acc_w[k][i][j];
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += blockDim.x)
      unsafeAtomicAdd(a.d_w_loc + (size_t)c * 5 + k, (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    Vec4<bf16>::store(dst + 4 * i, Vec4<float>::load(src + 4 * i));
}

inline int ni_of(int d) { return (d + 255) / 256; }

int launch_embed_fwd(const EmbedFwdArgs& a, void* stream);
}  // namespace

extern "C" {

int m3p_cast_f32_bf16(const float* src, void* dst, long long n, void* stream) {
  if (n <= 0 || (n % 4) != 0 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return M3P_EINVAL;
  const size_t n4 = (size_t)n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, n4);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_embed_assemble_fwd(const int64_t* tok, const void* emb_bf16, const float* pos, const void* img_proj,
                           const float* loc, const float* w_loc, const float* b_loc, const float* g_img,
                           const float* be_img, const float* g_emb, const float* be_emb, const int32_t* totlen,
                           void* h, void* z, float* mean_emb, float* rstd_emb, void* e, float* mean_img,
                           float* rstd_img, int B, int T, int R, int d, uint32_t seed_img, uint32_t seed_emb,
                           uint32_t thresh24, float inv_keep, const void* img_rows, void* stream) {
  if (B <= 0 || T < 0 || R < 0 || d <= 0 || (d % 4) != 0 || d > 1024) return M3P_EINVAL;
  if (img_rows && ((uintptr_t)img_rows & 7)) return M3P_EINVAL;
  EmbedFwdArgs a = {tok, (const bf16*)emb_bf16, pos, (const bf16*)img_proj, loc, w_loc, b_loc, g_img, be_img, g_emb, be_emb,
                    totlen, (bf16*)h, (bf16*)z, mean_emb, rstd_emb, (bf16*)e, mean_img, rstd_img, B, T, R, d,
                    seed_img, seed_emb, thresh24, inv_keep, (bf16*)const_cast<void*>(img_rows), img_rows ? 2 : 0};
  return launch_embed_fwd(a, stream);
}

int m3p_embed_image_rows_fwd(const void* img_proj, const float* loc, const float* w_loc, const float* b_loc,
                             const float* g_img, const float* be_img, void* e, float* mean_img, float* rstd_img,
                             void* img_rows, int B, int R, int d, uint32_t seed_img, uint32_t thresh24, float inv_keep,
                             void* stream) {
  if (B <= 0 || R <= 0 || d <= 0 || (d % 4) != 0 || d > 1024 || !img_rows || ((uintptr_t)img_rows & 7)) return M3P_EINVAL;
  EmbedFwdArgs a = {nullptr, nullptr, nullptr, (const bf16*)img_proj, loc, w_loc, b_loc, g_img, be_img, nullptr, nullptr,
                    nullptr, nullptr, nullptr, nullptr, nullptr, (bf16*)e, mean_img, rstd_img, B, 0, R, d,
                    seed_img, 0u, thresh24, inv_keep, (bf16*)img_rows, 1};
  return launch_embed_fwd(a, stream);
}

}  // extern "C"

namespace {
int launch_embed_fwd(const EmbedFwdArgs& a, void* stream) {
  const int B = a.B, T = a.T, R = a.R, d = a.d;
  const int rows = B * (R + T);
  const int blocks = (rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096;
  hipStream_t st = (hipStream_t)stream;
  switch (ni_of(d)) {
    case 1: hipLaunchKernelGGL(embed_fwd_kernel<1>, dim3(blocks), dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL(embed_fwd_kernel<2>, dim3(blocks), dim3(256), 0, st, a); break;
    case 3: hipLaunchKernelGGL(embed_fwd_kernel<3>, dim3(blocks), dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL(embed_fwd_kernel<4>, dim3(blocks), dim3(256), 0, st, a); break;
  }
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}
}  // namespace

extern "C" {

int m3p_embed_assemble_bwd(const void* dh, const void* z, const float* mean_emb, const float* rstd_emb,
                           const float* g_emb, const void* e, const float* mean_img, const float* rstd_img,
                           const float* g_img, const int64_t* tok, const int32_t* totlen, const float* loc,
                           void* dz_scratch, void* de, float* d_g_emb, float* d_be_emb, float* d_pos, float* d_emb,
                           void* d_tok_rows, float* d_g_img, float* d_be_img, float* d_b_img, float* d_b_loc, float* d_w_loc,
                           int B, int T, int R, int d, int pad_index, uint32_t seed_img, uint32_t seed_emb,
                           uint32_t thresh24, float inv_keep, int phase, void* stream) {
  if (B <= 0 || T < 0 || R < 0 || d <= 0 || (d % 4) != 0 || d > 1024 || phase < 0 || phase > 2) return M3P_EINVAL;
  int bsplit = 1;
  while (bsplit < 16 && (R + T) * bsplit < M3P_EMB_BWD_BLOCKS && B / (4 * bsplit) >= 8) bsplit *= 2;
  EmbedBwdArgs a = {(const bf16*)dh, (const bf16*)z, mean_emb, rstd_emb, g_emb, (const bf16*)e, mean_img, rstd_img, g_img,
                    tok, totlen, loc, (bf16*)dz_scratch, (bf16*)de, d_g_emb, d_be_emb, d_pos, d_emb,
                    (bf16*)d_tok_rows, d_g_img, d_be_img, d_b_img, d_b_loc, d_w_loc, B, T, R, d, pad_index, bsplit,
                    seed_img, seed_emb, thresh24, inv_keep};
  hipStream_t st = (hipStream_t)stream;
  const int S = R + T;
#define M3P_EMB_BWD(NI)                                                                              \
  do {                                                                                               \
    if (phase != 2) hipLaunchKernelGGL(embed_bwd_rows_kernel<NI>, dim3(S* bsplit), dim3(256), 0, st, a);             \
    if (R > 0 && phase != 1) hipLaunchKernelGGL(embed_bwd_img_kernel<NI>, dim3(R* bsplit), dim3(256), 0, st, a);   \
  } while (0)
  switch (ni_of(d)) {
    case 1: M3P_EMB_BWD(1); break;
    case 2: M3P_EMB_BWD(2); break;
    case 3: M3P_EMB_BWD(3); break;
    default: M3P_EMB_BWD(4); break;
  }
#undef M3P_EMB_BWD
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

}  // extern "C"

// Elementwise pieces of the AoA image refiner (M3P/src/model/transformer.py:287-422) that the
// encoder's kernels do not already provide.  The refiner's GEMMs, LayerNorms, attention and GELU run on the
// encoder kernels; what is left is
//   * dropout as its own pass, on a strided [rows, cols] view and optionally added to a residual
//     (SublayerConnection: x + dropout(sublayer(norm(x))), :381-394; the dropout on the concatenated AoA input,
//     :366; TransformerFFN's own dropout :226 in front of the sublayer's) - the view form lets the
//     concatenation torch.cat([attended, query], -1) and its dropout be written as two calls into the halves
//     of one [rows, 2d] buffer with the RNG indexed by position in that buffer;
//   * the gated linear unit nn.GLU of the AoA layer (:317): y = a * sigmoid(b) for ab = [a | b].
// All HBM-bound, 8-byte bf16x4 accesses, counter-based keep mask (common.hpp) so backward regenerates it.
#include "common.hpp"

namespace {

// y[r][c] = (res ? res[r][c] : 0) + (keep(r * rng_ld + rng_col0 + c) ? x[r][c] * inv_keep : 0)
__global__ __launch_bounds__(256) void dropout_rows_kernel(const bf16* __restrict__ x, int ldx, const bf16* __restrict__ res,
                                                           int ldres, bf16* __restrict__ y, int ldy, int rows, int cols4,
                                                           uint32_t rng_ld, uint32_t rng_col0, uint32_t seed,
                                                           uint32_t thresh24, float inv_keep) {
  const size_t n = (size_t)rows * cols4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols4), c = (int)(i - (size_t)r * cols4) * 4;
    f32x4 v = Vec4<bf16>::load(x + (size_t)r * ldx + c);
    if (thresh24) {
      const uint32_t base = (uint32_t)r * rng_ld + rng_col0 + (uint32_t)c;
      bool kp[4];
      m3p_keep_run<4>(base, seed, thresh24, kp);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = kp[j] ? v[j] * inv_keep : 0.f;
    }
    if (res) v = round_bf16(v) + Vec4<bf16>::load(res + (size_t)r * ldres + c);   // the dropped branch is a bf16 tensor in the reference
    Vec4<bf16>::store(y + (size_t)r * ldy + c, v);
  }
}

__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// ab [rows][2d] (pitch ld_ab): y[r][c] = ab[r][c] * sigmoid(ab[r][d + c])
__global__ __launch_bounds__(256) void glu_fwd_kernel(const bf16* __restrict__ ab, int ld_ab, bf16* __restrict__ y, int rows, int d4) {
  const size_t n = (size_t)rows * d4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / d4), c = (int)(i - (size_t)r * d4) * 4;
    const f32x4 a = Vec4<bf16>::load(ab + (size_t)r * ld_ab + c);
    const f32x4 b = Vec4<bf16>::load(ab + (size_t)r * ld_ab + 4 * d4 + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = a[j] * sigmoid_f(b[j]);
    Vec4<bf16>::store(y + (size_t)r * 4 * d4 + c, o);
  }
}

// dab[r][c] = dy * sigmoid(b) ; dab[r][d + c] = dy * a * sigmoid(b) * (1 - sigmoid(b))
__global__ __launch_bounds__(256) void glu_bwd_kernel(const bf16* __restrict__ ab, int ld_ab, const bf16* __restrict__ dy,
                                                      bf16* __restrict__ dab, int rows, int d4) {
  const size_t n = (size_t)rows * d4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / d4), c = (int)(i - (size_t)r * d4) * 4;
    const f32x4 a = Vec4<bf16>::load(ab + (size_t)r * ld_ab + c);
    const f32x4 b = Vec4<bf16>::load(ab + (size_t)r * ld_ab + 4 * d4 + c);
    const f32x4 g = Vec4<bf16>::load(dy + (size_t)r * 4 * d4 + c);
    f32x4 da, db;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float s = sigmoid_f(b[j]);
      da[j] = g[j] * s;
      db[j] = g[j] * a[j] * s * (1.0f - s);
    }
    Vec4<bf16>::store(dab + (size_t)r * ld_ab + c, da);
    Vec4<bf16>::store(dab + (size_t)r * ld_ab + 4 * d4 + c, db);
  }
}

inline int grid_for(size_t n) {
  const size_t b = (n + 255) / 256;
  return (int)(b < 8192 ? (b ? b : 1) : 8192);
}

}  // namespace

extern "C" {

int m3p_dropout_rows(const void* x, int ldx, const void* res, int ldres, void* y, int ldy, int rows, int cols,
                     uint32_t rng_ld, uint32_t rng_col0, uint32_t seed, uint32_t thresh24, float inv_keep, void* stream) {
  if (rows <= 0 || cols <= 0 || (cols % 4) != 0 || (ldx % 4) != 0 || (ldy % 4) != 0 || (res && (ldres % 4) != 0)) return M3P_EINVAL;
  if (((uintptr_t)x & 7) || ((uintptr_t)y & 7) || ((uintptr_t)res & 7) || (rng_col0 % 4) != 0 || (rng_ld % 4) != 0) return M3P_EINVAL;
  hipLaunchKernelGGL(dropout_rows_kernel, dim3(grid_for((size_t)rows * (cols / 4))), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)x, ldx, (const bf16*)res, ldres, (bf16*)y, ldy, rows, cols / 4, rng_ld, rng_col0, seed,
                     thresh24, inv_keep);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_glu_fwd(const void* ab, int ld_ab, void* y, int rows, int d, void* stream) {
  if (rows <= 0 || d <= 0 || (d % 4) != 0 || (ld_ab % 4) != 0 || ld_ab < 2 * d) return M3P_EINVAL;
  if (((uintptr_t)ab & 7) || ((uintptr_t)y & 7)) return M3P_EINVAL;
  hipLaunchKernelGGL(glu_fwd_kernel, dim3(grid_for((size_t)rows * (d / 4))), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)ab, ld_ab, (bf16*)y, rows, d / 4);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_glu_bwd(const void* ab, int ld_ab, const void* dy, void* dab, int rows, int d, void* stream) {
  if (rows <= 0 || d <= 0 || (d % 4) != 0 || (ld_ab % 4) != 0 || ld_ab < 2 * d) return M3P_EINVAL;
  if (((uintptr_t)ab & 7) || ((uintptr_t)dy & 7) || ((uintptr_t)dab & 7)) return M3P_EINVAL;
  hipLaunchKernelGGL(glu_bwd_kernel, dim3(grid_for((size_t)rows * (d / 4))), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)ab, ld_ab, (const bf16*)dy, (bf16*)dab, rows, d / 4);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

}  // extern "C"

// Hardware-semantics probes: tiny kernels that exercise exactly the MFMA operand / result
// lane maps and the ds_read_b64_tr_b16 gather that gemm.hip and attention.hip rely on, so
// tests/test_hw_probes.py can pin those assumptions on the real chip.
#include "common.hpp"
#include "../../include/m3p_hip.h"

namespace {

__global__ void probe_mfma_kernel(const bf16* __restrict__ a, const bf16* __restrict__ w, float* __restrict__ d,
                                  int* __restrict__ rowcol) {
  const int l = threadIdx.x;
  // operand map: lane l holds X[l & 15][8 * (l >> 4) .. +8]
  bf16x8 af = *reinterpret_cast<const bf16x8*>(a + (l & 15) * 32 + 8 * (l >> 4));
  bf16x8 wf = *reinterpret_cast<const bf16x8*>(w + (l & 15) * 32 + 8 * (l >> 4));
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, wf, acc, 0, 0, 0);
  // result map: D[i][j], j = l & 15, i = 4 * (l >> 4) + r ; D = A (16x32) * W^T (32x16)
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * (l >> 4) + r, j = l & 15;
    d[i * 16 + j] = acc[r];
    rowcol[(l * 4 + r) * 2 + 0] = i;
    rowcol[(l * 4 + r) * 2 + 1] = j;
  }
}

__global__ void probe_tr16_kernel(const short* __restrict__ tile, short* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 16];
  const int l = threadIdx.x;
  for (int i = l; i < 64 * 16; i += 64) lds[i] = tile[i];
  __syncthreads();
  // 16-lane group g reads the 4x16 block of rows 4g..4g+3: lane t supplies the address of
  // row 4g + (t >> 2), cols 4 * (t & 3) .. +3 and receives column t of that block.
  const int g = l >> 4, t = l & 15;
  const short* p = lds + (4 * g + (t >> 2)) * 16 + 4 * (t & 3);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}

}  // namespace

extern "C" {

const char* m3p_version(void) { return "m3p_hip 0.1 gfx950"; }

int m3p_probe_mfma_16x16x32(const void* a, const void* w, float* d, int* rowcol, void* stream) {
  hipLaunchKernelGGL(probe_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16*)a, (const bf16*)w, d, rowcol);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_probe_tr16(const void* tile, void* out, void* stream) {
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const short*)tile, (short*)out);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

}  // extern "C"

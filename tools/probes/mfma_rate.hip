// MFMA issue-rate probe (development aid, not product code): what the matrix pipe sustains on bf16 16x16x32 against 32x32x16
// with 128 accumulator registers per wave and register-resident operands, one or two waves per SIMD, random operands.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/mfma_rate.hip -o tools/probes/mfma_rate.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(512) void mfma_rate_kernel(const bf16x8* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = threadIdx.x;
  bf16x8 a[4], b[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = in[(tid + 64 * i) & 4095];
#pragma unroll
  for (int i = 0; i < 8; ++i) b[i] = in[(tid * 3 + 64 * i + 17) & 4095];
  float s = 0.f;
  if (MODE == 0) {
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
  } else if (MODE == 2) {
    // fp8 e4m3 x e4m3 on the block-scaled form (scales fixed at 1): 16 x 16 x 128, 65536 FLOP, 8 passes of 4 clocks at the 2x rate
    typedef __attribute__((ext_vector_type(8))) int i32x8;
    i32x8 a8[4], b8[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const bf16x8 lo = a[i], hi = a[(i + 1) & 3]; a8[i] = __builtin_bit_cast(i32x8, __builtin_shufflevector(lo, hi, 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15)); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { const bf16x8 lo = b[i], hi = b[(i + 1) & 7]; b8[i] = __builtin_bit_cast(i32x8, __builtin_shufflevector(lo, hi, 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15)); }
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[i], b8[j], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
  } else {
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i + 2 * kk], b[j + 4 * kk], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15];
  }
  if (s == 12345.678f) out[tid] = s;
}

// -> milliseconds for `iters` iterations (each: 32 x 16x16x32 or 16 x 32x32x16 per wave = 524288 FLOP per wave either way)
extern "C" __attribute__((visibility("default"))) float mfma_rate(int mode, int blocks, int threads, int iters, const void* in, void* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto k = mode == 2 ? mfma_rate_kernel<2> : (mode ? mfma_rate_kernel<1> : mfma_rate_kernel<0>);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, (const bf16x8*)in, (float*)out, 16);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, (const bf16x8*)in, (float*)out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms;
}

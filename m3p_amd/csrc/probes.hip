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

// v_mfma_scale_f32_16x16x128_f8f6f4 (the fp8 GEMM's instruction; gfx950 has no unscaled K = 128 form): operand lane l
// holds the 32 consecutive bytes X[l & 15][32 * (l >> 4) .. +32]; both block scales E8M0 = 127 (x 1.0); the result map
// is the 16x16 one of the bf16 instruction.  FA / FW: 0 = fp8 e4m3, 1 = bf8 e5m2 (OCP encodings).
typedef __attribute__((ext_vector_type(8))) int i32x8;
template <int FA, int FW>
__global__ void probe_mfma_fp8_kernel(const unsigned char* __restrict__ a, const unsigned char* __restrict__ w, float* __restrict__ d) {
  const int l = threadIdx.x;
  const i32x8 af = *reinterpret_cast<const i32x8*>(a + (l & 15) * 128 + 32 * (l >> 4));
  const i32x8 wf = *reinterpret_cast<const i32x8*>(w + (l & 15) * 128 + 32 * (l >> 4));
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af, wf, acc, FA, FW, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int r = 0; r < 4; ++r) d[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
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

namespace {
// v_permlane16_swap_b32 x, y: lane l holds (x, y) = (l, 100 + l) before; out[l] = x, out[64 + l] = y after
__global__ void probe_permlane16_swap_kernel(unsigned* out) {
  unsigned x = threadIdx.x, y = 100 + threadIdx.x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
  out[threadIdx.x] = x;
  out[64 + threadIdx.x] = y;
}
}  // namespace

extern "C" {

const char* m3p_version(void) { return "m3p_hip 0.1 gfx950"; }

int m3p_probe_mfma_16x16x32(const void* a, const void* w, float* d, int* rowcol, void* stream) {
  hipLaunchKernelGGL(probe_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16*)a, (const bf16*)w, d, rowcol);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_probe_mfma_fp8_16x16x128(const void* a, const void* w, float* d, int a_is_bf8, void* stream) {
  if (a_is_bf8)
    hipLaunchKernelGGL((probe_mfma_fp8_kernel<1, 0>), dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned char*)a, (const unsigned char*)w, d);
  else
    hipLaunchKernelGGL((probe_mfma_fp8_kernel<0, 0>), dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned char*)a, (const unsigned char*)w, d);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_probe_permlane16_swap(void* out_2x64_u32, void* stream) {
  hipLaunchKernelGGL(probe_permlane16_swap_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned*)out_2x64_u32);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

int m3p_probe_tr16(const void* tile, void* out, void* stream) {
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const short*)tile, (short*)out);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------
// Issue-cost probe (debug export, not part of the ABI header): one wave per SIMD runs
// `iters` rounds of {8 independent MFMAs, one memory instruction of kind `mode`} and
// reports its s_memtime ticks.  mode 0: none; 1: global_load_lds vaddr64; 2: global_load_lds
// saddr + 32-bit offset; 3: global_load_dwordx4 -> VGPR; 4: buffer_load_dwordx4 ... lds;
// 5: ds_read_b128.
// ---------------------------------------------------------------------------------
namespace {
template <int MODE>
__global__ __launch_bounds__(256) void probe_issue_kernel(const bf16* __restrict__ src, unsigned long long* __restrict__ out,
                                                          int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bf16* p = src + ((size_t)blockIdx.x * 4 + wid) * 4096 + lane * 8;   // 1 KB per wave round-robin over 8 KB
  const uint32_t lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wid * 8192;
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (bf16)1.0f; fb[i] = (bf16)0.5f; }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 sink = fa;
  uint4 gsink = {0, 0, 0, 0};
  // buffer resource for mode 4
  __attribute__((ext_vector_type(4))) uint32_t rsrc;
  rsrc[0] = (uint32_t)(uintptr_t)src; rsrc[1] = (uint32_t)((uintptr_t)src >> 32); rsrc[2] = 0x7fffffff; rsrc[3] = 0x00020000;
  const uint32_t voff = (uint32_t)((((size_t)blockIdx.x * 4 + wid) * 4096 + lane * 8) * 2);
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  u32x4 wdata = {(uint32_t)lane, 1u, 2u, 3u};
  uint32_t dummy = lane, d1 = lane, d2 = lane, d3 = lane;
  const uint32_t inv_addr = lds_addr + lane * 16;
  if (MODE == 7) asm volatile("s_mov_b32 m0, %0" :: "s"(lds_addr));
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const int sub = (it & 3) * 512;   // elements: 1 KB steps inside the wave's 8-KB window
#pragma unroll
    for (int k = 0; k < 8; ++k)
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[k]) : "v"(fa), "v"(fb));
    if (MODE == 1) {
      __builtin_amdgcn_global_load_lds(GLB_PTR(p + sub), LDS_PTR(smem + wid * 8192 + (it & 3) * 1024), 16, 0, 0);
    } else if (MODE == 2) {
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                   :: "v"(voff + sub * 2), "s"(src), "s"(lds_addr + (it & 3) * 1024) : "memory");
    } else if (MODE == 3) {
      u32x4 v;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p + sub) : "memory");
      if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
    } else if (MODE == 16) {
      asm volatile("ds_write_b128 %0, %1" :: "v"(inv_addr), "v"(wdata) : "memory");
      if ((it & 7) == 7) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    } else if (MODE == 17) {
      u32x4 v;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(v) : "v"(voff + sub * 2), "s"(src) : "memory");
      if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
    } else if (MODE == 4) {
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                   :: "v"(voff + sub * 2), "s"(rsrc), "s"(lds_addr + (it & 3) * 1024) : "memory");
    } else if (MODE == 5) {
      bf16x8 r;
      asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(lds_addr + lane * 16 + (it & 3) * 1024));
      if ((it & 7) == 7) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); sink[0] += r[0]; }
    }
    else if (MODE == 6) {
      asm volatile("v_add_u32 %0, %0, 1" : "+v"(dummy));
    } else if (MODE == 7) {
      asm volatile("global_load_lds_dwordx4 %0, off" :: "v"(p) : "memory");      // m0 set once outside the loop
    } else if (MODE == 8) {
      bf16x8 r;
      asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(inv_addr));
      if ((it & 7) == 7) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); sink[0] += r[0]; }
    } else if (MODE == 9) {
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(lds_addr + (it & 3) * 1024));
    } else if (MODE == 10) {
      asm volatile("v_add_u32 %0, %0, 1\n\tv_add_u32 %1, %1, 1\n\tv_add_u32 %2, %2, 1\n\tv_add_u32 %3, %3, 1" : "+v"(dummy), "+v"(d1), "+v"(d2), "+v"(d3));
    }
    else if (MODE == 11) {
      asm volatile("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4" : "+v"(dummy), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(0x9E3779B1u));
    } else if (MODE == 12) {
      asm volatile("v_mul_u32_u24 %0, %0, %4\n\tv_mul_u32_u24 %1, %1, %4\n\tv_mul_u32_u24 %2, %2, %4\n\tv_mul_u32_u24 %3, %3, %4" : "+v"(dummy), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(0x9E3779u));
    } else if (MODE == 13) {
      asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3" : "+v"(dummy), "+v"(d1), "+v"(d2), "+v"(d3));
    } else if (MODE == 14) {
      asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n\tv_mad_u32_u24 %1, %1, %4, %2\n\tv_mad_u32_u24 %2, %2, %4, %3\n\tv_mad_u32_u24 %3, %3, %4, %0" : "+v"(dummy), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(0x9E3779u));
    } else if (MODE == 15) {
      asm volatile("v_mul_hi_u32 %0, %0, %4\n\tv_mul_hi_u32 %1, %1, %4\n\tv_mul_hi_u32 %2, %2, %4\n\tv_mul_hi_u32 %3, %3, %4" : "+v"(dummy), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(0x9E3779B1u));
    }
    if ((MODE == 1 || MODE == 2 || MODE == 4 || MODE == 7) && (it & 7) == 7) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = (float)sink[0] + (float)gsink.x + (float)(dummy + d1 + d2 + d3);
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  if (lane == 0) { out[(blockIdx.x * 4 + wid) * 2] = t1 - t0; out[(blockIdx.x * 4 + wid) * 2 + 1] = (unsigned long long)s; }
}
}  // namespace

extern "C" __attribute__((visibility("default"))) int m3p_debug_probe_issue(int mode, const void* src, unsigned long long* out,
                                                                            int iters, int nblocks, void* stream) {
  const size_t lds = 4 * 8192;
#define LAUNCH(MODE) hipLaunchKernelGGL(probe_issue_kernel<MODE>, dim3(nblocks), dim3(256), lds, (hipStream_t)stream, (const bf16*)src, out, iters)
  switch (mode) {
    case 0: LAUNCH(0); break;
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 3: LAUNCH(3); break;
    case 4: LAUNCH(4); break;
    case 5: LAUNCH(5); break;
    case 6: LAUNCH(6); break;
    case 7: LAUNCH(7); break;
    case 8: LAUNCH(8); break;
    case 9: LAUNCH(9); break;
    case 10: LAUNCH(10); break;
    case 11: LAUNCH(11); break;
    case 12: LAUNCH(12); break;
    case 13: LAUNCH(13); break;
    case 14: LAUNCH(14); break;
    case 15: LAUNCH(15); break;
    case 16: LAUNCH(16); break;
    default: LAUNCH(17); break;
  }
#undef LAUNCH
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------
// Semantics probe (debug export): does the immediate offset of global_load_lds_dwordx4 move
// the LDS destination as well as the global source?  One wave loads 1 KB from src + 2048
// with offset:1024 into an 8-KB zeroed LDS window whose M0 base is window + 512; the whole
// window is copied out so the host can see where the KB landed.
// ---------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(64) void probe_dma_offset_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint32_t win[2048];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2048; i += 64) win[i] = 0xdeadbeefu;
  __syncthreads();
  const uint32_t lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)win + 512;
  const uint32_t* p = src + 512 + lane * 4;     // byte address src + 2048 + lane*16
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\ts_waitcnt vmcnt(0)"
               :: "v"(p), "s"(lds_addr) : "memory");
  __syncthreads();
  for (int i = lane; i < 2048; i += 64) out[i] = win[i];
}
}  // namespace

extern "C" __attribute__((visibility("default"))) int m3p_debug_probe_dma_offset(const void* src, void* out, void* stream) {
  hipLaunchKernelGGL(probe_dma_offset_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint32_t*)src, (uint32_t*)out);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------
// Ingest probe (debug export): how many bytes per clock can one CU pull out of L2?  Every
// workgroup (4 waves) streams `rounds` x 32 KB; the 32 workgroups of an XCD walk the same
// 2-MB window (L2-resident).  mode 0: global_load_lds 16 B/lane, 128-B row segments (a wave
// instruction = 8 rows x 128 B with a row pitch of 6 KB); mode 1: same with 64-B row segments
// (16 rows x 64 B); mode 2: global_load_dwordx4 into VGPRs, 128-B segments; mode 3: mode 0 but
// fully contiguous 1 KB per instruction.
// ---------------------------------------------------------------------------------
namespace {
template <int MODE>
__global__ __launch_bounds__(256) void probe_ingest_kernel(const char* __restrict__ src, unsigned long long* __restrict__ out,
                                                           int rounds) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7;
  const char* base = src + (size_t)xcd * (2u << 20);
  constexpr int PITCH = 6144;
  uint4 sink = {0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < rounds; ++r) {
    // 32 KB per round per workgroup = 8 instructions per wave
    const int win = ((r * 37 + (blockIdx.x >> 3) * 5) & 63) * 32768;        // 64 windows of 32 KB inside the 2 MB
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = wid * 8 + i;        // instruction index 0..31 inside the round
      const char* p;
      if (MODE == 0 || MODE == 2) p = base + ((win / PITCH) * PITCH + (size_t)(q * 8 + (lane >> 3)) * PITCH + (lane & 7) * 16) % (2u << 20);
      else if (MODE == 1) p = base + ((win / PITCH) * PITCH + (size_t)(q * 16 + (lane >> 2)) * PITCH + (lane & 3) * 16) % (2u << 20);
      else p = base + win + q * 1024 + lane * 16;
      if (MODE == 2) {
        typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
        u32x4 v;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p) : "memory");   // result never read: issue + return only
      } else {
        __builtin_amdgcn_global_load_lds(GLB_PTR(p), LDS_PTR(smem + (r & 3) * 32768 + q * 1024), 16, 0, 0);
      }
    }
    if (r >= 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 4 + wid] = t1 - t0 + sink.x;
}
}  // namespace

extern "C" __attribute__((visibility("default"))) int m3p_debug_probe_ingest(int mode, const void* src, unsigned long long* out,
                                                                             int rounds, int nblocks, void* stream) {
  const size_t lds = 4 * 32768;
#define LAUNCH(MODE)                                                                                                   \
  do {                                                                                                                 \
    hipFuncSetAttribute((const void*)probe_ingest_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(probe_ingest_kernel<MODE>, dim3(nblocks), dim3(256), lds, (hipStream_t)stream, (const char*)src, out, rounds); \
  } while (0)
  switch (mode) {
    case 0: LAUNCH(0); break;
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    default: LAUNCH(3); break;
  }
#undef LAUNCH
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------
// Cleaner issue-cost probe: 8 unrolled rounds of {8 MFMAs, one memory instruction} per loop
// trip, waits only once per trip with generous counts (no latency exposure, no per-round
// branch).  mode 0 none, 1 LDS-DMA (invariant M0, vaddr64 invariant), 2 global_load_dwordx4
// saddr + voff (invariant), 3 ds_read_b128, 4 ds_write_b128, 5 global_load vaddr64 invariant,
// 6 LDS-DMA with per-round v_lshl_add_u64 address update.
// ---------------------------------------------------------------------------------
namespace {
typedef __attribute__((ext_vector_type(4))) uint32_t pu32x4;
template <int MODE>
__global__ __launch_bounds__(256) void probe_issue2_kernel(const bf16* __restrict__ src, unsigned long long* __restrict__ out,
                                                           int trips) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bf16* p = src + ((size_t)blockIdx.x * 4 + wid) * 4096 + lane * 8;
  const uint32_t lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wid * 8192;
  const uint32_t inv_addr = lds_addr + lane * 16;
  const uint32_t voff = (uint32_t)((((size_t)blockIdx.x * 4 + wid) * 4096 + lane * 8) * 2);
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (bf16)1.0f; fb[i] = (bf16)0.5f; }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  pu32x4 wdata = {(uint32_t)lane, 1u, 2u, 3u};
  asm volatile("s_mov_b32 m0, %0" :: "s"(lds_addr));
  const bf16* pp = p;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < trips; ++it) {
#pragma unroll
    for (int rd = 0; rd < 8; ++rd) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[k]) : "v"(fa), "v"(fb));
      if (MODE == 1) asm volatile("global_load_lds_dwordx4 %0, off" :: "v"(p) : "memory");
      else if (MODE == 2) { pu32x4 v; asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(v) : "v"(voff), "s"(src) : "memory"); }
      else if (MODE == 3) { bf16x8 r; asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(inv_addr)); }
      else if (MODE == 4) asm volatile("ds_write_b128 %0, %1" :: "v"(inv_addr), "v"(wdata) : "memory");
      else if (MODE == 5) { pu32x4 v; asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p) : "memory"); }
      else if (MODE == 6) {
        asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n\tglobal_load_lds_dwordx4 %0, off" : "+v"(pp) : "s"((unsigned long long)((rd & 1) ? 1024 : -1024)) : "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(32) lgkmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  if (lane == 0) { out[(blockIdx.x * 4 + wid) * 2] = t1 - t0; out[(blockIdx.x * 4 + wid) * 2 + 1] = (unsigned long long)s; }
}
}  // namespace

extern "C" __attribute__((visibility("default"))) int m3p_debug_probe_issue2(int mode, const void* src, unsigned long long* out,
                                                                             int trips, int nblocks, void* stream) {
  const size_t lds = 4 * 8192;
#define LAUNCH(MODE) hipLaunchKernelGGL(probe_issue2_kernel<MODE>, dim3(nblocks), dim3(256), lds, (hipStream_t)stream, (const bf16*)src, out, trips)
  switch (mode) {
    case 0: LAUNCH(0); break;
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 3: LAUNCH(3); break;
    case 4: LAUNCH(4); break;
    case 5: LAUNCH(5); break;
    default: LAUNCH(6); break;
  }
#undef LAUNCH
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------
// Stand-in for a collective's kernel (debug export, not part of the ABI header): `nblocks` workgroups of 512 threads
// stream `bytes` from src to dst - like RCCL's channels, each workgroup wants a CU of its own and lives as long as its
// share of the copy takes.  tools/cu_reserve_ab.py runs it on a side stream beside the training step to price the CUs the
// persistent GEMM grids leave free under data parallelism (m3p_set_persistent_grid).
// ---------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(512) void side_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 512) dst[i] = src[i];
}
}  // namespace
extern "C" __attribute__((visibility("default"))) int m3p_debug_side_copy(const void* src, void* dst, size_t bytes, int nblocks,
                                                                           void* stream) {
  if (nblocks <= 0 || (bytes & 15)) return M3P_EINVAL;
  hipLaunchKernelGGL(side_copy_kernel, dim3(nblocks), dim3(512), 0, (hipStream_t)stream, (const f32x4*)src, (f32x4*)dst, bytes >> 4);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

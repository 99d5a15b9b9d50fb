// MFMA GEMMs for the M3P hot path on gfx950 (MI355X).
//
//   gemm_nt   : C[M,N] = epi(A[M,K] x W[N,K]^T)    both operands K-contiguous (nn.Linear)
//   gemm_wgrad: dW[N,K] += dY[M,N]^T x X[M,K]      both operands contraction-strided,
//               fragments come out of LDS through ds_read_b64_tr_b16 (hardware transpose)
//
// Kernels, in the order the launchers prefer them:
//   gemm_nt_w4_kernel / gemm_wgrad_w4_kernel   four waves, 256x256 tile, 128x128 per wave with the
//       256 accumulators pinned in AGPRs, 64-deep K-tiles in two 64-KB LDS stages; full-tile shapes.
//       Their bodies (the MFMA / memory-instruction schedule) are generated: tools/gen/gen_w4.py.
//   gemm_nt_ring_kernel / gemm_wgrad_ring_kernel   eight waves, 256x128 tile, 64x64 per wave,
//       three-stage 48-KB ring; ragged shapes, the dGELU epilogue, the vocabulary matrices.
//   gemm_nt_streamk_kernel   contraction-split NT with fp32 atomics (vocabulary data gradient).
//   gemm_nt_kernel / gemm_wgrad_kernel   128x128, two stages: small-M fallbacks.
// Common: v_mfma_f32_16x16x32_bf16 with swapped operands (a lane owns 4 consecutive output
// columns), HBM -> LDS by global_load_lds_dwordx4 (no VGPR round trip).  The LDS image of a
// K-contiguous tile is [rows][64] bf16 (128 B rows); because an LDS-DMA writes lane-linear,
// the bank swizzle (16-B chunk ^= row & 7) is applied to the per-lane SOURCE address and
// again on the ds_read_b128 (cdna guide rule 21), which makes the fragment reads
// conflict-free.  Persistent kernels deal tiles so that each XCD (private L2) walks a
// strip-blocked run of tiles that share their operand panels.
#include <mutex>
#include <type_traits>
#include "common.hpp"
#include "../../include/m3p_hip.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

static int g_variant = 1;  // debug: 0 = force the 128x128 2-stage kernel, 1 = auto (skinny for M <= 128, eight-wave 256x256 for full tiles, ring otherwise),
                           // 2 = force the four-wave 256x256 kernel, 3 = force ring, 6 = force eight-wave 256x256, 7 = the round-1 auto choice (w4 / ring), 9 = auto without the skinny kernel
static int g_ablate = 0;  // debug: timeline kernels only (bit0 = no fragment reads, bit1 = no LDS-DMA)
static int g_w8_strip = 0; // debug: eight-wave kernel's strip rule (bits 0-3: strip width in tiles, 0 = the default rule; bit 4: odd strips walk M backwards)
extern "C" void m3p_debug_set_variant(int v) { g_variant = v & 0xff; g_ablate = (v >> 8) & 0xff; g_w8_strip = (v >> 16) & 0xff; }     // (declared in the header's developer section)

namespace {

constexpr int BK = 64;            // contraction elements per LDS stage
constexpr int ROWB = BK * 2;      // bytes per LDS row of a K-contiguous tile (128)

// ---------------------------------------------------------------------------------
// NT kernel
// ---------------------------------------------------------------------------------
template <int EPI>
__device__ __forceinline__ void epilogue_store(const M3PEpilogue& ep, bf16* __restrict__ C, int ldc,
                                               int M, int N, int m, int n, f32x4 acc, f32x4& csum) {
  if (m >= M || n >= N) return;
  const bool full = (n + 3 < N);
  float v[4] = {acc[0], acc[1], acc[2], acc[3]};
  const float alpha = (ep.alpha == 0.f) ? 1.f : ep.alpha;
  if (EPI == M3P_EPI_NONE || EPI == M3P_EPI_BIAS || EPI == M3P_EPI_RES) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= alpha;
  }
  if (EPI == M3P_EPI_BIAS || EPI == M3P_EPI_BIAS_GELU || EPI == M3P_EPI_BIAS_DROP_RES) {
    if (ep.bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < N) v[r] += ep.bias[n + r];
    }
  }
  if (EPI == M3P_EPI_BIAS) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n + r < ep.scale_cols) v[r] *= ep.scale;
  }
  if (EPI == M3P_EPI_BIAS_GELU) {
    bf16* U = reinterpret_cast<bf16*>(ep.out2) + (size_t)m * ep.ld_out2 + n;
    bf16 ur[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { ur[r] = (bf16)v[r]; v[r] = gelu_erf_f((float)ur[r]); }
    if (full) *reinterpret_cast<bf16x4*>(U) = bf16x4{ur[0], ur[1], ur[2], ur[3]};
    else
      for (int r = 0; r < 4; ++r) if (n + r < N) U[r] = ur[r];
  }
  if (EPI == M3P_EPI_BIAS_DROP_RES) {
    if (ep.thresh24) {
      const uint32_t base = (uint32_t)m * (uint32_t)N + (uint32_t)n;
      bool kp[4];
      m3p_keep_run<4>(base, ep.seed, ep.thresh24, kp);        // (any N: the run may start on an odd element)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = kp[r] ? v[r] * ep.inv_keep : 0.f;
    }
  }
  if (EPI == M3P_EPI_BIAS_DROP_RES || EPI == M3P_EPI_RES || EPI == M3P_EPI_DGELU || EPI == M3P_EPI_MUL) {
    const bf16* X = reinterpret_cast<const bf16*>(ep.aux) + (size_t)m * ep.ld_aux + n;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (full) { bf16x4 t = *reinterpret_cast<const bf16x4*>(X); a[0] = (float)t[0]; a[1] = (float)t[1]; a[2] = (float)t[2]; a[3] = (float)t[3]; }
    else
      for (int r = 0; r < 4; ++r) if (n + r < N) a[r] = (float)X[r];
    if (EPI == M3P_EPI_DGELU) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= gelu_erf_grad_f(a[r]);
    } else if (EPI == M3P_EPI_MUL) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= a[r];
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += a[r];
    }
  }
  bf16 o[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = (bf16)v[r];
  bf16* Cp = C + (size_t)m * ldc + n;
  if (full) *reinterpret_cast<bf16x4*>(Cp) = bf16x4{o[0], o[1], o[2], o[3]};
  else
    for (int r = 0; r < 4; ++r) if (n + r < N) Cp[r] = o[r];
  if (EPI == M3P_EPI_DGELU || EPI == M3P_EPI_MUL) {
#pragma unroll
    for (int r = 0; r < 4; ++r) csum[r] += (n + r < N) ? (float)o[r] : 0.f;
  }
}

template <int BM, int BN, int EPI>
__global__ __launch_bounds__(64 * (BM / 64) * (BN / 64))
void gemm_nt_kernel(const bf16* __restrict__ A, int lda, const bf16* __restrict__ W, int ldw,
                    bf16* __restrict__ C, int ldc, int M, int N, int K, M3PEpilogue ep,
                    int tiles_m, int tiles_n) {
  constexpr int WAVES_M = BM / 64, WAVES_N = BN / 64, NWAVES = WAVES_M * WAVES_N;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = id / tiles_n, tn = id - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- staging: one wave instruction = 8 rows x 128 B, lane -> (row l>>3, LDS chunk l&7),
  //      global chunk = LDS chunk ^ (row & 7)
  const int sr = lane >> 3, sc = (lane & 7) ^ sr;
  const bf16* a_src[BM / 8 / NWAVES];
  const bf16* w_src[BN / 8 / NWAVES];
#pragma unroll
  for (int i = 0; i < BM / 8 / NWAVES; ++i) {
    int row = (wid + i * NWAVES) * 8 + sr;
    int gm = min(m0 + row, M - 1);
    a_src[i] = A + (size_t)gm * lda + sc * 8;
  }
#pragma unroll
  for (int i = 0; i < BN / 8 / NWAVES; ++i) {
    int row = (wid + i * NWAVES) * 8 + sr;
    int gn = min(n0 + row, N - 1);
    w_src[i] = W + (size_t)gn * ldw + sc * 8;
  }
  auto stage = [&](int kt, int s) {
    char* sa = smem + s * STAGE;
    char* sb = sa + A_BYTES;
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < BM / 8 / NWAVES; ++i)
      __builtin_amdgcn_global_load_lds(GLB_PTR(a_src[i] + k0), LDS_PTR(sa + (wid + i * NWAVES) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < BN / 8 / NWAVES; ++i)
      __builtin_amdgcn_global_load_lds(GLB_PTR(w_src[i] + k0), LDS_PTR(sb + (wid + i * NWAVES) * 1024), 16, 0, 0);
  };

  // ---- fragment addressing
  const int wm = wid / WAVES_N, wn = wid - wm * WAVES_N;
  const int fr = lane & 15, fg = lane >> 4;
  int a_off[2], b_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int ch = ((fg + 4 * ks) ^ (fr & 7)) * 16;
    a_off[ks] = (wm * 64 + fr) * ROWB + ch;
    b_off[ks] = A_BYTES + (wn * 64 + fr) * ROWB + ch;
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / BK;
  stage(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(kt + 1, cur ^ 1);
    const char* sbase = smem + cur * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sbase + a_off[ks] + i * 16 * ROWB);
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(sbase + b_off[ks] + j * 16 * ROWB);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          // first operand = W rows (n), second = A rows (m): D[n][m] -> lane holds
          // m = l&15 and four consecutive n = 4*(l>>4)+r  (8-byte bf16 stores along N)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue
  f32x4 csum[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) csum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + fr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fg * 4;
      epilogue_store<EPI>(ep, C, ldc, M, N, m, n, acc[i][j], csum[j]);
    }
  }
  if ((EPI == M3P_EPI_DGELU || EPI == M3P_EPI_MUL) && ep.colsum) {
    // reduce over the 16 lanes that share fg (different rows), then one atomic per column
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = csum[j][r];
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
        const int n = n0 + wn * 64 + j * 16 + fg * 4 + r;
        if (fr == 0 && n < N) unsafeAtomicAdd(ep.colsum + n, s);
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// NT kernel, ring version (the production path for large M): 256x128 output tile, 8 waves
// (4 x 2, 64x64 each), three 48-KB LDS stages (144 of the CU's 160 KB).  HBM -> LDS loads run
// TWO K-tiles ahead of the MFMAs and are retired with a COUNTED s_waitcnt vmcnt(6) (one
// tile = 6 LDS-DMA instructions per wave stays in flight across the barrier), so the
// workgroup never drains its memory pipeline inside the K loop; one raw s_barrier per K-tile.
// MFMA operand fragments are double-buffered per 32-deep k-step: the ds_read_b128s of the
// next k-step are issued before the 16 MFMAs of the current one.
// ---------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------
// Epilogue of an interior wave tile through LDS: the MFMA result layout gives a lane 4
// consecutive columns of one row (8-byte pieces, 32-B row segments per store instruction);
// staging 32 rows x 64 columns in LDS turns that into 16 bytes per lane and whole 128-B lines
// per 8 lanes.  `rows` are two 16-row MFMA tile rows (x 4 column tiles) starting at global row mrow0.
// ---------------------------------------------------------------------------------
constexpr int EP_PITCH = 144;                 // bytes per staged row: 128 + 16 (16-B aligned, rotates banks)
constexpr int EP_HALF = 32 * EP_PITCH;        // 4608 B per wave half-tile
// byte offset of (row, byte column) inside a wave's staging rows.  SW: 128-byte pitch with the 16-byte chunk index XORed
// with row & 7 instead of the padded pitch - 32 rows are exactly 4 KB, so eight waves' staging fits the 32 KB a CU has left
// beside two 64-KB stages (the eight-wave kernel then never touches a stage in its epilogue)
template <bool SW>
__device__ __forceinline__ int ep_off(int row, int bcol) {
  return SW ? row * 128 + ((((bcol >> 4) ^ (row & 7)) << 4) | (bcol & 15)) : row * EP_PITCH + bcol;
}
// The 8-byte accesses in the accumulator layout (lane = row fr of 16, four columns) put rows r and r + 8 of a 16-row block on
// the same banks (the chunk swizzle only knows row & 7): a two-way conflict on every such instruction (r03 counters: LDS bank
// conflicts in 5-9 % of the eight-wave kernel's cycles, all of them in the epilogue).  In the swizzled layout the 8-byte half
// inside the 16-byte chunk is therefore flipped for rows with bit 3 set (ep_off8); the 16-byte row-layout side of the same
// staging rows swaps the halves in registers (flip_halves - its rows are it * 8 + lane / 8, so bit 3 is a compile-time fact).
template <bool SW>
__device__ __forceinline__ int ep_off8(int row, int bcol) {
  return ep_off<SW>(row, bcol) ^ (SW ? (row & 8) : 0);
}
template <class V4>
__device__ __forceinline__ V4 flip_halves(const V4& t, bool flip) {
  return flip ? V4{t[2], t[3], t[0], t[1]} : t;
}

// LDS accesses the compiler does not see (ALDS = true in the epilogue helpers below).  Why: a kernel whose K loop keeps
// buffer_load ... lds transfers in flight makes the compiler put s_waitcnt vmcnt(0) in front of EVERY LDS access it knows of
// (the transfer might alias it) - and vmcnt(0) also waits for the previous piece's global stores, a full memory round trip
// per staged piece.  The staging buffers never alias the stages; with these the waits are ours (lgkm_wait ties the wait to
// the values it is for, so that their uses cannot be scheduled in front of it).
typedef unsigned int u32x4_lds __attribute__((ext_vector_type(4)));
#ifndef M3P_EPI_ALDS
#define M3P_EPI_ALDS 0       // 1: the eight-wave kernel's epilogues stage through these too.  Measured neutral there (two waves per
#endif                       // SIMD: the partner wave runs under the vmcnt(0)) - QKV 150 / 150 us, lin1 + GELU + byte 246 / 243, byte dgrad 214 / 216
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
__device__ __forceinline__ void lds_w64(uint32_t a, const bf16x4& v) { asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(v)); }
__device__ __forceinline__ void lds_w128(uint32_t a, const u32x4_lds& v) { asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(v)); }
__device__ __forceinline__ void lds_r64(uint32_t a, bf16x4& v) { asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a)); }
__device__ __forceinline__ void lds_r128(uint32_t a, u32x4_lds& v) { asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); }

// gelu_erf'(u) of a bf16 u is a function of 16 bits: the dGELU epilogue looks it up instead of
// evaluating erf/exp (~20 exposed VALU instructions per element).  The table covers |u| in
// [2^-15, 8) (exponents -15..2 x 128 mantissas = 2304 fp32 entries, 9 KB of LDS), built from what
// gelu_erf_grad_f returns for that bf16 value; gelu'(-u) = 1 - gelu'(u).
// M3P_GQ_TRIM: the leaner form of the lin1 + GELU + byte epilogue (1, default since late round 4): |x| as a source modifier of
// the first multiply, Phi(x) = 1/2 + copysign(1/2 - tail, x), and the code converted AND packed by v_cvt_pk_u8_f32.
// What is established: the instruction rounds to nearest (tests/test_gemm.py: the form with an extra +0.5 fails the codes'
// bias check, this one passes), and ONE in-process A/B of the three edits TOGETHER: 240.5 -> 231.8 us on the 41984 x 3072 x 768
// product (tools/ab_gemm.py).  Not established: which of the three edits the 8.7 us come from (never built separately), and
// the effect on the step.  0 = the first round-4 form, kept for A/B builds (tools/build_variant.sh).
#ifndef M3P_GQ_TRIM
#define M3P_GQ_TRIM 1
#endif
#ifndef M3P_W8_SPARE_EPILOGUE
#define M3P_W8_SPARE_EPILOGUE 1
#endif
#ifndef M3P_MQ_PREFETCH
#define M3P_MQ_PREFETCH 0
#endif
#ifndef M3P_MQ_EARLYREQ
#define M3P_MQ_EARLYREQ 0
#endif
#ifndef M3P_LSE_NT
#define M3P_LSE_NT 1      // the 2.4 GB of logits leave with non-temporal stores: 1883 -> 1847 us (profiles/r06_lse_nt.txt)
#endif
#ifndef M3P_DGELU_LUT
#define M3P_DGELU_LUT 1
#endif
constexpr int GELU_TAB_LO = 0x3800, GELU_TAB_HI = 0x4100, GELU_TAB_N = GELU_TAB_HI - GELU_TAB_LO;
// The table holds gelu'(|u|) - 1/2, an odd function of u: gelu'(u) = 1/2 + copysign(tab[|u|], u).  Seven VALU
// instructions and one LDS read per element (and, sub, med3, lshl_add, bfi, add, + the multiply); the first table
// entry serves every |u| < 2^-15 (error 2.4e-5, a tenth of a bf16 ulp of 1/2), the last every |u| >= 8 (gelu' = 1).
__device__ __forceinline__ void gelu_grad_table_fill(float* tab, int tid, int nthreads) {
  for (int i = tid; i < GELU_TAB_N; i += nthreads) {
    const uint16_t bits = (uint16_t)(GELU_TAB_LO + i);
    tab[i] = gelu_erf_grad_f((float)__builtin_bit_cast(bf16, bits)) - 0.5f;
  }
}
__device__ __forceinline__ float gelu_grad_lookup(const float* tab, bf16 u) {
  const uint32_t b = __builtin_bit_cast(uint16_t, u);
  const int idx = min(max((int)(b & 0x7FFFu) - GELU_TAB_LO, 0), GELU_TAB_N - 1);   // (compiles to v_med3_i32)
  const uint32_t t = __builtin_bit_cast(uint32_t, tab[idx]);
  return 0.5f + __builtin_bit_cast(float, (t & 0x7FFFFFFFu) | ((b << 16) & 0x80000000u));
}

// bias values of the four 16-column tiles starting at column nw for this lane (columns nw + 16 j + 4 (lane >> 4) ..+3)
template <int EPI>
__device__ __forceinline__ void load_bias4(const M3PEpilogue& ep, int nw, int lane, f32x4 (&biasv)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    biasv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if ((EPI == M3P_EPI_BIAS || EPI == M3P_EPI_BIAS_GELU || EPI == M3P_EPI_BIAS_DROP_RES) && ep.bias)
      biasv[j] = *reinterpret_cast<const f32x4*>(ep.bias + nw + j * 16 + (lane >> 4) * 4);
  }
}

// this lane's share of the 32 x 64 aux tile (residual / pre-activation) at (mrow0, nw)
template <int EPI>
__device__ __forceinline__ void load_aux(const M3PEpilogue& ep, int mrow0, int nw, int lane, bf16x4 (&auxv)[2][4]) {
  constexpr bool kAux = (EPI == M3P_EPI_BIAS_DROP_RES || EPI == M3P_EPI_RES || EPI == M3P_EPI_DGELU || EPI == M3P_EPI_MUL);
  if (!kAux) return;
  const bf16* X = reinterpret_cast<const bf16*>(ep.aux) + (size_t)(mrow0 + (lane & 15)) * ep.ld_aux + nw + (lane >> 4) * 4;
#pragma unroll
  for (int ii = 0; ii < 2; ++ii)
#pragma unroll
    for (int j = 0; j < 4; ++j) auxv[ii][j] = *reinterpret_cast<const bf16x4*>(X + (size_t)(ii * 16) * ep.ld_aux + j * 16);
}

// The same aux half-tile fetched as whole 128-byte row segments (16 B per lane, 8 lanes per row, 4 instructions
// instead of 8) and transposed into the accumulator layout through the wave-private staging rows `r1`; load_aux's
// direct form touches sixteen 32-byte row pieces per instruction.  Measured: dGELU 0.349 -> 0.340 ms, the
// residual epilogues unchanged (the aux tile costs ~60 us per 41984 x 3072 GEMM either way - see DESIGN.md).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // (HIP's uint4 is a struct: arrays of it passed by reference end up in scratch)

// 16-byte output-row store.  M3P_ST_POLICY (A/B builds): 0 plain, 1 nt, 2 sc1 (write-through, line dropped from this XCD's L2),
// 3 sc0 sc1, 4 sc1 nt
#ifndef M3P_ST_POLICY
#define M3P_ST_POLICY 0
#endif
template <class P>
__device__ __forceinline__ void st16(P* p, const u32x4& v) {
#if M3P_ST_POLICY == 0
  *reinterpret_cast<u32x4*>(p) = v;
#elif M3P_ST_POLICY == 1
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
#elif M3P_ST_POLICY == 2
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
#elif M3P_ST_POLICY == 3
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
#else
  asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
#endif
}
template <bool NT>
__device__ __forceinline__ void st16p(void* p, const u32x4& v) {
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
  else *reinterpret_cast<u32x4*>(p) = v;
}
template <int EPI>
__device__ __forceinline__ void load_aux_rows_issue(const M3PEpilogue& ep, int mrow0, int nw, int lane, u32x4 (&t)[4]) {
  constexpr bool kAux = (EPI == M3P_EPI_BIAS_DROP_RES || EPI == M3P_EPI_RES || EPI == M3P_EPI_DGELU || EPI == M3P_EPI_MUL);
  if (!kAux) return;
  const int srow = lane >> 3, sch = lane & 7;
  const bf16* X = reinterpret_cast<const bf16*>(ep.aux) + (size_t)(mrow0 + srow) * ep.ld_aux + nw + sch * 8;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
#ifdef M3P_ABL_NOAUX      // (timing ablation: what the aux tile's trip from memory costs the epilogue - results are garbage)
    t[it] = u32x4{(uint32_t)lane, 0x3f803f80u, 0x3f803f80u, (uint32_t)it};
#else
    t[it] = *reinterpret_cast<const u32x4*>(X + (size_t)(it * 8) * ep.ld_aux);
#endif
  }
}
template <int EPI, bool SW = false, bool ALDS = false>
__device__ __forceinline__ void load_aux_rows_finish(int lane, char* r1, const u32x4 (&t)[4], bf16x4 (&auxv)[2][4]) {
  constexpr bool kAux = (EPI == M3P_EPI_BIAS_DROP_RES || EPI == M3P_EPI_RES || EPI == M3P_EPI_DGELU || EPI == M3P_EPI_MUL);
  if (!kAux) return;
  const int srow = lane >> 3, sch = lane & 7;
  const int fr = lane & 15, fg = lane >> 4;
  if (ALDS) {
    const uint32_t la = lds_addr(r1);
#pragma unroll
    for (int it = 0; it < 4; ++it) lds_w128(la + ep_off<SW>(it * 8 + srow, sch * 16), flip_halves(t[it], SW && (it & 1)));
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j) lds_r64(la + ep_off8<SW>(ii * 16 + fr, (j * 16 + fg * 4) * 2), auxv[ii][j]);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(auxv[0][0]), "+v"(auxv[0][1]), "+v"(auxv[0][2]), "+v"(auxv[0][3]),
                                         "+v"(auxv[1][0]), "+v"(auxv[1][1]), "+v"(auxv[1][2]), "+v"(auxv[1][3]));
    return;
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) *reinterpret_cast<u32x4*>(r1 + ep_off<SW>(it * 8 + srow, sch * 16)) = flip_halves(t[it], SW && (it & 1));
#pragma unroll
  for (int ii = 0; ii < 2; ++ii)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      auxv[ii][j] = *reinterpret_cast<const bf16x4*>(r1 + ep_off8<SW>(ii * 16 + fr, (j * 16 + fg * 4) * 2));
}
template <int EPI, bool SW = false, bool ALDS = false>
__device__ __forceinline__ void load_aux_rows(const M3PEpilogue& ep, int mrow0, int nw, int lane, char* r1, bf16x4 (&auxv)[2][4]) {
  u32x4 t[4];
  load_aux_rows_issue<EPI>(ep, mrow0, nw, lane, t);
  load_aux_rows_finish<EPI, SW, ALDS>(lane, r1, t, auxv);
}

// epilogue_half in two parts: the arithmetic on a 32 x 64 piece and its staging writes (accumulator layout -> r1) ...
template <int EPI, bool SW = false, bool ALDS = false>
__device__ __forceinline__ void epilogue_half_write(const M3PEpilogue& ep, int N, int mrow0, int nw, char* r1, const f32x4 (&rows)[2][4],
                                                    const f32x4 (&biasv)[4], const bf16x4 (&auxv)[2][4], int lane, f32x4 (&csum)[4],
                                                    bf16x4 (&ukeep)[2][4], const float* gtab = nullptr) {
  const int fr = lane & 15, fg = lane >> 4;
  constexpr bool kAux = (EPI == M3P_EPI_BIAS_DROP_RES || EPI == M3P_EPI_RES || EPI == M3P_EPI_DGELU || EPI == M3P_EPI_MUL);
  const float alpha = (ep.alpha == 0.f) ? 1.f : ep.alpha;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = nw + j * 16 + fg * 4;
    f32x4 bias4 = biasv[j];        // (zeros when the epilogue has no bias; fetched by the caller ahead of time)
    // alpha, bias and the per-column scale fold into ONE fma per element: the constants are per
    // column, i.e. per (j, r), not per row (the epilogue runs with the MFMA pipe idle - every
    // instruction in it is exposed)
    f32x4 mulc = {alpha, alpha, alpha, alpha};
    if (EPI == M3P_EPI_BIAS) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < ep.scale_cols) { mulc[r] *= ep.scale; bias4[r] *= ep.scale; }
    }
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const int rl = ii * 16 + fr;                  // row inside the staged half
      const int mrow = mrow0 + rl;                  // global row
      const int lo = ep_off8<SW>(rl, (j * 16 + fg * 4) * 2);
      f32x4 v = rows[ii][j];
      if (EPI == M3P_EPI_NONE || EPI == M3P_EPI_BIAS || EPI == M3P_EPI_RES) v = v * mulc + bias4;
      else if (EPI != M3P_EPI_DGELU && EPI != M3P_EPI_MUL) v += bias4;      // (those two have no bias: 32 exposed adds per half)
      if (EPI == M3P_EPI_BIAS_GELU) {
        ukeep[ii][j] = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf_f((float)ukeep[ii][j][r]);
      }
      if (EPI == M3P_EPI_BIAS_DROP_RES && ep.thresh24) {
        const uint32_t base = (uint32_t)mrow * (uint32_t)N + (uint32_t)n;      // (whole tiles: N and n are multiples of 4)
        bool kp[4];
        m3p_keep_even<4>(base, ep.seed, ep.thresh24, kp);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = kp[r] ? v[r] * ep.inv_keep : 0.f;
      }
      if (kAux) {
        const bf16x4 t = auxv[ii][j];       // residual / pre-activation tile, fetched by the caller ahead of time
        const f32x4 a = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
        if (EPI == M3P_EPI_DGELU) {
          if (gtab) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= gelu_grad_lookup(gtab, t[r]);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= gelu_erf_grad_f(a[r]);
          }
        } else if (EPI == M3P_EPI_MUL) {
          v *= a;
        } else {
          v += a;
        }
      }
      const bf16x4 ob = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
      if (ALDS) lds_w64(lds_addr(r1) + lo, ob);
      else *reinterpret_cast<bf16x4*>(r1 + lo) = ob;
      if (EPI == M3P_EPI_DGELU || EPI == M3P_EPI_MUL) csum[j] += f32x4{(float)ob[0], (float)ob[1], (float)ob[2], (float)ob[3]};
    }
  }
}
// ... the staged piece read back as rows (16 bytes per lane, 8 lanes per 128-byte row segment) ...
template <bool SW, bool ALDS = false>
__device__ __forceinline__ void epilogue_rows_read(const char* r1, int lane, u32x4 (&R)[4]) {
  const int srow = lane >> 3, sch = lane & 7;
  if (ALDS) {      // (issued only: the caller waits - lgkm_wait_rows - and flips the halves of rows 8-15, 24-31 then)
    const uint32_t la = lds_addr(r1);
#pragma unroll
    for (int it = 0; it < 4; ++it) lds_r128(la + ep_off<SW>(it * 8 + srow, sch * 16), R[it]);
    return;
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) R[it] = flip_halves(*reinterpret_cast<const u32x4*>(r1 + ep_off<SW>(it * 8 + srow, sch * 16)), SW && (it & 1));
}
// (16-row pieces: two row instructions)  read + wait + store in one go, asm accesses
__device__ __forceinline__ void epilogue_rows16_flush_alds(bf16* __restrict__ C, int ldc, int mrow0, int nw, const char* r1, int lane) {
  const int srow = lane >> 3, sch = lane & 7;
  const uint32_t la = lds_addr(r1);
  u32x4 R[2];
  lds_r128(la + ep_off<true>(srow, sch * 16), R[0]);
  lds_r128(la + ep_off<true>(8 + srow, sch * 16), R[1]);
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(R[0]), "+v"(R[1]));
  bf16* Cp = C + (size_t)mrow0 * ldc + nw + sch * 8;
  st16(Cp + (size_t)srow * ldc, R[0]);
  st16(Cp + (size_t)(8 + srow) * ldc, flip_halves(R[1], true));
}
// the wait that belongs to epilogue_rows_read<SW, true>: `younger` (8 or 0) LDS instructions of this wave may stay in flight
template <bool SW>
__device__ __forceinline__ void lgkm_wait_rows(u32x4 (&R)[4], bool younger8) {
  if (younger8) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]));
  else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]));
#pragma unroll
  for (int it = 0; it < 4; ++it) R[it] = flip_halves(R[it], SW && (it & 1));
}
// ... and stored.  (Apart so that a kernel with nobody else on its SIMD can keep a piece's LDS round trip in flight under
// the next piece's accumulator reads and arithmetic: gemm_nt_w4_kernel.)
__device__ __forceinline__ void epilogue_rows_store(bf16* __restrict__ C, int ldc, int mrow0, int nw, int lane, const u32x4 (&R)[4]) {
  const int srow = lane >> 3, sch = lane & 7;
  bf16* Cp = C + (size_t)mrow0 * ldc + nw + sch * 8;
#pragma unroll
  for (int it = 0; it < 4; ++it) st16(Cp + (size_t)(it * 8 + srow) * ldc, R[it]);
}
template <int EPI, bool SW = false, bool ALDS = false>
__device__ __forceinline__ void epilogue_half(const M3PEpilogue& ep, bf16* __restrict__ C, int ldc, int N,
                                              int mrow0, int nw, char* r1, const f32x4 (&rows)[2][4],
                                              const f32x4 (&biasv)[4], const bf16x4 (&auxv)[2][4], int lane, f32x4 (&csum)[4],
                                              const float* gtab = nullptr) {
  const int fr = lane & 15, fg = lane >> 4;
  const int srow = lane >> 3, sch = lane & 7;
  bf16x4 ukeep[2][4];
  constexpr bool kA = ALDS && EPI != M3P_EPI_BIAS_GELU;      // (the two-output epilogue keeps plain accesses: not on the hot path)
  epilogue_half_write<EPI, SW, kA>(ep, N, mrow0, nw, r1, rows, biasv, auxv, lane, csum, ukeep, gtab);
  if (kA) {
    u32x4 R[4];
    epilogue_rows_read<SW, true>(r1, lane, R);
    lgkm_wait_rows<SW>(R, false);
    epilogue_rows_store(C, ldc, mrow0, nw, lane, R);
    return;
  }
  bf16* Cp = C + (size_t)mrow0 * ldc + nw + sch * 8;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + srow;
    st16(Cp + (size_t)row * ldc, flip_halves(*reinterpret_cast<const u32x4*>(r1 + ep_off<SW>(row, sch * 16)), SW && (it & 1)));
  }
  if (EPI == M3P_EPI_BIAS_GELU) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
        *reinterpret_cast<bf16x4*>(r1 + ep_off8<SW>(ii * 16 + fr, (j * 16 + fg * 4) * 2)) = ukeep[ii][j];
    bf16* Up = reinterpret_cast<bf16*>(ep.out2) + (size_t)mrow0 * ep.ld_out2 + nw + sch * 8;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + srow;
      st16(Up + (size_t)row * ep.ld_out2, flip_halves(*reinterpret_cast<const u32x4*>(r1 + ep_off<SW>(row, sch * 16)), SW && (it & 1)));
    }
  }
}


// 16-row pieces of the multiply epilogues (dGELU / MUL: out = acc * f(aux), column sums) for the eight-wave kernel: the
// aux rows of piece h + 1 are requested while piece h is computed (8 registers per piece), staging is 2 KB per wave in
// the swizzled layout - eight waves' staging and the 9-KB derivative table fit the 32 KB beside the two stages.
__device__ __forceinline__ void aux16_issue(const M3PEpilogue& ep, int mrow0, int nw, int lane, u32x4 (&t)[2]) {
  const int srow = lane >> 3, sch = lane & 7;
  const bf16* X = reinterpret_cast<const bf16*>(ep.aux) + (size_t)(mrow0 + srow) * ep.ld_aux + nw + sch * 8;
  t[0] = *reinterpret_cast<const u32x4*>(X);
  t[1] = *reinterpret_cast<const u32x4*>(X + (size_t)8 * ep.ld_aux);
}
template <int EPI>
__device__ __forceinline__ void epilogue_piece16(const M3PEpilogue& ep, bf16* __restrict__ C, int ldc, int mrow0, int nw, char* r1,
                                                 const f32x4 (&rows)[4], const u32x4 (&t)[2], int lane, f32x4 (&csum)[4],
                                                 const float* gtab) {
  const int fr = lane & 15, fg = lane >> 4;
  const int srow = lane >> 3, sch = lane & 7;
  // aux rows -> accumulator layout through the staging rows
  *reinterpret_cast<u32x4*>(r1 + ep_off<true>(srow, sch * 16)) = t[0];
  *reinterpret_cast<u32x4*>(r1 + ep_off<true>(8 + srow, sch * 16)) = flip_halves(t[1], true);
  bf16x4 auxv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) auxv[j] = *reinterpret_cast<const bf16x4*>(r1 + ep_off8<true>(fr, (j * 16 + fg * 4) * 2));
  // the sixteen table reads of a piece are independent: all addresses first, all reads in flight together, one wait (left
  // inside the per-element expression the compiler chains address -> read -> wait -> multiply sixteen times: ~130 clocks each)
  float tv[4][4];
  if (EPI == M3P_EPI_DGELU && gtab) {
    int idx[4][4];
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x2 raw = __builtin_bit_cast(u32x2, auxv[j]);      // four bf16 bit patterns in two dwords
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t b = (raw[r >> 1] >> (16 * (r & 1))) & 0x7FFFu;
        idx[j][r] = min(max((int)b - GELU_TAB_LO, 0), GELU_TAB_N - 1);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) tv[j][r] = gtab[idx[j][r]];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x4 v = rows[j];
    const bf16x4 a = auxv[j];
    if (EPI == M3P_EPI_DGELU) {
      if (gtab) {
        {
          typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
          const u32x2 raw = __builtin_bit_cast(u32x2, a);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint32_t sign = (raw[r >> 1] << (16 * (1 - (r & 1)))) & 0x80000000u;
            const uint32_t tb = __builtin_bit_cast(uint32_t, tv[j][r]);
            v[r] *= 0.5f + __builtin_bit_cast(float, (tb & 0x7FFFFFFFu) | sign);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= gelu_erf_grad_f((float)a[r]);
      }
    } else {
      v *= f32x4{(float)a[0], (float)a[1], (float)a[2], (float)a[3]};
    }
    const bf16x4 ob = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    *reinterpret_cast<bf16x4*>(r1 + ep_off8<true>(fr, (j * 16 + fg * 4) * 2)) = ob;
    csum[j] += f32x4{(float)ob[0], (float)ob[1], (float)ob[2], (float)ob[3]};
  }
  bf16* Cp = C + (size_t)mrow0 * ldc + nw + sch * 8;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int row = it * 8 + srow;
    st16(Cp + (size_t)row * ldc, flip_halves(*reinterpret_cast<const u32x4*>(r1 + ep_off<true>(row, sch * 16)), it & 1));
  }
}

// 16-row piece of the byte-derivative epilogue (M3P_EPI_MULQ): out = acc * decode(code), column sums.  `q` holds this
// lane's sixteen codes of the piece in accumulator order (dword j = columns 16 j + 4 fg .. + 3 of row fr: the fragment-order
// layout of common.hpp, fetched by the caller with one 16-byte load) - no aux trip through LDS, three VALU per element.

// Round 6: the 8-bit copy of a 16 x 64 output piece for the fp8 product that consumes this output (M3PEpilogue::out8): four
// values of a lane -> one dword (v_cvt_pk_fp8 / _bf8 on the scaled, saturated fp32 values), 16 x 64 bytes through the wave's
// staging rows (64-byte pitch, the 16-byte chunk index XORed with (row >> 1) & 3: two lanes per bank on the way in, the
// minimum for 64 four-byte writes) and out as ONE 16-byte store per lane (4 lanes per 64-byte row segment).  The staging rows
// are the ones the bf16 piece just left: LDS instructions of a wave execute in order, so the writes below cannot pass the
// row reads in front of them.  `amax` = this lane's running max |value| (folded into *amax8 once, at the end of the kernel).
template <bool BF8>
__device__ __forceinline__ uint32_t pack8(const f32x4& v, float scale, float& amax) {
  constexpr float kMax = BF8 ? 57344.f : 448.f;
  float f[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    amax = fmaxf(amax, fabsf(v[r]));
    f[r] = __builtin_amdgcn_fmed3f(v[r] * scale, -kMax, kMax);
  }
  int w = 0;
  if (BF8) { w = __builtin_amdgcn_cvt_pk_bf8_f32(f[0], f[1], w, false); w = __builtin_amdgcn_cvt_pk_bf8_f32(f[2], f[3], w, true); }
  else { w = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w, false); w = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w, true); }
  return (uint32_t)w;
}
__device__ __forceinline__ void lds_w32(uint32_t a, uint32_t v) { asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(v)); }
__device__ __forceinline__ void piece8_write(char* r8, const uint32_t (&w8)[4], int lane) {
  const int fr = lane & 15, fg = lane >> 4;
  const uint32_t la = lds_addr(r8);
  asm volatile("" ::: "memory");      // (callers that stage the bf16 piece with plain C++ accesses: none of those may sink below here)
#pragma unroll
  for (int j = 0; j < 4; ++j) lds_w32(la + fr * 64 + ((j ^ ((fr >> 1) & 3)) << 4) + fg * 4, w8[j]);
}
__device__ __forceinline__ void piece8_store(uint8_t* __restrict__ o8, int ld8, int mrow0, int nw, char* r8, int lane) {
  const uint32_t la = lds_addr(r8);
  const int row = lane >> 2, c = lane & 3;
  u32x4_lds R;
  lds_r128(la + row * 64 + (c << 4), R);
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(R) :: "memory");      // (... and none of the next piece's may rise above here)
  st16p<true>(o8 + (size_t)(mrow0 + row) * ld8 + nw + ((c ^ ((row >> 1) & 3)) << 4), u32x4{R[0], R[1], R[2], R[3]});
}
__device__ __forceinline__ void piece8_flush(uint8_t* __restrict__ o8, int ld8, int mrow0, int nw, char* r1, const uint32_t (&w8)[4], int lane) {
  piece8_write(r1, w8, lane);
  piece8_store(o8, ld8, mrow0, nw, r1, lane);
}
// M3P_MQ_ABL (timing ablations, results are garbage): 1 = the codes are not read (the caller passes lane numbers), 2 = dU is
// neither staged nor stored, 4 = no column sums
#ifndef M3P_MQ_ABL
#define M3P_MQ_ABL 0
#endif
#ifndef M3P_MQ_NT
#define M3P_MQ_NT 1          // dU rows leave with non-temporal stores: 209 -> 194 us on the 41984 x 3072 x 768 product (profiles/r06_store_policy.txt)
#endif
#ifndef M3P_MQ_CSUMV
#define M3P_MQ_CSUMV 1
#endif
#ifndef M3P_MQ_NTLOAD
#define M3P_MQ_NTLOAD 0
#endif
template <int O8 = 0>      // 0: no 8-bit copy, 1: e4m3, 2: e5m2
__device__ __forceinline__ void epilogue_pieceq(bf16* __restrict__ C, int ldc, int mrow0, int nw, char* r1, const f32x4 (&rows)[4],
                                                const u32x4& q, int lane, f32x4 (&csum)[4], uint8_t* __restrict__ o8 = nullptr, int ld8 = 0,
                                                float scale8 = 1.f, float* amax = nullptr, char* r8 = nullptr) {
  const int fr = lane & 15, fg = lane >> 4;
  const int srow = lane >> 3, sch = lane & 7;
  uint32_t w8[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t w = q[j];
    f32x4 g = f32x4{(float)(w & 0xFFu), (float)((w >> 8) & 0xFFu), (float)((w >> 16) & 0xFFu), (float)(w >> 24)};   // v_cvt_f32_ubyte0..3
    g = g * GQ_STEP - GQ_OFF;
    const f32x4 v = rows[j] * g;
    const bf16x4 ob = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    if (M3P_MQ_ABL & 2) asm volatile("" :: "v"(ob));
    else if (M3P_EPI_ALDS || O8) lds_w64(lds_addr(r1) + ep_off8<true>(fr, (j * 16 + fg * 4) * 2), ob);
    else *reinterpret_cast<bf16x4*>(r1 + ep_off8<true>(fr, (j * 16 + fg * 4) * 2)) = ob;
    // column sums (lin1's bias gradient) of the fp32 products: one add per element (round 6; summing the bf16-rounded values
    // cost an unpack per element on top and is no closer to the fp32 reference's sum)
    if (!(M3P_MQ_ABL & 4)) { if (M3P_MQ_CSUMV) csum[j] += v; else csum[j] += f32x4{(float)ob[0], (float)ob[1], (float)ob[2], (float)ob[3]}; }
    if (O8) w8[j] = (O8 == 2) ? pack8<true>(v, scale8, *amax) : pack8<false>(v, scale8, *amax);
  }
  if (O8) piece8_write(r8, w8, lane);
  if (M3P_MQ_ABL & 2) return;
  if (M3P_EPI_ALDS) { epilogue_rows16_flush_alds(C, ldc, mrow0, nw, r1, lane); return; }
  bf16* Cp = C + (size_t)mrow0 * ldc + nw + sch * 8;
  if (O8) {
    // every staging access of this form is one the compiler does not see (no vmcnt(0) in front of each: with the 8-bit rows'
    // stores in flight as well, those waits cost ~50 us per launch); both pieces' row reads share one wait
    const uint32_t la = lds_addr(r1), l8 = lds_addr(r8);
    const int row8 = lane >> 2, c8 = lane & 3;
    u32x4_lds Ra, Rb, R8;
    lds_r128(la + ep_off<true>(srow, sch * 16), Ra);
    lds_r128(la + ep_off<true>(8 + srow, sch * 16), Rb);
    lds_r128(l8 + row8 * 64 + (c8 << 4), R8);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Ra), "+v"(Rb), "+v"(R8));
    st16p<M3P_MQ_NT != 0>(Cp + (size_t)srow * ldc, u32x4{Ra[0], Ra[1], Ra[2], Ra[3]});
    st16p<M3P_MQ_NT != 0>(Cp + (size_t)(8 + srow) * ldc, u32x4{Rb[2], Rb[3], Rb[0], Rb[1]});
    st16p<true>(o8 + (size_t)(mrow0 + row8) * ld8 + nw + ((c8 ^ ((row8 >> 1) & 3)) << 4), u32x4{R8[0], R8[1], R8[2], R8[3]});
    return;
  }
  u32x4 R2[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int row = it * 8 + srow;
    R2[it] = flip_halves(*reinterpret_cast<const u32x4*>(r1 + ep_off<true>(row, sch * 16)), it & 1);
  }
  // (the 8-bit piece has staging rows of its own, r8: its writes were issued with the bf16 piece's - below - and its row read
  //  shares the wait of theirs; the first version took a second, serialised trip through r1: 277 against 207 us per launch)
#pragma unroll
  for (int it = 0; it < 2; ++it) st16p<M3P_MQ_NT != 0>(Cp + (size_t)(it * 8 + srow) * ldc, R2[it]);
  if (O8) piece8_store(o8, ld8, mrow0, nw, r8, lane);
}

// 32-row piece of the vocabulary projection's epilogue (M3P_EPI_BIAS_LSE): logits = acc + bias as for M3P_EPI_BIAS, plus what
// the cross-entropy needs of them while they are still in registers - per (row, 64-column block of this wave) the block's
// maximum and sum of exp(logit - max) over the columns < V, written to stats[block][row] (consecutive rows of a block are
// consecutive: 16 lanes store 128 bytes).  A row's log-sum-exp is then a reduction over N / 64 pairs instead of a second
// pass over the 2.4 GB of logits.  The statistics are taken on the fp32 values (the stored logits are their bf16 rounding).
__device__ __forceinline__ void epilogue_half_lse(bf16* __restrict__ C, int ldc, float2* __restrict__ stats, int V, int mrow0, int nw,
                                                  char* r1, const f32x4 (&rows0)[4], const f32x4 (&rows1)[4],
                                                  const f32x4 (&biasv)[4], int lane, bool edge) {
  constexpr float kLog2e = 1.4426950408889634f;
  const int fr = lane & 15, fg = lane >> 4;
  const int srow = lane >> 3, sch = lane & 7;
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) {
    f32x4 x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x[j] = (ii ? rows1[j] : rows0[j]) + biasv[j];
      const bf16x4 xb = bf16x4{(bf16)x[j][0], (bf16)x[j][1], (bf16)x[j][2], (bf16)x[j][3]};
      if (M3P_EPI_ALDS) lds_w64(lds_addr(r1) + ep_off8<true>(ii * 16 + fr, (j * 16 + fg * 4) * 2), xb);
      else *reinterpret_cast<bf16x4*>(r1 + ep_off8<true>(ii * 16 + fr, (j * 16 + fg * 4) * 2)) = xb;
      // the statistics are taken on the ROUNDED logits - the very values the cross-entropy's gradient pass and its target term
      // read back - so that softmax rows sum to one and the gradient rows to zero exactly as with a statistics pass over the
      // stored logits (one shift per element)
#pragma unroll
      for (int r = 0; r < 4; ++r) x[j][r] = (float)xb[r];
    }
    if (edge) {        // (wave-uniform: only the last column tile holds columns >= V)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (nw + j * 16 + fg * 4 + r >= V) x[j][r] = -INFINITY;
    }
    f32x4 m4 = x[0];
#pragma unroll
    for (int j = 1; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) m4[r] = fmaxf(m4[r], x[j][r]);
    float m = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float mb = (m == -INFINITY) ? 0.f : m * kLog2e;
    f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 a = x[j] * kLog2e - mb;
#pragma unroll
      for (int r = 0; r < 4; ++r) s4[r] += __builtin_amdgcn_exp2f(a[r]);
    }
    float sm = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    sm += __shfl_xor(sm, 16, 64);
    sm += __shfl_xor(sm, 32, 64);
    if (fg == 0) stats[mrow0 + ii * 16 + fr] = float2{m, sm};
  }
  if (M3P_EPI_ALDS) {
    u32x4 R[4];
    epilogue_rows_read<true, true>(r1, lane, R);
    lgkm_wait_rows<true>(R, false);
    epilogue_rows_store(C, ldc, mrow0, nw, lane, R);
    return;
  }
  bf16* Cp = C + (size_t)mrow0 * ldc + nw + sch * 8;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + srow;
    st16p<M3P_LSE_NT != 0>(Cp + (size_t)row * ldc, flip_halves(*reinterpret_cast<const u32x4*>(r1 + ep_off<true>(row, sch * 16)), it & 1));
  }
}

// 32-row piece of the FFN1 epilogue that leaves BOTH things the layer needs of u = acc + bias (M3P_EPI_BIAS_GELUQ):
// h = gelu_erf(u) staged to row order like any output tile, and gelu_erf'(u) as one byte per element written
// straight from the accumulator layout in fragment order (16 bytes per lane and 16-row block: 1 KB per wave instruction,
// no staging) - what M3P_EPI_MULQ reads back the same way.  u itself is never stored.  The arithmetic is written on
// four-element vectors so that the polynomial runs on packed f32 instructions (the epilogue has the VALU to itself).
// M3P_GQ_ABL (timing ablations of this epilogue, results are garbage): 1 = no arithmetic (h = x, code = bits of x),
// 2 = the codes are not stored, 4 = h is neither staged nor stored
#ifndef M3P_GQ_ABL
#define M3P_GQ_ABL 0
#endif
__device__ __forceinline__ void epilogue_half_geluq(bf16* __restrict__ C, int ldc, uint8_t* __restrict__ qout, int mrow0, int nw,
                                                    char* r1, const f32x4 (&rows0)[4], const f32x4 (&rows1)[4],
                                                    const f32x4 (&biasv)[4], int lane) {
  const int fr = lane & 15, fg = lane >> 4;
  const int srow = lane >> 3, sch = lane & 7;
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) {
    u32x4 code;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // (x stays fp32: with u never stored there is no bf16 copy anything else would have to agree with)
      const f32x4 x = (ii ? rows1[j] : rows0[j]) + biasv[j];
#if M3P_GQ_ABL & 1
      {
        code[j] = __builtin_bit_cast(uint32_t, x[0]);
        const bf16x4 hb = bf16x4{(bf16)x[0], (bf16)x[1], (bf16)x[2], (bf16)x[3]};
        if (M3P_GQ_ABL & 4) asm volatile("" :: "v"(hb));
        else *reinterpret_cast<bf16x4*>(r1 + ep_off8<true>(ii * 16 + fr, (j * 16 + fg * 4) * 2)) = hb;
        continue;
      }
#endif
      // gelu_parts (common.hpp) on a vector: Phi(|x|) = 1 - (poly(t) t e) / 2, t = 1 / (1 + p z), z = |x| / sqrt 2, e = exp(-z^2)
      f32x4 z, t, e;
#if M3P_GQ_TRIM
#pragma unroll
      for (int r = 0; r < 4; ++r) z[r] = fabsf(x[r]) * 0.70710678118654752440f;      // (one v_mul with the |x| modifier)
#else
#pragma unroll
      for (int r = 0; r < 4; ++r) z[r] = fabsf(x[r]);
      z *= 0.70710678118654752440f;
#endif
      const f32x4 den = z * 0.3275911f + 1.0f;
      const f32x4 ez = z * z * -1.4426950408889634f;          // exp(-z^2) = exp2(-z^2 log2 e)
#pragma unroll
      for (int r = 0; r < 4; ++r) { t[r] = __builtin_amdgcn_rcpf(den[r]); e[r] = __builtin_amdgcn_exp2f(ez[r]); }
      f32x4 poly = t * 1.061405429f - 1.453152027f;
      poly = poly * t + 1.421413741f;
      poly = poly * t - 0.284496736f;
      poly = poly * t + 0.254829592f;
      const f32x4 tail = poly * t * e * 0.5f;                 // 1 - Phi(|x|)
      f32x4 cdf;
#if M3P_GQ_TRIM
      {       // Phi(x) = 1/2 + copysign(1/2 - tail, x): one packed subtract, a bit-field insert of x's sign, one packed add
        const f32x4 half = 0.5f - tail;
#pragma unroll
        for (int r = 0; r < 4; ++r) cdf[r] = __builtin_copysignf(half[r], x[r]);
        cdf += 0.5f;
      }
#else
#pragma unroll
      for (int r = 0; r < 4; ++r) cdf[r] = (x[r] >= 0.f) ? 1.0f - tail[r] : tail[r];
#endif
      const f32x4 hv = x * cdf;
      const f32x4 gd = x * e * 0.39894228040143267794f + cdf;                  // gelu'(x) = Phi + x phi
#if M3P_GQ_TRIM
      // v_cvt_pk_u8_f32 converts AND drops the byte into place: one instruction per element where the other form has a
      // conversion plus a shift / or.  It rounds to nearest - pinned by tests/test_gemm.py (with +0.5 added the codes' rms
      // error fails the 0.4-step bar), hence no +0.5 here (M3P_GQ_TRIM == 2 is the failing form, kept so the test can be re-run)
      const f32x4 qf = gd * GQ_INV + (GQ_OFF * GQ_INV + (M3P_GQ_TRIM == 2 ? 0.5f : 0.0f));
      uint32_t cw = 0;
      cw = __builtin_amdgcn_cvt_pk_u8_f32(qf[0], 0, cw); cw = __builtin_amdgcn_cvt_pk_u8_f32(qf[1], 1, cw);
      cw = __builtin_amdgcn_cvt_pk_u8_f32(qf[2], 2, cw); cw = __builtin_amdgcn_cvt_pk_u8_f32(qf[3], 3, cw);
      code[j] = cw;
#else
      const f32x4 qf = gd * GQ_INV + (GQ_OFF * GQ_INV + 0.5f);                 // in [1.7, 253.3): truncation = round to nearest
      code[j] = (uint32_t)qf[0] | ((uint32_t)qf[1] << 8) | ((uint32_t)qf[2] << 16) | ((uint32_t)qf[3] << 24);
#endif
      if (M3P_GQ_ABL & 4) { const bf16x4 hb = bf16x4{(bf16)hv[0], (bf16)hv[1], (bf16)hv[2], (bf16)hv[3]}; asm volatile("" :: "v"(hb)); }
      else if (M3P_EPI_ALDS) lds_w64(lds_addr(r1) + ep_off8<true>(ii * 16 + fr, (j * 16 + fg * 4) * 2), bf16x4{(bf16)hv[0], (bf16)hv[1], (bf16)hv[2], (bf16)hv[3]});
      else *reinterpret_cast<bf16x4*>(r1 + ep_off8<true>(ii * 16 + fr, (j * 16 + fg * 4) * 2)) = bf16x4{(bf16)hv[0], (bf16)hv[1], (bf16)hv[2], (bf16)hv[3]};
    }
    if (M3P_GQ_ABL & 2) asm volatile("" :: "v"(code));
    else st16(qout + ii * 1024 + lane * 16, code);
  }
  if (M3P_GQ_ABL & 4) return;
  if (M3P_EPI_ALDS) {
    u32x4 R[4];
    epilogue_rows_read<true, true>(r1, lane, R);
    lgkm_wait_rows<true>(R, false);
    epilogue_rows_store(C, ldc, mrow0, nw, lane, R);
    return;
  }
  bf16* Cp = C + (size_t)mrow0 * ldc + nw + sch * 8;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + srow;
    st16(Cp + (size_t)row * ldc, flip_halves(*reinterpret_cast<const u32x4*>(r1 + ep_off<true>(row, sch * 16)), it & 1));
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the lin1 + GELU + byte epilogue from ONE table read per element (M3P_GQ_LUT, default).
// The round-4 form above evaluates erf and the Gaussian on the VALU - ~17 issue slots and two quarter-rate
// transcendentals per element, with the matrix pipe idle: measured (profiles/r06_gq_ablation.txt) the arithmetic alone is
// 46-64 us of the 232-us launch.  Both results are functions of x, and to the accuracy a bf16 output and a one-byte code can
// carry they are functions of the upper 16 bits of x:
//     h(x)     = max(x, 0) - |x| T(|x|)          T = 1 - Phi, the upper Gaussian tail
//     code(x)  = c(|x|) ^ (x < 0 ? 0xFF : 0)     c = the byte code of gelu'(|x|) (common.hpp: the grid is symmetric about 1/2)
// so one 32-bit table word per bf16 magnitude holds T as an fp32 whose low mantissa byte is replaced by c (T keeps 15
// mantissa bits: relative error 2^-16, far below the bf16 rounding of h).  The index is the TRUNCATED bf16 magnitude of the
// fp32 x (one v_bfe - no rounding instruction), clamped to |x| in [2^-11, 8) (one v_med3); the entry is evaluated at the
// MIDDLE of its truncation bucket, which makes the index error that of round-to-nearest (|dx| <= 2^-9 |x|, what a bf16
// pre-activation would carry anyway).  x itself enters h in full fp32 (fma(-|x|, T, max(x, 0))).
// Per element: v_bfe, v_med3, v_lshl_add (address), ds_read_b32, v_max, v_fma, half a v_cvt_pk_bf16 and 7/4 of an
// instruction for the code (v_perm gathers the four code bytes and - selectors 9 / 11 - the four sign masks) = ~7 VALU + 1 LDS
// read; the bias rides in the accumulators (they start an output tile at the bias instead of zero), so there is no add.
// 16-row pieces: staging 2 KB per wave + 1 KB for the 8-bit copy, the 7-KB table beside them (31 of the 32 KB the two stages leave).
// ---------------------------------------------------------------------------------------------------------------------
#ifndef M3P_GQ_LUT
#define M3P_GQ_LUT 1
#endif
#ifndef M3P_GQ_NT
#define M3P_GQ_NT 1          // h rows and codes leave with non-temporal stores (r06_store_policy.txt)
#endif
// |x| in [2^-11, 8): 14 exponents x 128 mantissas = 1792 words, 7 KB (below 2^-11 the first entry serves: T = 0.4998 against
// 1/2 - 0.4 |x|, an error of 4e-4 relative in h, a fifth of its bf16 rounding; gelu' = 1/2 +- 4e-4 -> codes 128 / 127 either way).
// The 2 KB this saves against the dGELU table's range [2^-15, 8) are what lets the 8-bit copy of h have staging rows of its own.
constexpr int GQ_TAB_LO = 0x3A00, GQ_TAB_HI = 0x4100, GQ_TAB_N = GQ_TAB_HI - GQ_TAB_LO;
constexpr int GQ_TAB_BYTES = GQ_TAB_N * 4;
__device__ __forceinline__ void geluq_table_fill(uint32_t* tab, int tid, int nthreads) {
  for (int i = tid; i < GQ_TAB_N; i += nthreads) {
    // middle of the truncation bucket of bf16 magnitude GQ_TAB_LO + i (the last bucket stands for every |x| >= 8, the first
    // for every |x| < 2^-11: T = 1/2 there, code 128 - half a step from gelu'(0) = 1/2 like its mirror image 127)
    const float xm = __builtin_bit_cast(float, ((uint32_t)(GQ_TAB_LO + i) << 16) | 0x8000u);
    const float tail = 0.5f * erfcf(xm * 0.70710678118654752440f);
    const float gd = (1.0f - tail) + xm * 0.39894228040143267794f * __expf(-0.5f * xm * xm);
    tab[i] = (__builtin_bit_cast(uint32_t, tail) & 0xFFFFFF00u) | gelu_grad_code(gd);
  }
}
__device__ __forceinline__ void lds_r32(uint32_t a, uint32_t& v) { asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a)); }
// one 16-row piece (this wave's 16 x 64 block at (mrow0, nw)): `rows` = accumulators INCLUDING the bias, tab_off = LDS byte
// address of the table minus 4 * GQ_TAB_LO, r1 = the wave's 2 KB of staging rows, qout = where the piece's 1024 code bytes go
template <int O8 = 0>      // 0: no 8-bit copy of h, 1: e4m3
__device__ __forceinline__ void epilogue_piece_geluq_lut(bf16* __restrict__ C, int ldc, uint8_t* __restrict__ qout, int mrow0, int nw,
                                                         char* r1, const f32x4 (&rows)[4], int lane, uint32_t tab_off,
                                                         uint8_t* __restrict__ o8 = nullptr, int ld8 = 0, float scale8 = 1.f, float* amax = nullptr,
                                                         char* r8 = nullptr) {
  const int fr = lane & 15, fg = lane >> 4;
  const int srow = lane >> 3, sch = lane & 7;
  uint32_t w8[4];
  uint32_t w[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float xf = rows[j][r];       // (through a scalar: __builtin_bit_cast applied to a vector ELEMENT reads element 0 - seen in the ISA)
      const uint32_t xb = __builtin_bit_cast(uint32_t, xf);
      const uint32_t mag = __builtin_amdgcn_ubfe(xb, 16, 15);
      const uint32_t idx = min(max(mag, (uint32_t)GQ_TAB_LO), (uint32_t)(GQ_TAB_HI - 1));     // (v_med3_u32)
      lds_r32((idx << 2) + tab_off, w[j][r]);
    }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[0][2]), "+v"(w[0][3]), "+v"(w[1][0]), "+v"(w[1][1]), "+v"(w[1][2]), "+v"(w[1][3]),
                                        "+v"(w[2][0]), "+v"(w[2][1]), "+v"(w[2][2]), "+v"(w[2][3]), "+v"(w[3][0]), "+v"(w[3][1]), "+v"(w[3][2]), "+v"(w[3][3]));
  const uint32_t la = lds_addr(r1);
  u32x4 code;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x4 hv;
    float xs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xs[r] = rows[j][r];
      // two instructions, written out: from C the compiler canonicalises the MFMA result first (v_max x, x) and forms -|x| with
      // an extra v_or to feed a packed fma (5 issue slots per element instead of 2)
      float mx;
      asm("v_max_f32 %0, 0, %1" : "=v"(mx) : "v"(xs[r]));
      asm("v_fma_f32 %0, -|%1|, %2, %3" : "=v"(hv[r]) : "v"(xs[r]), "v"(w[j][r]), "v"(mx));
    }
    // code bytes 0 / 1 / 2 / 3 = low byte of w[j][0..3]; sign masks (0xFF where x < 0) by the sign-replicating selectors
    const uint32_t x0 = __builtin_bit_cast(uint32_t, xs[0]), x1 = __builtin_bit_cast(uint32_t, xs[1]);
    const uint32_t x2 = __builtin_bit_cast(uint32_t, xs[2]), x3 = __builtin_bit_cast(uint32_t, xs[3]);
    const uint32_t blo = __builtin_amdgcn_perm(w[j][1], w[j][0], 0x0c0c0400u), bhi = __builtin_amdgcn_perm(w[j][3], w[j][2], 0x04000c0cu);
    const uint32_t slo = __builtin_amdgcn_perm(x1, x0, 0x0c0c0b09u), shi = __builtin_amdgcn_perm(x3, x2, 0x0b090c0cu);
    code[j] = (blo | bhi) ^ (slo | shi);
    lds_w64(la + ep_off8<true>(fr, (j * 16 + fg * 4) * 2), bf16x4{(bf16)hv[0], (bf16)hv[1], (bf16)hv[2], (bf16)hv[3]});
    if (O8) w8[j] = pack8<false>(hv, scale8, *amax);
  }
  st16p<M3P_GQ_NT != 0>(qout + lane * 16, code);
  if (O8) piece8_write(r8, w8, lane);      // (staging rows of its own: its round trip shares the bf16 piece's wait)
  u32x4 R[2];
  u32x4_lds R8;
  lds_r128(la + ep_off<true>(srow, sch * 16), R[0]);
  lds_r128(la + ep_off<true>(8 + srow, sch * 16), R[1]);
  const int row8 = lane >> 2, c8 = lane & 3;
  if (O8) {
    lds_r128(lds_addr(r8) + row8 * 64 + (c8 << 4), R8);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(R[0]), "+v"(R[1]), "+v"(R8));
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(R[0]), "+v"(R[1]));
  }
  bf16* Cp = C + (size_t)mrow0 * ldc + nw + sch * 8;
  st16p<M3P_GQ_NT != 0>(Cp + (size_t)srow * ldc, R[0]);
  st16p<M3P_GQ_NT != 0>(Cp + (size_t)(8 + srow) * ldc, flip_halves(R[1], true));
  if (O8) st16p<true>(o8 + (size_t)(mrow0 + row8) * ld8 + nw + ((c8 ^ ((row8 >> 1) & 3)) << 4), u32x4{R8[0], R8[1], R8[2], R8[3]});
}

#if defined(M3P_RING_TL) || defined(M3P_W8_TL) || defined(M3P_WG_TL)
__device__ unsigned long long g_ring_tl[256 * 8 * 16];  // debug build: per-wave cycle sums of the eight-wave kernel's segments
                                                        // ([256][8][8] segments, then [256][8][8] K-tile phases of the w8 kernel)
#endif

// ---------------------------------------------------------------------------------
// NT kernel, persistent ring version (the production path): one 8-wave workgroup per CU walks
// a list of 256x128 output tiles; the K-tiles of all its output tiles form ONE continuous
// stream through three 48-KB LDS stages (144 of the CU's 160 KB):
//   * HBM -> LDS loads (global_load_lds) run TWO K-tiles ahead of the MFMAs — across output
//     tile boundaries too, so the next tile's operands arrive during this tile's epilogue —
//     and are retired with a COUNTED s_waitcnt vmcnt(6) (one K-tile = 6 LDS-DMA instructions
//     per wave stays in flight across the barrier); one raw s_barrier per K-tile;
//   * MFMA operand fragments are double-buffered per 32-deep k-step with inline-asm
//     ds_read_b128 (our wait, not the compiler's): 8 reads are issued, the 16 MFMAs of the
//     other fragment set hide them (cdna guide 5.7 form iii);
//   * no per-tile workgroup launch (measured: 3.2 us of a 17 us tile), no drained pipeline;
//   * the epilogue stages its bf16 tile in the LDS stage that was just consumed.
// Tile order: round r of the persistent loop gives XCD x the 32 consecutive tile ids
// [256 r + 32 x, +32): n-tiles of the same A row-panel share that XCD's L2.
// ---------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(512)
void gemm_nt_ring_kernel(const bf16* __restrict__ A, int lda, const bf16* __restrict__ W, int ldw,
                         bf16* __restrict__ C, int ldc, int M, int N, int K, M3PEpilogue ep,
                         int tiles_m, int tiles_n, int m_fast) {
  constexpr int BM = 256, BN = 128, NWAVES = 8;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = tiles_m * tiles_n;
  // tile id -> (tm, tn): n-fastest shares the A row-panel between neighbours (weights small
  // enough for L2/MALL); m-fastest shares the W panel instead (vocabulary projection: W = 384 MB)
  // strips pay off once a full row of n-tiles no longer fits beside the A panels (measured:
  // 24 n-tiles 805 -> 914 TF with 8-wide strips; 18 n-tiles lose ~5 %, so those stay n-fastest)
  const int n_strips = (tiles_n >= 24) ? (tiles_n + 7) / 8 : 1;
  const int strip_w = (tiles_n + n_strips - 1) / n_strips;
  // (default order: strips of ~8 n-tiles walked m-major, so the 32 consecutive tiles an XCD
  //  holds in one round form a ~4 x 8 block: 4 A row-panels + 8 W panels are live per XCD
  //  instead of 1.3 + 24 -> fewer unique bytes per step in the 4-MB L2)
  auto split_tile = [&](int t, int& tm, int& tn) {
    if (m_fast) { tn = t / tiles_m; tm = t - tn * tiles_m; return; }
    const int strip = t / (tiles_m * strip_w);
    const int rem = t - strip * tiles_m * strip_w;
    const int bn = min(strip_w, tiles_n - strip * strip_w);
    tm = rem / bn;
    tn = strip * strip_w + (rem - tm * bn);
  };
  const int nwg = gridDim.x;
  // persistent schedule: sequence index q -> tile id
  const int per_xcd = nwg >> 3;                       // workgroups per XCD (grid is a multiple of 8)
  const int slot = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  auto tile_of = [&](int q) { return q * nwg + slot; };
  const int my_tiles = (ntiles > slot) ? (ntiles - slot + nwg - 1) / nwg : 0;
  if (my_tiles == 0) return;
  // dGELU: derivative table behind the three stages (visible to everyone after the first K-tile barrier)
  float* gtab = (EPI == M3P_EPI_DGELU && M3P_DGELU_LUT) ? reinterpret_cast<float*>(smem + 3 * STAGE) : nullptr;
  if (EPI == M3P_EPI_DGELU && M3P_DGELU_LUT) gelu_grad_table_fill(gtab, tid, 512);
  const int nk = K / BK;
  const int total = my_tiles * nk;

  // ---- load cursor
  const int sr = lane >> 3, sc = (lane & 7) ^ sr;
  const bf16* a_src[4];
  const bf16* w_src[2];
  int l_q = 0, l_kt = 0;
  auto set_load_tile = [&](int q) {
    const int t = tile_of(q);
    int tm, tn;
    split_tile(t, tm, tn);
#pragma unroll
    for (int i = 0; i < 4; ++i) a_src[i] = A + (size_t)min(tm * BM + (wid + i * NWAVES) * 8 + sr, M - 1) * lda + sc * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) w_src[i] = W + (size_t)min(tn * BN + (wid + i * NWAVES) * 8 + sr, N - 1) * ldw + sc * 8;
  };
  // one K-tile of the stream = 6 LDS-DMA instructions per wave (4 of A, 2 of W).  They are
  // issued ONE AT A TIME between groups of MFMAs (issue_load(s, 0..5)): the CU's texture path
  // takes ~16 cycles per 1-KiB instruction, and a wave that sits in a burst of six cannot
  // issue its own MFMAs meanwhile (measured: 19 % of wave time in the issue burst, 21 % at the
  // barrier waiting for the waves still issuing).
  auto issue_load = [&](int s, int piece) {
    char* sa = smem + s * STAGE;
    const int k0 = l_kt * BK;
    if (piece < 4)
      __builtin_amdgcn_global_load_lds(GLB_PTR(a_src[piece] + k0), LDS_PTR(sa + (wid + piece * NWAVES) * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds(GLB_PTR(w_src[piece - 4] + k0), LDS_PTR(sa + A_BYTES + (wid + (piece - 4) * NWAVES) * 1024), 16, 0, 0);
  };
  auto load_done = [&]() {
    if (++l_kt == nk) { l_kt = 0; ++l_q; if (l_q < my_tiles) set_load_tile(l_q); }
  };
  auto stage_next = [&](int s) {
#pragma unroll
    for (int pc = 0; pc < 6; ++pc) issue_load(s, pc);
    load_done();
  };

  // ---- fragment addressing
  const int wm = wid >> 1, wn = wid & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  uint32_t a_addr[2], b_addr[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int ch = ((fg + 4 * ks) ^ (fr & 7)) * 16;
    a_addr[ks] = lds0 + (wm * 64 + fr) * ROWB + ch;
    b_addr[ks] = lds0 + A_BYTES + (wn * 64 + fr) * ROWB + ch;
  }
#define M3P_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
  auto read_set = [&](uint32_t aa, uint32_t ba, bf16x8 (&af)[4], bf16x8 (&wf)[4]) {
    M3P_DSR(wf[0], ba, 0); M3P_DSR(af[0], aa, 0);
    M3P_DSR(wf[1], ba, 2048); M3P_DSR(wf[2], ba, 4096); M3P_DSR(wf[3], ba, 6144);
    M3P_DSR(af[1], aa, 2048); M3P_DSR(af[2], aa, 4096); M3P_DSR(af[3], aa, 6144);
  };
#define M3P_LGKM0() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mfma_batch = [&](const bf16x8 (&af)[4], const bf16x8 (&wf)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
  };
  // (Tried and measured neutral-to-negative on MI355X, kept out: issuing the six LDS-DMAs one
  //  at a time between MFMAs; giving the two waves of a SIMD different burst positions —
  //  before/after MFMA batch 1, or after the mid-step barrier; a 256x256 tile with 32-deep
  //  stages.  See DESIGN.md §4 "what did not work".)

  // ---- prologue of the stream
  set_load_tile(0);
  stage_next(0);
  if (total > 1) {
    stage_next(1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  bf16x8 af0[4], wf0[4], af1[4], wf1[4];
  read_set(a_addr[0], b_addr[0], af0, wf0);
  M3P_LGKM0();
  int cur = 0;      // stage of the current K-tile
  int c_q = 0, c_kt = 0;
  const bool io_aligned = ((ldc & 7) == 0) && (((uintptr_t)C & 15) == 0) &&
                          (!(EPI == M3P_EPI_BIAS_GELU) || (((ep.ld_out2 & 7) == 0) && (((uintptr_t)ep.out2 & 15) == 0))) &&
                          (!ep.bias || (((uintptr_t)ep.bias & 15) == 0)) &&
                          (!ep.aux || (((ep.ld_aux & 7) == 0) && (((uintptr_t)ep.aux & 15) == 0)));
  f32x4 csum[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) csum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int csum_nw = -1;            // first column of the 64-wide block the sums belong to
  auto flush_csum = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sfl = csum[j][r];
        sfl += __shfl_xor(sfl, 1, 64); sfl += __shfl_xor(sfl, 2, 64);
        sfl += __shfl_xor(sfl, 4, 64); sfl += __shfl_xor(sfl, 8, 64);
        const int n = csum_nw + j * 16 + fg * 4 + r;
        if (fr == 0 && n < N) unsafeAtomicAdd(ep.colsum + n, sfl);
        csum[j][r] = 0.f;
      }
  };
#ifdef M3P_RING_TL
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tl0 = __builtin_amdgcn_s_memtime(), tl1;
#define RING_TSEG(k) do { tl1 = __builtin_amdgcn_s_memtime(); tacc[k] += tl1 - tl0; tl0 = tl1; } while (0)
#else
#define RING_TSEG(k) do { } while (0)
#endif
  for (int step = 0; step < total; ++step) {
    const int nxt = (cur == 2) ? 0 : cur + 1;
    const int nx2 = (nxt == 2) ? 0 : nxt + 1;
    const bool more2 = (step + 2 < total);
    if (more2) stage_next(nx2);
    read_set(a_addr[1] + cur * STAGE, b_addr[1] + cur * STAGE, af1, wf1);
    __builtin_amdgcn_sched_barrier(0);   // reads first, then the MFMAs that hide them
    mfma_batch(af0, wf0);
    // k-step-1 fragments are in; every LDS read of this K-tile is complete; the next K-tile
    // has landed for this wave (the 6 LDS-DMAs just issued stay in flight) -> publish
    M3P_LGKM0();
    if (more2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    read_set(a_addr[0] + nxt * STAGE, b_addr[0] + nxt * STAGE, af0, wf0);   // stale after the last K-tile: unused
    __builtin_amdgcn_sched_barrier(0);
    mfma_batch(af1, wf1);
    M3P_LGKM0();

    if (++c_kt == nk) {
      RING_TSEG(0);
      // ---- epilogue of output tile c_q; stage `cur` is free (all waves passed the barrier above)
      c_kt = 0;
      const int t = tile_of(c_q);
      ++c_q;
      int tm, tn;
      split_tile(t, tm, tn);
      const int m0 = tm * BM, n0 = tn * BN;
      const int mw = m0 + wm * 64, nw = n0 + wn * 64;
      // column sums (bias gradient of the dGELU / MUL epilogues) stay in registers across this workgroup's tiles
      // and are flushed when its column block changes: with the strip order that is once per strip, not once
      // per tile (per tile the 64 shuffles + 16 atomics cost 5.4 k cycles, and the atomics - older than the next
      // K-tile's LDS-DMAs in the vmcnt order - stalled the K loop: +29 % on the whole launch)
      if ((EPI == M3P_EPI_DGELU || EPI == M3P_EPI_MUL) && ep.colsum && csum_nw != nw) {
        if (csum_nw >= 0) flush_csum();
        csum_nw = nw;
      }
      const bool fast = io_aligned && (m0 + BM <= M) && (n0 + BN <= N);
      if (fast) {
        char* r1 = smem + cur * STAGE + wid * 6144;
        // both halves' aux rows are requested up front: the second half's round trip hides behind the first half's work
        u32x4 auxt0[4], auxt1[4];
        load_aux_rows_issue<EPI>(ep, mw, nw, lane, auxt0);
        load_aux_rows_issue<EPI>(ep, mw + 32, nw, lane, auxt1);
        auto do_half = [&](const int hf, const u32x4 (&auxt)[4]) {
          f32x4 rows[2][4];
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j) rows[ii][j] = acc[2 * hf + ii][j];
          f32x4 biasv[4];
          bf16x4 auxv[2][4];
          load_bias4<EPI>(ep, nw, lane, biasv);
          RING_TSEG(1);
          load_aux_rows_finish<EPI>(lane, r1, auxt, auxv);
#ifdef M3P_RING_TL
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
          RING_TSEG(2);
          epilogue_half<EPI>(ep, C, ldc, N, mw + 32 * hf, nw, r1, rows, biasv, auxv, lane, csum, gtab);
          RING_TSEG(3);
        };
        do_half(0, auxt0);
        do_half(1, auxt1);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            epilogue_store<EPI>(ep, C, ldc, M, N, mw + i * 16 + fr, nw + j * 16 + fg * 4, acc[i][j], csum[j]);
      }
      RING_TSEG(4);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (step + 1 < total) {
        // the staging region is the slot the next iteration's loads go into: nobody may issue
        // them before every wave has finished its LDS round trip
        M3P_LGKM0();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      RING_TSEG(5);
    }
    cur = nxt;
  }
  if ((EPI == M3P_EPI_DGELU || EPI == M3P_EPI_MUL) && ep.colsum && csum_nw >= 0) flush_csum();
#ifdef M3P_RING_TL
  RING_TSEG(0);
  if (lane == 0 && EPI == M3P_EPI_DGELU)
    for (int k = 0; k < 8; ++k) g_ring_tl[(blockIdx.x * 8 + wid) * 8 + k] = tacc[k];
#endif
#undef RING_TSEG
#undef M3P_DSR
#undef M3P_LGKM0
}


// ---------------------------------------------------------------------------------
// NT kernel, "w8" version: 256x256 output tile, EIGHT waves (2 x 4), 128x64 per wave, two 64-KB LDS stages.
// tools/gemm_timeline.py on the four-wave kernel: a wave alone on its SIMD loses ~32 clocks of matrix pipe to each of its
// 16 LDS-DMA instructions per K-tile and issues 16x16x32 MFMAs every ~18 clocks instead of 16 - 29 % of the K loop - and
// pays the whole epilogue (20 % of a K = 768 launch) with the pipe idle; neither depends on the MFMA shape.  With two
// waves per SIMD the partner's MFMAs fill those holes.  What made the earlier eight-wave kernels LDS-bound was 64x64 per
// wave (256 KB of fragment reads per K-tile and CU); 128x64 per wave reads 192 KB (94 B/clk/CU over the 2048 clocks of a
// K-tile) and halves the LDS-DMAs per wave (8 per K-tile).  This is the geometry the CDNA4 guide's 256^2 template runs.
// Pipeline per K-tile (one barrier): read k-step 1 | 32 MFMAs on k-step 0 | lgkm, vmcnt(0) (K-tile +1 was requested one
// full K-tile ago), s_barrier | request K-tile +2 into the stage just vacated | read k-step 0 of K-tile +1 | 32 MFMAs on
// k-step 1.  The epilogue stages through the vacated stage (the request for K-tile +2 waits for it on an output tile's
// last K-tile).  128 accumulator + 96 fragment registers per lane: the compiler's allocation (no literal AGPR numbers).
// ---------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ const T* uniform_ptr(const T* p) {      // a wave-uniform pointer the compiler may not know to be one
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<const T*>(((uint64_t)hi << 32) | lo);
}

template <int EPI, bool DYN = false, bool O8 = false>      // O8: BIAS_GELUQ / MULQ also leave the 8-bit copy (M3PEpilogue::out8)
__global__ __launch_bounds__(512)
void gemm_nt_w8_kernel(const bf16* __restrict__ A, int lda, const bf16* __restrict__ W, int ldw,
                       bf16* __restrict__ C, int ldc, int M, int N, int K, M3PEpilogue ep,
                       int tiles_m, int tiles_n, int* __restrict__ tile_ctr, int strip_arg) {
  constexpr int BM = 256, BN = 256, NWAVES = 8;
  constexpr int A_BYTES = BM * ROWB, STAGE = (BM + BN) * ROWB;      // 32 KB, 64 KB
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = tiles_m * tiles_n;
  const int strip_req = strip_arg & 15;
  // (round 6: nine n-tiles - the QKV projection - in three strips of three instead of one row of nine: 131.4 -> 124.1 us on
  //  one box, 132 -> 131 on two others - never slower; profiles/r06_qkv_strips.txt)
  const int n_strips = strip_req ? (tiles_n + strip_req - 1) / strip_req : ((tiles_n >= 9) ? (tiles_n + 3) / 4 : 1);
  const int strip_w = (tiles_n + n_strips - 1) / n_strips;
  const bool serpentine = (strip_arg & 16) != 0;
  auto split_tile = [&](int t, int& tm, int& tn) {
    const int strip = t / (tiles_m * strip_w);
    const int rem = t - strip * tiles_m * strip_w;
    const int bn = min(strip_w, tiles_n - strip * strip_w);
    tm = rem / bn;
    tn = strip * strip_w + (rem - tm * bn);
    if (serpentine && (strip & 1)) tm = tiles_m - 1 - tm;
  };
  const int nwg = gridDim.x;
  const int per_xcd = nwg >> 3;
  const int xcd = blockIdx.x & 7;
  const int slot = xcd * per_xcd + (blockIdx.x >> 3);
  if (slot >= ntiles) return;
  // Tile schedule.  Static: workgroup `slot` takes tiles slot, slot + nwg, ...  Dynamic (tile_ctr != NULL, data parallelism):
  // the first tile is the static one, every further tile is popped from its XCD's queue - the same tiles in the same order
  // (position p of XCD x = tile (p / per_xcd) * nwg + x * per_xcd + p % per_xcd), but a workgroup whose CU is shared with a
  // collective's kernel simply pops fewer of them instead of holding the whole launch back (profiles/r03_dynamic_tiles.txt).
  // One lane pops (an L2 atomic) while the loader is three K-tiles into a tile, hands the result to the other waves through
  // four bytes of wave 0's epilogue staging rows (idle inside the K loop) across two K-tile barriers, and the loader picks
  // it up when it moves on nk - 3 K-tiles later: nothing waits for the atomic.
  const int nk = K / BK;
  constexpr bool dyn = DYN;       // (a separate instantiation: the queue's bookkeeping costs a dozen registers)
  auto tile_of = [&](int q) { return q * nwg + slot; };
  const int my_tiles = (ntiles - slot + nwg - 1) / nwg;      // (static schedule)
  const int total = my_tiles * nk;
  int t_ld = slot, t_cmp = slot, t_next = -1;               // (dynamic schedule: tiles being loaded / computed, the popped one)
  int pop_state = 0;        // 1: popped this K-tile (wave 0 publishes after the barrier), 2: published (everyone reads after the next)
  unsigned popped = 0;
  // kSpare: the epilogue stages through the 32 KB beside the two stages (swizzled 128-byte rows: 4 KB per wave, or 2 KB
  // for the 16-row pieces of the multiply epilogues, whose derivative table lives there too) instead of the stage the
  // output tile's last K-tile vacated: the next tile's K-tile 1 is then requested on schedule and nobody has to meet at a
  // barrier after the epilogue.
  constexpr bool kSpare = M3P_W8_SPARE_EPILOGUE;
  constexpr bool kMulE = (EPI == M3P_EPI_DGELU || EPI == M3P_EPI_MUL || EPI == M3P_EPI_MULQ);
  constexpr bool kAuxE = (EPI == M3P_EPI_BIAS_DROP_RES || EPI == M3P_EPI_RES || kMulE);
  constexpr bool kGqLut = (EPI == M3P_EPI_BIAS_GELUQ) && M3P_GQ_LUT;
  constexpr int kTabBytes = (EPI == M3P_EPI_DGELU && M3P_DGELU_LUT) ? GELU_TAB_N * (int)sizeof(float) : (kGqLut ? GQ_TAB_BYTES : 0);
  float* gtab = (EPI == M3P_EPI_DGELU && M3P_DGELU_LUT) ? reinterpret_cast<float*>(smem + 2 * STAGE) : nullptr;
  if (EPI == M3P_EPI_DGELU && M3P_DGELU_LUT) gelu_grad_table_fill(gtab, tid, 512);
  if (kGqLut) geluq_table_fill(reinterpret_cast<uint32_t*>(smem + 2 * STAGE), tid, 512);      // (the prologue's barrier publishes it)
  int* const mbox = reinterpret_cast<int*>(smem + 2 * STAGE + kTabBytes);

  // ---- load cursor: one LDS-DMA = 8 rows x 128 B; piece i of wave w covers row group w + 8 i of an operand
  //      (lane -> row l >> 3, LDS slot l & 7, source chunk slot ^ (row & 7)); groups are 64 rows apart
  const int sr = lane >> 3, sc = (lane & 7) ^ sr;
  const bf16* a_src;
  const bf16* w_src;
  const size_t a_step = (size_t)64 * lda, w_step = (size_t)64 * ldw;
  int l_kt = 0, l_q = 0;
  bool l_alive = true;      // (dynamic) the load cursor points at a K-tile of this workgroup's stream
#ifndef M3P_W8_IMM
#define M3P_W8_IMM 1
#endif
#ifndef M3P_W8_SADDR
#define M3P_W8_SADDR 1
#endif
  // M3P_W8_IMM: a wave stages 32 CONSECUTIVE rows of each operand (pieces 1 KB apart in LDS) and the four pieces of an operand
  // share one M0: the destination is picked by the instruction's immediate offset (-2048 .. +1024), which moves the global
  // source by the same bytes - compensated in the source pointer (tools/probe_dma_offset.py; the four-wave kernels do the
  // same).  Per LDS-DMA that saves the M0 write and its hazard nop (probe_issue.py: ~32 of the ~70 ticks a piece costs).
#ifndef M3P_W8_BUFDMA
#define M3P_W8_BUFDMA 1     // the transfers as buffer_load_dwordx4 ... lds (resource = the wave's slice of the operand tile, scalar offset = K-tile + piece)
#endif
  __amdgpu_buffer_rsrc_t a_rsrc, w_rsrc;
  auto set_load_tile = [&](int t) {
    int tm, tn;
    split_tile(t, tm, tn);
    if (M3P_W8_IMM) {       // (uniform: the lane's part is a_lane / w_lane)
      a_src = A + (size_t)(tm * BM + wid * 32) * lda;
      w_src = W + (size_t)(tn * BN + wid * 32) * ldw;
      if (M3P_W8_BUFDMA) {
        a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(uniform_ptr(a_src)), 0, 0xffffffff, 0x00020000);
        w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(uniform_ptr(w_src)), 0, 0xffffffff, 0x00020000);
      }
    } else {
      a_src = A + (size_t)(tm * BM + wid * 8 + sr) * lda + sc * 8;
      w_src = W + (size_t)(tn * BN + wid * 8 + sr) * ldw + sc * 8;
    }
  };
  const size_t a_step8 = (size_t)8 * lda, w_step8 = (size_t)8 * ldw;
  // scalar base + 32-bit lane offset: the address form the LDS-DMA takes without any per-piece vector arithmetic
  const uint32_t a_lane = (uint32_t)(sr * lda + sc * 8) * 2u, w_lane = (uint32_t)(sr * ldw + sc * 8) * 2u;
  auto issue_load = [&](int s, int piece) {
    char* sa = smem + s * STAGE;
    const int k0 = l_kt * BK;
    if (M3P_W8_IMM && M3P_W8_BUFDMA) {
      const int pc = piece & 3;
      char* base = sa + (piece < 4 ? 0 : A_BYTES) + wid * 4096;
      const uint32_t soff = __builtin_amdgcn_readfirstlane((uint32_t)k0 * 2u + (uint32_t)pc * (uint32_t)((piece < 4 ? lda : ldw) * 16) - (uint32_t)pc * 1024u);
#define W8_LDB(IMM) __builtin_amdgcn_raw_ptr_buffer_load_lds(piece < 4 ? a_rsrc : w_rsrc, LDS_PTR(base), 16, piece < 4 ? a_lane : w_lane, soff, IMM, 0)
      switch (pc) {
        case 0: W8_LDB(0); break;
        case 1: W8_LDB(1024); break;
        case 2: W8_LDB(2048); break;
        default: W8_LDB(3072); break;
      }
#undef W8_LDB
      return;
    }
    if (M3P_W8_IMM) {
      const int pc = piece & 3;
      char* base = sa + (piece < 4 ? 0 : A_BYTES) + wid * 4096 + 2048;
      // (source compensated by the immediate: (pc - 2) * 1024 bytes = (pc - 2) * 512 elements)
      // (uniform_ptr keeps the sum scalar: otherwise the lane offset is folded in first, loop-invariantly, and every piece pays
      //  a 64-bit vector add again)
      const bf16* row = uniform_ptr((piece < 4 ? a_src + pc * a_step8 : w_src + pc * w_step8) + k0 - (pc - 2) * 512);
      uint32_t lane_off = piece < 4 ? a_lane : w_lane;
      if (M3P_W8_SADDR) asm volatile("" : "+v"(lane_off));      // (the zero-extension has to sit beside the DMA for the scalar-base form to be selected)
      const char* src = reinterpret_cast<const char*>(row) + lane_off;
      switch (pc) {
        case 0: __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(base), 16, -2048, 0); break;
        case 1: __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(base), 16, -1024, 0); break;
        case 2: __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(base), 16, 0, 0); break;
        default: __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(base), 16, 1024, 0); break;
      }
      return;
    }
    if (piece < 4)
      __builtin_amdgcn_global_load_lds(GLB_PTR(a_src + piece * a_step + k0), LDS_PTR(sa + (wid + piece * NWAVES) * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds(GLB_PTR(w_src + (piece - 4) * w_step + k0), LDS_PTR(sa + A_BYTES + (wid + (piece - 4) * NWAVES) * 1024), 16, 0, 0);
  };
  auto load_done = [&]() {
    if constexpr (!dyn) {
      if (++l_kt == nk) { l_kt = 0; ++l_q; if (l_q < my_tiles) set_load_tile(tile_of(l_q)); }
    } else {
      ++l_kt;
      if (l_kt == 3) {
        pop_state = 1;
        if (wid == 0 && lane == 0) popped = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(tile_ctr) + xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (l_kt == nk) {
        l_kt = 0;
        t_ld = t_next;
        t_next = -1;
        if (t_ld >= 0) set_load_tile(t_ld); else l_alive = false;
      }
    }
  };
  auto stage_next = [&](int s) {
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) issue_load(s, pc);
    load_done();
  };

  // ---- fragment addressing (128-B rows, chunk ^= row & 7): A fragment i at + i * 2048, W fragment j at + j * 2048
  const int wm = wid >> 2, wn = wid & 3;
  const int fr = lane & 15, fg = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  uint32_t a_addr[2], b_addr[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int ch = ((fg + 4 * ks) ^ (fr & 7)) * 16;
    a_addr[ks] = lds0 + (wm * 128 + fr) * ROWB + ch;
    b_addr[ks] = lds0 + A_BYTES + (wn * 64 + fr) * ROWB + ch;
  }
#define W8_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
  // fragments are fetched a QUARTER K-tile (16 MFMAs) ahead: four W fragments per k-step, four A fragments per half
  // (64 fragment registers instead of the 96 of a whole k-step ahead - with 128 accumulators that is the difference
  // between fitting the 256 registers two waves per SIMD leave and spilling inside the K loop)
  auto read_w = [&](uint32_t ba, bf16x8 (&wf)[4]) {
    W8_DSR(wf[0], ba, 0); W8_DSR(wf[1], ba, 2048); W8_DSR(wf[2], ba, 4096); W8_DSR(wf[3], ba, 6144);
  };
  auto read_a_lo = [&](uint32_t aa, bf16x8 (&af)[4]) {
    W8_DSR(af[0], aa, 0); W8_DSR(af[1], aa, 2048); W8_DSR(af[2], aa, 4096); W8_DSR(af[3], aa, 6144);
  };
  auto read_a_hi = [&](uint32_t aa, bf16x8 (&af)[4]) {
    W8_DSR(af[0], aa, 8192); W8_DSR(af[1], aa, 10240); W8_DSR(af[2], aa, 12288); W8_DSR(af[3], aa, 14336);
  };
#define W8_LGKM0() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (kGqLut) {      // the accumulators of an output tile start at its bias (this lane's columns nw + 16 j + 4 fg .. + 3)
    int tm0, tn0;
    split_tile(slot, tm0, tn0);
    const float* bp = ep.bias + tn0 * BN + wn * 64 + fg * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bp + j * 16);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i][j] = b;
    }
  }
#ifndef M3P_W8_SETPRIO
#define M3P_W8_SETPRIO 1
#endif
#ifndef M3P_W8_SPREAD
#define M3P_W8_SPREAD 1
#endif
#ifndef M3P_W8_INTERLEAVE
#define M3P_W8_INTERLEAVE 0
#endif
  // (ld_s, ld_p0, ld_n): with M3P_W8_INTERLEAVE the quarter's LDS-DMA pieces ld_p0 .. ld_p0 + ld_n - 1 (stage ld_s) are
  //  issued one after every fourth MFMA instead of in front of the cluster
  auto mfma_q = [&](auto half_c, const bf16x8 (&af)[4], const bf16x8 (&wf)[4], int ld_s = 0, int ld_p0 = 0, int ld_n = 0) {
    constexpr int I0 = 4 * decltype(half_c)::value;
    if (M3P_W8_SETPRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[I0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[I0 + i][j], 0, 0, 0);
      if (M3P_W8_INTERLEAVE && i < ld_n) {
        __builtin_amdgcn_sched_barrier(0);
        issue_load(ld_s, ld_p0 + i);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (M3P_W8_SETPRIO == 1) __builtin_amdgcn_s_setprio(0);
  };
  // M3P_W8_SETPRIO == 2 (A/B build): static priority for the younger half of the workgroup instead of flips around every
  // MFMA cluster (MI355X_MICROARCH.md "Two waves per SIMD", item 4)
  if (M3P_W8_SETPRIO == 2 && wid >= 4) __builtin_amdgcn_s_setprio(1);
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;

#ifdef M3P_W8_TL
  // debug build: s_memtime sums per segment (0 K loop, 1 bias + aux fetch / wait, 2 epilogue pieces, 3 restart after the
  // epilogue, 4 prologue) and the absolute start / end stamps (5, 6) of every wave  (tools/w8_timeline.py)
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tl0 = __builtin_amdgcn_s_memtime(), tl1;
  tacc[5] = tl0;
#define W8_TSEG(k) do { tl1 = __builtin_amdgcn_s_memtime(); tacc[k] += tl1 - tl0; tl0 = tl1; } while (0)
  // phases of a K-tile: stamps are issued (not waited for) in front of each quarter's closing lgkmcnt wait, behind the vmcnt
  // wait and behind the barrier; the sums are taken behind quarter 3's wait, when all of them have returned
  unsigned long long kq[8] = {0, 0, 0, 0, 0, 0, 0, 0}, kst[7];
#define W8_KSTAMP(i) asm volatile("s_memtime %0" : "=s"(kst[i]))
#define W8_KSTAMP0() do { W8_KSTAMP(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
#else
#define W8_TSEG(k) do { } while (0)
#define W8_KSTAMP(i) do { } while (0)
#define W8_KSTAMP0() do { } while (0)
#endif
  // ---- prologue: K-tiles 0 and 1 into stages 0 and 1
  set_load_tile(slot);
#if M3P_MQ_PREFETCH
  if (EPI == M3P_EPI_MULQ) {       // (the first output tile's codes: see the epilogue)
    int tm2, tn2;
    split_tile(slot, tm2, tn2);
    const uint8_t* pf = reinterpret_cast<const uint8_t*>(ep.aux) + gq_block_offset(tm2, tn2, tiles_n, wid, 0) + lane * 128;
    __builtin_amdgcn_global_load_lds(GLB_PTR(pf), LDS_PTR(smem + 2 * STAGE + 8 * 2048 + wid * 256), 4, 0, 0);
  }
#endif
  stage_next(0);
#define W8_H1 (dyn ? h1d : (step + 1 < total))
#define W8_MORE2 (dyn ? l_alive : more2)
  bool h1d = l_alive;        // (dynamic) K-tile step + 1 of the stream exists (was, or is being, requested)
  if (dyn ? h1d : (total > 1)) {
    stage_next(1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  bf16x8 fa0[4], fa1[4], fw0[4], fw1[4];
  read_w(b_addr[0], fw0);
  read_a_lo(a_addr[0], fa0);
  W8_LGKM0();
  int c_kt = 0, c_q = 0;
  const bool io_aligned = ((ldc & 7) == 0) && (((uintptr_t)C & 15) == 0) &&
                          (!(EPI == M3P_EPI_BIAS_GELU) || (((ep.ld_out2 & 7) == 0) && (((uintptr_t)ep.out2 & 15) == 0))) &&
                          (!ep.bias || (((uintptr_t)ep.bias & 15) == 0)) &&
                          (!ep.aux || (((ep.ld_aux & 7) == 0) && (((uintptr_t)ep.aux & 15) == 0)));
  f32x4 csum[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) csum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int csum_nw = -1;
  auto flush_csum = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sfl = csum[j][r];
        sfl += __shfl_xor(sfl, 1, 64); sfl += __shfl_xor(sfl, 2, 64);
        sfl += __shfl_xor(sfl, 4, 64); sfl += __shfl_xor(sfl, 8, 64);
        const int n = csum_nw + j * 16 + fg * 4 + r;
        if (fr == 0 && n < N) unsafeAtomicAdd(ep.colsum + n, sfl);
        csum[j][r] = 0.f;
      }
  };
  float amax_run = 0.f;       // (BIAS_GELUQ with an 8-bit copy: this lane's max |output| over the launch; MULQ keeps it in LDS)
  if constexpr (O8 && EPI == M3P_EPI_MULQ) reinterpret_cast<float*>(smem + 2 * STAGE + 8 * 2048)[tid] = 0.f;
  bool spread_pending = false;
#ifdef M3P_W8_TL
  W8_TSEG(4);
#endif
  W8_KSTAMP0();
  for (int step = 0; dyn || step < total; ++step) {
    const int cur = step & 1, nxt = cur ^ 1;
    const bool last_kt = (c_kt + 1 == nk);
    // K-tile step + 2 exists (it is requested during this step).  Dynamic: known once quarter 1 has advanced the load cursor
    const bool more2 = dyn ? false : (step + 2 < total);
    const uint32_t so = cur * STAGE;
    const bool pend = M3P_W8_SPREAD && spread_pending;      // the request for K-tile +1 started in the previous quarter 3
    // quarter 0: k-step 0, rows 0-63 | fetch k-step 0, rows 64-127
    if (pend && !M3P_W8_INTERLEAVE) { issue_load(nxt, 3); issue_load(nxt, 4); issue_load(nxt, 5); }
    read_a_hi(a_addr[0] + so, fa1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_q(H0{}, fa0, fw0, nxt, 3, pend ? 3 : 0);
    W8_KSTAMP(1);
    W8_LGKM0();
    // quarter 1: k-step 0, rows 64-127 | fetch k-step 1: W and rows 0-63
    if (pend && !M3P_W8_INTERLEAVE) { issue_load(nxt, 6); issue_load(nxt, 7); }
    read_w(b_addr[1] + so, fw1);
    read_a_lo(a_addr[1] + so, fa0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_q(H1{}, fa1, fw0, nxt, 6, pend ? 2 : 0);
    if (pend) { load_done(); spread_pending = false; }
    W8_KSTAMP(2);
    W8_LGKM0();
    // quarter 2: k-step 1, rows 0-63 | fetch k-step 1, rows 64-127 (the last LDS read of this K-tile)
    read_a_hi(a_addr[1] + so, fa1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_q(H0{}, fa0, fw1);
    W8_KSTAMP(3);
    // every LDS read of this K-tile is complete and K-tile +1, requested one K-tile ago, has landed for this wave
    W8_LGKM0();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    W8_KSTAMP(4);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    W8_KSTAMP(5);
    __builtin_amdgcn_sched_barrier(0);
    if (dyn && pop_state) {
      if (pop_state == 1) {
        if (wid == 0 && lane == 0) {
          const int p = per_xcd + (int)popped, q = p / per_xcd;
          const int t = q * nwg + xcd * per_xcd + (p - q * per_xcd);
          *mbox = (t < ntiles) ? t : -1;
        }
        pop_state = 2;
      } else {
        t_next = __builtin_amdgcn_readfirstlane(*mbox);
        pop_state = 0;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // quarter 3: k-step 1, rows 64-127 | request K-tile +2 into the vacated stage | fetch K-tile +1's first fragments
    // (on an output tile's last K-tile both wait for the epilogue, which stages through that stage and wants the registers)
    // (epilogues with an aux tile ask for it FIRST: vmcnt retires in order, so a load issued behind the three LDS-DMAs
    //  would wait for them too - on an output tile's last K-tile the request moves into the epilogue, behind the aux loads)
    // (M3P_MQ_EARLYREQ, A/B: the byte-decode epilogue asks for K-tile + 2 on schedule and fetches its codes BEHIND the transfers)
    constexpr bool kReqLate = kAuxE && !(M3P_MQ_EARLYREQ && EPI == M3P_EPI_MULQ);
    const bool req = W8_MORE2 && (!last_kt || (kSpare && !kReqLate));
    if (req) {
      if (!M3P_W8_SPREAD) stage_next(cur);
      else if (!M3P_W8_INTERLEAVE) { issue_load(cur, 0); issue_load(cur, 1); issue_load(cur, 2); }
      if (M3P_W8_SPREAD) spread_pending = true;
    }
    if (W8_H1 && !last_kt) {
      read_w(b_addr[0] + nxt * STAGE, fw0);
      read_a_lo(a_addr[0] + nxt * STAGE, fa0);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_q(H1{}, fa1, fw1, cur, 0, (req && M3P_W8_SPREAD) ? 3 : 0);
    W8_KSTAMP(6);
    W8_LGKM0();
#ifdef M3P_W8_TL
#pragma unroll
    for (int i = 0; i < 6; ++i) kq[i] += kst[i + 1] - kst[i];
    kq[6] += 1;
    kst[0] = kst[6];
#endif

    if (++c_kt == nk) {
      // ---- epilogue of output tile c_q through the vacated stage `cur`
      W8_TSEG(0);
      c_kt = 0;
      // (dynamic, nk >= 8: the loader moved on to the next tile two K-tiles ago; -1 behind the last one)
      const int t = dyn ? t_cmp : tile_of(c_q);
      if (dyn) t_cmp = t_ld; else ++c_q;
      int tm, tn;
      split_tile(t, tm, tn);
      const int m0 = tm * BM, n0 = tn * BN;
      const int mw = m0 + wm * 128, nw = n0 + wn * 64;
      if (kMulE && ep.colsum && csum_nw != nw) {
        if (csum_nw >= 0) flush_csum();
        csum_nw = nw;
      }
      const bool fast = io_aligned && (m0 + BM <= M) && (n0 + BN <= N);
      if constexpr (EPI == M3P_EPI_BIAS_LSE) {
        if (fast) {
          char* r1 = smem + 2 * STAGE + wid * 4096;
          f32x4 biasv[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) biasv[j] = *reinterpret_cast<const f32x4*>(ep.bias + nw + j * 16 + (lane >> 4) * 4);
          const int V = ep.ld_out2;
          float2* st = reinterpret_cast<float2*>(ep.out2) + (size_t)(tn * 4 + wn) * M;
          const bool edge = (nw + 64 > V);
#pragma unroll
          for (int hf = 0; hf < 4; ++hf) {
            epilogue_half_lse(C, ldc, st, V, mw + 32 * hf, nw, r1, acc[2 * hf], acc[2 * hf + 1], biasv, lane, edge);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        W8_TSEG(2);
      } else if constexpr (kGqLut) {
        // (the launcher admits whole, aligned tiles only: `fast` always holds)
        f32x4 bnext[4];
        {
          const bool more_tiles = dyn ? (t_cmp >= 0) : (c_q < my_tiles);       // (c_q / t_cmp already name the NEXT output tile)
          int tm2 = tm, tn2 = tn;
          if (more_tiles) split_tile(dyn ? t_cmp : tile_of(c_q), tm2, tn2);
          const float* bp = ep.bias + tn2 * BN + wn * 64 + fg * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) bnext[j] = *reinterpret_cast<const f32x4*>(bp + j * 16);
        }
        {
          char* r1 = smem + 2 * STAGE + kTabBytes + wid * 2048;
          const uint32_t tab_off = lds0 + 2 * STAGE - 4 * GQ_TAB_LO;
          uint8_t* qo = reinterpret_cast<uint8_t*>(ep.out2) + gq_block_offset(tm, tn, tiles_n, wid, 0);
          if constexpr (O8) {      // + the e4m3 copy of h for an fp8 lin2
            const float sc8 = ep.scale8 ? *ep.scale8 : 1.f;
#pragma unroll
            for (int hp = 0; hp < 8; ++hp) {
              epilogue_piece_geluq_lut<1>(C, ldc, qo + 1024 * hp, mw + 16 * hp, nw, r1, acc[hp], lane, tab_off,
                                          reinterpret_cast<uint8_t*>(ep.out8), ep.ld_out8, sc8, &amax_run,
                                          smem + 2 * STAGE + kTabBytes + 8 * 2048 + wid * 1024);
              __builtin_amdgcn_sched_barrier(0);
            }
          } else {
#pragma unroll
            for (int hp = 0; hp < 8; ++hp) {
              epilogue_piece_geluq_lut(C, ldc, qo + 1024 * hp, mw + 16 * hp, nw, r1, acc[hp], lane, tab_off);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = bnext[j];
        W8_TSEG(2);
      } else if constexpr (EPI == M3P_EPI_BIAS_GELUQ) {
        if (fast) {
          char* r1 = smem + 2 * STAGE + wid * 4096;
          f32x4 biasv[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) biasv[j] = *reinterpret_cast<const f32x4*>(ep.bias + nw + j * 16 + (lane >> 4) * 4);
          uint8_t* qo = reinterpret_cast<uint8_t*>(ep.out2) + gq_block_offset(tm, tn, tiles_n, wid, 0);
#pragma unroll
          for (int hf = 0; hf < 4; ++hf) {
            epilogue_half_geluq(C, ldc, qo + 2048 * hf, mw + 32 * hf, nw, r1, acc[2 * hf], acc[2 * hf + 1], biasv, lane);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        W8_TSEG(2);
      } else if constexpr (EPI == M3P_EPI_MULQ) {
        // byte derivative in fragment order: one 16-byte load per lane and 16-row piece, two pieces ahead (the launcher admits
        // whole tiles only, so `fast` always holds).  Measured and removed: all eight code loads
        // up front with all eight LDS-DMA pieces of K-tile +2 behind them; 32-row pieces; the codes prefetched by LDS-DMA a
        // whole K-tile ahead into the idle staging rows - all within 1 % of this form, while a build that reads no codes at
        // all is 23 us faster: what the codes cost is their 129 MB on the fabric, not their latency.
        if (fast) {
          const uint8_t* qp = reinterpret_cast<const uint8_t*>(ep.aux) + gq_block_offset(tm, tn, tiles_n, wid, 0) + lane * 16;
          char* r1 = smem + 2 * STAGE + wid * 2048;
#if M3P_MQ_ABL & 1
#define MQ_LD(off) u32x4{(uint32_t)lane * 0x01010101u, 0x40404040u + (off), 0x80808080u, 0xc0c0c0c0u}
#else
#define MQ_LD(off) (M3P_MQ_NTLOAD ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(qp + (off))) : *reinterpret_cast<const u32x4*>(qp + (off)))
#endif
          u32x4 qa = MQ_LD(0), qb = MQ_LD(1024);
          __builtin_amdgcn_sched_barrier(0);
          if (W8_MORE2 && !M3P_MQ_EARLYREQ) { issue_load(cur, 0); issue_load(cur, 1); issue_load(cur, 2); spread_pending = true; }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (O8) {      // + the 8-bit copy of dU for an fp8 dx1 product
            const float sc8 = ep.scale8 ? *ep.scale8 : 1.f;
            uint8_t* o8 = reinterpret_cast<uint8_t*>(ep.out8);
            // (this lane's running max |dU| lives in LDS between output tiles: one more register across the K loop spills here)
            float* amax_slot = reinterpret_cast<float*>(smem + 2 * STAGE + 8 * 2048) + tid;
            float amax_t = *amax_slot;
#pragma unroll
            for (int hp = 0; hp < 8; ++hp) {
              u32x4 qc = qb;
              if (hp + 2 < 8) qc = MQ_LD((hp + 2) * 1024);
              char* r8 = smem + 2 * STAGE + 8 * 2048 + 2048 + wid * 1024;
              // (e5m2 only: the copy of a GRADIENT - the launcher refuses anything else, one code path in the instantiation)
              epilogue_pieceq<2>(C, ldc, mw + 16 * hp, nw, r1, acc[hp], qa, lane, csum, o8, ep.ld_out8, sc8, &amax_t, r8);
              __builtin_amdgcn_sched_barrier(0);
              qa = qb; qb = qc;
            }
            *amax_slot = amax_t;
          } else {
#pragma unroll
          for (int hp = 0; hp < 8; hp += 2) {
            u32x4 qc = qa, qd = qb;
            if (hp + 2 < 8) qc = MQ_LD((hp + 2) * 1024);
            epilogue_pieceq(C, ldc, mw + 16 * hp, nw, r1, acc[hp], qa, lane, csum);
            __builtin_amdgcn_sched_barrier(0);
            if (hp + 3 < 8) qd = MQ_LD((hp + 3) * 1024);
            epilogue_pieceq(C, ldc, mw + 16 * (hp + 1), nw, r1, acc[hp + 1], qb, lane, csum);
            __builtin_amdgcn_sched_barrier(0);
            qa = qc; qb = qd;
          }
          }
#undef MQ_LD
#if M3P_MQ_PREFETCH
          // the NEXT output tile's codes (this wave's 8 KB = 64 lines) pulled towards this XCD's L2 a whole K loop ahead: one
          // 4-byte LDS-DMA per lane, each touching another 128-byte line, into 256 idle bytes behind the staging rows
          if (dyn ? (t_cmp >= 0) : (c_q < my_tiles)) {
            int tm2, tn2;
            split_tile(dyn ? t_cmp : tile_of(c_q), tm2, tn2);
            const uint8_t* pf = reinterpret_cast<const uint8_t*>(ep.aux) + gq_block_offset(tm2, tn2, tiles_n, wid, 0) + lane * 128;
            __builtin_amdgcn_global_load_lds(GLB_PTR(pf), LDS_PTR(smem + 2 * STAGE + 8 * 2048 + wid * 256), 4, 0, 0);
          }
#endif
        }
        W8_TSEG(2);
      } else if (fast && kSpare && kMulE) {
        // 16-row pieces, the aux rows of the next piece in flight while this one is computed
        char* r1 = smem + 2 * STAGE + kTabBytes + wid * 2048;
        u32x4 ta[2], tb[2];
        aux16_issue(ep, mw, nw, lane, ta);
        __builtin_amdgcn_sched_barrier(0);
        if (W8_MORE2) { issue_load(cur, 0); issue_load(cur, 1); issue_load(cur, 2); spread_pending = true; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int hp = 0; hp < 8; hp += 2) {
          aux16_issue(ep, mw + 16 * (hp + 1), nw, lane, tb);
          epilogue_piece16<EPI>(ep, C, ldc, mw + 16 * hp, nw, r1, acc[hp], ta, lane, csum, gtab);
          __builtin_amdgcn_sched_barrier(0);
          if (hp + 2 < 8) aux16_issue(ep, mw + 16 * (hp + 2), nw, lane, ta);
          epilogue_piece16<EPI>(ep, C, ldc, mw + 16 * (hp + 1), nw, r1, acc[hp + 1], tb, lane, csum, gtab);
          __builtin_amdgcn_sched_barrier(0);
        }
        W8_TSEG(2);
      } else if (fast) {
        char* r1 = kSpare ? smem + 2 * STAGE + wid * 4096 : smem + cur * STAGE + wid * 6144;
        f32x4 biasv[4];
        load_bias4<EPI>(ep, nw, lane, biasv);
        u32x4 aux0[4];
        if (kSpare && kAuxE) {
          load_aux_rows_issue<EPI>(ep, mw, nw, lane, aux0);
          __builtin_amdgcn_sched_barrier(0);
          if (W8_MORE2) { issue_load(cur, 0); issue_load(cur, 1); issue_load(cur, 2); spread_pending = true; }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int hf = 0; hf < 4; ++hf) {
          f32x4 rows[2][4];
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j) rows[ii][j] = acc[2 * hf + ii][j];
          bf16x4 auxv[2][4];
          constexpr bool kAlds = M3P_EPI_ALDS && kSpare;      // (staging accesses the compiler does not see: no vmcnt(0) per piece)
          if (kSpare && kAuxE && hf == 0) load_aux_rows_finish<EPI, kSpare, kAlds>(lane, r1, aux0, auxv);
          else load_aux_rows<EPI, kSpare, kAlds>(ep, mw + 32 * hf, nw, lane, r1, auxv);
#ifdef M3P_W8_TL
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          W8_TSEG(1);
#endif
          epilogue_half<EPI, kSpare, kAlds>(ep, C, ldc, N, mw + 32 * hf, nw, r1, rows, biasv, auxv, lane, csum, gtab);
          W8_TSEG(2);
          __builtin_amdgcn_sched_barrier(0);      // one piece at a time: hoisted loads of the next piece cost registers
        }
      } else {
        if (kSpare && kAuxE && W8_MORE2) { issue_load(cur, 0); issue_load(cur, 1); issue_load(cur, 2); spread_pending = true; }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            epilogue_store<EPI>(ep, C, ldc, M, N, mw + i * 16 + fr, nw + j * 16 + fg * 4, acc[i][j], csum[j]);
      }
      if constexpr (!kGqLut) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (W8_H1) {
        if (!kSpare) {
          // the staging rows are where K-tile +2 goes: nobody requests it before every wave is done with its round trip
          W8_LGKM0();
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          if (W8_MORE2) stage_next(cur);
        }
        read_w(b_addr[0] + nxt * STAGE, fw0);
        read_a_lo(a_addr[0] + nxt * STAGE, fa0);
        W8_LGKM0();
      }
      W8_TSEG(3);
      W8_KSTAMP0();
    }
    if (dyn) {
      if (!h1d) break;
      h1d = l_alive;
    }
  }
  if (kMulE && ep.colsum && csum_nw >= 0) flush_csum();
  if constexpr (O8 && (EPI == M3P_EPI_BIAS_GELUQ || EPI == M3P_EPI_MULQ)) {
    if (ep.amax8) {
      // one candidate per workgroup, and only if it beats what is there (same-address atomics serialise at the memory side:
      // optim.hip, quant_fp8_kernel); the staging rows are free now
      float* wmax = reinterpret_cast<float*>(smem + 2 * STAGE + kTabBytes);
      if constexpr (EPI == M3P_EPI_MULQ) amax_run = reinterpret_cast<float*>(smem + 2 * STAGE + 8 * 2048)[tid];
      const float mxw = wave_max(amax_run);
      __syncthreads();
      if (lane == 0) wmax[wid] = mxw;
      __syncthreads();
      if (tid == 0) {
        float mx = wmax[0];
#pragma unroll
        for (int w = 1; w < NWAVES; ++w) mx = fmaxf(mx, wmax[w]);
        unsigned int* slot = reinterpret_cast<unsigned int*>(ep.amax8);
        if (__float_as_uint(mx) > __builtin_nontemporal_load(slot)) atomicMax(slot, __float_as_uint(mx));
      }
    }
  }
#ifdef M3P_W8_TL
  tacc[6] = __builtin_amdgcn_s_memtime();
  if (lane == 0)
    for (int k = 0; k < 8; ++k) {
      g_ring_tl[(blockIdx.x * 8 + wid) * 8 + k] = tacc[k];
      g_ring_tl[256 * 8 * 8 + (blockIdx.x * 8 + wid) * 8 + k] = kq[k];
    }
#endif
#undef W8_H1
#undef W8_MORE2
#undef W8_TSEG
#undef W8_KSTAMP
#undef W8_KSTAMP0
#undef W8_DSR
#undef W8_LGKM0
}

// ---------------------------------------------------------------------------------
// fp8 NT kernel (BASELINE configs[3]: fp8 MFMA GEMMs): the eight-wave 256x256 kernel above with 8-bit operands on
// v_mfma_scale_f32_16x16x128_f8f6f4 (block scales fixed at 1: gfx950 has no unscaled K = 128 form; per-tensor scales are
// applied to the accumulators).  C (bf16) = epilogue(descale_a * descale_b * A8[M,K] x W8[N,K]^T).
// An LDS row is again 128 bytes - now 128 contraction elements - so staging, swizzle and fragment addresses are the
// bf16 kernel's; what was two 16-byte k-steps there is ONE MFMA operand here: lane group g = lane >> 4 holds the 16-byte
// chunks g and g + 4 of its row (the contraction order is permuted identically for both operands, which a sum does not
// see).  32 MFMAs of 32 clocks per K-tile and wave; quarters of 8 MFMAs (two A tiles x four W tiles) with the next
// quarter's A fragments - and, in the last quarter, the next K-tile's W fragments - fetched underneath: 96 fragment registers.
// A_BF8: the A operand is bf8 (e5m2, gradients); W is fp8 (e4m3) always.
// ---------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
template <int EPI, bool A_BF8>
__global__ __launch_bounds__(512)
void gemm_nt_w8f8_kernel(const uint8_t* __restrict__ A, int lda, const uint8_t* __restrict__ W, int ldw,
                         bf16* __restrict__ C, int ldc, int M, int N, int K, M3PEpilogue ep,
                         int tiles_m, int tiles_n) {
  constexpr int BM = 256, BN = 256, NWAVES = 8, BK8 = 128;
  constexpr int A_BYTES = BM * ROWB, STAGE = (BM + BN) * ROWB;      // 32 KB, 64 KB
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = tiles_m * tiles_n;
  constexpr int strip_arg = 0;
  const int strip_req = strip_arg & 15;
  const int n_strips = strip_req ? (tiles_n + strip_req - 1) / strip_req : ((tiles_n >= 12) ? (tiles_n + 3) / 4 : 1);
  const int strip_w = (tiles_n + n_strips - 1) / n_strips;
  const bool serpentine = (strip_arg & 16) != 0;
  auto split_tile = [&](int t, int& tm, int& tn) {
    const int strip = t / (tiles_m * strip_w);
    const int rem = t - strip * tiles_m * strip_w;
    const int bn = min(strip_w, tiles_n - strip * strip_w);
    tm = rem / bn;
    tn = strip * strip_w + (rem - tm * bn);
    if (serpentine && (strip & 1)) tm = tiles_m - 1 - tm;
  };
  const int nwg = gridDim.x;
  const int per_xcd = nwg >> 3;
  const int slot = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  auto tile_of = [&](int q) { return q * nwg + slot; };
  const int my_tiles = (ntiles > slot) ? (ntiles - slot + nwg - 1) / nwg : 0;
  if (my_tiles == 0) return;
  // the epilogue stages through the 32 KB beside the two stages, like the bf16 kernel (see there)
  constexpr bool kSpare = M3P_W8_SPARE_EPILOGUE;
  constexpr bool kMulE = (EPI == M3P_EPI_DGELU || EPI == M3P_EPI_MUL);
  constexpr bool kAuxE = (EPI == M3P_EPI_BIAS_DROP_RES || EPI == M3P_EPI_RES || kMulE);
  constexpr bool kGqLut = (EPI == M3P_EPI_BIAS_GELUQ) && M3P_GQ_LUT;
  constexpr int kTabBytes = (EPI == M3P_EPI_DGELU && M3P_DGELU_LUT) ? GELU_TAB_N * (int)sizeof(float) : (kGqLut ? GQ_TAB_BYTES : 0);
  float* gtab = (EPI == M3P_EPI_DGELU && M3P_DGELU_LUT) ? reinterpret_cast<float*>(smem + 2 * STAGE) : nullptr;
  if (EPI == M3P_EPI_DGELU && M3P_DGELU_LUT) gelu_grad_table_fill(gtab, tid, 512);
  if (kGqLut) geluq_table_fill(reinterpret_cast<uint32_t*>(smem + 2 * STAGE), tid, 512);      // (the prologue's barrier publishes it)
  const int nk = K / BK8;
  const int total = my_tiles * nk;
  // per-tensor scales of the two operands (device scalars: delayed scaling keeps them on the device)
  const float dsc = (ep.descale_a ? *ep.descale_a : 1.f) * (ep.descale_b ? *ep.descale_b : 1.f);

  const int sr = lane >> 3, sc = (lane & 7) ^ sr;
  const uint8_t* a_src;
  const uint8_t* w_src;
  const size_t a_step = (size_t)64 * lda, w_step = (size_t)64 * ldw;
  int l_q = 0, l_kt = 0;
  auto set_load_tile = [&](int q) {
    int tm, tn;
    split_tile(tile_of(q), tm, tn);
    a_src = A + (size_t)(tm * BM + wid * 8 + sr) * lda + sc * 16;
    w_src = W + (size_t)(tn * BN + wid * 8 + sr) * ldw + sc * 16;
  };
  auto issue_load = [&](int s, int piece) {
    char* sa = smem + s * STAGE;
    const int k0 = l_kt * BK8;
    if (piece < 4)
      __builtin_amdgcn_global_load_lds(GLB_PTR(a_src + piece * a_step + k0), LDS_PTR(sa + (wid + piece * NWAVES) * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds(GLB_PTR(w_src + (piece - 4) * w_step + k0), LDS_PTR(sa + A_BYTES + (wid + (piece - 4) * NWAVES) * 1024), 16, 0, 0);
  };
  auto load_done = [&]() {
    if (++l_kt == nk) { l_kt = 0; ++l_q; if (l_q < my_tiles) set_load_tile(l_q); }
  };
  auto stage_next = [&](int s) {
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) issue_load(s, pc);
    load_done();
  };

  const int wm = wid >> 2, wn = wid & 3;
  const int fr = lane & 15, fg = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  uint32_t a_addr[2], b_addr[2];        // the two 16-byte chunks (fg, fg + 4) of this lane's row
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int ch = ((fg + 4 * ks) ^ (fr & 7)) * 16;
    a_addr[ks] = lds0 + (wm * 128 + fr) * ROWB + ch;
    b_addr[ks] = lds0 + A_BYTES + (wn * 64 + fr) * ROWB + ch;
  }
  // an 8-register MFMA operand = two 16-byte LDS reads into the halves of one register tuple (plain loads: the
  // register allocator then places the halves adjacently; inline-asm reads into two 4-register values cost a copy each)
  typedef i32x8 Frag;
  auto frag = [&](uint32_t o0, uint32_t o1) {
    const i32x4 lo = *reinterpret_cast<const i32x4*>(smem + o0), hi = *reinterpret_cast<const i32x4*>(smem + o1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto read_w = [&](uint32_t so, Frag (&wf)[4]) {
    const uint32_t b0 = b_addr[0] - lds0 + so, b1 = b_addr[1] - lds0 + so;
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[j] = frag(b0 + j * 2048, b1 + j * 2048);
  };
  auto read_a = [&](auto q_c, uint32_t so, Frag (&af)[2]) {       // A tiles 2q, 2q + 1
    constexpr int Q = decltype(q_c)::value;
    const uint32_t a0 = a_addr[0] - lds0 + so + Q * 4096, a1 = a_addr[1] - lds0 + so + Q * 4096;
    af[0] = frag(a0, a1);
    af[1] = frag(a0 + 2048, a1 + 2048);
  };
#define F8_LGKM0() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // (ld_s, ld_p0, ld_n): LDS-DMA pieces issued in front of the cluster (spread over the quarters like the bf16 kernel)
  auto mfma_q = [&](auto q_c, const Frag (&af)[2], const Frag (&wf)[4]) {
    constexpr int I0 = 2 * decltype(q_c)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[I0 + i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[j], af[i], acc[I0 + i][j], 0, A_BF8 ? 1 : 0,
                                                                          0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    __builtin_amdgcn_s_setprio(0);
  };
  using Q0 = std::integral_constant<int, 0>;
  using Q1 = std::integral_constant<int, 1>;
  using Q2 = std::integral_constant<int, 2>;
  using Q3 = std::integral_constant<int, 3>;

  set_load_tile(0);
  stage_next(0);
  if (total > 1) {
    stage_next(1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  Frag fa0[2], fa1[2], fw[4];
  read_w(0, fw);
  read_a(Q0{}, 0, fa0);
  F8_LGKM0();
  int c_q = 0;
  const bool io_aligned = ((ldc & 7) == 0) && (((uintptr_t)C & 15) == 0) &&
                          (!ep.bias || (((uintptr_t)ep.bias & 15) == 0)) &&
                          (!ep.aux || (((ep.ld_aux & 7) == 0) && (((uintptr_t)ep.aux & 15) == 0)));
  f32x4 csum[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) csum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int csum_nw = -1;
  auto flush_csum = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sfl = csum[j][r];
        sfl += __shfl_xor(sfl, 1, 64); sfl += __shfl_xor(sfl, 2, 64);
        sfl += __shfl_xor(sfl, 4, 64); sfl += __shfl_xor(sfl, 8, 64);
        const int n = csum_nw + j * 16 + fg * 4 + r;
        if (fr == 0 && n < N) unsafeAtomicAdd(ep.colsum + n, sfl);
        csum[j][r] = 0.f;
      }
  };
  bool spread_pending = false;
  int step = 0;
  // One K-tile.  LAST (compile time): the output tile's last K-tile - no request for K-tile +2 and no fragment prefetch
  // (the epilogue stages through the vacated stage and wants the registers); peeling it keeps both bodies straight-line
  // code (a run-time condition around the prefetch makes phi copies of 48 fragment registers and lets MFMAs sink).
  auto ktile = [&](auto last_c) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_c)::value;
    const int cur = step & 1, nxt = cur ^ 1;
    const bool more2 = (step + 2 < total);
    const uint32_t so = cur * STAGE;
    const bool pend = spread_pending;
    if (pend) { issue_load(nxt, 3); issue_load(nxt, 4); issue_load(nxt, 5); }
    read_a(Q1{}, so, fa1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_q(Q0{}, fa0, fw);
    F8_LGKM0();
    if (pend) { issue_load(nxt, 6); issue_load(nxt, 7); load_done(); spread_pending = false; }
    read_a(Q2{}, so, fa0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_q(Q1{}, fa1, fw);
    F8_LGKM0();
    read_a(Q3{}, so, fa1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_q(Q2{}, fa0, fw);
    F8_LGKM0();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!LAST || (kSpare && !kAuxE)) {
      if (more2) { issue_load(cur, 0); issue_load(cur, 1); issue_load(cur, 2); spread_pending = true; }
    }
    if (!LAST) read_a(Q0{}, nxt * STAGE, fa0);           // (stale bytes after the very last K-tile: never used)
    __builtin_amdgcn_sched_barrier(0);
    // last quarter W-tile-major: the W fragments are single-buffered (a second set of 32 registers does not exist
    // beside 128 accumulators), so W tile j of the next K-tile is fetched as soon as the two MFMAs that read tile j
    // have been issued
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[6][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw[j], fa1[0], acc[6][j], 0, A_BF8 ? 1 : 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      acc[7][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw[j], fa1[1], acc[7][j], 0, A_BF8 ? 1 : 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      if (!LAST) {
        const uint32_t b0 = b_addr[0] - lds0 + nxt * STAGE + j * 2048, b1 = b_addr[1] - lds0 + nxt * STAGE + j * 2048;
        fw[j] = frag(b0, b1);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    F8_LGKM0();
    ++step;
  };
  for (int tq = 0; tq < my_tiles; ++tq) {
    for (int kt = 0; kt + 1 < nk; ++kt) ktile(std::false_type{});
    ktile(std::true_type{});
    {
      const int cur = (step - 1) & 1, nxt = cur ^ 1;
      const bool more2 = (step + 1 < total);       // (step was advanced past the tile's last K-tile)
      const bool more1 = (step < total);
      const int t = tile_of(c_q);
      ++c_q;
      int tm, tn;
      split_tile(t, tm, tn);
      const int m0 = tm * BM, n0 = tn * BN;
      const int mw = m0 + wm * 128, nw = n0 + wn * 64;
      if ((EPI == M3P_EPI_DGELU || EPI == M3P_EPI_MUL) && ep.colsum && csum_nw != nw) {
        if (csum_nw >= 0) flush_csum();
        csum_nw = nw;
      }
      const bool fast = io_aligned && (m0 + BM <= M) && (n0 + BN <= N);
      if (fast && kSpare && kMulE) {
        char* r1 = smem + 2 * STAGE + kTabBytes + wid * 2048;
        u32x4 ta[2], tb[2];
        aux16_issue(ep, mw, nw, lane, ta);
        __builtin_amdgcn_sched_barrier(0);
        if (more2) { issue_load(cur, 0); issue_load(cur, 1); issue_load(cur, 2); spread_pending = true; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int hp = 0; hp < 8; hp += 2) {
          f32x4 rows[4];
          aux16_issue(ep, mw + 16 * (hp + 1), nw, lane, tb);
#pragma unroll
          for (int j = 0; j < 4; ++j) rows[j] = acc[hp][j] * dsc;
          epilogue_piece16<EPI>(ep, C, ldc, mw + 16 * hp, nw, r1, rows, ta, lane, csum, gtab);
          __builtin_amdgcn_sched_barrier(0);
          if (hp + 2 < 8) aux16_issue(ep, mw + 16 * (hp + 2), nw, lane, ta);
#pragma unroll
          for (int j = 0; j < 4; ++j) rows[j] = acc[hp + 1][j] * dsc;
          epilogue_piece16<EPI>(ep, C, ldc, mw + 16 * (hp + 1), nw, r1, rows, tb, lane, csum, gtab);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if (fast) {
        char* r1 = kSpare ? smem + 2 * STAGE + wid * 4096 : smem + cur * STAGE + wid * 6144;
        f32x4 biasv[4];
        load_bias4<EPI>(ep, nw, lane, biasv);
        u32x4 aux0[4];
        if (kSpare && kAuxE) {
          load_aux_rows_issue<EPI>(ep, mw, nw, lane, aux0);
          __builtin_amdgcn_sched_barrier(0);
          if (more2) { issue_load(cur, 0); issue_load(cur, 1); issue_load(cur, 2); spread_pending = true; }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int hf = 0; hf < 4; ++hf) {
          f32x4 rows[2][4];
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j) rows[ii][j] = acc[2 * hf + ii][j] * dsc;
          bf16x4 auxv[2][4];
          if (kSpare && kAuxE && hf == 0) load_aux_rows_finish<EPI, kSpare>(lane, r1, aux0, auxv);
          else load_aux_rows<EPI, kSpare>(ep, mw + 32 * hf, nw, lane, r1, auxv);
          epilogue_half<EPI, kSpare>(ep, C, ldc, N, mw + 32 * hf, nw, r1, rows, biasv, auxv, lane, csum, gtab);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        if (kSpare && kAuxE && more2) { issue_load(cur, 0); issue_load(cur, 1); issue_load(cur, 2); spread_pending = true; }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            epilogue_store<EPI>(ep, C, ldc, M, N, mw + i * 16 + fr, nw + j * 16 + fg * 4, acc[i][j] * dsc, csum[j]);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (more1) {
        if (!kSpare) {
          F8_LGKM0();
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          if (more2) stage_next(cur);
        }
        read_w(nxt * STAGE, fw);
        read_a(Q0{}, nxt * STAGE, fa0);
        F8_LGKM0();
      }
    }
  }
  if ((EPI == M3P_EPI_DGELU || EPI == M3P_EPI_MUL) && ep.colsum && csum_nw >= 0) flush_csum();
#undef F8_LGKM0
}


struct WgCursor {
  int c, t, mt, len;   // chunk, tile, K-tile inside the chunk, K-tiles in this chunk
};
// second product of a paired weight-gradient launch (tiles_i * tiles_j == 0: none)
struct WgradProblem {
  const bf16* dY; const bf16* X; float* dW;
  int lddy, ldx, lddw, tiles_i, tiles_j;
};

// ---------------------------------------------------------------------------------
// NT kernel, stream-K version with fp32 atomic output: Cf[M,N] += alpha * A[M,K] W[N,K]^T for
// few-tile / very-long-K problems (the data gradient of the vocabulary projection:
// M = n_pred = 4864, N = 768, K = V_pad = 250 048 -> only 114 tiles of 256x128 but 3907
// K-tiles each).  The (k-chunk, tile, K-tile) stream is dealt out exactly like the
// weight-gradient kernel's: chunk = one workgroup's share, so all 256 CUs are busy and the six
// N-tiles that share an A panel walk the same K range at the same time (A — the 2.4 GB
// dlogits — is then read from HBM once instead of six times).  Same ring pipeline as
// gemm_nt_ring_kernel; partial tiles are flushed with fp32 atomics.
// ---------------------------------------------------------------------------------
// W_KN: the second operand is given as W[K, N] (row = contraction index, N contiguous) instead of
// W[N, K]: its 64 x 128 tile is staged with 256-B rows and read through ds_read_b64_tr_b16 like the X
// operand of the weight-gradient kernel, so the vocabulary data gradient dH = dlogits . E reads the
// embedding matrix as it is (no transposed copy; rows >= k_valid - the padding of V - are clamped:
// their dlogits columns are exact zeros).
template <bool W_KN>
__global__ __launch_bounds__(512)
void gemm_nt_streamk_kernel(const bf16* __restrict__ A, int lda, const bf16* __restrict__ W, int ldw,
                            float* __restrict__ Cf, int ldc, int M, int N, int K, float alpha,
                            int tiles_m, int tiles_n, int CHUNK, int k_valid) {
  constexpr int BM = 256, BN = 128, NWAVES = 8;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntile = tiles_m * tiles_n;
  const int nk = K / BK;
  const long long total_all = (long long)ntile * nk;
  const int nwg = gridDim.x;
  const int per_xcd = nwg >> 3;
  const int slot = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const long long share = (total_all + nwg - 1) / nwg;
  const long long g0l = share * slot;
  if (g0l >= total_all) return;
  const int total = (int)((g0l + share <= total_all ? g0l + share : total_all) - g0l);

  auto locate = [&](long long g) {
    WgCursor cu;
    const long long full = (long long)ntile * CHUNK;
    cu.c = (int)(g / full);
    const int rem = (int)(g - cu.c * full);
    cu.len = min(CHUNK, nk - cu.c * CHUNK);
    cu.t = rem / cu.len;
    cu.mt = rem - cu.t * cu.len;
    return cu;
  };
  auto advance = [&](WgCursor& cu) {
    if (++cu.mt == cu.len) {
      cu.mt = 0;
      if (++cu.t == ntile) { cu.t = 0; ++cu.c; cu.len = min(CHUNK, nk - cu.c * CHUNK); }
    }
  };

  const int sr = lane >> 3, sc = (lane & 7) ^ sr;
  WgCursor lc = locate(g0l);
  auto stage_next = [&](int s) {
    char* sa = smem + s * STAGE;
    char* sb = sa + A_BYTES;
    const int tm = lc.t / tiles_n, tn = lc.t - tm * tiles_n;
    const int k0 = (lc.c * CHUNK + lc.mt) * BK + sc * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = min(tm * BM + (wid + i * NWAVES) * 8 + sr, M - 1);
      __builtin_amdgcn_global_load_lds(GLB_PTR(A + (size_t)row * lda + k0), LDS_PTR(sa + (wid + i * NWAVES) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (W_KN) {
        const int rb = wid + i * NWAVES;            // 4-row group of the 64 x 128 tile
        const int row = rb * 4 + (lane >> 4);
        const int f = (row & 3) | (((row >> 3) & 1) << 2);
        const int gc = (lane & 15) ^ (f << 1);
        const int col = min(tn * (BN / 8) + gc, (N + 7) / 8 - 1) * 8;
        const int krow = min((lc.c * CHUNK + lc.mt) * BK + row, k_valid - 1);
        __builtin_amdgcn_global_load_lds(GLB_PTR(W + (size_t)krow * ldw + col), LDS_PTR(sb + rb * 1024), 16, 0, 0);
      } else {
        const int row = min(tn * BN + (wid + i * NWAVES) * 8 + sr, N - 1);
        __builtin_amdgcn_global_load_lds(GLB_PTR(W + (size_t)row * ldw + k0), LDS_PTR(sb + (wid + i * NWAVES) * 1024), 16, 0, 0);
      }
    }
    advance(lc);
  };

  const int wm = wid >> 1, wn = wid & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  uint32_t a_addr[2], b_addr[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int ch = ((fg + 4 * ks) ^ (fr & 7)) * 16;
    a_addr[ks] = lds0 + (wm * 64 + fr) * ROWB + ch;
    b_addr[ks] = lds0 + A_BYTES + (wn * 64 + fr) * ROWB + ch;
  }
  // W_KN fragments: tr16 addressing of the weight-gradient kernel's X operand (256-B rows)
  const int t_row = fg * 8 + (fr >> 2);
  const int t_sw = ((fr >> 2) | ((fg & 1) << 2)) << 1;
  uint32_t x_addr[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int qx = wn * 8 + 2 * c + ((fr & 3) >> 1);
    x_addr[c] = lds0 + A_BYTES + t_row * 256 + ((qx ^ t_sw) << 4) + ((fr & 1) << 3);
  }
#define M3P_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define M3P_TRH(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
  // ks = 0 / 1 selects the 32-deep k-step; so = byte offset of the stage
  auto read_set = [&](int ks, uint32_t so, bf16x8 (&af)[4], bf16x8 (&wf)[4], s16x4 (&wh)[4][2]) {
    const uint32_t aa = a_addr[ks] + so, ba = b_addr[ks] + so;
    if (W_KN) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (ks == 0) { M3P_TRH(wh[c][0], x_addr[c] + so, 0); M3P_TRH(wh[c][1], x_addr[c] + so, 1024); }
        else { M3P_TRH(wh[c][0], x_addr[c] + so, 8192); M3P_TRH(wh[c][1], x_addr[c] + so, 9216); }
      }
    } else {
      M3P_DSR(wf[0], ba, 0); M3P_DSR(wf[1], ba, 2048); M3P_DSR(wf[2], ba, 4096); M3P_DSR(wf[3], ba, 6144);
    }
    M3P_DSR(af[0], aa, 0); M3P_DSR(af[1], aa, 2048); M3P_DSR(af[2], aa, 4096); M3P_DSR(af[3], aa, 6144);
  };
#define M3P_LGKM0() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mfma_batch = [&](const bf16x8 (&af)[4], const bf16x8 (&wf_in)[4], const s16x4 (&wh)[4][2]) {
    bf16x8 wf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      wf[j] = W_KN ? __builtin_bit_cast(bf16x8, __builtin_shufflevector(wh[j][0], wh[j][1], 0, 1, 2, 3, 4, 5, 6, 7)) : wf_in[j];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
  };

  stage_next(0);
  if (total > 1) {
    stage_next(1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  bf16x8 af0[4], wf0[4], af1[4], wf1[4];
  s16x4 wh0[4][2], wh1[4][2];
  read_set(0, 0, af0, wf0, wh0);
  M3P_LGKM0();
  WgCursor cc = locate(g0l);
  int cur = 0;
  for (int step = 0; step < total; ++step) {
    const int nxt = (cur == 2) ? 0 : cur + 1;
    const int nx2 = (nxt == 2) ? 0 : nxt + 1;
    const bool more2 = (step + 2 < total);
    if (more2) stage_next(nx2);
    read_set(1, cur * STAGE, af1, wf1, wh1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_batch(af0, wf0, wh0);
    M3P_LGKM0();
    if (more2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    read_set(0, nxt * STAGE, af0, wf0, wh0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_batch(af1, wf1, wh1);
    M3P_LGKM0();
    if ((cc.mt + 1 == cc.len) || (step + 1 == total)) {
      const int tm = cc.t / tiles_n, tn = cc.t - tm * tiles_n;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = tm * BM + wm * 64 + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = tn * BN + wn * 64 + j * 16 + fg * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (m < M && n + r < N) unsafeAtomicAdd(Cf + (size_t)m * ldc + n + r, alpha * acc[i][j][r]);
            acc[i][j][r] = 0.f;
          }
        }
      }
    }
    advance(cc);
    cur = nxt;
  }
#undef M3P_DSR
#undef M3P_TRH
#undef M3P_LGKM0
}

// W4-BEGIN (generated by tools/gen/gen_w4.py from tools/gen/w4_template.hip - edit those)
// ---------------------------------------------------------------------------------
// NT kernel, "w4" version: 256x256 output tile, FOUR waves (one per SIMD), each wave owns a
// 128x128 sub-tile = 8x8 MFMA tiles (256 accumulator registers of the 512 a lone wave has).
// Why: with 64x64 per wave (ring kernel above) every 32-deep k-step makes a wave read
// (64+64) rows x 64 B from LDS for 16 MFMAs; eight waves then pull 128 KB of LDS reads per
// 64-deep K-tile = 1024 clocks at the LDS's 128 B/clk - exactly the 1030 clocks the MFMAs
// of that K-tile need, before the 48 KB of LDS-DMA writes are even counted: the ring kernel
// is LDS-bandwidth bound (measured 2057 clk per K-tile).  128x128 per wave halves LDS bytes
// per FLOP, 256x256 halves L2 -> LDS bytes per FLOP (64 KB per 2048 MFMA clocks = 32 B/clk/CU).
// K-tiles are 64 deep with full 128-B rows: the first version of this kernel staged 32-deep
// tiles (64-B row segments) and stalled on L2 ingest - tools/probe_ingest.py measures
// 33 B/clk/CU for LDS-DMA with 64-B segments against 64 B/clk/CU with 128-B segments.
// Pipeline (two 64-KB stages, one s_barrier per 128 MFMAs):
//   phase 1 of K-tile j: 64 MFMAs on k-step 0 (registers) | ds_read k-step 1 of tile j
//                        | the last LDS-DMAs of tile j+1
//   mid:  lgkmcnt(0), vmcnt -> tile j+1 has landed, s_barrier (everyone is done with stage j)
//   phase 2: 64 MFMAs on k-step 1 | ds_read k-step 0 of tile j+1 | first LDS-DMAs of tile j+2
//            into stage j
// The 16 LDS-DMAs of a K-tile are spread over phase 2 and the start of the next phase 1: issued
// back to back in phase 2 alone they ask the texture path for its full 64 B/clk and the issuing
// waves (alone on their SIMDs, nothing else to run) stall on the queue.  Memory instructions
// themselves are free in the shadow of an MFMA (tools/probe_issue2.py: +1 clock per 8 MFMAs).
// One M0 per operand and K-tile: LDS destinations are selected by the immediate offset, which
// moves source and destination together (tools/probe_dma_offset.py).
// The epilogue runs from a private 4.5-KB staging area per wave; the loads of the next
// output tile are already in flight under it.
// ---------------------------------------------------------------------------------
template <int EPI, bool TL = false, int ABL = 0>
__global__ __launch_bounds__(256)
void gemm_nt_w4_kernel(const bf16* __restrict__ A, int lda, const bf16* __restrict__ W, int ldw,
                       bf16* __restrict__ C, int ldc, int M, int N, int K, M3PEpilogue ep,
                       int tiles_m, int tiles_n, int m_fast, unsigned long long* __restrict__ dbg = nullptr) {
  // TL: debug instantiation that accumulates s_memtime per pipeline segment (tools/gemm_timeline.py);
  // ABL (TL only): bit 0 = no fragment reads, bit 1 = no LDS-DMA, bit 2 = every K-tile re-reads the first
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tl0 = TL ? __builtin_amdgcn_s_memtime() : 0, tl1;
#define W4_TSEG(k) do { if (TL) { tl1 = __builtin_amdgcn_s_memtime(); tacc[k] += tl1 - tl0; tl0 = tl1; } } while (0)
  constexpr int BM = 256, BN = 256, KT = 64;
  constexpr int A_BYTES = BM * KT * 2, STAGE = (BM + BN) * KT * 2;     // 32 KB, 64 KB
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = tiles_m * tiles_n;
  const int n_strips = (tiles_n >= 12) ? (tiles_n + 3) / 4 : 1;
  const int strip_w = (tiles_n + n_strips - 1) / n_strips;
  auto split_tile = [&](int t, int& tm, int& tn) {
    if (m_fast) { tn = t / tiles_m; tm = t - tn * tiles_m; return; }
    const int strip = t / (tiles_m * strip_w);
    const int rem = t - strip * tiles_m * strip_w;
    const int bn = min(strip_w, tiles_n - strip * strip_w);
    tm = rem / bn;
    tn = strip * strip_w + (rem - tm * bn);
  };
  const int nwg = gridDim.x;
  const int per_xcd = nwg >> 3;
  const int slot = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  auto tile_of = [&](int q) { return q * nwg + slot; };
  const int my_tiles = (ntiles > slot) ? (ntiles - slot + nwg - 1) / nwg : 0;
  if (my_tiles == 0) return;
  const int nk = K / KT;
  const int total = my_tiles * nk;

  // ---- load cursor.  One LDS-DMA instruction = 1 KB = 8 tile rows of 128 B; lane l fills slot
  // (l & 7) of row (l >> 3), which must hold 16-B chunk slot ^ (row & 7).  Wave w stages rows
  // [64w, 64w + 64) of each operand = one contiguous 8-KB slice per operand: M0 = slice + 4096,
  // the eight instructions differ only in the immediate -4096..3072 (sources pre-compensated:
  // the +2048 elements in the base pointers and the -512 per row group undo the immediates).
  const int l_row = lane >> 3;
  const int l_col = ((lane & 7) ^ l_row) * 8;
  // (only full tiles come here - the launcher sends ragged shapes to the ring kernel - so the
  //  eight row groups of a slice are a uniform stride apart and two pointers are enough)
  // Source of piece p = (wave-uniform pointer: this K-tile's slice + 8 p rows, less what the immediate adds) + (the lane's 32-bit
  // byte offset, one register per operand for the whole kernel): the LDS-DMA's scalar-base address form, no vector arithmetic
  // per piece and no per-piece address registers.  (Sixteen 64-bit lane addresses live across the K-tile made the compiler
  // park values in the AGPRs - which are this kernel's accumulators.)
  const uint32_t a_lane = (uint32_t)(l_row * lda + l_col) * 2u, w_lane = (uint32_t)(l_row * ldw + l_col) * 2u;
  const size_t a_step8 = (size_t)8 * lda, w_step8 = (size_t)8 * ldw;
  const bf16* a_tile;       // this wave's slice of the tile the load cursor points at, K-tile 0
  const bf16* w_tile;
  const bf16* a_base;       // ... at the cursor's K-tile
  const bf16* w_base;
#ifndef M3P_W4_BUFDMA
#define M3P_W4_BUFDMA 1
#endif
  __amdgpu_buffer_rsrc_t a_rsrc, w_rsrc;
  int l_q = 0, l_kt = 0;
  auto set_load_tile = [&](int q) {
    int tm, tn;
    split_tile(tile_of(q), tm, tn);
    a_base = a_tile = A + (size_t)(tm * BM + wid * 64) * lda;
    w_base = w_tile = W + (size_t)(tn * BN + wid * 64) * ldw;
    if (M3P_W4_BUFDMA) {
      a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(uniform_ptr(a_tile)), 0, 0xffffffff, 0x00020000);
      w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(uniform_ptr(w_tile)), 0, 0xffffffff, 0x00020000);
    }
  };
#define W4_LD1(PTR, IMM) __builtin_amdgcn_global_load_lds(GLB_PTR(PTR), LDS_PTR(sl), 16, IMM, 0)
#define W4_LDB(IMM) __builtin_amdgcn_raw_ptr_buffer_load_lds(piece < 8 ? a_rsrc : w_rsrc, LDS_PTR(sl), 16, piece < 8 ? a_lane : w_lane, soff, IMM, 0)
  // M3P_W4_BUFDMA: the transfer as buffer_load_dwordx4 ... lds (resource = this wave's slice of the operand tile, scalar offset
  // = K-tile + piece, one 32-bit lane offset for the whole kernel) instead of global_load_lds_dwordx4.  The immediate of the
  // buffer form is unsigned 12-bit: pieces 0-3 and 4-7 of an operand get an M0 each (slice, slice + 4 KB), immediates 0..3072.
  auto issue_load = [&](int s, int piece) {
    const int pc = piece & 7;
    if (M3P_W4_BUFDMA) {
      char* sl = smem + s * STAGE + (piece < 8 ? 0 : A_BYTES) + wid * 8192 + (pc >> 2) * 4096;
      const uint32_t k_off = (ABL & 4) ? 0u : (uint32_t)l_kt * (KT * 2);
      const uint32_t soff = __builtin_amdgcn_readfirstlane(k_off + (uint32_t)pc * (uint32_t)((piece < 8 ? lda : ldw) * 16) - (uint32_t)(pc & 3) * 1024u);
      switch (pc & 3) {
        case 0: W4_LDB(0); break;
        case 1: W4_LDB(1024); break;
        case 2: W4_LDB(2048); break;
        default: W4_LDB(3072); break;
      }
      return;
    }
    char* sl = smem + s * STAGE + (piece < 8 ? 0 : A_BYTES) + wid * 8192 + 4096;
    const bf16* row = uniform_ptr((piece < 8 ? a_base + pc * a_step8 : w_base + pc * w_step8) - (pc - 4) * 512);
    uint32_t lane_off = piece < 8 ? a_lane : w_lane;
    asm volatile("" : "+v"(lane_off));      // (the zero-extension has to sit beside the DMA for the scalar-base form to be selected)
    const char* src = reinterpret_cast<const char*>(row) + lane_off;
    switch (pc) {
      case 0: W4_LD1(src, -4096); break;
      case 1: W4_LD1(src, -3072); break;
      case 2: W4_LD1(src, -2048); break;
      case 3: W4_LD1(src, -1024); break;
      case 4: W4_LD1(src, 0); break;
      case 5: W4_LD1(src, 1024); break;
      case 6: W4_LD1(src, 2048); break;
      default: W4_LD1(src, 3072); break;
    }
  };
  auto load_done = [&]() {
    // (past the end of the stream the last tile's K-tiles are requested again into stages nobody reads)
    if (++l_kt == nk) { l_kt = 0; ++l_q; if (l_q < my_tiles) set_load_tile(l_q); }
    if (!(ABL & 4)) { a_base = a_tile + l_kt * KT; w_base = w_tile + l_kt * KT; }
  };

  // ---- fragment addressing (as in the ring kernel: 128-B rows, chunk ^= row & 7)
  const int wm = wid >> 1, wn = wid & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  uint32_t a_addr[2], b_addr[2];      // per k-step; fragment i at + i * 2048
  uint32_t a_addr1[2], b_addr1[2];    // the same in stage 1 (the K loop is unrolled by two: the stage is a compile-time fact)
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int ch = ((fg + 4 * ks) ^ (fr & 7)) * 16;
    a_addr[ks] = lds0 + (wm * 128 + fr) * 128 + ch;
    b_addr[ks] = lds0 + A_BYTES + (wn * 128 + fr) * 128 + ch;
    a_addr1[ks] = a_addr[ks] + STAGE;
    b_addr1[ks] = b_addr[ks] + STAGE;
  }
#define W4_DSR(dst, addr, off) do { if (!(ABL & 1)) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr)); } while (0)
#define W4_LGKM0() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

  // The 256 accumulator registers live in a[0:255] under OUR control: every MFMA and every
  // accumulator read is inline asm with literal AGPR numbers (tile (i,j) = a[(8i+j)*4 .. +3]).
  // Left to the register allocator (builtin MFMAs, or asm with "+a" operands) the compiler
  // shuffled accumulators between AGPRs, VGPRs and scratch inside the K loop.  The compiler
  // itself never allocates AGPRs in this kernel (checked in the ISA: no v_accvgpr_* outside
  // ASMSTART/ASMEND); the empty asm below makes the kernel descriptor reserve all 256.
  asm volatile("" ::: "a0", "a255");
#define W4_ACC(I, J) "a[((" #I ")*8+(" #J "))*4:((" #I ")*8+(" #J "))*4+3]"
#define W4_M(FA, FW, I, J) do { if (FIRST) asm volatile("v_mfma_f32_16x16x32_bf16 " W4_ACC(I, J) ", %1, %0, 0" :: "v"(FA[I]), "v"(FW[J])); \
                                else asm volatile("v_mfma_f32_16x16x32_bf16 " W4_ACC(I, J) ", %1, %0, " W4_ACC(I, J) :: "v"(FA[I]), "v"(FW[J])); } while (0)
#define W4_L(PIECE) do { if (!(ABL & 2)) issue_load(s_cur, PIECE); __builtin_amdgcn_sched_barrier(0); } while (0)
#define W4_LP(PIECE) do { if (!(ABL & 2) && PEND) issue_load(s_cur ^ 1, PIECE); __builtin_amdgcn_sched_barrier(0); } while (0)

  bf16x8 fa0[8], fw0[8], fa1[8], fw1[8];
#ifndef M3P_W4_SCHED2
#define M3P_W4_SCHED2 1
#endif
#define W4_WAIT_LGKM(N) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define W4_WAIT_VM(N) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define W4_BAR() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define W4_LD() do { load_done(); __builtin_amdgcn_sched_barrier(0); } while (0)
#if M3P_W4_SCHED2
  // One K-tile (tile j in stage s, tile j+1 landing in stage s^1) = 128 MFMAs with every memory instruction in their shadow.
  // What decides the schedule is the time a global -> LDS transfer is given to land:
  //   MFMA   1..15  k-step 1 of tile j: W fragments out of stage s        (k-step 0 is in registers since the previous K-tile)
  //         17..43  ... and the A fragments
  //         20      lgkmcnt + barrier: nobody reads W of stage s any more  -> W of tile j+2 starts arriving there (21..49)
  //         50      lgkmcnt(0) + barrier: nor A                            -> A of tile j+2 (53..)
  //        107      vmcnt(16) + barrier: tile j+1 (requested one K-tile ago) has landed for everyone
  //        108..123 k-step 0 of tile j+1 into the registers k-step 0 of tile j vacated at MFMA 63
  // so a transfer has between 1.2 and 1.7 K-tiles (2500-3500 clocks) to land where the two-phase form above gives the last
  // five pieces of a K-tile 46 MFMAs (740 clocks, less than an HBM miss), and the three barriers sit where their condition
  // has long been true.
  auto ktile = [&](auto first_c, auto stage_c) {
    constexpr bool FIRST0 = decltype(first_c)::value;   // first K-tile of an output tile: C operand = 0 in k-step 0
    constexpr int s_cur = decltype(stage_c)::value;
    const uint32_t ra1 = s_cur ? a_addr1[1] : a_addr[1], rb1 = s_cur ? b_addr1[1] : b_addr[1];
    const uint32_t ra0n = s_cur ? a_addr[0] : a_addr1[0], rb0n = s_cur ? b_addr[0] : b_addr1[0];
    __builtin_amdgcn_sched_barrier(0);
    {
      constexpr bool FIRST = FIRST0;
    W4_M(fa0, fw0, 0, 0);
      W4_M(fa0, fw0, 0, 1); W4_DSR(fw1[0], rb1, 0);
      W4_M(fa0, fw0, 0, 2);
      W4_M(fa0, fw0, 0, 3); W4_DSR(fw1[1], rb1, 2048);
      W4_M(fa0, fw0, 0, 4);
      W4_M(fa0, fw0, 0, 5); W4_DSR(fw1[2], rb1, 4096);
      W4_M(fa0, fw0, 0, 6);
      W4_M(fa0, fw0, 0, 7); W4_DSR(fw1[3], rb1, 6144);
      W4_M(fa0, fw0, 1, 0);
      W4_M(fa0, fw0, 1, 1); W4_DSR(fw1[4], rb1, 8192);
      W4_M(fa0, fw0, 1, 2);
      W4_M(fa0, fw0, 1, 3); W4_DSR(fw1[5], rb1, 10240);
      W4_M(fa0, fw0, 1, 4);
      W4_M(fa0, fw0, 1, 5); W4_DSR(fw1[6], rb1, 12288);
      W4_M(fa0, fw0, 1, 6);
      W4_M(fa0, fw0, 1, 7); W4_DSR(fw1[7], rb1, 14336);
      W4_M(fa0, fw0, 2, 0);
      W4_M(fa0, fw0, 2, 1); W4_DSR(fa1[0], ra1, 0);
      W4_M(fa0, fw0, 2, 2);
      W4_M(fa0, fw0, 2, 3); W4_DSR(fa1[1], ra1, 2048);
      W4_M(fa0, fw0, 2, 4); W4_WAIT_LGKM(2); W4_BAR(); W4_TSEG(0);
      W4_M(fa0, fw0, 2, 5); W4_L(8);
      W4_M(fa0, fw0, 2, 6);
      W4_M(fa0, fw0, 2, 7); W4_DSR(fa1[2], ra1, 4096);
      W4_M(fa0, fw0, 3, 0);
      W4_M(fa0, fw0, 3, 1);
      W4_M(fa0, fw0, 3, 2); W4_L(9);
      W4_M(fa0, fw0, 3, 3); W4_DSR(fa1[3], ra1, 6144);
      W4_M(fa0, fw0, 3, 4);
      W4_M(fa0, fw0, 3, 5);
      W4_M(fa0, fw0, 3, 6);
      W4_M(fa0, fw0, 3, 7); W4_DSR(fa1[4], ra1, 8192); W4_L(10);
      W4_M(fa0, fw0, 4, 0);
      W4_M(fa0, fw0, 4, 1);
      W4_M(fa0, fw0, 4, 2);
      W4_M(fa0, fw0, 4, 3); W4_DSR(fa1[5], ra1, 10240);
      W4_M(fa0, fw0, 4, 4); W4_L(11);
      W4_M(fa0, fw0, 4, 5);
      W4_M(fa0, fw0, 4, 6);
      W4_M(fa0, fw0, 4, 7); W4_DSR(fa1[6], ra1, 12288);
      W4_M(fa0, fw0, 5, 0);
      W4_M(fa0, fw0, 5, 1); W4_L(12);
      W4_M(fa0, fw0, 5, 2);
      W4_M(fa0, fw0, 5, 3); W4_DSR(fa1[7], ra1, 14336);
      W4_M(fa0, fw0, 5, 4);
      W4_M(fa0, fw0, 5, 5);
      W4_M(fa0, fw0, 5, 6); W4_L(13);
      W4_M(fa0, fw0, 5, 7);
      W4_M(fa0, fw0, 6, 0);
      W4_M(fa0, fw0, 6, 1);
      W4_M(fa0, fw0, 6, 2); W4_WAIT_LGKM(0); W4_BAR(); W4_TSEG(1);
      W4_M(fa0, fw0, 6, 3);
      W4_M(fa0, fw0, 6, 4); W4_L(14);
      W4_M(fa0, fw0, 6, 5);
      W4_M(fa0, fw0, 6, 6);
      W4_M(fa0, fw0, 6, 7);
      W4_M(fa0, fw0, 7, 0);
      W4_M(fa0, fw0, 7, 1); W4_L(15);
      W4_M(fa0, fw0, 7, 2);
      W4_M(fa0, fw0, 7, 3);
      W4_M(fa0, fw0, 7, 4);
      W4_M(fa0, fw0, 7, 5);
      W4_M(fa0, fw0, 7, 6); W4_L(0);
      W4_M(fa0, fw0, 7, 7);
    }
    {
      constexpr bool FIRST = false;
    W4_M(fa1, fw1, 0, 0);
      W4_M(fa1, fw1, 0, 1);
      W4_M(fa1, fw1, 0, 2);
      W4_M(fa1, fw1, 0, 3); W4_L(1);
      W4_M(fa1, fw1, 0, 4);
      W4_M(fa1, fw1, 0, 5);
      W4_M(fa1, fw1, 0, 6);
      W4_M(fa1, fw1, 0, 7);
      W4_M(fa1, fw1, 1, 0); W4_L(2);
      W4_M(fa1, fw1, 1, 1);
      W4_M(fa1, fw1, 1, 2);
      W4_M(fa1, fw1, 1, 3);
      W4_M(fa1, fw1, 1, 4);
      W4_M(fa1, fw1, 1, 5); W4_L(3);
      W4_M(fa1, fw1, 1, 6);
      W4_M(fa1, fw1, 1, 7);
      W4_M(fa1, fw1, 2, 0);
      W4_M(fa1, fw1, 2, 1);
      W4_M(fa1, fw1, 2, 2); W4_L(4);
      W4_M(fa1, fw1, 2, 3);
      W4_M(fa1, fw1, 2, 4);
      W4_M(fa1, fw1, 2, 5);
      W4_M(fa1, fw1, 2, 6);
      W4_M(fa1, fw1, 2, 7); W4_L(5);
      W4_M(fa1, fw1, 3, 0);
      W4_M(fa1, fw1, 3, 1);
      W4_M(fa1, fw1, 3, 2);
      W4_M(fa1, fw1, 3, 3);
      W4_M(fa1, fw1, 3, 4); W4_L(6);
      W4_M(fa1, fw1, 3, 5);
      W4_M(fa1, fw1, 3, 6);
      W4_M(fa1, fw1, 3, 7);
      W4_M(fa1, fw1, 4, 0);
      W4_M(fa1, fw1, 4, 1); W4_L(7);
      W4_M(fa1, fw1, 4, 2); W4_LD();
      W4_M(fa1, fw1, 4, 3);
      W4_M(fa1, fw1, 4, 4);
      W4_M(fa1, fw1, 4, 5);
      W4_M(fa1, fw1, 4, 6);
      W4_M(fa1, fw1, 4, 7);
      W4_M(fa1, fw1, 5, 0);
      W4_M(fa1, fw1, 5, 1);
      W4_M(fa1, fw1, 5, 2);
      W4_M(fa1, fw1, 5, 3); W4_TSEG(2); W4_WAIT_VM(16); W4_TSEG(3); W4_BAR(); W4_TSEG(4);
      W4_M(fa1, fw1, 5, 4); W4_DSR(fw0[0], rb0n, 0);
      W4_M(fa1, fw1, 5, 5); W4_DSR(fw0[1], rb0n, 2048);
      W4_M(fa1, fw1, 5, 6); W4_DSR(fw0[2], rb0n, 4096);
      W4_M(fa1, fw1, 5, 7); W4_DSR(fw0[3], rb0n, 6144);
      W4_M(fa1, fw1, 6, 0); W4_DSR(fw0[4], rb0n, 8192);
      W4_M(fa1, fw1, 6, 1); W4_DSR(fw0[5], rb0n, 10240);
      W4_M(fa1, fw1, 6, 2); W4_DSR(fw0[6], rb0n, 12288);
      W4_M(fa1, fw1, 6, 3); W4_DSR(fw0[7], rb0n, 14336);
      W4_M(fa1, fw1, 6, 4); W4_DSR(fa0[0], ra0n, 0);
      W4_M(fa1, fw1, 6, 5); W4_DSR(fa0[1], ra0n, 2048);
      W4_M(fa1, fw1, 6, 6); W4_DSR(fa0[2], ra0n, 4096);
      W4_M(fa1, fw1, 6, 7); W4_DSR(fa0[3], ra0n, 6144);
      W4_M(fa1, fw1, 7, 0); W4_DSR(fa0[4], ra0n, 8192);
      W4_M(fa1, fw1, 7, 1); W4_DSR(fa0[5], ra0n, 10240);
      W4_M(fa1, fw1, 7, 2); W4_DSR(fa0[6], ra0n, 12288);
      W4_M(fa1, fw1, 7, 3); W4_DSR(fa0[7], ra0n, 14336);
      W4_M(fa1, fw1, 7, 4);
      W4_M(fa1, fw1, 7, 5);
      W4_M(fa1, fw1, 7, 6);
      W4_M(fa1, fw1, 7, 7); W4_WAIT_LGKM(0); W4_TSEG(5);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
#else
  // phase 1: k-step 0 of the current K-tile from registers; fetch its k-step 1 fragments; finish
  // the LDS-DMA list phase 2 of the previous iteration started (`pend`)
  auto phase1 = [&](auto first_c, auto stage_c, auto pend_c) {
    constexpr bool FIRST = decltype(first_c)::value;   // first K-tile of an output tile: C operand = 0
    constexpr int s_cur = decltype(stage_c)::value;
    constexpr bool PEND = decltype(pend_c)::value;     // (false in a workgroup's very first step only)
    const uint32_t ra1 = s_cur ? a_addr1[1] : a_addr[1], rb1 = s_cur ? b_addr1[1] : b_addr[1];
    __builtin_amdgcn_sched_barrier(0);
    W4_M(fa0, fw0, 0, 0);
    W4_M(fa0, fw0, 0, 1); W4_DSR(fw1[0], rb1, 0);
    W4_M(fa0, fw0, 0, 2); W4_LP(11);
    W4_M(fa0, fw0, 0, 3); W4_DSR(fw1[1], rb1, 2048);
    W4_M(fa0, fw0, 0, 4);
    W4_M(fa0, fw0, 0, 5); W4_DSR(fw1[2], rb1, 4096);
    W4_M(fa0, fw0, 0, 6); W4_LP(12);
    W4_M(fa0, fw0, 0, 7); W4_DSR(fw1[3], rb1, 6144);
    W4_M(fa0, fw0, 1, 0);
    W4_M(fa0, fw0, 1, 1); W4_DSR(fw1[4], rb1, 8192);
    W4_M(fa0, fw0, 1, 2); W4_LP(13);
    W4_M(fa0, fw0, 1, 3); W4_DSR(fw1[5], rb1, 10240);
    W4_M(fa0, fw0, 1, 4);
    W4_M(fa0, fw0, 1, 5); W4_DSR(fw1[6], rb1, 12288);
    W4_M(fa0, fw0, 1, 6); W4_LP(14);
    W4_M(fa0, fw0, 1, 7); W4_DSR(fw1[7], rb1, 14336);
    W4_M(fa0, fw0, 2, 0);
    W4_M(fa0, fw0, 2, 1); W4_DSR(fa1[0], ra1, 0);
    W4_M(fa0, fw0, 2, 2); W4_LP(15);
    W4_M(fa0, fw0, 2, 3); W4_DSR(fa1[1], ra1, 2048);
    W4_M(fa0, fw0, 2, 4);
    W4_M(fa0, fw0, 2, 5); W4_DSR(fa1[2], ra1, 4096);
    W4_M(fa0, fw0, 2, 6);
    W4_M(fa0, fw0, 2, 7); W4_DSR(fa1[3], ra1, 6144);
    W4_M(fa0, fw0, 3, 0);
    W4_M(fa0, fw0, 3, 1); W4_DSR(fa1[4], ra1, 8192);
    W4_M(fa0, fw0, 3, 2);
    W4_M(fa0, fw0, 3, 3); W4_DSR(fa1[5], ra1, 10240);
    W4_M(fa0, fw0, 3, 4);
    W4_M(fa0, fw0, 3, 5); W4_DSR(fa1[6], ra1, 12288);
    W4_M(fa0, fw0, 3, 6);
    W4_M(fa0, fw0, 3, 7); W4_DSR(fa1[7], ra1, 14336);
    W4_M(fa0, fw0, 4, 0);
    W4_M(fa0, fw0, 4, 1);
    W4_M(fa0, fw0, 4, 2);
    W4_M(fa0, fw0, 4, 3);
    W4_M(fa0, fw0, 4, 4);
    W4_M(fa0, fw0, 4, 5);
    W4_M(fa0, fw0, 4, 6);
    W4_M(fa0, fw0, 4, 7);
    W4_M(fa0, fw0, 5, 0);
    W4_M(fa0, fw0, 5, 1);
    W4_M(fa0, fw0, 5, 2);
    W4_M(fa0, fw0, 5, 3);
    W4_M(fa0, fw0, 5, 4);
    W4_M(fa0, fw0, 5, 5);
    W4_M(fa0, fw0, 5, 6);
    W4_M(fa0, fw0, 5, 7);
    W4_M(fa0, fw0, 6, 0);
    W4_M(fa0, fw0, 6, 1);
    W4_M(fa0, fw0, 6, 2);
    W4_M(fa0, fw0, 6, 3);
    W4_M(fa0, fw0, 6, 4);
    W4_M(fa0, fw0, 6, 5);
    W4_M(fa0, fw0, 6, 6);
    W4_M(fa0, fw0, 6, 7);
    W4_M(fa0, fw0, 7, 0);
    W4_M(fa0, fw0, 7, 1);
    W4_M(fa0, fw0, 7, 2);
    W4_M(fa0, fw0, 7, 3);
    W4_M(fa0, fw0, 7, 4);
    W4_M(fa0, fw0, 7, 5);
    W4_M(fa0, fw0, 7, 6);
    W4_M(fa0, fw0, 7, 7);
    __builtin_amdgcn_sched_barrier(0);
    if (PEND) load_done();
  };
  // phase 2: k-step 1; the stage just vacated by everyone (barrier) starts receiving K-tile +2,
  // and k-step 0 of the next K-tile comes out of the other stage
  auto phase2 = [&](auto stage_c) {
    constexpr bool FIRST = false;
    constexpr int s_cur = decltype(stage_c)::value;
    const uint32_t ra0n = s_cur ? a_addr[0] : a_addr1[0], rb0n = s_cur ? b_addr[0] : b_addr1[0];
    __builtin_amdgcn_sched_barrier(0);
    W4_M(fa1, fw1, 0, 0); W4_L(0);
    W4_M(fa1, fw1, 0, 1); W4_DSR(fw0[0], rb0n, 0);
    W4_M(fa1, fw1, 0, 2);
    W4_M(fa1, fw1, 0, 3); W4_DSR(fw0[1], rb0n, 2048);
    W4_M(fa1, fw1, 0, 4);
    W4_M(fa1, fw1, 0, 5); W4_DSR(fw0[2], rb0n, 4096);
    W4_M(fa1, fw1, 0, 6); W4_L(1);
    W4_M(fa1, fw1, 0, 7); W4_DSR(fw0[3], rb0n, 6144);
    W4_M(fa1, fw1, 1, 0);
    W4_M(fa1, fw1, 1, 1); W4_DSR(fw0[4], rb0n, 8192);
    W4_M(fa1, fw1, 1, 2);
    W4_M(fa1, fw1, 1, 3); W4_DSR(fw0[5], rb0n, 10240);
    W4_M(fa1, fw1, 1, 4); W4_L(2);
    W4_M(fa1, fw1, 1, 5); W4_DSR(fw0[6], rb0n, 12288);
    W4_M(fa1, fw1, 1, 6);
    W4_M(fa1, fw1, 1, 7); W4_DSR(fw0[7], rb0n, 14336);
    W4_M(fa1, fw1, 2, 0);
    W4_M(fa1, fw1, 2, 1); W4_DSR(fa0[0], ra0n, 0);
    W4_M(fa1, fw1, 2, 2); W4_L(3);
    W4_M(fa1, fw1, 2, 3); W4_DSR(fa0[1], ra0n, 2048);
    W4_M(fa1, fw1, 2, 4);
    W4_M(fa1, fw1, 2, 5); W4_DSR(fa0[2], ra0n, 4096);
    W4_M(fa1, fw1, 2, 6);
    W4_M(fa1, fw1, 2, 7); W4_DSR(fa0[3], ra0n, 6144);
    W4_M(fa1, fw1, 3, 0); W4_L(4);
    W4_M(fa1, fw1, 3, 1); W4_DSR(fa0[4], ra0n, 8192);
    W4_M(fa1, fw1, 3, 2);
    W4_M(fa1, fw1, 3, 3); W4_DSR(fa0[5], ra0n, 10240);
    W4_M(fa1, fw1, 3, 4);
    W4_M(fa1, fw1, 3, 5); W4_DSR(fa0[6], ra0n, 12288);
    W4_M(fa1, fw1, 3, 6); W4_L(5);
    W4_M(fa1, fw1, 3, 7); W4_DSR(fa0[7], ra0n, 14336);
    W4_M(fa1, fw1, 4, 0);
    W4_M(fa1, fw1, 4, 1);
    W4_M(fa1, fw1, 4, 2);
    W4_M(fa1, fw1, 4, 3);
    W4_M(fa1, fw1, 4, 4); W4_L(6);
    W4_M(fa1, fw1, 4, 5);
    W4_M(fa1, fw1, 4, 6);
    W4_M(fa1, fw1, 4, 7);
    W4_M(fa1, fw1, 5, 0);
    W4_M(fa1, fw1, 5, 1);
    W4_M(fa1, fw1, 5, 2); W4_L(7);
    W4_M(fa1, fw1, 5, 3);
    W4_M(fa1, fw1, 5, 4);
    W4_M(fa1, fw1, 5, 5);
    W4_M(fa1, fw1, 5, 6);
    W4_M(fa1, fw1, 5, 7);
    W4_M(fa1, fw1, 6, 0); W4_L(8);
    W4_M(fa1, fw1, 6, 1);
    W4_M(fa1, fw1, 6, 2);
    W4_M(fa1, fw1, 6, 3);
    W4_M(fa1, fw1, 6, 4);
    W4_M(fa1, fw1, 6, 5);
    W4_M(fa1, fw1, 6, 6); W4_L(9);
    W4_M(fa1, fw1, 6, 7);
    W4_M(fa1, fw1, 7, 0);
    W4_M(fa1, fw1, 7, 1);
    W4_M(fa1, fw1, 7, 2);
    W4_M(fa1, fw1, 7, 3);
    W4_M(fa1, fw1, 7, 4);
    W4_M(fa1, fw1, 7, 5);
    W4_M(fa1, fw1, 7, 6); W4_L(10);
    W4_M(fa1, fw1, 7, 7);
    __builtin_amdgcn_sched_barrier(0);
  };
#endif

  // ---- prologue: K-tiles 0 and 1 into stages 0 and 1
  set_load_tile(0);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int pc = 0; pc < 16; ++pc) issue_load(t, pc);
    load_done();
  }
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  W4_DSR(fw0[0], b_addr[0], 0); W4_DSR(fw0[1], b_addr[0], 2048); W4_DSR(fw0[2], b_addr[0], 4096); W4_DSR(fw0[3], b_addr[0], 6144);
  W4_DSR(fw0[4], b_addr[0], 8192); W4_DSR(fw0[5], b_addr[0], 10240); W4_DSR(fw0[6], b_addr[0], 12288); W4_DSR(fw0[7], b_addr[0], 14336);
  W4_DSR(fa0[0], a_addr[0], 0); W4_DSR(fa0[1], a_addr[0], 2048); W4_DSR(fa0[2], a_addr[0], 4096); W4_DSR(fa0[3], a_addr[0], 6144);
  W4_DSR(fa0[4], a_addr[0], 8192); W4_DSR(fa0[5], a_addr[0], 10240); W4_DSR(fa0[6], a_addr[0], 12288); W4_DSR(fa0[7], a_addr[0], 14336);
  W4_LGKM0();

  int c_q = 0, c_kt = 0;
  const bool io_aligned = ((ldc & 7) == 0) && (((uintptr_t)C & 15) == 0) &&
                          (!(EPI == M3P_EPI_BIAS_GELU) || (((ep.ld_out2 & 7) == 0) && (((uintptr_t)ep.out2 & 15) == 0))) &&
                          (!ep.bias || (((uintptr_t)ep.bias & 15) == 0)) &&
                          (!ep.aux || (((ep.ld_aux & 7) == 0) && (((uintptr_t)ep.aux & 15) == 0)));
  char* r1 = smem + 2 * STAGE + wid * EP_HALF;
  f32x4 bias_lo[4], bias_hi[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bias_lo[j] = bias_hi[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto kstep = [&](auto stage_c, int step) {
    if (c_kt == 0) {
      // bias values of this output tile: fetched now, used after the last K-tile (a load inside
      // the epilogue is a stall with nothing to hide behind)
      int btm, btn;
      split_tile(tile_of(c_q), btm, btn);
      if (btn * BN + BN <= N && io_aligned) {
        load_bias4<EPI>(ep, btn * BN + wn * 128, lane, bias_lo);
        load_bias4<EPI>(ep, btn * BN + wn * 128 + 64, lane, bias_hi);
      }
#if M3P_W4_SCHED2
      ktile(std::true_type{}, stage_c);
    } else {
      ktile(std::false_type{}, stage_c);
    }
#else
      if (step == 0) phase1(std::true_type{}, stage_c, std::false_type{});
      else phase1(std::true_type{}, stage_c, std::true_type{});
    } else {
      phase1(std::false_type{}, stage_c, std::true_type{});
    }
    W4_TSEG(0);
    W4_LGKM0();
    W4_TSEG(1);
    // everything this wave has in flight is K-tile step+1 (and older epilogue stores)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    W4_TSEG(2);
    __builtin_amdgcn_s_barrier();      // K-tile step+1 visible to all; stage s_cur fully read by all
    asm volatile("" ::: "memory");
    W4_TSEG(3);
    phase2(stage_c);
    W4_TSEG(0);
    W4_LGKM0();
    W4_TSEG(1);
#endif
    if (++c_kt == nk) {
      // ---- epilogue of output tile c_q out of the wave-private staging area
      c_kt = 0;
      // the compiler's hazard recogniser does not see the asm MFMAs: the wait for the last
      // accumulator write before v_accvgpr_read is ours
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
      int tm, tn;
      split_tile(tile_of(c_q), tm, tn);
      ++c_q;
      const int m0 = tm * BM, n0 = tn * BN;
      const int mw = m0 + wm * 128, nw = n0 + wn * 128;
      const bool fast = io_aligned && (m0 + BM <= M) && (n0 + BN <= N);
      f32x4 csum[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) csum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#define W4_RD(II, JJ, I, J)                                                         \
  asm volatile("v_accvgpr_read_b32 %0, a[((" #I ")*8+(" #J "))*4+0]\n\t"            \
               "v_accvgpr_read_b32 %1, a[((" #I ")*8+(" #J "))*4+1]\n\t"            \
               "v_accvgpr_read_b32 %2, a[((" #I ")*8+(" #J "))*4+2]\n\t"            \
               "v_accvgpr_read_b32 %3, a[((" #I ")*8+(" #J "))*4+3]"                \
               : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3));                           \
  rows[II][JJ] = f32x4{t0, t1, t2, t3}
#define W4_SLICE(RG, CH)                                                                      \
  W4_RD(0, 0, 2 * RG, 4 * CH); W4_RD(0, 1, 2 * RG, 4 * CH + 1); W4_RD(0, 2, 2 * RG, 4 * CH + 2); W4_RD(0, 3, 2 * RG, 4 * CH + 3); \
  W4_RD(1, 0, 2 * RG + 1, 4 * CH); W4_RD(1, 1, 2 * RG + 1, 4 * CH + 1); W4_RD(1, 2, 2 * RG + 1, 4 * CH + 2); W4_RD(1, 3, 2 * RG + 1, 4 * CH + 3)
#ifndef M3P_W4_PIPE_EPI
#define M3P_W4_PIPE_EPI 1
#endif
      // The plain epilogues with LDS accesses the compiler does not see (see lds_w64 ...): with transfers for the next output
      // tile in flight it puts s_waitcnt vmcnt(0) in front of every staging access it knows of, i.e. every piece waits for the
      // previous piece's global stores (~1150 clocks a piece, 9.2 k per output tile, a fifth of a K = 768 launch -
      // tools/gemm_timeline.py).  One swizzled 4-KB buffer inside each wave's 4.5-KB staging area (146 KB of LDS in all: a
      // collective's kernel can still share the CU).
      constexpr bool kPipe = M3P_W4_PIPE_EPI && (EPI == M3P_EPI_NONE || EPI == M3P_EPI_BIAS || EPI == M3P_EPI_RES || EPI == M3P_EPI_BIAS_DROP_RES);
      if (kPipe && fast) {
        char* rb = smem + 2 * STAGE + wid * EP_HALF;      // (one swizzled 4-KB buffer inside the wave's 4.5-KB staging area)
        u32x4 tq[4];
        load_aux_rows_issue<EPI>(ep, mw, nw, lane, tq);
#pragma nounroll
        for (int p = 0; p < 8; ++p) {
          const int ch = p >> 2, rg = p & 3;
          f32x4 rows[2][4];
          float t0, t1, t2, t3;
          switch (p) {
            case 0: W4_SLICE(0, 0); break;
            case 1: W4_SLICE(1, 0); break;
            case 2: W4_SLICE(2, 0); break;
            case 3: W4_SLICE(3, 0); break;
            case 4: W4_SLICE(0, 1); break;
            case 5: W4_SLICE(1, 1); break;
            case 6: W4_SLICE(2, 1); break;
            default: W4_SLICE(3, 1); break;
          }
          char* rc = rb;     // (one buffer: a wave's LDS instructions complete in order, piece p + 1's writes queue behind piece p's reads)
          bf16x4 aux_cur[2][4];
          load_aux_rows_finish<EPI, true, true>(lane, rc, tq, aux_cur);
          if (p + 1 < 8) load_aux_rows_issue<EPI>(ep, mw + 32 * ((p + 1) & 3), nw + 64 * ((p + 1) >> 2), lane, tq);
          bf16x4 ukeep[2][4];
          epilogue_half_write<EPI, true, true>(ep, N, mw + 32 * rg, nw + 64 * ch, rc, rows, ch ? bias_hi : bias_lo, aux_cur, lane, csum, ukeep);
          // (read back at once: straight-line code between the asm reads and their wait, so that no compiler-made copy of the
          //  destination registers can slip in between; what the asm accesses buy is the absence of vmcnt(0) - the stores of
          //  piece p are in flight under piece p + 1)
          u32x4 R[4];
          epilogue_rows_read<true, true>(rc, lane, R);
          lgkm_wait_rows<true>(R, false);
          epilogue_rows_store(C, ldc, mw + 32 * rg, nw + 64 * ch, lane, R);
        }
      } else
#pragma nounroll
      for (int p = 0; p < 8; ++p) {
        const int ch = p >> 2, rg = p & 3;        // column half outer: the bias-gradient sums run over rows
        // (fetching the residual / pre-activation tile of piece p+1 during piece p was measured
        //  neutral for the dropout-residual epilogue and 3 % slower for the plain residual one)
        bf16x4 aux_cur[2][4];
        if (fast) {
          // (dGELU / MUL here are never launched - launch_nt keeps it on the eight-wave kernel - and the 16 extra
          //  transient registers of the row-wise fetch make the compiler spill into our AGPRs)
          if (EPI == M3P_EPI_DGELU || EPI == M3P_EPI_MUL) load_aux<EPI>(ep, mw + 32 * rg, nw + 64 * ch, lane, aux_cur);
          else load_aux_rows<EPI>(ep, mw + 32 * rg, nw + 64 * ch, lane, r1, aux_cur);
        }
        f32x4 rows[2][4];
        float t0, t1, t2, t3;
        switch (p) {
          case 0: W4_SLICE(0, 0); break;
          case 1: W4_SLICE(1, 0); break;
          case 2: W4_SLICE(2, 0); break;
          case 3: W4_SLICE(3, 0); break;
          case 4: W4_SLICE(0, 1); break;
          case 5: W4_SLICE(1, 1); break;
          case 6: W4_SLICE(2, 1); break;
          default: W4_SLICE(3, 1); break;
        }
        const int mrow0 = mw + 32 * rg, ncol0 = nw + 64 * ch;
        if (fast) {
          epilogue_half<EPI>(ep, C, ldc, N, mrow0, ncol0, r1, rows, ch ? bias_hi : bias_lo, aux_cur, lane, csum);
        } else {
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
              epilogue_store<EPI>(ep, C, ldc, M, N, mrow0 + ii * 16 + fr, ncol0 + jj * 16 + fg * 4, rows[ii][jj], csum[jj]);
        }
        if ((EPI == M3P_EPI_DGELU || EPI == M3P_EPI_MUL) && rg == 3) {
          if (ep.colsum) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float sfl = csum[j][r];
                sfl += __shfl_xor(sfl, 1, 64); sfl += __shfl_xor(sfl, 2, 64);
                sfl += __shfl_xor(sfl, 4, 64); sfl += __shfl_xor(sfl, 8, 64);
                const int n = ncol0 + j * 16 + fg * 4 + r;
                if (fr == 0 && n < N) unsafeAtomicAdd(ep.colsum + n, sfl);
              }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) csum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
#undef W4_SLICE
#undef W4_RD
      W4_TSEG(M3P_W4_SCHED2 ? 6 : 4);
    }
  };
  for (int step = 0; step < total; step += 2) {
    kstep(std::integral_constant<int, 0>{}, step);
    if (step + 1 < total) kstep(std::integral_constant<int, 1>{}, step + 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // junk loads of the tail must not outlive the LDS allocation
  if (TL) {
    W4_TSEG(M3P_W4_SCHED2 ? 7 : 6);
    if (lane == 0)
      for (int k = 0; k < 8; ++k) dbg[((size_t)blockIdx.x * 8 + wid) * 8 + k] = tacc[k];
  }
#undef W4_TSEG
#undef W4_LD1
#undef W4_LDB
#undef W4_ACC
#undef W4_DSR
#undef W4_LGKM0
#undef W4_M
#undef W4_L
#undef W4_LP
#undef W4_WAIT_LGKM
#undef W4_WAIT_VM
#undef W4_BAR
#undef W4_LD
}

// W4-END

// ---------------------------------------------------------------------------------
// NT kernel for a handful of rows (M <= 128: one decoding step of the causal decoder, heads on a batch of [CLS] rows):
// latency / HBM work - every weight row is read once, straight from global memory into the MFMA operand registers (a
// lane's 16-byte fragment is 8 consecutive contraction elements of one W row), no LDS staging, no persistent schedule.
// SPLITK: a workgroup owns 16 output columns and its four waves a quarter of the contraction each (small N: 48
// workgroups x 4 waves for N = 768 instead of 12), partial sums folded through LDS.  Otherwise a workgroup owns 64
// columns, one wave per 16 (the vocabulary projection: N = 250 002).  Epilogues as everywhere (epilogue_store).
// ---------------------------------------------------------------------------------
template <int EPI, int MT, bool SPLITK>
__global__ __launch_bounds__(256)
void gemm_nt_skinny_kernel(const bf16* __restrict__ A, int lda, const bf16* __restrict__ W, int ldw,
                           bf16* __restrict__ C, int ldc, int M, int N, int K, M3PEpilogue ep) {
  __shared__ f32x4 red[SPLITK ? 3 : 1][MT][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int n0 = SPLITK ? blockIdx.x * 16 : (blockIdx.x * 4 + wid) * 16;
  const int klen = SPLITK ? K / 4 : K;
  const int kbeg = SPLITK ? wid * klen : 0;
  if (!SPLITK && n0 >= N) return;
  const bf16* wp = W + (size_t)min(n0 + fr, N - 1) * ldw + kbeg + fg * 8;
  const bf16* ap[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) ap[i] = A + (size_t)min(i * 16 + fr, M - 1) * lda + kbeg + fg * 8;
  f32x4 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int k = 0; k < klen; k += 32) {
    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wp + k);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(ap[i] + k);
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af, acc[i], 0, 0, 0);
    }
  }
  if (SPLITK) {
    if (wid > 0) {
#pragma unroll
      for (int i = 0; i < MT; ++i) red[wid - 1][i][lane] = acc[i];
    }
    __syncthreads();
    if (wid > 0) return;
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] += red[0][i][lane] + red[1][i][lane] + red[2][i][lane];
  }
  f32x4 unused = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < MT; ++i) epilogue_store<EPI>(ep, C, ldc, M, N, i * 16 + fr, n0 + fg * 4, acc[i], unused);
}

template <int EPI, int MT>
int launch_nt_skinny(const bf16* A, int lda, const bf16* W, int ldw, bf16* C, int ldc, int M, int N, int K,
                     const M3PEpilogue& ep, hipStream_t st) {
  if (N <= 8192)
    hipLaunchKernelGGL((gemm_nt_skinny_kernel<EPI, MT, true>), dim3((N + 15) / 16), dim3(256), 0, st, A, lda, W, ldw, C, ldc, M, N, K, ep);
  else
    hipLaunchKernelGGL((gemm_nt_skinny_kernel<EPI, MT, false>), dim3((N + 63) / 64), dim3(256), 0, st, A, lda, W, ldw, C, ldc, M, N, K, ep);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

static int hw_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
    n -= n % 8;   // the persistent schedule deals tiles out per XCD
    if (n <= 0) n = 8;
  }
  return n;
}
// workgroups of a persistent grid: one per CU, or what m3p_set_persistent_grid left of them (data parallelism: RCCL's
// kernels need CUs of their own - a persistent GEMM workgroup fills its CU's LDS, nothing can co-reside with it)
static int g_persistent_grid = 0;
static int num_cus() {
  const int hw = hw_cus();
  return (g_persistent_grid > 0 && g_persistent_grid < hw) ? g_persistent_grid : hw;
}

// dynamic tile queues of the eight-wave kernel (m3p_set_tile_queue): a ring of 8-counter slots, one slot per launch
// The ring belongs to ONE stream - the first one a queued launch arrives on after m3p_set_tile_queue (slots are cleared by a
// memset on that stream, behind everything it has launched): a launch on any other stream takes the static schedule, so a
// slot is never handed out while a kernel of another stream may still be popping from it.  Slot hand-out is under a mutex
// (forward launches from the Python thread, backward from the autograd thread; ctypes drops the GIL around the calls).
static int* g_tq_pool = nullptr;
static int g_tq_slots = 0, g_tq_next = 0;
static hipStream_t g_tq_stream = nullptr;
static bool g_tq_bound = false;
static std::mutex g_tq_mu;
// -> counters of a fresh slot, or nullptr (no queue / foreign stream / memset failed: *err set)
static int* tq_acquire(hipStream_t st, hipError_t* err) {
  *err = hipSuccess;
  std::lock_guard<std::mutex> lk(g_tq_mu);
  if (!g_tq_pool) return nullptr;
  if (!g_tq_bound) { g_tq_stream = st; g_tq_bound = true; }
  if (st != g_tq_stream) return nullptr;
  if (g_tq_next == g_tq_slots) {       // ring used up: clear it behind everything this stream has launched so far
    *err = hipMemsetAsync(g_tq_pool, 0, (size_t)g_tq_slots * 8 * sizeof(int), st);
    if (*err != hipSuccess) return nullptr;
    g_tq_next = 0;
  }
  return g_tq_pool + 8 * g_tq_next++;
}

#ifndef M3P_W8_MAX_W_BYTES
#define M3P_W8_MAX_W_BYTES (1LL << 40)      // (no limit: the vocabulary projection - W = 384 MB - measured 2.18 -> 1.85 ms on this kernel)
#endif
#ifndef M3P_W4_MIN_K
#define M3P_W4_MIN_K 2048
#endif
static bool tq_enabled() {
  std::lock_guard<std::mutex> lk(g_tq_mu);
  return g_tq_pool != nullptr;
}

// ---------------------------------------------------------------------------------------------------------------------
// WHICH KERNEL RUNS AN NT PRODUCT - the one place that decides (m3p_gemm_nt_plan reports it; launch_nt switches on it).
// Rows are tried top to bottom; "whole tiles" = M >= 1024, M % 256 == 0, N % 256 == 0, N >= 512, K % 64 == 0 (the entry point has
// already checked K % 64, lda / ldw % 8 and 16-byte operand bases).
//
//   shape / epilogue                                      kernel                     why (measured where)
//   ----------------------------------------------------  -------------------------  -------------------------------------------------
//   MULQ / BIAS_GELUQ / BIAS_LSE (whole tiles only,       eight-wave 256x256         their byte / statistics layouts ARE that kernel's
//     anything else -> M3P_EINVAL)                                                     (tile, wave, row block, lane) order
//   M <= 128, K % 128 == 0, not DGELU / MUL               skinny (no LDS)            one decoding step / [CLS] rows: every W row read once
//   whole tiles, K >= 2048, not DGELU / MUL,              four-wave 256x256          K-tile in one piece, buffer-form transfers: dx1 159 vs
//     no tile queue armed                                                              174 us, dh 125 vs 131, lin2 fwd 169 vs 177 (round 4)
//   whole tiles (any K)                                   eight-wave 256x256         second wave per SIMD hides epilogue + LDS-DMA issue:
//     + tile queue armed, > 1 tile per workgroup,           (queue instantiation)      q/k/v 137 vs 148 us at K = 768; vocabulary projection
//       K >= 512, not DGELU / MUL                                                      1.85 vs 2.18 ms
//   M >= 1024 otherwise (ragged M or N, N < 512)          ring 256x128, 3 stages     any shape; strip order for the vocabulary matrices
//   M < 1024                                              128x128, 2 stages          small-M fallback (cfg1, tests)
//
// Developer overrides (m3p_debug_set_variant, never set by the product): 0 forces the last row, 2 the four-wave kernel wherever it
// applies, 3 the ring kernel, 6 the eight-wave kernel, 7 round 1's choice (four-wave / ring), 9 = auto without the skinny kernel.
// ---------------------------------------------------------------------------------------------------------------------
static int nt_plan(int epi, int M, int N, int K, int* queue) {
  if (queue) *queue = 0;
  const bool whole = M >= 1024 && (M % 256) == 0 && (N % 256) == 0 && N >= 512 && (K % 64) == 0;
  if (epi == M3P_EPI_MULQ || epi == M3P_EPI_BIAS_GELUQ || epi == M3P_EPI_BIAS_LSE) return whole ? M3P_KERN_NT_W8 : M3P_EINVAL;
  const bool mul_epi = epi == M3P_EPI_DGELU || epi == M3P_EPI_MUL;
  if (M <= 128 && g_variant >= 1 && g_variant != 9 && (K % 128) == 0 && !mul_epi) return M3P_KERN_NT_SKINNY;
  // (dGELU epilogue on the four-wave kernel: 240 VGPRs, measured slower in the step)
  const bool deep = N >= 512 && 2LL * N * K <= (64LL << 20) && epi != M3P_EPI_DGELU;
  const bool armed = tq_enabled();     // (the tile queue of data parallelism lives in the eight-wave kernel)
  const bool w4_first = g_variant == 1 && deep && K >= M3P_W4_MIN_K && epi != M3P_EPI_MUL && !armed;
  if (whole && (g_variant == 1 || g_variant == 6) && !w4_first && 2LL * N * K <= M3P_W8_MAX_W_BYTES) {
    // dynamic tile queue only where a workgroup takes several tiles (not for the multiply epilogues: with the queue's
    // bookkeeping on top of their aux pieces and column sums the instantiation spills ~100 bytes per lane)
    int grid = num_cus();
    const int ntiles = (M / 256) * (N / 256);
    if (ntiles < grid) grid = (ntiles + 7) / 8 * 8;
    if (queue) *queue = (!mul_epi && armed && ntiles > grid && K >= 8 * BK) ? 1 : 0;
    return M3P_KERN_NT_W8;
  }
  if (M >= 1024 && (M % 256) == 0 && (N % 256) == 0 && (g_variant == 2 || ((g_variant == 1 || g_variant == 7) && deep)))
    return M3P_KERN_NT_W4;
  if (M >= 1024 && g_variant >= 1) return M3P_KERN_NT_RING;
  return M3P_KERN_NT_128;
}

template <int EPI>
int launch_nt(const bf16* A, int lda, const bf16* W, int ldw, bf16* C, int ldc, int M, int N, int K,
              const M3PEpilogue& ep, hipStream_t st) {
  int want_queue = 0;
  const int plan = nt_plan(EPI, M, N, K, &want_queue);
  if (plan == M3P_KERN_NT_SKINNY) {
    if constexpr (EPI != M3P_EPI_DGELU && EPI != M3P_EPI_MUL) {
      if (M <= 16) return launch_nt_skinny<EPI, 1>(A, lda, W, ldw, C, ldc, M, N, K, ep, st);
      if (M <= 32) return launch_nt_skinny<EPI, 2>(A, lda, W, ldw, C, ldc, M, N, K, ep, st);
      if (M <= 64) return launch_nt_skinny<EPI, 4>(A, lda, W, ldw, C, ldc, M, N, K, ep, st);
      return launch_nt_skinny<EPI, 8>(A, lda, W, ldw, C, ldc, M, N, K, ep, st);
    }
  }
  if (plan == M3P_KERN_NT_W8) {
    const int tiles_m = M / 256, tiles_n = N / 256;
    const size_t lds = 2 * 512 * ROWB + (EPI == M3P_EPI_DGELU && M3P_DGELU_LUT ? GELU_TAB_N * sizeof(float) : 0) +
                       (M3P_W8_SPARE_EPILOGUE ? ((EPI == M3P_EPI_DGELU || EPI == M3P_EPI_MUL) ? 8 * 2048 : 8 * 4096) : 0);
    auto kern = gemm_nt_w8_kernel<EPI>;
    static bool attr_set8 = false;
    if (!attr_set8) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      attr_set8 = true;
    }
    int grid = num_cus();
    const int ntiles = tiles_m * tiles_n;
    if (ntiles < grid) grid = (ntiles + 7) / 8 * 8;
    // dynamic tile queue (ep.tile_ctr: eight zeroed counters, one per XCD) only where a workgroup takes several tiles
    // (not for the multiply epilogues: with the queue's bookkeeping on top of their aux pieces and column sums the instantiation
    //  spills ~100 bytes per lane - 0.31 against 0.28 ms on the dGELU product, more than the queue returns on 12 launches a step)
    constexpr bool kQueueOk = !(EPI == M3P_EPI_DGELU || EPI == M3P_EPI_MUL);
    int* ctr = nullptr;
    if (kQueueOk && want_queue) {
      hipError_t e;
      ctr = tq_acquire(st, &e);
      if (e != hipSuccess) return (int)e;
    }
    if constexpr (kQueueOk) if (ctr) {
      auto kern_d = gemm_nt_w8_kernel<EPI, true>;
      static bool attr_set8d = false;
      if (!attr_set8d) {
        hipError_t e = hipFuncSetAttribute((const void*)kern_d, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set8d = true;
      }
      hipLaunchKernelGGL(kern_d, dim3(grid), dim3(512), lds, st, A, lda, W, ldw, C, ldc, M, N, K, ep, tiles_m, tiles_n, ctr, g_w8_strip);
      M3P_CHECK_LAUNCH();
      return M3P_OK;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, A, lda, W, ldw, C, ldc, M, N, K, ep, tiles_m, tiles_n, ctr, g_w8_strip);
    M3P_CHECK_LAUNCH();
    return M3P_OK;
  }
  if (plan == M3P_KERN_NT_W4) {
    constexpr int BM = 256, BN = 256;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const size_t lds = 2 * (BM + BN) * 128 + 4 * EP_HALF;
    auto kern = gemm_nt_w4_kernel<EPI, false>;
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      attr_set = true;
    }
    int grid = num_cus();
    const int ntiles = tiles_m * tiles_n;
    if (ntiles < grid) grid = (ntiles + 7) / 8 * 8;
    const long long wbytes = 2LL * N * K, abytes = 2LL * M * K;
    const int m_fast = (wbytes > (64LL << 20) && abytes < wbytes) ? 1 : 0;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, A, lda, W, ldw, C, ldc, M, N, K, ep, tiles_m, tiles_n, m_fast, (unsigned long long*)nullptr);
    M3P_CHECK_LAUNCH();
    return M3P_OK;
  }
  if (plan == M3P_KERN_NT_RING) {
    constexpr int BM = 256, BN = 128;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const size_t lds = 3 * (BM + BN) * ROWB + (EPI == M3P_EPI_DGELU && M3P_DGELU_LUT ? GELU_TAB_N * sizeof(float) : 0);
    auto kern = gemm_nt_ring_kernel<EPI>;
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      attr_set = true;
    }
    int grid = num_cus();                       // one persistent workgroup per CU
    const int ntiles = tiles_m * tiles_n;
    if (ntiles < grid) grid = (ntiles + 7) / 8 * 8;
    const long long wbytes = 2LL * N * K, abytes = 2LL * M * K;
    // (m-fastest order for a huge W used to win; with the 8-wide strips an XCD's 32 tiles are a 4 x 8 block
    //  whose A and W panels fit its L2 together: vocabulary projection 2.20 -> 2.05 ms, 9.3 -> ~2 GB of reads)
    const int m_fast = ((g_ablate & 8) && wbytes > (64LL << 20) && abytes < wbytes) ? 1 : 0;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, A, lda, W, ldw, C, ldc, M, N, K, ep, tiles_m, tiles_n, m_fast);
    M3P_CHECK_LAUNCH();
    return M3P_OK;
  }
  constexpr int BM = 128, BN = 128;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const size_t lds = 2 * (BM + BN) * ROWB;
  auto kern = gemm_nt_kernel<BM, BN, EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), lds, st, A, lda, W, ldw, C, ldc, M, N, K, ep,
                     tiles_m, tiles_n);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

// M3P_EPI_MULQ: the byte-derivative epilogue exists on the eight-wave kernel only (its aux layout IS that kernel's tiling)
// M3P_EPI_BIAS_GELUQ: likewise (it writes that layout); M3P_EPI_BIAS_LSE: its statistics are per 64-column wave block
template <int EPI>
static int launch_nt_gq(const bf16* A, int lda, const bf16* W, int ldw, bf16* C, int ldc, int M, int N, int K,
                        const M3PEpilogue& ep, hipStream_t st) {
  if (M < 1024 || (M % 256) || (N % 256) || N < 512 || (K % 64) || (lda % 8) || (ldw % 8) || (ldc % 8) || ((uintptr_t)C & 15))
    return M3P_EINVAL;
  if (EPI == M3P_EPI_MULQ && (!ep.aux || ((uintptr_t)ep.aux & 15))) return M3P_EINVAL;
  if ((EPI == M3P_EPI_BIAS_GELUQ || EPI == M3P_EPI_BIAS_LSE) && (!ep.out2 || ((uintptr_t)ep.out2 & 15) || !ep.bias || ((uintptr_t)ep.bias & 15)))
    return M3P_EINVAL;
  if (EPI == M3P_EPI_BIAS_LSE && (ep.ld_out2 <= 0 || ep.ld_out2 > N)) return M3P_EINVAL;
  const int tiles_m = M / 256, tiles_n = N / 256;
  const size_t lds = 2 * 512 * ROWB + ((EPI == M3P_EPI_BIAS_GELUQ && M3P_GQ_LUT) ? GQ_TAB_BYTES + 8 * 2048 + 8 * 1024 : 8 * (EPI != M3P_EPI_MULQ ? 4096 : 2048)) +
                     (EPI == M3P_EPI_MULQ ? 2048 + 8 * 1024 : 0);      // (MULQ: behind the staging rows 2 KB - the 8-bit copy's running maxima / the
                                                                       //  prefetch experiment - and the 8-bit copy's own staging rows, 1 KB per wave)
  int grid = num_cus();
  const int ntiles = tiles_m * tiles_n;
  if (ntiles < grid) grid = (ntiles + 7) / 8 * 8;
  if constexpr (EPI == M3P_EPI_BIAS_GELUQ || EPI == M3P_EPI_MULQ) {
    if (ep.out8) {      // the instantiation that also leaves the 8-bit copy (its own registers: the plain one must not pay for it)
      if (((uintptr_t)ep.out8 & 15) || (ep.ld_out8 & 15) || ep.ld_out8 < N || !ep.scale8 || !ep.amax8) return M3P_EINVAL;
      // the copy of an activation is e4m3, of a gradient e5m2 - what the fp8 product's operand slots take (include/m3p_hip.h)
      if ((EPI == M3P_EPI_MULQ) != (ep.out8_bf8 != 0)) return M3P_EINVAL;
      auto kern8 = gemm_nt_w8_kernel<EPI, false, true>;
      static bool attr_set8 = false;
      if (!attr_set8) {
        hipError_t e = hipFuncSetAttribute((const void*)kern8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set8 = true;
      }
      hipLaunchKernelGGL(kern8, dim3(grid), dim3(512), lds, st, A, lda, W, ldw, C, ldc, M, N, K, ep, tiles_m, tiles_n, (int*)nullptr, g_w8_strip);
      M3P_CHECK_LAUNCH();
      return M3P_OK;
    }
  }
  auto kern = gemm_nt_w8_kernel<EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, A, lda, W, ldw, C, ldc, M, N, K, ep, tiles_m, tiles_n, (int*)nullptr, g_w8_strip);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

// ---------------------------------------------------------------------------------
// weight-gradient kernel: dW[i,j] += alpha * sum_m dY[m,i] X[m,j]
// LDS tiles are [64 m][128 cols] bf16 (256-B rows, as in HBM); MFMA operands need 8
// consecutive m per lane for one column -> two ds_read_b64_tr_b16 per fragment.
// ---------------------------------------------------------------------------------
constexpr int WG_T = 128;           // output tile edge
constexpr int WG_ROWB = WG_T * 2;   // 256 B per LDS row
constexpr int WG_TILE_BYTES = BK * WG_ROWB;  // 16 KB

__device__ __forceinline__ bf16x4 lds_tr16(const char* p) {
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  return __builtin_bit_cast(bf16x4, v);
}

__global__ __launch_bounds__(256)
void gemm_wgrad_kernel(const bf16* __restrict__ dY, int lddy, const bf16* __restrict__ X, int ldx,
                       float* __restrict__ dW, int lddw, int M, int N, int K, float alpha,
                       int tiles_i, int tiles_j, int m_chunk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntile = tiles_i * tiles_j;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int split = id / ntile, tile = id - split * ntile;
  const int ti = tile / tiles_j, tj = tile - ti * tiles_j;
  const int i0 = ti * WG_T, j0 = tj * WG_T;
  const int m_begin = split * m_chunk;
  const int m_end = min(M, m_begin + m_chunk);
  if (m_begin >= m_end) return;

  // staging: one wave instruction = 4 rows x 256 B; lane -> (row l>>4, LDS chunk l&15).
  // Bank swizzle for the transpose reads: a ds_read_b64_tr_b16 pass covers 32 lanes = 8 tile
  // rows {8g + j : g in 0..1, j in 0..3} x one 32-B segment each; all rows sit on the same
  // banks (256-B pitch), so the 32-B segment index is XORed with f(row) = (row & 3) |
  // ((row >> 3) & 1) << 2, which is distinct for those 8 rows -> conflict-free.  The LDS-DMA
  // writes lane-linear, so the permutation is applied to the SOURCE chunk (cdna guide rule 21).
  const int sr = lane >> 4, sc = lane & 15;
  const int n_chunks = (N + 7) / 8, k_chunks = (K + 7) / 8;
  int ycol[4], xcol[4];   // per 4-row group handled by this wave (rb = wid + 4*i  ->  (rb >> 1) & 1 == (i*4 + wid) >> 1 & 1)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rb = wid + i * 4;
    const int f = sr | (((rb >> 1) & 1) << 2);
    const int gc = sc ^ (f << 1);
    // clamp the 8-column chunk so a ragged N / K never reads past the row pitch
    ycol[i] = min(i0 / 8 + gc, n_chunks - 1) * 8;
    xcol[i] = min(j0 / 8 + gc, k_chunks - 1) * 8;
  }

  auto stage = [&](int mt, int s) {
    char* sy = smem + s * 2 * WG_TILE_BYTES;
    char* sx = sy + WG_TILE_BYTES;
    const int mbase = m_begin + mt * BK;
    const bool fullt = (mbase + BK <= m_end);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rb = wid + i * 4;            // group of 4 rows
      const int m = mbase + rb * 4 + sr;
      if (fullt) {
        __builtin_amdgcn_global_load_lds(GLB_PTR(dY + (size_t)m * lddy + ycol[i]), LDS_PTR(sy + rb * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLB_PTR(X + (size_t)m * ldx + xcol[i]), LDS_PTR(sx + rb * 1024), 16, 0, 0);
      } else {
        // ragged last tile: rows >= m_end contribute zeros
        uint4 vy = make_uint4(0, 0, 0, 0), vx = make_uint4(0, 0, 0, 0);
        if (m < m_end) {
          vy = *reinterpret_cast<const uint4*>(dY + (size_t)m * lddy + ycol[i]);
          vx = *reinterpret_cast<const uint4*>(X + (size_t)m * ldx + xcol[i]);
        }
        *reinterpret_cast<uint4*>(sy + rb * 1024 + lane * 16) = vy;
        *reinterpret_cast<uint4*>(sx + rb * 1024 + lane * 16) = vx;
      }
    }
  };

  const int wi = wid >> 1, wj = wid & 1;
  const int ft = lane & 15, fg = lane >> 4;
  // tr16 address of lane (t,g) for k-step ks, half jj, sub-tile c:
  //   row = ks*32 + g*8 + jj*4 + (t>>2);  col = wave_base + c*16 + (t&3)*4  (8-byte piece)
  //   chunk = col / 8 swizzled with f(row) << 1, f(row) = (t>>2) | (g&1)<<2 for every ks, jj
  const int frow = (fg * 8 + (ft >> 2)) * WG_ROWB;
  const int fsw = ((ft >> 2) | ((fg & 1) << 2)) << 1;
  int y_off[4], x_off[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int qy = wi * 8 + 2 * c + ((ft & 3) >> 1), qx = wj * 8 + 2 * c + ((ft & 3) >> 1);
    y_off[c] = frow + ((qy ^ fsw) << 4) + ((ft & 1) << 3);
    x_off[c] = WG_TILE_BYTES + frow + ((qx ^ fsw) << 4) + ((ft & 1) << 3);
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nmt = (m_end - m_begin + BK - 1) / BK;
  stage(0, 0);
  __syncthreads();
  for (int mt = 0; mt < nmt; ++mt) {
    const int cur = mt & 1;
    if (mt + 1 < nmt) stage(mt + 1, cur ^ 1);
    const char* sbase = smem + cur * 2 * WG_TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 yf[4], xf[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const char* py = sbase + y_off[c] + ks * 32 * WG_ROWB;
        const char* px = sbase + x_off[c] + ks * 32 * WG_ROWB;
        bf16x4 y0 = lds_tr16(py), y1 = lds_tr16(py + 4 * WG_ROWB);
        bf16x4 x0 = lds_tr16(px), x1 = lds_tr16(px + 4 * WG_ROWB);
        yf[c] = bf16x8{y0[0], y0[1], y0[2], y0[3], y1[0], y1[1], y1[2], y1[3]};
        xf[c] = bf16x8{x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[a], xf[b], acc[a][b], 0, 0, 0);
    }
    __syncthreads();
  }

  // D[i][j]: lane holds j = l&15, i = 4*(l>>4)+r
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int j = j0 + wj * 64 + b * 16 + ft;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wi * 64 + a * 16 + fg * 4 + r;
        if (i < N && j < K) unsafeAtomicAdd(dW + (size_t)i * lddw + j, alpha * acc[a][b][r]);
      }
    }
}

// WGW4-BEGIN (generated by tools/gen/gen_w4.py from tools/gen/wgrad_w4_template.hip - edit those)
// ---------------------------------------------------------------------------------
// Weight-gradient kernel, four-wave version of the stream-K ring kernel below/above:
// dW[i,j] += alpha * sum_m dY[m,i] X[m,j] with 256(i) x 256(j) output tiles, 128x128 per wave
// (accumulators pinned in AGPRs as in gemm_nt_w4_kernel), 64-row K-tiles of M in two 64-KB LDS
// stages ([64 m][256 cols] bf16, 512-B rows for both operands), the same two-phase schedule
// (one s_barrier per 128 MFMAs, LDS-DMAs of K-tile +2 spread over phase 2 and the next phase 1).
// Both operands are contraction-strided, so fragments come out of LDS through
// ds_read_b64_tr_b16 with the 32-B-segment swizzle of the ring kernel.  Stream-K bookkeeping
// (chunk == one workgroup's share, round-robin for many tiles) is the ring kernel's; a segment
// ends with fp32 atomics straight from the AGPRs, the next one starts with C = 0.
// Full tiles only (N % 256 == 0, K % 256 == 0, M % 64 == 0): everything else stays on the ring kernel.
// ---------------------------------------------------------------------------------
// YROWS (the vocabulary DATA gradient dH [n, d] += dlogits [n, V] x E [V, d], round 4): the first operand is given with the
// contraction index CONTIGUOUS - dY_a[i * lddy + m], rows = output rows - i.e. as an NT GEMM's activation panel; its K-tile is
// staged as 256 rows x 128 B (chunk ^= row & 7 on the source, like the NT kernels) and its fragments are plain ds_read_b128,
// which deliver exactly what the two transposing reads deliver for a contraction-strided operand: eight consecutive
// contraction elements of one output row per lane.  Everything else - the second operand's transposing reads, the MFMA
// stream, the (tile, chunk) schedule, the workspace flush and the reduction - is the weight-gradient kernel's.
template <bool YROWS>
__global__ __launch_bounds__(256)
void gemm_wgrad_w4_kernel(const bf16* __restrict__ dY_a, int lddy_a, const bf16* __restrict__ X_a, int ldx_a,
                          float* __restrict__ dW_a, int lddw_a, int M, int N, int K, float alpha,
                          int tiles_i, int tiles_j_a, int WR_CHUNK, float* __restrict__ ws, int* __restrict__ ws_tile,
                          int dbg_flags, int overwrite, WgradProblem pb) {
  // Two products over the same M in one launch (pb.tiles != 0; one-segment-per-workgroup mode only): the tiles of product b
  // follow those of product a in the tile numbering, every workgroup picks its product once, before the K loop.  36 tiles of
  // out_lin + q/k/v then share one launch, one end-of-kernel flush and one reduction instead of 9 tiles x 28 chunks beside
  // 27 x 9.
  const bf16* dY = dY_a;
  const bf16* X = X_a;
  float* dW = dW_a;
  int lddy = lddy_a, ldx = ldx_a, lddw = lddw_a, tiles_j = tiles_j_a;
  int tile_id0 = 0;
  constexpr int TI = 256, TJ = 256, KT = 64;
  constexpr int ROWB = 512;                          // bytes per LDS row of either operand
  constexpr int Y_BYTES = KT * ROWB, STAGE = 2 * Y_BYTES;      // 32 KB, 64 KB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntile_a = tiles_i * tiles_j_a;
  const int ntile = ntile_a + pb.tiles_i * pb.tiles_j;
  const int nmt = M / KT;
  const long long total_all = (long long)ntile * nmt;
  const int nwg = gridDim.x;
  const int per_xcd = nwg >> 3;
  const int slot = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  // Schedule.  Few tiles / long M (every layer weight): each workgroup owns ONE (tile, M-chunk)
  // segment: C = nwg / ntile chunks per tile, slot = chunk * ntile + tile, so the 32 workgroups of
  // an XCD walk the same rows of M on neighbouring tiles and share them in L2, and every
  // workgroup flushes exactly once.  (Dealing the ragged remainder out stream-K style left two
  // workgroups with 18 tile flushes each: 340 us for a 150-us kernel.)  nwg - C * ntile
  // workgroups stay idle (1.6 % for 36 tiles).  Many tiles (WR_CHUNK == 0): whole tiles round-robin.
  const bool rr = (WR_CHUNK == 0);
  // first-segment partials go to this workgroup's private slot of the workspace with plain stores
  // (ws_tile[slot] = tile id, -1 = nothing there); wgrad_reduce_kernel folds the slots into dW.
  if (ws && tid == 0) ws_tile[slot] = -1;
  int total, seg_tile, seg_m0, seg_len;
  if (rr) {
    const int my_tiles = (ntile > slot) ? (ntile - slot + nwg - 1) / nwg : 0;
    if (my_tiles == 0) return;
    total = my_tiles * nmt;
    seg_tile = slot; seg_m0 = 0; seg_len = nmt;
  } else {
    const int C = max(nwg / ntile, 1);
    if (slot >= C * ntile) return;
    const int c = slot / ntile;
    seg_tile = slot - c * ntile;
    if (seg_tile >= ntile_a) {
      dY = pb.dY; X = pb.X; dW = pb.dW; lddy = pb.lddy; ldx = pb.ldx; lddw = pb.lddw; tiles_j = pb.tiles_j;
      seg_tile -= ntile_a;
      tile_id0 = ntile_a;
    }
    seg_m0 = (int)((long long)c * nmt / C);
    seg_len = (int)((long long)(c + 1) * nmt / C) - seg_m0;
    total = seg_len;
    if (total <= 0) return;
  }
  (void)total_all;
  auto locate = [&](int) {
    WgCursor cu;
    cu.c = 0; cu.t = seg_tile; cu.mt = 0; cu.len = seg_len;
    return cu;
  };
  auto advance = [&](WgCursor& cu) {
    if (++cu.mt == cu.len) { cu.mt = 0; cu.t += nwg; }     // (round-robin mode: next tile; otherwise the stream ends here)
  };
  const int g0 = 0;

  // ---- staging.  One wave instruction = 2 rows x 512 B: lane -> (row l >> 5, 16-B slot l & 31).
  // Wave w stages rows [16w, 16w + 16) of each operand = one contiguous 8-KB slice (M0 trick as
  // in the NT kernel).  Segment swizzle of the ring kernel: slot ^= 2 * ((row & 3) | ((row >> 3) & 1) << 2).
  WgCursor lc = locate(g0);
  int l_issued = 0;
  const int l_hi = lane >> 5, l_pos = lane & 31;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  int y_off[8], x_off[8];      // element offsets of this lane's 16 bytes inside the K-tile, immediates compensated
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int row = wid * 16 + 2 * p + l_hi;
    const int f = (row & 3) | (((row >> 3) & 1) << 2);
    const int gc = l_pos ^ (f << 1);
    y_off[p] = row * lddy + gc * 8 - (-4096 + 1024 * p) / 2;
    x_off[p] = row * ldx + gc * 8 - (-4096 + 1024 * p) / 2;
    if (YROWS) {      // piece p of wave w = rows 64 w + 8 p .. + 7 of the 256-row panel, 128 B each: lane -> (row l >> 3, slot l & 7)
      const int yrow = wid * 64 + 8 * p + (lane >> 3);
      y_off[p] = yrow * lddy + (((lane & 7) ^ (yrow & 7)) * 8) - (-4096 + 1024 * p) / 2;
    }
  }
  const bf16* y_base;
  const bf16* x_base;
#ifndef M3P_WG_BUFDMA
#define M3P_WG_BUFDMA 1
#endif
  // M3P_WG_BUFDMA: the transfers as buffer_load_dwordx4 ... lds - resource = the K-tile's base, scalar offset = the piece's
  // rows, a 32-bit lane offset (four per operand: the swizzle of a row depends on the piece's parity and half) - instead of
  // global_load_lds_dwordx4 with a 64-bit lane address per piece.  In the NT kernel the buffer form costs the issuing wave
  // ~16 clocks a piece where the global form costs ~37 (tools/gemm_timeline.py).  The buffer form's immediate is unsigned
  // 12-bit: pieces 0-3 and 4-7 of an operand get an M0 each.
  __amdgpu_buffer_rsrc_t y_rsrc, x_rsrc;
  uint32_t y_voff[4], x_voff[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int f = ((2 * (k & 1) + l_hi) & 3) | ((k >> 1) << 2);        // pieces p with (p & 1, p >= 4) = (k & 1, k >> 1)
    const int gc = l_pos ^ (f << 1);
    y_voff[k] = (uint32_t)(l_hi * lddy + gc * 8) * 2u;
    x_voff[k] = (uint32_t)(l_hi * ldx + gc * 8) * 2u;
    if (YROWS) y_voff[k] = (uint32_t)((lane >> 3) * lddy + (((lane & 7) ^ (lane >> 3)) * 8)) * 2u;
  }
  auto set_rsrc = [&]() {
    if (M3P_WG_BUFDMA) {
      y_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(uniform_ptr(y_base)), 0, 0xffffffff, 0x00020000);
      x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(uniform_ptr(x_base)), 0, 0xffffffff, 0x00020000);
    }
  };
  auto set_load_ktile = [&]() {
    const int ti = lc.t / tiles_j, tj = lc.t - ti * tiles_j;
    const size_t mbase = (size_t)(seg_m0 + lc.mt) * KT;
    y_base = YROWS ? dY + (size_t)(ti * TI) * lddy + mbase : dY + mbase * lddy + ti * TI;
    x_base = X + mbase * ldx + tj * TJ;
    set_rsrc();
  };
  set_load_ktile();
  // (written as instructions in the scalar-base form - SGPR pair + 32-bit lane offset, no 64-bit vector add per piece - the
  //  global loads measured the same, 6.88 against 6.80 ms: the adds fit the free issue slots between two MFMAs.)
#define WG_LD1(PTR, IMM) __builtin_amdgcn_global_load_lds(GLB_PTR(PTR), LDS_PTR(sl), 16, IMM, 0)
#define WG_LDB(IMM) __builtin_amdgcn_raw_ptr_buffer_load_lds(piece < 8 ? y_rsrc : x_rsrc, LDS_PTR(sl), 16, voff, soff, IMM, 0)
  auto issue_load = [&](int s, int piece) {
    if (M3P_WG_BUFDMA) {
      const int pc = piece & 7, k = (pc & 1) + 2 * (pc >> 2);
      char* sl = smem + s * STAGE + (piece < 8 ? 0 : Y_BYTES) + wid * 8192 + (pc >> 2) * 4096;
      const uint32_t voff = piece < 8 ? y_voff[YROWS ? 0 : k] : x_voff[k];
      const uint32_t rows = (piece < 8 && YROWS) ? (uint32_t)(wid * 64 + 8 * pc) : (uint32_t)(wid * 16 + 2 * pc);
      const uint32_t soff = __builtin_amdgcn_readfirstlane(rows * (uint32_t)((piece < 8 ? lddy : ldx) * 2) - (uint32_t)(pc & 3) * 1024u);
      switch (pc & 3) {
        case 0: WG_LDB(0); break;
        case 1: WG_LDB(1024); break;
        case 2: WG_LDB(2048); break;
        default: WG_LDB(3072); break;
      }
      return;
    }
    char* sl = smem + s * STAGE + (piece < 8 ? 0 : Y_BYTES) + wid * 8192 + 4096;
    const bf16* src = (piece < 8) ? y_base + y_off[piece & 7] : x_base + x_off[piece & 7];
    switch (piece & 7) {
      case 0: WG_LD1(src, -4096); break;
      case 1: WG_LD1(src, -3072); break;
      case 2: WG_LD1(src, -2048); break;
      case 3: WG_LD1(src, -1024); break;
      case 4: WG_LD1(src, 0); break;
      case 5: WG_LD1(src, 1024); break;
      case 6: WG_LD1(src, 2048); break;
      default: WG_LD1(src, 3072); break;
    }
  };
  const size_t y_step = YROWS ? (size_t)KT : (size_t)KT * lddy, x_step = (size_t)KT * ldx;
  auto load_done = [&]() {
    // past the end of this workgroup's stream the last K-tile is re-loaded into a stage nobody
    // reads again (keeps the loop body and the vmcnt bookkeeping uniform).
    // Within a tile the bases just move on by 64 rows: recomputing them (a division by tiles_j, two 64-bit products) was ~60
    // scalar instructions per K-tile in front of the mid-step barrier, with the matrix pipe running dry behind them.
    // (one segment per workgroup - every layer weight: branch-free, a select and two 64-bit adds)
    if (!rr) {
      const bool more = ++l_issued < total;
      y_base += more ? y_step : 0;
      x_base += more ? x_step : 0;
      set_rsrc();
    } else if (++l_issued < total) {
      if (++lc.mt == lc.len) { lc.mt = 0; lc.t += nwg; set_load_ktile(); }
      else { y_base += y_step; x_base += x_step; set_rsrc(); }
    }
  };

  // ---- fragment addressing (tr16): lane (t = l & 15, g = l >> 4) reads row 8g + (t >> 2) (+4 for the
  // second half, +32 for k-step 1), 8-byte piece (t & 3) of 16-column sub-tile c
  const int wi = wid >> 1, wj = wid & 1;
  const int ft = lane & 15, fg = lane >> 4;
  const int frow = fg * 8 + (ft >> 2);
  const int fsw = ((ft >> 2) | ((fg & 1) << 2)) << 1;
  // (one address per fragment column AND stage: the stage offset, 64 KB, is beyond the instruction's 16-bit immediate, and a
  //  v_add per read - 32 per K-tile - is not free for a wave alone on its SIMD: every vector instruction between two MFMAs of
  //  the same wave delays the second one.  7.74 -> 7.45 ms over the step's weight gradients, tools/ab_wgrad.py.)
  uint32_t y_addr[2][8], x_addr[2][8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int qy = wi * 16 + 2 * c + ((ft & 3) >> 1), qx = wj * 16 + 2 * c + ((ft & 3) >> 1);
    y_addr[0][c] = lds0 + frow * ROWB + ((qy ^ fsw) << 4) + ((ft & 1) << 3);
    x_addr[0][c] = lds0 + Y_BYTES + frow * ROWB + ((qx ^ fsw) << 4) + ((ft & 1) << 3);
    y_addr[1][c] = y_addr[0][c] + STAGE;
    x_addr[1][c] = x_addr[0][c] + STAGE;
  }
  // YROWS: fragment c of k-step ks = 16 bytes of row 128 wi + 16 c + ft at chunk (fg + 4 ks) ^ (row & 7)
  uint32_t yk_addr[2][2][8];      // [k-step][stage][c]
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      yk_addr[ks][0][c] = lds0 + (wi * 128 + 16 * c + ft) * 128 + (((fg + 4 * ks) ^ (ft & 7)) << 4);
      yk_addr[ks][1][c] = yk_addr[ks][0][c] + STAGE;
    }
  // one fragment = two tr16 reads (rows +0 / +4); OFF selects the k-step (0 / 16384).  The outputs are EARLY-CLOBBER: without
  // the '&' the compiler may give the first read's destination the address register (it did, in 21 of the kernel's 80 pairs),
  // and when the wave stalls between the two reads for longer than the LDS latency the first read's data IS the second
  // read's address - one half-fragment of wrong (finite) data, about once in 300 launches when the operands were freshly
  // allocated (slow first touches), never with warm ones (tools/wgrad_stress2.py; found by tests/test_gemm.py failing once
  // in ~20 runs of the suite)
#define WG_TR2(LO, HI, ADDR, OFF) asm volatile("ds_read_b64_tr_b16 %0, %2 offset:" #OFF "\n\tds_read_b64_tr_b16 %1, %2 offset:" #OFF "+2048" \
                                               : "=&v"(LO), "=&v"(HI) : "v"(ADDR))
// the first operand's fragment c of k-step KS from stage S: two transposing reads, or (YROWS) one ds_read_b128
#define WG_YRD(C, S, KS) do { if constexpr (YROWS) asm volatile("ds_read_b128 %0, %1" : "=v"(yq[C]) : "v"(yk_addr[KS][S][C])); \
                              else if constexpr ((KS) == 0) WG_TR2(yl[C], yh[C], y_addr[S][C], 0);                           \
                              else WG_TR2(yl[C], yh[C], y_addr[S][C], 16384); } while (0)
#define WG_LGKM0() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

  asm volatile("" ::: "a0", "a255");     // reserve all 256 AGPRs (see gemm_nt_w4_kernel)
  // acc tile (b, a) = a[((b)*8+(a))*4 .. +3] holds D[i = 16a + 4(l >> 4) + r][j = 16b + (l & 15)]
#define WG_ACC(B, A) "a[((" #B ")*8+(" #A "))*4:((" #B ")*8+(" #A "))*4+3]"
#define WG_M(YF, XF, B, A) do { if (FIRST) asm volatile("v_mfma_f32_16x16x32_bf16 " WG_ACC(B, A) ", %0, %1, 0" :: "v"(YF[A]), "v"(XF[B])); \
                                else asm volatile("v_mfma_f32_16x16x32_bf16 " WG_ACC(B, A) ", %0, %1, " WG_ACC(B, A) :: "v"(YF[A]), "v"(XF[B])); } while (0)
  // (the ablation switches of tools/gemm_timeline.py are compile-time: as run-time tests they were a scalar compare + branch per
  //  LDS-DMA - 21 per K-tile - in a wave that has no partner on its SIMD to issue around them)
#ifdef M3P_WG_DBG
#define WG_DBG(bit) (dbg_flags & (bit))
#else
#define WG_DBG(bit) 0
#endif
#define WG_L(PIECE) do { if (!WG_DBG(2)) issue_load(s_cur, PIECE); __builtin_amdgcn_sched_barrier(0); } while (0)
#define WG_LP(PIECE) do { if (PEND && !WG_DBG(2)) issue_load(s_cur ^ 1, PIECE); __builtin_amdgcn_sched_barrier(0); } while (0)
  auto frag = [](const s16x4& lo, const s16x4& hi) {
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  s16x4 yl[8], yh[8], xl[8], xh[8];     // raw halves of the fragments being fetched
  bf16x8 yq[8];                         // (YROWS: the first operand's fragments arrive whole)
  auto yfrag = [&](int c) { if constexpr (YROWS) return yq[c]; else return frag(yl[c], yh[c]); };
  bf16x8 yf0[8], xf0[8], yf1[8], xf1[8];
  // the stage is a compile-time fact of each phase body (the K loop is unrolled by two): fragment addresses and LDS-DMA
  // destinations are then plain registers / immediates
  auto phase1 = [&](auto first_c, auto stage_c, auto pend_c) {
    constexpr bool FIRST = decltype(first_c)::value;
    constexpr int s_cur = decltype(stage_c)::value;
    constexpr bool PEND = decltype(pend_c)::value;     // (false in a workgroup's very first step only: nothing to top up yet)
    __builtin_amdgcn_sched_barrier(0);
    WG_M(yf0, xf0, 0, 0);
    WG_M(yf0, xf0, 0, 1); WG_YRD(0, s_cur, 1);
    WG_M(yf0, xf0, 0, 2); WG_LP(11);
    WG_M(yf0, xf0, 0, 3);
    WG_M(yf0, xf0, 0, 4);
    WG_M(yf0, xf0, 0, 5); WG_YRD(1, s_cur, 1);
    WG_M(yf0, xf0, 0, 6); WG_LP(12);
    WG_M(yf0, xf0, 0, 7);
    WG_M(yf0, xf0, 1, 0);
    WG_M(yf0, xf0, 1, 1); WG_YRD(2, s_cur, 1);
    WG_M(yf0, xf0, 1, 2); WG_LP(13);
    WG_M(yf0, xf0, 1, 3);
    WG_M(yf0, xf0, 1, 4);
    WG_M(yf0, xf0, 1, 5); WG_YRD(3, s_cur, 1);
    WG_M(yf0, xf0, 1, 6); WG_LP(14);
    WG_M(yf0, xf0, 1, 7);
    WG_M(yf0, xf0, 2, 0);
    WG_M(yf0, xf0, 2, 1); WG_YRD(4, s_cur, 1);
    WG_M(yf0, xf0, 2, 2); WG_LP(15);
    WG_M(yf0, xf0, 2, 3);
    WG_M(yf0, xf0, 2, 4);
    WG_M(yf0, xf0, 2, 5); WG_YRD(5, s_cur, 1);
    WG_M(yf0, xf0, 2, 6);
    WG_M(yf0, xf0, 2, 7);
    WG_M(yf0, xf0, 3, 0);
    WG_M(yf0, xf0, 3, 1); WG_YRD(6, s_cur, 1);
    WG_M(yf0, xf0, 3, 2);
    WG_M(yf0, xf0, 3, 3);
    WG_M(yf0, xf0, 3, 4);
    WG_M(yf0, xf0, 3, 5); WG_YRD(7, s_cur, 1);
    WG_M(yf0, xf0, 3, 6);
    WG_M(yf0, xf0, 3, 7);
    WG_M(yf0, xf0, 4, 0);
    WG_M(yf0, xf0, 4, 1); WG_TR2(xl[0], xh[0], x_addr[s_cur][0], 16384);
    WG_M(yf0, xf0, 4, 2);
    WG_M(yf0, xf0, 4, 3);
    WG_M(yf0, xf0, 4, 4);
    WG_M(yf0, xf0, 4, 5); WG_TR2(xl[1], xh[1], x_addr[s_cur][1], 16384);
    WG_M(yf0, xf0, 4, 6);
    WG_M(yf0, xf0, 4, 7);
    WG_M(yf0, xf0, 5, 0);
    WG_M(yf0, xf0, 5, 1); WG_TR2(xl[2], xh[2], x_addr[s_cur][2], 16384);
    WG_M(yf0, xf0, 5, 2);
    WG_M(yf0, xf0, 5, 3);
    WG_M(yf0, xf0, 5, 4);
    WG_M(yf0, xf0, 5, 5); WG_TR2(xl[3], xh[3], x_addr[s_cur][3], 16384);
    WG_M(yf0, xf0, 5, 6);
    WG_M(yf0, xf0, 5, 7);
    WG_M(yf0, xf0, 6, 0);
    WG_M(yf0, xf0, 6, 1); WG_TR2(xl[4], xh[4], x_addr[s_cur][4], 16384);
    WG_M(yf0, xf0, 6, 2);
    WG_M(yf0, xf0, 6, 3);
    WG_M(yf0, xf0, 6, 4);
    WG_M(yf0, xf0, 6, 5); WG_TR2(xl[5], xh[5], x_addr[s_cur][5], 16384);
    WG_M(yf0, xf0, 6, 6);
    WG_M(yf0, xf0, 6, 7);
    WG_M(yf0, xf0, 7, 0);
    WG_M(yf0, xf0, 7, 1); WG_TR2(xl[6], xh[6], x_addr[s_cur][6], 16384);
    WG_M(yf0, xf0, 7, 2);
    WG_M(yf0, xf0, 7, 3);
    WG_M(yf0, xf0, 7, 4);
    WG_M(yf0, xf0, 7, 5); WG_TR2(xl[7], xh[7], x_addr[s_cur][7], 16384);
    WG_M(yf0, xf0, 7, 6);
    WG_M(yf0, xf0, 7, 7);
    __builtin_amdgcn_sched_barrier(0);
    if (PEND) load_done();
  };
  auto phase2 = [&](auto stage_c) {
    constexpr bool FIRST = false;
    constexpr int s_cur = decltype(stage_c)::value;
    __builtin_amdgcn_sched_barrier(0);
    WG_M(yf1, xf1, 0, 0); WG_L(0);
    WG_M(yf1, xf1, 0, 1); WG_YRD(0, s_cur ^ 1, 0);
    WG_M(yf1, xf1, 0, 2);
    WG_M(yf1, xf1, 0, 3);
    WG_M(yf1, xf1, 0, 4);
    WG_M(yf1, xf1, 0, 5); WG_YRD(1, s_cur ^ 1, 0);
    WG_M(yf1, xf1, 0, 6); WG_L(1);
    WG_M(yf1, xf1, 0, 7);
    WG_M(yf1, xf1, 1, 0);
    WG_M(yf1, xf1, 1, 1); WG_YRD(2, s_cur ^ 1, 0);
    WG_M(yf1, xf1, 1, 2);
    WG_M(yf1, xf1, 1, 3);
    WG_M(yf1, xf1, 1, 4); WG_L(2);
    WG_M(yf1, xf1, 1, 5); WG_YRD(3, s_cur ^ 1, 0);
    WG_M(yf1, xf1, 1, 6);
    WG_M(yf1, xf1, 1, 7);
    WG_M(yf1, xf1, 2, 0);
    WG_M(yf1, xf1, 2, 1); WG_YRD(4, s_cur ^ 1, 0);
    WG_M(yf1, xf1, 2, 2); WG_L(3);
    WG_M(yf1, xf1, 2, 3);
    WG_M(yf1, xf1, 2, 4);
    WG_M(yf1, xf1, 2, 5); WG_YRD(5, s_cur ^ 1, 0);
    WG_M(yf1, xf1, 2, 6);
    WG_M(yf1, xf1, 2, 7);
    WG_M(yf1, xf1, 3, 0); WG_L(4);
    WG_M(yf1, xf1, 3, 1); WG_YRD(6, s_cur ^ 1, 0);
    WG_M(yf1, xf1, 3, 2);
    WG_M(yf1, xf1, 3, 3);
    WG_M(yf1, xf1, 3, 4);
    WG_M(yf1, xf1, 3, 5); WG_YRD(7, s_cur ^ 1, 0);
    WG_M(yf1, xf1, 3, 6); WG_L(5);
    WG_M(yf1, xf1, 3, 7);
    WG_M(yf1, xf1, 4, 0);
    WG_M(yf1, xf1, 4, 1); WG_TR2(xl[0], xh[0], x_addr[s_cur ^ 1][0], 0);
    WG_M(yf1, xf1, 4, 2);
    WG_M(yf1, xf1, 4, 3);
    WG_M(yf1, xf1, 4, 4); WG_L(6);
    WG_M(yf1, xf1, 4, 5); WG_TR2(xl[1], xh[1], x_addr[s_cur ^ 1][1], 0);
    WG_M(yf1, xf1, 4, 6);
    WG_M(yf1, xf1, 4, 7);
    WG_M(yf1, xf1, 5, 0);
    WG_M(yf1, xf1, 5, 1); WG_TR2(xl[2], xh[2], x_addr[s_cur ^ 1][2], 0);
    WG_M(yf1, xf1, 5, 2); WG_L(7);
    WG_M(yf1, xf1, 5, 3);
    WG_M(yf1, xf1, 5, 4);
    WG_M(yf1, xf1, 5, 5); WG_TR2(xl[3], xh[3], x_addr[s_cur ^ 1][3], 0);
    WG_M(yf1, xf1, 5, 6);
    WG_M(yf1, xf1, 5, 7);
    WG_M(yf1, xf1, 6, 0); WG_L(8);
    WG_M(yf1, xf1, 6, 1); WG_TR2(xl[4], xh[4], x_addr[s_cur ^ 1][4], 0);
    WG_M(yf1, xf1, 6, 2);
    WG_M(yf1, xf1, 6, 3);
    WG_M(yf1, xf1, 6, 4);
    WG_M(yf1, xf1, 6, 5); WG_TR2(xl[5], xh[5], x_addr[s_cur ^ 1][5], 0);
    WG_M(yf1, xf1, 6, 6); WG_L(9);
    WG_M(yf1, xf1, 6, 7);
    WG_M(yf1, xf1, 7, 0);
    WG_M(yf1, xf1, 7, 1); WG_TR2(xl[6], xh[6], x_addr[s_cur ^ 1][6], 0);
    WG_M(yf1, xf1, 7, 2);
    WG_M(yf1, xf1, 7, 3);
    WG_M(yf1, xf1, 7, 4);
    WG_M(yf1, xf1, 7, 5); WG_TR2(xl[7], xh[7], x_addr[s_cur ^ 1][7], 0);
    WG_M(yf1, xf1, 7, 6); WG_L(10);
    WG_M(yf1, xf1, 7, 7);
    __builtin_amdgcn_sched_barrier(0);
  };

#ifndef M3P_WG_SCHED2
#define M3P_WG_SCHED2 1
#endif
#define WG_WAIT_LGKM(N) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define WG_WAIT_VM(N) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define WG_BAR() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define WG_LD() do { load_done(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define WG_SET1() do { _Pragma("unroll") for (int c = 0; c < 8; ++c) { yf1[c] = yfrag(c); xf1[c] = frag(xl[c], xh[c]); } __builtin_amdgcn_sched_barrier(0); } while (0)
#define WG_SET0() do { _Pragma("unroll") for (int c = 0; c < 8; ++c) { yf0[c] = yfrag(c); xf0[c] = frag(xl[c], xh[c]); } __builtin_amdgcn_sched_barrier(0); } while (0)
  // M3P_WG_SCHED2: the K-tile in one piece, as in gemm_nt_w4_kernel - the first operand's region of the stage is released by
  // a barrier as soon as its k-step-1 fragments are in registers (MFMA 20), the second's at MFMA 50, the next K-tile is waited
  // for at MFMA 107 (vmcnt(16): this K-tile's own sixteen transfers stay in flight): a transfer has 1.2-1.7 K-tiles to land
  // where the two-phase form gives the last five of a K-tile 46 MFMAs and waits ~200-270 clocks per K-tile at its vmcnt(0).
  auto wktile = [&](auto first_c, auto stage_c) {
    constexpr bool FIRST0 = decltype(first_c)::value;
    constexpr int s_cur = decltype(stage_c)::value;
    __builtin_amdgcn_sched_barrier(0);
    {
      constexpr bool FIRST = FIRST0;
    WG_M(yf0, xf0, 0, 0);
      WG_M(yf0, xf0, 0, 1); WG_YRD(0, s_cur, 1);
      WG_M(yf0, xf0, 0, 2);
      WG_M(yf0, xf0, 0, 3); WG_YRD(1, s_cur, 1);
      WG_M(yf0, xf0, 0, 4);
      WG_M(yf0, xf0, 0, 5); WG_YRD(2, s_cur, 1);
      WG_M(yf0, xf0, 0, 6);
      WG_M(yf0, xf0, 0, 7); WG_YRD(3, s_cur, 1);
      WG_M(yf0, xf0, 1, 0);
      WG_M(yf0, xf0, 1, 1); WG_YRD(4, s_cur, 1);
      WG_M(yf0, xf0, 1, 2);
      WG_M(yf0, xf0, 1, 3); WG_YRD(5, s_cur, 1);
      WG_M(yf0, xf0, 1, 4);
      WG_M(yf0, xf0, 1, 5); WG_YRD(6, s_cur, 1);
      WG_M(yf0, xf0, 1, 6);
      WG_M(yf0, xf0, 1, 7); WG_YRD(7, s_cur, 1);
      WG_M(yf0, xf0, 2, 0);
      WG_M(yf0, xf0, 2, 1); WG_TR2(xl[0], xh[0], x_addr[s_cur][0], 16384);
      WG_M(yf0, xf0, 2, 2);
      WG_M(yf0, xf0, 2, 3); WG_TR2(xl[1], xh[1], x_addr[s_cur][1], 16384);
      WG_M(yf0, xf0, 2, 4); WG_WAIT_LGKM(4); WG_BAR();
      WG_M(yf0, xf0, 2, 5); WG_L(0);
      WG_M(yf0, xf0, 2, 6);
      WG_M(yf0, xf0, 2, 7); WG_TR2(xl[2], xh[2], x_addr[s_cur][2], 16384);
      WG_M(yf0, xf0, 3, 0);
      WG_M(yf0, xf0, 3, 1);
      WG_M(yf0, xf0, 3, 2); WG_L(1);
      WG_M(yf0, xf0, 3, 3); WG_TR2(xl[3], xh[3], x_addr[s_cur][3], 16384);
      WG_M(yf0, xf0, 3, 4);
      WG_M(yf0, xf0, 3, 5);
      WG_M(yf0, xf0, 3, 6);
      WG_M(yf0, xf0, 3, 7); WG_TR2(xl[4], xh[4], x_addr[s_cur][4], 16384); WG_L(2);
      WG_M(yf0, xf0, 4, 0);
      WG_M(yf0, xf0, 4, 1);
      WG_M(yf0, xf0, 4, 2);
      WG_M(yf0, xf0, 4, 3); WG_TR2(xl[5], xh[5], x_addr[s_cur][5], 16384);
      WG_M(yf0, xf0, 4, 4); WG_L(3);
      WG_M(yf0, xf0, 4, 5);
      WG_M(yf0, xf0, 4, 6);
      WG_M(yf0, xf0, 4, 7); WG_TR2(xl[6], xh[6], x_addr[s_cur][6], 16384);
      WG_M(yf0, xf0, 5, 0);
      WG_M(yf0, xf0, 5, 1); WG_L(4);
      WG_M(yf0, xf0, 5, 2);
      WG_M(yf0, xf0, 5, 3); WG_TR2(xl[7], xh[7], x_addr[s_cur][7], 16384);
      WG_M(yf0, xf0, 5, 4);
      WG_M(yf0, xf0, 5, 5);
      WG_M(yf0, xf0, 5, 6); WG_L(5);
      WG_M(yf0, xf0, 5, 7);
      WG_M(yf0, xf0, 6, 0);
      WG_M(yf0, xf0, 6, 1);
      WG_M(yf0, xf0, 6, 2); WG_WAIT_LGKM(0); WG_SET1(); WG_BAR();
      WG_M(yf0, xf0, 6, 3);
      WG_M(yf0, xf0, 6, 4); WG_L(6);
      WG_M(yf0, xf0, 6, 5);
      WG_M(yf0, xf0, 6, 6);
      WG_M(yf0, xf0, 6, 7);
      WG_M(yf0, xf0, 7, 0);
      WG_M(yf0, xf0, 7, 1); WG_L(7);
      WG_M(yf0, xf0, 7, 2);
      WG_M(yf0, xf0, 7, 3);
      WG_M(yf0, xf0, 7, 4);
      WG_M(yf0, xf0, 7, 5);
      WG_M(yf0, xf0, 7, 6); WG_L(8);
      WG_M(yf0, xf0, 7, 7);
    }
    {
      constexpr bool FIRST = false;
    WG_M(yf1, xf1, 0, 0);
      WG_M(yf1, xf1, 0, 1);
      WG_M(yf1, xf1, 0, 2);
      WG_M(yf1, xf1, 0, 3); WG_L(9);
      WG_M(yf1, xf1, 0, 4);
      WG_M(yf1, xf1, 0, 5);
      WG_M(yf1, xf1, 0, 6);
      WG_M(yf1, xf1, 0, 7);
      WG_M(yf1, xf1, 1, 0); WG_L(10);
      WG_M(yf1, xf1, 1, 1);
      WG_M(yf1, xf1, 1, 2);
      WG_M(yf1, xf1, 1, 3);
      WG_M(yf1, xf1, 1, 4);
      WG_M(yf1, xf1, 1, 5); WG_L(11);
      WG_M(yf1, xf1, 1, 6);
      WG_M(yf1, xf1, 1, 7);
      WG_M(yf1, xf1, 2, 0);
      WG_M(yf1, xf1, 2, 1);
      WG_M(yf1, xf1, 2, 2); WG_L(12);
      WG_M(yf1, xf1, 2, 3);
      WG_M(yf1, xf1, 2, 4);
      WG_M(yf1, xf1, 2, 5);
      WG_M(yf1, xf1, 2, 6);
      WG_M(yf1, xf1, 2, 7); WG_L(13);
      WG_M(yf1, xf1, 3, 0);
      WG_M(yf1, xf1, 3, 1);
      WG_M(yf1, xf1, 3, 2);
      WG_M(yf1, xf1, 3, 3);
      WG_M(yf1, xf1, 3, 4); WG_L(14);
      WG_M(yf1, xf1, 3, 5);
      WG_M(yf1, xf1, 3, 6);
      WG_M(yf1, xf1, 3, 7);
      WG_M(yf1, xf1, 4, 0);
      WG_M(yf1, xf1, 4, 1); WG_L(15);
      WG_M(yf1, xf1, 4, 2); WG_LD();
      WG_M(yf1, xf1, 4, 3);
      WG_M(yf1, xf1, 4, 4);
      WG_M(yf1, xf1, 4, 5);
      WG_M(yf1, xf1, 4, 6);
      WG_M(yf1, xf1, 4, 7);
      WG_M(yf1, xf1, 5, 0);
      WG_M(yf1, xf1, 5, 1);
      WG_M(yf1, xf1, 5, 2);
      WG_M(yf1, xf1, 5, 3); WG_WAIT_VM(16); WG_BAR();
      WG_M(yf1, xf1, 5, 4); WG_YRD(0, s_cur ^ 1, 0);
      WG_M(yf1, xf1, 5, 5); WG_YRD(1, s_cur ^ 1, 0);
      WG_M(yf1, xf1, 5, 6); WG_YRD(2, s_cur ^ 1, 0);
      WG_M(yf1, xf1, 5, 7); WG_YRD(3, s_cur ^ 1, 0);
      WG_M(yf1, xf1, 6, 0); WG_YRD(4, s_cur ^ 1, 0);
      WG_M(yf1, xf1, 6, 1); WG_YRD(5, s_cur ^ 1, 0);
      WG_M(yf1, xf1, 6, 2); WG_YRD(6, s_cur ^ 1, 0);
      WG_M(yf1, xf1, 6, 3); WG_YRD(7, s_cur ^ 1, 0);
      WG_M(yf1, xf1, 6, 4); WG_TR2(xl[0], xh[0], x_addr[s_cur ^ 1][0], 0);
      WG_M(yf1, xf1, 6, 5); WG_TR2(xl[1], xh[1], x_addr[s_cur ^ 1][1], 0);
      WG_M(yf1, xf1, 6, 6); WG_TR2(xl[2], xh[2], x_addr[s_cur ^ 1][2], 0);
      WG_M(yf1, xf1, 6, 7); WG_TR2(xl[3], xh[3], x_addr[s_cur ^ 1][3], 0);
      WG_M(yf1, xf1, 7, 0); WG_TR2(xl[4], xh[4], x_addr[s_cur ^ 1][4], 0);
      WG_M(yf1, xf1, 7, 1); WG_TR2(xl[5], xh[5], x_addr[s_cur ^ 1][5], 0);
      WG_M(yf1, xf1, 7, 2); WG_TR2(xl[6], xh[6], x_addr[s_cur ^ 1][6], 0);
      WG_M(yf1, xf1, 7, 3); WG_TR2(xl[7], xh[7], x_addr[s_cur ^ 1][7], 0);
      WG_M(yf1, xf1, 7, 4);
      WG_M(yf1, xf1, 7, 5);
      WG_M(yf1, xf1, 7, 6);
      WG_M(yf1, xf1, 7, 7); WG_WAIT_LGKM(0); WG_SET0();
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: K-tiles 0 and 1 of the stream into stages 0 and 1
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int pc = 0; pc < 16; ++pc) issue_load(t, pc);
    load_done();
  }
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    WG_YRD(c, 0, 0);
    WG_TR2(xl[c], xh[c], x_addr[0][c], 0);
  }
  WG_LGKM0();
#pragma unroll
  for (int c = 0; c < 8; ++c) { yf0[c] = yfrag(c); xf0[c] = frag(xl[c], xh[c]); }

  WgCursor cc = locate(g0);
  bool first = true;
  int n_seg = 0;
  // -DM3P_WG_TL: s_memtime sums per segment of a K-tile (tools/wgrad_timeline.py): 0 phase 1 (64 MFMAs + reads + LDS-DMAs issued),
  // 1 its closing lgkmcnt(0), 2 vmcnt(0), 3 s_barrier, 4 phase 2, 5 its closing lgkmcnt(0), 6 step tail / flush, 7 K-tiles
#ifdef M3P_WG_TL
  unsigned long long wtl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long wt0 = __builtin_amdgcn_s_memtime(), wt1;
#define WG_TSEG(k) do { __builtin_amdgcn_sched_barrier(0); wt1 = __builtin_amdgcn_s_memtime(); wtl[k] += wt1 - wt0; wt0 = wt1; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WG_TSEG(k) do { } while (0)
#endif
  auto kstep = [&](auto stage_c, int step) {
#if M3P_WG_SCHED2
    if (first) wktile(std::true_type{}, stage_c);
    else wktile(std::false_type{}, stage_c);
    first = false;
#else
    WG_TSEG(6);
    if (step == 0) phase1(std::true_type{}, stage_c, std::false_type{});
    else if (first) phase1(std::true_type{}, stage_c, std::true_type{});
    else phase1(std::false_type{}, stage_c, std::true_type{});
    first = false;
    WG_TSEG(0);
    WG_LGKM0();
    WG_TSEG(1);
#pragma unroll
    for (int c = 0; c < 8; ++c) { yf1[c] = yfrag(c); xf1[c] = frag(xl[c], xh[c]); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WG_TSEG(2);
    __builtin_amdgcn_s_barrier();      // K-tile step+1 visible to all; stage s_cur fully read by all
    asm volatile("" ::: "memory");
    WG_TSEG(3);
    phase2(stage_c);
    WG_TSEG(4);
    WG_LGKM0();
    WG_TSEG(5);
#ifdef M3P_WG_TL
    wtl[7] += 1;
#endif
#pragma unroll
    for (int c = 0; c < 8; ++c) { yf0[c] = yfrag(c); xf0[c] = frag(xl[c], xh[c]); }

#endif

    const bool last_of_tile = (cc.mt + 1 == cc.len) || (step + 1 == total);
    if (last_of_tile) {
      // flush this (tile, chunk) segment with fp32 atomics; the next segment starts from C = 0
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // asm MFMAs: the accumulator-read hazard is ours
      const int ti = cc.t / tiles_j, tj = cc.t - ti * tiles_j;
      float* dbase = dW + (size_t)(ti * TI + wi * 128 + fg * 4) * lddw + tj * TJ + wj * 128 + ft;
      const bool to_ws = (ws != nullptr) && !rr && n_seg == 0 && (step + 1 == total);
      if (to_ws) {       // (only the last segment of a workgroup: no K-tile is in flight towards the stages any more)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      // the workgroups that share this output tile (one per chunk of M) flush at about the same
      // time: each starts at a different 16-column group so they do not queue on the same lines
#pragma nounroll
      for (int bb = 0; bb < 8; ++bb) {
        const int b = to_ws ? bb : ((bb + slot) & 7);
        float v[32];
#define WG_RD8(B)                                                                                   \
  _Pragma("unroll") for (int q = 0; q < 32; ++q) v[q] = 0.f;                                        \
  asm volatile("v_accvgpr_read_b32 %0, a[(" #B ")*32+0]\n\tv_accvgpr_read_b32 %1, a[(" #B ")*32+1]\n\t"   \
               "v_accvgpr_read_b32 %2, a[(" #B ")*32+2]\n\tv_accvgpr_read_b32 %3, a[(" #B ")*32+3]\n\t"   \
               "v_accvgpr_read_b32 %4, a[(" #B ")*32+4]\n\tv_accvgpr_read_b32 %5, a[(" #B ")*32+5]\n\t"   \
               "v_accvgpr_read_b32 %6, a[(" #B ")*32+6]\n\tv_accvgpr_read_b32 %7, a[(" #B ")*32+7]"       \
               : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]));   \
  asm volatile("v_accvgpr_read_b32 %0, a[(" #B ")*32+8]\n\tv_accvgpr_read_b32 %1, a[(" #B ")*32+9]\n\t"   \
               "v_accvgpr_read_b32 %2, a[(" #B ")*32+10]\n\tv_accvgpr_read_b32 %3, a[(" #B ")*32+11]\n\t" \
               "v_accvgpr_read_b32 %4, a[(" #B ")*32+12]\n\tv_accvgpr_read_b32 %5, a[(" #B ")*32+13]\n\t" \
               "v_accvgpr_read_b32 %6, a[(" #B ")*32+14]\n\tv_accvgpr_read_b32 %7, a[(" #B ")*32+15]"     \
               : "=v"(v[8]), "=v"(v[9]), "=v"(v[10]), "=v"(v[11]), "=v"(v[12]), "=v"(v[13]), "=v"(v[14]), "=v"(v[15])); \
  asm volatile("v_accvgpr_read_b32 %0, a[(" #B ")*32+16]\n\tv_accvgpr_read_b32 %1, a[(" #B ")*32+17]\n\t" \
               "v_accvgpr_read_b32 %2, a[(" #B ")*32+18]\n\tv_accvgpr_read_b32 %3, a[(" #B ")*32+19]\n\t" \
               "v_accvgpr_read_b32 %4, a[(" #B ")*32+20]\n\tv_accvgpr_read_b32 %5, a[(" #B ")*32+21]\n\t" \
               "v_accvgpr_read_b32 %6, a[(" #B ")*32+22]\n\tv_accvgpr_read_b32 %7, a[(" #B ")*32+23]"     \
               : "=v"(v[16]), "=v"(v[17]), "=v"(v[18]), "=v"(v[19]), "=v"(v[20]), "=v"(v[21]), "=v"(v[22]), "=v"(v[23])); \
  asm volatile("v_accvgpr_read_b32 %0, a[(" #B ")*32+24]\n\tv_accvgpr_read_b32 %1, a[(" #B ")*32+25]\n\t" \
               "v_accvgpr_read_b32 %2, a[(" #B ")*32+26]\n\tv_accvgpr_read_b32 %3, a[(" #B ")*32+27]\n\t" \
               "v_accvgpr_read_b32 %4, a[(" #B ")*32+28]\n\tv_accvgpr_read_b32 %5, a[(" #B ")*32+29]\n\t" \
               "v_accvgpr_read_b32 %6, a[(" #B ")*32+30]\n\tv_accvgpr_read_b32 %7, a[(" #B ")*32+31]"     \
               : "=v"(v[24]), "=v"(v[25]), "=v"(v[26]), "=v"(v[27]), "=v"(v[28]), "=v"(v[29]), "=v"(v[30]), "=v"(v[31]))
        switch (b) {
          case 0: { WG_RD8(0); } break;
          case 1: { WG_RD8(1); } break;
          case 2: { WG_RD8(2); } break;
          case 3: { WG_RD8(3); } break;
          case 4: { WG_RD8(4); } break;
          case 5: { WG_RD8(5); } break;
          case 6: { WG_RD8(6); } break;
          default: { WG_RD8(7); } break;
        }
#undef WG_RD8
        if (to_ws) {
          // through this wave's LDS patch (the K loop is over: the stages are free) so that the
          // workspace is written in full 128-B lines: lanes hold columns, lines run along rows.
          // 4-byte stores in 64-B pieces left the kernel waiting ~150 us after its last wave.
          float* lw = reinterpret_cast<float*>(smem) + wid * (128 * 36) + (fg * 4) * 36 + (b & 1) * 16 + ft;
#pragma unroll
          for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) lw[(a * 16 + r) * 36] = v[a * 4 + r];
          if (b & 1) {
            // patch = 128 rows x 32 columns (pitch 36 words); one instruction moves 8 rows x 128 B
            const float* lr = reinterpret_cast<const float*>(smem) + wid * (128 * 36) + (lane >> 3) * 36 + (lane & 7) * 4;
            float* wrow = ws + (size_t)slot * (TI * TJ + 272) + (size_t)(wi * 128 + (lane >> 3)) * TJ + wj * 128 + (b >> 1) * 32 + (lane & 7) * 4;
#pragma unroll
            for (int it = 0; it < 16; ++it)
              *reinterpret_cast<f32x4*>(wrow + it * 8 * TJ) = *reinterpret_cast<const f32x4*>(lr + it * 8 * 36);
          }
        } else {
          float* dcol = dbase + b * 16;
#pragma unroll
          for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              // overwrite (whole-tile round-robin mode on a gradient the caller knows to be zero: the vocabulary matrix's
              // first product of a step): a plain store - the atomic is a read-modify-write of 768 MB that is not in any cache
              if (overwrite) dcol[(size_t)(a * 16 + r) * lddw] = alpha * v[a * 4 + r];
              else if (!WG_DBG(1)) unsafeAtomicAdd(dcol + (size_t)(a * 16 + r) * lddw, alpha * v[a * 4 + r]);
        }
      }
      if (to_ws && tid == 0) ws_tile[slot] = tile_id0 + cc.t;
      ++n_seg;
      first = true;
    }
    advance(cc);
  };
  for (int step = 0; step < total; step += 2) {
    kstep(std::integral_constant<int, 0>{}, step);
    if (step + 1 < total) kstep(std::integral_constant<int, 1>{}, step + 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // junk loads of the tail must not outlive the LDS allocation
#ifdef M3P_WG_TL
  WG_TSEG(6);
  if (lane == 0)
    for (int k = 0; k < 8; ++k) g_ring_tl[(blockIdx.x * 4 + wid) * 8 + k] = wtl[k];
#endif
#undef WG_TSEG
#undef WG_LD1
#undef WG_LDB
#undef WG_TR2
#undef WG_YRD
#undef WG_LGKM0
#undef WG_ACC
#undef WG_M
#undef WG_L
#undef WG_DBG
#undef WG_LP
}

// dW[tile] += alpha * sum of the workspace slots that hold a partial of that tile.
// grid = (16 row groups, ntile); block = 256 threads, each 4 rows x 4 consecutive columns.
__global__ __launch_bounds__(256)
void wgrad_reduce_kernel(const float* __restrict__ ws, const int* __restrict__ ws_tile, int nslots,
                         float* __restrict__ dW, int lddw, int tiles_j, float alpha, int ntile_a, WgradProblem pb) {
  const int t = blockIdx.y, rg = blockIdx.x;
  int tl = t;
  if (t >= ntile_a) { tl = t - ntile_a; dW = pb.dW; lddw = pb.lddw; tiles_j = pb.tiles_j; }      // (second product of a paired launch)
  const int ti = tl / tiles_j, tj = tl - ti * tiles_j;
  const int col = (threadIdx.x & 63) * 4, row0 = rg * 16 + (threadIdx.x >> 6) * 4;
  f32x4 acc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
  bool any = false;
  // the producer of chunk c of tile t is workgroup slot c * ntile + t (see gemm_wgrad_w4_kernel)
  const int ntile = gridDim.y;
  for (int s = t; s < nslots; s += ntile) {
    if (ws_tile[s] != t) continue;
    any = true;
    const float* p = ws + (size_t)s * (65536 + 272) + row0 * 256 + col;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += *reinterpret_cast<const f32x4*>(p + r * 256);
  }
  if (!any) return;
  float* d = dW + (size_t)(ti * 256 + row0) * lddw + tj * 256 + col;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    f32x4 v = *reinterpret_cast<f32x4*>(d + (size_t)r * lddw);
    v += alpha * acc[r];
    *reinterpret_cast<f32x4*>(d + (size_t)r * lddw) = v;
  }
}

// WGW4-END
// ---------------------------------------------------------------------------------
// Weight-gradient kernel, persistent stream-K ring version (production path when M % 64 == 0):
// dW[i,j] += alpha * sum_m dY[m,i] X[m,j] with 256(i) x 128(j) output tiles.  The whole
// contraction space (all output tiles x all 64-row K-tiles of M) is ONE stream, ordered
// (m-chunk, tile, K-tile) with the chunk length equal to one workgroup's share of the stream:
// workgroup w then owns exactly (chunk w / ntile, tile w % ntile), so the 32 workgroups of an
// XCD walk the SAME rows of M at the same time on 32 neighbouring tiles and share their
// dY / X row blocks in that XCD's L2 (measured: L2 hit rate 5 % -> see DESIGN.md with a
// 32-K-tile chunk, every operand byte came from the fabric).  The ragged last chunk is
// dealt out stream-K style, so every workgroup still gets an equal share (no split-factor
// quantisation) and at most a few fp32-atomic flushes.  Each workgroup runs its share through
// the same three-stage LDS ring / counted-vmcnt / double-buffered-fragment pipeline as the NT
// kernel; fragments come out of LDS with ds_read_b64_tr_b16 (inline asm, conflict-free
// through the source-side segment swizzle).
// ---------------------------------------------------------------------------------
constexpr int WR_I = 256, WR_J = 128;
constexpr int WR_YROW = WR_I * 2, WR_XROW = WR_J * 2;             // 512 / 256 B per LDS row
constexpr int WR_YB = BK * WR_YROW, WR_XB = BK * WR_XROW;         // 32 KB + 16 KB
constexpr int WR_STAGE = WR_YB + WR_XB;

__global__ __launch_bounds__(512)
void gemm_wgrad_ring_kernel(const bf16* __restrict__ dY, int lddy, const bf16* __restrict__ X, int ldx,
                            float* __restrict__ dW, int lddw, int M, int N, int K, float alpha,
                            int tiles_i, int tiles_j, int WR_CHUNK) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntile = tiles_i * tiles_j;
  const int nmt = M / BK;
  const long long total_all = (long long)ntile * nmt;
  const int nwg = gridDim.x;
  const int per_xcd = nwg >> 3;
  const int slot = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  // Two schedules.  Few tiles / long M (every layer weight): stream-K with chunk == share, see
  // above.  Many tiles / short M (the tied vocabulary matrix: 5862 tiles, 76 K-tiles): tiles
  // are dealt round-robin (WR_CHUNK == 0), each done over all of M by one workgroup, so that
  // an XCD's 32 workgroups hold 32 consecutive tiles = the six j-tiles of ~5 dY panels.
  const bool rr = (WR_CHUNK == 0);
  int g0, total;
  if (rr) {
    const int my_tiles = (ntile > slot) ? (ntile - slot + nwg - 1) / nwg : 0;
    if (my_tiles == 0) return;
    g0 = 0;
    total = my_tiles * nmt;
  } else {
    // shares are whole chunk strips: workgroup `slot` owns stream positions [slot*share, +share)
    const long long share = (total_all + nwg - 1) / nwg;
    const long long g0l = share * slot;
    if (g0l >= total_all) return;
    g0 = (int)g0l;
    total = (int)((g0l + share <= total_all ? g0l + share : total_all) - g0l);
  }

  auto locate = [&](int g) {
    WgCursor cu;
    if (rr) { cu.c = 0; cu.t = slot; cu.mt = 0; cu.len = nmt; return cu; }
    const int full = ntile * WR_CHUNK;
    cu.c = g / full;
    const int rem = g - cu.c * full;
    cu.len = min(WR_CHUNK, nmt - cu.c * WR_CHUNK);
    cu.t = rem / cu.len;
    cu.mt = rem - cu.t * cu.len;
    return cu;
  };
  auto advance = [&](WgCursor& cu) {
    if (++cu.mt == cu.len) {
      cu.mt = 0;
      if (rr) { cu.t += nwg; return; }
      if (++cu.t == ntile) { cu.t = 0; ++cu.c; cu.len = min(WR_CHUNK, nmt - cu.c * WR_CHUNK); }
    }
  };

  // ---- staging.  dY: one wave instruction = 2 rows x 512 B, lane -> (row l>>5, pos l&31);
  //      X: 4 rows x 256 B, lane -> (row l>>4, pos l&15).  Segment swizzle as in the 128^2 kernel.
  const int n_chunks = (N + 7) / 8, k_chunks = (K + 7) / 8;
  WgCursor lc = locate(g0);
  auto stage_next = [&](int s) {
    char* sy = smem + s * WR_STAGE;
    char* sx = sy + WR_YB;
    const int ti = lc.t / tiles_j, tj = lc.t - ti * tiles_j;
    const int mbase = (lc.c * WR_CHUNK + lc.mt) * BK;   // (c == 0 in round-robin mode)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rb = wid + i * 8;               // 2-row group 0..31
      const int row = rb * 2 + (lane >> 5);
      const int f = (row & 3) | (((row >> 3) & 1) << 2);
      const int gc = (lane & 31) ^ (f << 1);
      const int col = min(ti * (WR_I / 8) + gc, n_chunks - 1) * 8;
      __builtin_amdgcn_global_load_lds(GLB_PTR(dY + (size_t)(mbase + row) * lddy + col), LDS_PTR(sy + rb * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rb = wid + i * 8;               // 4-row group 0..15
      const int row = rb * 4 + (lane >> 4);
      const int f = (row & 3) | (((row >> 3) & 1) << 2);
      const int gc = (lane & 15) ^ (f << 1);
      const int col = min(tj * (WR_J / 8) + gc, k_chunks - 1) * 8;
      __builtin_amdgcn_global_load_lds(GLB_PTR(X + (size_t)(mbase + row) * ldx + col), LDS_PTR(sx + rb * 1024), 16, 0, 0);
    }
    advance(lc);
  };

  // ---- fragment addressing (tr16): lane (t = l&15, g = l>>4) reads row g*8 + (t>>2) (+32 ks, +4 jj),
  //      8-byte piece (t&3) of 16-column sub-tile c
  const int wi = wid >> 1, wj = wid & 1;
  const int ft = lane & 15, fg = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int frow = fg * 8 + (ft >> 2);
  const int fsw = ((ft >> 2) | ((fg & 1) << 2)) << 1;
  uint32_t y_addr[4], x_addr[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int qy = wi * 8 + 2 * c + ((ft & 3) >> 1), qx = wj * 8 + 2 * c + ((ft & 3) >> 1);
    y_addr[c] = lds0 + frow * WR_YROW + ((qy ^ fsw) << 4) + ((ft & 1) << 3);
    x_addr[c] = lds0 + WR_YB + frow * WR_XROW + ((qx ^ fsw) << 4) + ((ft & 1) << 3);
  }
#define M3P_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
  // k-step 0 / 1 of a stage: row offsets 0 / 32 rows; second half of a fragment: +4 rows
  auto read_set0 = [&](uint32_t so, s16x4 (&y)[4][2], s16x4 (&x)[4][2]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      M3P_TR(y[c][0], y_addr[c] + so, 0); M3P_TR(y[c][1], y_addr[c] + so, 2048);
      M3P_TR(x[c][0], x_addr[c] + so, 0); M3P_TR(x[c][1], x_addr[c] + so, 1024);
    }
  };
  auto read_set1 = [&](uint32_t so, s16x4 (&y)[4][2], s16x4 (&x)[4][2]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      M3P_TR(y[c][0], y_addr[c] + so, 16384); M3P_TR(y[c][1], y_addr[c] + so, 18432);
      M3P_TR(x[c][0], x_addr[c] + so, 8192); M3P_TR(x[c][1], x_addr[c] + so, 9216);
    }
  };
#define M3P_LGKM0() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto frag = [](const s16x4 (&h)[2]) {
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7));
  };
  auto mfma_batch = [&](const s16x4 (&y)[4][2], const s16x4 (&x)[4][2]) {
    bf16x8 yf[4], xf[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { yf[c] = frag(y[c]); xf[c] = frag(x[c]); }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[a], xf[b], acc[a][b], 0, 0, 0);
  };

  stage_next(0);
  if (total > 1) {
    stage_next(1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  s16x4 y0[4][2], x0[4][2], y1[4][2], x1[4][2];
  read_set0(0, y0, x0);
  M3P_LGKM0();
  WgCursor cc = locate(g0);
  int cur = 0;
  for (int step = 0; step < total; ++step) {
    const int nxt = (cur == 2) ? 0 : cur + 1;
    const int nx2 = (nxt == 2) ? 0 : nxt + 1;
    const bool more2 = (step + 2 < total);
    if (more2) stage_next(nx2);
    read_set1(cur * WR_STAGE, y1, x1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_batch(y0, x0);
    M3P_LGKM0();
    if (more2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    read_set0(nxt * WR_STAGE, y0, x0);   // stale after the last K-tile: unused
    __builtin_amdgcn_sched_barrier(0);
    mfma_batch(y1, x1);
    M3P_LGKM0();

    const bool last_of_tile = (cc.mt + 1 == cc.len) || (step + 1 == total);
    if (last_of_tile) {
      // flush this (tile, chunk) segment: D[i][j], lane holds j = l&15, i = 4*(l>>4)+r
      const int ti = cc.t / tiles_j, tj = cc.t - ti * tiles_j;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int j = tj * WR_J + wj * 64 + b * 16 + ft;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = ti * WR_I + wi * 64 + a * 16 + fg * 4 + r;
            if (i < N && j < K) unsafeAtomicAdd(dW + (size_t)i * lddw + j, alpha * acc[a][b][r]);
            acc[a][b][r] = 0.f;
          }
        }
    }
    advance(cc);
    cur = nxt;
  }
#undef M3P_TR
#undef M3P_LGKM0
}

}  // namespace

namespace {
template <int EPI, bool A_BF8>
int launch_nt_fp8(const uint8_t* A, int lda, const uint8_t* W, int ldw, bf16* C, int ldc, int M, int N, int K,
                         const M3PEpilogue& ep, hipStream_t st) {
  const int tiles_m = M / 256, tiles_n = N / 256;
  const size_t lds = 2 * 512 * ROWB + (EPI == M3P_EPI_DGELU && M3P_DGELU_LUT ? GELU_TAB_N * sizeof(float) : 0) +
                     (M3P_W8_SPARE_EPILOGUE ? ((EPI == M3P_EPI_DGELU || EPI == M3P_EPI_MUL) ? 8 * 2048 : 8 * 4096) : 0);
  auto kern = gemm_nt_w8f8_kernel<EPI, A_BF8>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  int grid = num_cus();
  const int ntiles = tiles_m * tiles_n;
  if (ntiles < grid) grid = (ntiles + 7) / 8 * 8;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, A, lda, W, ldw, C, ldc, M, N, K, ep, tiles_m, tiles_n);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

}  // namespace

extern "C" {

int m3p_gemm_nt_bf16(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                     int epilogue, const M3PEpilogue* ep_in, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (K % BK) != 0 || (lda % 8) != 0 || (ldw % 8) != 0 || (ldc % 4) != 0)
    return M3P_EINVAL;   // (K % 64 == 0 also covers the 32-deep stages of the 256x256 kernel)
  if (((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)C & 7)) return M3P_EINVAL;
  M3PEpilogue ep = {};
  if (ep_in) ep = *ep_in;
  if ((epilogue == M3P_EPI_BIAS_DROP_RES || epilogue == M3P_EPI_RES || epilogue == M3P_EPI_DGELU || epilogue == M3P_EPI_MUL) &&
      (!ep.aux || (ep.ld_aux % 4) != 0))
    return M3P_EINVAL;
  if (epilogue == M3P_EPI_BIAS_GELU && (!ep.out2 || (ep.ld_out2 % 4) != 0)) return M3P_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const bf16* a = (const bf16*)A; const bf16* w = (const bf16*)W; bf16* c = (bf16*)C;
  switch (epilogue) {
    case M3P_EPI_NONE: return launch_nt<M3P_EPI_NONE>(a, lda, w, ldw, c, ldc, M, N, K, ep, st);
    case M3P_EPI_BIAS: return launch_nt<M3P_EPI_BIAS>(a, lda, w, ldw, c, ldc, M, N, K, ep, st);
    case M3P_EPI_BIAS_GELU: return launch_nt<M3P_EPI_BIAS_GELU>(a, lda, w, ldw, c, ldc, M, N, K, ep, st);
    case M3P_EPI_BIAS_DROP_RES: return launch_nt<M3P_EPI_BIAS_DROP_RES>(a, lda, w, ldw, c, ldc, M, N, K, ep, st);
    case M3P_EPI_RES: return launch_nt<M3P_EPI_RES>(a, lda, w, ldw, c, ldc, M, N, K, ep, st);
    case M3P_EPI_DGELU: return launch_nt<M3P_EPI_DGELU>(a, lda, w, ldw, c, ldc, M, N, K, ep, st);
    case M3P_EPI_MUL: return launch_nt<M3P_EPI_MUL>(a, lda, w, ldw, c, ldc, M, N, K, ep, st);
    case M3P_EPI_MULQ: return launch_nt_gq<M3P_EPI_MULQ>(a, lda, w, ldw, c, ldc, M, N, K, ep, st);
    case M3P_EPI_BIAS_GELUQ: return launch_nt_gq<M3P_EPI_BIAS_GELUQ>(a, lda, w, ldw, c, ldc, M, N, K, ep, st);
    case M3P_EPI_BIAS_LSE: return launch_nt_gq<M3P_EPI_BIAS_LSE>(a, lda, w, ldw, c, ldc, M, N, K, ep, st);
    default: return M3P_EINVAL;
  }
}

int m3p_gemm_nt_fp8(const void* A, int lda, int a_is_bf8, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                    int epilogue, const M3PEpilogue* ep_in, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (K % 128) != 0 || (lda % 16) != 0 || (ldw % 16) != 0 || (ldc % 8) != 0) return M3P_EINVAL;
  if ((M % 256) != 0 || (N % 256) != 0) return M3P_EINVAL;       // full 256x256 tiles only (the encoder layers' shapes)
  if (((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)C & 15)) return M3P_EINVAL;
  M3PEpilogue ep = {};
  if (ep_in) ep = *ep_in;
  if ((epilogue == M3P_EPI_BIAS_DROP_RES || epilogue == M3P_EPI_RES || epilogue == M3P_EPI_DGELU || epilogue == M3P_EPI_MUL) &&
      (!ep.aux || (ep.ld_aux % 8) != 0))
    return M3P_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const uint8_t* a = (const uint8_t*)A; const uint8_t* w = (const uint8_t*)W; bf16* c = (bf16*)C;
#define M3P_F8_CASE(E)                                                                                        \
  case E: return a_is_bf8 ? launch_nt_fp8<E, true>(a, lda, w, ldw, c, ldc, M, N, K, ep, st)                    \
                          : launch_nt_fp8<E, false>(a, lda, w, ldw, c, ldc, M, N, K, ep, st)
  switch (epilogue) {
    M3P_F8_CASE(M3P_EPI_NONE);
    M3P_F8_CASE(M3P_EPI_BIAS);
    M3P_F8_CASE(M3P_EPI_BIAS_DROP_RES);
    M3P_F8_CASE(M3P_EPI_RES);
    M3P_F8_CASE(M3P_EPI_DGELU);
    M3P_F8_CASE(M3P_EPI_MUL);
    default: return M3P_ENOTIMPL;
  }
#undef M3P_F8_CASE
}

// debug (only with -DM3P_RING_TL): per-wave cycle sums of the last dGELU launch of the eight-wave kernel
// [256 workgroups][8 waves][8 segments]: 0 K loop, 1 bias / row copies, 2 aux fetch + wait, 3 epilogue half (compute,
// staging, stores), 4 column sums, 5 zeroing + end barrier  (tools/ring_timeline.py)
__attribute__((visibility("default"))) int m3p_debug_ring_timeline(void* out, size_t bytes) {
#if defined(M3P_RING_TL) || defined(M3P_W8_TL) || defined(M3P_WG_TL)
  if (bytes > sizeof(g_ring_tl)) return M3P_EINVAL;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ring_tl), bytes);
#else
  (void)out; (void)bytes;
  return M3P_EINVAL;
#endif
}

__attribute__((visibility("default"))) int m3p_debug_gemm_timeline(const void* A, int lda, const void* W, int ldw, void* C, int ldc,
                                                                   int M, int N, int K, unsigned long long* dbg, void* stream) {
  // (only the four-wave kernel carries timeline instrumentation; full 256x256 tiles, K % 64 == 0)
  if ((M % 256) || (N % 256) || (K % 64)) return M3P_EINVAL;
  M3PEpilogue ep = {};
  const int tm = M / 256, tn = N / 256;
  const size_t lds4 = 2 * 512 * 128 + 4 * EP_HALF;
  auto k4 = gemm_nt_w4_kernel<M3P_EPI_NONE, true, 0>;
  if (g_ablate == 1) k4 = gemm_nt_w4_kernel<M3P_EPI_NONE, true, 1>;
  if (g_ablate == 2) k4 = gemm_nt_w4_kernel<M3P_EPI_NONE, true, 2>;
  if (g_ablate == 3) k4 = gemm_nt_w4_kernel<M3P_EPI_NONE, true, 3>;
  if (g_ablate == 4) k4 = gemm_nt_w4_kernel<M3P_EPI_NONE, true, 4>;
  hipFuncSetAttribute((const void*)k4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
  hipLaunchKernelGGL(k4, dim3(num_cus()), dim3(256), lds4, (hipStream_t)stream, (const bf16*)A, lda, (const bf16*)W, ldw,
                     (bf16*)C, ldc, M, N, K, ep, tm, tn, 0, dbg);
  return (int)hipGetLastError();
}



static int launch_streamk(const void* A, int lda, const void* W, int ldw, float* C, int ldc, int M, int N, int K, float alpha,
                          bool w_kn, int k_valid, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (K % BK) != 0 || (lda % 8) != 0 || (ldw % 8) != 0) return M3P_EINVAL;
  if (((uintptr_t)A & 15) || ((uintptr_t)W & 15)) return M3P_EINVAL;
  if (w_kn && (k_valid <= 0 || k_valid > K)) return M3P_EINVAL;
  const int tiles_m_all = (M + 255) / 256, tiles_n = (N + 127) / 128;
  const size_t lds = 3 * (256 + 128) * ROWB;
  auto kern = w_kn ? gemm_nt_streamk_kernel<true> : gemm_nt_streamk_kernel<false>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  const int grid = num_cus();
  const int nk = K / BK;
  // More tiles than workgroups: one launch would hand every workgroup 1.x whole tiles, so the workgroups
  // sit at 256 different K positions and the second operand (384 MB for the vocabulary) streams from
  // HBM once per tile row (measured: 551 TF/s at M = 19456 against 896 at M = 4864).  Row blocks of at
  // most one tile per workgroup keep every launch in the regime the kernel was laid out for: all
  // workgroups walk the same K range together and W is shared through L2 / MALL.
  const int n_launch = (tiles_m_all * tiles_n + grid - 1) / grid;
  const int rows_per = ((tiles_m_all + n_launch - 1) / n_launch) * 256;
  for (int m0 = 0; m0 < M; m0 += rows_per) {
    const int m = (M - m0 < rows_per) ? M - m0 : rows_per;
    const int tiles_m = (m + 255) / 256;
    long long share = ((long long)tiles_m * tiles_n * nk + grid - 1) / grid;
    const int chunk = (int)(share < nk ? (share < 1 ? 1 : share) : nk);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, (hipStream_t)stream, (const bf16*)A + (size_t)m0 * lda, lda,
                       (const bf16*)W, ldw, C + (size_t)m0 * ldc, ldc, m, N, K, alpha, tiles_m, tiles_n, chunk, k_valid);
    M3P_CHECK_LAUNCH();
  }
  return M3P_OK;
}

int m3p_gemm_nt_streamk_f32(const void* A, int lda, const void* W, int ldw, float* C, int ldc, int M, int N, int K,
                            float alpha, void* stream) {
  return launch_streamk(A, lda, W, ldw, C, ldc, M, N, K, alpha, false, K, stream);
}

int m3p_gemm_nn_streamk_f32(const void* A, int lda, const void* W, int ldw, int k_valid, float* C, int ldc, int M, int N,
                            int K, float alpha, void* stream) {
  return launch_streamk(A, lda, W, ldw, C, ldc, M, N, K, alpha, true, k_valid, stream);
}

int m3p_set_persistent_grid(int workgroups) {
  if (workgroups < 0 || (workgroups % 8) != 0) return M3P_EINVAL;
  g_persistent_grid = workgroups;
  return M3P_OK;
}

int m3p_set_tile_queue(int32_t* counters, int n_slots) {
  if (counters && (n_slots <= 0 || ((uintptr_t)counters & 3))) return M3P_EINVAL;
  std::lock_guard<std::mutex> lk(g_tq_mu);
  g_tq_pool = counters;
  g_tq_slots = counters ? n_slots : 0;
  g_tq_next = 0;
  g_tq_bound = false;        // the next queued launch binds the ring to its stream
  return M3P_OK;
}

size_t m3p_gemm_wgrad_workspace_bytes(void) {
  const size_t grid = (size_t)hw_cus();
  return grid * (65536 + 272) * sizeof(float) + grid * sizeof(int);
}

// four-wave kernel + reduction; pb: optional second product over the same M (paired launch), nullptr = none
static bool wgrad_w4_ok(int M, int N, int K, int lddy, int ldx, const void* dY, const void* X) {
  return g_variant >= 1 && g_variant != 3 && (M % 64) == 0 && M >= 4096 && (N % 256) == 0 && (K % 256) == 0 &&
         (lddy % 8) == 0 && (ldx % 8) == 0 && !((uintptr_t)dY & 15) && !((uintptr_t)X & 15);
}
extern "C++" {
template <bool YROWS = false>
static int launch_wgrad_w4(const void* dY, int lddy, const void* X, int ldx, float* dW, int lddw, int M, int N, int K,
                           float alpha, void* workspace, size_t workspace_bytes, void* stream, const WgradProblem* pb_in,
                           bool store = false) {
  const int ti = N / 256, tj = K / 256;
  WgradProblem pb = {};
  if (pb_in) pb = *pb_in;
  const int ntile = ti * tj + pb.tiles_i * pb.tiles_j;
  const size_t lds = 2 * 65536;
  static bool attr_set_w = false;
  if (!attr_set_w) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_wgrad_w4_kernel<YROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set_w = true;
  }
  const int grid = num_cus();
  const int nmt = M / 64;
  long long share = ((long long)ntile * nmt + grid - 1) / grid;
  int chunk = (int)(share < nmt ? (share < 1 ? 1 : share) : nmt);
  if ((long long)ntile >= 4LL * grid) chunk = 0;   // many tiles: round-robin whole tiles
  if (pb_in && chunk == 0) return M3P_EINVAL;       // (pairs only in the one-segment-per-workgroup mode; the caller checks)
  if (store && chunk != 0) return M3P_ENOTIMPL;     // (a plain store needs every tile flushed exactly once: whole-tile round-robin mode)
  // workspace for first-segment partials: one 256-KB slot per workgroup + a tile-id word each, owned by the
  // CALLER (m3p_gemm_wgrad_workspace_bytes): launches that may overlap on different streams need one each.
  // Without it the partial tiles go to dW with fp32 atomics (slower: DESIGN.md section 4).
  float* ws = nullptr;
  if (workspace && workspace_bytes >= m3p_gemm_wgrad_workspace_bytes() && (((uintptr_t)workspace & 15) == 0) && chunk != 0 &&
      (lddw % 4) == 0 && (((uintptr_t)dW & 15) == 0) && (!pb_in || ((pb.lddw % 4) == 0 && (((uintptr_t)pb.dW & 15) == 0))))
    ws = (float*)workspace;
  int* ws_tile = ws ? (int*)(ws + (size_t)grid * (65536 + 272)) : nullptr;
  hipLaunchKernelGGL(gemm_wgrad_w4_kernel<YROWS>, dim3(grid), dim3(256), lds, (hipStream_t)stream, (const bf16*)dY, lddy,
                     (const bf16*)X, ldx, dW, lddw, M, N, K, alpha, ti, tj, chunk, ws, ws_tile, g_ablate, store ? 1 : 0, pb);
  M3P_CHECK_LAUNCH();
  if (ws) {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(16, ntile), dim3(256), 0, (hipStream_t)stream, (const float*)ws,
                       (const int*)ws_tile, grid, dW, lddw, tj, alpha, ti * tj, pb);
    M3P_CHECK_LAUNCH();
  }
  return M3P_OK;
}
}  // extern "C++"

// ---------------------------------------------------------------------------------------------------------------------
// WHICH KERNEL RUNS A WEIGHT GRADIENT dW[N,K] += dY[M,N]^T X[M,K] (m3p_gemm_wgrad_plan reports it):
//
//   shape                                                  kernel                                   why
//   -----------------------------------------------------  ---------------------------------------  --------------------------------
//   M % 64 == 0, M >= 4096, N % 256 == 0, K % 256 == 0,    four-wave 256x256, one (tile, M-chunk)   1063 / 858 / 983 TF on FFN1 / FFN2 /
//     >= 9 output tiles, < 4 tiles per CU                    segment per workgroup; partial tiles     QKV against 845 / 751 / 808 on the
//                                                            go to the caller's workspace and a       ring kernel; no atomics, summed in
//                                                            reduce kernel (16 x ntile blocks) sums   slot order (bit-reproducible)
//                                                            them into dW in slot order
//   same, >= 4 tiles per CU (the vocabulary matrix)        four-wave, whole tiles round-robin       each tile flushed once: fp32 atomics,
//                                                                                                    or plain stores (m3p_gemm_wgrad_store_bf16)
//   M % 64 == 0, M >= 4096 otherwise                       ring stream-K 256x128 (atomics)          ragged N / K
//   anything else                                          128x128 split-M (atomics)                small M (cfg1, tests)
// ---------------------------------------------------------------------------------------------------------------------
static int wgrad_plan(int M, int N, int K, bool aligned) {
  if (aligned && g_variant >= 1 && g_variant != 3 && (M % 64) == 0 && M >= 4096 && (N % 256) == 0 && (K % 256) == 0 &&
      (N / 256) * (K / 256) >= ((g_ablate & 16) ? 1 : 9))
    return ((long long)(N / 256) * (K / 256) >= 4LL * num_cus()) ? M3P_KERN_WGRAD_W4_TILES : M3P_KERN_WGRAD_W4_CHUNKS;
  if (g_variant >= 1 && (M % BK) == 0 && M >= 4096) return M3P_KERN_WGRAD_RING;
  return M3P_KERN_WGRAD_128;
}

int m3p_gemm_nt_plan(int M, int N, int K, int epilogue) {
  if (M <= 0 || N <= 0 || K <= 0 || (K % BK) != 0 || epilogue < 0 || epilogue > M3P_EPI_BIAS_LSE) return M3P_EINVAL;
  int queue = 0;
  const int plan = nt_plan(epilogue, M, N, K, &queue);
  return (plan == M3P_KERN_NT_W8 && queue) ? M3P_KERN_NT_W8_QUEUE : plan;
}

int m3p_gemm_wgrad_plan(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return M3P_EINVAL;
  return wgrad_plan(M, N, K, true);
}

// Cf[M,N] += alpha * A[M,K] x W[K,N] on the four-wave kernel (YROWS form): the weight-gradient machinery with the roles
//   contraction = K, output rows = M (first operand A, contraction-contiguous), output columns = N (second operand W, rows = contraction)
int m3p_gemm_nn_w4_f32(const void* A, int lda, const void* W, int ldw, float* C, int ldc, int M, int N, int K, float alpha,
                       void* workspace, size_t workspace_bytes, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !W || !C) return M3P_EINVAL;
  if ((K % 64) || K < 4096 || (M % 256) || (N % 256) || (lda % 8) || (ldw % 8) || lda < K || ldw < N || ldc < N ||
      ((uintptr_t)A & 15) || ((uintptr_t)W & 15) || g_variant < 1 || g_variant == 3)
    return M3P_ENOTIMPL;
  return launch_wgrad_w4<true>(A, lda, W, ldw, C, ldc, K, M, N, alpha, workspace, workspace_bytes, stream, nullptr);
}

int m3p_gemm_wgrad_pair_bf16(const void* dYa, int lddya, const void* Xa, int ldxa, float* dWa, int lddwa, int Na, int Ka,
                             const void* dYb, int lddyb, const void* Xb, int ldxb, float* dWb, int lddwb, int Nb, int Kb,
                             int M, float alpha, void* workspace, size_t workspace_bytes, void* stream) {
  if (M <= 0 || Na <= 0 || Ka <= 0 || Nb <= 0 || Kb <= 0) return M3P_EINVAL;
  const bool pair = wgrad_w4_ok(M, Na, Ka, lddya, ldxa, dYa, Xa) && wgrad_w4_ok(M, Nb, Kb, lddyb, ldxb, dYb, Xb) &&
                    workspace && workspace_bytes >= m3p_gemm_wgrad_workspace_bytes() && (((uintptr_t)workspace & 15) == 0) &&
                    // (the workspace form's own conditions - launch_wgrad_w4 - so that a pair never lands on the atomic flush)
                    (lddwa % 4) == 0 && (lddwb % 4) == 0 && (((uintptr_t)dWa | (uintptr_t)dWb) & 15) == 0 &&
                    lddwa >= Ka && lddwb >= Kb &&
                    (long long)((Na / 256) * (Ka / 256) + (Nb / 256) * (Kb / 256)) * 2 <= num_cus();
  if (!pair) {
    int rc = m3p_gemm_wgrad_bf16(dYa, lddya, Xa, ldxa, dWa, lddwa, M, Na, Ka, alpha, workspace, workspace_bytes, stream);
    if (rc != M3P_OK) return rc;
    return m3p_gemm_wgrad_bf16(dYb, lddyb, Xb, ldxb, dWb, lddwb, M, Nb, Kb, alpha, workspace, workspace_bytes, stream);
  }
  if (lddya < Na || ldxa < Ka || lddyb < Nb || ldxb < Kb) return M3P_EINVAL;
  WgradProblem pb = {(const bf16*)dYb, (const bf16*)Xb, dWb, lddyb, ldxb, lddwb, Nb / 256, Kb / 256};
  return launch_wgrad_w4<>(dYa, lddya, Xa, ldxa, dWa, lddwa, M, Na, Ka, alpha, workspace, workspace_bytes, stream, &pb);
}

int m3p_gemm_wgrad_store_bf16(const void* dY, int lddy, const void* X, int ldx, float* dW, int lddw, int M, int N, int K,
                              float alpha, void* workspace, size_t workspace_bytes, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (lddy % 8) != 0 || (ldx % 8) != 0 || !dW || lddw < K) return M3P_EINVAL;
  if (lddy < N || ldx < K || ((uintptr_t)dY & 15) || ((uintptr_t)X & 15)) return M3P_EINVAL;
  if (!wgrad_w4_ok(M, N, K, lddy, ldx, dY, X)) return M3P_ENOTIMPL;
  return launch_wgrad_w4<>(dY, lddy, X, ldx, dW, lddw, M, N, K, alpha, workspace, workspace_bytes, stream, nullptr, true);
}

int m3p_gemm_wgrad_bf16(const void* dY, int lddy, const void* X, int ldx, float* dW, int lddw, int M, int N, int K,
                        float alpha, void* workspace, size_t workspace_bytes, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (lddy % 8) != 0 || (ldx % 8) != 0 || !dW || lddw < K) return M3P_EINVAL;
  if (lddy < ((N + 7) / 8) * 8 || ldx < ((K + 7) / 8) * 8) return M3P_EINVAL;
  if (((uintptr_t)dY & 15) || ((uintptr_t)X & 15)) return M3P_EINVAL;
  const int plan = wgrad_plan(M, N, K, wgrad_w4_ok(M, N, K, lddy, ldx, dY, X));
  if (plan == M3P_KERN_WGRAD_W4_CHUNKS || plan == M3P_KERN_WGRAD_W4_TILES) {      // (768 x 768 = 9 tiles included since the four-wave kernel lost its scalar overhead: 63 against 74 us on the ring kernel)
    return launch_wgrad_w4<>(dY, lddy, X, ldx, dW, lddw, M, N, K, alpha, workspace, workspace_bytes, stream, nullptr);
  }
  if (plan == M3P_KERN_WGRAD_RING) {
    const int ti = (N + WR_I - 1) / WR_I, tj = (K + WR_J - 1) / WR_J;
    const size_t lds = 3 * WR_STAGE;
    static bool attr_set_r = false;
    if (!attr_set_r) {
      hipError_t e = hipFuncSetAttribute((const void*)gemm_wgrad_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      attr_set_r = true;
    }
    const int grid = num_cus();
    const int nmt = M / BK;
    long long share = ((long long)ti * tj * nmt + grid - 1) / grid;
    int chunk = (int)(share < nmt ? (share < 1 ? 1 : share) : nmt);
    if ((long long)ti * tj >= 4LL * grid) chunk = 0;   // many tiles: round-robin whole tiles
    hipLaunchKernelGGL(gemm_wgrad_ring_kernel, dim3(grid), dim3(512), lds, (hipStream_t)stream, (const bf16*)dY, lddy,
                       (const bf16*)X, ldx, dW, lddw, M, N, K, alpha, ti, tj, chunk);
    M3P_CHECK_LAUNCH();
    return M3P_OK;
  }
  const int tiles_i = (N + WG_T - 1) / WG_T, tiles_j = (K + WG_T - 1) / WG_T;
  const int ntile = tiles_i * tiles_j;
  // split the contraction so that ~2 blocks per CU are in flight; chunk is a multiple of 64
  int split = (512 + ntile - 1) / ntile;
  const int max_split = (M + BK - 1) / BK;
  if (split > max_split) split = max_split;
  if (split < 1) split = 1;
  int m_chunk = ((M + split - 1) / split + BK - 1) / BK * BK;
  split = (M + m_chunk - 1) / m_chunk;
  const size_t lds = 4 * WG_TILE_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_wgrad_kernel, dim3(ntile * split), dim3(256), lds, (hipStream_t)stream,
                     (const bf16*)dY, lddy, (const bf16*)X, ldx, dW, lddw, M, N, K, alpha, tiles_i, tiles_j, m_chunk);
  M3P_CHECK_LAUNCH();
  return M3P_OK;
}

}  // extern "C"

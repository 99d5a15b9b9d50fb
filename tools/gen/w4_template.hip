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
@KT_A@
    }
    {
      constexpr bool FIRST = false;
@KT_B@
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
@PHASE1@
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
@PHASE2@
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


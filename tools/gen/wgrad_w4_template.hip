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
@PHASE1@
    __builtin_amdgcn_sched_barrier(0);
    if (PEND) load_done();
  };
  auto phase2 = [&](auto stage_c) {
    constexpr bool FIRST = false;
    constexpr int s_cur = decltype(stage_c)::value;
    __builtin_amdgcn_sched_barrier(0);
@PHASE2@
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
@WKT_A@
    }
    {
      constexpr bool FIRST = false;
@WKT_B@
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


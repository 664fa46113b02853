// Tap-GEMM kernel template and its launchers.  Included by tapgemm_inst.hip, which is compiled once per
// (mode, dtype) pair (-DL2S_INST_MODE=.. -DL2S_INST_ET=..) so the 48 kernel instantiations build in parallel.
#pragma once
// Tap-GEMM: the one dense-contraction kernel family of the path (Linear / Conv1d / Conv2d / ConvTranspose1d phases).
//
//   C[o(m), n] = epi( sum_tap sum_c A[src(m,tap), c] * W[n, tap*Cin + c] )
//
// gfx950 design: 4 or 8 waves, block tile BM x BN, K-tile 64 (two v_mfma_f32_16x16x32 k-steps per 16x16
// sub-tile).  Both operands are staged global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no
// VGPR round trip, no ds_write): the per-lane SOURCE address carries the im2col gather (Linear / Conv1d / Conv2d taps),
// the zero fill (lanes outside the tensor read a 16-byte zero page) and the XOR swizzle (chunk ^ (row & 7) on 128-byte
// rows) that makes every ds_read_b128 fragment read bank-conflict free; the LDS destination stays lane-linear.
// 2- or 3-deep LDS ring (up to 144 KiB): with 3 stages two tiles stay in flight behind a counted s_waitcnt vmcnt(N)
// and a raw s_barrier (a __syncthreads() would drain the DMA queue), one barrier per tile; 256-row tiles run 8 waves.
// Blocks are persistent: each walks its list of output tiles with the K-loop flattened across tiles, so the ring keeps
// streaming the next tile's operands under the current tile's epilogue; tiles are dealt so that the blocks of one XCD
// (blockIdx % 8) work on neighbouring tiles and share operand panels in that XCD's L2.  The MFMA is issued
// "swapped" (W rows as the A operand) so each lane ends with 4 consecutive output channels of one output row: bias /
// residual / store are 8- or 16-byte vectors.
#include "tapgemm_common.h"
#include "tapgemm_tiles.h"

namespace {

constexpr int BK = 64;        // K per tile = two MFMA k-steps of 32
constexpr int CPR = BK / 8;   // 16-byte chunks per LDS row

using namespace l2s;

// (4-wave tiles must fit two waves per SIMD - 256 VGPRs - so that the two blocks per CU their LDS admits are resident:
//  unbounded, the 128x128 tile took 332 registers and ran one block per CU at 0.6x of its speed with the cap)
template <typename ET, int BM, int BN, int WM_, int WN_, int MODE, int STAGES, int EPI, bool UNI>
__global__ __launch_bounds__(WM_* WN_ * 64, (WM_ * WN_ <= 4 ? 2 : 1)) void tapgemm_kernel(const l2s_gemm_desc p, const int tilesM,
                                                                 const int tilesN, const int chunk,
                                                                 const int band) {
  constexpr int NWAVES = WM_ * WN_;
  constexpr int WAVE_M = BM / WM_, WAVE_N = BN / WN_;
  constexpr int MI = WAVE_M / 16, NI = WAVE_N / 16;
  constexpr int A_INSTR = BM * CPR / 64, W_INSTR = BN * CPR / 64;  // 1-KiB LDS-DMA wave-instructions per tile
  constexpr int A_PER_W = A_INSTR / NWAVES;
  constexpr int W_PER_W = (W_INSTR + NWAVES - 1) / NWAVES;
  constexpr int BUF = (BM + BN) * BK;  // elements per LDS stage: A image then W image
  static_assert(A_INSTR % NWAVES == 0 && (W_INSTR % NWAVES == 0 || W_PER_W == 1), "DMA split");
  static_assert(STAGES == 2 || STAGES == 3, "pipeline depth");
  static_assert(MI <= 8 && NI <= 4, "fragment unroll");

  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];  // STAGES * BUF elements

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN_, wn = wave % WN_;
  const int Cin = p.Cin;
  const int Ktot = Cin * p.ntaps;
  const int nk = (Ktot + BK - 1) / BK;
  const float inv_cin = 1.0f / (float)Cin;
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(&g_zero16);

  // ---- persistent tile schedule ------------------------------------------------------------------------------
  // Linear tile id L walks bands of `band` M-tiles: inside a band tm runs fastest, then tn, then the next band, then the
  // next group.  The host sizes the band so that the operand panels one XCD's contiguous range touches are smallest
  // (band = 1: tn fastest; band = tilesM: tm fastest).  XCD x (= blockIdx % 8: blocks b and b+8 share an L2) owns the
  // contiguous range [x*chunk, (x+1)*chunk); its blocks take L = lo + slot + i*slots.  Placement only affects speed.
  const int ntiles = tilesM * tilesN * (p.groups > 0 ? p.groups : 1);
  const int slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int lo = (blockIdx.x & 7) * chunk;
  const int hi = lo + chunk < ntiles ? lo + chunk : ntiles;
  const int my_n = (lo + slot < hi) ? (hi - lo - slot + slots - 1) / slots : 0;
  if (my_n == 0) return;
  const int total = my_n * nk;  // K-tiles this block computes, flattened over its output tiles
  auto tile_coords = [&](int i, int& m0, int& n0, int& grp) {
    const int L = lo + slot + i * slots;
    const int per_grp = tilesM * tilesN;
    grp = L / per_grp;
    const int l = L - grp * per_grp;
    const int bsz = band * tilesN;               // tiles in a full band
    const int bi = l / bsz, idx = l - bi * bsz;
    const int rows = tilesM - bi * band < band ? tilesM - bi * band : band;  // the last band may be shorter
    const int tn = idx / rows;
    m0 = (bi * band + idx - tn * rows) * BM;
    n0 = tn * BN;
  };

  // ---- LDS-DMA staging assignment ---------------------------------------------------------------------------
  // LDS image: position q = row*8 + cpos holds global chunk (cpos ^ (row & 7)) of that row (XOR swizzle applied on the
  // SOURCE address; the DMA destination is lane-linear).  One wave-instruction fills positions [64*i, 64*i+64).
  // Rows past M / N are clamped to the last valid row (their outputs are never stored), so only the conv padding and
  // the K tail need the zero page.  The per-tile address math is kept to a few VALU ops per DMA: one tap per K-tile
  // whenever Cin % 64 == 0 (tracked incrementally, no division), per-lane taps otherwise.
  const int srow = lane >> 3;                     // row within the instruction's 8 rows
  const int schunk = (lane & 7) ^ (srow & 7);     // global chunk this lane fetches (same for every instruction)
  const bool ktail = UNI ? false : (Ktot % BK) != 0;   // a uniform-tap problem has Cin % 64 == 0, hence no K tail
  // UNI (conv modes): Cin % 64 == 0, so a K-tile lies inside ONE tap - compile-time, so the DMA issue of the K loop is
  // branch-free and the rarely used per-lane-tap path does not sit in the hot loop's instruction footprint
  constexpr bool uni_tap = UNI;
  const uint16_t* a_ptr[A_PER_W];
  int a_t[A_PER_W], a_x[A_PER_W];
  const uint16_t* w_ptr[W_PER_W];
  int run_tap = 0, run_c = 0, run_ky = 0, run_kx = 0;  // (tap, channel offset) of the next K-tile to issue

  auto setup_issue = [&](int i) {  // operand row pointers of this block's i-th output tile
    int m0, n0, grp;
    tile_coords(i, m0, n0, grp);
    const uint16_t* A = (const uint16_t*)p.A + grp * p.a_gstride;
    const uint16_t* W = (const uint16_t*)p.W + (int64_t)grp * p.w_gstride;
#pragma unroll
    for (int j = 0; j < A_PER_W; ++j) {
      int m = m0 + (wave * A_PER_W + j) * 8 + srow;
      m = m < p.M ? m : p.M - 1;
      a_t[j] = 0; a_x[j] = 0;
      if (MODE == L2S_MODE_LINEAR) {
        a_ptr[j] = A + (int64_t)m * p.lda + schunk * 8;
      } else if (MODE == L2S_MODE_CONV1D) {
        const int b = m / p.T_out, t = m - b * p.T_out;
        a_ptr[j] = A + (int64_t)b * p.T_in * p.lda;
        a_t[j] = t * p.stride + p.off;
      } else {
        const int hw = p.Ho * p.Wo;
        const int img = m / hw, rem = m - img * hw;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        a_ptr[j] = A + (int64_t)img * p.Hi * p.Wi * p.lda;
        a_t[j] = oy * p.stride - p.pad;
        a_x[j] = ox * p.stride - p.pad;
      }
    }
#pragma unroll
    for (int j = 0; j < W_PER_W; ++j) {
      int n = n0 + (wave * W_PER_W + j) * 8 + srow;
      n = n < p.N ? n : p.N - 1;
      w_ptr[j] = W + (int64_t)n * Ktot + schunk * 8;
    }
    run_tap = 0; run_c = 0; run_ky = 0; run_kx = 0;
  };

  // dma_issue only issues (no state change, no branch in the LINEAR / uniform-tap paths) so that it can share a basic
  // block with MFMAs; `live` = false turns every source into the zero page (the dummy issues that keep the last
  // K-tiles of a block on the same straight-line path).  dma_advance moves the tap cursor afterwards.
  auto dma_issue = [&](int kt, int buf, const bool live) {  // kt runs 0,1,2,... within a tile
    const int k0 = kt * BK;
    uint16_t* dstA = lds + buf * BUF;
    uint16_t* dstW = dstA + BM * BK;
    const bool kok = !ktail || (k0 + schunk * 8 < Ktot);
    if (MODE == L2S_MODE_LINEAR) {
#pragma unroll
      for (int j = 0; j < A_PER_W; ++j) {
        const uint16_t* g = a_ptr[j] + k0;
        g = (live && (!ktail || kok)) ? g : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(dstA + (wave * A_PER_W + j) * 512), 16, 0, 0);
      }
    } else if (uni_tap) {
      const int coff = run_c + schunk * 8;
#pragma unroll
      for (int j = 0; j < A_PER_W; ++j) {
        bool ok;
        int off;
        if (MODE == L2S_MODE_CONV1D) {
          const int st = a_t[j] + run_tap * p.dil;
          ok = (unsigned)st < (unsigned)p.T_in;
          off = st * p.lda + coff;
        } else {
          const int iy = a_t[j] + run_ky, ix = a_x[j] + run_kx;
          ok = ((unsigned)iy < (unsigned)p.Hi) && ((unsigned)ix < (unsigned)p.Wi);
          off = (iy * p.Wi + ix) * p.lda + coff;
        }
        const uint16_t* g = (ok && live) ? a_ptr[j] + off : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(dstA + (wave * A_PER_W + j) * 512), 16, 0, 0);
      }
    } else {
      const int kk = k0 + schunk * 8;
      const int tap = (int)(((float)kk + 0.5f) * inv_cin);
      const int cc = kk - tap * Cin;
#pragma unroll
      for (int j = 0; j < A_PER_W; ++j) {
        bool ok = kok;
        int off;
        if (MODE == L2S_MODE_CONV1D) {
          const int st = a_t[j] + tap * p.dil;
          ok = ok && ((unsigned)st < (unsigned)p.T_in);
          off = st * p.lda + cc;
        } else {
          const int ky = tap / p.KW, kx = tap - ky * p.KW;
          const int iy = a_t[j] + ky, ix = a_x[j] + kx;
          ok = ok && ((unsigned)iy < (unsigned)p.Hi) && ((unsigned)ix < (unsigned)p.Wi);
          off = (iy * p.Wi + ix) * p.lda + cc;
        }
        const uint16_t* g = (ok && live) ? a_ptr[j] + off : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(dstA + (wave * A_PER_W + j) * 512), 16, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < W_PER_W; ++j) {
      if (W_INSTR >= NWAVES || wave * W_PER_W + j < W_INSTR) {  // wave-uniform (compile-time true for the wide tiles)
        const uint16_t* g = w_ptr[j] + k0;
        g = (live && (!ktail || kok)) ? g : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(dstW + (wave * W_PER_W + j) * 512), 16, 0, 0);
      }
    }
  };
  auto dma_advance = [&]() {  // tap cursor of the uniform-tap path: one K-tile further
    if (MODE != L2S_MODE_LINEAR && uni_tap) {
      run_c += BK;
      if (run_c >= Cin) {
        run_c = 0;
        ++run_tap;
        if (++run_kx == p.KW) { run_kx = 0; ++run_ky; }
      }
    }
  };

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int lm = lane & 15, lg = lane >> 4;
  // fragment of k-step ks lives at position row*8 + ((ks*4 + lg) ^ (row & 7)); row & 7 == lm & 7 for every sub-tile
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)lds;
  const uint32_t k0_off = (uint32_t)(lm * CPR + ((0 + lg) ^ (lm & 7))) * 16;   // byte offsets inside a sub-tile
  const uint32_t k1_off = (uint32_t)(lm * CPR + ((4 + lg) ^ (lm & 7))) * 16;
  const uint32_t a_frag_off = (uint32_t)(wm * WAVE_M) * (BK * 2);
  const uint32_t w_frag_off = (uint32_t)(BM + wn * WAVE_N) * (BK * 2);
  auto read_frags = [&](frag16(&fa)[MI], frag16(&fw)[NI], uint32_t aa, uint32_t aw) {
    // sub-tile i sits 16 rows = 2048 bytes further: immediate offsets
    lds_read_b128<0>(fa[0], aa);
    if (MI > 1) lds_read_b128<2048>(fa[MI > 1 ? 1 : 0], aa);
    if (MI > 2) lds_read_b128<4096>(fa[MI > 2 ? 2 : 0], aa);
    if (MI > 3) lds_read_b128<6144>(fa[MI > 3 ? 3 : 0], aa);
    if (MI > 4) lds_read_b128<8192>(fa[MI > 4 ? 4 : 0], aa);
    if (MI > 5) lds_read_b128<10240>(fa[MI > 5 ? 5 : 0], aa);
    if (MI > 6) lds_read_b128<12288>(fa[MI > 6 ? 6 : 0], aa);
    if (MI > 7) lds_read_b128<14336>(fa[MI > 7 ? 7 : 0], aa);
    lds_read_b128<0>(fw[0], aw);
    if (NI > 1) lds_read_b128<2048>(fw[NI > 1 ? 1 : 0], aw);
    if (NI > 2) lds_read_b128<4096>(fw[NI > 2 ? 2 : 0], aw);
    if (NI > 3) lds_read_b128<6144>(fw[NI > 3 ? 3 : 0], aw);
  };

  // ---- main loop: STAGES-deep LDS ring, up to STAGES-1 K-tiles in flight, one barrier per K-tile ----------------
  // K-tile g is ordered for this wave's ds_reads by: the issuing waves' counted vmcnt (their DMA of g retired), then
  // the barrier every reader passes.  The same barrier proves every wave finished reading the stage of g-1, which the
  // DMA issued right after it overwrites.  vmcnt counts in issue order and every later operation (an epilogue's loads
  // and stores) is younger than the DMA being waited for, so the counted wait can over-wait but never under-wait.
  const bool wave_has_w = (W_INSTR >= NWAVES) || (wave < W_INSTR);
  int s_i = 0, s_kt = 0, s_stage = 0, issued = 0;  // issue cursor
  auto issue_advance = [&]() {   // cursor bookkeeping of one issued K-tile (branches live here, away from the MFMAs)
    dma_advance();
    ++issued;
    s_stage = s_stage + 1 == STAGES ? 0 : s_stage + 1;
    if (++s_kt == nk) {
      s_kt = 0;
      if (++s_i < my_n) setup_issue(s_i);
    }
  };
  auto issue_next = [&]() {
    dma_issue(s_kt, s_stage, true);
    issue_advance();
  };
  setup_issue(0);
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (issued < total) issue_next();

  int c_i = 0, stage = 0;  // compute cursor
  // The loop is software-pipelined ACROSS the per-K-tile barrier: a K-tile is two 32-deep MFMA steps (k0, k1); the
  // barrier that hands over the next ring stage sits between them, so the wait for the next K-tile's DMA, the barrier
  // skew, the next DMA issue and the LDS round trip of the next k0 fragments all run under 16 MFMAs already issued.
  auto wait_stage = [&](int gi) {  // K-tile gi has landed (K-tile gi+1 may stay in flight)
    if (STAGES == 3 && issued - gi - 1 > 0) {
      if (wave_has_w) wait_vmcnt<A_PER_W + W_PER_W>(); else wait_vmcnt<A_PER_W>();
    } else {
      wait_vmcnt<0>();
    }
  };
  int g = 0;  // K-tiles consumed so far (all tiles of this block)
  for (int ti = 0; ti < my_n; ++ti) {
    frag16 fa0[MI], fw0[NI], fa1[MI], fw1[NI];
    {  // first K-tile of the output tile: nothing of it is in registers yet
      const uint32_t sbase = lds_base + (uint32_t)stage * (BUF * 2);
      wait_stage(g);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (issued < total) issue_next();
      read_frags(fa0, fw0, sbase + a_frag_off + k0_off, sbase + w_frag_off + k0_off);
      lds_wait();
    }
    for (int kt = 0; kt < nk; ++kt, ++g) {
      const uint32_t sbase = lds_base + (uint32_t)stage * (BUF * 2);
      read_frags(fa1, fw1, sbase + a_frag_off + k1_off, sbase + w_frag_off + k1_off);  // under the k0 MFMA step
      __builtin_amdgcn_sched_barrier(0);   // reads first: hipcc otherwise hoists the (register-only) MFMAs above them
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = ET::mfma(fw0[j], fa0[i], acc[i][j]);
      lds_wait();                                    // this wave has read everything it needs from the stage
      const int nstage = stage + 1 == STAGES ? 0 : stage + 1;
      const bool more_k = kt + 1 < nk;
      if (more_k) {
        wait_stage(g + 1);
        __builtin_amdgcn_s_barrier();                // K-tile g+1 visible to all; the stage of K-tile g is free
        asm volatile("" ::: "memory");
        // ... and is refilled right away.  The DMA issue (a dummy from the zero page once the block has nothing left to
        // fetch), the next k0 fragment reads and the k1 MFMA step form ONE basic block, and the sched_group_barriers ask
        // for "2 MFMAs, 1 DMA" so the address math and the issue slots of the DMAs hide under the MFMAs instead of
        // delaying them (both waves of a SIMD leave the barrier together, so nobody else would feed the MFMA pipe).
        const bool live = issued < total;
        const uint32_t nbase = lds_base + (uint32_t)nstage * (BUF * 2);
        read_frags(fa0, fw0, nbase + a_frag_off + k0_off, nbase + w_frag_off + k0_off);  // under the k1 MFMA step
        dma_issue(s_kt, s_stage, live);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = ET::mfma(fw1[j], fa1[i], acc[i][j]);
#pragma unroll
        for (int q = 0; q < A_PER_W + W_PER_W; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMA
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read (LDS-DMA)
        }
        lds_wait();
        if (live) issue_advance();
      } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = ET::mfma(fw1[j], fa1[i], acc[i][j]);
      }
      stage = nstage;
    }
    int m0, n0, grp;
    tile_coords(c_i++, m0, n0, grp);

    // ---- epilogue (tapgemm_common.h): LDS-transposed, 16-byte row-contiguous loads/stores --------------------------
    // The scratch lives in the ring stage that was just consumed; the other stages keep receiving the next tile's DMA.
    constexpr int SCRB = epilogue_scratch_bytes<MI, NI>();
    static_assert(NWAVES * SCRB <= BUF * 2, "epilogue scratch must fit one ring stage");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // every wave has finished reading the consumed stage
    const int free_stage = stage == 0 ? STAGES - 1 : stage - 1;
    const uint32_t scr = lds_base + (uint32_t)free_stage * (BUF * 2) + (uint32_t)wave * SCRB;
    epilogue<ET, MI, NI, EPI>(p, acc, scr, lane, m0 + wm * WAVE_M, n0 + wn * WAVE_N, grp, [&](int m) -> int64_t {
      return m < p.M ? (int64_t)m * p.out_row_mul + p.out_row_add : (int64_t)-1;
    });
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
}

template <typename ET, int BM, int BN, int WM_, int WN_, int MODE, int STAGES, int EPI, bool UNI>
int launch_tile(const l2s_gemm_desc& d, hipStream_t st) {
  constexpr int SMEM = STAGES * (BM + BN) * BK * 2;
  constexpr int BPC_LDS = (160 * 1024) / SMEM;                         // blocks per CU the LDS admits
  constexpr int BPC = BPC_LDS < (32 / (WM_ * WN_)) ? BPC_LDS : (32 / (WM_ * WN_));
  auto kern = tapgemm_kernel<ET, BM, BN, WM_, WN_, MODE, STAGES, EPI, UNI>;
  static L2sSmemOptIn opt_in;  // > 64 KB of dynamic LDS: opt-in per instantiation and device
  if (int e = l2s_smem_opt_in(kern, SMEM, opt_in)) return e;
  const int G = d.groups > 0 ? d.groups : 1;
  const int tilesM = (d.M + BM - 1) / BM, tilesN = (d.N + BN - 1) / BN;
  const int ntiles = tilesM * tilesN * G;
  const int chunk = (ntiles + 7) / 8;                                  // tiles per XCD
  const int slots = chunk < 32 * BPC ? chunk : 32 * BPC;               // blocks per XCD (32 CUs each)
  // bytes of operand panels one XCD's contiguous tile range touches under either tile order
  const double ap = (double)BM * d.Cin * 2.0, wp = (double)BN * d.Cin * d.ntaps * 2.0;
  auto cdivi = [](int a, int b) { return (a + b - 1) / b; };
  int band = 1;
  double best = 1e300;
  for (int b = 1; b <= tilesM; ++b) {  // a range of `chunk` tiles spans ~b A-panels and ~chunk/b W-panels (capped)
    const int wn = cdivi(chunk, b) < tilesN ? cdivi(chunk, b) : tilesN;
    const int an = b * cdivi(chunk, b * tilesN);
    const double fp = ap * (an < tilesM ? an : tilesM) + wp * wn;
    if (fp < best) { best = fp; band = b; }
  }
  static const int band_env = [] { const char* e = getenv("L2S_BAND"); return e ? atoi(e) : 0; }();  // tuning aid
  if (band_env > 0) band = band_env < tilesM ? band_env : tilesM;
  hipLaunchKernelGGL(kern, dim3(8 * slots), dim3(WM_ * WN_ * 64), SMEM, st, d, tilesM, tilesN, chunk, band);
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

template <typename ET, int MODE, int EPI, bool UNI>
int launch_mode_uni(const l2s_gemm_desc& d, hipStream_t st) {
  switch (pick_tile(d.M, d.N, d.groups > 0 ? d.groups : 1)) {
    case 128016: return launch_tile<ET, 128, 16, 4, 1, MODE, 2, EPI, UNI>(d, st);
    case 128032: return launch_tile<ET, 128, 32, 4, 1, MODE, 3, EPI, UNI>(d, st);
    case 256128: return launch_tile<ET, 256, 128, 4, 2, MODE, 3, EPI, UNI>(d, st);
    case 256064: return launch_tile<ET, 256, 64, 4, 2, MODE, 3, EPI, UNI>(d, st);
    case 128128: return launch_tile<ET, 128, 128, 2, 2, MODE, 2, EPI, UNI>(d, st);
    case 128064: return launch_tile<ET, 128, 64, 2, 2, MODE, 3, EPI, UNI>(d, st);
    default: return launch_tile<ET, 64, 64, 2, 2, MODE, 3, EPI, UNI>(d, st);
  }
}

template <typename ET, int MODE, int EPI>
int launch_mode(const l2s_gemm_desc& d, hipStream_t st) {
  if constexpr (MODE == L2S_MODE_LINEAR) return launch_mode_uni<ET, MODE, EPI, false>(d, st);
  else return (d.Cin % BK == 0) ? launch_mode_uni<ET, MODE, EPI, true>(d, st) : launch_mode_uni<ET, MODE, EPI, false>(d, st);
}

}  // namespace

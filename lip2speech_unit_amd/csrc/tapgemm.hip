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
#include "l2s_common.h"
#include <cstdlib>

namespace {

constexpr int BK = 64;        // K per tile = two MFMA k-steps of 32
constexpr int CPR = BK / 8;   // 16-byte chunks per LDS row

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// source of every zero-filled 16-byte chunk (conv padding, M/N/K tails): LDS-DMA cannot write an immediate
__device__ const uint4 g_zero16 = {0u, 0u, 0u, 0u};

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// Fragment reads are inline asm on purpose: for a compiler-visible LDS load hipcc (ROCm 7.2) inserts s_waitcnt vmcnt(0)
// while any LDS-DMA is outstanding, which would drain the tiles we keep in flight.  The reads are ordered against the
// DMA by the counted vmcnt + barrier of the main loop and against the MFMAs by lds_wait() below.
template <int OFF>
__device__ __forceinline__ void lds_read_b128(frag16& f, uint32_t addr) {
  u32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  f.u = make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);  // keep register-only MFMAs below the wait (they do not touch memory)
}

template <int OFF>
__device__ __forceinline__ void lds_write_f4(uint32_t addr, f32x4_t v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ f32x4_t lds_read_f4(uint32_t addr) {
  f32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename ET, int BM, int BN, int WM_, int WN_, int MODE, int STAGES>
__global__ __launch_bounds__(WM_* WN_ * 64) void tapgemm_kernel(const l2s_gemm_desc p, const int tilesM,
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
  static_assert(MI <= 4 && NI <= 4, "fragment unroll");

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
  const bool ktail = (Ktot % BK) != 0;
  const bool uni_tap = (MODE != L2S_MODE_LINEAR) && (Cin % BK == 0);
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

  auto dma_issue = [&](int kt, int buf) {  // kt runs 0,1,2,... within a tile
    const int k0 = kt * BK;
    uint16_t* dstA = lds + buf * BUF;
    uint16_t* dstW = dstA + BM * BK;
    const bool kok = !ktail || (k0 + schunk * 8 < Ktot);
    if (MODE == L2S_MODE_LINEAR) {
#pragma unroll
      for (int j = 0; j < A_PER_W; ++j) {
        const uint16_t* g = a_ptr[j] + k0;
        if (ktail) g = kok ? g : zero;
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
        const uint16_t* g = ok ? a_ptr[j] + off : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(dstA + (wave * A_PER_W + j) * 512), 16, 0, 0);
      }
      run_c += BK;
      if (run_c >= Cin) {
        run_c = 0;
        ++run_tap;
        if (++run_kx == p.KW) { run_kx = 0; ++run_ky; }
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
        const uint16_t* g = ok ? a_ptr[j] + off : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(dstA + (wave * A_PER_W + j) * 512), 16, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < W_PER_W; ++j) {
      if (wave * W_PER_W + j < W_INSTR) {  // wave-uniform
        const uint16_t* g = w_ptr[j] + k0;
        if (ktail) g = kok ? g : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(dstW + (wave * W_PER_W + j) * 512), 16, 0, 0);
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
  auto issue_next = [&]() {
    dma_issue(s_kt, s_stage);
    ++issued;
    s_stage = s_stage + 1 == STAGES ? 0 : s_stage + 1;
    if (++s_kt == nk) {
      s_kt = 0;
      if (++s_i < my_n) setup_issue(s_i);
    }
  };
  setup_issue(0);
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (issued < total) issue_next();

  int c_i = 0, c_kt = 0, stage = 0;  // compute cursor
  for (int g = 0; g < total; ++g) {
    if (STAGES == 3 && issued - g - 1 > 0) {
      if (wave_has_w) wait_vmcnt<A_PER_W + W_PER_W>(); else wait_vmcnt<A_PER_W>();  // K-tile g+1 may stay in flight
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const uint32_t sbase = lds_base + (uint32_t)stage * (BUF * 2);
    const uint32_t aA = sbase + a_frag_off, aW = sbase + w_frag_off;
    frag16 fa0[MI], fw0[NI], fa1[MI], fw1[NI];
    read_frags(fa0, fw0, aA + k0_off, aW + k0_off);
    if (issued < total) issue_next();  // address math + DMA issue run under the fragment reads' latency
    lds_wait();
    read_frags(fa1, fw1, aA + k1_off, aW + k1_off);  // in flight under the first 32-deep MFMA step
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = ET::mfma(fw0[j], fa0[i], acc[i][j]);
    lds_wait();
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = ET::mfma(fw1[j], fa1[i], acc[i][j]);
    stage = stage + 1 == STAGES ? 0 : stage + 1;
    if (++c_kt < nk) continue;
    c_kt = 0;
    int m0, n0, grp;
    tile_coords(c_i++, m0, n0, grp);

    // ---- epilogue -------------------------------------------------------------------------------------------------
    // The MFMA leaves each lane with 4 consecutive channels of 16 different rows: storing that directly costs one
    // partial cache line per lane (store-issue bound, ~0.5 us per 16x16 sub-tile).  Instead each wave transposes 16 rows
    // at a time through a private LDS scratch (in the ring stage that was just consumed; the other stages keep
    // receiving the next tile's DMA) so that a lane owns 8 consecutive channels of one row: bias / residual / accumulate
    // loads and the stores become 16-byte accesses and a wave-instruction covers whole 128-byte row segments.
    constexpr int LPR = WAVE_N / 8;               // lanes per row after the transpose
    constexpr int SROW = WAVE_N + 4;              // scratch row stride in floats
    static_assert(NWAVES * 16 * SROW * 4 <= BUF * 2, "epilogue scratch must fit one ring stage");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // every wave has finished reading the consumed stage
    const int free_stage = stage == 0 ? STAGES - 1 : stage - 1;
    const uint32_t scr = lds_base + (uint32_t)free_stage * (BUF * 2) + (uint32_t)wave * (16 * SROW * 4);
    const uint32_t scr_w = scr + (uint32_t)(lm * SROW + lg * 4) * 4;
    const int rr = lane / LPR, cc = lane - rr * LPR;
    const uint32_t scr_r = scr + (uint32_t)(rr * SROW + cc * 8) * 4;
    const bool lane_on = lane < 16 * LPR;
    const int flags = p.flags;
    const int n = n0 + wn * WAVE_N + cc * 8;
    const bool ok_lo = lane_on && (n < p.N), ok_hi = lane_on && (n + 4 < p.N);
    const int col = grp * p.c_gstride + n;
    const float* bias = p.bias ? p.bias + grp * p.N : nullptr;
    const float* slope = (p.act == L2S_ACT_PRELU) ? p.slope + grp * p.N : nullptr;
    float bv[8], sv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bv[e] = 0.f; sv[e] = 0.f; }
    if (bias && ok_lo) { const float4 q = *reinterpret_cast<const float4*>(bias + n); bv[0] = q.x; bv[1] = q.y; bv[2] = q.z; bv[3] = q.w; }
    if (bias && ok_hi) { const float4 q = *reinterpret_cast<const float4*>(bias + n + 4); bv[4] = q.x; bv[5] = q.y; bv[6] = q.z; bv[7] = q.w; }
    if (slope && ok_lo) { const float4 q = *reinterpret_cast<const float4*>(slope + n); sv[0] = q.x; sv[1] = q.y; sv[2] = q.z; sv[3] = q.w; }
    if (slope && ok_hi) { const float4 q = *reinterpret_cast<const float4*>(slope + n + 4); sv[4] = q.x; sv[5] = q.y; sv[6] = q.z; sv[7] = q.w; }
    const bool has_res = (flags & (L2S_F_RES_PRE | L2S_F_RES_POST)) != 0;
    auto ld8 = [&](const void* base, int64_t row, int ld, bool f32, float(&out)[8]) {  // 8 values at (row, col)
      if (f32) {
        const float* q = (const float*)base + row * ld + col;
        if (ok_lo) { const float4 t = *reinterpret_cast<const float4*>(q); out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w; }
        if (ok_hi) { const float4 t = *reinterpret_cast<const float4*>(q + 4); out[4] = t.x; out[5] = t.y; out[6] = t.z; out[7] = t.w; }
      } else {
        const uint16_t* q = (const uint16_t*)base + row * ld + col;
        if (ok_lo) {
          const uint2 t = *reinterpret_cast<const uint2*>(q);
          out[0] = ET::to_f32((uint16_t)(t.x & 0xffff)); out[1] = ET::to_f32((uint16_t)(t.x >> 16));
          out[2] = ET::to_f32((uint16_t)(t.y & 0xffff)); out[3] = ET::to_f32((uint16_t)(t.y >> 16));
        }
        if (ok_hi) {
          const uint2 t = *reinterpret_cast<const uint2*>(q + 4);
          out[4] = ET::to_f32((uint16_t)(t.x & 0xffff)); out[5] = ET::to_f32((uint16_t)(t.x >> 16));
          out[6] = ET::to_f32((uint16_t)(t.y & 0xffff)); out[7] = ET::to_f32((uint16_t)(t.y >> 16));
        }
      }
    };
    auto st8_16 = [&](void* base, int64_t row, int ld, const float(&v)[8]) {  // 8 values stored as 16-bit
      uint16_t* q = (uint16_t*)base + row * ld + col;
      uint4 t;
      t.x = (uint32_t)ET::from_f32(v[0]) | ((uint32_t)ET::from_f32(v[1]) << 16);
      t.y = (uint32_t)ET::from_f32(v[2]) | ((uint32_t)ET::from_f32(v[3]) << 16);
      t.z = (uint32_t)ET::from_f32(v[4]) | ((uint32_t)ET::from_f32(v[5]) << 16);
      t.w = (uint32_t)ET::from_f32(v[6]) | ((uint32_t)ET::from_f32(v[7]) << 16);
      if (ok_hi && (((uintptr_t)q & 15) == 0)) {
        *reinterpret_cast<uint4*>(q) = t;
      } else {
        if (ok_lo) *reinterpret_cast<uint2*>(q) = make_uint2(t.x, t.y);
        if (ok_hi) *reinterpret_cast<uint2*>(q + 4) = make_uint2(t.z, t.w);
      }
    };
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      // 16 rows of this wave's sub-tile -> scratch (N-tile j at floats [16j, 16j+16) of a row) -> 8 channels per lane
      lds_write_f4<0>(scr_w, acc[i][0]);
      if (NI > 1) lds_write_f4<64>(scr_w, acc[i][NI > 1 ? 1 : 0]);
      if (NI > 2) lds_write_f4<128>(scr_w, acc[i][NI > 2 ? 2 : 0]);
      if (NI > 3) lds_write_f4<192>(scr_w, acc[i][NI > 3 ? 3 : 0]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const f32x4_t lo = lds_read_f4<0>(scr_r), hi = lds_read_f4<16>(scr_r);
      lds_wait();
      const int m = m0 + wm * WAVE_M + i * 16 + rr;
      if (!lane_on || m >= p.M || !ok_lo) continue;
      const int64_t o = (int64_t)m * p.out_row_mul + p.out_row_add;
      bool keep = true;
      if (flags & L2S_F_MASK) {
        const int clip = (int)(o / p.mask_T);
        const int t = (int)(o - (int64_t)clip * p.mask_T);
        keep = t < p.lens[clip] * p.mask_mul;
      }
      float rv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, cv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (has_res) ld8(p.R, o, p.ldr, (flags & L2S_F_RES_F32) != 0, rv);
      if (flags & L2S_F_ACCUM) ld8(p.C, o, p.ldc, (flags & L2S_F_OUT_F32) != 0, cv);
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (v[e] + bv[e]) * p.alpha;
      if (flags & L2S_F_RES_PRE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      }
      switch (p.act) {
        case L2S_ACT_RELU:
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          break;
        case L2S_ACT_GELU:
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = l2s_gelu(v[e]);
          break;
        case L2S_ACT_SWISH:
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = l2s_swish(v[e]);
          break;
        case L2S_ACT_PRELU:
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * sv[e];
          break;
        case L2S_ACT_LRELU:
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * p.act_slope;
          break;
        case L2S_ACT_TANH:
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
          break;
        default: break;
      }
      if (flags & L2S_F_RES_POST) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      }
      if (flags & L2S_F_ACCUM) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += cv[e];
      }
      if (!keep) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
      }
      if (flags & L2S_F_OUT_F32) {
        float* q = (float*)p.C + o * p.ldc + col;
        *reinterpret_cast<float4*>(q) = make_float4(v[0], v[1], v[2], v[3]);
        if (ok_hi) *reinterpret_cast<float4*>(q + 4) = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        st8_16(p.C, o, p.ldc, v);
      }
      if (flags & L2S_F_DUAL) {
        float w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = v[e] >= 0.f ? v[e] : v[e] * p.slope2;
        st8_16(p.C2, o, p.ldc2, w);
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
}

// ---- host side: tile choice and persistent grid ---------------------------------------------------------------
struct TileCfg { int bm, bn; float eff; };
// eff = measured steady-state speed relative to the 256x128 tile (tools/gemm_bench.py); cost = tiles on the busiest CU x tile size / eff
inline int pick_tile(int M, int N, int G) {
  auto cdiv = [](int a, int b) { return (a + b - 1) / b; };
  static const int forced = [] { const char* e = getenv("L2S_FORCE_TILE"); return e ? atoi(e) : 0; }();  // tuning aid
  if (forced) return forced;
  if (N <= 16) return 128016;
  if (N <= 32) return 128032;
  static const TileCfg cands[] = {{256, 128, 1.0f}, {256, 64, 0.8f}, {128, 128, 0.8f}, {128, 64, 0.62f}, {64, 64, 0.4f}};
  int best = 0;
  float best_cost = 1e30f;
  for (const TileCfg& c : cands) {
    if (c.bn == 128 && N < 128) continue;
    const long nt = (long)cdiv(M, c.bm) * cdiv(N, c.bn) * G;
    const float cost = (float)cdiv((int)nt, 256) * (float)(c.bm * c.bn) / c.eff;
    if (cost < best_cost) { best_cost = cost; best = c.bm * 1000 + c.bn; }
  }
  return best;
}

template <typename ET, int BM, int BN, int WM_, int WN_, int MODE, int STAGES>
int launch_tile(const l2s_gemm_desc& d, hipStream_t st) {
  constexpr int SMEM = STAGES * (BM + BN) * BK * 2;
  constexpr int BPC_LDS = (160 * 1024) / SMEM;                         // blocks per CU the LDS admits
  constexpr int BPC = BPC_LDS < (32 / (WM_ * WN_)) ? BPC_LDS : (32 / (WM_ * WN_));
  auto kern = tapgemm_kernel<ET, BM, BN, WM_, WN_, MODE, STAGES>;
  static bool attr_set = false;  // >64 KiB of dynamic LDS needs the opt-in once per instantiation
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
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
  hipLaunchKernelGGL(kern, dim3(8 * slots), dim3(WM_ * WN_ * 64), SMEM, st, d, tilesM, tilesN, chunk, band);
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

template <typename ET, int MODE>
int launch_mode(const l2s_gemm_desc& d, hipStream_t st) {
  switch (pick_tile(d.M, d.N, d.groups > 0 ? d.groups : 1)) {
    case 128016: return launch_tile<ET, 128, 16, 4, 1, MODE, 2>(d, st);
    case 128032: return launch_tile<ET, 128, 32, 4, 1, MODE, 3>(d, st);
    case 256128: return launch_tile<ET, 256, 128, 4, 2, MODE, 3>(d, st);
    case 256064: return launch_tile<ET, 256, 64, 4, 2, MODE, 3>(d, st);
    case 256065: return launch_tile<ET, 256, 64, 4, 1, MODE, 3>(d, st);  // experiment: 4 waves of 64x64
    case 128128: return launch_tile<ET, 128, 128, 2, 2, MODE, 2>(d, st);
    case 128064: return launch_tile<ET, 128, 64, 2, 2, MODE, 3>(d, st);
    default: return launch_tile<ET, 64, 64, 2, 2, MODE, 3>(d, st);
  }
}

template <typename ET>
int launch_dtype(const l2s_gemm_desc& d, hipStream_t st) {
  switch (d.mode) {
    case L2S_MODE_LINEAR: return launch_mode<ET, L2S_MODE_LINEAR>(d, st);
    case L2S_MODE_CONV1D: return launch_mode<ET, L2S_MODE_CONV1D>(d, st);
    case L2S_MODE_CONV2D: return launch_mode<ET, L2S_MODE_CONV2D>(d, st);
    default: return L2S_EINVAL;
  }
}

}  // namespace

extern "C" int l2s_tapgemm(const l2s_gemm_desc* hd, void* stream) {
  if (!hd) return L2S_EINVAL;
  l2s_gemm_desc d = *hd;
  if (!d.A || !d.W || !d.C) return L2S_EINVAL;
  if (d.M <= 0 || d.N <= 0 || d.Cin <= 0 || d.ntaps <= 0) return L2S_ESHAPE;
  if (d.groups <= 0) d.groups = 1;
  if (d.out_row_mul <= 0) d.out_row_mul = 1;
  // 16-byte vector loads of A/W chunks, 8/16-byte vector epilogue
  if ((d.Cin & 7) || (d.lda & 7) || (d.N & 3) || (d.ldc & 3) || (d.a_gstride & 7) || (d.c_gstride & 3) ||
      (d.w_gstride & 7))
    return L2S_EALIGN;
  if (((uintptr_t)d.A & 15) || ((uintptr_t)d.W & 15) || ((uintptr_t)d.C & 15)) return L2S_EALIGN;
  if ((d.flags & (L2S_F_RES_PRE | L2S_F_RES_POST)) && (!d.R || (d.ldr & 3))) return L2S_EINVAL;
  if ((d.flags & L2S_F_DUAL) && (!d.C2 || (d.ldc2 & 3))) return L2S_EINVAL;
  if ((d.flags & L2S_F_MASK) && (!d.lens || d.mask_T <= 0 || d.mask_mul <= 0)) return L2S_EINVAL;
  if (d.act == L2S_ACT_PRELU && !d.slope) return L2S_EINVAL;
  if (d.mode == L2S_MODE_CONV1D && (d.T_out <= 0 || d.T_in <= 0)) return L2S_ESHAPE;
  if (d.mode == L2S_MODE_CONV2D && (d.Ho <= 0 || d.Wo <= 0 || d.Hi <= 0 || d.Wi <= 0 || d.KW <= 0)) return L2S_ESHAPE;
  if (d.mode != L2S_MODE_LINEAR && (int64_t)d.Cin * d.ntaps > (1 << 15)) return L2S_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (d.dtype == L2S_F16) return launch_dtype<ElemF16>(d, st);
  if (d.dtype == L2S_BF16) return launch_dtype<ElemBF16>(d, st);
  return L2S_EINVAL;
}

extern "C" int l2s_tapgemm_variant(const l2s_gemm_desc* hd) {
  if (!hd || hd->M <= 0 || hd->N <= 0) return L2S_EINVAL;
  return pick_tile(hd->M, hd->N, hd->groups > 0 ? hd->groups : 1);
}

// Phase-staggered tap-GEMM (Linear / Conv1d / Conv2d with Cin % 64 == 0): 256x256 output tile, 8 waves (2 x 4) of
// 128x64, for layers with N >= 256 whose epilogue is a lean 16-bit family, a 16-bit residual / dual family or the fp32
// residual stream (QKV, FC1(+GELU), out-proj, FC2, conformer FFN / pointwise convs: fairseq q,k,v,out / fc1,fc2,
// espnet positionwise_feed_forward.py:28-30, attention.py:50-53, convolution.py:26-45; ResNet layer3/4 convs,
// resnet.py:43-69; the C = 256 ResBlock convs of the vocoder, hifigan models.py:30-70).
//
// The 256x128 kernel of tapgemm_kernel.h synchronises all 8 waves once per K-tile, so the two waves of a SIMD always
// do the same thing at the same time: both wait on LDS, then both want the MFMA pipe.  Here the K loop is cut into
// PHASES of 16 MFMAs (one 64x32 quadrant of the wave tile x K=64) with two barriers each, and the upper wave row
// (waves 4-7, the second wave of every SIMD) runs ONE BARRIER BEHIND the lower one: while one wave of a SIMD issues its
// 16 MFMAs the other issues its fragment reads and LDS-DMA for the next phase, so the MFMA pipe and the LDS/VMEM pipes
// are busy in alternation by construction (the "8-phase" idiom of cdna_hip_programming.md section 5).
//
// K-tile = 64; its operands are staged as four QUARTER tiles ordered by first use inside the K-tile:
//   QA0  A rows {0-63, 128-191}  (the upper 64 rows of both wave rows)     first read in phase 0
//   QB0  W rows {wc*64 + 0..31}   (the first 32 columns of all four wave columns)     phase 0 (kept for phase 3)
//   QB1  W rows {wc*64 + 32..63}                                           phase 1
//   QA1  A rows {64-127, 192-255}                                          phase 2
// 2 K-tiles x 4 quarters x 16 KB = 128 KB of LDS.  One quarter is staged per phase (2 LDS-DMA instructions per wave),
// six stream elements ahead of the phase counter: every quarter has >= 5 phases to land, is overwritten >= 2 phases
// after its last read (WAR across the stagger), and `s_waitcnt vmcnt(8)` in every phase (4 younger quarters x 2) retires
// exactly what the NEXT phase reads, one phase and one barrier before it is read (RAW).  Quadrant order per K-tile:
// (rows 0-63, cols 0-31) -> (0-63, 32-63) -> (64-127, 32-63) -> (64-127, 0-31): 8+4, 4, 8, 0 fragment reads.
// Persistent blocks (the quarter stream runs across the block's output tiles), XCD-aware banded tile order and
// epilogue_fast16 are shared with tapgemm_kernel.h.
#pragma once
#include "tapgemm_common.h"
#include <cstdlib>

using namespace l2s;

namespace {

constexpr int PBM = 256, PBN = 256, PBK = 64;
constexpr int Q_B = 128 * PBK * 2;                       // one quarter tile: 128 rows x 128 B = 16 KB
// wave-private epilogue scratch behind the ring: epilogue_fast16 (one row group per round, padded rows) for the lean
// families, the swizzled 4 KB fp32 transposition of epilogue_impl for the 16-bit residual / dual families
constexpr bool p_generic(int epi) { return epi == L2S_EPI_G16A || epi == L2S_EPI_G16B; }
constexpr int p_scr_b(int epi) { return p_generic(epi) ? 16 * 64 * 4 : 16 * (4 * 32 + 16); }
constexpr int p_smem(int epi) { return 8 * Q_B + 8 * p_scr_b(epi); }   // 128 KB + 18 KB, or exactly 160 KB

// Diagnostic build (-DL2S_PHASE_STAMPS, tools/phase_stamps.py): waves 0 and 7 of every block accumulate s_memtime deltas of
// [0] K loops, [1] epilogues, [2] the wait at the first barrier after an epilogue (the block's slowest wave), and count tiles.
#ifdef L2S_PHASE_STAMPS
__device__ unsigned long long* g_phase_stamps = nullptr;
#define PHSTAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ph_acc[i] += now_ - ph_last; ph_last = now_; }
#else
#define PHSTAMP(i)
#endif

// MODE_T 3 (K-block table, include/lip2speech_hip.h: l2s_gemm_desc::ktab): LINEAR staging, but the launch is `groups` problems
// over the same A / W, and a problem's K is a LIST of kblk-wide column blocks of A paired with column blocks of W - a convolution
// on a small map with its image as ONE row of A: output position g sums only the taps that fall inside the map (ResNet layer3 / 4:
// 21 / 40 % of a 3x3 convolution's taps on a 6 x 6 / 3 x 3 map are padding).  Per tile the table row (block count, A and W element
// offsets) is loaded into SGPRs at the stream cursor; the K-tile count varies from tile to tile.
constexpr int PG_KTAB = 3;
template <typename ET, int MODE_T, int EPI>
__global__ __launch_bounds__(512) void phasegemm_kernel(const l2s_gemm_desc p, const int tilesM, const int tilesN,
                                                        const int chunk, const int band) {
  constexpr bool KT = MODE_T == PG_KTAB;
  constexpr int MODE = KT ? (int)L2S_MODE_LINEAR : MODE_T;
  constexpr int MI = 8, NI = 4;
  // the 16-bit families store straight from the MFMA layout (tapgemm_common.h: epilogue_direct16, whole lines through the lane
  // exchange): W fragments are read in the paired row order, the W quarters carry the paired swizzle key
#ifdef L2S_NO_PAIRED     // (A/B switch of the diagnostic builds)
  constexpr bool PAIRED = false;
#else
  constexpr bool PAIRED = EPI <= L2S_EPI_G16A;   // (residual + dual + mask, G16B, measured 11 % slower on it: it keeps epilogue_impl)
#endif
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int K = p.Cin * p.ntaps;         // conv modes: tap-major K, a K-tile lies inside one tap (Cin % 64 == 0)
  const int nk = K / PBK;
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(&g_zero16);

  // ---- persistent tile schedule (as tapgemm_kernel.h) --------------------------------------------------------------
  const int tiles_mn = tilesM * tilesN;
  const int ntiles = tiles_mn * (KT ? p.groups : 1);
  const int slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int lo = (blockIdx.x & 7) * chunk;
  const int hi = lo + chunk < ntiles ? lo + chunk : ntiles;
  const int my_n = (lo + slot < hi) ? (hi - lo - slot + slots - 1) / slots : 0;
  if (my_n == 0) return;
  const int total_q = my_n * nk * 4;                     // quarter tiles this block consumes
  auto tile_coords = [&](int i, int& m0, int& n0, int& g) {   // g: the tile's problem (KT), else 0
    int l = lo + slot + i * slots;
    g = 0;
    if constexpr (KT) {
      // M-tile-major: the problems (output positions) of one M-tile run back to back on neighbouring CUs, so the 256 image rows they
      // all read stay in L2 / Infinity Cache (problem-major order re-fetched A from HBM once per problem and tap: HBM-bound)
      const int per_m = p.groups * tilesN;
      const int mt = l / per_m, r = l - mt * per_m;
      g = r / tilesN;
      m0 = mt * PBM;
      n0 = (r - g * tilesN) * PBN;
      return;
    }
    const int bsz = band * tilesN;
    const int bi = l / bsz, idx = l - bi * bsz;
    const int rows = tilesM - bi * band < band ? tilesM - bi * band : band;
    const int tn = idx / rows;
    m0 = (bi * band + idx - tn * rows) * PBM;
    n0 = tn * PBN;
  };

  // ---- staging: quarter = 128 rows; instruction i covers quarter rows 8i..8i+7; wave w issues i = 2w, 2w+1 ----------
  // lane -> (row lane>>3, slot lane&7) fetches chunk slot ^ (row & 7): the XOR swizzle of the 128-byte rows lives in
  // the source address, the LDS image is lane-linear.
  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ (srow & 7);
  const uint16_t* qa_ptr[2][2];    // [rh][instr]: row pointer (LINEAR) / clip or image base (conv modes)
  int qa_t[2][2], qa_x[2][2];      // conv modes: input time (CONV1D) / input row and column (CONV2D) of tap 0
  const uint16_t* qb_ptr[2][2];    // [ch][instr]
  int run_tap = 0, run_c = 0, run_ky = 0, run_kx = 0;   // (tap, channel offset) of the K-tile being staged
  // LINEAR: the eight rows of a staging instruction are one uniform row group, so its address is an SGPR base (tile origin,
  // row group, K-tile) plus ONE per-lane 32-bit offset for A and one of two for W - no per-row pointer registers (16 VGPRs
  // less: the fp32-residual family spilled around its epilogue) and no 64-bit vector adds in the K loop.  The launcher admits
  // M % 8 == 0 only (a row group is inside the matrix or clamped whole); N % 8 == 0 holds for every launch.
  // The bases of the 4 + 4 row groups a wave stages (A: [rh][instr], W: [ch][instr]) are computed once per tile, at the stream
  // cursor (round 4: the products row * pitch used to be recomputed by every staging instruction - ~9 SALU each, 70 per K-tile).
  const char* sa_base[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  const char* sw_base[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  uint32_t a_lane = 0, w_lane[2] = {0, 0};
  if constexpr (MODE == L2S_MODE_LINEAR) {
    a_lane = (uint32_t)(srow * p.lda * 2 + (schunk << 4));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int within = 16 * (wave & 1) + 8 * j + srow;
      const int bchunk = PAIRED ? ((lane & 7) ^ paired_w_key(within)) : schunk;
      w_lane[j] = (uint32_t)(srow * K * 2 + (bchunk << 4));
    }
  }
  // KT: the cursor inside the staged tile's table row: byte offsets of the current K-tile in A and in W, and those of the NEXT block,
  // fetched by scalar loads one block ahead (a register copy of the row indexed by the block counter ends up in scratch: hipcc turns
  // a select chain back into an indexed array)
  // (the table is read through the CONSTANT address space: a uniform load from a plain global pointer compiles to a vector load whose
  // s_waitcnt vmcnt(0) would drain the six quarter tiles in flight at every block change; from address space 4 it is an s_load)
  typedef const int32_t __attribute__((address_space(4)))* ktab_ptr_t;
  constexpr int KTN = L2S_KTAB_MAX;
  const ktab_ptr_t kt_tab = (ktab_ptr_t)(uintptr_t)p.ktab;
  ktab_ptr_t kt_row = kt_tab;
  int nk_stage = nk, kt_blk = 0, kt_win = 0;
  uint32_t kt_cur_a = 0, kt_cur_w = 0, kt_nxt_a = 0, kt_nxt_w = 0;
  const int kpb = KT ? p.Cin / PBK : 1;                  // K-tiles per block (Cin = the block width)
  auto setup_issue = [&](int i) {
    int m0, n0, tg;
    tile_coords(i, m0, n0, tg);
    if constexpr (KT) {
      kt_row = kt_tab + tg * (2 + 2 * KTN);
      nk_stage = kt_row[0] * kpb;
      kt_blk = 0; kt_win = 0;
      kt_cur_a = (uint32_t)kt_row[2] * 2u; kt_cur_w = (uint32_t)kt_row[2 + KTN] * 2u;
      kt_nxt_a = (uint32_t)kt_row[3] * 2u; kt_nxt_w = (uint32_t)kt_row[3 + KTN] * 2u;
    }
#ifdef L2S_ABL_SAMETILE    // (diagnostic) every block streams the operands of tile (0, 0): real data from L2, nothing from HBM
    m0 = 0; n0 = 0;
#endif
    if constexpr (MODE == L2S_MODE_LINEAR) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          int mg = m0 + (wave >> 2) * 128 + h * 64 + 16 * (wave & 3) + 8 * j;
          mg = mg < p.M ? mg : p.M - 8;
          sa_base[h][j] = (const char*)p.A + (int64_t)mg * p.lda * 2;
          int ng = n0 + (wave >> 1) * 64 + h * 32 + 16 * (wave & 1) + 8 * j;
          ng = ng < p.N ? ng : p.N - 8;
          sw_base[h][j] = (const char*)p.W + (int64_t)ng * K * 2;
        }
      return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // quarter row 16w + 8j + srow  ->  A row (w>>2)*128 + h*64 + 16*(w&3) + 8j + srow ; W row (w>>1)*64 + h*32 + 16*(w&1) + 8j + srow
        int m = m0 + (wave >> 2) * 128 + h * 64 + 16 * (wave & 3) + 8 * j + srow;
        m = m < p.M ? m : p.M - 1;             // rows past M / N are clamped (their outputs are never stored)
        qa_t[h][j] = 0; qa_x[h][j] = 0;
        if (MODE == L2S_MODE_LINEAR) {
          qa_ptr[h][j] = (const uint16_t*)p.A + (int64_t)m * p.lda + schunk * 8;
        } else if (MODE == L2S_MODE_CONV1D) {
          const int b = m / p.T_out, t = m - b * p.T_out;
          qa_ptr[h][j] = (const uint16_t*)p.A + (int64_t)b * p.T_in * p.lda + schunk * 8;
          qa_t[h][j] = t * p.stride + p.off;
        } else {
          const int hw = p.Ho * p.Wo;
          const int img = m / hw, rem = m - img * hw;
          const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
          qa_ptr[h][j] = (const uint16_t*)p.A + (int64_t)img * p.Hi * p.Wi * p.lda + schunk * 8;
          qa_t[h][j] = oy * p.stride - p.pad;
          qa_x[h][j] = ox * p.stride - p.pad;
        }
        const int within = 16 * (wave & 1) + 8 * j + srow;          // row inside the wave column's 32-row share of the quarter
        int n = n0 + (wave >> 1) * 64 + h * 32 + within;
        n = n < p.N ? n : p.N - 1;
        const int bchunk = PAIRED ? ((lane & 7) ^ paired_w_key(within)) : schunk;
        qb_ptr[h][j] = (const uint16_t*)p.W + (int64_t)n * K + bchunk * 8;
      }
  };
  // LDS slot of quarter e (0 QA0, 1 QB0, 2 QB1, 3 QA1) of a K-tile with parity b
  auto slot_off = [&](int b, int e) -> uint32_t { return (uint32_t)((b * 4 + e) * Q_B); };
  int s_i = 0, s_kt = 0, s_e = 0, s_par = 0, staged = 0;   // stream cursor: (tile, K-tile, element), K-tile parity
  // `e_tag`: the element being staged; the stream runs six elements ahead of the phase counter, so inside phase Q it is (Q + 2) & 3 -
  // a compile-time constant (the prologue stages elements 0..3, 0, 1).  -1 = read the cursor (only the conv modes' prologue).
  auto stage_one = [&](auto e_tag) {   // exactly 2 LDS-DMA instructions per wave, dummies from the zero page past the end
    constexpr int E = decltype(e_tag)::value;
    const bool live = staged < total_q;
    const int k0 = s_kt * PBK;
    const int se = E >= 0 ? E : s_e;
    uint16_t* dst = lds + (slot_off(s_par, se) >> 1) + wave * 1024;   // 2 instructions x 512 elements per wave
    const bool is_a = (se == 0) || (se == 3);
    const int h = (se == 2 || se == 3) ? 1 : 0;
    if constexpr (MODE == L2S_MODE_LINEAR) {
      // (past the block's last quarter the cursor stays on it: the trailing stagings re-fetch it into a slot nobody reads)
      static_assert(E >= 0, "LINEAR stages with a compile-time element");
      const uint32_t kb = (uint32_t)k0 * 2u;
      const uint32_t kba = KT ? kt_cur_a : kb, kbw = KT ? kt_cur_w : kb;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const char* g = is_a ? sa_base[h][j] + (size_t)kba + (size_t)a_lane : sw_base[h][j] + (size_t)kbw + (size_t)w_lane[j];
#ifdef L2S_ABL_ZEROSRC     // (diagnostic) every DMA reads the zero page: the LDS write side without the HBM / L2 side
        g = (const char*)zero;
#endif
#if defined(L2S_ABL_WARMDMA)   // (diagnostic) the first eight quarters land (both parities hold real data), then no staging
        if (staged < 8) __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(dst + j * 512), 16, 0, 0);
#elif !defined(L2S_ABL_NODMA)
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(dst + j * 512), 16, 0, 0);
#endif
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint16_t* g;
      bool ok = live;
      if (!is_a) {
        g = qb_ptr[h][j] + k0;
      } else if (MODE == L2S_MODE_LINEAR) {
        g = qa_ptr[h][j] + k0;
      } else if (MODE == L2S_MODE_CONV1D) {
        const int st = qa_t[h][j] + run_tap * p.dil;
        ok = ok && ((unsigned)st < (unsigned)p.T_in);
        g = qa_ptr[h][j] + (st * p.lda + run_c);
      } else {
        const int iy = qa_t[h][j] + run_ky, ix = qa_x[h][j] + run_kx;
        ok = ok && ((unsigned)iy < (unsigned)p.Hi) && ((unsigned)ix < (unsigned)p.Wi);
        g = qa_ptr[h][j] + ((iy * p.Wi + ix) * p.lda + run_c);
      }
      g = ok ? g : zero;   // conv padding and the dummies past the block's end read the zero page
#ifdef L2S_ABL_ZEROSRC     // (diagnostic) every DMA reads the zero page: the LDS write side without the HBM / L2 side
      g = zero;
#endif
#ifndef L2S_ABL_NODMA      // (diagnostic) no staging at all: results are garbage, timing only
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(dst + j * 512), 16, 0, 0);
#endif
    }
  };
  auto stage_advance = [&]() {
    const bool live = staged < total_q;
    ++staged;
    if constexpr (MODE == L2S_MODE_LINEAR) {
      // the LDS slot sequence (element, parity) keeps running; the SOURCE cursor stops on the block's last K-tile
      if (++s_e == 4) {
        s_e = 0;
        s_par ^= 1;
        if constexpr (KT) {
          // (the tiles differ in K: the cursor stops on the last K-tile of the block's last tile)
          if (!(s_i + 1 == my_n && s_kt + 1 == nk_stage)) {
            if (++s_kt == nk_stage) {
              s_kt = 0;
              setup_issue(++s_i);
            } else if (++kt_win == kpb) {
              kt_win = 0;
              ++kt_blk;
              kt_cur_a = kt_nxt_a; kt_cur_w = kt_nxt_w;
              const int nb = kt_blk + 1 < KTN ? kt_blk + 1 : KTN - 1;
              kt_nxt_a = (uint32_t)kt_row[2 + nb] * 2u; kt_nxt_w = (uint32_t)kt_row[2 + KTN + nb] * 2u;
            } else {
              kt_cur_a += PBK * 2; kt_cur_w += PBK * 2;
            }
          }
        } else if (staged < total_q && ++s_kt == nk) {
          s_kt = 0;
          setup_issue(++s_i);
        }
      }
      return;
    }
    if (++s_e == 4) {   // next K-tile
      s_e = 0;
      s_par ^= 1;
      if (MODE != L2S_MODE_LINEAR) {
        run_c += PBK;
        if (run_c >= p.Cin) {
          run_c = 0;
          ++run_tap;
          if (++run_kx == p.KW) { run_kx = 0; ++run_ky; }
        }
      }
      if (live && ++s_kt == nk) {
        s_kt = 0;
        run_tap = 0; run_c = 0; run_ky = 0; run_kx = 0;
        if (++s_i < my_n) setup_issue(s_i);
      }
    }
  };

  // ---- fragments ------------------------------------------------------------------------------------------------------
  const int lm = lane & 15, lg = lane >> 4;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)lds;
  const uint32_t k0_off = (uint32_t)(lm * 128 + (((0 + lg) ^ (lm & 7)) << 4));
  const uint32_t k1_off = (uint32_t)(lm * 128 + (((4 + lg) ^ (lm & 7)) << 4));
  const uint32_t a_row = (uint32_t)(wr * 64) * 128;      // this wave's 64 rows inside a QA quarter
  const uint32_t b_row = (uint32_t)(wc * 32) * 128;      // this wave's 32 rows inside a QB quarter
  frag16 fa[4][2], fb[2][2][2];                          // fa[i][ks]; fb[ch][j][ks] (both column halves stay resident)
  auto read_a = [&](uint32_t qbase) {
    const uint32_t a0 = qbase + a_row + k0_off, a1 = qbase + a_row + k1_off;
    lds_read_b128<0>(fa[0][0], a0); lds_read_b128<2048>(fa[1][0], a0); lds_read_b128<4096>(fa[2][0], a0); lds_read_b128<6144>(fa[3][0], a0);
    lds_read_b128<0>(fa[0][1], a1); lds_read_b128<2048>(fa[1][1], a1); lds_read_b128<4096>(fa[2][1], a1); lds_read_b128<6144>(fa[3][1], a1);
  };
  // plain order: block s of the quarter = rows 16 s + lm of the wave's 32; paired order: rows 8 (lm >> 2) + 4 s + (lm & 3)
  uint32_t bk_off[2][2];      // [s][ks]
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      bk_off[s2][ks] = PAIRED ? paired_w_off(lm, s2, 4 * ks + lg) : (uint32_t)(s2 * 2048) + (ks ? k1_off : k0_off);
  auto read_b = [&](frag16(&f)[2][2], uint32_t qbase) {
    const uint32_t b = qbase + b_row;
    lds_read_b128<0>(f[0][0], b + bk_off[0][0]); lds_read_b128<0>(f[1][0], b + bk_off[1][0]);
    lds_read_b128<0>(f[0][1], b + bk_off[0][1]); lds_read_b128<0>(f[1][1], b + bk_off[1][1]);
  };

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: stream elements 0..5, elements 0 and 1 landed (phase 0 reads them), then the stagger ------------------
  setup_issue(0);
  stage_one(std::integral_constant<int, 0>{}); stage_advance();
  stage_one(std::integral_constant<int, 1>{}); stage_advance();
  stage_one(std::integral_constant<int, 2>{}); stage_advance();
  stage_one(std::integral_constant<int, 3>{}); stage_advance();
  stage_one(std::integral_constant<int, 0>{}); stage_advance();
  stage_one(std::integral_constant<int, 1>{}); stage_advance();
  wait_vmcnt<8>();
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();             // the upper wave row runs one barrier behind from here on

  // one phase: Q = 0..3 (compile time), par = parity of the K-tile being computed
  auto phase = [&](auto q_tag, const int par) {
    constexpr int Q = decltype(q_tag)::value;
#ifndef L2S_ABL_NOREAD       // (diagnostic) no fragment reads
    if (Q == 0) { read_b(fb[0], lds_base + slot_off(par, 1)); __builtin_amdgcn_sched_barrier(0); read_a(lds_base + slot_off(par, 0)); }
    if (Q == 1) read_b(fb[1], lds_base + slot_off(par, 2));
    if (Q == 2) read_a(lds_base + slot_off(par, 3));
#endif
    stage_one(std::integral_constant<int, (Q + 2) & 3>{});
    __builtin_amdgcn_sched_barrier(0);
    wait_vmcnt<8>();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    lds_wait();
    __builtin_amdgcn_s_setprio(1);
    constexpr int RH = (Q >= 2) ? 1 : 0, CH = (Q == 1 || Q == 2) ? 1 : 0;
#ifdef L2S_ABL_MFMA32         // (diagnostic, TIMING ONLY: operands are not in the 32x32x16 layout) same FLOPs, registers and LDS
    {                         // traffic issued as 8 x v_mfma_f32_32x32x16_f16 per phase instead of 16 x 16x16x32
      typedef __attribute__((ext_vector_type(16))) float f32x16_t;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        f32x16_t c;
#pragma unroll
        for (int e = 0; e < 16; ++e) c[e] = acc[RH * 4 + mb * 2 + (e >> 3)][CH * 2 + ((e >> 2) & 1)][e & 3];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[CH][kk >> 1][kk & 1].h, fa[mb * 2 + (kk >> 1)][kk & 1].h, c, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[RH * 4 + mb * 2 + (e >> 3)][CH * 2 + ((e >> 2) & 1)][e & 3] = c[e];
      }
    }
#elif !defined(L2S_ABL_NOMFMA)       // (diagnostic) no matrix instructions
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[RH * 4 + i][CH * 2 + j] = ET::mfma(fb[CH][j][0], fa[i][0], acc[RH * 4 + i][CH * 2 + j]);
        acc[RH * 4 + i][CH * 2 + j] = ET::mfma(fb[CH][j][1], fa[i][1], acc[RH * 4 + i][CH * 2 + j]);
      }
#endif
    __builtin_amdgcn_s_setprio(0);
    // Nothing may sit between the last MFMA and the barrier: the partner wave of this SIMD starts its MFMAs behind it.  Without the
    // fence AFTER the barrier hipcc hoists the next phase's address arithmetic above it (round 4, read off the ISA: ~20 SALU / VALU
    // instructions per phase on the MFMA path; csrc/respair256.hip measured 800 -> 610 cycles per phase from this and leaner staging)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    stage_advance();
  };

  int par = 0;
#ifdef L2S_PHASE_STAMPS
  unsigned long long ph_acc[4] = {0, 0, 0, 0};
  unsigned long long ph_last = __builtin_amdgcn_s_memtime();
  const unsigned long long ph_t0 = ph_last;
#endif
  // KT: the K-tile count of the tile being computed (read one tile ahead: a scalar load with a whole K loop to land)
  auto nk_of = [&](int ti) -> int {
    if (ti >= my_n) return 0;
    const int l = lo + slot + ti * slots;
    const int per_m = p.groups * tilesN;
    return kt_tab[((l % per_m) / tilesN) * (2 + 2 * KTN)] * kpb;
  };
  int nk_next = KT ? nk_of(0) : nk;
  for (int ti = 0; ti < my_n; ++ti) {
    const int nk_c = nk_next;
    if constexpr (KT) nk_next = nk_of(ti + 1);
    for (int kt = 0; kt < nk_c; ++kt) {
#ifdef L2S_PHASE_STAMPS
      if (kt == 1) PHSTAMP(2)     // the first K-tile after an epilogue: includes waiting for the block's slowest wave
#endif
      phase(std::integral_constant<int, 0>{}, par);
      phase(std::integral_constant<int, 1>{}, par);
      phase(std::integral_constant<int, 2>{}, par);
      phase(std::integral_constant<int, 3>{}, par);
      par ^= 1;
    }
    PHSTAMP(0)
    int m0, n0, grp;                                       // grp: column block / bias row of this problem (the epilogues' group argument)
    tile_coords(ti, m0, n0, grp);
    const uint32_t scr = lds_base + 8 * Q_B + (uint32_t)wave * p_scr_b(EPI);   // wave-private, behind the quarter slots
    auto rowmap = [&](int m) -> int64_t { return m < p.M ? (int64_t)m * p.out_row_mul + p.out_row_add : (int64_t)-1; };
    if constexpr (EPI == L2S_EPI_X32) {
      epilogue_stream32x<ET, MI, NI>(p, acc, lane, m0 + wr * 128, n0 + wc * 64, rowmap);     // no scratch
    } else if constexpr (EPI == L2S_EPI_S32) {
      epilogue_direct32<ET, MI, NI>(p, acc, lane, m0 + wr * 128, n0 + wc * 64, grp, rowmap);   // no scratch at all
    } else if constexpr (PAIRED) {
      epilogue_direct16<ET, MI, NI, EPI, decltype(rowmap), NoHook, true>(p, acc, lane, m0 + wr * 128, n0 + wc * 64, grp, rowmap);
    } else if constexpr (EPI == L2S_EPI_G16A) {
      epilogue_impl<ET, MI, NI, F_G16A, true, true>(p, acc, scr, lane, m0 + wr * 128, n0 + wc * 64, grp, rowmap);
    } else if constexpr (EPI == L2S_EPI_G16B) {
      epilogue_impl<ET, MI, NI, F_G16B, true, true>(p, acc, scr, lane, m0 + wr * 128, n0 + wc * 64, grp, rowmap);
    } else {
      epilogue_fast16<ET, MI, NI, (EPI - L2S_EPI_F16) / 2, ((EPI - L2S_EPI_F16) & 1) != 0, 1>(
          p, acc, scr, lane, m0 + wr * 128, n0 + wc * 64, grp, rowmap);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    PHSTAMP(1)
  }
  wait_vmcnt<0>();   // no LDS-DMA (the trailing dummies) may outlive the block's LDS allocation
#ifdef L2S_PHASE_STAMPS
  if (lane == 0 && (wave == 0 || wave == 7) && g_phase_stamps) {
    unsigned long long* o = g_phase_stamps + ((int64_t)blockIdx.x * 2 + (wave ? 1 : 0)) * 8;
    for (int i = 0; i < 3; ++i) o[i] = ph_acc[i];
    o[3] = (unsigned long long)my_n;
    o[4] = __builtin_amdgcn_s_memtime() - ph_t0;
    o[5] = (unsigned long long)nk;
  }
#endif
}

template <typename ET, int MODE, int EPI>
int launch_phase(const l2s_gemm_desc& d, hipStream_t st) {
  auto kern = phasegemm_kernel<ET, MODE, EPI>;
  static L2sSmemOptIn opt_in;  // > 64 KB of dynamic LDS: opt-in per instantiation and device
  if (int e = l2s_smem_opt_in(kern, p_smem(EPI), opt_in)) return e;
  const int tilesM = (d.M + PBM - 1) / PBM, tilesN = (d.N + PBN - 1) / PBN;
  const int ntiles = tilesM * tilesN * (MODE == PG_KTAB ? d.groups : 1);
  const int chunk = (ntiles + 7) / 8;
  const int slots = chunk < 32 ? chunk : 32;
  const double ap = (double)PBM * d.Cin * 2.0, wp = (double)PBN * d.Cin * d.ntaps * 2.0;
  auto cdivi = [](int a, int b) { return (a + b - 1) / b; };
  int band = 1;
  double best = 1e300;
  for (int b = 1; b <= tilesM; ++b) {
    const int wn = cdivi(chunk, b) < tilesN ? cdivi(chunk, b) : tilesN;
    const int an = b * cdivi(chunk, b * tilesN);
    const double fp = ap * (an < tilesM ? an : tilesM) + wp * wn;
    if (fp < best) { best = fp; band = b; }
  }
  hipLaunchKernelGGL(kern, dim3(8 * slots), dim3(512), p_smem(EPI), st, d, tilesM, tilesN, chunk, band);
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}


#ifdef L2S_PHASE_STAMPS
int phase_set_stamps(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_stamps), &buf, sizeof(buf)); }
#endif

template <typename ET, int MODE>
int launch_phase_mode(const l2s_gemm_desc& d, hipStream_t st) {
  if constexpr (MODE == PG_KTAB) {   // bias + linear-family activation -> 16 bit, without / with a 16-bit residual in front of it
    switch (pick_epilogue(d.flags, d.act)) {
      case 2: return launch_phase<ET, MODE, 2>(d, st);
      case L2S_EPI_G16A: return launch_phase<ET, MODE, L2S_EPI_G16A>(d, st);
      default: return L2S_EUNSUPPORTED;
    }
  } else {
    switch (pick_epilogue(d.flags, d.act)) {
      case 0: return launch_phase<ET, MODE, 0>(d, st);
      case 1: return launch_phase<ET, MODE, 1>(d, st);
      case 2: return launch_phase<ET, MODE, 2>(d, st);
      case 3: return launch_phase<ET, MODE, 3>(d, st);
      case 4: return launch_phase<ET, MODE, 4>(d, st);
      case 5: return launch_phase<ET, MODE, 5>(d, st);
      case L2S_EPI_G16A: return launch_phase<ET, MODE, L2S_EPI_G16A>(d, st);
      case L2S_EPI_G16B: return launch_phase<ET, MODE, L2S_EPI_G16B>(d, st);
      case L2S_EPI_S32: return launch_phase<ET, MODE, L2S_EPI_S32>(d, st);
      case L2S_EPI_ALL: return is_x32(d.flags, d.act) ? launch_phase<ET, MODE, L2S_EPI_X32>(d, st) : (int)L2S_EUNSUPPORTED;
      default: return L2S_EUNSUPPORTED;
    }
  }
}

}  // namespace

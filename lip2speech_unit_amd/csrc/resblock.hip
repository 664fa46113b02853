// Fused HiFi-GAN ResBlock1 for the narrow vocoder stages (C = 16 / 32 channels at 32 000 / 64 000 samples per clip).
//   speech-resynthesis/models.py:34-41: for (c1,c2,d) in zip(convs1, convs2, (1,3,5)):
//       xt = c2(leaky_relu(c1(leaky_relu(x)), 0.1)) ; x = xt + x
//   and :103-109: xs = sum over the three ResBlocks (k = 3, 7, 11) of one stage.
// As six separate tap-GEMMs these stages are HBM-bound: every conv re-reads and re-writes a 65 MB activation (~19 passes
// per stage).  Here one block keeps a time tile (+ halo) of the activation in LDS, runs all six convolutions of a
// ResBlock on MFMA straight out of LDS and touches HBM twice: one read of leaky_relu(x), one accumulate into the fp32 xs.
// l2s_resstage_fused runs the three ResBlocks of a stage back to back in one block: the tile is read from HBM once (the
// second and third read hit L2), the running sum xs is re-read from L2 by the block that wrote it, and the two launch
// boundaries with their serial load / compute / store phases disappear.
//
// LDS holds XL = leaky_relu(x) (the conv input) and T1 = leaky_relu(c1(.)); the residual x is recovered from XL by the
// exact inverse of leaky_relu (x = xl >= 0 ? xl : xl / slope = min(xl, xl / slope) for 0 < slope <= 1), so no second
// copy of x is kept.  Rows outside the clip are zero in both buffers (the reference's zero padding); rows whose
// receptive field leaves the tile hold finite garbage that never reaches the tile's own output rows (halo = sum of the
// six receptive half-widths).
//
// Inner loop.  With N = C <= 32 an MFMA 16x16x32 consumes a fresh 1 KB activation fragment per one (C = 16) or two
// (C = 32) instructions and the epilogue (LeakyReLU, inverse LeakyReLU of the residual, 16-bit pack) is 25-50 VALU
// instructions per 16 rows - more issue time than the MFMAs themselves for k = 3.  A wave therefore runs a three-stage
// software pipeline over its 16-row groups: the fragment reads of group i+1 and the residual reads of group i are issued
// first, then the MFMAs of group i (their fragments were requested one iteration earlier), then the epilogue of group
// i-1 - three independent instruction streams in one basic block for the LDS, matrix and vector pipes.
#include <cstdlib>
#include <type_traits>

#include "l2s_common.h"

namespace {

#ifdef RB_STAMPS
__device__ unsigned long long* g_rb_stamps = nullptr;     // diagnostic build: per-(block, wave) cycle counts of the phases
#define RB_T() __builtin_amdgcn_s_memtime()
#endif
constexpr int RB_GUARD = 32;  // zero guard rows on each side: the widest single tap offset is 5*5 = 25 rows

// Tile configuration of the one-ResBlock kernel.  RS = row stride (elements): dense 32-byte rows make the C=16 fragment
// reads conflict-free; C=32, k = 3 / 7 use dense 64-byte rows (2-way conflicts, measured irrelevant) and a 384 / 368-sample
// tile so that two blocks fit the 160 KB of LDS; for k = 11 the halo (2 x 60 rows) makes the small tile a loss, it keeps
// 512 samples, padded 96-byte rows (conflict-free) and one block per CU.  NW = waves per block, BPC = blocks per CU the
// LDS allows; the register allocation is bounded by BPC*NW/4 waves per SIMD (the pipeline keeps two fragment sets and
// the whole conv's weights in registers: 2 x 44 + 88 VGPRs at C = 32, k = 11).
template <int C, int K> struct RBCfg {
  static constexpr bool STAGE = false;
  static constexpr int RS = C == 16 ? 16 : (K == 11 ? 48 : 32);
  static constexpr int TT = (C == 32 && K != 11) ? (K == 7 ? 368 : 384) : 512;
  static constexpr int NW = (C == 32 && K == 11) ? 8 : 4;
  static constexpr int BPC = C == 16 ? 3 : (K == 11 ? 1 : 2);
  static constexpr int WPE = (BPC * NW + 3) / 4;
  static constexpr int TLB = 8;                  // 16-byte chunks of the input tile per thread and batch
  template <int KK> static constexpr bool wb2() { return !(C == 32 && KK != 3) && !(C == 16 && KK == 11); }
};
// The stage kernel runs the three ResBlocks in one block, so they share the k = 11 geometry, and it keeps the fp32 running sum
// of the three ResBlocks IN REGISTERS at C = 32: a wave's share of the tile's own rows is TT / (16 NW) 16-row groups x C / 16 accumulator
// quads = 32 VGPRs per lane for both widths.  Between the ResBlocks the sum costs no HBM traffic and no LDS (an fp32 tile in
// LDS was built first and lost: it halves the blocks per CU at C = 16 and forces dense conflicted rows and shorter tiles at
// C = 32 - profiles/r04_resstage_sum_ab.log).  For that a wave must own the SAME output rows in the last conv of all three
// ResBlocks: the tile's own rows start `lead` = halo rounded up to 32 rows behind buffer row 0 (a group boundary whatever the
// ResBlock), the last conv computes exactly the own groups, statically unrolled, pair p of 16-row groups on wave p mod NW.
// The sum is live through the bodies of the second and third ResBlock, so the ResBlock whose body has no 32 registers to spare
// runs first: 11 -> 7 -> 3 at C = 32 (k = 11: 249 of 256).  C = 16 (168 registers at three waves per SIMD) keeps the sum in the
// global buffer, k = 3 -> 7 -> 11 (see the kernel).
#ifndef RS_RSUM          // (A/B switch of the variant builds: 0 = the running sum in the global fp32 buffer, rounds 2-3)
#define RS_RSUM 1
#endif
template <int C, int K> struct RSCfg {
  static constexpr bool STAGE = true;
  static constexpr int RS = C == 16 ? 16 : 48;
  static constexpr int TT = 512;
  static constexpr int NW = C == 32 ? 8 : 4;
  static constexpr int BPC = C == 16 ? 3 : 1;
  static constexpr int WPE = (BPC * NW + 3) / 4;
  static constexpr int TLB = C == 16 ? 4 : 6;    // covers the whole tile for every legal dilation set (RB <= 768 rows)
  template <int KK> static constexpr bool wb2() { return KK != 11; }
};
// 16-row groups per pipeline stage: two at C = 16 (NI = 1) so that two MFMA chains are in flight; k = 11 keeps one (two
// fragment sets of 2 x 6 x 4 VGPRs do not fit 3 waves per SIMD) and splits its k-steps into two chains instead.
template <int C, int K> constexpr int rb_gp() { return (C == 16 && K != 11) ? 2 : 1; }
template <int C, int K> constexpr int rb_nch() { return (C == 16 && K == 11) ? 2 : 1; }
constexpr int rb_kpad(int C, int K) { return ((K * C + 31) / 32) * 32; }
constexpr int rb_cprp(int C, int K) { return ((rb_kpad(C, K) / 8 + 1) / 4) * 4 + 2; }   // >= KPAD/8, = 2 (mod 4)
// rows in front of the tile's own rows: the halo; in the stage kernel rounded up to whole 32-row pairs (see RSCfg)
template <typename Cfg> __host__ __device__ constexpr int rb_lead(int H) { return Cfg::STAGE ? ((H + 31) / 32) * 32 : H; }
// rows of a ResBlock's LDS buffers (lead + tile + halo, whole pipeline stages), and the dynamic LDS it needs: two activation
// buffers + one conv's weights
template <int C, int K, typename Cfg> __host__ __device__ constexpr int rb_rows(int TT, int H) {
  return ((TT + rb_lead<Cfg>(H) + H + 16 * rb_gp<C, K>() - 1) / (16 * rb_gp<C, K>())) * (16 * rb_gp<C, K>());
}
template <int C, int K, typename Cfg> int rb_smem(int TT, int H) {
  return 2 * (rb_rows<C, K, Cfg>(TT, H) + 2 * RB_GUARD) * Cfg::RS * 2 + (Cfg::template wb2<K>() ? 2 : 1) * C * rb_cprp(C, K) * 16;
}

struct RbArgs {
  const uint16_t* xl_in;   // [B*T, C] 16-bit leaky_relu(x)
  float* xs;               // [B*T, C] fp32 running sum over the stage's ResBlocks
  uint16_t* xl_out;        // [B*T, C] 16-bit leaky_relu(xs) or NULL
  const int32_t* lens;
  int len_mul, T;
  float slope;
  int xs_final;            // resstage: 0 = the caller only reads xl_out; the last ResBlock then leaves xs unwritten
};

// Batch `base` of the input tile of ResBlock K (halo H = HALF * (d0 + d1 + d2 + 3) rows on each side) -> registers.  All of
// a batch's global loads are in flight before the first LDS store (one load -> store per trip costs a full memory latency
// per trip: 10-12 k cycles per tile, a quarter of the kernel).
template <int C, int K, typename Cfg>
__device__ __forceinline__ void rb_tile_fetch(uint4 (&tv)[Cfg::TLB], const RbArgs& a, const int b, const int t0, const int TT,
                                              const int d0, const int d1, const int d2, const int lim, const int base) {
  constexpr int NT = Cfg::NW * 64;
  const int H = ((K - 1) / 2) * (d0 + d1 + d2 + 3);
  const int RB = rb_rows<C, K, Cfg>(TT, H) + 2 * RB_GUARD;
  const int g0 = t0 - rb_lead<Cfg>(H) - RB_GUARD;  // global time of buffer row 0
  const int tl_total = RB * (C / 8);
  const int tid = threadIdx.x;
#pragma unroll
  for (int u = 0; u < Cfg::TLB; ++u) {
    const int i = base + u * NT + tid;
    const int rb = i / (C / 8), ch = i - rb * (C / 8);
    const int t = g0 + rb;
    tv[u] = make_uint4(0, 0, 0, 0);
    if (i < tl_total && t >= 0 && t < lim) tv[u] = *reinterpret_cast<const uint4*>(a.xl_in + ((int64_t)b * a.T + t) * C + ch * 8);
  }
}

// One ResBlock on the time tile [t0, t0 + TT) of clip b.  The whole block calls it; it starts by overwriting both LDS
// buffers (callers running several ResBlocks in a row need no barrier in between: the last conv ends with one) and
// ends with the tile's rows of xs (and xl_out) written.
// `pre`: the first batch of this ResBlock's input tile, fetched by the caller (NULL: fetched here); `next_fetch()` is
// called before the last conv, so that a following ResBlock's tile travels under it.
// SP: where the running sum over a stage's ResBlocks lives.  -1: in the global fp32 buffer xs (read when `accumulate`,
// written when `write_xs`) - the one-ResBlock kernel.  0 / 1 / 2: first / middle / last ResBlock of the stage kernel, the
// sum in the registers S (first: written; middle: added to; last: added, the total goes to xs / xl_out).  The additions
// are the global form's, in the same order: (first + middle) + last in fp32.
template <int C, typename Cfg> constexpr int rb_ns() { return Cfg::STAGE ? Cfg::TT / (16 * Cfg::NW) : 1; }
template <typename ET, int C, int K, typename Cfg, int SP, typename NextFetch>
__device__ __forceinline__ void rb_body(uint16_t* sm, f32x4_t (&S)[rb_ns<C, Cfg>()][C / 16], const RbArgs& a,
                                        const uint16_t* __restrict__ w,
                                        const float* __restrict__ bias, const int b, const int t0, const int TT,
                                        const int d0, const int d1, const int d2, const bool accumulate,
                                        uint16_t* xl_out, uint4 (*pre)[Cfg::TLB], NextFetch next_fetch,
                                        const bool write_xs = true) {
  constexpr int HALF = (K - 1) / 2;
  constexpr int RS = Cfg::RS;                         // row stride in elements
  constexpr int NI = C / 16;
  constexpr int KPAD = rb_kpad(C, K);
  constexpr int STEPS = KPAD / 32;
  constexpr int NW = Cfg::NW, NT = NW * 64, GP = rb_gp<C, K>();
  const int T = a.T;
  const float slope = a.slope;
  float* __restrict__ xs = a.xs;

  const int tid = threadIdx.x, lane = tid & 63;
  const int lm = lane & 15, lg = lane >> 4;
  const int H = HALF * (d0 + d1 + d2 + 3);
  const int HL = rb_lead<Cfg>(H);                    // buffer row of the tile's first own row (t = t0)
  const int R = rb_rows<C, K, Cfg>(TT, H);
  const int RB = R + 2 * RB_GUARD;
  uint16_t* XL = sm;
  uint16_t* T1 = sm + RB * RS;
  // Conv weights reach the waves' registers through LDS: read straight from global, the waves of a block would each pull
  // C*KPAD*2 bytes through the CU's 64 B/clk vector-memory path after every conv's barrier (13-21 % of the kernel);
  // here the block fetches the NEXT conv's C*KPAD*2 bytes once, into registers while the current conv computes, and
  // the waves fill their fragments from LDS.  Weight rows are padded to a stride of 2 (mod 4) 16-byte slots, which makes
  // the 16-lane groups of ds_read_b128 (MI355X_MICROARCH.md, LDS: {0-3,12-15,20-27}, ...) hit 16 different slots.
  // With two weight buffers (WB2: where the LDS allows) the next conv's weights are committed while the current conv
  // still runs and a conv costs one barrier instead of two.
  constexpr int CPR = KPAD / 8;                        // 16-byte chunks per weight row
  constexpr int CPRP = rb_cprp(C, K);                  // ... as stored in LDS
  constexpr bool WB2 = Cfg::template wb2<K>();
  uint16_t* WB = sm + 2 * RB * RS;
  constexpr int WB_ELEMS = C * CPRP * 8;
  constexpr int WCH = C * CPR;                         // chunks per conv
  constexpr int WPT = (WCH + NT - 1) / NT;             // chunks per thread
  uint4 wpre[WPT];
  auto w_fetch = [&](int cv) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int l = tid + i * NT;
      wpre[i] = l < WCH ? *reinterpret_cast<const uint4*>(w + (int64_t)cv * C * KPAD + l * 8) : make_uint4(0, 0, 0, 0);
    }
  };
  auto w_commit = [&](int cv) {                        // weights of conv cv -> its LDS buffer
    uint16_t* wb = WB + ((WB2 && (cv & 1)) ? WB_ELEMS : 0);
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int l = tid + i * NT;
      const int n = l / CPR, c = l - n * CPR;
      if (l < WCH) *reinterpret_cast<uint4*>(wb + (n * CPRP + c) * 8) = wpre[i];
    }
  };
#ifdef RB_STAMPS
  unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long st_t0 = RB_T(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  w_fetch(0);
  // the bias of a conv is fetched one conv ahead (first use: the accumulator init): the barrier below each conv's
  // fragment reads drains vmcnt, so a load issued just before it would put an L2 round trip on every conv's critical path
  f32x4_t bs_next[NI];
  auto bias_fetch = [&](int cv) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const float4 q = *reinterpret_cast<const float4*>(bias + cv * C + ni * 16 + lg * 4);
      bs_next[ni] = f32x4_t{q.x, q.y, q.z, q.w};
    }
  };
  bias_fetch(0);
  int lim = a.lens ? a.lens[b] * a.len_mul : T;
  lim = lim < T ? lim : T;
  const float inv_slope = 1.0f / slope;

  // ---- load XL tile (zero outside the clip), clear T1 guards ----
  constexpr int TLB = Cfg::TLB;
  const int tl_total = RB * (C / 8);
  for (int base = 0; base < tl_total; base += TLB * NT) {
    uint4 tv[TLB];
    if (base == 0 && pre) {
#pragma unroll
      for (int u = 0; u < TLB; ++u) tv[u] = (*pre)[u];
    } else {
      rb_tile_fetch<C, K, Cfg>(tv, a, b, t0, TT, d0, d1, d2, lim, base);
    }
#pragma unroll
    for (int u = 0; u < TLB; ++u) {
      const int i = base + u * NT + tid;
      const int rb = i / (C / 8), ch = i - rb * (C / 8);
      if (i < tl_total) {
        *reinterpret_cast<uint4*>(XL + rb * RS + ch * 8) = tv[u];
        if (rb < RB_GUARD || rb >= RB - RB_GUARD) *reinterpret_cast<uint4*>(T1 + rb * RS + ch * 8) = make_uint4(0, 0, 0, 0);
      }
    }
  }
  w_commit(0);
  __syncthreads();
#ifdef RB_STAMPS
  st_acc[0] = RB_T() - st_t0;
#endif


  // One convolution.  KIND 0: c1 (XL -> T1 = leaky_relu(conv + b)); 1: c2 (T1 -> XL = leaky_relu(conv + b + x)); 2: the
  // last c2 (T1 -> global: xs (+)= conv + b + x, xl_out = leaky_relu of it).
  // A conv computes only the 16-row groups whose outputs a later conv (or the tile's own rows) still reads: the halo a conv has
  // to cover shrinks by every receptive half-width behind it (k = 11, dilations 1/3/5: 55, 50, 35, 30, 5, 0 rows per side
  // instead of 60 six times: -9 % of the ResBlock's MFMAs and epilogues).  `rem`: halo rows still needed on each side of
  // the tile AFTER the conv; its inputs then lie inside the rows the conv before it computed.
  int rem_halo = H;
  auto run_conv = [&](auto kind_tag, const int cv, const int d) {
    constexpr int KIND = decltype(kind_tag)::value;
    rem_halo -= HALF * d;
    const int st_lo = (HL - rem_halo) / (16 * GP), st_hi = (HL + TT + rem_halo + 16 * GP - 1) / (16 * GP);
    const uint16_t* src = KIND == 0 ? XL : T1;
#ifdef RB_STAMPS
    const unsigned long long st_c0 = RB_T();
#endif
    // weights of this conv -> registers (MFMA A operand: lane = output channel lm of N-tile ni, k-chunk lg)
    frag16 wf[NI][STEPS];
    {
      const uint16_t* wb = WB + ((WB2 && (cv & 1)) ? WB_ELEMS : 0);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
          wf[ni][s].u = *reinterpret_cast<const uint4*>(wb + ((ni * 16 + lm) * CPRP + s * 4 + lg) * 8);
    }
    f32x4_t bs[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bs[ni] = bs_next[ni];
    if (!WB2) __syncthreads();                         // every wave holds its fragments: the weight buffer is free
    if (cv < 5) { w_fetch(cv + 1); bias_fetch(cv + 1); }   // in flight during this conv
    // stages are dealt round-robin: wave, wave + NW, ...  (handing them out dynamically from an LDS counter, to even out
    // the oldest-first issue priority between the two waves of a SIMD, measured 5 % slower: atomic + exec-mask overhead)
    int next_stage = st_lo + __builtin_amdgcn_readfirstlane(tid >> 6);
    auto grab = [&]() -> int { const int g = next_stage; next_stage += NW; return g; };
    // per-lane source offset of each k-step: K index = tap*C + c
    int aoff[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      int tap, chunk;
      if (C == 32) { tap = s; chunk = lg; } else { tap = 2 * s + (lg >> 1); chunk = lg & 1; }
      tap = tap < K ? tap : K - 1;  // K padding: weights are zero there, keep the address in range
      aoff[s] = ((tap - HALF) * d) * RS + chunk * 8;
    }

    struct Frags { frag16 fa[GP][STEPS]; };
    struct Epi {
      f32x4_t acc[GP][NI];
      uint2 res[GP][NI];          // KIND >= 1: this lane's 4 channels of XL (the residual), per N-tile
      float4 old[GP][NI];         // KIND 2 with accumulate: xs before this ResBlock
    };
    // stage st covers the 16-row groups st*GP + j, j < GP
    auto row_of = [&](int st, int j) { return (st * GP + j) * 16 + lm; };
    auto load_fa = [&](Frags& F, int st) {
#pragma unroll
      for (int j = 0; j < GP; ++j) {
        const uint16_t* ap = src + (RB_GUARD + row_of(st, j)) * RS;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) F.fa[j][s].u = *reinterpret_cast<const uint4*>(ap + aoff[s]);
      }
    };
    auto load_res = [&](Epi& E, int st) {
      if (KIND == 0) return;
#pragma unroll
      for (int j = 0; j < GP; ++j) {
        const int r = row_of(st, j);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          E.res[j][ni] = *reinterpret_cast<const uint2*>(XL + (RB_GUARD + r) * RS + ni * 16 + lg * 4);
        if (KIND == 2) {
          const int t = t0 - HL + r;
          const bool own = accumulate && r >= HL && r < HL + TT && t < T;
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#if defined(RS_ABL_NOSUM) || defined(RS_ABL_NOLOAD)   // (diagnostic, TIMING ONLY: wrong sums) the global-sum stage kernel without the sum's HBM round trips / loads / stores
            E.old[j][ni] = make_float4(0.f, 0.f, 0.f, 0.f);
#else
            E.old[j][ni] = own ? *reinterpret_cast<const float4*>(xs + ((int64_t)b * T + t) * C + ni * 16 + lg * 4)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
        }
      }
    };
    auto mfmas = [&](const Frags& F, Epi& E) {
      if (rb_nch<C, K>() == 1) {
#pragma unroll
        for (int j = 0; j < GP; ++j)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) E.acc[j][ni] = bs[ni];                     // bias rides in the accumulator
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
#pragma unroll
          for (int j = 0; j < GP; ++j)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) E.acc[j][ni] = ET::mfma(wf[ni][s], F.fa[j][s], E.acc[j][ni]);
      } else {                                                                       // GP = NI = 1: even / odd k-steps
        f32x4_t a0 = bs[0], a1 = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
          if (s & 1) a1 = ET::mfma(wf[0][s], F.fa[0][s], a1);
          else a0 = ET::mfma(wf[0][s], F.fa[0][s], a0);
        }
        E.acc[0][0] = a0 + a1;
      }
    };
    auto lrelu_pack = [&](const float (&v)[4]) {
      float l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e)                  // max(v, v*slope) for 0 < slope <= 1, as one v_med3_f32 (fmaxf would
        l[e] = __builtin_amdgcn_fmed3f(v[e], v[e] * slope, __builtin_inff());   // add a canonicalising v_max per input)
      return make_uint2(ET::pack2(l[0], l[1]), ET::pack2(l[2], l[3]));
    };
    auto epilogue = [&](const Epi& E, int st) {
#pragma unroll
      for (int j = 0; j < GP; ++j) {
        const int r = row_of(st, j);
        const int t = t0 - HL + r;                   // global time of the row
        const bool inclip = (t >= 0) && (t < lim);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int n = ni * 16 + lg * 4;
          float v[4] = {E.acc[j][ni][0], E.acc[j][ni][1], E.acc[j][ni][2], E.acc[j][ni][3]};
          if (KIND == 0) {
            uint2 q = lrelu_pack(v);
            if (!inclip) q = make_uint2(0, 0);
            *reinterpret_cast<uint2*>(T1 + (RB_GUARD + r) * RS + n) = q;
          } else {
            const uint2 q0 = E.res[j][ni];
            const float xr[4] = {ET::to_f32((uint16_t)(q0.x & 0xffff)), ET::to_f32((uint16_t)(q0.x >> 16)),
                                 ET::to_f32((uint16_t)(q0.y & 0xffff)), ET::to_f32((uint16_t)(q0.y >> 16))};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += __builtin_amdgcn_fmed3f(xr[e], xr[e] * inv_slope, -__builtin_inff());   // min(.,.)
            if (KIND == 1) {
              uint2 q = lrelu_pack(v);
              if (!inclip) q = make_uint2(0, 0);
              *reinterpret_cast<uint2*>(XL + (RB_GUARD + r) * RS + n) = q;
            } else if (r >= HL && r < HL + TT && t < T) {
              // ResBlock output for the tile's own rows: accumulate into xs (models.py:105-108)
              if (!inclip) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
              const float4 o = E.old[j][ni];
              v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
#if defined(RS_ABL_NOSUM) || defined(RS_ABL_NOSTORE)
              if (write_xs && xl_out)
#else
              if (write_xs)
#endif
                *reinterpret_cast<float4*>(xs + ((int64_t)b * T + t) * C + n) = make_float4(v[0], v[1], v[2], v[3]);
              if (xl_out) *reinterpret_cast<uint2*>(xl_out + ((int64_t)b * T + t) * C + n) = lrelu_pack(v);
            }
          }
        }
      }
    };

#ifdef RB_STAMPS
    const unsigned long long st_c1 = RB_T();
    st_acc[1] += st_c1 - st_c0;
#endif
    if constexpr (KIND == 2 && SP >= 0) {
      // The stage kernel's last conv: exactly the tile's own rows, one 16-row group per step, fully unrolled so that the
      // running sum S[i] is a register.  Step i of wave w: group 2 (w + NW (i / 2)) + (i & 1) - the same rows in every
      // ResBlock of the stage.  Same pipeline as below (fragments of step i+1 || MFMAs of step i || epilogue of step i-1),
      // same MFMA chains and the same additions as the global form: bit-identical results.
      constexpr int NS = rb_ns<C, Cfg>();
      static_assert(Cfg::TT % (32 * NW) == 0, "own rows: whole pairs of 16-row groups per wave");
      const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
      auto row_i = [&](int i) { return HL + 16 * (2 * (wv + NW * (i >> 1)) + (i & 1)) + lm; };
      constexpr int NB2 = (NI == 1 && rb_nch<C, K>() == 1) ? 4 : 2;     // ring of 2 steps x G2 groups (below)
      frag16 fa2[NB2][STEPS];
      f32x4_t acc2[NB2][NI];
      uint2 res2[NB2][NI];
      auto lfa = [&](frag16 (&f)[STEPS], int i) {
        const uint16_t* ap = src + (RB_GUARD + row_i(i)) * RS;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) f[s].u = *reinterpret_cast<const uint4*>(ap + aoff[s]);
      };
      auto epi2 = [&](int i) {
        const int r = row_i(i);
        const int t = t0 - HL + r;                   // >= t0: an own row
        const bool inclip = t < lim;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int n = ni * 16 + lg * 4;
          float v[4] = {acc2[i & (NB2 - 1)][ni][0], acc2[i & (NB2 - 1)][ni][1], acc2[i & (NB2 - 1)][ni][2], acc2[i & (NB2 - 1)][ni][3]};
          const uint2 q0 = res2[i & (NB2 - 1)][ni];
          const float xr[4] = {ET::to_f32((uint16_t)(q0.x & 0xffff)), ET::to_f32((uint16_t)(q0.x >> 16)),
                               ET::to_f32((uint16_t)(q0.y & 0xffff)), ET::to_f32((uint16_t)(q0.y >> 16))};
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += __builtin_amdgcn_fmed3f(xr[e], xr[e] * inv_slope, -__builtin_inff());
          if (!inclip) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
          if (SP == 0) {
            S[i][ni] = f32x4_t{v[0] + 0.f, v[1] + 0.f, v[2] + 0.f, v[3] + 0.f};      // (+ 0: the global form's -0 -> +0)
          } else if (SP == 1) {
            S[i][ni] = f32x4_t{v[0] + S[i][ni][0], v[1] + S[i][ni][1], v[2] + S[i][ni][2], v[3] + S[i][ni][3]};
          } else if (t < T) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += S[i][ni][e];
            if (write_xs) *reinterpret_cast<float4*>(xs + ((int64_t)b * T + t) * C + n) = make_float4(v[0], v[1], v[2], v[3]);
            if (xl_out) *reinterpret_cast<uint2*>(xl_out + ((int64_t)b * T + t) * C + n) = lrelu_pack(v);
          }
        }
      };
      // G2 16-row groups per step: two where a group is ONE dependent MFMA chain (C = 16, k = 3 / 7), so that two chains are in
      // flight as in the pipelined form (GP = 2)
      constexpr int G2 = (NI == 1 && rb_nch<C, K>() == 1) ? 2 : 1;
      static_assert(NS % G2 == 0, "whole steps");
      auto mm = [&](int i) {
        if (rb_nch<C, K>() == 1) {
#pragma unroll
          for (int g = 0; g < G2; ++g)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc2[(i + g) & (2 * G2 - 1)][ni] = bs[ni];
#pragma unroll
          for (int s = 0; s < STEPS; ++s)
#pragma unroll
            for (int g = 0; g < G2; ++g)
#pragma unroll
              for (int ni = 0; ni < NI; ++ni)
                acc2[(i + g) & (2 * G2 - 1)][ni] = ET::mfma(wf[ni][s], fa2[(i + g) & (2 * G2 - 1)][s], acc2[(i + g) & (2 * G2 - 1)][ni]);
        } else {
          f32x4_t a0 = bs[0], a1 = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < STEPS; ++s) {
            if (s & 1) a1 = ET::mfma(wf[0][s], fa2[i & 1][s], a1);
            else a0 = ET::mfma(wf[0][s], fa2[i & 1][s], a0);
          }
          acc2[i & 1][0] = a0 + a1;
        }
      };
#pragma unroll
      for (int g = 0; g < G2; ++g) lfa(fa2[g], g);
#pragma unroll
      for (int i = 0; i < NS; i += G2) {
#pragma unroll
        for (int g = 0; g < G2; ++g) {
          if (i + G2 + g < NS) lfa(fa2[(i + G2 + g) & (2 * G2 - 1)], i + G2 + g);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            res2[(i + g) & (2 * G2 - 1)][ni] = *reinterpret_cast<const uint2*>(XL + (RB_GUARD + row_i(i + g)) * RS + ni * 16 + lg * 4);
        }
        mm(i);
        if (i > 0) {
#pragma unroll
          for (int g = 0; g < G2; ++g) epi2(i - G2 + g);
        }
        __builtin_amdgcn_sched_barrier(0);   // (unrolled: without it the scheduler hoists every step's fragment reads to the top and spills)
      }
#pragma unroll
      for (int g = 0; g < G2; ++g) epi2(NS - G2 + g);
    } else {
    // Iteration st: issue the fragment reads of stage st+1 and the residual reads of stage st, then the MFMAs of stage
    // st (their fragments were requested one iteration ago), then the epilogue of stage st-1.
    Frags F0, F1;
    Epi E0, E1;
    const int last = st_hi - 1;
    int prev = 0, cur = grab(), nxt = grab();
    if (cur <= last) {
      load_fa(F0, cur);
      load_fa(F1, nxt < last ? nxt : last);
      load_res(E0, cur);
      mfmas(F0, E0);
      prev = cur; cur = nxt; nxt = grab();
      while (true) {
        if (cur > last) { epilogue(E0, prev); break; }
        load_fa(F0, nxt < last ? nxt : last);
        load_res(E1, cur);
        mfmas(F1, E1);
        epilogue(E0, prev);
        prev = cur; cur = nxt; nxt = grab();
        if (cur > last) { epilogue(E1, prev); break; }
        load_fa(F1, nxt < last ? nxt : last);
        load_res(E0, cur);
        mfmas(F0, E0);
        epilogue(E1, prev);
        prev = cur; cur = nxt; nxt = grab();
      }
    }
    }
#ifdef RB_STAMPS
    const unsigned long long st_c2 = RB_T();
    st_acc[KIND == 2 ? 3 : 2] += st_c2 - st_c1;
#endif
    if (cv < 5) w_commit(cv + 1);
    __syncthreads();
#ifdef RB_STAMPS
    st_acc[4] += RB_T() - st_c2;
#endif
  };

  run_conv(std::integral_constant<int, 0>{}, 0, d0);
  run_conv(std::integral_constant<int, 1>{}, 1, 1);
  run_conv(std::integral_constant<int, 0>{}, 2, d1);
  run_conv(std::integral_constant<int, 1>{}, 3, 1);
  run_conv(std::integral_constant<int, 0>{}, 4, d2);
  next_fetch();
  run_conv(std::integral_constant<int, 2>{}, 5, 1);
#ifdef RB_STAMPS
  if (lane == 0 && g_rb_stamps) {
    const int wave = tid >> 6;
    unsigned long long* o = g_rb_stamps + (((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 3 + (K == 3 ? 0 : (K == 7 ? 1 : 2))) * 8;
    for (int i = 0; i < 5; ++i) o[i] = st_acc[i];
    o[5] = RB_T() - st_t0;
    o[6] = __builtin_amdgcn_s_memrealtime() - st_r0;
    o[7] = st_t0;
  }
#endif
}

template <typename ET, int C, int K>
__global__ __launch_bounds__((RBCfg<C, K>::NW * 64), (RBCfg<C, K>::WPE)) void resblock_kernel(
    RbArgs a, const uint16_t* __restrict__ w, const float* __restrict__ bias, int TT, int d0, int d1, int d2,
    int accumulate) {
  extern __shared__ __attribute__((aligned(16))) uint16_t sm[];
  f32x4_t S[1][C / 16];
  rb_body<ET, C, K, RBCfg<C, K>, -1>(sm, S, a, w, bias, blockIdx.y, blockIdx.x * TT, TT, d0, d1, d2, accumulate != 0, a.xl_out,
                                 nullptr, [] {});
}

// The three ResBlocks (k = 3, 7, 11) of one stage on one tile: xs = rb3(x) + rb7(x) + rb11(x), xl_out = leaky_relu(xs).
struct RsW { const uint16_t* w[3]; const float* bias[3]; int d[3][3]; };

template <typename ET, int C>
__global__ __launch_bounds__((RSCfg<C, 11>::NW * 64), (RSCfg<C, 11>::WPE)) void resstage_kernel(RbArgs a, RsW p, int TT) {
  extern __shared__ __attribute__((aligned(16))) uint16_t sm[];
  const int b = blockIdx.y, t0 = blockIdx.x * TT;
  using Cfg = RSCfg<C, 11>;
  int lim = a.lens ? a.lens[b] * a.len_mul : a.T;
  lim = lim < a.T ? lim : a.T;
  constexpr int SP0 = RS_RSUM ? 0 : -1, SP1 = RS_RSUM ? 1 : -1, SP2 = RS_RSUM ? 2 : -1;
  f32x4_t S[rb_ns<C, Cfg>()][C / 16];                 // the stage's running sum over this wave's own rows
  // the next ResBlock's input tile (same rows, its own halo: the re-read hits L2) is fetched into registers under the
  // current one's last conv instead of at the start of the next one, where nothing could overlap it
  uint4 tnext[Cfg::TLB];
  const bool wx = a.xs_final != 0 || a.xl_out == nullptr;
#define RS_D(j) p.d[j][0], p.d[j][1], p.d[j][2]
  if constexpr (C == 32) {
    // 11 -> 7 -> 3; the next ResBlock's tile travels under the current one's last conv at both hand-overs (with the sum filling up
    // under the k = 11 one hipcc still fits the fp16 instantiation into 256 registers; measured 7.89 -> 7.87 ms: within noise, kept)
    constexpr bool PF1 = true;
    rb_body<ET, C, 11, Cfg, SP0>(sm, S, a, p.w[2], p.bias[2], b, t0, TT, RS_D(2), false, nullptr, nullptr, [&] {
      if constexpr (PF1) rb_tile_fetch<C, 7, Cfg>(tnext, a, b, t0, TT, RS_D(1), lim, 0);
    });
    rb_body<ET, C, 7, Cfg, SP1>(sm, S, a, p.w[1], p.bias[1], b, t0, TT, RS_D(1), true, nullptr, PF1 ? &tnext : nullptr,
                                [&] { rb_tile_fetch<C, 3, Cfg>(tnext, a, b, t0, TT, RS_D(0), lim, 0); });
    rb_body<ET, C, 3, Cfg, SP2>(sm, S, a, p.w[0], p.bias[0], b, t0, TT, RS_D(0), true, a.xl_out, &tnext, [] {}, wx);
  } else {
    // C = 16 keeps the sum in the global fp32 buffer, k = 3 -> 7 -> 11 (rounds 2-3).  At three waves per SIMD (168 registers)
    // hipcc carries the sum's 32 registers through the k = 3 body only (7 -> 3: 144 VGPRs) and spills them around the k = 7 and
    // k = 11 bodies in every order tried (63-131 dwords of scratch, 7.07 ms against 6.32); 7 -> 3 on registers, one round trip of
    // the partial sum through the global buffer (prefetched a conv ahead) and then k = 11 halves the sum's traffic and measured
    // 6.26-6.40 ms against 6.16-6.33: not selected (profiles/r04_resstage_sum_ab.log).
    rb_body<ET, C, 3, Cfg, -1>(sm, S, a, p.w[0], p.bias[0], b, t0, TT, RS_D(0), false, nullptr, nullptr,
                               [&] { rb_tile_fetch<C, 7, Cfg>(tnext, a, b, t0, TT, RS_D(1), lim, 0); });
    // (the k = 7 body runs at 147 of the 168 registers: no room for a tile in flight under its last conv)
    rb_body<ET, C, 7, Cfg, -1>(sm, S, a, p.w[1], p.bias[1], b, t0, TT, RS_D(1), true, nullptr, &tnext, [] {});
    rb_body<ET, C, 11, Cfg, -1>(sm, S, a, p.w[2], p.bias[2], b, t0, TT, RS_D(2), true, a.xl_out, nullptr, [] {}, wx);
  }
#undef RS_D
}

template <typename ET, int C, int K>
int launch_rb(const RbArgs& a, const void* w, const float* bias, int B, int d0, int d1, int d2, int accumulate,
              hipStream_t st) {
  using Cfg = RBCfg<C, K>;
  const int H = ((K - 1) / 2) * (d0 + d1 + d2 + 3);
  const int TT = Cfg::TT;
  const int smem = rb_smem<C, K, Cfg>(TT, H);
  auto kern = resblock_kernel<ET, C, K>;
  static L2sSmemOptIn opt_in;
  if (int e = l2s_smem_opt_in(kern, 160 * 1024, opt_in)) return e;
  if (smem > 160 * 1024) return L2S_EUNSUPPORTED;
  dim3 grid((a.T + TT - 1) / TT, B);
  hipLaunchKernelGGL(kern, grid, dim3(Cfg::NW * 64), smem, st, a, (const uint16_t*)w, bias, TT, d0, d1, d2, accumulate);
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

template <typename ET>
int dispatch_rb(int C, int K, const RbArgs& a, const void* w, const float* bias, int B, int d0, int d1, int d2,
                int accumulate, hipStream_t st) {
#define RB_CASE(CC, KK) \
  if (C == CC && K == KK) return launch_rb<ET, CC, KK>(a, w, bias, B, d0, d1, d2, accumulate, st);
  RB_CASE(16, 3) RB_CASE(16, 7) RB_CASE(16, 11) RB_CASE(32, 3) RB_CASE(32, 7) RB_CASE(32, 11)
#undef RB_CASE
  return L2S_EUNSUPPORTED;
}

template <typename ET, int C>
int launch_rs(const RbArgs& a, const RsW& p, int B, hipStream_t st) {
  using Cfg = RSCfg<C, 11>;
  const int TT = Cfg::TT;   // (compile time: the last conv's unrolled own-row loop; round 3's L2S_RS_TT sweep is in DESIGN.md)
  const int h3 = 1 * (p.d[0][0] + p.d[0][1] + p.d[0][2] + 3), h7 = 3 * (p.d[1][0] + p.d[1][1] + p.d[1][2] + 3),
            h11 = 5 * (p.d[2][0] + p.d[2][1] + p.d[2][2] + 3);
  const int s3 = rb_smem<C, 3, Cfg>(TT, h3), s7 = rb_smem<C, 7, Cfg>(TT, h7), s11 = rb_smem<C, 11, Cfg>(TT, h11);
  int smem = s3 > s7 ? s3 : s7;
  smem = smem > s11 ? smem : s11;
  auto kern = resstage_kernel<ET, C>;
  static L2sSmemOptIn opt_in;
  if (int e = l2s_smem_opt_in(kern, 160 * 1024, opt_in)) return e;
  if (smem > 160 * 1024) return L2S_EUNSUPPORTED;
  dim3 grid((a.T + TT - 1) / TT, B);
  hipLaunchKernelGGL(kern, grid, dim3(Cfg::NW * 64), smem, st, a, p, TT);
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

bool rb_dil_ok(int d) { return d >= 1 && d <= 5; }

}  // namespace

#ifdef RB_STAMPS
extern "C" int l2s_debug_rb_stamps(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_rb_stamps), &buf, sizeof(buf)); }
#endif

extern "C" int l2s_resblock_fused(const void* xl, const void* w, const float* bias, float* xs, void* xl_out,
                                  const int32_t* lens, int len_mul, int B, int T, int C, int k, int d0, int d1, int d2,
                                  int accumulate, float slope, int dtype, void* stream) {
  if (!xl || !w || !bias || !xs) return L2S_EINVAL;
  if (B <= 0 || T <= 0) return L2S_ESHAPE;
  // slope in (0, 1]: the kernel's LeakyReLU is max(v, v*slope) and its inverse min(xl, xl/slope)
  if (!rb_dil_ok(d0) || !rb_dil_ok(d1) || !rb_dil_ok(d2) || !(slope > 0.f) || slope > 1.f) return L2S_EUNSUPPORTED;
  if (lens && len_mul <= 0) return L2S_EINVAL;
  if (((uintptr_t)xl & 15) || ((uintptr_t)w & 15) || ((uintptr_t)xs & 15) || ((uintptr_t)xl_out & 7)) return L2S_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  const RbArgs a{(const uint16_t*)xl, xs, (uint16_t*)xl_out, lens, len_mul, T, slope, 1};
  if (dtype == L2S_F16) return dispatch_rb<ElemF16>(C, k, a, w, bias, B, d0, d1, d2, accumulate, st);
  if (dtype == L2S_BF16) return dispatch_rb<ElemBF16>(C, k, a, w, bias, B, d0, d1, d2, accumulate, st);
  return L2S_EINVAL;
}

extern "C" int l2s_resstage_fused(const void* xl, const void* const* w, const float* const* bias, const int* ks,
                                  const int* dils, int n_blocks, float* xs, void* xl_out, const int32_t* lens,
                                  int len_mul, int B, int T, int C, float slope, int xs_final, int dtype, void* stream) {
  if (!xl || !w || !bias || !ks || !dils || !xs) return L2S_EINVAL;
  if (!xs_final && !xl_out) return L2S_EINVAL;      // nothing would be written
  if (B <= 0 || T <= 0) return L2S_ESHAPE;
  // resblock_kernel_sizes = [3, 7, 11] (configs/*/multi_input.json): the one stage layout that is built
  if (n_blocks != 3 || ks[0] != 3 || ks[1] != 7 || ks[2] != 11 || (C != 16 && C != 32)) return L2S_EUNSUPPORTED;
  if (!(slope > 0.f) || slope > 1.f) return L2S_EUNSUPPORTED;
  if (lens && len_mul <= 0) return L2S_EINVAL;
  if (((uintptr_t)xl & 15) || ((uintptr_t)xs & 15) || ((uintptr_t)xl_out & 7)) return L2S_EALIGN;
  RsW p;
  for (int j = 0; j < 3; ++j) {
    if (!w[j] || !bias[j]) return L2S_EINVAL;
    if ((uintptr_t)w[j] & 15) return L2S_EALIGN;
    p.w[j] = (const uint16_t*)w[j];
    p.bias[j] = bias[j];
    for (int m = 0; m < 3; ++m) {
      if (!rb_dil_ok(dils[j * 3 + m])) return L2S_EUNSUPPORTED;
      p.d[j][m] = dils[j * 3 + m];
    }
  }
  hipStream_t st = (hipStream_t)stream;
  const RbArgs a{(const uint16_t*)xl, xs, (uint16_t*)xl_out, lens, len_mul, T, slope, xs_final};
  if (dtype == L2S_F16) return C == 16 ? launch_rs<ElemF16, 16>(a, p, B, st) : launch_rs<ElemF16, 32>(a, p, B, st);
  if (dtype == L2S_BF16) return C == 16 ? launch_rs<ElemBF16, 16>(a, p, B, st) : launch_rs<ElemBF16, 32>(a, p, B, st);
  return L2S_EINVAL;
}

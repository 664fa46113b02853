// Tile choice of the tap-GEMM (shared by the dispatcher and the per-mode instantiations).
#pragma once
#include <cstdlib>

namespace l2s {
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

// Epilogue families.  Every tap-GEMM kernel is compiled once per family so that a launch carries ONE epilogue's code and
// nothing else: measured on MI355X, every piece of epilogue code that a launch does not need (a GELU branch, the row-mask
// path, the other flag families) costs 5-20 % of the whole kernel through instruction-cache misses at each tile end.
enum { L2S_EPI_F16 = 0,        // 0..5: bias / alpha / activation / (row mask) -> 16-bit (epilogue_fast16)
                               //   + 0 no activation, + 2 linear family {relu, prelu, lrelu}, + 4 gelu; + 1 with row mask
       L2S_EPI_G16A = 6,       // 16-bit residual before/after a linear-family activation
       L2S_EPI_G16B = 7,       // ... + second LeakyReLU output + row mask (vocoder ResBlock convs, transposed-conv phases)
       L2S_EPI_S32 = 8,        // fp32 residual stream: fp32 / 16-bit residual, fp32 output, linear-family activation
       L2S_EPI_ALL = 9,        // everything else (accumulate, swish / tanh, mixed combinations)
       L2S_EPI_COUNT = 10,
       // phase-staggered kernel only (not a pick_epilogue() result): the ResBlock-sum update of the un-fused vocoder stage,
       // xs (+)= conv + b + r16 with a row mask and an optional LeakyReLU'd 16-bit copy - a subset of L2S_EPI_ALL
       L2S_EPI_X32 = 10 };
// flags of L2S_EPI_X32: fp32 output, 16-bit residual after the (absent) activation, optional accumulate / dual / mask
inline bool is_x32(int flags, int act) {
  const int need = 0x2 /*RES_POST*/ | 0x20 /*OUT_F32*/, may = 0x4 /*ACCUM*/ | 0x8 /*DUAL*/ | 0x10 /*MASK*/;
  return act == 0 && (flags & need) == need && (flags & ~(need | may)) == 0;
}
inline int pick_epilogue(int flags, int act) {
  const bool lin = act == 0 /*NONE*/ || act == 1 /*RELU*/ || act == 4 /*PRELU*/ || act == 5 /*LRELU*/;
  if ((flags & ~0x10) == 0) {
    const int m = (flags & 0x10) ? 1 : 0;
    if (act == 0) return L2S_EPI_F16 + 0 + m;
    if (lin) return L2S_EPI_F16 + 2 + m;
    if (act == 2 /*GELU*/) return L2S_EPI_F16 + 4 + m;
    return L2S_EPI_ALL;
  }
  if ((flags & ~(0x1 | 0x2)) == 0 && lin) return L2S_EPI_G16A;
  if ((flags & ~(0x1 | 0x2 | 0x8 | 0x10)) == 0 && lin) return L2S_EPI_G16B;
  if ((flags & ~(0x1 | 0x2 | 0x20 | 0x40)) == 0 && (flags & 0x20) && lin) return L2S_EPI_S32;
  return L2S_EPI_ALL;
}

}  // namespace l2s

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

}  // namespace l2s

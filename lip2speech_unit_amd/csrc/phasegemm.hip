// Phase-staggered tap-GEMM, dispatch side: the selection heuristic and the per-(dtype, mode) entry points of
// phasegemm_kernel.h (256x256 output tile, 8 waves of 128x64; built by phasegemm_inst.hip).
#include "l2s_common.h"
#include "tapgemm_tiles.h"
#include <cstdlib>

int l2s_phasegemm_f16_m0(const l2s_gemm_desc& d, hipStream_t st);
int l2s_phasegemm_f16_m1(const l2s_gemm_desc& d, hipStream_t st);
int l2s_phasegemm_f16_m2(const l2s_gemm_desc& d, hipStream_t st);
int l2s_phasegemm_bf16_m0(const l2s_gemm_desc& d, hipStream_t st);
int l2s_phasegemm_bf16_m1(const l2s_gemm_desc& d, hipStream_t st);
int l2s_phasegemm_bf16_m2(const l2s_gemm_desc& d, hipStream_t st);
int l2s_phasegemm_f16_m3(const l2s_gemm_desc& d, hipStream_t st);    // K-block table (l2s_gemm_desc::ktab)
int l2s_phasegemm_bf16_m3(const l2s_gemm_desc& d, hipStream_t st);

// A launch with a K-block table is served by this kernel or not at all (no other kernel reads the table)
int l2s_phasegemm_ktab_launch(const l2s_gemm_desc& d, hipStream_t st) {
  if (d.mode != L2S_MODE_LINEAR || d.groups < 1) return L2S_EINVAL;
  const int fam = l2s::pick_epilogue(d.flags, d.act);
  if (fam != 2 && fam != l2s::L2S_EPI_G16A) return L2S_EUNSUPPORTED;
  if ((d.Cin % 64) || d.N < 256 || d.M < 256 || (d.M & 7) || (d.N & 7) || (d.lda & 7) || (d.ldc & 7) || (d.c_gstride & 63) ||
      (d.ldr & 7) || d.out_row_mul != 1 || d.out_row_add != 0)
    return L2S_EUNSUPPORTED;
  return d.dtype == L2S_F16 ? l2s_phasegemm_f16_m3(d, st) : l2s_phasegemm_bf16_m3(d, st);
}

// Is the phase-staggered 256x256 kernel the better choice for this descriptor?  (called by l2s_tapgemm)
bool l2s_phasegemm_eligible(const l2s_gemm_desc& d) {
  static const int mode = [] { const char* e = getenv("L2S_PHASEGEMM"); return e ? atoi(e) : 1; }();  // 0 off, 1 auto, 2 force
  if (mode == 0) return false;
  if (d.groups > 1 || d.mode < 0 || d.mode > 2 || d.ktab) return false;
  if (d.mode == L2S_MODE_LINEAR && d.ntaps != 1) return false;
  if (d.mode == L2S_MODE_CONV1D && (d.T_out <= 0 || d.T_in <= 0)) return false;
  if (d.mode == L2S_MODE_CONV2D && (d.Ho <= 0 || d.Wo <= 0 || d.KW <= 0 || d.ntaps % d.KW)) return false;
  const int fam = l2s::pick_epilogue(d.flags, d.act);
  // every family but the catch-all: with accumulate + fp32 output + dual on top of the 128 accumulator registers of a
  // 128x64 wave tile the epilogue spills (measured 0.6x of the 256x128 kernel); the 16-bit residual / dual families
  // (swizzled 4 KB fp32 transposition) were measured 4-12 % faster here
  // ... except the ResBlock-sum update (L2S_EPI_X32), which has its own register-lean epilogue (epilogue_stream32x)
  if (fam == l2s::L2S_EPI_ALL && !(l2s::is_x32(d.flags, d.act) && !(d.ldc & 3) && !(d.ldr & 3) && !(d.ldc2 & 3))) return false;
  static const int res_on = [] { const char* e = getenv("L2S_PHASEGEMM_RES"); return e ? atoi(e) : 1; }();  // A/B switch
  if (!res_on && (fam == l2s::L2S_EPI_G16A || fam == l2s::L2S_EPI_G16B)) return false;
  if (fam == l2s::L2S_EPI_S32 && ((d.ldc & 3) || (d.ldr & 3))) return false;
  if ((d.Cin % 64) || d.N < 256 || d.M < 256 || (d.lda & 7)) return false;
  if (d.mode == L2S_MODE_LINEAR && ((d.M & 7) || (d.N & 7))) return false;   // uniform 8-row staging groups (phasegemm_kernel.h)
  if (mode == 2) return true;
  // whole rounds of 256 blocks: tiles of this kernel vs tiles of the 256x128 kernel (measured relative speed 1.2), and
  // enough K-tiles per block to amortise the six-quarter prologue (measured: a single tile of K <= 1024 loses)
  auto cdivl = [](long a, long b) { return (a + b - 1) / b; };
  const long tm = cdivl(d.M, 256);
  const long rounds_big = cdivl(tm * cdivl(d.N, 256), 256);
  if (rounds_big * ((long)d.Cin * d.ntaps / 64) < 20) return false;
  // (round 2 kept the fp32 residual GEMMs with K <= 1024 on the two-block 256x128 kernel, which hid one block's serialised
  // residual round trips under the other's K loop; with epilogue_stream32 those round trips are gone and this kernel is faster
  // for them too: out-proj 204 vs 234 us, the conformer's N = K = 512 projections 155 vs 172 us at 640 clips)
  const double big = (double)rounds_big * 2.0 / 1.2;
  const double reg = (double)cdivl(tm * cdivl(d.N, 128), 256);
  return big < reg;
}

// widegemm_kernel.h (built by widegemm_inst.hip): four waves of 128x128 for the Linear layers.  Measured equal to the 8-wave kernel
// (round 3, DESIGN.md section 3: +2-3 % in a pure K loop, -0...5 % on the encoder / conformer shapes where its exposed epilogue
// weighs more), so it is NOT part of the product library: `make WIDE=1` / tools/build_variant.sh wide compile it in
// (-DL2S_WITH_WIDEGEMM) and L2S_WIDEGEMM=1 then routes every eligible Linear to it (A/B builds only).
#ifdef L2S_WITH_WIDEGEMM
int l2s_widegemm_f16(const l2s_gemm_desc& d, hipStream_t st);
int l2s_widegemm_bf16(const l2s_gemm_desc& d, hipStream_t st);
static bool wide_ok(const l2s_gemm_desc& d) {
  static const int mode = [] { const char* e = getenv("L2S_WIDEGEMM"); return e ? atoi(e) : 0; }();
  if (mode == 0 || d.mode != L2S_MODE_LINEAR) return false;
  const int fam = l2s::pick_epilogue(d.flags, d.act);
  if (fam > l2s::L2S_EPI_G16A && fam != l2s::L2S_EPI_S32) return false;
  return !(d.M & 7) && !(d.N & 7);        // a staging instruction's 8 rows are inside the matrix or clamped whole
}
#endif

int l2s_phasegemm_launch(const l2s_gemm_desc& d, hipStream_t st) {
  const bool h = d.dtype == L2S_F16;
#ifdef L2S_WITH_WIDEGEMM
  if (wide_ok(d)) return h ? l2s_widegemm_f16(d, st) : l2s_widegemm_bf16(d, st);
#endif
  switch (d.mode) {
    case L2S_MODE_LINEAR: return h ? l2s_phasegemm_f16_m0(d, st) : l2s_phasegemm_bf16_m0(d, st);
    case L2S_MODE_CONV1D: return h ? l2s_phasegemm_f16_m1(d, st) : l2s_phasegemm_bf16_m1(d, st);
    case L2S_MODE_CONV2D: return h ? l2s_phasegemm_f16_m2(d, st) : l2s_phasegemm_bf16_m2(d, st);
    default: return L2S_EINVAL;
  }
}

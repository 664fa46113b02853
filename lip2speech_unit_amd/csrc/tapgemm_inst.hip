// One (mode, dtype) slice of the tap-GEMM: see tapgemm_kernel.h.  Built six times by the Makefile.
#include "tapgemm_kernel.h"

#define L2S_CAT2(a, b) a##b
#define L2S_CAT(a, b) L2S_CAT2(a, b)
#if L2S_INST_ET == 0
using InstET = ElemF16;
#define L2S_INST_NAME L2S_CAT(l2s_tapgemm_f16_m, L2S_INST_MODE)
#else
using InstET = ElemBF16;
#define L2S_INST_NAME L2S_CAT(l2s_tapgemm_bf16_m, L2S_INST_MODE)
#endif

int L2S_INST_NAME(const l2s_gemm_desc& d, hipStream_t st) { return launch_mode<InstET, L2S_INST_MODE>(d, st); }

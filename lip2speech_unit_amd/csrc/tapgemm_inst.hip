// One (dtype, mode, epilogue family) slice of the tap-GEMM: see tapgemm_kernel.h.  Built 60 times by the Makefile.
#include "tapgemm_kernel.h"

#define L2S_CAT3_(a, b, c, d) a##b##c##d
#define L2S_CAT3(a, b, c, d) L2S_CAT3_(a, b, c, d)
#if L2S_INST_ET == 0
using InstET = ElemF16;
#define L2S_INST_NAME L2S_CAT3(l2s_tapgemm_f16_m, L2S_INST_MODE, _e, L2S_INST_EPI)
#else
using InstET = ElemBF16;
#define L2S_INST_NAME L2S_CAT3(l2s_tapgemm_bf16_m, L2S_INST_MODE, _e, L2S_INST_EPI)
#endif

int L2S_INST_NAME(const l2s_gemm_desc& d, hipStream_t st) { return launch_mode<InstET, L2S_INST_MODE, L2S_INST_EPI>(d, st); }

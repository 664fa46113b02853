// One (dtype, mode) slice of the phase-staggered kernel: see phasegemm_kernel.h.  Built 6 times by the Makefile.
#include "phasegemm_kernel.h"

#define L2S_PCAT_(a, b) a##b
#define L2S_PCAT(a, b) L2S_PCAT_(a, b)
#if L2S_INST_ET == 0
using InstET = ElemF16;
#define L2S_INST_NAME L2S_PCAT(l2s_phasegemm_f16_m, L2S_INST_MODE)
#else
using InstET = ElemBF16;
#define L2S_INST_NAME L2S_PCAT(l2s_phasegemm_bf16_m, L2S_INST_MODE)
#endif

int L2S_INST_NAME(const l2s_gemm_desc& d, hipStream_t st) { return launch_phase_mode<InstET, L2S_INST_MODE>(d, st); }

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

#if defined(L2S_PHASE_STAMPS) && L2S_INST_ET == 0 && L2S_INST_MODE == 0
extern "C" int l2s_debug_phase_stamps(void* buf) { return phase_set_stamps(buf); }   // fp16 LINEAR slice only (tools/phase_stamps.py)
#endif

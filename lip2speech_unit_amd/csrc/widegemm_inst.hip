// One dtype slice of the wide-wave-tile Linear kernel: see widegemm_kernel.h.  Built twice by the Makefile.
#include "widegemm_kernel.h"

#if L2S_INST_ET == 0
int l2s_widegemm_f16(const l2s_gemm_desc& d, hipStream_t st) { return launch_wide_epi<ElemF16>(d, st); }
#else
int l2s_widegemm_bf16(const l2s_gemm_desc& d, hipStream_t st) { return launch_wide_epi<ElemBF16>(d, st); }
#endif

// ABI identification.
#include "l2s_common.h"
extern "C" int l2s_abi_version(void) { return L2S_ABI_VERSION; }
extern "C" const char* l2s_build_info(void) { return "lip2speech_hip gfx950 wave64 mfma16x16x32 " __DATE__; }

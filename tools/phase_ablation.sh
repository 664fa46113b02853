# K-loop ablations / variants of the phase-staggered GEMM (diagnostic builds, see tools/build_variant.sh and the L2S_ABL_* switches in
# csrc/phasegemm_kernel.h).  usage: bash tools/phase_ablation.sh <variant> [<variant> ...]
for v in "$@"; do
  echo "=== $v"
  L2S_LIB_PATH=build_ab/$v/liblip2speech_hip.so L2S_PHASEGEMM=2 python tools/phase_stamps.py 640 2>&1 | grep -v amdgpu.ids | grep -A2 "enc qkv\|enc fc2"
done

#!/bin/bash
# Build an A/B or diagnostic variant of the library next to the product one:
#   tools/build_variant.sh <name> "<extra hipcc flags>" <source slice> [<source slice> ...]
# recompiles only the named slices of lip2speech_unit_amd/csrc with the extra flags into build_ab/<name>/ and links them with
# the product build's other objects into build_ab/<name>/liblip2speech_hip.so (select it with L2S_LIB_PATH).
# A slice is a .hip file name, or tapgemm_inst:e<E>_m<M>_f<F> / phasegemm_inst:e<E>_m<M> for the template slices.
set -e
name=$1; flags=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/lip2speech_unit_amd/csrc
out=$root/build_ab/$name
mkdir -p "$out"
make -C "$src" -j8 ARCH=gfx950 > /dev/null
objs=$(ls "$src"/build/*.o)
for sl in "$@"; do
  case $sl in
    tapgemm_inst:*) k=${sl#tapgemm_inst:}; e=${k%%_*}; r=${k#*_}; m=${r%%_*}; f=${r#*_}
      o=tapgemm_inst_${k}.o; defs="-DL2S_INST_ET=${e#e} -DL2S_INST_MODE=${m#m} -DL2S_INST_EPI=${f#f}"; file=tapgemm_inst.hip ;;
    phasegemm_inst:*) k=${sl#phasegemm_inst:}; e=${k%%_*}; m=${k#*_}
      o=phasegemm_inst_${k}.o; defs="-DL2S_INST_ET=${e#e} -DL2S_INST_MODE=${m#m}"; file=phasegemm_inst.hip ;;
    *) o=${sl%.hip}.o; defs=""; file=$sl ;;
  esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $defs $flags -c "$src/$file" -o "$out/$o"
  objs=$(echo "$objs" | grep -v "/$o\$"; echo "$out/$o")
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/liblip2speech_hip.so" $objs
echo "$out/liblip2speech_hip.so"

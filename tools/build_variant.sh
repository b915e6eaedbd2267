#!/usr/bin/env bash
# Build a variant of libsr_yolo2.so for A/B runs on one GPU box (select it with Y2_LIB=<path>):
#   tools/build_variant.sh <name> [--src-rev <git-rev>] [extra hipcc flags for the two conv translation units...]
# The conv kernels are rebuilt with the extra flags (or taken from another git revision); everything else is linked
# from the regular build directory.  Output: sr_object_detection_amd/libsr_yolo2_<name>.so (git-ignored, travels with gpurun).
set -euo pipefail
name=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/sr_object_detection_amd/csrc
B=$C/build_$name
mkdir -p "$B"
src=$C
if [ "${1:-}" = "--src-rev" ]; then
  rev=$2; shift 2
  src=$B/src; mkdir -p "$src"
  for f in y2_conv.hip y2_conv_f16.hip y2_conv_shared.hpp y2_common.hpp; do git -C "$ROOT" show "$rev:sr_object_detection_amd/csrc/$f" > "$src/$f"; done
fi
make -C "$C" -j8 >/dev/null
FLAGS="--offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -std=c++17 -I$ROOT/include -I$C/host -I$C -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS "$@" -c "$src/y2_conv.hip" -o "$B/y2_conv.o" &
/opt/rocm/bin/hipcc $FLAGS "$@" -c "$src/y2_conv_f16.hip" -o "$B/y2_conv_f16.o" &
wait
objs=$(ls "$C"/build/*.o | grep -v -e /y2_conv.o -e /y2_conv_f16.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/sr_object_detection_amd/libsr_yolo2_$name.so" $objs "$B/y2_conv.o" "$B/y2_conv_f16.o" -lm -lstdc++ -ldl
echo "built sr_object_detection_amd/libsr_yolo2_$name.so"

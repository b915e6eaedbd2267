#!/usr/bin/env bash
# Compile the reference's own CPU path (plain C, no -DGPU / -DOPENCV) from the
# sources where they lie under /root/reference into oracle/_ref/.  Nothing is
# copied into the repo; oracle/_ref/ is git-ignored but travels to the GPU box.
#
# Notes:
#  * -iquote, not -I: src_yolo2/unistd.h is a Windows shim that would shadow <unistd.h>
#  * image.c is excluded (does not compile without OpenCV); its symbols stay
#    unresolved in the shared object -- they are only reached from drawing /
#    training code, never from the forward path driven here.
#  * -ffp-contract=off and no -march so rounding is the x86-64 baseline (no FMA).
set -euo pipefail
REF=${REFERENCE_ROOT:-/root/reference}
SRC=$REF/src_yolo2
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
if [ ! -d "$SRC" ]; then
  echo "build_ref: $SRC not present (GPU box?) -- keeping prebuilt files in $OUT" >&2
  exit 0
fi
mkdir -p "$OUT/obj"
FILES="network parser option_list list utils blas gemm im2col col2im activations
convolutional_layer batchnorm_layer maxpool_layer reorg_layer route_layer region_layer
softmax_layer avgpool_layer cost_layer box tree layer cuda data connected_layer crop_layer
detection_layer dropout_layer gru_layer rnn_layer crnn_layer local_layer normalization_layer
shortcut_layer activation_layer deconvolutional_layer matrix detector"
CFLAGS="-O2 -w -fPIC -fopenmp -ffp-contract=off -iquote $SRC"
objs=""
for f in $FILES; do
  gcc $CFLAGS -c "$SRC/$f.c" -o "$OUT/obj/$f.o" &
  objs="$objs $OUT/obj/$f.o"
done
wait
gcc -shared -fopenmp -o "$OUT/libdarknet_ref.so" $objs -lm -lpthread
# driver (our code) that calls the reference API and dumps tensors
gcc -O2 -w -fopenmp -ffp-contract=off -iquote "$SRC" "$HERE/ref_driver.c" \
    -o "$OUT/ref_driver" -L"$OUT" -ldarknet_ref -Wl,-rpath,'$ORIGIN' \
    -Wl,--unresolved-symbols=ignore-in-shared-libs -lm -lpthread
echo "build_ref: built $OUT/libdarknet_ref.so and $OUT/ref_driver"

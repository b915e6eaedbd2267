#!/usr/bin/env bash
# Forced-tile sweep behind the host's tile cost model (pick_variant, y2_conv.hip): TFLOP/s per conv layer with every
# layer forced to one tile shape (Y2_CONV_TILE), plus the model's own choice.   tools/tile_sweep.sh <workload> [steps]
w=${1:-yolo608_b32}; steps=${2:-5}
for t in auto 192x256 256x128 128x128 128x64 64x64 256x64 128x32; do
  if [ $t = auto ]; then unset Y2_CONV_TILE; else export Y2_CONV_TILE=$t; fi
  echo "== $t"
  python tools/layer_profile.py $w $steps 2>/dev/null | awk '/conv_/ {printf "%s:%s(%s) ", $1, $NF, $(NF-1)} /^total/ {print ""; print $0}'
done

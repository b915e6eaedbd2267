#!/usr/bin/env bash
# hybrid stream-K of the fp32 192x256 kernel: the yolo.cfg 608 b32 layer shapes with and without it (order alternating),
# then the in-kernel stamps (tools/build_variant.sh stamps -DY2_F32_STAMPS) of both forms and of the hybrid
# instantiation walking every tile whole (Y2_SKH_NOSPLIT: code generation alone)
set -uo pipefail
mkdir -p gpurun_out
O=gpurun_out/skh_probe.txt; : > $O
for shape in ${SHAPES:-"76 128 256" "38 256 512" "19 512 1024" "19 1024 1024"}; do
  for r in 1 2; do
    if [ $r = 1 ]; then A="Y2_SKH=0"; B="Y2_SKH=-1"; else A="Y2_SKH=-1"; B="Y2_SKH=0"; fi
    echo "$A: $(env $A python tools/conv_probe.py $shape 3 1 leaky 32 7 2>/dev/null | tail -n 1)" >> $O
    echo "$B: $(env $B python tools/conv_probe.py $shape 3 1 leaky 32 7 2>/dev/null | tail -n 1)" >> $O
  done
  echo "nosplit: $(env Y2_SKH_NOSPLIT=1 python tools/conv_probe.py $shape 3 1 leaky 32 7 2>/dev/null | tail -n 1)" >> $O
  for E in "Y2_SKH=0" "Y2_SKH=-1" "Y2_SKH_NOSPLIT=1"; do
    echo "$E stamps: $(env $E Y2_F32_STAMPS=1 Y2_LIB=$PWD/sr_object_detection_amd/libsr_yolo2_stamps.so python tools/conv_probe.py $shape 3 1 leaky 32 2 2>&1 | grep 'f32 stamps conv_mfma_f32_192' | tail -n 1)" >> $O
  done
done
cat $O

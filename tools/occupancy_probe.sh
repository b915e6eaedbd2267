#!/usr/bin/env bash
# How does a K-tile of the fp16 256x256 kernel scale with the number of workgroups running?  One 3x3 layer
# (28x28 256->512, batch 128: 784 tiles) on capped persistent grids; then the in-kernel clock of the stamped build.
set -uo pipefail
mkdir -p gpurun_out
export Y2_PROBE_HALF=1 Y2_SK=0 Y2_TAIL=0
for r in 1 2; do
for g in 16 32 64 128 192 224 256; do
  echo -n "grid $g: "; Y2_CONV_GRID=$g python tools/conv_probe.py 28 256 512 3 1 leaky 128 5 2>/dev/null | tail -1
done
done
for g in 32 128 256; do
  echo "stamped build, grid $g"
  Y2_LIB=$PWD/sr_object_detection_amd/libsr_yolo2_stamps.so Y2_P8_STAMPS=1 Y2_CONV_GRID=$g python tools/conv_probe.py 28 256 512 3 1 leaky 128 3 2>&1 | grep "p8 stamps" | tail -2
done

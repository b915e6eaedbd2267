#!/usr/bin/env bash
# timing ablation of the fp16 256x256 kernel's staging traffic (results are garbage): one 28x28 256->512 and one 14x14 512->1024 layer
set -uo pipefail
export Y2_PROBE_HALF=1 Y2_SK=0 Y2_TAIL=0 Y2_LIB=$PWD/sr_object_detection_amd/libsr_yolo2_ablate.so
for r in 1 2; do
for dbg in 0 128 512 768; do
  echo -n "Y2_DBG=$dbg: "; Y2_DBG=$dbg python tools/conv_probe.py 28 256 512 3 1 leaky 128 5 2>/dev/null | tail -1
  echo -n "Y2_DBG=$dbg: "; Y2_DBG=$dbg python tools/conv_probe.py 14 512 1024 3 1 leaky 128 5 2>/dev/null | tail -1
done
done

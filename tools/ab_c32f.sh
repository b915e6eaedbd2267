#!/usr/bin/env bash
set -uo pipefail
mkdir -p gpurun_out
for W in yolo608_b32 yolo416_b8; do
for r in 1 2; do
  Y2_NO_C32F=1 python tools/layer_profile.py $W 6 > gpurun_out/c32f_${W}_off_$r.log 2>&1
  python tools/layer_profile.py $W 6 > gpurun_out/c32f_${W}_on_$r.log 2>&1
done
done
grep -h "x32 *->64\|^total" gpurun_out/c32f_*.log

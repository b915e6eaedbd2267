#!/usr/bin/env bash
set -uo pipefail
mkdir -p gpurun_out
W=darknet19_448_b128_f16
for r in 1 2; do
  Y2_NO_C64=1 python tools/layer_profile.py $W 8 > gpurun_out/c64_off_$r.log 2>&1
  python tools/layer_profile.py $W 8 > gpurun_out/c64_on_$r.log 2>&1
done
grep -h "112x112 *x64\|^total" gpurun_out/c64_*.log

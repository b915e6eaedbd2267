#!/usr/bin/env bash
# A/B of the stream-K plan on one box (interleaved rounds): per-layer times of darknet19_448 b128 fp16
set -uo pipefail
mkdir -p gpurun_out
W=${1:-darknet19_448_b128_f16}
for r in 1 2; do
  Y2_SK=0 python tools/layer_profile.py $W 8 > gpurun_out/sk_off_$r.log 2>&1
  python tools/layer_profile.py $W 8 > gpurun_out/sk_model_$r.log 2>&1
  Y2_SK_MARGIN=2 python tools/layer_profile.py $W 8 > gpurun_out/sk_all_$r.log 2>&1
  Y2_SK_MARGIN=2 Y2_SK_MINK=8 python tools/layer_profile.py $W 8 > gpurun_out/sk_all_mink8_$r.log 2>&1
  echo "round $r done"
done
tail -n 1 gpurun_out/sk_*.log

#!/usr/bin/env bash
# sweep of the tail tile behind the fp16 256x256 kernel (darknet19_448 b128): per-layer times, interleaved with the
# plain run and the stream-K run on ONE box
set -uo pipefail
mkdir -p gpurun_out
W=${1:-darknet19_448_b128_f16}
for r in 1 2; do
  Y2_SK=0 Y2_TAIL=0 python tools/layer_profile.py $W 8 > gpurun_out/tl_off_$r.log 2>&1
  Y2_TAIL=0 Y2_SK_MARGIN=2 Y2_SK_MINK=8 python tools/layer_profile.py $W 8 > gpurun_out/tl_sk_$r.log 2>&1
  for t in 256x128 256x64 128x128 128x64 64x64; do
    Y2_TAIL_TILE=$t python tools/layer_profile.py $W 8 > gpurun_out/tl_${t}_$r.log 2>&1
  done
  python tools/layer_profile.py $W 8 > gpurun_out/tl_model_$r.log 2>&1
  echo "round $r done"
done
tail -qn 1 gpurun_out/tl_*.log

#!/usr/bin/env bash
# A/B of two builds of the library on one box: tools/ab_lib.sh <variant-name> <workload> [rounds]   (variant: tools/build_variant.sh)
set -uo pipefail
mkdir -p gpurun_out
V=$1; W=$2; N=${3:-2}
for r in $(seq 1 $N); do
  Y2_LIB=$PWD/sr_object_detection_amd/libsr_yolo2_$V.so python tools/layer_profile.py $W 6 > gpurun_out/ab_${V}_${W}_$r.log 2>&1
  python tools/layer_profile.py $W 6 > gpurun_out/ab_cur_${W}_$r.log 2>&1
done
for r in $(seq 1 $N); do echo "$V $r: $(tail -n 1 gpurun_out/ab_${V}_${W}_$r.log)"; echo "cur $r: $(tail -n 1 gpurun_out/ab_cur_${W}_$r.log)"; done

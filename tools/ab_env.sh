#!/usr/bin/env bash
# A/B of an environment switch on one box, order alternating:  tools/ab_env.sh "VAR=value" <workload> [rounds]
set -uo pipefail
mkdir -p gpurun_out
E=$1; W=$2; N=${3:-2}
for r in $(seq 1 $N); do
  if [ $((r % 2)) = 1 ]; then
    env $E python tools/layer_profile.py $W 8 > gpurun_out/abe_set_${W}_$r.log 2>&1; python tools/layer_profile.py $W 8 > gpurun_out/abe_def_${W}_$r.log 2>&1
  else
    python tools/layer_profile.py $W 8 > gpurun_out/abe_def_${W}_$r.log 2>&1; env $E python tools/layer_profile.py $W 8 > gpurun_out/abe_set_${W}_$r.log 2>&1
  fi
done
for r in $(seq 1 $N); do echo "$E $r: $(tail -n 1 gpurun_out/abe_set_${W}_$r.log)"; echo "default $r: $(tail -n 1 gpurun_out/abe_def_${W}_$r.log)"; done

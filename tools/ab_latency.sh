#!/usr/bin/env bash
# batch-1 resident frame time of two builds, interleaved:  tools/ab_latency.sh <variant> [rounds]
set -uo pipefail
V=$1; N=${2:-3}
for r in $(seq 1 $N); do
  for lib in $V cur; do
    L=""; [ $lib != cur ] && L=$PWD/sr_object_detection_amd/libsr_yolo2_$lib.so
    for w in tiny416_b1 yolo416_b1; do
      echo -n "$lib $w: "; Y2_LIB=$L python bench.py --workload $w --cpu-iters 0 --host-input off --latency-iters 0 --steps 300 --warmup 30 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['ms_per_step'])"
    done
  done
done

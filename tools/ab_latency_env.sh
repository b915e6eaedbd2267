#!/usr/bin/env bash
# batch-1 latency block of bench.py (test_detector_img / Detector::detect / device part) with and without an environment
# switch, order alternating:  tools/ab_latency_env.sh "VAR=value" [rounds]
set -uo pipefail
E=$1; N=${2:-2}
run() { python bench.py --workload tiny416_b1 --cpu-iters 0 --host-input off --clock-probe 0 --steps 200 --warmup 30 --latency-iters 300 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); l=d['latency']['nets']
print('pipelined %.4f ms/frame |' % d['ms_per_step'], ' | '.join('%s: img %.4f det %.4f dev %.4f' % (k.split()[0], v['test_detector_img']['p50_ms'], v['Detector_detect']['p50_ms'], v['device_only']['p50_ms']) for k, v in l.items()))"; }
for r in $(seq 1 $N); do
  if [ $((r % 2)) = 1 ]; then echo "$E $r: $(env $E bash -c "$(declare -f run); run")"; echo "default $r: $(run)";
  else echo "default $r: $(run)"; echo "$E $r: $(env $E bash -c "$(declare -f run); run")"; fi
done

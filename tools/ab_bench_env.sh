#!/usr/bin/env bash
# A/B of an environment switch in the benchmark's own context (sustained, pipelined steps), order alternating:
#   tools/ab_bench_env.sh "VAR=value" [workload] [rounds]
set -uo pipefail
mkdir -p gpurun_out
E=$1; W=${2:-yolo608_b32}; N=${3:-2}
run() { python bench.py --workload $W --steps 40 --warmup 10 --latency-iters 0 --clock-probe 0 --cpu-iters 0 --host-input off 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'dominant %.1f %s frac %.4f' % (r['achieved'], r['unit'], r['frac']))"; }
for r in $(seq 1 $N); do
  if [ $((r % 2)) = 1 ]; then echo "$E $r: $(env $E bash -c "$(declare -f run); W=$W run")"; echo "default $r: $(run)";
  else echo "default $r: $(run)"; echo "$E $r: $(env $E bash -c "$(declare -f run); W=$W run")"; fi
done

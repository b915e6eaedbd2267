mkdir -p gpurun_out/b1
for g in 0 1 0 1; do Y2_GRAPH=$g python bench.py --workload tiny416_b1 --cpu-iters 0 --host-input off --steps 300 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph=$g tiny', d['value'], d['ms_per_step'])"; done
for g in 0 1; do Y2_GRAPH=$g python bench.py --workload yolo608_b1 --cpu-iters 0 --host-input off --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph=$g yolo608b1', d['value'], d['ms_per_step'])"; done

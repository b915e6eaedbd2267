mkdir -p gpurun_out/t
for i in 1 2 3 4; do
  timeout -k 10 300 python -m pytest tests -m gpu -q -x -s > gpurun_out/t/fix_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc"; grep -E " passed| failed|Memory access" gpurun_out/t/fix_$i.log | tail -2
  if [ $rc != 0 ] || grep -q "Memory access fault" gpurun_out/t/fix_$i.log; then exit 1; fi
done
for i in 1 2; do python bench.py --workload darknet19_448_b128_f16 --cpu-iters 0 --host-input off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('d19 f16', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['achieved'])"; done
python bench.py --workload yolo608_b32_f16 --cpu-iters 0 --host-input off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('yolo608 f16', d['value'], d['ms_per_step'], d['roofline']['frac'])"

set -o pipefail
mkdir -p gpurun_out/sk
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/sk/pytest.log 2>&1; rc=$?; tail -4 gpurun_out/sk/pytest.log; [ $rc = 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for w in tiny416_b1 yolo608_b1 yolo416_b8; do
for m in new old new old; do
  if [ $m = old ]; then export Y2_SPLITK_TWO_PASS=1; else unset Y2_SPLITK_TWO_PASS; fi
  python bench.py --workload $w --cpu-iters 0 --host-input off --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $m', d['value'], d['ms_per_step'])"
done; done
unset Y2_SPLITK_TWO_PASS
python tools/layer_profile.py tiny416_b1 16 2>/dev/null | tail -18

set -o pipefail
R=${1:-r03}
mkdir -p gpurun_out/final
bash tools/collect_profiles.sh yolo608_b32 ${R}_608b32 > gpurun_out/final/cp_608.log 2>&1 && echo cp608 ok &&
bash tools/collect_profiles.sh darknet19_448_b128_f16 ${R}_d19f16 > gpurun_out/final/cp_d19.log 2>&1 && echo cpd19 ok &&
bash tools/collect_profiles.sh yolo416_b8 ${R}_416b8 > gpurun_out/final/cp_416.log 2>&1 && echo cp416 ok &&
bash tools/collect_profiles.sh yolo9000_544_b8 ${R}_9k544b8 > gpurun_out/final/cp_9k.log 2>&1 && echo cp9k ok &&
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err && echo bench ok &&
for w in darknet19_448_b128_f16 yolo416_b8 yolo9000_544_b8 tiny416_b1; do python bench.py --workload $w --cpu-iters 0 --latency-iters 0 > gpurun_out/final/bench_$w.json 2>/dev/null || exit 1; done
cat gpurun_out/final/bench_default.json | cut -c1-600

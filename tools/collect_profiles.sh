#!/usr/bin/env bash
# Everything the `roofline` object of bench.py is checked against, for one workload, on the GPU box:
#   tools/collect_profiles.sh <workload> <tag>      -> gpurun_out/prof_<tag>/ (copy the summaries into profiles/)
#  1. rocprofv3 --kernel-trace --stats of the bench command      -> <tag>_kernel_stats.csv + the bench line under the profiler
#  2. three separate --pmc passes (never combined with other trace domains) of a shorter run, summarised per conv launch
#     by tools/pmc_summary.py (clock, MFMA-busy, HBM bytes with the gfx950 FETCH_SIZE x2 correction, L2 hit rate)
# rocprofv3 is given `python3 bench.py ...` directly (no env / bash -c hop: the profiler's preloaded library has already
# initialised the GPU).
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
w=$1; tag=$2
out=gpurun_out/prof_$tag; mkdir -p "$out"
args="--workload $w --cpu-iters 0 --host-input off --autotune 0 --latency-iters 0 --clock-probe 0"
rocprofv3 --kernel-trace --stats -f csv -d "$out/kt" -o kt -- python3 bench.py $args --steps 10 --warmup 3 > "$out/${tag}_bench_under_rocprof.json" 2> "$out/kt.err" || { tail -5 "$out/kt.err"; exit 1; }
find "$out/kt" -name "*kernel_stats.csv" -exec cp {} "$out/${tag}_kernel_stats.csv" \;
export Y2_BENCH_DUMP_KERNELS=$PWD/$out/layer_kernels.json
i=0
for pmc in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pmc -f csv -d "$out/p$i" -o p$i -- python3 bench.py $args --steps 3 --warmup 2 > /dev/null 2> "$out/p$i.err" || { tail -5 "$out/p$i.err"; exit 1; }
  find "$out/p$i" -name "*counter_collection.csv" -exec cp {} "$out/p${i}_counter_collection.csv" \;
  find "$out/p$i" -name "*kernel_trace.csv" -exec cp {} "$out/p${i}_kernel_trace.csv" \;
done
python3 tools/pmc_summary.py "$out" --names "$out/layer_kernels.json" --json "$out/${tag}_pmc_traffic.json" --workload "$w" --wl-name "$w" > "$out/${tag}_pmc_summary.txt" 2>&1
tail -40 "$out/${tag}_pmc_summary.txt"
rm -rf "$out"/kt "$out"/p1 "$out"/p2 "$out"/p3

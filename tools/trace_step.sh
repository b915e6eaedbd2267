#!/usr/bin/env bash
# per-dispatch kernel durations of ONE bench step (forward + decode + NMS + collect; the last of a short run) under
# rocprofv3 --kernel-trace:   tools/trace_step.sh <workload> <tag> [VAR=value ...]   -> gpurun_out/step_<tag>.txt
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
w=$1; tag=$2; shift 2
for kv in "$@"; do export "$kv"; done
out=gpurun_out/stept_$tag; rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --memory-copy-trace -f csv -d "$out" -o kt -- python3 bench.py --workload $w --cpu-iters 0 --host-input off --latency-iters 0 --steps 6 --warmup 3 > "$out/run.log" 2>&1 || { tail -5 "$out/run.log"; exit 1; }
python3 - "$out" > gpurun_out/step_$tag.txt <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:64]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %s B" % (r.get("Direction", "?"), r.get("Size", "?"))))
rows.sort()
firsts = [i for i, r in enumerate(rows) if "conv_first" in r[2]]
a, b = firsts[-2], firsts[-1]
prev = None
tot = 0.0
for s, e, n in rows[a:b]:
    gap = (s - prev) / 1e3 if prev else 0.0
    print("%8.1f us  gap %6.1f  %s" % ((e - s) / 1e3, gap, n))
    prev = e
    tot += (e - s) / 1e3
print("sum of kernels+copies %.1f us; first start to last end %.1f us" % (tot, (rows[b - 1][1] - rows[a][0]) / 1e3))
PY
rm -rf "$out"
cat gpurun_out/step_$tag.txt

#!/usr/bin/env bash
# per-dispatch kernel durations of ONE forward pass (the last of a short run) under rocprofv3 --kernel-trace:
#   tools/trace_forward.sh <workload> <tag> [VAR=value ...]      -> gpurun_out/trace_<tag>.txt
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
w=$1; tag=$2; shift 2
for kv in "$@"; do export "$kv"; done
out=gpurun_out/trace_$tag; rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace -f csv -d "$out" -o kt -- python3 tools/layer_profile.py $w 3 > "$out/run.log" 2>&1 || { tail -5 "$out/run.log"; exit 1; }
python3 - "$out" > gpurun_out/trace_$tag.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last forward = from the last dispatch of the first-layer kernel on
idx = max(i for i, r in enumerate(rows) if "conv_first" in r["Kernel_Name"])
prev_end = None
tot = 0
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    name = r["Kernel_Name"].split("(")[0][:70]
    print("%9.1f us  gap %6.1f  grid %6s  %s" % ((e - s) / 1e3, gap, r.get("Grid_Size", "?"), name))
    prev_end = e
    tot += (e - s) / 1e3
print("sum of kernels %.1f us" % tot)
PY
rm -rf "$out"
tail -3 gpurun_out/trace_$tag.txt

#!/usr/bin/env bash
# yolo9000 544 b8: the sparse tree detection chain against the dense one, in the benchmark's context (order alternating)
set -uo pipefail
tools/ab_bench_env.sh "Y2_DETECT_SEPARATE=1" yolo9000_544_b8 ${1:-2}

#!/bin/bash
# kernel-trace pass only -> gpurun_out/<tag>/kernel_stats.csv
set -e
TAG=${1:-trace}; STEPS=5; WARM=2; N=$((STEPS + WARM)); ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline $BENCH_EXTRA > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log"
python3 tools/pmc_summary.py stats "$OUT/trace" $N "$OUT/kernel_stats.csv"
rm -rf "$OUT/trace"
head -40 "$OUT/kernel_stats.csv"

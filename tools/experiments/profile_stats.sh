#!/bin/bash
# One rocprofv3 kernel-trace pass over the headline bench with extra bench arguments -> gpurun_out/<tag>/kernel_stats.csv.
# Usage (on the GPU box): bash tools/profile_stats.sh r03_mixed --precise mixed
set -e
TAG=${1:-prof}; shift
STEPS=5; WARM=2; N=$((STEPS + WARM + 1))   # + the untimed probe step of bench.py
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --parity-leg none "$@" > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log"
python3 tools/pmc_summary.py stats "$OUT/trace" $N "$OUT/kernel_stats.csv"
python3 tools/pmc_summary.py timeline "$OUT/trace" $N "$OUT/step_timeline.txt"
rm -rf "$OUT/trace"
head -40 "$OUT/kernel_stats.csv"

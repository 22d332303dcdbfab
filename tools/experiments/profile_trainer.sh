#!/bin/bash
# rocprofv3 kernel trace of harness.EndToEndTrainer iterations -> gpurun_out/<tag>_trainer_{kernel_stats.csv,timeline.txt}
# Usage (GPU box): bash tools/profile_trainer.sh r03 2 1        (batch 2, hipGraphs on)
set -e
TAG=$1; B=${2:-2}; GR=${3:-1}; IT=6
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=$ROOT/gpurun_out/${TAG}_trainer
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 tools/prof_trainer.py $B $IT $GR > "$OUT/log.txt" 2> "$OUT/trace.log"
python3 tools/pmc_summary.py stats "$OUT/trace" $((IT + 5)) "$ROOT/gpurun_out/${TAG}_trainer_kernel_stats.csv"
python3 tools/pmc_summary.py timeline "$OUT/trace" $((IT + 5)) "$ROOT/gpurun_out/${TAG}_trainer_timeline.txt"
rm -rf "$OUT/trace"
tail -1 "$OUT/log.txt"

#!/bin/bash
# Three rocprofv3 passes over the headline bench (kernel trace, FETCH_SIZE, WRITE_SIZE) -> gpurun_out/<tag>_*.
# Usage (on the GPU box): bash tools/profile_step.sh r01_final
set -e
TAG=${1:-prof}
STEPS=5; WARM=2; N=$((STEPS + WARM + 1))   # + the untimed probe step of bench.py
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --fast-leg none > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- python3 bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --fast-leg none > /dev/null 2> "$OUT/fetch.log"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- python3 bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --fast-leg none > /dev/null 2> "$OUT/write.log"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/mfma" -o m -- python3 bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --fast-leg none > /dev/null 2> "$OUT/mfma.log"
python3 tools/pmc_summary.py mfma "$OUT/mfma" $N "$OUT/pmc_mfma.json"
python3 tools/pmc_summary.py stats "$OUT/trace" $N "$OUT/kernel_stats.csv"
python3 tools/pmc_summary.py timeline "$OUT/trace" $N "$OUT/step_timeline.txt"
python3 tools/pmc_summary.py pmc "$OUT/fetch" "$OUT/write" $N "$OUT/pmc_traffic.json"
rm -rf "$OUT/trace" "$OUT/fetch" "$OUT/write" "$OUT/mfma"
head -12 "$OUT/kernel_stats.csv"

#!/bin/bash
# rocprofv3 kernel-trace summaries of the non-headline configs -> gpurun_out/<tag>_<which>_kernel_stats.csv
# Usage (GPU box): bash tools/profile_configs.sh r02 g2 d2 u2 v128
set -e
TAG=$1; shift
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
for W in "$@"; do
  OUT=$ROOT/gpurun_out/${TAG}_cfg_$W
  mkdir -p "$OUT"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 tools/prof_config.py $W 3 > "$OUT/log.txt" 2> "$OUT/trace.log"
  python3 tools/pmc_summary.py stats "$OUT/trace" 5 "$ROOT/gpurun_out/${TAG}_cfg_${W}_kernel_stats.csv"
  python3 tools/pmc_summary.py timeline "$OUT/trace" 5 "$ROOT/gpurun_out/${TAG}_cfg_${W}_timeline.txt"
  rm -rf "$OUT/trace"
  tail -1 "$OUT/log.txt"
  head -14 "$ROOT/gpurun_out/${TAG}_cfg_${W}_kernel_stats.csv"; tail -1 "$ROOT/gpurun_out/${TAG}_cfg_${W}_kernel_stats.csv"
done

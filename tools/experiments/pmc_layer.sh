#!/bin/bash
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
for L in "d43 16 1024 1024" "u40 256 128 64" "inc3 256 64 64"; do
  set -- $L
  OUT=$ROOT/gpurun_out/pmcl_$1; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -o p -- python3 tools/_onelayer.py $1 $2 $3 $4 > /dev/null 2> $OUT/log.txt
  python3 - "$OUT" "$1" <<'PY'
import csv, glob, sys, collections
d, name = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(float); n = 0
for r in csv.DictReader(open(f)):
    if "conv3x3_big" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
print(name, {k: round(v / acc["SQ_WAVE_CYCLES"], 4) for k, v in acc.items()})
PY
done

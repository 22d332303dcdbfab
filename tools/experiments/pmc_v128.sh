set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_v128; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o f -- python3 tools/prof_config.py v128 2 > $OUT/log.txt 2> $OUT/err.txt
python3 - <<'P'
import csv, glob, collections
f=glob.glob('gpurun_out/pmc_v128/f/**/*counter_collection.csv', recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if r['Counter_Name']=='FETCH_SIZE']
rows.sort(key=lambda r:int(r['Dispatch_Id']))
# last step: print wgrad launches
out=[]
for r in rows:
    n=r['Kernel_Name']
    if 'wgrad3x3' in n or 'conv3x3_dma' in n:
        out.append((int(r['Dispatch_Id']), n.split('(')[0][-45:], float(r['Counter_Value'])*2*1024/1e6, r.get('Grid_Size','?')))
for o in out[-60:]: print("%6d %-45s %9.1f MB  grid %s"%o)
P
rm -rf $OUT/f

#!/bin/bash
# held clock of the conv kernel with and without its DMA traffic (GRBM_GUI_ACTIVE / 8 / duration per dispatch)
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/dma_clock; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/p" -o c -- python3 tools/ablate_conv_dma.py > "$OUT/run.txt" 2> "$OUT/err.txt"
python3 - "$OUT" <<'PY'
import sys, csv, glob, collections
out=sys.argv[1]
f=glob.glob(out+'/p/**/*counter_collection.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
# group per dispatch
d=collections.OrderedDict()
for r in rows:
    k=r['Dispatch_Id']
    e=d.setdefault(k, {'name':r['Kernel_Name'], 'start':int(r['Start_Timestamp']), 'end':int(r['End_Timestamp'])})
    e[r['Counter_Name']]=float(r['Counter_Value'])
res=[]
for k,e in d.items():
    if 'conv3x3_dma' not in e['name']: continue
    dur=(e['end']-e['start'])*1e-9
    if 'GRBM_GUI_ACTIVE' in e:
        res.append((e['name'][:40], dur*1e6, e['GRBM_GUI_ACTIVE']/8/dur/1e9, e.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(1024*e['GRBM_GUI_ACTIVE']/8)))
# the ablate script runs per layer 6 variants x 12 launches (2 warm-up + 10): print per group of 12 the mean
grp=12
for i in range(0, len(res), grp):
    g=res[i:i+grp]
    if not g: break
    print(f"{i//grp:3d} {g[0][0]:40s} dur {sum(x[1] for x in g)/len(g):7.1f} us  clock {sum(x[2] for x in g)/len(g):5.2f} GHz  mfma_busy {sum(x[3] for x in g)/len(g):5.2f}")
PY

#!/bin/bash
# per-kernel durations of the partitioned group-by (rocprofv3 --kernel-trace): tools/gb_trace.sh <tag> <keys> [tuning]
TAG=$1; KEYS=${2:-100000}; TUN=${3:-}
R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$TAG -- python3 $R/tools/bench_groupby.py 1000000000 $KEYS "$TUN" 10 > $R/gpurun_out/$TAG.log 2>&1
grep "GROUP BY s" $R/gpurun_out/$TAG.log
python3 - $R/gpurun_out/$TAG <<'PY'
import csv,sys,glob,collections
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
seq=[(r['Kernel_Name'].split('(')[0].replace('qe::',''),(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6) for r in rows if 'gb_' in r['Kernel_Name']]
i=0
out=[]
while i<len(seq):
    if seq[i][0]=='qe_gb_count':
        j=i+1
        while j<len(seq) and seq[j][0]!='qe_gb_count': j+=1
        out.append(seq[i:j]); i=j
    else: i+=1
for k in (1,len(out)-2):
    if 0<=k<len(out): print(' | '.join(f"{n} {t:.3f}" for n,t in out[k]), ' total', round(sum(t for _,t in out[k]),3))
PY

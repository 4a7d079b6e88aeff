#!/bin/bash
# per-kernel durations of the kernel-per-node pipeline (rocprofv3 --kernel-trace --stats): tools/pn_trace.sh <tag> [workload, default config2]
TAG=$1
R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python3 $R/bench.py --workload ${2:-config2} --exec-mode per_node --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/$TAG.log 2>&1
python3 - $R/gpurun_out/$TAG $R/gpurun_out/$TAG.log <<'PY'
import csv,sys,glob,json
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if float(r['AverageNs'])>50000: print(f"  {r['Name'][:44]:44s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e6:.3f} ms")
for l in open(sys.argv[2]):
    if l.startswith('{'):
        j=json.loads(l); print('  ms_per_step', round(j['ms_per_step'],3), 'kernel_ms', round(j['roofline']['kernel_ms'],3))
PY

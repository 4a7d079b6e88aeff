#!/bin/bash
# hash-partitioned GROUP BY, lines of records against header records: per-kernel times (kernel trace) and the scatter's phases
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp QE_HP_FROM=20000
KEYS=${KEYS:-100000}
for T in "0,0,0,0,0,0" "0,0,0,0,0,33554432"; do
  D=gpurun_out/hp_ab/$(echo $T | tr ',' '_'); mkdir -p $D
  echo "== tuning $T"
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 tools/bench_groupby_numeric.py 1000000000 $KEYS $T > $D/out.txt 2> $D/err.txt || { echo failed; exit 1; }
  tail -1 $D/out.txt
  python3 - $D <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    rows = [r for r in rows if r['Kernel_Name'].startswith('qe_gb')][-3:]
    for r in rows:
        print(f"   {r['Kernel_Name'][:28]:28s} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6:8.3f} ms lds {r.get('LDS_Block_Size')}")
PY
  T64=$(echo $T | awk -F, '{printf "%s,%s,%s,%s,%s,%d", $1,$2,$3,$4,$5,$6+64}')
  timeout -k 10 200 python3 tools/bench_groupby_numeric.py 1000000000 $KEYS $T64 2>&1 | grep phases | tail -1
done

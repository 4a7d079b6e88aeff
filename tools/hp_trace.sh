#!/bin/bash
# per-kernel times (rocprofv3 --kernel-trace --stats) of the hash-partitioned GROUP BY for a few settings
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
ROWS=${ROWS:-1000000000}
i=0
while read -r KEYS P SH; do
  [ -z "$KEYS" ] && continue
  i=$((i+1)); D=gpurun_out/hp_trace/$i; mkdir -p $D
  echo "== keys $KEYS parts $P shift $SH"
  export QE_HP_FROM=50000; if [ "$P" != "-" ]; then export QE_HP_PARTS=$P QE_HP_SHIFT=$SH; else unset QE_HP_PARTS QE_HP_SHIFT; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 tools/bench_groupby_numeric.py $ROWS $KEYS > $D/out.txt 2> $D/err.txt || { echo failed; exit 1; }
  tail -1 $D/out.txt
  python3 - $D <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Name'].startswith(('qe_gb', 'qe_fused', 'qe_ht', 'exclusive', 'qe_')):
            print(f"   {r['Name'][:34]:34s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:8.3f} ms min {float(r['MinNs'])/1e6:8.3f}")
PY
done <<< "${CASES:-100000 256 11
100000 512 11
1000000 - -}"

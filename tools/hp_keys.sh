#!/bin/bash
# hash-partitioned GROUP BY over key counts (default sizing): kernel ms of the last executions
cd "${GRAFT_REPO_ROOT:-.}"
for KEYS in ${KEYSET:-30000 100000 300000 1000000}; do
  echo "== keys $KEYS"
  timeout -k 10 150 python3 tools/bench_groupby_numeric.py ${ROWS:-1000000000} $KEYS ${TUNING:-0} 2>&1 | grep -v amdgpu.ids | tail -2 || exit 1
done

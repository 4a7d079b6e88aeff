#!/bin/bash
# hash-partitioned GROUP BY: lean (16-byte records + row ids) against header records, default sizing
cd "${GRAFT_REPO_ROOT:-.}"
for KEYS in ${KEYSET:-30000 100000 300000 1000000}; do
  for T in "0" "0,0,0,0,0,33554432"; do
    echo "== keys $KEYS tuning $T"
    QE_HP_FROM=20000 timeout -k 10 150 python3 tools/bench_groupby_numeric.py ${ROWS:-1000000000} $KEYS $T 2>&1 | grep -v amdgpu.ids | tail -2 || exit 1
  done
done

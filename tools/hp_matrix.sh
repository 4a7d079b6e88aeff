#!/bin/bash
# kernel times of the hash-partitioned GROUP BY over partition counts / table sizes (one process per setting: the knobs are read once)
cd "${GRAFT_REPO_ROOT:-.}"
ROWS=${ROWS:-1000000000}
for KEYS in ${KEYSET:-100000 1000000}; do
  for SET in ${SETS:-"512 9" "512 10" "512 11" "256 10" "256 11" "1024 10"}; do
    set -- $SET
    echo "== keys $KEYS parts $1 shift $2"
    QE_HP_FROM=50000 QE_HP_PARTS=$1 QE_HP_SHIFT=$2 timeout -k 10 150 python3 tools/bench_groupby_numeric.py $ROWS $KEYS 2>&1 | grep -v amdgpu.ids | tail -2 || exit 1
  done
done

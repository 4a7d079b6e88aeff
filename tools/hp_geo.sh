cd "${GRAFT_REPO_ROOT:-.}"
for SET in "256 11 256,4" "256 11 256,2" "128 11 256,4" "256 11 512,2"; do
  set -- $SET
  echo "== parts $1 shift $2 tuning $3"
  QE_HP_FROM=50000 QE_HP_PARTS=$1 QE_HP_SHIFT=$2 timeout -k 10 150 python3 tools/bench_groupby_numeric.py 1000000000 100000 $3,0,0,0,64 2>&1 | grep -v amdgpu.ids | tail -2 || exit 1
done

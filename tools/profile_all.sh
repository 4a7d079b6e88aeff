#!/bin/bash
# Run on the GPU box (through gpurun).  For every case: one un-profiled bench run (settles and persists the plan's
# geometry choice, gives the un-profiled numbers), then rocprofv3 in THREE separate passes of the same command
# (--kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE: the two counters do not fit one pass and PMC passes never
# carry trace options -- MI355X_MICROARCH.md, rocprofv3 PMC slots).  The program follows `--` directly (python3).
# Raw output -> gpurun_out/prof_<tag>/<case>/ ; tools/summarize_profile.py condenses it into profiles/.
#   usage: tools/profile_all.sh <tag> [case ...]     (no case = all)
set -o pipefail
TAG=${1:-r03}; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
declare -A CASES
CASES[cfg2]="bench.py --workload config2"
CASES[cfg2_sel001]="bench.py --workload config2 --selectivity 0.01"
CASES[cfg2_sel010]="bench.py --workload config2 --selectivity 0.10"
CASES[cfg2_sel050]="bench.py --workload config2 --selectivity 0.50"
CASES[cfg2_sel100]="bench.py --workload config2 --selectivity 1.00"
CASES[cfg2_null]="bench.py --workload config2 --null-pct 1"
CASES[cfg2_swapped]="bench.py --workload config2_swapped"
CASES[cfg3]="bench.py --workload config3"
CASES[cfg4]="bench.py --workload config4"
CASES[cfg2_pernode]="bench.py --workload config2 --exec-mode per_node"
CASES[q6agg]="tools/bench_q6.py"
CASES[groupby]="tools/bench_groupby.py 1000000000 1000,100000,1000000"
CASES[groupby_numeric]="tools/bench_groupby_numeric.py 1000000000 100000,300000"
ORDER="cfg2 cfg2_null cfg2_swapped cfg2_sel001 cfg2_sel010 cfg2_sel050 cfg2_sel100 cfg3 cfg4 cfg2_pernode q6agg groupby groupby_numeric"
[ $# -gt 0 ] && ORDER="$*"
cd /tmp
for C in $ORDER; do
    CMD=${CASES[$C]}
    [ -z "$CMD" ] && { echo "unknown case $C"; continue; }
    D=$OUT/$C; mkdir -p $D
    SCRIPT=$ROOT/${CMD%% *}; ARGS=${CMD#* }; [ "$ARGS" == "$CMD" ] && ARGS=""
    PLAIN=""; PROF=""
    if [[ $CMD == bench.py* ]]; then PLAIN="--steps 10 --warmup 8 --no-cpu-baseline"; PROF="--steps 5 --warmup 2 --profile-run"; fi
    echo "== $C: $CMD"
    timeout -k 10 300 python3 $SCRIPT $ARGS $PLAIN > $D/plain.out 2> $D/plain.err || { echo "$C: plain run failed"; tail -3 $D/plain.err; continue; }
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $SCRIPT $ARGS $PROF > $D/trace.out 2> $D/trace.err || echo "$C: kernel-trace pass failed"
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -- python3 $SCRIPT $ARGS $PROF > $D/pmc_fetch.out 2> $D/pmc_fetch.err || echo "$C: FETCH_SIZE pass failed"
    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -- python3 $SCRIPT $ARGS $PROF > $D/pmc_write.out 2> $D/pmc_write.err || echo "$C: WRITE_SIZE pass failed"
    tail -c 600 $D/plain.out
done
cd $ROOT
python3 tools/summarize_profile.py $OUT $TAG

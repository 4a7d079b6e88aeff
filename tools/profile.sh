#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats of the default bench command and
# the HBM traffic counters in separate passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit
# one pass).  Summaries land in gpurun_out/prof_<tag>/ ; copy the ones to keep into profiles/.
set -o pipefail
TAG=${1:-r01}
ROOT=$(pwd)
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace -- $BENCH > $ROOT/$OUT/trace_bench.json 2> $ROOT/$OUT/trace.err || echo "kernel-trace run failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/pmc_fetch -- $BENCH > $ROOT/$OUT/pmc_fetch_bench.json 2> $ROOT/$OUT/pmc_fetch.err || echo "pmc FETCH_SIZE run failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/$OUT/pmc_write -- $BENCH > $ROOT/$OUT/pmc_write_bench.json 2> $ROOT/$OUT/pmc_write.err || echo "pmc WRITE_SIZE run failed"
cd $ROOT
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt

#!/bin/bash
# SQ counter passes over the hash-partitioned GROUP BY (100 000 DOUBLE keys, 1 B rows): which unit the scatter and the
# aggregation passes wait on.  usage (GPU box): tools/hp_pmc.sh [keys] [rows]
KEYS=${1:-100000}; ROWS=${2:-1000000000}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp QE_HP_FROM=50000
D=gpurun_out/hp_pmc; mkdir -p $D
timeout -k 10 200 python3 tools/bench_groupby_numeric.py $ROWS $KEYS > $D/plain.out 2>&1 || exit 1
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_LDS_ATOMIC_RETURN" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_INSTS_LDS_ATOMIC"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $D/pass$i -- python3 tools/bench_groupby_numeric.py $ROWS $KEYS > $D/pass$i.out 2> $D/pass$i.err || { echo "pass $i failed"; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/hp_pmc/pass*/')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        if not k.startswith('qe_gb') and not k.startswith('qe_fused') and not k.startswith('qe_ht'): continue
        print(k, {c: (len(v), round(sum(v[-2:]) / len(v[-2:]))) for c, v in cs.items()})
PY

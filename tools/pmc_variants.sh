#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of ablation variants of the fused kernel (diagnostic).
set -o pipefail
ROOT=$(pwd); OUT=gpurun_out/pmc_variants; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for C in WRITE_SIZE FETCH_SIZE; do
rocprofv3 --pmc $C --output-format csv -d $ROOT/$OUT/$C -- python3 $ROOT/tools/sweep.py --rounds 1 --reps 2 --rows 1000000000 --variants "$1" > $ROOT/$OUT/$C.log 2>&1 || echo "$C failed"
done
cd $ROOT
python3 - <<'PY'
import csv,glob
for c in ("WRITE_SIZE","FETCH_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc_variants/{c}/**/*counter_collection.csv", recursive=True):
        rows=[r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("qe_fused") and r["Counter_Name"]==c]
        seen=[]
        for r in rows:
            key=r["Kernel_Id"]
            if not seen or seen[-1][0]!=key: seen.append([key,[]])
            seen[-1][1].append(float(r["Counter_Value"])*1024/1e9)
        for k,v in seen: print(c, "kernel_id",k, "n",len(v), "avg GB %.3f"%(sum(v)/len(v)))
PY

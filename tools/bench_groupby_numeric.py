#!/usr/bin/env python3
"""GROUP BY over a DOUBLE column with many distinct keys (Tripdata.kt:27-31's shape at scale): SELECT k, MIN(v), MAX(v) over 1 B
rows, per execution: which form ran and its kernel time.  usage: bench_groupby_numeric.py [rows] [keys,..] [tuning] [minmax|sumcount|avg]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from queryengine_amd import engine as E, native as N
from queryengine_amd.ast import ColumnExpression
from queryengine_amd.datatypes import DataType
from queryengine_amd.workloads import GenColumn

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
KEYS = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [20000, 100000, 1000000]
TUNING = [int(x) for x in sys.argv[3].split(',') if x] if len(sys.argv) > 3 else []
AGGS = {"minmax": ([N.AGG_MIN, N.AGG_MAX], "MIN(v), MAX(v)"), "sumcount": ([N.AGG_SUM, N.AGG_COUNT], "SUM(v), COUNT(v)"),
        "avg": ([N.AGG_AVG], "AVG(v)")}[sys.argv[4] if len(sys.argv) > 4 else "minmax"]
for nkeys in KEYS:
    ctx = E.Context(device=0, profile=True, tuning=TUNING)
    cols = [GenColumn("k", DataType.DOUBLE, N.GEN_F64_MOD, 0, modulus=nkeys), GenColumn("v", DataType.DOUBLE, N.GEN_F64_UNIT, 1)]
    b = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in cols], rows)
    k, v = ColumnExpression("k", 0, DataType.DOUBLE), ColumnExpression("v", 1, DataType.DOUBLE)
    args = ([ctx.compile(k)], [ctx.compile(v) for _ in AGGS[0]], AGGS[0])
    for rep in range(5):
        r = E.filter_groupby(ctx, b, None, *args); ng = r.count; r.free()
        ms = ctx.kernel_time()[0]
        print(f"GROUP BY DOUBLE k ({nkeys} distinct), {AGGS[1]}: rep {rep} form {ctx.last_form} groups {ng} kernel {ms:.3f} ms "
              f"= {rows * 16 / ms / 1e6 / 8000:.2f} of 8 TB/s on 16 B/row", flush=True)
    b.free(); ctx.close()

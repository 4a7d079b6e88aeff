#!/usr/bin/env python3
"""Group-by aggregation throughput: SELECT s, SUM(v), COUNT(v) FROM t [WHERE v < 0.5] over 1 B rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from queryengine_amd import engine as E, native as N, workloads as W
from queryengine_amd.ast import ColumnExpression, Function, FunctionExpression, NumericLiteralExpression
from queryengine_amd.datatypes import DataType

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
KEYS = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [10, 1000, 100000]
TUNING = [int(x) for x in sys.argv[3].split(',') if x] if len(sys.argv) > 3 else []
for nkeys in KEYS:
    wl = W.config4(rows, nkeys=nkeys)
    ctx = E.Context(device=0, profile=True, tuning=TUNING)
    b = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], rows)
    s, v = ColumnExpression("s", 0, DataType.STRING), ColumnExpression("v", 1, DataType.DOUBLE)
    flt = FunctionExpression(Function.CMP_LT, [v, NumericLiteralExpression(0.5)], DataType.BOOLEAN)
    for f in (None, flt):
        cf = ctx.compile(f) if f is not None else None
        args = ([ctx.compile(s)], [ctx.compile(v), ctx.compile(v)], [N.AGG_SUM, N.AGG_COUNT])
        r = E.filter_groupby(ctx, b, cf, *args); ng = r.count; r.free()
        ctx.reset_kernel_time()
        for _ in range(5):
            r = E.filter_groupby(ctx, b, cf, *args); r.free()
        _, tot, n = ctx.kernel_time()
        ms = tot / n
        print(f"GROUP BY s ({nkeys} keys){' WHERE v < 0.5' if f is not None else ''}: groups {ng} kernel {ms:.3f} ms "
              f"{rows * 12 / ms / 1e6:.0f} GB/s ({rows / ms / 1e6:.1f} G rows/s)", flush=True)
    b.free(); ctx.close()

# Tripdata.kt:27-31 shape: SELECT passenger_count, MIN(fare_amount), MAX(fare_amount) -- grouping by a DOUBLE column (hashed form)
from queryengine_amd.workloads import GenColumn
for nkeys in ([10, 100000] if len(sys.argv) <= 4 else [int(x) for x in sys.argv[4].split(',')]):
    ctx = E.Context(device=0, profile=True, tuning=TUNING)
    cols = [GenColumn("k", DataType.DOUBLE, N.GEN_F64_MOD, 0, modulus=nkeys), GenColumn("v", DataType.DOUBLE, N.GEN_F64_UNIT, 1)]
    b = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in cols], rows)
    k, v = ColumnExpression("k", 0, DataType.DOUBLE), ColumnExpression("v", 1, DataType.DOUBLE)
    args = ([ctx.compile(k)], [ctx.compile(v), ctx.compile(v)], [N.AGG_MIN, N.AGG_MAX])
    r = E.filter_groupby(ctx, b, None, *args); ng = r.count; r.free()
    ctx.reset_kernel_time()
    for _ in range(5):
        r = E.filter_groupby(ctx, b, None, *args); r.free()
    _, tot, n = ctx.kernel_time()
    ms = tot / n
    print(f"GROUP BY DOUBLE k ({nkeys} distinct values), MIN(v), MAX(v): groups {ng} kernel {ms:.3f} ms "
          f"{rows * 16 / ms / 1e6:.0f} GB/s = {rows * 16 / ms / 1e6 / 8000:.2f} of 8 TB/s ({rows / ms / 1e6:.1f} G rows/s)", flush=True)
    b.free(); ctx.close()

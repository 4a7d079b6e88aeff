#!/usr/bin/env python3
"""Wall time of the FIRST executions of a GROUP BY over a DOUBLE column with many keys (the plan knows nothing about its keys yet):
usage: hp_first.py [rows] [keys,..]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from queryengine_amd import engine as E, native as N
from queryengine_amd.ast import ColumnExpression
from queryengine_amd.datatypes import DataType
from queryengine_amd.workloads import GenColumn

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
for nkeys in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "20000,100000,1000000").split(',')]:
    ctx = E.Context(device=0, profile=True)
    cols = [GenColumn("k", DataType.DOUBLE, N.GEN_F64_MOD, 0, modulus=nkeys), GenColumn("v", DataType.DOUBLE, N.GEN_F64_UNIT, 1)]
    b = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in cols], rows)
    k, v = ColumnExpression("k", 0, DataType.DOUBLE), ColumnExpression("v", 1, DataType.DOUBLE)
    args = ([ctx.compile(k)], [ctx.compile(v), ctx.compile(v)], [N.AGG_MIN, N.AGG_MAX])
    E.prepare_groupby(ctx, b, None, *args)          # the JIT of the plan itself is not what is timed (forms it switches to still compile)
    for rep in range(3):
        ctx.synchronize()
        t0 = time.perf_counter()
        r = E.filter_groupby(ctx, b, None, *args)
        ctx.synchronize()
        t1 = time.perf_counter()
        ng = r.count; r.free()
        print(f"{nkeys} keys: execution {rep}: {1e3 * (t1 - t0):9.1f} ms wall, form {ctx.last_form}, {ng} groups, last bracket {ctx.kernel_time()[0]:.2f} ms", flush=True)
    b.free(); ctx.close()

#!/usr/bin/env python3
"""TPC-H Q6 shape as ONE fused filter -> multiply -> SUM kernel: config 3's predicate with
SUM(l_extendedprice * l_discount) instead of a projected column -- no output column at all."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from queryengine_amd import engine as E, native as N, workloads as W

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 600_037_902
wl = W.config3(rows)
TUNING = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else []
ctx = E.Context(device=0, profile=True, tuning=TUNING)
b = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], rows)
cf, ce = ctx.compile(wl.filter), [ctx.compile(p) for p in wl.projections]
vals, nsel = E.filter_aggregate(ctx, b, cf, ce, [N.AGG_SUM])
ctx.reset_kernel_time()
for _ in range(10):
    E.filter_aggregate(ctx, b, cf, ce, [N.AGG_SUM])
_, tot, n = ctx.kernel_time()
ms = tot / n
print(f"Q6 shape (tuning {TUNING}), {rows} rows: SUM = {vals[0]!r} over {nsel} rows; kernel {ms:.3f} ms, {rows * 28 / ms / 1e6:.0f} GB/s algorithmic "
      f"({rows * 28 / ms / 1e6 / 8000:.2f} of 8 TB/s), {rows / ms / 1e6:.1f} G rows/s", flush=True)

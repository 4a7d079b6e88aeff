#!/usr/bin/env python3
"""config 2 with nullable inputs (1 % nulls per column: validity bitmaps are read, nullable outputs are packed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from queryengine_amd import engine as E, workloads as W
from sel_probe import timed

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
for tuning in ([], [256, 6], [256, 4]):
    ctx = E.Context(device=0, profile=True, tuning=tuning)
    for null_pct in (0, 1):
        wl = W.config2(rows, null_pct=null_pct)
        b = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], rows)
        ms, mn, nout = timed(ctx, b, ctx.compile(wl.filter), [ctx.compile(p) for p in wl.projections], 7)
        print(f"tuning {tuning} null_pct {null_pct}: nout {nout} kernel {ms:.3f} ms (min {mn:.3f})", flush=True)
        b.free()
    ctx.close()

#!/usr/bin/env python3
"""Selectivity sweep of config2 next to its ceilings: the same projections WITHOUT a filter (24 B read + 16 B written per
row at the row index, no compaction, no look-back) and the pure read stream.  One process, one GPU."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from queryengine_amd import engine as E
from queryengine_amd import workloads as W


def timed(ctx, b, cf, cp, reps):
    E.prepare(ctx, b, cf, cp)
    r = E.filter_project(ctx, b, cf, cp); nout = r.count; r.free()
    ts = []
    for _ in range(reps):
        r = E.filter_project(ctx, b, cf, cp); r.free()
        ts.append(ctx.kernel_time()[0])
    ts.sort()
    return ts[len(ts) // 2], ts[0], nout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--sels", default="0.01,0.05,0.1,0.25,0.5,1.0")
    ap.add_argument("--tuning", default="")
    args = ap.parse_args()
    ctx = E.Context(device=0, profile=True, tuning=[int(x) for x in args.tuning.split(",") if x])
    print("stream_read_gbps", ctx.stream_read_bandwidth(8 << 30, 5), flush=True)
    wl = W.config2(args.rows)
    b = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], args.rows)
    cp = [ctx.compile(p) for p in wl.projections]
    ms, mn, nout = timed(ctx, b, None, cp, args.reps)
    print(f"no filter: {ms:.3f} ms (min {mn:.3f})  {(24 + 16) * args.rows / ms / 1e6:.0f} GB/s", flush=True)
    for s in [float(x) for x in args.sels.split(",")]:
        c_limit = 0.5 if s <= 0.5 else 1.0
        w = W.config2(args.rows, a_limit=round(1000 * s / c_limit), c_limit=c_limit)
        ms, mn, nout = timed(ctx, b, ctx.compile(w.filter), cp, args.reps)
        gb = w.algorithmic_bytes(args.rows, nout)
        print(f"sel {s}: nout {nout}  {ms:.3f} ms (min {mn:.3f})  {gb / ms / 1e6:.0f} GB/s  frac {gb / ms / 1e6 / 8000:.3f}", flush=True)


if __name__ == "__main__":
    main()

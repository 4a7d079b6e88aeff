#!/usr/bin/env python3
"""Kernel tuning sweep on one GPU: times the fused filter+project kernel of a workload under
different geometry knobs (interleaved rounds in ONE process), plus the filter+COUNT aggregate
kernel (same loads and predicate, no compaction) and the plain streaming-read kernel."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from queryengine_amd import engine as E
from queryengine_amd import native as N
from queryengine_amd import workloads as W


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--reps", type=int, default=9)
    ap.add_argument("--variants", default="256,4,0,0,32;256,8,0,0,32;256,2,0,0,32;256,4,0,8,32;256,4,0,0,128;256,4,0,0,8;256,4,2,0,32")
    args = ap.parse_args()
    wl = W.WORKLOADS[args.workload](args.rows)
    variants = [tuple(int(x) for x in v.split(",")) for v in args.variants.split(";") if v]
    ctxs = []
    base = E.Context(device=0, profile=True)
    print("stream_read_gbps", base.stream_read_bandwidth(8 << 30, 5), flush=True)
    batch = E.DeviceBatch.generate(base, [c.spec(base) for c in wl.columns], args.rows)
    import ctypes as C
    # other contexts share the columns zero-copy (qe_batch_wrap_device)
    views = []
    for j in range(batch.ncols):
        pass
    results = {}
    for v in variants:
        ctx = E.Context(device=0, profile=True, tuning=list(v))
        b = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], args.rows) if False else None
        ctxs.append((v, ctx))
    # one shared batch per context would need 24 GB each: instead run variants sequentially on `base` data by
    # re-creating the context options is not possible, so generate per context lazily and free after timing
    for rnd in range(args.rounds):
        for v, ctx in ctxs:
            b = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], args.rows)
            cf = ctx.compile(wl.filter)
            cp = [ctx.compile(p) for p in wl.projections]
            E.prepare(ctx, b, cf, cp)
            r = E.filter_project(ctx, b, cf, cp); nout = r.count; r.free()
            times = []
            for _ in range(args.reps):
                r = E.filter_project(ctx, b, cf, cp); r.free()
                times.append(ctx.kernel_time()[0])
            times.sort()
            ms = times[len(times) // 2]
            gbps = wl.algorithmic_bytes(args.rows, nout) / (ms * 1e-3) / 1e9
            results.setdefault(v, []).append(ms)
            print(f"round {rnd} variant threads,unroll,nt,bpc,spc={v}: kernel median {ms:.3f} ms (min {times[0]:.3f})  {gbps:.0f} GB/s  nout {nout}", flush=True)
            if rnd == 0 and v == ctxs[0][0]:
                ctx.reset_kernel_time()
                for _ in range(args.reps):
                    E.filter_aggregate(ctx, b, cf, [cp[0]], [N.AGG_COUNT])
                _, tot, n = ctx.kernel_time()
                print(f"   filter+COUNT aggregate kernel (reads only the predicate columns): {tot / n:.3f} ms", flush=True)
                ctx.reset_kernel_time()
                for _ in range(args.reps):
                    E.filter_aggregate(ctx, b, cf, cp, [N.AGG_SUM] * len(cp))
                _, tot, n = ctx.kernel_time()
                print(f"   filter+SUM(every projection) aggregate kernel (all columns, no compaction): {tot / n:.3f} ms  "
                      f"{args.rows * wl.read_bytes_per_row() / (tot / n * 1e-3) / 1e9:.0f} GB/s", flush=True)
            b.free()
            ctx.trim()
    print(json.dumps({",".join(map(str, k)): min(v) for k, v in results.items()}))


if __name__ == "__main__":
    main()

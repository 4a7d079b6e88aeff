#!/usr/bin/env python3
"""Interleaved timing of kernel variants (qe_options.tuning lists) of ONE workload on one GPU: every variant gets its own
context (own plans) over ONE shared device batch (qe_batch_wrap_device would need plumbing: the batch is generated per
context only when the columns fit twice, else variants run one after the other).  Prints kernel-time median / min per variant
and round, plus the filter+COUNT aggregate (predicate columns only, no compaction) as the floor of the filter stage.
usage: exp_variants.py <workload> <rows> "name=t0,t1,..;name=..." [reps] [rounds]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from queryengine_amd import engine as E
from queryengine_amd import native as N
from queryengine_amd import workloads as W


def main():
    wname, rows = sys.argv[1], int(sys.argv[2])
    variants = []
    for item in sys.argv[3].split(";"):
        if not item:
            continue
        name, _, t = item.partition("=")
        variants.append((name, [int(x) for x in t.split(",") if x]))
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 7
    rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 2
    mk = {"config2": W.config2, "config3": W.config3, "config4": W.config4, "config2n": lambda n: W.config2(n, null_pct=1),
          "config2r": None}[wname]
    wl = mk(rows)
    base = E.Context(device=0, profile=True)
    print("stream_read_gbps", round(base.stream_read_bandwidth(8 << 30, 5)), flush=True)
    batch = E.DeviceBatch.generate(base, [c.spec(base) for c in wl.columns], rows)
    res = {}
    ctxs = [(name, t, E.Context(device=0, profile=True, tuning=t)) for name, t in variants]
    for rnd in range(rounds):
        for name, t, ctx in ctxs:
            # the batch belongs to `base`; a batch handle is only a set of device pointers, so another context of the same
            # device may read it (contexts differ in stream, pool and plan cache)
            cf = ctx.compile(wl.filter)
            cp = [ctx.compile(p) for p in wl.projections]
            try:
                E.prepare(ctx, batch, cf, cp)
                for _ in range(2):
                    r = E.filter_project(ctx, batch, cf, cp); nout = r.count; r.free()
                ts = []
                for _ in range(reps):
                    r = E.filter_project(ctx, batch, cf, cp); r.free()
                    ts.append(ctx.kernel_time()[0])
            except N.QeError as exc:
                print(f"round {rnd} {name:28s} {t}: FAILED {exc}", flush=True)
                continue
            ts.sort()
            res.setdefault(name, []).append(ts[len(ts) // 2])
            print(f"round {rnd} {name:28s} {t}: median {ts[len(ts)//2]:.3f} min {ts[0]:.3f} ms  form {ctx.last_form} nout {nout}", flush=True)
            if rnd == 0 and name == ctxs[0][0]:
                ctx.reset_kernel_time()
                for _ in range(reps):
                    E.filter_aggregate(ctx, batch, cf, [cp[0]], [N.AGG_COUNT])
                _, tot, n = ctx.kernel_time()
                print(f"   filter+COUNT aggregate kernel: {tot / n:.3f} ms", flush=True)
    print(json.dumps({k: min(v) for k, v in res.items()}))


if __name__ == "__main__":
    main()

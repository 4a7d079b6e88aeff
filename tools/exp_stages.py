#!/usr/bin/env python3
"""What every load stage of cfg 3 costs: the filter's AND chain cut after 1, 2, 3 columns (+ the projection's column), timed
as the fused filter+COUNT aggregate (no output, no ordering) and as the ring kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from queryengine_amd import engine as E
from queryengine_amd import native as N
from queryengine_amd import workloads as W
from queryengine_amd.ast import Function as Fn, FunctionExpression as FE, NumericLiteralExpression as Num
from queryengine_amd.datatypes import DataType as T


def conj(wl, k):
    out = []

    def split(e):
        if isinstance(e, FE) and e.function == Fn.AND:
            split(e.operands[0]); split(e.operands[1])
        else:
            out.append(e)
    split(wl.filter)
    e = out[0]
    for c in out[1:k]:
        e = FE(Fn.AND, [e, c], T.BOOLEAN)
    return e


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 600_037_902
    tuning = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else []
    wl = W.config3(rows)
    ctx = E.Context(device=0, profile=True, tuning=tuning)
    print("stream_read_gbps", round(ctx.stream_read_bandwidth(8 << 30, 5)))
    batch = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], rows)
    from queryengine_amd.ast import ColumnExpression as Col
    sh, di, qt, pr = [Col(c.name, i, c.type) for i, c in enumerate(wl.columns)]
    cases = [("sh only -> sh", conj(wl, 2), [sh]), ("sh,di -> di", conj(wl, 4), [di]), ("sh,di,qt -> qt", conj(wl, 5), [qt]),
             ("sh,di,qt -> pr*di (cfg 3)", conj(wl, 5), wl.projections)]
    for name, flt, projs in cases:
        cf, cp = ctx.compile(flt), [ctx.compile(p) for p in projs]
        for _ in range(3):
            r = E.filter_project(ctx, batch, cf, cp); nout = r.count; r.free()
        ts = []
        for _ in range(7):
            r = E.filter_project(ctx, batch, cf, cp); r.free()
            ts.append(ctx.kernel_time()[0])
        ts.sort()
        ctx.reset_kernel_time()
        for _ in range(7):
            E.filter_aggregate(ctx, batch, cf, [cp[0]], [N.AGG_COUNT])
        _, tot, n = ctx.kernel_time()
        ctx.reset_kernel_time()
        for _ in range(7):
            E.filter_aggregate(ctx, batch, cf, cp, [N.AGG_SUM] * len(cp))
        _, tot2, n2 = ctx.kernel_time()
        print(f"{name:28s} kept {nout / rows:.4f}  ring kernel {ts[len(ts) // 2]:.3f} ms  filter+COUNT {tot / n:.3f} ms  filter+SUM(proj) {tot2 / n2:.3f} ms", flush=True)


if __name__ == "__main__":
    main()

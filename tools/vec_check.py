import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from queryengine_amd import engine as E, workloads as W
for wl in (W.config2(3_000_017), W.config2(3_000_017, null_pct=1), W.config3(3_000_017)):
    outs = []
    for t in ([0], [0,0,5]):
        ctx = E.Context(device=0, tuning=t)
        b = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], wl.default_rows)
        cf = ctx.compile(wl.filter); cp=[ctx.compile(p) for p in wl.projections]
        for _ in range(3):
            r = E.filter_project(ctx, b, cf, cp)
            cols = r.to_columns(); form = ctx.last_form
            r.free()
        outs.append(cols); print(wl.name, t, form, len(cols[0].data))
    for a, b_ in zip(*outs):
        assert (a.valid is None) == (b_.valid is None)
        if a.valid is not None:
            assert np.array_equal(a.valid, b_.valid)
            assert np.array_equal(a.data[a.valid].view(np.uint8), b_.data[b_.valid].view(np.uint8))
        else:
            assert np.array_equal(a.data.view(np.uint8), b_.data.view(np.uint8))
print("VEC PARITY OK")

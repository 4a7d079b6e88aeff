#!/usr/bin/env python3
"""Diagnostic: per-tile timeline of the dense single-pass kernel (debug bit 32 of tuning[5]): ticket -> count published ->
resolved -> stored, in 10 ns ticks of s_memrealtime.  Shows where a tile's time goes and how long it had to wait for the
slowest earlier tile (the inherent part of the look-back wait)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from queryengine_amd import engine as E, workloads as W

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
sel = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
extra = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tun = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0, 0, 0, 0, 0]
tun = (tun + [0] * 8)[:8]
tun[5] |= 16384 | 32 | extra
path = "/tmp/qe_dense_trace.bin"
os.environ["QE_TRACE_FILE"] = path
ctx = E.Context(device=0, profile=True, tuning=tun)
c_limit = 0.5 if sel <= 0.5 else 1.0
wl = W.config2(rows, a_limit=round(1000 * sel / c_limit), c_limit=c_limit)
batch = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], rows)
cf, cp = ctx.compile(wl.filter), [ctx.compile(p) for p in wl.projections]
for _ in range(3):
    r = E.filter_project(ctx, batch, cf, cp); r.free()
print(f"kernel {ctx.kernel_time()[0]:.3f} ms (traced build), tuning {tun}")
t = np.fromfile(path, dtype=np.uint64).reshape(-1, 4).astype(np.int64)
t0 = t[:, 0].min()
tk, pub, res, done = (t[:, i] - t0 for i in range(4))
us = 0.01


def pct(name, x):
    q = np.percentile(x, [50, 90, 99, 100]) * us
    print(f"  {name:44s} mean {x.mean() * us:8.2f} us  p50 {q[0]:8.2f}  p90 {q[1]:8.2f}  p99 {q[2]:8.2f}  max {q[3]:8.2f}")


print(f"tiles {len(t)}, span {(done.max()) * us / 1000:.3f} ms")
pct("ticket -> count published (hold)", pub - tk)
pct("published -> resolved (look-back wait)", res - pub)
pct("resolved -> stored", done - res)
pct("ticket -> stored (a workgroup's cycle)", done - tk)
runmax = np.maximum.accumulate(pub)
prev = np.concatenate([[0], runmax[:-1]])
pct("slowest EARLIER tile published after mine by", np.maximum(prev - pub, 0))
pct("resolved later than that (poll + sleep)", res - np.maximum(prev, pub))
order = np.argsort(tk, kind="stable")
print("  ticket order == tile order:", bool(np.all(np.diff(tk) >= -200)))
# throughput over time
edges = np.linspace(0, done.max(), 11)
h, _ = np.histogram(done, edges)
print("  tiles stored per tenth of the run:", h.tolist())
batch.free(); ctx.close()

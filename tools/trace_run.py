#!/usr/bin/env python3
"""Diagnostic: per-chunk timeline of the fused kernel (ticket / streamed / resolved timestamps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from queryengine_amd import engine as E, workloads as W

spc = int(sys.argv[1]) if len(sys.argv) > 1 else 16
prio = int(sys.argv[3]) if len(sys.argv) > 3 else 0
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/trace.bin"
os.environ["QE_TRACE_FILE"] = out
wl = W.config2(1_000_000_000)
ctx = E.Context(device=0, profile=True, tuning=[256, 8, prio, 300, spc, 32, 1])
b = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], wl.default_rows)
cf = ctx.compile(wl.filter); cp = [ctx.compile(p) for p in wl.projections]
for _ in range(3):
    r = E.filter_project(ctx, b, cf, cp); r.free()
print("kernel ms", ctx.kernel_time()[0])
t = np.fromfile(out, dtype=np.uint64).reshape(-1, 4)
t0 = t[:, 0].min()
tick = (t[:, 0] - t0) * 0.01   # us (100 MHz)
streamed = (t[:, 1] - t0) * 0.01
resolved = (t[:, 2] - t0) * 0.01
wslot = (t[:, 3] >> np.uint64(40)).astype(np.int64)
xcc = ((t[:, 3] >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
hwid = (t[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
dur = streamed - tick
print("chunks", len(t), "span us", resolved.max())
print("chunk duration us: mean %.1f std %.1f min %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (dur.mean(), dur.std(), dur.min(), *np.percentile(dur, [50, 90, 99]), dur.max()))
for x in range(8):
    m = xcc == x
    if m.any():
        print("  xcc", x, "chunks", m.sum(), "mean dur %.1f" % dur[m].mean())
wv = wslot % 4
for w in range(4):
    print("  wave-in-WG", w, "mean dur %.1f" % dur[wv == w].mean())
pred_done = np.maximum.accumulate(streamed)
pred_done = np.concatenate([[0], pred_done[:-1]])
lag = pred_done - streamed
print("lag (all predecessors streamed - own streamed) us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f frac>0 %.2f" % (lag.mean(), *np.percentile(lag, [50, 90, 99]), lag.max(), (lag > 0).mean()))
rl = resolved - streamed
print("resolve latency us: mean %.1f p50 %.1f p90 %.1f p99 %.1f" % (rl.mean(), *np.percentile(rl, [50, 90, 99])))
print("ticket monotonic violations:", int((np.diff(tick) < 0).sum()))
idx = np.argsort(-dur)[:10]
for i in idx:
    print("  slow chunk", i, "dur %.1f" % dur[i], "xcc", xcc[i], "wslot", wslot[i], "tick %.1f" % tick[i])
# per wave slot: mean duration distribution
import collections
per = collections.defaultdict(list)
for d, w in zip(dur, wslot): per[w].append(d)
means = np.array([np.mean(v) for v in per.values()])
print("per-wave mean duration: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f (n waves %d)" % (means.min(), *np.percentile(means, [10, 50, 90]), means.max(), len(means)))

# hardware placement: HW_ID bits (gfx9 layout: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13] ...)
simd = (hwid >> 4) & 3
cu = (hwid >> 8) & 15
sh = (hwid >> 12) & 1
se = (hwid >> 13) & 7
print("hwid sample", [hex(int(h)) for h in hwid[:6]])
for name, arr, n in (("simd", simd, 4), ("cu", cu, 16), ("sh", sh, 2), ("se", se, 8)):
    print(" ", name, " ".join("%d:%.0f(%d)" % (v, dur[arr == v].mean(), (arr == v).sum()) for v in range(n) if (arr == v).any()))
# per physical CU (xcc, se, sh, cu): mean duration and number of distinct wave slots
key = xcc * 4096 + se * 256 + sh * 16 + cu
ukeys = np.unique(key)
rows = []
for k in ukeys:
    m = key == k
    rows.append((dur[m].mean(), len(np.unique(wslot[m])), m.sum(), int(k)))
rows.sort()
print("physical CUs seen:", len(rows))
print("fastest CUs (mean dur, nwaves, chunks, key):", [(round(a, 1), b, c, hex(d)) for a, b, c, d in rows[:6]])
print("slowest CUs:", [(round(a, 1), b, c, hex(d)) for a, b, c, d in rows[-6:]])
nw = np.array([r[1] for r in rows]); md = np.array([r[0] for r in rows])
for n in np.unique(nw):
    print("  CUs with", n, "resident waves:", (nw == n).sum(), "mean chunk dur %.1f" % md[nw == n].mean())

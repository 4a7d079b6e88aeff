#!/usr/bin/env python3
"""Selectivity sweep of config 2 (1 B rows) through the three forms of the fused executor: the LDS-ring single pass
(default at low selectivity), the two-pass form (count -> scan -> ordered write; round 1's high-selectivity form) and the
dense single pass (round 2: chunk == sub-tile, blocking look-back, direct ordered stores).  One context per form on the
same generated data; kernel time = HIP events (qe_ctx_kernel_time), median of --reps.  Prints a table and one JSON line."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from queryengine_amd import engine as E, workloads as W


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--sels", default="0.01,0.05,0.1,0.25,0.5,0.75,1.0")
    ap.add_argument("--forms", default="ring,two_pass,dense")
    ap.add_argument("--tuning", default="", help="extra tuning slots 0..4 for every form, e.g. 512,8")
    args = ap.parse_args()
    base = [int(x) for x in args.tuning.split(",") if x] + [0] * 8
    bits = {"ring": 1024 | 32768 | 8192, "ring_choice": 1024 | 32768, "two_pass": 512, "dense": 16384}
    out = {}
    for form in args.forms.split(","):
        t = list(base[:8])
        t[5] |= bits[form]
        ctx = E.Context(device=0, profile=True, tuning=t)
        wl0 = W.config2(args.rows)
        batch = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl0.columns], args.rows)
        for sel in [float(x) for x in args.sels.split(",")]:
            c_limit = 0.5 if sel <= 0.5 else 1.0
            wl = W.config2(args.rows, a_limit=round(1000 * sel / c_limit), c_limit=c_limit)
            cf, cp = ctx.compile(wl.filter), [ctx.compile(p) for p in wl.projections]
            E.prepare(ctx, batch, cf, cp)
            nout = 0
            for _ in range(8 if form == "ring_choice" else 2):
                r = E.filter_project(ctx, batch, cf, cp); nout = r.count; r.free()
            times = []
            for _ in range(args.reps):
                r = E.filter_project(ctx, batch, cf, cp); r.free()
                times.append(ctx.kernel_time()[0])
            times.sort()
            ms = times[len(times) // 2]
            alg = wl.algorithmic_bytes(args.rows, nout)
            out.setdefault(form, {})[sel] = {"ms": ms, "min_ms": times[0], "frac": alg / (ms * 1e-3) / 8e12, "nout": nout}
            print(f"{form:12s} sel {sel:5.2f} (kept {nout / args.rows:.4f}): kernel median {ms:7.3f} ms (min {times[0]:7.3f})  "
                  f"{alg / (ms * 1e-3) / 1e9:6.0f} GB/s algorithmic = {alg / (ms * 1e-3) / 8e12:.3f} of 8 TB/s", flush=True)
        batch.free()
        ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

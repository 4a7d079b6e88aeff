#!/usr/bin/env python3
"""PCIe-inclusive cost of the boundary: qe_batch_create from pageable host buffers ("pin to HBM
once") and qe_result_column_to_host of a result, timed end to end on the host clock."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from queryengine_amd import engine as E
from queryengine_amd import workloads as W
from queryengine_amd.table import Column

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ctx = E.Context(device=0)
wl = W.config2(n)
rng = np.random.default_rng(1)
cols = [Column(wl.columns[0].type, rng.integers(0, 1000, n, dtype=np.int64), None),
        Column(wl.columns[1].type, rng.integers(0, 2 ** 31, n, dtype=np.int64), None),
        Column(wl.columns[2].type, rng.random(n), None)]
nbytes = 24 * n
for rep in range(3):
    t0 = time.perf_counter()
    b = E.DeviceBatch.from_columns(ctx, cols)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(f"qe_batch_create {nbytes / 1e9:.1f} GB from pageable host memory: {dt * 1e3:.1f} ms = {nbytes / dt / 1e9:.1f} GB/s", flush=True)
    if rep < 2:
        b.free()
cf, cp = ctx.compile(wl.filter), [ctx.compile(p) for p in wl.projections]
E.prepare(ctx, b, cf, cp)
for rep in range(3):
    t0 = time.perf_counter()
    r = E.filter_project(ctx, b, cf, cp)
    t1 = time.perf_counter()
    host = r.to_columns()
    t2 = time.perf_counter()
    out_bytes = r.count * 16
    print(f"filter_project {n} rows: {1e3 * (t1 - t0):.2f} ms; result {r.count} rows to host {1e3 * (t2 - t1):.1f} ms = "
          f"{out_bytes / (t2 - t1) / 1e9:.1f} GB/s; end to end incl. H2D would be {1e3 * (dt + t2 - t0):.0f} ms "
          f"= {n / (dt + t2 - t0) / 1e9:.2f} G rows/s", flush=True)
    r.free()

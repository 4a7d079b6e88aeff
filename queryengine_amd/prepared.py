"""Plan-time preparation of the BASELINE workloads: generate + hiprtc-compile their fused kernels
into the JIT cache without a GPU (the analogue of building the physical plan ahead of execution)."""
from __future__ import annotations

import numpy as np

from . import engine as E
from . import workloads as W
from .table import Column

_NP = {8: None}


def _schema_columns(wl):
    cols = []
    for c in wl.columns:
        npdt = np.float64 if c.type.name == "DOUBLE" else (np.int64 if c.type.name == "INT64" else np.int32)
        valid = np.array([False, True]) if c.null_pct else None
        cols.append(Column(c.type, np.zeros(2, dtype=npdt), valid, c.dictionary))
    return cols


def prewarm(verbose: bool = False) -> int:
    ctx = E.Context(device=None)
    n = 0
    for wl in (W.config1(), W.config2(), W.config2(null_pct=1), W.config3(), W.config4()):
        batch = E.DeviceBatch.describe(ctx, _schema_columns(wl))
        cf = ctx.compile(wl.filter) if wl.filter is not None else None
        cp = [ctx.compile(p) for p in wl.projections]
        E.prepare(ctx, batch, cf, cp)
        n += 1
        if verbose:
            print("prepared", wl.name)
    ctx.close()
    return n

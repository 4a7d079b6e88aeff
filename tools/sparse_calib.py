#!/usr/bin/env python3
"""What does a sparse streaming read cost?  (late materialisation: load a column only where the predicate so far holds)
Run once per percentage: QE_CALIB_SPARSE_PCT is read when the library first launches the read kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from queryengine_amd import engine as E
ctx = E.Context(device=0)
gbps = ctx.stream_read_bandwidth(8 << 30, 7)
print(f"pct {os.environ.get('QE_CALIB_SPARSE_PCT', 'dense')}: 8 GiB in {(8 << 30) / gbps / 1e6:.3f} ms  ({gbps:.0f} GB/s dense-equivalent)", flush=True)

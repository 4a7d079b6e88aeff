#!/usr/bin/env python3
"""What does a trickle of writes cost a 24 GB read stream? (roofline calibration)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from queryengine_amd import engine as E
ctx = E.Context(device=0)
n = 24 << 30
for we in (0, 16, 2):
    ms, wb = ctx.stream_read_write_time(n, we, 5)
    print(f"write_every {we:3d} (no gating): {ms:.3f} ms  read {n/1e9:.1f} GB  written {wb/1e9:.3f} GB  -> {(n+wb)/ms/1e6:.0f} GB/s total")
# gate the writes into device-wide windows (period / length in 10 ns ticks of the shared 100 MHz clock)
for period_us, len_us in ((200, 20), (1000, 50)):
    for we in (2,):
        code = we + 1000 * period_us + 10000000 * len_us
        ms, wb = ctx.stream_read_write_time(n, code, 5)
        print(f"write_every {we} window {len_us} us every {period_us} us: {ms:.3f} ms  written {wb/1e9:.3f} GB")

# same written bytes, longer contiguous runs per write event (QE_CALIB_BLOCKS x 512 B every QE_CALIB_BLOCKS-th time)
for blocks in (1, 4, 16, 64):
    os.environ["QE_CALIB_BLOCKS"] = str(blocks)
    ms, wb = ctx.stream_read_write_time(n, 2, 5)
    print(f"write 0.8 GB as runs of {blocks * 512} B per wave: {ms:.3f} ms")

os.environ["QE_CALIB_BLOCKS"] = "1"
for kind, name in ((0, "plain"), (1, "nontemporal"), (2, "sc1 (agent-scope atomic store)"), (3, "system-scope atomic store")):
    os.environ["QE_CALIB_STORE"] = str(kind)
    ms, wb = ctx.stream_read_write_time(n, 2, 5)
    print(f"0.8 GB of 512 B {name} stores: {ms:.3f} ms")

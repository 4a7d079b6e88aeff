#!/usr/bin/env python3
"""Try the C ABI's exchange step (qe_comm_init / qe_gather, RCCL direct) with TWO ranks that share ONE GPU: two processes,
rank 0's ncclUniqueId handed over through a file.  RCCL may refuse two ranks on one device; then this prints the refusal and
exits 0 -- the multi-rank placement code is covered by tests/test_gpu_distributed.py::test_result_concat_equals_one_pass, the
rank / offset logic by the world-size-2 gloo tests.  When it does run, rank 0 checks the gathered result against a
single-process pass over the whole table, bit for bit."""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(rank, world, idfile, q):
    import numpy as np
    from queryengine_amd import engine as E, native as N, workloads as W
    from queryengine_amd.distributed import shard_range
    try:
        ctx = E.Context(device=0)
        n = 3_000_017
        wl = W.config2(n, null_pct=1)
        if rank == 0:
            uid = ctx.comm_unique_id()
            with open(idfile + ".tmp", "wb") as f:
                f.write(uid)
            os.rename(idfile + ".tmp", idfile)
        else:
            for _ in range(600):
                if os.path.exists(idfile):
                    break
                time.sleep(0.05)
            uid = open(idfile, "rb").read()
        ctx.comm_init(world, rank, uid)
        begin, end = shard_range(n, rank, world)
        batch = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], end - begin, row_begin=begin)
        cf, cp = ctx.compile(wl.filter), [ctx.compile(p) for p in wl.projections]
        res = E.filter_project(ctx, batch, cf, cp)
        g = ctx.gather(res, 0)
        ok = None
        if rank == 0:
            whole = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], n, row_begin=0)
            want = E.filter_project(ctx, whole, cf, cp)
            ok = g.count == want.count
            for a, b in zip(g.to_columns(), want.to_columns()):
                av = a.valid if a.valid is not None else np.ones(len(a), bool)
                bv = b.valid if b.valid is not None else np.ones(len(b), bool)
                ok = ok and np.array_equal(av, bv) and np.array_equal(a.data[av].view(np.uint64), b.data[bv].view(np.uint64))
            parts = ctx.allgather_host(bytes([rank]) * 8)
            ok = ok and parts == [bytes([0]) * 8, bytes([1]) * 8]
        else:
            ctx.allgather_host(bytes([rank]) * 8)
        q.put((rank, "ok" if ok or rank != 0 else "MISMATCH", res.count))
        ctx.comm_destroy()
        ctx.close()
    except Exception as exc:   # noqa: BLE001
        q.put((rank, f"{type(exc).__name__}: {exc}", -1))


if __name__ == "__main__":
    mp.set_start_method("spawn")
    idfile = f"/tmp/qe_uid_{os.getpid()}"
    q = mp.Queue()
    ps = [mp.Process(target=worker, args=(r, 2, idfile, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = []
    try:
        for _ in range(2):
            out.append(q.get(timeout=120))
    except Exception:
        out.append(("?", "timeout", -1))
    for p in ps:
        p.join(10)
        if p.is_alive():
            p.kill()
    print(sorted(out, key=str))

#!/usr/bin/env python3
"""bench.py -- filter+project throughput of the fused HIP path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path -- qe_filter_project through the C ABI -- over one
HBM-resident synthetic batch (config 2: SELECT a + b, c * 2.0 FROM t WHERE a < 100 AND c < 0.5,
int64/int64/f64, 1 B rows per GPU).  Inputs are generated on the device before the timed region.
Multi-GPU: one process per GPU, rows range-sharded by global row index, no data-path collective
during the scan (weak scaling: each rank owns --rows rows).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 GB/s achievable


def host_cpu_share() -> int:
    """CPUs this process may really use: scheduler affinity capped by the cgroup CPU quota (a GPU box gives one GPU's job a
    share of the host, not all of its hardware threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, -(-int(txt[0]) // int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, -(-q // per)))
            break
        except Exception:
            continue
    return max(1, min(n, int(os.environ.get("QE_BENCH_CPU_THREADS", "16"))))


def cpu_baseline(workload, budget_s=12.0):
    """The oracle (a C port of the reference's row-at-a-time evaluator) timed on one host core
    over a bounded sample of the same workload."""
    import numpy as np
    from oracle import qe_oracle as O
    from queryengine_amd.table import Column
    O.build()

    def sample(n):
        cols = []
        for c in workload.columns:
            s = O.GenSpec()
            s.kind, s.col_id, s.modulus, s.offset, s.step = c.kind if c.kind != 6 else 1, c.col_id, c.modulus, c.offset, c.step
            s.aux_col_id, s.null_pct = c.aux_col_id, c.null_pct
            npdt = {8: np.float64 if c.type.name == "DOUBLE" else np.int64, 4: np.int32}[c.width]
            data, valid = O.generate(s, 42, 0, n, npdt)
            cols.append(Column(c.type, data, valid, c.dictionary))
        return cols

    n = 1_000_000
    cols = sample(n)
    t0 = time.perf_counter()
    O.filter_project(cols, workload.filter, workload.projections, O.BYTECODE_COMPILER)
    dt = time.perf_counter() - t0
    # scale the sample so one pass is ~2 s, then repeat passes until the budget is spent
    n = int(min(100_000_000, max(n, n * 2.0 / max(dt, 1e-3))))
    cols = sample(n)
    passes, total = 0, 0.0
    while total < budget_s and passes < 50:
        t0 = time.perf_counter()
        O.filter_project(cols, workload.filter, workload.projections, O.BYTECODE_COMPILER)
        total += time.perf_counter() - t0
        passes += 1
    columnar = None
    if workload.name == "config2" and workload.columns[0].null_pct == 0:
        # SURVEY 8(d) "cpu-columnar": the same query hand-specialised, all host cores (oracle/qe_columnar.c) -- the strong CPU baseline
        try:
            from queryengine_amd.ast import NumericLiteralExpression
            lits = []

            def walk(e):
                if isinstance(e, NumericLiteralExpression):
                    lits.append(e.value)
                for o in getattr(e, "operands", ()):
                    walk(o)
            walk(workload.filter)
            a_lim, c_lim = lits[0], lits[1]
            a, b, c = cols[0].data, cols[1].data, cols[2].data
            out = (np.empty(n + 1, dtype=np.int64), np.empty(n + 1, dtype=np.float64))
            nthr = host_cpu_share()
            _, _, used = O.columnar_config2(a, b, c, a_lim, c_lim, nthr, out)      # warm up (page in the output buffers)
            cp, ct = 0, 0.0
            while ct < 4.0 and cp < 200:
                t0 = time.perf_counter()
                O.columnar_config2(a, b, c, a_lim, c_lim, nthr, out)
                ct += time.perf_counter() - t0
                cp += 1
            columnar = {"value": n * cp / ct, "unit": "rows/s", "cores": used, "kind": "port-columnar",
                        "sample": f"{cp} passes over {n} rows, oracle/qe_columnar.c (hand-specialised columnar loop for this "
                                  f"query, two passes, OpenMP), {ct:.1f} s on {used} host threads"}
        except Exception as exc:   # the baseline is a report, never a reason to lose the bench line
            columnar = {"error": str(exc)}
    return {"value": n * passes / total, "unit": "rows/s", "cores": 1, "kind": "port", "columnar": columnar,
            "sample": f"{passes} passes over {n} rows of {workload.name} through oracle/qe_oracle.c (row-at-a-time C port "
                      f"of the reference evaluator, BYTECODE_COMPILER semantics; the reference is single-threaded), "
                      f"{total:.1f} s on 1 host core"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)   # the first two executions of a plan also time its two kernel geometries
    ap.add_argument("--rows", type=int, default=None, help="rows per GPU (default: the workload's BASELINE size, 1e9 for config2)")
    ap.add_argument("--workload", default="config2", choices=["config1", "config2", "config3", "config4"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tuning", default="", help="comma separated kernel tuning knobs (threads,unroll,nt,blocks_per_cu)")
    ap.add_argument("--exec-mode", default="fused", choices=["fused", "per_node"])
    ap.add_argument("--selectivity", type=float, default=None, help="config2 only: target selectivity (changes the literals)")
    ap.add_argument("--gather", action="store_true", help="also time the RCCL gather of the result to rank 0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py --gpus {args.gpus} must be launched with torch.distributed.run (one rank per GPU)", file=sys.stderr)
            sys.exit(2)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: queryengine_amd has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    rehearsal = os.environ.get("QE_BENCH_REHEARSAL") == "1"   # N ranks sharing GPU 0 over gloo: exercises the N>1 code on a 1-GPU box
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from queryengine_amd import engine as E
    from queryengine_amd import native as N
    from queryengine_amd import workloads as W

    if args.rows is None:
        args.rows = W.WORKLOADS[args.workload]().default_rows
    wl = W.WORKLOADS[args.workload](args.rows)
    if args.selectivity is not None and args.workload == "config2":
        # a < 1000 * s / c_limit with c < c_limit: sweep 1 %, 10 %, 50 %, 100 % as BASELINE.md section 3 asks
        s_ = max(0.0, min(1.0, args.selectivity))
        c_limit = 0.5 if s_ <= 0.5 else 1.0
        wl = W.config2(args.rows, a_limit=round(1000 * s_ / c_limit), c_limit=c_limit)
    tuning = [int(x) for x in args.tuning.split(",") if x]
    ctx = E.Context(device=local_rank, profile=True, tuning=tuning,
                    exec_mode=N.EXEC_FUSED if args.exec_mode == "fused" else N.EXEC_PER_NODE)
    nrows = args.rows
    batch = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], nrows, row_begin=rank * nrows, seed=42)
    cf = ctx.compile(wl.filter) if wl.filter is not None else None
    cp = [ctx.compile(p) for p in wl.projections]
    E.prepare(ctx, batch, cf, cp)    # plan time (buildPhysicalPlan): JIT compile, not part of a step

    def step():
        r = E.filter_project(ctx, batch, cf, cp)   # returns after the result count is known on the host
        n = r.count
        r.free()                                   # buffers go back to the context pool (re-open pattern)
        return n

    def barrier():
        if world > 1:
            dist.barrier()
        ctx.synchronize()
        torch.cuda.synchronize()

    nout = 0
    for _ in range(args.warmup):
        nout = step()
    ctx.reset_kernel_time()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nout = step()
    barrier()
    dt = time.perf_counter() - t0
    _, kernel_ms_total, launches = ctx.kernel_time()

    t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
    cnt = torch.tensor([nout], dtype=torch.int64, device="cpu" if rehearsal else "cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    dt_max = float(t.item())
    total_out = int(cnt.item())

    gather_info = None
    if args.gather:
        from queryengine_amd import distributed as QD
        gather_info = QD.time_gather(ctx, batch, cf, cp, world, rank)

    if rank == 0:
        kernel_ms = kernel_ms_total / max(1, launches)
        alg_bytes = wl.algorithmic_bytes(nrows, nout)
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        try:
            stream_gbps = ctx.stream_read_bandwidth(min(8 << 30, max(1 << 28, nrows * 8)), 5)
        except Exception:
            stream_gbps = None
        # HBM traffic per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction +
        # WRITE_SIZE, separate passes; profiles/README.md) -- only valid for the workload it was collected on
        traffic, traffic_src = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if tj.get("workload") == wl.name and tj.get("rows") == nrows and args.exec_mode == "fused":
                traffic, traffic_src = tj["hbm_bytes_per_launch"], tj["source"]
        except Exception:
            pass
        out = {
            "metric": "rows/sec filter+project over int64/f64 batch (1B rows per GPU); achieved HBM GB/s in roofline",
            "value": world * nrows * args.steps / dt_max,
            "unit": "rows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int64/f64", "data": "synthetic",
            "config": {"workload": f"{wl.name}: {wl.sql}", "rows_per_gpu": nrows, "rows_total": world * nrows,
                       "selected_rows_total": total_out, "exec_mode": args.exec_mode,
                       "sharding": "contiguous row ranges by global row index, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_gbps": (traffic / (kernel_ms * 1e-3) / 1e9) if traffic and kernel_ms > 0 else None,
                         "kernel": ("qe_fp_count + gb_scan + qe_fp_write (two-pass form, chosen at selectivity >= 0.6)"
                                    if args.exec_mode == "fused" and wl.filter is not None and nout >= 0.6 * nrows else
                                    "qe_fused" if args.exec_mode == "fused" else "per-node kernels"),
                         "kernel_ms": kernel_ms,
                         "note": "achieved = SURVEY 8(d) algorithmic bytes (every input column in full + output rows) / kernel time; "
                                 "the kernel loads later filter / projection columns only for rows still alive (late "
                                 "materialisation), so measured HBM traffic can be BELOW the algorithmic bytes", "algorithmic_bytes_per_launch": alg_bytes,
                         "measured_stream_read_gbps": stream_gbps},
        }
        if gather_info is not None:
            out["gather"] = gather_info
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(out), flush=True)
    batch.free()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

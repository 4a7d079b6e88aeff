#!/usr/bin/env python3
"""bench.py -- filter+project throughput of the fused HIP path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path -- qe_filter_project through the C ABI -- over one
HBM-resident synthetic batch (config 2: SELECT a + b, c * 2.0 FROM t WHERE a < 100 AND c < 0.5,
int64/int64/f64).  Inputs are generated on the device before the timed region.

Rows per GPU (unless --rows is given):
  N = 1   the configuration BASELINE.json's metric is quoted on: 1 B rows (configs[1]).
  N > 1   BASELINE.json configs[4] ("cfg 5"): 10 B rows range-sharded over 8 GPUs = 1.25 B rows per GPU.  The SAME shard
          size is used at N = 2 and 4 (2.5 B / 5 B rows in all), so per-GPU work is fixed as N grows (weak scaling) and
          N = 8 is exactly cfg 5.  Rank r owns global rows [r * rows, (r + 1) * rows); no data-path collective during
          the scan.  The materialising exchange (qe_gather: RCCL through the C ABI) is timed separately and reported
          in "gather" (with and without it); `value` is the scan, which is what a plan whose root consumes its
          shard in place (an aggregation) pays.

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
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


GATHER_TIMEOUT_S = 120
CFG5_ROWS_PER_GPU = 10_000_000_000 // 8      # BASELINE.json configs[4]


def traffic_key(workload: str, rows: int, selectivity, exec_mode: str, null_pct: int = 0) -> str:
    return (f"{workload}|rows={rows}|sel={'default' if selectivity is None else format(selectivity, 'g')}|{exec_mode}" +
            (f"|null={null_pct}" if null_pct else ""))


def code_hash(ctx, E, batch, cf, cp, exec_mode: str) -> str:
    """Identity of the code a traffic measurement belongs to: the generated kernel source (fused) or the per-node sources."""
    h = hashlib.sha1()
    if exec_mode == "fused":   # the plan's generated source + the generator itself (the dense / two-pass forms come from it too)
        h.update(E.generated_source(ctx, batch, cf, cp).encode())
        for f in ("qe_codegen.cpp", "qe_api.cpp"):   # the generator, and the executor that picks form / order / geometry
            h.update(open(os.path.join(ROOT, "queryengine_amd", "csrc", f), "rb").read())
        return h.hexdigest()[:16]
    for f in ("qe_pernode.cpp", "qe_pernode_kernels.hip"):
        h.update(open(os.path.join(ROOT, "queryengine_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def median(xs):
    xs = sorted(xs)
    n = len(xs)
    return 0.0 if n == 0 else (xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)   # the first executions of a plan also time its kernel geometries
    ap.add_argument("--rows", type=int, default=None, help="rows per GPU (default: see the module docstring)")
    ap.add_argument("--workload", default="config2", choices=["config1", "config2", "config3", "config4", "config2_swapped"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tuning", default="", help="comma separated kernel tuning knobs (qe_options.tuning, DESIGN.md 3.1a)")
    ap.add_argument("--exec-mode", default="fused", choices=["fused", "per_node"])
    ap.add_argument("--selectivity", type=float, default=None, help="config2 only: target selectivity (changes the literals)")
    ap.add_argument("--null-pct", type=int, default=0, help="config2 only: ~this % of NULLs in every input column (validity bitmaps "
                    "are read, nullable outputs are packed): SURVEY 8(d)'s nullable variant")
    ap.add_argument("--gather", action="store_true", help="time the materialising exchange too (default when --gpus > 1)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--profile-run", action="store_true",
                    help="under rocprofv3: exactly warmup + steps passes of the hot path and nothing else on the device")
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
        if world > 1 and args.workload in ("config2", "config2_swapped"):
            args.rows = CFG5_ROWS_PER_GPU
    wl = W.WORKLOADS[args.workload](args.rows)
    if args.selectivity is not None and args.workload == "config2":
        # a < 1000 * s / c_limit with c < c_limit: sweep 1 %, 10 %, 50 %, 100 % as BASELINE.md section 3 asks
        s_ = max(0.0, min(1.0, args.selectivity))
        c_limit = 0.5 if s_ <= 0.5 else 1.0
        wl = W.config2(args.rows, a_limit=round(1000 * s_ / c_limit), c_limit=c_limit, null_pct=args.null_pct)
    elif args.null_pct and args.workload in ("config2", "config2_swapped"):
        wl = W.WORKLOADS[args.workload](args.rows, null_pct=args.null_pct)
    tuning = [int(x) for x in args.tuning.split(",") if x]
    ctx = E.Context(device=local_rank, profile=True, tuning=tuning,
                    exec_mode=N.EXEC_FUSED if args.exec_mode == "fused" else N.EXEC_PER_NODE)
    nrows = args.rows
    batch = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], nrows, row_begin=rank * nrows, seed=42)
    cf = ctx.compile(wl.filter) if wl.filter is not None else None
    cp = [ctx.compile(p) for p in wl.projections]
    E.prepare(ctx, batch, cf, cp)    # plan time (buildPhysicalPlan): JIT compile, not part of a step
    # still plan time: on a fresh JIT cache the first executions of a plan on a large batch time its two kernel geometries
    # (best of 3 each) and persist the choice; do that before the warm-up so that no timed step is an exploring one
    if args.exec_mode == "fused":
        for _ in range(12):
            if E.chosen_geometry(ctx, batch, cf, cp)[0] != -1:
                break
            r = E.filter_project(ctx, batch, cf, cp)
            r.free()

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
    step_ms, kern_ms = [], []
    barrier()
    t0 = time.perf_counter()
    tprev = t0
    for _ in range(args.steps):
        nout = step()                              # ends with the result count on the host: a step boundary is a sync
        tnow = time.perf_counter()
        step_ms.append((tnow - tprev) * 1e3)
        tprev = tnow
        kern_ms.append(ctx.kernel_time()[0])
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

    # end to end including the result's way back to the host (SURVEY 8d "Timing"): step + every output column on the host.
    # e2e_with_d2h_ms: qe_result_to_host -- pinned staging owned by the library (pooled: pinned once, the first repetition
    # pays for it and is not the minimum), what a JVM host wraps as MemorySegments; e2e_with_d2h_pageable_ms: the columns copied
    # into the CALLER's pageable buffers (qe_result_column_to_host: pinned chunks + a few memcpy threads)
    e2e_ms, e2e_pageable_ms = None, None
    if not args.profile_run:
        try:
            import ctypes as C
            import numpy as np
            reps = []
            for _ in range(4):
                ctx.synchronize()
                t1 = time.perf_counter()
                r = E.filter_project(ctx, batch, cf, cp)
                h = r.to_host().wait()
                reps.append((time.perf_counter() - t1) * 1e3)
                h.free()
                r.free()
            e2e_ms = min(reps)
            r = E.filter_project(ctx, batch, cf, cp)
            bufs = []
            for c in range(r.ncols):
                v = r.view(c)
                words = (r.count + 63) // 64
                nbytes = words * 8 if v.type == 2 else r.count * (8 if v.type in (1, 3) else 4)
                bufs.append((np.empty(max(nbytes, 8), dtype=np.uint8), np.empty(max(words, 1), dtype=np.uint64)))
            r.free()
            reps = []
            for _ in range(3):
                ctx.synchronize()
                t1 = time.perf_counter()
                r = E.filter_project(ctx, batch, cf, cp)
                for c, (d_, v_) in enumerate(bufs):
                    N.check(ctx.handle, ctx._lib.qe_result_column_to_host(ctx.handle, r.handle, c, d_.ctypes.data, v_.ctypes.data))
                reps.append((time.perf_counter() - t1) * 1e3)
                r.free()
            e2e_pageable_ms = min(reps)
        except Exception as exc:
            print(f"e2e measurement failed: {type(exc).__name__}: {exc}", file=sys.stderr)

    if rank == 0:
        kernel_ms = kernel_ms_total / max(1, launches)
        alg_bytes = wl.algorithmic_bytes(nrows, nout)
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        stream_gbps = None
        if not args.profile_run:
            try:
                stream_gbps = ctx.stream_read_bandwidth(min(8 << 30, max(1 << 28, nrows * 8)), 5)
            except Exception:
                stream_gbps = None
        # HBM traffic per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE,
        # separate passes; tools/profile_all.sh) -- reported only when it was collected on this workload AND this code
        key = traffic_key(wl.name, nrows, args.selectivity, args.exec_mode, args.null_pct)
        chash = code_hash(ctx, E, batch, cf, cp, args.exec_mode)
        traffic, traffic_src, traffic_stale = None, None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            ent = tj.get("entries", {}).get(key)
            if ent is not None:
                if ent.get("code_hash") == chash:
                    traffic, traffic_src = ent["hbm_bytes_per_launch"], ent.get("source")
                else:
                    traffic_stale = f"profiles/traffic.json holds {key} for code {ent.get('code_hash')}, this run is {chash}"
        except Exception:
            pass
        conj_order = None
        if args.exec_mode == "fused" and cf is not None:
            try:
                conj_order = E.conjunct_order(ctx, batch, cf, cp)   # written-order indices of the filter's conjuncts in evaluation order
            except Exception:
                conj_order = None
        geometry = None
        if args.exec_mode == "fused":   # which of the plan's two kernel geometries the measured choice kept (and whether it came from the JIT cache)
            try:
                g_, cached_ = E.chosen_geometry(ctx, batch, cf, cp)
                geometry = {"chosen": {-1: "undecided", 0: "default", 1: "wide", 2: "mid"}.get(g_, str(g_)), "from_jit_cache": cached_}
            except Exception:
                geometry = None
        out = {
            "metric": "rows/sec filter+project over int64/f64 batch; achieved HBM GB/s in roofline",
            "value": world * nrows * args.steps / dt_max,
            "unit": "rows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "ms_median": median(step_ms), "ms_min": min(step_ms) if step_ms else None,
            "e2e_with_d2h_ms": e2e_ms, "e2e_with_d2h_pageable_ms": e2e_pageable_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int64/f64", "data": "synthetic",
            "config": {"workload": f"{wl.name}: {wl.sql}", "rows_per_gpu": nrows, "rows_total": world * nrows,
                       "selected_rows_total": total_out, "exec_mode": args.exec_mode,
                       "sharding": "contiguous row ranges by global row index, no data-path collective",
                       "rows_rule": "N=1: 1 B rows (BASELINE configs[1]); N>1: 1.25 B rows per GPU (configs[4]: 10 B / 8), same shard at N=2,4"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_stale": traffic_stale, "traffic_key": key, "code_hash": chash,
                         "traffic_gbps": (traffic / (kernel_ms * 1e-3) / 1e9) if traffic and kernel_ms > 0 else None,
                         # the honest second fraction: bytes the kernel really MOVED (PMC) / kernel time / peak
                         "frac_moved": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic and kernel_ms > 0 else None,
                         "kernel": ({N.FORM_DENSE: "qe_fused (dense single-pass form, chosen from selectivity 0.12 on)",
                                     N.FORM_RING: "qe_fused (LDS-ring single-pass form)",
                                     N.FORM_LOCAL: "qe_fl_scan + scan + qe_fl_move (local form: dependency-free scan into per-chunk slots, "
                                                   "chosen up to selectivity 0.03; kernel_ms covers all of them)",
                                     N.FORM_TWO_PASS: "qe_fp_count + scan + qe_fp_write (two-pass form)",
                                     N.FORM_NO_FILTER: "qe_fused (projection only)"}.get(ctx.last_form, "qe_fused")
                                    if args.exec_mode == "fused" else "per-node kernels"),
                         "conjunct_order": conj_order,
                         "kernel_ms": kernel_ms, "kernel_ms_median": median(kern_ms), "kernel_ms_min": min(kern_ms) if kern_ms else None,
                         "geometry": geometry,
                         "note": "achieved = SURVEY 8(d) algorithmic bytes (every input column in full + output rows) / kernel time; "
                                 "the kernel loads later filter / projection columns only for rows still alive (late "
                                 "materialisation), so measured HBM traffic can be BELOW the algorithmic bytes: frac_moved is "
                                 "the fraction of peak the kernel really sustains", "algorithmic_bytes_per_launch": alg_bytes,
                         "measured_stream_read_gbps": stream_gbps},
        }
        if not args.no_cpu_baseline and world == 1 and not args.profile_run:
            out["cpu_baseline"] = cpu_baseline(wl)
    else:
        out = None

    # The exchange runs LAST and under a watchdog: it is the one part of this file that no multi-GPU box has executed before
    # the driver's own run, and a collective that never returns must not cost the bench line.  On expiry rank 0 prints the
    # line (with the timeout recorded) and every rank leaves without tearing down the stuck communicator.
    import threading

    def give_up():
        # the line goes out first (it carries gather.error), then the process leaves NON-ZERO: a collective that hung must
        # not look like a clean run to whoever keys on the exit code
        if rank == 0 and out is not None:
            g = out.get("gather") or {}
            g["error"] = f"timeout: the exchange did not finish within {GATHER_TIMEOUT_S} s"
            out["gather"] = g
            print(json.dumps(out), flush=True)
        os._exit(3)

    watchdog = threading.Timer(GATHER_TIMEOUT_S, give_up)
    watchdog.daemon = True
    watchdog.start()
    # the materialising exchange (cfg 5: "with and without the RCCL gather"); never part of `value`
    gather_info = None
    # a rehearsal (N ranks on ONE GPU) runs the exchange only when the caller names a transport that accepts several ranks
    # per device (QE_RCCL_LIBRARY; RCCL itself refuses that)
    exchange_possible = not rehearsal or world == 1 or bool(os.environ.get("QE_RCCL_LIBRARY"))
    if (args.gather or world > 1) and not args.no_gather and not args.profile_run and exchange_possible:
        try:
            from queryengine_amd import distributed as QD
            if world > 1:
                QD.comm_init(ctx)
            else:
                ctx.comm_init(1, 0, ctx.comm_unique_id())
            gather_info = QD.time_gather(ctx, batch, cf, cp, world, rank)
            if world > 1:
                tg = torch.tensor([gather_info["ms"], gather_info["scan_plus_gather_ms"]], dtype=torch.float64,
                                  device="cpu" if rehearsal else "cuda")
                dist.all_reduce(tg, op=dist.ReduceOp.MAX)
                gather_info["ms"], gather_info["scan_plus_gather_ms"] = float(tg[0].item()), float(tg[1].item())
            gather_info["rows_per_s_with_gather"] = world * nrows / (gather_info["scan_plus_gather_ms"] * 1e-3)
            if rank == 0 and out is not None:
                out["gather"] = dict(gather_info)     # what is known so far survives a hang in the overlapped form below
            # the overlapped form (scan in slices, every slice's rows travelling while the next one is scanned): one call
            try:
                ov = QD.time_gather_overlapped(ctx, batch, cf, cp, world, rank)
                if world > 1:
                    tv = torch.tensor([ov["ms"]], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
                    dist.all_reduce(tv, op=dist.ReduceOp.MAX)
                    ov["ms"] = float(tv[0].item())
                ov["rows_per_s"] = world * nrows / (ov["ms"] * 1e-3)
                gather_info["overlapped"] = ov
            except Exception as exc:
                gather_info["overlapped"] = {"error": f"{type(exc).__name__}: {exc}"}
        except Exception as exc:      # the exchange is a report beside the bench line, never a reason to lose it
            gather_info = {"error": f"{type(exc).__name__}: {exc}"}

    watchdog.cancel()
    if rank == 0:
        if gather_info is not None:
            out["gather"] = gather_info
        print(json.dumps(out), flush=True)
    if world > 1:   # the line is out: a teardown that hangs after a failed exchange must not hold the job
        sys.stdout.flush()
        late = threading.Timer(60, lambda: os._exit(4))     # the line is out, but a teardown that hangs is not a clean exit
        late.daemon = True
        late.start()
    batch.free()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""One rank of the multi-rank exchange tests (tests/test_gpu_gather_multirank.py): a fresh process that shares GPU 0 with
the other ranks, runs the PRODUCT's qe_filter_project on its row-range shard and the PRODUCT's qe_gather / qe_comm_* through
the C ABI.  The bytes between the processes travel through tests/transport (QE_RCCL_LIBRARY), because RCCL itself refuses
two ranks on one device -- the code under test (qe_comm.cpp: header exchange, offsets, send / recv pairing, bitmap
placement, error paths) is the product's own.

usage: gather_worker.py <rank> <world> <uid file> <report file>
Every scenario appends {"name", "ok", "detail"} to the rank's report; a scenario that expects an error checks that EVERY
rank got it (the driver compares the reports)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    rank, world, idfile, report_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    from helpers import B, D, I64, S, assert_columns_equal, col, fn, num
    from queryengine_amd import Column, DataType, Function as Fn
    from queryengine_amd import engine as E, native as N, workloads as W
    from queryengine_amd.distributed import shard_range

    report = []

    def note(name, ok, detail=""):
        report.append({"name": name, "ok": bool(ok), "detail": str(detail)[:500]})
        with open(report_path + ".tmp", "w") as f:
            json.dump(report, f)
        os.replace(report_path + ".tmp", report_path)

    ctx = E.Context(device=0)
    try:
        if rank == 0:
            uid = ctx.comm_unique_id()
            with open(idfile + ".tmp", "wb") as f:
                f.write(uid)
            os.replace(idfile + ".tmp", idfile)
        else:
            for _ in range(1200):
                if os.path.exists(idfile):
                    break
                time.sleep(0.05)
            uid = open(idfile, "rb").read()
        ctx.comm_init(world, rank, uid)
        note("comm_init", ctx.comm_nranks == world and ctx.comm_rank == rank)

        # ---- control data: qe_comm_allgather_host in rank order ----
        parts = ctx.allgather_host(bytes([rank + 1]) * 24)
        note("allgather_host", parts == [bytes([r + 1]) * 24 for r in range(world)])

        def compare(got, want, what):
            try:
                assert got.count == want.count, f"count {got.count} != {want.count}"
                for i, (g, w) in enumerate(zip(got.to_columns(), want.to_columns())):
                    assert_columns_equal(g, w, f"{what} column {i}")
                return True, f"{got.count} rows"
            except AssertionError as exc:
                return False, exc

        # ---- cfg 2 with ~1 % nulls + two BOOLEAN projections, ragged shards, an EMPTY shard, root != 0 ----
        def cfg2_case(name, sizes, root):
            wl = W.config2(sum(sizes), null_pct=1)
            projs = list(wl.projections) + [fn(Fn.CMP_LT, col("c", 2, D), num(0.25)), fn(Fn.CMP_GT, col("b", 1, I64), num(7))]
            cf, cp = ctx.compile(wl.filter), [ctx.compile(p) for p in projs]
            begin = sum(sizes[:rank])
            batch = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], sizes[rank], row_begin=begin)
            local = E.filter_project(ctx, batch, cf, cp)
            g = ctx.gather(local, root)
            if rank == root:
                whole_b = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], sum(sizes), row_begin=0)
                whole = E.filter_project(ctx, whole_b, cf, cp)
                ok, detail = compare(g, whole, name)
                note(name, ok and g.count % 64 != 0, detail)
                whole.free(); whole_b.free(); g.free()
            else:
                note(name, g is None, f"local {local.count} rows")
            local.free(); batch.free()

        sizes = [64 * 700, 0, 64 * 1111 + 17][:world] if world >= 3 else [64 * 700, 64 * 1111 + 17]
        cfg2_case("cfg2_nullable_ragged_empty_shard_root1", sizes, 1 % world)
        # equal 64-aligned shards as distributed.shard_range hands them out, root 0
        n = 200_000
        cfg2_case("cfg2_shard_range_root0", [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)], 0)

        # ---- the OVERLAPPED form: scan in slices, every slice's rows on their way while the next one is scanned ----
        def overlapped_case(name, sizes, root, nslices):
            wl = W.config2(sum(sizes), null_pct=1)
            projs = list(wl.projections) + [fn(Fn.CMP_LT, col("c", 2, D), num(0.25)), fn(Fn.CMP_GT, col("b", 1, I64), num(7))]
            cf, cp = ctx.compile(wl.filter), [ctx.compile(p) for p in projs]
            begin = sum(sizes[:rank])
            batch = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], sizes[rank], row_begin=begin)
            g = ctx.filter_project_gather(batch, cf, cp, root, nslices)
            if rank == root:
                whole_b = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], sum(sizes), row_begin=0)
                whole = E.filter_project(ctx, whole_b, cf, cp)
                ok, detail = compare(g, whole, name)
                note(name, ok, detail)
                whole.free(); whole_b.free(); g.free()
            else:
                note(name, g is None)
            batch.free()
        # several 16 Ki-row chunks per slice, ragged tails, ranks with different numbers of slices, an empty shard
        big = [64 * 5000 + 17, 64 * 900, 64 * 3000 + 5][:world] if world >= 3 else [64 * 5000 + 17, 64 * 900]
        overlapped_case("overlapped_gather_ragged_root0", big, 0, 4)
        overlapped_case("overlapped_gather_root_last_16_slices", big, world - 1, 16)
        overlapped_case("overlapped_gather_empty_shard", [64 * 2000, 0, 64 * 700 + 9][:world] if world >= 3 else [0, 64 * 700 + 9], 1 % world, 3)

        # no Filter node: every row travels (BOOLEAN values as bitmap words across slice boundaries that are not multiples of 64 kept rows)
        def overlapped_no_filter():
            k = 40_000 + 1000 * rank
            base = 1_000_000 * rank
            data = np.arange(base, base + k, dtype=np.float64)
            valid = (np.arange(k) % 5 != 0) if rank % 2 == 0 else None
            b = E.DeviceBatch.from_columns(ctx, [Column(DataType.DOUBLE, data, valid)])
            projs = [col("x", 0, D), fn(Fn.CMP_LT, col("x", 0, D), num(base + 20_000))]
            g = ctx.filter_project_gather(b, None, [ctx.compile(p) for p in projs], 0, 2)
            if rank == 0:
                want_d, want_v, want_b = [], [], []
                for r in range(world):
                    kr = 40_000 + 1000 * r
                    d_ = np.arange(1_000_000 * r, 1_000_000 * r + kr, dtype=np.float64)
                    want_d.append(d_)
                    want_v.append((np.arange(kr) % 5 != 0) if r % 2 == 0 else np.ones(kr, bool))
                    want_b.append(d_ < 1_000_000 * r + 20_000)
                want_d, want_v, want_b = np.concatenate(want_d), np.concatenate(want_v), np.concatenate(want_b)
                got = g.to_columns()
                ok = np.array_equal(got[0].valid, want_v) and np.array_equal(got[0].data[want_v], want_d[want_v])
                ok = ok and np.array_equal(got[1].valid, want_v) and np.array_equal(got[1].data[want_v], want_b[want_v])
                note("overlapped_gather_no_filter_mixed_validity", ok, f"{g.count} rows")
                g.free()
            else:
                note("overlapped_gather_no_filter_mixed_validity", g is None)
            b.free()
        overlapped_no_filter()

        # ---- a shard WITHOUT a validity bitmap next to shards with one (the missing bitmap counts as ones) ----
        def mixed_validity(root):
            k = 1000 + 37 * rank
            base = 10_000 * rank
            data = np.arange(base, base + k, dtype=np.float64)
            valid = (np.arange(k) % 3 != 0) if rank % 2 == 1 else None
            b = E.DeviceBatch.from_columns(ctx, [Column(DataType.DOUBLE, data, valid)])
            local = E.filter_project(ctx, b, None, [ctx.compile(col("x", 0, D))])
            g = ctx.gather(local, root)
            if rank == root:
                want_d, want_v = [], []
                for r in range(world):
                    kr = 1000 + 37 * r
                    want_d.append(np.arange(10_000 * r, 10_000 * r + kr, dtype=np.float64))
                    want_v.append((np.arange(kr) % 3 != 0) if r % 2 == 1 else np.ones(kr, bool))
                want_d, want_v = np.concatenate(want_d), np.concatenate(want_v)
                got = g.to_columns()[0]
                ok = got.valid is not None and np.array_equal(got.valid, want_v) and np.array_equal(got.data[want_v], want_d[want_v])
                note("mixed_validity", ok, f"{g.count} rows")
                g.free()
            else:
                note("mixed_validity", g is None)
            local.free(); b.free()
        mixed_validity(0)

        # ---- STRING columns travel as codes: same dictionary everywhere is required, and checked on every rank ----
        def strings(name, entries, root, expect_error):
            codes = ((np.arange(500 + 11 * rank) * 7 + rank) % 3).astype(np.int32)
            b = E.DeviceBatch.from_columns(ctx, [Column(DataType.STRING, codes, None, list(entries))])
            local = E.filter_project(ctx, b, None, [ctx.compile(col("s", 0, S))])
            try:
                g = ctx.gather(local, root)
                if expect_error:
                    note(name, False, "no error")
                elif rank == root:
                    want = []
                    for r in range(world):
                        want += [["a", "b", "c"][(i * 7 + r) % 3] for i in range(500 + 11 * r)]
                    got = g.to_columns()[0]
                    note(name, [got.dictionary[c] for c in got.data] == want, f"{g.count} rows")
                    g.free()
                else:
                    note(name, g is None)
            except N.QeError as exc:
                note(name, expect_error and exc.code == 1, exc)
            local.free(); b.free()
        strings("shared_dictionary", ["a", "b", "c"], world - 1, False)
        strings("dictionary_mismatch_is_invalid_arg_on_every_rank", ["a", "c", "b"] if rank == world - 1 else ["a", "b", "c"], 0, True)

        # ---- a result of another schema on one rank: QE_ERR_INVALID_ARG on EVERY rank, nobody hangs ----
        def schema_mismatch():
            data = np.arange(100, dtype=np.float64)
            b = E.DeviceBatch.from_columns(ctx, [Column(DataType.DOUBLE, data, None)])
            proj = fn(Fn.CMP_LT, col("x", 0, D), num(50)) if rank == world - 1 else col("x", 0, D)
            local = E.filter_project(ctx, b, None, [ctx.compile(proj)])
            try:
                ctx.gather(local, 0)
                note("schema_mismatch_is_invalid_arg_on_every_rank", False, "no error")
            except N.QeError as exc:
                note("schema_mismatch_is_invalid_arg_on_every_rank", exc.code == 1, exc)
            local.free(); b.free()
        schema_mismatch()

        # ---- the communicator still works after the refused calls ----
        cfg2_case("gather_after_errors", [64 * 10 + 3] * world, 0)
        with_err = None
        try:
            ctx.gather(E.filter_project(ctx, E.DeviceBatch.from_columns(ctx, [Column(DataType.DOUBLE, np.zeros(4), None)]), None,
                                        [ctx.compile(col("x", 0, D))]), world + 3)
        except N.QeError as exc:
            with_err = exc.code
        note("root_out_of_range", with_err == 1)
        ctx.comm_destroy()
        note("done", ctx.comm_nranks == 0)
    except Exception as exc:   # noqa: BLE001
        import traceback
        note("exception", False, traceback.format_exc()[-480:])
        ctx.close()
        sys.exit(1)
    ctx.close()


if __name__ == "__main__":
    main()

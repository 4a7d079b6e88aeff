"""The exchange step on the GPU box: RCCL (backend nccl) with the ranks that fit one GPU (world 1),
zero-copy tensor views of result columns, gather in rank order.  The multi-rank logic itself is
covered on CPU by tests/test_distributed_cpu.py (gloo, world size 2)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_result_views_and_gather_world1(gpu_ctx, oracle):
    import torch
    import torch.distributed as dist
    from queryengine_amd import engine as E
    from queryengine_amd import workloads as W
    from queryengine_amd.distributed import gather_result, result_column_tensor, shard_range

    wl = W.config2(300_000, null_pct=1)
    begin, end = shard_range(wl.default_rows, 0, 1)
    batch = E.DeviceBatch.generate(gpu_ctx, [c.spec(gpu_ctx) for c in wl.columns], end - begin, row_begin=begin)
    res = E.filter_project(gpu_ctx, batch, gpu_ctx.compile(wl.filter), [gpu_ctx.compile(p) for p in wl.projections])
    host = res.to_columns()
    # zero-copy torch views of the HBM result buffers
    for c in range(res.ncols):
        data, valid = result_column_tensor(res, c)
        assert data.is_cuda and data.shape[0] == res.count
        assert np.array_equal(data.cpu().numpy().view(np.uint64), host[c].data.view(np.uint64))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        gathered = gather_result(res, 0)     # no communicator on gpu_ctx: the torch.distributed restatement
        for (g, gv), h in zip(gathered, host):
            hv = h.valid if h.valid is not None else np.ones(len(h), dtype=bool)
            if gv is not None:
                assert np.array_equal(gv.cpu().numpy().astype(bool), hv)
            assert np.array_equal(g.cpu().numpy()[hv].view(np.uint64), h.data[hv].view(np.uint64))
        # aggregation across shards (here one): AVG is carried as SUM + COUNT partials and finished after the merge
        from helpers import AGG_CASE, agg_case_columns
        from queryengine_amd.distributed import sharded_filter_aggregate, sharded_filter_groupby
        cols = agg_case_columns(0, 30_000)
        flt, keys, exprs, aggs = AGG_CASE()
        ab = E.DeviceBatch.from_columns(gpu_ctx, cols)
        cf, ck, ce = gpu_ctx.compile(flt), [gpu_ctx.compile(k) for k in keys], [gpu_ctx.compile(e) for e in exprs]
        want, _ = oracle.filter_aggregate(cols, flt, exprs, aggs, oracle.BYTECODE_COMPILER)
        assert sharded_filter_aggregate(gpu_ctx, ab, cf, ce, aggs) == want
        assert sharded_filter_groupby(gpu_ctx, ab, cf, ck, ce, aggs) == oracle.filter_groupby(cols, flt, keys, exprs, aggs,
                                                                                              oracle.BYTECODE_COMPILER)
        ab.free()
    finally:
        dist.destroy_process_group()
    res.free()
    batch.free()


def _cfg2_nullable_parts(ctx, sizes, begin=0):
    """config 2 with ~1 % nulls over consecutive row ranges of the given sizes: (batches, compiled filter, projections)."""
    from queryengine_amd import engine as E
    from queryengine_amd import workloads as W
    from queryengine_amd import Function as Fn
    from helpers import B, D, I64, col, fn, num
    wl = W.config2(sum(sizes), null_pct=1)
    # a BOOLEAN projection (bitmap values), a nullable DOUBLE, a nullable INT64 and a non-nullable INT32-free literal compare
    projs = list(wl.projections) + [fn(Fn.CMP_LT, col("c", 2, D), num(0.25)), fn(Fn.CMP_GT, col("b", 1, I64), num(7))]
    batches = []
    for n in sizes:
        batches.append(E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], n, row_begin=begin))
        begin += n
    return wl, batches, ctx.compile(wl.filter), [ctx.compile(p) for p in projs], projs


def test_result_concat_equals_one_pass(gpu_ctx, oracle):
    """qe_result_concat: results of consecutive row ranges, concatenated, are bit for bit the result of one pass over the
    whole table -- value columns at row offsets, BOOLEAN / validity bitmaps shifted as words across ragged counts, an
    empty part in the middle.  The same placement code is qe_gather's."""
    from queryengine_amd import engine as E
    from helpers import assert_columns_equal
    sizes = [64 * 700, 64 * 3, 0, 64 * 1111, 64 * 50 + 17]     # shard boundaries are 64-row aligned; the last part is ragged
    wl, batches, cf, cp, projs = _cfg2_nullable_parts(gpu_ctx, sizes)
    parts = [E.filter_project(gpu_ctx, b, cf, cp) for b in batches]
    assert len({p.count % 64 for p in parts}) > 2                # the bitmap segments really start at odd bit offsets
    whole_batch = E.DeviceBatch.generate(gpu_ctx, [c.spec(gpu_ctx) for c in wl.columns], sum(sizes), row_begin=0)
    whole = E.filter_project(gpu_ctx, whole_batch, cf, cp)
    cat = gpu_ctx.concat(parts)
    assert cat.count == whole.count == sum(p.count for p in parts)
    for i, (g, w) in enumerate(zip(cat.to_columns(), whole.to_columns())):
        assert_columns_equal(g, w, f"column {i}")
    # and against the oracle on the host copy of the same rows
    cols = [whole_batch.column_to_host(i) for i in range(whole_batch.ncols)]
    want = oracle.filter_project(cols, wl.filter, projs, oracle.BYTECODE_COMPILER)
    for i, (g, w) in enumerate(zip(cat.to_columns(), want)):
        assert_columns_equal(g, w, f"column {i} vs oracle")
    # parts without a validity bitmap next to parts with one: the missing bitmap counts as all ones
    from queryengine_amd import Column, DataType
    a = Column(DataType.DOUBLE, np.arange(100, dtype=np.float64))
    b = Column(DataType.DOUBLE, np.arange(100, 230, dtype=np.float64), np.arange(130) % 3 != 0)
    from helpers import D, col
    pa = E.filter_project(gpu_ctx, E.DeviceBatch.from_columns(gpu_ctx, [a]), None, [gpu_ctx.compile(col("x", 0, D))])
    pb = E.filter_project(gpu_ctx, E.DeviceBatch.from_columns(gpu_ctx, [b]), None, [gpu_ctx.compile(col("x", 0, D))])
    mixed = gpu_ctx.concat([pa, pb, pa]).to_columns()[0]
    assert np.array_equal(mixed.valid, np.concatenate([np.ones(100, bool), b.valid, np.ones(100, bool)]))
    assert np.array_equal(mixed.data[mixed.valid], np.concatenate([a.data, b.data, a.data])[mixed.valid])
    for r in parts + [whole, cat, pa, pb]:
        r.free()
    for bt in batches + [whole_batch]:
        bt.free()


def test_c_abi_gather_over_rccl_world1(native_lib, oracle):
    """qe_comm_unique_id / qe_comm_init / qe_gather / qe_comm_allgather_host through the C ABI on RCCL itself (no
    torch.distributed anywhere): the ranks that fit one GPU (world 1).  The multi-rank placement is the code
    test_result_concat_equals_one_pass covers; rank order / offsets are rehearsed at world 2 on gloo (CPU suite)."""
    from queryengine_amd import engine as E
    from queryengine_amd import native as N
    from helpers import assert_columns_equal
    ctx = E.Context(device=0)
    try:
        assert ctx.comm_nranks == 0 and ctx.comm_rank == -1
        wl, batches, cf, cp, projs = _cfg2_nullable_parts(ctx, [64 * 900 + 5])
        res = E.filter_project(ctx, batches[0], cf, cp)
        with pytest.raises(N.QeError) as ei:     # no communicator yet
            ctx.gather(res, 0)
        assert ei.value.code == 7                # QE_ERR_COMM
        uid = ctx.comm_unique_id()
        assert len(uid) == 128
        ctx.comm_init(1, 0, uid)
        assert ctx.comm_nranks == 1 and ctx.comm_rank == 0
        g = ctx.gather(res, 0)
        assert g is not None and g.count == res.count
        for i, (a, b) in enumerate(zip(g.to_columns(), res.to_columns())):
            assert_columns_equal(a, b, f"column {i}")
        assert ctx.allgather_host(b"0123456789abcdef") == [b"0123456789abcdef"]
        with pytest.raises(N.QeError):
            ctx.gather(res, 3)                   # root out of range
        g.free(); res.free(); batches[0].free()
        ctx.comm_destroy()
        assert ctx.comm_nranks == 0
    finally:
        ctx.close()

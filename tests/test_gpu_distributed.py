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
        gathered = gather_result(res, 0)
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

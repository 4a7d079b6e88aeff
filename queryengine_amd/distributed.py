"""Multi-GPU: row-range sharding + the one exchange step of the path.

filter+project is row independent, so a scan shards by CONTIGUOUS row ranges in rank order
(SURVEY 8e): rank r owns rows [begin_r, end_r) of the global table, generates / pins only those,
and runs the identical fused kernel on its own GPU and stream with NO communication.  Only when a
plan materialises its result on one rank is there an exchange: an all-gather of the per-rank
output counts, then a variable-length gather implemented as direct point-to-point sends grouped
into one batch (RCCL: ncclGroupStart/End around ncclSend/ncclRecv), so that every peer uses its
OWN xGMI link into the root instead of a ring bounded by one link.  Concatenation in rank order
preserves the reference's output order (FilterOperator.kt:17-22 is order preserving).

The same code runs over gloo with CPU tensors (tests) and over nccl (= RCCL) with GPU tensors.
"""
from __future__ import annotations

import time
from typing import List, Optional, Sequence, Tuple

import numpy as np

_TORCH_DTYPE = {"f8": "float64", "i8": "int64", "i4": "int32", "u8": "int64", "u1": "uint8"}


def shard_range(nrows_total: int, rank: int, world: int, align: int = 64) -> Tuple[int, int]:
    """Rows [begin, end) of rank `rank`: equal contiguous shares, boundaries aligned to `align`
    rows so that validity / boolean bitmap words never straddle two shards."""
    per = -(-nrows_total // world)
    per = -(-per // align) * align
    begin = min(nrows_total, rank * per)
    end = min(nrows_total, begin + per)
    return begin, end


def gatherv(tensor, dst: int = 0, group=None):
    """Variable-length gather of 1-D tensors to rank `dst`, concatenated in rank order.

    Returns the concatenated tensor on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = torch.tensor([tensor.shape[0]], dtype=torch.int64, device=tensor.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    if rank == dst:
        out = torch.empty(sum(counts), dtype=tensor.dtype, device=tensor.device)
        offs = np.concatenate([[0], np.cumsum(counts)])
        ops = []
        for src in range(world):
            if counts[src] == 0:
                continue
            view = out[int(offs[src]):int(offs[src + 1])]
            if src == rank:
                view.copy_(tensor)
            else:
                ops.append(dist.P2POp(dist.irecv, view, src, group))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        return out
    if counts[rank] > 0:
        for r in dist.batch_isend_irecv([dist.P2POp(dist.isend, tensor.contiguous(), dst, group)]):
            r.wait()
    return None


class _DeviceArray:
    """Zero-copy view of a qe_result column for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr: int, n: int, typestr: str, owner):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 3}
        self._owner = owner


def result_column_tensor(result, col: int):
    """(values tensor, validity-words tensor | None) viewing the result's HBM buffers (no copy)."""
    import torch
    from .datatypes import DataType
    v = result.view(col)
    n = int(v.count)
    t = DataType(v.type)
    dev = torch.device("cuda", result.ctx.device)
    if t == DataType.BOOLEAN:
        ts, cnt = "<i8", (n + 63) // 64
    else:
        ts, cnt = {DataType.DOUBLE: "<f8", DataType.INT64: "<i8", DataType.INT32: "<i4", DataType.STRING: "<i4"}[t], n
    if cnt == 0:
        tt = {"<f8": torch.float64, "<i8": torch.int64, "<i4": torch.int32}[ts]
        return torch.empty(0, dtype=tt, device=dev), None
    data = torch.as_tensor(_DeviceArray(int(v.data), cnt, ts, result), device=dev)
    valid = None
    if v.validity:
        valid = torch.as_tensor(_DeviceArray(int(v.validity), (n + 63) // 64, "<i8", result), device=dev)
    return data, valid


def _unpack_bits(words, n: int):
    """int64 bitmap words -> uint8 per row (torch, any device); plumbing around the exchange only."""
    import torch
    if n == 0:
        return torch.empty(0, dtype=torch.uint8, device=words.device)
    shifts = torch.arange(64, dtype=torch.int64, device=words.device)
    bits = (words.unsqueeze(1) >> shifts) & 1
    return bits.reshape(-1)[:n].to(torch.uint8)


def gather_result(result, dst: int = 0, group=None):
    """Materialise a sharded result on rank `dst`: per output column the values (BOOLEAN: uint8 per
    row) and, if nullable, a uint8 validity vector, concatenated in rank order.  Returns a list of
    (values, valid|None) torch tensors on `dst`, None elsewhere."""
    import torch.distributed as dist
    from .datatypes import DataType
    out = []
    n = result.count
    for c in range(result.ncols):
        data, valid = result_column_tensor(result, c)
        if DataType(result.view(c).type) == DataType.BOOLEAN:
            data = _unpack_bits(data, n)
        g = gatherv(data, dst, group)
        gv = gatherv(_unpack_bits(valid, n), dst, group) if valid is not None else None
        out.append((g, gv))
    return out if dist.get_rank(group) == dst else None


def time_gather(ctx, batch, cf, cp, world: int, rank: int, reps: int = 3):
    """bench.py --gather: time the materialising exchange separately from the scan."""
    import torch
    import torch.distributed as dist
    from . import engine as E
    res = E.filter_project(ctx, batch, cf, cp)
    best = None
    total = 0
    for _ in range(reps):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cols = gather_result(res, 0) if world > 1 else [result_column_tensor(res, c) for c in range(res.ncols)]
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        if rank == 0 and cols:
            total = int(cols[0][0].shape[0])
    nbytes = 0
    if rank == 0:
        nbytes = sum(int(c[0].numel()) * c[0].element_size() for c in cols)
    res.free()
    return {"ms": best * 1e3, "rows_on_root": total, "bytes_on_root": nbytes,
            "gbps_into_root": (nbytes / best / 1e9) if best else None,
            "method": "count all-gather + grouped ncclSend/ncclRecv (direct peer links), rank order"}

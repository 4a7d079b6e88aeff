"""Multi-GPU: row-range sharding + the one exchange step of the path.

filter+project is row independent, so a scan shards by CONTIGUOUS row ranges in rank order
(SURVEY 8e): rank r owns rows [begin_r, end_r) of the global table, generates / pins only those,
and runs the identical fused kernel on its own GPU and stream with NO communication.  Only when a
plan materialises its result on one rank is there an exchange: an all-gather of the per-rank
output counts, then a variable-length gather implemented as direct point-to-point sends grouped
into one batch (RCCL: ncclGroupStart/End around ncclSend/ncclRecv), so that every peer uses its
OWN xGMI link into the root instead of a ring bounded by one link.  Concatenation in rank order
preserves the reference's output order (FilterOperator.kt:17-22 is order preserving).

The PRODUCT exchange is libqe_hip.so's own (qe_comm_init / qe_gather / qe_comm_allgather_host, RCCL called directly
from the C ABI -- what a JVM host binds): `comm_init` bootstraps it and `gather_result` / `allreduce_aggregates` use it
whenever the context holds a communicator.  The torch.distributed forms below (`gatherv`, the gloo branches) restate the
same offset / order logic for the world-size-2 CPU tests, where no GPU and no RCCL exist.
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


def comm_init(ctx, group=None) -> None:
    """Bootstrap the context's RCCL communicator: rank 0 creates the ncclUniqueId through the C ABI, the host channel
    that carries its 128 bytes to the other ranks is torch.distributed here (a JVM host uses its own RPC)."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [ctx.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    ctx.comm_init(world, rank, box[0])


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
    """Materialise a sharded result on rank `dst`.

    With a communicator on the result's context (comm_init) this is qe_gather: the concatenated `engine.Result` on
    `dst`, None elsewhere.  Without one (torch.distributed only) it returns, on `dst`, a list of (values, valid|None)
    torch tensors per output column (BOOLEAN / validity as uint8 per row), None elsewhere."""
    import torch.distributed as dist
    if result.ctx.comm_nranks > 0:
        return result.ctx.gather(result, dst)
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


# ---- aggregation across shards ("Cross-GPU: reduce of the per-GPU partials") -------------------
_MIN, _MAX, _SUM, _COUNT, _AVG = 0, 1, 2, 3, 4     # ast/Functions.kt:24-26


def expand_partial_aggregates(aggs: Sequence[int]):
    """Per-shard accumulators that can be merged: AVG becomes SUM + COUNT of the same input (Accumulators.kt:82-107 keeps
    exactly that pair).  Returns (partial_fns, source_index per partial, recipe) where recipe[i] lists the partial slots
    of aggregate i."""
    fns: List[int] = []
    src: List[int] = []
    recipe: List[Tuple[int, ...]] = []
    for i, a in enumerate(aggs):
        a = int(a)
        if a == _AVG:
            recipe.append((len(fns), len(fns) + 1))
            fns += [_SUM, _COUNT]
            src += [i, i]
        else:
            recipe.append((len(fns),))
            fns.append(a)
            src.append(i)
    return fns, src, recipe


def _java_min(a: float, b: float) -> float:
    """java.lang.Math.min: NaN wins, -0.0 < 0.0."""
    if a != a or b != b:
        return float("nan")
    if a == 0.0 and b == 0.0:
        return a if np.signbit(a) else b
    return a if a < b else b


def _java_max(a: float, b: float) -> float:
    if a != a or b != b:
        return float("nan")
    if a == 0.0 and b == 0.0:
        return b if np.signbit(a) else a
    return a if a > b else b


def _merge_partial(fn: int, acc: Optional[float], value: Optional[float]) -> Optional[float]:
    if value is None:
        return acc
    if acc is None:
        return value
    if fn == _MIN:
        return _java_min(acc, value)
    if fn == _MAX:
        return _java_max(acc, value)
    return acc + value           # SUM, COUNT


def finish_partials(aggs: Sequence[int], recipe, partial: Sequence[Optional[float]]) -> List[Optional[float]]:
    """Partial slots -> final accumulator values (Accumulators.kt:26-107: empty => null, COUNT => count)."""
    out: List[Optional[float]] = []
    for a, slots in zip(aggs, recipe):
        if int(a) == _AVG:
            s, c = partial[slots[0]], partial[slots[1]]
            out.append(None if not c or s is None else s / c)
        elif int(a) == _COUNT:
            out.append(partial[slots[0]] or 0.0)
        else:
            out.append(partial[slots[0]])
    return out


def combine_aggregate_partials(partial_fns: Sequence[int], per_rank: Sequence[Sequence[Optional[float]]]) -> List[Optional[float]]:
    """Fold the shards' partials in RANK order (a fixed order: every rank computes the identical result)."""
    acc: List[Optional[float]] = [None] * len(partial_fns)
    for values in per_rank:
        for j, fn in enumerate(partial_fns):
            acc[j] = _merge_partial(fn, acc[j], values[j])
    return acc


def allreduce_aggregates(local: Sequence[Optional[float]], partial_fns: Sequence[int], group=None, device=None, ctx=None):
    """All ranks contribute their partial accumulators ({value, valid} pairs, 16 B per aggregate) with one all-gather and
    fold them in rank order.  MIN/MAX/COUNT are exact; SUM is the sum of the shards' sums (deterministic for a world size)."""
    import torch
    import torch.distributed as dist
    if ctx is not None and ctx.comm_nranks > 0:       # the C ABI's own all-gather (RCCL), no torch in the path
        payload = np.array([[0.0 if v is None else float(v), 0.0 if v is None else 1.0] for v in local], dtype=np.float64)
        per_rank = []
        for blob in ctx.allgather_host(payload.tobytes()):
            h = np.frombuffer(blob, dtype=np.float64).reshape(-1, 2)
            per_rank.append([float(h[j, 0]) if h[j, 1] != 0.0 else None for j in range(h.shape[0])])
        return combine_aggregate_partials(partial_fns, per_rank)
    world = dist.get_world_size(group)
    mine = torch.tensor([[0.0 if v is None else float(v), 0.0 if v is None else 1.0] for v in local],
                        dtype=torch.float64, device=device).reshape(-1, 2)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    per_rank = []
    for p in parts:
        h = p.cpu().numpy()
        per_rank.append([float(h[j, 0]) if h[j, 1] != 0.0 else None for j in range(h.shape[0])])
    return combine_aggregate_partials(partial_fns, per_rank)


def sharded_filter_aggregate(ctx, batch, cf, exprs, aggs: Sequence[int], group=None):
    """GlobalAggregation over a row-range sharded table: the fused filter+aggregate kernel on every rank's shard, then ONE
    small all-gather.  `exprs` are CompiledExpressions (one per aggregate).  Returns the final values on every rank."""
    import torch
    from . import engine as E
    fns, src, recipe = expand_partial_aggregates(aggs)
    local, _ = E.filter_aggregate(ctx, batch, cf, [exprs[i] for i in src], fns)
    merged = allreduce_aggregates(local, fns, group, device=torch.device("cuda", ctx.device), ctx=ctx)
    return finish_partials(aggs, recipe, merged)


def merge_group_partials(partial_fns: Sequence[int], per_rank_groups):
    """Group-by across shards.  per_rank_groups[r] = list of (key tuple, [partial values]) in rank r's insertion order.
    Shards are contiguous row ranges in rank order, so the global first occurrence of a key is in the lowest rank that
    has it: walking the ranks in order and appending unseen keys reproduces the reference's LinkedHashMap order
    (GroupByAggregationOperator.kt:22)."""
    merged = {}
    for groups in per_rank_groups:
        for key, values in groups:
            key = tuple(_nan_key(k) for k in key)
            acc = merged.get(key)
            if acc is None:
                merged[key] = list(values)
            else:
                for j, fn in enumerate(partial_fns):
                    acc[j] = _merge_partial(fn, acc[j], values[j])
    return [(tuple(None if k is _NAN else k for k in key), acc) for key, acc in merged.items()]


class _NaN:
    def __repr__(self):
        return "NaN"


_NAN = _NaN()


def _nan_key(k):
    """Double.equals: all NaNs are one key (and -0.0 != 0.0, which Python's dict would merge -- keys here are
    dictionary strings / booleans, floats only pass through for completeness)."""
    if isinstance(k, float) and k != k:
        return _NAN
    return k


def allgather_groups(local_groups, partial_fns: Sequence[int], group=None):
    """Every rank receives every shard's (small) group table and merges in rank order."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    tables = [None] * world
    dist.all_gather_object(tables, local_groups, group=group)
    return merge_group_partials(partial_fns, tables)


def sharded_filter_groupby(ctx, batch, cf, keys, exprs, aggs: Sequence[int], group=None):
    """GroupByAggregation over a sharded table: fused group-by kernel per shard, then one all-gather of the group tables.
    Returns rows [key values..., aggregate values...] in global insertion order on every rank."""
    from . import engine as E
    fns, src, recipe = expand_partial_aggregates(aggs)
    res = E.filter_groupby(ctx, batch, cf, keys, [exprs[i] for i in src], fns)
    cols = res.to_columns()
    res.free()
    nk = len(keys)
    local = [(tuple(c.value(i) for c in cols[:nk]), [c.value(i) for c in cols[nk:]]) for i in range(len(cols[0]) if cols else 0)]
    merged = allgather_groups(local, fns, group)
    return [list(k) + finish_partials(aggs, recipe, acc) for k, acc in merged]


def time_gather(ctx, batch, cf, cp, world: int, rank: int, reps: int = 3):
    """bench.py: time the materialising exchange (qe_gather through the C ABI) separately from the scan, and the scan +
    gather back to back ("for cfg 5 with and without the RCCL gather")."""
    import torch
    import torch.distributed as dist
    from . import engine as E

    def barrier():
        ctx.synchronize()
        if world > 1:
            dist.barrier()

    res = E.filter_project(ctx, batch, cf, cp)
    best, best_both = None, None
    rows_on_root, nbytes = 0, 0
    for _ in range(reps):
        barrier()
        t0 = time.perf_counter()
        g = ctx.gather(res, 0)
        barrier()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        if g is not None:
            rows_on_root = g.count
            nbytes = 0
            for c in range(g.ncols):
                v = g.view(c)
                nbytes += (g.count + 63) // 64 * 8 if v.type == 2 else g.count * (8 if v.type in (1, 3) else 4)
                if v.validity:
                    nbytes += (g.count + 63) // 64 * 8
            g.free()
    res.free()
    for _ in range(reps):
        barrier()
        t0 = time.perf_counter()
        r = E.filter_project(ctx, batch, cf, cp)
        g = ctx.gather(r, 0)
        barrier()
        dt = time.perf_counter() - t0
        best_both = dt if best_both is None else min(best_both, dt)
        if g is not None:
            g.free()
        r.free()
    return {"ms": best * 1e3, "scan_plus_gather_ms": best_both * 1e3, "rows_on_root": rows_on_root, "bytes_on_root": nbytes,
            "gbps_into_root": (nbytes / best / 1e9) if best else None,
            "method": "qe_gather (C ABI, RCCL direct): ncclAllGather of result headers + one ncclGroupStart/End of "
                      "ncclRecv at final offsets / ncclSend per column (direct peer links), rank order, bitmap words"}


def time_gather_overlapped(ctx, batch, cf, cp, world: int, rank: int, nslices: int = 8, reps: int = 3):
    """bench.py: the scan and the exchange in ONE overlapped call (qe_filter_project_gather: count pre-pass, the shard scanned
    in slices, every slice's rows on their way to the root while the next slice is scanned), to be read against
    time_gather's scan_plus_gather_ms."""
    import torch.distributed as dist

    def barrier():
        ctx.synchronize()
        if world > 1:
            dist.barrier()

    best, rows_on_root = None, 0
    for _ in range(reps):
        barrier()
        t0 = time.perf_counter()
        g = ctx.filter_project_gather(batch, cf, cp, 0, nslices)
        barrier()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        if g is not None:
            rows_on_root = g.count
            g.free()
    return {"ms": best * 1e3, "rows_on_root": rows_on_root, "slices": nslices,
            "method": "qe_filter_project_gather (C ABI): count pre-pass + one all-gather of per-slice counts, then per slice the scan "
                      "(compute stream) and grouped ncclSend / ncclRecv at final offsets (copy stream)"}

"""Object layer over the C ABI: Context, Dictionary, DeviceBatch, CompiledExpression, Result.

Each class owns one opaque handle of include/qe_hip.h and frees it with the
matching ``*_free``.  Nothing here computes: all evaluation happens in
libqe_hip.so on the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from . import ast as A
from . import native as N
from .datatypes import DataType
from .program import serialize
from .table import Column, pack_bitmap, unpack_bitmap

_NP = {DataType.DOUBLE: np.float64, DataType.INT64: np.int64, DataType.INT32: np.int32, DataType.STRING: np.int32}


class Context:
    """qe_ctx: one device + one HIP stream.  ``device=None`` -> planning-only (QE_DEVICE_NONE)."""

    def __init__(self, device: Optional[int] = 0, exec_mode: int = N.EXEC_FUSED, cmp_semantics: int = N.CMP_TOTAL_ORDER,
                 profile: bool = False, result_capacity_rows: int = 0, jit_cache_dir: Optional[str] = None,
                 tuning: Sequence[int] = ()):
        self._lib = N.lib()
        opts = N.Options()
        opts.struct_size = C.sizeof(N.Options)
        opts.exec_mode = exec_mode
        opts.cmp_semantics = cmp_semantics
        opts.profile = 1 if profile else 0
        opts.result_capacity_rows = result_capacity_rows
        opts.jit_cache_dir = jit_cache_dir.encode() if jit_cache_dir else None
        for i, t in enumerate(tuning):
            opts.tuning[i] = int(t)
        h = C.c_void_p()
        st = self._lib.qe_ctx_create(N.DEVICE_NONE if device is None else device, C.byref(opts), C.byref(h))
        N.check(None, st)
        self.handle = h
        self.device = device
        self._dicts: Dict[tuple, "Dictionary"] = {}

    def close(self) -> None:
        if getattr(self, "handle", None):
            self._dicts.clear()
            self._lib.qe_ctx_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_exec_mode(self, mode: int) -> None:
        N.check(self.handle, self._lib.qe_ctx_set_exec_mode(self.handle, mode))

    def set_cmp_semantics(self, mode: int) -> None:
        N.check(self.handle, self._lib.qe_ctx_set_cmp_semantics(self.handle, mode))

    def kernel_time(self):
        last, total, n = C.c_double(), C.c_double(), C.c_int64()
        N.check(self.handle, self._lib.qe_ctx_kernel_time(self.handle, C.byref(last), C.byref(total), C.byref(n)))
        return last.value, total.value, n.value

    def reset_kernel_time(self) -> None:
        N.check(self.handle, self._lib.qe_ctx_reset_kernel_time(self.handle))

    @property
    def last_form(self) -> int:
        """qe_ctx_last_form: native.FORM_* of the last filter_project on this context (-1: none yet)."""
        return int(self._lib.qe_ctx_last_form(self.handle))

    def synchronize(self) -> None:
        N.check(self.handle, self._lib.qe_ctx_synchronize(self.handle))

    def trim(self) -> None:
        N.check(self.handle, self._lib.qe_ctx_trim(self.handle))

    def stream_read_bandwidth(self, nbytes: int = 1 << 30, reps: int = 5) -> float:
        out = C.c_double()
        N.check(self.handle, self._lib.qe_stream_read_bandwidth(self.handle, nbytes, reps, C.byref(out)))
        return out.value

    def stream_read_write_time(self, nbytes: int, write_every: int, reps: int = 5):
        """(ms, bytes written) of a read stream with a trickle of writes: roofline calibration."""
        ms, wb = C.c_double(), C.c_double()
        N.check(self.handle, self._lib.qe_stream_read_write_time(self.handle, nbytes, write_every, reps, C.byref(ms), C.byref(wb)))
        return ms.value, wb.value

    # ---- the exchange step of a sharded scan (qe_comm_* / qe_gather) ----
    def comm_unique_id(self) -> bytes:
        """ncclUniqueId created by this rank (rank 0 hands it to the others through the host's own channel)."""
        buf = C.create_string_buffer(N.COMM_ID_BYTES)
        N.check(self.handle, self._lib.qe_comm_unique_id(self.handle, buf))
        return buf.raw

    def comm_init(self, nranks: int, rank: int, unique_id: bytes) -> None:
        assert len(unique_id) == N.COMM_ID_BYTES
        buf = C.create_string_buffer(unique_id, N.COMM_ID_BYTES)
        N.check(self.handle, self._lib.qe_comm_init(self.handle, nranks, rank, buf))

    @property
    def comm_rank(self) -> int:
        return int(self._lib.qe_comm_rank(self.handle))

    @property
    def comm_nranks(self) -> int:
        return int(self._lib.qe_comm_nranks(self.handle))

    def comm_destroy(self) -> None:
        self._lib.qe_comm_destroy(self.handle)

    def gather(self, result: "Result", root: int = 0) -> Optional["Result"]:
        """qe_gather (collective): the concatenation of every rank's result in rank order on `root`, None elsewhere."""
        h = C.c_void_p()
        N.check(self.handle, self._lib.qe_gather(self.handle, result.handle, root, C.byref(h)))
        return Result(self, h) if h.value else None

    def filter_project_gather(self, batch: "DeviceBatch", filter: Optional["CompiledExpression"], projections: Sequence["CompiledExpression"],
                              root: int = 0, nslices: int = 0) -> Optional["Result"]:
        """qe_filter_project_gather (collective): scan of this rank's shard in slices with every slice's rows travelling to
        `root` while the next slice is scanned; the whole result on `root`, None elsewhere."""
        h = C.c_void_p()
        N.check(self.handle, self._lib.qe_filter_project_gather(self.handle, batch.handle, filter.handle if filter else None,
                                                                _expr_array(projections), len(projections), root, nslices, C.byref(h)))
        return Result(self, h) if h.value else None

    def allgather_host(self, payload: bytes) -> List[bytes]:
        """qe_comm_allgather_host (collective): every rank's `payload` (equal sizes), in rank order."""
        n = self.comm_nranks
        recv = C.create_string_buffer(len(payload) * n)
        N.check(self.handle, self._lib.qe_comm_allgather_host(self.handle, payload, len(payload), recv))
        return [recv.raw[i * len(payload):(i + 1) * len(payload)] for i in range(n)]

    def order_by(self, result: "Result", column: int) -> "Result":
        """qe_result_order_by: the rows of `result` sorted stably by `column` (0-based) with compareValues order."""
        h = C.c_void_p()
        N.check(self.handle, self._lib.qe_result_order_by(self.handle, result.handle, column, C.byref(h)))
        return Result(self, h)

    def concat(self, parts: Sequence["Result"]) -> "Result":
        """qe_result_concat: results of this device, concatenated in the given order."""
        arr = (C.c_void_p * max(1, len(parts)))(*[p.handle for p in parts])
        h = C.c_void_p()
        N.check(self.handle, self._lib.qe_result_concat(self.handle, arr, len(parts), C.byref(h)))
        return Result(self, h)

    def dictionary(self, entries: Sequence[str]) -> "Dictionary":
        key = tuple(entries)
        d = self._dicts.get(key)
        if d is None:
            d = Dictionary(self, entries)
            self._dicts[key] = d
        return d

    def compile(self, expr: A.Expression) -> "CompiledExpression":
        return CompiledExpression(self, expr)


class Dictionary:
    def __init__(self, ctx: Context, entries: Sequence[str]):
        self.ctx = ctx
        self.entries = list(entries)
        enc = [s.encode("utf-8") for s in self.entries]
        arr = (C.c_char_p * max(1, len(enc)))(*enc)
        h = C.c_void_p()
        N.check(ctx.handle, ctx._lib.qe_dict_create(ctx.handle, len(enc), arr, C.byref(h)))
        self.handle = h

    def __del__(self):
        try:
            if self.handle and self.ctx.handle:
                self.ctx._lib.qe_dict_free(self.ctx.handle, self.handle)
        except Exception:
            pass
        self.handle = None


def _col_descs(ctx: Context, columns: Sequence[Column], keep: list, schema_only: bool = False):
    descs = (N.ColDesc * max(1, len(columns)))()
    for j, c in enumerate(columns):
        descs[j].type = int(c.type)
        if c.type == DataType.BOOLEAN:
            data = pack_bitmap(c.data)
        else:
            data = np.ascontiguousarray(c.data)
        keep.append(data)
        descs[j].data = None if schema_only else data.ctypes.data
        if c.valid is not None:
            v = pack_bitmap(c.valid)
            keep.append(v)
            descs[j].validity = v.ctypes.data
        if c.type == DataType.STRING:
            d = ctx.dictionary(c.dictionary)
            keep.append(d)
            descs[j].dict = d.handle
    return descs


class DeviceBatch:
    """qe_batch: columns resident in HBM ("pin once")."""

    def __init__(self, ctx: Context, handle, keep=None):
        self.ctx = ctx
        self.handle = handle
        self._keep = keep

    @staticmethod
    def from_columns(ctx: Context, columns: Sequence[Column]) -> "DeviceBatch":
        keep: list = []
        nrows = len(columns[0]) if columns else 0
        descs = _col_descs(ctx, columns, keep)
        h = C.c_void_p()
        N.check(ctx.handle, ctx._lib.qe_batch_create(ctx.handle, nrows, len(columns), descs, C.byref(h)))
        return DeviceBatch(ctx, h, [k for k in keep if isinstance(k, Dictionary)])

    @staticmethod
    def describe(ctx: Context, columns: Sequence[Column]) -> "DeviceBatch":
        """Schema-only batch for plan-time work (no device memory)."""
        keep: list = []
        nrows = len(columns[0]) if columns else 0
        descs = _col_descs(ctx, columns, keep, schema_only=True)
        h = C.c_void_p()
        N.check(ctx.handle, ctx._lib.qe_batch_describe(ctx.handle, nrows, len(columns), descs, C.byref(h)))
        return DeviceBatch(ctx, h, keep)

    @staticmethod
    def generate(ctx: Context, specs: Sequence[N.GenSpec], nrows: int, row_begin: int = 0, seed: int = 42,
                 keep=None) -> "DeviceBatch":
        arr = (N.GenSpec * max(1, len(specs)))(*specs)
        h = C.c_void_p()
        N.check(ctx.handle, ctx._lib.qe_batch_generate(ctx.handle, seed, row_begin, nrows, len(specs), arr, C.byref(h)))
        return DeviceBatch(ctx, h, keep)

    @property
    def nrows(self) -> int:
        return int(self.ctx._lib.qe_batch_nrows(self.handle))

    @property
    def ncols(self) -> int:
        return int(self.ctx._lib.qe_batch_ncols(self.handle))

    def column_type(self, col: int) -> DataType:
        return DataType(self.ctx._lib.qe_batch_column_type(self.handle, col))

    def column_to_host(self, col: int, row_begin: int = 0, nrows: Optional[int] = None, dictionary=None) -> Column:
        n = self.nrows - row_begin if nrows is None else nrows
        t = self.column_type(col)
        nwords = (n + 63) // 64
        valid_words = np.zeros(max(1, nwords), dtype=np.uint64)
        data = np.zeros(max(1, nwords), dtype=np.uint64) if t == DataType.BOOLEAN else np.zeros(max(1, n), dtype=_NP[t])
        N.check(self.ctx.handle, self.ctx._lib.qe_batch_column_to_host(
            self.ctx.handle, self.handle, col, row_begin, n, data.ctypes.data, valid_words.ctypes.data))
        valid = unpack_bitmap(valid_words, n)
        vals = unpack_bitmap(data, n) if t == DataType.BOOLEAN else data[:n]
        return Column(t, vals, valid, dictionary)

    def free(self) -> None:
        if self.handle and self.ctx.handle:
            self.ctx._lib.qe_batch_free(self.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class CompiledExpression:
    """qe_expr: the analogue of the RowCallable returned by compileExpression (Compiler.kt:20-26)."""

    def __init__(self, ctx: Context, expr: A.Expression):
        self.ctx = ctx
        self.expr = expr
        prog = serialize(expr)
        h = C.c_void_p()
        N.check(ctx.handle, ctx._lib.qe_expr_compile(ctx.handle, prog, len(prog), C.byref(h)))
        self.handle = h

    @property
    def result_type(self) -> DataType:
        return DataType(self.ctx._lib.qe_expr_result_type(self.handle))

    def __del__(self):
        try:
            if self.handle and self.ctx.handle:
                self.ctx._lib.qe_expr_free(self.ctx.handle, self.handle)
        except Exception:
            pass
        self.handle = None


def _expr_array(exprs: Sequence[CompiledExpression]):
    return (C.c_void_p * max(1, len(exprs)))(*[e.handle for e in exprs])


class Result:
    """qe_result: compacted output columns in HBM + row count."""

    def __init__(self, ctx: Context, handle):
        self.ctx = ctx
        self.handle = handle

    @property
    def count(self) -> int:
        return int(self.ctx._lib.qe_result_count(self.handle))

    @property
    def ncols(self) -> int:
        return int(self.ctx._lib.qe_result_ncols(self.handle))

    def view(self, col: int) -> N.ColView:
        v = N.ColView()
        N.check(self.ctx.handle, self.ctx._lib.qe_result_column(self.handle, col, C.byref(v)))
        return v

    def column_to_host(self, col: int) -> Column:
        v = self.view(col)
        t = DataType(v.type)
        n = int(v.count)
        nwords = (n + 63) // 64
        valid_words = np.zeros(max(1, nwords), dtype=np.uint64)
        data = np.zeros(max(1, nwords), dtype=np.uint64) if t == DataType.BOOLEAN else np.zeros(max(1, n), dtype=_NP[t])
        N.check(self.ctx.handle, self.ctx._lib.qe_result_column_to_host(
            self.ctx.handle, self.handle, col, data.ctypes.data, valid_words.ctypes.data))
        valid = unpack_bitmap(valid_words, n)
        vals = unpack_bitmap(data, n) if t == DataType.BOOLEAN else data[:n]
        dictionary = None
        if t == DataType.STRING:
            m = self.ctx._lib.qe_dict_size(v.dict)
            dictionary = [self.ctx._lib.qe_dict_entry(v.dict, i).decode("utf-8") for i in range(m)]
        return Column(t, vals, valid, dictionary)

    def to_columns(self) -> List[Column]:
        return [self.column_to_host(i) for i in range(self.ncols)]

    def to_host(self) -> "HostResult":
        """qe_result_to_host: START copying every column into pinned host memory owned by the library (copy stream) and
        return at once; HostResult.wait() blocks until the bytes are there."""
        h = C.c_void_p()
        N.check(self.ctx.handle, self.ctx._lib.qe_result_to_host(self.ctx.handle, self.handle, C.byref(h)))
        return HostResult(self.ctx, h, self)

    def free(self) -> None:
        if self.handle and self.ctx.handle:
            self.ctx._lib.qe_result_free(self.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class HostResult:
    """qe_host_result: the columns of a result in PINNED host memory owned by the library (zero-copy numpy views)."""

    def __init__(self, ctx: Context, handle, source: Result):
        self.ctx = ctx
        self.handle = handle
        self._source = source        # the device result must outlive the copies

    def wait(self) -> "HostResult":
        N.check(self.ctx.handle, self.ctx._lib.qe_host_result_wait(self.ctx.handle, self.handle))
        self._source = None
        return self

    @property
    def count(self) -> int:
        return int(self.ctx._lib.qe_host_result_count(self.handle))

    @property
    def ncols(self) -> int:
        return int(self.ctx._lib.qe_host_result_ncols(self.handle))

    def column_views(self, col: int):
        """(values, validity words | None) as numpy arrays VIEWING the pinned buffers (valid until free()); BOOLEAN values
        and validity are bitmap words (row i = word i >> 6, bit i & 63)."""
        self.wait()
        v = N.ColView()
        N.check(self.ctx.handle, self.ctx._lib.qe_host_result_column(self.handle, col, C.byref(v)))
        t = DataType(v.type)
        n = int(v.count)
        nwords = (n + 63) // 64

        def view(ptr, npdt, cnt):
            if cnt == 0 or not ptr:
                return np.zeros(0, dtype=npdt)
            buf = (C.c_char * (cnt * np.dtype(npdt).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=npdt, count=cnt)
        data = view(v.data, np.uint64, nwords) if t == DataType.BOOLEAN else view(v.data, _NP[t], n)
        valid = view(v.validity, np.uint64, nwords) if v.validity else None
        return data, valid

    def column(self, col: int) -> Column:
        """A Column that OWNS its data (copied out of the pinned buffers), for tests."""
        data, valid = self.column_views(col)
        v = N.ColView()
        N.check(self.ctx.handle, self.ctx._lib.qe_host_result_column(self.handle, col, C.byref(v)))
        t = DataType(v.type)
        n = int(v.count)
        vals = unpack_bitmap(data, n) if t == DataType.BOOLEAN else data.copy()
        dictionary = None
        if t == DataType.STRING:
            m = self.ctx._lib.qe_dict_size(v.dict)
            dictionary = [self.ctx._lib.qe_dict_entry(v.dict, i).decode("utf-8") for i in range(m)]
        return Column(t, vals, unpack_bitmap(valid, n) if valid is not None else None, dictionary)

    def free(self) -> None:
        if self.handle and self.ctx.handle:
            self.ctx._lib.qe_host_result_free(self.ctx.handle, self.handle)
        self.handle = None
        self._source = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def filter_project(ctx: Context, batch: DeviceBatch, filter: Optional[CompiledExpression],
                   projections: Sequence[CompiledExpression]) -> Result:
    """qe_filter_project: Projection(Filter(Scan)) in one call."""
    h = C.c_void_p()
    N.check(ctx.handle, ctx._lib.qe_filter_project(ctx.handle, batch.handle, filter.handle if filter else None,
                                                   _expr_array(projections), len(projections), C.byref(h)))
    return Result(ctx, h)


def prepare(ctx: Context, batch: DeviceBatch, filter: Optional[CompiledExpression],
            projections: Sequence[CompiledExpression]) -> None:
    N.check(ctx.handle, ctx._lib.qe_filter_project_prepare(ctx.handle, batch.handle, filter.handle if filter else None,
                                                           _expr_array(projections), len(projections)))


def chosen_geometry(ctx: Context, batch: DeviceBatch, filter: Optional[CompiledExpression],
                    projections: Sequence[CompiledExpression]):
    """qe_filter_project_geometry: (chosen: -1 undecided / 0 default / 1 wide, came from the JIT cache?)."""
    chosen, cached = C.c_int32(), C.c_int32()
    N.check(ctx.handle, ctx._lib.qe_filter_project_geometry(ctx.handle, batch.handle, filter.handle if filter else None,
                                                            _expr_array(projections), len(projections), C.byref(chosen),
                                                            C.byref(cached)))
    return int(chosen.value), bool(cached.value)


def conjunct_order(ctx: Context, batch: DeviceBatch, filter: Optional[CompiledExpression],
                   projections: Sequence[CompiledExpression]):
    """qe_filter_project_conjunct_order: None while the plan has not measured its conjuncts yet, else the list of written-order
    indices in evaluation order."""
    order = (C.c_int32 * 16)()
    n = C.c_int32()
    N.check(ctx.handle, ctx._lib.qe_filter_project_conjunct_order(ctx.handle, batch.handle, filter.handle if filter else None,
                                                                  _expr_array(projections), len(projections), order, 16, C.byref(n)))
    return None if n.value < 0 else [int(order[i]) for i in range(min(n.value, 16))]


def generated_source(ctx: Context, batch: DeviceBatch, filter: Optional[CompiledExpression],
                     projections: Sequence[CompiledExpression]) -> str:
    out = C.c_char_p()
    N.check(ctx.handle, ctx._lib.qe_filter_project_source(ctx.handle, batch.handle, filter.handle if filter else None,
                                                          _expr_array(projections), len(projections), C.byref(out)))
    return out.value.decode("utf-8")


def filter_aggregate(ctx: Context, batch: DeviceBatch, filter: Optional[CompiledExpression],
                     exprs: Sequence[CompiledExpression], aggs: Sequence[int]):
    """qe_filter_aggregate: returns ([float|None per aggregate], selected row count)."""
    n = len(exprs)
    vals = (C.c_double * max(1, n))()
    valid = (C.c_uint8 * max(1, n))()
    nsel = C.c_int64()
    fns = (C.c_int32 * max(1, n))(*[int(a) for a in aggs])
    N.check(ctx.handle, ctx._lib.qe_filter_aggregate(ctx.handle, batch.handle, filter.handle if filter else None,
                                                     _expr_array(exprs), fns, n, vals, valid, C.byref(nsel)))
    return [float(vals[i]) if valid[i] else None for i in range(n)], int(nsel.value)


def prepare_aggregate(ctx: Context, batch: DeviceBatch, filter: Optional[CompiledExpression],
                      exprs: Sequence[CompiledExpression], aggs: Sequence[int]) -> None:
    fns = (C.c_int32 * max(1, len(exprs)))(*[int(a) for a in aggs])
    N.check(ctx.handle, ctx._lib.qe_filter_aggregate_prepare(ctx.handle, batch.handle, filter.handle if filter else None,
                                                             _expr_array(exprs), fns, len(exprs)))


def filter_groupby(ctx: Context, batch: DeviceBatch, filter: Optional[CompiledExpression], keys: Sequence[CompiledExpression],
                   exprs: Sequence[CompiledExpression], aggs: Sequence[int]) -> Result:
    """qe_filter_groupby: key columns + DOUBLE aggregate columns, one row per group in insertion order."""
    h = C.c_void_p()
    fns = (C.c_int32 * max(1, len(exprs)))(*[int(a) for a in aggs])
    N.check(ctx.handle, ctx._lib.qe_filter_groupby(ctx.handle, batch.handle, filter.handle if filter else None,
                                                   _expr_array(keys), len(keys), _expr_array(exprs), fns, len(exprs), C.byref(h)))
    return Result(ctx, h)


def prepare_groupby(ctx: Context, batch: DeviceBatch, filter: Optional[CompiledExpression], keys: Sequence[CompiledExpression],
                    exprs: Sequence[CompiledExpression], aggs: Sequence[int]) -> None:
    fns = (C.c_int32 * max(1, len(exprs)))(*[int(a) for a in aggs])
    N.check(ctx.handle, ctx._lib.qe_filter_groupby_prepare(ctx.handle, batch.handle, filter.handle if filter else None,
                                                           _expr_array(keys), len(keys), _expr_array(exprs), fns, len(exprs)))

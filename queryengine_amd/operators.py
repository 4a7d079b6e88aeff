"""Physical operators: the reference's pull protocol on top of the GPU path.

``Operator`` is ``operator/Operators.kt:5-11`` (``open / next / close``; ``next()`` returns one boxed
row or ``None``); ``forEach / map / mapTo`` are ``:13-32``.  The two GPU operators do all their work in
``open()`` -- exactly like the reference's blocking operators (``GlobalAggregationOperator.kt:10-25``)
-- and box rows lazily in ``next()``.  Operators are re-openable (``MemorySourceOperator.kt:10-12``;
the JMH harness re-runs open/next/close on one plan, ``T/SimpleSumBenchmark.java:63-94``): the input
batch is pinned to HBM once per (table, context) and reused by every ``open()``.

There is no CPU evaluation here: expressions only ever run inside libqe_hip.so.
"""
from __future__ import annotations

import math
from typing import Any, Callable, List, Optional, Sequence

from . import engine as E
from .ast import AggregationFunction, Expression
from .table import Column, ColumnarTable


class Operator:
    def open(self) -> None:
        raise NotImplementedError

    def close(self) -> None:
        raise NotImplementedError

    def next(self) -> Optional[List[Any]]:
        raise NotImplementedError

    # Closeable.use
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def forEach(op: Operator, consumer: Callable[[List[Any]], None]) -> None:
    """operator/Operators.kt:13-23"""
    op.open()
    try:
        while True:
            row = op.next()
            if row is None:
                break
            consumer(row)
    finally:
        op.close()


def mapTo(op: Operator, result: list, mapper: Callable[[List[Any]], Any]) -> list:
    forEach(op, lambda row: result.append(mapper(row)))
    return result


def map(op: Operator, mapper: Callable[[List[Any]], Any]) -> list:   # noqa: A001 (reference name)
    return mapTo(op, [], mapper)


def _compare_key(value: Any):
    """Sort key with the order of Kotlin's ``compareValues`` on the boxed types of the engine: null first, then
    ``Double.compareTo`` (total order: -0.0 < 0.0, NaN greatest), ``String.compareTo`` (UTF-16 code units),
    ``Boolean.compareTo`` (false < true), integers numerically."""
    if value is None:
        return (0,)
    if isinstance(value, bool):
        return (1, int(value))
    if isinstance(value, float):
        if value != value:
            return (1, 1, 0.0, 0)
        return (1, 0, value, 0 if (value == 0.0 and math.copysign(1.0, value) < 0) else 1)
    if isinstance(value, str):
        return (1, value.encode("utf-16-be", "surrogatepass"))
    return (1, 0, value, 1)


class OrderByOperator(Operator):
    """operator/OrderByOperator.kt:5-31: drain the source in ``open()``, stable ``sortBy`` on one column.

    A consumer of the path's output.  When the source is a GPU operator whose result sits in HBM
    (``result()``), the rows are sorted there (qe_result_order_by: key images, stable radix sort, gather) and boxed
    afterwards; any other source is drained and sorted on the host with the same ``compareValues`` order."""

    def __init__(self, source: Operator, index: int):
        self.source = source
        self.index = index
        self._iter = None
        self._sorted: Optional[E.Result] = None

    def open(self) -> None:
        if hasattr(self.source, "result") and hasattr(self.source, "ctx"):
            self.source.open()
            try:
                res = self.source.result()
                self._sorted = self.source.ctx.order_by(res, self.index)
            finally:
                self.source.close()
            cols = self._sorted.to_columns()
            n = self._sorted.count
            self._iter = ([c.value(i) for c in cols] for i in range(n))
            return
        data = mapTo(self.source, [], lambda row: list(row))
        data.sort(key=lambda row: _compare_key(row[self.index]))   # list.sort is stable, like java.util.List.sort
        self._iter = iter(data)

    def close(self) -> None:
        self._iter = None
        if self._sorted is not None:
            self._sorted.free()
            self._sorted = None

    def next(self) -> Optional[List[Any]]:
        if self._iter is None:
            raise RuntimeError("Operator not opened")               # OrderByOperator.kt:23
        return next(self._iter, None)


class ColumnarScanOperator(Operator):
    """Scan leaf over a ColumnarTable (replaces MemorySourceOperator.kt:5-36).

    As a row source it reuses ONE row buffer and nulls it on close, like the reference
    (``:8,15,26-32``).  The GPU operators do not pull rows from it: they take ``columns()``."""

    def __init__(self, table: ColumnarTable, projection: Sequence[str]):
        self.table = table
        self.projection = list(projection)
        self._columns = [table.column(name) for name in self.projection]   # raises "Unknown field" like MemoryTable.kt:11
        self._idx = 0
        self._row: List[Any] = [None] * len(self._columns)

    def columns(self) -> List[Column]:
        return self._columns

    def open(self) -> None:
        self._idx = 0

    def close(self) -> None:
        for j in range(len(self._row)):
            self._row[j] = None

    def next(self) -> Optional[List[Any]]:
        i = self._idx
        if i >= self.table.nrows:
            return None
        self._idx = i + 1
        for j, c in enumerate(self._columns):
            self._row[j] = c.value(i)
        return self._row

    def device_batch(self, ctx: E.Context) -> E.DeviceBatch:
        """Pin the projected columns to HBM once per context; later opens reuse the batch."""
        cache = self.table.__dict__.setdefault("_device_batches", {})
        # batches of contexts that were closed meanwhile are dropped (their handles died with the context)
        for k in [k for k, v in cache.items() if v.ctx.handle is None or v.handle is None]:
            del cache[k]
        key = (id(ctx), tuple(self.projection))
        b = cache.get(key)
        # id() values are recycled after a Context is collected: a hit only counts if the batch belongs to THIS context
        if b is None or b.ctx is not ctx or b.handle is None or ctx.handle is None:
            b = E.DeviceBatch.from_columns(ctx, self._columns)
            cache[key] = b
        return b


class GpuFilterProjectOperator(Operator):
    """Projection(Filter(Scan)) fused into one GPU operator.

    Replaces FilterOperator (operator/FilterOperator.kt:5-26) + ProjectionOperator
    (operator/ProjectionOperator.kt:5-21) / the generated CompiledProjectionOperator
    (evaluator/BytecodeCompiler.kt:37-132).  ``filter`` may be None."""

    def __init__(self, ctx: E.Context, source: ColumnarScanOperator, filter: Optional[Expression],
                 projections: Sequence[Expression]):
        self.ctx = ctx
        self.source = source
        # compileExpression at plan time (Planner.kt:35,44)
        self._filter = ctx.compile(filter) if filter is not None else None
        self._projections = [ctx.compile(p) for p in projections]
        self._result: Optional[E.Result] = None
        self._columns: Optional[List[Column]] = None
        self._idx = 0
        self._count = 0

    def open(self) -> None:
        self.source.open()
        batch = self.source.device_batch(self.ctx)
        self._result = E.filter_project(self.ctx, batch, self._filter, self._projections)
        self._count = self._result.count
        self._columns = None
        self._idx = 0

    def result(self) -> E.Result:
        """The columnar result in HBM (valid until close())."""
        if self._result is None:
            raise RuntimeError("Operator not initialized")   # CsvSourceOperator.kt:49
        return self._result

    def next(self) -> Optional[List[Any]]:
        if self._result is None:
            raise RuntimeError("Operator not initialized")
        if self._columns is None:
            self._columns = self._result.to_columns()       # one D2H per column, then rows are boxed lazily
        i = self._idx
        if i >= self._count:
            return None
        self._idx = i + 1
        return [c.value(i) for c in self._columns]          # a fresh row per call (ProjectionOperator.kt:18)

    def close(self) -> None:
        if self._result is not None:
            self._result.free()
        self._result = None
        self._columns = None
        self.source.close()


class GpuGlobalAggregationOperator(Operator):
    """GlobalAggregation(Projection(Filter(Scan))) as one fused GPU reduction.

    Replaces GlobalAggregationOperator (operator/GlobalAggregationOperator.kt:7-36) over the inner
    projection: one result row, nulls skipped, empty input => null (COUNT => count)."""

    def __init__(self, ctx: E.Context, source: ColumnarScanOperator, filter: Optional[Expression],
                 expressions: Sequence[Expression], aggregateFunctions: Sequence[AggregationFunction]):
        self.ctx = ctx
        self.source = source
        self._filter = ctx.compile(filter) if filter is not None else None
        self._exprs = [ctx.compile(e) for e in expressions]
        self._aggs = [int(a) for a in aggregateFunctions]
        self._row: Optional[List[Any]] = None

    def open(self) -> None:
        self.source.open()
        batch = self.source.device_batch(self.ctx)
        vals, _ = E.filter_aggregate(self.ctx, batch, self._filter, self._exprs, self._aggs)
        # CountAccumulator.finish returns an Int (Accumulators.kt:26-36)
        self._row = [int(v) if a == int(AggregationFunction.COUNT) else v for v, a in zip(vals, self._aggs)]

    def next(self) -> Optional[List[Any]]:
        res, self._row = self._row, None      # GlobalAggregationOperator.kt:32-36
        return res

    def close(self) -> None:
        self._row = None
        self.source.close()


class GpuGroupByAggregationOperator(Operator):
    """GroupByAggregation(Projection(Filter(Scan))) on the GPU.

    Replaces GroupByAggregationOperator (operator/GroupByAggregationOperator.kt:7-76) over the inner projection:
    one row [keys..., finished accumulators...] per group, in insertion order of the groups; null is a key value."""

    def __init__(self, ctx: E.Context, source: ColumnarScanOperator, filter: Optional[Expression],
                 keyExpressions: Sequence[Expression], expressions: Sequence[Expression],
                 aggregateFunctions: Sequence[AggregationFunction]):
        self.ctx = ctx
        self.source = source
        self._filter = ctx.compile(filter) if filter is not None else None
        self._keys = [ctx.compile(e) for e in keyExpressions]
        self._exprs = [ctx.compile(e) for e in expressions]
        self._aggs = [int(a) for a in aggregateFunctions]
        self._rows: Optional[List[List[Any]]] = None
        self._idx = 0

    def open(self) -> None:
        self.source.open()
        batch = self.source.device_batch(self.ctx)
        res = E.filter_groupby(self.ctx, batch, self._filter, self._keys, self._exprs, self._aggs)
        cols = res.to_columns()
        res.free()
        nk = len(self._keys)
        rows = []
        for i in range(len(cols[0]) if cols else 0):
            row = [c.value(i) for c in cols]
            for a, fn in enumerate(self._aggs):          # CountAccumulator.finish returns an Int (Accumulators.kt:26-36)
                if fn == int(AggregationFunction.COUNT) and row[nk + a] is not None:
                    row[nk + a] = int(row[nk + a])
            rows.append(row)
        self._rows = rows
        self._idx = 0

    def next(self) -> Optional[List[Any]]:
        if self._rows is None:
            raise RuntimeError("Operator not opened")      # GroupByAggregationOperator.kt:57
        if self._idx >= len(self._rows):
            return None
        self._idx += 1
        return self._rows[self._idx - 1]

    def close(self) -> None:
        self._rows = None
        self.source.close()


class GpuFinishProjectionOperator(Operator):
    """The projection the reference puts on top of an aggregation (``groupByFinish``, RewriteAggregates.kt:29-47):
    re-orders / combines the few aggregated rows.  The rows of the blocking source are pinned as one small batch and
    the expressions run on the GPU like any other projection (no CPU evaluation)."""

    def __init__(self, ctx: E.Context, source: Operator, sourceTypes: Sequence, expressions: Sequence[Expression]):
        self.ctx = ctx
        self.source = source
        self.sourceTypes = list(sourceTypes)
        self.expressions = list(expressions)
        self._compiled = [ctx.compile(e) for e in expressions]
        self._rows: Optional[List[List[Any]]] = None
        self._idx = 0

    def open(self) -> None:
        from .ast import ColumnExpression
        rows = map(self.source, lambda r: list(r))
        if not rows:
            self._rows = []
            self._idx = 0
            return
        cols = []
        for j, t in enumerate(self.sourceTypes):
            vals = [r[j] for r in rows]
            if t.name == "DOUBLE":
                vals = [float(v) if v is not None else None for v in vals]   # COUNT arrives as an Int
            cols.append(Column.from_values(t, vals))
        batch = E.DeviceBatch.from_columns(self.ctx, cols)
        res = E.filter_project(self.ctx, batch, None, self._compiled)
        out = res.to_columns()
        res.free()
        batch.free()
        result = []
        for i in range(len(rows)):
            row = [c.value(i) for c in out]
            for k, e in enumerate(self.expressions):   # a bare reference to a COUNT column keeps the Int
                if isinstance(e, ColumnExpression) and isinstance(rows[i][e.index], int) and not isinstance(rows[i][e.index], bool):
                    row[k] = rows[i][e.index]
            result.append(row)
        self._rows = result
        self._idx = 0

    def next(self) -> Optional[List[Any]]:
        if self._rows is None:
            raise RuntimeError("Operator not initialized")
        if self._idx >= len(self._rows):
            return None
        self._idx += 1
        return self._rows[self._idx - 1]

    def close(self) -> None:
        self._rows = None

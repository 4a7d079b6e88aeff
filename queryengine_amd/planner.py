"""Logical plan nodes and the physical-plan dispatch for the GPU modes.

``LogicalNode`` classes mirror ``evaluator/LogicalPlan.kt:7-12``.  ``buildLogicalPlan`` restates the
pipeline of ``evaluator/Planner.kt:7-28`` for the plan shapes the hot path covers (initialPlan ->
resolveSchema -> typeCheck -> aggregate split for all-aggregate select lists -> identity projection
removal); ``buildPhysicalPlan`` is the drop-in for the Filter/Projection branches of
``Planner.kt:30-63``: it pattern-matches Projection(Filter(Scan)) / Projection(Scan) / Filter(Scan)
and emits ONE fused GPU operator.  ``Mode`` extends ``evaluator/Compiler.kt:5-7``
with the GPU modes; the three JVM modes are not implemented here on purpose (no CPU path).
"""
from __future__ import annotations

import enum
from dataclasses import dataclass
from typing import Any, List, Optional, Sequence

from . import engine as E
from . import native as N
from .ast import (AggregationFunction, AggregationFunctionExpression, ColumnExpression, DefaultExpressionVisitor,
                  Expression, FunctionExpression, IdentifierExpression)
from .datatypes import Field, Schema
from .operators import (ColumnarScanOperator, GpuFilterProjectOperator, GpuFinishProjectionOperator,
                        GpuGlobalAggregationOperator, GpuGroupByAggregationOperator, Operator, OrderByOperator)
from .operators import map as op_map
from .sql import Query, parseQuery
from .table import Table, TableRegistry
from .typecheck import typeCheck


class Mode(enum.Enum):
    GPU_FUSED = "gpu_fused"          # JIT-fused single-pass kernel (the BYTECODE_COMPILER analogue)
    GPU_PER_NODE = "gpu_per_node"    # kernel per expression node (the INTERPRETER analogue)


class SchemaException(RuntimeError):
    """evaluator/ResolveSchema.kt:9"""


class LogicalNode:
    pass


@dataclass(frozen=True)
class LogicalScanNode(LogicalNode):
    table: str
    schema: Schema


@dataclass(frozen=True)
class LogicalFilterNode(LogicalNode):
    source: LogicalNode
    filter: Expression


@dataclass(frozen=True)
class LogicalAggregationNode(LogicalNode):
    source: LogicalNode
    groupCount: int
    aggregateFunctions: tuple


@dataclass(frozen=True)
class LogicalOrderByNode(LogicalNode):
    """evaluator/LogicalPlan.kt:12; ``index`` is the 1-based select position of ``ORDER BY n``."""
    source: LogicalNode
    index: int


@dataclass(frozen=True)
class LogicalProjectionNode(LogicalNode):
    source: LogicalNode
    expressions: tuple


class _ResolveSchema(DefaultExpressionVisitor):
    """evaluator/ResolveSchema.kt:49-69: column slots are assigned in order of first use."""

    def __init__(self, schema: Schema):
        self.schema = schema
        self.fields: List[Field] = []

    def visitIdentifier(self, expr: IdentifierExpression) -> Expression:
        for idx, f in enumerate(self.fields):
            if f.name == expr.name:
                return ColumnExpression(f.name, idx, f.type)
        field = self.schema[expr.name]
        if field is None:
            raise SchemaException(f"Could not find field {expr.name}")
        self.fields.append(field)
        return ColumnExpression(field.name, len(self.fields) - 1, field.type)


class InvalidAggregatesException(RuntimeError):
    """evaluator/RewriteAggregates.kt:7"""


def _count_aggregates(expr: Expression) -> int:
    """RewriteAggregates.kt:57-82 (CountAggregates)"""
    if isinstance(expr, AggregationFunctionExpression):
        if any(_count_aggregates(o) > 0 for o in expr.operands):
            raise InvalidAggregatesException("Nested aggregates are not allowed")
        return 1
    if isinstance(expr, FunctionExpression):
        return sum(_count_aggregates(o) for o in expr.operands)
    return 0


class _RewriteAggregates(DefaultExpressionVisitor):
    """RewriteAggregates.kt:84-97: every aggregate call becomes a column of the aggregation's output."""

    def __init__(self, groupExpressionCount: int):
        self.groupExpressionCount = groupExpressionCount
        self.aggregateInputs: List[Expression] = []
        self.aggregateFunctions: List[AggregationFunction] = []

    def visitAggregationFunction(self, expr):
        ops = self.visitOperands(expr.operands)
        idx = len(self.aggregateFunctions)
        self.aggregateFunctions.append(expr.function)
        self.aggregateInputs.append(ops[0])
        return ColumnExpression(expr.function.name, self.groupExpressionCount + idx, expr.dataType)


def rewriteAggregates(plan: "LogicalProjectionNode") -> Optional[LogicalNode]:
    """RewriteAggregates.kt:19-50: Projection with aggregates -> Projection[finish](Aggregation(Projection[inputs])).
    Non-aggregate select items are the (implicit) GROUP BY keys.  Returns None when there is no aggregate."""
    if isinstance(plan.source, LogicalFilterNode) and _count_aggregates(plan.source.filter) > 0:
        raise InvalidAggregatesException("Aggregate expressions not allowed in where clause")
    classified = [(e, _count_aggregates(e)) for e in plan.expressions]
    if sum(c for _, c in classified) == 0:
        return None
    groupCount = sum(1 for _, c in classified if c == 0)
    rw = _RewriteAggregates(groupCount)
    aggregateInput: List[Expression] = []
    groupByFinish: List[Expression] = []
    for expr, count in classified:
        if count > 0:
            groupByFinish.append(expr.accept(rw))
        else:
            index = len(aggregateInput)
            aggregateInput.append(expr)
            groupByFinish.append(ColumnExpression(f"_{index}", index, expr.dataType))
    aggregateInput.extend(rw.aggregateInputs)
    inp = LogicalProjectionNode(plan.source, tuple(aggregateInput))
    aggregate = LogicalAggregationNode(inp, groupCount, tuple(rw.aggregateFunctions))
    return LogicalProjectionNode(aggregate, tuple(groupByFinish))


def buildLogicalPlan(tableRegistry: TableRegistry, query: Query) -> LogicalNode:
    """Planner.kt:8-28 for the shapes of the hot path; ORDER BY wraps the finished plan (Planner.kt:13)."""
    if query.orderByColumn is not None:
        inner = buildLogicalPlan(tableRegistry, Query(query.select, query.from_, query.filter, None))
        return LogicalOrderByNode(inner, query.orderByColumn)
    schema = tableRegistry.getSchema(query.from_)
    resolver = _ResolveSchema(schema)
    # rewritePlan visits the Projection before its source (ResolveSchema.kt:24-33): SELECT list first, then WHERE
    select = [typeCheck(e.accept(resolver)) for e in query.select]
    flt = typeCheck(query.filter.accept(resolver)) if query.filter is not None else None
    scan: LogicalNode = LogicalScanNode(query.from_, Schema(resolver.fields))
    source = LogicalFilterNode(scan, flt) if flt is not None else scan
    rewritten = rewriteAggregates(LogicalProjectionNode(source, tuple(select)))
    if rewritten is not None:
        return rewritten
    # Optimizer.kt:33-35 removes an identity projection over the scan
    if flt is None and all(isinstance(e, ColumnExpression) and e.index == i for i, e in enumerate(select)) \
            and len(select) == len(resolver.fields):
        return scan
    return LogicalProjectionNode(source, tuple(select))


_contexts = {}


def default_context(mode: Mode, device: int = 0) -> E.Context:
    key = (mode, device)
    ctx = _contexts.get(key)
    if ctx is None or ctx.handle is None:
        ctx = E.Context(device=device, exec_mode=N.EXEC_FUSED if mode == Mode.GPU_FUSED else N.EXEC_PER_NODE)
        _contexts[key] = ctx
    return ctx


def _scan_of(tableRegistry: TableRegistry, node: LogicalScanNode) -> ColumnarScanOperator:
    op = tableRegistry.getTable(node.table).getScanOperator([f.name for f in node.schema.fields])   # Planner.kt:32
    if not isinstance(op, ColumnarScanOperator):
        raise TypeError("the GPU modes need a columnar scan leaf (ColumnarTable)")
    return op


def _match_filter_scan(node: LogicalNode):
    if isinstance(node, LogicalFilterNode) and isinstance(node.source, LogicalScanNode):
        return node.filter, node.source
    if isinstance(node, LogicalScanNode):
        return None, node
    return None


def buildPhysicalPlan(tableRegistry: TableRegistry, plan: LogicalNode, mode: Mode = Mode.GPU_FUSED,
                      ctx: Optional[E.Context] = None) -> Operator:
    """Planner.kt:30-63 with the Filter/Projection(/global Aggregation) subtree fused into one GPU operator."""
    ctx = ctx or default_context(mode)
    if isinstance(plan, LogicalOrderByNode):     # Planner.kt:58-61
        return OrderByOperator(buildPhysicalPlan(tableRegistry, plan.source, mode, ctx), plan.index - 1)
    if isinstance(plan, LogicalProjectionNode):
        m = _match_filter_scan(plan.source)
        if m is not None:
            flt, scan = m
            return GpuFilterProjectOperator(ctx, _scan_of(tableRegistry, scan), flt, plan.expressions)
    if isinstance(plan, (LogicalFilterNode, LogicalScanNode)):
        m = _match_filter_scan(plan)
        if m is not None:
            flt, scan = m
            # FilterOperator returns the scan row itself (FilterOperator.kt:21): every scan column passes through
            identity = [ColumnExpression(f.name, i, f.type) for i, f in enumerate(scan.schema.fields)]
            return GpuFilterProjectOperator(ctx, _scan_of(tableRegistry, scan), flt, identity)
    if isinstance(plan, LogicalAggregationNode) and isinstance(plan.source, LogicalProjectionNode):
        m = _match_filter_scan(plan.source.source)
        if m is not None:
            flt, scan = m
            exprs = plan.source.expressions
            if plan.groupCount == 0:   # Planner.kt:56-58
                return GpuGlobalAggregationOperator(ctx, _scan_of(tableRegistry, scan), flt, exprs, plan.aggregateFunctions)
            return GpuGroupByAggregationOperator(ctx, _scan_of(tableRegistry, scan), flt, exprs[:plan.groupCount],
                                                 exprs[plan.groupCount:], plan.aggregateFunctions)   # Planner.kt:54-55
    if isinstance(plan, LogicalProjectionNode) and isinstance(plan.source, LogicalAggregationNode):
        agg = plan.source
        source = buildPhysicalPlan(tableRegistry, agg, mode, ctx)
        from .datatypes import DataType
        types = [e.dataType for e in agg.source.expressions[:agg.groupCount]] + [DataType.DOUBLE] * len(agg.aggregateFunctions)
        if all(isinstance(e, ColumnExpression) and e.index == i for i, e in enumerate(plan.expressions)) \
                and len(plan.expressions) == len(types):
            return source                     # identity finish projection (Optimizer.kt:33-35)
        return GpuFinishProjectionOperator(ctx, source, types, plan.expressions)
    raise NotImplementedError(f"plan shape outside the GPU hot path: {plan!r}")


def query(registry, sql: str, mode: Mode = Mode.GPU_FUSED, table: Optional[Table] = None,
          ctx: Optional[E.Context] = None) -> List[List[Any]]:
    """Main.kt:11-26: ``query(registry, sql, mode)`` or ``query(tableName, table, sql, mode)``."""
    if isinstance(registry, str):
        if table is None:
            # positional form query(tableName, table, sql[, mode])
            raise TypeError("query(tableName, sql, table=...) needs a table")
        r = TableRegistry()
        r.register(registry, table)
        registry = r
    ast = parseQuery(sql)
    logical = buildLogicalPlan(registry, ast)
    physical = buildPhysicalPlan(registry, logical, mode, ctx)
    return op_map(physical, lambda row: row)

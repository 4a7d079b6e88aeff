// The seam: a GPU branch for buildPhysicalPlan (evaluator/Planner.kt:30-63).  The reference's
// Mode enum (evaluator/Compiler.kt:5-7) gains GPU; Projection(Filter(Scan)), Projection(Scan) and
// Filter(Scan) over a ColumnarSource collapse into ONE GpuFilterProjectOperator, Aggregation over them into ONE
// GpuAggregationOperator; every other plan
// shape falls through to the unmodified reference code.
package net.jhorstmann.queryengine.gpu

import net.jhorstmann.queryengine.ast.ColumnExpression
import net.jhorstmann.queryengine.data.TableRegistry
import net.jhorstmann.queryengine.evaluator.*
import net.jhorstmann.queryengine.operator.Operator
import java.lang.foreign.MemorySegment

fun buildGpuPhysicalPlan(ctx: MemorySegment, registry: TableRegistry, plan: LogicalNode): Operator {
    fun scanOf(n: LogicalNode): Pair<LogicalScanNode, net.jhorstmann.queryengine.ast.Expression?>? = when (n) {
        is LogicalScanNode -> n to null
        is LogicalFilterNode -> (n.source as? LogicalScanNode)?.let { it to n.filter }
        else -> null
    }
    // Aggregation(Projection(Filter(Scan))) -> one fused aggregate / group-by operator (Planner.kt:48-57)
    if (plan is LogicalAggregationNode) {
        val inner = plan.source as? LogicalProjectionNode
        val m = inner?.let { scanOf(it.source) }
        val table = m?.let { registry.getTable(it.first.table) }
        if (inner != null && m != null && table is ColumnarSource)
            return GpuAggregationOperator(ctx, table, m.second, inner.expressions, plan.groupCount, plan.aggregateFunctions)
    }
    val (projections, below) = when (plan) {
        is LogicalProjectionNode -> plan.expressions to plan.source
        else -> null to plan
    }
    val m = scanOf(below)
    if (m != null) {
        val (scan, filter) = m
        val table = registry.getTable(scan.table)
        if (table is ColumnarSource) {
            val exprs = projections ?: scan.schema.fields.mapIndexed { i, f -> ColumnExpression(f.name, i, f.type) }
            return GpuFilterProjectOperator(ctx, table, filter, exprs)
        }
    }
    // not on the GPU path: the reference's own planner (any of its three modes)
    return buildPhysicalPlan(registry, plan, Mode.BYTECODE_COMPILER)
}

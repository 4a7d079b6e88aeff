// Aggregation(Projection(Filter(Scan))) as ONE GPU operator behind the reference's Operator API: replaces
// GlobalAggregationOperator.kt:7-36 (groupCount == 0: one row, also over an empty input) and
// GroupByAggregationOperator.kt:7-76 (one row [keys..., accumulators...] per group in LinkedHashMap insertion order).
// Kotlin twin of queryengine_amd/operators.py:GpuGlobalAggregationOperator / GpuGroupByAggregationOperator and of
// queryengine_amd/host/qe_host.hpp:GpuAggregationOperator.  NOT compiled here (no JDK in the image).
package net.jhorstmann.queryengine.gpu

import net.jhorstmann.queryengine.ast.AggregationFunction
import net.jhorstmann.queryengine.ast.Expression
import net.jhorstmann.queryengine.operator.Operator
import java.lang.foreign.*
import java.lang.foreign.ValueLayout.*

class GpuAggregationOperator(
        private val ctx: MemorySegment,                 // qe_ctx*
        private val source: ColumnarSource,
        filter: Expression?,
        inputs: List<Expression>,                       // the inner projection: group keys first, then one input per aggregate
        private val groupCount: Int,
        private val functions: List<AggregationFunction>) : Operator() {

    private val arena = Arena.ofConfined()
    private val compiler = ExpressionCompiler(ctx, arena)
    private val filterExpr: MemorySegment = filter?.let { compiler.compile(it) } ?: MemorySegment.NULL
    private val exprs: MemorySegment = arena.allocate(ADDRESS, inputs.size.toLong()).also { arr ->
        inputs.forEachIndexed { i, e -> arr.setAtIndex(ADDRESS, i.toLong(), compiler.compile(e)) }
    }
    private val fns: MemorySegment = arena.allocate(JAVA_INT, maxOf(1, functions.size).toLong()).also { arr ->
        functions.forEachIndexed { i, f -> arr.setAtIndex(JAVA_INT, i.toLong(), f.ordinal) }   // MIN, MAX, SUM, COUNT, AVG = 0..4
    }
    private val nagg = functions.size
    private var batch: MemorySegment = MemorySegment.NULL
    private var rows: List<Array<Any?>>? = null
    private var iter: Iterator<Array<Any?>>? = null

    private fun finish(i: Int, valid: Boolean, v: Double): Any? = when {
        functions[i] == AggregationFunction.COUNT -> v.toInt()          // CountAccumulator.finish is an Int (Accumulators.kt:26-36)
        !valid -> null                                                  // empty => null (Accumulators.kt:47-53)
        else -> v
    }

    override fun open() {
        if (batch == MemorySegment.NULL) {
            val out = arena.allocate(ADDRESS)
            QeNative.check(ctx, QeNative.qe_batch_create.invokeExact(ctx, source.rowCount, source.columnCount,
                    source.columnDescs(arena), out) as Int)
            batch = out.get(ADDRESS, 0)
        }
        if (groupCount == 0) {
            val values = arena.allocate(JAVA_DOUBLE, maxOf(1, nagg).toLong())
            val valid = arena.allocate(JAVA_BYTE, maxOf(1, nagg).toLong())
            val selected = arena.allocate(JAVA_LONG)
            QeNative.check(ctx, QeNative.qe_filter_aggregate.invokeExact(ctx, batch, filterExpr, exprs, fns, nagg,
                    values, valid, selected) as Int)
            rows = listOf(Array(nagg) { finish(it, valid.getAtIndex(JAVA_BYTE, it.toLong()) != 0.toByte(), values.getAtIndex(JAVA_DOUBLE, it.toLong())) })
        } else {
            val out = arena.allocate(ADDRESS)
            val aggExprs = exprs.asSlice(ADDRESS.byteSize() * groupCount)
            QeNative.check(ctx, QeNative.qe_filter_groupby.invokeExact(ctx, batch, filterExpr, exprs, groupCount, aggExprs, fns, nagg, out) as Int)
            val result = out.get(ADDRESS, 0)
            val n = QeNative.qe_result_count.invokeExact(result) as Long
            val cols = Array(groupCount + nagg) { HostColumn.fetch(ctx, result, it, arena) }
            rows = (0 until n).map { r ->
                Array<Any?>(groupCount + nagg) { c ->
                    val v = cols[c].box(r)
                    if (c < groupCount) v else finish(c - groupCount, v != null, (v as Double?) ?: 0.0)
                }
            }
            QeNative.qe_result_free.invokeExact(ctx, result)
        }
        iter = rows!!.iterator()
    }

    override fun next(): Array<Any?>? {
        val it = iter ?: throw IllegalStateException("Operator not opened")   // GroupByAggregationOperator.kt:57
        return if (it.hasNext()) it.next() else null
    }

    override fun close() {
        iter = null
        rows = null
    }
}

/** compileExpression (Compiler.kt:20-26) for the GPU: serialise, qe_expr_compile (verifies and types the program). */
internal class ExpressionCompiler(private val ctx: MemorySegment, private val arena: Arena) {
    fun compile(e: Expression): MemorySegment {
        val prog = ProgramSerializer.serialize(e)
        val seg = arena.allocate(prog.size.toLong()).also { it.copyFrom(MemorySegment.ofArray(prog)) }
        val out = arena.allocate(ADDRESS)
        QeNative.check(ctx, QeNative.qe_expr_compile.invokeExact(ctx, seg, prog.size.toLong(), out) as Int)
        return out.get(ADDRESS, 0)
    }
}

// The drop-in operator: Projection(Filter(Scan)) fused into one GPU call behind the reference's
// unchanged Operator API (operator/Operators.kt:5-11).  Kotlin twin of
// queryengine_amd/operators.py:GpuFilterProjectOperator.  All work happens in open(), like the
// reference's blocking operators (GlobalAggregationOperator.kt:10-25); next() boxes rows lazily.
package net.jhorstmann.queryengine.gpu

import net.jhorstmann.queryengine.ast.Expression
import net.jhorstmann.queryengine.data.DataType
import net.jhorstmann.queryengine.operator.Operator
import java.lang.foreign.*
import java.lang.foreign.ValueLayout.*

/** A scan leaf that can hand over whole columns (the columnar replacement of MemoryTable). */
interface ColumnarSource {
    val rowCount: Long
    /** off-heap column buffers in the layouts of qe_col_desc */
    fun columnDescs(arena: Arena): MemorySegment
    val columnCount: Int
    /** a source that already holds its columns in HBM (GpuCsvTable: qe_csv_pin) hands the qe_batch* over; default: none */
    fun deviceBatch(ctx: MemorySegment): MemorySegment = MemorySegment.NULL
}

class GpuFilterProjectOperator(
        private val ctx: MemorySegment,                 // qe_ctx*
        private val source: ColumnarSource,
        filter: Expression?,
        projections: List<Expression>) : Operator() {

    private val arena = Arena.ofConfined()
    private val filterExpr: MemorySegment = filter?.let { compile(it) } ?: MemorySegment.NULL
    private val projExprs: MemorySegment = arena.allocate(ADDRESS, projections.size.toLong()).also { arr ->
        projections.forEachIndexed { i, e -> arr.setAtIndex(ADDRESS, i.toLong(), compile(e)) }
    }
    private val nproj = projections.size
    private var batch: MemorySegment = MemorySegment.NULL    // pinned to HBM once, reused by every open()
    private var result: MemorySegment = MemorySegment.NULL
    private var hostResult: MemorySegment = MemorySegment.NULL   // qe_host_result*: the result's columns in pinned host memory
    private var columns: Array<HostColumn>? = null
    private var idx = 0L
    private var count = 0L

    private fun compile(e: Expression): MemorySegment {     // compileExpression, Compiler.kt:20-26
        val prog = ProgramSerializer.serialize(e)
        val seg = arena.allocate(prog.size.toLong()).also { it.copyFrom(MemorySegment.ofArray(prog)) }
        val out = arena.allocate(ADDRESS)
        QeNative.check(ctx, QeNative.qe_expr_compile.invokeExact(ctx, seg, prog.size.toLong(), out) as Int)
        return out.get(ADDRESS, 0)
    }

    override fun open() {
        if (batch == MemorySegment.NULL) batch = source.deviceBatch(ctx)
        if (batch == MemorySegment.NULL) {
            val out = arena.allocate(ADDRESS)
            QeNative.check(ctx, QeNative.qe_batch_create.invokeExact(ctx, source.rowCount, source.columnCount,
                    source.columnDescs(arena), out) as Int)
            batch = out.get(ADDRESS, 0)
        }
        val out = arena.allocate(ADDRESS)
        QeNative.check(ctx, QeNative.qe_filter_project.invokeExact(ctx, batch, filterExpr, projExprs, nproj, out) as Int)
        result = out.get(ADDRESS, 0)
        count = QeNative.qe_result_count.invokeExact(result) as Long
        columns = null
        idx = 0
        // the rows' way to the host starts NOW, on the context's copy stream, into pinned memory owned by the library:
        // it runs beside whatever the caller does before its first next() (and beside the next batch's scan)
        val hout = arena.allocate(ADDRESS)
        QeNative.check(ctx, QeNative.qe_result_to_host.invokeExact(ctx, result, hout) as Int)
        hostResult = hout.get(ADDRESS, 0)
    }

    /** the qe_result* of the last open(): for a downstream GPU consumer (qe_gather, qe_result_order_by) */
    fun result(): MemorySegment = result.also { check(it != MemorySegment.NULL) { "Operator not initialized" } }

    override fun next(): Array<Any?>? {
        check(result != MemorySegment.NULL) { "Operator not initialized" }
        if (idx >= count) return null
        val cols = columns ?: run {
            QeNative.check(ctx, QeNative.qe_host_result_wait.invokeExact(ctx, hostResult) as Int)
            Array(nproj) { HostColumn.view(ctx, hostResult, it, arena) }.also { columns = it }   // pinned memory: no second copy
        }
        val i = idx++
        return Array(nproj) { cols[it].box(i) }                 // fresh row per call (ProjectionOperator.kt:18)
    }

    override fun close() {
        if (hostResult != MemorySegment.NULL) QeNative.qe_host_result_free.invokeExact(ctx, hostResult)   // waits for a copy in flight
        hostResult = MemorySegment.NULL
        if (result != MemorySegment.NULL) QeNative.qe_result_free.invokeExact(ctx, result)
        result = MemorySegment.NULL
        columns = null
    }
}

/** One result column copied to the host; box() produces the reference's boxed values. */
class HostColumn(val type: DataType?, val typeCode: Int, val data: MemorySegment, val validity: MemorySegment, val dict: List<String>?) {
    fun box(i: Long): Any? {
        if (validity != MemorySegment.NULL && (validity.getAtIndex(JAVA_LONG, i shr 6) ushr (i and 63).toInt()) and 1L == 0L) return null
        return when (typeCode) {
            0 -> dict!![data.getAtIndex(JAVA_INT, i)]
            1 -> data.getAtIndex(JAVA_DOUBLE, i)
            2 -> (data.getAtIndex(JAVA_LONG, i shr 6) ushr (i and 63).toInt()) and 1L != 0L
            3 -> data.getAtIndex(JAVA_LONG, i)
            else -> data.getAtIndex(JAVA_INT, i)
        }
    }

    companion object {
        /** Column `col` of a qe_host_result (after qe_host_result_wait): segments over the library's pinned buffers. */
        fun view(ctx: MemorySegment, host: MemorySegment, col: Int, arena: Arena): HostColumn {
            val v = arena.allocate(QeNative.COL_VIEW)
            QeNative.check(ctx, QeNative.qe_host_result_column.invokeExact(host, col, v) as Int)
            val type = v.get(JAVA_INT, 0)
            val n = v.get(JAVA_LONG, 24)
            val words = (n + 63) / 64
            val width = when (type) { 1, 3 -> 8L; 2 -> 0L; else -> 4L }
            val data = v.get(ADDRESS, 8).reinterpret(maxOf(8, if (type == 2) words * 8 else n * width))
            val vptr = v.get(ADDRESS, 16)
            // validity == NULL: no NULL in the column (box() checks for it)
            val valid = if (vptr == MemorySegment.NULL) MemorySegment.NULL else vptr.reinterpret(maxOf(8, words * 8))
            val dictSeg = v.get(ADDRESS, 32)
            val dict = if (type == 0) (0 until (QeNative.qe_dict_size.invokeExact(dictSeg) as Int)).map {
                (QeNative.qe_dict_entry.invokeExact(dictSeg, it) as MemorySegment).reinterpret(65536).getString(0)
            } else null
            return HostColumn(DataType.values().getOrNull(type), type, data, valid, dict)
        }

        /** Column `col` of a device result copied into caller-owned (arena) memory: qe_result_column_to_host. */
        fun fetch(ctx: MemorySegment, result: MemorySegment, col: Int, arena: Arena): HostColumn {
            val view = arena.allocate(QeNative.COL_VIEW)
            QeNative.check(ctx, QeNative.qe_result_column.invokeExact(result, col, view) as Int)
            val type = view.get(JAVA_INT, 0)
            val n = view.get(JAVA_LONG, 24)
            val words = (n + 63) / 64
            val width = when (type) { 1, 3 -> 8L; 2 -> 0L; else -> 4L }
            val data = arena.allocate(maxOf(8, if (type == 2) words * 8 else n * width))
            val valid = arena.allocate(maxOf(8, words * 8))
            QeNative.check(ctx, QeNative.qe_result_column_to_host.invokeExact(ctx, result, col, data, valid) as Int)
            val dictSeg = view.get(ADDRESS, 32)
            val dict = if (type == 0) (0 until (QeNative.qe_dict_size.invokeExact(dictSeg) as Int)).map {
                (QeNative.qe_dict_entry.invokeExact(dictSeg, it) as MemorySegment).reinterpret(65536).getString(0)
            } else null
            return HostColumn(DataType.values().getOrNull(type), type, data, valid, dict)
        }
    }
}

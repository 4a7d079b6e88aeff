// A CSV-backed table whose scan leaf is columnar and pinned to HBM: the drop-in for CsvTable / UnivocityCsvTable
// (data/CsvTable.kt:12-29, data/UnivocityCsvTable.kt:10-26).  The file is converted ONCE by qe_csv_parse_file -- same rules
// as CsvSourceOperator.next (operator/CsvSourceOperator.kt:52-76): header lookup by name, empty or missing field = null,
// String.toBoolean, String.toDouble -- into one array per field and copied to the GPU by qe_csv_pin.  NOT compiled in this
// repository (no JDK in the build image); see INTEGRATION.md.
package net.jhorstmann.queryengine.gpu

import net.jhorstmann.queryengine.data.DataType
import net.jhorstmann.queryengine.data.Schema
import net.jhorstmann.queryengine.data.Table
import net.jhorstmann.queryengine.operator.Operator
import java.io.File
import java.lang.foreign.Arena
import java.lang.foreign.MemorySegment
import java.lang.foreign.ValueLayout.*

class GpuCsvTable(private val ctx: MemorySegment, private val file: File, override val schema: Schema) : Table(), ColumnarSource {
    private var table: MemorySegment = MemorySegment.NULL      // qe_csv_table*
    private var batch: MemorySegment = MemorySegment.NULL      // qe_batch* (every schema field, in schema order)

    override val rowCount: Long get() = QeNative.qe_csv_nrows.invokeExact(parsed()) as Long
    override val columnCount: Int get() = schema.fields.size
    override fun columnDescs(arena: Arena): MemorySegment = throw IllegalStateException("GpuCsvTable hands over a device batch")
    private fun parsed(): MemorySegment { deviceBatch(ctx); return table }

    /** parse + pin on first use; later scans and re-opened operators reuse the resident batch */
    override fun deviceBatch(ctx: MemorySegment): MemorySegment {
        if (batch == MemorySegment.NULL) Arena.ofConfined().use { a ->
            val n = schema.fields.size
            val names = a.allocate(ADDRESS, n.toLong())
            val types = a.allocate(JAVA_INT, n.toLong())
            schema.fields.forEachIndexed { i, f ->
                names.setAtIndex(ADDRESS, i.toLong(), a.allocateFrom(f.name))
                types.setAtIndex(JAVA_INT, i.toLong(), f.type.ordinal)          // STRING, DOUBLE, BOOLEAN = 0, 1, 2 (data/Schema.kt:3-5)
            }
            val out = a.allocate(ADDRESS)
            // a field missing from the header / a malformed number: status 1 with the reference's message
            QeNative.check(ctx, QeNative.qe_csv_parse_file.invokeExact(ctx, a.allocateFrom(file.path), n, names, types, out) as Int)
            table = out.get(ADDRESS, 0)
            QeNative.check(ctx, QeNative.qe_csv_pin.invokeExact(ctx, table, out) as Int)
            batch = out.get(ADDRESS, 0)
        }
        return batch
    }

    private fun columnIndex(name: String): Int = schema.fields.indexOfFirst { it.name == name }
    private fun columnType(index: Int): DataType = schema.fields[index].type

    /** data/Table.kt:8 -- a plain scan (no Filter / Projection above it) is a projection of bare columns */
    override fun getScanOperator(projection: List<String>): Operator =
            GpuFilterProjectOperator(ctx, this, null, projection.map { name ->
                val i = columnIndex(name)
                if (i < 0) throw IllegalStateException("projected field $name not found in schema")   // CsvSourceOperator.kt:25-26
                net.jhorstmann.queryengine.ast.ColumnExpression(name, i, columnType(i))
            })

    fun close() {
        if (batch != MemorySegment.NULL) QeNative.qe_batch_free.invokeExact(ctx, batch)
        if (table != MemorySegment.NULL) QeNative.qe_csv_free.invokeExact(ctx, table)
        batch = MemorySegment.NULL; table = MemorySegment.NULL
    }
}

// Expression tree -> postfix program bytes (include/qe_hip.h "expression program").
// The Kotlin twin of queryengine_amd/program.py; an ExpressionVisitor<Unit> over the reference's own
// AST (ast/Expressions.kt:6-62), using Function.ordinal and DataType.ordinal as the wire values.
package net.jhorstmann.queryengine.gpu

import net.jhorstmann.queryengine.ast.*
import java.io.ByteArrayOutputStream
import java.nio.ByteBuffer
import java.nio.ByteOrder

internal class ProgramSerializer : ExpressionVisitor<Unit> {
    private val out = ByteArrayOutputStream().apply { write(byteArrayOf('Q'.code.toByte(), 'E'.code.toByte(), 'X'.code.toByte(), 1)) }

    private fun le(n: Int, f: ByteBuffer.() -> Unit) = out.write(ByteBuffer.allocate(n).order(ByteOrder.LITTLE_ENDIAN).apply(f).array())

    override fun visitIdentifier(expr: IdentifierExpression) = throw IllegalStateException("Identifier not expected during evaluation")
    override fun visitNumericLiteral(expr: NumericLiteralExpression) = le(9) { put(2); putDouble(expr.value) }
    override fun visitBooleanLiteral(expr: BooleanLiteralExpression) = le(2) { put(3); put(if (expr.value) 1 else 0) }
    override fun visitStringLiteral(expr: StringLiteralExpression) {
        val b = expr.value.toByteArray(Charsets.UTF_8)
        le(3) { put(4); putShort(b.size.toShort()) }
        out.write(b)
    }
    override fun visitColumn(expr: ColumnExpression) = le(4) { put(1); put(expr.dataType.ordinal.toByte()); putShort(expr.index.toShort()) }
    override fun visitFunction(expr: FunctionExpression) {
        expr.operands.forEach { it.accept(this) }                       // operands first: postfix
        le(3) { put(16); put(expr.function.ordinal.toByte()); put((expr.dataTypeNullable?.ordinal ?: 0xFF).toByte()) }
    }
    override fun visitAggregationFunction(expr: AggregationFunctionExpression) =
            throw IllegalStateException("Unexpected aggregation expression in expression compiler")

    fun bytes(): ByteArray = out.toByteArray()

    companion object {
        fun serialize(expr: Expression): ByteArray = ProgramSerializer().also { expr.accept(it) }.bytes()
    }
}

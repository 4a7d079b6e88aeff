"""Expression tree -> postfix program bytes of the C ABI (include/qe_hip.h).

This is the serialiser a Kotlin ``ExpressionVisitor<Unit>`` would implement on
the JVM side: operands first, then the function, so the
library can verify it with a plain operand stack the way the reference's
``MaxStackVisitor`` walks its trees (evaluator/BytecodeCompiler.kt:177-196).
"""
from __future__ import annotations

import struct

from . import ast as A

OP_COLUMN, OP_NUM_LITERAL, OP_BOOL_LITERAL, OP_STR_LITERAL, OP_FUNCTION = 1, 2, 3, 4, 16
HEADER = b"QEX\x01"


class _Serializer(A.ExpressionVisitor):
    def __init__(self):
        self.out = bytearray(HEADER)

    def visitIdentifier(self, expr):
        raise RuntimeError("Identifier not expected during evaluation")   # Interpreter.kt:9-11

    def visitNumericLiteral(self, expr):
        self.out += struct.pack("<Bd", OP_NUM_LITERAL, float(expr.value))

    def visitBooleanLiteral(self, expr):
        self.out += struct.pack("<BB", OP_BOOL_LITERAL, 1 if expr.value else 0)

    def visitStringLiteral(self, expr):
        b = expr.value.encode("utf-8")
        if len(b) > 0xFFFF:
            raise ValueError("string literal too long")
        self.out += struct.pack("<BH", OP_STR_LITERAL, len(b)) + b

    def visitColumn(self, expr):
        self.out += struct.pack("<BBH", OP_COLUMN, int(expr.dataType), expr.index)

    def visitFunction(self, expr):
        for op in expr.operands:
            op.accept(self)
        t = 0xFF if expr.dataTypeNullable is None else int(expr.dataTypeNullable)
        self.out += struct.pack("<BBB", OP_FUNCTION, expr.function.ordinal, t)

    def visitAggregationFunction(self, expr):
        raise RuntimeError("Unexpected aggregation expression in expression compiler")   # Interpreter.kt:111-113


def serialize(expr: A.Expression) -> bytes:
    s = _Serializer()
    expr.accept(s)
    return bytes(s.out)

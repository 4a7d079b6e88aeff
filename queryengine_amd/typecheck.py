"""Type assignment for function nodes -- the host-side twin of the checks libqe_hip repeats
when it verifies a program.

Follows ``evaluator/TypeCheck.kt:38-133`` with three deliberate differences:
  * AND/OR require BOTH operands BOOLEAN (TypeCheck.kt:79-85 demands operands[0] == DOUBLE: a bug
    that makes every ``p AND q`` fail);
  * unary operators check one operand (TypeCheck.kt:50-52 indexes operands[1]);
  * numeric = {DOUBLE, INT64, INT32} with Java binary numeric promotion (extension types).
"""
from __future__ import annotations

from .ast import (AggregationFunction, AggregationFunctionExpression, DefaultExpressionVisitor, Expression, Function,
                  FunctionExpression)
from .datatypes import DataType, promote


class TypeCheckException(RuntimeError):
    """evaluator/TypeCheck.kt:8"""


def _invalid(function, operands) -> TypeCheckException:
    return TypeCheckException(f"Invalid operand types for [{function.name}] [{', '.join(o.dataType.name for o in operands)}]")


class _TypeCheckVisitor(DefaultExpressionVisitor):
    def visitFunction(self, expr: FunctionExpression) -> Expression:
        ops = self.visitOperands(expr.operands)
        f = expr.function
        if len(ops) != f.arity:
            raise TypeCheckException(f"[{f.name}] expects {f.arity} operands, got {len(ops)}")
        B = DataType.BOOLEAN
        if f in (Function.UNARY_MINUS, Function.UNARY_PLUS):
            if not ops[0].dataType.is_numeric:
                raise _invalid(f, ops)
            return expr.with_(ops, ops[0].dataType)
        if f in (Function.ADD, Function.SUB, Function.MUL, Function.DIV, Function.MOD):
            t = promote(ops[0].dataType, ops[1].dataType)
            if t is None:
                raise _invalid(f, ops)
            return expr.with_(ops, t)
        if f == Function.NOT:
            if ops[0].dataType != B:
                raise _invalid(f, ops)
            return expr.with_(ops, B)
        if f in (Function.CMP_EQ, Function.CMP_NE):
            if promote(ops[0].dataType, ops[1].dataType) is None and ops[0].dataType != ops[1].dataType:
                raise _invalid(f, ops)
            return expr.with_(ops, B)
        if f in (Function.CMP_LT, Function.CMP_LE, Function.CMP_GE, Function.CMP_GT):
            if promote(ops[0].dataType, ops[1].dataType) is None:   # TypeCheck.kt:70-76: numeric only
                raise _invalid(f, ops)
            return expr.with_(ops, B)
        if f in (Function.AND, Function.OR):
            if ops[0].dataType != B or ops[1].dataType != B:
                raise _invalid(f, ops)
            return expr.with_(ops, B)
        if f == Function.IF:
            if ops[0].dataType != B:
                raise _invalid(f, ops)
            t = promote(ops[1].dataType, ops[2].dataType)
            if t is None:
                if ops[1].dataType != ops[2].dataType:
                    raise _invalid(f, ops)
                t = ops[1].dataType
            return expr.with_(ops, t)
        raise TypeCheckException(f"unknown function {f}")

    def visitAggregationFunction(self, expr: AggregationFunctionExpression) -> Expression:
        ops = self.visitOperands(expr.operands)
        f = expr.function
        if f in (AggregationFunction.MIN, AggregationFunction.MAX, AggregationFunction.SUM, AggregationFunction.AVG):
            if not ops[0].dataType.is_numeric:
                raise TypeCheckException(f"Invalid operand types for aggregation [{f.name}] [{ops[0].dataType.name}]")
        elif f != AggregationFunction.COUNT:
            raise TypeCheckException(f"aggregation [{f.name}] is not implemented (TODO() in the reference, Accumulators.kt:16-17)")
        return AggregationFunctionExpression(f, ops, DataType.DOUBLE, expr.accumulatorIndex)   # TypeCheck.kt:108-120


def typeCheck(expr: Expression) -> Expression:
    return expr.accept(_TypeCheckVisitor())

"""Expression tree: the input language of the hot path.

Mirrors ``ast/Expressions.kt:6-62`` (node classes), ``ast/Functions.kt:3-26``
(``FunctionType``, ``Function`` with type and arity; ordinals identical to the
Kotlin enum so that the serialised program is what a Kotlin-side serialiser
would emit from ``Function.ordinal``) and ``ast/ExpressionVisitor.kt:3-13``.
"""
from __future__ import annotations

import enum
from dataclasses import dataclass
from typing import Generic, List, Optional, Sequence, TypeVar

from .datatypes import DataType

R = TypeVar("R")


class FunctionType(enum.IntEnum):
    LOGIC = 0
    ARITHMETIC = 1
    COMPARISON = 2


class Function(enum.Enum):
    # (ordinal, type, arity) -- ast/Functions.kt:7-22
    AND = (0, FunctionType.LOGIC, 2)
    OR = (1, FunctionType.LOGIC, 2)
    IF = (2, FunctionType.LOGIC, 3)
    NOT = (3, FunctionType.LOGIC, 1)
    UNARY_MINUS = (4, FunctionType.ARITHMETIC, 1)
    UNARY_PLUS = (5, FunctionType.ARITHMETIC, 1)
    MUL = (6, FunctionType.ARITHMETIC, 2)
    DIV = (7, FunctionType.ARITHMETIC, 2)
    MOD = (8, FunctionType.ARITHMETIC, 2)
    ADD = (9, FunctionType.ARITHMETIC, 2)
    SUB = (10, FunctionType.ARITHMETIC, 2)
    CMP_LT = (11, FunctionType.COMPARISON, 2)
    CMP_LE = (12, FunctionType.COMPARISON, 2)
    CMP_GE = (13, FunctionType.COMPARISON, 2)
    CMP_GT = (14, FunctionType.COMPARISON, 2)
    CMP_EQ = (15, FunctionType.COMPARISON, 2)
    CMP_NE = (16, FunctionType.COMPARISON, 2)

    @property
    def ordinal(self) -> int:
        return self.value[0]

    @property
    def type(self) -> FunctionType:
        return self.value[1]

    @property
    def arity(self) -> int:
        return self.value[2]


class AggregationFunction(enum.IntEnum):
    # ast/Functions.kt:24-26
    MIN = 0
    MAX = 1
    SUM = 2
    COUNT = 3
    AVG = 4
    ANY = 5
    ALL = 6


class ExpressionVisitor(Generic[R]):
    def visitIdentifier(self, expr: "IdentifierExpression") -> R: raise NotImplementedError
    def visitNumericLiteral(self, expr: "NumericLiteralExpression") -> R: raise NotImplementedError
    def visitBooleanLiteral(self, expr: "BooleanLiteralExpression") -> R: raise NotImplementedError
    def visitStringLiteral(self, expr: "StringLiteralExpression") -> R: raise NotImplementedError
    def visitFunction(self, expr: "FunctionExpression") -> R: raise NotImplementedError
    def visitAggregationFunction(self, expr: "AggregationFunctionExpression") -> R: raise NotImplementedError
    def visitColumn(self, expr: "ColumnExpression") -> R: raise NotImplementedError


class Expression:
    @property
    def dataType(self) -> DataType:
        raise NotImplementedError

    def accept(self, visitor: ExpressionVisitor[R]) -> R:
        raise NotImplementedError


@dataclass(frozen=True)
class IdentifierExpression(Expression):
    name: str

    @property
    def dataType(self) -> DataType:
        raise RuntimeError("Unresolved identifier")

    def accept(self, visitor):
        return visitor.visitIdentifier(self)


@dataclass(frozen=True)
class NumericLiteralExpression(Expression):
    """Every numeric literal is a Double (ast/Expressions.kt:17-21)."""
    value: float

    @property
    def dataType(self) -> DataType:
        return DataType.DOUBLE

    def accept(self, visitor):
        return visitor.visitNumericLiteral(self)


@dataclass(frozen=True)
class BooleanLiteralExpression(Expression):
    value: bool

    @property
    def dataType(self) -> DataType:
        return DataType.BOOLEAN

    def accept(self, visitor):
        return visitor.visitBooleanLiteral(self)


@dataclass(frozen=True)
class StringLiteralExpression(Expression):
    value: str

    @property
    def dataType(self) -> DataType:
        return DataType.STRING

    def accept(self, visitor):
        return visitor.visitStringLiteral(self)


@dataclass(frozen=True)
class FunctionExpression(Expression):
    function: Function
    operands: Sequence[Expression]
    dataTypeNullable: Optional[DataType] = None

    def __post_init__(self):
        object.__setattr__(self, "operands", tuple(self.operands))

    @property
    def dataType(self) -> DataType:
        if self.dataTypeNullable is None:
            raise RuntimeError("Data type not initialized")
        return self.dataTypeNullable

    def with_(self, operands: Sequence[Expression], dataType: DataType) -> "FunctionExpression":
        return FunctionExpression(self.function, tuple(operands), dataType)

    def accept(self, visitor):
        return visitor.visitFunction(self)


@dataclass(frozen=True)
class AggregationFunctionExpression(Expression):
    function: AggregationFunction
    operands: Sequence[Expression]
    dataTypeNullable: Optional[DataType] = None
    accumulatorIndex: int = -1

    def __post_init__(self):
        object.__setattr__(self, "operands", tuple(self.operands))

    @property
    def dataType(self) -> DataType:
        if self.dataTypeNullable is None:
            raise RuntimeError("Data type not initialized")
        return self.dataTypeNullable

    def accept(self, visitor):
        return visitor.visitAggregationFunction(self)


@dataclass(frozen=True)
class ColumnExpression(Expression):
    name: str
    index: int
    _dataType: DataType

    @property
    def dataType(self) -> DataType:
        return self._dataType

    def accept(self, visitor):
        return visitor.visitColumn(self)


class DefaultExpressionVisitor(ExpressionVisitor[Expression]):
    """Identity rewrite (ast/DefaultExpressionVisitor.kt:3-25)."""

    def visitOperands(self, operands):
        return [op.accept(self) for op in operands]

    def visitIdentifier(self, expr): return expr
    def visitNumericLiteral(self, expr): return expr
    def visitBooleanLiteral(self, expr): return expr
    def visitStringLiteral(self, expr): return expr
    def visitColumn(self, expr): return expr

    def visitFunction(self, expr):
        return FunctionExpression(expr.function, self.visitOperands(expr.operands), expr.dataTypeNullable)

    def visitAggregationFunction(self, expr):
        return AggregationFunctionExpression(expr.function, self.visitOperands(expr.operands),
                                             expr.dataTypeNullable, expr.accumulatorIndex)

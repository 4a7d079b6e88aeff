"""queryengine_amd -- MI355X-native drop-in for the Filter/Projection hot path
of jhorstmann/queryengine."""
from .datatypes import DataType, Field, Schema, promote  # noqa: F401
from .ast import (  # noqa: F401
    Function, FunctionType, AggregationFunction, Expression, IdentifierExpression, NumericLiteralExpression,
    BooleanLiteralExpression, StringLiteralExpression, FunctionExpression, AggregationFunctionExpression,
    ColumnExpression, ExpressionVisitor, DefaultExpressionVisitor,
)
from .table import Column, ColumnarTable, Table, TableRegistry, pack_bitmap, unpack_bitmap  # noqa: F401

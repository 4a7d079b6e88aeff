"""Shared helpers of the parity tests: expression builders, random data, column comparison."""
from __future__ import annotations

import math
import random
from typing import List, Optional, Sequence

import numpy as np

from queryengine_amd import (BooleanLiteralExpression, Column, ColumnExpression, DataType, Function,
                             FunctionExpression, NumericLiteralExpression, StringLiteralExpression, promote)

D, I64, I32, B, S = DataType.DOUBLE, DataType.INT64, DataType.INT32, DataType.BOOLEAN, DataType.STRING
Fn = Function


def col(name: str, index: int, t: DataType) -> ColumnExpression:
    return ColumnExpression(name, index, t)


def num(v: float) -> NumericLiteralExpression:
    return NumericLiteralExpression(float(v))


def fn(f: Function, *ops, t: Optional[DataType] = None) -> FunctionExpression:
    """Pre-typed FunctionExpression, as CompilerTest.kt:28,40,53 builds them by hand."""
    if t is None:
        t = infer_type(f, ops)
    return FunctionExpression(f, list(ops), t)


def infer_type(f: Function, ops) -> DataType:
    if f in (Fn.AND, Fn.OR, Fn.NOT) or f.name.startswith("CMP_"):
        return B
    if f == Fn.IF:
        p = promote(ops[1].dataType, ops[2].dataType)
        return p if p is not None else ops[1].dataType
    if f in (Fn.UNARY_MINUS, Fn.UNARY_PLUS):
        return ops[0].dataType
    return promote(ops[0].dataType, ops[1].dataType)


def assert_columns_equal(got: Column, want: Column, what: str = "") -> None:
    """Bit-exact comparison: same type, length, null positions; values identical where valid
    (any NaN == any NaN: Java only defines NaN through doubleToLongBits' canonical form)."""
    assert got.type == want.type, f"{what}: type {got.type} != {want.type}"
    assert len(got) == len(want), f"{what}: length {len(got)} != {len(want)}"
    n = len(got)
    gv = got.valid if got.valid is not None else np.ones(n, dtype=bool)
    wv = want.valid if want.valid is not None else np.ones(n, dtype=bool)
    bad = np.nonzero(gv != wv)[0]
    assert bad.size == 0, f"{what}: validity differs at rows {bad[:8]} (got {gv[bad[:8]]})"
    if got.type == S:
        g = [got.dictionary[c] if v else None for c, v in zip(got.data, gv)]
        w = [want.dictionary[c] if v else None for c, v in zip(want.data, wv)]
        assert g == w, f"{what}: strings differ"
        return
    g, w = got.data[gv], want.data[wv]
    if got.type == D:
        gb, wb = g.view(np.uint64), w.view(np.uint64)
        same = (gb == wb) | (np.isnan(g) & np.isnan(w))
    else:
        same = g == w
    bad = np.nonzero(~same)[0]
    assert bad.size == 0, f"{what}: values differ at valid-row ordinals {bad[:8]}: got {g[bad[:8]]!r} want {w[bad[:8]]!r}"


SPECIAL_F64 = [0.0, -0.0, 1.0, -1.0, 0.5, 2.5, 100.0, -7.5, 7.5, float("inf"), float("-inf"), float("nan"),
               5e-324, -5e-324, 1.7976931348623157e308, 2.0 ** 53, -(2.0 ** 53), 1e-300, 3.0, 99.99999999999999]
SPECIAL_I64 = [0, 1, -1, 2, -2, 100, 99, 101, 2 ** 31, -(2 ** 31), 2 ** 53, -(2 ** 53), 2 ** 63 - 1, -(2 ** 63), 7, -7,
               2 ** 53 + 1, -(2 ** 53 + 1), 2 ** 53 + 2, -(2 ** 53 + 2), 2 ** 53 - 1, -(2 ** 53 - 1)]
# literals around the exact-integer limit of a double: (double)int64 OP literal must behave like the widened compare
# (BytecodeCompiler.kt:298-320), also where 2^53 and 2^53+1 collapse onto the same double
BOUNDARY_LITERALS = [2.0 ** 53, -(2.0 ** 53), 2.0 ** 53 - 1, -(2.0 ** 53 - 1), 2.0 ** 53 + 2, 2.0 ** 63, -(2.0 ** 63)]
SPECIAL_I32 = [0, 1, -1, 2, -2, 100, 99, 101, 2 ** 31 - 1, -(2 ** 31), 7, -7, 8766, 9131]


def random_column(rng: np.random.Generator, t: DataType, n: int, null_frac: float = 0.0, special: bool = True,
                  dictionary: Optional[List[str]] = None) -> Column:
    if t == D:
        data = rng.normal(0, 100, n)
        if special and n:
            idx = rng.integers(0, n, max(1, n // 4))
            data[idx] = rng.choice(np.array(SPECIAL_F64), idx.size)
    elif t == I64:
        data = rng.integers(-1000, 1000, n, dtype=np.int64)
        if special and n:
            idx = rng.integers(0, n, max(1, n // 6))
            data[idx] = rng.choice(np.array(SPECIAL_I64, dtype=np.int64), idx.size)
    elif t == I32:
        data = rng.integers(-1000, 1000, n, dtype=np.int32)
        if special and n:
            idx = rng.integers(0, n, max(1, n // 6))
            data[idx] = rng.choice(np.array(SPECIAL_I32, dtype=np.int32), idx.size)
    elif t == B:
        data = rng.integers(0, 2, n).astype(bool)
    elif t == S:
        dictionary = dictionary or ["k%04d" % i for i in range(16)]
        data = rng.integers(0, len(dictionary), n, dtype=np.int32)
    else:
        raise ValueError(t)
    valid = None
    if null_frac > 0 and n:
        valid = rng.random(n) >= null_frac
    return Column(t, data, valid, dictionary)


class ExprGen:
    """Random well-typed expression trees over a given schema (list of (name, type))."""

    def __init__(self, rng: random.Random, schema: Sequence, allow_string: bool = True):
        self.rng = rng
        self.schema = list(schema)
        self.allow_string = allow_string

    def cols_of(self, pred) -> List[ColumnExpression]:
        return [col(n, i, t) for i, (n, t) in enumerate(self.schema) if pred(t)]

    def numeric(self, depth: int) -> FunctionExpression:
        r = self.rng
        leaves = self.cols_of(lambda t: t.is_numeric)
        if depth <= 0 or r.random() < 0.25:
            if leaves and r.random() < 0.75:
                return r.choice(leaves)
            return num(r.choice([0.0, 1.0, 2.0, 10.0, 100.0, 0.5, -3.0, 1e9, 7.0, 99.5] + BOUNDARY_LITERALS))
        k = r.random()
        if k < 0.1:
            return fn(Fn.UNARY_MINUS, self.numeric(depth - 1))
        if k < 0.15:
            return fn(Fn.UNARY_PLUS, self.numeric(depth - 1))
        if k < 0.25:
            a, b = self.numeric(depth - 1), self.numeric(depth - 1)
            return fn(Fn.IF, self.boolean(depth - 1), a, b)
        f = r.choice([Fn.ADD, Fn.SUB, Fn.MUL, Fn.DIV, Fn.MOD])
        return fn(f, self.numeric(depth - 1), self.numeric(depth - 1))

    def boolean(self, depth: int):
        r = self.rng
        bcols = self.cols_of(lambda t: t == B)
        if depth <= 0 or r.random() < 0.15:
            if bcols and r.random() < 0.6:
                return r.choice(bcols)
            if r.random() < 0.2:
                return BooleanLiteralExpression(r.random() < 0.5)
            return fn(r.choice([Fn.CMP_LT, Fn.CMP_LE, Fn.CMP_GE, Fn.CMP_GT, Fn.CMP_EQ, Fn.CMP_NE]),
                      self.numeric(0), self.numeric(0))
        k = r.random()
        if k < 0.35:
            return fn(r.choice([Fn.CMP_LT, Fn.CMP_LE, Fn.CMP_GE, Fn.CMP_GT, Fn.CMP_EQ, Fn.CMP_NE]),
                      self.numeric(depth - 1), self.numeric(depth - 1))
        if k < 0.7:
            return fn(r.choice([Fn.AND, Fn.OR]), self.boolean(depth - 1), self.boolean(depth - 1))
        if k < 0.8:
            return fn(Fn.NOT, self.boolean(depth - 1))
        if k < 0.87:
            return fn(Fn.IF, self.boolean(depth - 1), self.boolean(depth - 1), self.boolean(depth - 1))
        scols = self.cols_of(lambda t: t == S)
        if scols and self.allow_string:
            c = r.choice(scols)
            d = self._dict_of(c)
            lit = r.choice(d + ["absent"])
            return fn(r.choice([Fn.CMP_EQ, Fn.CMP_NE]), c, StringLiteralExpression(lit))
        if bcols:
            return fn(r.choice([Fn.CMP_EQ, Fn.CMP_NE, Fn.CMP_LT, Fn.CMP_GE]), r.choice(bcols), self.boolean(depth - 1))
        return fn(Fn.CMP_LT, self.numeric(depth - 1), self.numeric(depth - 1))

    dicts = {}

    def _dict_of(self, c):
        return self.dicts.get(c.name, ["k0000"])


# ---- a sharded aggregation case: every value is a function of the GLOBAL row index -----------------------------------
_AGG_DICT = [f"k{i:02d}" for i in range(12)]


def agg_case_columns(begin: int, end: int) -> List[Column]:
    """Columns s (STRING, nullable, keys k06.. appear only in the second half of a 30 000-row table), p (BOOLEAN),
    x (INT64), y (DOUBLE with nulls, integer valued: sums are exact in any order)."""
    i = np.arange(begin, end, dtype=np.int64)
    h = (i * 2654435761) % 1000003
    code = np.where(i < 15000, h % 6, h % 12).astype(np.int32)
    s_valid = (h % 17) != 0
    p = (h % 3) == 0
    x = (h % 1000).astype(np.int64)
    y = ((h % 2001) - 1000).astype(np.float64)
    y_valid = (h % 11) != 0
    return [Column(S, code, s_valid, list(_AGG_DICT)), Column(B, p, None), Column(I64, x, None), Column(D, y, y_valid)]


def AGG_CASE():
    """(filter, group keys, aggregate inputs, aggregate functions) over agg_case_columns."""
    from queryengine_amd import AggregationFunction as AF
    s, p, x, y = col("s", 0, S), col("p", 1, B), col("x", 2, I64), col("y", 3, D)
    flt = fn(Fn.CMP_LT, x, num(900))
    keys = [s, p]
    exprs = [y, y, fn(Fn.ADD, x, y), x, y]
    aggs = [int(AF.MIN), int(AF.MAX), int(AF.SUM), int(AF.COUNT), int(AF.AVG)]
    return flt, keys, exprs, aggs

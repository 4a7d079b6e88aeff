"""The BASELINE.json configurations as (synthetic column specs, filter, projections).

Data are counter-based: value(col, i) depends only on the
GLOBAL row index, so every GPU shard, the host and the CPU oracle generate
identical columns without any transfer.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

from . import native as N
from .ast import (ColumnExpression, Expression, Function, FunctionExpression, NumericLiteralExpression,
                  StringLiteralExpression)
from .datatypes import DataType

D, I64, I32, B, S = DataType.DOUBLE, DataType.INT64, DataType.INT32, DataType.BOOLEAN, DataType.STRING
Fn = Function


def _fn(f, *ops, t):
    return FunctionExpression(f, list(ops), t)


def _num(v):
    return NumericLiteralExpression(float(v))


@dataclass
class GenColumn:
    name: str
    type: DataType
    kind: int
    col_id: int
    modulus: int = 0
    offset: int = 0
    step: float = 0.0
    aux_col_id: int = 0
    null_pct: int = 0
    dictionary: Optional[List[str]] = None

    def spec(self, ctx=None) -> N.GenSpec:
        s = N.GenSpec()
        s.kind, s.col_id, s.modulus, s.offset = self.kind, self.col_id, self.modulus, self.offset
        s.step, s.aux_col_id, s.null_pct = self.step, self.aux_col_id, self.null_pct
        if self.dictionary is not None and ctx is not None:
            s.dict = ctx.dictionary(self.dictionary).handle
        return s

    @property
    def width(self) -> int:
        return 8 if self.type in (D, I64) else 4


@dataclass
class Workload:
    name: str
    sql: str
    columns: List[GenColumn]
    filter: Optional[Expression]
    projections: List[Expression]
    default_rows: int
    expected_selectivity: float

    def read_bytes_per_row(self) -> float:
        """algorithmic input bytes per row: column widths (+1/8 byte per validity bitmap)"""
        return sum(c.width + (0.125 if c.null_pct else 0.0) for c in self.columns)

    def write_bytes_per_output_row(self) -> int:
        return sum(8 if p.dataType in (D, I64) else 4 for p in self.projections)

    def algorithmic_bytes(self, nrows: int, nout: int) -> float:
        return nrows * self.read_bytes_per_row() + nout * self.write_bytes_per_output_row()


def config1(nrows: int = 1_000_000) -> Workload:
    """SELECT a + b FROM t WHERE a < 100 -- reference types (DOUBLE), 10 % selectivity"""
    a, b = ColumnExpression("a", 0, D), ColumnExpression("b", 1, D)
    return Workload("config1", "SELECT a + b FROM t WHERE a < 100",
                    [GenColumn("a", D, N.GEN_F64_MOD, 0, modulus=1000), GenColumn("b", D, N.GEN_F64_MOD, 1, modulus=2 ** 31)],
                    _fn(Fn.CMP_LT, a, _num(100), t=B), [_fn(Fn.ADD, a, b, t=D)], nrows, 0.10)


def config2(nrows: int = 1_000_000_000, a_limit: float = 100.0, c_limit: float = 0.5, null_pct: int = 0) -> Workload:
    """SELECT a + b, c * 2.0 FROM t WHERE a < 100 AND c < 0.5 -- int64/int64/f64, 5 % selectivity.
    The headline workload of BASELINE.json (`metric` is quoted on it)."""
    a, b, c = ColumnExpression("a", 0, I64), ColumnExpression("b", 1, I64), ColumnExpression("c", 2, D)
    flt = _fn(Fn.AND, _fn(Fn.CMP_LT, a, _num(a_limit), t=B), _fn(Fn.CMP_LT, c, _num(c_limit), t=B), t=B)
    return Workload("config2", f"SELECT a + b, c * 2.0 FROM t WHERE a < {a_limit:g} AND c < {c_limit:g}",
                    [GenColumn("a", I64, N.GEN_I64_MOD, 0, modulus=1000, null_pct=null_pct),
                     GenColumn("b", I64, N.GEN_I64_MOD, 1, modulus=2 ** 31, null_pct=null_pct),
                     GenColumn("c", D, N.GEN_F64_UNIT, 2, null_pct=null_pct)],
                    flt, [_fn(Fn.ADD, a, b, t=I64), _fn(Fn.MUL, c, _num(2.0), t=D)], nrows,
                    (a_limit / 1000.0) * c_limit)


def config3(nrows: int = 600_037_902) -> Workload:
    """TPC-H Q6-style range predicate over SF100-shaped lineitem columns (dates as int32 days)."""
    sh, di = ColumnExpression("l_shipdate", 0, I32), ColumnExpression("l_discount", 1, D)
    qt, pr = ColumnExpression("l_quantity", 2, D), ColumnExpression("l_extendedprice", 3, D)
    flt = _fn(Fn.AND, _fn(Fn.AND, _fn(Fn.AND, _fn(Fn.AND,
              _fn(Fn.CMP_GE, sh, _num(8766), t=B), _fn(Fn.CMP_LT, sh, _num(9131), t=B), t=B),
              _fn(Fn.CMP_GE, di, _num(0.05), t=B), t=B), _fn(Fn.CMP_LE, di, _num(0.07), t=B), t=B),
              _fn(Fn.CMP_LT, qt, _num(24), t=B), t=B)
    return Workload("config3",
                    "SELECT l_extendedprice * l_discount FROM lineitem WHERE l_shipdate >= 8766 AND l_shipdate < 9131 "
                    "AND l_discount >= 0.05 AND l_discount <= 0.07 AND l_quantity < 24",
                    [GenColumn("l_shipdate", I32, N.GEN_I32_MOD, 0, modulus=2526, offset=8036),
                     GenColumn("l_discount", D, N.GEN_F64_STEP, 1, modulus=11, step=0.01),
                     GenColumn("l_quantity", D, N.GEN_F64_MOD, 2, modulus=50, offset=1),
                     GenColumn("l_extendedprice", D, N.GEN_F64_PRICE, 3, aux_col_id=2)],
                    flt, [_fn(Fn.MUL, pr, di, t=D)], nrows, (365 / 2526) * (3 / 11) * (23 / 50))


def config4(nrows: int = 1_000_000_000, nkeys: int = 1000, key: str = "k0042") -> Workload:
    """SELECT s, v FROM t WHERE s = 'k0042' -- dictionary-encoded string equality + gather-project"""
    dictionary = ["k%04d" % i for i in range(nkeys)]
    s, v = ColumnExpression("s", 0, S), ColumnExpression("v", 1, D)
    return Workload("config4", f"SELECT s, v FROM t WHERE s = '{key}'",
                    [GenColumn("s", S, N.GEN_DICT_MOD, 0, modulus=nkeys, dictionary=dictionary),
                     GenColumn("v", D, N.GEN_F64_UNIT, 1)],
                    _fn(Fn.CMP_EQ, s, StringLiteralExpression(key), t=B), [s, v], nrows, 1.0 / nkeys)


def config2_swapped(nrows: int = 1_000_000_000, null_pct: int = 0) -> Workload:
    """config 2 with its conjuncts written the other way round: SELECT a + b, c * 2.0 FROM t WHERE c < 0.5 AND a < 100.
    Evaluated as written it loads c in full and a for half of the rows; the plan's measured conjunct order makes it config 2."""
    wl = config2(nrows, null_pct=null_pct)
    a, c = wl.filter.operands[0], wl.filter.operands[1]
    wl.filter = _fn(Fn.AND, c, a, t=B)
    wl.name = "config2_swapped"
    wl.sql = "SELECT a + b, c * 2.0 FROM t WHERE c < 0.5 AND a < 100"
    return wl


WORKLOADS = {"config1": config1, "config2": config2, "config3": config3, "config4": config4, "config2_swapped": config2_swapped}

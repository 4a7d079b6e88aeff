"""Columnar tables: the scan leaf of the hot path.

The reference's scan leaf is ``MemoryTable`` (``data/MemoryTable.kt:7-19``), a
row-major ``List<List<Any?>>`` walked by ``MemorySourceOperator``
(``operator/MemorySourceOperator.kt:5-36``).  Its columnar replacement keeps
the same catalogue contract -- ``Table.getScanOperator(projection)``
(``data/Table.kt:6-10``) -- but stores one contiguous numpy array per column
(+ optional validity mask, + dictionary for STRING), which is what is pinned to
HBM once by ``qe_batch_create``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from .datatypes import DataType, Field, Schema

_NP_DTYPE = {
    DataType.DOUBLE: np.float64,
    DataType.INT64: np.int64,
    DataType.INT32: np.int32,
    DataType.BOOLEAN: np.bool_,
    DataType.STRING: np.int32,   # dictionary codes
}


@dataclass
class Column:
    type: DataType
    data: np.ndarray
    valid: Optional[np.ndarray] = None          # bool per row, None = all valid
    dictionary: Optional[List[str]] = None      # STRING only: code -> string

    def __post_init__(self):
        self.data = np.ascontiguousarray(self.data, dtype=_NP_DTYPE[self.type])
        if self.valid is not None:
            self.valid = np.ascontiguousarray(self.valid, dtype=np.bool_)
            if self.valid.shape != self.data.shape:
                raise ValueError("validity mask shape mismatch")
            if bool(self.valid.all()):
                self.valid = None
        if self.type == DataType.STRING and self.dictionary is None:
            raise ValueError("STRING column needs a dictionary")

    def __len__(self) -> int:
        return int(self.data.shape[0])

    @staticmethod
    def from_values(type: DataType, values: Sequence[Any], dictionary: Optional[List[str]] = None) -> "Column":
        """Build from boxed values (``None`` = null), like one column of a MemoryTable."""
        n = len(values)
        valid = np.array([v is not None for v in values], dtype=np.bool_)
        if type == DataType.STRING:
            if dictionary is None:
                dictionary = []
                for v in values:
                    if v is not None and v not in dictionary:
                        dictionary.append(v)
            index = {s: i for i, s in enumerate(dictionary)}
            data = np.array([index[v] if v is not None else 0 for v in values], dtype=np.int32)
        else:
            zero = False if type == DataType.BOOLEAN else 0
            data = np.array([v if v is not None else zero for v in values], dtype=_NP_DTYPE[type])
        if n == 0:
            data = np.zeros(0, dtype=_NP_DTYPE[type])
        return Column(type, data, valid, dictionary)

    def value(self, i: int) -> Any:
        """Boxed value of row i (None = null)."""
        if self.valid is not None and not self.valid[i]:
            return None
        v = self.data[i]
        if self.type == DataType.STRING:
            return self.dictionary[int(v)]
        if self.type == DataType.DOUBLE:
            return float(v)
        if self.type == DataType.BOOLEAN:
            return bool(v)
        return int(v)

    def to_list(self) -> List[Any]:
        return [self.value(i) for i in range(len(self))]


def pack_bitmap(mask: np.ndarray) -> np.ndarray:
    """bool per row -> uint64 words, row i at word i>>6 bit i&63 (LSB first)."""
    n = int(mask.shape[0])
    nwords = (n + 63) // 64
    padded = np.zeros(nwords * 64, dtype=np.uint8)
    padded[:n] = mask.astype(np.uint8)
    return np.packbits(padded, bitorder="little").view(np.uint64).copy()


def unpack_bitmap(words: np.ndarray, n: int) -> np.ndarray:
    bits = np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little")
    return bits[:n].astype(np.bool_)


class Table:
    """data/Table.kt:6-10"""
    schema: Schema

    def getScanOperator(self, projection: List[str]):
        raise NotImplementedError


class ColumnarTable(Table):
    def __init__(self, schema: Schema, columns: Sequence[Column]):
        if len(schema.fields) != len(columns):
            raise ValueError("schema/column count mismatch")
        n = len(columns[0]) if columns else 0
        for f, c in zip(schema.fields, columns):
            if f.type != c.type:
                raise ValueError(f"column {f.name}: type {c.type} != schema {f.type}")
            if len(c) != n:
                raise ValueError("ragged columns")
        self.schema = schema
        self.columns = list(columns)
        self.nrows = n

    @staticmethod
    def from_rows(schema: Schema, rows: Sequence[Sequence[Any]]) -> "ColumnarTable":
        """Same constructor shape as ``MemoryTable(schema, values)``: row-major boxed values."""
        cols = [Column.from_values(f.type, [r[j] for r in rows]) for j, f in enumerate(schema.fields)]
        return ColumnarTable(schema, cols)

    def column(self, name: str) -> Column:
        idx = self.schema.index_of(name)
        if idx < 0:
            raise ValueError(f"Unknown field {name}")   # MemoryTable.kt:11
        return self.columns[idx]

    def getScanOperator(self, projection: List[str]):
        from .operators import ColumnarScanOperator
        return ColumnarScanOperator(self, projection)


class TableRegistry:
    """data/TableRegistry.kt:5-19"""

    def __init__(self):
        self._tables: Dict[str, Table] = {}

    def register(self, name: str, table: Table) -> None:
        self._tables[name] = table

    def getTable(self, name: str) -> Table:
        t = self._tables.get(name)
        if t is None:
            raise ValueError(f"Unknown table {name}")
        return t

    def getSchema(self, name: str) -> Schema:
        return self.getTable(name).schema

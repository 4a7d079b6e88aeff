"""Data types and schema of the hot path.

Mirrors ``data/Schema.kt:3-13`` of the reference (``DataType {STRING, DOUBLE,
BOOLEAN}``, ``Field``, ``Schema``).  The ordinals of the three reference types
are kept (they are what the C-ABI program encoding carries); ``INT64`` and
``INT32`` are build-defined extensions: Java
``long``/``int`` semantics, pinned against the DOUBLE-only reference by keeping
test integers inside +-2^53.

Physical layout on the device:
  STRING  -> int32 dictionary codes + a ``Dictionary``
  DOUBLE  -> f64
  BOOLEAN -> value bitmap, 1 bit/row, LSB-first in uint64 words
  INT64   -> i64,  INT32 -> i32
Every column may carry a validity bitmap (absent = all valid).
"""
from __future__ import annotations

import enum
from dataclasses import dataclass
from typing import List, Optional


class DataType(enum.IntEnum):
    STRING = 0
    DOUBLE = 1
    BOOLEAN = 2
    # extensions (not in the reference)
    INT64 = 3
    INT32 = 4

    @property
    def is_numeric(self) -> bool:
        return self in (DataType.DOUBLE, DataType.INT64, DataType.INT32)


def promote(a: DataType, b: DataType) -> Optional[DataType]:
    """Binary numeric promotion (JLS 5.6.2) over {INT32, INT64, DOUBLE}."""
    if not (a.is_numeric and b.is_numeric):
        return None
    if DataType.DOUBLE in (a, b):
        return DataType.DOUBLE
    if DataType.INT64 in (a, b):
        return DataType.INT64
    return DataType.INT32


@dataclass(frozen=True)
class Field:
    name: str
    type: DataType


class Schema:
    def __init__(self, fields: List[Field]):
        self.fields = list(fields)
        self._by_name = {f.name: f for f in self.fields}

    def __getitem__(self, name: str) -> Optional[Field]:
        return self._by_name.get(name)

    def index_of(self, name: str) -> int:
        for i, f in enumerate(self.fields):
            if f.name == name:
                return i
        return -1

    def __repr__(self) -> str:
        return f"Schema({self.fields!r})"

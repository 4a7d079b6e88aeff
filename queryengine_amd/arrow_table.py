"""Apache Arrow <-> columnar table: a columnar wire / in-memory format on either side of the hot path.

The reference only has row-major ``MemoryTable`` and CSV scan leaves (``data/MemoryTable.kt``, ``data/CsvTable.kt``);
this is an extension in the spirit of SURVEY 8f row 3 ("columnar scan sources").  Arrow's physical layouts ARE the
layouts of ``qe_col_desc``: contiguous float64 / int64 / int32 value buffers, LSB-first validity and boolean
bitmaps, dictionary<int32, utf8> = codes + dictionary -- so ingestion is a reinterpretation of buffers, not a
conversion of values (numpy views where Arrow allows zero copy), and a result goes back the same way.

Host-side plumbing: nothing here evaluates an expression.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from .datatypes import DataType, Field, Schema
from .table import Column, ColumnarTable


def _column_from_arrow(arr) -> Column:
    import pyarrow as pa
    import pyarrow.compute as pc
    if isinstance(arr, pa.ChunkedArray):
        arr = arr.combine_chunks() if arr.num_chunks != 1 else arr.chunk(0)
    n = len(arr)
    valid = None
    if arr.null_count:
        valid = np.asarray(arr.is_valid().to_numpy(zero_copy_only=False), dtype=np.bool_)
    t = arr.type
    if pa.types.is_string(t) or pa.types.is_large_string(t):
        arr = pc.dictionary_encode(arr)
        t = arr.type
    if pa.types.is_dictionary(t):
        if not (pa.types.is_string(t.value_type) or pa.types.is_large_string(t.value_type)):
            raise TypeError(f"dictionary of {t.value_type} is not supported (dictionary<int, string> is)")
        codes = arr.indices.fill_null(0).to_numpy(zero_copy_only=False).astype(np.int32, copy=False)
        return Column(DataType.STRING, codes, valid, [s.as_py() for s in arr.dictionary])
    if pa.types.is_float64(t) or pa.types.is_float32(t) or pa.types.is_float16(t):
        data = arr.cast(pa.float64()).fill_null(0.0).to_numpy(zero_copy_only=False)
        return Column(DataType.DOUBLE, data, valid)
    if pa.types.is_int64(t):
        return Column(DataType.INT64, arr.fill_null(0).to_numpy(zero_copy_only=False), valid)
    if pa.types.is_int32(t) or pa.types.is_int16(t) or pa.types.is_int8(t) or pa.types.is_date32(t):
        return Column(DataType.INT32, arr.cast(pa.int32()).fill_null(0).to_numpy(zero_copy_only=False), valid)
    if pa.types.is_boolean(t):
        return Column(DataType.BOOLEAN, arr.fill_null(False).to_numpy(zero_copy_only=False), valid)
    raise TypeError(f"Arrow type {t} has no engine type (DOUBLE, INT64, INT32, BOOLEAN, STRING)")


def table_from_arrow(table, n: Optional[int] = None) -> ColumnarTable:
    """``pyarrow.Table`` / ``RecordBatch`` -> ``ColumnarTable`` (field names kept; every column pinned to HBM once by the
    GPU operators that scan it)."""
    fields: List[Field] = []
    cols: List[Column] = []
    for name, arr in zip(table.schema.names, table.columns):
        c = _column_from_arrow(arr)
        fields.append(Field(name, c.type))
        cols.append(c)
    return ColumnarTable(Schema(fields), cols)


def columns_to_arrow(names: Sequence[str], columns: Sequence[Column]):
    """Result columns (``Result.to_columns()``) -> ``pyarrow.Table``: DOUBLE/INT64/INT32/BOOLEAN arrays with validity,
    STRING as dictionary<int32, string>."""
    import pyarrow as pa
    arrays = []
    for c in columns:
        mask = None if c.valid is None else ~c.valid
        if c.type == DataType.STRING:
            codes = pa.array(c.data, type=pa.int32(), mask=mask)
            arrays.append(pa.DictionaryArray.from_arrays(codes, pa.array(c.dictionary or [], type=pa.string())))
        else:
            t = {DataType.DOUBLE: pa.float64(), DataType.INT64: pa.int64(), DataType.INT32: pa.int32(), DataType.BOOLEAN: pa.bool_()}[c.type]
            arrays.append(pa.array(c.data, type=t, mask=mask))
    return pa.table(arrays, names=list(names))

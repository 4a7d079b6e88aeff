"""CSV -> columnar table: the on-disk step in front of the hot path.

Follows the conversion rules of the reference's CSV scan leaves -- ``data/CsvTable.kt:12-29`` +
``operator/CsvSourceOperator.kt:52-76`` (commons-csv ``CSVFormat.DEFAULT.withFirstRecordAsHeader()
.withDelimiter(',').withIgnoreEmptyLines(true)``) and ``data/UnivocityCsvTable.kt:52-66``:

* the first record is the header; projected fields are located by header name;
* a missing trailing field or an empty string is NULL;
* STRING as is; BOOLEAN = ``String.toBoolean()`` (case-insensitive "true"); DOUBLE = ``String.toDouble()``
  (``java.lang.Double.parseDouble``: surrounding whitespace, NaN, Infinity, hex floats, d/f suffix).

Instead of boxing rows it builds one contiguous array per column (+ validity, + a dictionary in order of
first appearance for STRING), i.e. exactly what ``qe_batch_create`` pins to HBM.  Parsing is host-side I/O and
happens once per table, never inside a step.
"""
from __future__ import annotations

import csv
import re
from typing import Dict, List, Optional, Sequence

import numpy as np

from .datatypes import DataType, Field, Schema
from .table import Column, ColumnarTable

_DEC = re.compile(r"^[+-]?(?:\d+\.?\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?)[dDfF]?$")
_HEX = re.compile(r"^[+-]?0[xX](?:[0-9a-fA-F]+\.?[0-9a-fA-F]*|\.[0-9a-fA-F]+)[pP][+-]?\d+[dDfF]?$")


class NumberFormatException(ValueError):
    """java.lang.NumberFormatException"""


def java_parse_double(text: str) -> float:
    """java.lang.Double.parseDouble as invoked by Kotlin's String.toDouble()."""
    s = text.strip(" \t\n\r\x0b\x0c" + "".join(chr(c) for c in range(0x1d)))   # Java trims chars <= U+0020
    body = s.lstrip("+-")
    if body in ("NaN",):
        return float("nan")
    if body == "Infinity":
        return float("-inf") if s.startswith("-") else float("inf")
    if _HEX.match(s):
        return float.fromhex(s.rstrip("dDfF"))
    if _DEC.match(s):
        return float(s.rstrip("dDfF"))
    raise NumberFormatException(f'For input string: "{text}"')


def kotlin_to_boolean(text: str) -> bool:
    """String.toBoolean(): java.lang.Boolean.parseBoolean"""
    return text.lower() == "true"


def read_csv_columns(path: str, schema: Schema, projection: Optional[Sequence[str]] = None, encoding: str = "utf-8") -> ColumnarTable:
    """Parse `path` into a ColumnarTable holding the projected fields (default: every schema field)."""
    names = list(projection) if projection is not None else [f.name for f in schema.fields]
    fields = []
    for name in names:
        f = schema[name]
        if f is None:
            raise RuntimeError(f"projected field {name} not found in schema")          # CsvSourceOperator.kt:25-26
        if f.type not in (DataType.STRING, DataType.BOOLEAN, DataType.DOUBLE):
            raise TypeError("CSV sources carry the reference's three types only")
        fields.append(f)
    with open(path, newline="", encoding=encoding) as fh:
        reader = csv.reader(fh, delimiter=",", quotechar='"', doublequote=True)
        header = None
        for rec in reader:
            if rec:                    # withIgnoreEmptyLines(true)
                header = rec
                break
        if header is None:
            header = []
        hmap: Dict[str, int] = {}
        for i, h in enumerate(header):
            hmap[h] = i          # a duplicated header resolves to its LAST occurrence (commons-csv 1.8 builds its map with put())
        idx = []
        for f in fields:
            if f.name not in hmap:
                raise RuntimeError(f"projected field {f.name} not found in csv headers")   # :27-28
            idx.append(hmap[f.name])
        raw: List[List[Optional[str]]] = [[] for _ in fields]
        for rec in reader:
            if not rec:
                continue
            n = len(rec)
            for k, i in enumerate(idx):
                v = rec[i] if i < n else None          # :59, :71-73
                raw[k].append(v if v else None)        # isNullOrEmpty -> null
    cols = []
    for f, values in zip(fields, raw):
        n = len(values)
        valid = np.fromiter((v is not None for v in values), dtype=np.bool_, count=n)
        if f.type == DataType.DOUBLE:
            data = np.fromiter((java_parse_double(v) if v is not None else 0.0 for v in values), dtype=np.float64, count=n)
            cols.append(Column(DataType.DOUBLE, data, valid))
        elif f.type == DataType.BOOLEAN:
            data = np.fromiter((kotlin_to_boolean(v) if v is not None else False for v in values), dtype=np.bool_, count=n)
            cols.append(Column(DataType.BOOLEAN, data, valid))
        else:
            dictionary: List[str] = []
            index: Dict[str, int] = {}
            codes = np.zeros(n, dtype=np.int32)
            for j, v in enumerate(values):
                if v is None:
                    continue
                c = index.get(v)
                if c is None:
                    c = len(dictionary)
                    index[v] = c
                    dictionary.append(v)
                codes[j] = c
            cols.append(Column(DataType.STRING, codes, valid, dictionary))
    return ColumnarTable(Schema(fields), cols)


def read_csv_native(ctx, path_or_bytes, schema: Schema, projection: Optional[Sequence[str]] = None) -> ColumnarTable:
    """The same conversion through the C ABI (qe_csv_parse / qe_csv_parse_file in libqe_hip.so: what a JVM host binds; a
    planning-only context suffices -- parsing is host work).  Returns a ColumnarTable over copies of the parsed columns;
    ``table.native`` keeps the qe_csv_table handle so that `pin_csv` can copy it to HBM without another conversion."""
    import ctypes as C
    from . import native as N
    names = list(projection) if projection is not None else [f.name for f in schema.fields]
    fields = []
    for name in names:
        f = schema[name]
        if f is None:
            raise RuntimeError(f"projected field {name} not found in schema")          # CsvSourceOperator.kt:25-26
        fields.append(f)
    lib = N.lib()
    cnames = (C.c_char_p * max(1, len(fields)))(*[f.name.encode("utf-8") for f in fields])
    ctypes_ = (C.c_int32 * max(1, len(fields)))(*[int(f.type) for f in fields])
    h = C.c_void_p()
    if isinstance(path_or_bytes, (bytes, bytearray)):
        st = lib.qe_csv_parse(ctx.handle, bytes(path_or_bytes), len(path_or_bytes), len(fields), cnames, ctypes_, C.byref(h))
    else:
        st = lib.qe_csv_parse_file(ctx.handle, str(path_or_bytes).encode("utf-8"), len(fields), cnames, ctypes_, C.byref(h))
    if st != 0:
        msg = (lib.qe_last_error(ctx.handle) or b"").decode("utf-8", "replace")
        if "not found in csv headers" in msg:
            raise RuntimeError(msg)
        if "NumberFormatException" in msg:
            raise NumberFormatException(msg)
        raise N.QeError(st, msg)
    n = int(lib.qe_csv_nrows(h))
    nwords = max(1, (n + 63) // 64)
    cols = []
    for j, f in enumerate(fields):
        d = N.ColDesc()
        N.check(ctx.handle, lib.qe_csv_column(h, j, C.byref(d)))
        valid = None
        if d.validity:
            words = np.ctypeslib.as_array(C.cast(d.validity, C.POINTER(C.c_uint64)), shape=(nwords,)).copy()
            valid = np.unpackbits(words.view(np.uint8), bitorder="little")[:n].astype(np.bool_)
        if f.type == DataType.DOUBLE:
            data = np.ctypeslib.as_array(C.cast(d.data, C.POINTER(C.c_double)), shape=(max(n, 1),))[:n].copy()
            cols.append(Column(DataType.DOUBLE, data, valid))
        elif f.type == DataType.BOOLEAN:
            words = np.ctypeslib.as_array(C.cast(d.data, C.POINTER(C.c_uint64)), shape=(nwords,)).copy()
            cols.append(Column(DataType.BOOLEAN, np.unpackbits(words.view(np.uint8), bitorder="little")[:n].astype(np.bool_), valid))
        else:
            codes = np.ctypeslib.as_array(C.cast(d.data, C.POINTER(C.c_int32)), shape=(max(n, 1),))[:n].copy()
            m = lib.qe_dict_size(d.dict)
            dictionary = [lib.qe_dict_entry(d.dict, i).decode("utf-8") for i in range(m)]
            cols.append(Column(DataType.STRING, codes, valid, dictionary))
    t = ColumnarTable(Schema(fields), cols)
    t.native = _NativeCsv(ctx, h)
    return t


class _NativeCsv:
    """Owner of a qe_csv_table handle."""

    def __init__(self, ctx, handle):
        self.ctx, self.handle = ctx, handle

    def pin(self):
        """qe_csv_pin: the parsed columns -> HBM (one H2D per column), as an engine.DeviceBatch."""
        import ctypes as C
        from . import engine as E
        from . import native as N
        h = C.c_void_p()
        N.check(self.ctx.handle, self.ctx._lib.qe_csv_pin(self.ctx.handle, self.handle, C.byref(h)))
        return E.DeviceBatch(self.ctx, h)

    def __del__(self):
        try:
            if self.handle and self.ctx.handle:
                self.ctx._lib.qe_csv_free(self.ctx.handle, self.handle)
        except Exception:
            pass
        self.handle = None


class CsvColumnarTable(ColumnarTable):
    """Drop-in for ``CsvTable(file, schema)`` (data/CsvTable.kt:12): parsed once, scanned columnar."""

    def __init__(self, path: str, schema: Schema):
        t = read_csv_columns(path, schema)
        super().__init__(t.schema, t.columns)
        self.path = path

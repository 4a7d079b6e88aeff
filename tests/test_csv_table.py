"""CSV -> columns conversion rules of CsvSourceOperator.kt:52-76 / UnivocityCsvTable.kt:52-66 (host side)."""
import math
import os

import numpy as np
import pytest

from queryengine_amd import DataType, Field, Schema
from queryengine_amd.csv_table import CsvColumnarTable, NumberFormatException, java_parse_double, kotlin_to_boolean, read_csv_columns

S, D, B = DataType.STRING, DataType.DOUBLE, DataType.BOOLEAN

CSV = '''id,country,price,paid,note
1,DE,100.5,true,"hello, world"

2,"AT",  2.5e1 ,TRUE,"say ""hi"""
3,,NaN,false,
4,CH,0x1p3,yes
5,DE,-Infinity,,x
6,DE,7d,False,y,extra
'''


def test_java_parse_double_and_boolean():
    assert java_parse_double(" 2.5e1 ") == 25.0 and java_parse_double("7d") == 7.0 and java_parse_double("0x1p3") == 8.0
    assert java_parse_double(".5") == 0.5 and java_parse_double("5.") == 5.0 and java_parse_double("+1E2f") == 100.0
    assert math.isnan(java_parse_double("NaN")) and java_parse_double("-Infinity") == float("-inf")
    for bad in ("", "abc", "1_0", "inf", "nan", "1e", "0x10", "1,5", "--1"):
        with pytest.raises(NumberFormatException):
            java_parse_double(bad)
    assert kotlin_to_boolean("true") and kotlin_to_boolean("TrUe") and not kotlin_to_boolean("yes") and not kotlin_to_boolean("1")


def test_csv_rules(tmp_path):
    p = tmp_path / "orders.csv"
    p.write_text(CSV, encoding="utf-8")
    schema = Schema([Field("id", S), Field("country", S), Field("price", D), Field("paid", B), Field("note", S)])
    t = CsvColumnarTable(str(p), schema)
    assert t.nrows == 6                                           # the empty line is ignored
    rows = [[c.value(i) for c in t.columns] for i in range(t.nrows)]
    assert rows[0] == ["1", "DE", 100.5, True, "hello, world"]
    assert rows[1] == ["2", "AT", 25.0, True, 'say "hi"']
    assert rows[2][:2] == ["3", None] and math.isnan(rows[2][2]) and rows[2][3:] == [False, None]   # empty -> null
    assert rows[3] == ["4", "CH", 8.0, False, None]               # short record -> null; "yes" is not true
    assert rows[4] == ["5", "DE", float("-inf"), None, "x"]
    assert rows[5] == ["6", "DE", 7.0, False, "y"]                # extra fields ignored
    assert t.column("country").dictionary == ["DE", "AT", "CH"]   # first appearance order
    # projection by header name, any order; unknown names raise like the reference
    sub = read_csv_columns(str(p), schema, ["price", "id"])
    assert [f.name for f in sub.schema.fields] == ["price", "id"] and sub.columns[1].to_list()[:2] == ["1", "2"]
    with pytest.raises(RuntimeError):
        read_csv_columns(str(p), schema, ["nope"])
    with pytest.raises(RuntimeError):
        read_csv_columns(str(p), Schema([Field("missing", S)]), ["missing"])


def test_native_csv_parser_matches_the_python_restatement(tmp_path, native_lib):
    """qe_csv_parse / qe_csv_parse_file (C ABI, what the JVM host binds) against the Python restatement of the same rules on
    the fixture above and on quoting / line-end / Unicode edge cases; a planning-only context suffices (host work)."""
    from queryengine_amd import engine as E
    from queryengine_amd.csv_table import read_csv_native
    ctx = E.Context(device=None, jit_cache_dir=str(tmp_path / "jit"))
    schema = Schema([Field("id", S), Field("country", S), Field("price", D), Field("paid", B), Field("note", S)])
    more = ("id,country,price,paid,note\r\n"                      # CRLF line ends
            '7,"multi\nline",1e-3,true,"a,b"\r\n'
            "\r\n"
            '8,Z\u00fcrich \U0001F600,+.5,tRuE,""\r'              # a lone CR ends a record too; quoted empty string = NULL
            "9,,Infinity,,last")                                  # no line end after the last record
    for k, text in enumerate((CSV, more)):
        p = tmp_path / f"t{k}.csv"
        p.write_bytes(text.encode("utf-8"))
        for proj in (None, ["price", "id"], ["note", "paid", "country"]):
            want = read_csv_columns(str(p), schema, proj)
            for src in (str(p), text.encode("utf-8")):
                got = read_csv_native(ctx, src, schema, proj)
                assert got.nrows == want.nrows and [f.name for f in got.schema.fields] == [f.name for f in want.schema.fields]
                for g, w in zip(got.columns, want.columns):
                    assert g.type == w.type and g.dictionary == w.dictionary
                    gv = g.valid if g.valid is not None else np.ones(len(g), bool)
                    wv = w.valid if w.valid is not None else np.ones(len(w), bool)
                    assert np.array_equal(gv, wv)
                    if g.type == D:
                        assert np.array_equal(g.data[gv].view(np.uint64), w.data[wv].view(np.uint64))
                    else:
                        assert np.array_equal(g.data[gv], w.data[wv])
    t = read_csv_native(ctx, more.encode("utf-8"), schema)
    assert t.column("country").to_list() == ["multi\nline", "Z\u00fcrich \U0001F600", None]
    assert t.column("note").to_list() == ["a,b", None, "last"] and t.column("price").to_list() == [0.001, 0.5, float("inf")]
    # errors speak the reference's language
    with pytest.raises(RuntimeError):
        read_csv_native(ctx, CSV.encode(), Schema([Field("missing", S)]))
    with pytest.raises(RuntimeError):       # a byte order mark stays part of the first header name (FileReader(file, UTF_8))
        read_csv_native(ctx, ("\ufeff" + CSV).encode("utf-8"), schema, ["id"])
    for bad in ("1_0", "inf", "nan", "0x10", "1e", "abc", "1,5"):
        with pytest.raises(NumberFormatException):
            read_csv_native(ctx, f'price\n"{bad}"\n'.encode(), Schema([Field("price", D)]))
    ok = read_csv_native(ctx, b"price\n 0x1.8p1 \n7D\n-NaN\n1e400\n", Schema([Field("price", D)])).columns[0].data
    assert ok[0] == 3.0 and ok[1] == 7.0 and math.isnan(ok[2]) and ok[3] == float("inf")
    with pytest.raises(Exception):
        read_csv_native(ctx, b'a\n"unterminated\n', Schema([Field("a", S)]))
    # a duplicated header name resolves to its LAST occurrence: commons-csv 1.8's CSVFormat.DEFAULT allows duplicate headers
    # and fills its header map with put() (a later column replaces an earlier one).  Self-derived expectation: the library is
    # not in the image and the reference holds no fixture with duplicate headers -- parity unpinned.
    dup = b"x,y,x\n1,2,3\n4,5,\n"
    sch = Schema([Field("x", D), Field("y", D)])
    pd_ = tmp_path / "dup.csv"
    pd_.write_bytes(dup)
    for t2 in (read_csv_native(ctx, dup, sch), read_csv_columns(str(pd_), sch)):
        assert t2.column("x").to_list() == [3.0, None] and t2.column("y").to_list() == [2.0, 5.0]
    ctx.close()


@pytest.mark.gpu
def test_native_csv_table_pinned_and_queried(tmp_path, gpu_ctx, oracle):
    """qe_csv_parse_file -> qe_csv_pin -> qe_filter_project: the CSV scan leaf through the C ABI end to end."""
    from queryengine_amd import ColumnExpression, Function, FunctionExpression, NumericLiteralExpression
    from queryengine_amd import engine as E
    from queryengine_amd.csv_table import read_csv_native
    p = tmp_path / "orders.csv"
    p.write_text(CSV, encoding="utf-8")
    schema = Schema([Field("id", S), Field("country", S), Field("price", D), Field("paid", B), Field("note", S)])
    t = read_csv_native(gpu_ctx, str(p), schema)
    batch = t.native.pin()
    assert batch.nrows == 6 and batch.ncols == 5
    price, paid, country = ColumnExpression("price", 2, D), ColumnExpression("paid", 3, B), ColumnExpression("country", 1, S)
    flt = FunctionExpression(Function.AND, [paid, FunctionExpression(Function.CMP_LT, [price, NumericLiteralExpression(1000.0)], B)], B)
    projs = [country, FunctionExpression(Function.MUL, [price, NumericLiteralExpression(2.0)], D)]
    res = E.filter_project(gpu_ctx, batch, gpu_ctx.compile(flt), [gpu_ctx.compile(e) for e in projs])
    got = res.to_columns()
    want = oracle.filter_project(t.columns, flt, projs, oracle.BYTECODE_COMPILER)
    assert got[0].to_list() == want[0].to_list() == ["DE", "AT"] and got[1].to_list() == want[1].to_list() == [201.0, 50.0]
    res.free(); batch.free()


@pytest.mark.gpu
def test_query_over_csv_table(tmp_path, gpu_ctx):
    from queryengine_amd.planner import Mode, query
    p = tmp_path / "orders.csv"
    p.write_text(CSV, encoding="utf-8")
    schema = Schema([Field("id", S), Field("country", S), Field("price", D), Field("paid", B), Field("note", S)])
    t = CsvColumnarTable(str(p), schema)
    rows = query("orders", "SELECT id, price * 2 FROM orders WHERE paid AND price < 1000", Mode.GPU_FUSED, table=t, ctx=gpu_ctx)
    assert rows == [["1", 201.0], ["2", 50.0]]
    rows = query("orders", "SELECT country, COUNT(id), SUM(price) FROM orders WHERE price < 1000 AND price > 0", Mode.GPU_FUSED, table=t, ctx=gpu_ctx)
    assert rows == [["DE", 2, 107.5], ["AT", 1, 25.0], ["CH", 1, 8.0]]

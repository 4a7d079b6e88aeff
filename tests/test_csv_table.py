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

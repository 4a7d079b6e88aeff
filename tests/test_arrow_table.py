"""Arrow ingestion / export around the path (host-side; the GPU part checks a query over an Arrow-born table)."""
import numpy as np
import pytest

pa = pytest.importorskip("pyarrow")

from queryengine_amd import DataType
from queryengine_amd.arrow_table import columns_to_arrow, table_from_arrow


def _arrow_table():
    return pa.table({
        "id": pa.array([1, 2, None, 4, 5], type=pa.int64()),
        "price": pa.array([10.5, None, 30.0, float("nan"), -0.0], type=pa.float64()),
        "paid": pa.array([True, False, None, True, True]),
        "country": pa.array(["DE", "AT", None, "DE", "CH"]),
        "day": pa.array([8766, 8767, 8768, None, 9000], type=pa.int32()),
    })


def test_arrow_round_trip_keeps_values_nulls_and_dictionaries():
    t = table_from_arrow(_arrow_table())
    assert [f.type for f in t.schema.fields] == [DataType.INT64, DataType.DOUBLE, DataType.BOOLEAN, DataType.STRING, DataType.INT32]
    assert [t.column("country").value(i) for i in range(5)] == ["DE", "AT", None, "DE", "CH"]
    assert t.column("country").dictionary == ["DE", "AT", "CH"]            # first-appearance order, like the CSV source
    assert [t.column("id").value(i) for i in range(5)] == [1, 2, None, 4, 5]
    assert t.column("paid").value(2) is None and t.column("paid").value(1) is False
    back = columns_to_arrow([f.name for f in t.schema.fields], t.columns)
    want = _arrow_table()
    for name in want.schema.names:
        a, b = back.column(name).to_pylist(), want.column(name).to_pylist()
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert (x is None and y is None) or (isinstance(x, float) and x != x and y != y) or x == y, (name, x, y)
    # chunked input and a pre-encoded dictionary column are accepted too
    chunked = pa.chunked_array([pa.array(["x", "y"]), pa.array(["y", None])])
    t2 = table_from_arrow(pa.table({"s": chunked, "d": pa.array(["k", "k", "m", None]).dictionary_encode()}))
    assert [t2.column("s").value(i) for i in range(4)] == ["x", "y", "y", None]
    assert [t2.column("d").value(i) for i in range(4)] == ["k", "k", "m", None]
    with pytest.raises(TypeError):
        table_from_arrow(pa.table({"b": pa.array([b"raw"])}))


@pytest.mark.gpu
def test_query_over_an_arrow_table(gpu_ctx):
    from queryengine_amd.planner import Mode, query
    t = table_from_arrow(_arrow_table())
    rows = query("orders", "SELECT id, price * 2 FROM orders WHERE paid AND country = 'DE'", Mode.GPU_FUSED, table=t, ctx=gpu_ctx)
    assert rows[0] == [1, 21.0] and rows[1][0] == 4 and rows[1][1] != rows[1][1]        # NaN * 2
    rows = query("orders", "SELECT country, COUNT(id), SUM(day) FROM orders WHERE day < 9500", Mode.GPU_FUSED, table=t, ctx=gpu_ctx)
    assert rows == [["DE", 1, 8766.0], ["AT", 1, 8767.0], [None, 0, 8768.0], ["CH", 1, 9000.0]]

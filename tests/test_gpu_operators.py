"""The reference's operator / query API on the GPU path: query(), buildPhysicalPlan, re-openable
operators, Filter-only plans, the aggregate golden values of SimpleSumBenchmark."""
import json
import os

import numpy as np
import pytest

from queryengine_amd import (Column, ColumnarTable, DataType, Field, Schema, TableRegistry)
from queryengine_amd import native as N
from queryengine_amd.operators import GpuFilterProjectOperator, GpuGlobalAggregationOperator, forEach, map as op_map
from queryengine_amd.planner import Mode, buildLogicalPlan, buildPhysicalPlan, query
from queryengine_amd.sql import parseQuery

pytestmark = pytest.mark.gpu
D, I64, I32, B, S = DataType.DOUBLE, DataType.INT64, DataType.INT32, DataType.BOOLEAN, DataType.STRING
GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))


def _orders_table():
    # the table of Main.kt:29-45 (id, country, net_price, net_shipping_cost), typed DOUBLE like the reference
    schema = Schema([Field("id", S), Field("country", S), Field("net_price", D), Field("net_shipping_cost", D)])
    rows = [["1", "DE", 100.0, 5.0], ["2", "DE", 200.0, 10.0], ["3", "AT", 100.0, 5.0], ["4", "CH", 40.0, 10.0],
            ["5", "DE", 50.0, None], ["6", None, 10.0, 1.0]]
    return ColumnarTable.from_rows(schema, rows)


@pytest.mark.parametrize("mode", [Mode.GPU_FUSED, Mode.GPU_PER_NODE])
def test_query_filter_project(mode, gpu_ctx, gpu_ctx_per_node):
    ctx = gpu_ctx if mode == Mode.GPU_FUSED else gpu_ctx_per_node
    t = _orders_table()
    rows = query("orders", "SELECT net_price + net_shipping_cost, country FROM orders WHERE net_price >= 50", mode, table=t, ctx=ctx)
    assert rows == [[105.0, "DE"], [210.0, "DE"], [105.0, "AT"], [None, "DE"]]
    # Filter(Scan): identity projection removed (Optimizer.kt:33-35) -> the scan row passes through
    rows = query("orders", "SELECT id, country FROM orders WHERE country = 'DE'", mode, table=t, ctx=ctx)
    assert rows == [["1", "DE"], ["2", "DE"], ["5", "DE"]]
    # null predicate rows are dropped (FilterOperator.kt:20), IF / literals / unary minus folded by the parser
    rows = query("orders", "SELECT IF net_shipping_cost > 5 THEN 'big' ELSE 'small' END, -1.5 * net_price FROM orders "
                 "WHERE NOT (country = 'AT')", mode, table=t, ctx=ctx)
    assert rows == [["small", -150.0], ["big", -300.0], ["big", -60.0], [None, -75.0]]
    # NOT binds tighter than comparison in the reference's grammar (Query.g4:31-34): (NOT country) = 'AT'
    from queryengine_amd.typecheck import TypeCheckException
    with pytest.raises(TypeCheckException):
        query("orders", "SELECT id FROM orders WHERE NOT country = 'AT'", mode, table=t, ctx=ctx)


@pytest.mark.parametrize("mode", [Mode.GPU_FUSED, Mode.GPU_PER_NODE])
def test_operator_is_reopenable_and_pins_batch_once(mode, gpu_ctx, gpu_ctx_per_node):
    """T/SimpleSumBenchmark.java:63-94 re-runs open/next/close on one plan."""
    ctx = gpu_ctx if mode == Mode.GPU_FUSED else gpu_ctx_per_node
    rng = np.random.default_rng(0)
    n = 5000
    t = ColumnarTable(Schema([Field("a", I64), Field("c", D)]),
                      [Column(I64, rng.integers(0, 1000, n)), Column(D, rng.random(n))])
    r = TableRegistry()
    r.register("t", t)
    plan = buildLogicalPlan(r, parseQuery("SELECT a + a, c FROM t WHERE a < 100 AND c < 0.5"))
    op = buildPhysicalPlan(r, plan, mode, ctx)
    assert isinstance(op, GpuFilterProjectOperator)
    with pytest.raises(RuntimeError):
        op.next()                                         # "Operator not initialized"
    first = op_map(op, lambda row: row)
    second = op_map(op, lambda row: row)
    assert first == second and len(first) > 0
    keep = (t.columns[0].data < 100) & (t.columns[1].data < 0.5)
    assert [row[0] for row in first] == list(2 * t.columns[0].data[keep])
    assert len(t.__dict__["_device_batches"]) == 1        # pinned to HBM once, reused by both opens


def test_simple_sum_benchmark_known_answers(gpu_ctx):
    """SELECT SUM(foo + 10*bar) FROM table, foo = bar = (double)(i / 1000): 0.0 and 5494500000.0
    (SimpleSumBenchmark.java:41-53); every partial sum is an integer < 2^53, so the tree reduction is exact."""
    for case in GOLDEN["aggregate_cases"]:
        n = case["size"]
        v = (np.arange(n) // 1000).astype(np.float64)
        t = ColumnarTable(Schema([Field("foo", D), Field("bar", D)]), [Column(D, v), Column(D, v.copy())])
        r = TableRegistry()
        r.register("table", t)
        op = buildPhysicalPlan(r, buildLogicalPlan(r, parseQuery(case["sql"])), Mode.GPU_FUSED, gpu_ctx)
        assert isinstance(op, GpuGlobalAggregationOperator)
        for _ in range(2):
            assert op_map(op, lambda row: row) == [case["expected"]]


def test_global_aggregates_match_oracle(gpu_ctx, oracle):
    """MIN / MAX / SUM / COUNT / AVG with nulls, a filter, an empty selection (Accumulators.kt:26-107)."""
    from queryengine_amd import ColumnExpression, Function, FunctionExpression, NumericLiteralExpression
    from queryengine_amd import engine as E
    rng = np.random.default_rng(3)
    n = 100_000
    a = Column(D, rng.normal(0, 10, n), rng.random(n) > 0.1)
    b = Column(I64, rng.integers(-50, 50, n))
    A_, B_ = ColumnExpression("a", 0, D), ColumnExpression("b", 1, I64)
    exprs = [A_, A_, A_, A_, A_, B_, FunctionExpression(Function.MUL, [A_, NumericLiteralExpression(2.0)], D)]
    aggs = [oracle.MIN, oracle.MAX, oracle.SUM, oracle.COUNT, oracle.AVG, oracle.SUM, oracle.SUM]
    batch = E.DeviceBatch.from_columns(gpu_ctx, [a, b])
    for flt in (None, FunctionExpression(Function.CMP_LT, [B_, NumericLiteralExpression(0.0)], B),
                FunctionExpression(Function.CMP_LT, [B_, NumericLiteralExpression(-1000.0)], B)):
        cf = gpu_ctx.compile(flt) if flt is not None else None
        got, nsel = E.filter_aggregate(gpu_ctx, batch, cf, [gpu_ctx.compile(e) for e in exprs], aggs)
        want, wsel = oracle.filter_aggregate([a, b], flt, exprs, aggs, oracle.BYTECODE_COMPILER)
        assert nsel == wsel
        for g, w, fn in zip(got, want, aggs):
            if w is None:
                assert g is None
            elif fn in (oracle.MIN, oracle.MAX, oracle.COUNT):
                assert g == w                                  # order independent: exact
            else:
                # SUM / AVG: fixed-shape tree vs the reference's sequential order; bound n * eps * sum|x|
                assert abs(g - w) <= 1e-9 * max(1.0, abs(w)), (g, w)
    batch.free()


def test_query_test_group_by_multiple_columns(gpu_ctx):
    """The reference's only end-to-end test (T/evaluator/QueryTest.kt:15-30) through query()."""
    schema = Schema([Field("foo", S), Field("bar", S), Field("num", D)])
    t = ColumnarTable.from_rows(schema, [["a", "A", 1.0], ["a", "B", 2.0], ["a", "B", 3.0], ["b", "B", 4.0],
                                         ["b", "B", None], ["c", None, None]])
    actual = query("table", "SELECT bar, SUM(num), foo FROM table", Mode.GPU_FUSED, table=t, ctx=gpu_ctx)
    assert actual == [["A", 1.0, "a"], ["B", 5.0, "a"], ["B", 4.0, "b"], [None, None, "c"]]


def test_main_kt_orders_example(gpu_ctx):
    """Main.kt:29-50 (derived): SELECT SUM(net_price+net_shipping_cost)*1.25, country FROM orders -> CH 62.5, AT 131.25,
    DE 456.25 in the order of ORDER BY 1 (OrderByOperator.kt:9-12 on top of the GPU operators)."""
    schema = Schema([Field("id", S), Field("country", S), Field("net_price", D), Field("net_shipping_cost", D)])
    rows = [["1", "DE", 100.0, 5.0], ["2", "DE", 200.0, 10.0], ["3", "AT", 100.0, 5.0], ["4", "CH", 40.0, 10.0], ["5", "DE", 50.0, 0.0]]
    t = ColumnarTable.from_rows(schema, rows)
    actual = query("orders", "SELECT SUM(net_price + net_shipping_cost) * 1.25, country FROM orders", Mode.GPU_FUSED, table=t, ctx=gpu_ctx)
    assert sorted(actual) == [[62.5, "CH"], [131.25, "AT"], [456.25, "DE"]]
    ordered = query("orders", "SELECT SUM(net_price + net_shipping_cost) * 1.25, country FROM orders ORDER BY 1", Mode.GPU_FUSED,
                    table=t, ctx=gpu_ctx)
    assert ordered == [[62.5, "CH"], [131.25, "AT"], [456.25, "DE"]]
    ordered = query("orders", "SELECT id, net_price FROM orders WHERE net_price >= 50 ORDER BY 2", Mode.GPU_FUSED, table=t, ctx=gpu_ctx)
    assert ordered == [["5", 50.0], ["1", 100.0], ["3", 100.0], ["2", 200.0]]      # stable: id 1 before id 3
    # insertion order of the groups: DE first
    assert [r[1] for r in actual] == ["DE", "AT", "CH"]
    # COUNT keeps the reference's Int (Accumulators.kt:26-36), expressions over aggregates run on the GPU
    actual = query("orders", "SELECT country, COUNT(id), MAX(net_price) / MIN(net_price) FROM orders WHERE net_price > 45", Mode.GPU_FUSED,
                   table=t, ctx=gpu_ctx)
    assert actual == [["DE", 3, 4.0], ["AT", 1, 1.0]]


def test_group_by_matches_oracle(gpu_ctx, oracle):
    """Dictionary + boolean keys with nulls, every accumulator, a filter; 200k rows; groups in insertion order."""
    from queryengine_amd import ColumnExpression, Function, FunctionExpression, NumericLiteralExpression
    from queryengine_amd import engine as E
    rng = np.random.default_rng(5)
    n = 200_000
    d = ["k%03d" % i for i in range(37)]
    s = Column(S, rng.integers(0, len(d), n).astype(np.int32), rng.random(n) > 0.05, d)
    p = Column(B, rng.random(n) > 0.5, rng.random(n) > 0.1)
    x = Column(D, np.round(rng.normal(0, 100, n)), rng.random(n) > 0.2)        # integer-valued: exact sums in any order
    y = Column(I64, rng.integers(-1000, 1000, n))
    Sx, P, X, Y = ColumnExpression("s", 0, S), ColumnExpression("p", 1, B), ColumnExpression("x", 2, D), ColumnExpression("y", 3, I64)
    keys = [Sx, P]
    exprs = [X, X, X, X, X, FunctionExpression(Function.ADD, [Y, Y], I64)]
    aggs = [oracle.SUM, oracle.MIN, oracle.MAX, oracle.COUNT, oracle.AVG, oracle.SUM]
    flt = FunctionExpression(Function.CMP_LT, [Y, NumericLiteralExpression(500.0)], B)
    batch = E.DeviceBatch.from_columns(gpu_ctx, [s, p, x, y])
    for f in (None, flt):
        res = E.filter_groupby(gpu_ctx, batch, gpu_ctx.compile(f) if f is not None else None,
                               [gpu_ctx.compile(k) for k in keys], [gpu_ctx.compile(e) for e in exprs], aggs)
        cols = res.to_columns()
        got = [[c.value(i) for c in cols] for i in range(res.count)]
        res.free()
        want = oracle.filter_groupby([s, p, x, y], f, keys, exprs, aggs, oracle.BYTECODE_COMPILER)
        assert len(got) == len(want) and len(got) > 100
        for g, w in zip(got, want):
            assert g[:2] == w[:2]                      # same groups, same (insertion) order
            for a, b, fn in zip(g[2:], w[2:], aggs):
                if b is None or fn != oracle.AVG:
                    assert a == b, (g, w)
                else:
                    assert abs(a - b) <= 1e-12 * max(1.0, abs(b))
    batch.free()


@pytest.mark.parametrize("case", ["partitioned", "partitioned_wide_domain", "global_atomics"])
def test_group_by_large_domain_matches_oracle(case, oracle):
    """Domains whose accumulator table does not fit LDS: the partitioned path (count -> scan -> scatter -> per-partition
    LDS aggregation) and, with debug bit 256, the global-atomic path.  Ragged last chunk, null keys, null values, a
    filter; groups must come out in the reference's insertion order (GroupByAggregationOperator.kt:22)."""
    from queryengine_amd import ColumnExpression, Function, FunctionExpression, NumericLiteralExpression
    from queryengine_amd import engine as E
    rng = np.random.default_rng(11)
    wide = case == "partitioned_wide_domain"
    n = 400_003 if wide else 300_017
    nkeys = 300_000 if wide else 3000
    ctx = E.Context(device=0, tuning=[0, 0, 0, 0, 0, 256, 0, 0] if case == "global_atomics" else [])
    d = ["k%06d" % i for i in range(nkeys)]
    s = Column(S, rng.integers(0, nkeys, n).astype(np.int32), rng.random(n) > 0.05, d)
    p = Column(B, rng.random(n) > 0.5, rng.random(n) > 0.1)
    x = Column(D, np.round(rng.normal(0, 100, n)), rng.random(n) > 0.2)
    y = Column(I64, rng.integers(-1000, 1000, n))
    Sx, P, X, Y = ColumnExpression("s", 0, S), ColumnExpression("p", 1, B), ColumnExpression("x", 2, D), ColumnExpression("y", 3, I64)
    keys = [Sx, P]
    if wide:
        exprs, aggs = [X], [oracle.SUM]
    else:
        exprs = [X, X, X, X, X, FunctionExpression(Function.ADD, [Y, Y], I64)]
        aggs = [oracle.SUM, oracle.MIN, oracle.MAX, oracle.COUNT, oracle.AVG, oracle.SUM]
    flt = FunctionExpression(Function.CMP_LT, [Y, NumericLiteralExpression(500.0)], B)
    batch = E.DeviceBatch.from_columns(ctx, [s, p, x, y])
    for f in (None, flt):
        res = E.filter_groupby(ctx, batch, ctx.compile(f) if f is not None else None,
                               [ctx.compile(k) for k in keys], [ctx.compile(e) for e in exprs], aggs)
        cols = res.to_columns()
        got = [[c.value(i) for c in cols] for i in range(res.count)]
        res.free()
        want = oracle.filter_groupby([s, p, x, y], f, keys, exprs, aggs, oracle.BYTECODE_COMPILER)
        assert len(got) == len(want) and len(got) > 5000
        for g, w in zip(got, want):
            assert g[:2] == w[:2]
            for a, b, fn in zip(g[2:], w[2:], aggs):
                if b is None or fn != oracle.AVG:
                    assert a == b, (g, w)
                else:
                    assert abs(a - b) <= 1e-12 * max(1.0, abs(b))
    batch.free()
    ctx.close()


@pytest.mark.parametrize("n", [1, 63, 1025, 4096, 4097, 65535, 65536, 65537, 131072 + 5, 200_000])
def test_group_by_partitioned_sizes_around_tiles_and_chunks(n, oracle):
    """The partitioned path works on workgroup tiles of 4096 rows (one 1024-row sub-tile per wave) and chunks of 16 tiles,
    and pads every tile's records of a partition to whole 128-byte lines: batches smaller than a tile (idle waves), exactly
    one tile / one chunk, one row more, and ragged tails must all give the reference's groups in insertion order."""
    from queryengine_amd import ColumnExpression, Function, FunctionExpression, NumericLiteralExpression
    from queryengine_amd import engine as E
    rng = np.random.default_rng(n)
    nkeys = 5000
    ctx = E.Context(device=0)
    d = ["k%05d" % i for i in range(nkeys)]
    s = Column(S, rng.integers(0, nkeys, n).astype(np.int32), None, d)
    x = Column(D, np.round(rng.normal(0, 100, n)), rng.random(n) > 0.1)
    y = Column(D, np.round(rng.normal(0, 10, n)))
    Sx, X, Y = ColumnExpression("s", 0, S), ColumnExpression("x", 1, D), ColumnExpression("y", 2, D)
    flt = FunctionExpression(Function.CMP_LT, [Y, NumericLiteralExpression(5.0)], B)
    batch = E.DeviceBatch.from_columns(ctx, [s, x, y])
    for f, exprs, aggs in ((None, [Y, Y], [oracle.SUM, oracle.COUNT]), (flt, [X, Y, X], [oracle.SUM, oracle.MAX, oracle.COUNT]),
                           (FunctionExpression(Function.CMP_LT, [Y, NumericLiteralExpression(-1000.0)], B), [Y], [oracle.MIN])):
        res = E.filter_groupby(ctx, batch, ctx.compile(f) if f is not None else None, [ctx.compile(Sx)], [ctx.compile(e) for e in exprs], aggs)
        cols = res.to_columns()
        got = [[c.value(i) for c in cols] for i in range(res.count)]
        res.free()
        want = oracle.filter_groupby([s, x, y], f, [Sx], exprs, aggs, oracle.BYTECODE_COMPILER)
        assert got == want, (n, aggs)
    batch.free()
    ctx.close()


def test_group_by_1m_keys_min_max_full_size_properties(oracle):
    """SELECT k, MIN(v), MAX(v) over 1 B rows with 1 000 000 distinct DOUBLE keys (Tripdata.kt:27-31's shape at scale): the
    hash-partitioned form with counter-less 32-byte table entries (no input can be NULL, nothing counts), 4096 buckets per
    partition, records in 128-byte lines.  Properties: every key is a group; per group MIN <= MAX inside [0, 1); the smallest MIN
    and the largest MAX are the independent global MIN / MAX; with 1000 values per key MIN and MAX sit near the ends; the first
    groups are the keys of the first rows in their order (the oracle walks that window)."""
    from queryengine_amd import ColumnExpression
    from queryengine_amd import engine as E
    from queryengine_amd.workloads import GenColumn
    n, nkeys = 1_000_000_000, 1_000_000
    ctx = E.Context(device=0)
    gen = [GenColumn("k", D, N.GEN_F64_MOD, 0, modulus=nkeys), GenColumn("v", D, N.GEN_F64_UNIT, 1)]
    K, V = ColumnExpression("k", 0, D), ColumnExpression("v", 1, D)
    batch = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in gen], n)
    (gmin, gmax), nsel = E.filter_aggregate(ctx, batch, None, [ctx.compile(V), ctx.compile(V)], [oracle.MIN, oracle.MAX])
    assert nsel == n
    for rep in range(2):
        res = E.filter_groupby(ctx, batch, None, [ctx.compile(K)], [ctx.compile(V), ctx.compile(V)], [oracle.MIN, oracle.MAX])
        cols, ngroups = res.to_columns(), res.count
        res.free()
        assert ngroups == nkeys
        lo, hi = cols[1].data, cols[2].data
        assert cols[1].valid is None or cols[1].valid.all()
        assert (lo <= hi).all() and lo.min() == gmin and hi.max() == gmax and 0.0 <= gmin and gmax < 1.0
        assert np.median(lo) < 0.005 and np.median(hi) > 0.995                    # 1000 uniform values per key
        assert np.array_equal(np.sort(cols[0].data), np.arange(nkeys, dtype=np.float64))          # every key exactly once
        m = 20_000
        host = [batch.column_to_host(j, 0, m) for j in range(batch.ncols)]
        want = oracle.filter_groupby(host, None, [K], [V, V], [oracle.MIN, oracle.MAX], oracle.BYTECODE_COMPILER)
        assert [cols[0].value(i) for i in range(len(want))] == [w[0] for w in want]
        # (the first execution too: the id build's first table fills up -- more than 32 768 keys -- and this form takes over with the
        # widest tables, 256 partitions first, then 512)
        assert ctx.last_form == N.FORM_GROUPBY_HASH_PARTITIONED
    batch.free()
    ctx.close()


@pytest.mark.parametrize("kind", ["dictionary_100k", "double_100k", "double_400k_hash_partitioned"])
def test_group_by_full_size_properties(kind, oracle):
    """1 B rows through the partitioned passes (100 000 dictionary keys), through the dense-id path (100 000 distinct DOUBLE
    keys) and through the hash-partitioned form (400 000 distinct DOUBLE keys): properties that do not need a second engine -- every key is a group, COUNT adds up to the rows the filter keeps (an
    independent filter + COUNT aggregate), SUM adds up to the independent SUM within the reassociation bound, the first
    groups are the keys of the first rows in their order of first appearance (the oracle walks that window row by row)."""
    from queryengine_amd import ColumnExpression, Function, FunctionExpression, NumericLiteralExpression
    from queryengine_amd import engine as E
    from queryengine_amd.workloads import GenColumn
    n, nkeys = 1_000_000_000, (400_000 if kind.startswith("double_400k") else 100_000)
    # "double_100k" keeps the dense-id path (debug bit 16777216 forbids the hash-partitioned form, which would take over at this key count)
    ctx = E.Context(device=0, tuning=[0, 0, 0, 0, 0, 16777216] if kind == "double_100k" else [])
    if kind == "dictionary_100k":
        d = ["k%06d" % i for i in range(nkeys)]
        gen = [GenColumn("k", S, N.GEN_DICT_MOD, 0, modulus=nkeys, dictionary=d), GenColumn("v", D, N.GEN_F64_UNIT, 1)]
        K = ColumnExpression("k", 0, S)
    else:
        gen = [GenColumn("k", D, N.GEN_F64_MOD, 0, modulus=nkeys), GenColumn("v", D, N.GEN_F64_UNIT, 1)]
        K = ColumnExpression("k", 0, D)
    V = ColumnExpression("v", 1, D)
    batch = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in gen], n)
    flt = FunctionExpression(Function.CMP_LT, [V, NumericLiteralExpression(0.25)], B)
    cf = ctx.compile(flt)
    for rep in range(2):   # DOUBLE keys: the first execution finds out that the keys do not fit LDS, the second starts with ids
        res = E.filter_groupby(ctx, batch, cf, [ctx.compile(K)], [ctx.compile(V), ctx.compile(V)], [oracle.SUM, oracle.COUNT])
        cols, ngroups = res.to_columns(), res.count
        res.free()
        assert ngroups == nkeys
        counts, sums = cols[2].data, cols[1].data
        (tot_sum, tot_cnt), nsel = E.filter_aggregate(ctx, batch, cf, [ctx.compile(V), ctx.compile(V)], [oracle.SUM, oracle.COUNT])
        assert int(counts.sum()) == int(tot_cnt) == nsel and abs(nsel / n - 0.25) < 1e-3
        assert abs(float(sums.sum()) - tot_sum) <= 1e-9 * tot_sum
        assert counts.min() > 0.5 * nsel / nkeys and counts.max() < 2 * nsel / nkeys          # uniform keys
        # insertion order: the groups the oracle finds in the first rows come first, in the same order
        m = 20_000
        host = [batch.column_to_host(j, 0, m, dictionary=gen[j].dictionary) for j in range(batch.ncols)]
        want = oracle.filter_groupby(host, flt, [K], [V, V], [oracle.SUM, oracle.COUNT], oracle.BYTECODE_COMPILER)
        head = [cols[0].value(i) for i in range(len(want))]
        assert head == [w[0] for w in want]
        if kind.startswith("double_400k"):
            assert ctx.last_form == N.FORM_GROUPBY_HASH_PARTITIONED    # from the first execution on (more than 32 768 keys)
        elif kind == "double_100k":
            assert ctx.last_form == N.FORM_GROUPBY_HASHED
    batch.free()
    ctx.close()


@pytest.mark.parametrize("path", ["lds", "partitioned", "global_atomics"])
def test_group_by_counter_sharing_with_mixed_nullability(path, oracle):
    """Aggregates over non-nullable inputs share ONE row counter per group, nullable ones keep their own: every order
    and mix of the two kinds must finish to the reference's accumulators (COUNT / AVG / empty => null included)."""
    import itertools
    from queryengine_amd import ColumnExpression, Function, FunctionExpression, NumericLiteralExpression
    from queryengine_amd import engine as E
    rng = np.random.default_rng(31)
    n = 120_001
    nkeys = 12 if path == "lds" else 4000
    ctx = E.Context(device=0, tuning=[0, 0, 0, 0, 0, 256, 0, 0] if path == "global_atomics" else [])
    d = ["k%05d" % i for i in range(nkeys)]
    s = Column(S, rng.integers(0, nkeys, n).astype(np.int32), rng.random(n) > 0.02, d)
    x = Column(D, np.round(rng.normal(0, 50, n)), rng.random(n) > 0.3)          # nullable, integer valued
    y = Column(I64, rng.integers(-100, 100, n))                                # non-nullable
    z = Column(D, np.round(rng.normal(0, 5, n)))                               # non-nullable
    Sx, X, Y, Z = ColumnExpression("s", 0, S), ColumnExpression("x", 1, D), ColumnExpression("y", 2, I64), ColumnExpression("z", 3, D)
    pool = [(X, oracle.SUM), (Y, oracle.COUNT), (X, oracle.COUNT), (Z, oracle.AVG), (Y, oracle.MIN), (X, oracle.MAX), (Z, oracle.SUM)]
    flt = FunctionExpression(Function.CMP_LT, [Z, NumericLiteralExpression(4.0)], B)
    batch = E.DeviceBatch.from_columns(ctx, [s, x, y, z])
    picks = [pool, pool[::-1], pool[1:4], [pool[1]], [pool[2]], pool[3:6]] + [list(p) for p in itertools.islice(itertools.permutations(pool, 3), 7, 60, 13)]
    for sel in picks:
        exprs, aggs = [e for e, _ in sel], [a for _, a in sel]
        res = E.filter_groupby(ctx, batch, ctx.compile(flt), [ctx.compile(Sx)], [ctx.compile(e) for e in exprs], aggs)
        cols = res.to_columns()
        got = [[c.value(i) for c in cols] for i in range(res.count)]
        res.free()
        want = oracle.filter_groupby([s, x, y, z], flt, [Sx], exprs, aggs, oracle.BYTECODE_COMPILER)
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert g[0] == w[0]
            for a, b, f in zip(g[1:], w[1:], aggs):
                if b is None or f != oracle.AVG:
                    assert a == b, (g, w, aggs)
                else:
                    assert abs(a - b) <= 1e-12 * max(1.0, abs(b))
    batch.free()
    ctx.close()


def _rows_equal(got, want, nkeys, aggs, oracle):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        for a, b in zip(g[:nkeys], w[:nkeys]):                 # same groups, same (insertion) order; Double.equals on keys
            if isinstance(b, float):
                assert a is not None and (a == b and np.signbit(a) == np.signbit(b) or (a != a and b != b)), (g, w)
            else:
                assert a == b, (g, w)
        for a, b, fn in zip(g[nkeys:], w[nkeys:], aggs):
            if b is None or fn != oracle.AVG:
                assert a == b or (a != a and b != b), (g, w)
            else:
                assert abs(a - b) <= 1e-12 * max(1.0, abs(b))


@pytest.mark.parametrize("case", ["double_specials", "double_specials_many", "int64_many", "int64_many_global_atomics", "mixed_keys", "grows",
                                  "double_specials_many_hash_partitioned", "int64_many_hash_partitioned", "mixed_keys_hash_partitioned",
                                  "grows_hash_partitioned", "int64_many_hash_partitioned_records", "grows_hash_partitioned_records",
                                  "nocount_hash_partitioned", "onevalue_hash_partitioned", "mixed_keys_hash_partitioned_device"])
def test_group_by_numeric_keys_hashed_matches_oracle(case, oracle):
    """GROUP BY over DOUBLE / INT64 / INT32 keys (GroupByAggregationOperator.kt:33-37 groups on any boxed key tuple;
    Tripdata.kt:27-31 groups by a DOUBLE column): the hashed form.  Key equality is List<Any?>.equals -> Double.equals
    (all NaNs one group, -0.0 and 0.0 two groups), NULL is a key; groups in the reference's insertion order; few keys stay
    in the workgroups' LDS tables; many are resolved to dense ids first (qe_ht_build: a global table that grows when more than
    half full) and aggregated by the dense / partitioned kernels; debug bit 131072 keeps the global-atomic form instead."""
    from queryengine_amd import ColumnExpression, Function, FunctionExpression, NumericLiteralExpression
    from queryengine_amd import engine as E
    rng = np.random.default_rng(77)
    # debug bit 131072 keeps the global-atomic form (the fallback for more than 2^20 keys) instead of dense ids
    # debug bit 8388608 forces the hash-partitioned form (round 3: rows scattered by key hash, one LDS hash table per partition)
    # from the first execution on, with 64 partitions; "grows" (570 k keys) overflows those tables: the execution falls back and
    # the next one runs with more partitions
    # its records live in 128-byte lines of R records {value / key words, row ids, flag bytes}: R = 3 / 4 / 2 for the 4 / 3 / 6 words of
    # these cases; "_records" (debug bit 33554432) keeps the first layout, one {header, words} record padded to a power of two
    records = case.endswith("_records")
    if records:
        case = case[:-len("_records")]
    # results of 4096 groups and more are finished on the device (ordered by first row, accumulators finished, validity bitmaps);
    # "_device" (debug bit 67108864) sends a small result of STRING / DOUBLE / BOOLEAN key tuples that way too
    on_device = case.endswith("_device")
    if on_device:
        case = case[:-len("_device")]
    hp = case.endswith("_hash_partitioned")
    if hp:
        case = case[:-len("_hash_partitioned")]
    ctx = E.Context(device=0, tuning=[0, 0, 0, 0, 0, 131072] if case.endswith("global_atomics") else
                    [0, 0, 0, 0, 0, 8388608 | (33554432 if records else 0) | (67108864 if on_device else 0)] if hp else [])
    I32 = DataType.INT32
    if case.startswith("double_specials"):
        n = 150_001
        pool = np.array([0.0, -0.0, 1.0, 2.5, float("nan"), float("inf"), -float("inf"), 6.0, -1.0,
                         np.frombuffer(np.uint64(0x7ff8000000000123).tobytes(), dtype=np.float64)[0]])   # a NaN with a payload
        if case.endswith("many"):   # the same special values among 5000 ordinary keys: the dense-id path (16-byte entries)
            pool = np.concatenate([pool, rng.normal(0, 1e6, 5000)])
        k = Column(D, pool[rng.integers(0, len(pool), n)], rng.random(n) > 0.03)
        keys = [ColumnExpression("k", 0, D)]
        cols = [k]
    elif case.startswith("int64_many"):
        n = 600_011
        k = Column(I64, rng.integers(-2 ** 62, 2 ** 62, 150_000)[rng.integers(0, 150_000, n)], rng.random(n) > 0.01)
        keys = [ColumnExpression("k", 0, I64)]
        cols = [k]
    elif case == "onevalue":
        n = 500_009                                           # one key word + one value word (6 records per line), a chunk boundary inside
        k = Column(D, np.round(rng.normal(0, 9000, n)))
        keys = [ColumnExpression("k", 0, D)]
        cols = [k]
    elif case == "nocount":
        n = 400_003                                           # ~40 k keys, nothing nullable, nothing that counts: table entries without counter words
        k = Column(D, np.round(rng.normal(0, 12000, n)))
        keys = [ColumnExpression("k", 0, D)]
        cols = [k]
    elif case == "grows":
        n = 900_001                                           # ~570 k distinct keys: the 65 536-entry table grows twice
        k = Column(I32, rng.integers(0, 1_000_000, n).astype(np.int32))
        keys = [ColumnExpression("k", 0, I32)]
        cols = [k]
    else:
        n = 250_003
        d = ["k%02d" % i for i in range(9)]
        s = Column(S, rng.integers(0, len(d), n).astype(np.int32), rng.random(n) > 0.05, d)
        k = Column(D, np.round(rng.normal(0, 3, n)), rng.random(n) > 0.05)
        b = Column(B, rng.random(n) > 0.5)
        keys = [ColumnExpression("s", 0, S), FunctionExpression(Function.MUL, [ColumnExpression("k", 1, D), NumericLiteralExpression(0.5)], D),
                ColumnExpression("b", 2, B)]
        cols = [s, k, b]
    x = Column(D, np.round(rng.normal(0, 100, n)), None if case == "nocount" else rng.random(n) > 0.2)   # integer valued: sums exact in any order
    y = Column(I64, rng.integers(-1000, 1000, n))
    nc = len(cols)
    X, Y = ColumnExpression("x", nc, D), ColumnExpression("y", nc + 1, I64)
    cols = cols + [x, y]
    exprs = [X, X, X, X, X, FunctionExpression(Function.ADD, [Y, Y], I64)]
    aggs = [oracle.SUM, oracle.MIN, oracle.MAX, oracle.COUNT, oracle.AVG, oracle.SUM]
    if case == "nocount":
        exprs, aggs = [X, X, X, FunctionExpression(Function.ADD, [Y, Y], I64)], [oracle.SUM, oracle.MIN, oracle.MAX, oracle.SUM]
    if case == "onevalue":
        exprs, aggs = [X, X, X], [oracle.COUNT, oracle.AVG, oracle.MAX]
    flt = FunctionExpression(Function.CMP_LT, [Y, NumericLiteralExpression(500.0)], B)
    batch = E.DeviceBatch.from_columns(ctx, cols)
    forms = []
    for f in (None, flt):
        for rep in range(3 if hp else 2):                     # the second run starts with the capacity that sufficed
            res = E.filter_groupby(ctx, batch, ctx.compile(f) if f is not None else None,
                                   [ctx.compile(kk) for kk in keys], [ctx.compile(e) for e in exprs], aggs)
            cs = res.to_columns()
            got = [[c.value(i) for c in cs] for i in range(res.count)]
            res.free()
            forms.append(ctx.last_form)
        want = oracle.filter_groupby(cols, f, keys, exprs, aggs, oracle.BYTECODE_COMPILER)
        assert len(want) > (5 if case != "double_specials" else 9)
        _rows_equal(got, want, len(keys), aggs, oracle)
    from queryengine_amd import native as N
    if hp:   # the hash-partitioned form really ran (from the first execution on; "grows" only once its tables are large enough)
        assert forms[-1] == N.FORM_GROUPBY_HASH_PARTITIONED, forms
        # (64 partitions hold 131 072 buckets: the 150 000 / 570 000 keys of "int64_many" / "grows" overflow them once, that
        # execution falls back and reports the key count, the next one sizes its partitions from it)
        assert case in ("grows", "int64_many", "nocount", "onevalue") or all(f_ == N.FORM_GROUPBY_HASH_PARTITIONED for f_ in forms), forms
    else:
        assert all(f_ == N.FORM_GROUPBY_HASHED for f_ in forms), forms
    batch.free()
    ctx.close()


def test_group_by_heavy_hitter_key_leaves_the_hash_partitioned_form(oracle):
    """One key owns 60 % of 5 M rows, 50 000 more share the rest.  The plan's first execution finds many keys (the id build's first
    table fills) and turns to the hash-partitioned form, whose count pass then shows one partition with most of the records: ONE
    workgroup would aggregate it alone, so the form is given up for this plan (the dense-id path slices its partitions).  Same
    groups as the oracle, in its order, on every execution."""
    from queryengine_amd import ColumnExpression
    from queryengine_amd import engine as E
    rng = np.random.default_rng(5)
    n = 5_000_000
    k = np.where(rng.random(n) < 0.6, 0.0, np.floor(rng.random(n) * 50_000.0) + 1.0)
    v = np.round(rng.normal(0, 100, n))
    cols = [Column(D, k), Column(D, v)]
    K, V = ColumnExpression("k", 0, D), ColumnExpression("v", 1, D)
    aggs = [oracle.SUM, oracle.COUNT, oracle.MAX]
    want = oracle.filter_groupby(cols, None, [K], [V, V, V], aggs, oracle.BYTECODE_COMPILER)
    ctx = E.Context(device=0)
    batch = E.DeviceBatch.from_columns(ctx, cols)
    for rep in range(3):
        res = E.filter_groupby(ctx, batch, None, [ctx.compile(K)], [ctx.compile(V), ctx.compile(V), ctx.compile(V)], aggs)
        cs = res.to_columns()
        got = [[c.value(i) for c in cs] for i in range(res.count)]
        res.free()
        assert ctx.last_form == N.FORM_GROUPBY_HASHED
        _rows_equal(got, want, 1, aggs, oracle)
    batch.free()
    ctx.close()


def test_tripdata_query_shape_group_by_double_column(gpu_ctx):
    """Tripdata.kt:27-31: SELECT passenger_count, MIN(fare_amount), MAX(fare_amount) FROM tripdata -- grouping by a DOUBLE
    column, through query() and the planner (implicit GROUP BY, RewriteAggregates.kt:21-47)."""
    from queryengine_amd import ColumnarTable, Field, Schema, TableRegistry
    from queryengine_amd.planner import Mode, query
    rng = np.random.default_rng(3)
    n = 50_000
    pc = rng.integers(0, 7, n).astype(np.float64)
    fare = np.round(rng.uniform(2.5, 80.0, n), 2)
    t = ColumnarTable(Schema([Field("passenger_count", D), Field("fare_amount", D)]), [Column(D, pc), Column(D, fare)])
    reg = TableRegistry()
    reg.register("tripdata", t)
    rows = query(reg, "SELECT passenger_count, MIN(fare_amount), MAX(fare_amount) FROM tripdata", Mode.GPU_FUSED, ctx=gpu_ctx)
    seen = []
    for v in pc:
        if v not in seen:
            seen.append(v)
    assert [r[0] for r in rows] == seen                        # LinkedHashMap insertion order
    for key, lo, hi in rows:
        sel = fare[pc == key]
        assert lo == sel.min() and hi == sel.max()


@pytest.mark.parametrize("keytype", ["double", "int64", "string", "boolean", "int32"])
def test_order_by_on_the_device_matches_compare_values(gpu_ctx, keytype):
    """qe_result_order_by: OrderByOperator.kt:9-12 -- stable sortBy { row[index] as Comparable } = compareValues: null first,
    Double.compareTo (-0.0 < 0.0, NaN last), String.compareTo (UTF-16 code units), false < true; every other column
    follows its row; ties keep their input order (200 k rows, many ties, nullable key and payload)."""
    from queryengine_amd import ColumnExpression
    from queryengine_amd import engine as E
    from queryengine_amd.operators import _compare_key
    rng = np.random.default_rng(9)
    n = 200_003
    if keytype == "double":
        pool = np.array([0.0, -0.0, 1.5, -1.5, float("nan"), float("inf"), -float("inf"), 1e300, -1e-300, 3.0])
        key = Column(D, np.where(rng.random(n) < 0.5, pool[rng.integers(0, len(pool), n)], rng.normal(0, 1e3, n)), rng.random(n) > 0.05)
    elif keytype == "int64":
        pool = np.array([0, -1, 1, 2 ** 63 - 1, -(2 ** 63), 2 ** 53 + 1, 2 ** 63 - 2], dtype=np.int64)
        key = Column(I64, np.where(rng.random(n) < 0.3, pool[rng.integers(0, len(pool), n)], rng.integers(-50, 50, n)), rng.random(n) > 0.05)
    elif keytype == "int32":
        key = Column(I32, rng.integers(-2 ** 31, 2 ** 31 - 1, n).astype(np.int32) // (1 << 20), rng.random(n) > 0.05)
    elif keytype == "string":
        d = ["b", "a", "", "B", "\uff5e", "\U0001F600", "aa", "Z\u00fc", "zz", "a "]      # (the C ABI's strings are NUL terminated)
        key = Column(S, rng.integers(0, len(d), n).astype(np.int32), rng.random(n) > 0.05, d)
    else:
        key = Column(B, rng.random(n) > 0.5, rng.random(n) > 0.1)
    rowid = Column(I64, np.arange(n, dtype=np.int64))
    flag = Column(B, rng.random(n) > 0.3, rng.random(n) > 0.2)
    batch = E.DeviceBatch.from_columns(gpu_ctx, [key, rowid, flag])
    projs = [gpu_ctx.compile(ColumnExpression("k", 0, key.type)), gpu_ctx.compile(ColumnExpression("r", 1, I64)),
             gpu_ctx.compile(ColumnExpression("f", 2, B))]
    res = E.filter_project(gpu_ctx, batch, None, projs)
    srt = gpu_ctx.order_by(res, 0)
    k, r, f = srt.to_columns()
    want = sorted(range(n), key=lambda i: _compare_key(key.value(i)))          # stable, like java.util.List.sort
    assert np.array_equal(r.data, np.array(want, dtype=np.int64))
    for j in (0, 1, n // 3, n // 2, n - 2, n - 1):
        i = want[j]
        a, b = k.value(j), key.value(i)
        assert (a == b) or (a != a and b != b)
        assert f.value(j) == flag.value(i)
    res.free(); srt.free(); batch.free()


def test_order_by_query_sorts_on_the_device(gpu_ctx):
    """Main.kt:35-50 shape through query(): the OrderByOperator above a GPU operator sorts its result in HBM."""
    from queryengine_amd.planner import Mode, query
    rng = np.random.default_rng(4)
    n = 30_000
    a = rng.integers(0, 1000, n).astype(np.float64)
    c = rng.random(n)
    t = ColumnarTable(Schema([Field("a", D), Field("c", D)]), [Column(D, a), Column(D, c, rng.random(n) > 0.1)])
    reg = TableRegistry()
    reg.register("t", t)
    rows = query(reg, "SELECT a + 1, c FROM t WHERE a < 500 ORDER BY 2", Mode.GPU_FUSED, ctx=gpu_ctx)
    keep = a < 500
    cv = t.columns[1]
    exp = [[a[i] + 1, cv.value(i)] for i in np.nonzero(keep)[0]]
    from queryengine_amd.operators import _compare_key
    exp.sort(key=lambda r: _compare_key(r[1]))
    assert rows == exp and rows[0][1] is None


def test_result_to_host_pinned_path_returns_the_same_bytes(gpu_ctx):
    """qe_result_to_host (pinned staging owned by the library, copy stream, asynchronous start) hands out the same bytes as
    the column-by-column copy into caller buffers -- nullable INT64 / DOUBLE, BOOLEAN bitmaps, dictionary codes, an empty
    result -- also when the next scan runs while the copy is in flight, when the device result is freed before the wait,
    and when the pinned buffers are recycled."""
    from queryengine_amd import engine as E
    from queryengine_amd import workloads as W
    from helpers import D, I64, S, Fn, assert_columns_equal, col, fn, num
    ctx = gpu_ctx
    wl = W.config2(3_000_011, null_pct=1)
    projs = list(wl.projections) + [fn(Fn.CMP_LT, col("c", 2, D), num(0.25)), fn(Fn.CMP_GT, col("b", 1, I64), num(7))]
    batch = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], wl.default_rows)
    cf, cp = ctx.compile(wl.filter), [ctx.compile(p) for p in projs]
    for rnd in range(3):                                  # round 2 and 3 run on recycled pinned buffers
        res = E.filter_project(ctx, batch, cf, cp)
        want = res.to_columns()
        host = res.to_host()                              # copies started, not waited for
        other = E.filter_project(ctx, batch, cf, cp)      # the next scan beside the copy
        assert other.count == res.count
        other.free()
        if rnd == 1:
            res.free()                                    # waits for the copy that still reads it
        host.wait()
        assert host.count == len(want[0]) and host.ncols == len(want)
        for i, w in enumerate(want):
            assert_columns_equal(host.column(i), w, f"round {rnd} column {i}")
        data, valid = host.column_views(0)                # zero-copy views of the pinned memory
        assert data.dtype == np.int64 and len(data) == host.count and valid is not None
        host.free()
        res.free()
    # dictionary codes keep their dictionary; an empty result works
    wl4 = W.config4(200_000, nkeys=10, key="k0004")
    b4 = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl4.columns], wl4.default_rows)
    r4 = E.filter_project(ctx, b4, ctx.compile(wl4.filter), [ctx.compile(p) for p in wl4.projections])
    h4 = r4.to_host().wait()
    for i, w in enumerate(r4.to_columns()):
        assert_columns_equal(h4.column(i), w, f"cfg 4 column {i}")
    h4.free(); r4.free()
    r0 = E.filter_project(ctx, b4, ctx.compile(fn(Fn.CMP_EQ, col("s", 0, S), __import__("queryengine_amd").StringLiteralExpression("absent"))),
                          [ctx.compile(p) for p in wl4.projections])
    h0 = r0.to_host().wait()
    assert h0.count == 0 and len(h0.column(1).data) == 0
    h0.free(); r0.free(); b4.free(); batch.free()

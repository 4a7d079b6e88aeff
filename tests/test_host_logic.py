"""Host-side logic that needs no GPU: SQL front end, type check, plan construction, serialisation,
program verification (through the C ABI on a planning-only context), kernel generation + hiprtc."""
import struct

import numpy as np
import pytest

from queryengine_amd import (BooleanLiteralExpression, Column, ColumnExpression, ColumnarTable, DataType, Field, Function,
                             FunctionExpression, IdentifierExpression, NumericLiteralExpression, Schema,
                             StringLiteralExpression, TableRegistry, AggregationFunctionExpression, AggregationFunction)
from queryengine_amd import engine as E
from queryengine_amd import native as N
from queryengine_amd.planner import (LogicalAggregationNode, LogicalFilterNode, LogicalOrderByNode, LogicalProjectionNode,
                                     LogicalScanNode,
                                     SchemaException, buildLogicalPlan)
from queryengine_amd.program import serialize
from queryengine_amd.sql import SyntaxException, parseExpression, parseQuery
from queryengine_amd.typecheck import TypeCheckException, typeCheck

D, I64, I32, B, S = DataType.DOUBLE, DataType.INT64, DataType.INT32, DataType.BOOLEAN, DataType.STRING
Fn = Function


# ---- parser: the vectors of T/parser/ParserTest.kt:8-50 ------------------------------------------------
def test_parser_reference_vectors():
    assert parseExpression("foo") == IdentifierExpression("foo")
    assert parseExpression("123") == NumericLiteralExpression(123.0)
    assert parseExpression("-foo") == FunctionExpression(Fn.UNARY_MINUS, [IdentifierExpression("foo")])
    assert parseExpression("IF foo THEN 1 ELSE 2 END") == FunctionExpression(
        Fn.IF, [IdentifierExpression("foo"), NumericLiteralExpression(1.0), NumericLiteralExpression(2.0)])
    assert parseExpression("SUM(foo)") == AggregationFunctionExpression(AggregationFunction.SUM, [IdentifierExpression("foo")])
    q = parseQuery("SELECT foo, bar FROM baz WHERE foo > 1 ORDER BY 1")
    assert q.select == (IdentifierExpression("foo"), IdentifierExpression("bar")) and q.from_ == "baz"
    assert q.filter == FunctionExpression(Fn.CMP_GT, [IdentifierExpression("foo"), NumericLiteralExpression(1.0)])
    assert q.orderByColumn == 1


def test_parser_precedence_follows_grammar_alternative_order():
    """Query.g4:27-40: unary > mul > add > compare > AND > OR; NOT binds tighter than comparison."""
    a, b, c = (IdentifierExpression(x) for x in "abc")
    assert parseExpression("a + b * c") == FunctionExpression(Fn.ADD, [a, FunctionExpression(Fn.MUL, [b, c])])
    assert parseExpression("a - b - c") == FunctionExpression(Fn.SUB, [FunctionExpression(Fn.SUB, [a, b]), c])
    assert parseExpression("a < b AND b < c OR a = c") == FunctionExpression(Fn.OR, [
        FunctionExpression(Fn.AND, [FunctionExpression(Fn.CMP_LT, [a, b]), FunctionExpression(Fn.CMP_LT, [b, c])]),
        FunctionExpression(Fn.CMP_EQ, [a, c])])
    assert parseExpression("NOT a < b") == FunctionExpression(Fn.CMP_LT, [FunctionExpression(Fn.NOT, [a]), b])
    assert parseExpression("-1.5") == NumericLiteralExpression(-1.5)      # ExpressionAstBuilder.kt:104-110
    assert parseExpression("select") if False else True
    assert parseExpression("'it''s'") == StringLiteralExpression("it's")
    assert parseExpression('"my col" <> 2') == FunctionExpression(Fn.CMP_NE, [IdentifierExpression("my col"), NumericLiteralExpression(2.0)])
    assert parseExpression("TrUe") == BooleanLiteralExpression(True)
    with pytest.raises(SyntaxException):
        parseExpression("a +")
    with pytest.raises(SyntaxException):
        parseExpression("a ? b")


# ---- type check -------------------------------------------------------------------------------------------
def test_typecheck_rules():
    a, i, j, p, s = (ColumnExpression("a", 0, D), ColumnExpression("i", 1, I64), ColumnExpression("j", 2, I32),
                     ColumnExpression("p", 3, B), ColumnExpression("s", 4, S))
    assert typeCheck(FunctionExpression(Fn.ADD, [a, a])).dataType == D
    assert typeCheck(FunctionExpression(Fn.ADD, [i, j])).dataType == I64
    assert typeCheck(FunctionExpression(Fn.MUL, [j, j])).dataType == I32
    assert typeCheck(FunctionExpression(Fn.SUB, [i, NumericLiteralExpression(1.0)])).dataType == D
    assert typeCheck(FunctionExpression(Fn.UNARY_MINUS, [i])).dataType == I64
    # the reference rejects `bool AND bool` (TypeCheck.kt:79-85, a bug); the intended rule is implemented
    assert typeCheck(FunctionExpression(Fn.AND, [p, FunctionExpression(Fn.CMP_LT, [a, i])])).dataType == B
    assert typeCheck(FunctionExpression(Fn.IF, [p, i, a])).dataType == D
    assert typeCheck(FunctionExpression(Fn.CMP_EQ, [s, StringLiteralExpression("x")])).dataType == B
    for bad in (FunctionExpression(Fn.ADD, [a, p]), FunctionExpression(Fn.AND, [a, p]), FunctionExpression(Fn.NOT, [a]),
                FunctionExpression(Fn.CMP_LT, [s, s]), FunctionExpression(Fn.CMP_EQ, [s, a]),
                FunctionExpression(Fn.IF, [a, a, a]), FunctionExpression(Fn.IF, [p, a, s])):
        with pytest.raises(TypeCheckException):
            typeCheck(bad)


# ---- logical plans ---------------------------------------------------------------------------------------------
def _registry():
    t = ColumnarTable.from_rows(Schema([Field("a", I64), Field("b", I64), Field("c", D), Field("s", S)]),
                                [[1, 2, 0.25, "x"], [200, 3, 0.75, None]])
    r = TableRegistry()
    r.register("t", t)
    return r


def test_logical_plan_shapes_and_column_slot_order():
    r = _registry()
    plan = buildLogicalPlan(r, parseQuery("SELECT c * 2.0, a + b FROM t WHERE a < 100 AND c < 0.5"))
    assert isinstance(plan, LogicalProjectionNode) and isinstance(plan.source, LogicalFilterNode)
    scan = plan.source.source
    # ResolveSchema.kt:53-63 + :24-33: slots in order of first use, SELECT list before WHERE
    assert [f.name for f in scan.schema.fields] == ["c", "a", "b"]
    assert plan.expressions[0].operands[0] == ColumnExpression("c", 0, D)
    assert plan.source.filter.dataType == B
    with pytest.raises(SchemaException):
        buildLogicalPlan(r, parseQuery("SELECT nope FROM t"))
    assert isinstance(buildLogicalPlan(r, parseQuery("SELECT a, a FROM t")), LogicalProjectionNode)
    # distinct plain columns are an identity projection over the pruned scan: removed (Optimizer.kt:33-35)
    ident = buildLogicalPlan(r, parseQuery("SELECT b, a FROM t"))
    assert isinstance(ident, LogicalScanNode) and [f.name for f in ident.schema.fields] == ["b", "a"]
    fin = buildLogicalPlan(r, parseQuery("SELECT SUM(a + 10*b), COUNT(c) FROM t WHERE c < 0.5"))
    agg = fin.source
    assert isinstance(fin, LogicalProjectionNode) and isinstance(agg, LogicalAggregationNode) and agg.groupCount == 0
    assert agg.aggregateFunctions == (AggregationFunction.SUM, AggregationFunction.COUNT)
    grouped = buildLogicalPlan(r, parseQuery("SELECT s, SUM(a) FROM t"))       # implicit GROUP BY s
    assert grouped.source.groupCount == 1 and isinstance(grouped.source.source.source, LogicalScanNode)
    ordered = buildLogicalPlan(r, parseQuery("SELECT a + b, c FROM t WHERE a < 100 ORDER BY 2"))   # Planner.kt:13
    assert isinstance(ordered, LogicalOrderByNode) and ordered.index == 2
    assert isinstance(ordered.source, LogicalProjectionNode) and isinstance(ordered.source.source, LogicalFilterNode)
    ordered = buildLogicalPlan(r, parseQuery("SELECT s, SUM(a) FROM t ORDER BY 1"))    # RewriteAggregates.kt:50-53
    assert isinstance(ordered.source.source, LogicalAggregationNode)


def test_order_by_operator_follows_compare_values():
    """OrderByOperator.kt:9-12: stable sortBy { row[index] as Comparable } = compareValues: null first, then
    Double.compareTo (-0.0 < 0.0, NaN last), String.compareTo (UTF-16 code units), Boolean false < true."""
    from queryengine_amd.operators import Operator, OrderByOperator, map as op_map

    class Rows(Operator):
        def __init__(self, rows):
            self.rows, self.i = rows, None

        def open(self):
            self.i = 0

        def close(self):
            self.i = None

        def next(self):
            if self.i >= len(self.rows):
                return None
            self.i += 1
            return self.rows[self.i - 1]

    nan = float("nan")
    rows = [[1.0, "a"], [nan, "b"], [None, "c"], [-0.0, "d"], [0.0, "e"], [float("inf"), "f"], [None, "g"], [-1.0, "h"], [0.0, "i"]]
    op = OrderByOperator(Rows(rows), 0)
    assert [r[1] for r in op_map(op, lambda r: r)] == ["c", "g", "h", "d", "e", "i", "a", "f", "b"]
    assert [r[1] for r in op_map(op, lambda r: r)] == ["c", "g", "h", "d", "e", "i", "a", "f", "b"]   # re-openable
    with pytest.raises(RuntimeError):
        op.next()
    # U+FF5E (one code unit 0xFF5E) sorts AFTER U+1F600 (surrogates 0xD83D 0xDE00) in UTF-16 order, before it by code point
    strs = [["\uff5e"], ["\U0001F600"], ["B"], [None], ["a"], [""]]
    assert op_map(OrderByOperator(Rows(strs), 0), lambda r: r[0]) == [None, "", "B", "a", "\U0001F600", "\uff5e"]
    bools = [[True], [None], [False]]
    assert op_map(OrderByOperator(Rows(bools), 0), lambda r: r[0]) == [None, False, True]


# ---- serialisation + verification through the C ABI (planning-only context: no GPU needed) ------------------------
@pytest.fixture(scope="module")
def plan_ctx(native_lib, tmp_path_factory):
    ctx = E.Context(device=None, jit_cache_dir=str(tmp_path_factory.mktemp("jit")))
    yield ctx
    ctx.close()


def test_program_encoding_is_postfix():
    e = FunctionExpression(Fn.CMP_LT, [ColumnExpression("a", 3, I64), NumericLiteralExpression(100.0)], B)
    b = serialize(e)
    assert b[:4] == b"QEX\x01"
    assert b[4:8] == struct.pack("<BBH", 1, int(I64), 3)
    assert b[8:17] == struct.pack("<Bd", 2, 100.0)
    assert b[17:] == struct.pack("<BBB", 16, Fn.CMP_LT.ordinal, int(B))


def test_expression_verifier(plan_ctx):
    a, p = ColumnExpression("a", 0, D), ColumnExpression("p", 1, B)
    assert plan_ctx.compile(FunctionExpression(Fn.ADD, [a, a], D)).result_type == D
    assert plan_ctx.compile(FunctionExpression(Fn.ADD, [a, ColumnExpression("i", 2, I64)])).result_type == D   # inferred
    bad_programs = [
        b"",                                              # no header
        b"QEX\x02",                                       # wrong version
        b"QEX\x01",                                       # leaves nothing
        b"QEX\x01" + struct.pack("<BBB", 16, 9, 1),       # ADD on an empty stack (underflow)
        serialize(a) + serialize(a)[4:],                  # leaves two values
        b"QEX\x01" + struct.pack("<BBH", 1, 9, 0),        # bad column type
        serialize(a)[:-1],                                # truncated
        b"QEX\x01" + b"\x63",                             # unknown opcode
    ]
    for prog in bad_programs:
        h = __import__("ctypes").c_void_p()
        st = plan_ctx._lib.qe_expr_compile(plan_ctx.handle, prog, len(prog), __import__("ctypes").byref(h))
        assert st in (1, 2), prog
    for bad in (FunctionExpression(Fn.ADD, [a, p], D), FunctionExpression(Fn.AND, [a, p], B),
                FunctionExpression(Fn.ADD, [a, a], B)):   # declared type does not match inferred
        with pytest.raises(N.QeError) as ei:
            plan_ctx.compile(bad)
        assert ei.value.code == 2


def test_planning_context_generates_and_compiles_kernels_without_gpu(plan_ctx):
    from queryengine_amd import workloads as W
    from queryengine_amd import prepared
    for wl in (W.config1(), W.config2(null_pct=1), W.config3(), W.config4()):
        batch = E.DeviceBatch.describe(plan_ctx, prepared._schema_columns(wl))
        cf = plan_ctx.compile(wl.filter)
        cp = [plan_ctx.compile(p) for p in wl.projections]
        src = E.generated_source(plan_ctx, batch, cf, cp)
        assert "qe_fused" in src and "__builtin_nontemporal_load" in src and "qe_lookback" in src
        E.prepare(plan_ctx, batch, cf, cp)                 # hiprtc cross-compiles for gfx950 with no device
        if all(p.dataType.is_numeric for p in wl.projections):
            E.prepare_aggregate(plan_ctx, batch, cf, cp, [N.AGG_SUM] * len(cp))
            E.prepare_aggregate(plan_ctx, batch, None, cp[:1], [N.AGG_MIN])
    # but nothing can EXECUTE without a device: there is no CPU fallback
    with pytest.raises(N.QeError) as ei:
        E.filter_project(plan_ctx, batch, cf, cp)
    assert ei.value.code == 3
    with pytest.raises(N.QeError):
        E.DeviceBatch.from_columns(plan_ctx, [Column(D, np.zeros(4))])


def test_generated_source_specialises_null_handling(plan_ctx):
    a_nn = Column(D, np.zeros(4))
    a_n = Column(D, np.zeros(4), np.array([True, False, True, True]))
    e = FunctionExpression(Fn.CMP_LT, [ColumnExpression("a", 0, D), NumericLiteralExpression(1.0)], B)
    proj = [plan_ctx.compile(ColumnExpression("a", 0, D))]
    src_nn = E.generated_source(plan_ctx, E.DeviceBatch.describe(plan_ctx, [a_nn]), plan_ctx.compile(e), proj)
    src_n = E.generated_source(plan_ctx, E.DeviceBatch.describe(plan_ctx, [a_n]), plan_ctx.compile(e), proj)
    assert "kc0" not in src_nn and "kc0" in src_n and "stagevalid" in src_n


def test_column_type_mismatch_is_rejected(plan_ctx):
    batch = E.DeviceBatch.describe(plan_ctx, [Column(I64, np.zeros(4, dtype=np.int64))])
    with pytest.raises(N.QeError) as ei:
        E.generated_source(plan_ctx, batch, None, [plan_ctx.compile(ColumnExpression("a", 0, D))])
    assert ei.value.code == 2
    with pytest.raises(N.QeError):
        E.generated_source(plan_ctx, batch, None, [plan_ctx.compile(ColumnExpression("a", 5, I64))])


def test_rewrite_aggregates_reference_fixtures():
    """The three plan-shape fixtures of T/evaluator/RewriteAggregatesTest.kt:13-100."""
    from queryengine_amd.planner import InvalidAggregatesException, rewriteAggregates
    AF, AFE, CE = AggregationFunction, AggregationFunctionExpression, ColumnExpression
    scan = LogicalScanNode("table", Schema([Field("foo", D)]))
    got = rewriteAggregates(LogicalProjectionNode(scan, (AFE(AF.SUM, [CE("foo", 0, D)], D),)))
    assert got == LogicalProjectionNode(
        LogicalAggregationNode(LogicalProjectionNode(scan, (CE("foo", 0, D),)), 0, (AF.SUM,)), (CE("SUM", 0, D),))
    scan2 = LogicalScanNode("table", Schema([Field("foo", D), Field("bar", D)]))
    got = rewriteAggregates(LogicalProjectionNode(scan2, (AFE(AF.SUM, [CE("foo", 0, D)], D), AFE(AF.COUNT, [CE("bar", 1, D)], D),
                                                          AFE(AF.AVG, [CE("foo", 0, D)], D))))
    assert got == LogicalProjectionNode(
        LogicalAggregationNode(LogicalProjectionNode(scan2, (CE("foo", 0, D), CE("bar", 1, D), CE("foo", 0, D))), 0,
                               (AF.SUM, AF.COUNT, AF.AVG)),
        (CE("SUM", 0, D), CE("COUNT", 1, D), CE("AVG", 2, D)))
    div = FunctionExpression(Fn.DIV, [AFE(AF.COUNT, [CE("bar", 1, D)], D), AFE(AF.COUNT, [CE("foo", 0, D)], D)], D)
    got = rewriteAggregates(LogicalProjectionNode(scan2, (div,)))
    assert got == LogicalProjectionNode(
        LogicalAggregationNode(LogicalProjectionNode(scan2, (CE("bar", 1, D), CE("foo", 0, D))), 0, (AF.COUNT, AF.COUNT)),
        (FunctionExpression(Fn.DIV, [CE("COUNT", 0, D), CE("COUNT", 1, D)], D),))
    assert rewriteAggregates(LogicalProjectionNode(scan2, (CE("foo", 0, D),))) is None
    with pytest.raises(InvalidAggregatesException):
        rewriteAggregates(LogicalProjectionNode(scan2, (AFE(AF.SUM, [AFE(AF.SUM, [CE("foo", 0, D)], D)], D),)))


def test_late_materialisation_splits_the_filter_into_conjunct_stages(native_lib, tmp_path):
    """Generated source (planning-only context, no GPU): the filter's AND chain becomes qe_conj0 / qe_conj1, the
    projection-only column is loaded under the live-row guard; tuning[5] bit 2048 restores the load-everything form."""
    from queryengine_amd import workloads as W
    from queryengine_amd.prepared import _schema_columns
    wl = W.config2(1000)
    for tuning, staged in (([], True), ([0, 0, 0, 0, 0, 2048, 0, 0], False)):
        ctx = E.Context(device=None, jit_cache_dir=str(tmp_path / ("s" if staged else "u")), tuning=tuning)
        batch = E.DeviceBatch.describe(ctx, _schema_columns(wl))
        src = E.generated_source(ctx, batch, ctx.compile(wl.filter), [ctx.compile(p) for p in wl.projections])
        assert ("qe_conj0(" in src and "qe_conj1(" in src) == staged
        assert ("if (keep[u][0] || keep[u][1]) {" in src) == staged
        assert "qe_fp_count" in src and "qe_fp_write" in src       # the two-pass form lives in the same module
        ctx.close()


def test_plan_cache_keys_on_dictionary_identity_not_address(native_lib, tmp_path):
    """A C-ABI caller frees a batch and its dictionary, then creates a new dictionary: the allocator hands back the same
    address.  The fused-plan cache must not mistake it for the old one (round-1 bug: `s = 'b'` compiled to the OLD
    dictionary's code in half of the iterations).  Planning-only context, raw C ABI (the Python layer would keep every
    dictionary alive and hide it)."""
    import ctypes as C
    L = native_lib
    ctx = E.Context(device=None, jit_cache_dir=str(tmp_path))
    s_eq_b = ctx.compile(FunctionExpression(Fn.CMP_EQ, [ColumnExpression("s", 0, S), StringLiteralExpression("b")], B))
    proj = ctx.compile(ColumnExpression("s", 0, S))
    seen = set()
    for it in range(60):
        entries = [b"a", b"b"] if it % 2 == 0 else [b"b", b"a"]
        arr = (C.c_char_p * 2)(*entries)
        d = C.c_void_p()
        N.check(ctx.handle, L.qe_dict_create(ctx.handle, 2, arr, C.byref(d)))
        seen.add(d.value)
        desc = (N.ColDesc * 1)()
        desc[0].type = int(S)
        desc[0].dict = d
        bh = C.c_void_p()
        N.check(ctx.handle, L.qe_batch_describe(ctx.handle, 128, 1, desc, C.byref(bh)))
        out = C.c_char_p()
        projs = (C.c_void_p * 1)(proj.handle)
        N.check(ctx.handle, L.qe_filter_project_source(ctx.handle, bh, s_eq_b.handle, projs, 1, C.byref(out)))
        src = out.value.decode()
        want = 1 if it % 2 == 0 else 0
        assert f"(c0 == {want})" in src, f"iteration {it}: literal 'b' must compile to code {want}"
        L.qe_batch_free(ctx.handle, bh)
        L.qe_dict_free(ctx.handle, d)
    ctx.close()


def test_int_compare_shortcut_stops_below_2_53(plan_ctx):
    """Generated source: (double)int64_col OP integral literal is compared on the integers only for |L| < 2^53; at
    L = +-2^53 (where 2^53 and 2^53+1 convert to the same double) the widened compare stays."""
    x = Column(I64, np.zeros(4, dtype=np.int64))
    batch = E.DeviceBatch.describe(plan_ctx, [x])
    X = ColumnExpression("x", 0, I64)

    def src(lit):
        return E.generated_source(plan_ctx, batch, plan_ctx.compile(FunctionExpression(Fn.CMP_LE, [X, NumericLiteralExpression(lit)])),
                                  [plan_ctx.compile(X)])
    assert "(c0 <= 9007199254740991ll)" in src(2.0 ** 53 - 1)
    assert "(c0 <= -9007199254740991ll)" in src(-(2.0 ** 53 - 1))
    for lit in (2.0 ** 53, -(2.0 ** 53), 2.0 ** 60):
        text = src(lit)
        assert "ll)" not in text.split("qe_conj0")[1].split("}")[0], lit     # no integer literal compare in the conjunct
        assert "((double)c0)" in text


def test_dense_single_pass_kernel_is_generated_and_compiles(native_lib, tmp_path):
    """The dense form (plans that keep a large share of their rows): a workgroup per tile of QE_WAVES sub-tiles, the kept rows of a
    tile parked in the workgroup's LDS tile while the next tile is loaded, resolved one tile later and moved with coalesced
    stores.  Generated
    and hiprtc-compiled for the BASELINE plans on a planning-only context (no GPU)."""
    from queryengine_amd import workloads as W
    from queryengine_amd.prepared import _schema_columns
    ctx = E.Context(device=None, jit_cache_dir=str(tmp_path), tuning=[0, 0, 0, 0, 0, 16384, 0, 0])
    for wl in (W.config2(1000), W.config2(1000, null_pct=1), W.config3(1000), W.config4(1000)):
        batch = E.DeviceBatch.describe(ctx, _schema_columns(wl))
        cf, cp = ctx.compile(wl.filter), [ctx.compile(p) for p in wl.projections]
        src = E.generated_source(ctx, batch, cf, cp)
        assert "s_ticket" in src and "qe_dense_load<true>" in src and "qe_lookback_wait(p, ptile, lb, lane)" in src and "qe_dense_park(" in src
        assert "qe_conj0(" not in src and "QE_SUBS_PER_CHUNK 1u" in src       # every load up front, chunk == sub-tile
        E.prepare(ctx, batch, cf, cp)      # compiles the default, the wide AND the dense kernel
    ctx.close()


def test_group_by_plans_generate_and_compile_without_gpu(native_lib, tmp_path):
    """Every group-by form is generated and hiprtc-compiled on a planning-only context: the LDS-privatised table (incl. the
    1024-thread variant for tables up to 144 KiB), the partitioned passes (4-wave tiles, 8-wave tiles past 128 partitions,
    several record values, nullable aggregate inputs) and the hashed form with its id build pass (one key: 16-byte entries;
    several keys).  A compile error in any generated kernel fails here, before a GPU is involved."""
    from queryengine_amd import ColumnExpression, Function, FunctionExpression, NumericLiteralExpression
    from queryengine_amd import Column, DataType
    ctx = E.Context(device=None, jit_cache_dir=str(tmp_path))
    S, D, I64, B = DataType.STRING, DataType.DOUBLE, DataType.INT64, DataType.BOOLEAN
    for nkeys in (10, 2500, 100_000, 1_000_000):
        d = ["k%07d" % i for i in range(nkeys)]
        cols = [Column(S, np.zeros(2, dtype=np.int32), None, d), Column(D, np.zeros(2), np.array([True, False])), Column(I64, np.zeros(2, dtype=np.int64))]
        batch = E.DeviceBatch.describe(ctx, cols)
        K, X, Y = ColumnExpression("k", 0, S), ColumnExpression("x", 1, D), ColumnExpression("y", 2, I64)
        flt = FunctionExpression(Function.CMP_LT, [Y, NumericLiteralExpression(5.0)], B)
        for f, exprs, aggs in ((None, [Y, Y], [N.AGG_SUM, N.AGG_COUNT]), (flt, [X, Y, X], [N.AGG_SUM, N.AGG_MAX, N.AGG_AVG])):
            E.prepare_groupby(ctx, batch, ctx.compile(f) if f is not None else None, [ctx.compile(K)], [ctx.compile(e) for e in exprs], aggs)
    cols = [Column(D, np.zeros(2), np.array([True, False])), Column(I64, np.zeros(2, dtype=np.int64)), Column(D, np.zeros(2))]
    batch = E.DeviceBatch.describe(ctx, cols)
    K1, K2, V = ColumnExpression("a", 0, D), ColumnExpression("b", 1, I64), ColumnExpression("v", 2, D)
    E.prepare_groupby(ctx, batch, None, [ctx.compile(K1)], [ctx.compile(V), ctx.compile(V)], [N.AGG_MIN, N.AGG_MAX])
    E.prepare_groupby(ctx, batch, None, [ctx.compile(K1), ctx.compile(K2)], [ctx.compile(V)], [N.AGG_SUM])
    ctx.close()

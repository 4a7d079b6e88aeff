"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bar (BASELINE.json north_star): bit-exact for integer / bitmap / index results; the f64 results are
also required to be bit-exact here (0 ULP, tighter than the 1 ULP allowed) because the generated
kernels are compiled with -ffp-contract=off.
"""
import os
import random

import numpy as np
import pytest

from queryengine_amd import (BooleanLiteralExpression, Column, DataType, Function, NumericLiteralExpression,
                             StringLiteralExpression)
from queryengine_amd import engine as E
from queryengine_amd import native as N

from helpers import B, D, I32, I64, S, ExprGen, Fn, assert_columns_equal, col, fn, num, random_column

pytestmark = pytest.mark.gpu


def run_both(ctx, oracle, cols, flt, projs, mode=None):
    mode = oracle.BYTECODE_COMPILER if mode is None else mode
    batch = E.DeviceBatch.from_columns(ctx, cols)
    cf = ctx.compile(flt) if flt is not None else None
    cp = [ctx.compile(p) for p in projs]
    res = E.filter_project(ctx, batch, cf, cp)
    got = res.to_columns()
    want = oracle.filter_project(cols, flt, projs, mode)
    assert res.count == (len(want[0]) if want else res.count)
    for i, (g, w) in enumerate(zip(got, want)):
        assert_columns_equal(g, w, f"projection {i}")
    res.free()
    batch.free()
    return got


def test_config1_a_plus_b_where_a_lt_100(any_ctx, oracle):
    """BASELINE config 1: SELECT a + b FROM t WHERE a < 100 (reference types: DOUBLE)."""
    rng = np.random.default_rng(1)
    n = 100_000
    a = Column(D, rng.integers(0, 1000, n).astype(np.float64))
    b = Column(D, rng.integers(0, 2 ** 31, n).astype(np.float64))
    got = run_both(any_ctx, oracle, [a, b], fn(Fn.CMP_LT, col("a", 0, D), num(100)),
                   [fn(Fn.ADD, col("a", 0, D), col("b", 1, D))])
    assert 0.05 * n < len(got[0]) < 0.15 * n


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 127, 128, 129, 511, 512, 513, 16383, 16384, 16385, 70001, 200_000])
def test_ragged_sizes(any_ctx, oracle, n):
    """empty / ragged inputs around the 64-row bitmap word and the tile size"""
    rng = np.random.default_rng(n)
    a = random_column(rng, I64, n, special=False)
    c = random_column(rng, D, n, null_frac=0.2 if n else 0.0)
    flt = fn(Fn.AND, fn(Fn.CMP_LT, col("a", 0, I64), num(100)), fn(Fn.CMP_LT, col("c", 1, D), num(0.5)))
    run_both(any_ctx, oracle, [a, c], flt, [fn(Fn.ADD, col("a", 0, I64), col("a", 0, I64)),
                                            fn(Fn.MUL, col("c", 1, D), num(2.0))])


@pytest.mark.parametrize("sel", [0.0, 0.01, 0.5, 1.0])
def test_selectivity_extremes(any_ctx, oracle, sel):
    rng = np.random.default_rng(7)
    n = 50_000
    a = Column(D, rng.random(n))
    b = random_column(rng, I64, n)
    run_both(any_ctx, oracle, [a, b], fn(Fn.CMP_LT, col("a", 0, D), num(sel)),
             [col("b", 1, I64), fn(Fn.SUB, col("a", 0, D), num(1.0))])


def test_no_filter_projection_only(any_ctx, oracle):
    rng = np.random.default_rng(3)
    n = 10_000
    cols = [random_column(rng, D, n, null_frac=0.1), random_column(rng, D, n)]
    run_both(any_ctx, oracle, cols, None, [fn(Fn.ADD, col("a", 0, D), col("b", 1, D)),
                                           fn(Fn.DIV, col("a", 0, D), col("b", 1, D)),
                                           fn(Fn.MOD, col("a", 0, D), col("b", 1, D))])


def test_kleene_tables_with_null_columns(any_ctx, oracle):
    """AND / OR / NOT / IF over nullable BOOLEAN columns (CompilerTest.kt:55-65,81-91,107-111 as columns)."""
    vals = [True, False, None]
    p = Column.from_values(B, [x for x in vals for _ in vals] * 50)
    q = Column.from_values(B, [y for _ in vals for y in vals] * 50)
    P, Q = col("p", 0, B), col("q", 1, B)
    projs = [fn(Fn.AND, P, Q), fn(Fn.OR, P, Q), fn(Fn.NOT, P),
             fn(Fn.IF, P, StringLiteralExpression("t"), StringLiteralExpression("f")),
             fn(Fn.IF, P, Q, fn(Fn.NOT, Q)), fn(Fn.CMP_EQ, P, Q), fn(Fn.CMP_LT, P, Q)]
    got = run_both(any_ctx, oracle, [p, q], None, projs)
    assert got[0].to_list()[:9] == [True, False, None, False, False, False, None, False, None]
    assert got[1].to_list()[:9] == [True, True, True, True, False, None, True, None, None]
    assert got[3].to_list()[:9] == ["t", "t", "t", "f", "f", "f", None, None, None]
    # filter keeps only non-null true (FilterOperator.kt:20)
    kept = run_both(any_ctx, oracle, [p, q], fn(Fn.OR, P, Q), [P, Q])
    assert len(kept[0]) == 50 * 5


@pytest.mark.parametrize("mode_name", ["total", "ieee"])
def test_f64_special_values_all_comparisons(any_ctx, oracle, mode_name):
    """NaN / -0.0 / Inf cross product, both comparison semantics."""
    from helpers import SPECIAL_F64
    xs = [x for x in SPECIAL_F64 for _ in SPECIAL_F64]
    ys = [y for _ in SPECIAL_F64 for y in SPECIAL_F64]
    a, b = Column(D, np.array(xs)), Column(D, np.array(ys))
    A_, B_ = col("a", 0, D), col("b", 1, D)
    projs = [fn(f, A_, B_) for f in (Fn.CMP_LT, Fn.CMP_LE, Fn.CMP_GE, Fn.CMP_GT, Fn.CMP_EQ, Fn.CMP_NE)]
    projs += [fn(f, A_, B_) for f in (Fn.ADD, Fn.SUB, Fn.MUL, Fn.DIV, Fn.MOD)] + [fn(Fn.UNARY_MINUS, A_)]
    projs += [fn(Fn.CMP_LT, A_, num(0.0)), fn(Fn.CMP_GE, A_, num(float("nan"))), fn(Fn.CMP_GT, A_, num(100.0))]
    if mode_name == "total":
        any_ctx.set_cmp_semantics(N.CMP_TOTAL_ORDER)
        run_both(any_ctx, oracle, [a, b], None, projs, oracle.BYTECODE_COMPILER)
        run_both(any_ctx, oracle, [a, b], None, projs, oracle.INTERPRETER)
    else:
        any_ctx.set_cmp_semantics(N.CMP_IEEE)
        try:
            run_both(any_ctx, oracle, [a, b], None, projs, oracle.CLOSURE_COMPILER)
        finally:
            any_ctx.set_cmp_semantics(N.CMP_TOTAL_ORDER)


def test_fma_contraction_is_off(any_ctx, oracle):
    """a + 10*b must round twice like DMUL;DADD."""
    a = float.fromhex("0x1.acd7053aa42a3p-1")
    b = float.fromhex("0x1.1ce794bb05232p-1")
    cols = [Column(D, np.full(300, a)), Column(D, np.full(300, b))]
    got = run_both(any_ctx, oracle, cols, None, [fn(Fn.ADD, col("a", 0, D), fn(Fn.MUL, num(10.0), col("b", 1, D)))])
    assert got[0].data[0].hex() == "0x1.99bc5a911af12p+2"


def test_integer_extension_wrap_div_mod(any_ctx, oracle):
    from helpers import SPECIAL_I32, SPECIAL_I64
    xs = [x for x in SPECIAL_I64 for _ in SPECIAL_I64]
    ys = [y for _ in SPECIAL_I64 for y in SPECIAL_I64]
    a, b = Column(I64, np.array(xs, dtype=np.int64)), Column(I64, np.array(ys, dtype=np.int64))
    x32 = [x for x in SPECIAL_I32 for _ in SPECIAL_I32]
    y32 = [y for _ in SPECIAL_I32 for y in SPECIAL_I32]
    n = min(len(xs), len(x32))
    cols = [Column(I64, a.data[:n]), Column(I64, b.data[:n]), Column(I32, np.array(x32[:n], dtype=np.int32)),
            Column(I32, np.array(y32[:n], dtype=np.int32))]
    A_, B_, C_, D_ = col("a", 0, I64), col("b", 1, I64), col("c", 2, I32), col("d", 3, I32)
    projs = [fn(f, A_, B_) for f in (Fn.ADD, Fn.SUB, Fn.MUL, Fn.DIV, Fn.MOD)]
    projs += [fn(f, C_, D_) for f in (Fn.ADD, Fn.SUB, Fn.MUL, Fn.DIV, Fn.MOD)]
    run_both(any_ctx, oracle, cols, None, projs)
    projs = [fn(Fn.ADD, A_, C_), fn(Fn.MUL, C_, num(1.5)), fn(Fn.UNARY_MINUS, A_), fn(Fn.UNARY_MINUS, C_),
              fn(Fn.CMP_LT, A_, B_), fn(Fn.CMP_EQ, A_, C_), fn(Fn.CMP_GE, C_, num(100.0)), fn(Fn.CMP_LE, A_, num(99.5))]
    run_both(any_ctx, oracle, cols, None, projs)


def test_int64_compare_against_literals_at_the_2_53_boundary(any_ctx, oracle):
    """(double)int64 OP literal is the widened comparison (BytecodeCompiler.kt:298-320: Double.compare on the converted
    value).  At L = +-2^53 the integers 2^53 and 2^53+1 convert to the same double, so comparing on the integers there is
    wrong (round-1 bug in both executors); every literal of the pool, all six comparisons, both operand orders."""
    from helpers import BOUNDARY_LITERALS
    vals = [2 ** 53, -(2 ** 53), 2 ** 53 + 1, -(2 ** 53 + 1), 2 ** 53 + 2, -(2 ** 53 + 2), 2 ** 53 - 1, -(2 ** 53 - 1),
            2 ** 53 + 3, 2 ** 63 - 1, -(2 ** 63), 2 ** 63 - 512, 2 ** 63 - 513, 0, 1, -1]
    x = Column(I64, np.array(vals * 9, dtype=np.int64))
    x32 = Column(I32, np.array(([2 ** 31 - 1, -(2 ** 31), 0, 1, -1, 7, 100, -100] * 18), dtype=np.int32))
    X, Y = col("x", 0, I64), col("y", 1, I32)
    cmps = (Fn.CMP_LT, Fn.CMP_LE, Fn.CMP_GE, Fn.CMP_GT, Fn.CMP_EQ, Fn.CMP_NE)
    for lit in BOUNDARY_LITERALS + [2.0 ** 31, -(2.0 ** 31), 2.0 ** 31 - 1]:
        projs = [fn(f, X, num(lit)) for f in cmps] + [fn(f, num(lit), X) for f in cmps]
        projs += [fn(Fn.CMP_LE, Y, num(lit)), fn(Fn.CMP_EQ, num(lit), Y)]
        got = run_both(any_ctx, oracle, [x, x32], None, projs)
        if lit == 2.0 ** 53:   # the case the integer shortcut got wrong: (double)(2^53+1) == 2^53
            assert got[4].to_list()[2] is True and got[3].to_list()[2] is False
        # as a Filter too (the fused kernel's conjunct path)
        run_both(any_ctx, oracle, [x, x32], fn(Fn.CMP_LE, X, num(lit)), [X])
        run_both(any_ctx, oracle, [x, x32], fn(Fn.AND, fn(Fn.CMP_GT, X, num(lit)), fn(Fn.CMP_GE, Y, num(0))), [X, Y])


def test_integers_within_2_53_match_double_only_reference(any_ctx, oracle):
    """with |values| <= 2^53 the INT64 extension equals the DOUBLE-only reference."""
    rng = np.random.default_rng(11)
    n = 20_000
    ai = rng.integers(0, 1000, n, dtype=np.int64)
    bi = rng.integers(0, 2 ** 31, n, dtype=np.int64)
    flt_i = fn(Fn.CMP_LT, col("a", 0, I64), num(100))
    got = run_both(any_ctx, oracle, [Column(I64, ai), Column(I64, bi)], flt_i, [fn(Fn.ADD, col("a", 0, I64), col("b", 1, I64))])
    ref = oracle.filter_project([Column(D, ai.astype(np.float64)), Column(D, bi.astype(np.float64))],
                                fn(Fn.CMP_LT, col("a", 0, D), num(100)), [fn(Fn.ADD, col("a", 0, D), col("b", 1, D))],
                                oracle.BYTECODE_COMPILER)
    assert np.array_equal(got[0].data.astype(np.float64), ref[0].data)


def test_dictionary_equality_and_gather_project(any_ctx, oracle):
    """BASELINE config 4 shape: SELECT s, v FROM t WHERE s = 'k0042'"""
    rng = np.random.default_rng(5)
    n = 40_000
    d = ["k%04d" % i for i in range(1000)]
    s = random_column(rng, S, n, null_frac=0.05, dictionary=d)
    v = Column(D, rng.random(n))
    Sx, V = col("s", 0, S), col("v", 1, D)
    got = run_both(any_ctx, oracle, [s, v], fn(Fn.CMP_EQ, Sx, StringLiteralExpression("k0042")), [Sx, V])
    assert set(got[0].to_list()) <= {"k0042"}
    run_both(any_ctx, oracle, [s, v], fn(Fn.CMP_NE, Sx, StringLiteralExpression("nope")), [Sx])
    run_both(any_ctx, oracle, [s, v], fn(Fn.CMP_EQ, Sx, StringLiteralExpression("nope")), [Sx, V])
    run_both(any_ctx, oracle, [s, v], None,
             [fn(Fn.IF, fn(Fn.CMP_LT, V, num(0.5)), Sx, StringLiteralExpression("other")),
              fn(Fn.IF, fn(Fn.CMP_LT, V, num(0.5)), StringLiteralExpression("k0001"), Sx),
              fn(Fn.CMP_EQ, Sx, Sx)])


def test_q6_shape(any_ctx, oracle):
    """BASELINE config 3 shape: TPC-H Q6 predicate, int32 dates, f64 price/discount/quantity."""
    rng = np.random.default_rng(6)
    n = 60_000
    ship = Column(I32, rng.integers(8036, 10562, n, dtype=np.int32))
    disc = Column(D, rng.integers(0, 11, n).astype(np.float64) * 0.01)
    qty = Column(D, rng.integers(1, 51, n).astype(np.float64))
    price = Column(D, np.round(qty.data * rng.uniform(900, 2100, n), 2))
    SH, DI, QT, PR = col("l_shipdate", 0, I32), col("l_discount", 1, D), col("l_quantity", 2, D), col("l_extendedprice", 3, D)
    flt = fn(Fn.AND, fn(Fn.AND, fn(Fn.AND, fn(Fn.AND, fn(Fn.CMP_GE, SH, num(8766)), fn(Fn.CMP_LT, SH, num(9131))),
                                   fn(Fn.CMP_GE, DI, num(0.05))), fn(Fn.CMP_LE, DI, num(0.07))), fn(Fn.CMP_LT, QT, num(24)))
    got = run_both(any_ctx, oracle, [ship, disc, qty, price], flt, [fn(Fn.MUL, PR, DI)])
    assert 0 < len(got[0]) < n * 0.05


@pytest.mark.parametrize("seed", range(12 + int(os.environ.get("QE_FUZZ_EXTRA", "0"))))
def test_random_expression_trees(any_ctx, oracle, seed):
    """Differential test in the spirit of CompilerTest's @EnumSource(Mode): random typed trees,
    null-heavy data with NaN/-0.0/extreme integers, GPU vs oracle."""
    rnd = random.Random(seed)
    rng = np.random.default_rng(seed)
    n = 3000 + seed * 17
    schema = [("a", D), ("b", D), ("i", I64), ("j", I32), ("p", B), ("q", B), ("s", S)]
    dictionary = ["k%04d" % i for i in range(8)]
    cols = [random_column(rng, t, n, null_frac=rnd.choice([0.0, 0.1, 0.5]), dictionary=dictionary) for _, t in schema]
    g = ExprGen(rnd, schema)
    g.dicts = {"s": dictionary}
    for _ in range(4):
        flt = g.boolean(3) if rnd.random() < 0.8 else None
        projs = [g.numeric(3) if rnd.random() < 0.7 else g.boolean(2) for _ in range(rnd.randint(1, 3))]
        run_both(any_ctx, oracle, cols, flt, projs)


def test_generator_matches_oracle(gpu_ctx, oracle):
    """qe_batch_generate (device) == qo_generate (oracle) for every generator kind, global row index."""
    import ctypes as C
    kinds = [(N.GEN_I64_MOD, np.int64, dict(modulus=1000)), (N.GEN_I64_MOD, np.int64, dict(modulus=2 ** 31, offset=-5)),
             (N.GEN_I32_MOD, np.int32, dict(modulus=2526, offset=8036)), (N.GEN_F64_UNIT, np.float64, {}),
             (N.GEN_F64_MOD, np.float64, dict(modulus=1000)), (N.GEN_F64_STEP, np.float64, dict(modulus=11, step=0.01)),
             (N.GEN_F64_PRICE, np.float64, dict(aux_col_id=2))]
    specs, ospecs = [], []
    for cid, (kind, _, kw) in enumerate(kinds):
        s = N.GenSpec(); s.kind = kind; s.col_id = cid; s.modulus = kw.get("modulus", 0); s.offset = kw.get("offset", 0)
        s.step = kw.get("step", 0.0); s.aux_col_id = kw.get("aux_col_id", 0); s.null_pct = 1 if cid % 2 else 0
        specs.append(s)
        o = oracle.GenSpec(); o.kind = kind; o.col_id = cid; o.modulus = s.modulus; o.offset = s.offset; o.step = s.step
        o.aux_col_id = s.aux_col_id; o.null_pct = s.null_pct
        ospecs.append(o)
    n, row_begin = 10_000 + 37, 1 << 33
    batch = E.DeviceBatch.generate(gpu_ctx, specs, n, row_begin=row_begin, seed=42)
    for cid, (kind, npdt, _) in enumerate(kinds):
        got = batch.column_to_host(cid)
        data, valid = oracle.generate(ospecs[cid], 42, row_begin, n, npdt)
        want = Column(got.type, data, valid)
        assert_columns_equal(got, want, f"generator kind {kind}")
    batch.free()


def test_large_batch_properties(gpu_ctx, oracle):
    """Size-independent properties at a size the oracle cannot walk: count == popcount of the predicate
    evaluated by an independent plan, order preservation (projected row ids strictly increasing),
    and idempotence (second run identical)."""
    n = 50_000_000
    s0 = N.GenSpec(); s0.kind = N.GEN_I64_MOD; s0.col_id = 0; s0.modulus = 1000
    s1 = N.GenSpec(); s1.kind = N.GEN_F64_UNIT; s1.col_id = 2
    batch = E.DeviceBatch.generate(gpu_ctx, [s0, s1], n)
    A_, C_ = col("a", 0, I64), col("c", 1, D)
    flt = fn(Fn.AND, fn(Fn.CMP_LT, A_, num(100)), fn(Fn.CMP_LT, C_, num(0.5)))
    cf = gpu_ctx.compile(flt)
    proj = [gpu_ctx.compile(C_), gpu_ctx.compile(fn(Fn.ADD, A_, A_))]
    r1 = E.filter_project(gpu_ctx, batch, cf, proj)
    c1 = r1.to_columns()
    # independent count through the aggregate path
    vals, nsel = E.filter_aggregate(gpu_ctx, batch, cf, [gpu_ctx.compile(C_)], [N.AGG_COUNT])
    assert r1.count == nsel == int(vals[0])
    assert abs(r1.count / n - 0.05) < 0.001
    assert np.all(c1[0].data < 0.5) and np.all(c1[1].data < 200) and np.all(c1[1].data % 2 == 0)
    # oracle on a prefix: the first rows of the result are exactly the oracle's result on the first 200k rows
    m = 200_000
    pa, pc = batch.column_to_host(0, 0, m), batch.column_to_host(1, 0, m)
    want = oracle.filter_project([pa, pc], flt, [C_, fn(Fn.ADD, A_, A_)], oracle.BYTECODE_COMPILER)
    k = len(want[0])
    assert np.array_equal(c1[0].data[:k], want[0].data) and np.array_equal(c1[1].data[:k], want[1].data)
    r2 = E.filter_project(gpu_ctx, batch, cf, proj)
    c2 = r2.to_columns()
    assert np.array_equal(c1[0].data, c2[0].data) and np.array_equal(c1[1].data, c2[1].data)
    r1.free(); r2.free(); batch.free()


def test_full_size_1b_rows_properties(gpu_ctx, oracle):
    """BASELINE config 2 at its full size (1 B rows): properties that do not need the oracle to walk the
    batch -- the projected global row ids are strictly increasing (order preserved, nothing duplicated),
    every output row satisfies the predicate, count == the independent aggregate COUNT == popcount of the
    predicate over two far-apart windows walked by the oracle, and a second run is identical."""
    n = 1_000_000_000
    specs = []
    for kind, cid, mod in ((N.GEN_I64_MOD, 0, 1000), (N.GEN_I64_ROWID, 1, 0), (N.GEN_F64_UNIT, 2, 0)):
        s = N.GenSpec(); s.kind = kind; s.col_id = cid; s.modulus = mod
        specs.append(s)
    batch = E.DeviceBatch.generate(gpu_ctx, specs, n)
    A_, R_, C_ = col("a", 0, I64), col("rowid", 1, I64), col("c", 2, D)
    flt = fn(Fn.AND, fn(Fn.CMP_LT, A_, num(100)), fn(Fn.CMP_LT, C_, num(0.5)))
    cf = gpu_ctx.compile(flt)
    projs = [gpu_ctx.compile(R_), gpu_ctx.compile(fn(Fn.ADD, A_, R_)), gpu_ctx.compile(fn(Fn.MUL, C_, num(2.0)))]
    r1 = E.filter_project(gpu_ctx, batch, cf, projs)
    rid, apr, c2 = (c.data for c in r1.to_columns())
    assert abs(r1.count / n - 0.05) < 0.0005
    assert np.all(np.diff(rid) > 0) and rid[0] >= 0 and rid[-1] < n          # stable, no duplicates
    assert np.all(apr - rid < 100) and np.all(apr - rid >= 0) and np.all(c2 < 1.0)
    vals, nsel = E.filter_aggregate(gpu_ctx, batch, cf, [gpu_ctx.compile(C_)], [N.AGG_COUNT])
    assert nsel == r1.count == int(vals[0])
    # two windows checked row by row against the oracle (which is far too slow for 1 B rows)
    for begin in (0, 999_000_000 - 999_000_000 % 64):
        m = 500_000
        win = [batch.column_to_host(j, begin, m) for j in range(3)]
        want = oracle.filter_project(win, flt, [R_, fn(Fn.ADD, A_, R_), fn(Fn.MUL, C_, num(2.0))], oracle.BYTECODE_COMPILER)
        lo = np.searchsorted(rid, begin)
        k = len(want[0])
        assert np.array_equal(rid[lo:lo + k], want[0].data)
        assert np.array_equal(apr[lo:lo + k], want[1].data)
        assert np.array_equal(c2[lo:lo + k].view(np.uint64), want[2].data.view(np.uint64))
        assert lo + k == len(rid) or rid[lo + k] >= begin + m
    r2 = E.filter_project(gpu_ctx, batch, cf, projs)
    assert r2.count == r1.count and np.array_equal(r2.column_to_host(0).data, rid)
    r1.free(); r2.free(); batch.free()
    gpu_ctx.trim()


def test_full_size_1b_rows_nullable_properties(gpu_ctx, oracle):
    """SURVEY 8(d)'s nullable variant of config 2 at its full size (1 B rows, ~1 % NULLs in every input column: validity bitmaps
    read, nullable outputs packed) through the workload object bench.py runs with --null-pct 1, plus a global-row-id column.
    Row ids strictly increasing; count == an independent aggregate COUNT of the kept rows; the NULLs of every output column
    == kept rows - the aggregate COUNT(<that expression>) (COUNT skips NULLs: Accumulators.kt:26-36) -- a NULL b under a kept
    row must give a NULL a + b, never drop the row; the expected selectivity 0.05 * 0.99^2; two windows walked by the oracle."""
    from queryengine_amd import workloads as W
    n = 1_000_000_000
    wl = W.config2(n, null_pct=1)
    rid_spec = N.GenSpec(); rid_spec.kind = N.GEN_I64_ROWID; rid_spec.col_id = 99
    batch = E.DeviceBatch.generate(gpu_ctx, [c.spec(gpu_ctx) for c in wl.columns] + [rid_spec], n)
    R_ = col("rowid", 3, I64)
    proj_exprs = list(wl.projections) + [R_]
    cf, cp = gpu_ctx.compile(wl.filter), [gpu_ctx.compile(p) for p in proj_exprs]
    r1 = E.filter_project(gpu_ctx, batch, cf, cp)
    cols1 = r1.to_columns()
    rid = cols1[-1].data
    assert abs(r1.count / n - 0.05 * 0.99 * 0.99) < 0.0005
    assert np.all(np.diff(rid) > 0) and rid[0] >= 0 and rid[-1] < n
    vals, nsel = E.filter_aggregate(gpu_ctx, batch, cf, cp, [N.AGG_COUNT] * len(cp))
    assert nsel == r1.count == int(vals[-1])
    for i, c in enumerate(cols1[:-1]):
        nulls = 0 if c.valid is None else int((~c.valid).sum())
        assert nulls == r1.count - int(vals[i]), f"output {i}: {nulls} NULLs, aggregate says {r1.count - int(vals[i])}"
    assert int((~cols1[0].valid).sum()) > 0.005 * r1.count        # a + b is NULL where b is (a passed the filter: not NULL)
    assert cols1[1].valid is None or bool(cols1[1].valid.all())   # c * 2.0: c passed the filter, never NULL under a kept row
    for begin in (0, (n - 700_000) - (n - 700_000) % 64):
        m = 600_000
        win = [batch.column_to_host(j, begin, m) for j in range(4)]
        want = oracle.filter_project(win, wl.filter, proj_exprs, oracle.BYTECODE_COMPILER)
        lo = int(np.searchsorted(rid, begin))
        k = len(want[0])
        assert k > 0 and np.array_equal(rid[lo:lo + k], want[-1].data)
        for g, w in zip(cols1[:-1], want[:-1]):
            gv = None if g.valid is None else g.valid[lo:lo + k]
            assert_columns_equal(Column(g.type, g.data[lo:lo + k], gv), w, f"nullable cfg 2 window at {begin}")
    r1.free(); batch.free()
    gpu_ctx.trim()


def test_cfg5_rank7_shard_properties(gpu_ctx, oracle):
    """BASELINE configs[4] ("cfg 5": 10 B rows row-range sharded over 8 GPUs), the single-GPU half of it: RANK 7's shard --
    1.25 B rows whose GLOBAL row ids start at 7 x 1.25 B = 8.75e9 > 2^33 -- through the very workload object bench.py runs at
    N > 1 (W.config2: same generator specs, filter, projections) plus one appended global-row-id column.  Properties: the
    projected global row ids are strictly increasing inside [row_begin, row_begin + n) (order kept, nothing duplicated, the
    shard offset really applied), count == an independent aggregate COUNT, the expected selectivity, the device generator ==
    the oracle's generator at those global ids, two far-apart windows walked row by row by the oracle, a second run identical."""
    from queryengine_amd import workloads as W
    from queryengine_amd.distributed import shard_range
    total, world, rank = 10_000_000_000, 8, 7
    begin, end = shard_range(total, rank, world)
    n = end - begin
    assert n == 1_250_000_000 and begin == 8_750_000_000 and begin > 1 << 33
    wl = W.config2(n)
    specs = [c.spec(gpu_ctx) for c in wl.columns]
    rid_spec = N.GenSpec(); rid_spec.kind = N.GEN_I64_ROWID; rid_spec.col_id = 99
    batch = E.DeviceBatch.generate(gpu_ctx, specs + [rid_spec], n, row_begin=begin)
    R_ = col("rowid", len(wl.columns), I64)
    proj_exprs = list(wl.projections) + [R_]
    cf = gpu_ctx.compile(wl.filter)
    cp = [gpu_ctx.compile(p) for p in proj_exprs]
    r1 = E.filter_project(gpu_ctx, batch, cf, cp)
    cols1 = r1.to_columns()
    rid = cols1[-1].data
    assert abs(r1.count / n - 0.05) < 0.0005
    assert np.all(np.diff(rid) > 0) and rid[0] >= begin and rid[-1] < end
    vals, nsel = E.filter_aggregate(gpu_ctx, batch, cf, [gpu_ctx.compile(R_)], [N.AGG_COUNT])
    assert nsel == r1.count == int(vals[0])
    for local in (0, (n - 700_000) - (n - 700_000) % 64):
        m = 600_000
        win = [batch.column_to_host(j, local, m) for j in range(len(specs) + 1)]
        assert win[-1].data[0] == begin + local and win[-1].data[-1] == begin + local + m - 1
        for j, c in enumerate(wl.columns):      # the device generator at global ids > 2^33 == the oracle's
            o = oracle.GenSpec(); o.kind, o.col_id, o.modulus, o.offset, o.step = c.kind, c.col_id, c.modulus, c.offset, c.step
            o.aux_col_id, o.null_pct = c.aux_col_id, c.null_pct
            data, _ = oracle.generate(o, 42, begin + local, m, np.float64 if c.type == D else np.int64)
            assert np.array_equal(win[j].data.view(np.uint64), data.view(np.uint64)), f"generator, column {c.name}"
        want = oracle.filter_project(win, wl.filter, proj_exprs, oracle.BYTECODE_COMPILER)
        lo = int(np.searchsorted(rid, begin + local))
        k = len(want[0])
        assert k > 0 and np.array_equal(rid[lo:lo + k], want[-1].data)
        assert lo + k == len(rid) or rid[lo + k] >= begin + local + m
        for g, w in zip(cols1[:-1], want[:-1]):
            assert_columns_equal(Column(g.type, g.data[lo:lo + k], None), w, f"cfg 5 rank 7 window at {local}")
    r2 = E.filter_project(gpu_ctx, batch, cf, cp)
    assert r2.count == r1.count and np.array_equal(r2.column_to_host(len(cp) - 1).data, rid)
    r1.free(); r2.free(); batch.free()
    gpu_ctx.trim()


@pytest.mark.parametrize("name", ["config3", "config4", "config4_10keys"])
def test_full_size_bench_workloads_properties(gpu_ctx, oracle, name):
    """BASELINE configs 3 and 4 at their FULL size, through the very workload objects bench.py runs (W.config3() /
    W.config4(): same generator specs, same filter, same projections -- not look-alikes) plus one appended row-id column:
    projected global row ids strictly increasing (order kept, nothing duplicated), count == an independent aggregate
    COUNT, the expected selectivity, two far-apart windows walked row by row by the oracle, and a second run identical."""
    from queryengine_amd import workloads as W
    wl = {"config3": W.config3, "config4": W.config4, "config4_10keys": lambda: W.config4(nkeys=10, key="k0004")}[name]()
    n = wl.default_rows
    specs = [c.spec(gpu_ctx) for c in wl.columns]
    rid_spec = N.GenSpec(); rid_spec.kind = N.GEN_I64_ROWID; rid_spec.col_id = 99
    batch = E.DeviceBatch.generate(gpu_ctx, specs + [rid_spec], n)
    R_ = col("rowid", len(wl.columns), I64)
    proj_exprs = list(wl.projections) + [R_]
    cf = gpu_ctx.compile(wl.filter)
    cp = [gpu_ctx.compile(p) for p in proj_exprs]
    r1 = E.filter_project(gpu_ctx, batch, cf, cp)
    cols1 = r1.to_columns()
    rid = cols1[-1].data
    assert abs(r1.count / n - wl.expected_selectivity) < 0.02 * wl.expected_selectivity + 1e-5
    assert np.all(np.diff(rid) > 0) and rid[0] >= 0 and rid[-1] < n
    vals, nsel = E.filter_aggregate(gpu_ctx, batch, cf, [gpu_ctx.compile(R_)], [N.AGG_COUNT])
    assert nsel == r1.count == int(vals[0])
    dicts = [getattr(c, "dictionary", None) for c in wl.columns] + [None]
    for begin in (0, (n - 700_000) - (n - 700_000) % 64):
        m = 600_000
        win = [batch.column_to_host(j, begin, m, dictionary=dicts[j]) for j in range(len(dicts))]
        want = oracle.filter_project(win, wl.filter, proj_exprs, oracle.BYTECODE_COMPILER)
        lo = int(np.searchsorted(rid, begin))
        k = len(want[0])
        assert k > 0 and np.array_equal(rid[lo:lo + k], want[-1].data)
        assert lo + k == len(rid) or rid[lo + k] >= begin + m
        for g, w in zip(cols1[:-1], want[:-1]):
            assert_columns_equal(Column(g.type, g.data[lo:lo + k], None, g.dictionary), w, f"{name} window at {begin}")
    r2 = E.filter_project(gpu_ctx, batch, cf, cp)
    assert r2.count == r1.count and np.array_equal(r2.column_to_host(len(cp) - 1).data, rid)
    r1.free(); r2.free(); batch.free()
    gpu_ctx.trim()


def test_dense_form_is_chosen_after_a_high_selectivity_run(oracle):
    """The fused executor remembers the share of rows a plan kept: the first execution runs the LDS-ring single pass, the
    next one of a plan that kept >= 12 % runs the dense single pass (a workgroup per tile, one tile parked in LDS, resolved
    one tile later, coalesced ordered stores); a selective plan stays with the ring.  All give the reference's rows in input
    order.  Several tiles per workgroup and ragged tails (n is not a multiple of the tile)."""
    from queryengine_amd import engine as E
    from queryengine_amd import workloads as W
    n = 2_500_001
    ctx = E.Context(device=0)
    for a_limit, c_limit, want_form in ((1000, 1.0, N.FORM_DENSE), (1000, 0.5, N.FORM_DENSE), (300, 0.5, N.FORM_DENSE), (100, 0.5, N.FORM_RING)):
        wl = W.config2(n, a_limit=a_limit, c_limit=c_limit, null_pct=1)
        batch = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], n)
        host = [batch.column_to_host(j) for j in range(batch.ncols)]
        want = oracle.filter_project(host, wl.filter, wl.projections, oracle.BYTECODE_COMPILER)
        cf, cp = ctx.compile(wl.filter), [ctx.compile(p) for p in wl.projections]
        for rep in range(3):
            res = E.filter_project(ctx, batch, cf, cp)
            assert ctx.last_form == (N.FORM_RING if rep == 0 else want_form)
            got = res.to_columns()
            res.free()
            for g, w in zip(got, want):
                assert_columns_equal(g, w, f"a<{a_limit} c<{c_limit} rep {rep}")
        batch.free()
    ctx.close()


def test_first_execution_samples_its_selectivity_on_a_large_batch(oracle):
    """From 8 Mi rows on, the FIRST execution of a plan estimates its selectivity from 256 chunks spread over the batch (the
    count pass of the two-pass form with a chunk stride) and starts in the form that fits: dense when the plan keeps >= 12 %
    of its rows, the local form (dependency-free scan into per-chunk slots + one move) when it keeps <= 3 %, the LDS ring in
    between.  Same rows either way: two windows are walked by the oracle, the count must equal
    the next execution's, and debug bit 65536 switches the sample off (first execution = ring)."""
    from queryengine_amd import engine as E
    from queryengine_amd import workloads as W
    n = 12_000_017
    for a_limit, c_limit, want_form, tuning in ((1000, 1.0, N.FORM_DENSE, []), (400, 0.5, N.FORM_DENSE, []), (100, 0.5, N.FORM_RING, []),
                                                (20, 0.5, N.FORM_LOCAL, []), (20, 0.5, N.FORM_RING, [0, 0, 0, 0, 0, 524288]),
                                                (1000, 1.0, N.FORM_RING, [0, 0, 0, 0, 0, 65536])):
        ctx = E.Context(device=0, tuning=tuning)
        wl = W.config2(n, a_limit=a_limit, c_limit=c_limit)
        batch = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], n)
        cf, cp = ctx.compile(wl.filter), [ctx.compile(p) for p in wl.projections]
        res = E.filter_project(ctx, batch, cf, cp)
        assert ctx.last_form == want_form, (a_limit, c_limit, ctx.last_form)
        first = res.to_columns()
        res.free()
        res = E.filter_project(ctx, batch, cf, cp)
        second = res.to_columns()
        res.free()
        for g, w in zip(first, second):
            assert_columns_equal(g, w, f"a<{a_limit} c<{c_limit}: first vs second execution")
        # the head of the batch, row by row, against the oracle
        m = 70_000
        host = [batch.column_to_host(j, 0, m) for j in range(batch.ncols)]
        want = oracle.filter_project(host, wl.filter, wl.projections, oracle.BYTECODE_COMPILER)
        k = len(want[0])
        for g, w in zip(first, want):
            assert_columns_equal(Column(g.type, g.data[:k], None, g.dictionary), w, f"a<{a_limit} c<{c_limit}: head window")
        batch.free()
        ctx.close()


def test_local_form_falls_back_when_a_chunk_overflows_its_slot(oracle):
    """The local form parks a chunk's kept rows in a slot of ring_entries rows.  Kept rows that are CLUSTERED (here: one
    dense stretch in an otherwise empty batch -- 1.5 % overall, so the sample picks the local form) overflow the slots of
    the chunks they fall in: the kernel reports it, the SAME execution answers through the single-pass kernel (same rows,
    same order) and the plan does not try the local form again."""
    from queryengine_amd import engine as E
    n = 9_000_001
    a = np.full(n, 500, dtype=np.int64)
    lo, hi = 3_000_000, 3_135_000
    a[lo:hi] = 7
    rid = np.arange(n, dtype=np.int64)
    cols = [Column(I64, a, None), Column(I64, rid, None)]
    A_, R_ = col("a", 0, I64), col("rid", 1, I64)
    flt = fn(Fn.CMP_LT, A_, num(100))
    ctx = E.Context(device=0)
    batch = E.DeviceBatch.from_columns(ctx, cols)
    cf, cp = ctx.compile(flt), [ctx.compile(R_), ctx.compile(fn(Fn.ADD, A_, R_))]
    for rep in range(3):
        res = E.filter_project(ctx, batch, cf, cp)
        assert ctx.last_form == N.FORM_RING, (rep, ctx.last_form)     # rep 0: local tried, overflowed, fell back; later: not tried
        got = res.to_columns()
        res.free()
        assert len(got[0]) == hi - lo
        assert np.array_equal(got[0].data, rid[lo:hi]) and np.array_equal(got[1].data, rid[lo:hi] + 7)
    # the same data spread evenly keeps the local form
    a2 = np.where(rid % 67 == 0, 7, 500).astype(np.int64)
    b2 = E.DeviceBatch.from_columns(ctx, [Column(I64, a2, None), Column(I64, rid, None)])
    flt2 = fn(Fn.CMP_LT, A_, num(99))      # another plan (another literal): its own memory
    cf2 = ctx.compile(flt2)
    for rep in range(2):
        res = E.filter_project(ctx, b2, cf2, cp)
        assert ctx.last_form == N.FORM_LOCAL
        got = res.to_columns()
        res.free()
        assert np.array_equal(got[0].data, rid[rid % 67 == 0]) and np.array_equal(got[1].data, rid[rid % 67 == 0] + 7)
    batch.free(); b2.free()
    ctx.close()


@pytest.mark.parametrize("seed", range(8 + int(os.environ.get("QE_FUZZ_EXTRA", "0"))))
def test_random_conjunctive_filters(any_ctx, oracle, seed):
    """Filters that ARE top-level AND chains of 2-4 random boolean trees over null-heavy columns, several chunks of rows:
    the staged (late-materialisation) evaluation -- a conjunct is only evaluated, and its columns only loaded, for rows
    on which every earlier conjunct was a non-null TRUE -- must keep exactly the rows the reference's Kleene AND keeps."""
    rnd = random.Random(1000 + seed)
    rng = np.random.default_rng(1000 + seed)
    n = 40_000 + seed * 1111
    schema = [("a", D), ("b", D), ("i", I64), ("j", I32), ("p", B), ("q", B), ("s", S)]
    dictionary = ["k%04d" % i for i in range(8)]
    cols = [random_column(rng, t, n, null_frac=rnd.choice([0.0, 0.1, 0.5]), dictionary=dictionary) for _, t in schema]
    g = ExprGen(rnd, schema)
    g.dicts = {"s": dictionary}
    for _ in range(2):
        flt = g.boolean(2)
        for _ in range(rnd.randint(1, 3)):
            flt = fn(Fn.AND, flt, g.boolean(2), t=B) if rnd.random() < 0.5 else fn(Fn.AND, g.boolean(2), flt, t=B)
        projs = [g.numeric(2) if rnd.random() < 0.7 else g.boolean(2) for _ in range(rnd.randint(1, 3))]
        run_both(any_ctx, oracle, cols, flt, projs)


def test_string_ordering_and_cross_dictionary_comparisons(any_ctx, oracle):
    """String.compareTo on the device (BytecodeCompiler.kt:300-306): all six comparisons of a dictionary column with a
    literal, with a column of the SAME dictionary and with a column of a DIFFERENT dictionary (ranks in one merged
    UTF-16 order), plus IF over two dictionaries (union dictionary, remapped codes); nulls on both sides."""
    rng = np.random.default_rng(21)
    n = 30_011
    d1 = ["pear", "apple", "fig", "Fig", "", "zebra", "～", "\U0001F600", "apples", "é"]
    d2 = ["fig", "banana", "\U0001F600", "apple", "kiwi", "zebra ", "A"]
    s1 = random_column(rng, S, n, null_frac=0.1, dictionary=d1)
    s2 = random_column(rng, S, n, null_frac=0.1, dictionary=d1)
    t = random_column(rng, S, n, null_frac=0.1, dictionary=d2)
    v = Column(D, rng.random(n))
    S1, S2, T, V = col("s1", 0, S), col("s2", 1, S), col("t", 2, S), col("v", 3, D)
    cols = [s1, s2, t, v]
    lit = StringLiteralExpression
    for f in (Fn.CMP_LT, Fn.CMP_LE, Fn.CMP_GE, Fn.CMP_GT, Fn.CMP_EQ, Fn.CMP_NE):
        run_both(any_ctx, oracle, cols, fn(f, S1, lit("fig")), [S1, fn(f, lit("b"), S1), fn(f, S1, S2), fn(f, S1, T), fn(f, T, lit("～"))])
    run_both(any_ctx, oracle, cols, fn(Fn.AND, fn(Fn.CMP_GE, S1, T), fn(Fn.CMP_LT, T, lit("kiwi"))),
             [fn(Fn.IF, fn(Fn.CMP_LT, V, num(0.5)), S1, T), fn(Fn.IF, fn(Fn.CMP_LT, V, num(0.3)), T, S2), V])


@pytest.mark.parametrize("form", ["single_pass", "two_pass", "per_node"])
def test_result_capacity_rows_bounds_the_output_buffers(oracle, form):
    """qe_options.result_capacity_rows: output buffers are sized for that many rows; a result that fits is exact, one that
    does not is an error (status QE_ERR_INVALID_ARG), never a silent truncation -- and the context stays usable."""
    from queryengine_amd import workloads as W
    n = 100_000
    wl = W.config2(n, null_pct=1)                       # ~4.9 % -> about 4 900 rows
    kw = {"single_pass": dict(), "two_pass": dict(tuning=[0, 0, 0, 0, 0, 512, 0, 0]), "per_node": dict(exec_mode=N.EXEC_PER_NODE)}[form]
    for cap, fits in ((6000, True), (3000, False)):
        ctx = E.Context(device=0, result_capacity_rows=cap, **kw)
        batch = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], n)
        host = [batch.column_to_host(j) for j in range(batch.ncols)]
        want = oracle.filter_project(host, wl.filter, wl.projections, oracle.BYTECODE_COMPILER)
        cf, cp = ctx.compile(wl.filter), [ctx.compile(p) for p in wl.projections]
        for _ in range(2):
            if fits:
                res = E.filter_project(ctx, batch, cf, cp)
                for g, w in zip(res.to_columns(), want):
                    assert_columns_equal(g, w, f"cap {cap}")
                res.free()
            else:
                with pytest.raises(N.QeError) as ei:
                    E.filter_project(ctx, batch, cf, cp)
                assert "result_capacity_rows" in str(ei.value)
        batch.free()
        ctx.close()


def _workload_vs_oracle(ctx, oracle, wl, n, reps=1):
    batch = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], n)
    host = [batch.column_to_host(j, dictionary=getattr(c, "dictionary", None)) for j, c in enumerate(wl.columns)]
    want = oracle.filter_project(host, wl.filter, wl.projections, oracle.BYTECODE_COMPILER)
    cf = ctx.compile(wl.filter) if wl.filter is not None else None
    cp = [ctx.compile(p) for p in wl.projections]
    for rep in range(reps):
        res = E.filter_project(ctx, batch, cf, cp)
        got = res.to_columns()
        assert res.count == len(want[0])
        for g, w in zip(got, want):
            assert_columns_equal(g, w, f"{wl.name} n={n} rep {rep}")
        res.free()
    batch.free()


@pytest.mark.parametrize("tuning", [[256, 16, 0, 0, 20004], [128, 8, 0, 0, 20008]], ids=["wide", "mid"])
def test_wide_geometry_parity(oracle, tuning):
    """The second and third geometry of the fused kernel (wide: 16 load groups per sub-tile, 512-entry LDS rings, 4 sub-tiles
    per chunk; mid: the default sub-tile, 8 sub-tiles per chunk, 512-entry rings, 2 waves per workgroup), pinned through
    qe_options.tuning: ragged sizes, several chunks, low and high selectivity, dictionary shape."""
    from queryengine_amd import workloads as W
    ctx = E.Context(device=0, tuning=tuning)
    for n in (1, 2047, 2048, 2049, 8191, 8192, 70_001, 300_000):
        _workload_vs_oracle(ctx, oracle, W.config2(n), n)
    for a_limit, c_limit in ((1000, 1.0), (1000, 0.5), (10, 0.5)):
        _workload_vs_oracle(ctx, oracle, W.config2(150_001, a_limit=a_limit, c_limit=c_limit), 150_001)
    _workload_vs_oracle(ctx, oracle, W.config4(200_003), 200_003)
    _workload_vs_oracle(ctx, oracle, W.config1(100_000), 100_000)
    ctx.close()


def test_geometry_choice_on_a_large_batch(oracle, tmp_path):
    """From 32 Mi rows on, the first executions of a plan time its three geometries (best of 3 each) and the faster one is
    kept: every execution -- exploring or settled -- returns exactly the oracle's rows.  The decision is persisted next to
    the code object: a NEW context on the same JIT cache runs the same geometry without exploring (same plan => same
    geometry) -- unless the two candidates were closer than 7 % (the spread of one binary over the boxes of
    the pool): such a decision is measured again once per context."""
    from queryengine_amd import workloads as W
    n = 34_000_001
    cache = str(tmp_path / "jit")
    ctx = E.Context(device=0, jit_cache_dir=cache)
    wl = W.config2(n)
    batch = E.DeviceBatch.describe(ctx, [Column(c.type, np.zeros(2, dtype=np.int64 if c.type == I64 else np.float64)) for c in wl.columns])
    cf, cp = ctx.compile(wl.filter), [ctx.compile(p) for p in wl.projections]
    assert E.chosen_geometry(ctx, batch, cf, cp) == (-1, False)
    _workload_vs_oracle(ctx, oracle, wl, n, reps=8)
    assert E.chosen_geometry(ctx, batch, cf, cp)[0] == -1          # 8 of the 9 exploring executions (three candidates, best of 3)
    _workload_vs_oracle(ctx, oracle, wl, n, reps=2)
    chosen, cached = E.chosen_geometry(ctx, batch, cf, cp)
    assert chosen in (0, 1, 2) and not cached
    plain = W.config2(n)
    plain.filter = None                                   # a projection without a Filter takes part in the choice too
    _workload_vs_oracle(ctx, oracle, plain, n, reps=3)
    ctx.close()
    ctx2 = E.Context(device=0, jit_cache_dir=cache)
    batch2 = E.DeviceBatch.describe(ctx2, [Column(c.type, np.zeros(2, dtype=np.int64 if c.type == I64 else np.float64)) for c in wl.columns])
    cf2, cp2 = ctx2.compile(wl.filter), [ctx2.compile(p) for p in wl.projections]
    first = E.chosen_geometry(ctx2, batch2, cf2, cp2)
    if first[1]:                                           # a clear decision (the candidates were >= 7 % apart): reused as it is
        assert first == (chosen, True)
        _workload_vs_oracle(ctx2, oracle, wl, n, reps=1)   # runs the persisted geometry at once
        assert E.chosen_geometry(ctx2, batch2, cf2, cp2) == (chosen, True)
    else:                                                  # a close call (inside the box-to-box spread): measured again on this context
        assert first == (-1, False)
        _workload_vs_oracle(ctx2, oracle, wl, n, reps=10)
        again, cached = E.chosen_geometry(ctx2, batch2, cf2, cp2)
        assert again in (0, 1, 2) and not cached
    ctx2.close()


def test_conjuncts_are_ordered_by_measured_pass_rate(oracle):
    """`c < 0.5 AND a < 100` loads c for every row and a for half of them when evaluated as written; on its first execution on a
    large batch the plan measures every conjunct's pass rate and evaluates `a < 100` (passes 10 %) first, like the same filter
    written the other way round.  Same rows either way (a row is kept iff every conjunct is TRUE: FilterOperator.kt:20), bit
    for bit against the written-order plan, the oracle on two windows, and with the ordering switched off (debug bit
    1048576).  Nullable inputs: a NULL conjunct drops the row whatever the order."""
    from queryengine_amd import workloads as W
    n = 9_000_017
    wl = W.config2(n, null_pct=1)
    A_, B_, C_ = col("a", 0, I64), col("b", 1, I64), col("c", 2, D)
    swapped = fn(Fn.AND, fn(Fn.CMP_LT, C_, num(0.5)), fn(Fn.CMP_LT, A_, num(100)))
    three = fn(Fn.AND, fn(Fn.AND, fn(Fn.CMP_LT, C_, num(0.9)), fn(Fn.CMP_GE, B_, num(0))), fn(Fn.CMP_LT, A_, num(50)))
    results = {}
    for name, tuning in (("ordered", []), ("as_written", [0, 0, 0, 0, 0, 1048576])):
        ctx = E.Context(device=0, tuning=tuning)
        batch = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], n)
        cp = [ctx.compile(p) for p in wl.projections]
        for label, flt, want_order in (("written", wl.filter, [0, 1]), ("swapped", swapped, [1, 0]), ("three", three, [2, 0, 1])):
            cf = ctx.compile(flt)
            assert E.conjunct_order(ctx, batch, cf, cp) is None
            res = E.filter_project(ctx, batch, cf, cp)
            order = E.conjunct_order(ctx, batch, cf, cp)
            if name == "ordered":
                assert order == want_order, (label, order)
            else:
                assert order is None
            again = E.filter_project(ctx, batch, cf, cp)
            cols, cols2 = res.to_columns(), again.to_columns()
            for g, w in zip(cols, cols2):
                assert_columns_equal(g, w, f"{name} {label}: second execution")
            results[name, label] = cols
            res.free(); again.free()
        if name == "ordered":
            m = 80_000
            for begin in (0, (n - 200_000) - (n - 200_000) % 64):
                host = [batch.column_to_host(j, begin, m) for j in range(batch.ncols)]
                want = oracle.filter_project(host, swapped, wl.projections, oracle.BYTECODE_COMPILER)
                nsel = 0
                if begin:   # kept rows in front of the window: an independent aggregate COUNT over rows [0, begin)
                    head = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], begin)
                    _, nsel = E.filter_aggregate(ctx, head, ctx.compile(swapped), [ctx.compile(A_)], [N.AGG_COUNT])
                    head.free()
                lo, k = int(nsel), len(want[0])
                for g, w in zip(results[name, "swapped"], want):
                    assert_columns_equal(Column(g.type, g.data[lo:lo + k], None if g.valid is None else g.valid[lo:lo + k]), w, f"window at {begin}")
        batch.free()
        ctx.close()
    for label in ("written", "swapped", "three"):
        for g, w in zip(results["ordered", label], results["as_written", label]):
            assert_columns_equal(g, w, f"{label}: ordered vs as written")
    for g, w in zip(results["ordered", "written"], results["ordered", "swapped"]):
        assert_columns_equal(g, w, "written vs swapped")


def test_two_rows_per_lane_output_stores(oracle):
    """Measurement switch tuning[2] % 10 == 5: a resolved chunk's rows leave the LDS ring two per lane (16-byte stores for
    8-byte columns, one row peeled when the chunk's first position is odd).  Same rows as the oracle around the sub-tile and
    chunk boundaries, odd / even chunk counts, nullable outputs and a 4-byte output column."""
    from queryengine_amd import workloads as W
    ctx = E.Context(device=0, tuning=[0, 0, 5, 0, 0, 1024 | 524288 | 32768, 0, 0])
    for n in (1, 2, 3, 1023, 1025, 16384, 16385, 40_001, 300_007):
        for null_pct in (0, 3):
            wl = W.config2(n, a_limit=300, null_pct=null_pct)
            batch = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], n, row_begin=777 * 64)
            cols = [batch.column_to_host(j) for j in range(batch.ncols)]
            batch.free()
            projs = list(wl.projections) + [fn(Fn.CMP_LT, col("c", 2, D), num(0.25))]
            run_both(ctx, oracle, cols, wl.filter, projs)
    assert ctx.last_form == N.FORM_RING
    ctx.close()


@pytest.mark.parametrize("bits", [4194304, 4194304 | 262144, 2097152])
def test_stage0_prefetch_forced_and_off(oracle, bits):
    """Staged plans issue the stage-0 loads of the NEXT sub-tile at the start of the current one (by default only plans with
    >= 4 load stages, e.g. the Q6 shape).  Forced for every staged plan (debug bit 4194304; with 262144 through the local
    form, whose prefetch also crosses chunk boundaries) and switched off (2097152): the same rows as the oracle over sizes
    around the sub-tile / chunk boundaries, nullable inputs (the validity bits of stage 0 travel with the prefetch) included."""
    from queryengine_amd import workloads as W
    ctx = E.Context(device=0, tuning=[0, 0, 0, 0, 0, bits, 0, 0])
    for n in (1, 1023, 1024, 1025, 2047, 2049, 16384, 16385, 40_001, 300_007):
        for null_pct in (0, 3):
            wl = W.config2(n, a_limit=300, null_pct=null_pct)
            batch = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], n, row_begin=12345 * 64)
            cols = [batch.column_to_host(j) for j in range(batch.ncols)]
            batch.free()
            run_both(ctx, oracle, cols, wl.filter, wl.projections)
    wl = W.config3(250_003)
    batch = E.DeviceBatch.generate(ctx, [c.spec(ctx) for c in wl.columns], wl.default_rows)
    cols = [batch.column_to_host(j) for j in range(batch.ncols)]
    batch.free()
    run_both(ctx, oracle, cols, wl.filter, wl.projections)
    ctx.close()

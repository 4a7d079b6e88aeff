// test_host.cpp -- the reference's evaluator tests (T/evaluator/CompilerTest.kt:11-122 and
// T/ByteCodeCompilerTest.kt:12-40) restated against the C++ host mirror, running on the GPU through
// the C ABI in both execution modes (the reference parameterises over its three Modes).
// Exit code 0 = all assertions hold.  Needs a GPU: there is no CPU path.
#include <cmath>
#include <cstdio>

#include "qe_host.hpp"

using namespace queryengine;

static int failures = 0;
#define EXPECT(cond, what)                                                     \
    do {                                                                       \
        if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, what); failures++; } \
    } while (0)

static std::string show(const Value &v) {
    if (isNull(v)) return "null";
    if (auto d = std::get_if<double>(&v)) return std::to_string(*d);
    if (auto b = std::get_if<bool>(&v)) return *b ? "true" : "false";
    if (auto s = std::get_if<std::string>(&v)) return *s;
    if (auto l = std::get_if<int64_t>(&v)) return std::to_string(*l);
    return std::to_string(std::get<int32_t>(v));
}

// evaluate `expr` on every row of `rows` (the analogue of compileExpression(expr, mode)(row))
static std::vector<Value> evalRows(std::shared_ptr<Context> ctx, const Schema &schema, const std::vector<Row> &rows, ExpressionPtr expr) {
    TableRegistry reg;
    reg.registerTable("t", std::make_shared<ColumnarTable>(schema, rows));
    auto plan = std::make_shared<LogicalProjectionNode>(std::make_shared<LogicalScanNode>("t", schema), std::vector<ExpressionPtr>{expr});
    auto op = buildPhysicalPlan(reg, plan, ctx);
    return map<Value>(*op, [](const Row &r) { return r[0]; });
}

static void runAll(std::shared_ptr<Context> ctx, const char *modeName) {
    const Value N = std::monostate{};
    const Value T = true, F = false;
    // `should load column` (CompilerTest.kt:15-23)
    {
        Schema s{{{"foo", DataType::STRING}}};
        auto got = evalRows(ctx, s, {{std::string("foobar")}}, col("foo", 0, DataType::STRING));
        EXPECT(got.size() == 1 && got[0] == Value(std::string("foobar")), "should load column");
    }
    // `should add numeric literals` (:27-35): evaluated once per row of a one-row table
    {
        Schema s{{{"x", DataType::DOUBLE}}};
        auto got = evalRows(ctx, s, {{0.0}}, fn(Function::ADD, {num(1.0), num(2.0)}, DataType::DOUBLE));
        EXPECT(got.size() == 1 && got[0] == Value(3.0), "should add numeric literals");
    }
    // `should handle null values` (:39-47)
    {
        Schema s{{{"foo", DataType::DOUBLE}, {"bar", DataType::DOUBLE}}};
        auto got = evalRows(ctx, s, {{10.0, N}}, fn(Function::MUL, {col("foo", 0, DataType::DOUBLE), col("bar", 1, DataType::DOUBLE)}, DataType::DOUBLE));
        EXPECT(got.size() == 1 && isNull(got[0]), "should handle null values");
    }
    // `should handle null in boolean and / or expressions` (:53-73, :79-99)
    {
        Schema s{{{"p", DataType::BOOLEAN}, {"q", DataType::BOOLEAN}}};
        std::vector<Row> rows = {{T, T}, {T, F}, {T, N}, {F, T}, {F, F}, {F, N}, {N, T}, {N, F}, {N, N}};
        std::vector<Value> andTruth = {T, F, N, F, F, F, N, F, N};
        std::vector<Value> orTruth = {T, T, T, T, F, N, T, N, N};
        auto p = col("p", 0, DataType::BOOLEAN), q = col("q", 1, DataType::BOOLEAN);
        auto a = evalRows(ctx, s, rows, fn(Function::AND, {p, q}, DataType::BOOLEAN));
        auto o = evalRows(ctx, s, rows, fn(Function::OR, {p, q}, DataType::BOOLEAN));
        for (size_t i = 0; i < rows.size(); i++) {
            EXPECT(a[i] == andTruth[i], ("Expected (" + show(rows[i][0]) + " AND " + show(rows[i][1]) + ") == " + show(andTruth[i]) + " got " + show(a[i])).c_str());
            EXPECT(o[i] == orTruth[i], ("Expected (" + show(rows[i][0]) + " OR " + show(rows[i][1]) + ") == " + show(orTruth[i]) + " got " + show(o[i])).c_str());
        }
    }
    // `should handle null in if expressions` (:105-119)
    {
        Schema s{{{"cond", DataType::BOOLEAN}}};
        auto got = evalRows(ctx, s, {{T}, {F}, {N}}, fn(Function::IF, {col("cond", 0, DataType::BOOLEAN), str("t"), str("f")}, DataType::STRING));
        EXPECT(got.size() == 3 && got[0] == Value(std::string("t")) && got[1] == Value(std::string("f")) && isNull(got[2]), "should handle null in if expressions");
    }
    // ByteCodeCompilerTest.kt:14-37: project [foo+bar, bar+baz] over three rows
    {
        Schema s{{{"foo", DataType::DOUBLE}, {"bar", DataType::DOUBLE}, {"baz", DataType::DOUBLE}}};
        TableRegistry reg;
        reg.registerTable("table", std::make_shared<ColumnarTable>(s, std::vector<Row>{{10.0, 11.0, 12.0}, {20.0, 21.0, 22.0}, {30.0, 31.0, 32.0}}));
        auto D = DataType::DOUBLE;
        auto plan = std::make_shared<LogicalProjectionNode>(std::make_shared<LogicalScanNode>("table", s), std::vector<ExpressionPtr>{
            fn(Function::ADD, {col("foo", 0, D), col("bar", 1, D)}), fn(Function::ADD, {col("bar", 1, D), col("baz", 2, D)})});
        auto op = buildPhysicalPlan(reg, plan, ctx);
        std::vector<Row> want = {{21.0, 23.0}, {41.0, 43.0}, {61.0, 63.0}};
        for (int pass = 0; pass < 2; pass++) {   // operators are re-openable (SimpleSumBenchmark.java:63-94)
            auto got = map<Row>(*op, [](const Row &r) { return r; });
            EXPECT(got == want, "compileProjection [foo+bar, bar+baz]");
        }
    }
    // Filter(Scan): keep iff non-null true; the scan row passes through (FilterOperator.kt:14-25)
    {
        Schema s{{{"a", DataType::INT64}, {"c", DataType::DOUBLE}}};
        std::vector<Row> rows = {{(int64_t)5, 0.25}, {(int64_t)500, 0.1}, {N, 0.1}, {(int64_t)7, N}, {(int64_t)99, 0.49}};
        TableRegistry reg;
        reg.registerTable("t", std::make_shared<ColumnarTable>(s, rows));
        auto flt = fn(Function::AND, {fn(Function::CMP_LT, {col("a", 0, DataType::INT64), num(100)}), fn(Function::CMP_LT, {col("c", 1, DataType::DOUBLE), num(0.5)})});
        auto plan = std::make_shared<LogicalFilterNode>(std::make_shared<LogicalScanNode>("t", s), flt);
        auto op = buildPhysicalPlan(reg, plan, ctx);
        auto got = map<Row>(*op, [](const Row &r) { return r; });
        std::vector<Row> want = {{(int64_t)5, 0.25}, {(int64_t)99, 0.49}};
        EXPECT(got == want, "Filter(Scan) keeps non-null true rows in order");
    }
    // type errors surface as TypeCheckException at plan time
    {
        bool thrown = false;
        try {
            Schema s{{{"a", DataType::DOUBLE}, {"p", DataType::BOOLEAN}}};
            evalRows(ctx, s, {{1.0, true}}, fn(Function::ADD, {col("a", 0, DataType::DOUBLE), col("p", 1, DataType::BOOLEAN)}));
        } catch (const TypeCheckException &) {
            thrown = true;
        }
        EXPECT(thrown, "Invalid operand types must throw TypeCheckException");
    }
    std::printf("mode %s: done\n", modeName);
}

// Aggregation on top of the path, fused mode.
static void runAggregates(std::shared_ptr<Context> ctx) {
    const Value N = std::monostate{};
    auto D = DataType::DOUBLE, S = DataType::STRING;
    // QueryTest.kt:15-30: SELECT bar, SUM(num), foo FROM table -> keys (bar, foo) in insertion order; the finish
    // projection [bar, SUM, foo] is a re-ordering of [bar, foo, SUM]
    {
        Schema s{{{"foo", S}, {"bar", S}, {"num", D}}};
        std::vector<Row> rows = {{std::string("a"), std::string("A"), 1.0}, {std::string("a"), std::string("B"), 2.0},
                                 {std::string("a"), std::string("B"), 3.0}, {std::string("b"), std::string("B"), 4.0},
                                 {std::string("b"), std::string("B"), N},   {std::string("c"), N, N}};
        TableRegistry reg;
        reg.registerTable("table", std::make_shared<ColumnarTable>(s, rows));
        auto inputs = std::make_shared<LogicalProjectionNode>(std::make_shared<LogicalScanNode>("table", s),
            std::vector<ExpressionPtr>{col("bar", 1, S), col("foo", 0, S), col("num", 2, D)});
        auto plan = std::make_shared<LogicalAggregationNode>(inputs, 2, std::vector<AggregationFunction>{AggregationFunction::SUM});
        auto op = buildPhysicalPlan(reg, plan, ctx);
        std::vector<Row> want = {{std::string("A"), std::string("a"), 1.0}, {std::string("B"), std::string("a"), 5.0},
                                 {std::string("B"), std::string("b"), 4.0}, {N, std::string("c"), N}};
        for (int pass = 0; pass < 2; pass++) {
            auto got = map<Row>(*op, [](const Row &r) { return r; });
            EXPECT(got == want, "QueryTest group by (bar, foo) with SUM(num), insertion order");
        }
    }
    // SimpleSumBenchmark.java:41-53: foo = bar = (double)(i / 1000), SUM(foo + 10 * bar): N = 1000 -> 0.0, N = 1e6 -> 5494500000.0
    for (int64_t n : {(int64_t)1000, (int64_t)1000000}) {
        Schema s{{{"foo", D}, {"bar", D}}};
        std::vector<Row> rows;
        rows.reserve((size_t)n);
        for (int64_t i = 0; i < n; i++) rows.push_back({(double)(i / 1000), (double)(i / 1000)});
        TableRegistry reg;
        reg.registerTable("table", std::make_shared<ColumnarTable>(s, rows));
        auto expr = fn(Function::ADD, {col("foo", 0, D), fn(Function::MUL, {num(10), col("bar", 1, D)})});
        auto inputs = std::make_shared<LogicalProjectionNode>(std::make_shared<LogicalScanNode>("table", s), std::vector<ExpressionPtr>{expr, expr, expr});
        auto plan = std::make_shared<LogicalAggregationNode>(inputs, 0, std::vector<AggregationFunction>{
            AggregationFunction::SUM, AggregationFunction::COUNT, AggregationFunction::MAX});
        auto op = buildPhysicalPlan(reg, plan, ctx);
        auto got = map<Row>(*op, [](const Row &r) { return r; });
        const double want = n == 1000 ? 0.0 : 5494500000.0;
        EXPECT(got.size() == 1 && got[0].size() == 3 && got[0][0] == Value(want) && got[0][1] == Value((int64_t)n) &&
               got[0][2] == Value((double)((n - 1) / 1000) * 11.0), "SimpleSumBenchmark SUM(foo + 10 * bar)");
    }
    // empty input: SUM => null, COUNT => 0 (Accumulators.kt:26-53)
    {
        Schema s{{{"x", D}}};
        TableRegistry reg;
        reg.registerTable("t", std::make_shared<ColumnarTable>(s, std::vector<Row>{{1.0}, {2.0}}));
        auto flt = fn(Function::CMP_GT, {col("x", 0, D), num(5)});
        auto inputs = std::make_shared<LogicalProjectionNode>(std::make_shared<LogicalFilterNode>(std::make_shared<LogicalScanNode>("t", s), flt),
                                                              std::vector<ExpressionPtr>{col("x", 0, D), col("x", 0, D)});
        auto plan = std::make_shared<LogicalAggregationNode>(inputs, 0, std::vector<AggregationFunction>{AggregationFunction::SUM, AggregationFunction::COUNT});
        auto got = map<Row>(*buildPhysicalPlan(reg, plan, ctx), [](const Row &r) { return r; });
        EXPECT(got.size() == 1 && isNull(got[0][0]) && got[0][1] == Value((int64_t)0), "empty aggregation: SUM null, COUNT 0");
    }
    std::printf("aggregates: done\n");
}

int main() {
    try {
        runAll(std::make_shared<Context>(0, Mode::GPU_FUSED), "GPU_FUSED");
        runAll(std::make_shared<Context>(0, Mode::GPU_PER_NODE), "GPU_PER_NODE");
        runAggregates(std::make_shared<Context>(0, Mode::GPU_FUSED));
    } catch (const std::exception &e) {
        std::printf("FAIL exception: %s\n", e.what());
        return 2;
    }
    std::printf(failures ? "%d FAILURES\n" : "all host tests passed (%d failures)\n", failures);
    return failures ? 1 : 0;
}

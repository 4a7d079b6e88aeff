// qe_host.hpp -- C++ host-side mirror of the reference's operator / planner interface on top of
// the C ABI (include/qe_hip.h).  The reference is compiled JVM code whose toolchain is absent from the
// build image, so this header (and its Python twin, queryengine_amd/*.py) is what stands in for the
// Kotlin side: same class names, argument meaning and error behaviour, so that the tests read like the
// reference's own (host/test_host.cpp follows T/evaluator/CompilerTest.kt).
//
//   ast/Expressions.kt:6-62      -> Expression, *LiteralExpression, ColumnExpression, FunctionExpression
//   ast/Functions.kt:7-22        -> Function (ordinals identical)
//   data/Schema.kt:3-13          -> DataType (+INT64, INT32), Field, Schema
//   data/MemoryTable.kt:7-19     -> ColumnarTable (the columnar scan leaf)
//   operator/Operators.kt:5-32   -> Operator, forEach, map
//   evaluator/LogicalPlan.kt:7-12, Planner.kt:30-63 -> Logical*Node, buildPhysicalPlan(…, Mode::GPU_*)
//
// There is no CPU evaluation here: every expression runs inside libqe_hip.so on the GPU.
#pragma once

#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

#include "../../include/qe_hip.h"

namespace queryengine {

enum class DataType { STRING = 0, DOUBLE = 1, BOOLEAN = 2, INT64 = 3, INT32 = 4 };

enum class Function {
    AND = 0, OR, IF, NOT, UNARY_MINUS, UNARY_PLUS, MUL, DIV, MOD, ADD, SUB, CMP_LT, CMP_LE, CMP_GE, CMP_GT, CMP_EQ, CMP_NE
};

enum class Mode { GPU_FUSED, GPU_PER_NODE };

// boxed nullable value: the analogue of Any?
using Value = std::variant<std::monostate, double, bool, std::string, int64_t, int32_t>;
using Row = std::vector<Value>;
inline bool isNull(const Value &v) { return std::holds_alternative<std::monostate>(v); }

struct TypeCheckException : std::runtime_error { using std::runtime_error::runtime_error; };

// ---- expressions -----------------------------------------------------------------------------------
struct Expression {
    virtual ~Expression() = default;
    virtual void serialize(std::vector<uint8_t> &out) const = 0;
};
using ExpressionPtr = std::shared_ptr<const Expression>;

namespace detail {
template <typename T> inline void put(std::vector<uint8_t> &out, T v) {
    uint8_t b[sizeof(T)];
    std::memcpy(b, &v, sizeof(T));
    out.insert(out.end(), b, b + sizeof(T));
}
}  // namespace detail

struct NumericLiteralExpression : Expression {
    double value;
    explicit NumericLiteralExpression(double v) : value(v) {}
    void serialize(std::vector<uint8_t> &out) const override { out.push_back(QE_OP_NUM_LITERAL); detail::put(out, value); }
};
struct BooleanLiteralExpression : Expression {
    bool value;
    explicit BooleanLiteralExpression(bool v) : value(v) {}
    void serialize(std::vector<uint8_t> &out) const override { out.push_back(QE_OP_BOOL_LITERAL); out.push_back(value ? 1 : 0); }
};
struct StringLiteralExpression : Expression {
    std::string value;
    explicit StringLiteralExpression(std::string v) : value(std::move(v)) {}
    void serialize(std::vector<uint8_t> &out) const override {
        out.push_back(QE_OP_STR_LITERAL);
        detail::put(out, (uint16_t)value.size());
        out.insert(out.end(), value.begin(), value.end());
    }
};
struct ColumnExpression : Expression {
    std::string name;
    int index;
    DataType dataType;
    ColumnExpression(std::string n, int i, DataType t) : name(std::move(n)), index(i), dataType(t) {}
    void serialize(std::vector<uint8_t> &out) const override {
        out.push_back(QE_OP_COLUMN);
        out.push_back((uint8_t)dataType);
        detail::put(out, (uint16_t)index);
    }
};
struct FunctionExpression : Expression {
    Function function;
    std::vector<ExpressionPtr> operands;
    std::optional<DataType> dataTypeNullable;
    FunctionExpression(Function f, std::vector<ExpressionPtr> ops, std::optional<DataType> t = std::nullopt)
        : function(f), operands(std::move(ops)), dataTypeNullable(t) {}
    void serialize(std::vector<uint8_t> &out) const override {
        for (const auto &op : operands) op->serialize(out);   // postfix: operands first
        out.push_back(QE_OP_FUNCTION);
        out.push_back((uint8_t)function);
        out.push_back(dataTypeNullable ? (uint8_t)*dataTypeNullable : 0xFF);
    }
};

inline std::vector<uint8_t> serialize(const Expression &e) {
    std::vector<uint8_t> out = {'Q', 'E', 'X', 1};
    e.serialize(out);
    return out;
}

inline ExpressionPtr col(std::string n, int i, DataType t) { return std::make_shared<ColumnExpression>(std::move(n), i, t); }
inline ExpressionPtr num(double v) { return std::make_shared<NumericLiteralExpression>(v); }
inline ExpressionPtr str(std::string v) { return std::make_shared<StringLiteralExpression>(std::move(v)); }
inline ExpressionPtr fn(Function f, std::vector<ExpressionPtr> ops, std::optional<DataType> t = std::nullopt) {
    return std::make_shared<FunctionExpression>(f, std::move(ops), t);
}

// ---- context (RAII over qe_ctx) ----------------------------------------------------------------------
class Context {
public:
    explicit Context(int device = 0, Mode mode = Mode::GPU_FUSED) {
        qe_options o{};
        o.struct_size = sizeof o;
        o.exec_mode = mode == Mode::GPU_FUSED ? QE_EXEC_FUSED : QE_EXEC_PER_NODE;
        if (int st = qe_ctx_create(device, &o, &ctx_)) throw std::runtime_error(std::string("qe_ctx_create: ") + qe_last_error(nullptr) + " (" + std::to_string(st) + ")");
    }
    ~Context() { qe_ctx_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    qe_ctx *get() const { return ctx_; }
    void check(int st) const {
        if (st == QE_OK) return;
        std::string msg = qe_last_error(ctx_);
        if (st == QE_ERR_PROGRAM) throw TypeCheckException(msg);
        if (st == QE_ERR_INVALID_ARG) throw std::invalid_argument(msg);
        throw std::runtime_error(msg);
    }

private:
    qe_ctx *ctx_ = nullptr;
};

// ---- data ---------------------------------------------------------------------------------------------
struct Field { std::string name; DataType type; };
struct Schema {
    std::vector<Field> fields;
    int indexOf(const std::string &n) const {
        for (size_t i = 0; i < fields.size(); i++) if (fields[i].name == n) return (int)i;
        return -1;
    }
};

inline std::vector<uint64_t> packBitmap(const std::vector<bool> &bits) {
    std::vector<uint64_t> w((bits.size() + 63) / 64, 0);
    for (size_t i = 0; i < bits.size(); i++) if (bits[i]) w[i >> 6] |= 1ull << (i & 63);
    return w;
}

// One column in the layouts of qe_col_desc.
struct Column {
    DataType type;
    std::vector<double> f64;
    std::vector<int64_t> i64;
    std::vector<int32_t> i32;        // INT32 values or STRING dictionary codes
    std::vector<uint64_t> bits;      // BOOLEAN value bitmap
    std::vector<uint64_t> validity;  // empty = all valid
    std::vector<std::string> dictionary;
    size_t size = 0;
};

// Build a column from boxed values (like one column of MemoryTable's row-major List<List<Any?>>).
inline Column columnFromValues(DataType t, const std::vector<Value> &vals) {
    Column c;
    c.type = t;
    c.size = vals.size();
    std::vector<bool> valid(vals.size()), bits(vals.size());
    bool anyNull = false;
    for (size_t i = 0; i < vals.size(); i++) {
        const Value &v = vals[i];
        valid[i] = !isNull(v);
        anyNull |= !valid[i];
        switch (t) {
        case DataType::DOUBLE: c.f64.push_back(valid[i] ? std::get<double>(v) : 0.0); break;
        case DataType::INT64: c.i64.push_back(valid[i] ? std::get<int64_t>(v) : 0); break;
        case DataType::INT32: c.i32.push_back(valid[i] ? std::get<int32_t>(v) : 0); break;
        case DataType::BOOLEAN: bits[i] = valid[i] && std::get<bool>(v); break;
        case DataType::STRING: {
            int code = 0;
            if (valid[i]) {
                const std::string &s = std::get<std::string>(v);
                code = -1;
                for (size_t k = 0; k < c.dictionary.size(); k++) if (c.dictionary[k] == s) code = (int)k;
                if (code < 0) { code = (int)c.dictionary.size(); c.dictionary.push_back(s); }
            }
            c.i32.push_back(code);
            break;
        }
        }
    }
    if (t == DataType::BOOLEAN) c.bits = packBitmap(bits);
    if (anyNull) c.validity = packBitmap(valid);
    return c;
}

class Operator;

// data/Table.kt:6-10
struct Table {
    virtual ~Table() = default;
    virtual const Schema &schema() const = 0;
};

// The columnar replacement of MemoryTable: constructor shape MemoryTable(schema, rows).
class ColumnarTable : public Table {
public:
    ColumnarTable(Schema s, const std::vector<Row> &rows) : schema_(std::move(s)), nrows_(rows.size()) {
        for (size_t j = 0; j < schema_.fields.size(); j++) {
            std::vector<Value> vals;
            for (const Row &r : rows) vals.push_back(r.at(j));
            columns_.push_back(columnFromValues(schema_.fields[j].type, vals));
        }
    }
    const Schema &schema() const override { return schema_; }
    size_t rowCount() const { return nrows_; }
    const Column &column(const std::string &name) const {
        int i = schema_.indexOf(name);
        if (i < 0) throw std::invalid_argument("Unknown field " + name);   // MemoryTable.kt:11
        return columns_[i];
    }

private:
    Schema schema_;
    std::vector<Column> columns_;
    size_t nrows_;
};

struct TableRegistry {
    std::vector<std::pair<std::string, std::shared_ptr<Table>>> tables;
    void registerTable(const std::string &n, std::shared_ptr<Table> t) { tables.emplace_back(n, std::move(t)); }
    std::shared_ptr<Table> getTable(const std::string &n) const {
        for (auto &kv : tables) if (kv.first == n) return kv.second;
        throw std::invalid_argument("Unknown table " + n);
    }
};

// ---- operators (operator/Operators.kt:5-32) ------------------------------------------------------------
class Operator {
public:
    virtual ~Operator() = default;
    virtual void open() = 0;
    virtual void close() = 0;
    virtual std::optional<Row> next() = 0;   // nullopt = exhausted
};

inline void forEach(Operator &op, const std::function<void(const Row &)> &consumer) {
    op.open();
    try {
        while (auto row = op.next()) consumer(*row);
    } catch (...) {
        op.close();
        throw;
    }
    op.close();
}
template <typename T> std::vector<T> map(Operator &op, const std::function<T(const Row &)> &mapper) {
    std::vector<T> out;
    forEach(op, [&](const Row &r) { out.push_back(mapper(r)); });
    return out;
}

// Projection(Filter(Scan)) fused into one GPU operator; replaces FilterOperator.kt:5-26 +
// ProjectionOperator.kt:5-21 / BytecodeCompiler.kt:37-132.  Re-openable; the batch is pinned once.
class GpuFilterProjectOperator : public Operator {
public:
    GpuFilterProjectOperator(std::shared_ptr<Context> ctx, std::shared_ptr<ColumnarTable> table, std::vector<std::string> projection,
                             ExpressionPtr filter, std::vector<ExpressionPtr> projections)
        : ctx_(std::move(ctx)), table_(std::move(table)), scan_(std::move(projection)) {
        if (filter) filter_ = compile(*filter);                 // compileExpression at plan time (Planner.kt:35,44)
        for (auto &p : projections) projs_.push_back(compile(*p));
    }
    ~GpuFilterProjectOperator() override {
        close();
        for (qe_expr *e : projs_) qe_expr_free(ctx_->get(), e);
        if (filter_) qe_expr_free(ctx_->get(), filter_);
        if (batch_) qe_batch_free(ctx_->get(), batch_);
        for (qe_dict *d : dicts_) qe_dict_free(ctx_->get(), d);
    }
    void open() override {
        close();
        if (!batch_) pin();
        ctx_->check(qe_filter_project(ctx_->get(), batch_, filter_, projs_.data(), (int32_t)projs_.size(), &result_));
        count_ = qe_result_count(result_);
        idx_ = 0;
        fetched_ = false;
    }
    std::optional<Row> next() override {
        if (!result_) throw std::logic_error("Operator not initialized");   // CsvSourceOperator.kt:49
        if (!fetched_) fetch();
        if (idx_ >= count_) return std::nullopt;
        Row row;
        for (auto &c : out_) row.push_back(c.box(idx_));       // a fresh row per call (ProjectionOperator.kt:18)
        idx_++;
        return row;
    }
    void close() override {
        if (result_) qe_result_free(ctx_->get(), result_);
        result_ = nullptr;
        out_.clear();
    }
    int64_t resultCount() const { return count_; }

protected:
    struct HostColumn {
        int type = 0;
        std::vector<uint8_t> data;
        std::vector<uint64_t> validity;
        std::vector<std::string> dict;
        Value box(int64_t i) const {
            if (!((validity[i >> 6] >> (i & 63)) & 1)) return std::monostate{};
            switch (type) {
            case QE_DOUBLE: { double v; std::memcpy(&v, data.data() + 8 * i, 8); return v; }
            case QE_INT64: { int64_t v; std::memcpy(&v, data.data() + 8 * i, 8); return v; }
            case QE_INT32: { int32_t v; std::memcpy(&v, data.data() + 4 * i, 4); return v; }
            case QE_STRING: { int32_t v; std::memcpy(&v, data.data() + 4 * i, 4); return dict.at(v); }
            default: { uint64_t w; std::memcpy(&w, data.data() + 8 * (i >> 6), 8); return (bool)((w >> (i & 63)) & 1); }
            }
        }
    };
    qe_expr *compile(const Expression &e) {
        auto prog = serialize(e);
        qe_expr *out = nullptr;
        ctx_->check(qe_expr_compile(ctx_->get(), prog.data(), prog.size(), &out));
        return out;
    }
    void pin() {
        std::vector<qe_col_desc> descs;
        for (const std::string &name : scan_) {
            const Column &c = table_->column(name);
            qe_col_desc d{};
            d.type = (int32_t)c.type;
            switch (c.type) {
            case DataType::DOUBLE: d.data = c.f64.data(); break;
            case DataType::INT64: d.data = c.i64.data(); break;
            case DataType::BOOLEAN: d.data = c.bits.data(); break;
            default: d.data = c.i32.data(); break;
            }
            static const uint64_t dummy = 0;
            if (!d.data) d.data = &dummy;
            d.validity = c.validity.empty() ? nullptr : c.validity.data();
            if (c.type == DataType::STRING) {
                std::vector<const char *> ptrs;
                for (auto &s : c.dictionary) ptrs.push_back(s.c_str());
                qe_dict *dict = nullptr;
                ctx_->check(qe_dict_create(ctx_->get(), (int32_t)ptrs.size(), ptrs.data(), &dict));
                dicts_.push_back(dict);
                d.dict = dict;
            }
            descs.push_back(d);
        }
        ctx_->check(qe_batch_create(ctx_->get(), (int64_t)table_->rowCount(), (int32_t)descs.size(), descs.data(), &batch_));
    }
    void fetch() {
        for (int32_t c = 0; c < qe_result_ncols(result_); c++) {
            qe_col_view v{};
            ctx_->check(qe_result_column(result_, c, &v));
            HostColumn hc;
            hc.type = v.type;
            const size_t words = (size_t)(count_ + 63) / 64;
            const size_t width = (v.type == QE_DOUBLE || v.type == QE_INT64) ? 8 : 4;
            hc.data.resize(std::max<size_t>(8, v.type == QE_BOOLEAN ? words * 8 : width * (size_t)count_));
            hc.validity.assign(std::max<size_t>(1, words), 0);
            ctx_->check(qe_result_column_to_host(ctx_->get(), result_, c, hc.data.data(), hc.validity.data()));
            if (v.type == QE_STRING)
                for (int32_t k = 0; k < qe_dict_size(v.dict); k++) hc.dict.emplace_back(qe_dict_entry(v.dict, k));
            out_.push_back(std::move(hc));
        }
        fetched_ = true;
    }

    std::shared_ptr<Context> ctx_;
    std::shared_ptr<ColumnarTable> table_;
    std::vector<std::string> scan_;
    qe_expr *filter_ = nullptr;
    std::vector<qe_expr *> projs_;
    std::vector<qe_dict *> dicts_;
    qe_batch *batch_ = nullptr;
    qe_result *result_ = nullptr;
    std::vector<HostColumn> out_;
    int64_t count_ = 0, idx_ = 0;
    bool fetched_ = false;
};

enum class AggregationFunction { MIN = 0, MAX = 1, SUM = 2, COUNT = 3, AVG = 4 };   // ast/Functions.kt:24-26

// Aggregation(Projection(Filter(Scan))) on the GPU: GlobalAggregationOperator.kt:7-36 when groupCount == 0 (one row,
// even over an empty input), GroupByAggregationOperator.kt:7-76 otherwise (one row [keys..., accumulators...] per group
// in insertion order).  The first `groupCount` input expressions are the keys.  COUNT finishes as the reference's Int
// (Accumulators.kt:26-36): an int64 here.
class GpuAggregationOperator : public GpuFilterProjectOperator {
public:
    GpuAggregationOperator(std::shared_ptr<Context> ctx, std::shared_ptr<ColumnarTable> table, std::vector<std::string> projection,
                           ExpressionPtr filter, std::vector<ExpressionPtr> inputs, int groupCount,
                           std::vector<AggregationFunction> functions)
        : GpuFilterProjectOperator(std::move(ctx), std::move(table), std::move(projection), std::move(filter), std::move(inputs)),
          groupCount_(groupCount) {
        for (auto f : functions) fns_.push_back((int32_t)f);
        if ((size_t)groupCount_ + fns_.size() != projs_.size()) throw std::invalid_argument("keys + aggregates != input expressions");
    }
    void open() override {
        close();
        if (!batch_) pin();
        const int32_t nagg = (int32_t)fns_.size();
        if (groupCount_ == 0) {
            std::vector<double> vals((size_t)std::max(1, nagg));
            std::vector<uint8_t> valid((size_t)std::max(1, nagg));
            int64_t selected = 0;
            ctx_->check(qe_filter_aggregate(ctx_->get(), batch_, filter_, projs_.data(), fns_.data(), nagg, vals.data(), valid.data(), &selected));
            Row row;
            for (int32_t i = 0; i < nagg; i++) row.push_back(finish(i, valid[i] != 0, vals[i]));
            rows_.assign(1, row);
        } else {
            ctx_->check(qe_filter_groupby(ctx_->get(), batch_, filter_, projs_.data(), groupCount_, projs_.data() + groupCount_, fns_.data(),
                                          nagg, &result_));
            count_ = qe_result_count(result_);
            fetch();
            rows_.clear();
            for (int64_t r = 0; r < count_; r++) {
                Row row;
                for (int k = 0; k < groupCount_; k++) row.push_back(out_[k].box(r));
                for (int32_t i = 0; i < nagg; i++) {
                    const Value v = out_[groupCount_ + i].box(r);
                    row.push_back(finish(i, !isNull(v), isNull(v) ? 0.0 : std::get<double>(v)));
                }
                rows_.push_back(std::move(row));
            }
            qe_result_free(ctx_->get(), result_);
            result_ = nullptr;
            out_.clear();
        }
        opened_ = true;
        idx_ = 0;
    }
    std::optional<Row> next() override {
        if (!opened_) throw std::logic_error("Operator not opened");   // GroupByAggregationOperator.kt:57
        if (idx_ >= (int64_t)rows_.size()) return std::nullopt;
        return rows_[(size_t)idx_++];
    }
    void close() override {
        GpuFilterProjectOperator::close();
        rows_.clear();
        opened_ = false;
    }

private:
    Value finish(int32_t i, bool valid, double v) const {
        if (fns_[i] == (int32_t)AggregationFunction::COUNT) return (int64_t)v;
        if (!valid) return std::monostate{};
        return v;
    }
    int groupCount_;
    std::vector<int32_t> fns_;
    std::vector<Row> rows_;
    bool opened_ = false;
};

// ---- logical plan + physical dispatch (LogicalPlan.kt:7-12, Planner.kt:30-63) ----------------------------
struct LogicalNode { virtual ~LogicalNode() = default; };
struct LogicalScanNode : LogicalNode {
    std::string table;
    Schema schema;
    LogicalScanNode(std::string t, Schema s) : table(std::move(t)), schema(std::move(s)) {}
};
struct LogicalFilterNode : LogicalNode {
    std::shared_ptr<LogicalNode> source;
    ExpressionPtr filter;
    LogicalFilterNode(std::shared_ptr<LogicalNode> s, ExpressionPtr f) : source(std::move(s)), filter(std::move(f)) {}
};
struct LogicalProjectionNode : LogicalNode {
    std::shared_ptr<LogicalNode> source;
    std::vector<ExpressionPtr> expressions;
    LogicalProjectionNode(std::shared_ptr<LogicalNode> s, std::vector<ExpressionPtr> e) : source(std::move(s)), expressions(std::move(e)) {}
};

struct LogicalAggregationNode : LogicalNode {   // LogicalPlan.kt:11; the source is the projection of keys + aggregate inputs
    std::shared_ptr<LogicalNode> source;
    int groupCount;
    std::vector<AggregationFunction> aggregateFunctions;
    LogicalAggregationNode(std::shared_ptr<LogicalNode> s, int g, std::vector<AggregationFunction> f)
        : source(std::move(s)), groupCount(g), aggregateFunctions(std::move(f)) {}
};

// Projection(Filter(Scan)) / Projection(Scan) / Filter(Scan) / Scan -> ONE fused GPU operator;
// Aggregation(Projection(Filter(Scan))) -> ONE fused aggregate / group-by operator (Planner.kt:48-57).
inline std::unique_ptr<Operator> buildPhysicalPlan(const TableRegistry &registry, const std::shared_ptr<LogicalNode> &planIn,
                                                   std::shared_ptr<Context> ctx) {
    std::shared_ptr<LogicalNode> plan = planIn;
    auto agg = std::dynamic_pointer_cast<LogicalAggregationNode>(plan);
    if (agg) {
        plan = agg->source;
        if (!std::dynamic_pointer_cast<LogicalProjectionNode>(plan)) throw std::logic_error("aggregation needs its input projection");
    }
    std::shared_ptr<LogicalNode> below = plan;
    std::vector<ExpressionPtr> projections;
    bool hasProjection = false;
    if (auto p = std::dynamic_pointer_cast<LogicalProjectionNode>(plan)) {
        projections = p->expressions;
        below = p->source;
        hasProjection = true;
    }
    ExpressionPtr filter;
    if (auto f = std::dynamic_pointer_cast<LogicalFilterNode>(below)) {
        filter = f->filter;
        below = f->source;
    }
    auto scan = std::dynamic_pointer_cast<LogicalScanNode>(below);
    if (!scan) throw std::logic_error("plan shape outside the GPU hot path");
    auto table = std::dynamic_pointer_cast<ColumnarTable>(registry.getTable(scan->table));
    if (!table) throw std::logic_error("the GPU modes need a columnar scan leaf (ColumnarTable)");
    std::vector<std::string> names;
    for (auto &f : scan->schema.fields) names.push_back(f.name);   // Planner.kt:32
    if (!hasProjection)   // FilterOperator returns the scan row itself (FilterOperator.kt:21)
        for (size_t i = 0; i < scan->schema.fields.size(); i++)
            projections.push_back(col(scan->schema.fields[i].name, (int)i, scan->schema.fields[i].type));
    if (agg)
        return std::make_unique<GpuAggregationOperator>(std::move(ctx), table, names, filter, projections, agg->groupCount,
                                                        agg->aggregateFunctions);
    return std::make_unique<GpuFilterProjectOperator>(std::move(ctx), table, names, filter, projections);
}

}  // namespace queryengine

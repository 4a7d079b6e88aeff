/*
 * qe_oracle.h -- CPU ORACLE for the filter/project hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's row-at-a-time evaluator; it
 * is the checker that the HIP path is compared against.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product (queryengine_amd/, libqe_hip.so) never links, imports or calls it.
 *
 * Reference files restated (paths under /root/reference/src/main/java/net/jhorstmann/queryengine/):
 *   evaluator/Interpreter.kt:7-114       INTERPRETER semantics (primary)
 *   evaluator/ClosureCompiler.kt:65-132  CLOSURE_COMPILER differences (IEEE <,<=,>=,>; NOT(null)=null)
 *   evaluator/BytecodeCompiler.kt:286-322,324-506  BYTECODE_COMPILER (compare()-based comparisons)
 *   operator/FilterOperator.kt:14-25     keep iff result != null && true, order preserving
 *   operator/ProjectionOperator.kt:15-19 one output value per expression per row
 *   operator/MemorySourceOperator.kt:18-36  scan: copy projected values into ONE reused row buffer
 *   operator/GlobalAggregationOperator.kt:7-36 + evaluator/Accumulators.kt:5-107  (row f1 of SURVEY 8f)
 *
 * Pinning: the reference cannot be compiled here (Kotlin/JVM, no JDK in the
 * image -- SURVEY.md 8c), so the oracle is pinned by the golden vectors the
 * reference's own tests hold (tests/golden/reference_vectors.json, checked by
 * tests/test_oracle_golden.py).  Operators those tests do not cover
 * (comparisons, SUB/DIV/MOD/NEG, NaN/-0.0) are pinned only by the JLS
 * definitions they invoke: "parity unpinned by reference fixtures" for those.
 * INT64/INT32 are build-defined extensions (Java long/int semantics).
 */
#ifndef QE_ORACLE_H
#define QE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* data types: ordinals of data/Schema.kt:3-5 plus extensions */
enum { QO_STRING = 0, QO_DOUBLE = 1, QO_BOOLEAN = 2, QO_INT64 = 3, QO_INT32 = 4 };

/* evaluator modes: evaluator/Compiler.kt:5-7 */
enum { QO_INTERPRETER = 0, QO_CLOSURE_COMPILER = 1, QO_BYTECODE_COMPILER = 2 };

/* node kinds: ast/Expressions.kt */
enum { QO_COLUMN = 0, QO_NUMERIC_LITERAL = 1, QO_BOOLEAN_LITERAL = 2, QO_STRING_LITERAL = 3, QO_FUNCTION = 4 };

/* functions: ordinals of ast/Functions.kt:7-22 */
enum {
    QO_AND = 0, QO_OR, QO_IF, QO_NOT, QO_UNARY_MINUS, QO_UNARY_PLUS, QO_MUL, QO_DIV, QO_MOD, QO_ADD, QO_SUB,
    QO_CMP_LT, QO_CMP_LE, QO_CMP_GE, QO_CMP_GT, QO_CMP_EQ, QO_CMP_NE
};

/* aggregation functions: ast/Functions.kt:24-26 */
enum { QO_MIN = 0, QO_MAX, QO_SUM, QO_COUNT, QO_AVG };

/* status */
enum { QO_OK = 0, QO_THROWN = 1 /* the reference would throw (ClassCast / NPE) */, QO_BAD_ARG = 2 };

/* flat expression tree; children are indices into the same array */
typedef struct {
    int32_t kind;
    int32_t fn;
    int32_t dtype;     /* static type of the node (ColumnExpression.dataType etc.) */
    int32_t col;       /* QO_COLUMN: index into the scan row */
    int32_t nops;
    int32_t ops[3];
    double num;        /* QO_NUMERIC_LITERAL */
    int32_t bval;      /* QO_BOOLEAN_LITERAL */
    int32_t pad;
    const char *str;   /* QO_STRING_LITERAL (UTF-8, NUL terminated) */
} qo_node;

/* boxed nullable value: the analogue of Any? */
enum { QO_T_NULL = 0, QO_T_F64 = 1, QO_T_BOOL = 2, QO_T_STR = 3, QO_T_I64 = 4, QO_T_I32 = 5 };
typedef struct {
    int32_t tag;
    int32_t pad;
    union { double d; int32_t b; int64_t l; int32_t i; const char *s; } u;
} qo_value;

/* columnar view of the table the scan leaf walks (the oracle boxes each row) */
typedef struct {
    int32_t dtype;
    int32_t pad;
    const void *data;          /* f64 / i64 / i32 / u8 (BOOLEAN: 0|1) / i32 codes (STRING) */
    const uint8_t *valid;      /* one byte per row, 0 = null; NULL = all valid */
    const char *const *dict;   /* STRING: code -> UTF-8 string */
} qo_column;

typedef struct {
    int32_t dtype;
    int32_t pad;
    void *data;                /* capacity nrows: f64 / i64 / i32 / u8 / const char* */
    uint8_t *valid;            /* capacity nrows, one byte per row */
} qo_out_column;

/* evaluate one expression on one boxed row (RowCallable.invoke, Compiler.kt:9-11) */
int32_t qo_eval(const qo_node *nodes, int32_t root, const qo_value *row, int32_t mode, qo_value *out);

/* Projection(Filter(Scan)): filter_root < 0 means no filter.  Returns the
 * number of output rows, or -1 when the reference would have thrown (*err set). */
int64_t qo_filter_project(const qo_node *nodes, int32_t filter_root, const int32_t *proj_roots, int32_t nproj,
                          const qo_column *cols, int32_t ncols, int64_t nrows, int32_t mode,
                          qo_out_column *outs, int32_t *err);

/* GlobalAggregation(Projection(Filter(Scan))): one accumulator per expression
 * (Accumulators.kt); values accumulated sequentially in row order, nulls
 * skipped, empty => null (COUNT => 0). Results as doubles + null flags. */
int64_t qo_filter_aggregate(const qo_node *nodes, int32_t filter_root, const int32_t *expr_roots,
                            const int32_t *agg_fns, int32_t nagg,
                            const qo_column *cols, int32_t ncols, int64_t nrows, int32_t mode,
                            double *out_values, uint8_t *out_valid, int32_t *err);

/* GroupByAggregation(Projection(Filter(Scan))) (operator/GroupByAggregationOperator.kt:7-76): the first nkeys
 * expressions are the group key (boxed values compared with equals(): null == null, Double by bits with NaNs
 * collapsed), the following nagg expressions feed one accumulator each (Accumulators.kt).  Groups come out in
 * INSERTION order (LinkedHashMap, :22), which T/evaluator/QueryTest.kt:25-30 pins.
 * out_keys[k] has capacity max_groups rows; out_values / out_valid are [group][agg] row-major.
 * Returns the number of groups or -1 (*err set; QO_BAD_ARG also when max_groups is exceeded). */
int64_t qo_filter_groupby(const qo_node *nodes, int32_t filter_root, const int32_t *key_roots, int32_t nkeys,
                          const int32_t *expr_roots, const int32_t *agg_fns, int32_t nagg,
                          const qo_column *cols, int32_t ncols, int64_t nrows, int32_t mode, int64_t max_groups,
                          qo_out_column *out_keys, double *out_values, uint8_t *out_valid, int32_t *err);

/* synthetic column generator; same formulas as the device generator */
enum { QO_GEN_I64_MOD = 0, QO_GEN_I32_MOD = 1, QO_GEN_F64_UNIT = 2, QO_GEN_F64_MOD = 3, QO_GEN_F64_STEP = 4,
       QO_GEN_F64_PRICE = 5, QO_GEN_I64_ROWID = 7 };
typedef struct {
    int32_t kind;
    int32_t col_id;     /* stream id of this column */
    uint64_t modulus;
    int64_t offset;
    double step;
    int32_t aux_col_id; /* QO_GEN_F64_PRICE: stream id of the quantity column */
    int32_t null_pct;   /* 0 = not nullable */
} qo_gen_spec;

uint64_t qo_gen_raw(uint64_t seed, int32_t col_id, uint64_t row);
void qo_generate(const qo_gen_spec *spec, uint64_t seed, int64_t row_begin, int64_t nrows, void *data, uint8_t *valid);

/* helpers exposed for tests */
int32_t qo_double_compare(double a, double b);   /* java.lang.Double.compare */
int32_t qo_double_equals(double a, double b);    /* java.lang.Double.equals */

#ifdef __cplusplus
}
#endif
#endif

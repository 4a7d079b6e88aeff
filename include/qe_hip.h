/*
 * qe_hip.h -- C ABI of libqe_hip.so: the MI355X (gfx950) drop-in for the
 * Filter/Projection hot path of jhorstmann/queryengine.
 *
 * The reference has no native boundary of its own: its seam is
 * the Kotlin operator API.  Each entry point below names the reference
 * interface it sits beneath (paths relative to
 * /root/reference/src/main/java/net/jhorstmann/queryengine/); INTEGRATION.md
 * shows the Panama/JNI binding a maintainer would add on the Kotlin side.
 *
 * Conventions: plain C, no C++ types, no exceptions across the boundary.
 * Every call returns int32 status (QE_OK = 0); qe_last_error(ctx) gives the
 * message of the last failing call on that context.  The caller owns host
 * buffers it passes in; the library owns device memory and result objects
 * until the matching *_free.  One qe_ctx = one device + one HIP stream; a
 * context is NOT thread-safe, distinct contexts are independent (the
 * reference runs one thread per plan: operator/Operators.kt:5-32).
 */
#ifndef QE_HIP_H
#define QE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QE_ABI_VERSION 1

/* ---- status codes ---------------------------------------------------------- */
enum {
    QE_OK = 0,
    QE_ERR_INVALID_ARG = 1,   /* IllegalArgumentException analogue */
    QE_ERR_PROGRAM = 2,       /* malformed / ill-typed expression program (TypeCheckException analogue) */
    QE_ERR_HIP = 3,           /* HIP runtime / hiprtc failure, message carries the HIP error string */
    QE_ERR_OOM = 4,
    QE_ERR_UNSUPPORTED = 5,
    QE_ERR_INTERNAL = 6,      /* includes a look-back spin that hit its bound */
    QE_ERR_COMM = 7           /* RCCL failure */
};

/* ---- data types: data/Schema.kt:3-5 ordinals + extensions ------------------ */
enum {
    QE_STRING = 0,    /* int32 dictionary codes + qe_dict */
    QE_DOUBLE = 1,    /* f64 */
    QE_BOOLEAN = 2,   /* value bitmap: row i = word i>>6, bit i&63 (LSB first) */
    QE_INT64 = 3,     /* extension: Java long semantics */
    QE_INT32 = 4      /* extension: Java int semantics */
};

/* ---- expression program ----------------------------------------------------
 * An Expression tree (ast/Expressions.kt:6-62) is serialised in POSTFIX order
 * (operands first), little endian:
 *
 *   header : 'Q' 'E' 'X' <version=1>
 *   QE_OP_COLUMN       u8 op, u8 type, u16 index          ColumnExpression(name, index, dataType) :60-62
 *   QE_OP_NUM_LITERAL  u8 op, f64 value                   NumericLiteralExpression :17-21 (always DOUBLE)
 *   QE_OP_BOOL_LITERAL u8 op, u8 value                    BooleanLiteralExpression :23-27
 *   QE_OP_STR_LITERAL  u8 op, u16 nbytes, UTF-8 bytes     StringLiteralExpression :29-33
 *   QE_OP_FUNCTION     u8 op, u8 function, u8 type        FunctionExpression :36-45; function = Function.ordinal
 *                                                         (ast/Functions.kt:7-22), type = dataTypeNullable or 0xFF
 *
 * qe_expr_compile is the analogue of compileExpression (evaluator/Compiler.kt:20-26)
 * plus the bytecode verifier pass (BytecodeCompiler.kt:138, MaxStackVisitor :177-196):
 * it checks stack discipline, arity and operand types and infers result types.
 */
enum { QE_OP_COLUMN = 1, QE_OP_NUM_LITERAL = 2, QE_OP_BOOL_LITERAL = 3, QE_OP_STR_LITERAL = 4, QE_OP_FUNCTION = 16 };

/* ast/Functions.kt:7-22 ordinals */
enum {
    QE_FN_AND = 0, QE_FN_OR, QE_FN_IF, QE_FN_NOT, QE_FN_UNARY_MINUS, QE_FN_UNARY_PLUS, QE_FN_MUL, QE_FN_DIV,
    QE_FN_MOD, QE_FN_ADD, QE_FN_SUB, QE_FN_CMP_LT, QE_FN_CMP_LE, QE_FN_CMP_GE, QE_FN_CMP_GT, QE_FN_CMP_EQ,
    QE_FN_CMP_NE, QE_FN_COUNT_
};

/* ast/Functions.kt:24-26 ordinals (ANY/ALL are TODO() in the reference: Accumulators.kt:16-17) */
enum { QE_AGG_MIN = 0, QE_AGG_MAX = 1, QE_AGG_SUM = 2, QE_AGG_COUNT = 3, QE_AGG_AVG = 4 };

/* ---- options ---------------------------------------------------------------- */
enum {
    QE_EXEC_FUSED = 0,     /* one single-pass fused kernel per plan, JIT-specialised with hiprtc
                              (the on-device analogue of Mode.BYTECODE_COMPILER / compileProjection,
                              BytecodeCompiler.kt:37-132) */
    QE_EXEC_PER_NODE = 1   /* one precompiled kernel per expression node, filter first
                              (the analogue of Mode.INTERPRETER's tree walk, Interpreter.kt:29-109) */
};
enum {
    QE_CMP_TOTAL_ORDER = 0, /* java.lang.Double.compare: INTERPRETER + BYTECODE_COMPILER */
    QE_CMP_IEEE = 1         /* primitive <,<=,>=,>: CLOSURE_COMPILER (ClosureCompiler.kt:127-130) */
};

typedef struct {
    uint32_t struct_size;        /* = sizeof(qe_options) */
    int32_t exec_mode;           /* QE_EXEC_* */
    int32_t cmp_semantics;       /* QE_CMP_* */
    int32_t profile;             /* 1: bracket the dominant kernel with HIP events (qe_ctx_kernel_time) */
    int64_t result_capacity_rows;/* 0 = size result buffers for the worst case (every row kept) */
    const char *jit_cache_dir;   /* NULL = $QE_JIT_CACHE_DIR or <library dir>/jit_cache */
    int32_t tuning[8];           /* kernel tuning knobs, 0 = default; see DESIGN.md */
} qe_options;

typedef struct qe_ctx qe_ctx;
typedef struct qe_dict qe_dict;
typedef struct qe_batch qe_batch;
typedef struct qe_expr qe_expr;
typedef struct qe_result qe_result;

/* ---- context ---------------------------------------------------------------- */
/* device = QE_DEVICE_NONE gives a PLANNING-ONLY context: expressions can be
 * compiled and verified, plans generated and JIT-compiled into the cache (hiprtc
 * needs no GPU), but every call that would touch device memory fails with
 * QE_ERR_HIP -- there is no CPU execution path in this library. */
#define QE_DEVICE_NONE (-1)
int32_t qe_abi_version(void);
/* last error of a failed qe_ctx_create (ctx == NULL) or of ctx */
const char *qe_last_error(const qe_ctx *ctx);
int32_t qe_ctx_create(int32_t device, const qe_options *opts, qe_ctx **out);
void qe_ctx_destroy(qe_ctx *ctx);
int32_t qe_ctx_set_exec_mode(qe_ctx *ctx, int32_t exec_mode);
int32_t qe_ctx_set_cmp_semantics(qe_ctx *ctx, int32_t cmp_semantics);
/* HIP-event time of the dominant kernel (needs opts.profile): last launch, sum and count since reset */
int32_t qe_ctx_kernel_time(qe_ctx *ctx, double *last_ms, double *total_ms, int64_t *launches);
int32_t qe_ctx_reset_kernel_time(qe_ctx *ctx);
int32_t qe_ctx_synchronize(qe_ctx *ctx);
/* release cached device buffers back to the driver */
int32_t qe_ctx_trim(qe_ctx *ctx);

/* ---- dictionaries (STRING columns) -------------------------------------------- */
int32_t qe_dict_create(qe_ctx *ctx, int32_t nentries, const char *const *utf8, qe_dict **out);
int32_t qe_dict_size(const qe_dict *dict);
const char *qe_dict_entry(const qe_dict *dict, int32_t code);
void qe_dict_free(qe_ctx *ctx, qe_dict *dict);

/* ---- batches: the columnar scan leaf --------------------------------------------
 * Replaces MemoryTable / MemorySourceOperator (data/MemoryTable.kt:7-19,
 * operator/MemorySourceOperator.kt:5-36) beneath Table.getScanOperator
 * (data/Table.kt:8): columns are copied to HBM ONCE and stay resident across
 * repeated open()/close() of the operator (T/SimpleSumBenchmark.java:63-94). */
typedef struct {
    int32_t type;              /* QE_* data type */
    int32_t reserved;
    const void *data;          /* nrows elements (BOOLEAN: ceil(nrows/64) uint64 words) */
    const uint64_t *validity;  /* ceil(nrows/64) words, bit = 1 valid; NULL = all valid */
    const qe_dict *dict;       /* QE_STRING only */
} qe_col_desc;

/* host buffers -> device (H2D once) */
int32_t qe_batch_create(qe_ctx *ctx, int64_t nrows, int32_t ncols, const qe_col_desc *cols, qe_batch **out);
/* data/validity already are device pointers owned by the caller (zero copy) */
int32_t qe_batch_wrap_device(qe_ctx *ctx, int64_t nrows, int32_t ncols, const qe_col_desc *cols, qe_batch **out);

/* schema only (types, nullability = validity != NULL, dictionaries; data ignored): for plan-time
 * preparation (qe_filter_project_prepare / _source) without device memory */
int32_t qe_batch_describe(qe_ctx *ctx, int64_t nrows, int32_t ncols, const qe_col_desc *cols, qe_batch **out);

/* synthetic columns generated on the device from the GLOBAL row index;
 * avoids a 24 GB H2D for the 1 B-row configurations and makes shards reproducible */
enum { QE_GEN_I64_MOD = 0, QE_GEN_I32_MOD = 1, QE_GEN_F64_UNIT = 2, QE_GEN_F64_MOD = 3, QE_GEN_F64_STEP = 4,
       QE_GEN_F64_PRICE = 5, QE_GEN_DICT_MOD = 6,
       QE_GEN_I64_ROWID = 7 /* value = global row index: order-preservation checks */ };
typedef struct {
    int32_t kind;
    int32_t col_id;       /* random stream id */
    uint64_t modulus;
    int64_t offset;
    double step;
    int32_t aux_col_id;
    int32_t null_pct;     /* 0 = no validity bitmap */
    const qe_dict *dict;  /* QE_GEN_DICT_MOD */
} qe_gen_spec;
int32_t qe_batch_generate(qe_ctx *ctx, uint64_t seed, int64_t row_begin, int64_t nrows, int32_t ncols,
                          const qe_gen_spec *specs, qe_batch **out);

int64_t qe_batch_nrows(const qe_batch *batch);
int32_t qe_batch_ncols(const qe_batch *batch);
int32_t qe_batch_column_type(const qe_batch *batch, int32_t col);
/* device -> host copy of rows [row_begin, row_begin + nrows); row_begin must be a multiple of 64.
 * validity_out may be NULL; if the column has no validity bitmap it is filled with ones */
int32_t qe_batch_column_to_host(qe_ctx *ctx, const qe_batch *batch, int32_t col, int64_t row_begin, int64_t nrows,
                                void *data_out, uint64_t *validity_out);
void qe_batch_free(qe_ctx *ctx, qe_batch *batch);

/* ---- CSV text -> columns ------------------------------------------------------------------
 * The on-disk step in front of the path: replaces the row-at-a-time CSV scan leaves CsvTable / CsvSourceOperator
 * (data/CsvTable.kt:12-29, operator/CsvSourceOperator.kt:52-76) and UnivocityCsvTable / UnivocityCsvScanOperator
 * (data/UnivocityCsvTable.kt:10-69) beneath Table.getScanOperator (data/Table.kt:8).  Same conversion rules (first
 * record = header, fields located by header name, empty lines skipped, missing or empty field = NULL, String.toBoolean,
 * String.toDouble = java.lang.Double.parseDouble; a malformed number is QE_ERR_INVALID_ARG with the reference's
 * NumberFormatException text), but the result is one contiguous array per projected field in the qe_col_desc layout
 * (+ validity bitmap, + a dictionary in order of first appearance for STRING) that qe_csv_pin copies to HBM once.
 * types[] may hold QE_STRING, QE_DOUBLE, QE_BOOLEAN (data/Schema.kt:3-5).  Host-side, needs no device. */
typedef struct qe_csv_table qe_csv_table;
int32_t qe_csv_parse(qe_ctx *ctx, const char *utf8, size_t nbytes, int32_t nfields, const char *const *names,
                     const int32_t *types, qe_csv_table **out);
int32_t qe_csv_parse_file(qe_ctx *ctx, const char *path, int32_t nfields, const char *const *names, const int32_t *types,
                          qe_csv_table **out);
int64_t qe_csv_nrows(const qe_csv_table *table);
int32_t qe_csv_ncols(const qe_csv_table *table);
/* host view of one parsed column (pointers owned by the table; validity == NULL: no NULL in the column) */
int32_t qe_csv_column(const qe_csv_table *table, int32_t col, qe_col_desc *out);
/* = qe_batch_create on the table's columns ("pin to HBM once"); the batch keeps the dictionaries alive */
int32_t qe_csv_pin(qe_ctx *ctx, const qe_csv_table *table, qe_batch **out);
void qe_csv_free(qe_ctx *ctx, qe_csv_table *table);

/* ---- expressions ------------------------------------------------------------------ */
/* compileExpression(expression, mode): evaluator/Compiler.kt:20-26 */
int32_t qe_expr_compile(qe_ctx *ctx, const uint8_t *program, size_t len, qe_expr **out);
int32_t qe_expr_result_type(const qe_expr *expr);
void qe_expr_free(qe_ctx *ctx, qe_expr *expr);

/* ---- the hot path ------------------------------------------------------------------
 * Projection(Filter(Scan)) in one call: FilterOperator.next (operator/FilterOperator.kt:14-25)
 * + ProjectionOperator.next (operator/ProjectionOperator.kt:15-19) /
 * CompiledProjectionOperator (BytecodeCompiler.kt:37-132) for every row of the
 * batch, order preserving.  filter may be NULL (no Filter node).  The analogue of
 * the Filter/Projection branches of buildPhysicalPlan (evaluator/Planner.kt:33-46). */
int32_t qe_filter_project(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter,
                          const qe_expr *const *projections, int32_t nproj, qe_result **out);
/* plan-time preparation only (JIT compile + cache), no execution: what buildPhysicalPlan does */
int32_t qe_filter_project_prepare(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter,
                                  const qe_expr *const *projections, int32_t nproj);

/* GlobalAggregation(Projection(Filter(Scan))):
 * GlobalAggregationOperator.open (operator/GlobalAggregationOperator.kt:10-25) with
 * Accumulators.kt:26-107 semantics: nulls skipped, empty => null, COUNT => count.
 * SUM/AVG use a fixed-shape tree reduction (deterministic, not the reference's
 * sequential order: see DESIGN.md for the tolerance). */
int32_t qe_filter_aggregate(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter,
                            const qe_expr *const *exprs, const int32_t *agg_fns, int32_t nagg,
                            double *out_values, uint8_t *out_valid, int64_t *out_selected_rows);

/* GroupByAggregation(Projection(Filter(Scan))): GroupByAggregationOperator.open
 * (operator/GroupByAggregationOperator.kt:21-49).  The result has nkeys key columns followed by nagg DOUBLE
 * aggregate columns (NULL for an empty MIN/MAX/SUM/AVG, COUNT as a double), one row per group, in INSERTION
 * order of the groups (LinkedHashMap, :22; pinned by T/evaluator/QueryTest.kt:25-30); NULL is a key value.
 * Keys are expressions of any type -- the reference groups on the boxed key tuple (:33-37; Tripdata.kt:27-31 groups by a
 * DOUBLE column): STRING (dictionary) / BOOLEAN keys with at most 2^20 combinations index a dense table; DOUBLE / INT64 /
 * INT32 keys (Double.equals: all NaNs one group, -0.0 and 0.0 two) and larger combinations are hashed.
 * SUM/AVG use native f64 atomics: exact when every partial sum is representable, otherwise order dependent in the last
 * bits. */
int32_t qe_filter_groupby(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter,
                          const qe_expr *const *keys, int32_t nkeys,
                          const qe_expr *const *exprs, const int32_t *agg_fns, int32_t nagg, qe_result **out);

int32_t qe_filter_groupby_prepare(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter,
                                  const qe_expr *const *keys, int32_t nkeys,
                                  const qe_expr *const *exprs, const int32_t *agg_fns, int32_t nagg);

/* plan-time preparation of the aggregate plan (JIT compile + cache), no execution */
int32_t qe_filter_aggregate_prepare(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter,
                                    const qe_expr *const *exprs, const int32_t *agg_fns, int32_t nagg);

/* ---- results --------------------------------------------------------------------------- */
typedef struct {
    int32_t type;
    int32_t nullable;
    const void *data;          /* DEVICE pointer: count elements (BOOLEAN: bitmap words) */
    const uint64_t *validity;  /* DEVICE pointer or NULL */
    int64_t count;
    const qe_dict *dict;       /* QE_STRING: dictionary of the output codes (owned by the result) */
} qe_col_view;

int64_t qe_result_count(const qe_result *result);
int32_t qe_result_ncols(const qe_result *result);
int32_t qe_result_column(const qe_result *result, int32_t col, qe_col_view *out);
/* copy one output column to the CALLER's host buffers sized for qe_result_count rows (pageable memory is fine: the bytes
 * are staged through pinned chunks and copied out by a few host threads while the next chunk is on the link) */
int32_t qe_result_column_to_host(qe_ctx *ctx, const qe_result *result, int32_t col, void *data_out,
                                 uint64_t *validity_out);
void qe_result_free(qe_ctx *ctx, qe_result *result);

/* Result -> host, the way a caller that materialises rows wants it (Main.kt:18 `physicalPlan.map { it }`;
 * operator/Operators.kt:5-11 hands out host rows): every column is copied into PINNED host memory owned by the library
 * (pooled per context: a re-opened operator does not pin again) on the context's COPY stream.  qe_result_to_host only
 * starts the copies and returns -- the next qe_filter_project (compute stream) runs beside them; qe_host_result_wait blocks
 * until the bytes are there; qe_host_result_column then gives HOST pointers in the qe_col_view (same layouts as on the
 * device; validity NULL = no NULL in the column); qe_host_result_free returns the buffers to the pool.  `result` must stay
 * alive until the wait has returned (qe_result_free waits for a copy that still reads it).  Into pinned memory the link
 * runs at its own rate (0.8 GB: ~15 ms) where a copy into pageable memory (qe_result_column_to_host) is bound by host
 * memcpy; the kernel itself is not overlapped with the copy of ITS OWN result: it takes 3 ms, the copy 15. */
typedef struct qe_host_result qe_host_result;
int32_t qe_result_to_host(qe_ctx *ctx, const qe_result *result, qe_host_result **out);
int32_t qe_host_result_wait(qe_ctx *ctx, qe_host_result *host);
int64_t qe_host_result_count(const qe_host_result *host);
int32_t qe_host_result_ncols(const qe_host_result *host);
int32_t qe_host_result_column(const qe_host_result *host, int32_t col, qe_col_view *out);
void qe_host_result_free(qe_ctx *ctx, qe_host_result *host);
/* Concatenate results of ONE device, in the given order, into a new result (value columns at row offsets, bitmap
 * columns shifted into place as 64-row words): what a host does that feeds a table batch by batch and still hands the
 * reference's single Operator (operator/Operators.kt:5-11) the whole result.  Parts must share schema and dictionaries. */
int32_t qe_result_concat(qe_ctx *ctx, const qe_result *const *parts, int32_t nparts, qe_result **out);

/* ORDER BY <column> of a materialised result, on the device: OrderByOperator.open (operator/OrderByOperator.kt:9-15) sorts
 * the rows STABLY with Kotlin's compareValues -- NULL first, Double.compareTo (-0.0 < 0.0, NaN greatest), String.compareTo
 * (UTF-16 code units), false < true.  `column` is 0-based (the planner's ORDER BY <ordinal> is 1-based: Planner.kt:60). */
int32_t qe_result_order_by(qe_ctx *ctx, const qe_result *result, int32_t column, qe_result **out);

/* ---- the exchange step of a row-range sharded scan -------------------------------------------------------
 * One process (one qe_ctx) per GPU; rank r scans rows [r*N/P, (r+1)*N/P) with NO communication.  Only a plan whose root
 * materialises its result on one rank (evaluator/Planner.kt:30-63: ONE Operator yields the whole result; Main.kt:18
 * `physicalPlan.map { it }`) exchanges data.  RCCL directly: ncclAllGather of the per-rank result headers, then one
 * ncclGroupStart/End of ncclRecv (root, at the final offsets) / ncclSend (peers), every peer on its own xGMI link;
 * rank order = the reference's row order (operator/FilterOperator.kt:17-22 is order preserving).
 * Bootstrap: rank 0 calls qe_comm_unique_id and hands the 128 bytes to the other ranks through whatever channel the
 * host has (the JVM host: its own RPC; tests / bench.py: torch.distributed broadcast); then every rank calls
 * qe_comm_init.  Failures return QE_ERR_COMM. */
typedef struct { char internal[128]; } qe_comm_id;     /* = ncclUniqueId */
int32_t qe_comm_unique_id(qe_ctx *ctx, qe_comm_id *out);
int32_t qe_comm_init(qe_ctx *ctx, int32_t nranks, int32_t rank, const qe_comm_id *id);
int32_t qe_comm_rank(const qe_ctx *ctx);       /* -1 without a communicator */
int32_t qe_comm_nranks(const qe_ctx *ctx);     /* 0 without a communicator */
void qe_comm_destroy(qe_ctx *ctx);             /* also done by qe_ctx_destroy */
/* Collective.  On `root`, *out = the concatenation of every rank's result in rank order (free with qe_result_free);
 * on the other ranks *out = NULL.  `local` stays owned by the caller.
 * Every rank must hold a result of the same shape (same plan): column count and types are compared on every rank, and a
 * STRING column must carry THE SAME DICTIONARY (same entries, same order) on every rank -- codes travel, not strings, and
 * the root labels them with its own dictionary.  A host that parses each shard's text separately (qe_csv_parse_file per
 * rank) gets a first-appearance dictionary per shard and must pin ONE shared dictionary instead.  A mismatch is
 * QE_ERR_INVALID_ARG on EVERY rank (the fingerprints travel in the header all-gather), as is a schema mismatch; a rank that
 * cannot allocate its buffers makes the call fail on every rank before any transfer starts (QE_ERR_OOM). */
int32_t qe_gather(qe_ctx *ctx, const qe_result *local, int32_t root, qe_result **out);
/* Collective: Projection(Filter(Scan)) over this rank's shard AND the materialising exchange in one call, OVERLAPPED -- the
 * shard is scanned in `nslices` slices (<= 16; 0 = 8) and a slice's rows travel to the root (copy stream) while the next slice
 * is scanned (compute stream).  The root can only place rows at their final offset if every count is known in advance: the
 * call starts with a count pre-pass (the filter's columns only) and ONE all-gather of every rank's per-slice counts.  Same
 * result, order and checks as qe_filter_project + qe_gather.  On `root` *out = the whole result, elsewhere NULL. */
int32_t qe_filter_project_gather(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter, const qe_expr *const *projections,
                                 int32_t nproj, int32_t root, int32_t nslices, qe_result **out);
/* Collective, small control data (aggregate partials, counts): recv gets nranks * nbytes host bytes in rank order.
 * Aggregations over a sharded table fold the per-GPU partials in rank order on the host. */
int32_t qe_comm_allgather_host(qe_ctx *ctx, const void *send, size_t nbytes, void *recv);

/* ---- introspection ------------------------------------------------------------------------ */
/* HIP source the JIT would compile for this plan (NUL terminated, owned by ctx, valid until next call) */
int32_t qe_filter_project_source(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter,
                                 const qe_expr *const *projections, int32_t nproj, const char **out);
/* Which of the fused kernel's three geometries this plan runs with on large batches: -1 not decided yet (the first
 * executions on a batch of >= 32 Mi rows time all three, best of 3 each), 0 default, 1 wide (16 load groups per sub-tile,
 * 512-entry LDS rings, 2 waves per SIMD), 2 mid (the default sub-tile, 8 Ki-row chunks, 512-entry rings, 2 waves per
 * workgroup).  The decision is persisted next to the plan's code object in the JIT cache, so a later context / process runs the
 * same geometry without exploring (*out_from_cache = 1 when it came from there) -- unless the winner was less than 7 % ahead
 * (the spread of one binary over the boxes of a pool): such a decision is measured again once per context. */
int32_t qe_filter_project_geometry(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter,
                                   const qe_expr *const *projections, int32_t nproj, int32_t *out_chosen,
                                   int32_t *out_from_cache);
/* The order in which this plan evaluates the conjuncts of its filter's top-level AND chain (and so loads their columns):
 * out_order[k] = index, in WRITTEN order, of the conjunct evaluated k-th.  *out_nconj = number of conjuncts, or -1 while
 * the plan has not measured yet (its first execution on a batch of >= 8 Mi rows measures every conjunct's pass rate and
 * keeps the order that fetches the fewest lines; smaller batches run as written).  The reference's AND is lazy left to
 * right (evaluator/Interpreter.kt:54-72) -- with typed plans and no side effects any order keeps the same rows. */
int32_t qe_filter_project_conjunct_order(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter,
                                         const qe_expr *const *projections, int32_t nproj, int32_t *out_order, int32_t capacity,
                                         int32_t *out_nconj);
/* Which form the last qe_filter_project / qe_filter_groupby on this context ran in (-1: none yet).  The fused executor picks the form from the
 * share of rows the plan kept last time: the local form up to 3 % (large batches), the LDS-ring single pass below 12 %, the
 * dense single pass from there on. */
enum { QE_FORM_RING = 0, QE_FORM_TWO_PASS = 1, QE_FORM_DENSE = 2, QE_FORM_PER_NODE = 3, QE_FORM_NO_FILTER = 4,
       QE_FORM_LOCAL = 5 /* dependency-free scan into per-chunk slots + one move: plans that kept <= 3 % of their rows */,
       /* qe_filter_groupby: */ QE_FORM_GROUPBY_DENSE = 8, QE_FORM_GROUPBY_HASHED = 9,
       QE_FORM_GROUPBY_HASH_PARTITIONED = 10 /* many distinct numeric keys: rows scattered by key hash, an LDS hash table per partition */ };
int32_t qe_ctx_last_form(const qe_ctx *ctx);
/* measured device read bandwidth of a plain streaming kernel over nbytes (GB/s): roofline calibration */
int32_t qe_stream_read_bandwidth(qe_ctx *ctx, int64_t nbytes, int32_t reps, double *out_gbps);

/* calibration: time (ms) of streaming nbytes with one 512-byte block WRITTEN per wave every `write_every`
 * 8-KiB read iterations (0 = read only); what a trickle of writes costs a read stream on this device */
int32_t qe_stream_read_write_time(qe_ctx *ctx, int64_t nbytes, int32_t write_every, int32_t reps, double *out_ms,
                                  double *out_written_bytes);

#ifdef __cplusplus
}
#endif
#endif

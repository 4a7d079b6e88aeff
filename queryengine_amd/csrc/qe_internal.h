// qe_internal.h -- internal structures of libqe_hip.so (not part of the ABI).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/qe_hip.h"

namespace qe {

// ---- errors ---------------------------------------------------------------------
struct Error {
    int32_t code;
    std::string msg;
};
[[noreturn]] void fail(int32_t code, const std::string &msg);
void hip_check(hipError_t e, const char *what, const char *file, int line);
#define QE_HIP(x) ::qe::hip_check((x), #x, __FILE__, __LINE__)

// ---- IR: typed expression tree ------------------------------------------------------
enum NodeKind { N_COLUMN = 0, N_NUM = 1, N_BOOL = 2, N_STR = 3, N_FN = 4, N_CAST = 5 };

struct Node {
    int kind = 0;
    int fn = -1;        // QE_FN_* for N_FN
    int type = -1;      // QE_* data type of the value this node produces
    int col = -1;       // N_COLUMN
    double num = 0.0;   // N_NUM
    bool bval = false;  // N_BOOL
    std::string str;    // N_STR
    std::vector<int> ops;
};

struct Expr {
    std::vector<uint8_t> program;  // the serialised postfix program (cache key material)
    std::vector<Node> nodes;       // children before parents
    int root = -1;
    int max_stack = 0;             // MaxStackVisitor analogue
};

Expr decode_program(const uint8_t *program, size_t len);
int promote_type(int a, int b);
const char *type_name(int t);
const char *fn_name(int fn);

// ---- dictionaries ----------------------------------------------------------------------
struct DictData {
    // process-wide serial number, assigned when the dictionary is created: plan caches key on it, never on the
    // address (an allocator hands a freed dictionary's address to the next one)
    uint64_t id = next_id();
    static uint64_t next_id();
    std::vector<std::string> entries;
    std::unordered_map<std::string, int32_t> index;
    int32_t find(const std::string &s) const {
        auto it = index.find(s);
        return it == index.end() ? -1 : it->second;
    }
};

// java.lang.String.compareTo (BytecodeCompiler.kt:303, Interpreter.kt:104-107) on UTF-8 input: lexicographic order of
// the UTF-16 code units (a supplementary character sorts as its surrogate pair, i.e. BEFORE U+E000..U+FFFF).
int utf16_compare(const std::string &a, const std::string &b);
// Dense ranks of the strings of several lists in ONE merged compareTo order: rank[l][i] of lists[l][i]; equal strings get
// equal ranks, so every comparison of two strings is the same comparison of their ranks.
std::vector<std::vector<int32_t>> merged_ranks(const std::vector<const std::vector<std::string> *> &lists);

}  // namespace qe

struct qe_dict {
    std::shared_ptr<qe::DictData> d;
};

namespace qe {

// ---- device memory pool --------------------------------------------------------------------
// Result and scratch buffers are recycled across calls so that a re-opened
// operator (T/SimpleSumBenchmark.java:63-94) does not pay hipMalloc/hipFree
// (each a device synchronisation) per open().
class Pool {
public:
    void *alloc(size_t bytes);
    void release(void *p);
    void trim();
    ~Pool() { trim_all(); }
    size_t bytes_in_use = 0, bytes_cached = 0;

private:
    void trim_all();
    std::multimap<size_t, void *> free_;
    std::unordered_map<void *, size_t> live_;
};

// Pinned (page-locked) host buffers, recycled per context like the device pool: hipHostMalloc pins pages at ~1 GB/s-scale
// rates, a re-opened operator must not pay that per open() (T/SimpleSumBenchmark.java:63-94 re-runs one plan).
class PinnedPool {
public:
    void *alloc(size_t bytes);
    void release(void *p);
    void trim();
    ~PinnedPool() { trim(); }

private:
    std::multimap<size_t, void *> free_;
    std::unordered_map<void *, size_t> live_;
};

// Table entry of the hash-partitioned GROUP BY behind its key words: only the words the plan needs.  An aggregate's counter says
// whether it saw a non-null value, or counts (COUNT, AVG): aggregates over inputs that cannot be NULL share one (the first such
// aggregate's), and that one exists only when something counts; a nullable aggregate keeps its own.  COUNT has no accumulator.
// cw[i] / aw[i] = word of aggregate i's counter / accumulator (1-based behind the key words; 0 = none: the count reads as 1).
struct HpEntryLayout {
    std::vector<int> cw, aw, init_of;   // init_of[w - 1] = the aggregate whose accumulator word w is (-1: a counter)
    int words = 0;
};
inline HpEntryLayout hp_entry_layout(const std::vector<char> &nullable, const int32_t *agg_fns, int nagg) {
    HpEntryLayout L;
    L.cw.assign((size_t)nagg, 0);
    L.aw.assign((size_t)nagg, 0);
    int shared = -1;
    bool shared_counts = false;
    for (int i = 0; i < nagg; i++)
        if (!nullable[(size_t)i]) {
            if (shared < 0) shared = i;
            shared_counts = shared_counts || agg_fns[i] == QE_AGG_COUNT || agg_fns[i] == QE_AGG_AVG;
        }
    for (int i = 0; i < nagg; i++) {
        if (nullable[(size_t)i] || (i == shared && shared_counts)) {
            L.cw[(size_t)i] = ++L.words;
            L.init_of.push_back(-1);
        }
        if (agg_fns[i] != QE_AGG_COUNT) {
            L.aw[(size_t)i] = ++L.words;
            L.init_of.push_back(i);
        }
    }
    for (int i = 0; i < nagg; i++)
        if (!nullable[(size_t)i] && i != shared) L.cw[(size_t)i] = shared >= 0 ? L.cw[(size_t)shared] : 0;
    return L;
}

struct Column {
    int type = 0;
    void *data = nullptr;            // device
    uint64_t *validity = nullptr;    // device or null
    std::shared_ptr<DictData> dict;  // QE_STRING
    bool owned = true;
};

}  // namespace qe

struct qe_batch {
    int64_t nrows = 0;
    bool schema_only = false;
    std::vector<qe::Column> cols;
};

struct qe_expr {
    qe::Expr e;
};

namespace qe {

struct OutColumn {
    int type = 0;
    bool nullable = false;
    void *data = nullptr;          // device: values (BOOLEAN: bitmap words after packing)
    uint64_t *validity = nullptr;  // device bitmap or null
    void *bytes_data = nullptr;    // device scratch: BOOLEAN values as bytes before packing
    uint8_t *bytes_valid = nullptr;// device scratch: validity as bytes before packing
    std::shared_ptr<DictData> dict;
    qe_dict dict_handle;
    // QE_EXEC_PER_NODE: buffers shared with the executor's temporaries (released when reset)
    std::shared_ptr<void> hold_data, hold_valid;
};

}  // namespace qe

struct qe_result {
    int64_t count = 0;
    int64_t capacity = 0;
    std::vector<qe::OutColumn> cols;
};

namespace qe {

// ---- bound plan: expression + schema -> generated source + loaded kernel -------------------
struct BoundColumn {
    int type;
    bool nullable;
    std::shared_ptr<DictData> dict;
};

struct OutSpec {
    int type;
    bool nullable;
    std::shared_ptr<DictData> dict;
};

struct FusedGeometry {
    int threads = 256;
    int rows_per_lane = 2;
    int unroll = 8;          // load groups (of 128 rows) per sub-tile
    int subs_per_chunk = 16; // sub-tiles per chunk (one ticket + one look-back per chunk)
    int nbuf = 2;            // LDS buffers per wave = depth of the FIFO of streamed-but-unresolved chunks + 1
    int ring_entries = 256;  // LDS entries per wave, buffer and output column: a chunk's kept rows stay in LDS
                             // until it is resolved; only an overflow spills to the global staging slot
    int lookback_k = 1;      // descriptor windows (of 64) loaded per look-back round
    int gate_period_log2 = 0;   // output-write gate: period / window width in 10 ns ticks of s_memrealtime (0 = no gate)
    int gate_width_log2 = 0;
    int stagger = 0;         // grade the sizes of the first chunks (needs subs_per_chunk % 16 == 0)
    int min_waves = 4;       // __launch_bounds__ 2nd argument: waves per SIMD the register allocator must allow
    int prio_mode = 1;       // 0 off, 1 rotate s_setprio among the waves of a SIMD per sub-tile, 2 per chunk
    int resolve_at = 1;      // sub-tiles into the next chunk from which the previous chunk's resolve is tried
    int sub_rows() const { return 64 * rows_per_lane * unroll; }
    int chunk_rows() const { return sub_rows() * subs_per_chunk; }
    // rows per staging slot: the chunk plus an odd multiple of 64 rows, so that slot bases do not all
    // alias onto the same HBM channels (a power-of-two slot stride made every slot's small used prefix hit
    // the same few channels)
    int slot_rows() const { return chunk_rows() + 832; }
};

struct CodegenInput {
    const Expr *filter = nullptr;
    std::vector<const Expr *> projections;
    std::vector<int> agg_fns;  // non-empty: aggregate mode (one per projection)
    std::vector<const Expr *> group_keys;  // aggregate mode only: non-empty => GROUP BY these (STRING / BOOLEAN) expressions
    std::vector<BoundColumn> schema;
    int cmp_semantics = QE_CMP_TOTAL_ORDER;
    FusedGeometry geo;
    int nontemporal = 1;
    int vec_stores = 0;   // the ring kernel moves a resolved chunk's rows two per lane (16-byte stores for 8-byte columns)
    int nt_stores = 1;    // non-temporal stores for the output rows
    int hp_shift = 0;             // .. buckets per partition = 2^hp_shift (6 .. 11; 0 = 11), fewer when the entry is wide
    int hp_lines = 1;             // .. 0: records {header, values, key words} padded to a power of two instead of lines of records (measurement)
    int hp_parts = 0;             // hashed GROUP BY: > 0 = generate the HASH-PARTITIONED form with this many partitions (a power of two, <= 1024)
    std::vector<int> conj_order;  // evaluation order of the filter's conjuncts (a permutation of their written order); empty = as written
    int filter_load_stages = 0;   // > 0: at most this many load stages for the filter's columns (later conjuncts' columns join the last one)
    int prefetch = 1;     // staged filter+project plans: the stage-0 loads of the next sub-tile are issued at the start of this one
                          // (0 never, 1 plans with >= 4 load stages, 2 every staged plan)
    bool staged = true;   // late materialisation: evaluate the filter's AND chain conjunct by conjunct, load later columns for live rows only
    bool dense = false;   // filter+project only: generate the DENSE single-pass kernel (chunk == sub-tile, blocking look-back between
                          // evaluation and stores, rows go from the registers straight to their final position: no LDS ring, no
                          // staging) -- the form for plans that keep a large share of their rows
    int debug_mask = 0;   // ablation builds (wrong results): 1 no look-back, 2 no staging stores, 4 no move
};

struct CodegenOutput {
    std::string source;
    std::vector<OutSpec> outs;
    std::vector<int> used_cols;  // batch column index per kernel column slot
    bool has_filter = false;
    bool two_pass = false;       // the module also holds qe_fp_count / qe_fp_write (count + direct ordered write)
    bool dense = false;          // the module holds the dense single-pass kernel only (entry qe_fused)
    bool has_probe = false;      // the module holds qe_conj_probe (pass rate of every conjunct on its own)
    int nconj = 0;               // conjuncts of the filter's top-level AND chain (0: not staged)
    std::vector<std::vector<int>> conj_cols;   // per conjunct, in EVALUATION order: the kernel column slots it reads
    std::vector<int> col_width;  // bytes per row of every kernel column slot (0: bitmap)
    int fl_ring = 0;             // local form (qe_fl_scan / qe_fl_move): rows per chunk slot = LDS entries per wave
    std::vector<std::vector<int32_t>> aux_tables;   // int32 tables indexed by dictionary codes (string ranks, remaps): col[kMaxCols-1-k]
    // group-by mode: key columns of the result, their domain sizes (without the extra NULL code) and the
    // accumulator table geometry: ngroups rows of table_words u64 words {first row, (count, acc) per aggregate}
    std::vector<OutSpec> keys;
    std::vector<int> key_domain;
    long long ngroups = 0;
    int table_words = 0;
    bool table_in_lds = false;
    bool hashed = false;         // group-by over arbitrary key tuples (a DOUBLE / INT64 / INT32 key, or too many combinations for a
                                 // dense table): open-addressing hash table, entry = hash_words u64 words
    int hash_words = 0;          // {state, null bits, key words.., first row, (count, acc)..}
    int table_copies = 1;        // global group table: copies merged on the host (one per XCD)
    std::vector<int> cnt_src;    // per aggregate: the aggregate whose count word holds its count (non-nullable inputs all
                                 // see every kept row of the group: they share the first one's counter, one atomic less each)
    // partitioned group-by (domains that do not fit LDS): rows are first scattered into nparts key-range partitions of
    // part_groups (= 1 << part_shift) groups each, then every partition is aggregated in an LDS table
    bool partitioned = false;
    int part_shift = 0;
    int part_groups = 0;
    int nparts = 0;
    std::vector<int> val_slot;   // per aggregate: slot of its input value in a record (identical inputs share one)
    int nvals = 0;               // distinct aggregate inputs; a record is 1 + nvals u64 words {header, values}
    bool hp = false;             // hash-partitioned form of a hashed GROUP BY: dense partitioned passes over {partition, home bucket} pseudo
                                 // ids, records carry hp_key_words key words from value slot hp_key_slot on, entries = hash_words layout
    int hp_key_words = 0, hp_key_slot = 0, hp_shift = 11;   // 2^hp_shift buckets per partition
    int hp_line_recs = 0;        // .. > 0: records live in 128-byte lines {R x value / key words, R x 32-bit row id, R x flag byte}; R = this
};

CodegenOutput generate_fused_source(const CodegenInput &in);

struct Kernel {
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    int scratch = -1;   // spill bytes per lane (-1 unknown)
};

class Jit {
public:
    explicit Jit(std::string cache_dir) : cache_dir_(std::move(cache_dir)) {}
    ~Jit();
    // compile (or fetch from memory / disk cache) and load; needs a current device unless load == false
    Kernel get(const std::string &source, const char *entry, bool load = true);
    void reject(const std::string &source, int scratch);   // drop a spilling code object from the cache, keep the verdict
    static std::vector<char> compile(const std::string &source);
    static std::string source_key(const std::string &source);
    // persisted measured decisions (geometry choice) next to the code object; -1 = none
    int load_choice(const std::string &source, double *margin = nullptr) const;
    void store_choice(const std::string &source, int chosen, const std::string &note) const;
    static int scratch_bytes(const std::vector<char> &code);
    int compiles = 0, disk_hits = 0, mem_hits = 0, last_scratch = -1;

private:
    std::string cache_dir_;
    std::unordered_map<std::string, Kernel> loaded_;
};

struct Plan {
    CodegenOutput cg;
    Kernel kernel;
    FusedGeometry geo;
    bool aggregate = false;
    mutable std::vector<void *> aux_dev;      // device copies of cg.aux_tables (uploaded at the first execution)
    ~Plan() { for (void *q : aux_dev) (void)hipFree(q); }
    mutable double last_selectivity = -1.0;   // kept / scanned rows of the last execution (picks the two-pass form)
    mutable int64_t hash_capacity = 0;        // hashed group-by: entries of the global table that sufficed last time
    mutable int64_t id_capacity = 0;          // .. of the key -> dense id table (qe_ht_build)
    mutable bool use_ids = false;             // .. the keys did not fit the LDS table last time: resolve them to dense ids first
    mutable int64_t known_keys = -1;          // hashed group-by: groups the last execution produced (picks the hash-partitioned form)
    mutable bool hp_failed = false;           // .. a partition's LDS table filled up (or the plan does not build): never again
    mutable bool ids_overflow = false;        // .. more than 2^20 - 1 distinct keys: the id build cannot hold them, never try it again
    mutable std::vector<int> conj_order;      // filter+project: evaluation order of the conjuncts chosen from measured pass rates (empty: as written)
    mutable bool conj_decided = false;
    mutable bool local_overflowed = false;    // filter+project: a chunk of the local form kept more rows than its slot holds: never again
    int est_regs = 0;                         // register estimate of the plan's geometry (get_plan)
    bool explicit_geometry = false;           // unroll / chunk / ring were fixed through qe_options.tuning
};

// kernel parameter block of the generated fused kernel (must match the prelude in qe_codegen.cpp)
constexpr int kMaxCols = 16;
constexpr int kMaxOuts = 16;
struct FusedParams {
    const void *col[kMaxCols];
    const unsigned long long *colvalid[kMaxCols];
    void *out[kMaxOuts];
    unsigned char *outvalid[kMaxOuts];
    void *stage[kMaxOuts];           // per-wave-slot staging of compacted rows (QE_CHUNK_ROWS per slot)
    unsigned char *stagevalid[kMaxOuts];
    long long nrows;
    long long capacity;
    unsigned long long *desc;    // level-0 look-back descriptors, one per chunk (zeroed per launch)
    unsigned long long *l1;      // level-1 descriptors, one per block of 64 chunks (zeroed per launch)
    unsigned long long *blk;     // per-block packed {finished chunks, sum of counts} (zeroed per launch)
    unsigned int *ticket;        // chunk ticket counter (zeroed per launch)
    unsigned long long *total;   // out: number of selected rows
    unsigned int *error;         // out: nonzero when a bounded spin gave up
    double *agg_partial;         // aggregate mode: per-workgroup partials
    long long nchunks;
    long long stagger_chunks;
    long long stagger_rows;
    unsigned long long *trace;
    unsigned long long *stats;
};

}  // namespace qe

struct qe_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    qe_options opts{};
    std::string last_error;
    qe::Pool pool;
    std::unique_ptr<qe::Jit> jit;
    std::map<std::string, std::shared_ptr<qe::Plan>> plans;
    // geometry choice per fused filter+project plan: the first executions on a large batch time the default geometry and
    // the "wide" one (16 load groups per sub-tile, 512-entry LDS rings, 2 waves per SIMD); the faster one is kept
    struct GeoChoice {
        static constexpr int kCands = 3;   // 0 default, 1 wide, 2 mid (default sub-tile, 8 Ki-row chunks, 512-entry rings, 2 waves per workgroup)
        int chosen = -1;
        int runs[kCands] = {0, 0, 0};
        float best_ms[kCands] = {1e30f, 1e30f, 1e30f};
        bool from_cache = false;
    };
    std::map<const qe::Plan *, GeoChoice> geo_choice;   // plans live as long as the context: the pointer is never recycled
    std::map<int64_t, std::shared_ptr<qe::DictData>> id_dicts;   // placeholder dictionaries of 2^k entries: the domain of a dense-id key column
    std::string source_scratch;
    // profiling of the dominant kernel
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double last_ms = 0.0, total_ms = 0.0;
    int64_t launches = 0;
    // small persistent device scratch: ticket, total, error (+ pinned host mirror)
    unsigned int *d_ctrl = nullptr;      // [0]=ticket [1]=error, [2..3]=total (u64)
    unsigned long long *h_ctrl = nullptr;// pinned: total, error
    // RCCL communicator of the exchange step (qe_comm.cpp); null until qe_comm_init
    void *comm = nullptr;
    int comm_rank = -1, comm_nranks = 0;
    int last_form = -1;   // QE_FORM_* of the last qe_filter_project execution
    // result -> host (qe_result_to_host): a second stream, so that the copy of one batch's result runs beside the scan of the
    // next batch, and pinned staging owned by the context
    hipStream_t copy_stream = nullptr;
    qe::PinnedPool pinned;
    std::vector<struct qe_host_result *> host_results;   // alive host results (their copies may still read a qe_result)
};

struct qe_host_result {
    const qe_result *src = nullptr;    // the device result the copies read; nullptr once they have completed
    int64_t count = 0;
    struct Col {
        int type = 0;
        bool nullable = false;
        void *data = nullptr;          // pinned host memory
        uint64_t *validity = nullptr;  // pinned host memory or null
        std::shared_ptr<qe::DictData> dict;
        qe_dict dict_handle;
    };
    std::vector<Col> cols;
    hipEvent_t done = nullptr;
    bool waited = false;
};

// internals shared by qe_api.cpp and qe_comm.cpp (the overlapped scan + exchange)
std::vector<int64_t> qe_int_count_slices(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter, const qe_expr *const *projs, int32_t nproj,
                                         int64_t *slice_rows_io, int32_t nslices);
qe_result *qe_int_run_fused_slice(qe_ctx *ctx, const qe_batch *batch, int64_t row_begin, int64_t nrows, const qe_expr *filter,
                                  const qe_expr *const *projs, int32_t nproj);

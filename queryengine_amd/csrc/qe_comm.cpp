// qe_comm.cpp -- the ONE exchange step of the path: results of a row-range sharded scan are materialised on
// one rank.  RCCL is used directly (no PyTorch in the data path): ncclAllGather of the per-rank result headers, then ONE
// ncclGroupStart/End in which the root posts a ncclRecv per peer and column and every peer a ncclSend per column, so that
// each peer streams over its OWN xGMI link into the root (a fully connected mesh: a ring would be bound by one link).
// Value columns land at their final offset (rank order == the reference's row order, FilterOperator.kt:17-22 is order
// preserving); bitmap columns (BOOLEAN values, validity) travel as raw 64-row WORDS and are funnel-shifted into place on
// the root (bitmap_place_kernel) -- never expanded to a byte per row.
//
// The same placement code concatenates several results of ONE device (qe_result_concat): that is what a host does that
// feeds a table batch by batch and still hands the reference's single Operator the whole result (Planner.kt:30-63).
//
// librccl.so.1 is opened lazily (dlopen) at qe_comm_init, so the library itself loads on hosts without RCCL and shares
// the RCCL a host process may already have loaded (PyTorch bundles one with the same SONAME).
#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <functional>
#include <numeric>

#include "qe_internal.h"
#include "qe_kernels.h"

namespace qe {

// ---- the slice of the RCCL API this file uses (rccl.h: ncclResult_t = int, ncclSuccess = 0) ----------------------------
struct RcclApi {
    void *handle = nullptr;
    int (*GetUniqueId)(void *id) = nullptr;
    int (*CommInitRank)(void **comm, int nranks, qe_comm_id id, int rank) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*AllGather)(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t s) = nullptr;
    int (*Send)(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t s) = nullptr;
    int (*Recv)(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t s) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
constexpr int kNcclUint8 = 1;   // rccl.h: ncclUint8 = 1

static RcclApi load_rccl() {
    RcclApi api;
    // QE_RCCL_LIBRARY: another library with the same nine entry points (a site's own RCCL build; the test suite's transport)
    const char *names[] = {std::getenv("QE_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    std::string tried;
    for (const char *n : names) {
        if (!n || !*n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
        tried += std::string(tried.empty() ? "" : "; ") + dlerror();
    }
    if (!h) fail(QE_ERR_COMM, "RCCL is not available: " + tried);
    auto sym = [&](const char *name) {
        void *p = dlsym(h, name);
        if (!p) fail(QE_ERR_COMM, std::string("librccl lacks ") + name);
        return p;
    };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    api.handle = h;
    return api;
}

// initialised once, under the language's own lock; a failed load throws and is retried by the next call
static RcclApi &rccl() {
    static RcclApi api = load_rccl();
    return api;
}

static void nccl_check(int r, const char *what) {
    if (r == 0) return;
    const char *msg = rccl().GetErrorString ? rccl().GetErrorString(r) : "?";
    fail(QE_ERR_COMM, std::string(what) + " failed: " + (msg ? msg : "?") + " (" + std::to_string(r) + ")");
}
#define QE_NCCL(x) ::qe::nccl_check((x), #x)

static size_t width_of(int t) { return (t == QE_DOUBLE || t == QE_INT64) ? 8 : (t == QE_INT32 || t == QE_STRING) ? 4 : 0; }
static size_t words_of(int64_t n) { return (size_t)((n + 63) / 64); }

// Fingerprint of a dictionary: STRING columns travel as raw codes, so every rank must hold the SAME dictionary (same
// entries in the same order) or the root would decode a peer's codes to the wrong strings.
static uint64_t dict_fingerprint(const DictData *d) {
    if (!d) return 0;
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t n) {
        const unsigned char *c = (const unsigned char *)p;
        for (size_t i = 0; i < n; i++) h = (h ^ c[i]) * 1099511628211ull;
    };
    const uint64_t n = d->entries.size();
    mix(&n, 8);
    for (const std::string &e : d->entries) {
        const uint64_t len = e.size();
        mix(&len, 8);
        mix(e.data(), e.size());
    }
    return h ? h : 1;
}

// header every rank contributes to the all-gather: its row count and the shape of its result
constexpr int kGatherMaxCols = 16;
struct GatherHeader {
    int64_t count;
    int32_t ncols;
    uint32_t validity_mask;    // bit c: column c carries a validity bitmap on this rank
    uint64_t type_sig;         // 4 bits per column: the column types must agree on every rank
    uint64_t dict_fp[kGatherMaxCols];   // STRING columns: dict_fingerprint of the column's dictionary (0 otherwise)
};

static GatherHeader header_of(const qe_result *r, bool with_dicts = false) {
    GatherHeader h{};
    h.count = r->count;
    h.ncols = (int32_t)r->cols.size();
    for (size_t c = 0; c < r->cols.size() && c < (size_t)kGatherMaxCols; c++) {
        if (r->cols[c].validity) h.validity_mask |= 1u << c;
        h.type_sig |= (uint64_t)(r->cols[c].type & 15) << (4 * c);
        if (with_dicts && r->cols[c].type == QE_STRING) h.dict_fp[c] = dict_fingerprint(r->cols[c].dict.get());
    }
    return h;
}

// ncclGroupStart .. ncclGroupEnd that is closed on every path: an exception between the two must not leave the thread's
// group open (every later collective of the thread would be queued into it and never run)
struct RcclGroup {
    RcclApi &nc;
    bool open = false;
    explicit RcclGroup(RcclApi &api) : nc(api) {
        const int r = nc.GroupStart();
        if (r != 0) fail(QE_ERR_COMM, "ncclGroupStart failed (" + std::to_string(r) + ")");
        open = true;
    }
    void end() {
        open = false;
        const int r = nc.GroupEnd();
        if (r != 0) {
            const char *msg = nc.GetErrorString ? nc.GetErrorString(r) : nullptr;
            fail(QE_ERR_COMM, std::string("ncclGroupEnd failed: ") + (msg ? msg : "?") + " (" + std::to_string(r) + ")");
        }
    }
    ~RcclGroup() {
        if (open) (void)nc.GroupEnd();
    }
};

// The output result of a concatenation / gather: per column a values buffer for `total` rows and, if any part carries a
// validity bitmap, a validity bitmap (parts without one contribute ones).
static qe_result *make_output(qe_ctx *ctx, const qe_result *like, int64_t total, uint32_t any_validity) {
    std::unique_ptr<qe_result> out(new qe_result());
    out->count = total;
    out->capacity = total;
    try {
        for (size_t c = 0; c < like->cols.size(); c++) {
            const OutColumn &src = like->cols[c];
            OutColumn oc;
            oc.type = src.type;
            oc.dict = src.dict;
            oc.dict_handle.d = src.dict;
            oc.nullable = (any_validity >> c) & 1u;
            out->cols.push_back(oc);
            OutColumn &dst = out->cols.back();
            const size_t nb = src.type == QE_BOOLEAN ? words_of(total) * 8 : width_of(src.type) * (size_t)total;
            dst.data = ctx->pool.alloc(std::max<size_t>(nb, 16));
            if (src.type == QE_BOOLEAN && nb) QE_HIP(hipMemsetAsync(dst.data, 0, nb, ctx->stream));
            if (dst.nullable) {
                dst.validity = (uint64_t *)ctx->pool.alloc(std::max<size_t>(words_of(total) * 8, 16));
                if (total > 0) QE_HIP(hipMemsetAsync(dst.validity, 0, words_of(total) * 8, ctx->stream));
            }
        }
    } catch (...) {
        for (auto &c : out->cols) {
            ctx->pool.release(c.data);
            ctx->pool.release(c.validity);
        }
        throw;
    }
    return out.release();
}

static void free_output(qe_ctx *ctx, qe_result *r) {
    if (!r) return;
    for (auto &c : r->cols) {
        ctx->pool.release(c.data);
        ctx->pool.release(c.validity);
    }
    delete r;
}

// scratch buffers that go back to the pool when the call ends
struct Scratch {
    qe_ctx *ctx;
    std::vector<void *> bufs;
    void *get(size_t bytes) {
        void *p = ctx->pool.alloc(std::max<size_t>(bytes, 16));
        bufs.push_back(p);
        return p;
    }
    ~Scratch() { for (void *p : bufs) ctx->pool.release(p); }
};

static uint64_t *ones_bitmap(qe_ctx *ctx, Scratch &sc, int64_t n) {
    uint64_t *p = (uint64_t *)sc.get(words_of(n) * 8);
    QE_HIP(hipMemsetAsync(p, 0xff, words_of(n) * 8, ctx->stream));
    return p;
}

}  // namespace qe

using namespace qe;

template <typename F>
static int32_t guarded_comm(qe_ctx *ctx, F &&f) {
    try {
        f();
        return QE_OK;
    } catch (const Error &e) {
        if (ctx) ctx->last_error = e.msg;
        return e.code;
    } catch (const std::bad_alloc &) {
        if (ctx) ctx->last_error = "host out of memory";
        return QE_ERR_OOM;
    } catch (const std::exception &e) {
        if (ctx) ctx->last_error = e.what();
        return QE_ERR_INTERNAL;
    }
}

static void need_dev(const qe_ctx *ctx) {
    if (ctx->device < 0) fail(QE_ERR_HIP, "planning-only context (QE_DEVICE_NONE): this call needs a HIP device");
    QE_HIP(hipSetDevice(ctx->device));
}

extern "C" {

int32_t qe_comm_unique_id(qe_ctx *ctx, qe_comm_id *out) {
    if (!ctx || !out) return QE_ERR_INVALID_ARG;
    return guarded_comm(ctx, [&] { QE_NCCL(rccl().GetUniqueId(out)); });
}

int32_t qe_comm_init(qe_ctx *ctx, int32_t nranks, int32_t rank, const qe_comm_id *id) {
    if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return QE_ERR_INVALID_ARG;
    return guarded_comm(ctx, [&] {
        need_dev(ctx);
        if (ctx->comm) fail(QE_ERR_INVALID_ARG, "qe_comm_init: this context already has a communicator");
        void *comm = nullptr;
        QE_NCCL(rccl().CommInitRank(&comm, nranks, *id, rank));
        ctx->comm = comm;
        ctx->comm_rank = rank;
        ctx->comm_nranks = nranks;
    });
}

int32_t qe_comm_rank(const qe_ctx *ctx) { return ctx && ctx->comm ? ctx->comm_rank : -1; }
int32_t qe_comm_nranks(const qe_ctx *ctx) { return ctx && ctx->comm ? ctx->comm_nranks : 0; }

void qe_comm_destroy(qe_ctx *ctx) {
    if (!ctx || !ctx->comm) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    (void)rccl().CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_nranks = 0;
    ctx->comm_rank = -1;
}

// every rank contributes nbytes host bytes; recv gets nranks * nbytes in rank order (small control data: aggregate
// partials, counts).  SURVEY 8f: "Cross-GPU: reduce of the per-GPU partials" -- folded by the host in rank order.
int32_t qe_comm_allgather_host(qe_ctx *ctx, const void *send, size_t nbytes, void *recv) {
    if (!ctx || !send || !recv || nbytes == 0) return QE_ERR_INVALID_ARG;
    return guarded_comm(ctx, [&] {
        need_dev(ctx);
        if (!ctx->comm) fail(QE_ERR_COMM, "qe_comm_allgather_host: no communicator (qe_comm_init)");
        Scratch sc{ctx, {}};
        const size_t n = (size_t)ctx->comm_nranks;
        char *d_send = (char *)sc.get(nbytes), *d_recv = (char *)sc.get(nbytes * n);
        QE_HIP(hipMemcpyAsync(d_send, send, nbytes, hipMemcpyHostToDevice, ctx->stream));
        QE_NCCL(rccl().AllGather(d_send, d_recv, nbytes, kNcclUint8, ctx->comm, ctx->stream));
        QE_HIP(hipMemcpyAsync(recv, d_recv, nbytes * n, hipMemcpyDeviceToHost, ctx->stream));
        QE_HIP(hipStreamSynchronize(ctx->stream));
    });
}

// Concatenate results of ONE device in the given order (values at row offsets, bitmaps shifted into place).
int32_t qe_result_concat(qe_ctx *ctx, const qe_result *const *parts, int32_t nparts, qe_result **out) {
    if (!ctx || !out || nparts < 1 || !parts) return QE_ERR_INVALID_ARG;
    *out = nullptr;
    return guarded_comm(ctx, [&] {
        need_dev(ctx);
        int64_t total = 0;
        uint32_t any_validity = 0;
        for (int32_t i = 0; i < nparts; i++) {
            if (!parts[i]) fail(QE_ERR_INVALID_ARG, "qe_result_concat: null part");
            const GatherHeader h = header_of(parts[i]), h0 = header_of(parts[0]);
            if (h.ncols != h0.ncols || h.type_sig != h0.type_sig) fail(QE_ERR_INVALID_ARG, "qe_result_concat: parts differ in schema");
            if (h.ncols > 16) fail(QE_ERR_UNSUPPORTED, "qe_result_concat: more than 16 columns");
            for (size_t c = 0; c < parts[i]->cols.size(); c++)
                if (parts[i]->cols[c].dict != parts[0]->cols[c].dict &&
                    (!parts[i]->cols[c].dict || !parts[0]->cols[c].dict || parts[i]->cols[c].dict->entries != parts[0]->cols[c].dict->entries))
                    fail(QE_ERR_INVALID_ARG, "qe_result_concat: STRING columns must share one dictionary");
            total += h.count;
            any_validity |= h.validity_mask;
        }
        Scratch sc{ctx, {}};
        std::unique_ptr<qe_result, std::function<void(qe_result *)>> res(make_output(ctx, parts[0], total, any_validity),
                                                                          [ctx](qe_result *r) { free_output(ctx, r); });
        int64_t off = 0;
        for (int32_t i = 0; i < nparts; i++) {
            const int64_t n = parts[i]->count;
            if (n == 0) continue;
            for (size_t c = 0; c < res->cols.size(); c++) {
                const OutColumn &src = parts[i]->cols[c];
                OutColumn &dst = res->cols[c];
                if (src.type == QE_BOOLEAN)
                    launch_bitmap_place(ctx->stream, (uint64_t *)dst.data, off, (const uint64_t *)src.data, n);
                else
                    QE_HIP(hipMemcpyAsync((char *)dst.data + width_of(src.type) * (size_t)off, src.data, width_of(src.type) * (size_t)n,
                                          hipMemcpyDeviceToDevice, ctx->stream));
                if (dst.nullable)
                    launch_bitmap_place(ctx->stream, dst.validity, off, src.validity ? src.validity : ones_bitmap(ctx, sc, n), n);
            }
            off += n;
        }
        QE_HIP(hipGetLastError());
        QE_HIP(hipStreamSynchronize(ctx->stream));
        *out = res.release();
    });
}

// ORDER BY <column> on the device: OrderByOperator.open (operator/OrderByOperator.kt:9-15) sorts the materialised rows stably
// with compareValues -- null first, Double.compareTo (-0.0 < 0.0, NaN greatest), String.compareTo (UTF-16 code units),
// false < true.  Key images + stable LSD radix sort of (key, row id) + gather of every column (qe_sort.hip).
int32_t qe_result_order_by(qe_ctx *ctx, const qe_result *src, int32_t column, qe_result **out) {
    if (!ctx || !src || !out || column < 0 || column >= (int32_t)src->cols.size()) return QE_ERR_INVALID_ARG;
    *out = nullptr;
    return guarded_comm(ctx, [&] {
        need_dev(ctx);
        const int64_t n = src->count;
        if (n >= (1ll << 32)) fail(QE_ERR_UNSUPPORTED, "qe_result_order_by: more than 2^32 rows");
        uint32_t any_validity = 0;
        for (size_t c = 0; c < src->cols.size(); c++)
            if (src->cols[c].validity) any_validity |= 1u << c;
        std::unique_ptr<qe_result, std::function<void(qe_result *)>> res(make_output(ctx, src, n, any_validity),
                                                                          [ctx](qe_result *r) { free_output(ctx, r); });
        if (n > 0) {
            Scratch sc{ctx, {}};
            const OutColumn &kc = src->cols[(size_t)column];
            unsigned long long *keys[2] = {(unsigned long long *)sc.get((size_t)n * 8), (unsigned long long *)sc.get((size_t)n * 8)};
            uint32_t *rows[2] = {(uint32_t *)sc.get((size_t)n * 4), (uint32_t *)sc.get((size_t)n * 4)};
            uint32_t *hist = (uint32_t *)sc.get((size_t)((n + 1023) / 1024) * 16 * 4);
            SortKeyArgs ka{};
            ka.type = kc.type;
            ka.data = kc.data;
            ka.validity = (const unsigned long long *)kc.validity;
            ka.n = n;
            ka.keys = keys[0];
            ka.rows = rows[0];
            if (kc.type == QE_STRING) {   // String.compareTo order of the dictionary (UTF-16 code units), as dense ranks
                if (!kc.dict) fail(QE_ERR_INVALID_ARG, "STRING column without dictionary");
                const std::vector<std::vector<int32_t>> ranks = merged_ranks({&kc.dict->entries});
                int *d_ranks = (int *)sc.get(std::max<size_t>(ranks[0].size() * 4, 16));
                if (!ranks[0].empty()) QE_HIP(hipMemcpyAsync(d_ranks, ranks[0].data(), ranks[0].size() * 4, hipMemcpyHostToDevice, ctx->stream));
                QE_HIP(hipStreamSynchronize(ctx->stream));   // `ranks` is a host temporary
                ka.ranks = d_ranks;
                ka.nranks = (int)ranks[0].size();
            }
            launch_sort_keys(ctx->stream, ka);
            // which digits differ at all?
            unsigned long long *d_bits = (unsigned long long *)sc.get(16);
            const unsigned long long init[2] = {0ull, ~0ull};
            QE_HIP(hipMemcpyAsync(d_bits, init, 16, hipMemcpyHostToDevice, ctx->stream));
            launch_key_bits(ctx->stream, keys[0], n, d_bits);
            unsigned long long h_bits[2] = {0, 0};
            QE_HIP(hipMemcpyAsync(h_bits, d_bits, 16, hipMemcpyDeviceToHost, ctx->stream));
            QE_HIP(hipStreamSynchronize(ctx->stream));
            const unsigned long long varying = h_bits[0] & ~h_bits[1];
            int cur = 0;
            for (int shift = 0; shift < 64; shift += 4) {
                if (((varying >> shift) & 15ull) == 0) continue;   // the same digit in every key
                launch_radix_pass(ctx->stream, keys[cur], rows[cur], nullptr, n, shift, hist, keys[cur ^ 1], rows[cur ^ 1]);
                cur ^= 1;
            }
            if (kc.validity) {   // NULL rows in front (compareValues), in their input order
                launch_radix_pass(ctx->stream, keys[cur], rows[cur], kc.validity, n, 64, hist, keys[cur ^ 1], rows[cur ^ 1]);
                cur ^= 1;
            }
            for (size_t c = 0; c < src->cols.size(); c++) {
                const OutColumn &s_ = src->cols[c];
                OutColumn &d_ = res->cols[c];
                if (s_.type == QE_BOOLEAN) launch_gather_bits_rows(ctx->stream, (const uint64_t *)s_.data, rows[cur], n, (uint64_t *)d_.data);
                else launch_gather_rows(ctx->stream, (int)width_of(s_.type), s_.data, rows[cur], n, d_.data);
                if (d_.nullable) launch_gather_bits_rows(ctx->stream, s_.validity, rows[cur], n, d_.validity);
            }
            QE_HIP(hipGetLastError());
            QE_HIP(hipStreamSynchronize(ctx->stream));
        }
        *out = res.release();
    });
}

// Materialise a sharded result on rank `root`: *out is the concatenation in rank order there, NULL elsewhere.
// Collective: every rank of the communicator calls it with its local result (same plan => same column types; STRING
// columns must hold the same dictionary on every rank -- codes travel, not strings).
//
// Shape of the call, chosen so that NO rank can be left waiting inside RCCL for a peer that has already given up:
//   (1) all-gather of the result headers; every rank checks the SAME headers and so reaches the same verdict;
//   (2) every rank allocates all it needs (the root: the output and the bitmap staging; a peer: its all-ones validity
//       words) and the ranks all-gather one status word -- an allocation failure anywhere aborts the call everywhere;
//   (3) only then ONE ncclGroupStart/End of the transfers, with nothing that can throw in between except RCCL itself.
int32_t qe_gather(qe_ctx *ctx, const qe_result *local, int32_t root, qe_result **out) {
    if (!ctx || !local || !out) return QE_ERR_INVALID_ARG;
    *out = nullptr;
    return guarded_comm(ctx, [&] {
        need_dev(ctx);
        if (!ctx->comm) fail(QE_ERR_COMM, "qe_gather: no communicator (qe_comm_init)");
        const int nranks = ctx->comm_nranks, rank = ctx->comm_rank;
        if (root < 0 || root >= nranks) fail(QE_ERR_INVALID_ARG, "qe_gather: root out of range");
        // a rank whose local result cannot travel still takes part in the header exchange (with ncols = -1), so that the
        // others do not wait for it
        const bool too_wide = local->cols.size() > (size_t)kGatherMaxCols;
        RcclApi &nc = rccl();
        Scratch sc{ctx, {}};
        // (1) all-gather of the result headers: counts -> offsets; shapes and dictionaries are checked on every rank
        GatherHeader mine = header_of(local, true);
        if (too_wide) mine.ncols = -1;
        std::vector<GatherHeader> hdr((size_t)nranks);
        {
            GatherHeader *d_mine = (GatherHeader *)sc.get(sizeof(GatherHeader));
            GatherHeader *d_all = (GatherHeader *)sc.get(sizeof(GatherHeader) * (size_t)nranks);
            QE_HIP(hipMemcpyAsync(d_mine, &mine, sizeof mine, hipMemcpyHostToDevice, ctx->stream));
            QE_NCCL(nc.AllGather(d_mine, d_all, sizeof(GatherHeader), kNcclUint8, ctx->comm, ctx->stream));
            QE_HIP(hipMemcpyAsync(hdr.data(), d_all, sizeof(GatherHeader) * (size_t)nranks, hipMemcpyDeviceToHost, ctx->stream));
            QE_HIP(hipStreamSynchronize(ctx->stream));
        }
        int64_t total = 0;
        uint32_t any_validity = 0;
        std::vector<int64_t> offset((size_t)nranks, 0);
        const GatherHeader &ref = hdr[(size_t)root];    // every rank compares against the root's shape: one verdict everywhere
        for (int r = 0; r < nranks; r++) {
            if (hdr[r].ncols < 0) fail(QE_ERR_UNSUPPORTED, "qe_gather: rank " + std::to_string(r) + " holds more than 16 columns");
            if (hdr[r].ncols != ref.ncols || hdr[r].type_sig != ref.type_sig)
                fail(QE_ERR_INVALID_ARG, "qe_gather: rank " + std::to_string(r) + " holds a result of a different schema than rank " +
                                             std::to_string(root));
            for (int c = 0; c < ref.ncols; c++)
                if (hdr[r].dict_fp[c] != ref.dict_fp[c])
                    fail(QE_ERR_INVALID_ARG, "qe_gather: column " + std::to_string(c) + " of rank " + std::to_string(r) +
                                                 " has another dictionary than on rank " + std::to_string(root) +
                                                 " (STRING columns travel as codes: every rank must pin the same dictionary)");
            offset[r] = total;
            total += hdr[r].count;
            any_validity |= hdr[r].validity_mask;
        }
        const size_t ncols = local->cols.size();
        const int64_t n_me = mine.count;

        // (2) local allocations, then one status word per rank
        std::unique_ptr<qe_result, std::function<void(qe_result *)>> res(nullptr, [ctx](qe_result *r) { free_output(ctx, r); });
        std::vector<const uint64_t *> vsend(ncols, nullptr);   // what this rank contributes as validity words per column
        struct Staged { uint64_t *words; int64_t off, n; uint64_t *dst; };
        std::vector<Staged> staged;                            // root: received bitmap words waiting for their placement
        int32_t my_status = QE_OK;
        std::string my_error;
        try {
            if (n_me > 0) {
                const uint64_t *ones = nullptr;
                for (size_t c = 0; c < ncols; c++) {
                    if (!((any_validity >> c) & 1u)) continue;
                    if (local->cols[c].validity) vsend[c] = local->cols[c].validity;
                    else vsend[c] = ones ? ones : (ones = ones_bitmap(ctx, sc, n_me));
                }
            }
            if (rank == root) {
                res.reset(make_output(ctx, local, total, any_validity));
                for (int r = 0; r < nranks; r++) {
                    const int64_t n = hdr[r].count;
                    if (n == 0 || r == root) continue;
                    for (size_t c = 0; c < ncols; c++) {
                        OutColumn &dst = res->cols[c];
                        if (dst.type == QE_BOOLEAN) staged.push_back({(uint64_t *)sc.get(words_of(n) * 8), offset[r], n, (uint64_t *)dst.data});
                        if (dst.nullable) staged.push_back({(uint64_t *)sc.get(words_of(n) * 8), offset[r], n, dst.validity});
                    }
                }
            }
        } catch (const Error &e) {
            my_status = e.code;
            my_error = e.msg;
        } catch (const std::bad_alloc &) {
            my_status = QE_ERR_OOM;
            my_error = "host out of memory";
        }
        {
            std::vector<int32_t> status((size_t)nranks, 0);
            int32_t *d_mine = (int32_t *)sc.get(sizeof(int32_t));
            int32_t *d_all = (int32_t *)sc.get(sizeof(int32_t) * (size_t)nranks);
            QE_HIP(hipMemcpyAsync(d_mine, &my_status, sizeof my_status, hipMemcpyHostToDevice, ctx->stream));
            QE_NCCL(nc.AllGather(d_mine, d_all, sizeof(int32_t), kNcclUint8, ctx->comm, ctx->stream));
            QE_HIP(hipMemcpyAsync(status.data(), d_all, sizeof(int32_t) * (size_t)nranks, hipMemcpyDeviceToHost, ctx->stream));
            QE_HIP(hipStreamSynchronize(ctx->stream));
            if (my_status != QE_OK) fail(my_status, "qe_gather: " + my_error);
            for (int r = 0; r < nranks; r++)
                if (status[r] != QE_OK)
                    fail(status[r], "qe_gather: rank " + std::to_string(r) + " could not allocate its buffers; the exchange was not started");
        }

        // (3) the transfers.  Peer: one send per column buffer -- values, then validity words.  Root: values land at their
        // final offset, bitmap words in the staging areas allocated above (same order as the peer's sends).
        if (rank != root) {
            if (n_me > 0) {
                RcclGroup group(nc);
                for (size_t c = 0; c < ncols; c++) {
                    const OutColumn &src = local->cols[c];
                    const size_t nb = src.type == QE_BOOLEAN ? words_of(n_me) * 8 : width_of(src.type) * (size_t)n_me;
                    QE_NCCL(nc.Send(src.data, nb, kNcclUint8, root, ctx->comm, ctx->stream));
                    if (vsend[c]) QE_NCCL(nc.Send(vsend[c], words_of(n_me) * 8, kNcclUint8, root, ctx->comm, ctx->stream));
                }
                group.end();
            }
            QE_HIP(hipStreamSynchronize(ctx->stream));   // the local result (and the scratch) may be freed by the caller now
            return;
        }
        {
            RcclGroup group(nc);
            size_t si = 0;
            for (int r = 0; r < nranks; r++) {
                const int64_t n = hdr[r].count;
                if (n == 0 || r == root) continue;
                for (size_t c = 0; c < ncols; c++) {
                    OutColumn &dst = res->cols[c];
                    if (dst.type == QE_BOOLEAN)
                        QE_NCCL(nc.Recv(staged[si++].words, words_of(n) * 8, kNcclUint8, r, ctx->comm, ctx->stream));
                    else
                        QE_NCCL(nc.Recv((char *)dst.data + width_of(dst.type) * (size_t)offset[r], width_of(dst.type) * (size_t)n, kNcclUint8, r,
                                        ctx->comm, ctx->stream));
                    if (dst.nullable) QE_NCCL(nc.Recv(staged[si++].words, words_of(n) * 8, kNcclUint8, r, ctx->comm, ctx->stream));
                }
            }
            group.end();
        }
        // the root's own shard: device-to-device, same placement code as the peers' segments
        if (n_me > 0) {
            for (size_t c = 0; c < ncols; c++) {
                const OutColumn &src = local->cols[c];
                OutColumn &dst = res->cols[c];
                if (src.type == QE_BOOLEAN)
                    launch_bitmap_place(ctx->stream, (uint64_t *)dst.data, offset[root], (const uint64_t *)src.data, n_me);
                else
                    QE_HIP(hipMemcpyAsync((char *)dst.data + width_of(src.type) * (size_t)offset[root], src.data,
                                          width_of(src.type) * (size_t)n_me, hipMemcpyDeviceToDevice, ctx->stream));
                if (dst.nullable) launch_bitmap_place(ctx->stream, dst.validity, offset[root], vsend[c], n_me);
            }
        }
        for (const Staged &s_ : staged) launch_bitmap_place(ctx->stream, s_.dst, s_.off, s_.words, s_.n);
        QE_HIP(hipGetLastError());
        QE_HIP(hipStreamSynchronize(ctx->stream));
        *out = res.release();
    });
}

constexpr int kGatherMaxSlices = 16;

// Scan + exchange, OVERLAPPED: the shard is scanned in slices and a slice's rows travel to the root while
// the next slice is scanned.  The root can only place a peer's rows at their final offset if it knows every count in advance,
// so the call starts with a count pre-pass (the filter's columns only: qe_fp_count) and ONE all-gather of every rank's
// per-slice counts; after that nothing waits for a count any more:
//   every rank, slice k:  qe_filter_project on rows [k * S, (k + 1) * S) (compute stream)  ->  transfers of slice k on the COPY
//   stream (peer: one ncclSend per column buffer; root: one ncclRecv per peer and column at the final offset, bitmap words
//   through a staging area and bitmap_place; its own slice device-to-device) while slice k + 1 is scanned.
// Same result as qe_filter_project + qe_gather (rank order, input order), same checks (schema, dictionaries, status word).
// What it costs and buys at cfg 5 (1.25 B rows per GPU, 7 GB into the root): DESIGN.md 6.
int32_t qe_filter_project_gather(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter, const qe_expr *const *projections,
                                 int32_t nproj, int32_t root, int32_t nslices, qe_result **out) {
    if (!ctx || !batch || !out || nproj < 0 || (nproj > 0 && !projections)) return QE_ERR_INVALID_ARG;
    *out = nullptr;
    return guarded_comm(ctx, [&] {
        need_dev(ctx);
        if (!ctx->comm) fail(QE_ERR_COMM, "qe_filter_project_gather: no communicator (qe_comm_init)");
        const int nranks = ctx->comm_nranks, rank = ctx->comm_rank;
        if (root < 0 || root >= nranks) fail(QE_ERR_INVALID_ARG, "qe_filter_project_gather: root out of range");
        if (nproj > kGatherMaxCols) fail(QE_ERR_UNSUPPORTED, "qe_filter_project_gather: more than 16 columns");
        if (batch->schema_only) fail(QE_ERR_INVALID_ARG, "schema-only batch (qe_batch_describe) cannot be executed");
        const int K = std::max(1, std::min(nslices <= 0 ? 8 : nslices, kGatherMaxSlices));
        RcclApi &nc = rccl();
        Scratch sc{ctx, {}};
        // (0) count pre-pass: kept rows of every slice of this shard
        int64_t slice_rows = 0;
        std::vector<int64_t> counts = qe_int_count_slices(ctx, batch, filter, projections, nproj, &slice_rows, K);
        const int ns = (int)counts.size();     // <= K slices really exist (short shards have fewer)
        // (1) one all-gather: the usual header (shape of this rank's result: taken from a zero-row execution of the plan) + counts
        std::unique_ptr<qe_result, std::function<void(qe_result *)>> probe(qe_int_run_fused_slice(ctx, batch, 0, 0, filter, projections, nproj),
                                                                            [ctx](qe_result *r) { qe_result_free(ctx, r); });
        struct SliceHeader {
            GatherHeader h;
            int64_t nslices;
            int64_t count[kGatherMaxSlices];
        };
        SliceHeader mine{};
        mine.h = header_of(probe.get(), true);
        // a zero-row result carries no validity bitmap: nullability is the plan's (OutColumn::nullable)
        mine.h.validity_mask = 0;
        for (size_t c = 0; c < probe->cols.size(); c++)
            if (probe->cols[c].nullable) mine.h.validity_mask |= 1u << c;
        mine.nslices = ns;
        int64_t n_me = 0;
        for (int k = 0; k < ns; k++) { mine.count[k] = counts[(size_t)k]; n_me += counts[(size_t)k]; }
        mine.h.count = n_me;
        std::vector<SliceHeader> hdr((size_t)nranks);
        {
            SliceHeader *d_mine = (SliceHeader *)sc.get(sizeof(SliceHeader));
            SliceHeader *d_all = (SliceHeader *)sc.get(sizeof(SliceHeader) * (size_t)nranks);
            QE_HIP(hipMemcpyAsync(d_mine, &mine, sizeof mine, hipMemcpyHostToDevice, ctx->stream));
            QE_NCCL(nc.AllGather(d_mine, d_all, sizeof(SliceHeader), kNcclUint8, ctx->comm, ctx->stream));
            QE_HIP(hipMemcpyAsync(hdr.data(), d_all, sizeof(SliceHeader) * (size_t)nranks, hipMemcpyDeviceToHost, ctx->stream));
            QE_HIP(hipStreamSynchronize(ctx->stream));
        }
        int64_t total = 0;
        uint32_t any_validity = 0;
        std::vector<int64_t> offset((size_t)nranks, 0);
        const GatherHeader &ref = hdr[(size_t)root].h;
        int max_slices = 0;
        for (int r = 0; r < nranks; r++) {
            if (hdr[r].h.ncols != ref.ncols || hdr[r].h.type_sig != ref.type_sig)
                fail(QE_ERR_INVALID_ARG, "qe_filter_project_gather: rank " + std::to_string(r) + " runs a plan of a different result schema than rank " +
                                             std::to_string(root));
            for (int c = 0; c < ref.ncols; c++)
                if (hdr[r].h.dict_fp[c] != ref.dict_fp[c])
                    fail(QE_ERR_INVALID_ARG, "qe_filter_project_gather: column " + std::to_string(c) + " of rank " + std::to_string(r) +
                                                 " has another dictionary than on rank " + std::to_string(root));
            offset[r] = total;
            total += hdr[r].h.count;
            any_validity |= hdr[r].h.validity_mask;
            max_slices = std::max<int>(max_slices, (int)hdr[r].nslices);
        }
        const size_t ncols = probe->cols.size();
        // (2) allocations, then one status word per rank (as qe_gather)
        std::unique_ptr<qe_result, std::function<void(qe_result *)>> res(nullptr, [ctx](qe_result *r) { free_output(ctx, r); });
        struct Staged { uint64_t *words; int64_t off, n; uint64_t *dst; };
        std::vector<std::vector<Staged>> staged((size_t)max_slices);   // root: per slice, in the order the receives are posted
        int32_t my_status = QE_OK;
        std::string my_error;
        const uint64_t *ones = nullptr;   // validity words of a column that is nullable on another rank only
        try {
            if ((any_validity & ~mine.h.validity_mask) != 0 && n_me > 0) {
                uint64_t *w = (uint64_t *)sc.get(words_of(slice_rows) * 8);
                QE_HIP(hipMemsetAsync(w, 0xff, words_of(slice_rows) * 8, ctx->stream));
                ones = w;
            }
            if (rank == root) {
                res.reset(make_output(ctx, probe.get(), total, any_validity));
                for (int k = 0; k < max_slices; k++) {
                    for (int r = 0; r < nranks; r++) {
                        if (r == root || k >= hdr[r].nslices || hdr[r].count[k] == 0) continue;
                        int64_t off = offset[r];
                        for (int j = 0; j < k; j++) off += hdr[r].count[j];
                        const int64_t n = hdr[r].count[k];
                        for (size_t c = 0; c < ncols; c++) {
                            OutColumn &dst = res->cols[c];
                            if (dst.type == QE_BOOLEAN) staged[(size_t)k].push_back({(uint64_t *)sc.get(words_of(n) * 8), off, n, (uint64_t *)dst.data});
                            if (dst.nullable) staged[(size_t)k].push_back({(uint64_t *)sc.get(words_of(n) * 8), off, n, dst.validity});
                        }
                    }
                }
            }
        } catch (const Error &e) {
            my_status = e.code;
            my_error = e.msg;
        } catch (const std::bad_alloc &) {
            my_status = QE_ERR_OOM;
            my_error = "host out of memory";
        }
        {
            std::vector<int32_t> status((size_t)nranks, 0);
            int32_t *d_mine = (int32_t *)sc.get(sizeof(int32_t));
            int32_t *d_all = (int32_t *)sc.get(sizeof(int32_t) * (size_t)nranks);
            QE_HIP(hipMemcpyAsync(d_mine, &my_status, sizeof my_status, hipMemcpyHostToDevice, ctx->stream));
            QE_NCCL(nc.AllGather(d_mine, d_all, sizeof(int32_t), kNcclUint8, ctx->comm, ctx->stream));
            QE_HIP(hipMemcpyAsync(status.data(), d_all, sizeof(int32_t) * (size_t)nranks, hipMemcpyDeviceToHost, ctx->stream));
            QE_HIP(hipStreamSynchronize(ctx->stream));
            if (my_status != QE_OK) fail(my_status, "qe_filter_project_gather: " + my_error);
            for (int r = 0; r < nranks; r++)
                if (status[r] != QE_OK)
                    fail(status[r], "qe_filter_project_gather: rank " + std::to_string(r) + " could not allocate its buffers; the exchange was not started");
        }
        // (3) slices: scan on the compute stream, transfers on the copy stream
        std::vector<std::unique_ptr<qe_result, std::function<void(qe_result *)>>> parts;   // alive until their transfers have completed
        struct Drain {   // whatever happens, nothing in flight may outlive the buffers it reads or writes
            qe_ctx *c;
            ~Drain() { (void)hipStreamSynchronize(c->copy_stream); }
        } drain{ctx};
        int64_t my_off = offset[(size_t)rank];
        for (int k = 0; k < max_slices; k++) {
            qe_result *part = nullptr;
            int64_t n_k = 0;
            if (k < ns) {
                const int64_t b = (int64_t)k * slice_rows, e = std::min<int64_t>(batch->nrows, b + slice_rows);
                part = qe_int_run_fused_slice(ctx, batch, b, e - b, filter, projections, nproj);   // returns when the slice is complete
                parts.emplace_back(part, [ctx](qe_result *r) { qe_result_free(ctx, r); });
                n_k = part->count;
                if (n_k != counts[(size_t)k])
                    fail(QE_ERR_INTERNAL, "qe_filter_project_gather: slice " + std::to_string(k) + " kept " + std::to_string(n_k) + " rows, the count pass said " +
                                              std::to_string(counts[(size_t)k]));
            }
            if (rank != root) {
                if (n_k > 0) {
                    RcclGroup group(nc);
                    for (size_t c = 0; c < ncols; c++) {
                        const OutColumn &src = part->cols[c];
                        const size_t nb = src.type == QE_BOOLEAN ? words_of(n_k) * 8 : width_of(src.type) * (size_t)n_k;
                        QE_NCCL(nc.Send(src.data, nb, kNcclUint8, root, ctx->comm, ctx->copy_stream));
                        if ((any_validity >> c) & 1u)
                            QE_NCCL(nc.Send(src.validity ? src.validity : ones, words_of(n_k) * 8, kNcclUint8, root, ctx->comm, ctx->copy_stream));
                    }
                    group.end();
                }
                continue;
            }
            // root: this slice of every peer, then its own
            {
                bool any = false;
                for (int r = 0; r < nranks; r++) any = any || (r != root && k < hdr[r].nslices && hdr[r].count[k] > 0);
                if (any) {
                    RcclGroup group(nc);
                    size_t si = 0;
                    for (int r = 0; r < nranks; r++) {
                        if (r == root || k >= hdr[r].nslices || hdr[r].count[k] == 0) continue;
                        int64_t off = offset[r];
                        for (int j = 0; j < k; j++) off += hdr[r].count[j];
                        const int64_t n = hdr[r].count[k];
                        for (size_t c = 0; c < ncols; c++) {
                            OutColumn &dst = res->cols[c];
                            if (dst.type == QE_BOOLEAN)
                                QE_NCCL(nc.Recv(staged[(size_t)k][si++].words, words_of(n) * 8, kNcclUint8, r, ctx->comm, ctx->copy_stream));
                            else
                                QE_NCCL(nc.Recv((char *)dst.data + width_of(dst.type) * (size_t)off, width_of(dst.type) * (size_t)n, kNcclUint8, r,
                                                ctx->comm, ctx->copy_stream));
                            if (dst.nullable) QE_NCCL(nc.Recv(staged[(size_t)k][si++].words, words_of(n) * 8, kNcclUint8, r, ctx->comm, ctx->copy_stream));
                        }
                    }
                    group.end();
                }
                for (const Staged &s_ : staged[(size_t)k]) launch_bitmap_place(ctx->copy_stream, s_.dst, s_.off, s_.words, s_.n);
            }
            if (n_k > 0) {
                for (size_t c = 0; c < ncols; c++) {
                    const OutColumn &src = part->cols[c];
                    OutColumn &dst = res->cols[c];
                    if (src.type == QE_BOOLEAN)
                        launch_bitmap_place(ctx->copy_stream, (uint64_t *)dst.data, my_off, (const uint64_t *)src.data, n_k);
                    else
                        QE_HIP(hipMemcpyAsync((char *)dst.data + width_of(src.type) * (size_t)my_off, src.data, width_of(src.type) * (size_t)n_k,
                                              hipMemcpyDeviceToDevice, ctx->copy_stream));
                    if (dst.nullable) launch_bitmap_place(ctx->copy_stream, dst.validity, my_off, src.validity ? src.validity : ones, n_k);
                }
                my_off += n_k;
            }
        }
        QE_HIP(hipGetLastError());
        QE_HIP(hipStreamSynchronize(ctx->copy_stream));
        if (rank == root) *out = res.release();
    });
}

}  // extern "C"

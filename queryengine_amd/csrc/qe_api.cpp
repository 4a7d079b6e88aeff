// qe_api.cpp -- the C ABI of libqe_hip.so (include/qe_hip.h): contexts, HBM-resident
// batches, expression handles, the fused filter+project / filter+aggregate calls, results.
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <sstream>
#include <thread>

#include "qe_internal.h"
#include "qe_kernels.h"
#include "qe_pernode.h"
#include "qe_pernode_kernels.h"

namespace qe {

void fail(int32_t code, const std::string &msg) { throw Error{code, msg}; }

void hip_check(hipError_t e, const char *what, const char *file, int line) {
    if (e == hipSuccess) return;
    std::ostringstream s;
    s << what << " failed: " << hipGetErrorString(e) << " (" << (int)e << ") at " << file << ":" << line;
    fail(e == hipErrorOutOfMemory ? QE_ERR_OOM : QE_ERR_HIP, s.str());
}

uint64_t DictData::next_id() {
    static std::atomic<uint64_t> counter{0};
    return ++counter;
}

// ---- pool ------------------------------------------------------------------------------
void *Pool::alloc(size_t bytes) {
    bytes = std::max<size_t>(256, (bytes + 255) & ~size_t(255));
    auto it = free_.lower_bound(bytes);
    if (it != free_.end() && it->first <= bytes + bytes / 4) {
        void *p = it->second;
        bytes_cached -= it->first;
        bytes_in_use += it->first;
        live_[p] = it->first;
        free_.erase(it);
        return p;
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipErrorOutOfMemory && !free_.empty()) {
        (void)hipGetLastError();
        trim_all();
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        fail(QE_ERR_OOM, "hipMalloc of " + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
    }
    live_[p] = bytes;
    bytes_in_use += bytes;
    return p;
}

void Pool::release(void *p) {
    if (!p) return;
    auto it = live_.find(p);
    if (it == live_.end()) return;
    bytes_in_use -= it->second;
    bytes_cached += it->second;
    free_.emplace(it->second, p);
    live_.erase(it);
}

void Pool::trim_all() {
    for (auto &kv : free_) (void)hipFree(kv.second);
    free_.clear();
    bytes_cached = 0;
}
void Pool::trim() { trim_all(); }

void *PinnedPool::alloc(size_t bytes) {
    bytes = (std::max<size_t>(bytes, 64) + 4095) & ~(size_t)4095;
    auto it = free_.lower_bound(bytes);
    if (it != free_.end() && it->first <= bytes + bytes / 4 + (1u << 20)) {   // a cached buffer that is not wastefully large
        void *p = it->second;
        live_[p] = it->first;
        free_.erase(it);
        return p;
    }
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        trim();   // give the cached buffers back and try once more
        e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            fail(QE_ERR_OOM, "hipHostMalloc of " + std::to_string(bytes) + " bytes of pinned host memory failed: " + hipGetErrorString(e));
        }
    }
    live_[p] = bytes;
    return p;
}

void PinnedPool::release(void *p) {
    if (!p) return;
    auto it = live_.find(p);
    if (it == live_.end()) return;
    free_.emplace(it->second, p);
    live_.erase(it);
}

void PinnedPool::trim() {
    for (auto &kv : free_) (void)hipHostFree(kv.second);
    free_.clear();
}

}  // namespace qe

using namespace qe;

static thread_local std::string g_create_error;

template <typename F>
static int32_t guarded(qe_ctx *ctx, F &&f) {
    try {
        f();
        return QE_OK;
    } catch (const Error &e) {
        (ctx ? ctx->last_error : g_create_error) = e.msg;
        return e.code;
    } catch (const std::bad_alloc &) {
        (ctx ? ctx->last_error : g_create_error) = "host out of memory";
        return QE_ERR_OOM;
    } catch (const std::exception &e) {
        (ctx ? ctx->last_error : g_create_error) = e.what();
        return QE_ERR_INTERNAL;
    }
}

static void need_device(const qe_ctx *ctx) {
    if (ctx->device < 0)
        fail(QE_ERR_HIP, "planning-only context (QE_DEVICE_NONE): this call needs a HIP device; libqe_hip has no CPU fallback");
    QE_HIP(hipSetDevice(ctx->device));
}

static size_t type_width(int t) {
    switch (t) {
    case QE_DOUBLE: case QE_INT64: return 8;
    case QE_INT32: case QE_STRING: return 4;
    default: return 0;
    }
}
static size_t column_bytes(int t, int64_t n) {
    return t == QE_BOOLEAN ? (size_t)((n + 63) / 64) * 8 : type_width(t) * (size_t)n;
}
static size_t bitmap_bytes(int64_t n) { return (size_t)((n + 63) / 64) * 8; }

static std::string default_cache_dir() {
    if (const char *e = std::getenv("QE_JIT_CACHE_DIR")) return e;
    Dl_info info;
    if (dladdr((void *)&default_cache_dir, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        size_t s = p.rfind('/');
        if (s != std::string::npos) return p.substr(0, s) + "/jit_cache";
    }
    return "jit_cache";
}

extern "C" {

int32_t qe_abi_version(void) { return QE_ABI_VERSION; }

const char *qe_last_error(const qe_ctx *ctx) { return ctx ? ctx->last_error.c_str() : g_create_error.c_str(); }

int32_t qe_ctx_create(int32_t device, const qe_options *opts, qe_ctx **out) {
    if (!out) return QE_ERR_INVALID_ARG;
    *out = nullptr;
    qe_ctx *ctx = nullptr;
    int32_t st = guarded(nullptr, [&] {
        if (device == QE_DEVICE_NONE) {   // planning-only context: no HIP call at all
            ctx = new qe_ctx();
            ctx->device = device;
            if (opts) std::memcpy(&ctx->opts, opts, std::min<size_t>(opts->struct_size, sizeof(qe_options)));
            ctx->opts.struct_size = sizeof(qe_options);
            ctx->jit.reset(new Jit(ctx->opts.jit_cache_dir ? ctx->opts.jit_cache_dir : default_cache_dir()));
            ctx->opts.jit_cache_dir = nullptr;
            return;
        }
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev == 0) {
            (void)hipGetLastError();
            fail(QE_ERR_HIP, std::string("no HIP device available: ") + hipGetErrorString(e) +
                                 " (libqe_hip has no CPU fallback)");
        }
        if (device < 0 || device >= ndev) fail(QE_ERR_INVALID_ARG, "device ordinal out of range");
        ctx = new qe_ctx();
        ctx->device = device;
        if (opts) {
            size_t n = std::min<size_t>(opts->struct_size, sizeof(qe_options));
            std::memcpy(&ctx->opts, opts, n);
        }
        ctx->opts.struct_size = sizeof(qe_options);
        QE_HIP(hipSetDevice(device));
        QE_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        QE_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        QE_HIP(hipEventCreate(&ctx->ev0));
        QE_HIP(hipEventCreate(&ctx->ev1));
        QE_HIP(hipMalloc((void **)&ctx->d_ctrl, 256));
        QE_HIP(hipHostMalloc((void **)&ctx->h_ctrl, 256, hipHostMallocDefault));
        ctx->jit.reset(new Jit(ctx->opts.jit_cache_dir ? ctx->opts.jit_cache_dir : default_cache_dir()));
        ctx->opts.jit_cache_dir = nullptr;
    });
    if (st != QE_OK) {
        delete ctx;
        return st;
    }
    *out = ctx;
    return QE_OK;
}

void qe_ctx_destroy(qe_ctx *ctx) {
    if (!ctx) return;
    if (ctx->device < 0) {
        delete ctx;
        return;
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    while (!ctx->host_results.empty()) qe_host_result_free(ctx, ctx->host_results.back());
    ctx->pinned.trim();
    qe_comm_destroy(ctx);
    ctx->plans.clear();
    ctx->jit.reset();
    ctx->pool.trim();
    if (ctx->d_ctrl) (void)hipFree(ctx->d_ctrl);
    if (ctx->h_ctrl) (void)hipHostFree(ctx->h_ctrl);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    delete ctx;
}

int32_t qe_ctx_set_exec_mode(qe_ctx *ctx, int32_t m) {
    if (!ctx || (m != QE_EXEC_FUSED && m != QE_EXEC_PER_NODE)) return QE_ERR_INVALID_ARG;
    ctx->opts.exec_mode = m;
    return QE_OK;
}
int32_t qe_ctx_set_cmp_semantics(qe_ctx *ctx, int32_t m) {
    if (!ctx || (m != QE_CMP_TOTAL_ORDER && m != QE_CMP_IEEE)) return QE_ERR_INVALID_ARG;
    ctx->opts.cmp_semantics = m;
    return QE_OK;
}
int32_t qe_ctx_kernel_time(qe_ctx *ctx, double *last_ms, double *total_ms, int64_t *launches) {
    if (!ctx) return QE_ERR_INVALID_ARG;
    if (last_ms) *last_ms = ctx->last_ms;
    if (total_ms) *total_ms = ctx->total_ms;
    if (launches) *launches = ctx->launches;
    return QE_OK;
}
int32_t qe_ctx_reset_kernel_time(qe_ctx *ctx) {
    if (!ctx) return QE_ERR_INVALID_ARG;
    ctx->last_ms = ctx->total_ms = 0.0;
    ctx->launches = 0;
    return QE_OK;
}
int32_t qe_ctx_last_form(const qe_ctx *ctx) { return ctx ? ctx->last_form : -1; }
int32_t qe_ctx_synchronize(qe_ctx *ctx) {
    if (!ctx) return QE_ERR_INVALID_ARG;
    return guarded(ctx, [&] { need_device(ctx); QE_HIP(hipStreamSynchronize(ctx->stream)); });
}
int32_t qe_ctx_trim(qe_ctx *ctx) {
    if (!ctx) return QE_ERR_INVALID_ARG;
    return guarded(ctx, [&] {
        need_device(ctx);
        QE_HIP(hipStreamSynchronize(ctx->stream));
        ctx->pool.trim();
        ctx->pinned.trim();
    });
}

// ---- dictionaries ---------------------------------------------------------------------------
int32_t qe_dict_create(qe_ctx *ctx, int32_t nentries, const char *const *utf8, qe_dict **out) {
    if (!ctx || !out || nentries < 0 || (nentries > 0 && !utf8)) return QE_ERR_INVALID_ARG;
    return guarded(ctx, [&] {
        auto d = std::make_shared<DictData>();
        d->entries.reserve(nentries);
        for (int32_t i = 0; i < nentries; i++) {
            if (!utf8[i]) fail(QE_ERR_INVALID_ARG, "null dictionary entry");
            d->entries.emplace_back(utf8[i]);
            d->index.emplace(d->entries.back(), i);   // first occurrence wins
        }
        *out = new qe_dict{d};
    });
}
int32_t qe_dict_size(const qe_dict *dict) { return dict && dict->d ? (int32_t)dict->d->entries.size() : 0; }
const char *qe_dict_entry(const qe_dict *dict, int32_t code) {
    if (!dict || !dict->d || code < 0 || code >= (int32_t)dict->d->entries.size()) return nullptr;
    return dict->d->entries[code].c_str();
}
void qe_dict_free(qe_ctx *, qe_dict *dict) { delete dict; }

// ---- batches ------------------------------------------------------------------------------------
static void free_batch(qe_ctx *ctx, qe_batch *b) {
    if (!b) return;
    for (auto &c : b->cols)
        if (c.owned) {
            ctx->pool.release(c.data);
            ctx->pool.release(c.validity);
        }
    delete b;
}

static void check_col_desc(const qe_col_desc &d, int64_t nrows) {
    if (d.type < QE_STRING || d.type > QE_INT32) fail(QE_ERR_INVALID_ARG, "bad column type");
    if (nrows > 0 && !d.data) fail(QE_ERR_INVALID_ARG, "null column data");
    if (d.type == QE_STRING && (!d.dict || !d.dict->d)) fail(QE_ERR_INVALID_ARG, "STRING column needs a dictionary");
}

int32_t qe_batch_create(qe_ctx *ctx, int64_t nrows, int32_t ncols, const qe_col_desc *cols, qe_batch **out) {
    if (!ctx || !out || nrows < 0 || ncols < 0 || (ncols > 0 && !cols)) return QE_ERR_INVALID_ARG;
    *out = nullptr;
    qe_batch *b = nullptr;
    int32_t st = guarded(ctx, [&] {
        need_device(ctx);
        b = new qe_batch();
        b->nrows = nrows;
        for (int32_t j = 0; j < ncols; j++) {
            check_col_desc(cols[j], nrows);
            Column c;
            c.type = cols[j].type;
            if (c.type == QE_STRING) c.dict = cols[j].dict->d;
            size_t nb = column_bytes(c.type, nrows);
            c.data = ctx->pool.alloc(std::max<size_t>(nb, 16));
            b->cols.push_back(c);
            if (nb) QE_HIP(hipMemcpyAsync(c.data, cols[j].data, nb, hipMemcpyHostToDevice, ctx->stream));
            if (cols[j].validity && nrows > 0) {
                b->cols.back().validity = (uint64_t *)ctx->pool.alloc(bitmap_bytes(nrows));
                QE_HIP(hipMemcpyAsync(b->cols.back().validity, cols[j].validity, bitmap_bytes(nrows),
                                      hipMemcpyHostToDevice, ctx->stream));
            }
        }
        QE_HIP(hipStreamSynchronize(ctx->stream));   // host buffers may be reused by the caller
    });
    if (st != QE_OK) {
        free_batch(ctx, b);
        return st;
    }
    *out = b;
    return QE_OK;
}

int32_t qe_batch_wrap_device(qe_ctx *ctx, int64_t nrows, int32_t ncols, const qe_col_desc *cols, qe_batch **out) {
    if (!ctx || !out || nrows < 0 || ncols < 0 || (ncols > 0 && !cols)) return QE_ERR_INVALID_ARG;
    *out = nullptr;
    qe_batch *b = nullptr;
    int32_t st = guarded(ctx, [&] {
        b = new qe_batch();
        b->nrows = nrows;
        for (int32_t j = 0; j < ncols; j++) {
            check_col_desc(cols[j], nrows);
            if (((uintptr_t)cols[j].data & 15) || ((uintptr_t)cols[j].validity & 7))
                fail(QE_ERR_INVALID_ARG, "device column pointers must be 16-byte aligned (validity: 8)");
            Column c;
            c.type = cols[j].type;
            c.data = const_cast<void *>(cols[j].data);
            c.validity = const_cast<uint64_t *>(cols[j].validity);
            if (c.type == QE_STRING) c.dict = cols[j].dict->d;
            c.owned = false;
            b->cols.push_back(c);
        }
    });
    if (st != QE_OK) {
        delete b;
        return st;
    }
    *out = b;
    return QE_OK;
}

int32_t qe_batch_describe(qe_ctx *ctx, int64_t nrows, int32_t ncols, const qe_col_desc *cols, qe_batch **out) {
    if (!ctx || !out || nrows < 0 || ncols < 0 || (ncols > 0 && !cols)) return QE_ERR_INVALID_ARG;
    *out = nullptr;
    qe_batch *b = nullptr;
    int32_t st = guarded(ctx, [&] {
        b = new qe_batch();
        b->nrows = nrows;
        b->schema_only = true;
        for (int32_t j = 0; j < ncols; j++) {
            if (cols[j].type < QE_STRING || cols[j].type > QE_INT32) fail(QE_ERR_INVALID_ARG, "bad column type");
            if (cols[j].type == QE_STRING && (!cols[j].dict || !cols[j].dict->d))
                fail(QE_ERR_INVALID_ARG, "STRING column needs a dictionary");
            Column c;
            c.type = cols[j].type;
            c.validity = cols[j].validity ? (uint64_t *)(uintptr_t)8 : nullptr;   // nullability marker only
            if (c.type == QE_STRING) c.dict = cols[j].dict->d;
            c.owned = false;
            b->cols.push_back(c);
        }
    });
    if (st != QE_OK) {
        delete b;
        return st;
    }
    *out = b;
    return QE_OK;
}

int32_t qe_batch_generate(qe_ctx *ctx, uint64_t seed, int64_t row_begin, int64_t nrows, int32_t ncols,
                          const qe_gen_spec *specs, qe_batch **out) {
    if (!ctx || !out || nrows < 0 || row_begin < 0 || ncols < 0 || (ncols > 0 && !specs)) return QE_ERR_INVALID_ARG;
    *out = nullptr;
    qe_batch *b = nullptr;
    int32_t st = guarded(ctx, [&] {
        need_device(ctx);
        b = new qe_batch();
        b->nrows = nrows;
        for (int32_t j = 0; j < ncols; j++) {
            const qe_gen_spec &g = specs[j];
            Column c;
            switch (g.kind) {
            case QE_GEN_I64_MOD: case QE_GEN_I64_ROWID: c.type = QE_INT64; break;
            case QE_GEN_I32_MOD: c.type = QE_INT32; break;
            case QE_GEN_DICT_MOD:
                c.type = QE_STRING;
                if (!g.dict || !g.dict->d) fail(QE_ERR_INVALID_ARG, "QE_GEN_DICT_MOD needs a dictionary");
                if (g.modulus > g.dict->d->entries.size() || g.offset != 0)
                    fail(QE_ERR_INVALID_ARG, "QE_GEN_DICT_MOD codes exceed the dictionary");
                c.dict = g.dict->d;
                break;
            case QE_GEN_F64_UNIT: case QE_GEN_F64_MOD: case QE_GEN_F64_STEP: case QE_GEN_F64_PRICE: c.type = QE_DOUBLE; break;
            default: fail(QE_ERR_INVALID_ARG, "bad generator kind");
            }
            if (g.kind != QE_GEN_F64_UNIT && g.kind != QE_GEN_F64_PRICE && g.kind != QE_GEN_I64_ROWID && g.modulus == 0)
                fail(QE_ERR_INVALID_ARG, "generator modulus must be > 0");
            c.data = ctx->pool.alloc(std::max<size_t>(column_bytes(c.type, nrows), 16));
            if (g.null_pct > 0 && nrows > 0) c.validity = (uint64_t *)ctx->pool.alloc(bitmap_bytes(nrows));
            b->cols.push_back(c);
            launch_generate(ctx->stream, g, seed, row_begin, nrows, c.data, c.validity);
        }
        QE_HIP(hipGetLastError());
        QE_HIP(hipStreamSynchronize(ctx->stream));
    });
    if (st != QE_OK) {
        free_batch(ctx, b);
        return st;
    }
    *out = b;
    return QE_OK;
}

int64_t qe_batch_nrows(const qe_batch *b) { return b ? b->nrows : -1; }
int32_t qe_batch_ncols(const qe_batch *b) { return b ? (int32_t)b->cols.size() : -1; }
int32_t qe_batch_column_type(const qe_batch *b, int32_t col) {
    return (b && col >= 0 && col < (int32_t)b->cols.size()) ? b->cols[col].type : -1;
}

int32_t qe_batch_column_to_host(qe_ctx *ctx, const qe_batch *b, int32_t col, int64_t row_begin, int64_t nrows,
                                void *data_out, uint64_t *validity_out) {
    if (!ctx || !b || col < 0 || col >= (int32_t)b->cols.size() || row_begin < 0 || nrows < 0 ||
        row_begin + nrows > b->nrows || (row_begin & 63))
        return QE_ERR_INVALID_ARG;
    return guarded(ctx, [&] {
        need_device(ctx);
        const Column &c = b->cols[col];
        if (nrows == 0) return;
        if (data_out) {
            if (c.type == QE_BOOLEAN)
                QE_HIP(hipMemcpyAsync(data_out, (const char *)c.data + (row_begin / 64) * 8, bitmap_bytes(nrows),
                                      hipMemcpyDeviceToHost, ctx->stream));
            else
                QE_HIP(hipMemcpyAsync(data_out, (const char *)c.data + type_width(c.type) * (size_t)row_begin,
                                      type_width(c.type) * (size_t)nrows, hipMemcpyDeviceToHost, ctx->stream));
        }
        if (validity_out) {
            if (c.validity)
                QE_HIP(hipMemcpyAsync(validity_out, c.validity + row_begin / 64, bitmap_bytes(nrows),
                                      hipMemcpyDeviceToHost, ctx->stream));
            else
                std::memset(validity_out, 0xff, bitmap_bytes(nrows));
        }
        QE_HIP(hipStreamSynchronize(ctx->stream));
    });
}

void qe_batch_free(qe_ctx *ctx, qe_batch *b) {
    if (!ctx) return;
    free_batch(ctx, b);
}

// ---- expressions ---------------------------------------------------------------------------------
int32_t qe_expr_compile(qe_ctx *ctx, const uint8_t *program, size_t len, qe_expr **out) {
    if (!ctx || !out) return QE_ERR_INVALID_ARG;
    *out = nullptr;
    return guarded(ctx, [&] {
        Expr e = decode_program(program, len);
        *out = new qe_expr{std::move(e)};
    });
}
int32_t qe_expr_result_type(const qe_expr *e) { return e ? e->e.nodes[e->e.root].type : -1; }
void qe_expr_free(qe_ctx *, qe_expr *e) { delete e; }


int32_t qe_stream_read_write_time(qe_ctx *ctx, int64_t nbytes, int32_t write_every, int32_t reps, double *out_ms,
                                  double *out_written_bytes) {
    if (!ctx || nbytes < 4096 || reps < 1 || !out_ms) return QE_ERR_INVALID_ARG;
    return guarded(ctx, [&] {
        need_device(ctx);
        // write_every packs: [write_every % 1000] + 1000 * window_period_us + 1e7 * window_len_us
        const int we = write_every % 1000;
        const int64_t dst_bytes = we > 0 ? nbytes / 16 / we + (1 << 22) : (1 << 22);
        void *buf = ctx->pool.alloc((size_t)nbytes);
        void *dst = ctx->pool.alloc((size_t)dst_bytes);
        struct G { qe_ctx *c; void *a, *b; ~G() { c->pool.release(a); c->pool.release(b); } } g{ctx, buf, dst};
        QE_HIP(hipMemsetAsync(buf, 0x5a, (size_t)nbytes, ctx->stream));
        double best = 1e30;
        for (int r = 0; r <= reps; r++) {
            QE_HIP(hipEventRecord(ctx->ev0, ctx->stream));
            launch_stream_read_write(ctx->stream, buf, nbytes, (unsigned long long *)(ctx->d_ctrl + 8), dst, dst_bytes,
                                     write_every % 1000, ((write_every / 1000) % 10000) * 100, (write_every / 10000000) * 100,
                                     std::getenv("QE_CALIB_BLOCKS") ? std::atoi(std::getenv("QE_CALIB_BLOCKS")) : 1);
            QE_HIP(hipEventRecord(ctx->ev1, ctx->stream));
            QE_HIP(hipStreamSynchronize(ctx->stream));
            float ms = 0.f;
            QE_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
            if (r > 0) best = std::min(best, (double)ms);
        }
        *out_ms = best;
        if (out_written_bytes) {
            // iterations per wave = nvec / (8 * threads); one 512-byte block per write_every iterations
            const double iters = (double)(nbytes / 16) / (8.0 * 256 * 8 * 256);
            *out_written_bytes = we > 0 ? std::floor(iters / we) * 512.0 * (256 * 8 * 4) : 0.0;
        }
    });
}

}  // extern "C"

// ---- plans --------------------------------------------------------------------------------------------
namespace {

// a plan that kept at least this share of its rows last time runs the dense single-pass kernel next time (measured
// crossover against the LDS-ring kernel on cfg 2, 1 B rows: 10 % 4.14 vs 4.23 ms, 25 % 5.7 vs 4.5 ms)
constexpr double kDenseFromSelectivity = 0.12;
// a plan that kept at most this share of its rows last time runs the LOCAL form next time: scan without any inter-wave
// dependency into per-chunk slots, then a scan over the counts and one move
constexpr double kLocalUpToSelectivity = 0.03;
constexpr int64_t kSampleFromRows = 8ll << 20;   // batches from here on sample their selectivity before the first execution of a plan
constexpr int kScatterWgsPerCu = 2;   // workgroups per CU of the partitioned group-by's scatter pass

FusedGeometry geometry_of(const qe_ctx *ctx) {
    FusedGeometry g;
    const int t = ctx->opts.tuning[0], u = ctx->opts.tuning[1];
    if (t == 64 || t == 128 || t == 256 || t == 512 || t == 1024) g.threads = t;
    const int mw = ctx->opts.tuning[3] / 100;    // tuning[3] = 100 * min_waves + blocks_per_cu
    if (mw >= 1 && mw <= 8) g.min_waves = mw;
    if (u >= 1 && u <= 16) g.unroll = u;
    const int spc = ctx->opts.tuning[4] % 10000;   // tuning[4] = subs_per_chunk + 10000 * (LDS ring entries / 256)
    if (spc >= 1 && spc <= 4096) g.subs_per_chunk = spc;
    const int ringk = ctx->opts.tuning[4] / 10000;
    if (ringk == 1 || ringk == 2 || ringk == 4) g.ring_entries = 256 * ringk;
    const int lbk = ctx->opts.tuning[6] % 100;   // tuning[6] = lookback_k + 100 * gate_period_log2 + 10000 * gate_width_log2
    if (lbk >= 1 && lbk <= 16) g.lookback_k = lbk;
    const int gp = (ctx->opts.tuning[6] / 100) % 100, gw = ctx->opts.tuning[6] / 10000;
    if (gp >= 8 && gp <= 24 && gw >= 4 && gw < gp) { g.gate_period_log2 = gp; g.gate_width_log2 = gw; }
    const int pm = ctx->opts.tuning[2] / 10;   // tuning[2] = 10 * (prio_mode + 1) + nt ; 0 = default
    if (pm >= 1 && pm <= 4) g.prio_mode = pm - 1;   // 0 none, 1 per sub-tile mod 3, 2 per chunk mod 3, 3 per sub-tile mod 2
    const int nb = ctx->opts.tuning[7] / 100;   // tuning[7] = 100 * nbuf + (stagger / resolve_at code)
    if (nb >= 2 && nb <= 8) g.nbuf = nb;
    const int st7 = ctx->opts.tuning[7] % 100;
    if (st7 == 2) g.stagger = 0;
    if (st7 == 1 && g.subs_per_chunk % 16 == 0) g.stagger = 1;
    if (st7 >= 10) g.resolve_at = st7 - 10;   // 10 + n: resolve n sub-tiles into the next chunk
    return g;
}

std::shared_ptr<Plan> get_plan(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter,
                               const qe_expr *const *projs, int32_t nproj, const int32_t *agg_fns, bool load,
                               const qe_expr *const *keys = nullptr, int32_t nkeys = 0, int geo_cand = 0, bool dense = false,
                               const std::vector<int> *conj_order = nullptr, int hp_parts = 0, int hp_shift = 0) {
    if (nproj < 0 || (nproj > 0 && !projs)) fail(QE_ERR_INVALID_ARG, "bad projection list");
    CodegenInput in;
    in.filter = filter ? &filter->e : nullptr;
    for (int32_t i = 0; i < nproj; i++) {
        if (!projs[i]) fail(QE_ERR_INVALID_ARG, "null projection");
        in.projections.push_back(&projs[i]->e);
        if (agg_fns) {
            if (agg_fns[i] < QE_AGG_MIN || agg_fns[i] > QE_AGG_AVG) fail(QE_ERR_UNSUPPORTED, "unsupported aggregation function");
            in.agg_fns.push_back(agg_fns[i]);
        }
    }
    for (int32_t i = 0; i < nkeys; i++) {
        if (!keys || !keys[i]) fail(QE_ERR_INVALID_ARG, "null group key");
        in.group_keys.push_back(&keys[i]->e);
    }
    in.cmp_semantics = ctx->opts.cmp_semantics;
    if (conj_order) in.conj_order = *conj_order;
    in.hp_parts = hp_parts;
    // records in 128-byte lines pad in units of 6 / 4 / 3 / 2 where {header, words} records of 32 bytes pad in units of 4: they win
    // while a scatter tile holds runs of several records per partition (1 B rows, SELECT k, MIN(v), MAX(v), lines / records: 256
    // partitions, 100 000 keys 16.9 / 18.3 ms; 512 partitions, 500 000 keys 18.3 / 20.6, 1 M keys 24.6 / 26.7); with 1024 partitions
    // the stage of a tile's lines no longer fits a 4 Ki-row tile.  Debug bit 33554432: always {header, words} records.
    static const int lines_maxp = std::getenv("QE_HP_LINES_MAXP") ? std::atoi(std::getenv("QE_HP_LINES_MAXP")) : 512;
    in.hp_lines = (ctx->opts.tuning[5] & 33554432) == 0 && hp_parts <= lines_maxp ? 1 : 0;
    in.hp_shift = hp_shift;
    in.geo = geometry_of(ctx);
    const bool wide = geo_cand == 1;
    if (wide) {   // the second candidate of the geometry choice (qe_ctx::geo_choice)
        in.geo.unroll = 16;
        in.geo.subs_per_chunk = 4;
        in.geo.ring_entries = 512;
    } else if (geo_cand == 2) {
        // the third candidate (round 3): the default sub-tile, 8 Ki-row chunks whose kept rows fit 512-entry rings, two waves per
        // workgroup.  cfg 2, 1 B rows, three boxes: default / wide / this = 3.56 / 3.29 / 3.19, 3.47 / 3.47 / 3.19 and
        // (a fast box) 3.35 / 3.13 / 3.20 ms -- which one wins depends on the box, so it is measured like the other two
        in.geo.unroll = 8;
        in.geo.subs_per_chunk = 8;
        in.geo.ring_entries = 512;
        in.geo.threads = 128;
    }
    in.nontemporal = ctx->opts.tuning[2] % 10 == 2 ? 0 : 1;
    in.vec_stores = ctx->opts.tuning[2] % 10 == 5 ? 1 : 0;   // tuning[2] % 10 == 5: 16-byte output stores (measurement)
    in.nt_stores = ctx->opts.tuning[2] % 10 == 3 ? 0 : ctx->opts.tuning[2] % 10 == 4 ? 2 : 1;   // tuning[2] % 10: 2 plain loads, 3 plain output stores, 4 nt spill stores too
    in.debug_mask = ctx->opts.tuning[5] & 255;   // bits 256.. are host-side switches, not ablation builds
    in.staged = (ctx->opts.tuning[5] & 2048) == 0;
    in.prefetch = (ctx->opts.tuning[5] & 2097152) ? 0 : (ctx->opts.tuning[5] & 4194304) ? 2 : 1;   // debug bits: 2097152 no stage-0 prefetch, 4194304 always
    in.dense = dense && filter != nullptr && !agg_fns;
    if (in.dense) {   // the dense kernel has its own shape: a workgroup per tile of QE_WAVES sub-tiles, one tile parked in LDS
        in.geo.subs_per_chunk = 1;
        in.geo.stagger = 0;
        // measured on cfg 2, 1 B rows (tools/sel_sweep.py): 4 waves x 1024 rows = 4 Ki-row tiles (244 K tickets, 64 KiB of LDS
        // for 16-byte rows, two workgroups per CU) ran 4.0 - 6.4 ms from 1 % to 100 %; 2 Ki-row tiles are bound by the
        // single-address ticket rate (~77 tickets/us: 6.3 ms whatever the selectivity), 8 / 16 Ki-row tiles leave one
        // workgroup per CU (5 - 15 % slower)
        if (ctx->opts.tuning[0] == 0) in.geo.threads = 256;
        if (ctx->opts.tuning[1] == 0) in.geo.unroll = 8;
    }
    in.filter_load_stages = (ctx->opts.tuning[5] & 4096) ? 1 : 0;   // bit 4096: every filter column in the first load stage   // bit 2048: load every column for every row (no late materialisation)
    std::ostringstream key;
    key << "m" << (agg_fns ? 1 : 0) << (in.dense ? "D" : "") << "c" << in.cmp_semantics << "t" << in.geo.threads << "u" << in.geo.unroll << "s"
        << in.geo.subs_per_chunk << "n" << in.nontemporal << in.nt_stores << in.vec_stores << "L" << (in.staged ? 1 : 0) << "P" << in.prefetch << "." << in.filter_load_stages << "k" << in.geo.lookback_k << "G" << in.geo.gate_period_log2 << "." << in.geo.gate_width_log2 << "d" << in.debug_mask << "r" << in.geo.resolve_at << "g" << in.geo.stagger << "w" << in.geo.min_waves << "p" << in.geo.prio_mode << "b" << in.geo.nbuf << "R" << in.geo.ring_entries << "|";
    for (const Column &c : batch->cols) {
        in.schema.push_back(BoundColumn{c.type, c.validity != nullptr, c.dict});
        key << c.type << (c.validity ? 'n' : 'v') << (c.dict ? c.dict->id : 0) << ",";   // the dictionary's serial number, not its address
    }
    auto add_prog = [&](const Expr *e) {
        key << "|";
        if (e) key.write((const char *)e->program.data(), (std::streamsize)e->program.size());
    };
    add_prog(in.filter);
    for (const Expr *e : in.projections) add_prog(e);
    for (const Expr *e : in.group_keys) {
        key << "|k";
        add_prog(e);
    }
    if (agg_fns)
        for (int a : in.agg_fns) key << "|a" << a;
    if (in.hp_parts) key << "|H" << in.hp_parts << "." << in.hp_shift << "." << in.hp_lines;
    if (!in.conj_order.empty()) {
        key << "|O";
        for (int o : in.conj_order) key << o << ".";
    }
    const std::string k = key.str();
    auto it = ctx->plans.find(k);
    if (it != ctx->plans.end() && (it->second->kernel.fn || !load)) return it->second;
    auto plan = std::make_shared<Plan>();
    plan->cg = generate_fused_source(in);
    if (wide && !in.dense && !agg_fns && in.group_keys.empty()) {
        // A plan that reads few bytes per row (cfg 4: a 4-byte code and an 8-byte value) pays the per-chunk work -- ticket,
        // descriptors, look-back -- on few bytes: its second candidate keeps the 16 load groups but hands out 32 Ki-row chunks
        // (16 sub-tiles) with the default rings instead of 8 Ki-row chunks.  Measured on cfg 4, 1 B rows: default 1.02 ms,
        // 16 groups x 4 sub-tiles 0.97 ms, 16 groups x 16 sub-tiles 0.80 ms.  The choice itself stays measured (best of 3).
        size_t inbytes = 0;
        for (int c : plan->cg.used_cols) {
            const int t = in.schema[(size_t)c].type;
            inbytes += t == QE_BOOLEAN ? 0 : (t == QE_DOUBLE || t == QE_INT64) ? 8 : 4;
        }
        if (inbytes <= 16) {
            const FusedGeometry dflt = geometry_of(ctx);
            in.geo.subs_per_chunk = dflt.subs_per_chunk;
            in.geo.ring_entries = dflt.ring_entries;
            plan->cg = generate_fused_source(in);
        }
    }
    if (!in.group_keys.empty() && !plan->cg.hashed && !plan->cg.table_in_lds && ctx->opts.tuning[0] == 0 &&
        (size_t)plan->cg.ngroups * plan->cg.table_words * 8 <= 144 * 1024) {
        in.geo.threads = 1024;   // the table fits ONE workgroup's LDS: 16 waves per CU share it (see table_in_lds)
        plan->cg = generate_fused_source(in);
    }
    if (plan->cg.hp && ctx->opts.tuning[1] == 0 && ctx->opts.tuning[0] == 0) {
        // the scatter pass sorts a workgroup tile's records in ONE LDS stage: 8 waves x 512 rows x 32-byte records = 128 KiB (one
        // workgroup per CU) was the fastest of the shapes measured on 100 000 DOUBLE keys, 512 partitions: 4 waves x 1024 rows
        // 31.4 ms, 4 x 512 29.0, 8 x 512 25.6, 8 x 256 29.3; wider records take fewer rows per wave
        in.geo.threads = 512;
        int u = 4;
        auto lines_lds = [&](int uu) {   // the stage holds the tile's LINES: every partition's run padded to whole lines, in the worst case
            const size_t R = (size_t)std::max(1, plan->cg.hp_line_recs), P = (size_t)plan->cg.nparts;
            const size_t tile = (size_t)(in.geo.threads / 64) * 128 * uu, lines = (tile + (P <= 256 ? 2 : 1) * (R - 1) * P + R - 1) / R;
            return lines * 132 + ((P + 3) & ~(size_t)3) * (P <= 256 ? 24 : 16) + 64;   // (<= 256 partitions: the carry -- every run may also START with waiting records)
        };
        if (plan->cg.hp_line_recs && lines_lds(4) > 156 * 1024) {
            // wide records (two or three per line) in many partitions: the lines' stage would force a smaller tile, i.e. shorter runs and
            // more padding than the {header, words} records have
            in.hp_lines = 0;
            plan->cg = generate_fused_source(in);
        }
        if (plan->cg.hp_line_recs) {
            while (u > 1 && lines_lds(u) > 156 * 1024) u /= 2;
        } else {
            const size_t rec = (size_t)(1 + plan->cg.nvals) * 8;
            while (u > 1 && (size_t)(in.geo.threads / 64) * 128 * u * rec > 128 * 1024) u /= 2;
        }
        in.geo.unroll = u;
        plan->cg = generate_fused_source(in);
    }
    if (plan->cg.partitioned && !plan->cg.hp && plan->cg.nparts > 128 && ctx->opts.tuning[0] == 0 && in.geo.threads == 256 &&
        8 + plan->cg.part_shift + 13 <= 32 && (size_t)8 * in.geo.sub_rows() * (1 + plan->cg.nvals) * 8 <= 128 * 1024) {
        // many partitions: a tile of 8 waves (8 Ki rows, one workgroup per CU) holds twice the records per partition, so the
        // whole-line padding of the scatter pass costs half as much (1 M keys: 45 % -> 22 % more records)
        in.geo.threads = 512;
        plan->cg = generate_fused_source(in);
    }
    if (plan->cg.hashed && ctx->opts.tuning[1] == 0 && in.geo.unroll > 4) {
        in.geo.unroll = 4;   // hashed group-by: the key words of 2 * U rows live in registers next to the inputs; it is bound by atomics, not by loads in flight
        plan->cg = generate_fused_source(in);
    }
    if (in.dense) {
        // LDS budget of the parked tile: (waves * 128 * U) rows of every output column.  64 KiB lets two workgroups share a
        // CU (the second one streams while the first waits at its barriers); shrink the sub-tile, then the workgroup.
        size_t rowbytes = 0;
        for (const OutSpec &o : plan->cg.outs)
            rowbytes += (o.type == QE_BOOLEAN ? 1 : (o.type == QE_DOUBLE || o.type == QE_INT64) ? 8 : 4) + (o.nullable ? 1 : 0);
        rowbytes = std::max<size_t>(rowbytes, 1);
        auto tile_bytes = [&]() { return (size_t)(in.geo.threads / 64) * 128 * in.geo.unroll * rowbytes; };
        const size_t limit = 64 * 1024;
        while (tile_bytes() > limit && in.geo.unroll > 1 && ctx->opts.tuning[1] == 0) in.geo.unroll /= 2;
        while (tile_bytes() > limit && in.geo.threads > 128 && ctx->opts.tuning[0] == 0) in.geo.threads /= 2;
        if (tile_bytes() > 150 * 1024)
            fail(QE_ERR_UNSUPPORTED, "projection list too wide for the dense kernel's LDS tile (" + std::to_string(rowbytes) + " bytes per output row)");
        plan->cg = generate_fused_source(in);
    }
    if (!agg_fns && in.filter && !in.dense) {
        // LDS budget of the per-wave FIFO of chunk buffers: waves * nbuf * ring * (bytes per output row).
        // Narrow rows get 3 buffers; wide rows 2 buffers and, if need be, fewer waves per workgroup so that a
        // workgroup stays within the 160 KiB of a CU (and several workgroups still fit).
        size_t rowbytes = 0;
        for (const OutSpec &o : plan->cg.outs)
            rowbytes += (o.type == QE_BOOLEAN ? 1 : (o.type == QE_DOUBLE || o.type == QE_INT64) ? 8 : 4) + (o.nullable ? 1 : 0);
        rowbytes = std::max<size_t>(rowbytes, 1);
        if (ctx->opts.tuning[7] / 100 == 0) in.geo.nbuf = 2;
        if (ctx->opts.tuning[0] == 0) {
            const size_t limit = 96 * 1024;
            while (in.geo.threads > 64 && (size_t)(in.geo.threads / 64) * in.geo.nbuf * in.geo.ring_entries * rowbytes > limit)
                in.geo.threads /= 2;
        }
        if ((size_t)(in.geo.threads / 64) * in.geo.nbuf * in.geo.ring_entries * rowbytes > 156 * 1024)
            fail(QE_ERR_UNSUPPORTED, "projection list too wide for the fused kernel's LDS buffers (" + std::to_string(rowbytes) +
                                         " bytes per output row); use QE_EXEC_PER_NODE");
        plan->cg = generate_fused_source(in);
    }
    if (ctx->opts.tuning[3] / 100 == 0) {
        // Register budget.  Per lane a sub-tile holds 2*U rows of every input column, later of every output
        // column, plus one VGPR per boolean per row: shrink the sub-tile of very wide plans first.  Then ask
        // for the highest occupancy (__launch_bounds__ waves per SIMD) that compiles WITHOUT SCRATCH: a spill
        // turns into HBM traffic (84 B/lane of scratch cost 2.7 GB of extra writes per 1 B rows when measured).
        auto dwords = [](int t) { return (t == QE_DOUBLE || t == QE_INT64) ? 2 : 1; };
        for (;;) {
            int in_dw = 0, out_dw = 0, nbool = 1, in_nulls = 0;
            for (int c : plan->cg.used_cols) {
                in_dw += dwords(in.schema[c].type);
                in_nulls += in.schema[c].nullable ? 1 : 0;
            }
            for (const OutSpec &o : plan->cg.outs) {
                out_dw += dwords(o.type);
                nbool += o.nullable ? 1 : 0;
            }
            // the validity bits of ALL nullable inputs share one register per load group (qe_vb)
            const int est = 2 * in.geo.unroll * (std::max(in_dw, out_dw) + nbool - 1) + (in_nulls ? in.geo.unroll : 0) + 54;
            // three waves per SIMD need <= 168 VGPRs: a half-size sub-tile at 3 waves beat the full one at 2 waves
            // (cfg 2 with nullable inputs: 4.47 vs 6.16 ms per 1 B rows)
            plan->est_regs = est;
            if (((est > 168 && in.geo.unroll > 4) || (est > 300 && in.geo.unroll > 2)) && ctx->opts.tuning[1] == 0 && !wide) {
                in.geo.unroll /= 2;
                if (!in.dense) in.geo.subs_per_chunk *= 2;   // keep the chunk size (the dense kernel's chunk IS the sub-tile)
                plan->cg = generate_fused_source(in);
                continue;
            }
            // (nullable cfg 2, est 122: 3 waves per SIMD 3.54 ms, 4 waves 3.69 ms -- the request also shapes the register allocation)
            in.geo.min_waves = est <= 120 ? 4 : est <= 168 ? 3 : est <= 256 ? 2 : 1;
            break;
        }
        for (;;) {
            plan->cg = generate_fused_source(in);
            plan->kernel = ctx->jit->get(plan->cg.source, "qe_fused", false);   // compile (or cache hit) only
            if (ctx->jit->last_scratch <= 0) break;
            const bool can_retry = in.geo.min_waves > 1 || (in.geo.unroll > 2 && ctx->opts.tuning[1] == 0);
            if (can_retry) ctx->jit->reject(plan->cg.source, ctx->jit->last_scratch);   // superseded below: it does not stay in the cache
            if (in.geo.min_waves > 2) {
                in.geo.min_waves--;
            } else if (in.geo.unroll > 2 && ctx->opts.tuning[1] == 0) {
                in.geo.unroll /= 2;            // a smaller sub-tile rather than one wave per SIMD
                if (!in.dense) in.geo.subs_per_chunk *= 2;
                in.geo.min_waves = 3;
            } else if (in.geo.min_waves > 1) {
                in.geo.min_waves--;
            } else {
                break;
            }
        }
    }
    plan->geo = in.geo;
    plan->explicit_geometry = ctx->opts.tuning[0] != 0 || ctx->opts.tuning[1] != 0 || ctx->opts.tuning[3] != 0 || ctx->opts.tuning[4] != 0 ||
                              ctx->opts.tuning[7] != 0;
    plan->aggregate = agg_fns != nullptr;
    plan->kernel = ctx->jit->get(plan->cg.source, "qe_fused", load);
    ctx->plans[k] = plan;
    return plan;
}

int device_cus(int device) {
    int cus = 0;
    QE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    return cus > 0 ? cus : 256;
}

int blocks_per_cu(const qe_ctx *ctx, const Plan &plan) {
    if (ctx->opts.tuning[3] % 100 > 0) return ctx->opts.tuning[3] % 100;
    int nb = 0;
    hipError_t e = hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, plan.kernel.fn, plan.geo.threads, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        nb = 2;
    }
    return std::max(1, std::min(nb, 8));
}

void fill_inputs(FusedParams &p, const qe_batch *batch, const Plan &plan) {
    std::memset(&p, 0, sizeof p);
    for (size_t s = 0; s < plan.cg.used_cols.size(); s++) {
        const Column &c = batch->cols[plan.cg.used_cols[s]];
        p.col[s] = c.data;
        p.colvalid[s] = (const unsigned long long *)c.validity;
    }
    p.nrows = batch->nrows;
    // plan constant lookup tables (string ranks / code remaps) ride in the unused tail of col[]
    if (plan.aux_dev.size() != plan.cg.aux_tables.size()) {
        for (const std::vector<int32_t> &t : plan.cg.aux_tables) {
            void *d = nullptr;
            QE_HIP(hipMalloc(&d, std::max<size_t>(t.size() * 4, 16)));
            plan.aux_dev.push_back(d);
            if (!t.empty()) QE_HIP(hipMemcpy(d, t.data(), t.size() * 4, hipMemcpyHostToDevice));
        }
    }
    for (size_t k = 0; k < plan.aux_dev.size(); k++) p.col[kMaxCols - 1 - k] = plan.aux_dev[k];
}

void launch_fused(qe_ctx *ctx, const Plan &plan, FusedParams &p, int grid, bool timed = false) {
    void *args[] = {&p};
    timed = timed || ctx->opts.profile;
    if (timed) QE_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    QE_HIP(hipModuleLaunchKernel(plan.kernel.fn, grid, 1, 1, plan.geo.threads, 1, 1, 0, ctx->stream, args, nullptr));
    if (timed) QE_HIP(hipEventRecord(ctx->ev1, ctx->stream));
}

void collect_time(qe_ctx *ctx) {
    if (!ctx->opts.profile) return;
    float ms = 0.f;
    QE_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->last_ms = ms;
    ctx->total_ms += ms;
    ctx->launches++;
}

void free_result(qe_ctx *ctx, qe_result *r) {
    if (!r) return;
    for (auto &c : r->cols) {
        if (c.hold_data || c.hold_valid) {   // per-node results share buffers with pool-backed temporaries
            c.hold_data.reset();
            c.hold_valid.reset();
            continue;
        }
        ctx->pool.release(c.data);
        ctx->pool.release(c.validity);
        ctx->pool.release(c.bytes_data);
        ctx->pool.release(c.bytes_valid);
    }
    delete r;
}

// Pass rate of every conjunct on its own (qe_conj_probe), then the evaluation order that fetches the fewest 128-byte lines.
// Model: a column of w bytes per row, needed where a share d of the rows is still alive, costs w * (1 - (1 - d)^(128 / w)) bytes
// per row (whole lines are fetched); conjuncts are taken as independent; the projections' columns are read at the final
// density whatever the order.  Up to 6 conjuncts: every permutation; more: as written.  Returns {} for "as written".
std::vector<int> choose_conjunct_order(qe_ctx *ctx, const qe_batch *batch, const Plan &plan) {
    const int K = plan.cg.nconj;
    if (K < 2 || K > 6 || (int)plan.cg.conj_cols.size() != K) return {};
    const int64_t n = batch->nrows, sub_rows = plan.geo.sub_rows();
    const int64_t full_subs = n / sub_rows;
    const int64_t S = std::min<int64_t>(256, full_subs);
    if (S < 16) return {};
    hipFunction_t f_probe = nullptr;
    QE_HIP(hipModuleGetFunction(&f_probe, plan.kernel.module, "qe_conj_probe"));
    uint32_t *d_cnt = (uint32_t *)ctx->pool.alloc(64 * 4);
    struct G { qe_ctx *c; void *q; ~G() { c->pool.release(q); } } g{ctx, d_cnt};
    FusedParams pp;
    fill_inputs(pp, batch, plan);
    pp.nchunks = S;
    pp.stagger_rows = (full_subs / S) * sub_rows;
    pp.blk = (unsigned long long *)d_cnt;
    void *args[] = {&pp};
    const int waves = plan.geo.threads / 64;
    QE_HIP(hipMemsetAsync(d_cnt, 0, 64 * 4, ctx->stream));
    QE_HIP(hipModuleLaunchKernel(f_probe, (unsigned)((S + waves - 1) / waves), 1, 1, plan.geo.threads, 1, 1, 0, ctx->stream, args, nullptr));
    uint32_t h[64];
    QE_HIP(hipMemcpyAsync(h, d_cnt, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    QE_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<double> rate((size_t)K);
    for (int i = 0; i < K; i++) rate[(size_t)i] = (double)h[i] / (double)(S * sub_rows);
    auto lines = [](double d, int w) { return w <= 0 ? 0.0 : (double)w * (1.0 - std::pow(1.0 - std::min(1.0, std::max(0.0, d)), 128.0 / (double)w)); };
    const size_t ncols = plan.cg.col_width.size();
    auto cost_of = [&](const std::vector<int> &perm) {
        std::vector<char> loaded(ncols, 0);
        double alive = 1.0, cost = 0.0;
        for (int k : perm) {
            for (int c : plan.cg.conj_cols[(size_t)k])
                if (!loaded[(size_t)c]) { loaded[(size_t)c] = 1; cost += lines(alive, plan.cg.col_width[(size_t)c]); }
            alive *= rate[(size_t)k];
        }
        return cost;   // (+ the projection-only columns at the final density: the same for every order)
    };
    std::vector<int> perm((size_t)K), best;
    for (int i = 0; i < K; i++) perm[(size_t)i] = i;
    const std::vector<int> identity = perm;
    const double written = cost_of(identity);
    double best_cost = written;
    do {
        const double c = cost_of(perm);
        if (c < best_cost - 1e-9) { best_cost = c; best = perm; }
    } while (std::next_permutation(perm.begin(), perm.end()));
    // a different order must save at least 3 % of the filter's bytes: otherwise the written order stays (one kernel fewer to build)
    if (best.empty() || best_cost > 0.97 * written) return {};
    return best;
}

qe_result *run_fused(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter, const qe_expr *const *projs,
                     int32_t nproj) {
    auto plan = get_plan(ctx, batch, filter, projs, nproj, nullptr, true);
    const int64_t n = batch->nrows;
    // Geometry choice ("measure, don't guess"): which of the two sub-tile geometries is faster depends on the plan's
    // shape (cfg 2: the wide one by 5-8 %; cfg 3: the default by 25 %), so on a large batch the first two executions of a
    // plan time one each and the faster one is kept.  Same rows, same order either way.
    const std::shared_ptr<Plan> base = plan;
    // Conjunct order (round 3): the load stages follow the evaluation order of the filter's AND chain.  As written,
    // `c < 0.5 AND a < 100` reads c in full and a for half of the rows; evaluated as `a < 100 AND c < 0.5` it reads a in full, c
    // where a passes.  On its first execution on a large batch a plan measures the pass rate of every conjunct on its own
    // (qe_conj_probe over 256 sub-tiles spread over the batch), the host picks the order that fetches the fewest 128-byte
    // lines, and the plan keeps it.  Same rows, same order: a row is kept iff every conjunct is TRUE.
    if (base->cg.has_probe && !base->conj_decided && n >= kSampleFromRows && (ctx->opts.tuning[5] & 1048576) == 0) {
        base->conj_decided = true;
        base->conj_order = choose_conjunct_order(ctx, batch, *base);
    }
    std::vector<int> order = base->conj_order;
    if (!order.empty()) plan = get_plan(ctx, batch, filter, projs, nproj, nullptr, true, nullptr, 0, false, false, &order);
    const std::shared_ptr<Plan> obase = plan;   // the plan in the chosen order, default geometry: what the memories below hang on
    qe_ctx::GeoChoice *choice = nullptr;
    int cand = 0;
    {
        const int k2 = plan->geo.unroll > 0 ? (plan->est_regs - 54) / (2 * plan->geo.unroll) : 99;   // registers per row pair
        const bool eligible = !plan->explicit_geometry && plan->est_regs > 0 && n >= (32ll << 20) &&   // (a plain projection too: 6.5 vs 7.3 ms)
                              32 * k2 + 54 <= 256 && (ctx->opts.tuning[5] & 8192) == 0;
        if (eligible) {
            const bool fresh = ctx->geo_choice.find(obase.get()) == ctx->geo_choice.end();
            choice = &ctx->geo_choice[obase.get()];
            if (fresh) {   // a decision measured earlier (another context / process) is kept: same plan => same geometry
                // ... unless the two candidates were closer than the box-to-box spread of one binary (+-7 %) when
                // it was measured: such a decision is measured again once per context, on THIS box
                double margin = 1.0;
                const int saved = ctx->jit->load_choice(obase->cg.source, &margin);
                if (saved >= 0 && margin >= 0.07) { choice->chosen = saved; choice->from_cache = true; }
            }
            // exploring: the candidates alternate, kGeoRuns timed executions each, best time wins
            cand = choice->chosen;
            if (cand < 0) {   // the candidate with the fewest runs so far
                cand = 0;
                for (int c = 1; c < qe_ctx::GeoChoice::kCands; c++)
                    if (choice->runs[c] < choice->runs[cand]) cand = c;
            }
            if (cand >= 1) {
                try {
                    plan = get_plan(ctx, batch, filter, projs, nproj, nullptr, true, nullptr, 0, cand, false, order.empty() ? nullptr : &order);
                } catch (const Error &) {   // this candidate does not build for this plan: it is out of the race
                    choice->runs[cand] = 1 << 20;
                    if (choice->chosen == cand) choice->chosen = 0;
                    cand = 0;
                    plan = obase;
                }
            }
        }
    }
    const bool exploring = choice && choice->chosen < 0;
    std::unique_ptr<qe_result, std::function<void(qe_result *)>> res(new qe_result(),
                                                                      [ctx](qe_result *r) { free_result(ctx, r); });
    int64_t cap = ctx->opts.result_capacity_rows > 0 ? std::min<int64_t>(ctx->opts.result_capacity_rows, n) : n;
    res->capacity = cap;
    for (const OutSpec &os : plan->cg.outs) {
        OutColumn oc;
        oc.type = os.type;
        oc.nullable = os.nullable;
        oc.dict = os.dict;
        oc.dict_handle.d = os.dict;
        res->cols.push_back(oc);
    }
    if (n == 0) return res.release();
    FusedParams p;
    fill_inputs(p, batch, *plan);
    for (size_t i = 0; i < res->cols.size(); i++) {
        OutColumn &oc = res->cols[i];
        if (oc.type == QE_BOOLEAN) {
            oc.bytes_data = ctx->pool.alloc((size_t)std::max<int64_t>(cap, 1));
            p.out[i] = oc.bytes_data;
        } else {
            oc.data = ctx->pool.alloc(std::max<size_t>(type_width(oc.type) * (size_t)cap, 16));
            p.out[i] = oc.data;
        }
        if (oc.nullable) {
            oc.bytes_valid = (uint8_t *)ctx->pool.alloc((size_t)std::max<int64_t>(cap, 1));
            p.outvalid[i] = oc.bytes_valid;
        }
    }
    // Two-pass form for HIGH selectivity (chosen from the selectivity this plan showed last time): count the kept rows
    // per chunk, scan, then stream again and store every kept row straight at its final position -- no look-back, no
    // LDS ring, no staging round trip (which costs 48 B per kept row instead of 16 once the ring overflows).
    const bool force_two_pass = (ctx->opts.tuning[5] & 512) != 0, never_two_pass = (ctx->opts.tuning[5] & 1024) != 0;
    // DENSE single-pass form (round 2): chunk == sub-tile, outputs wait in the registers during a blocking look-back and
    // go straight to their final position -- one read of every input, one write of every kept row at ANY selectivity.
    // Chosen from the selectivity the plan showed last time; supersedes the two-pass form (kept behind debug bit 512).
    const bool force_dense = (ctx->opts.tuning[5] & 16384) != 0, never_dense = (ctx->opts.tuning[5] & 32768) != 0;
    const double dense_from = std::getenv("QE_DENSE_FROM") ? std::atof(std::getenv("QE_DENSE_FROM")) : kDenseFromSelectivity;
    // The FIRST execution of a plan knows nothing about its selectivity (a ring kernel that keeps every row costs 12 ms per
    // 1 B rows where the dense form needs 6.3): estimate it from 256 chunks spread evenly over the batch -- the count pass
    // of the two-pass form over ~4 M rows, filter columns only, a few tens of microseconds.
    if (plan->cg.has_filter && plan->cg.two_pass && base->last_selectivity < 0 && !force_dense && !never_dense && !force_two_pass &&
        n >= kSampleFromRows && (ctx->opts.tuning[5] & 65536) == 0) {
        const int64_t chunk_rows = plan->geo.chunk_rows();
        const int64_t full_chunks = n / chunk_rows;
        const int64_t S = std::min<int64_t>(256, full_chunks);
        if (S > 0) {
            const int waves = plan->geo.threads / 64;
            hipFunction_t f_count = nullptr;
            QE_HIP(hipModuleGetFunction(&f_count, plan->kernel.module, "qe_fp_count"));
            uint32_t *d_sample = (uint32_t *)ctx->pool.alloc((size_t)S * 4);
            struct SG { qe_ctx *c; void *q; ~SG() { c->pool.release(q); } } sg{ctx, d_sample};
            FusedParams ps = p;
            ps.nchunks = S;
            ps.stagger_chunks = full_chunks / S;
            ps.blk = (unsigned long long *)d_sample;
            void *sargs[] = {&ps};
            QE_HIP(hipModuleLaunchKernel(f_count, (unsigned)((S + waves - 1) / waves), 1, 1, plan->geo.threads, 1, 1, 0, ctx->stream, sargs, nullptr));
            std::vector<uint32_t> h((size_t)S);
            QE_HIP(hipMemcpyAsync(h.data(), d_sample, (size_t)S * 4, hipMemcpyDeviceToHost, ctx->stream));
            QE_HIP(hipStreamSynchronize(ctx->stream));
            unsigned long long kept = 0;
            for (uint32_t v : h) kept += v;
            base->last_selectivity = (double)kept / (double)(S * chunk_rows);
        }
    }
    const bool dense = plan->cg.has_filter && !never_dense && !force_two_pass && (force_dense || base->last_selectivity >= dense_from);
    const bool two_pass = !dense && plan->cg.has_filter && plan->cg.two_pass && !never_two_pass && n < (1ll << 32) &&
                          (force_two_pass || base->last_selectivity >= 0.6);   // measured crossover on cfg 2: 0.55 - 0.6
    unsigned long long total = 0;
    // LOCAL form (round 3): for plans that keep a few per cent of their rows at most.  The scan has no inter-wave dependency --
    // no ticket, no descriptor, no look-back: every wave parks the kept rows of its (statically strided) chunks in per-chunk
    // slots of ring_entries rows; a scan over the per-chunk counts and ONE move kernel put them at their final place.  The
    // chunk is sized from the selectivity the plan showed so that it keeps ~ring_entries / 4 rows; a chunk that overflows its
    // slot is reported by the kernel, this execution falls back to the single-pass kernel and the plan stays there.
    const bool force_local = (ctx->opts.tuning[5] & 262144) != 0, never_local = (ctx->opts.tuning[5] & 524288) != 0;
    const double local_upto = std::getenv("QE_LOCAL_UPTO") ? std::atof(std::getenv("QE_LOCAL_UPTO")) : kLocalUpToSelectivity;
    bool local = plan->cg.has_filter && plan->cg.two_pass && !dense && !two_pass && !never_local && !base->local_overflowed &&
                 (force_local || (n >= kSampleFromRows && base->last_selectivity >= 0 && base->last_selectivity <= local_upto));
    if (local) {
        const int64_t sub_rows = plan->geo.sub_rows(), ring = plan->cg.fl_ring;
        const double sel = base->last_selectivity;
        // expected kept rows per chunk = slot / fill: 2.5 leaves 1.5 slots of head room over the mean (a binomial count of ~200
        // has a standard deviation of ~14); larger chunks were measured faster (cfg 3: 2 Ki rows 2.07 ms, 8 Ki rows 1.71 ms)
        static const double fill = std::getenv("QE_LOCAL_FILL") ? std::atof(std::getenv("QE_LOCAL_FILL")) : 2.5;
        int64_t subs = sel > 0 ? (int64_t)((double)ring / (fill * sel) / (double)sub_rows) : 32;
        if (sel < 0) subs = 1;                               // nothing known (forced): the smallest chunk
        subs = std::max<int64_t>(1, std::min<int64_t>(subs, 32));
        if (ctx->opts.tuning[4] % 10000 > 0) subs = ctx->opts.tuning[4] % 10000;   // explicit sub-tiles per chunk (measurement)
        const int64_t crow = subs * sub_rows;
        const int64_t nchunks = (n + crow - 1) / crow;
        if (nchunks * ring >= (1ll << 32)) local = false;    // 32-bit offsets of the move
        if (local) {
            const int waves = plan->geo.threads / 64;
            hipFunction_t f_scan = nullptr, f_move = nullptr;
            QE_HIP(hipModuleGetFunction(&f_scan, plan->kernel.module, "qe_fl_scan"));
            QE_HIP(hipModuleGetFunction(&f_move, plan->kernel.module, "qe_fl_move"));
            std::vector<void *> scratch;
            struct LocalGuard {
                qe_ctx *c; std::vector<void *> &v;
                ~LocalGuard() { for (void *q : v) c->pool.release(q); }
            } lg{ctx, scratch};
            const int64_t nsum = (nchunks + 1023) / 1024;
            uint32_t *d_counts = (uint32_t *)ctx->pool.alloc((size_t)nchunks * 4);
            scratch.push_back(d_counts);
            uint32_t *d_offsets = (uint32_t *)ctx->pool.alloc((size_t)nchunks * 4);
            scratch.push_back(d_offsets);
            uint32_t *d_sums = (uint32_t *)ctx->pool.alloc((size_t)(nsum + 1) * 4);
            scratch.push_back(d_sums);
            FusedParams lp = p;
            for (size_t i = 0; i < res->cols.size(); i++) {
                const OutColumn &oc = res->cols[i];
                const size_t w = oc.type == QE_BOOLEAN ? 1 : type_width(oc.type);
                lp.stage[i] = ctx->pool.alloc((size_t)nchunks * (size_t)ring * w);
                scratch.push_back(lp.stage[i]);
                if (oc.nullable) {
                    lp.stagevalid[i] = (unsigned char *)ctx->pool.alloc((size_t)nchunks * (size_t)ring);
                    scratch.push_back(lp.stagevalid[i]);
                }
            }
            lp.capacity = cap;
            lp.nchunks = nchunks;
            lp.stagger_rows = crow;
            lp.blk = (unsigned long long *)d_counts;
            lp.l1 = (unsigned long long *)d_offsets;
            lp.error = ctx->d_ctrl + 1;
            lp.total = (unsigned long long *)(ctx->d_ctrl + 2);
            // 20 waves per CU: measured on cfg 3 (600 M rows) 12 / 16 / 20 / 24 waves per CU = 1.99 / 1.85 / 1.78 / 1.85 ms, on cfg 4
            // 0.665 / 0.654 / 0.661 / 0.670 ms -- past 20 the extra streams cost more than the extra loads in flight bring
            int occ = 0;   // of qe_fl_scan itself (it has its own register allocation: no occupancy request, so that it never spills)
            if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&occ, f_scan, plan->geo.threads, 0) != hipSuccess) {
                (void)hipGetLastError();
                occ = 2;
            }
            occ = std::max(1, std::min(occ, 8));
            const int bpc = ctx->opts.tuning[3] % 100 > 0 ? ctx->opts.tuning[3] % 100 : std::min(occ, std::max(1, 20 / waves));
            const int64_t max_grid = (int64_t)device_cus(ctx->device) * bpc;
            const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((nchunks + waves - 1) / waves, max_grid));
            const int mgrid = (int)std::max<int64_t>(1, std::min<int64_t>((nchunks + 3) / 4, (int64_t)device_cus(ctx->device) * 8));
            void *largs[] = {&lp};
            QE_HIP(hipMemsetAsync(ctx->d_ctrl, 0, 96, ctx->stream));
            if (ctx->opts.profile) QE_HIP(hipEventRecord(ctx->ev0, ctx->stream));
            QE_HIP(hipModuleLaunchKernel(f_scan, grid, 1, 1, plan->geo.threads, 1, 1, 0, ctx->stream, largs, nullptr));
            pn::exclusive_scan_u32(ctx->stream, d_counts, d_offsets, d_sums, nchunks, lp.total);
            QE_HIP(hipModuleLaunchKernel(f_move, mgrid, 1, 1, 256, 1, 1, 0, ctx->stream, largs, nullptr));
            if (ctx->opts.profile) QE_HIP(hipEventRecord(ctx->ev1, ctx->stream));
            QE_HIP(hipMemcpyAsync(ctx->h_ctrl, ctx->d_ctrl, 96, hipMemcpyDeviceToHost, ctx->stream));
            QE_HIP(hipStreamSynchronize(ctx->stream));
            collect_time(ctx);
            const unsigned int *hc = (const unsigned int *)ctx->h_ctrl;
            if (hc[1] == 5) {            // some chunk kept more rows than its slot holds: this plan's rows are not spread evenly
                local = false;
                base->local_overflowed = true;
            } else if (hc[1] != 0) {
                fail(QE_ERR_INTERNAL, "local form: unexpected error flag");
            } else {
                total = ctx->h_ctrl[1];
            }
        }
    }
    if (local) {
        // done above
    } else if (dense) {
        auto dplan = get_plan(ctx, batch, filter, projs, nproj, nullptr, true, nullptr, 0, false, true);
        // a tile = the sub-tiles of one workgroup's waves; descriptors are per TILE
        const int waves = dplan->geo.threads / 64;
        const int64_t tile_rows = (int64_t)dplan->geo.sub_rows() * waves;
        const int64_t nchunks = (n + tile_rows - 1) / tile_rows;
        if (nchunks >= (1ll << 31)) fail(QE_ERR_UNSUPPORTED, "batch too large for 32-bit tile tickets");
        const int64_t max_grid = (int64_t)device_cus(ctx->device) * blocks_per_cu(ctx, *dplan);
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(nchunks, max_grid));
        const int64_t nblocks = (nchunks + 63) / 64;
        const size_t desc_words = (size_t)nchunks + 2 * (size_t)nblocks;
        unsigned long long *desc = (unsigned long long *)ctx->pool.alloc(desc_words * 8);
        struct DG { qe_ctx *c; void *q; ~DG() { c->pool.release(q); } } dg{ctx, desc};
        FusedParams dp = p;                       // same inputs and outputs
        fill_inputs(dp, batch, *dplan);           // (the dense plan has its own column slots / aux tables)
        for (size_t i = 0; i < res->cols.size(); i++) { dp.out[i] = p.out[i]; dp.outvalid[i] = p.outvalid[i]; }
        dp.capacity = cap;
        dp.desc = desc;
        dp.l1 = desc + nchunks;
        dp.blk = desc + nchunks + nblocks;
        dp.ticket = ctx->d_ctrl;
        dp.error = ctx->d_ctrl + 1;
        dp.total = (unsigned long long *)(ctx->d_ctrl + 2);
        dp.nchunks = nchunks;
        void *trace = nullptr;
        struct TG { qe_ctx *c; void **q; ~TG() { c->pool.release(*q); } } tg{ctx, &trace};
        if (ctx->opts.tuning[5] & 32) {   // diagnostic build: per tile {t_ticket, t_published, t_resolved, t_stored} in 10 ns ticks
            trace = ctx->pool.alloc((size_t)nchunks * 32);
            dp.trace = (unsigned long long *)trace;
            QE_HIP(hipMemsetAsync(trace, 0, (size_t)nchunks * 32, ctx->stream));
        }
        QE_HIP(hipMemsetAsync(ctx->d_ctrl, 0, 96, ctx->stream));
        QE_HIP(hipMemsetAsync(desc, 0, desc_words * 8, ctx->stream));
        launch_fused(ctx, *dplan, dp, grid);
        QE_HIP(hipMemcpyAsync(ctx->h_ctrl, ctx->d_ctrl, 96, hipMemcpyDeviceToHost, ctx->stream));
        QE_HIP(hipStreamSynchronize(ctx->stream));
        collect_time(ctx);
        if (trace && std::getenv("QE_TRACE_FILE")) {
            std::vector<unsigned long long> tr((size_t)nchunks * 4);
            QE_HIP(hipMemcpy(tr.data(), trace, tr.size() * 8, hipMemcpyDeviceToHost));
            if (FILE *f = std::fopen(std::getenv("QE_TRACE_FILE"), "wb")) {
                std::fwrite(tr.data(), 8, tr.size(), f);
                std::fclose(f);
            }
        }
        const unsigned int *hc = (const unsigned int *)ctx->h_ctrl;
        if (hc[1] != 0)
            fail(QE_ERR_INTERNAL, hc[1] == 2 ? "dense kernel: ticket grant never posted (spin limit)"
                                             : "dense kernel: look-back spin limit reached (chunk descriptor never published)");
        total = ctx->h_ctrl[1];
    } else if (two_pass) {
        const int64_t chunk_rows = plan->geo.chunk_rows();
        const int waves = plan->geo.threads / 64;
        const int64_t nchunks = (n + chunk_rows - 1) / chunk_rows;
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((nchunks + waves - 1) / waves, (int64_t)device_cus(ctx->device) * 8));
        hipFunction_t f_count = nullptr, f_write = nullptr;
        QE_HIP(hipModuleGetFunction(&f_count, plan->kernel.module, "qe_fp_count"));
        QE_HIP(hipModuleGetFunction(&f_write, plan->kernel.module, "qe_fp_write"));
        uint32_t *d_counts = (uint32_t *)ctx->pool.alloc((size_t)nchunks * 4);
        struct CG { qe_ctx *c; void *q; ~CG() { c->pool.release(q); } } cg_guard{ctx, d_counts};
        p.capacity = cap;
        p.nchunks = nchunks;
        p.blk = (unsigned long long *)d_counts;
        p.error = ctx->d_ctrl + 1;
        p.total = (unsigned long long *)(ctx->d_ctrl + 2);
        void *args[] = {&p};
        QE_HIP(hipMemsetAsync(ctx->d_ctrl, 0, 96, ctx->stream));
        if (ctx->opts.profile) QE_HIP(hipEventRecord(ctx->ev0, ctx->stream));
        QE_HIP(hipModuleLaunchKernel(f_count, grid, 1, 1, plan->geo.threads, 1, 1, 0, ctx->stream, args, nullptr));
        launch_gb_scan(ctx->stream, d_counts, nchunks, 1, p.total);   // counts -> exclusive offsets, *total = kept rows
        QE_HIP(hipModuleLaunchKernel(f_write, grid, 1, 1, plan->geo.threads, 1, 1, 0, ctx->stream, args, nullptr));
        if (ctx->opts.profile) QE_HIP(hipEventRecord(ctx->ev1, ctx->stream));
        QE_HIP(hipMemcpyAsync(ctx->h_ctrl, ctx->d_ctrl, 96, hipMemcpyDeviceToHost, ctx->stream));
        QE_HIP(hipStreamSynchronize(ctx->stream));
        collect_time(ctx);
        total = ctx->h_ctrl[1];
    } else {
        const int64_t chunk_rows = plan->geo.chunk_rows();
        const int waves = plan->geo.threads / 64;
        const int64_t max_grid = (int64_t)device_cus(ctx->device) * blocks_per_cu(ctx, *plan);
        // the first `stagger_chunks` chunks (one per resident wave) have graded sizes ((7c mod 16)+1)/16
        int64_t stagger_chunks = plan->geo.stagger ? max_grid * waves : 0, stagger_rows = 0, nchunks = 0;
        {
            const int64_t sixteenth = chunk_rows / 16;
            int64_t c = 0, rows = 0;
            // whole periods of 16 chunks cover 136 sixteenths
            const int64_t periods = std::min<int64_t>(stagger_chunks / 16, n / (136 * sixteenth));
            c = periods * 16;
            rows = periods * 136 * sixteenth;
            while (c < stagger_chunks && rows < n) {
                rows += (((7 * c) & 15) + 1) * sixteenth;
                c++;
            }
            if (rows >= n) {
                nchunks = c;
                stagger_chunks = c;        // every chunk is a staggered one
                stagger_rows = rows;
            } else {
                stagger_rows = rows;
                nchunks = stagger_chunks + (n - rows + chunk_rows - 1) / chunk_rows;
            }
        }
        if (nchunks >= (1ll << 31)) fail(QE_ERR_UNSUPPORTED, "batch too large for 32-bit chunk tickets");
        const int grid = (int)std::min<int64_t>((nchunks + waves - 1) / waves, max_grid);
        static const bool dbg_grid = std::getenv("QE_DEBUG_GRID") != nullptr;
        if (dbg_grid) std::fprintf(stderr, "[qe] ring kernel: grid %d x %d threads (%d blocks per CU), unroll %d, %d sub-tiles per chunk, ring %d, min_waves %d\n",
                                   grid, plan->geo.threads, blocks_per_cu(ctx, *plan), plan->geo.unroll, plan->geo.subs_per_chunk, plan->geo.ring_entries, plan->geo.min_waves);
        // scratch: look-back descriptors + one staging slot (chunk_rows rows per output column) per resident wave
        std::vector<void *> scratch;
        struct ScratchGuard {
            qe_ctx *c; std::vector<void *> &v;
            ~ScratchGuard() { for (void *q : v) c->pool.release(q); }
        } sg{ctx, scratch};
        // descriptors: [nchunks] level 0, then [nblocks] level 1, then [nblocks] block counters -- one allocation,
        // zeroed by ONE memset on the stream before every launch
        const int64_t nblocks = (nchunks + 63) / 64;
        const size_t desc_words = (size_t)nchunks + 2 * (size_t)nblocks;
        unsigned long long *desc = (unsigned long long *)ctx->pool.alloc(desc_words * 8);
        scratch.push_back(desc);
        if (plan->cg.has_filter) {
            const size_t slots = (size_t)grid * waves * plan->geo.nbuf;   // one staging (overflow) slot per LDS buffer
            for (size_t i = 0; i < res->cols.size(); i++) {
                const OutColumn &oc = res->cols[i];
                const size_t w = oc.type == QE_BOOLEAN ? 1 : type_width(oc.type);
                p.stage[i] = ctx->pool.alloc(slots * (size_t)plan->geo.slot_rows() * w);
                scratch.push_back(p.stage[i]);
                if (oc.nullable) {
                    p.stagevalid[i] = (unsigned char *)ctx->pool.alloc(slots * (size_t)plan->geo.slot_rows());
                    scratch.push_back(p.stagevalid[i]);
                }
            }
        }
        p.capacity = cap;
        p.desc = desc;
        p.l1 = desc + nchunks;
        p.blk = desc + nchunks + nblocks;
        p.ticket = ctx->d_ctrl;
        p.error = ctx->d_ctrl + 1;
        p.total = (unsigned long long *)(ctx->d_ctrl + 2);
        p.nchunks = nchunks;
        p.stagger_chunks = stagger_chunks;
        p.stagger_rows = stagger_rows;
        p.stats = (unsigned long long *)(ctx->d_ctrl + 16);   // bytes 64..95 of the control block
        if (ctx->opts.tuning[5] & 32) {
            p.trace = (unsigned long long *)ctx->pool.alloc((size_t)nchunks * 32);
            scratch.push_back(p.trace);
            QE_HIP(hipMemsetAsync(p.trace, 0, (size_t)nchunks * 32, ctx->stream));
        }
        // flags, tickets and descriptors are re-zeroed on the stream before EVERY launch
        QE_HIP(hipMemsetAsync(ctx->d_ctrl, 0, 96, ctx->stream));
        QE_HIP(hipMemsetAsync(desc, 0, desc_words * 8, ctx->stream));
        launch_fused(ctx, *plan, p, grid, exploring);
        QE_HIP(hipMemcpyAsync(ctx->h_ctrl, ctx->d_ctrl, 96, hipMemcpyDeviceToHost, ctx->stream));
        QE_HIP(hipStreamSynchronize(ctx->stream));
        collect_time(ctx);
        if (exploring) {   // kGeoRuns timed executions per candidate (best of), then the faster geometry is kept for this plan
            constexpr int kGeoRuns = 3;   // one sample each was not reproducible: box / allocation noise is +-7 %, the gap 5-9 %
            float ms = 0.f;
            QE_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
            choice->best_ms[cand] = std::min(choice->best_ms[cand], ms);
            choice->runs[cand]++;
            bool all_done = true;
            for (int c = 0; c < qe_ctx::GeoChoice::kCands; c++) all_done = all_done && choice->runs[c] >= kGeoRuns;
            if (all_done) {
                // another geometry must beat the default by a margin (2 %): ties go to the default, which holds fewer registers
                int best = 0;
                for (int c = 1; c < qe_ctx::GeoChoice::kCands; c++)
                    if (choice->best_ms[c] < 0.98f * choice->best_ms[0] && (best == 0 || choice->best_ms[c] < choice->best_ms[best])) best = c;
                choice->chosen = best;
                char note[200];
                std::snprintf(note, sizeof note, "default %.4f ms, wide %.4f ms, mid %.4f ms (best of %d each, %lld rows)", choice->best_ms[0],
                              choice->best_ms[1], choice->best_ms[2], kGeoRuns, (long long)n);
                ctx->jit->store_choice(obase->cg.source, choice->chosen, note);
            }
        }
        if ((ctx->opts.tuning[5] & 32) && std::getenv("QE_TRACE_FILE")) {
            std::vector<unsigned long long> tr((size_t)nchunks * 4);
            QE_HIP(hipMemcpy(tr.data(), p.trace, tr.size() * 8, hipMemcpyDeviceToHost));
            if (FILE *f = std::fopen(std::getenv("QE_TRACE_FILE"), "wb")) {
                std::fwrite(tr.data(), 8, tr.size(), f);
                std::fclose(f);
            }
        }
        if (ctx->opts.tuning[5] & 16)
            std::fprintf(stderr, "[qe stats] chunks %llu failed_tries %llu forced_waits %llu blocking_spins %llu\n", ctx->h_ctrl[11],
                         ctx->h_ctrl[8], ctx->h_ctrl[10], ctx->h_ctrl[9]);
        const unsigned int *hc = (const unsigned int *)ctx->h_ctrl;
        if (hc[1] != 0) fail(QE_ERR_INTERNAL, "fused kernel: look-back spin limit reached (tile descriptor never published)");
        total = ctx->h_ctrl[1];
    }
    ctx->last_form = !plan->cg.has_filter ? QE_FORM_NO_FILTER : local ? QE_FORM_LOCAL : dense ? QE_FORM_DENSE : two_pass ? QE_FORM_TWO_PASS : QE_FORM_RING;
    plan->last_selectivity = base->last_selectivity = (double)total / (double)n;
    if ((int64_t)total > cap)
        fail(QE_ERR_INVALID_ARG, "result has " + std::to_string(total) + " rows but result_capacity_rows is " +
                                     std::to_string(cap));
    res->count = (int64_t)total;
    // nullable / boolean outputs were written one byte per row: pack them into bitmaps
    bool packed = false;
    for (OutColumn &oc : res->cols) {
        if (oc.type == QE_BOOLEAN) {
            oc.data = ctx->pool.alloc(std::max<size_t>(bitmap_bytes(res->count), 16));
            launch_pack_bytes(ctx->stream, (const uint8_t *)oc.bytes_data, res->count, (uint64_t *)oc.data);
            packed = true;
        }
        if (oc.nullable) {
            oc.validity = (uint64_t *)ctx->pool.alloc(std::max<size_t>(bitmap_bytes(res->count), 16));
            launch_pack_bytes(ctx->stream, oc.bytes_valid, res->count, oc.validity);
            packed = true;
        }
    }
    if (packed) {
        QE_HIP(hipGetLastError());
        QE_HIP(hipStreamSynchronize(ctx->stream));
        for (OutColumn &oc : res->cols) {
            ctx->pool.release(oc.bytes_data);
            ctx->pool.release(oc.bytes_valid);
            oc.bytes_data = nullptr;
            oc.bytes_valid = nullptr;
        }
    }
    return res.release();
}

qe_result *run_groupby_dense(qe_ctx *ctx, const qe_batch *batch, const std::shared_ptr<Plan> &plan, const int32_t *agg_fns, int32_t nagg);

// GroupByAggregation over arbitrary key tuples (a DOUBLE / INT64 / INT32 key, or more key combinations than a dense table
// holds): the hashed form.  Global open-addressing table, grown (x8) and the kernel run again when it got more than half
// full; the used entries are collected on the device, sorted by smallest row id on the host (LinkedHashMap insertion order,
// GroupByAggregationOperator.kt:22) and finished like the dense form's (Accumulators.kt:26-107).
// Key columns of a hashed group-by result: row j's {null bits, key words..} come from `key_of(j)`.
template <typename KeyOf>
void append_key_columns(qe_ctx *ctx, const CodegenOutput &cg, qe_result *res, int64_t m, KeyOf key_of,
                        std::vector<std::vector<unsigned long long>> &keep64, std::vector<std::vector<int32_t>> &keep32) {
    const int NK = (int)cg.keys.size();
    const size_t words = (size_t)std::max<int64_t>(1, (m + 63) / 64);
    auto upload = [&](const void *src, size_t bytes) -> void * {
        void *d = ctx->pool.alloc(std::max<size_t>(bytes, 16));
        if (bytes) QE_HIP(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return d;
    };
    for (int k = 0; k < NK; k++) {
        OutColumn oc;
        oc.type = cg.keys[k].type;
        oc.dict = cg.keys[k].dict;
        oc.dict_handle.d = oc.dict;
        std::vector<unsigned long long> valid(words, 0), vals64((size_t)std::max<int64_t>(m, 1), 0), bits(words, 0);
        std::vector<int32_t> vals32((size_t)std::max<int64_t>(m, 1), 0);
        bool any_null = false;
        for (int64_t j = 0; j < m; j++) {
            const unsigned long long *e = key_of(j);   // {null bits, key words..}
            if ((e[0] >> k) & 1ull) { any_null = true; continue; }
            valid[j >> 6] |= 1ull << (j & 63);
            const unsigned long long kw = e[1 + k];
            vals64[j] = kw;                       // DOUBLE: the canonical bits ARE the value; INT64: the value
            vals32[j] = (int32_t)(int64_t)kw;     // INT32 / dictionary codes
            if (kw) bits[j >> 6] |= 1ull << (j & 63);
        }
        oc.nullable = any_null;
        if (oc.type == QE_BOOLEAN) {
            keep64.push_back(bits);
            oc.data = upload(keep64.back().data(), words * 8);
        } else if (oc.type == QE_DOUBLE || oc.type == QE_INT64) {
            keep64.push_back(vals64);
            oc.data = upload(keep64.back().data(), (size_t)m * 8);
        } else {
            keep32.push_back(vals32);
            oc.data = upload(keep32.back().data(), (size_t)m * 4);
        }
        if (any_null) {
            keep64.push_back(valid);
            oc.validity = (uint64_t *)upload(keep64.back().data(), words * 8);
        }
        res->cols.push_back(oc);
    }
}

// Hashed group-by whose keys do not fit the LDS table: (1) qe_ht_build gives every key a dense id (global open-addressing
// table, a 64 KiB id cache per workgroup; after its first rows a key is only READ) and writes the id of every kept row;
// (2) the dense group-by -- LDS-privatised table or the partitioned passes -- runs on the id column; (3) the ids of the result
// rows are turned back into key values.  nullptr: more keys than a dense table takes (the caller keeps the global-atomic form).
qe_result *run_groupby_ids(qe_ctx *ctx, const qe_batch *batch, const Plan &plan, const qe_expr *filter, const qe_expr *const *exprs,
                           const int32_t *agg_fns, int32_t nagg, bool *many_keys) {
    const CodegenOutput &cg = plan.cg;
    const int NK = (int)cg.keys.size(), BW = NK == 1 ? 2 : 3 + NK;   // single-key plans: 16-byte entries {key, state | null bits | id}
    const int64_t n = batch->nrows;
    std::vector<void *> temps;
    struct GT { qe_ctx *c; std::vector<void *> *t; ~GT() { for (void *q : *t) c->pool.release(q); } } gt{ctx, &temps};
    auto talloc = [&](size_t bytes) { void *q = ctx->pool.alloc(std::max<size_t>(bytes, 16)); temps.push_back(q); return q; };
    uint32_t *d_ids = (uint32_t *)talloc((size_t)n * 4);
    hipFunction_t f_build = nullptr;
    QE_HIP(hipModuleGetFunction(&f_build, plan.kernel.module, "qe_ht_build"));
    int64_t C = plan.id_capacity > 0 ? plan.id_capacity : (1ll << 16);
    int64_t D = 0;
    unsigned long long *d_keys = nullptr;
    double build_ms = 0.0;
    for (;;) {
        unsigned long long *d_tab = (unsigned long long *)talloc((size_t)C * BW * 8);
        d_keys = (unsigned long long *)talloc((size_t)C * (1 + NK) * 8);
        QE_HIP(hipMemsetAsync(d_tab, 0, (size_t)C * BW * 8, ctx->stream));
        QE_HIP(hipMemsetAsync(ctx->d_ctrl, 0, 96, ctx->stream));
        const int64_t sub_rows = plan.geo.sub_rows();
        const int64_t ntiles = (n + sub_rows - 1) / sub_rows;
        const int waves = plan.geo.threads / 64;
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((ntiles + waves - 1) / waves, (int64_t)device_cus(ctx->device) * 4));
        FusedParams p;
        fill_inputs(p, batch, plan);
        p.agg_partial = (double *)d_tab;
        p.capacity = C;
        p.ticket = ctx->d_ctrl;
        p.error = ctx->d_ctrl + 1;
        p.desc = d_keys;
        p.blk = (unsigned long long *)d_ids;
        void *args[] = {&p};
        if (ctx->opts.profile) QE_HIP(hipEventRecord(ctx->ev0, ctx->stream));
        QE_HIP(hipModuleLaunchKernel(f_build, grid, 1, 1, plan.geo.threads, 1, 1, 0, ctx->stream, args, nullptr));
        if (ctx->opts.profile) QE_HIP(hipEventRecord(ctx->ev1, ctx->stream));
        QE_HIP(hipMemcpyAsync(ctx->h_ctrl, ctx->d_ctrl, 32, hipMemcpyDeviceToHost, ctx->stream));
        QE_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->opts.profile) {
            float ms = 0.f;
            QE_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
            build_ms += ms;
        }
        const unsigned int *hc = (const unsigned int *)ctx->h_ctrl;
        if (hc[1] != 0) {   // more than half full: a bigger table, again (the launch stopped at once: an attempt costs little)
            if (many_keys && plan.id_capacity <= 0) {   // the first table of a plan that knows nothing yet: the caller takes over
                *many_keys = true;
                return nullptr;
            }
            if (C >= (1ll << 22)) return nullptr;
            C *= 4;
            continue;
        }
        D = hc[0];
        break;
    }
    plan.id_capacity = C;
    if (D > (1ll << 20) - 1) return nullptr;   // a dense table takes 2^20 groups (the id domain + its NULL code)
    // (2) the dense group-by over [the batch's columns.., ids]: the id column is a dictionary-coded key whose dictionary is a
    // placeholder of 2^k entries (only its size matters: the domain of the group id)
    int64_t dom = 2;
    while (dom < D) dom *= 2;
    dom = std::min<int64_t>(dom, (1ll << 20) - 1);
    std::shared_ptr<DictData> &idd = ctx->id_dicts[dom];
    if (!idd) {
        idd = std::make_shared<DictData>();
        idd->entries.resize((size_t)dom);
    }
    qe_batch tmp;
    tmp.nrows = n;
    tmp.cols = batch->cols;
    for (Column &c : tmp.cols) c.owned = false;
    Column idc;
    idc.type = QE_STRING;
    idc.data = d_ids;
    idc.validity = nullptr;
    idc.dict = idd;
    idc.owned = false;
    tmp.cols.push_back(idc);
    qe_expr kx;
    {
        Node nd;
        nd.kind = N_COLUMN;
        nd.type = QE_STRING;
        nd.col = (int)batch->cols.size();
        kx.e.nodes.push_back(nd);
        kx.e.root = 0;
        kx.e.max_stack = 1;
        const char tag[] = "\xEEid-column";
        kx.e.program.assign(tag, tag + sizeof tag - 1);
        kx.e.program.push_back((uint8_t)batch->cols.size());
    }
    const qe_expr *kp[1] = {&kx};
    auto dplan = get_plan(ctx, &tmp, filter, exprs, nagg, agg_fns, true, kp, 1);
    if (dplan->cg.hashed) fail(QE_ERR_INTERNAL, "dense-id plan came out hashed");
    std::unique_ptr<qe_result, std::function<void(qe_result *)>> r(run_groupby_dense(ctx, &tmp, dplan, agg_fns, nagg),
                                                                   [ctx](qe_result *q) { free_result(ctx, q); });
    if (ctx->opts.profile) {   // one step = build pass + dense passes
        ctx->last_ms += build_ms;
        ctx->total_ms += build_ms;
    }
    // (3) ids of the result rows (column 0, in insertion order) -> key values
    const int64_t m = r->count;
    std::vector<int32_t> rid((size_t)std::max<int64_t>(m, 1));
    std::vector<unsigned long long> hkeys((size_t)std::max<int64_t>(D, 1) * (1 + NK));
    if (m > 0) QE_HIP(hipMemcpyAsync(rid.data(), r->cols[0].data, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (D > 0) QE_HIP(hipMemcpyAsync(hkeys.data(), d_keys, (size_t)D * (1 + NK) * 8, hipMemcpyDeviceToHost, ctx->stream));
    QE_HIP(hipStreamSynchronize(ctx->stream));
    std::unique_ptr<qe_result, std::function<void(qe_result *)>> res(new qe_result(), [ctx](qe_result *q) { free_result(ctx, q); });
    res->count = m;
    res->capacity = m;
    std::vector<std::vector<unsigned long long>> keep64;
    std::vector<std::vector<int32_t>> keep32;
    append_key_columns(ctx, cg, res.get(), m, [&](int64_t j) { return &hkeys[(size_t)rid[(size_t)j] * (1 + NK)]; }, keep64, keep32);
    for (size_t c = 1; c < r->cols.size(); c++) {   // the aggregates move over as they are
        res->cols.push_back(r->cols[c]);
        r->cols[c].data = nullptr;
        r->cols[c].validity = nullptr;
    }
    QE_HIP(hipStreamSynchronize(ctx->stream));
    return res.release();
}

// The same on the device, for results of many groups (1 M groups cost the host ~70 ms and crossed the link twice): the entries are
// ordered by first row (their keys = the first rows, a stable LSD radix sort over as many 4-bit digits as the batch's row ids
// have), then one kernel writes every result column -- key values, finished accumulators, validity bitmaps -- in that order.  The
// host only learns which columns hold a NULL anywhere (a column without one carries no bitmap, as the host path decides).
qe_result *finish_hashed_groups_on_device(qe_ctx *ctx, const CodegenOutput &cg, const unsigned long long *d_entries, int64_t m, int64_t nrows,
                                          const int32_t *agg_fns, int32_t nagg) {
    const int W = cg.hash_words, NK = (int)cg.keys.size();
    std::vector<void *> temps;
    struct GT { qe_ctx *c; std::vector<void *> *t; ~GT() { for (void *q : *t) c->pool.release(q); } } gt{ctx, &temps};
    auto talloc = [&](size_t bytes) { void *q = ctx->pool.alloc(std::max<size_t>(bytes, 16)); temps.push_back(q); return q; };
    unsigned long long *keys[2] = {(unsigned long long *)talloc((size_t)m * 8), (unsigned long long *)talloc((size_t)m * 8)};
    uint32_t *rows[2] = {(uint32_t *)talloc((size_t)m * 4), (uint32_t *)talloc((size_t)m * 4)};
    uint32_t *hist = (uint32_t *)talloc((size_t)16 * (size_t)((m + 1023) / 1024) * 4);
    launch_group_sort_keys(ctx->stream, d_entries, W, 2 + NK, m, keys[0], rows[0]);
    int bits = 1;
    while (bits < 62 && (1ll << bits) < nrows) bits++;
    int cur = 0;
    for (int shift = 0; shift < bits; shift += 4) {
        launch_radix_pass(ctx->stream, keys[cur], rows[cur], nullptr, m, shift, hist, keys[1 - cur], rows[1 - cur]);
        cur = 1 - cur;
    }
    std::unique_ptr<qe_result, std::function<void(qe_result *)>> res(new qe_result(), [ctx](qe_result *r) { free_result(ctx, r); });
    res->count = m;
    res->capacity = m;
    const size_t words = (size_t)std::max<int64_t>(1, (m + 63) / 64);
    GroupFinishArgs a{};
    a.entries = d_entries;
    a.rows = rows[cur];
    a.m = m;
    a.words = W;
    a.nkeys = NK;
    a.nagg = nagg;
    unsigned int *d_flags = (unsigned int *)talloc(64);
    QE_HIP(hipMemsetAsync(d_flags, 0, 64, ctx->stream));
    a.flags = d_flags;
    for (int k = 0; k < NK; k++) {
        OutColumn oc;
        oc.type = cg.keys[k].type;
        oc.dict = cg.keys[k].dict;
        oc.dict_handle.d = oc.dict;
        const size_t bytes = oc.type == QE_BOOLEAN ? words * 8 : (oc.type == QE_DOUBLE || oc.type == QE_INT64) ? (size_t)m * 8 : (size_t)m * 4;
        oc.data = ctx->pool.alloc(std::max<size_t>(bytes, 16));
        oc.validity = (uint64_t *)ctx->pool.alloc(words * 8);
        res->cols.push_back(oc);
        a.key_type[k] = oc.type;
        a.key_data[k] = oc.data;
        a.key_valid[k] = (unsigned long long *)oc.validity;
    }
    for (int i = 0; i < nagg; i++) {
        OutColumn oc;
        oc.type = QE_DOUBLE;
        oc.data = ctx->pool.alloc(std::max<size_t>((size_t)m * 8, 16));
        oc.validity = (uint64_t *)ctx->pool.alloc(words * 8);
        res->cols.push_back(oc);
        a.agg_fn[i] = agg_fns[i];
        a.cnt_src[i] = cg.cnt_src[(size_t)i];
        a.agg_data[i] = (double *)oc.data;
        a.agg_valid[i] = (unsigned long long *)oc.validity;
    }
    launch_group_finish(ctx->stream, a);
    unsigned int flags[16] = {};
    QE_HIP(hipMemcpyAsync(flags, d_flags, 64, hipMemcpyDeviceToHost, ctx->stream));
    QE_HIP(hipGetLastError());
    QE_HIP(hipStreamSynchronize(ctx->stream));
    for (int c = 0; c < NK + nagg; c++) {   // a column without a NULL carries no bitmap
        OutColumn &oc = res->cols[(size_t)c];
        oc.nullable = flags[c < NK ? c : 4 + (c - NK)] != 0;
        if (!oc.nullable) {
            ctx->pool.release(oc.validity);
            oc.validity = nullptr;
        }
    }
    return res.release();
}

// The groups of a hashed GROUP BY, finished on the host: `dense` holds m entries of cg.hash_words words {state, null bits, key
// words.., first row, (count, acc)..}.  Insertion order = ascending first row (LinkedHashMap, GroupByAggregationOperator.kt:22);
// accumulators finish as Accumulators.kt:26-107 says.
qe_result *finish_hashed_groups(qe_ctx *ctx, const CodegenOutput &cg, const unsigned long long *dense, int64_t m,
                                const int32_t *agg_fns, int32_t nagg) {
    const int W = cg.hash_words, NK = (int)cg.keys.size(), ACC = 2 + NK;
    std::vector<std::pair<unsigned long long, int64_t>> order;
    order.reserve((size_t)m);
    for (int64_t g = 0; g < m; g++) order.emplace_back(dense[(size_t)g * W + ACC], g);
    if (m < 4096) {
        std::sort(order.begin(), order.end());
    } else {
        // first rows are distinct row ids < 2^42: three stable passes of a 14-bit radix sort (std::sort took ~80 of the 125 ms the
        // host spent finishing 1 M groups)
        std::vector<std::pair<unsigned long long, int64_t>> tmp(order.size());
        std::vector<size_t> cnt((size_t)1 << 14);
        unsigned long long all = 0;
        for (const auto &o : order) all |= o.first;
        for (int pass = 0; pass < 5 && (all >> (14 * pass)) != 0; pass++) {
            const int sh = 14 * pass;
            std::fill(cnt.begin(), cnt.end(), 0);
            for (const auto &o : order) cnt[(size_t)((o.first >> sh) & 0x3fffull)]++;
            size_t run = 0;
            for (size_t &c : cnt) { const size_t t = c; c = run; run += t; }
            for (const auto &o : order) tmp[cnt[(size_t)((o.first >> sh) & 0x3fffull)]++] = o;
            order.swap(tmp);
        }
    }
    std::unique_ptr<qe_result, std::function<void(qe_result *)>> res(new qe_result(), [ctx](qe_result *r) { free_result(ctx, r); });
    res->count = m;
    res->capacity = m;
    const size_t words = (size_t)std::max<int64_t>(1, (m + 63) / 64);
    auto upload = [&](const void *src, size_t bytes) -> void * {
        void *d = ctx->pool.alloc(std::max<size_t>(bytes, 16));
        if (bytes) QE_HIP(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return d;
    };
    std::vector<std::vector<unsigned long long>> keep64;   // host staging must outlive the async copies
    std::vector<std::vector<int32_t>> keep32;
    append_key_columns(ctx, cg, res.get(), m, [&](int64_t j) { return &dense[(size_t)order[(size_t)j].second * W + 1]; }, keep64, keep32);
    std::vector<std::vector<double>> keep_vals;
    for (int i = 0; i < nagg; i++) {
        OutColumn oc;
        oc.type = QE_DOUBLE;
        std::vector<double> vals((size_t)std::max<int64_t>(m, 1), 0.0);
        std::vector<unsigned long long> valid(words, 0);
        bool any_null = false;
        for (int64_t j = 0; j < m; j++) {
            const unsigned long long *e = &dense[(size_t)order[j].second * W] + ACC;   // {first row, (count, acc)..}
            const unsigned long long cnt = e[1 + 2 * cg.cnt_src[i]];
            const unsigned long long raw = e[2 + 2 * i];
            double v = 0.0;
            bool ok = true;
            switch (agg_fns[i]) {
            case QE_AGG_COUNT: v = (double)cnt; break;                       // Accumulators.kt:26-36
            case QE_AGG_SUM: std::memcpy(&v, &raw, 8); ok = cnt != 0; break;  // :47-53 empty => null
            case QE_AGG_AVG: std::memcpy(&v, &raw, 8); ok = cnt != 0; if (ok) v /= (double)cnt; break;
            default: {                                                        // MIN / MAX: undo the ordered key
                long long key = (long long)raw;
                long long b = key ^ ((key >> 63) & 0x7fffffffffffffffll);
                std::memcpy(&v, &b, 8);
                ok = cnt != 0;
            }
            }
            if (ok) valid[j >> 6] |= 1ull << (j & 63);
            else { any_null = true; v = 0.0; }
            vals[j] = v;
        }
        oc.nullable = any_null;
        keep_vals.push_back(std::move(vals));
        oc.data = upload(keep_vals.back().data(), (size_t)m * 8);
        if (any_null) {
            keep64.push_back(std::move(valid));
            oc.validity = (uint64_t *)upload(keep64.back().data(), words * 8);
        }
        res->cols.push_back(oc);
    }
    QE_HIP(hipStreamSynchronize(ctx->stream));
    return res.release();
}

// `many_keys` (optional): set -- and nullptr returned, nothing decided for the plan -- when the id build's FIRST table fills up
// (more than 32 768 keys): the caller has a better form for that many keys than a grown id table.
qe_result *run_groupby_hashed(qe_ctx *ctx, const qe_batch *batch, const Plan &plan, const qe_expr *filter, const qe_expr *const *exprs,
                              const int32_t *agg_fns, int32_t nagg, bool *many_keys) {
    const CodegenOutput &cg = plan.cg;
    const bool ids_allowed = (ctx->opts.tuning[5] & 131072) == 0 && batch->nrows < (1ll << 32);   // debug bit 131072: keep the global-atomic form
    bool ids_failed = plan.ids_overflow;   // more keys than a dense table takes, found out by an earlier execution: straight to the global-atomic form
    if (plan.use_ids && ids_allowed && !ids_failed) {
        qe_result *r = run_groupby_ids(ctx, batch, plan, filter, exprs, agg_fns, nagg, many_keys);
        if (r) return r;
        if (many_keys && *many_keys) return nullptr;
        plan.ids_overflow = ids_failed = true;
        plan.use_ids = false;
    }
    const int W = cg.hash_words, NK = (int)cg.keys.size(), ACC = 2 + NK;
    if (W > 40) fail(QE_ERR_UNSUPPORTED, "too many GROUP BY keys + aggregates for one hash entry");
    const int64_t n = batch->nrows;
    HtInit init{};
    init.words = W;
    for (int w = 0; w < W; w++) init.word[w] = 0;
    init.word[ACC] = ~0ull;   // smallest row id
    for (int i = 0; i < nagg; i++)
        init.word[ACC + 2 + 2 * i] = agg_fns[i] == QE_AGG_MIN ? 0x7fffffffffffffffull : agg_fns[i] == QE_AGG_MAX ? 0x8000000000000000ull : 0ull;
    std::vector<unsigned long long> dense;
    int64_t m = 0;
    if (n > 0) {
        int64_t C = plan.hash_capacity > 0 ? plan.hash_capacity : (1ll << 16);
        for (;;) {
            unsigned long long *d_tab = (unsigned long long *)ctx->pool.alloc((size_t)C * W * 8);
            struct G1 { qe_ctx *c; void *p; ~G1() { c->pool.release(p); } } g1{ctx, d_tab};
            launch_ht_init(ctx->stream, d_tab, C, init);
            QE_HIP(hipMemsetAsync(ctx->d_ctrl, 0, 96, ctx->stream));   // [0] entries in use (p.ticket), [1] error (p.error)
            const int64_t sub_rows = plan.geo.sub_rows();
            const int64_t ntiles = (n + sub_rows - 1) / sub_rows;
            const int waves = plan.geo.threads / 64;
            const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((ntiles + waves - 1) / waves, (int64_t)device_cus(ctx->device) * 4));
            FusedParams p;
            fill_inputs(p, batch, plan);
            p.agg_partial = (double *)d_tab;
            p.capacity = C;
            p.ticket = ctx->d_ctrl;
            p.error = ctx->d_ctrl + 1;
            p.stagger_chunks = ids_allowed && !ids_failed ? 1 : 0;   // a key that finds no room in LDS stops the launch (error 4) instead of going global
            launch_fused(ctx, plan, p, grid);
            QE_HIP(hipMemcpyAsync(ctx->h_ctrl, ctx->d_ctrl, 16, hipMemcpyDeviceToHost, ctx->stream));
            QE_HIP(hipStreamSynchronize(ctx->stream));
            const unsigned int *hc = (const unsigned int *)ctx->h_ctrl;
            if (hc[1] == 4) {   // the keys do not fit the LDS table: dense ids from now on (this execution included)
                plan.use_ids = true;
                qe_result *r = run_groupby_ids(ctx, batch, plan, filter, exprs, agg_fns, nagg, many_keys);
                if (r) return r;
                if (many_keys && *many_keys) return nullptr;
                ids_failed = true;   // more keys than a dense table takes: the global-atomic form after all -- and remembered,
                plan.ids_overflow = true;   // so that later executions do not run the failing id build again
                plan.use_ids = false;
                continue;
            }
            collect_time(ctx);
            if (hc[1] != 0) {   // more than half full (or a probe sequence ran out): a bigger table, again
                if (C >= (1ll << 28)) fail(QE_ERR_UNSUPPORTED, "GROUP BY produced more than 2^27 groups");
                C *= 8;
                continue;
            }
            plan.hash_capacity = C;
            const int64_t used = hc[0];
            // more than a few dozen keys: probing the LDS table costs more than resolving ids first (300 keys: 15 ms here,
            // 6.6 ms as build pass + dense LDS group-by; crossover ~70 keys) -- the next executions of this plan go that way
            static const int64_t ids_from = std::getenv("QE_IDS_FROM") ? std::atoll(std::getenv("QE_IDS_FROM")) : 64;
            if (used > ids_from && ids_allowed && !plan.ids_overflow) plan.use_ids = true;
            unsigned long long *d_dense = (unsigned long long *)ctx->pool.alloc((size_t)std::max<int64_t>(used, 1) * W * 8);
            struct G2 { qe_ctx *c; void *p; ~G2() { c->pool.release(p); } } g2{ctx, d_dense};
            QE_HIP(hipMemsetAsync(ctx->d_ctrl, 0, 16, ctx->stream));
            launch_ht_collect(ctx->stream, d_tab, C, W, d_dense, ctx->d_ctrl);
            dense.resize((size_t)used * W);
            if (used > 0) QE_HIP(hipMemcpyAsync(dense.data(), d_dense, dense.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
            QE_HIP(hipGetLastError());
            QE_HIP(hipStreamSynchronize(ctx->stream));
            m = used;
            break;
        }
    }
    return finish_hashed_groups(ctx, cg, dense.data(), m, agg_fns, nagg);
}

// HASH-PARTITIONED form of a hashed GROUP BY with many distinct keys (round 3): count -> scan -> scatter of
// {header, aggregate inputs, key words} records by key HASH (the dense partitioned passes over a pseudo group id) -> ONE
// workgroup per partition aggregates its records in an LDS hash table and appends the used entries to the result.  Every pass
// streams; no gather, no global atomic per row.  nullptr: a partition held more distinct keys than its table has buckets (the
// caller takes another path and remembers).
// `skewed` (optional) is set, and nullptr returned before the scatter pass, when one partition holds several times its share of
// the records (a key that owns a large part of the rows): ONE workgroup aggregates a partition, so that workgroup would run
// alone for most of the pass -- the dense-id path slices its partitions and does not mind.
qe_result *run_groupby_hp(qe_ctx *ctx, const qe_batch *batch, const std::shared_ptr<Plan> &plan, const int32_t *agg_fns, int32_t nagg,
                          bool *skewed = nullptr) {
    const CodegenOutput &cg = plan->cg;
    const int64_t n = batch->nrows;
    const int P = cg.nparts, HW = cg.hash_words;
    const int waves = plan->geo.threads / 64;
    const int64_t chunk_rows = plan->geo.chunk_rows() * waves;
    const int64_t nchunks = (n + chunk_rows - 1) / chunk_rows;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(nchunks, (int64_t)device_cus(ctx->device) * 8));
    hipFunction_t f_count = nullptr, f_scatter = nullptr, f_agg = nullptr;
    QE_HIP(hipModuleGetFunction(&f_count, plan->kernel.module, "qe_gb_count"));
    QE_HIP(hipModuleGetFunction(&f_scatter, plan->kernel.module, "qe_gb_scatter"));
    QE_HIP(hipModuleGetFunction(&f_agg, plan->kernel.module, "qe_gb_aggregate"));
    std::vector<void *> temps;
    struct GT { qe_ctx *c; std::vector<void *> *t; ~GT() { for (void *q : *t) c->pool.release(q); } } gt{ctx, &temps};
    auto talloc = [&](size_t bytes) { void *q = ctx->pool.alloc(std::max<size_t>(bytes, 16)); temps.push_back(q); return q; };
    uint32_t *d_counts = (uint32_t *)talloc((size_t)nchunks * P * 4);
    unsigned long long *d_start = (unsigned long long *)talloc((size_t)(P + 1) * 8);
    FusedParams p;
    fill_inputs(p, batch, *plan);
    p.nchunks = nchunks;
    p.blk = (unsigned long long *)d_counts;
    void *args[] = {&p};
    if (ctx->opts.profile) QE_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    QE_HIP(hipModuleLaunchKernel(f_count, grid, 1, 1, plan->geo.threads, 1, 1, 0, ctx->stream, args, nullptr));
    launch_gb_scan(ctx->stream, d_counts, nchunks, P, d_start);
    std::vector<unsigned long long> start((size_t)P + 1, 0);
    QE_HIP(hipMemcpyAsync(start.data(), d_start, (size_t)P * 8, hipMemcpyDeviceToHost, ctx->stream));
    QE_HIP(hipStreamSynchronize(ctx->stream));
    unsigned long long m_records = 0;
    for (int j = 0; j < P; j++) {
        const unsigned long long cnt = start[j];
        start[j] = m_records;
        m_records += cnt;
    }
    start[P] = m_records;
    if (m_records >= (1ull << 32)) return nullptr;   // record positions are 32-bit in the scatter pass
    if (skewed) {
        unsigned long long largest = 0;
        for (int j = 0; j < P; j++) largest = std::max(largest, start[j + 1] - start[j]);
        if (m_records > (1ull << 22) && largest > 3 * (m_records / (unsigned long long)P) + 65536) {
            *skewed = true;
            if (ctx->opts.profile) {   // (the bracket the caller reads must be closed)
                QE_HIP(hipEventRecord(ctx->ev1, ctx->stream));
                QE_HIP(hipStreamSynchronize(ctx->stream));
                collect_time(ctx);
            }
            return nullptr;
        }
    }
    QE_HIP(hipMemcpyAsync(d_start, start.data(), (size_t)(P + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    // the groups' entries come back through pinned staging (64 MB for 1 M groups: 1.5 ms instead of ~15 ms into pageable memory)
    unsigned long long *dense = nullptr;
    struct PG { qe_ctx *c; unsigned long long **p; ~PG() { if (*p) c->pinned.release(*p); } } pg{ctx, &dense};
    int64_t m = 0;
    if (m_records > 0) {
        p.l1 = d_start;
        const int rec_words = 1 + cg.nvals;
        static const bool dbg_times = std::getenv("QE_DEBUG_TIMES") != nullptr;
        const auto t_alloc0 = std::chrono::steady_clock::now();
        if (cg.hp_line_recs) p.desc = (unsigned long long *)talloc((size_t)(m_records / cg.hp_line_recs + 2) * 128);   // whole lines + the spare line
        else p.desc = (unsigned long long *)talloc((size_t)(m_records + 16) * 8 * rec_words);
        if (dbg_times)
            std::fprintf(stderr, "run_groupby_hp: record array of %.2f GB from the pool in %.1f ms\n",
                         (cg.hp_line_recs ? (double)(m_records / cg.hp_line_recs + 2) * 128 : (double)(m_records + 16) * 8 * rec_words) / 1e9,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_alloc0).count());
        const int sgrid = (int)std::max<int64_t>(1, std::min<int64_t>(grid, (int64_t)device_cus(ctx->device) * (plan->geo.threads >= 512 ? 1 : kScatterWgsPerCu)));
        hipDeviceptr_t dbg = nullptr;
        size_t dbg_bytes = 0;
        if (ctx->opts.tuning[5] & 64) {
            QE_HIP(hipModuleGetGlobal(&dbg, &dbg_bytes, plan->kernel.module, "qe_dbg"));
            QE_HIP(hipMemsetAsync(dbg, 0, dbg_bytes, ctx->stream));
        }
        QE_HIP(hipModuleLaunchKernel(f_scatter, sgrid, 1, 1, plan->geo.threads, 1, 1, 0, ctx->stream, args, nullptr));
        if (dbg) {   // diagnostic build: shader clocks per phase, summed over the waves
            unsigned long long h[8] = {};
            QE_HIP(hipMemcpyAsync(h, dbg, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
            QE_HIP(hipStreamSynchronize(ctx->stream));
            const double waves_total = (double)sgrid * waves;
            std::fprintf(stderr, "qe_gb_scatter (hash-partitioned) phases, clocks per wave (grid %d x %d waves): issue loads %.0f | flush (stores) %.0f | "
                         "LDS sort %.0f | wait loads + evaluate %.0f | chunk drain %.0f\n", sgrid, waves, h[0] / waves_total,
                         h[1] / waves_total, h[2] / waves_total, h[3] / waves_total, h[4] / waves_total);
        }
        const int64_t cap = (int64_t)P * cg.part_groups;   // every bucket of every partition: cannot be exceeded
        unsigned long long *d_out = (unsigned long long *)talloc((size_t)cap * HW * 8);
        p.agg_partial = (double *)d_out;
        p.capacity = cap;
        p.ticket = ctx->d_ctrl;
        p.error = ctx->d_ctrl + 1;
        QE_HIP(hipMemsetAsync(ctx->d_ctrl, 0, 96, ctx->stream));
        QE_HIP(hipModuleLaunchKernel(f_agg, P, 1, 1, 1024, 1, 1, 0, ctx->stream, args, nullptr));
        if (ctx->opts.profile) QE_HIP(hipEventRecord(ctx->ev1, ctx->stream));
        QE_HIP(hipMemcpyAsync(ctx->h_ctrl, ctx->d_ctrl, 16, hipMemcpyDeviceToHost, ctx->stream));
        QE_HIP(hipStreamSynchronize(ctx->stream));
        collect_time(ctx);
        const unsigned int *hc = (const unsigned int *)ctx->h_ctrl;
        if (hc[1] != 0) return nullptr;   // 6: some partition's table filled up
        m = hc[0];
        static const int64_t device_finish_from = std::getenv("QE_GROUPS_ON_DEVICE_FROM") ? std::atoll(std::getenv("QE_GROUPS_ON_DEVICE_FROM")) : 4096;
        if (m >= ((ctx->opts.tuning[5] & 67108864) ? 1 : device_finish_from) && (int)cg.keys.size() <= 4 && nagg <= 8)   // (debug bit 67108864: always)
            return finish_hashed_groups_on_device(ctx, cg, d_out, m, batch->nrows, agg_fns, nagg);
        if (m > 0) {
            dense = (unsigned long long *)ctx->pinned.alloc((size_t)m * HW * 8);
            QE_HIP(hipMemcpyAsync(dense, d_out, (size_t)m * HW * 8, hipMemcpyDeviceToHost, ctx->stream));
        }
        QE_HIP(hipStreamSynchronize(ctx->stream));
    } else {
        if (ctx->opts.profile) QE_HIP(hipEventRecord(ctx->ev1, ctx->stream));
        QE_HIP(hipStreamSynchronize(ctx->stream));
        collect_time(ctx);
    }
    return finish_hashed_groups(ctx, cg, dense, m, agg_fns, nagg);
}

// GroupByAggregation over a dense group id (dictionary / boolean keys): LDS-privatised table, partitioned passes or global
// atomics, the groups finished on the host in insertion order.
qe_result *run_groupby_dense(qe_ctx *ctx, const qe_batch *batch, const std::shared_ptr<Plan> &plan, const int32_t *agg_fns, int32_t nagg) {
    const CodegenOutput &cg = plan->cg;
    const int64_t G = cg.ngroups;
    const int W = cg.table_words;
    // global accumulator table, initialised from the host (smallest row = ~0, MIN/MAX keys at their identity)
    const int copies = cg.table_copies;
    std::vector<unsigned long long> tab((size_t)G * W * copies);
    for (int64_t g = 0; g < G * copies; g++) {
        unsigned long long *e = &tab[(size_t)g * W];
        e[0] = ~0ull;
        for (int i = 0; i < nagg; i++) {
            e[1 + 2 * i] = 0;
            e[2 + 2 * i] = agg_fns[i] == QE_AGG_MIN ? 0x7fffffffffffffffull : agg_fns[i] == QE_AGG_MAX ? 0x8000000000000000ull : 0ull;
        }
    }
    unsigned long long *d_tab = (unsigned long long *)ctx->pool.alloc(tab.size() * 8);
    struct G1 { qe_ctx *c; void *p; ~G1() { c->pool.release(p); } } g1{ctx, d_tab};
    QE_HIP(hipMemcpyAsync(d_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    const int64_t n = batch->nrows;
    const bool no_partition = (ctx->opts.tuning[5] & 256) != 0;   // debug bit 256: keep the global-atomic path (A/B measurements, tests)
    if (n > 0 && n < (1ll << 32) && cg.partitioned && !no_partition) {   // record positions are 32-bit in the scatter pass
        // Domain too large for an LDS table: count -> scan -> scatter -> per-partition LDS aggregation
        // (two streaming passes over the input and one over the records instead of one global atomic per value).
        const int P = cg.nparts;
        const int waves = plan->geo.threads / 64;
        // a chunk = subs_per_chunk workgroup tiles (one sub-tile per wave each): the unit both passes hand to a workgroup
        const int64_t chunk_rows = plan->geo.chunk_rows() * waves;
        const int64_t nchunks = (n + chunk_rows - 1) / chunk_rows;
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(nchunks, (int64_t)device_cus(ctx->device) * 8));
        hipFunction_t f_count = nullptr, f_scatter = nullptr;
        QE_HIP(hipModuleGetFunction(&f_count, plan->kernel.module, "qe_gb_count"));
        QE_HIP(hipModuleGetFunction(&f_scatter, plan->kernel.module, "qe_gb_scatter"));
        std::vector<void *> temps;
        struct GT { qe_ctx *c; std::vector<void *> *t; ~GT() { for (void *q : *t) c->pool.release(q); } } gt{ctx, &temps};
        auto talloc = [&](size_t bytes) { void *q = ctx->pool.alloc(std::max<size_t>(bytes, 16)); temps.push_back(q); return q; };
        uint32_t *d_counts = (uint32_t *)talloc((size_t)nchunks * P * 4);
        unsigned long long *d_start = (unsigned long long *)talloc((size_t)(P + 1) * 8);
        FusedParams p;
        fill_inputs(p, batch, *plan);
        p.nchunks = nchunks;
        p.blk = (unsigned long long *)d_counts;
        void *args[] = {&p};
        if (ctx->opts.profile) QE_HIP(hipEventRecord(ctx->ev0, ctx->stream));
        QE_HIP(hipModuleLaunchKernel(f_count, grid, 1, 1, plan->geo.threads, 1, 1, 0, ctx->stream, args, nullptr));
        launch_gb_scan(ctx->stream, d_counts, nchunks, P, d_start);
        std::vector<unsigned long long> start((size_t)P + 1, 0);
        QE_HIP(hipMemcpyAsync(start.data(), d_start, (size_t)P * 8, hipMemcpyDeviceToHost, ctx->stream));
        QE_HIP(hipStreamSynchronize(ctx->stream));
        unsigned long long m_records = 0;
        for (int j = 0; j < P; j++) {
            const unsigned long long cnt = start[j];
            start[j] = m_records;
            m_records += cnt;
        }
        start[P] = m_records;
        QE_HIP(hipMemcpyAsync(d_start, start.data(), (size_t)(P + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        if (m_records > 0) {
            p.l1 = d_start;
            const int rec_words = 1 + cg.nvals;
            if (m_records >= (1ull << 32)) fail(QE_ERR_UNSUPPORTED, "partitioned GROUP BY: more than 2^32 records");
            p.desc = (unsigned long long *)talloc((size_t)(m_records + 16) * 8 * rec_words);   // + the spare line the scatter's idle threads write
            // two workgroups per CU: the 64 KiB LDS stage of the tile sort lets two share a CU (one sorts and stores while the
            // other waits for its loads)
            static const int scatter_wgs = std::getenv("QE_GB_SCATTER_WGS_PER_CU") ? std::atoi(std::getenv("QE_GB_SCATTER_WGS_PER_CU")) : kScatterWgsPerCu;
            const int sgrid = (int)std::max<int64_t>(1, std::min<int64_t>(grid, (int64_t)device_cus(ctx->device) * (plan->geo.threads >= 512 ? 1 : std::max(1, scatter_wgs))));
            hipDeviceptr_t dbg = nullptr;
            size_t dbg_bytes = 0;
            if (ctx->opts.tuning[5] & 64) {
                QE_HIP(hipModuleGetGlobal(&dbg, &dbg_bytes, plan->kernel.module, "qe_dbg"));
                QE_HIP(hipMemsetAsync(dbg, 0, dbg_bytes, ctx->stream));
            }
            QE_HIP(hipModuleLaunchKernel(f_scatter, sgrid, 1, 1, plan->geo.threads, 1, 1, 0, ctx->stream, args, nullptr));
            if (dbg) {   // diagnostic build: shader clocks per phase, summed over the waves
                unsigned long long h[8] = {};
                QE_HIP(hipMemcpyAsync(h, dbg, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
                QE_HIP(hipStreamSynchronize(ctx->stream));
                const double waves_total = (double)sgrid * waves;
                std::fprintf(stderr, "qe_gb_scatter phases, clocks per wave (grid %d x %d waves): issue loads %.0f | flush (stores) %.0f | "
                             "LDS sort %.0f | wait loads + evaluate %.0f | chunk drain %.0f\n", sgrid, waves, h[0] / waves_total,
                             h[1] / waves_total, h[2] / waves_total, h[3] / waves_total, h[4] / waves_total);
            }
            // pass 3 (generated per plan): ~512 workgroups, one LDS table each, merged into the global table
            static const int agg_wgs = std::getenv("QE_GB_AGG_WGS") ? std::atoi(std::getenv("QE_GB_AGG_WGS")) : 512;
            const int slices = std::max(1, std::min(64, agg_wgs / P));
            const size_t lds = (size_t)cg.part_groups * W * 8;
            const int agg_threads = lds > 48 * 1024 ? 1024 : 256;   // a table that leaves room for one workgroup per CU: make it a big one
            hipFunction_t f_agg = nullptr;
            QE_HIP(hipModuleGetFunction(&f_agg, plan->kernel.module, "qe_gb_aggregate"));
            p.agg_partial = (double *)d_tab;
            QE_HIP(hipModuleLaunchKernel(f_agg, slices, P, 1, agg_threads, 1, 1, 0, ctx->stream, args, nullptr));
        }
        if (ctx->opts.profile) QE_HIP(hipEventRecord(ctx->ev1, ctx->stream));
        QE_HIP(hipStreamSynchronize(ctx->stream));   // the temporaries go back to the pool when this scope ends
    } else if (n > 0) {
        const int64_t sub_rows = plan->geo.sub_rows();
        const int64_t ntiles = (n + sub_rows - 1) / sub_rows;
        const int waves = plan->geo.threads / 64;
        const int wgs_per_cu = plan->geo.threads >= 1024 ? 1 : 4;   // a 1024-thread workgroup owns its CU (and merges its table once)
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((ntiles + waves - 1) / waves, (int64_t)device_cus(ctx->device) * wgs_per_cu));
        FusedParams p;
        fill_inputs(p, batch, *plan);
        p.agg_partial = (double *)d_tab;
        launch_fused(ctx, *plan, p, grid);
    }
    QE_HIP(hipMemcpyAsync(tab.data(), d_tab, tab.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    QE_HIP(hipStreamSynchronize(ctx->stream));
    if (n > 0) collect_time(ctx);
    // fold the per-XCD copies into copy 0, in XCD order
    for (int c = 1; c < copies; c++) {
        for (int64_t g = 0; g < G; g++) {
            unsigned long long *d = &tab[(size_t)g * W];
            const unsigned long long *e = &tab[((size_t)c * G + g) * W];
            if (e[0] == ~0ull) continue;
            d[0] = std::min(d[0], e[0]);
            for (int i = 0; i < nagg; i++) {
                if (e[1 + 2 * cg.cnt_src[i]] == 0) continue;
                if (cg.cnt_src[i] == i) d[1 + 2 * i] += e[1 + 2 * i];
                if (agg_fns[i] == QE_AGG_SUM || agg_fns[i] == QE_AGG_AVG) {
                    double a, b;
                    std::memcpy(&a, &d[2 + 2 * i], 8);
                    std::memcpy(&b, &e[2 + 2 * i], 8);
                    a += b;
                    std::memcpy(&d[2 + 2 * i], &a, 8);
                } else if (agg_fns[i] == QE_AGG_MIN) {
                    d[2 + 2 * i] = (unsigned long long)std::min((long long)d[2 + 2 * i], (long long)e[2 + 2 * i]);
                } else if (agg_fns[i] == QE_AGG_MAX) {
                    d[2 + 2 * i] = (unsigned long long)std::max((long long)d[2 + 2 * i], (long long)e[2 + 2 * i]);
                }
            }
        }
    }
    // groups in insertion order = ascending first row (LinkedHashMap order, GroupByAggregationOperator.kt:22)
    std::vector<std::pair<unsigned long long, int64_t>> order;
    for (int64_t g = 0; g < G; g++)
        if (tab[(size_t)g * W] != ~0ull) order.emplace_back(tab[(size_t)g * W], g);
    std::sort(order.begin(), order.end());
    const int64_t m = (int64_t)order.size();
    std::unique_ptr<qe_result, std::function<void(qe_result *)>> res(new qe_result(), [ctx](qe_result *r) { free_result(ctx, r); });
    res->count = m;
    res->capacity = m;
    const size_t words = (size_t)std::max<int64_t>(1, (m + 63) / 64);
    auto upload = [&](const void *src, size_t bytes) -> void * {
        void *d = ctx->pool.alloc(std::max<size_t>(bytes, 16));
        if (bytes) QE_HIP(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return d;
    };
    std::vector<std::vector<unsigned long long>> keep_words;   // host staging must outlive the async copies
    std::vector<std::vector<int32_t>> keep_codes;
    std::vector<std::vector<double>> keep_vals;
    int64_t stride = 1;
    for (size_t k = 0; k < cg.keys.size(); k++) {
        const int domain = cg.key_domain[k];
        OutColumn oc;
        oc.type = cg.keys[k].type;
        oc.dict = cg.keys[k].dict;
        oc.dict_handle.d = oc.dict;
        std::vector<unsigned long long> valid(words, 0), bits(words, 0);
        std::vector<int32_t> codes((size_t)std::max<int64_t>(m, 1), 0);
        bool any_null = false;
        for (int64_t j = 0; j < m; j++) {
            const int code = (int)((order[j].second / stride) % (domain + 1));
            if (code == domain) { any_null = true; continue; }
            valid[j >> 6] |= 1ull << (j & 63);
            codes[j] = code;
            if (code) bits[j >> 6] |= 1ull << (j & 63);
        }
        oc.nullable = any_null;
        if (oc.type == QE_BOOLEAN) {
            keep_words.push_back(bits);
            oc.data = upload(keep_words.back().data(), words * 8);
        } else {
            keep_codes.push_back(codes);
            oc.data = upload(keep_codes.back().data(), (size_t)m * 4);
        }
        if (any_null) {
            keep_words.push_back(valid);
            oc.validity = (uint64_t *)upload(keep_words.back().data(), words * 8);
        }
        res->cols.push_back(oc);
        stride *= (domain + 1);
    }
    for (int i = 0; i < nagg; i++) {
        OutColumn oc;
        oc.type = QE_DOUBLE;
        std::vector<double> vals((size_t)std::max<int64_t>(m, 1), 0.0);
        std::vector<unsigned long long> valid(words, 0);
        bool any_null = false;
        for (int64_t j = 0; j < m; j++) {
            const unsigned long long *e = &tab[(size_t)order[j].second * W];
            const unsigned long long cnt = e[1 + 2 * cg.cnt_src[i]];
            const unsigned long long raw = e[2 + 2 * i];
            double v = 0.0;
            bool ok = true;
            switch (agg_fns[i]) {
            case QE_AGG_COUNT: v = (double)cnt; break;                       // Accumulators.kt:26-36
            case QE_AGG_SUM: std::memcpy(&v, &raw, 8); ok = cnt != 0; break;  // :47-53 empty => null
            case QE_AGG_AVG: std::memcpy(&v, &raw, 8); ok = cnt != 0; if (ok) v /= (double)cnt; break;
            default: {                                                        // MIN / MAX: undo the ordered key
                long long key = (long long)raw;
                long long b = key ^ ((key >> 63) & 0x7fffffffffffffffll);
                std::memcpy(&v, &b, 8);
                ok = cnt != 0;
            }
            }
            if (ok) valid[j >> 6] |= 1ull << (j & 63);
            else { any_null = true; v = 0.0; }
            vals[j] = v;
        }
        oc.nullable = any_null;
        keep_vals.push_back(vals);
        oc.data = upload(keep_vals.back().data(), (size_t)m * 8);
        if (any_null) {
            keep_words.push_back(valid);
            oc.validity = (uint64_t *)upload(keep_words.back().data(), words * 8);
        }
        res->cols.push_back(oc);
    }
    QE_HIP(hipStreamSynchronize(ctx->stream));
    return res.release();
}

}  // namespace

// ---- internals of the overlapped scan + exchange (qe_comm.cpp: qe_filter_project_gather) ---------------------------------
// kept rows of every slice of `slice_rows` rows (a multiple of the plan's chunk) of the batch: ONE launch of the count pass
// (qe_fp_count: the filter's columns only, late materialisation included) and a 4-byte read-back per chunk
std::vector<int64_t> qe_int_count_slices(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter, const qe_expr *const *projs, int32_t nproj,
                                  int64_t *slice_rows_io, int32_t nslices) {
    const int64_t n = batch->nrows;
    auto plan = get_plan(ctx, batch, filter, projs, nproj, nullptr, true);
    const int64_t chunk_rows = plan->geo.chunk_rows();
    int64_t slice_rows = (n + nslices - 1) / std::max(1, nslices);
    slice_rows = std::max<int64_t>(chunk_rows, (slice_rows + chunk_rows - 1) / chunk_rows * chunk_rows);
    *slice_rows_io = slice_rows;
    const int64_t ns = n > 0 ? (n + slice_rows - 1) / slice_rows : 0;
    std::vector<int64_t> counts((size_t)ns, 0);
    if (n == 0) return counts;
    if (!filter || !plan->cg.two_pass) {   // no Filter node: every row is kept
        for (int64_t k = 0; k < ns; k++) counts[(size_t)k] = std::min(slice_rows, n - k * slice_rows);
        return counts;
    }
    const int64_t nchunks = (n + chunk_rows - 1) / chunk_rows;
    const int waves = plan->geo.threads / 64;
    hipFunction_t f_count = nullptr;
    QE_HIP(hipModuleGetFunction(&f_count, plan->kernel.module, "qe_fp_count"));
    uint32_t *d_counts = (uint32_t *)ctx->pool.alloc((size_t)nchunks * 4);
    struct G { qe_ctx *c; void *q; ~G() { c->pool.release(q); } } g{ctx, d_counts};
    FusedParams p;
    fill_inputs(p, batch, *plan);
    p.nchunks = nchunks;
    p.blk = (unsigned long long *)d_counts;
    void *args[] = {&p};
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((nchunks + waves - 1) / waves, (int64_t)device_cus(ctx->device) * 8));
    QE_HIP(hipModuleLaunchKernel(f_count, grid, 1, 1, plan->geo.threads, 1, 1, 0, ctx->stream, args, nullptr));
    std::vector<uint32_t> h((size_t)nchunks);
    QE_HIP(hipMemcpyAsync(h.data(), d_counts, (size_t)nchunks * 4, hipMemcpyDeviceToHost, ctx->stream));
    QE_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t per = slice_rows / chunk_rows;
    for (int64_t c = 0; c < nchunks; c++) counts[(size_t)(c / per)] += h[(size_t)c];
    return counts;
}

// Projection(Filter(Scan)) over rows [row_begin, row_begin + nrows) of the batch (row_begin a multiple of 64: bitmap words do
// not straddle the cut): a view of the batch's columns, no copy
qe_result *qe_int_run_fused_slice(qe_ctx *ctx, const qe_batch *batch, int64_t row_begin, int64_t nrows, const qe_expr *filter,
                           const qe_expr *const *projs, int32_t nproj) {
    if (row_begin % 64 != 0 || row_begin < 0 || nrows < 0 || row_begin + nrows > batch->nrows) fail(QE_ERR_INTERNAL, "bad slice");
    qe_batch view;
    view.nrows = nrows;
    for (const Column &c : batch->cols) {
        Column v = c;
        v.owned = false;
        if (c.data) v.data = (char *)c.data + (c.type == QE_BOOLEAN ? (size_t)(row_begin / 64) * 8 : type_width(c.type) * (size_t)row_begin);
        if (c.validity) v.validity = c.validity + row_begin / 64;
        view.cols.push_back(v);
    }
    return run_fused(ctx, &view, filter, projs, nproj);
}

extern "C" {

int32_t qe_filter_project(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter,
                          const qe_expr *const *projections, int32_t nproj, qe_result **out) {
    if (!ctx || !batch || !out) return QE_ERR_INVALID_ARG;
    *out = nullptr;
    return guarded(ctx, [&] {
        need_device(ctx);
        if (batch->schema_only) fail(QE_ERR_INVALID_ARG, "schema-only batch (qe_batch_describe) cannot be executed");
        if (ctx->opts.exec_mode == QE_EXEC_PER_NODE) {
            *out = run_per_node(ctx, batch, filter, projections, nproj);
            ctx->last_form = QE_FORM_PER_NODE;
        } else
            *out = run_fused(ctx, batch, filter, projections, nproj);
    });
}

int32_t qe_filter_project_prepare(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter,
                                  const qe_expr *const *projections, int32_t nproj) {
    if (!ctx || !batch) return QE_ERR_INVALID_ARG;
    return guarded(ctx, [&] {
        if (ctx->device >= 0) need_device(ctx);
        if (ctx->opts.exec_mode == QE_EXEC_FUSED) {
            auto pl = get_plan(ctx, batch, filter, projections, nproj, nullptr, ctx->device >= 0);
            // the second geometry candidate (qe_ctx::geo_choice) is compiled ahead of time as well, so that the choice on
            // the device never waits for the JIT
            const int k2 = pl->geo.unroll > 0 ? (pl->est_regs - 54) / (2 * pl->geo.unroll) : 99;
            if (!pl->explicit_geometry && pl->est_regs > 0 && 32 * k2 + 54 <= 256 && (ctx->opts.tuning[5] & 8192) == 0)
                for (int c = 1; c < qe_ctx::GeoChoice::kCands; c++)
                    try {
                        (void)get_plan(ctx, batch, filter, projections, nproj, nullptr, ctx->device >= 0, nullptr, 0, c);
                    } catch (const Error &) {   // optional candidates: the default plan above is what prepare guarantees
                    }
            // the dense single-pass kernel (plans that keep a large share of their rows) is compiled ahead of time as well
            if (filter && (ctx->opts.tuning[5] & 32768) == 0)
                (void)get_plan(ctx, batch, filter, projections, nproj, nullptr, ctx->device >= 0, nullptr, 0, false, true);
        }
    });
}

int32_t qe_filter_project_source(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter,
                                 const qe_expr *const *projections, int32_t nproj, const char **out) {
    if (!ctx || !batch || !out) return QE_ERR_INVALID_ARG;
    return guarded(ctx, [&] {
        const bool dense = filter && (ctx->opts.tuning[5] & 16384) != 0;   // a context forced into the dense form shows that kernel
        auto plan = get_plan(ctx, batch, filter, projections, nproj, nullptr, false, nullptr, 0, false, dense);
        ctx->source_scratch = plan->cg.source;
        *out = ctx->source_scratch.c_str();
    });
}

int32_t qe_filter_project_geometry(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter,
                                   const qe_expr *const *projections, int32_t nproj, int32_t *out_chosen, int32_t *out_from_cache) {
    if (!ctx || !batch || !out_chosen) return QE_ERR_INVALID_ARG;
    return guarded(ctx, [&] {
        auto plan = get_plan(ctx, batch, filter, projections, nproj, nullptr, false);
        if (!plan->conj_order.empty())   // the geometry memory belongs to the plan in its chosen conjunct order
            plan = get_plan(ctx, batch, filter, projections, nproj, nullptr, false, nullptr, 0, false, false, &plan->conj_order);
        *out_chosen = -1;
        if (out_from_cache) *out_from_cache = 0;
        auto it = ctx->geo_choice.find(plan.get());
        if (it != ctx->geo_choice.end()) {
            *out_chosen = it->second.chosen;
            if (out_from_cache) *out_from_cache = it->second.from_cache ? 1 : 0;
        } else {
            double margin = 1.0;
            const int saved = ctx->jit->load_choice(plan->cg.source, &margin);
            if (saved >= 0 && margin >= 0.07) {   // (a closer call is measured again by the next execution on this context)
                *out_chosen = saved;
                if (out_from_cache) *out_from_cache = 1;
            }
        }
    });
}

int32_t qe_filter_project_conjunct_order(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter,
                                         const qe_expr *const *projections, int32_t nproj, int32_t *out_order, int32_t capacity,
                                         int32_t *out_nconj) {
    if (!ctx || !batch || !out_nconj || capacity < 0 || (capacity > 0 && !out_order)) return QE_ERR_INVALID_ARG;
    return guarded(ctx, [&] {
        auto plan = get_plan(ctx, batch, filter, projections, nproj, nullptr, false);
        const int K = plan->cg.nconj;
        *out_nconj = plan->conj_decided ? K : -1;
        for (int i = 0; i < K && i < capacity; i++) out_order[i] = plan->conj_order.empty() ? i : plan->conj_order[(size_t)i];
    });
}

int32_t qe_filter_aggregate_prepare(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter,
                                    const qe_expr *const *exprs, const int32_t *agg_fns, int32_t nagg) {
    if (!ctx || !batch || nagg <= 0 || !exprs || !agg_fns) return QE_ERR_INVALID_ARG;
    return guarded(ctx, [&] {
        if (ctx->device >= 0) need_device(ctx);
        (void)get_plan(ctx, batch, filter, exprs, nagg, agg_fns, ctx->device >= 0);
    });
}

int32_t qe_filter_aggregate(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter, const qe_expr *const *exprs,
                            const int32_t *agg_fns, int32_t nagg, double *out_values, uint8_t *out_valid,
                            int64_t *out_selected_rows) {
    if (!ctx || !batch || nagg <= 0 || !exprs || !agg_fns || !out_values || !out_valid) return QE_ERR_INVALID_ARG;
    return guarded(ctx, [&] {
        need_device(ctx);
        if (batch->schema_only) fail(QE_ERR_INVALID_ARG, "schema-only batch (qe_batch_describe) cannot be executed");
        auto plan = get_plan(ctx, batch, filter, exprs, nagg, agg_fns, true);
        const int64_t n = batch->nrows;
        const int64_t sub_rows = plan->geo.sub_rows();
        const int64_t ntiles = (n + sub_rows - 1) / sub_rows;
        // fixed grid => fixed reduction tree => bitwise reproducible sums on a given device
        const int waves = plan->geo.threads / 64;
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((ntiles + waves - 1) / waves,
                                                                      (int64_t)device_cus(ctx->device) * 8));
        const int stride = 2 * nagg + 1;
        std::vector<double> partial((size_t)grid * stride, 0.0);
        if (ntiles > 0) {
            FusedParams p;
            fill_inputs(p, batch, *plan);
            double *d_partial = (double *)ctx->pool.alloc(partial.size() * 8);
            struct G { qe_ctx *c; void *p; ~G() { c->pool.release(p); } } g{ctx, d_partial};
            p.agg_partial = d_partial;
            p.nchunks = ntiles;
            launch_fused(ctx, *plan, p, grid);
            QE_HIP(hipMemcpyAsync(partial.data(), d_partial, partial.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
            QE_HIP(hipStreamSynchronize(ctx->stream));
            collect_time(ctx);
        }
        int64_t nsel = 0;
        for (int i = 0; i < nagg; i++) {
            const int fn = agg_fns[i];
            double acc = fn == QE_AGG_MIN ? INFINITY : fn == QE_AGG_MAX ? -INFINITY : 0.0;
            double cnt = 0;
            if (ntiles > 0)
                for (int b = 0; b < grid; b++) {
                    const double a = partial[(size_t)b * stride + 2 * i], c = partial[(size_t)b * stride + 2 * i + 1];
                    cnt += c;
                    if (fn == QE_AGG_MIN) acc = (a != a || acc != acc) ? (acc != acc ? acc : a)
                                              : (a == 0.0 && acc == 0.0 ? (std::signbit(acc) ? acc : a) : std::min(acc, a));
                    else if (fn == QE_AGG_MAX) acc = (a != a || acc != acc) ? (acc != acc ? acc : a)
                                                   : (a == 0.0 && acc == 0.0 ? (std::signbit(acc) ? a : acc) : std::max(acc, a));
                    else acc += a;
                }
            if (fn == QE_AGG_COUNT) {           // Accumulators.kt:26-36
                out_values[i] = cnt;
                out_valid[i] = 1;
            } else if (cnt == 0) {              // :47-53 empty => null
                out_values[i] = 0.0;
                out_valid[i] = 0;
            } else {
                out_values[i] = fn == QE_AGG_AVG ? acc / cnt : acc;   // :101-107
                out_valid[i] = 1;
            }
        }
        if (ntiles > 0)
            for (int b = 0; b < grid; b++) nsel += (int64_t)partial[(size_t)b * stride + 2 * nagg];
        if (out_selected_rows) *out_selected_rows = nsel;
    });
}

int32_t qe_filter_groupby_prepare(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter, const qe_expr *const *keys,
                                  int32_t nkeys, const qe_expr *const *exprs, const int32_t *agg_fns, int32_t nagg) {
    if (!ctx || !batch || nkeys <= 0 || !keys || nagg <= 0 || !exprs || !agg_fns) return QE_ERR_INVALID_ARG;
    return guarded(ctx, [&] {
        if (ctx->device >= 0) need_device(ctx);
        auto plan = get_plan(ctx, batch, filter, exprs, nagg, agg_fns, ctx->device >= 0, keys, nkeys);
        if (plan->cg.hashed && (ctx->opts.tuning[5] & 8388608) != 0 && nkeys <= 4 && nagg <= 8)   // forced hash-partitioned form: its plan too
            (void)get_plan(ctx, batch, filter, exprs, nagg, agg_fns, ctx->device >= 0, keys, nkeys, false, false, nullptr, 64, 8);
    });
}

int32_t qe_filter_groupby(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter, const qe_expr *const *keys, int32_t nkeys,
                          const qe_expr *const *exprs, const int32_t *agg_fns, int32_t nagg, qe_result **out) {
    if (!ctx || !batch || !out || nkeys <= 0 || !keys || nagg <= 0 || !exprs || !agg_fns) return QE_ERR_INVALID_ARG;
    *out = nullptr;
    return guarded(ctx, [&] {
        need_device(ctx);
        if (batch->schema_only) fail(QE_ERR_INVALID_ARG, "schema-only batch (qe_batch_describe) cannot be executed");
        auto plan = get_plan(ctx, batch, filter, exprs, nagg, agg_fns, true, keys, nkeys);
        const CodegenOutput &cg = plan->cg;
        if (cg.hashed) {
            // many distinct keys (known from an earlier execution of this plan): the hash-partitioned form -- every pass
            // streams -- instead of the id build + dense passes (100 000 DOUBLE keys, 1 B rows: 24 ms that way)
            // measured, SELECT k, MIN(v), MAX(v) over 1 B rows, id build + dense passes against this form: 30 000 keys 21 / 14.7 ms,
            // 100 000 keys 24.0 / 15.3, 300 000 keys 32.1 / 17.8, 1 000 000 keys 57 / 24.4 ; below 25 000 keys, this form / ids:
            // 20 000 keys 14.9 / 20.2, 10 000 keys 15.9 / 16.9, 5000 keys 16.3 / 16.8, 3000 keys 17.2 / 18.3 (fewer fit the LDS-privatised table)
            static const int64_t hp_from = std::getenv("QE_HP_FROM") ? std::atoll(std::getenv("QE_HP_FROM")) : 4000;
            const bool hp_forced = (ctx->opts.tuning[5] & 8388608) != 0, hp_never = (ctx->opts.tuning[5] & 16777216) != 0;
            const int64_t n = batch->nrows;
            const bool hp_possible = !hp_never && !plan->hp_failed && n > 0 && n < (1ll << 32) && nkeys <= 4 && nagg <= 8;
            static const int env_parts = std::getenv("QE_HP_PARTS") ? std::atoi(std::getenv("QE_HP_PARTS")) : 0;
            static const int env_shift = std::getenv("QE_HP_SHIFT") ? std::atoi(std::getenv("QE_HP_SHIFT")) : 0;
            static const int env_fill = std::getenv("QE_HP_FILL") ? std::atoi(std::getenv("QE_HP_FILL")) : 16;
            // the widest table a partition may have: entry = {first row, key words.., the counters and accumulators the plan needs}:
            // {first row, key, MIN, MAX} and {first row, key, count, SUM} are 32 bytes: 4096 buckets in 128 KiB
            int max_shift = 12;
            {
                bool keys_nullable = false;
                std::vector<char> agg_nullable;
                for (int i = 0; i < nagg; i++) agg_nullable.push_back(cg.outs[(size_t)i].nullable ? 1 : 0);
                for (const OutSpec &ks : cg.keys) keys_nullable = keys_nullable || ks.nullable;
                const int64_t entry_bytes = 8 * (1 + nkeys + (keys_nullable ? 1 : 0) + hp_entry_layout(agg_nullable, agg_fns, nagg).words);
                while (max_shift > 8 && (entry_bytes << max_shift) > 144 * 1024) max_shift--;
            }
            // one attempt with P partitions of 2^shift buckets: the result, or nullptr when some partition's table filled up (or the form
            // is not to be had: hp_failed)
            auto try_hp = [&](int P, int shift) -> qe_result * {
                std::shared_ptr<Plan> hplan;
                try {
                    hplan = get_plan(ctx, batch, filter, exprs, nagg, agg_fns, true, keys, nkeys, false, false, nullptr, P, shift);
                } catch (const Error &) {   // e.g. an entry too wide for the LDS table: the other forms stay
                    plan->hp_failed = true;
                    return nullptr;
                }
                if (!hplan || !hplan->cg.hp) return nullptr;
                qe_result *r = nullptr;
                bool skewed = false;
                try {
                    r = run_groupby_hp(ctx, batch, hplan, agg_fns, nagg, &skewed);
                } catch (const Error &e) {
                    // its record array (21 - 64 bytes per kept row) did not fit beside the batch: the dense-id path needs 4 + 16
                    // bytes per row -- this plan stays with that one (run_groupby_hp releases what it had allocated)
                    if (e.code != QE_ERR_OOM) throw;
                    plan->hp_failed = true;
                }
                if (skewed) plan->hp_failed = true;   // a property of the data: this plan stays with the forms that slice their work
                if (r) {
                    plan->known_keys = r->count;
                    ctx->last_form = QE_FORM_GROUPBY_HASH_PARTITIONED;
                }
                return r;
            };
            if (hp_possible && (hp_forced || (plan->known_keys >= hp_from && n >= (4ll << 20)))) {
                // ONE workgroup aggregates a partition (128 partitions left half the chip idle: 21.8 ms for the aggregation of 1 B
                // records).  Fewer partitions make longer runs per scatter tile -- less padding to whole lines --, more partitions keep
                // the tables sparse.  Buckets for ~16x the keys seen, 256 .. 4096 per partition: a wave leaves the probe loop after its LONGEST
                // probe sequence, so the tables are as sparse as the LDS allows (100 000 keys, 256 partitions x 1024 / 2048 / 4096 buckets:
                // 23.4 / 17 / 15.5 ms)
                const int64_t keys_seen = std::max<int64_t>(plan->known_keys, 1);
                // few partitions = long runs per scatter tile = little padding, and the probe loop tolerates full tables better than the
                // scatter tolerates short runs (1 M keys, 1 B rows: 512 partitions half full 24.6 ms, 1024 partitions a quarter full
                // 37.1 ms; 500 000 keys: 256 partitions half full 22.6 ms, 512 a quarter full 18.3 ms): 256 partitions up to 30 %, 512
                // up to 50 %, 1024 beyond
                int P = keys_seen * 10 <= ((int64_t)256 << max_shift) * 3 ? 256 : keys_seen * 2 <= ((int64_t)512 << max_shift) ? 512 : 1024;
                if (hp_forced && plan->known_keys <= 0) P = 64;
                if (env_parts >= 2) P = env_parts;
                int shift = 8;
                while (shift < max_shift && ((int64_t)P << shift) < keys_seen * env_fill) shift++;
                if (env_shift >= 6) shift = env_shift;
                if (qe_result *r = try_hp(P, shift)) {
                    *out = r;
                    return;
                }
                // some partition's table filled up: with 1024 partitions there is nothing larger to try; otherwise the run below
                // reports how many keys there are and the next execution sizes its partitions from that
                if (P >= 1024) plan->hp_failed = true;
            }
            // The FIRST execution of a plan does not know its keys.  The LDS tables and the id build find out cheaply that there are
            // many (their launches stop at once when a table fills up: more than 32 768 keys fill the id build's first table); from
            // there the hash-partitioned form takes over with the widest tables -- 256, then 512, then 1024 partitions when a table
            // overflows -- instead of growing the id table (100 000 keys, 1 B rows: a first execution of 48 ms that way).
            bool many_keys = false;
            const bool first_hp = hp_possible && !hp_forced && plan->known_keys < 0 && plan->id_capacity <= 0 && n >= (4ll << 20);
            *out = run_groupby_hashed(ctx, batch, *plan, filter, exprs, agg_fns, nagg, first_hp ? &many_keys : nullptr);
            if (!*out && many_keys) {
                for (int P = 256; P <= 1024 && !*out && !plan->hp_failed; P *= 2) *out = try_hp(P, max_shift);
                if (*out) return;
                *out = run_groupby_hashed(ctx, batch, *plan, filter, exprs, agg_fns, nagg, nullptr);   // more keys than 1024 tables hold
            }
            if (*out) plan->known_keys = (*out)->count;
            ctx->last_form = QE_FORM_GROUPBY_HASHED;
            return;
        }
        *out = run_groupby_dense(ctx, batch, plan, agg_fns, nagg);
        ctx->last_form = QE_FORM_GROUPBY_DENSE;
    });
}

// ---- results ----------------------------------------------------------------------------------------
int64_t qe_result_count(const qe_result *r) { return r ? r->count : -1; }
int32_t qe_result_ncols(const qe_result *r) { return r ? (int32_t)r->cols.size() : -1; }

int32_t qe_result_column(const qe_result *r, int32_t col, qe_col_view *out) {
    if (!r || !out || col < 0 || col >= (int32_t)r->cols.size()) return QE_ERR_INVALID_ARG;
    const OutColumn &c = r->cols[col];
    out->type = c.type;
    out->nullable = c.nullable ? 1 : 0;
    out->data = c.data;
    out->validity = c.validity;
    out->count = r->count;
    out->dict = c.dict ? &c.dict_handle : nullptr;
    return QE_OK;
}

// Device -> a caller's PAGEABLE buffer.  One hipMemcpy into pageable memory ran at 7 GB/s here (0.8 GB in 108 ms) while the
// link does ~55 GB/s into pinned memory: the bytes go through two pinned staging chunks on the copy stream, and the chunk
// that has arrived is copied into the caller's buffer (by a few host threads: one memcpy thread does ~10 GB/s) while the
// next one is on the link.
static void parallel_memcpy(void *dst, const void *src, size_t n) {
    const size_t kMin = 4u << 20;
    unsigned nthr = (unsigned)std::min<size_t>(4, n / kMin);
    if (nthr <= 1) {
        std::memcpy(dst, src, n);
        return;
    }
    std::vector<std::thread> ts;
    const size_t per = ((n / nthr) + 4095) & ~(size_t)4095;
    for (unsigned t = 1; t < nthr; t++) {
        const size_t off = (size_t)t * per;
        if (off >= n) break;
        const size_t len = std::min(per, n - off);
        ts.emplace_back([=] { std::memcpy((char *)dst + off, (const char *)src + off, len); });
    }
    std::memcpy(dst, src, std::min(per, n));
    for (auto &t : ts) t.join();
}

static void staged_d2h(qe_ctx *ctx, void *dst, const void *src, size_t n) {
    const size_t kChunk = 32u << 20;
    if (n <= (1u << 20)) {   // small: not worth the staging
        QE_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, ctx->copy_stream));
        QE_HIP(hipStreamSynchronize(ctx->copy_stream));
        return;
    }
    void *stage[2] = {ctx->pinned.alloc(std::min(n, kChunk)), ctx->pinned.alloc(std::min(n, kChunk))};
    struct G { qe_ctx *c; void **s; ~G() { c->pinned.release(s[0]); c->pinned.release(s[1]); } } g{ctx, stage};
    hipEvent_t ev[2];
    QE_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    QE_HIP(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    struct EG { hipEvent_t *e; ~EG() { (void)hipEventDestroy(e[0]); (void)hipEventDestroy(e[1]); } } eg{ev};
    const size_t nchunks = (n + kChunk - 1) / kChunk;
    auto issue = [&](size_t k) {
        const size_t off = k * kChunk, len = std::min(kChunk, n - off);
        QE_HIP(hipMemcpyAsync(stage[k & 1], (const char *)src + off, len, hipMemcpyDeviceToHost, ctx->copy_stream));
        QE_HIP(hipEventRecord(ev[k & 1], ctx->copy_stream));
    };
    issue(0);
    for (size_t k = 0; k < nchunks; k++) {
        if (k + 1 < nchunks) issue(k + 1);
        QE_HIP(hipEventSynchronize(ev[k & 1]));
        const size_t off = k * kChunk, len = std::min(kChunk, n - off);
        parallel_memcpy((char *)dst + off, stage[k & 1], len);
    }
}

int32_t qe_result_column_to_host(qe_ctx *ctx, const qe_result *r, int32_t col, void *data_out, uint64_t *validity_out) {
    if (!ctx || !r || col < 0 || col >= (int32_t)r->cols.size()) return QE_ERR_INVALID_ARG;
    return guarded(ctx, [&] {
        need_device(ctx);
        const OutColumn &c = r->cols[col];
        if (r->count == 0) return;
        if (data_out) staged_d2h(ctx, data_out, c.data, column_bytes(c.type, r->count));
        if (validity_out) {
            if (c.validity) staged_d2h(ctx, validity_out, c.validity, bitmap_bytes(r->count));
            else std::memset(validity_out, 0xff, bitmap_bytes(r->count));
        }
    });
}

// Result -> PINNED host memory owned by the library, on the context's copy stream: the call returns at once, so the scan of
// the next batch (compute stream) runs beside the copy; qe_host_result_wait blocks until the bytes are there.  What a host
// that materialises rows (Main.kt:18 `physicalPlan.map { it }`, Operators.kt:5-11) reads them from -- no second copy.
int32_t qe_result_to_host(qe_ctx *ctx, const qe_result *r, qe_host_result **out) {
    if (!ctx || !r || !out) return QE_ERR_INVALID_ARG;
    *out = nullptr;
    return guarded(ctx, [&] {
        need_device(ctx);
        std::unique_ptr<qe_host_result> h(new qe_host_result());
        h->count = r->count;
        try {
            for (const OutColumn &c : r->cols) {
                qe_host_result::Col hc;
                hc.type = c.type;
                hc.nullable = c.validity != nullptr;
                hc.dict = c.dict;
                hc.dict_handle.d = c.dict;
                h->cols.push_back(hc);
                qe_host_result::Col &d = h->cols.back();
                d.data = ctx->pinned.alloc(std::max<size_t>(column_bytes(c.type, r->count), 64));
                if (c.validity) d.validity = (uint64_t *)ctx->pinned.alloc(std::max<size_t>(bitmap_bytes(r->count), 64));
            }
            QE_HIP(hipEventCreateWithFlags(&h->done, hipEventDisableTiming));
            if (r->count > 0) {
                for (size_t i = 0; i < r->cols.size(); i++) {
                    const OutColumn &c = r->cols[i];
                    QE_HIP(hipMemcpyAsync(h->cols[i].data, c.data, column_bytes(c.type, r->count), hipMemcpyDeviceToHost, ctx->copy_stream));
                    if (c.validity)
                        QE_HIP(hipMemcpyAsync(h->cols[i].validity, c.validity, bitmap_bytes(r->count), hipMemcpyDeviceToHost, ctx->copy_stream));
                }
            }
            QE_HIP(hipEventRecord(h->done, ctx->copy_stream));
        } catch (...) {
            (void)hipStreamSynchronize(ctx->copy_stream);
            for (auto &c : h->cols) {
                ctx->pinned.release(c.data);
                ctx->pinned.release(c.validity);
            }
            if (h->done) (void)hipEventDestroy(h->done);
            throw;
        }
        h->src = r;
        ctx->host_results.push_back(h.get());
        *out = h.release();
    });
}

int32_t qe_host_result_wait(qe_ctx *ctx, qe_host_result *h) {
    if (!ctx || !h) return QE_ERR_INVALID_ARG;
    return guarded(ctx, [&] {
        if (h->waited) return;
        QE_HIP(hipEventSynchronize(h->done));
        h->waited = true;
        h->src = nullptr;
    });
}

int64_t qe_host_result_count(const qe_host_result *h) { return h ? h->count : -1; }
int32_t qe_host_result_ncols(const qe_host_result *h) { return h ? (int32_t)h->cols.size() : -1; }

int32_t qe_host_result_column(const qe_host_result *h, int32_t col, qe_col_view *out) {
    if (!h || !out || col < 0 || col >= (int32_t)h->cols.size()) return QE_ERR_INVALID_ARG;
    const qe_host_result::Col &c = h->cols[col];
    out->type = c.type;
    out->nullable = c.nullable ? 1 : 0;
    out->data = c.data;
    out->validity = c.validity;
    out->count = h->count;
    out->dict = c.type == QE_STRING ? &c.dict_handle : nullptr;
    return QE_OK;
}

void qe_host_result_free(qe_ctx *ctx, qe_host_result *h) {
    if (!ctx || !h) return;
    if (!h->waited && h->done) (void)hipEventSynchronize(h->done);   // the copies write into the buffers released below
    for (auto &c : h->cols) {
        ctx->pinned.release(c.data);
        ctx->pinned.release(c.validity);
    }
    if (h->done) (void)hipEventDestroy(h->done);
    auto &v = ctx->host_results;
    v.erase(std::remove(v.begin(), v.end(), h), v.end());
    delete h;
}

void qe_result_free(qe_ctx *ctx, qe_result *r) {
    if (!ctx) return;
    // a copy to the host that still reads this result must finish before its buffers go back to the pool
    for (qe_host_result *h : ctx->host_results)
        if (h->src == r) {
            if (h->done) (void)hipEventSynchronize(h->done);
            h->waited = true;
            h->src = nullptr;
        }
    free_result(ctx, r);
}

int32_t qe_stream_read_bandwidth(qe_ctx *ctx, int64_t nbytes, int32_t reps, double *out_gbps) {
    if (!ctx || nbytes < 4096 || reps < 1 || !out_gbps) return QE_ERR_INVALID_ARG;
    return guarded(ctx, [&] {
        need_device(ctx);
        void *buf = ctx->pool.alloc((size_t)nbytes);
        struct G { qe_ctx *c; void *p; ~G() { c->pool.release(p); } } g{ctx, buf};
        QE_HIP(hipMemsetAsync(buf, 0x5a, (size_t)nbytes, ctx->stream));
        launch_stream_read(ctx->stream, buf, nbytes, (unsigned long long *)(ctx->d_ctrl + 8));
        QE_HIP(hipStreamSynchronize(ctx->stream));
        double best = 1e30;
        // the achievable read rate depends on how many waves stream: 8 per CU (2 workgroups) reach ~7 TB/s where 32 reach ~6.3
        // (tools/copy_calib.hip) -- the calibration reports the best of 2 / 4 / 8 workgroups per CU
        for (int wgs : {2, 4, 8}) {
            for (int r = 0; r < reps; r++) {
                QE_HIP(hipEventRecord(ctx->ev0, ctx->stream));
                launch_stream_read(ctx->stream, buf, nbytes, (unsigned long long *)(ctx->d_ctrl + 8), wgs);
                QE_HIP(hipEventRecord(ctx->ev1, ctx->stream));
                QE_HIP(hipStreamSynchronize(ctx->stream));
                float ms = 0.f;
                QE_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
                best = std::min(best, (double)ms);
            }
        }
        *out_gbps = (double)(nbytes / 16 * 16) / (best * 1e-3) / 1e9;
    });
}

}  // extern "C"

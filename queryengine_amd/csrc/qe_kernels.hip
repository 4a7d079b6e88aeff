// qe_kernels.hip -- precompiled gfx950 kernels that are not plan-specific:
// the synthetic column generator, byte->bitmap packing of nullable/boolean
// outputs, and the streaming-read calibration kernel used for the roofline.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "qe_kernels.h"

namespace qe {

typedef unsigned long long u64;
typedef long long i64;
typedef unsigned int u32;

__device__ __forceinline__ u64 mix64(u64 z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
#define QE_GOLDEN 0x9E3779B97F4A7C15ull
__device__ __forceinline__ u64 gen_raw(u64 seed, int col_id, u64 row) {
    return mix64(QE_GOLDEN * (u64)(col_id + 1) + row * QE_GOLDEN + seed);
}

struct GenArgs {
    int kind, col_id, aux_col_id, null_pct;
    u64 modulus, seed;
    i64 offset, row_begin, nrows;
    double step;
    void *data;
    u64 *validity;
};

// One thread per row, a wave per 64 rows: the validity word of a wave is one __ballot.
__global__ void __launch_bounds__(256) generate_kernel(const GenArgs a) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    const i64 padded = (a.nrows + 63) & ~63ll;
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < padded; k += stride) {
        const bool in = k < a.nrows;
        const u64 i = (u64)(a.row_begin + k);
        const u64 x = gen_raw(a.seed, a.col_id, i);
        if (in) {
            switch (a.kind) {
            case QE_GEN_I64_MOD: ((i64 *)a.data)[k] = (i64)(x % a.modulus) + a.offset; break;
            case QE_GEN_I64_ROWID: ((i64 *)a.data)[k] = (i64)i; break;
            case QE_GEN_I32_MOD:
            case QE_GEN_DICT_MOD: ((int *)a.data)[k] = (int)((i64)(x % a.modulus) + a.offset); break;
            case QE_GEN_F64_UNIT: ((double *)a.data)[k] = (double)(x >> 11) * 0x1.0p-53; break;
            case QE_GEN_F64_MOD: ((double *)a.data)[k] = (double)((i64)(x % a.modulus) + a.offset); break;
            case QE_GEN_F64_STEP: ((double *)a.data)[k] = (double)((i64)(x % a.modulus) + a.offset) * a.step; break;
            case QE_GEN_F64_PRICE: {
                const u64 q = gen_raw(a.seed, a.aux_col_id, i) % 50 + 1;
                const u64 cents = 90000 + x % 120000;
                ((double *)a.data)[k] = (double)(i64)(q * cents) / 100.0;
                break;
            }
            default: break;
            }
        }
        if (a.validity) {
            const bool valid = in && (mix64(x + QE_GOLDEN) % 100 >= (u64)a.null_pct);
            const u64 w = __ballot(valid);
            if ((threadIdx.x & 63) == 0) a.validity[k >> 6] = w;
        }
    }
}

void launch_generate(hipStream_t s, const qe_gen_spec &spec, uint64_t seed, int64_t row_begin, int64_t nrows,
                     void *data, uint64_t *validity) {
    if (nrows <= 0) return;
    GenArgs a;
    a.kind = spec.kind; a.col_id = spec.col_id; a.aux_col_id = spec.aux_col_id; a.null_pct = spec.null_pct;
    a.modulus = spec.modulus ? spec.modulus : 1; a.seed = seed; a.offset = spec.offset; a.row_begin = row_begin;
    a.nrows = nrows; a.step = spec.step; a.data = data; a.validity = (u64 *)validity;
    const int64_t blocks = (nrows + 255) / 256;
    const int grid = (int)(blocks < 16384 ? blocks : 16384);
    hipLaunchKernelGGL(generate_kernel, dim3(grid), dim3(256), 0, s, a);
}

__global__ void __launch_bounds__(256) pack_bytes_kernel(const unsigned char *bytes, i64 n, u64 *words) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    const i64 padded = (n + 63) & ~63ll;
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < padded; k += stride) {
        const bool b = k < n && bytes[k] != 0;
        const u64 w = __ballot(b);
        if ((threadIdx.x & 63) == 0) words[k >> 6] = w;
    }
}

void launch_pack_bytes(hipStream_t s, const uint8_t *bytes, int64_t n, uint64_t *words) {
    if (n <= 0) return;
    const int64_t blocks = (n + 255) / 256;
    const int grid = (int)(blocks < 8192 ? blocks : 8192);
    hipLaunchKernelGGL(pack_bytes_kernel, dim3(grid), dim3(256), 0, s, bytes, (i64)n, (u64 *)words);
}

typedef u64 u64x2 __attribute__((ext_vector_type(2)));

// Read-only stream: a wave reads 4 KiB contiguous per step (4 loads of 16 B per lane in flight), waves stride over the buffer
// (tools/copy_calib.hip measured this shape at 7.0 - 7.4 TB/s with 8 waves per CU; 8 strided loads per lane reached 6.3 - 6.5).
__global__ void __launch_bounds__(256) stream_read_kernel(const u64x2 *src, i64 nvec, u64 *sink) {
    const int lane = threadIdx.x & 63;
    const i64 wave = (i64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = (i64)gridDim.x * (blockDim.x >> 6);
    const i64 nsteps = nvec / 256;
    u64 acc = 0;
    for (i64 it = wave; it < nsteps; it += nwaves) {
        u64x2 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __builtin_nontemporal_load(src + (it * 4 + j) * 64 + lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc ^= v[j].x ^ v[j].y;
    }
    for (i64 i = nsteps * 256 + wave * 64 + lane; i < nvec; i += nwaves * 64) {
        const u64x2 v = __builtin_nontemporal_load(src + i);
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x0123456789abcdefull) sink[0] = acc;   // practically never: keeps the loads alive
}

// Calibration of what a TRICKLE of writes costs a read stream: the same read loop, but every
// `write_every`-th iteration each wave also stores one 512-byte block (8 B per lane) to a private,
// sequential position of `dst`.
__global__ void __launch_bounds__(256) stream_read_write_kernel(const u64x2 *src, i64 nvec, u64 *sink, u64 *dst, i64 dst_words_per_wave,
                                                                int write_every, int window_period, int window_len, int blocks_per_event, int store_kind) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 wave = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    u64 *out = dst + wave * dst_words_per_wave;
    i64 wpos = 0;
    u64 acc = 0;
    int it = 0, pending = 0;
    for (; i + 7 * stride < nvec; i += 8 * stride, ++it) {
        u64x2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(src + i + j * stride);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j].x ^ v[j].y;
        if (write_every > 0 && it % (write_every * blocks_per_event) == 0) pending += blocks_per_event;   // same bytes, longer runs
        // window_period == 0: write at once (a trickle).  Otherwise every wave of the device holds its blocks
        // until the shared 100 MHz clock says the write window is open, so that writes reach HBM in bursts.
        bool open = true;
        if (window_period > 0) open = (__builtin_amdgcn_s_memrealtime() % (u64)window_period) < (u64)window_len;
        if (pending > 0 && open) {
            for (; pending > 0 && wpos + 64 <= dst_words_per_wave; --pending) {
                if (store_kind == 1) __builtin_nontemporal_store(acc + pending, out + wpos + lane);
                else if (store_kind == 2) __hip_atomic_store(out + wpos + lane, acc + pending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (store_kind == 3) __hip_atomic_store(out + wpos + lane, acc + pending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                else out[wpos + lane] = acc + pending;
                wpos += 64;
            }
        }
    }
    for (; pending > 0 && wpos + 64 <= dst_words_per_wave; --pending) {
        out[wpos + lane] = acc + pending;
        wpos += 64;
    }
    if (acc == 0x0123456789abcdefull) sink[0] = acc;
}

void launch_stream_read_write(hipStream_t s, const void *src, int64_t nbytes, unsigned long long *sink, void *dst,
                              int64_t dst_bytes, int write_every, int window_period, int window_len, int blocks_per_event) {
    const int store_kind = getenv("QE_CALIB_STORE") ? atoi(getenv("QE_CALIB_STORE")) : 0;
    const i64 nvec = nbytes / 16;
    if (nvec <= 0) return;
    const int grid = 256 * 8;
    const i64 waves = (i64)grid * 4;
    hipLaunchKernelGGL(stream_read_write_kernel, dim3(grid), dim3(256), 0, s, (const u64x2 *)src, nvec, (u64 *)sink, (u64 *)dst,
                       (i64)(dst_bytes / 8 / waves), write_every, window_period, window_len, blocks_per_event < 1 ? 1 : blocks_per_event, store_kind);
}

// Calibration of SPARSE streaming reads (late materialisation): the same loop, but a lane loads its 16 bytes only when a
// hash of the vector index falls below `pct` -- what fraction of the dense time does a `pct` % selection cost?
template <int KIND>
__device__ __forceinline__ u64x2 sparse_load(const u64x2 *p) {
    if (KIND == 1) return *p;
    if (KIND == 6) {   // 8-byte loads (what a 4-byte column's 2 rows per lane use): HALF the bytes of the 16-byte form
        u64x2 v;
        v.x = __builtin_nontemporal_load((const u64 *)p);
        v.y = 0;
        return v;
    }
    if (KIND >= 2) {   // cache-policy variants through the ISA bits: 2 = sc0 sc1, 3 = sc1, 4 = sc0, 5 = sc0 sc1 nt
        u64x2 v;
        if (KIND == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
        else if (KIND == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
        else if (KIND == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
    return __builtin_nontemporal_load(p);
}

template <int KIND>
__global__ void __launch_bounds__(256) stream_read_sparse_kernel(const u64x2 *src, i64 nvec, u64 *sink, unsigned pct) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 acc = 0;
    for (; i + 7 * stride < nvec; i += 8 * stride) {
        u64x2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u64 idx = (u64)(i + j * stride);
            u64 h = idx * 0x9E3779B97F4A7C15ull;
            h ^= h >> 29;
            h *= 0xBF58476D1CE4E5B9ull;
            h ^= h >> 32;
            v[j].x = 0; v[j].y = 0;
            if ((unsigned)(h % 100u) < pct) v[j] = sparse_load<KIND>(KIND == 6 ? (const u64x2 *)((const u64 *)src + idx) : src + idx);   // 6: contiguous 8 B per lane
        }
        if (KIND >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the asm loads are invisible to the compiler's counters (timing only)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j].x ^ v[j].y;
    }
    if (acc == 0x0123456789abcdefull) sink[0] = acc;
}

void launch_stream_read(hipStream_t s, const void *src, int64_t nbytes, unsigned long long *sink, int wgs_per_cu) {
    const i64 nvec = nbytes / 16;
    if (nvec <= 0) return;
    static const int sparse_pct = std::getenv("QE_CALIB_SPARSE_PCT") ? std::atoi(std::getenv("QE_CALIB_SPARSE_PCT")) : -1;
    static const int sparse_kind = std::getenv("QE_CALIB_SPARSE_KIND") ? std::atoi(std::getenv("QE_CALIB_SPARSE_KIND")) : 0;
    if (sparse_pct >= 0) {
        const dim3 g(256 * 8), b(256);
        const u64x2 *sp = (const u64x2 *)src;
        switch (sparse_kind) {
        case 1: hipLaunchKernelGGL(stream_read_sparse_kernel<1>, g, b, 0, s, sp, nvec, (u64 *)sink, (unsigned)sparse_pct); break;
        case 2: hipLaunchKernelGGL(stream_read_sparse_kernel<2>, g, b, 0, s, sp, nvec, (u64 *)sink, (unsigned)sparse_pct); break;
        case 3: hipLaunchKernelGGL(stream_read_sparse_kernel<3>, g, b, 0, s, sp, nvec, (u64 *)sink, (unsigned)sparse_pct); break;
        case 4: hipLaunchKernelGGL(stream_read_sparse_kernel<4>, g, b, 0, s, sp, nvec, (u64 *)sink, (unsigned)sparse_pct); break;
        case 5: hipLaunchKernelGGL(stream_read_sparse_kernel<5>, g, b, 0, s, sp, nvec, (u64 *)sink, (unsigned)sparse_pct); break;
        case 6: hipLaunchKernelGGL(stream_read_sparse_kernel<6>, g, b, 0, s, sp, nvec, (u64 *)sink, (unsigned)sparse_pct); break;
        default: hipLaunchKernelGGL(stream_read_sparse_kernel<0>, g, b, 0, s, sp, nvec, (u64 *)sink, (unsigned)sparse_pct); break;
        }
    }
    else
        hipLaunchKernelGGL(stream_read_kernel, dim3(256 * (wgs_per_cu > 0 ? wgs_per_cu : 8)), dim3(256), 0, s, (const u64x2 *)src, nvec, (u64 *)sink);
}

// ---- partitioned group-by -------------------------------------------------------------------------------------------
// One workgroup per partition walks the chunks in order: exclusive scan of counts[chunk][part] along the chunk axis.
__global__ void __launch_bounds__(256) gb_scan_kernel(u32 *counts, i64 nchunks, int nparts, u64 *totals) {
    __shared__ u32 s_wave[4];
    const int part = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u64 running = 0;
    for (i64 c0 = 0; c0 < nchunks; c0 += 256) {
        const i64 c = c0 + threadIdx.x;
        const u32 v = c < nchunks ? counts[c * nparts + part] : 0u;
        u32 incl = v;   // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        u32 before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) before += s_wave[w];
            total += s_wave[w];
        }
        if (c < nchunks) counts[c * nparts + part] = (u32)(running + before + incl - v);
        running += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[part] = running;
}

void launch_gb_scan(hipStream_t s, uint32_t *counts, int64_t nchunks, int nparts, unsigned long long *totals) {
    if (nparts <= 0) return;
    hipLaunchKernelGGL(gb_scan_kernel, dim3((unsigned)nparts), dim3(256), 0, s, counts, (i64)nchunks, nparts, (u64 *)totals);
}

// ---- bitmap segments (result concatenation / gather) -----------------------------------------------------------------
// Place `nbits` bits of `src` (bit i = word i>>6, bit i&63) at bit offset `dst_off` of `dst`.  One thread owns one
// destination word, so the read-modify-write of the two boundary words is race free inside a launch; segments are placed
// by launches serialised on one stream.  Words: 64 rows each, funnel-shifted -- never expanded to a byte per row.
__global__ void __launch_bounds__(256) bitmap_place_kernel(u64 *dst, i64 dst_off, const u64 *src, i64 nbits) {
    const i64 w0 = dst_off >> 6, w1 = (dst_off + nbits - 1) >> 6;
    const i64 nsrc = (nbits + 63) >> 6;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 w = w0 + (i64)blockIdx.x * blockDim.x + threadIdx.x; w <= w1; w += stride) {
        const i64 s = w * 64 - dst_off;          // source bit index of this word's bit 0 (negative in the first word)
        u64 val;
        if (s >= 0) {
            const i64 k = s >> 6;
            const u32 sh = (u32)(s & 63);
            val = src[k] >> sh;
            if (sh != 0 && k + 1 < nsrc) val |= src[k + 1] << (64u - sh);
        } else {
            val = src[0] << (u32)(-s);
        }
        const i64 lo = s < 0 ? -s : 0;                                   // first bit of this word that belongs to the segment
        const i64 hi = nbits - s < 64 ? nbits - s : 64;                  // one past its last bit
        const u64 mask = (hi >= 64 ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
        dst[w] = (dst[w] & ~mask) | (val & mask);
    }
}

void launch_bitmap_place(hipStream_t s, uint64_t *dst, int64_t dst_bit_offset, const uint64_t *src, int64_t nbits) {
    if (nbits <= 0) return;
    const int64_t words = ((dst_bit_offset + nbits - 1) >> 6) - (dst_bit_offset >> 6) + 1;
    const int64_t blocks = (words + 255) / 256;
    hipLaunchKernelGGL(bitmap_place_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, (u64 *)dst,
                       (i64)dst_bit_offset, (const u64 *)src, (i64)nbits);
}

// ---- hashed group-by: table initialisation and collection of the used entries ---------------------------------------------
__global__ void __launch_bounds__(256) ht_init_kernel(u64 *tab, i64 nentries, HtInit init) {
    const i64 total = nentries * init.words;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) tab[i] = init.word[i % init.words];
}
void launch_ht_init(hipStream_t s, unsigned long long *tab, int64_t nentries, const HtInit &init) {
    if (nentries <= 0) return;
    const int64_t blocks = (nentries * init.words + 255) / 256;
    hipLaunchKernelGGL(ht_init_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, (u64 *)tab, (i64)nentries, init);
}
// every READY entry (state word == 2) is copied to dense[slot], slot drawn from *counter (order: irrelevant, the host sorts
// the groups by their smallest row id)
__global__ void __launch_bounds__(256) ht_collect_kernel(const u64 *tab, i64 nentries, int words, u64 *dense, unsigned int *counter) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < nentries; e += stride) {
        const u64 *src = tab + e * words;
        if (src[0] != 2ull) continue;
        const u32 slot = atomicAdd(counter, 1u);
        u64 *dst = dense + (i64)slot * words;
        for (int w = 0; w < words; ++w) dst[w] = src[w];
    }
}
void launch_ht_collect(hipStream_t s, const unsigned long long *tab, int64_t nentries, int words, unsigned long long *dense,
                       unsigned int *counter) {
    if (nentries <= 0) return;
    const int64_t blocks = (nentries + 255) / 256;
    hipLaunchKernelGGL(ht_collect_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, (const u64 *)tab, (i64)nentries,
                       words, (u64 *)dense, counter);
}

// ---- groups of a hashed GROUP BY finished on the device: insertion order = ascending first row (LinkedHashMap,
// GroupByAggregationOperator.kt:22), accumulators finished as Accumulators.kt:26-107 says -- what finish_hashed_groups does on
// the host, for results of many groups (1 M groups: ~70 ms of host work and two trips over the link otherwise) -----------------
__global__ void __launch_bounds__(256) group_sort_keys_kernel(const u64 *entries, int words, int first_row_word, i64 m, u64 *keys, u32 *rows) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) {
        keys[i] = entries[i * words + first_row_word];
        rows[i] = (u32)i;
    }
}
void launch_group_sort_keys(hipStream_t s, const unsigned long long *entries, int words, int first_row_word, int64_t m,
                            unsigned long long *keys, uint32_t *rows) {
    if (m <= 0) return;
    hipLaunchKernelGGL(group_sort_keys_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, (const u64 *)entries, words, first_row_word,
                       (i64)m, (u64 *)keys, (u32 *)rows);
}
// thread j finishes result row j (64 consecutive rows per wave: one bitmap word per ballot)
__global__ void __launch_bounds__(256) group_finish_kernel(const GroupFinishArgs a) {
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = j < a.m;
    const int lane = threadIdx.x & 63;
    const u64 *e = a.entries + (live ? (i64)a.rows[j] : 0) * a.words;
    const u64 knull = live ? e[1] : 0ull;
    for (int k = 0; k < a.nkeys; ++k) {
        const bool null = ((knull >> k) & 1ull) != 0ull;
        const u64 kw = live && !null ? e[2 + k] : 0ull;
        const int t = a.key_type[k];
        if (t == QE_DOUBLE || t == QE_INT64) { if (live) ((u64 *)a.key_data[k])[j] = kw; }
        else if (t == QE_BOOLEAN) {
            const u64 bits = __ballot(live && kw != 0ull);
            if (lane == 0 && live) ((u64 *)a.key_data[k])[j >> 6] = bits;
        } else { if (live) ((int *)a.key_data[k])[j] = (int)(i64)kw; }
        const u64 vb = __ballot(live && !null);
        if (lane == 0 && live) a.key_valid[k][j >> 6] = vb;
        if (live && null) a.flags[k] = 1u;
    }
    const u64 *acc = e + 2 + a.nkeys;   // {first row, (count, acc)..}
    for (int i = 0; i < a.nagg; ++i) {
        const u64 cnt = live ? acc[1 + 2 * a.cnt_src[i]] : 0ull;
        const u64 raw = live ? acc[2 + 2 * i] : 0ull;
        double v = 0.0;
        bool ok = true;
        switch (a.agg_fn[i]) {
        case QE_AGG_COUNT: v = (double)cnt; break;                                  // Accumulators.kt:26-36
        case QE_AGG_SUM: v = __builtin_bit_cast(double, raw); ok = cnt != 0ull; break;   // :47-53 empty => null
        case QE_AGG_AVG: v = __builtin_bit_cast(double, raw); ok = cnt != 0ull; if (ok) v /= (double)cnt; break;
        default: {                                                                          // MIN / MAX: undo the ordered key
            const i64 key = (i64)raw;
            v = __builtin_bit_cast(double, key ^ ((key >> 63) & 0x7fffffffffffffffll));
            ok = cnt != 0ull;
        }
        }
        if (!ok) v = 0.0;
        if (live) a.agg_data[i][j] = v;
        const u64 vb = __ballot(live && ok);
        if (lane == 0 && live) a.agg_valid[i][j >> 6] = vb;
        if (live && !ok) a.flags[4 + i] = 1u;
    }
}
void launch_group_finish(hipStream_t s, const GroupFinishArgs &a) {
    if (a.m <= 0) return;
    hipLaunchKernelGGL(group_finish_kernel, dim3((unsigned)((a.m + 255) / 256)), dim3(256), 0, s, a);
}

}  // namespace qe

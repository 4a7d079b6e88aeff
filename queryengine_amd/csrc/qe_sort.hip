// qe_sort.hip -- ORDER BY on the device ("last"): OrderByOperator.open (operator/OrderByOperator.kt:9-15)
// materialises its source and sorts the rows STABLY by one column with Kotlin's compareValues: null first, then
// Double.compareTo (-0.0 < 0.0, NaN greatest) / String.compareTo (UTF-16 code units) / Boolean.compareTo (false < true).
//
// Here the source's result already sits in HBM: (1) the key column becomes one u64 per row whose UNSIGNED order is that
// order (0 = null), (2) the (key, row id) pairs go through a stable LSD radix sort, 4 bits per pass, passes whose digit
// is the same for every key skipped, (3) every column of the result is gathered through the sorted row ids.  A
// different roofline from the scan (16 read-write passes over 12-byte pairs in the worst case), and not part of any
// BASELINE configuration.
#include <hip/hip_runtime.h>

#include "qe_kernels.h"

namespace qe {

typedef unsigned long long u64;
typedef long long i64;
typedef unsigned int u32;

// ---- sort keys -----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool bit_at(const u64 *bm, i64 i) { return (bm[i >> 6] >> (i & 63)) & 1ull; }

__global__ void __launch_bounds__(256) sort_key_kernel(const SortKeyArgs a) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        u64 k = 0;   // NULL: before every value (compareValues)
        if (!a.validity || bit_at(a.validity, i)) {
            switch (a.type) {
            case QE_DOUBLE: {   // Double.compareTo: IEEE order with -0.0 < 0.0, every NaN equal and greatest
                const double d = ((const double *)a.data)[i];
                i64 b = d != d ? 0x7ff8000000000000ll : __builtin_bit_cast(i64, d);
                b ^= (b >> 63) & 0x7fffffffffffffffll;            // negative values: reverse their order
                k = ((u64)b ^ 0x8000000000000000ull);             // signed -> unsigned order
                break;
            }
            case QE_INT64: k = (u64)((const i64 *)a.data)[i] ^ 0x8000000000000000ull; break;
            case QE_INT32: k = (u64)(i64)((const int *)a.data)[i] ^ 0x8000000000000000ull; break;
            case QE_STRING: {   // rank of the code in the dictionary's String.compareTo order (table from the host)
                const int c = ((const int *)a.data)[i];
                k = (u64)((u32)c < (u32)a.nranks ? a.ranks[c] : 0);
                break;
            }
            default: k = bit_at((const u64 *)a.data, i) ? 1ull : 0ull; break;   // BOOLEAN bitmap
            }
        }   // (NULL rows keep key 0 and are moved in front by one last pass on the validity bit: launch_radix_pass with shift 64)
        a.keys[i] = k;
        a.rows[i] = (u32)i;
    }
}
void launch_sort_keys(hipStream_t s, const SortKeyArgs &a) {
    if (a.n <= 0) return;
    const i64 blocks = (a.n + 255) / 256;
    hipLaunchKernelGGL(sort_key_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, a);
}

// OR and AND of all keys: a 4-bit digit whose bits are all equal in both is the same for every key -> its pass is skipped
__global__ void __launch_bounds__(256) key_bits_kernel(const u64 *keys, i64 n, u64 *or_and) {
    u64 o = 0, a = ~0ull;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { o |= keys[i]; a &= keys[i]; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { o |= __shfl_xor(o, d, 64); a &= __shfl_xor(a, d, 64); }
    if ((threadIdx.x & 63) == 0) { atomicOr(or_and, o); atomicAnd(or_and + 1, a); }
}
void launch_key_bits(hipStream_t s, const unsigned long long *keys, int64_t n, unsigned long long *or_and) {
    if (n <= 0) return;
    const i64 blocks = (n + 255) / 256;
    hipLaunchKernelGGL(key_bits_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, (const u64 *)keys, (i64)n, (u64 *)or_and);
}

// ---- one stable radix pass (4 bits): histogram per block of 1024 elements -> scan -> stable scatter ----------------------
constexpr int kSortBlock = 1024;

// digit of element i: 4 key bits at `shift`, or (shift == 64) the validity bit of its row: NULL (0) before everything else
__device__ __forceinline__ int sort_digit(const u64 *keys, const u32 *rows, const u64 *validity, i64 i, int shift) {
    if (shift < 64) return (int)((keys[i] >> shift) & 15ull);
    return bit_at(validity, (i64)rows[i]) ? 1 : 0;
}

__global__ void __launch_bounds__(256) radix_hist_kernel(const u64 *keys, const u32 *rows, const u64 *validity, i64 n, int shift, u32 *hist, i64 nblocks) {
    __shared__ u32 s_cnt[16];
    if (threadIdx.x < 16) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const i64 base = (i64)blockIdx.x * kSortBlock;
    for (int r = 0; r < 4; ++r) {
        const i64 i = base + r * 256 + threadIdx.x;
        if (i < n) atomicAdd(&s_cnt[sort_digit(keys, rows, validity, i, shift)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 16) hist[(i64)threadIdx.x * nblocks + blockIdx.x] = s_cnt[threadIdx.x];   // bucket-major: one scan gives the offsets
}

// exclusive scan of hist[16 * nblocks] by ONE workgroup (the table is small: 16 counters per 1024 rows)
__global__ void __launch_bounds__(1024) radix_scan_kernel(u32 *hist, i64 total) {
    __shared__ u32 s_wave[16];
    __shared__ u32 s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (i64 b = 0; b < total; b += 1024) {
        const i64 i = b + threadIdx.x;
        const u32 v = i < total ? hist[i] : 0u;
        u32 incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        u32 before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        if (i < total) hist[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + incl;
        __syncthreads();
    }
}

// stable scatter: the block's 1024 elements in 4 rounds of 256 (index order); rank of an element = elements of its digit in
// earlier rounds + in earlier waves of its round + in lower lanes of its wave (__ballot per digit, mbcnt)
__global__ void __launch_bounds__(256) radix_scatter_kernel(const u64 *keys, const u32 *rows, const u64 *validity, i64 n, int shift, const u32 *offsets,
                                                            i64 nblocks, u64 *keys_out, u32 *rows_out) {
    __shared__ u32 s_wave_cnt[4][16];
    __shared__ u32 s_running[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 16) s_running[threadIdx.x] = offsets[(i64)threadIdx.x * nblocks + blockIdx.x];
    __syncthreads();
    const i64 base = (i64)blockIdx.x * kSortBlock;
    for (int r = 0; r < 4; ++r) {
        const i64 i = base + r * 256 + threadIdx.x;
        const bool in = i < n;
        const u64 k = in ? keys[i] : 0ull;
        const u32 row = in ? rows[i] : 0u;
        const int digit = in ? sort_digit(keys, rows, validity, i, shift) : 16;
        u32 my_rank = 0;
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            const u64 m = __ballot(digit == d);
            if (digit == d) my_rank = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
            if (lane == 0) s_wave_cnt[wave][d] = (u32)__popcll(m);
        }
        __syncthreads();
        if (in) {
            u32 pos = s_running[digit] + my_rank;
            for (int w = 0; w < wave; ++w) pos += s_wave_cnt[w][digit];
            keys_out[pos] = k;
            rows_out[pos] = row;
        }
        __syncthreads();
        if (threadIdx.x < 16) s_running[threadIdx.x] += s_wave_cnt[0][threadIdx.x] + s_wave_cnt[1][threadIdx.x] + s_wave_cnt[2][threadIdx.x] + s_wave_cnt[3][threadIdx.x];
        __syncthreads();
    }
}

void launch_radix_pass(hipStream_t s, const unsigned long long *keys, const uint32_t *rows, const uint64_t *validity, int64_t n, int shift,
                       uint32_t *hist, unsigned long long *keys_out, uint32_t *rows_out) {
    if (n <= 0) return;
    const i64 nblocks = (n + kSortBlock - 1) / kSortBlock;
    hipLaunchKernelGGL(radix_hist_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, (const u64 *)keys, rows, (const u64 *)validity, (i64)n, shift, hist, nblocks);
    hipLaunchKernelGGL(radix_scan_kernel, dim3(1), dim3(1024), 0, s, hist, 16 * nblocks);
    hipLaunchKernelGGL(radix_scatter_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, (const u64 *)keys, rows, (const u64 *)validity, (i64)n, shift,
                       (const u32 *)hist, nblocks, (u64 *)keys_out, rows_out);
}

// ---- gather of one bitmap through the sorted row ids (value columns use the per-node gather) ---------------------------
__global__ void __launch_bounds__(256) gather_bits_rows_kernel(const u64 *src, const u32 *rows, i64 n, u64 *out) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    const i64 padded = (n + 63) & ~63ll;
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < padded; j += stride) {
        const bool b = j < n && bit_at(src, (i64)rows[j]);
        const u64 w = __ballot(b);
        if ((threadIdx.x & 63) == 0) out[j >> 6] = w;
    }
}
void launch_gather_bits_rows(hipStream_t s, const uint64_t *src, const uint32_t *rows, int64_t n, uint64_t *out) {
    if (n <= 0) return;
    const i64 blocks = (n + 255) / 256;
    hipLaunchKernelGGL(gather_bits_rows_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, (const u64 *)src, rows, (i64)n, (u64 *)out);
}

template <typename T> __global__ void __launch_bounds__(256) gather_rows_kernel(const T *src, const u32 *rows, i64 n, T *out) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) out[j] = src[rows[j]];
}
void launch_gather_rows(hipStream_t s, int width, const void *src, const uint32_t *rows, int64_t n, void *out) {
    if (n <= 0) return;
    const i64 blocks = (n + 255) / 256;
    const dim3 g((unsigned)(blocks < 8192 ? blocks : 8192));
    if (width == 8) hipLaunchKernelGGL(gather_rows_kernel<u64>, g, dim3(256), 0, s, (const u64 *)src, rows, (i64)n, (u64 *)out);
    else hipLaunchKernelGGL(gather_rows_kernel<u32>, g, dim3(256), 0, s, (const u32 *)src, rows, (i64)n, (u32 *)out);
}

}  // namespace qe

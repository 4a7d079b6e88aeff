// qe_pernode_kernels.hip -- one precompiled gfx950 kernel per expression node kind
// (SURVEY 2.1 kernel inventory): arithmetic, negate, cast, comparison -> bitmap via
// __ballot, Kleene logic on 64-row words, IF select, and the filter's stable
// compaction (word popcount -> scan -> index expansion -> gather).
//
// Semantics per node follow evaluator/Interpreter.kt:94-107 (JVM DADD..DREM,
// Double.compare / equals) exactly as the fused kernels do; validity is handled
// by the executor with word-parallel bitmap kernels (k = ka & kb etc.).
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "qe_pernode_kernels.h"
#include "../../include/qe_hip.h"

namespace qe {
namespace pn {

typedef unsigned long long u64;
typedef long long i64;
typedef unsigned int u32;

template <typename T> struct Ld {
    const T *p;
    T s;
    const u32 *idx;   // selection vector (Opnd::idx) or null
    __device__ __forceinline__ T operator[](i64 i) const { return p ? (idx ? p[idx[i]] : p[i]) : s; }
};

template <typename T> static Ld<T> mk(const Opnd &o);
template <> Ld<double> mk<double>(const Opnd &o) { return Ld<double>{(const double *)o.ptr, o.f, o.idx}; }
template <> Ld<i64> mk<i64>(const Opnd &o) { return Ld<i64>{(const i64 *)o.ptr, (i64)o.i, o.idx}; }
template <> Ld<int> mk<int>(const Opnd &o) { return Ld<int>{(const int *)o.ptr, (int)o.i, o.idx}; }

static inline int grid_for(int64_t n, int per_block = 256, int cap = 16384) {
    int64_t b = (n + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ---- arithmetic ------------------------------------------------------------------------------------
template <typename T, int OP> struct ArithOp;
template <int OP> struct ArithOp<double, OP> {
    static __device__ __forceinline__ double apply(double a, double b) {
        if (OP == A_ADD) return a + b;
        if (OP == A_SUB) return a - b;
        if (OP == A_MUL) return a * b;
        if (OP == A_DIV) return a / b;
        return fmod(a, b);   // DREM
    }
};
template <int OP> struct ArithOp<i64, OP> {
    static __device__ __forceinline__ i64 apply(i64 a, i64 b) {
        if (OP == A_ADD) return (i64)((u64)a + (u64)b);
        if (OP == A_SUB) return (i64)((u64)a - (u64)b);
        if (OP == A_MUL) return (i64)((u64)a * (u64)b);
        const i64 d = b == 0 ? 1 : b;   // row is NULL anyway (nonzero bitmap)
        if (OP == A_DIV) return d == -1 ? (i64)(0ull - (u64)a) : a / d;
        return d == -1 ? 0 : a % d;
    }
};
template <int OP> struct ArithOp<int, OP> {
    static __device__ __forceinline__ int apply(int a, int b) {
        if (OP == A_ADD) return (int)((u32)a + (u32)b);
        if (OP == A_SUB) return (int)((u32)a - (u32)b);
        if (OP == A_MUL) return (int)((u32)a * (u32)b);
        const int d = b == 0 ? 1 : b;
        if (OP == A_DIV) return d == -1 ? (int)(0u - (u32)a) : a / d;
        return d == -1 ? 0 : a % d;
    }
};

// Two rows per lane and load: 16 bytes for the 8-byte types (global_load_dwordx4, 1 KiB contiguous per wave instruction),
// 4 pairs in flight per lane before the first use; a scalar operand lives in SGPRs (no column of copies).
template <typename T> struct Pair { T x, y; };
template <typename T> struct V2;
template <> struct V2<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct V2<i64> { typedef i64 type __attribute__((ext_vector_type(2))); };
template <> struct V2<int> { typedef int type __attribute__((ext_vector_type(2))); };
template <typename T> using Vec2 = typename V2<T>::type;

template <typename T> __device__ __forceinline__ Pair<T> ld2(const Ld<T> &a, i64 pair) {
    if (!a.p) return Pair<T>{a.s, a.s};
    if (a.idx) {   // through the selection vector: the two row ids in one 8-byte load, then two gathers
        typedef u32 U2 __attribute__((ext_vector_type(2)));
        const U2 r = __builtin_nontemporal_load((const U2 *)a.idx + pair);
        return Pair<T>{a.p[r.x], a.p[r.y]};
    }
    const Vec2<T> v = __builtin_nontemporal_load((const Vec2<T> *)a.p + pair);
    return Pair<T>{v.x, v.y};
}

template <typename T, int OP>
__global__ void __launch_bounds__(256) k_arith(Ld<T> a, Ld<T> b, T *out, i64 n) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    const i64 npairs = n >> 1;
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < npairs; i += 4 * stride) {
        Pair<T> x[4], y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[j] = ld2(a, i + j * stride); y[j] = ld2(b, i + j * stride); }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Vec2<T> r;
            r.x = ArithOp<T, OP>::apply(x[j].x, y[j].x);
            r.y = ArithOp<T, OP>::apply(x[j].y, y[j].y);
            ((Vec2<T> *)out)[i + j * stride] = r;
        }
    }
    for (; i < npairs; i += stride) {
        const Pair<T> x = ld2(a, i), y = ld2(b, i);
        Vec2<T> r;
        r.x = ArithOp<T, OP>::apply(x.x, y.x);
        r.y = ArithOp<T, OP>::apply(x.y, y.y);
        ((Vec2<T> *)out)[i] = r;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) out[n - 1] = ArithOp<T, OP>::apply(a[n - 1], b[n - 1]);
}

template <typename T> static void arith_t(hipStream_t s, int op, const Opnd &a, const Opnd &b, void *out, int64_t n) {
    const int g = grid_for((n + 1) / 2, 256, 256 * 8);
    Ld<T> la = mk<T>(a), lb = mk<T>(b);
    switch (op) {
    case A_ADD: hipLaunchKernelGGL((k_arith<T, A_ADD>), dim3(g), dim3(256), 0, s, la, lb, (T *)out, (i64)n); break;
    case A_SUB: hipLaunchKernelGGL((k_arith<T, A_SUB>), dim3(g), dim3(256), 0, s, la, lb, (T *)out, (i64)n); break;
    case A_MUL: hipLaunchKernelGGL((k_arith<T, A_MUL>), dim3(g), dim3(256), 0, s, la, lb, (T *)out, (i64)n); break;
    case A_DIV: hipLaunchKernelGGL((k_arith<T, A_DIV>), dim3(g), dim3(256), 0, s, la, lb, (T *)out, (i64)n); break;
    default: hipLaunchKernelGGL((k_arith<T, A_MOD>), dim3(g), dim3(256), 0, s, la, lb, (T *)out, (i64)n); break;
    }
}

void arith(hipStream_t s, int type, int op, Opnd a, Opnd b, void *out, int64_t n) {
    if (n <= 0) return;
    if (type == QE_DOUBLE) arith_t<double>(s, op, a, b, out, n);
    else if (type == QE_INT64) arith_t<i64>(s, op, a, b, out, n);
    else arith_t<int>(s, op, a, b, out, n);
}

template <typename T> __global__ void __launch_bounds__(256) k_neg(const T *a, T *out, i64 n);
template <> __global__ void __launch_bounds__(256) k_neg<double>(const double *a, double *out, i64 n) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = -a[i];   // DNEG
}
template <> __global__ void __launch_bounds__(256) k_neg<i64>(const i64 *a, i64 *out, i64 n) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (i64)(0ull - (u64)a[i]);
}
template <> __global__ void __launch_bounds__(256) k_neg<int>(const int *a, int *out, i64 n) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (int)(0u - (u32)a[i]);
}
void negate(hipStream_t s, int type, const void *a, void *out, int64_t n) {
    if (n <= 0) return;
    const int g = grid_for(n);
    if (type == QE_DOUBLE) hipLaunchKernelGGL(k_neg<double>, dim3(g), dim3(256), 0, s, (const double *)a, (double *)out, (i64)n);
    else if (type == QE_INT64) hipLaunchKernelGGL(k_neg<i64>, dim3(g), dim3(256), 0, s, (const i64 *)a, (i64 *)out, (i64)n);
    else hipLaunchKernelGGL(k_neg<int>, dim3(g), dim3(256), 0, s, (const int *)a, (int *)out, (i64)n);
}

template <typename F, typename T>
__global__ void __launch_bounds__(256) k_cast(const F *a, T *out, i64 n) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (T)a[i];
}
void cast(hipStream_t s, int from, int to, const void *a, void *out, int64_t n) {
    if (n <= 0) return;
    const int g = grid_for(n);
    if (from == QE_INT64 && to == QE_DOUBLE) hipLaunchKernelGGL((k_cast<i64, double>), dim3(g), dim3(256), 0, s, (const i64 *)a, (double *)out, (i64)n);
    else if (from == QE_INT32 && to == QE_DOUBLE) hipLaunchKernelGGL((k_cast<int, double>), dim3(g), dim3(256), 0, s, (const int *)a, (double *)out, (i64)n);
    else hipLaunchKernelGGL((k_cast<int, i64>), dim3(g), dim3(256), 0, s, (const int *)a, (i64 *)out, (i64)n);
}

template <typename T> __global__ void __launch_bounds__(256) k_fill(T v, T *out, i64 n) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = v;
}
void fill(hipStream_t s, int type, Opnd v, void *out, int64_t n) {
    if (n <= 0) return;
    const int g = grid_for(n);
    if (type == QE_DOUBLE) hipLaunchKernelGGL(k_fill<double>, dim3(g), dim3(256), 0, s, v.f, (double *)out, (i64)n);
    else if (type == QE_INT64) hipLaunchKernelGGL(k_fill<i64>, dim3(g), dim3(256), 0, s, (i64)v.i, (i64 *)out, (i64)n);
    else hipLaunchKernelGGL(k_fill<int>, dim3(g), dim3(256), 0, s, (int)v.i, (int *)out, (i64)n);
}

// ---- comparison -> bitmap -----------------------------------------------------------------------------
__device__ __forceinline__ i64 canon_bits(double d) { return d != d ? 0x7ff8000000000000ll : __builtin_bit_cast(i64, d); }
__device__ __forceinline__ int dcmp(double a, double b) {   // java.lang.Double.compare
    if (a < b) return -1;
    if (a > b) return 1;
    const i64 x = canon_bits(a), y = canon_bits(b);
    return x == y ? 0 : (x < y ? -1 : 1);
}

template <typename T, int CMP, int IEEE> struct CmpOp {
    static __device__ __forceinline__ bool apply(T a, T b) {
        if (CMP == C_LT) return a < b;
        if (CMP == C_LE) return a <= b;
        if (CMP == C_GE) return a >= b;
        if (CMP == C_GT) return a > b;
        if (CMP == C_EQ) return a == b;
        return a != b;
    }
};
template <int CMP> struct CmpOp<double, CMP, 0> {   // total order: INTERPRETER / BYTECODE_COMPILER
    static __device__ __forceinline__ bool apply(double a, double b) {
        if (CMP == C_EQ) return canon_bits(a) == canon_bits(b);
        if (CMP == C_NE) return canon_bits(a) != canon_bits(b);
        const int c = dcmp(a, b);
        if (CMP == C_LT) return c < 0;
        if (CMP == C_LE) return c <= 0;
        if (CMP == C_GE) return c >= 0;
        return c > 0;
    }
};
template <int CMP> struct CmpOp<double, CMP, 1> {   // CLOSURE_COMPILER: IEEE <,<=,>=,>; equals() for ==, !=
    static __device__ __forceinline__ bool apply(double a, double b) {
        if (CMP == C_EQ) return canon_bits(a) == canon_bits(b);
        if (CMP == C_NE) return canon_bits(a) != canon_bits(b);
        if (CMP == C_LT) return a < b;
        if (CMP == C_LE) return a <= b;
        if (CMP == C_GE) return a >= b;
        return a > b;
    }
};

// A wave handles chunks of 4 adjacent groups of 128 rows (4 KiB of an 8-byte column): lane l loads rows 2l, 2l+1 of each
// group with ONE load (16 bytes for the 8-byte types, 1 KiB contiguous per wave instruction), all 4 in flight before the
// first use.  Lane l holds the results of rows 2l, 2l+1; word 0 of a group (rows 0..63) wants row i in bit i: lane i fetches
// the pair of lane i >> 1 (word 1: lane 32 + (i >> 1)) with a cross-lane permute and ballots its bit i & 1.  The chunk's 8
// result words leave in ONE store instruction (lanes 0..7, 64 contiguous bytes): a store per word from a single lane kept
// the address unit busy for a whole wave instruction each (15.6 M of them per 1 B rows: 1.6 - 1.9 ms per 8 GB column
// instead of 1.3), and so did interleaving the even / odd ballots with scalar bit tricks (~60 SALU instructions per group
// on the CU's one scalar unit).  Row i = word i >> 6, bit i & 63; bits past the last row are 0.
template <typename T, int CMP, int IEEE, int K>
__global__ void __launch_bounds__(256) k_cmp(Ld<T> a, Ld<T> b, u64 *out, i64 n) {
    const int lane = threadIdx.x & 63;
    const i64 cstride = (i64)gridDim.x * (blockDim.x >> 6);
    const i64 nchunks = (n + 511) >> 9, nfull = n >> 9;
    const i64 nw = (n + 63) >> 6;
    const int src0 = (lane >> 1) << 2, src1 = (32 + (lane >> 1)) << 2;
    // K chunks (a grid stride apart) per step: 4 K loads per operand in flight per lane
    for (i64 c0 = (i64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); c0 < nchunks; c0 += K * cstride) {
        int pair[K][4];
        if (c0 + (K - 1) * cstride < nfull) {
            Pair<T> x[K][4], y[K][4];
#pragma unroll
            for (int q = 0; q < K; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    x[q][j] = ld2(a, ((c0 + q * cstride) * 4 + j) * 64 + lane);
                    y[q][j] = ld2(b, ((c0 + q * cstride) * 4 + j) * 64 + lane);
                }
#pragma unroll
            for (int q = 0; q < K; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    pair[q][j] = (CmpOp<T, CMP, IEEE>::apply(x[q][j].x, y[q][j].x) ? 1 : 0) | (CmpOp<T, CMP, IEEE>::apply(x[q][j].y, y[q][j].y) ? 2 : 0);
        } else {   // near the ragged end: row by row
            // (written with explicit flags: `if (r < n && apply(..)) pair |= 1` came out of hipcc 7.2 with a scalar-mask
            //  sequence that lost the Double.compare tie-break of the even rows -- caught by the special-values parity test)
#pragma unroll
            for (int q = 0; q < K; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const i64 r = ((c0 + q * cstride) * 4 + j) * 128 + 2 * lane;
                    bool r0 = false, r1 = false;
                    if (r < n) r0 = CmpOp<T, CMP, IEEE>::apply(a[r], b[r]);
                    if (r + 1 < n) r1 = CmpOp<T, CMP, IEEE>::apply(a[r + 1], b[r + 1]);
                    pair[q][j] = (r0 ? 1 : 0) | (r1 ? 2 : 0);
                }
        }
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const i64 c = c0 + q * cstride;
            u64 mine = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p0 = __builtin_amdgcn_ds_bpermute(src0, pair[q][j]);
                const int p1 = __builtin_amdgcn_ds_bpermute(src1, pair[q][j]);
                const u64 w0 = __ballot((p0 >> (lane & 1)) & 1);
                const u64 w1 = __ballot((p1 >> (lane & 1)) & 1);
                if (lane == 2 * j) mine = w0;
                if (lane == 2 * j + 1) mine = w1;
            }
            if (lane < 8 && c * 8 + lane < nw) out[c * 8 + lane] = mine;
        }
    }
}

// Variant without cross-lane traffic: lane l takes rows l and l + 64 of every 128-row group (two loads of one element each
// instead of one load of two), so the two ballots ARE the group's two bitmap words -- no ds_bpermute, no per-lane shifts.
template <typename T, int CMP, int IEEE, int K>
__global__ void __launch_bounds__(256) k_cmp_rows(Ld<T> a, Ld<T> b, u64 *out, i64 n) {
    const int lane = threadIdx.x & 63;
    const i64 cstride = (i64)gridDim.x * (blockDim.x >> 6);
    const i64 nchunks = (n + 511) >> 9, nfull = n >> 9;
    const i64 nw = (n + 63) >> 6;
    for (i64 c0 = (i64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); c0 < nchunks; c0 += K * cstride) {
        bool r0[K][4], r1[K][4];
        if (c0 + (K - 1) * cstride < nfull) {
            T x0[K][4], x1[K][4], y0[K][4], y1[K][4];
#pragma unroll
            for (int q = 0; q < K; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const i64 r = ((c0 + q * cstride) * 4 + j) * 128 + lane;
                    x0[q][j] = a[r]; x1[q][j] = a[r + 64];
                    y0[q][j] = b[r]; y1[q][j] = b[r + 64];
                }
#pragma unroll
            for (int q = 0; q < K; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    r0[q][j] = CmpOp<T, CMP, IEEE>::apply(x0[q][j], y0[q][j]);
                    r1[q][j] = CmpOp<T, CMP, IEEE>::apply(x1[q][j], y1[q][j]);
                }
        } else {
#pragma unroll
            for (int q = 0; q < K; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const i64 r = ((c0 + q * cstride) * 4 + j) * 128 + lane;
                    bool t0 = false, t1 = false;
                    if (r < n) t0 = CmpOp<T, CMP, IEEE>::apply(a[r], b[r]);
                    if (r + 64 < n) t1 = CmpOp<T, CMP, IEEE>::apply(a[r + 64], b[r + 64]);
                    r0[q][j] = t0; r1[q][j] = t1;
                }
        }
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const i64 c = c0 + q * cstride;
            u64 mine = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u64 w0 = __ballot(r0[q][j]), w1 = __ballot(r1[q][j]);
                if (lane == 2 * j) mine = w0;
                if (lane == 2 * j + 1) mine = w1;
            }
            if (lane < 8 && c * 8 + lane < nw) out[c * 8 + lane] = mine;
        }
    }
}

template <typename T, int IEEE, int K> static void cmp_tk(hipStream_t s, int cmp, const Opnd &a, const Opnd &b, u64 *out, int64_t n) {
    static const int wgs_per_cu = getenv("QE_PN_CMP_WGS") ? atoi(getenv("QE_PN_CMP_WGS")) : 4;   // 16 waves per CU: 1.25 - 1.28 ms per 8 GB column; 32 waves 1.43 ms
    const int g = grid_for((n + 511) / 512, 4, 256 * wgs_per_cu);   // 4 waves per workgroup, K 512-row chunks per wave and step
    Ld<T> la = mk<T>(a), lb = mk<T>(b);
    // 4-byte columns (INT32, dictionary codes) take the row-per-lane form: k_cmp is bound by its per-group work, not by bytes
    // (~800 G rows/s for 8- and 4-byte columns alike), and without the cross-lane merge an int column runs 0.77 -> 0.43 ms per
    // 600 M rows; for 8-byte columns the two 8-byte loads cost more than the merge saves (1.51 against 1.22 ms per 1 B rows)
    if (sizeof(T) == 4) {
        switch (cmp) {
        case C_LT: hipLaunchKernelGGL((k_cmp_rows<T, C_LT, IEEE, K>), dim3(g), dim3(256), 0, s, la, lb, out, (i64)n); break;
        case C_LE: hipLaunchKernelGGL((k_cmp_rows<T, C_LE, IEEE, K>), dim3(g), dim3(256), 0, s, la, lb, out, (i64)n); break;
        case C_GE: hipLaunchKernelGGL((k_cmp_rows<T, C_GE, IEEE, K>), dim3(g), dim3(256), 0, s, la, lb, out, (i64)n); break;
        case C_GT: hipLaunchKernelGGL((k_cmp_rows<T, C_GT, IEEE, K>), dim3(g), dim3(256), 0, s, la, lb, out, (i64)n); break;
        case C_EQ: hipLaunchKernelGGL((k_cmp_rows<T, C_EQ, IEEE, K>), dim3(g), dim3(256), 0, s, la, lb, out, (i64)n); break;
        default: hipLaunchKernelGGL((k_cmp_rows<T, C_NE, IEEE, K>), dim3(g), dim3(256), 0, s, la, lb, out, (i64)n); break;
        }
        return;
    }
    switch (cmp) {
    case C_LT: hipLaunchKernelGGL((k_cmp<T, C_LT, IEEE, K>), dim3(g), dim3(256), 0, s, la, lb, out, (i64)n); break;
    case C_LE: hipLaunchKernelGGL((k_cmp<T, C_LE, IEEE, K>), dim3(g), dim3(256), 0, s, la, lb, out, (i64)n); break;
    case C_GE: hipLaunchKernelGGL((k_cmp<T, C_GE, IEEE, K>), dim3(g), dim3(256), 0, s, la, lb, out, (i64)n); break;
    case C_GT: hipLaunchKernelGGL((k_cmp<T, C_GT, IEEE, K>), dim3(g), dim3(256), 0, s, la, lb, out, (i64)n); break;
    case C_EQ: hipLaunchKernelGGL((k_cmp<T, C_EQ, IEEE, K>), dim3(g), dim3(256), 0, s, la, lb, out, (i64)n); break;
    default: hipLaunchKernelGGL((k_cmp<T, C_NE, IEEE, K>), dim3(g), dim3(256), 0, s, la, lb, out, (i64)n); break;
    }
}
template <typename T, int IEEE> static void cmp_t(hipStream_t s, int cmp, const Opnd &a, const Opnd &b, u64 *out, int64_t n) {
    static const int k = getenv("QE_PN_CMP_K") ? atoi(getenv("QE_PN_CMP_K")) : 1;
    if (k == 2) cmp_tk<T, IEEE, 2>(s, cmp, a, b, out, n);
    else if (k == 4) cmp_tk<T, IEEE, 4>(s, cmp, a, b, out, n);
    else cmp_tk<T, IEEE, 1>(s, cmp, a, b, out, n);   // K = 2 (8 loads per operand in flight) measured the same: 1.34 - 1.41 ms per 8 GB
}

void compare(hipStream_t s, int type, int cmp, int ieee, Opnd a, Opnd b, uint64_t *out, int64_t n) {
    if (n <= 0) return;
    if (type == QE_DOUBLE) {
        if (ieee) cmp_t<double, 1>(s, cmp, a, b, (u64 *)out, n);
        else cmp_t<double, 0>(s, cmp, a, b, (u64 *)out, n);
    } else if (type == QE_INT64) cmp_t<i64, 0>(s, cmp, a, b, (u64 *)out, n);
    else cmp_t<int, 0>(s, cmp, a, b, (u64 *)out, n);
}

template <typename T> __global__ void __launch_bounds__(256) k_nonzero(Ld<T> b, u64 *out, i64 n) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    const i64 padded = (n + 63) & ~63ll;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < padded; i += stride) {
        const bool r = i < n && b[i < n ? i : 0] != 0;
        const u64 w = __ballot(r);
        if ((threadIdx.x & 63) == 0) out[i >> 6] = w;
    }
}
void nonzero(hipStream_t s, int type, Opnd b, uint64_t *out, int64_t n) {
    if (n <= 0) return;
    const int g = grid_for(n);
    if (type == QE_INT64) hipLaunchKernelGGL(k_nonzero<i64>, dim3(g), dim3(256), 0, s, mk<i64>(b), (u64 *)out, (i64)n);
    else hipLaunchKernelGGL(k_nonzero<int>, dim3(g), dim3(256), 0, s, mk<int>(b), (u64 *)out, (i64)n);
}

// ---- word-parallel logic (64 rows per word) ------------------------------------------------------------
__global__ void __launch_bounds__(256) k_word_op(int op, const u64 *a, const u64 *b, u64 *out, i64 nw) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += stride) {
        const u64 x = a[i], y = b[i];
        u64 r;
        switch (op) {
        case W_AND: r = x & y; break;
        case W_OR: r = x | y; break;
        case W_XOR: r = x ^ y; break;
        case W_ANDNOT: r = x & ~y; break;
        case W_ORNOT: r = x | ~y; break;
        case W_XNOR: r = ~(x ^ y); break;
        case W_NOTAND: r = ~x & y; break;
        default: r = ~x | y; break;
        }
        out[i] = r;
    }
}
void word_op(hipStream_t s, int op, const uint64_t *a, const uint64_t *b, uint64_t *out, int64_t nw) {
    if (nw <= 0) return;
    hipLaunchKernelGGL(k_word_op, dim3(grid_for(nw)), dim3(256), 0, s, op, (const u64 *)a, (const u64 *)b, (u64 *)out, (i64)nw);
}
__global__ void __launch_bounds__(256) k_word_not(const u64 *a, u64 *out, i64 nw) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += stride) out[i] = ~a[i];
}
void word_not(hipStream_t s, const uint64_t *a, uint64_t *out, int64_t nw) {
    if (nw <= 0) return;
    hipLaunchKernelGGL(k_word_not, dim3(grid_for(nw)), dim3(256), 0, s, (const u64 *)a, (u64 *)out, (i64)nw);
}
__global__ void __launch_bounds__(256) k_word_fill(u64 v, u64 *out, i64 nw) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += stride) out[i] = v;
}
void word_fill(hipStream_t s, uint64_t v, uint64_t *out, int64_t nw) {
    if (nw <= 0) return;
    hipLaunchKernelGGL(k_word_fill, dim3(grid_for(nw)), dim3(256), 0, s, (u64)v, (u64 *)out, (i64)nw);
}

// Kleene three-valued AND / OR (Interpreter.kt:54-91; SURVEY 2.1 word-parallel form)
__global__ void __launch_bounds__(256) k_kleene(int is_and, const u64 *va, const u64 *ka, const u64 *vb, const u64 *kb,
                                                u64 *vout, u64 *kout, i64 nw) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += stride) {
        const u64 xa = ka ? ka[i] : ~0ull, xb = kb ? kb[i] : ~0ull;
        const u64 a = va[i] & xa, b = vb[i] & xb;   // value bits under a null are 0
        u64 v, k;
        if (is_and) {
            v = a & b;
            k = (xa & xb) | (xa & ~a) | (xb & ~b);   // false dominates null
        } else {
            v = a | b;
            k = (xa & xb) | a | b;                   // true dominates null
        }
        vout[i] = v;
        if (kout) kout[i] = k;
    }
}
void kleene(hipStream_t s, bool is_and, const uint64_t *va, const uint64_t *ka, const uint64_t *vb, const uint64_t *kb,
            uint64_t *vout, uint64_t *kout, int64_t nw) {
    if (nw <= 0) return;
    hipLaunchKernelGGL(k_kleene, dim3(grid_for(nw)), dim3(256), 0, s, is_and ? 1 : 0, (const u64 *)va, (const u64 *)ka,
                       (const u64 *)vb, (const u64 *)kb, (u64 *)vout, (u64 *)kout, (i64)nw);
}

// ---- IF -----------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_select(const u64 *cond, Ld<T> t, Ld<T> e, T *out, i64 n) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const bool c = (cond[i >> 6] >> (i & 63)) & 1ull;
        out[i] = c ? t[i] : e[i];
    }
}
void select(hipStream_t s, int type, const uint64_t *cond, Opnd t, Opnd e, void *out, int64_t n) {
    if (n <= 0) return;
    const int g = grid_for(n);
    if (type == QE_DOUBLE) hipLaunchKernelGGL(k_select<double>, dim3(g), dim3(256), 0, s, (const u64 *)cond, mk<double>(t), mk<double>(e), (double *)out, (i64)n);
    else if (type == QE_INT64) hipLaunchKernelGGL(k_select<i64>, dim3(g), dim3(256), 0, s, (const u64 *)cond, mk<i64>(t), mk<i64>(e), (i64 *)out, (i64)n);
    else hipLaunchKernelGGL(k_select<int>, dim3(g), dim3(256), 0, s, (const u64 *)cond, mk<int>(t), mk<int>(e), (int *)out, (i64)n);
}
__global__ void __launch_bounds__(256) k_select_words(const u64 *c, const u64 *t, const u64 *e, u64 *out, i64 nw) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += stride) out[i] = (c[i] & t[i]) | (~c[i] & e[i]);
}
void select_words(hipStream_t s, const uint64_t *c, const uint64_t *t, const uint64_t *e, uint64_t *out, int64_t nw) {
    if (nw <= 0) return;
    hipLaunchKernelGGL(k_select_words, dim3(grid_for(nw)), dim3(256), 0, s, (const u64 *)c, (const u64 *)t, (const u64 *)e, (u64 *)out, (i64)nw);
}

// ---- filter: bitmap -> ascending row ids (stable) -------------------------------------------------------------
__device__ __forceinline__ u64 keep_word(const u64 *v, const u64 *k, i64 w, i64 n) {
    u64 x = v[w];
    if (k) x &= k[w];                               // FilterOperator.kt:20: non-null AND true
    const i64 rem = n - w * 64;
    if (rem < 64) x &= (1ull << rem) - 1ull;        // bits past the last row
    return x;
}
__global__ void __launch_bounds__(256) k_word_popcounts(const u64 *v, const u64 *k, i64 n, u32 *counts, i64 nw) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 w = (i64)blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += stride) counts[w] = (u32)__popcll(keep_word(v, k, w, n));
}
void word_popcounts(hipStream_t s, const uint64_t *v, const uint64_t *k, int64_t n, uint32_t *counts, int64_t nw) {
    if (nw <= 0) return;
    hipLaunchKernelGGL(k_word_popcounts, dim3(grid_for(nw)), dim3(256), 0, s, (const u64 *)v, (const u64 *)k, (i64)n, counts, (i64)nw);
}

// exclusive scan of u32 counts in three passes: block sums (1024 per block), scan of the block sums by ONE
// workgroup, per-block scan with offset.  Wave-level scans use DPP-free shuffles; 16 waves per block.
__device__ __forceinline__ u32 wave_incl_scan(u32 x, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    return x;
}
__device__ __forceinline__ u32 block_excl_scan_1024(u32 x, u32 *lds, u32 &block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 incl = wave_incl_scan(x, lane);
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        const u32 t = lane < 16 ? lds[lane] : 0u;
        const u32 ti = wave_incl_scan(t, lane);
        if (lane < 16) lds[16 + lane] = ti - t;
        if (lane == 15) lds[32] = ti;
    }
    __syncthreads();
    block_total = lds[32];
    const u32 r = lds[16 + wave] + incl - x;
    __syncthreads();
    return r;
}
__global__ void __launch_bounds__(1024) k_scan_block_sums(const u32 *in, u32 *block_sums, i64 n) {
    __shared__ u32 lds[40];
    const i64 i = (i64)blockIdx.x * 1024 + threadIdx.x;
    u32 total;
    (void)block_excl_scan_1024(i < n ? in[i] : 0u, lds, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}
__global__ void __launch_bounds__(1024) k_scan_of_sums(u32 *block_sums, i64 nblocks, u64 *total_out) {
    __shared__ u32 lds[40];
    u64 carry = 0;
    for (i64 base = 0; base < nblocks; base += 1024) {
        const i64 i = base + threadIdx.x;
        const u32 x = i < nblocks ? block_sums[i] : 0u;
        u32 total;
        const u32 e = block_excl_scan_1024(x, lds, total);
        if (i < nblocks) block_sums[i] = (u32)(carry + e);
        carry += total;
    }
    if (threadIdx.x == 0) *total_out = carry;
}
__global__ void __launch_bounds__(1024) k_scan_final(const u32 *in, const u32 *block_offsets, u32 *out, i64 n) {
    __shared__ u32 lds[40];
    const i64 i = (i64)blockIdx.x * 1024 + threadIdx.x;
    u32 total;
    const u32 e = block_excl_scan_1024(i < n ? in[i] : 0u, lds, total);
    if (i < n) out[i] = block_offsets[blockIdx.x] + e;
}
void exclusive_scan_u32(hipStream_t s, const uint32_t *in, uint32_t *out, uint32_t *block_sums, int64_t n,
                        unsigned long long *total) {
    if (n <= 0) return;
    const int64_t nblocks = (n + 1023) / 1024;
    hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)nblocks), dim3(1024), 0, s, in, block_sums, (i64)n);
    hipLaunchKernelGGL(k_scan_of_sums, dim3(1), dim3(1024), 0, s, block_sums, (i64)nblocks, (u64 *)total);
    hipLaunchKernelGGL(k_scan_final, dim3((unsigned)nblocks), dim3(1024), 0, s, in, block_sums, out, (i64)n);
}

// A LANE per bitmap word: lane l walks the set bits of word w0 + l and writes their row ids from word_offsets[w] on.  The 64
// words of a wave are adjacent and so are their output ranges, so step i of the walk is one store instruction with every
// lane that still has a bit active, all within a few hundred bytes.  (One wave per word -- lanes = bits -- issued a store
// instruction with ~3 active lanes per word at 5 % selectivity: 1.43 ms per 1 B rows.)
// Round 2: when the wave's 64 words hold at most 1024 kept rows (the usual case below ~25 %) the row ids first meet in a 4 KiB
// LDS buffer of the wave and leave in whole 256-byte store instructions: the direct form issues up to max-bits-per-word store
// instructions of 64 lanes x 4 bytes spread over ~12 lines each (0.27 ms per 1 B rows at 10 %).
__global__ void __launch_bounds__(256) k_expand_indices(const u64 *v, const u64 *k, i64 n, const u32 *word_offsets,
                                                        u32 *indices, i64 nw) {
    __shared__ u32 s_buf[4][1024];
    u32 *buf = s_buf[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    const i64 nw_pad = (nw + 63) & ~63ll;   // whole waves: the wave-level steps below need every lane
    for (i64 w = (i64)blockIdx.x * blockDim.x + threadIdx.x; w < nw_pad; w += stride) {
        u64 x = w < nw ? keep_word(v, k, w, n) : 0ull;
        u32 pos = w < nw ? word_offsets[w] : 0u;
        const u32 base = (u32)(w * 64);
        const u32 first = (u32)__builtin_amdgcn_readfirstlane((int)pos);                                   // the wave's words are adjacent
        const u32 cnt = (u32)__popcll(x);
        // total of the wave: an inclusive scan is not needed, offsets are already exclusive -- last valid lane's pos + cnt
        u32 endpos = pos + cnt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const u32 t = (u32)__shfl_xor((int)endpos, o, 64);
            endpos = t > endpos ? t : endpos;
        }
        const u32 total = endpos - first;
        if (total <= 1024u) {
            u32 q = pos - first;
            while (x != 0) {
                buf[q++] = base + (u32)__builtin_ctzll(x);
                x &= x - 1;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (u32 j = (u32)lane; j < total; j += 64u) indices[first + j] = buf[j];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        } else {
            while (x != 0) {
                indices[pos++] = base + (u32)__builtin_ctzll(x);
                x &= x - 1;
            }
        }
    }
}
void expand_indices(hipStream_t s, const uint64_t *v, const uint64_t *k, int64_t n, const uint32_t *word_offsets,
                    uint32_t *indices, int64_t nw) {
    if (nw <= 0) return;
    hipLaunchKernelGGL(k_expand_indices, dim3(grid_for(nw, 256, 256 * 8)), dim3(256), 0, s, (const u64 *)v, (const u64 *)k, (i64)n,
                       word_offsets, indices, (i64)nw);
}

// Gather of up to 8 value columns at the kept row ids, one output row per lane and step, 4 rows in flight per lane: every
// lane of every load carries a row (a masked one-row-per-lane compaction straight from the bitmap was measured 2.8x
// slower at 5 % selectivity: the address unit spends its cycles per wave instruction, not per active lane).
// Round 2, second try at 10 %: streaming the column and compacting through the keep bitmap (masked stores, or an LDS buffer per
// 512-row chunk and one coalesced store) took 1.71 - 1.78 ms per 8 GB column against 1.35 ms for this gather.
__global__ void __launch_bounds__(256) k_gather_multi(const GatherArgs a) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; j + 3 * stride < a.m; j += 4 * stride) {
        u32 r[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = __builtin_nontemporal_load(a.idx + j + q * stride);
        for (int c = 0; c < a.ncols; ++c) {
            if (a.width[c] == 8) {
                u64 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = ((const u64 *)a.src[c])[r[q]];
#pragma unroll
                for (int q = 0; q < 4; ++q) ((u64 *)a.dst[c])[j + q * stride] = v[q];
            } else {
                u32 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = ((const u32 *)a.src[c])[r[q]];
#pragma unroll
                for (int q = 0; q < 4; ++q) ((u32 *)a.dst[c])[j + q * stride] = v[q];
            }
        }
    }
    for (; j < a.m; j += stride) {
        const u32 r = a.idx[j];
        for (int c = 0; c < a.ncols; ++c) {
            if (a.width[c] == 8) ((u64 *)a.dst[c])[j] = ((const u64 *)a.src[c])[r];
            else ((u32 *)a.dst[c])[j] = ((const u32 *)a.src[c])[r];
        }
    }
}
void gather_multi(hipStream_t s, const GatherArgs &a) {
    if (a.m <= 0 || a.ncols <= 0) return;
    hipLaunchKernelGGL(k_gather_multi, dim3(grid_for((a.m + 3) / 4, 256, 256 * 8)), dim3(256), 0, s, a);
}

template <typename T> __global__ void __launch_bounds__(256) k_gather(const T *src, const u32 *idx, T *out, i64 m) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) out[j] = src[idx[j]];
}
void gather(hipStream_t s, int type, const void *src, const uint32_t *idx, void *out, int64_t m) {
    if (m <= 0) return;
    const int g = grid_for(m);
    if (type == QE_DOUBLE || type == QE_INT64) hipLaunchKernelGGL(k_gather<u64>, dim3(g), dim3(256), 0, s, (const u64 *)src, idx, (u64 *)out, (i64)m);
    else hipLaunchKernelGGL(k_gather<u32>, dim3(g), dim3(256), 0, s, (const u32 *)src, idx, (u32 *)out, (i64)m);
}
__global__ void __launch_bounds__(256) k_gather_bits(const u64 *src, const u32 *idx, u64 *out, i64 m) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    const i64 padded = (m + 63) & ~63ll;
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < padded; j += stride) {
        bool b = false;
        if (j < m) {
            const u32 r = idx[j];
            b = (src[r >> 6] >> (r & 63)) & 1ull;
        }
        const u64 w = __ballot(b);
        if ((threadIdx.x & 63) == 0) out[j >> 6] = w;
    }
}
void gather_bits(hipStream_t s, const uint64_t *src, const uint32_t *idx, uint64_t *out, int64_t m) {
    if (m <= 0) return;
    hipLaunchKernelGGL(k_gather_bits, dim3(grid_for(m)), dim3(256), 0, s, (const u64 *)src, idx, (u64 *)out, (i64)m);
}

__global__ void __launch_bounds__(256) k_lookup_codes(const int *table, int ntable, const int *codes, int *out, i64 n) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const int c = codes[j];
        out[j] = (unsigned)c < (unsigned)ntable ? table[c] : -1;
    }
}
void lookup_codes(hipStream_t s, const int32_t *table, int32_t ntable, const int32_t *codes, int32_t *out, int64_t n) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_lookup_codes, dim3(grid_for(n)), dim3(256), 0, s, (const int *)table, (int)ntable, (const int *)codes, (int *)out, (i64)n);
}

}  // namespace pn
}  // namespace qe

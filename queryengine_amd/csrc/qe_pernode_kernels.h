// qe_pernode_kernels.h -- launchers of the precompiled kernel-per-expression-node kernels (gfx950).
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdint>

namespace qe {
namespace pn {

enum Arith { A_ADD = 0, A_SUB, A_MUL, A_DIV, A_MOD };
enum Cmp { C_LT = 0, C_LE, C_GE, C_GT, C_EQ, C_NE };
enum WordOp { W_AND = 0, W_OR, W_XOR, W_ANDNOT /* a & ~b */, W_ORNOT /* a | ~b */, W_XNOR, W_NOTAND /* ~a & b */, W_NOTOR /* ~a | b */ };

// operand: column pointer, or (ptr == nullptr) a scalar broadcast from an SGPR
struct Opnd {
    const void *ptr;
    double f;
    long long i;
    // selection vector: element j of the operand is ptr[idx[j]] (a batch column read through the row ids of a narrowed domain,
    // instead of a gathered copy); null = ptr[j]
    const uint32_t *idx = nullptr;
};

// type codes follow QE_* (QE_DOUBLE=1, QE_INT64=3, QE_INT32=4; STRING codes are handled as INT32)
void arith(hipStream_t s, int type, int op, Opnd a, Opnd b, void *out, int64_t n);
void negate(hipStream_t s, int type, const void *a, void *out, int64_t n);
void cast(hipStream_t s, int from, int to, const void *a, void *out, int64_t n);
void fill(hipStream_t s, int type, Opnd v, void *out, int64_t n);
// comparison -> value bitmap (one __ballot per 64 rows); bits of rows >= n are 0
void compare(hipStream_t s, int type, int cmp, int ieee, Opnd a, Opnd b, uint64_t *out, int64_t n);
// `b != 0` as a bitmap (integer division by zero => NULL)
void nonzero(hipStream_t s, int type, Opnd b, uint64_t *out, int64_t n);
void word_op(hipStream_t s, int op, const uint64_t *a, const uint64_t *b, uint64_t *out, int64_t nwords);
void word_not(hipStream_t s, const uint64_t *a, uint64_t *out, int64_t nwords);
void word_fill(hipStream_t s, uint64_t v, uint64_t *out, int64_t nwords);
// Kleene AND / OR on (value, known) bitmaps; ka / kb may be null (= all known); kout may be null if both are
void kleene(hipStream_t s, bool is_and, const uint64_t *va, const uint64_t *ka, const uint64_t *vb, const uint64_t *kb,
            uint64_t *vout, uint64_t *kout, int64_t nwords);
// IF: out = cond ? t : e for value columns; for bitmaps use select_words
void select(hipStream_t s, int type, const uint64_t *cond, Opnd t, Opnd e, void *out, int64_t n);
void select_words(hipStream_t s, const uint64_t *cond, const uint64_t *t, const uint64_t *e, uint64_t *out, int64_t nwords);
// selection vector: keep bitmap (& optional known bitmap) -> ascending row ids (stable compaction)
void word_popcounts(hipStream_t s, const uint64_t *v, const uint64_t *k, int64_t n, uint32_t *counts, int64_t nwords);
void exclusive_scan_u32(hipStream_t s, const uint32_t *in, uint32_t *out, uint32_t *block_sums, int64_t n,
                        unsigned long long *total);
void expand_indices(hipStream_t s, const uint64_t *v, const uint64_t *k, int64_t n, const uint32_t *word_offsets,
                    uint32_t *indices, int64_t nwords);
// gather of up to 8 value columns (4- or 8-byte elements) at the kept row ids in ONE launch (the ids are read once)
struct GatherArgs {
    const void *src[8];
    void *dst[8];
    int width[8];
    int ncols;
    const uint32_t *idx;
    long long m;
};
void gather_multi(hipStream_t s, const GatherArgs &a);
void gather(hipStream_t s, int type, const void *src, const uint32_t *idx, void *out, int64_t m);
void gather_bits(hipStream_t s, const uint64_t *src, const uint32_t *idx, uint64_t *out, int64_t m);
// out[i] = table[codes[i]] (a code outside the table -- the garbage under a NULL -- reads as -1): string ranks / code remaps
void lookup_codes(hipStream_t s, const int32_t *table, int32_t ntable, const int32_t *codes, int32_t *out, int64_t n);

}  // namespace pn
}  // namespace qe

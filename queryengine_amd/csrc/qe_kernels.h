// qe_kernels.h -- host-callable launchers of the precompiled (AOT) gfx950 kernels.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdint>
#include "../../include/qe_hip.h"

namespace qe {

// synthetic column generator; data/validity are device pointers
void launch_generate(hipStream_t s, const qe_gen_spec &spec, uint64_t seed, int64_t row_begin, int64_t nrows,
                     void *data, uint64_t *validity);
// bytes (0/1 per row) -> bitmap words (row i = word i>>6 bit i&63)
void launch_pack_bytes(hipStream_t s, const uint8_t *bytes, int64_t n, uint64_t *words);
// plain streaming read of nbytes (16 B per lane), result folded into sink[0] so nothing is elided
void launch_stream_read(hipStream_t s, const void *src, int64_t nbytes, unsigned long long *sink, int wgs_per_cu = 8);

// read stream + a trickle of writes (one 512-byte block per wave every `write_every` read iterations of 8 KiB)
void launch_stream_read_write(hipStream_t s, const void *src, int64_t nbytes, unsigned long long *sink, void *dst,
                              int64_t dst_bytes, int write_every, int window_period, int window_len, int blocks_per_event);

// ---- partitioned group-by (domains that do not fit LDS) ----
// counts[chunk][part] (kept rows) -> in place: offset of (chunk, part) inside its partition; totals[part] = rows of the partition
void launch_gb_scan(hipStream_t s, uint32_t *counts, int64_t nchunks, int nparts, unsigned long long *totals);


// hashed group-by: fill a table of nentries entries with the per-word pattern of an empty entry; copy the entries in use
// (state word == 2) to a dense array, *counter = how many
struct HtInit {
    int words;
    unsigned long long word[40];
};
void launch_ht_init(hipStream_t s, unsigned long long *tab, int64_t nentries, const HtInit &init);
void launch_ht_collect(hipStream_t s, const unsigned long long *tab, int64_t nentries, int words, unsigned long long *dense,
                       unsigned int *counter);

// ---- groups of a hashed GROUP BY finished on the device (qe_kernels.hip) ----
// entries: m x `words` u64 {state, null bits of the keys, key words.., first row, (count, acc) per aggregate}
// sort keys: keys[i] = first row of entry i, rows[i] = i; after the radix passes rows[] lists the entries in insertion order
void launch_group_sort_keys(hipStream_t s, const unsigned long long *entries, int words, int first_row_word, int64_t m,
                            unsigned long long *keys, uint32_t *rows);
struct GroupFinishArgs {
    const unsigned long long *entries;
    const unsigned int *rows;        // entry of result row j
    long long m;
    int words, nkeys, nagg;
    int key_type[4];                 // QE_* of the key columns
    void *key_data[4];               // DOUBLE / INT64: u64[m]; INT32 / STRING: i32[m]; BOOLEAN: bitmap words
    unsigned long long *key_valid[4];
    int agg_fn[8], cnt_src[8];       // QE_AGG_*, and which aggregate's counter says whether aggregate i saw a value
    double *agg_data[8];
    unsigned long long *agg_valid[8];
    unsigned int *flags;             // [k] = 1: some group has a NULL in key k; [4 + i]: aggregate i is NULL somewhere
};
void launch_group_finish(hipStream_t s, const GroupFinishArgs &a);

// ---- ORDER BY (qe_sort.hip) ----
// keys[i] = order-preserving u64 image of row i of the key column (0 under a NULL), rows[i] = i
struct SortKeyArgs {
    int type;                       // QE_* of the key column
    const void *data;
    const unsigned long long *validity;
    const int *ranks;               // QE_STRING: compareTo rank per dictionary code
    int nranks;
    long long n;
    unsigned long long *keys;
    unsigned int *rows;
};
void launch_sort_keys(hipStream_t s, const SortKeyArgs &a);
// or_and[0] |= every key, or_and[1] &= every key (caller initialises to {0, ~0})
void launch_key_bits(hipStream_t s, const unsigned long long *keys, int64_t n, unsigned long long *or_and);
// one stable LSD radix pass over the 4 key bits at `shift` (shift == 64: over the validity bit of each element's row, NULL
// first); hist: 16 * ceil(n / 1024) u32 of scratch
void launch_radix_pass(hipStream_t s, const unsigned long long *keys, const uint32_t *rows, const uint64_t *validity, int64_t n, int shift,
                       uint32_t *hist, unsigned long long *keys_out, uint32_t *rows_out);
void launch_gather_rows(hipStream_t s, int width, const void *src, const uint32_t *rows, int64_t n, void *out);
void launch_gather_bits_rows(hipStream_t s, const uint64_t *src, const uint32_t *rows, int64_t n, uint64_t *out);

// place nbits bits of src at bit offset dst_bit_offset of dst (bitmap words; concatenation of results / gather)
void launch_bitmap_place(hipStream_t s, uint64_t *dst, int64_t dst_bit_offset, const uint64_t *src, int64_t nbits);

}  // namespace qe

/*
 * qe_columnar.c -- STRONG CPU BASELINE for config 2.  TEST / BENCH INFRASTRUCTURE ONLY (lives under oracle/).
 *
 * SURVEY.md 8(d) asks for a "cpu-columnar" baseline next to the row-at-a-time port: the same query
 *     SELECT a + b, c * 2.0 FROM t WHERE a < 100 AND c < 0.5          (a, b INT64; c DOUBLE; no nulls)
 * written the way a columnar CPU engine would run it -- hand-specialised, branch-free predicate, two passes
 * (count per block, exclusive scan, compacting write), all host cores via OpenMP.  It is NOT the reference's
 * algorithm (the reference boxes every row, operator/FilterOperator.kt:14-25) and is never shipped or called by the
 * product; tests/test_oracle_golden.py checks it bit-for-bit against qe_oracle.c, bench.py times it as
 * cpu_baseline.columnar.
 *
 * Semantics kept from the oracle: (double)a compared with Double.compare against 100.0 and c against c_limit --
 * for literals that are neither NaN nor +-0 that is the IEEE '<' (a NaN in c is "greater": not kept, as here);
 * a + b wraps like a Java long; c * 2.0 is one IEEE multiply.
 */
#include <stdint.h>
#include <stdlib.h>
#include <omp.h>

#define QC_BLOCK 65536

int64_t qc_config2(const int64_t *a, const int64_t *b, const double *c, int64_t n, double a_limit, double c_limit,
                   int64_t *out0, double *out1, int32_t nthreads, int32_t *threads_used) {
    const int64_t nblocks = (n + QC_BLOCK - 1) / QC_BLOCK;
    int64_t *counts = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nblocks + 1));
    if (!counts) return -1;
    if (nthreads > 0) omp_set_num_threads(nthreads);
    int used = 1;
#pragma omp parallel
    {
#pragma omp single
        used = omp_get_num_threads();
#pragma omp for schedule(static)
        for (int64_t blk = 0; blk < nblocks; blk++) {
            const int64_t lo = blk * QC_BLOCK, hi = lo + QC_BLOCK < n ? lo + QC_BLOCK : n;
            int64_t cnt = 0;
            for (int64_t i = lo; i < hi; i++) cnt += ((double)a[i] < a_limit) & (c[i] < c_limit);
            counts[blk + 1] = cnt;
        }
    }
    counts[0] = 0;
    for (int64_t blk = 0; blk < nblocks; blk++) counts[blk + 1] += counts[blk];
#pragma omp parallel for schedule(static)
    for (int64_t blk = 0; blk < nblocks; blk++) {
        const int64_t lo = blk * QC_BLOCK, hi = lo + QC_BLOCK < n ? lo + QC_BLOCK : n;
        int64_t pos = counts[blk];
        for (int64_t i = lo; i < hi; i++) {
            /* a store only for kept rows: an unconditional store + conditional advance would let the last dropped row of
             * a block scribble over the first output of the next block, which another thread may already have written */
            if (((double)a[i] < a_limit) & (c[i] < c_limit)) {
                out0[pos] = (int64_t)((uint64_t)a[i] + (uint64_t)b[i]);
                out1[pos] = c[i] * 2.0;
                pos++;
            }
        }
    }
    const int64_t total = counts[nblocks];
    free(counts);
    if (threads_used) *threads_used = used;
    return total;
}

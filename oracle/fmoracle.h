/* oracle/fmoracle.h — CPU restatement of the reference's backward-search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, imported by or executed
 * from the product (fmindex-collection_amd/, include/).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it, and only as the checker / reported baseline.
 *
 * Parity status: PINNED —
 *   * string layer (layouts byte for byte, rank / prefix_rank / symbol) against the real
 *     reference headers compiled in place (oracle/ref_driver.cpp -> oracle/_ref/libfmref.so);
 *   * search-scheme tables (h2, pigeon_opt, backtracking, expand, limitToHamming,
 *     createUniformPartition, isValid, isComplete) against the same library;
 *   * FMIndex / BiFMIndex construction, cursors, search_no_errors, search_backtracking,
 *     search_ng26<Edit=false>, locate: against the golden vectors of the reference's own
 *     tests (tests/golden/reference_tests.json: checkFMIndex.cpp, checkBiFMIndex.cpp,
 *     check*Cursor.cpp, checkSearches.cpp, checkSearchBacktracking.cpp) and against
 *     brute-force text scans (the SA interval of a pattern is a mathematical function of
 *     the text).  Those reference layers cannot be compiled here: they include utils.h,
 *     which needs libsais / mmser headers that are fetched from the network.
 *
 * Every function cites the reference file:line (relative to
 * /root/reference/src/fmindex-collection/) whose behaviour it restates.
 */
#ifndef FMORACLE_H
#define FMORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* layout ids — identical numbering in include/fmgpu.h and oracle/ref_driver.cpp */
enum ora_layout {
    ORA_IB8 = 0, ORA_IB16 = 1, ORA_IB32 = 2, ORA_IB16A = 3,  /* string/InterleavedBitvector.h:168-173 */
    ORA_IBP16 = 4,                                           /* string/InterleavedBitvectorPrefix.h:204-209 */
    ORA_EPR8 = 5, ORA_EPR16 = 6, ORA_EPR32 = 7,              /* string/InterleavedEPR.h:222-227 */
    ORA_EPRV2_8 = 8, ORA_EPRV2_16 = 9, ORA_EPRV2_32 = 10,    /* string/InterleavedEPRV2.h:289-309 */
    ORA_WAVELET = 11,                                        /* string/Wavelet.h:27-28 */
    ORA_EPRV3_8 = 12, ORA_EPRV3_16 = 13, ORA_EPRV3_32 = 14,  /* string/EPRV3.h:263-272 */
    ORA_EPRV4 = 15,                                          /* string/EPRV4.h:14 */
    ORA_EPRV5 = 16,                                          /* string/EPRV5.h:14 */
    ORA_IEPRV7 = 17,                                         /* string/InterleavedEPRV7.h:15 */
    ORA_FBV_64_64K = 18, ORA_FBV_512_64K = 19, ORA_FBV_2048_64K = 20,   /* string/FlattenedBitvectors2L.h:274-279 (512_64k: the default String of FMIndex, fmindex/FMIndex.h:14) */
    ORA_LAYOUT_COUNT = 21
};

/* ---------------------------------------------------------------- strings with rank support */
typedef struct ora_string ora_string;

ora_string* ora_string_build(int layout, int sigma, const uint8_t* symbols, uint64_t n);
void        ora_string_free(ora_string* s);
uint64_t    ora_string_size(const ora_string* s);
int         ora_string_sigma(const ora_string* s);
int         ora_string_layout(const ora_string* s);
uint64_t    ora_rank(const ora_string* s, uint64_t idx, uint64_t symb);
uint64_t    ora_prefix_rank(const ora_string* s, uint64_t idx, uint64_t symb);
uint64_t    ora_symbol(const ora_string* s, uint64_t idx);
/* mathematically correct all_ranks_and_prefix_ranks (string/concepts.h:50-64), rs/prs hold sigma entries */
void        ora_all_ranks_and_prefix_ranks(const ora_string* s, uint64_t idx, uint64_t* rs, uint64_t* prs);
/* raw arrays in the reference's in-memory layout.
 * blocked layouts: part 0 = blocks, part 1 = superBlocks ([k][sigma] u64)
 * wavelet:         part node*4 + {0: superblocks u64, 1: blocks u8, 2: bits u64, 3: totalLength u64}
 * EPRV3/4/5/7:     part 0 = bits (InBits per 64 rows; V7: the packed {bits, level0} structs), part 1 = superBlocks,
 *                  part 2.. = counter levels bottom-up (V3: blocks_; V4: level0, level1, level2; V5: level0, level1; V7: level1 at part 3)
 * FlattenedBitvectors2L: part 0 = bits (bitct bitsets of l1_bits per block), part 1 = l0 ([k][sigma+1] u64), part 2 = l1 ([k][sigma+1] u16).
 *                  This header includes ../utils.h (libsais, mmser) and cannot be compiled here: the array layout is restated from the
 *                  source text and pinned only through the reference's String unit-test vectors (values, not bytes). */
int         ora_string_raw(const ora_string* s, int part, const void** ptr, uint64_t* bytes);
uint64_t    ora_string_block_stride(const ora_string* s);
uint64_t    ora_string_bits_offset(const ora_string* s);

/* ---------------------------------------------------------------- sampled suffix array (SparseArray) */
typedef struct ora_dense_vector {            /* DenseVector.h:26-205 */
    uint64_t* data; uint64_t nwords;
    uint64_t bitCount; uint8_t bits; uint64_t largestValue; uint64_t commonDivisor;
} ora_dense_vector;

typedef struct ora_sparse {                  /* suffixarray/SparseArray.h:31-76 */
    uint64_t n;                              /* number of rows */
    uint64_t* l0; uint64_t nl0;              /* bitvector/Bitvector2L.h:26-171, <512, 65536> */
    uint16_t* l1; uint64_t nl1;
    uint64_t* bits; uint64_t nbitwords;      /* 8 words per 512-bit block */
    ora_dense_vector field[2];               /* DenseMultiVector<tuple<u32,u32>>: seqId, pos */
    uint64_t nvalues;
} ora_sparse;

ora_sparse* ora_sparse_build(uint64_t n, const uint8_t* has, const uint64_t* seq, const uint64_t* pos);
void        ora_sparse_free(ora_sparse* s);
int         ora_sparse_value(const ora_sparse* s, uint64_t idx, uint64_t* seq, uint64_t* pos);  /* 1 if present */
uint64_t    ora_dense_access(const ora_dense_vector* v, uint64_t i);
ora_dense_vector* ora_dense_build(const uint64_t* values, uint64_t n, uint64_t largest, uint64_t divisor);   /* DenseVector.h:57-61, :84-99 (largest = divisor = 0: from the values) */
ora_dense_vector* ora_dense_concat(const ora_dense_vector* a, const ora_dense_vector* b);                     /* DenseVector.h:38-50 */
uint64_t    ora_dense_size(const ora_dense_vector* v);
void        ora_dense_free(ora_dense_vector* v);
uint64_t    ora_sparse_rank(const ora_sparse* s, uint64_t idx);                                               /* bitvector/Bitvector2L.h:123-142 */

/* ---------------------------------------------------------------- FMIndex / BiFMIndex */
typedef struct ora_index {
    int sigma, layout, bidirectional;
    uint64_t n;
    ora_string* bwt;
    ora_string* bwt_rev;                     /* NULL for FMIndex */
    uint64_t C[258];                         /* sigma+1 entries used */
    ora_sparse* sa;
} ora_index;

/* utils.h:97-129 (suffix order of the plain byte string, shorter suffix first) */
int ora_suffix_array(const uint8_t* text, uint64_t n, uint64_t* sa);
/* utils.h:145-163 */
void ora_bwt_from_sa(const uint8_t* text, uint64_t n, const uint64_t* sa, uint8_t* bwt);

/* fmindex/FMIndex.h:58-104, fmindex/BiFMIndex.h:107-167 (delimiters on, no reversed input):
 * sequences given concatenated in seqs[], seq i = seqs[seq_off[i] .. seq_off[i+1]) */
ora_index* ora_index_build(int layout, int sigma, const uint8_t* seqs, const uint64_t* seq_off, uint64_t nseq,
                           uint64_t sampling_rate, int bidirectional);
/* fmindex/FMIndex.h:30-34, fmindex/BiFMIndex.h:40-51: from BWT(s) + sampled SA description */
ora_index* ora_index_from_bwt(int layout, int sigma, const uint8_t* bwt, const uint8_t* bwt_rev, uint64_t n,
                              const uint8_t* has, const uint64_t* seq, const uint64_t* pos);
void       ora_index_free(ora_index* x);
/* spreads the pages of the occurrence tables over the NUMA nodes of `nthreads` OpenMP threads (parallel first touch of a copy); only
 * bench.py's cpu_baseline uses it, so that the all-core figure is not bound by one memory controller */
void       ora_index_spread(ora_index* x, int nthreads);

/* cursors: fmindex/FMIndexCursor.h:33-53, fmindex/BiFMIndexCursor.h:58-128, :180-190 */
typedef struct ora_cursor { uint64_t lb, lb_rev, len; } ora_cursor;
ora_cursor ora_cursor_init(const ora_index* x);
ora_cursor ora_extend_left(const ora_index* x, ora_cursor c, uint64_t symb);
ora_cursor ora_extend_right(const ora_index* x, ora_cursor c, uint64_t symb);
void       ora_extend_left_all(const ora_index* x, ora_cursor c, ora_cursor* out /* sigma */);
void       ora_extend_right_all(const ora_index* x, ora_cursor c, ora_cursor* out /* sigma */);

/* fmindex/FMIndex.h:113-124, fmindex/BiFMIndex.h:176-202 */
void ora_locate(const ora_index* x, uint64_t row, uint64_t* seq, uint64_t* pos, uint64_t* steps);

/* ---------------------------------------------------------------- searches */
typedef struct ora_hit { uint64_t qidx, lb, lb_rev, len, errors; } ora_hit;

/* search/SearchNoErrors.h:12-26 per query; out_steps (optional) = executed extensions */
void ora_search_exact(const ora_index* x, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq,
                      uint64_t* out_lb, uint64_t* out_len, uint64_t* out_steps, int nthreads);
/* the batched form, search/SearchNoErrors.h:28-86 (BatchSize cursors advanced round-robin); same results */
void     ora_search_exact_batched(const ora_index* x, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq,
                                  uint64_t* out_lb, uint64_t* out_len, int batch, int nthreads);

/* search/Backtracking.h:42-102 — hits in the reference's callback order; returns total count (may exceed cap) */
uint64_t ora_search_backtracking(const ora_index* x, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq,
                                 uint64_t max_errors, ora_hit* out, uint64_t cap, uint64_t* out_nodes);

/* search/SearchNg26.h:18-433 with Edit=false; scheme flattened [search][part].
 * max_hits_per_query = the `n` of search_n (SearchNg26.h:407-423), UINT64_MAX for unlimited.
 * per-query hit ranges are written to out_qcount (optional, nq entries); returns total hits. */
uint64_t ora_search_ng26_hamming(const ora_index* x, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq,
                                 int nsearch, int nparts, const uint64_t* pi, const uint64_t* l, const uint64_t* u,
                                 const uint64_t* partition /* nparts, or NULL = uniform per query length */,
                                 uint64_t max_hits_per_query,
                                 ora_hit* out, uint64_t cap, uint64_t* out_qcount, uint64_t* out_nodes, int nthreads);
/* search_ng26::search<Edit> (search/SearchNg26.h:18-366, :407-433) with both values of Edit: edit != 0 adds insertions / deletions
 * (the :146-218 and :286-362 branches).  For edit == 0 it must equal ora_search_ng26_hamming. */
uint64_t ora_search_ng26(const ora_index* x, int edit, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq,
                         int nsearch, int nparts, const uint64_t* pi, const uint64_t* l, const uint64_t* u,
                         const uint64_t* partition, uint64_t max_hits_per_query,
                         ora_hit* out, uint64_t cap, uint64_t* out_qcount, uint64_t* out_nodes, int nthreads);

/* ---------------------------------------------------------------- search schemes (flattened [search][part]) */
int  ora_scheme_h2(uint64_t N, uint64_t minK, uint64_t K, uint64_t* pi, uint64_t* l, uint64_t* u);           /* search_scheme/generator/h2.h:128-153; returns #searches (K+1), parts = N */
int  ora_scheme_pigeon_opt(uint64_t minK, uint64_t K, uint64_t* pi, uint64_t* l, uint64_t* u);              /* generator/pigeon.h:54-102; parts = K+1 */
int  ora_scheme_pigeon_trivial(uint64_t minK, uint64_t K, uint64_t* pi, uint64_t* l, uint64_t* u);          /* generator/pigeon.h:14-52 */
int  ora_scheme_backtracking(uint64_t N, uint64_t minK, uint64_t K, uint64_t* pi, uint64_t* l, uint64_t* u); /* generator/backtracking.h:14-21 */
/* search/SearchNg21.h:205-240 (search, search_n) over an expanded scheme: nsearch rows of `len` entries */
uint64_t ora_search_ng21(const ora_index* x, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq,
                         int nsearch, uint64_t len, const uint64_t* pi, const uint64_t* l, const uint64_t* u,
                         uint64_t max_hits_per_query, ora_hit* out, uint64_t cap, uint64_t* out_qcount, uint64_t* out_nodes);
void ora_uniform_partition(uint64_t parts, uint64_t total, uint64_t* out);                                    /* search_scheme/expand.h:324-343 */
/* search_scheme/expand.h:146-165: expands every search to newLen parts, drops invalid ones; returns #searches kept */
int  ora_scheme_expand(int nsearch, uint64_t parts, const uint64_t* pi, const uint64_t* l, const uint64_t* u,
                       uint64_t newLen, uint64_t* opi, uint64_t* ol, uint64_t* ou);
void ora_scheme_limit_to_hamming(int nsearch, uint64_t parts, uint64_t* l, uint64_t* u);                      /* expand.h:301-319 */
int  ora_scheme_is_valid(int nsearch, uint64_t parts, const uint64_t* pi, const uint64_t* l, const uint64_t* u);      /* isValid.h:55-93 */
int  ora_scheme_is_complete(int nsearch, uint64_t parts, const uint64_t* pi, const uint64_t* l, const uint64_t* u,
                            uint64_t minK, uint64_t maxK);                                                     /* isComplete.h:69-84 */
double ora_scheme_node_count_hamming(int nsearch, uint64_t parts, const uint64_t* l, const uint64_t* u, uint64_t sigma); /* nodeCount.h:19-57 */

#ifdef __cplusplus
}
#endif
#endif

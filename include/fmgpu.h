/* fmgpu.h — C-ABI of libfmgpu.so, the MI355X (gfx950) backward-search engine.
 *
 * This is the drop-in boundary for the reference's batched search path.  The reference
 * (SGSSGene/fmindex-collection, header-only C++23) has no FFI; the interface this ABI replaces is
 * the set of free function templates and member functions cited at each entry point below
 * (paths relative to src/fmindex-collection/ in the reference).  The C++ mirror of the reference's
 * template API that calls this ABI lives in include/fmc_gpu.hpp; INTEGRATION.md shows the binding
 * a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative fmgpu_status; fmgpu_last_error() returns a
 *     thread-local message for the last failure on the calling thread; no C++ exception crosses
 *     the ABI.
 *   - data pointers may be host or device pointers (detected per pointer with
 *     hipPointerGetAttributes); host buffers are staged through HBM by the library, device buffers
 *     are used in place.  `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *     Calls that receive host output buffers return after the results have landed; calls that
 *     only touch device buffers are asynchronous on `stream`.
 *   - an index handle is immutable after creation and may be used concurrently from several host
 *     threads (each call brings its own stream / buffers).
 *   - symbols are ranks in [0, sigma) exactly as in the reference (0 = sequence delimiter);
 *     row indices, interval bounds and counts are uint64_t like the reference's size_t.
 *   - row width: an index of fewer than 2^32 - 64 rows is held in 32-bit device tables and may use every optional accelerator table below;
 *     a larger one (up to 2^40 rows; the reference switches to a 64-bit suffix array at 2^31 rows, utils.h:243-247) is held in 64-bit-row
 *     tables: construction, exact search (with the interval and walk tables of fmgpu_index_accelerate_exact, 16-byte entries), search_ng26
 *     (Hamming and edit distance; equal-length Hamming batches on the lean kernel like 32-bit rows), search_ng21, search_backtracking, locate,
 *     cursor steps, String_c queries and the index file work on it; the multi-symbol-step table, the k-mismatch tables (LF^1..3, walk, prefix),
 *     the locate answer table and the one-word transport forms return FMGPU_ERR_UNSUPPORTED (fmgpu_index_row_bits tells which).
 *   - derived occurrence tables: beside the layout it is handed, an index keeps what its kernels read fastest, built on the device at creation,
 *     construction and load and counted in device_bytes — sigma = 5: a symbol-pair table (1 byte per row; exact search takes two symbols per
 *     step; both row widths) and, for a BiFMIndex with 32-bit rows, dense DNA blocks (0.5 byte per row and direction; the equal-length k-mismatch kernel); a Wavelet
 *     or EPR / EPRV2 bwt with 6 <= sigma <= 29: a symbol-plane table (2 bytes per row; exact search takes one memory line per step and interval end instead of
 *     one per tree level); a sigma = 5 string handed over as InterleavedEPR* / InterleavedEPRV2* blocks or as a Wavelet: the one-symbol block table every
 *     other DNA layout is held in (1 byte per row), and with it the two tables above — every layout searches at the same speed.
 *     Results do not depend on them (fmgpu_set_option: FMGPU_OPT_PAIR_TABLE / _DENSE_DNA / _SYMBOL_PLANES / _EXPAND_DNA = 0 keep them out).
 *   - the library reads no environment variable: what used to be FMGPU_* switches are options set through fmgpu_set_option.
 */
#ifndef FMGPU_H
#define FMGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMGPU_ABI_VERSION 6

typedef enum fmgpu_status {
    FMGPU_OK = 0,
    FMGPU_ERR_INVALID = -1,      /* bad argument (null pointer, sigma out of range, sizes that do not match the layout) */
    FMGPU_ERR_UNSUPPORTED = -2,  /* valid request this build cannot serve (e.g. an accelerator table on a 64-bit-row index, n >= 2^40 rows) */
    FMGPU_ERR_HIP = -3,          /* a HIP runtime call failed; message holds hipGetErrorString */
    FMGPU_ERR_NO_DEVICE = -4,    /* no gfx950 device visible */
    FMGPU_ERR_CAPACITY = -5,     /* result buffer too small; *out_count holds the required record count */
    FMGPU_ERR_NOMEM = -6
} fmgpu_status;

/* occurrence-table ("string with rank support") layouts of the reference, string/<file>.h */
typedef enum fmgpu_layout {
    FMGPU_IB8 = 0, FMGPU_IB16 = 1, FMGPU_IB32 = 2, FMGPU_IB16A = 3, /* InterleavedBitvector.h:168-173 */
    FMGPU_IBP16 = 4,                                                 /* InterleavedBitvectorPrefix.h:204-209 */
    FMGPU_EPR8 = 5, FMGPU_EPR16 = 6, FMGPU_EPR32 = 7,                /* InterleavedEPR.h:222-227 */
    FMGPU_EPRV2_8 = 8, FMGPU_EPRV2_16 = 9, FMGPU_EPRV2_32 = 10,      /* InterleavedEPRV2.h:289-309 */
    FMGPU_WAVELET = 11,                                              /* Wavelet.h:27-28 over bitvector::Bitvector */
    FMGPU_EPRV3_8 = 12, FMGPU_EPRV3_16 = 13, FMGPU_EPRV3_32 = 14,    /* EPRV3.h:263-272 */
    FMGPU_EPRV4 = 15,                                                /* EPRV4.h:14 */
    FMGPU_EPRV5 = 16,                                                /* EPRV5.h:14 */
    FMGPU_IEPRV7 = 17,                                               /* InterleavedEPRV7.h:15 */
    FMGPU_FBV_64_64K = 18, FMGPU_FBV_512_64K = 19, FMGPU_FBV_2048_64K = 20   /* FlattenedBitvectors2L.h:274-279; _512_64k is FMIndex's default String (fmindex/FMIndex.h:14) */
} fmgpu_layout;

/* One reference String object, described by the arrays it already holds in host memory.
 * Blocked layouts: `blocks` = String::blocks.data() (sizeof(Block) stride),
 *                  `super_blocks` = String::superBlocks.data() ([k][sigma] uint64).
 * Wavelet:         `nodes` = bit_ceil(sigma) node descriptors = Wavelet::bitvector[i].{superblocks,blocks,bits,totalLength}.
 * EPRV3/4/5/7:     `blocks` = String::bits.data() (one InBits per 64 rows; V7: the packed {bits, level0} structs),
 *                  `super_blocks` = String::superBlocks.data(), `levels[]` = the counter arrays bottom-up:
 *                  EPRV3 {blocks_}, EPRV4 {level0, level1, level2}, EPRV5 {level0, level1}, InterleavedEPRV7 {NULL, level1}.
 * FlattenedBitvectors2L: `blocks` = String::bits.data() (bitct bitsets of l1_bits per block), `super_blocks` = String::l0.data()
 *                  ([k][sigma+1] uint64, n_super_blocks = k), `levels[0]` = String::l1.data() ([k][sigma+1] uint16). */
typedef struct fmgpu_wavelet_node {
    const uint64_t* superblocks; uint64_t n_superblocks;   /* bitvector/Bitvector.h:31 */
    const uint8_t*  blocks;      uint64_t n_blocks;        /* :32 */
    const uint64_t* bits;        uint64_t n_bits;          /* :33 */
    uint64_t total_length;                                 /* :34 */
} fmgpu_wavelet_node;

typedef struct fmgpu_string_desc {
    int32_t  layout;              /* fmgpu_layout */
    int32_t  sigma;               /* String::Sigma, 2..256 */
    uint64_t n;                   /* String::size() */
    const void*     blocks;       uint64_t blocks_bytes;
    const uint64_t* super_blocks; uint64_t n_super_blocks;
    const fmgpu_wavelet_node* nodes; uint64_t n_nodes;
    const void* levels[3];        uint64_t level_bytes[3];
} fmgpu_string_desc;

/* suffixarray::SparseArray<std::tuple<uint32_t,uint32_t>, Bitvector2L<512,65536>> (suffixarray/SparseArray.h:31-76):
 * presence bitvector (bitvector/Bitvector2L.h:30-33) + two bit-packed DenseVectors (DenseVector.h:26-34). */
typedef struct fmgpu_dense_vector_desc {
    const uint64_t* data; uint64_t n_words;
    uint64_t bit_count; uint32_t bits; uint64_t largest_value; uint64_t common_divisor;
} fmgpu_dense_vector_desc;

typedef struct fmgpu_sparse_array_desc {
    uint64_t n;                                   /* rows (== string n) */
    const uint64_t* l0;   uint64_t n_l0;
    const uint16_t* l1;   uint64_t n_l1;
    const uint64_t* bits; uint64_t n_bit_words;   /* 8 words per 512-bit block */
    fmgpu_dense_vector_desc field[2];             /* documents.data[0] = seqId, data[1] = pos */
} fmgpu_sparse_array_desc;

/* FMIndex (fmindex/FMIndex.h:21-23) or BiFMIndex (fmindex/BiFMIndex.h:31-35) */
typedef struct fmgpu_index_desc {
    fmgpu_string_desc bwt;
    const fmgpu_string_desc* bwt_rev;             /* NULL => unidirectional FMIndex */
    const uint64_t* C;                            /* sigma+1 entries */
    const fmgpu_sparse_array_desc* annotated_array; /* NULL => locate unavailable */
} fmgpu_index_desc;

typedef struct fmgpu_index* fmgpu_index_t;

/* one reported cursor: search/SearchNg26.h:398-403 delegate(qidx, cursor, errors).  Order: the search kernels emit records in no particular
 * order; inside a read, ascending (errors >> 8, seq) is the reference's callback order — `seq` is either the position of the report within its
 * read, or (search_ng26 with <= 2 substitutions / <= 3 edit errors and search_ng21, where idle lanes take over subtrees of a large read) the
 * low 32 bits of a path key whose upper bits sit in errors[8..31].  fmgpu_hits_sort orders the records and leaves seq = the dense callback
 * position, errors = the error count; a caller that reads raw records takes the error count from errors & 0xff. */
typedef struct fmgpu_hit {
    uint64_t qidx, lb, lb_rev, len;
    uint32_t errors, seq;
} fmgpu_hit;

/* search_scheme::Scheme flattened [search][part] (search_scheme/Search.h:19-27) */
typedef struct fmgpu_scheme {
    int32_t n_searches, n_parts;
    const uint64_t* pi; const uint64_t* l; const uint64_t* u;
    const uint64_t* partition;   /* n_parts entries, or NULL = createUniformPartition per query length (expand.h:324-343) */
    int32_t edit;                /* 0: Hamming distance (search_ng26::search<false>); != 0: edit distance (search<true>, the reference's default) */
    int32_t reserved;
} fmgpu_scheme;

/* an EXPANDED scheme, one {pi, l, u} entry per query symbol (search_scheme/expand.h:146-165), flattened [search][length]:
 * what search_ng21 walks (search/SearchNg21.h:184-200 prepare_reorder) */
typedef struct fmgpu_expanded_scheme {
    int32_t  n_searches, reserved;
    uint64_t length;             /* entries per search = symbols of a query that are searched */
    const uint64_t* pi; const uint64_t* l; const uint64_t* u;
} fmgpu_expanded_scheme;

typedef struct fmgpu_stats {
    uint64_t lf_steps;       /* exact search: executed extensions;  k-mismatch: visited nodes (cursor extensions) */
    uint64_t hits;           /* records produced */
    float    kernel_ms;      /* duration of the dominant kernel alone, measured with hipEvents on `stream` (0 if not requested) */
    float    prepass_ms;     /* k-mismatch searches: host-measured duration of the pass that orders the batch's hand-out (reads of high-copy repeats first: a flag
                                kernel, a sample read-back, a partition) when it ran, else 0 — part of a call's wall time, never of kernel_ms (same size and
                                offset as the `reserved` word of ABI 4) */
    /* what the dominant kernel actually asked of the memory system on the index tables, counted by the kernel itself (0 for kernels that do
     * not count): table_bytes = sum over issued table loads of the entry bytes consumed (a 12-byte block entry, an 8-byte walk entry, a
     * 64-byte block of an extend-all, a 16-byte frame ...), table_accesses = number of such accesses that can each touch a different
     * memory line (two loads into the same 64-byte block count once).  Query and result traffic is coalesced and not included. */
    uint64_t table_bytes;
    uint64_t table_accesses;
    uint64_t table_steps;    /* exact search from an interval table in front of the pair table: LF steps that the entries stood for (entries read x their symbols); else 0 */
} fmgpu_stats;

/* Library options: process-wide, read when a handle is created / a call starts (set them before, not during, the calls they concern).
 * The first seven and FMGPU_OPT_BUCKET_ROWS choose what a handle holds or how a batch / a construction is prepared — results never depend on them; FORCE_WIDE, KERNEL_SELECT and FAIL_SCRATCH are test hooks. */
typedef enum fmgpu_option {
    FMGPU_OPT_PAIR_TABLE = 0,      /* 1 (default): a sigma = 5 bwt gets the symbol-pair table (exact search takes two symbols per step) */
    FMGPU_OPT_DENSE_DNA = 1,       /* 1: both strings of a sigma = 5 BiFMIndex with 32-bit rows get dense DNA blocks (equal-length k-mismatch kernel) */
    FMGPU_OPT_SYMBOL_PLANES = 2,   /* 1: a Wavelet / EPR / EPRV2 bwt with 6 <= sigma <= 29 gets the symbol-plane table (one line per LF step and end) */
    FMGPU_OPT_EXPAND_DNA = 3,      /* 1: sigma = 5 strings handed over as EPR / EPRV2 blocks or as a Wavelet are expanded into the one-symbol block table at creation */
    FMGPU_OPT_LF_TABLE = 4,        /* 1: the explicit LF mapping is built at creation (fmgpu_index_accelerate_lf adds / drops it later) */
    FMGPU_OPT_FUSED_LOCATE = 5,    /* 1: the sampled suffix array's presence bits are fused into the sigma <= 5 blocks (one line per locate step) */
    FMGPU_OPT_HEAVY_FIRST = 6,     /* 1: k-mismatch batches are handed out with the reads of high-copy repeats first */
    FMGPU_OPT_FORCE_WIDE = 7,      /* test hook, 0: 1 = every new handle is held in 64-bit-row tables whatever its size */
    FMGPU_OPT_KERNEL_SELECT = 8,   /* test / A-B hook, 0: FMGPU_SEL_* bits — which of several result-identical kernels serves a call */
    FMGPU_OPT_FAIL_SCRATCH = 9,    /* test hook, 0: k = the k-th allocation of the next per-thread call scratch fails */
    FMGPU_OPT_BUCKET_ROWS = 10,    /* 0 (default): the bucketed suffix sorters cut their buckets as large as the free device memory allows; k > 0: k rows per bucket at most */
    FMGPU_OPT_SUFFIX_SORTER = 11,  /* which suffix sorter fmgpu_build_index uses (results do not depend on it).  0 (default): by the memory each needs —
                                    * 1: all suffixes at once: suffix array + rank array + keys of all rows (30 / 42 bytes per row with 32- / 64-bit rows beside the text);
                                    * 2: bucket by bucket with the inverse suffix array as rank array (6 / 10 bytes per row beside the text + one bucket + ~60 bytes per row that shares
                                    *    its first 12-21 symbols with another): prefix doubling on the ties, the suffix array itself is never held;
                                    * 3: bucket by bucket without any array of n entries (2 bytes per row + one bucket): ties are broken by reading further symbols, a text with
                                    *    very long exact repeats is refused (FMGPU_ERR_UNSUPPORTED) */
    FMGPU_OPT_COUNT_ = 12
} fmgpu_option;
/* bits of FMGPU_OPT_KERNEL_SELECT: each takes a call off the kernel the library would pick (the parity tests run every kernel through them) */
#define FMGPU_SEL_GENERAL_DFS      (1 << 1)   /* search_ng26 / ng21: the general kernels (k_scheme, k_scheme_edit, k_ng21) */
#define FMGPU_SEL_NO_PREFIX_TABLE  (1 << 2)
#define FMGPU_SEL_NO_LF3           (1 << 3)   /* no LF^1..3 walk table */
#define FMGPU_SEL_NO_LF_GENERAL    (1 << 4)   /* no LF table in the general kernels */
#define FMGPU_SEL_NO_WALK_TABLE    (1 << 5)
#define FMGPU_SEL_NO_LENGTH_BUCKETS (1 << 6)
#define FMGPU_SEL_EXACT_ON_TREE    (1 << 21)  /* exact search on the wavelet levels although the symbol-plane table exists (k_exact_m) */
#define FMGPU_SEL_EXACT_ONE_SYMBOL (1 << 22)  /* exact search in one-symbol steps although the pair table exists (k_exact_a) */
#define FMGPU_SEL_LOCATE_PER_LANE  (1 << 23)  /* locate with one row per lane (k_locate_fused) instead of the quad-cooperative kernel */
#define FMGPU_SEL_NO_SHARING       (1 << 24)  /* no work sharing between the lanes of a wave */
#define FMGPU_SEL_NO_EXACT_LUT     (1 << 25)  /* exact search does not start from the interval table in front of the pair table */
#define FMGPU_SEL_NO_BOARD         (1 << 26)  /* no work sharing between the waves of a launch (the lanes of a wave still share) */
#define FMGPU_SEL_LEAN_FORMAT_A    (1 << 29)  /* k_scheme_lean on the one-symbol blocks although dense DNA blocks exist */
#define FMGPU_SEL_NO_LEAN          (1 << 30)  /* k_scheme_fast<PLAIN> instead of k_scheme_lean */
#define FMGPU_SEL_ALL (FMGPU_SEL_GENERAL_DFS | FMGPU_SEL_NO_PREFIX_TABLE | FMGPU_SEL_NO_LF3 | FMGPU_SEL_NO_LF_GENERAL | FMGPU_SEL_NO_WALK_TABLE | FMGPU_SEL_NO_LENGTH_BUCKETS | \
                       FMGPU_SEL_EXACT_ON_TREE | FMGPU_SEL_EXACT_ONE_SYMBOL | FMGPU_SEL_LOCATE_PER_LANE | FMGPU_SEL_NO_SHARING | FMGPU_SEL_NO_EXACT_LUT | FMGPU_SEL_NO_BOARD | FMGPU_SEL_LEAN_FORMAT_A | FMGPU_SEL_NO_LEAN)
int         fmgpu_set_option(int32_t option, int64_t value);
int         fmgpu_get_option(int32_t option, int64_t* value);

int         fmgpu_abi_version(void);
const char* fmgpu_last_error(void);
int         fmgpu_device_count(int* count);
int         fmgpu_set_device(int device);   /* hipSetDevice for the calling thread.  A handle lives on the device that was current when it was created;
                                               calls on it must be made with that device current (one process per GPU needs a single call at start-up) */

/* index upload: copies (and re-lays out for HBM) the arrays; the caller keeps ownership of host memory.
 * replaces: FMIndex(span bwt, SparseArray) fmindex/FMIndex.h:30-34, BiFMIndex(...) fmindex/BiFMIndex.h:40-51 */
int fmgpu_index_create(const fmgpu_index_desc* desc, fmgpu_index_t* out);
int fmgpu_index_destroy(fmgpu_index_t h);
int fmgpu_index_info(fmgpu_index_t h, uint64_t* n, int32_t* sigma, int32_t* layout, int32_t* bidirectional, uint64_t* device_bytes);
int fmgpu_index_row_bits(fmgpu_index_t h, int32_t* bits);   /* 32 or 64: the width of the device tables this index is held in */
/* what the handle's bwt is held in at this moment (FMGPU_FMT_* bits): tells which kernel serves a search — e.g. exact search takes the pair table when
 * FMGPU_FMT_PAIRS is set, the symbol planes when FMGPU_FMT_PLANES is set and the handle has no one-symbol blocks, the interval / k-step / walk tables when any exists */
#define FMGPU_FMT_BLOCKS     (1u << 0)   /* one-symbol block table (the InterleavedBitvector* layouts, or the expansion of another layout) */
#define FMGPU_FMT_PAIRS      (1u << 1)   /* symbol-pair table */
#define FMGPU_FMT_DENSE      (1u << 2)   /* dense DNA blocks */
#define FMGPU_FMT_PLANES     (1u << 3)   /* symbol-plane table */
#define FMGPU_FMT_TREE       (1u << 4)   /* multi-ary wavelet tree (a Wavelet string) */
#define FMGPU_FMT_REFERENCE  (1u << 5)   /* EPR / EPRV2 blocks read in place */
#define FMGPU_FMT_LF         (1u << 6)   /* explicit LF mapping */
#define FMGPU_FMT_KSTEP      (1u << 7)   /* multi-symbol-step table */
#define FMGPU_FMT_INTERVALS  (1u << 8)   /* interval table of fmgpu_index_accelerate_exact */
#define FMGPU_FMT_WALK       (1u << 9)   /* walk tables */
#define FMGPU_FMT_PREFIX     (1u << 10)  /* prefix table of fmgpu_index_accelerate_search */
#define FMGPU_FMT_LOCATE     (1u << 11)  /* locate answer table */
#define FMGPU_FMT_FUSED      (1u << 12)  /* presence bits of the sampled suffix array fused into the blocks */
int fmgpu_index_formats(fmgpu_index_t h, uint32_t* mask);

/* The library's own index file — replaces saveIndex / loadIndex (fmindex/diskStorage.h:12-27) for a handle of this library: a header, a description of
 * the handle and every device array as it sits in HBM, each with a checksum; include_tables != 0 also stores whatever optional tables the handle
 * holds at that moment (LF, k-step, interval, walk, prefix, locate tables, Format A expansion), so that a process start costs one read and one copy per
 * array instead of a suffix sort and the table construction.  fmgpu_index_load creates the handle on the calling thread's current device; a file
 * that is truncated, damaged (checksums), of another format / ABI version or byte order, or whose description does not fit its own n / sigma / layouts (every
 * array size is checked against what creation would allocate, before anything is allocated) is refused with an error code and nothing is created.
 * NOT the reference's cereal format: its byte layout for the mmser members cannot be pinned without a reference-written file. */
int fmgpu_index_save(fmgpu_index_t h, const char* path, int32_t include_tables);
int fmgpu_index_load(const char* path, fmgpu_index_t* out);
/* A copy of the handle on the calling thread's CURRENT device: every array the handle holds (optional tables included) travels device to device (hipMemcpyPeer: over
 * xGMI, no host copy), the derived tables are rebuilt there.  SURVEY 8e: "upload once to GPU0 then hipMemcpyPeer rather than 8 PCIe uploads"; fmgpu_replicas_load uses it. */
int fmgpu_index_clone(fmgpu_index_t h, fmgpu_index_t* out);

/* The explicit LF mapping (one word per row and direction: LF(row) = C[s] + rank(row, s) of the row's own symbol s): one-load one-row
 * search nodes and locate steps, and what the walk tables are built from.  Built at creation unless FMGPU_OPT_LF_TABLE is 0; enable = 0 drops it (the walk tables must have been dropped before), enable != 0 builds it.  Without it the index is the
 * bit-packed occurrence table alone (GRCh38: 3.1 GB per direction).  Results are unchanged. */
int fmgpu_index_accelerate_lf(fmgpu_index_t h, int32_t enable);

/* Optional accelerator for fmgpu_search_exact: a k-symbol-step occurrence table (one table entry advances a cursor by `kstep`
 * symbols, so a query touches 1/kstep as many HBM lines).  Built on the device from the index itself; needs
 * (sigma-1)^kstep <= 255 contexts and 16 * (sigma-1)^kstep / 64 bytes per row of HBM (DNA, kstep 3: 16 B/row).  Results of every search
 * stay identical; kstep = 1 removes the k-step table.  Same idea as the reference's BiFMIndexKStep (fmindex/BiFMIndexKStep.h).
 * For InterleavedEPR* / InterleavedEPRV2* / Wavelet indices any kstep >= 1 first expands the occurrence table on the device into the
 * one-line-per-LF-step block format the InterleavedBitvector* layouts are held in (12 * sigma / 64 bytes per row and direction; a
 * Wavelet step otherwise touches bit_width(sigma-1) lines); every search kernel then reads that table, fmgpu_string_query keeps
 * answering from the native layout.  kstep = 0 removes the k-step table and the expansion. */
int fmgpu_index_accelerate(fmgpu_index_t h, int32_t kstep);   /* (all fmgpu_index_accelerate* calls modify the handle: not concurrently with searches on it) */

/* fmgpu_index_accelerate plus two more optional tables for fmgpu_search_exact (results unchanged):
 *   lut_len > 0: the interval of every string of `lut_len` symbols ((sigma-1)^lut_len entries of 8 bytes; DNA, 12 symbols: 134 MB) — a query
 *                starts from the entry of its last lut_len symbols instead of lut_len wide-interval steps;
 *   walk != 0:   per row LF^J and the J symbols met on the way, J = 32 / bit_width(sigma-2) (DNA: 16 symbols, protein: 6; 8 bytes per row):
 *                once the interval is one row, J query symbols are checked and consumed with one load;
 *   walk >= 2:   additionally LF^(2J) and the 2J symbols (12 bytes per row): 32 bp / 12 aa per load while that many symbols remain.
 * 64-bit-row indices: kstep <= 1 (the multi-symbol-step table holds 32-bit counts), every entry of the interval and walk tables is 16 bytes. */
int fmgpu_index_accelerate_exact(fmgpu_index_t h, int32_t kstep, int32_t lut_len, int32_t walk);

/* Optional accelerators for fmgpu_search_scheme on a BiFMIndex (results unchanged):
 *   prefix_len > 0: table of the bidirectional SA interval of every string of `prefix_len` symbols ((sigma-1)^prefix_len entries of 16 bytes; DNA,
 *                   11 symbols: 67 MB; 16 symbols: 69 GB, at most 2^32 entries) — the always-exact first part of a search (u[0] = 0, search_scheme/generator/h2.h) starts from its entry;
 *   walk & 1:       per row and direction LF, LF^2, LF^3 (12 bytes): a cursor of one row advances up to three symbols per load;
 *   walk & 2:       per row and direction LF^J and the J symbols met (8 bytes, J = 32 / bit_width(sigma-2)): with 2-bit symbols (sigma <= 5) a
 *                   one-row cursor advances 16 symbols per load wherever 16 steps of a search go in one direction. */
int fmgpu_index_accelerate_search(fmgpu_index_t h, int32_t prefix_len, int32_t walk);

/* Optional accelerator for fmgpu_locate (results unchanged): every row is located once and its (seqId, pos, steps) answer kept,
 * 12 bytes per row — one load per located row instead of ~samplingRate/2 LF steps with a presence-bit probe each.  enable = 0 drops it. */
int fmgpu_index_accelerate_locate(fmgpu_index_t h, int32_t enable);

/* String_c batch evaluation (string/concepts.h:25-87): what[i] selects 0 = rank(idx,symb), 1 = prefix_rank(idx,symb),
 * 2 = symbol(idx); which = 0 -> bwt, 1 -> bwtRev */
int fmgpu_string_query(fmgpu_index_t h, int which, const uint64_t* idx, const uint8_t* symb, const uint8_t* what,
                       uint64_t count, uint64_t* out, void* stream);

/* search_no_errors::search (search/SearchNoErrors.h:12-26 per query / :28-86 batched): out_lb/out_len = cursor after the
 * last executed extension (len == 0: no occurrence) */
int fmgpu_search_exact(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq,
                       uint64_t* out_lb, uint64_t* out_len, fmgpu_stats* stats, void* stream);

/* the same search, each cursor as ONE word  lb << 32 | len  (32-bit-row indices only): the 8-byte-per-read form in which a rank's
 * intervals travel to the gathering rank ("only an RCCL gather of the resulting SA intervals") */
int fmgpu_search_exact_packed(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq,
                              uint64_t* out_interval, fmgpu_stats* stats, void* stream);

/* A profile of the same search: out_depth[q] = query symbols consumed until the cursor holds at most one row (0 rows included), or
 * length + 1 if it still holds several rows at the end.  (Tells how much of a batch the one-row walk tables can serve; bench.py reports
 * the distribution for each text.) */
int fmgpu_search_exact_depth(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, uint32_t* out_depth, void* stream);

/* Cursor steps, batched: FMIndexCursor / BiFMIndexCursor::extendLeft(symb), extendRight(symb) (fmindex/FMIndexCursor.h:33-37,
 * fmindex/BiFMIndexCursor.h:113-128) for `count` cursors {lb, lb_rev, len}, or — symb == NULL — extendLeft() / extendRight() over all
 * symbols (FMIndexCursor.h:38-53, BiFMIndexCursor.h:58-82): then every cursor yields sigma cursors, out[i * sigma + c].
 * direction 0 = left, 1 = right (BiFMIndex only).  lb_rev / out_lb_rev may be NULL for a unidirectional FMIndex.
 * symbolLeft / symbolRight of a cursor (BiFMIndexCursor.h:180-190) are fmgpu_string_query(what = 2) on bwt at lb / on bwtRev at lb_rev. */
int fmgpu_cursor_extend(fmgpu_index_t h, int32_t direction, uint64_t count, const uint64_t* lb, const uint64_t* lb_rev, const uint64_t* len,
                        const uint8_t* symb, uint64_t* out_lb, uint64_t* out_lb_rev, uint64_t* out_len, void* stream);

/* search_ng26::search<Edit=false>(index, queries, scheme, partition, delegate, n) (search/SearchNg26.h:426-433);
 * BiFMIndex only.  max_hits_per_query = n (UINT64_MAX = unlimited).  Records are appended in no particular order across
 * queries; fmgpu_hits_sort restores the reference's callback order (see fmgpu_hit).  *out_count = records produced (also when > capacity). */
int fmgpu_search_scheme(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq,
                        const fmgpu_scheme* scheme, uint64_t max_hits_per_query,
                        fmgpu_hit* out, uint64_t capacity, uint64_t* out_count, fmgpu_stats* stats, void* stream);

/* search_ng21::search(index, queries, search_scheme, delegate) and search_n(..., n, delegate) (search/SearchNg21.h:205-240):
 * edit-distance search over an expanded scheme; BiFMIndex only.  max_hits_per_query = n (UINT64_MAX = search).  Queries shorter than
 * scheme->length (which the reference would read out of bounds) produce nothing; of longer ones the first `length` symbols' positions
 * pi[] are searched, as in the reference.  Records as for fmgpu_search_scheme; errors <= 127.  search_best / search_best_n
 * (:242-293) are host loops over this call (first scheme of a list with any hit), see the host mirrors. */
int fmgpu_search_ng21(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq,
                      const fmgpu_expanded_scheme* scheme, uint64_t max_hits_per_query,
                      fmgpu_hit* out, uint64_t capacity, uint64_t* out_count, fmgpu_stats* stats, void* stream);

/* search_backtracking::search(index, queries, maxErrors, delegate) (search/Backtracking.h:85-89); FMIndex or BiFMIndex */
int fmgpu_search_backtracking(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq,
                              uint64_t max_errors, fmgpu_hit* out, uint64_t capacity, uint64_t* out_count,
                              fmgpu_stats* stats, void* stream);

/* FMIndex::locate / BiFMIndex::locate (fmindex/FMIndex.h:113-124, fmindex/BiFMIndex.h:176-202), one SA row per entry:
 * out_seq/out_pos = sampled entry reached, out_steps = LF steps walked (text position = pos + steps, locate.h:46-56) */
int fmgpu_locate(fmgpu_index_t h, const uint64_t* rows, uint64_t count,
                 uint64_t* out_seq, uint64_t* out_pos, uint64_t* out_steps, fmgpu_stats* stats, void* stream);

/* the 16-byte transport form of hit records (what a rank sends to the gathering rank): out[2k] = qidx:32 | lb:32,
 * out[2k+1] = len:32 | errors:8 | seq:24; lb_rev is dropped (it only serves further extension of the cursor).  Needs qidx, lb, len < 2^32,
 * errors < 256, seq < 2^24; a record that does not fit makes the call return FMGPU_ERR_UNSUPPORTED (checked on the device; the call
 * synchronises `stream`). */
int fmgpu_hits_pack16(const fmgpu_hit* hits, uint64_t count, uint64_t* out, void* stream);

/* the 24-byte transport form, for records as the search kernels emit them (not yet in callback order): out[3k] = qidx:32 | lb:32,
 * out[3k+1] = len:32 | errors:32, out[3k+2] = lb_rev:32 | seq:32 — the whole record, order key included, for rows and read numbers below 2^32
 * (FMGPU_ERR_UNSUPPORTED otherwise; the call synchronises `stream`). */
int fmgpu_hits_pack24(const fmgpu_hit* hits, uint64_t count, uint64_t* out, void* stream);

/* Puts `count` hit records (host or device memory) into the reference's callback order — ascending qidx, inside a query the order the
 * delegate is called in (search/SearchNg26.h:385-390) — with a stable device radix sort on (qidx, errors >> 8, seq), and normalises them:
 * seq = position of the record within its query, errors = the error count (upper bits cleared). */
int fmgpu_hits_sort(fmgpu_hit* hits, uint64_t count, void* stream);

/* GPU index construction from sequences — replaces FMIndex(Sequences, samplingRate, threads) (fmindex/FMIndex.h:58-104) and
 * BiFMIndex(Sequences, samplingRate, threads) (fmindex/BiFMIndex.h:107-167), i.e. libsais (utils.h:97-129) + the String /
 * SparseArray constructors.  Sequence i = seqs[seq_off[i] .. seq_off[i+1]); a 0 delimiter follows every sequence
 * (utils.h:382-411).  `layout` names the reference String type being replaced: Wavelet is held as wavelet lines, every blocked layout as the LF-ready block table (the answers of a String_c do not depend on its layout).
 * keep_host != 0 additionally returns host copies of the by-products through fmgpu_built_get:
 *   part 0 BWT bytes, 1 BWT of the reversed text (BiFMIndex), 2 C (sigma+1 u64), 3 l0, 4 l1, 5 presence bits,
 *   6 / 7 DenseVector words of seqId / pos, 8 {bitCount, bits, largestValue, commonDivisor} x 2 (u64). */
typedef struct fmgpu_built* fmgpu_built_t;
int fmgpu_build_index(const uint8_t* seqs, const uint64_t* seq_off, uint64_t nseq, int32_t sigma, int32_t layout,
                      uint64_t sampling_rate, int32_t bidirectional, int32_t keep_host,
                      fmgpu_index_t* out, fmgpu_built_t* built);
int fmgpu_built_get(fmgpu_built_t b, int32_t part, const void** ptr, uint64_t* bytes);
int fmgpu_built_free(fmgpu_built_t b);

/* ---- one index on several GPUs of a node, for a caller that is ONE process (SURVEY 8b `fmgpu_set_devices`, 8e): the index file is read ONCE, onto the first listed
 * device, and copied from there to the others device to device (fmgpu_index_clone; the file is read again for a device the copy fails on;
 * ndev <= 0: every visible device; a device may be listed twice), a batch is cut into contiguous ranges of queries, every replica searches its range
 * on its own device from its own host thread and writes into its range of the caller's HOST arrays (queries are independent and the index is read-only: there
 * is no exchange between replicas; ranks of a multi-process job gather with RCCL instead — bench.py).  Results equal the single-handle calls' on the same
 * batch (hit records: the same set, query numbers of the whole batch; fmgpu_hits_sort orders them).  stats: sums, kernel_ms / prepass_ms = the slowest replica's.
 * All buffers must be host memory (FMGPU_ERR_INVALID otherwise).  fmgpu_replicas_info: count, the device of each replica, the first replica's handle (borrowed). */
typedef struct fmgpu_replicas* fmgpu_replicas_t;
int fmgpu_replicas_load(const char* path, const int32_t* devices, int32_t ndev, fmgpu_replicas_t* out);
int fmgpu_replicas_destroy(fmgpu_replicas_t r);
int fmgpu_replicas_peer_copies(fmgpu_replicas_t r, int32_t* count);   /* replicas that were made by a device-to-device copy of the first one (the others read the file) */
int fmgpu_replicas_info(fmgpu_replicas_t r, int32_t* count, int32_t* devices, int32_t capacity, fmgpu_index_t* first);
int fmgpu_replicas_search_exact(fmgpu_replicas_t r, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, uint64_t* out_lb, uint64_t* out_len, fmgpu_stats* stats);
int fmgpu_replicas_search_scheme(fmgpu_replicas_t r, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, const fmgpu_scheme* scheme, uint64_t max_hits_per_query,
                                 fmgpu_hit* out, uint64_t capacity, uint64_t* out_count, fmgpu_stats* stats);
int fmgpu_replicas_search_ng21(fmgpu_replicas_t r, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, const fmgpu_expanded_scheme* scheme, uint64_t max_hits_per_query,
                               fmgpu_hit* out, uint64_t capacity, uint64_t* out_count, fmgpu_stats* stats);
int fmgpu_replicas_locate(fmgpu_replicas_t r, const uint64_t* rows, uint64_t count, uint64_t* out_seq, uint64_t* out_pos, uint64_t* out_steps, fmgpu_stats* stats);

/* device memory helpers for callers that keep queries / results resident in HBM */
int fmgpu_malloc(void** ptr, uint64_t bytes);
int fmgpu_free(void* ptr);
int fmgpu_memcpy_h2d(void* dst, const void* src, uint64_t bytes);
int fmgpu_memcpy_d2h(void* dst, const void* src, uint64_t bytes);
int fmgpu_synchronize(void* stream);

#ifdef __cplusplus
}
#endif
#endif

// fmgpu_common.h — internal to libfmgpu.so: device formats, occurrence-table accessors, host helpers.
//
// Row width.  Every translation unit with kernels is compiled twice: FMGPU_WIDE=0 (namespace fmgpu32, idx_t = uint32_t, n < 2^32 - 64 rows,
// all accelerator tables) and FMGPU_WIDE=1 (namespace fmgpu64, idx_t = uint64_t, n < 2^40 rows: the plain occurrence tables, exact search,
// search_ng26<Hamming>, search_backtracking, locate, construction).  The reference is size_t throughout and switches to a 64-bit suffix array at
// n >= 2^31 (utils.h:243-247).  fmgpu_abi.hip holds the extern "C" entry points and picks the namespace from the handle (IndexHeader::wide).
// The ABI speaks uint64_t like the reference's size_t either way.
//
// Device formats (HBM layout).
//
//  Format A  ("LF-ready interleaved bitvectors", built from the reference's InterleavedBitvector* /
//             InterleavedBitvectorPrefix* arrays by k_convert_ib):
//     one block per 64 rows, entry c of block B at byte  B*bstride + 12*c :
//         u32 cnt   = C[c] + #{ j < 64B : s[j] == c }          (C folded in: LF needs no second table)
//         u64 bits  = bit k set  <=>  s[64B + k] == c           (row p <-> bit p&63 of block p>>6)
//     sigma <= 5: bstride = 64 (one HBM line per block, 4 spare bytes); otherwise bstride = 12*sigma.
//     LF(i, c) = cnt + popc(bits & lowmask(i & 63)),  rank(i, c) = LF(i, c) - C[c].
//     Wide rows: cnt is relative to the block's super-block (2^30 rows); super[(i >> 30) * sigma + c] (u64, a few hundred bytes, cache
//     resident) holds the rest:  LF(i, c) = super + cnt + popc(...).
//     Fused presence bits (sigma <= 5, the bwt of an index with a sampled suffix array): the 8 bytes of entry 0's bitmap hold the SparseArray's
//     presence bits of the block's 64 rows instead (ViewA::fused) — locate reads "is this row sampled", the row's symbol and its LF from ONE
//     line per step.  The delimiter's own LF is not lost: the ranks of all symbols at i add up to i, so
//         LF(i, 0) = i + (C[1] + ... + C[sigma-1]) - (LF(i, 1) + ... + LF(i, sigma-1))      (ViewA::ksum holds the sum of C)
//     and a row holds the delimiter iff no other symbol's bit is set.  Every reader of Format A goes through OccA, which applies this.
//     The reference stores u16 counts relative to a 65 536-row super-block plus a u64 super-block table
//     (string/InterleavedBitvector.h:13-60) and shifts rows by one bit (row p <-> bit (p+1)&63 of block
//     (p+1)>>6); both are normalised away at upload, results are identical.
//
//  Format D  ("dense DNA": sigma = 5 strings of a BiFMIndex with 32-bit rows, beside Format A — what the k-mismatch kernel on the plain index reads):
//     one block of 32 bytes per 64 rows:  u32 occ[4] = C[c] + #{ j < 64B : s[j] == c } for c = 1..4;  u64 p0, p1 = bit r: bit 0 / bit 1 of (s[64B + r] - 1).
//     The depth-first kernels are bound by the number of vector-memory instructions a node costs (16 bytes per lane and instruction at most; one more
//     access per interval end, same line: 107 -> 140 ms): two per end here, three from Format A.  A delimiter row is written as code 0 and would count as an
//     'A'; the few rows that hold one (one per sequence) are listed — `dense_ex`, ascending — and blocks that may hold one are marked in a 2048-bit filter
//     (block number mod 2048): a marked block takes a slow path that consults the list.  Built when the string has at most 256 delimiters.
//
//  Format P  ("pairs": the bwt of a sigma = 5 index, beside Format A — what exact search reads, two symbols per step):
//     one line of 128 bytes per 128 rows:  u32 cnt[16] = (first row of the interval of "xy") + #{ j < 128L : the pair of row j is (x, y) }, index
//     (x-1)*4 + (y-1);  then for rows 0..63 and for rows 64..127 four u64 planes = bit k of the pair code of each row.  The pair of row j is
//     (s[LF(j)], s[j]) — the two symbols in front of suffix j — so with lb' = cnt[xy] + (rows of the pair before lb) a search prepends "xy" in ONE
//     step:  LF_x(LF_y(i)) = C[x] + rank_x(C[y]) + #{ j < i : pair(j) = (x, y) }  (LF keeps the order of the rows of one symbol).  An exact search
//     is bound by the random 128-byte line fills it causes (tools/membench.hip: 52-55 G lines/s whatever is read of a line), one per step and
//     interval end; this format halves the lines of a query.  Rows whose pair holds a delimiter (two per sequence) carry code 0, are left out of
//     the counts and are listed (`pairs_ex`, ascending) behind a filter on the line number; built when there are at most 512 of them.
//     64-bit rows (n < 2^38): the line's counts are relative to its super-block of 2^30 rows, `pairs_super[sb][16]` holds the interval start + the count before it.
//
//  Format S  ("symbol planes": the bwt of an index with 6 <= sigma <= 29 that arrived as a Wavelet or as EPR / EPRV2 blocks, beside Format M / R — what exact search reads there, one line per LF step and end):
//     one line of 128 bytes per 64 rows:  u64 plane[5] = bit k of each row's symbol;  then, bit-packed into the 88 bytes behind them, sigma numbers of
//     B = flat_count_bits(sigma) = min(28, 704 / sigma) bits (sigma = 28: 25, sigma = 29: 24) = the rows before the line that hold symbol c, counted from the start of the
//     line's super-block of 2^B rows (two LDS words and one funnel shift);  flat_super[sb][c] = C[c] + the rows holding c before super-block sb (24 * sigma + 320 <= 1024
//     bits: sigma <= 29).  LF(i, c) = flat_super[i >> B][c] + count[c] + popcount(rows of the line below i whose five plane bits spell c).  The multi-ary wavelet
//     tree takes two lines per step and end for sigma = 28 (two levels); the super table (6.7 KB at 2 x 10^9 rows, 19 KB as 5-byte entries at 4.5 x 10^9) is staged in LDS.  2 bytes per row.
//
//  Format R  (reference layout as is — InterleavedEPR*, InterleavedEPRV2*): blocks + superBlocks copied verbatim.
//
//  Format W  (the reference's binary wavelet tree, one 64-byte line per 384 node bits; built from Wavelet::bitvector[*] at upload and only
//     kept until Format M has been derived from it): { u64 hdr0 = ones before the line (within the node); u64 hdr1 = cum[1..5], 9 bits each;
//     u64 bits[6] }; nodes concatenated, line offset of node k in node_base[k].
//
//  Format M  (multi-ary wavelet tree — what a Wavelet string is searched in): the bit_width(sigma-1) symbol bits are cut from the top into
//     digits of 3, 3, 2 bits (5 bits: 3 + 2; the reference's own two-level form is string/MultiaryWavelet.h:27-33).  Level l has one node
//     per value of the digits above it, holding the level's digit of every symbol with that prefix in text order.  A node is an array of
//     blocks of 64 positions:  { u32 cnt[2^d] = occurrences of each digit value before the block (within the node; wide rows: relative to
//     the 2^30-position super-block);  u64 plane[d] = bit k of the digit of the 64 positions }  — 64 bytes for d = 3, 32 for d = 2, 16 for
//     d = 1: one memory line per node rank, two (sigma = 28) instead of five dependent lines per LF step and end.
//     rank_v(i) = cnt[v] + popc(AND_k (plane[k] ^ ~bit_k(v)) & lowmask(i & 63)).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/fmgpu.h"

#ifndef FMGPU_WIDE
#define FMGPU_WIDE 0
#endif
#if FMGPU_WIDE
#define FMGPU_NS fmgpu64
#else
#define FMGPU_NS fmgpu32
#endif

// ====================================================================================================================== shared (width-independent)
namespace fmgpu {

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
int hip_fail(hipError_t e, const char* what);
const char* last_error_cstr();

#define FM_HIP(call)                                                   \
    do {                                                               \
        hipError_t e_ = (call);                                        \
        if (e_ != hipSuccess) return ::fmgpu::hip_fail(e_, #call);     \
    } while (0)

// after a `<<<>>>`: a rejected launch configuration is reported by hipGetLastError only (hipDeviceSynchronize stays silent)
#define FM_LAUNCHED(what)                                              \
    do {                                                               \
        hipError_t e_ = hipGetLastError();                             \
        if (e_ != hipSuccess) return ::fmgpu::hip_fail(e_, what);      \
    } while (0)

bool is_device_pointer(const void* p);

// grid of 256-thread blocks covering `threads` threads.  A kernel whose domain may exceed the grid limit runs a grid-stride loop and passes the
// number of blocks it wants at most; every other launch fails with FMGPU_ERR_UNSUPPORTED instead of being rejected by the runtime.
constexpr uint64_t kMaxGridBlocks = 0x7fffffffull;
inline int grid_of(uint64_t threads, dim3* out, uint64_t cap_blocks = 0) {
    uint64_t blocks = (threads + 255) / 256;
    if (blocks == 0) blocks = 1;
    if (cap_blocks) blocks = blocks < cap_blocks ? blocks : cap_blocks;
    // (2^32 threads and more: the dispatch packet holds the grid size in 32 bits and the runtime cuts a larger launch short without an error)
    if (blocks > kMaxGridBlocks || blocks * 256 > 0xffffffffull) return fail(FMGPU_ERR_UNSUPPORTED, "a launch of " + std::to_string(threads) + " threads exceeds the grid limit");
    *out = dim3((unsigned)blocks);
    return 0;
}
#define FM_GRID(var, threads)       dim3 var; do { int rc_ = ::fmgpu::grid_of((threads), &var); if (rc_) return rc_; } while (0)

// Library options (fmgpu_set_option, include/fmgpu.h): which derived tables a new handle gets and which of several result-identical kernels serves a call.
// The shipped library reads NO environment variable: tests, bench.py and the A/B tools set options through the ABI.  Builds made with -DFMGPU_DEV
// (make DEV=1; tools/k2_*_probe.py) additionally honour the FMGPU_DEV_* environment knobs — count-only runs, per-read node dumps, tuning fields, residency
// overrides — and take FMGPU_DEV_FLAGS as the initial kernel selection, so that a stray environment variable cannot make the shipped library drop records
// or write outside a caller's buffer.
int64_t opt(int option);                      // current value of an fmgpu_option (fmgpu_abi.hip)
inline bool opt_on(int option) { return opt(option) != 0; }
inline const char* dev_env(const char* name) {
#ifdef FMGPU_DEV
    return getenv(name);
#else
    (void)name; return nullptr;
#endif
}
// bits of FMGPU_OPT_KERNEL_SELECT (FMGPU_SEL_* in include/fmgpu.h); a DEV build ORs the non-selection bits of FMGPU_DEV_FLAGS in (1 = count hits only, ...)
inline int kernel_flags() {
    int f = (int)opt(FMGPU_OPT_KERNEL_SELECT) & FMGPU_SEL_ALL;
#ifdef FMGPU_DEV
    if (const char* e = getenv("FMGPU_DEV_FLAGS")) f |= atoi(e);
#endif
    return f;
}

struct DBuf {    // RAII device allocation
    void* p = nullptr; size_t bytes = 0;
    int alloc(size_t b) {
        release(); bytes = b ? b : 8;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            p = nullptr; bytes = 0; (void)hipGetLastError();
            return fail(e == hipErrorOutOfMemory ? FMGPU_ERR_NOMEM : FMGPU_ERR_HIP, "hipMalloc(" + std::to_string(b) + " bytes): " + hipGetErrorString(e));
        }
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    void* take() { void* q = p; p = nullptr; bytes = 0; return q; }     // hands the allocation over (to a handle)
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
    ~DBuf() { release(); }
    DBuf() = default;
    DBuf(const DBuf&) = delete; DBuf& operator=(const DBuf&) = delete;
};

// A temporary device mirror of a caller buffer: device pointers are used in place, host pointers are staged.
struct Staged {
    void* dev = nullptr;
    void* host = nullptr;
    size_t bytes = 0;
    bool owned = false, writeback = false;
    hipStream_t stream = nullptr;
    int in(const void* p, size_t nbytes, hipStream_t s);      // read-only input
    int out(void* p, size_t nbytes, hipStream_t s);           // output (copied back by finish())
    int finish();                                             // D2H of outputs (synchronises when anything was staged)
    ~Staged();
};

struct Built {   // host copies of construction by-products (fmgpu_build_index with keep_host)
    std::vector<std::vector<uint8_t>> part;
};

// first member of both widths' Index: what the extern "C" layer needs to route a handle
constexpr uint32_t kIndexMagic = 0x464d4758u;   // "FMGX"
struct IndexHeader { uint32_t magic = kIndexMagic; int32_t wide = 0; int32_t device = 0; };

// rows below this are indexed with 32-bit device tables (FMGPU_FORCE_WIDE=1 sends every new index to the 64-bit build: a test knob)
constexpr uint64_t kNarrowLimit = 0xffffffffull - 64;
constexpr uint64_t kWideLimit = 1ull << 40;
bool want_wide(uint64_t n);

// per host thread and device: small device buffers and events that every search call needs (allocating and freeing them per call costs more
// than the bookkeeping they serve — hipFree synchronises the device).  `ctr` serves calls that report stats (they synchronise before
// returning, so it is idle between calls), `sink` the others (never read).
constexpr uint32_t kCounterStripes = 64;      // step counters of the one-thread-per-query kernels are striped (a single word would serialise one atomic per wave)
constexpr uint32_t kCounterKinds = 4;         // [0] steps, [1] table bytes, [2] table accesses, [3] steps served by interval-table entries — kCounterStripes words each
struct CallScratch {
    unsigned long long* ctr = nullptr; unsigned long long* sink = nullptr; unsigned long long* len2 = nullptr;
    unsigned long long* pinned = nullptr;                         // host side of the small read-backs (a pageable target costs a staging copy each)
    void* frames = nullptr; size_t frames_bytes = 0;              // frame stacks of the DFS kernels, kept between calls up to kFrameCache bytes
    void* dfs_ctr = nullptr;                                      // their Counters (a DFS call synchronises before it returns: one at a time per thread)
    void* order = nullptr; size_t order_bytes = 0;                // hand-out order of a batch and the workspace that makes it (heavy reads first), kept between calls
    void* board = nullptr; uint32_t board_slots = 0;              // the work boards of the depth-first kernels (sharing between the waves of a launch), made on first use: one per concurrent launch
    hipStream_t dfs_streams[8] = {};                              // a ragged batch: the launches of its read lengths run side by side on these (made on first use)
    hipEvent_t dfs_events[9] = {};
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    void drop();
};
int call_scratch(CallScratch** out);          // the calling thread's scratch for its current device (created on first use; all-or-nothing)

}  // namespace fmgpu

// ====================================================================================================================== per row width
namespace FMGPU_NS {
using namespace fmgpu;

#if FMGPU_WIDE
using idx_t = uint64_t;
#else
using idx_t = uint32_t;
#endif
constexpr bool kWide = FMGPU_WIDE != 0;
// entries of the exact-search tables (fmgpu_index_accelerate_exact).  32-bit rows: interval {lb, len} 8 bytes, walk {LF^J row, code} 8 bytes,
// double walk {LF^2J row, code, code} 12 bytes.  64-bit rows: 16 bytes each — {lb, len} as two u64; {row lo, row hi, code, 0}; {row lo, row hi, code, code};
// a walk that meets a delimiter has row = all ones.
constexpr size_t kSlutEntryBytes = kWide ? 16 : 8, kWalkEntryBytes = kWide ? 16 : 8, kWalk2EntryBytes = kWide ? 16 : 12;
constexpr uint32_t kSuperShift = 30;          // wide rows: counts are kept relative to super-blocks of 2^30 rows / node positions
constexpr idx_t kNoRow = ~(idx_t)0;

enum Family : int { FAM_A = 0, FAM_EPR = 1, FAM_EPRV2 = 2, FAM_WAVELET = 3 };

// ------------------------------------------------------------------ device-side views (kernel arguments)
struct ViewA {            // Format A
    const uint8_t* blk;
    uint32_t bstride;
    uint32_t sigma;
    const idx_t* C;       // sigma+1 entries (device)
    const uint64_t* super;// wide rows only: [row >> 30][sigma]
    uint32_t fused;       // 1: entry 0's bitmap holds the sampled suffix array's presence bits (see above)
    idx_t ksum;           // C[1] + ... + C[sigma-1] (mod 2^width), for LF(i, 0) of a fused table
};

struct ViewR {            // Format R (EPR / EPRV2 reference layout)
    const uint8_t* blk;
    const uint64_t* super;
    const idx_t* C;
    uint32_t stride, bits_off, bt, sigma, bitct;
    uint32_t rows;        // rows per block (EPR: 64 / bitct, EPRV2: 64)
    uint32_t period_shift;// EPRV2: log2(rows per super-block), EPR: 0 (uses period)
    uint32_t period;      // EPR: rows per super-block (not a power of two)
    uint64_t maskEven, bitMask;
};

struct ViewW {            // Format W (transient)
    const uint64_t* lines;      // 8 u64 per line
    const uint32_t* node_base;  // line offset per node
    const idx_t* C;
    uint32_t sigma, bitct;
};

constexpr int kMaxLevelsM = 3;
constexpr int kMaxNodesM = 1 + 8 + 64;
struct LevelM { uint32_t bits, shift, stride, first_node; };     // digit = (c >> shift) & ((1 << bits) - 1); node = first_node + (c >> (shift + bits))
struct ViewM {            // Format M
    const uint8_t* data;
    const uint64_t* node_off;   // byte offset of every node's block array (kMaxNodesM entries at most)
    const idx_t* C;
    uint32_t sigma, bitct, nlevels, nnodes;
    LevelM lv[kMaxLevelsM];
    const uint64_t* super;      // wide rows only: counts of the super-blocks, [node_super[node] + (pos >> 30)][8]
    const uint32_t* node_super; // wide rows only: first super-block row of every node
};
// digits of a symbol of `bitct` bits, from the top
inline void digits_of(uint32_t bitct, uint32_t* d, uint32_t* n) {
    static const uint8_t tab[9][3] = {{1, 0, 0}, {1, 0, 0}, {2, 0, 0}, {3, 0, 0}, {2, 2, 0}, {3, 2, 0}, {3, 3, 0}, {3, 2, 2}, {3, 3, 2}};
    *n = 0;
    for (int k = 0; k < 3; ++k) if (tab[bitct][k]) d[(*n)++] = tab[bitct][k];
}

// ------------------------------------------------------------------ small device helpers
__device__ __forceinline__ uint64_t lowmask(uint32_t k) { return (1ull << k) - 1ull; }   // k in [0, 63]
__device__ __forceinline__ uint32_t popc64(uint64_t v) { return (uint32_t)__popcll(v); }

struct EntryA { idx_t cnt; uint64_t bits; };

__device__ __forceinline__ EntryA load_entry_a(const ViewA& v, idx_t i, uint32_t c) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(v.blk + (size_t)(i >> 6) * v.bstride + c * 12u);
    EntryA e;
    uint32_t a = p[0], b = p[1], d = p[2];
    e.cnt = a;
    if constexpr (kWide) e.cnt += v.super[(size_t)(i >> kSuperShift) * v.sigma + c];
    e.bits = (uint64_t)b | ((uint64_t)d << 32);
    return e;
}

// Format A entry writer shared by the converters and the builder: `total` = C[c] + occurrences before the block, `base` = the same for the first
// block of the block's super-block (wide rows: the entry keeps total - base, the first block of a super-block publishes base)
__device__ __forceinline__ void put_entry_count_a(uint8_t* out, uint64_t* super, uint64_t B, uint32_t c, uint32_t sigma, uint32_t bstride, uint64_t total, uint64_t base) {
    uint32_t* o = reinterpret_cast<uint32_t*>(out + B * bstride + 12ull * c);
    if constexpr (kWide) {
        o[0] = (uint32_t)(total - base);
        if ((B & ((1ull << (kSuperShift - 6)) - 1ull)) == 0) super[(B >> (kSuperShift - 6)) * sigma + c] = total;
    } else o[0] = (uint32_t)total;
}
__device__ __forceinline__ void put_entry_a(uint8_t* out, uint64_t* super, uint64_t B, uint32_t c, uint32_t sigma, uint32_t bstride, uint64_t total, uint64_t base, uint64_t bits) {
    put_entry_count_a(out, super, B, c, sigma, bstride, total, base);
    uint32_t* o = reinterpret_cast<uint32_t*>(out + B * bstride + 12ull * c);
    o[1] = (uint32_t)bits; o[2] = (uint32_t)(bits >> 32);
}
__device__ __forceinline__ uint64_t super_first_block(uint64_t B) { return (B >> (kSuperShift - 6)) << (kSuperShift - 6); }

// ------------------------------------------------------------------ occurrence-table accessors
// Every Occ exposes:
//   lf2(a, b, c, &ra, &rb)      LF(a,c), LF(b,c)   (= rank + C[c]) with both ends' loads issued together
//   rank(i, c), prefix_rank(i, c), symbol(i)
//   all2(a, b, lfa[], lfb[])    LF for every symbol at both ends (extend-all)
//   sigma()
template <int SIGMA>   // SIGMA > 0: compile-time alphabet, 0: runtime
struct OccA {
    ViewA v;
    static constexpr int kMaxSigma = SIGMA > 0 ? SIGMA : 256;
    __device__ __forceinline__ uint32_t sigma() const { return SIGMA > 0 ? (uint32_t)SIGMA : v.sigma; }

    // LF(i, 0) of a fused table: i + sum C - sum over the other symbols (each one entry of the same block)
    __device__ __forceinline__ idx_t lf0_fused(idx_t i) const {
        idx_t r = i + v.ksum;
        const uint32_t s = sigma();
        for (uint32_t c = 1; c < s; ++c) { EntryA e = load_entry_a(v, i, c); r -= e.cnt + popc64(e.bits & lowmask((uint32_t)i & 63u)); }
        return r;
    }
    __device__ __forceinline__ void lf2(idx_t a, idx_t b, uint32_t c, idx_t& ra, idx_t& rb) const {
        if (v.fused && c == 0) { ra = lf0_fused(a); rb = lf0_fused(b); return; }
        EntryA ea = load_entry_a(v, a, c);
        EntryA eb = load_entry_a(v, b, c);
        ra = ea.cnt + popc64(ea.bits & lowmask((uint32_t)a & 63u));
        rb = eb.cnt + popc64(eb.bits & lowmask((uint32_t)b & 63u));
    }
    __device__ __forceinline__ idx_t lf(idx_t i, uint32_t c) const {
        if (v.fused && c == 0) return lf0_fused(i);
        EntryA e = load_entry_a(v, i, c);
        return e.cnt + popc64(e.bits & lowmask((uint32_t)i & 63u));
    }
    __device__ __forceinline__ idx_t rank(idx_t i, uint32_t c) const { return lf(i, c) - v.C[c]; }
    __device__ __forceinline__ idx_t prefix_rank(idx_t i, uint32_t c) const {
        idx_t r = 0;
        for (uint32_t d = 0; d < c; ++d) r += rank(i, d);
        return r;
    }
    __device__ __forceinline__ uint32_t symbol(idx_t i) const {
        const uint32_t s = sigma();
        uint32_t bit = (uint32_t)i & 63u;
        if (v.fused) {                                           // the delimiter row is the one no other symbol claims
            for (uint32_t c = 1; c < s; ++c) { EntryA e = load_entry_a(v, i, c); if ((e.bits >> bit) & 1ull) return c; }
            return 0;
        }
        for (uint32_t c = 0; c + 1 < s; ++c) {
            EntryA e = load_entry_a(v, i, c);
            if ((e.bits >> bit) & 1ull) return c;
        }
        return s - 1;
    }
    // symbol and LF of that symbol from one pass over the block (locate)
    __device__ __forceinline__ idx_t lf_symbol(idx_t i, uint32_t& symb) const {
        const uint32_t s = sigma();
        uint32_t bit = (uint32_t)i & 63u;
        if (v.fused) {
            for (uint32_t c = 1; c < s; ++c) {
                EntryA e = load_entry_a(v, i, c);
                if ((e.bits >> bit) & 1ull) { symb = c; return e.cnt + popc64(e.bits & lowmask(bit)); }
            }
            symb = 0;
            return lf0_fused(i);
        }
        uint32_t c = 0;
        EntryA e = load_entry_a(v, i, 0);
        while (c + 1 < s && !((e.bits >> bit) & 1ull)) { ++c; e = load_entry_a(v, i, c); }
        symb = c;
        return e.cnt + popc64(e.bits & lowmask(bit));
    }
    // all2 in two halves for SIGMA <= 5 (one 64-byte block per end): the loads alone, so that a caller can issue them next to other lanes'
    // loads of a divergent wave before anything is consumed, and the arithmetic
    __device__ __forceinline__ void load2(idx_t a, idx_t b, uint32_t* da, uint32_t* db) const {
        const uint4* pa = reinterpret_cast<const uint4*>(v.blk + (size_t)(a >> 6) * 64u);
        const uint4* pb = reinterpret_cast<const uint4*>(v.blk + (size_t)(b >> 6) * 64u);
#pragma unroll
        for (int k = 0; k < 4; ++k) { uint4 t = pa[k]; da[4 * k] = t.x; da[4 * k + 1] = t.y; da[4 * k + 2] = t.z; da[4 * k + 3] = t.w; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { uint4 t = pb[k]; db[4 * k] = t.x; db[4 * k + 1] = t.y; db[4 * k + 2] = t.z; db[4 * k + 3] = t.w; }
    }
    __device__ __forceinline__ void all2_of(const uint32_t* da, const uint32_t* db, idx_t a, idx_t b, idx_t* lfa, idx_t* lfb) const {
        const uint64_t ma = lowmask((uint32_t)a & 63u), mb = lowmask((uint32_t)b & 63u);
        constexpr uint32_t S = (uint32_t)(SIGMA > 0 ? SIGMA : 1);
#pragma unroll
        for (uint32_t c = 0; c < S; ++c) {
            lfa[c] = da[3 * c] + popc64(((uint64_t)da[3 * c + 1] | ((uint64_t)da[3 * c + 2] << 32)) & ma);
            lfb[c] = db[3 * c] + popc64(((uint64_t)db[3 * c + 1] | ((uint64_t)db[3 * c + 2] << 32)) & mb);
            if constexpr (kWide) {
                lfa[c] += v.super[(size_t)(a >> kSuperShift) * S + c];
                lfb[c] += v.super[(size_t)(b >> kSuperShift) * S + c];
            }
        }
        if (v.fused) {                                           // entry 0's bitmap holds presence bits: the delimiter's LF from the other symbols'
            idx_t ra = a + v.ksum, rb = b + v.ksum;
#pragma unroll
            for (uint32_t c = 1; c < S; ++c) { ra -= lfa[c]; rb -= lfb[c]; }
            lfa[0] = ra; lfb[0] = rb;
        }
    }
    template <int MS>
    __device__ __forceinline__ void all2(idx_t a, idx_t b, idx_t* lfa, idx_t* lfb) const {
        const uint32_t s = sigma();
        if (SIGMA > 0 && SIGMA <= 5) {
            // 64-byte block = one line: fetch it whole (4 x dwordx4); both ends' loads are issued back to back (one round trip): a wave holds 64
            // out-of-phase lanes, so "rare" second fetches would otherwise be paid by the whole wave in nearly every iteration
            uint32_t da[16], db[16];
            load2(a, b, da, db);
            all2_of(da, db, a, b, lfa, lfb);
            return;
        }
        if (MS <= 32) {
#pragma unroll
            for (uint32_t c = 0; c < (uint32_t)MS; ++c) {
                if (c < s) {
                    EntryA ea = load_entry_a(v, a, c);
                    EntryA eb = load_entry_a(v, b, c);
                    lfa[c] = ea.cnt + popc64(ea.bits & lowmask((uint32_t)a & 63u));
                    lfb[c] = eb.cnt + popc64(eb.bits & lowmask((uint32_t)b & 63u));
                }
            }
        } else {
            for (uint32_t c = 0; c < s; ++c) {
                EntryA ea = load_entry_a(v, a, c);
                EntryA eb = load_entry_a(v, b, c);
                lfa[c] = ea.cnt + popc64(ea.bits & lowmask((uint32_t)a & 63u));
                lfb[c] = eb.cnt + popc64(eb.bits & lowmask((uint32_t)b & 63u));
            }
        }
        if (v.fused) {                                           // (fused tables have sigma <= 5: this path only for a runtime-sigma instantiation)
            idx_t ra = a + v.ksum, rb = b + v.ksum;
            for (uint32_t c = 1; c < s; ++c) { ra -= lfa[c]; rb -= lfb[c]; }
            lfa[0] = ra; lfb[0] = rb;
        }
    }
};

// Format R: reference layout read in place.  EPR: string/InterleavedEPR.h:63-103, :154-178; EPRV2: string/InterleavedEPRV2.h:28-105, :191-213
template <bool V2>
struct OccR {
    ViewR v;
    static constexpr int kMaxSigma = 256;
    __device__ __forceinline__ uint32_t sigma() const { return v.sigma; }

    __device__ __forceinline__ uint32_t cnt(size_t b, uint32_t c) const {
        const uint8_t* p = v.blk + b * v.stride + c * v.bt;
        if (v.bt == 2) return *reinterpret_cast<const uint16_t*>(p);
        if (v.bt == 1) return *p;
        return *reinterpret_cast<const uint32_t*>(p);
    }
    __device__ __forceinline__ uint64_t word(size_t b, uint32_t k) const {
        return *reinterpret_cast<const uint64_t*>(v.blk + b * v.stride + v.bits_off + 8u * k);
    }
    __device__ __forceinline__ uint64_t have(size_t b, uint32_t symb) const {   // EPRV2 symbol-match mask
        uint64_t r = ~0ull;
        for (uint32_t i = 0; i < v.bitct; ++i) {
            uint64_t inv = (~symb >> i) & 1u;
            r &= word(b, i) ^ (0ull - inv);
        }
        return r;
    }
    // EPR: one bit per slot (at bit slot*bitct) set where slot value <= symb
    __device__ __forceinline__ uint64_t le_mask(uint64_t in, uint32_t symb) const {
        uint64_t rb = 0;     // rb[symb], InterleavedEPR.h:49-61
        uint64_t mk = (uint64_t)symb | (1ull << v.bitct);
        for (uint32_t i = 0; i < 64u / v.bitct; i += 2) rb = (rb << (2 * v.bitct)) | mk;
        uint64_t te = ((rb - (in & v.maskEven)) & v.bitMask) >> v.bitct;
        uint64_t to = (rb - ((in >> v.bitct) & v.maskEven)) & v.bitMask;
        return te | to;
    }
    __device__ __forceinline__ void split(idx_t i, size_t& b, size_t& sb, uint32_t& bit) const {
        if (V2) { b = (size_t)(i >> 6); sb = v.period_shift >= 32 ? (size_t)((uint64_t)i >> 32) : (size_t)(i >> v.period_shift); bit = (uint32_t)i & 63u; }
        else    { b = (size_t)(i / v.rows); sb = (size_t)(i / v.period); bit = (uint32_t)(i % v.rows); }
    }
    __device__ __forceinline__ idx_t rank(idx_t i, uint32_t c) const {
        size_t b, sb; uint32_t bit; split(i, b, sb, bit);
        uint64_t sup = v.super[sb * v.sigma + c];
        if (V2) {
            uint64_t m = have(b, c);
            uint32_t in = bit == 0 ? 0u : popc64(m << (64u - bit));
            return (idx_t)(sup + cnt(b, c) + in);
        } else {
            uint64_t in = word(b, 0);
            uint64_t lim = (1ull << (bit * v.bitct)) - 1ull;
            uint32_t hi = popc64(le_mask(in, c) & lim);
            uint32_t lo = c == 0 ? 0u : popc64(le_mask(in, c - 1) & lim);
            return (idx_t)(sup + cnt(b, c) + hi - lo);
        }
    }
    __device__ __forceinline__ idx_t lf(idx_t i, uint32_t c) const { return rank(i, c) + v.C[c]; }
    __device__ __forceinline__ void lf2(idx_t a, idx_t b, uint32_t c, idx_t& ra, idx_t& rb) const {
        ra = lf(a, c); rb = lf(b, c);
    }
    __device__ __forceinline__ idx_t prefix_rank(idx_t i, uint32_t c) const {
        idx_t r = 0;
        for (uint32_t d = 0; d < c; ++d) r += rank(i, d);
        return r;
    }
    __device__ __forceinline__ uint32_t symbol(idx_t i) const {
        size_t b, sb; uint32_t bit; split(i, b, sb, bit);
        if (V2) {
            uint32_t s = 0;
            for (uint32_t k = v.bitct; k > 0; --k) s = (s << 1) | (uint32_t)((word(b, k - 1) >> bit) & 1ull);
            return s;
        }
        return (uint32_t)((word(b, 0) >> (bit * v.bitct)) & ((1ull << v.bitct) - 1ull));
    }
    __device__ __forceinline__ idx_t lf_symbol(idx_t i, uint32_t& symb) const { symb = symbol(i); return lf(i, symb); }
    template <int MS>
    __device__ __forceinline__ void all2(idx_t a, idx_t b, idx_t* lfa, idx_t* lfb) const {
        for (uint32_t c = 0; c < v.sigma && c < (uint32_t)MS; ++c) { lfa[c] = lf(a, c); lfb[c] = lf(b, c); }
    }
};

// Format W: string/Wavelet.h:77-102 over one-line node ranks (only used to read the symbols back when Format M is derived from uploaded node arrays)
struct OccW {
    ViewW v;
    __device__ __forceinline__ idx_t node_rank(uint32_t id, idx_t i, uint32_t* bit_out) const {
        const idx_t line = i / 384u; const uint32_t r = (uint32_t)(i - line * 384u);
        const uint32_t k = r >> 6, part = r & 63u;
        const uint64_t* L = v.lines + ((size_t)v.node_base[id] + (size_t)line) * 8u;
        const uint64_t h0 = L[0], h1 = L[1];
        const uint64_t w = L[2 + k];
        const uint32_t cum = k ? (uint32_t)(h1 >> (9u * (k - 1u))) & 0x1ffu : 0u;
        if (bit_out) *bit_out = (uint32_t)((w >> part) & 1ull);
        return (idx_t)h0 + cum + popc64(w & lowmask(part));
    }
    __device__ __forceinline__ uint32_t symbol(idx_t i) const {
        uint32_t s = 0;
        for (uint32_t b = 0; b < v.bitct; ++b) {
            uint32_t id = ((1u << b) - 1u) + s;
            uint32_t bit;
            idx_t r = node_rank(id, i, &bit);
            s = (s << 1) | bit;
            i = bit ? r : i - r;
        }
        return s;
    }
};

// Format M: the multi-ary wavelet tree
struct OccM {
    ViewM v;
    static constexpr int kMaxSigma = 256;
    __device__ __forceinline__ uint32_t sigma() const { return v.sigma; }

    __device__ __forceinline__ const uint8_t* block(const LevelM& L, uint32_t node, idx_t i) const {
        return v.data + v.node_off[node] + (size_t)(i >> 6) * L.stride;
    }
    static __device__ __forceinline__ uint64_t match(const uint64_t* pl, uint32_t bits, uint32_t val) {
        uint64_t m = ~0ull;
        for (uint32_t k = 0; k < bits; ++k) m &= pl[k] ^ (0ull - (uint64_t)((~val >> k) & 1u));
        return m;
    }
    __device__ __forceinline__ idx_t super_of(uint32_t node, idx_t i, uint32_t val) const {
        if constexpr (kWide) return (idx_t)v.super[((size_t)v.node_super[node] + (size_t)(i >> kSuperShift)) * 8u + val];
        else return 0;
    }
    __device__ __forceinline__ idx_t node_rank(const LevelM& L, uint32_t node, idx_t i, uint32_t val) const {
        const uint8_t* b = block(L, node, i);
        const uint32_t cnt = reinterpret_cast<const uint32_t*>(b)[val];
        const uint64_t* pl = reinterpret_cast<const uint64_t*>(b + (4u << L.bits));
        return super_of(node, i, val) + cnt + popc64(match(pl, L.bits, val) & lowmask((uint32_t)i & 63u));
    }
    __device__ __forceinline__ idx_t rank(idx_t i, uint32_t c) const {
        for (uint32_t l = 0; l < v.nlevels; ++l) {
            const LevelM L = v.lv[l];
            i = node_rank(L, L.first_node + (c >> (L.shift + L.bits)), i, (c >> L.shift) & ((1u << L.bits) - 1u));
        }
        return i;
    }
    __device__ __forceinline__ idx_t lf(idx_t i, uint32_t c) const { return rank(i, c) + v.C[c]; }
    __device__ __forceinline__ void lf2(idx_t a, idx_t b, uint32_t c, idx_t& ra, idx_t& rb) const {
        for (uint32_t l = 0; l < v.nlevels; ++l) {        // both ends descend the same node path: the two dependent chains interleave
            const LevelM L = v.lv[l];
            const uint32_t node = L.first_node + (c >> (L.shift + L.bits)), val = (c >> L.shift) & ((1u << L.bits) - 1u);
            const idx_t x = node_rank(L, node, a, val), y = node_rank(L, node, b, val);
            a = x; b = y;
        }
        ra = a + v.C[c]; rb = b + v.C[c];
    }
    __device__ __forceinline__ idx_t prefix_rank(idx_t i, uint32_t c) const {     // symbols < c among the first i
        idx_t acc = 0;
        if (c >= (1u << v.bitct)) return i;                                       // (c == 2^bitct == sigma: every symbol is smaller)
        for (uint32_t l = 0; l < v.nlevels; ++l) {
            const LevelM L = v.lv[l];
            const uint32_t node = L.first_node + (c >> (L.shift + L.bits)), val = (c >> L.shift) & ((1u << L.bits) - 1u);
            for (uint32_t u = 0; u < val; ++u) acc += node_rank(L, node, i, u);   // same prefix, smaller digit (one block: the same line)
            i = node_rank(L, node, i, val);
        }
        return acc;
    }
    __device__ __forceinline__ idx_t lf_symbol(idx_t i, uint32_t& symb) const {
        uint32_t s = 0;
        for (uint32_t l = 0; l < v.nlevels; ++l) {
            const LevelM L = v.lv[l];
            const uint32_t node = L.first_node + s;
            const uint8_t* b = block(L, node, i);
            const uint64_t* pl = reinterpret_cast<const uint64_t*>(b + (4u << L.bits));
            uint64_t p[3] = {0, 0, 0};
            for (uint32_t k = 0; k < L.bits; ++k) p[k] = pl[k];
            const uint32_t bit = (uint32_t)i & 63u;
            uint32_t val = 0;
            for (uint32_t k = 0; k < L.bits; ++k) val |= (uint32_t)((p[k] >> bit) & 1ull) << k;
            const uint32_t cnt = reinterpret_cast<const uint32_t*>(b)[val];
            i = super_of(node, i, val) + cnt + popc64(match(p, L.bits, val) & lowmask(bit));
            s = (s << L.bits) | val;
        }
        symb = s;
        return i + v.C[s];
    }
    __device__ __forceinline__ uint32_t symbol(idx_t i) const { uint32_t s; (void)lf_symbol(i, s); return s; }
    template <int MS>
    __device__ __forceinline__ void all2(idx_t a, idx_t b, idx_t* lfa, idx_t* lfb) const {
        for (uint32_t c = 0; c < v.sigma && c < (uint32_t)MS; ++c) lf2(a, b, c, lfa[c], lfb[c]);
    }
};

// sampled suffix array on the device: the reference's arrays verbatim (suffixarray/SparseArray.h:31-76)
struct ViewSA {
    const uint64_t* l0; const uint16_t* l1; const uint64_t* bits;
    const uint64_t* f0; const uint64_t* f1;
    uint32_t bits0, bits1; uint64_t div0, div1;
};

__device__ __forceinline__ bool sa_present(const ViewSA& s, idx_t i) { return (s.bits[i >> 6] >> ((uint32_t)i & 63u)) & 1ull; }
__device__ __forceinline__ uint64_t sa_rank(const ViewSA& s, idx_t i) {   // bitvector/Bitvector2L.h:123-142
    uint32_t bitId = (uint32_t)i & 511u;
    const uint64_t* w = s.bits + (size_t)(i >> 9) * 8u;
    uint32_t cnt = 0;
    for (uint32_t k = 0; k < (bitId >> 6); ++k) cnt += popc64(w[k]);
    if (bitId & 63u) cnt += popc64(w[bitId >> 6] & lowmask(bitId & 63u));
    return s.l0[i >> 16] + s.l1[i >> 9] + cnt;
}
__device__ __forceinline__ uint64_t dense_access(const uint64_t* data, uint32_t bits, uint64_t div, uint64_t i) {   // DenseVector.h:154-182
    uint64_t begin = i * bits, end = begin + bits - 1;
    uint64_t s = begin >> 6, e = end >> 6;
    uint32_t off = (uint32_t)(begin & 63u);
    uint64_t mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
    uint64_t val = data[s] >> off;
    if (s != e) val |= data[e] << (64u - off);
    return (val & mask) * div;
}

// ------------------------------------------------------------------ host side
struct DevString {
    int layout = 0, family = 0, sigma = 0, bitct = 0;
    uint64_t n = 0;
    void* blk = nullptr;       // Format A blocks / Format R blocks / Format M node blocks
    void* aux = nullptr;       // Format R superBlocks / Format M node offsets (wide: + node_super + super rows)
    void* sup = nullptr;       // wide Format A: super-block counts
    size_t blk_bytes = 0, aux_bytes = 0, sup_bytes = 0;
    ViewA va{}; ViewR vr{}; ViewM vm{};
    // explicit LF mapping: lf_table[i] = C[s[i]] + rank(i, s[i])  (n entries; the symbol is recovered from C).
    // One word per row buys one-load single-row DFS nodes and one-load locate steps; fmgpu_index_accelerate_lf drops / adds it
    // (FMGPU_LF_TABLE=0: not built at creation).
    idx_t* lf_table = nullptr;
    // multi-symbol-step table (fmgpu_index_accelerate): block B, context w at  kblk + (B * kcodes + w) * 16 :
    //   { u32 cnt = LF_k(64B, w); u32 bits_lo; u32 bits_hi; u32 0 }   (bit r: the kstep symbols preceding suffix 64B+r spell w;
    //   the first 12 bytes are fetched with one dwordx3 load, like a Format A entry)
    uint8_t* kblk = nullptr; uint32_t kstep = 0, kcodes = 0; size_t kblk_bytes = 0;
    // walk table (fmgpu_index_accelerate_search): walk3[3*i .. 3*i+2] = LF(i), LF^2(i), LF^3(i)
    idx_t* walk3 = nullptr;
    // exact-search accelerators (fmgpu_index_accelerate_exact):
    //   slut[code] = {lb, len} of the backward search of the slut_len symbols with code = sum (c_t - 1) * (sigma-1)^t, c_0 = the LAST symbol;
    //   walkj[row] = {LF^J(row), sum (s_t - 1) << (walk_bits * t)} with s_1.. the symbols met (s_0 = BWT symbol of row), or {~0, 0} if a delimiter is met
    uint2* slut = nullptr; uint32_t slut_len = 0; uint64_t slut_entries = 0;
    uint2* walkj = nullptr; uint32_t walk_J = 0, walk_bits = 0;
    //   walk2j[3*row ..] = {LF^(2J)(row), code of symbols 0 .. J-1, code of symbols J .. 2J-1} (walk = 2 in fmgpu_index_accelerate_exact), or {~0, 0, 0}
    uint32_t* walk2j = nullptr;
    // Format A shadow of a Format R / M string (fmgpu_index_accelerate, kstep >= 1): the searches then read `va` (one line per
    // LF step instead of one per level); fmgpu_string_query keeps answering from the native format.
    // Format D (see the head of this file): dense DNA blocks + the ascending list of the rows that hold a delimiter (u32, `dense_nex` of them)
    void* dense = nullptr; size_t dense_bytes = 0; uint32_t* dense_ex = nullptr; uint32_t dense_nex = 0;
    // Format P (see the head of this file): pair lines of the bwt + the ascending list of the rows left out of them (u32, `pairs_nex` of them)
    uint8_t* pairs = nullptr; size_t pairs_bytes = 0; idx_t* pairs_ex = nullptr; uint32_t pairs_nex = 0;
    idx_t* pairs_super = nullptr; uint32_t pairs_nsb = 0;     // 64-bit rows: the line counts are relative to super-blocks of 2^30 rows, [pairs_nsb][16] holds the rest
    // Format S (see the head of this file): one line per 64 rows of a Wavelet bwt with 6 <= sigma <= 29 + the counts at the start of every 2^24 rows ([flat_nsb][sigma], C folded in)
    uint8_t* flat = nullptr; size_t flat_bytes = 0; idx_t* flat_super = nullptr; uint32_t flat_nsb = 0;
    void* shadow = nullptr; size_t shadow_bytes = 0;   // (shadow_bytes = blocks + super table)
    void* shadow_sup = nullptr; size_t shadow_sup_bytes = 0;
    int search_family() const { return shadow ? (int)FAM_A : family; }
};

// builds s.shadow / s.va from the string's own symbols; defined in fmgpu_build.hip
int build_format_a_shadow(DevString& s, const idx_t* dC, hipStream_t stream);
// Format M from a device array of symbols (string/Wavelet.h:61-71 restated as bulk passes); defined in fmgpu_build.hip
int make_format_m(const uint8_t* symbols, uint64_t n, uint32_t sigma, const idx_t* dC, DevString& s, int layout, hipStream_t stream);

// Suffix sorting without the suffix array (fmgpu_bucketsort.hip): the suffixes are sorted bucket by bucket (a bucket = a range of rows) and every bucket's text positions are handed to
// `sink`, in ascending row order: sink(first_row, pos, count, scratch, scratch_bytes) — scratch: device memory the sink may use until it returns (8 bytes per row of the largest bucket)
using SuffixSink = std::function<int(uint64_t first_row, const idx_t* pos, uint64_t count, void* scratch, size_t scratch_bytes)>;
int sort_suffixes_bucketed(const uint8_t* text, uint64_t n, uint32_t sigma, uint64_t bucket_rows, const SuffixSink& sink, hipStream_t stream);
// ... with the inverse suffix array as the rank array of a prefix doubling on the ties: rank[p] (n entries, device) = the row of suffix p
int sort_suffixes_isa(const uint8_t* text, uint64_t n, uint32_t sigma, uint64_t bucket_rows, idx_t* rank, hipStream_t stream);

struct Index;
// 0 if the calling thread's current device is the one the handle lives on; defined in fmgpu_index.hip
int on_handle_device(const Index* x);

// fills s.lf_table from the device string (all layouts); defined in fmgpu_index.hip
int build_lf_table(DevString& s, hipStream_t stream);
// builds Format D beside a sigma = 5 Format A string (no-op where it does not apply); defined in fmgpu_index.hip
int build_dense_dna(DevString& s, hipStream_t stream);
// moves the sampled suffix array's presence bits into the bwt's Format A blocks (sigma <= 5; see "Fused presence bits"); defined in fmgpu_index.hip
int fuse_presence_bits(Index* x, hipStream_t stream);
// builds Format P beside the bwt of a sigma = 5 index with 32-bit rows (no-op where it does not apply; adds its bytes to device_bytes); defined in fmgpu_index.hip
int build_pair_table(Index* x, hipStream_t stream);
// sigma = 5 strings that arrived in another layout's own format (EPR / EPRV2 blocks read in place, the multi-ary wavelet tree) get the Format A expansion at once, so
// that every search of the DNA path runs on the same kernels whatever layout the caller's index has (FMGPU_SHADOW=0: only on fmgpu_index_accelerate); adds nothing to
// device_bytes itself; defined in fmgpu_index.hip
int auto_shadow(Index* x, hipStream_t stream);
// Format S: bits of a count inside a line (= log2 of the rows per super-block); 28 at most, so that a super-block's scan stays short
constexpr uint32_t flat_count_bits(uint32_t sigma) { return sigma == 0u ? 28u : ((704u / sigma) < 28u ? (704u / sigma) : 28u); }
// builds Format S beside the Wavelet bwt of an index with 6 <= sigma <= 29 (no-op where it does not apply; adds its bytes to device_bytes); defined in fmgpu_index.hip
int build_flat_table(Index* x, hipStream_t stream);
void free_string(DevString& s);

struct Index {
    IndexHeader hdr;
    DevString bwt, rev;
    bool bidirectional = false, has_sa = false;
    idx_t* dC = nullptr;
    uint64_t hC[258] = {0};
    // sampled SA
    void *sa_l0 = nullptr, *sa_l1 = nullptr, *sa_bits = nullptr, *sa_f0 = nullptr, *sa_f1 = nullptr;
    size_t sa_bytes[5] = {0, 0, 0, 0, 0};          // allocation sizes of l0, l1, bits, f0, f1 (the index file stores them as they are)
    ViewSA vsa{};
    // fmgpu_index_accelerate_locate: the (seqId, pos, steps) answer of every row, 3 x u32 per row (or null)
    uint32_t* loc_tab = nullptr;
    size_t device_bytes = 0;
    // prefix table (fmgpu_index_accelerate_search): lut[code(w)] = { lb, lbRev, len, symbols consumed before the interval emptied (or L) }
    uint4* lut = nullptr; uint32_t lut_len = 0; uint64_t lut_entries = 0;
    Index() { hdr.wide = kWide ? 1 : 0; }
};

template <class F>
static int dispatch_native(const DevString& s, F&& f) {      // the string's own format (fmgpu_string_query, table builders)
    switch (s.family) {
    case FAM_A:
        if (s.sigma == 5) return f(OccA<5>{s.va}, std::integral_constant<int, 5>{});
        if (s.sigma <= 32) return f(OccA<0>{s.va}, std::integral_constant<int, 32>{});
        return f(OccA<0>{s.va}, std::integral_constant<int, 256>{});
    case FAM_EPR:
        if (s.sigma <= 32) return f(OccR<false>{s.vr}, std::integral_constant<int, 32>{});
        return f(OccR<false>{s.vr}, std::integral_constant<int, 256>{});
    case FAM_EPRV2:
        if (s.sigma <= 32) return f(OccR<true>{s.vr}, std::integral_constant<int, 32>{});
        return f(OccR<true>{s.vr}, std::integral_constant<int, 256>{});
    default:
        if (s.sigma <= 32) return f(OccM{s.vm}, std::integral_constant<int, 32>{});
        return f(OccM{s.vm}, std::integral_constant<int, 256>{});
    }
}
template <class F>
static int dispatch_occ(const DevString& s, F&& f) {         // what the searches read (the Format A shadow if there is one)
    if (s.shadow) {
        if (s.sigma == 5) return f(OccA<5>{s.va}, std::integral_constant<int, 5>{});
        if (s.sigma <= 32) return f(OccA<0>{s.va}, std::integral_constant<int, 32>{});
        return f(OccA<0>{s.va}, std::integral_constant<int, 256>{});
    }
    return dispatch_native(s, std::forward<F>(f));
}

}  // namespace FMGPU_NS

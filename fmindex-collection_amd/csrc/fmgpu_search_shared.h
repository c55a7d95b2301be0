// fmgpu_search_shared.h — what the search translation units (fmgpu_exact.hip, fmgpu_search.hip, fmgpu_locate.hip) share: wave-level helpers, the step counters
// of the one-thread-per-query kernels, query readers and LDS staging, the DFS kernels' machinery (scheme tables, path keys, symbol sets, hit rings), and the host
// helpers of the launchers.  Internal to libfmgpu.so; included once per row width like fmgpu_common.h.
#pragma once
#include "fmgpu_common.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include <hipcub/hipcub.hpp>

namespace FMGPU_NS {

typedef __attribute__((address_space(3))) uint32_t lds_word;
typedef uint32_t __attribute__((ext_vector_type(4))) flat_u32x4;

// the cursor of query q: two 64-bit arrays as the reference's cursor fields, or (out_len == nullptr) one word lb << 32 | len — the transport
// form of fmgpu_search_exact_packed (32-bit rows only)
__device__ __forceinline__ void store_interval(uint64_t* __restrict__ out_lb, uint64_t* __restrict__ out_len, uint64_t q, idx_t lb, idx_t len) {
    if (kWide || out_len) { out_lb[q] = lb; out_len[q] = len; }
    else out_lb[q] = ((uint64_t)lb << 32) | (uint64_t)len;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// counters of the one-thread-per-query kernels: [0] executed steps, [1] table bytes consumed, [2] table accesses, [3] steps that an interval-table entry stood for (fmgpu_stats), each striped
// over kCounterStripes words (a single word would serialise one atomic per wave — 156 k of them for 10 M queries — behind each other)
__device__ __forceinline__ void add_counters(unsigned long long* __restrict__ ctr, uint32_t steps, uint32_t bytes, uint32_t accesses, uint32_t table_steps = 0u) {
    const uint32_t ts = wave_sum(steps), tb = wave_sum(bytes), ta = wave_sum(accesses), tt = wave_sum(table_steps);
    if ((threadIdx.x & 63u) == 0 && ts) {
        const uint32_t stripe = blockIdx.x & (kCounterStripes - 1u);
        atomicAdd(&ctr[stripe], (unsigned long long)ts);
        atomicAdd(&ctr[kCounterStripes + stripe], (unsigned long long)tb);
        atomicAdd(&ctr[2u * kCounterStripes + stripe], (unsigned long long)ta);
        if (tt) atomicAdd(&ctr[3u * kCounterStripes + stripe], (unsigned long long)tt);
    }
}

// ---- exact search on Format A without accelerator tables --------------------------------------------------------
// The query symbols are fetched as aligned 64-bit words one word ahead of use, so that the only load on the
// dependent chain of an LF step is the occurrence-table entry; the second interval end re-uses the first end's
// entry when both fall into the same 64-row block (the common case once the interval is short).
struct QueryReader {
    const uint64_t* base;   // 8-byte aligned
    uint64_t pos;           // absolute byte position (relative to base) of the next symbol to hand out (moving down)
    uint64_t cw, nw;        // current word, next (lower) word
    __device__ __forceinline__ void init(const uint8_t* qbuf, uint64_t off, uint32_t m) {
        uint64_t mis = (uint64_t)qbuf & 7ull;
        base = reinterpret_cast<const uint64_t*>((uint64_t)qbuf - mis);
        pos = off + mis + m - 1;                       // m >= 1
        uint64_t w = pos >> 3;
        cw = base[w];
        nw = w ? base[w - 1] : 0;
    }
    __device__ __forceinline__ uint32_t next() {
        uint32_t c = (uint32_t)(cw >> ((pos & 7ull) * 8ull)) & 0xffu;
        if ((pos & 7ull) == 0) {                       // crossing into the lower word: rotate and prefetch
            uint64_t w = pos >> 3;
            cw = nw;
            nw = w >= 2 ? base[w - 2] : 0;
        }
        --pos;
        return c;
    }
};

// ------------------------------------------------------------------ DFS machinery
constexpr int kMaxParts = 16;
constexpr int kMaxSearches = 16;

struct SchemeDev {             // flattened [search][part]; values fit a byte (errors <= 255, parts <= 16)
    int S, P;
    uint8_t pi[kMaxSearches * kMaxParts], l[kMaxSearches * kMaxParts], u[kMaxSearches * kMaxParts];
    uint32_t partition[kMaxParts];   // used when uniform == 0
    uint32_t psum;                   // sum of partition[] (queries of another length are skipped)
    int uniform;
    int dev_flags;                   // dev knobs: 1 = count hits per lane only (no records)
    int use_key, sharing;            // k_scheme: hit records carry path keys / idle lanes take subtrees from the busy lanes of their wave
};

struct Counters { unsigned long long hits, nodes, next, table_bytes, table_accesses; };

// ---- work sharing between the lanes of a wave in k_scheme_fast -------------------------------------------------------------------------
// The work of a k-mismatch search is heavy-tailed on a repeat-rich text: the median read visits ~200 nodes, a read from a satellite array
// half a million (measured on the genome-like text: 0.02 % of the reads hold 13 % of all nodes), and a depth-first walk of one read by one
// lane takes as long as its node count.  A lane that has spent kShareNodes nodes on its current read therefore offers the BOTTOM frame of its
// stack — the untried siblings of its shallowest branching node, the largest piece of work it still owns — to the lanes of its wave that
// are out of work: the frame travels by lane shuffles, the staged read by an LDS column copy, no atomic and no global traffic beyond the
// three frame words.  (A device-wide task queue was tried first: one queue head for thousands of waiting waves serialised the hand-over at
// ~1.3 us per task — 2 M tasks, 8 s — and was dropped.)  This needs an order of the hit records that does not depend on who found them:
constexpr uint32_t kShareNodes = 2;
// ... and only from a read that has proven heavy: with offers from every read the hand-over ran in nearly every iteration of a wave (some lane is always
// out of work) and cost more than it returned (uniform text, plain index: 56.8 -> 39.8 ms once reads of fewer than 64 nodes stopped offering)
[[maybe_unused]] constexpr uint32_t kShareHeavy = 64;

// Path key: the callback order of the reference is the depth-first order in which every node tries its match child first and its substitution
// children in ascending symbol order (SearchNg26.h:171-218).  For hits of one read that is the lexicographic order of
//   (search, [m - step of the 1st substitution, its symbol], [m - step of the 2nd substitution, its symbol])   with "no substitution" = 0:
// a later first substitution is met earlier on the way back up.  24 bits per substitution; the key travels in fmgpu_hit::seq (low 32 bits) and
// the upper 24 bits of fmgpu_hit::errors until fmgpu_hits_sort orders the records by (qidx, key) and turns it into the dense callback index.
// the key of an ancestor that had made `e` substitutions: the fields of the later ones cleared (so frames need not carry keys)
__device__ __forceinline__ uint64_t key_prefix(uint64_t key, uint32_t e) {
    return e == 0u ? key & (0xffull << 48) : (e == 1u ? key & ~0xffffffull : key);
}
__device__ __forceinline__ uint64_t key_with(uint64_t key, uint32_t e_before, uint32_t m, uint32_t step, uint32_t symb) {
    if (e_before >= 2u) return key;
    return key | ((uint64_t)(((m - step) << 8) | symb) << (24u * (1u - e_before)));
}



// lane-interleaved frame stack: frame d of lane g at word (d * nlanes + g) of three u64 planes
struct StackView { uint64_t *p0, *p1, *p2, *p3; uint64_t nlanes; uint32_t depth; };   // frame planes (the edit-distance kernels use the block as 32-byte records)

struct Cur { idx_t lb, lbRev, len; };

__device__ __forceinline__ void emit_hit(fmgpu_hit* out, uint64_t cap, Counters* ctr, uint64_t qidx, Cur c, uint32_t e, uint32_t seq) {
    unsigned long long k = atomicAdd(&ctr->hits, 1ull);
    if (k < cap) {
        fmgpu_hit h;
        h.qidx = qidx; h.lb = c.lb; h.lb_rev = c.lbRev; h.len = c.len; h.errors = e; h.seq = seq;
        out[k] = h;
    }
}

// ---- symbol sets and children ----------------------------------------------------------------------------------
// MAXSIG <= 32: one register word, arrays stay in registers (fully unrolled selects).  MAXSIG = 256: eight words, the
// LF arrays live in scratch and are indexed dynamically (the reference itself does O(sigma) work per extend-all).
template <int MAXSIG>
struct SymSet {
    static constexpr int W = (MAXSIG + 31) / 32;
    uint32_t w[W];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < W; ++i) w[i] = 0;
    }
    __device__ __forceinline__ bool test(uint32_t s) const {
        if (s >= (uint32_t)MAXSIG) return false;
        if (W == 1) return (w[0] >> s) & 1u;
        uint32_t r = 0;
#pragma unroll
        for (int i = 0; i < W; ++i) if ((s >> 5) == (uint32_t)i) r = w[i];
        return (r >> (s & 31u)) & 1u;
    }
    __device__ __forceinline__ void remove(uint32_t s) {
        if (s >= (uint32_t)MAXSIG) return;
#pragma unroll
        for (int i = 0; i < W; ++i) if ((s >> 5) == (uint32_t)i) w[i] &= ~(1u << (s & 31u));
    }
    __device__ __forceinline__ void insert(uint32_t s) {
        if (s >= (uint32_t)MAXSIG) return;
#pragma unroll
        for (int i = 0; i < W; ++i) if ((s >> 5) == (uint32_t)i) w[i] |= 1u << (s & 31u);
    }
    __device__ __forceinline__ bool any() const {
        uint32_t r = 0;
#pragma unroll
        for (int i = 0; i < W; ++i) r |= w[i];
        return r != 0;
    }
    __device__ __forceinline__ uint32_t first() const {     // lowest member; caller checks any()
        uint32_t r = 0xffffffffu;
#pragma unroll
        for (int i = W - 1; i >= 0; --i) if (w[i]) r = (uint32_t)i * 32u + (uint32_t)__ffs((int)w[i]) - 1u;
        return r;
    }
    __device__ __forceinline__ void clear_below(uint32_t s) {   // drop members < s
#pragma unroll
        for (int i = 0; i < W; ++i) {
            uint32_t lo = (uint32_t)i * 32u;
            if (s >= lo + 32u) w[i] = 0;
            else if (s > lo) w[i] &= ~((1u << (s - lo)) - 1u);
        }
    }
};

template <int MAXSIG>
__device__ __forceinline__ SymSet<MAXSIG> alive_set(const idx_t* lfa, const idx_t* lfb, uint32_t sigma) {
    SymSet<MAXSIG> m; m.clear();
    if (MAXSIG <= 32) {
#pragma unroll
        for (uint32_t d = 0; d < (uint32_t)MAXSIG; ++d) if (d < sigma && lfb[d] != lfa[d]) m.w[0] |= 1u << d;
    } else {
        for (uint32_t d = 0; d < sigma; ++d) if (lfb[d] != lfa[d]) m.w[d >> 5] |= 1u << (d & 31u);
    }
    return m;
}

// kid cursor of symbol s from the LF values at both ends; `right` mirrors the roles (fmindex/BiFMIndexCursor.h:58-82)
template <int MAXSIG>
__device__ __forceinline__ Cur kid_of(const idx_t* lfa, const idx_t* lfb, Cur cur, uint32_t s, bool right, uint32_t sigma) {
    idx_t pre = 0, la = 0, lb = 0;
    if (MAXSIG <= 32) {
#pragma unroll
        for (uint32_t d = 0; d < (uint32_t)MAXSIG; ++d) {
            if (d < s && d < sigma) pre += lfb[d] - lfa[d];
            if (d == s) { la = lfa[d]; lb = lfb[d]; }
        }
    } else {
        for (uint32_t d = 0; d < s; ++d) pre += lfb[d] - lfa[d];
        la = lfa[s]; lb = lfb[s];
    }
    Cur k;
    k.len = lb - la;
    if (right) { k.lbRev = la; k.lb = cur.lb + pre; }
    else       { k.lb = la; k.lbRev = cur.lbRev + pre; }
    return k;
}

// ---- work sharing inside a wave: who hands a subtree to whom (used by every depth-first kernel).  idlem / offerm = ballots of the lanes that are out of work / that
// offer the bottom frame of their stack; the i-th idle lane is paired with the i-th offering lane.
struct WavePairs {
    uint64_t idlem, offerm, below; uint32_t pairs;
    __device__ __forceinline__ WavePairs(uint64_t idle_mask, uint64_t offer_mask, uint32_t lane)
        : idlem(idle_mask), offerm(offer_mask), below((1ull << lane) - 1ull), pairs((uint32_t)min(__popcll(idle_mask), __popcll(offer_mask))) {}
    __device__ __forceinline__ bool gives(bool offer) const { return offer && (uint32_t)__popcll(offerm & below) < pairs; }
    __device__ __forceinline__ bool takes(bool idle) const { return idle && (uint32_t)__popcll(idlem & below) < pairs; }
    __device__ __forceinline__ int partner(bool take) const {       // the (rank + 1)-th offering lane, for the idle lane of that rank (lane 0's for everyone else)
        uint64_t om = offerm;
        for (uint32_t t = take ? (uint32_t)__popcll(idlem & below) : 0u; t > 0; --t) om &= om - 1ull;
        return (int)__ffsll((unsigned long long)om) - 1;
    }
};

constexpr uint32_t kNoResume = 0xffffffffu;

// ---- work sharing between the WAVES of a launch: the board ---------------------------------------------------------------------------------
// Sharing inside a wave spreads a heavy read over 64 lanes.  On a repeat-rich text that is not enough for the heaviest reads: 125 k reads of the genome text took the
// edit-distance kernel 85 ms and 2 M reads 147 ms (tools/batch_scaling_probe.py) — the launch ended when the ONE wave that held a read of a satellite array (millions of
// nodes) was done, thousands of waves having left long before.  Round 2's device-wide task queue failed on its queue head: every heavy lane pushed, every idle lane popped, one
// address served millions of atomics at ~1.3 us each.  The board is driven by DEMAND instead, so that nothing touches it while every wave has work:
//   * a wave that is out of work once the batch is handed out leaves the count of working waves, asks for a batch (one fetch-add: its index) and sleeps on THAT batch's
//     header word — a line of its own — until the batch is published or the launch is over (`state` = 0: no working wave, no untaken batch; it cannot rise again).  At most
//     kBoardWaiters waves wait; the others leave (a poll is a read of memory);
//   * a working wave looks at the board every kBoardPeriod-th pass — one load — and only if some lane of it has a subtree to give; if batches are asked for it takes the
//     next index (one fetch-add), writes the bottom frames of ALL its offering lanes into that batch — up to kBoardFramesPerLane per lane, 64 subtrees in all — and publishes
//     the header; the sleeper wakes with that many lanes of work, which prove heavy in turn and give again.
// Hit records carry path keys (fmgpu_hits_sort orders them whoever found them), as for the sharing inside a wave: the board is on exactly when that sharing is.
// No co-residency is assumed: a wave that starts late finds the batch handed out, asks, sees state = 0 and leaves.
// What it took (tools/board_sweep.sh, development build): (1) polls are RELAXED loads — an acquire load invalidates the CU's L1 at every poll; (2) no fence anywhere (below);
// (3) a giver reserves with ONE fetch-add — a compare-and-swap loop cost (givers)^2 atomics once thousands of waves gave at once (7.5 x 10^7 attempts for 1.7 x 10^5 batches;
// a reservation took milliseconds, the launch 1.1 s instead of 0.11); with these, a hand-over costs the giver 2.4 us and a look 0.6 us, and a low threshold is best.
constexpr uint32_t kBoardBatches = 8192;      // ring of batches (a batch is taken at once by the wave that asked for it: one per waiting wave is ever in use)
constexpr uint32_t kBoardWords = 14;          // dwords of a task (edit distance: two 16-byte frame halves, the path key, read number, search number)
constexpr uint32_t kBoardPeriod = 8;          // passes of a working wave between two looks at the board (a look: one load; a hand-over: ~2.4 us of the giving wave)
constexpr uint32_t kBoardHeavy = 256;         // nodes a lane must have spent on its read (or subtree) before it gives to another WAVE.  Measured on the genome text, k = 2 edit distance,
                                              // kernel ms for 125 k / 1 M / 2 M reads (development build; without the board 85 / 110 / 147): heavy 8192: 92 / 101 / 124, 2048: 44 / 79 / 127,
                                              // 512: 30 / 78 / 123, 256 and 64: 25 / 77 / 121 (period 4-8; period 64: 55 / 90 / 135)
constexpr uint32_t kBoardFramesPerLane = 8;  // frames one lane gives per hand-over at most (a batch holds 64 subtrees; few offering lanes fill it from the bottoms of their stacks)
constexpr uint32_t kBoardSpinCap = 1u << 21;  // polls (~30 s) of a waiting wave per give-up period: after eight periods in which NO wave published a batch it leaves and flags the launch as
                                              // failed — a bug must not hang the card, and a long healthy launch must not be failed
constexpr uint32_t kBoardWaiters = 1024;      // waves that wait at the board at most: a poll is a read of memory (sc1), and 4096 waves polling every 3 us took the channels their headers
                                              // live on — a giver's compare-and-swap on the same pages then took milliseconds (tools/board_sweep.sh); the other idle waves leave
struct WorkBoard {
    unsigned long long state;                 // low 32 bits: waves that hold work; high 32: published batches nobody has taken yet
    unsigned long long failed;                // a waiting wave gave up (kBoardSpinCap)
    unsigned long long tasks;                 // subtrees handed over (a count for the development build's log)
    uint32_t heavy, period;                   // kBoardHeavy / kBoardPeriod of this launch (written by the host: development builds can vary them)
    uint32_t waiters, cas_tries;              // kBoardWaiters of this launch; development build: compare-and-swap attempts of the givers
    unsigned long long dev[7];                // development build: wave-cycles (s_memtime) waiting / giving / looking, their counts, waves, cycles of all waves
    unsigned long long pad0[4];
    unsigned long long ht;                    // (a line of its own) high 32 bits: batches asked for; low 32: batches published (never more than asked for)
    unsigned long long pad1[15];
    unsigned long long hdr[kBoardBatches][8]; // [.][0]: (index + 1) << 8 | tasks of the batch with that index, written last by its giver; (index + 1) << 8 | 0xff once its taker has read it;
                                              // 64 bytes apart: the waiting waves' polls spread over the memory channels
    uint32_t task[kBoardBatches][kBoardWords][64];
};
constexpr size_t kBoardResetBytes = 256 + sizeof(unsigned long long) * 8 * kBoardBatches;     // what a launch starts from zero

__device__ __forceinline__ unsigned long long wave_bcast64(unsigned long long v, int src) {
    return ((unsigned long long)__shfl((uint32_t)(v >> 32), src, 64) << 32) | __shfl((uint32_t)v, src, 64);
}
// Visibility between waves on different CUs / XCDs without a fence: every word of the board is written by an agent-scope atomic or an `sc1` (write-through) store and read by an
// agent-scope atomic or an `sc1` load; a giver drains its stores (s_waitcnt vmcnt(0)) before it stores the header, a taker loads the tasks only after its poll has matched the header
// (MI355X_MICROARCH.md, "hand-offs measured with sc1 loads in place of the acquire").  A release / acquire fence here would write back and invalidate the XCD's L2 at every hand-over:
// the first version did (__threadfence) and made the launch up to eight times SLOWER — thousands of hand-overs each flushed the cache every other wave of the XCD walks its index through.
// every wave of the launch, before it asks for its first reads (the count has arrived before the hand-out counter moves: the add returns, and is waited for)
__device__ __forceinline__ void board_enter(WorkBoard* b, uint32_t lane) {
    unsigned long long old = 0;
    if (lane == 0) old = atomicAdd(&b->state, 1ull);
    asm volatile("s_waitcnt vmcnt(0)" :: "v"((uint32_t)old) : "memory");
}
// a wave without work, the batch handed out: returns the number of tasks of the batch it was given (their dwords at task[*slot][.][0 .. k - 1]), or 0 = the launch is over
__device__ __forceinline__ uint32_t board_wait(WorkBoard* b, uint32_t lane, uint32_t* slot, uint32_t* index) {
    unsigned long long asked = 0; uint32_t stay = 0;
    if (lane == 0) {
        atomicAdd(&b->state, ~0ull);
        const unsigned long long v = __hip_atomic_load(&b->ht, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stay = (int32_t)((uint32_t)(v >> 32) - (uint32_t)v) < (int32_t)b->waiters ? 1u : 0u;      // (enough waves wait already: this one is not needed; published > asked: a batch waits for THIS wave)
        if (stay) asked = atomicAdd(&b->ht, 1ull << 32) >> 32;
    }
    if (!__shfl(stay, 0, 64)) return 0;
    const uint32_t idx = (uint32_t)wave_bcast64(asked, 0);
    *slot = idx % kBoardBatches; *index = idx;
    uint32_t quiet = 0, seen_published = 0;                        // (lane 0) give-up bookkeeping: periods of kBoardSpinCap polls in which no wave published anything
    for (uint32_t spin = 0;; ++spin) {
        unsigned long long h = 0, st = 1;
        if (lane == 0) {
            // (relaxed, never acquire: an acquire load invalidates the CU's L1 at every poll — with fifteen waiting waves beside it, the one wave that still walks the heaviest read
            // then takes every block from L2)
            h = __hip_atomic_load(&b->hdr[*slot][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((spin & 7u) == 7u) st = __hip_atomic_load(&b->state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (spin == kBoardSpinCap) {                            // ~30 s of waiting: a long launch whose waves still publish is healthy; eight such periods without ANY batch published are not
                const uint32_t pub = (uint32_t)__hip_atomic_load(&b->ht, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (pub != seen_published) { seen_published = pub; quiet = 0; } else ++quiet;
                spin = 0;
                if (quiet == 8u) { atomicAdd(&b->failed, 1ull); st = 0; }
            }
        }
        h = wave_bcast64(h, 0); st = wave_bcast64(st, 0);
        if ((h >> 8) == (unsigned long long)idx + 1ull) {
            if (lane == 0) atomicAdd(&b->state, 1ull - (1ull << 32));      // one more working wave, one batch fewer on the board — in one step
            return (uint32_t)(h & 0xffu);
        }
        if (st == 0) return 0;
        __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127);      // ~14 us between polls
    }
}
// a working wave with subtrees to give: the index of the batch it may fill, if a wave waits (wave-uniform).  ONE fetch-add, no compare-and-swap loop: in the tail of a launch
// thousands of waves give at once, and a loop that retries until it wins costs (givers)^2 atomics on one word (measured: 7.5 x 10^7 compare-and-swaps for 1.7 x 10^5 batches,
// a reservation took milliseconds).  Two givers that both saw the last request may both publish: the second batch has no waiting wave yet — the next wave that runs out of
// work asks, gets exactly that index, and finds it published (every working wave asks when it is done, and a wave never leaves while a published batch is untaken)
__device__ __forceinline__ bool board_reserve(WorkBoard* b, uint32_t lane, uint32_t* slot, uint32_t* index) {
    unsigned long long v = 0; uint32_t ok = 0;
    if (lane == 0) {
        v = __hip_atomic_load(&b->ht, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int32_t)((uint32_t)(v >> 32) - (uint32_t)v) > 0) {    // asked > published
            atomicAdd(&b->state, 1ull << 32);                      // (counted as published before its header can be seen)
            v = atomicAdd(&b->ht, 1ull);
            ok = 1;
#ifdef FMGPU_DEV
            atomicAdd(&b->cas_tries, 1u);
#endif
        }
    }
    ok = __shfl(ok, 0, 64);
    *index = (uint32_t)wave_bcast64(v, 0);
    *slot = *index % kBoardBatches;
    if (ok && lane == 0) {
        // the slot is this batch's once the batch that used it a ring earlier has been TAKEN (its taker says so in the header).  Normally long past: a wave between its reservation and
        // its header spends microseconds, a ring is thousands of reservations — but a wave can be held up (two processes on one card), and a batch written over an untaken one would
        // lose subtrees silently.  The wait is for another wave's progress, never for this one's.
        const unsigned long long free_mark = *index >= kBoardBatches ? (((unsigned long long)(*index - kBoardBatches + 1u) << 8) | 0xffull) : 0ull;
        for (uint32_t spin = 0; __hip_atomic_load(&b->hdr[*slot][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != free_mark; ++spin) {
            if (spin == kBoardSpinCap) { atomicAdd(&b->failed, 1ull); break; }
            __builtin_amdgcn_s_sleep(32);
        }
    }
    return ok != 0;
}
// a taker, after its lanes have read their tasks: the slot may be written again
__device__ __forceinline__ void board_release(WorkBoard* b, uint32_t lane, uint32_t slot, uint32_t index) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(&b->hdr[slot][0], ((unsigned long long)(index + 1u) << 8) | 0xffull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void board_put(WorkBoard* b, uint32_t slot, uint32_t word, uint32_t task, uint32_t value) {      // a giver's lane: one dword of its task, written through
    __hip_atomic_store(&b->task[slot][word][task], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void board_publish(WorkBoard* b, uint32_t lane, uint32_t slot, uint32_t index, uint32_t tasks) {      // all lanes, after the givers wrote their tasks
#ifdef FMGPU_DEV
    if (lane == 0) atomicAdd(&b->tasks, (unsigned long long)tasks);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the wave's task stores (and the count above) have been written through
    if (lane == 0) __hip_atomic_store(&b->hdr[slot][0], ((unsigned long long)(index + 1u) << 8) | tasks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t board_word(const WorkBoard* b, uint32_t slot, uint32_t word, uint32_t task) {      // (past the reader's L1: the slot may have been read before)
    return __hip_atomic_load(&b->task[slot][word][task], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t lane) {
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { uint32_t y = __shfl_up(x, off, 64); if (lane >= (uint32_t)off) x += y; }
    return x - v;
}

// ---- wave-level reservations for the DFS kernels ----------------------------------------------------------------------------------
// One atomicAdd per hit (or per query handed out) queues millions of atomics behind each other on one address: ~9 ns each, more than the
// searches themselves.  Instead every lane of a wave passes a synchronous section in each loop iteration (a lane without work idles until
// the whole wave is done): queries are handed out with one reservation for all lanes that want one, and hit records go to a ring per WAVE in
// LDS — the slot from an LDS atomic on the ring's fill count — that the whole wave writes out with ONE reservation once it holds kWaveRingFlush
// records.  (Round 2 kept two slots per LANE and flushed when one lane's were full: a lane in a repeat fills its two while the other 63 are
// empty — one reservation per ~6 records, 8.8 M returning atomics on one word per 10 M reads, near the ~88 M/s a single word sustains.)
constexpr uint32_t kWaveHitBuf = 2;                               // (sizes the LDS area: 2 x 64 record slots per wave)
constexpr uint32_t kHitWords = kWide ? 10u : 7u;                  // [qidx lo, qidx hi, lb, lbRev, len, e, seq (, high words of lb, lbRev, len)]
constexpr uint32_t kWaveHitWords = kWaveHitBuf * kHitWords * 256u;   // per block: 4 waves x [word][128 slots]; slot 0 of word 0 is the ring's fill count
constexpr uint32_t kWaveRingSlots = kWaveHitBuf * 64u - 1u;       // 127 records per wave
constexpr uint32_t kWaveRingFlush = 64u;                          // written out once this many are waiting: 63 more fit (one per lane and iteration), a surplus goes out one by one

__device__ __forceinline__ uint64_t wave_hand_out_at(bool want, unsigned long long* next, uint32_t lane) {     // all lanes call; valid for lanes with `want`
    const uint64_t wm = __ballot(want);
    if (!wm) return 0;
    const uint32_t leader = (uint32_t)__ffsll((unsigned long long)wm) - 1u;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(next, (unsigned long long)__popcll(wm));
    base = ((unsigned long long)__shfl((uint32_t)(base >> 32), leader, 64) << 32) | __shfl((uint32_t)base, leader, 64);
    return base + (uint64_t)__popcll(wm & ((1ull << lane) - 1ull));
}
__device__ __forceinline__ uint64_t wave_hand_out(bool want, Counters* ctr, uint32_t lane) { return wave_hand_out_at(want, &ctr->next, lane); }
__device__ __forceinline__ uint32_t* wave_ring(uint32_t* s_hb) { return s_hb + (threadIdx.x >> 6) * (kWaveHitBuf * kHitWords * 64u); }
__device__ __forceinline__ void wave_ring_init(uint32_t* s_hb) {  // every wave, before its first record (LDS operations of one wave execute in order)
    if ((threadIdx.x & 63u) == 0) __hip_atomic_store((lds_word*)wave_ring(s_hb), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ uint32_t wave_ring_fill(uint32_t* s_hb) {   // wave-uniform
    return __hip_atomic_load((lds_word*)wave_ring(s_hb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ void wave_keep_hit(uint32_t* s_hb, uint32_t& nh, fmgpu_hit* out, uint64_t cap, Counters* ctr, uint64_t q, Cur r, uint32_t e, uint32_t seq) {
    uint32_t* ring = wave_ring(s_hb);
    const uint32_t slot = 1u + __hip_atomic_fetch_add((lds_word*)ring, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    constexpr uint32_t W = kWaveHitBuf * 64u;                      // slots per word plane
    if (slot <= kWaveRingSlots) {
        uint32_t* h = ring + slot;
        h[0] = (uint32_t)q; h[W] = (uint32_t)(q >> 32); h[2 * W] = (uint32_t)r.lb; h[3 * W] = (uint32_t)r.lbRev; h[4 * W] = (uint32_t)r.len; h[5 * W] = e; h[6 * W] = seq;
        if constexpr (kWide) { h[7 * W] = (uint32_t)((uint64_t)r.lb >> 32); h[8 * W] = (uint32_t)((uint64_t)r.lbRev >> 32); h[9 * W] = (uint32_t)((uint64_t)r.len >> 32); }
    } else emit_hit(out, cap, ctr, q, r, e, seq);                  // the ring is full (more than 63 records since the wave last looked): this one goes out alone
    (void)nh;
}
__device__ __forceinline__ void wave_flush_hits(uint32_t* s_hb, uint32_t& nh, uint32_t lane, fmgpu_hit* out, uint64_t cap, Counters* ctr) {   // all lanes call
    uint32_t* ring = wave_ring(s_hb);
    const uint32_t total = min(wave_ring_fill(s_hb), kWaveRingSlots);
    constexpr uint32_t W = kWaveHitBuf * 64u;
    if (total) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&ctr->hits, (unsigned long long)total);
        base = ((unsigned long long)__shfl((uint32_t)(base >> 32), 0, 64) << 32) | __shfl((uint32_t)base, 0, 64);
        for (uint32_t k = 1u + lane; k <= total; k += 64u) {
            const uint32_t* h = ring + k;
            const unsigned long long at = base + (k - 1u);
            if (at < cap) {
                fmgpu_hit rec;
                rec.qidx = (uint64_t)h[0] | ((uint64_t)h[W] << 32); rec.lb = h[2 * W]; rec.lb_rev = h[3 * W]; rec.len = h[4 * W];
                if constexpr (kWide) { rec.lb |= (uint64_t)h[7 * W] << 32; rec.lb_rev |= (uint64_t)h[8 * W] << 32; rec.len |= (uint64_t)h[9 * W] << 32; }
                rec.errors = h[5 * W]; rec.seq = h[6 * W];
                out[at] = rec;
            }
        }
        if (lane == 0) __hip_atomic_store((lds_word*)ring, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    nh = 0;
}


// one-row cursors with the explicit LF table (DevString::lf_table): LF(row) in one 4-byte load; the row's symbol is the k with
// C[k] <= LF(row) < C[k+1] (C staged in LDS)
struct LfView { const idx_t* fw; const idx_t* rv; const idx_t* C; };
__device__ __forceinline__ uint32_t symbol_of_lf_lds(const idx_t* sC, uint32_t sigma, idx_t t) {
    uint32_t lo = 0, hi = sigma;                    // invariant: sC[lo] <= t < sC[hi]   (sC[sigma] = n > t)
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (sC[mid] <= t) lo = mid; else hi = mid; }
    return lo;
}

// eight query bytes -> eight nibbles (a byte >= sigma, sigma <= 15, becomes 15 = "not a symbol"), with word operations
__device__ __forceinline__ uint32_t pack_nibbles8(uint64_t x, uint32_t sigma) {
    const uint64_t k1 = 0x0101010101010101ull;
    const uint64_t t = (((x & (0x7full * k1)) + (uint64_t)(0x80u - sigma) * k1) | x) & (0x80ull * k1);   // bit 7 of a byte set <=> byte >= sigma
    uint64_t y = (x | ((t >> 7) * 0xffull)) & (0x0full * k1);
    y = (y | (y >> 4)) & 0x00ff00ff00ff00ffull;
    y = (y | (y >> 8)) & 0x0000ffff0000ffffull;
    y = (y | (y >> 16)) & 0x00000000ffffffffull;
    return (uint32_t)y;
}
// ---- per-lane query staging in LDS ------------------------------------------------------------------------------
// A DFS visits a few hundred nodes per query; reading the query symbol of every node from global memory costs a
// second random line per node (half a million lanes' query lines do not survive in L2).  Each lane therefore copies
// its query once into LDS: word w of lane t at  lds[w * 256 + t]  (bank = t mod 32/64: conflict-free), 8 symbols per
// word as nibbles (sigma <= 15; 15 = "not a symbol") or 4 symbols per word as bytes.
struct QStage {
    uint32_t* lds;          // this block's staging area
    uint32_t words;         // words per query (0 = staging disabled: read global memory)
    uint32_t nib;           // 1 = 4-bit symbols
};
__device__ __forceinline__ void qstage_load(const QStage& st, const uint8_t* qbuf, uint64_t off, uint32_t m, uint32_t sigma) {
    if (!st.words) return;
    const uint64_t addr = (uint64_t)qbuf + off;
    const uint32_t mis = (uint32_t)(addr & 7ull);
    const uint64_t* base = reinterpret_cast<const uint64_t*>(addr - mis);
    const uint32_t last = (mis + m - 1u) >> 3;                     // last aligned word that holds query bytes (m >= 1)
    uint64_t lo = base[0];
    const uint32_t per = st.nib ? 8u : 4u;
    uint32_t wi = 0;
    for (uint32_t k = 0; k * 8u < m; ++k) {
        uint64_t hi = (k + 1 <= last) ? base[k + 1] : 0ull;
        uint64_t x = mis ? ((lo >> (8u * mis)) | (hi << (64u - 8u * mis))) : lo;   // query bytes 8k .. 8k+7
        lo = hi;
        if (st.nib) {
            st.lds[wi * 256u + threadIdx.x] = pack_nibbles8(x, sigma); ++wi;
        } else {
            st.lds[wi * 256u + threadIdx.x] = (uint32_t)x; ++wi;
            if ((k * 8u + 4u) < m) { st.lds[wi * 256u + threadIdx.x] = (uint32_t)(x >> 32); ++wi; }
        }
    }
    (void)per;
}
__device__ __forceinline__ uint32_t qstage_get(const QStage& st, const uint8_t* qs, uint32_t p) {
    if (!st.words) return qs[p];
    if (st.nib) { uint32_t v = (st.lds[(p >> 3) * 256u + threadIdx.x] >> ((p & 7u) * 4u)) & 15u; return v == 15u ? 255u : v; }
    return (st.lds[(p >> 2) * 256u + threadIdx.x] >> ((p & 3u) * 8u)) & 255u;
}

// wave-synchronous staging of one query per lane: all loads of a chunk are issued before the first is consumed
__device__ __forceinline__ void qstage_load_sync(const QStage& st, const uint8_t* qbuf, uint64_t off, uint32_t m, uint32_t sigma, bool active, uint32_t maxm) {
    if (!st.words) return;                                         // staging disabled (very long queries): qstage_get reads global memory
    const uint64_t addr = (uint64_t)qbuf + off;
    const uint32_t mis = (uint32_t)(addr & 7ull);
    const uint64_t* base = reinterpret_cast<const uint64_t*>(addr - mis);
    const uint32_t nw = active ? ((mis + m + 7u) >> 3) : 0u;        // aligned 64-bit words that hold query bytes
    uint64_t carry = 0;
    uint32_t wi = 0;
    for (uint32_t k0 = 0; k0 <= ((maxm + 14u) >> 3); k0 += 8) {     // uniform trip count (maxm >= every lane's m); word nw flushes the last bytes
        uint64_t r[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) r[k] = (k0 + k < nw) ? base[k0 + k] : 0ull;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            // query bytes 8(k0+k-1) .. +7 are completed by word k0+k:  x = (carry >> 8mis) | (r[k] << (64 - 8mis))
            const uint32_t widx = k0 + k;
            if (widx == 0) { carry = r[k]; continue; }
            uint64_t x = mis ? ((carry >> (8u * mis)) | (r[k] << (64u - 8u * mis))) : carry;
            carry = r[k];
            const uint32_t p0 = (widx - 1u) * 8u;                   // first query position in x
            if (active && p0 < m) {
                if (st.nib) {
                    st.lds[wi * 256u + threadIdx.x] = pack_nibbles8(x, sigma); ++wi;
                } else {
                    st.lds[wi * 256u + threadIdx.x] = (uint32_t)x; ++wi;
                    if (p0 + 4u < m) { st.lds[wi * 256u + threadIdx.x] = (uint32_t)(x >> 32); ++wi; }
                }
            }
        }
    }
}


// the 16 query symbols of a stretch as a 32-bit code (2 bits per symbol, the symbol consumed first in the low bits), from the nibble staging:
// pos = query position of the stretch's first step, right = positions ascend.  valid = all 16 symbols are in 1 .. 4.
__device__ __forceinline__ uint32_t query_code16(const QStage& qst, uint32_t pos, bool right, bool& valid) {
    const uint32_t p0 = right ? pos : pos - 15u;                 // lowest query position of the stretch
    const uint32_t w0 = qst.lds[(p0 >> 3) * 256u + threadIdx.x], w1 = qst.lds[((p0 >> 3) + 1u) * 256u + threadIdx.x];
    const uint32_t w2 = (p0 & 7u) ? qst.lds[((p0 >> 3) + 2u) * 256u + threadIdx.x] : 0u;
    const uint32_t sh = 4u * (p0 & 7u);
    uint64_t x = ((uint64_t)w0 | ((uint64_t)w1 << 32)) >> sh;
    if (sh) x |= (uint64_t)w2 << (64u - sh);
    const uint64_t v = x - 0x1111111111111111ull;                // nibbles 1..4 -> 0..3
    valid = ((v & ~x & 0x8888888888888888ull) == 0ull) && ((v & 0xccccccccccccccccull) == 0ull);
    uint64_t t = v & 0x3333333333333333ull;
    t = (t | (t >> 2)) & 0x0f0f0f0f0f0f0f0full; t = (t | (t >> 4)) & 0x00ff00ff00ff00ffull;
    t = (t | (t >> 8)) & 0x0000ffff0000ffffull; t = (t | (t >> 16)) & 0x00000000ffffffffull;
    uint32_t qc = (uint32_t)t;                                   // symbol at position p0 + k in bits 2k
    if (!right) { qc = __brev(qc); qc = ((qc >> 1) & 0x55555555u) | ((qc & 0x55555555u) << 1); }   // ... at position pos - k
    return qc;
}

// ------------------------------------------------------------------ host launchers
static int step_counters(bool want, hipStream_t stream, unsigned long long** out) {
    CallScratch* sc = nullptr;
    int rc = call_scratch(&sc); if (rc) return rc;
    if (want) FM_HIP(hipMemsetAsync(sc->ctr, 0, (size_t)kCounterStripes * kCounterKinds * 8, stream));
    *out = want ? sc->ctr : sc->sink;
    return 0;
}
// totals[0] executed steps, [1] table bytes, [2] table accesses
static int read_step_counters(const unsigned long long* dev, hipStream_t stream, unsigned long long* totals) {
    CallScratch* sc = nullptr;
    int rc = call_scratch(&sc); if (rc) return rc;
    unsigned long long* h = sc->pinned;
    FM_HIP(hipMemcpyAsync(h, dev, (size_t)kCounterStripes * kCounterKinds * 8, hipMemcpyDeviceToHost, stream));
    FM_HIP(hipStreamSynchronize(stream));
    for (unsigned kind = 0; kind < kCounterKinds; ++kind) {
        unsigned long long t = 0;
        for (unsigned k = 0; k < kCounterStripes; ++k) t += h[kind * kCounterStripes + k];
        totals[kind] = t;
    }
    return 0;
}

struct EventTimer {       // the thread's cached event pair (one timed call at a time per host thread)
    hipEvent_t a = nullptr, b = nullptr; hipStream_t s; bool on;
    EventTimer(hipStream_t s_, bool on_) : s(s_), on(on_) {
        CallScratch* sc = nullptr;
        if (on && call_scratch(&sc) == 0) { a = sc->ev_a; b = sc->ev_b; } else on = false;
    }
    void start() { if (on) (void)hipEventRecord(a, s); }
    void stop() { if (on) (void)hipEventRecord(b, s); }
    float ms() { float v = 0; if (on) { (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&v, a, b); } return v; }
};

// longest and shortest query and the total symbol count of a batch whose offsets live in HBM: one reduction kernel, one small copy,
// one synchronisation (per-block partial results reduced on the host: no atomics, nothing to initialise)
constexpr unsigned kLenBlocks = 1024;
static __global__ __launch_bounds__(256) void k_len_range(const uint64_t* __restrict__ qoff, uint64_t nq, unsigned long long* __restrict__ out) {
    __shared__ unsigned long long s_v[4], s_w[4];
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = 0, w = ~0ull;
    for (uint64_t q = t; q < nq; q += (uint64_t)gridDim.x * blockDim.x) { unsigned long long l = qoff[q + 1] - qoff[q]; v = l > v ? l : v; w = l < w ? l : w; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_xor(v, off, 64); v = o > v ? o : v;
        unsigned long long p = __shfl_xor(w, off, 64); w = p < w ? p : w;
    }
    if ((threadIdx.x & 63u) == 0) { s_v[threadIdx.x >> 6] = v; s_w[threadIdx.x >> 6] = w; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) { v = s_v[i] > v ? s_v[i] : v; w = s_w[i] < w ? s_w[i] : w; }
        out[2 * blockIdx.x] = v; out[2 * blockIdx.x + 1] = w;
        if (blockIdx.x == 0) out[2 * gridDim.x] = qoff[nq];
    }
}

static int query_shape(const uint64_t* dqoff, uint64_t nq, hipStream_t stream, uint32_t* out_max, uint32_t* out_min, uint64_t* out_total) {
    CallScratch* sc = nullptr;
    int rc = call_scratch(&sc); if (rc) return rc;
    unsigned long long* d = sc->len2;
    const unsigned blocks = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nq + 255) / 256, kLenBlocks));
    k_len_range<<<dim3(blocks), dim3(256), 0, stream>>>(dqoff, nq, d);
    unsigned long long* h = sc->pinned;
    hipError_t e = hipMemcpyAsync(h, d, ((size_t)2 * blocks + 1) * 8, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return hip_fail(e, "k_len_range");
    unsigned long long mx = 0, mn = ~0ull;
    for (unsigned b = 0; b < blocks; ++b) { mx = std::max(mx, h[2 * b]); mn = std::min(mn, h[2 * b + 1]); }
    *out_max = (uint32_t)std::min<unsigned long long>(mx, 0xffffffffull);
    *out_min = (uint32_t)std::min<unsigned long long>(mn, 0xffffffffull);
    if (out_total) *out_total = h[2 * blocks];
    return 0;
}
static int query_len_range(const uint64_t* dqoff, uint64_t nq, hipStream_t stream, uint32_t* out_max, uint32_t* out_min) {
    return query_shape(dqoff, nq, stream, out_max, out_min, nullptr);
}

// expands a scheme for queries of length m into the fast kernel's per-step table (see k_scheme_fast); false if it does not fit.

}  // namespace FMGPU_NS

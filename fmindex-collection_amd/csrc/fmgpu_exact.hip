// fmgpu_exact.hip — search_no_errors::search (search/SearchNoErrors.h:12-86), one query per lane:
//  k_exact_p      two symbols per step on the pair lines (Format P), optionally behind an interval table
//  k_exact_a      one symbol per step on the one-symbol blocks (Format A)
//  k_exact_s      one line per step on the symbol planes (Format S), optionally behind an interval table
//  k_exact_m      the multi-ary wavelet tree (Format M)
//  k_exact_kstep  table-driven: interval table, k-symbol-step table, LF^J walk tables
//  k_exact        any layout; k_exact_depth: the profile of a batch (symbols until one row)
#include "fmgpu_search_shared.h"

namespace FMGPU_NS {

// ------------------------------------------------------------------ exact search
template <class Occ>
__global__ __launch_bounds__(256) void k_exact(Occ occ, const uint8_t* __restrict__ qbuf, const uint64_t* __restrict__ qoff,
                                               uint64_t nq, idx_t n, uint64_t* __restrict__ out_lb, uint64_t* __restrict__ out_len,
                                               unsigned long long* __restrict__ steps_total) {
    uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t steps = 0;
    if (q < nq) {
        uint64_t o = qoff[q];
        uint32_t m = (uint32_t)(qoff[q + 1] - o);
        const uint8_t* s = qbuf + o;
        const uint32_t sigma = occ.sigma();
        idx_t lb = 0, len = n;
        for (uint32_t i = m; i-- > 0;) {
            uint32_t c = s[i];
            ++steps;
            if (c >= sigma) { lb = 0; len = 0; break; }      // not a rank of this alphabet: no occurrence
            idx_t ra, rb;
            occ.lf2(lb, lb + len, c, ra, rb);                 // fmindex/FMIndexCursor.h:33-37
            lb = ra; len = rb - ra;
            if (len == 0) break;
        }
        store_interval(out_lb, out_len, q, lb, len);
    }
    add_counters(steps_total, steps, 0u, 0u);
}

// symbols a query consumes until its interval is a single row (or empty): out[q] = that count, or its length + 1 if the interval still
// holds several rows at the end — the quantity that decides how much of a read the one-row walk tables can serve (bench.py reports its distribution)
template <class Occ>
__global__ __launch_bounds__(256) void k_exact_depth(Occ occ, const uint8_t* __restrict__ qbuf, const uint64_t* __restrict__ qoff, uint64_t nq, idx_t n,
                                                     uint32_t* __restrict__ out) {
    uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    uint64_t o = qoff[q];
    uint32_t m = (uint32_t)(qoff[q + 1] - o);
    const uint8_t* s = qbuf + o;
    const uint32_t sigma = occ.sigma();
    idx_t lb = 0, len = n;
    uint32_t depth = m + 1, done = 0;
    for (uint32_t i = m; i-- > 0 && len > 1;) {
        uint32_t c = s[i];
        ++done;
        if (c >= sigma) { len = 0; break; }
        idx_t ra, rb;
        occ.lf2(lb, lb + len, c, ra, rb);
        lb = ra; len = rb - ra;
    }
    if (len <= 1) depth = done;
    out[q] = depth;
}

template <int SIGMA>
__device__ __noinline__ void lf0_pair(const OccA<SIGMA>& occ, idx_t a, idx_t b, idx_t& ra, idx_t& rb) { ra = occ.lf0_fused(a); rb = occ.lf0_fused(b); }
template <int SIGMA>
__global__ __launch_bounds__(256) void k_exact_a(OccA<SIGMA> occ, const uint8_t* __restrict__ qbuf, const uint64_t* __restrict__ qoff,
                                                 uint64_t nq, idx_t n, uint64_t* __restrict__ out_lb, uint64_t* __restrict__ out_len,
                                                 unsigned long long* __restrict__ steps_total) {
    uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t steps = 0, acc = 0;
    if (q < nq) {
        uint64_t o = qoff[q];
        uint32_t m = (uint32_t)(qoff[q + 1] - o);
        const uint32_t sigma = occ.sigma();
        idx_t lb = 0, len = n;
        if (m) {
            QueryReader qr; qr.init(qbuf, o, m);
            for (uint32_t i = 0; i < m; ++i) {
                uint32_t c = qr.next();
                ++steps;
                if (c >= sigma) { lb = 0; len = 0; break; }
                const idx_t a = lb, b = lb + len;
                idx_t ra, rb;
                if (c == 0 && occ.v.fused) { lf0_pair(occ, a, b, ra, rb); acc += 2; }   // a delimiter in the query on a table whose entry 0 carries presence bits (rare; out of line)
                else {
                    EntryA ea = load_entry_a(occ.v, a, c);
                    EntryA eb = ea;
                    ++acc;
                    if ((a >> 6) != (b >> 6)) { eb = load_entry_a(occ.v, b, c); ++acc; }
                    ra = ea.cnt + popc64(ea.bits & lowmask((uint32_t)a & 63u));
                    rb = eb.cnt + popc64(eb.bits & lowmask((uint32_t)b & 63u));
                }
                lb = ra; len = rb - ra;
                if (len == 0) break;
            }
        }
        store_interval(out_lb, out_len, q, lb, len);
    }
    add_counters(steps_total, steps, 12u * acc, acc);
}

// ---- exact search in two-symbol steps on Format P (fmgpu_common.h): one 128-byte line per interval end and PAIR of symbols.  An exact search is bound
// by the random line fills it causes (tools/membench.hip: 52-55 G dependent lines/s, whatever is read of a line), so halving the lines of a read halves
// its time.  A pair whose interval comes out empty is taken again in one-symbol steps (Format A), which yields the row and the step count a
// one-symbol search ends with; so is a pair that holds a delimiter or a byte outside the alphabet, and the last symbol of a read of odd length.
constexpr uint32_t kPairFilterBits = 32768;          // line number mod this: the ~50 listed rows of a genome mark 0.15 % of the lines
// ---- lines fetched by the eight lanes of an octet together (k_exact_p, k_exact_s).  A lane that reads 44-68 bytes of its own random line with four or five load
// instructions pays as many address translations and passes through the texture path per line, and that — not the line fills — bounds such a kernel
// (tools/membench.hip modes 8 / 9 / 4: 33 / 47 / 44 G lines/s on a 3.1 GB table, 24 / 22 / 25 on a 4.2 GB one, where one load per line keeps 51).  Here
// instruction k of a round has the eight lanes of every octet load the eight 16-byte pieces of the line of the octet's lane k — one coalesced 128-byte
// request and one translation per line — straight into LDS (LDS-DMA: piece j of lane 8o + k's line lands at region k, offset 128 o + 16 j), from where the
// owner reads what it needs.  The loops are wave-uniform (reads that are over ride along with a dummy line); waves do not synchronise with each other.
constexpr uint32_t kCoopRegion = 1024u + 16u;        // bytes per region (64 pieces + padding that spreads the owners' reads over the LDS banks)
__device__ __forceinline__ void coop_round(const uint8_t* __restrict__ flat, uint32_t line, uint32_t lane, lds_word* wave_lds) {
#pragma unroll
    for (uint32_t k = 0; k < 8u; ++k) {
        const uint32_t l = __shfl(line, (int)((lane & ~7u) | k), 64);
        const uint8_t* g = flat + (size_t)l * 128u + (lane & 7u) * 16u;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(wave_lds + k * (kCoopRegion / 4u)), 16, 0, 0);
    }
}
// rows of the pair before row i in the owner's line (now in LDS) + the pair's count
__device__ __forceinline__ uint32_t pair_rank_lds(const lds_word* own, uint32_t i, uint32_t pc) {
    const uint32_t cnt = own[pc];
    const flat_u32x4 w0 = *reinterpret_cast<const __attribute__((address_space(3))) flat_u32x4*>(own + 16), w1 = *reinterpret_cast<const __attribute__((address_space(3))) flat_u32x4*>(own + 20);
    const flat_u32x4 w2 = *reinterpret_cast<const __attribute__((address_space(3))) flat_u32x4*>(own + 24), w3 = *reinterpret_cast<const __attribute__((address_space(3))) flat_u32x4*>(own + 28);
    const uint32_t i0 = (pc & 1u) ? 0u : ~0u, i1 = (pc & 2u) ? 0u : ~0u, i2 = (pc & 4u) ? 0u : ~0u, i3 = (pc & 8u) ? 0u : ~0u;
    const uint32_t off = i & 127u;
    const uint32_t m0 = off >= 32u ? ~0u : (1u << off) - 1u;
    const uint32_t m1 = off >= 64u ? ~0u : (off > 32u ? (1u << (off - 32u)) - 1u : 0u);
    const uint32_t m2 = off >= 96u ? ~0u : (off > 64u ? (1u << (off - 64u)) - 1u : 0u);
    const uint32_t m3 = off > 96u ? (1u << (off - 96u)) - 1u : 0u;
    // rows 0..63: w0 = planes 0, 1 (lo, hi words each), w1 = planes 2, 3; rows 64..127: w2, w3
    const uint32_t h0 = (w0.x ^ i0) & (w0.z ^ i1) & (w1.x ^ i2) & (w1.z ^ i3), h1 = (w0.y ^ i0) & (w0.w ^ i1) & (w1.y ^ i2) & (w1.w ^ i3);
    const uint32_t h2 = (w2.x ^ i0) & (w2.z ^ i1) & (w3.x ^ i2) & (w3.z ^ i3), h3 = (w2.y ^ i0) & (w2.w ^ i1) & (w3.y ^ i2) & (w3.w ^ i3);
    return cnt + __popc(h0 & m0) + __popc(h1 & m1) + __popc(h2 & m2) + __popc(h3 & m3);
}
// slut != null (fmgpu_index_accelerate_exact(h, 1, lutL, 0) on a handle with the pair table): a read whose last lutL symbols are all in 1..4 starts from the interval-table
// entry of those symbols (one 8 / 16-byte load from a table of 4^lutL entries — 12 symbols: 134 MB, Infinity-Cache resident — instead of lutL / 2 pair steps whose two interval
// ends lie in two lines each); an empty entry is walked from the start instead, so the miss row and the step count stay the one-symbol search's.
__global__ __launch_bounds__(256) void k_exact_p(OccA<5> occ, const uint8_t* __restrict__ pairs, const idx_t* __restrict__ ex, uint32_t nex, const idx_t* __restrict__ psuper,
                                                 const void* __restrict__ slut, uint32_t lutL,
                                                 const uint8_t* __restrict__ qbuf, const uint64_t* __restrict__ qoff,
                                                 uint64_t nq, idx_t n, uint64_t* __restrict__ out_lb, uint64_t* __restrict__ out_len,
                                                 unsigned long long* __restrict__ steps_total) {
    extern __shared__ uint32_t s_coop[];                            // 4 waves x 8 regions
    __shared__ uint32_t s_filt[kPairFilterBits / 32u];
    __shared__ idx_t s_ex[512];
    for (uint32_t t = threadIdx.x; t < kPairFilterBits / 32u; t += 256u) s_filt[t] = 0u;
    for (uint32_t t = threadIdx.x; t < 512u; t += 256u) s_ex[t] = t < nex ? ex[t] : ~(idx_t)0;
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < 512u; t += 256u) {
        const idx_t r = s_ex[t];
        if (r != ~(idx_t)0) { const uint32_t bk = (uint32_t)(r >> 7) & (kPairFilterBits - 1u); atomicOr(&s_filt[bk >> 5], 1u << (bk & 31u)); }
    }
    __syncthreads();
    // listed rows in [first row of i's line, i): they sit in the planes as code 0 and are in no count
    auto listed_before = [&](idx_t i) -> uint32_t {
        const uint32_t bk = (uint32_t)(i >> 7) & (kPairFilterBits - 1u);
        if (!((s_filt[bk >> 5] >> (bk & 31u)) & 1u)) return 0u;
        const idx_t first = i & ~(idx_t)127;
        uint32_t c = 0;
        for (uint32_t t = 0; t < 512u && s_ex[t] < i; ++t) if (s_ex[t] >= first) ++c;
        return c;
    };
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    lds_word* const wave_lds = (lds_word*)(s_coop + wave * 8u * (kCoopRegion / 4u));
    const lds_word* const own = wave_lds + (lane & 7u) * (kCoopRegion / 4u) + (lane >> 3) * 32u;
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t steps = 0, acc = 0, acc2 = 0, m = 0;
    idx_t lb = 0, len = n;
    QueryReader qr;
    if (q < nq) {
        const uint64_t o = qoff[q];
        m = (uint32_t)(qoff[q + 1] - o);
        if (m) qr.init(qbuf, o, m);
    }
    // one symbol, as k_exact_a does it; false once the search is over
    auto single = [&](uint32_t c) -> bool {
        ++steps;
        if (c >= 5u) { lb = 0; len = 0; return false; }
        const idx_t a = lb, b = lb + len;
        idx_t ra, rb;
        if (c == 0 && occ.v.fused) { lf0_pair(occ, a, b, ra, rb); acc += 2; }
        else {
            EntryA ea = load_entry_a(occ.v, a, c);
            EntryA eb = ea;
            ++acc;
            if ((a >> 6) != (b >> 6)) { eb = load_entry_a(occ.v, b, c); ++acc; }
            ra = ea.cnt + popc64(ea.bits & lowmask((uint32_t)a & 63u));
            rb = eb.cnt + popc64(eb.bits & lowmask((uint32_t)b & 63u));
        }
        lb = ra; len = rb - ra;
        return len != 0;
    };
    bool alive = m != 0;
    uint32_t done = 0, lut_steps = 0, acc3 = 0;                     // symbols of the read consumed so far; steps an interval-table entry stood for; such entries read
    if (slut && alive && m >= lutL && n > 1) {
        uint32_t code = 0; bool valid = true;
        for (uint32_t t = 0; t < lutL; ++t) { const uint32_t c = qr.next(); valid = valid && c - 1u < 4u; code |= ((c - 1u) & 3u) << (2u * t); }
        idx_t elb = 0, elen = 0;
        if (valid) {
            if constexpr (kWide) { const ulonglong2 en = reinterpret_cast<const ulonglong2*>(slut)[code]; elb = (idx_t)en.x; elen = (idx_t)en.y; }
            else { const uint2 en = reinterpret_cast<const uint2*>(slut)[code]; elb = en.x; elen = en.y; }
            ++acc3;
        }
        if (elen != 0) { lb = elb; len = elen; done = lutL; steps = lutL; lut_steps = lutL; }
        else qr.init(qbuf, qoff[q], m);                             // (a foreign byte among the symbols, or a string the text does not hold: from the start, step by step)
    }
    for (;;) {                                                      // every lane of the wave takes its next two symbols (or is done)
        if (done >= m) alive = false;
        if (!__ballot(alive)) break;
        const bool two = alive && done + 2u <= m;
        uint32_t y = 0, x = 0;
        if (alive) { y = qr.next(); if (two) x = qr.next(); }
        done += 2u;
        const bool pairable = two && y - 1u < 4u && x - 1u < 4u;
        bool stepped = false;
        if (__ballot(pairable)) {
            const uint32_t pc = pairable ? (x - 1u) * 4u + (y - 1u) : 0u;
            const idx_t a = lb, b = lb + len;
            const uint32_t la = pairable ? (uint32_t)(a >> 7) : 0u, lbn = pairable ? (uint32_t)(b >> 7) : 0u;      // (n < 2^38: a line number fits 31 bits)
            const bool far = pairable && la != lbn;
            coop_round(pairs, la, lane, wave_lds);
            __builtin_amdgcn_s_waitcnt(0x0f70);                     // vmcnt(0): the round's pieces are in LDS
            asm volatile("" ::: "memory");
            idx_t ra = 0, rb = 0;
            if (pairable) {
                acc2 += far ? 2u : 1u;
                ra = pair_rank_lds(own, (uint32_t)a, pc);
                if (!far) rb = pair_rank_lds(own, (uint32_t)b, pc);
            }
            if (__ballot(far)) {                                    // the other end's lines, where they are other lines (the first ~14 symbols of a read)
                __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): every lane has read what it needs of the first round
                asm volatile("" ::: "memory");
                coop_round(pairs, far ? lbn : 0u, lane, wave_lds);
                __builtin_amdgcn_s_waitcnt(0x0f70);
                asm volatile("" ::: "memory");
                if (far) rb = pair_rank_lds(own, (uint32_t)b, pc);
            }
            if (pairable) {
                if constexpr (kWide) {                              // the line's counts are relative to its super-block of 2^30 rows
                    const idx_t sa = psuper[(size_t)(a >> kSuperShift) * 16u + pc];
                    ra += sa; rb += (a >> kSuperShift) == (b >> kSuperShift) ? sa : psuper[(size_t)(b >> kSuperShift) * 16u + pc];
                }
                if (pc == 0u) { ra -= listed_before(a); rb -= listed_before(b); }
                if (rb > ra) { lb = ra; len = rb - ra; steps += 2u; stepped = true; }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                     // the next round overwrites the regions
            asm volatile("" ::: "memory");
        }
        if (alive && !stepped) {                                    // rare: a pair that came out empty or cannot be a pair, the last symbol of an odd read
            alive = single(y);
            if (alive && two) alive = single(x);
        }
    }
    if (q < nq) store_interval(out_lb, out_len, q, lb, len);
    add_counters(steps_total, steps, 12u * acc + 68u * acc2 + (uint32_t)kSlutEntryBytes * acc3, acc + acc2 + acc3, lut_steps);
}

// ---- exact search on Format S (fmgpu_common.h): ONE 128-byte line per LF step and interval end where the multi-ary wavelet tree of sigma = 28 takes two —
// the five symbol planes of the line's 64 rows, the symbol's 24-bit count before the line and the super-block's count (a small table, LDS or L2).
// The lines are fetched by the lanes of an OCTET together: a lane that reads 44 bytes of its own random line with four load instructions pays four address
// translations and four passes through the texture path per line, and that — not the line fills — bounds the kernel (tools/membench.hip modes 8 / 9 / 4:
// 33 / 47 / 44 G lines/s on a 3.1 GB table, 24 / 22 / 25 on a 4.2 GB one, where one load per line keeps 51).  Here instruction k of a round has the eight lanes of
// every octet load the eight 16-byte pieces of the line of the octet's lane k — one coalesced 128-byte request and one translation per line — straight into
// LDS (LDS-DMA: piece j of lane 8o + k's line lands at region k, offset 128 o + 16 j), from where the owner reads its planes and count.  The loop is
// wave-uniform (reads that are over ride along with a dummy line); waves do not synchronise with each other.
// rows before row i that hold symbol c in the owner's line (now in LDS) + the line's count of c
__device__ __forceinline__ uint32_t flat_rank_lds(const lds_word* own, uint32_t i, uint32_t c, uint32_t cbits) {
    const flat_u32x4 p01 = *reinterpret_cast<const __attribute__((address_space(3))) flat_u32x4*>(own);
    const flat_u32x4 p23 = *reinterpret_cast<const __attribute__((address_space(3))) flat_u32x4*>(own + 4);
    const uint32_t p4l = own[8], p4h = own[9];
    // the symbol's count: cbits bits at bit c * cbits of the 88 bytes behind the planes (the word behind the last one may be the next line's first: masked away)
    const uint32_t bitpos = c * cbits;
    const lds_word* g = own + 10u + (bitpos >> 5);
    const uint32_t cnt = __funnelshift_r(g[0], g[1], bitpos & 31u) & ((1u << cbits) - 1u);
    const uint32_t i0 = (c & 1u) ? 0u : ~0u, i1 = (c & 2u) ? 0u : ~0u, i2 = (c & 4u) ? 0u : ~0u, i3 = (c & 8u) ? 0u : ~0u, i4 = (c & 16u) ? 0u : ~0u;
    const uint32_t off = i & 63u;
    const uint32_t mlo = off >= 32u ? ~0u : (1u << off) - 1u, mhi = off > 32u ? (1u << (off - 32u)) - 1u : 0u;
    const uint32_t lo = (p01.x ^ i0) & (p01.z ^ i1) & (p23.x ^ i2) & (p23.z ^ i3) & (p4l ^ i4);
    const uint32_t hi = (p01.y ^ i0) & (p01.w ^ i1) & (p23.y ^ i2) & (p23.w ^ i3) & (p4h ^ i4);
    return cnt + __popc(lo & mlo) + __popc(hi & mhi);
}
// the super table in LDS: 32-bit rows as they are; 64-bit rows as a low word + a high byte per entry (5 instead of 8 bytes: sigma = 28, 4.5 x 10^9 rows: 19 KB;
// read through L2 instead the kernel took 13 % longer on the same text)
// Every workgroup stages its own copy of the table, beside the 33 KB of four waves' regions: 6.7 KB at 2 x 10^9 rows, 19 KB at 4.5 x 10^9, 42 KB at 10^10 (sigma = 28) — three, three and
// two 256-lane workgroups per CU.  The kernel is instantiated for workgroups of 256 and 512 lanes (a CU has 160 KB of LDS and one workgroup may take all of it); the launch takes the size
// that keeps the most waves resident (more than 16 gain nothing: they queue up at the memory system), and on a tie the LARGER one — half as many copies of the table are staged and held.
// Measured, 10 M x 40 aa, kernel ms with 256 / 512 lanes: 2.0 x 10^9 rows 7.98 / 7.53 (16 / 16 waves); 4.5 x 10^9 rows 7.95 / 8.38 (12 / 8); 10^10 rows 9.72 / 9.11 (8 / 8; the
// table read through L2 instead, 16 waves: 9.85 — at that size the random lines miss the translation caches, and residency is not what bounds the kernel).
constexpr size_t kLdsPerCu = 160 * 1024;
constexpr size_t kFlatSuperLdsMax = 88 * 1024;      // (with the 66.6 KB of eight waves' regions: one workgroup of 512 lanes per CU; sigma = 28: 2.1 x 10^10 rows)
__host__ __device__ constexpr size_t flat_super_lds_bytes(uint32_t entries) { return kWide ? ((size_t)entries * 5u + 15u) / 16u * 16u : (size_t)entries * 4u; }
__host__ constexpr size_t flat_lds_bytes(uint32_t block, uint32_t super_entries) { return (size_t)(block / 64u) * 8u * (1024u + 16u) + 16u + flat_super_lds_bytes(super_entries); }
inline uint32_t flat_block_lanes(uint32_t super_entries) {
    uint32_t best = 256, best_waves = 0;
    for (uint32_t blk : {256u, 512u}) {
        const size_t lds = flat_lds_bytes(blk, super_entries);
        if (lds > kLdsPerCu) continue;
        const uint32_t waves = (uint32_t)std::min<size_t>(16u, (kLdsPerCu / lds) * (blk / 64u));
        if (waves >= best_waves) { best_waves = waves; best = blk; }
    }
    return best;
}
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_exact_s(const uint8_t* __restrict__ flat, const idx_t* __restrict__ super, uint32_t sigma, uint32_t cbits,
                                                 const void* __restrict__ slut, uint32_t lutL,
                                                 const uint8_t* __restrict__ qbuf, const uint64_t* __restrict__ qoff,
                                                 uint64_t nq, idx_t n, uint64_t* __restrict__ out_lb, uint64_t* __restrict__ out_len,
                                                 unsigned long long* __restrict__ steps_total, uint32_t super_lds) {
    extern __shared__ uint32_t s_flat[];                            // BLOCK / 64 waves x 8 regions | 16 bytes | the super table, when it fits (super_lds entries)
    uint32_t* const s_lo = s_flat + (uint32_t)(BLOCK / 64) * 8u * (kCoopRegion / 4u) + 4u;
    uint8_t* const s_hi = reinterpret_cast<uint8_t*>(s_lo + super_lds);
    for (uint32_t t = threadIdx.x; t < super_lds; t += (uint32_t)BLOCK) {
        const idx_t v = super[t];
        s_lo[t] = (uint32_t)v;
        if constexpr (kWide) s_hi[t] = (uint8_t)((uint64_t)v >> 32);
    }
    __syncthreads();
    auto sup = [&](size_t at) -> idx_t {
        if (!super_lds) return super[at];
        if constexpr (kWide) return (idx_t)s_lo[at] | ((idx_t)s_hi[at] << 32);
        else return (idx_t)s_lo[at];
    };
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    lds_word* const wave_lds = (lds_word*)(s_flat + wave * 8u * (kCoopRegion / 4u));
    const lds_word* const own = wave_lds + (lane & 7u) * (kCoopRegion / 4u) + (lane >> 3) * 32u;
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t steps = 0, acc = 0, m = 0;
    idx_t lb = 0, len = n;
    QueryReader qr;
    if (q < nq) {
        const uint64_t o = qoff[q];
        m = (uint32_t)(qoff[q + 1] - o);
        if (m) qr.init(qbuf, o, m);
    }
    bool alive = m != 0;
    uint32_t done = 0, lut_steps = 0, acc3 = 0;                     // symbols of the read consumed so far; steps an interval-table entry stood for; such entries read
    if (slut && alive && m >= lutL && n > 1) {
        uint32_t code = 0, mul = 1; bool valid = true;
        for (uint32_t t = 0; t < lutL; ++t) { const uint32_t c = qr.next(); valid = valid && c - 1u < sigma - 1u; code += (c - 1u) * mul; mul *= sigma - 1u; }
        idx_t elb = 0, elen = 0;
        if (valid) {
            if constexpr (kWide) { const ulonglong2 en = reinterpret_cast<const ulonglong2*>(slut)[code]; elb = (idx_t)en.x; elen = (idx_t)en.y; }
            else { const uint2 en = reinterpret_cast<const uint2*>(slut)[code]; elb = en.x; elen = en.y; }
            ++acc3;
        }
        if (elen != 0) { lb = elb; len = elen; done = lutL; steps = lutL; lut_steps = lutL; }
        else qr.init(qbuf, qoff[q], m);
    }
    for (;;) {
        if (done >= m) alive = false;                               // (a shorter read of the wave is done)
        if (!__ballot(alive)) break;
        uint32_t c = 0;
        if (alive) {
            c = qr.next();
            ++steps; ++done;
            if (c >= sigma) { lb = 0; len = 0; alive = false; }
        }
        const idx_t a = lb, b = lb + len;
        const uint32_t la = alive ? (uint32_t)(a >> 6) : 0u, lbn = alive ? (uint32_t)(b >> 6) : 0u;
        const bool far = alive && la != lbn;
        coop_round(flat, la, lane, wave_lds);
        idx_t sa = 0, sb = 0;
        if (alive) {
            sa = sup((size_t)(a >> cbits) * sigma + c);
            sb = sa;
            if ((a >> cbits) != (b >> cbits)) sb = sup((size_t)(b >> cbits) * sigma + c);
            acc += far ? 2u : 1u;
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);                         // vmcnt(0): the round's pieces are in LDS
        asm volatile("" ::: "memory");
        idx_t ra = 0, rb = 0;
        if (alive) {
            ra = sa + flat_rank_lds(own, (uint32_t)a, c, cbits);
            if (!far) rb = sb + flat_rank_lds(own, (uint32_t)b, c, cbits);
        }
        if (__ballot(far)) {                                        // the other end's lines, where they are other lines (the first log_sigma(n) steps of a read)
            __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0): every lane has read what it needs of the first round
            asm volatile("" ::: "memory");
            coop_round(flat, far ? lbn : 0u, lane, wave_lds);
            __builtin_amdgcn_s_waitcnt(0x0f70);
            asm volatile("" ::: "memory");
            if (far) rb = sb + flat_rank_lds(own, (uint32_t)b, c, cbits);
        }
        if (alive) {
            lb = ra; len = rb - ra;
            if (len == 0) alive = false;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                         // the next round overwrites the regions
        asm volatile("" ::: "memory");
    }
    if (q < nq) store_interval(out_lb, out_len, q, lb, len);
    add_counters(steps_total, steps, 44u * acc + (uint32_t)kSlutEntryBytes * acc3, acc + acc3, lut_steps);
}

// ---- exact search over Format M (multi-ary wavelet tree) -----------------------------------------------------------
// One LF step = one node rank per level (string/Wavelet.h:104-119 with the levels fused into digits of 3 / 2 bits): per level and interval end one
// block of 64 positions = one memory line — the count of the digit value before the block and the digit's bit planes.  Both ends walk the same
// node path; once the interval is short they sit in the same block and the second end re-uses the first end's loads.  The query is staged in
// LDS, the node offsets and C[] too, so that the only global loads of a step are the block reads on the dependent chain.
template <int D>
__device__ __forceinline__ void load_block_m(const uint8_t* blk, uint32_t val, uint32_t& cnt, uint64_t (&pl)[3]) {
    cnt = reinterpret_cast<const uint32_t*>(blk)[val];
    if constexpr (D == 3) {
        const uint4 x = *reinterpret_cast<const uint4*>(blk + 32); const uint2 y = *reinterpret_cast<const uint2*>(blk + 48);
        pl[0] = (uint64_t)x.x | ((uint64_t)x.y << 32); pl[1] = (uint64_t)x.z | ((uint64_t)x.w << 32); pl[2] = (uint64_t)y.x | ((uint64_t)y.y << 32);
    } else if constexpr (D == 2) {
        const uint4 x = *reinterpret_cast<const uint4*>(blk + 16);
        pl[0] = (uint64_t)x.x | ((uint64_t)x.y << 32); pl[1] = (uint64_t)x.z | ((uint64_t)x.w << 32); pl[2] = 0;
    } else {
        const uint2 x = *reinterpret_cast<const uint2*>(blk + 8);
        pl[0] = (uint64_t)x.x | ((uint64_t)x.y << 32); pl[1] = 0; pl[2] = 0;
    }
}
template <int D>
__device__ __forceinline__ uint64_t match_m(const uint64_t (&pl)[3], uint32_t val) {
    uint64_t m = pl[0] ^ (0ull - (uint64_t)(~val & 1u));
    if constexpr (D >= 2) m &= pl[1] ^ (0ull - (uint64_t)((~val >> 1) & 1u));
    if constexpr (D >= 3) m &= pl[2] ^ (0ull - (uint64_t)((~val >> 2) & 1u));
    return m;
}
// one level of the descent for both interval ends
template <int D, int SHIFT, int FIRST>
__device__ __forceinline__ void level_m(const ViewM& v, const uint64_t* s_off, const uint32_t* nsup, const uint64_t* sup, uint32_t c, idx_t& a, idx_t& b, uint32_t& bytes, uint32_t& acc) {
    constexpr uint32_t stride = D == 3 ? 64u : (D == 2 ? 32u : 16u);
    const uint32_t val = (c >> SHIFT) & ((1u << D) - 1u), node = (uint32_t)FIRST + (c >> (SHIFT + D));
    const uint8_t* nb = v.data + s_off[node];
    uint32_t ca, cb; uint64_t pa[3], pb[3];
    load_block_m<D>(nb + (size_t)(a >> 6) * stride, val, ca, pa);
    bytes += 4u + 8u * D; ++acc;
    if ((a >> 6) != (b >> 6)) { load_block_m<D>(nb + (size_t)(b >> 6) * stride, val, cb, pb); bytes += 4u + 8u * D; ++acc; }
    else { cb = ca; pb[0] = pa[0]; pb[1] = pa[1]; pb[2] = pa[2]; }
    idx_t xa = ca + popc64(match_m<D>(pa, val) & lowmask((uint32_t)a & 63u));
    idx_t xb = cb + popc64(match_m<D>(pb, val) & lowmask((uint32_t)b & 63u));
    if constexpr (kWide) {                                          // counts are relative to super-blocks of 2^30 positions: the rest from the super table (LDS when it is small)
        const size_t row = nsup[node];
        xa += (idx_t)sup[(row + (size_t)(a >> kSuperShift)) * 8u + val];
        xb += (idx_t)sup[(row + (size_t)(b >> kSuperShift)) * 8u + val];
    }
    a = xa; b = xb;
}
template <int D0, int D1, int D2>
__global__ __launch_bounds__(256) void k_exact_m(ViewM v, const uint8_t* __restrict__ qbuf, const uint64_t* __restrict__ qoff, uint64_t nq, idx_t n,
                                                 uint64_t* __restrict__ out_lb, uint64_t* __restrict__ out_len, unsigned long long* __restrict__ steps_total,
                                                 uint32_t qwords, uint32_t super_rows) {
    extern __shared__ uint32_t s_query[];
    __shared__ uint64_t s_off[kMaxNodesM];
    __shared__ idx_t s_C[257];
    // 64-bit rows: the super table (a few rows of 8 counts per node: 13 rows for 4.5 x 10^9 residues) and the nodes' first rows staged in LDS — from global
    // memory they were two more dependent loads per level and interval end on the chain of every LF step (protein_wide 0.41-0.46 of the roofline vs 0.56 with 32-bit rows)
    constexpr uint32_t kSuperLds = kWide ? 96u : 1u;
    __shared__ uint64_t s_sup[kSuperLds * 8u];
    __shared__ uint32_t s_nsup[kWide ? kMaxNodesM : 1];
    const uint32_t sigma = v.sigma;
    for (uint32_t i = threadIdx.x; i < v.nnodes; i += blockDim.x) s_off[i] = v.node_off[i];
    for (uint32_t i = threadIdx.x; i <= sigma; i += blockDim.x) s_C[i] = v.C[i];
    const uint64_t* sup = nullptr; const uint32_t* nsup = nullptr;
    if constexpr (kWide) {
        for (uint32_t i = threadIdx.x; i < v.nnodes; i += blockDim.x) s_nsup[i] = v.node_super[i];
        nsup = s_nsup; sup = v.super;
        if (super_rows <= kSuperLds) { for (uint32_t i = threadIdx.x; i < super_rows * 8u; i += blockDim.x) s_sup[i] = v.super[i]; sup = s_sup; }
    }
    __syncthreads();
    constexpr int BITCT = D0 + D1 + D2;
    const QStage qst{s_query, qwords, 0u};
    uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t steps = 0, bytes = 0, acc = 0;
    if (q < nq) {
        uint64_t o = qoff[q];
        uint32_t m = (uint32_t)(qoff[q + 1] - o);
        const uint8_t* qs = qbuf + o;
        if (m) qstage_load(qst, qbuf, o, m, sigma);
        idx_t a = 0, b = n;
        for (uint32_t i = m; i-- > 0;) {
            uint32_t c = qstage_get(qst, qs, i);
            ++steps;
            if (c >= sigma) { a = b = 0; break; }
            level_m<D0, BITCT - D0, 0>(v, s_off, nsup, sup, c, a, b, bytes, acc);
            if constexpr (D1 > 0) level_m<D1, BITCT - D0 - D1, 1>(v, s_off, nsup, sup, c, a, b, bytes, acc);
            if constexpr (D2 > 0) level_m<D2, 0, 1 + (1 << D0)>(v, s_off, nsup, sup, c, a, b, bytes, acc);
            a += s_C[c]; b += s_C[c];
            if (a == b) break;
        }
        store_interval(out_lb, out_len, q, a, b - a);
    }
    add_counters(steps_total, steps, bytes, acc);
}

// ---- exact search over the multi-symbol-step table (fmgpu_index_accelerate) ---------------------------------------
// One table entry advances the cursor by K query symbols, so a query touches 1/K as many lines.  A chunk that holds a
// symbol outside [1, sigma), or that empties the interval, is (re-)walked with single steps so that the reported cursor
// and step count are exactly those of search/SearchNoErrors.h:12-26.  The query is staged in LDS up front (one query per
// lane: the staging is wave-synchronous by construction); the next chunk's context code is fetched from LDS while the
// table entries of the current chunk are in flight, and both interval ends are loaded together.
struct ExactAccel {                                 // (entry shapes by row width: fmgpu_common.h)
    const uint8_t* kblk; uint32_t K, ncodes;        // k-symbol-step table (or null; 32-bit rows only)
    const void* slut; uint32_t lutL;                // interval of the query's last lutL symbols (or null)
    const void* walk; uint32_t J, wbits;            // per row LF^J + the J symbols met (or null)
    const void* walk2;                              // per row LF^(2J) + the 2J symbols met as two codes (or null)
};

template <class Occ>
__global__ __launch_bounds__(256) void k_exact_kstep(Occ occ, ExactAccel ac, uint32_t R,
                                                     const uint8_t* __restrict__ qbuf, const uint64_t* __restrict__ qoff, uint64_t nq, idx_t n,
                                                     uint64_t* __restrict__ out_lb, uint64_t* __restrict__ out_len,
                                                     unsigned long long* __restrict__ steps_total, uint32_t qwords, uint32_t qnib, uint32_t maxm) {
    extern __shared__ uint32_t s_dyn[];
    const QStage qst{s_dyn, qwords, qnib};
    uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = q < nq;
    const uint64_t o = active ? qoff[q] : 0;
    const uint32_t m = active ? (uint32_t)(qoff[q + 1] - o) : 0;
    const uint8_t* sq = qbuf + o;
    const uint32_t sigma = occ.sigma();
    const uint32_t K = ac.K, ncodes = ac.ncodes;
    qstage_load_sync(qst, qbuf, o, m, sigma, active && m != 0, maxm);
    uint32_t steps = 0, tbytes = 0, tacc = 0;          // executed extensions; table bytes consumed / table accesses issued (fmgpu_stats)
    if (active) {
        idx_t lb = 0, len = n;
        uint32_t done = 0;                                       // symbols consumed (from the right end)
        // code of the `cnt` symbols ending at position m-1-from, radix R (tables indexed by contexts) or `shift` bits per symbol (walk table)
        auto code_of = [&](uint32_t from, uint32_t cnt, uint32_t shift, bool& valid) -> uint32_t {
            uint32_t code = 0, mul = 1; valid = m - from >= cnt;
            if (valid) for (uint32_t t = 0; t < cnt; ++t) {
                uint32_t c = qstage_get(qst, sq, m - 1 - from - t);
                valid = valid && c >= 1 && c < sigma;
                if (shift) code |= (c - 1) << (shift * t); else { code += (c - 1) * mul; mul *= R; }
            }
            return code;
        };
        if (ac.slut && n > 1) {                                  // the last lutL symbols at once; an empty entry is walked step by step instead
            bool v = false;                                      // (the reference's cursor and step count at the failing step are part of the result)
            uint32_t code;
            if (qst.words && qst.nib && R == 4u && ac.lutL <= 16u && m >= 16u) {       // radix 4 = 2 bits per symbol: the word-level code, cut to lutL symbols
                code = query_code16(qst, m - 1u, false, v);
                if (ac.lutL < 16u) code &= (1u << (2u * ac.lutL)) - 1u;
                if (!v) code = code_of(0, ac.lutL, 0, v);       // (an odd symbol among the 16: decide on the lutL symbols alone)
            } else code = code_of(0, ac.lutL, 0, v);
            if (v) {
                idx_t elb, elen;
                if constexpr (kWide) { const ulonglong2 en = reinterpret_cast<const ulonglong2*>(ac.slut)[code]; elb = (idx_t)en.x; elen = (idx_t)en.y; }
                else { const uint2 en = reinterpret_cast<const uint2*>(ac.slut)[code]; elb = en.x; elen = en.y; }
                tbytes += (uint32_t)kSlutEntryBytes; ++tacc;
                if (elen != 0) { lb = elb; len = elen; done = ac.lutL; steps = ac.lutL; }
            }
        }
        // main phase: one table load per iteration.  One row left: J (or 2J) symbols per load from the walk tables; otherwise K symbols from the
        // context table.  A step that would empty the interval (or meets an odd symbol) ends the phase WITHOUT touching the cursor:
        // its single-step walk is left to the tail phase, where the lanes of the wave are convergent again.
        // The lanes of a wave sit in different kinds of steps, so an iteration first decides every lane's kind and address (registers and LDS
        // only), then issues ONE 16-byte load for all of them (every table entry is dword-aligned and the tables carry 16 bytes of slack),
        // and only then looks at what came back: one memory round trip per iteration instead of one per kind of step present in the wave.
        const bool nib16 = qst.words && qst.nib && ac.wbits == 2u && ac.J == 16u;         // DNA: 16 staged nibbles -> one 32-bit code with a few word operations
        // A walk entry whose symbols differ from the query's tells where: the symbols before that place are matches and still go K at a time
        // through the context table (walks switched off up to `limit`); only the differing step itself and < K symbols before it are single steps.
        bool walks = true; uint32_t limit = 0;
        for (;;) {
            uint32_t kind = 0, q0 = 0, q1 = 0;                  // 1: 2J symbols, 2: J symbols, 3: K symbols from the context table
            const uint8_t* p0 = nullptr;
            const idx_t a = lb, b = lb + len;
            if (!walks) {
                bool valid = false;
                if (!ac.kblk || done + K > limit) break;
                const uint32_t code = code_of(done, K, 0, valid);
                if (!valid) break;
                kind = 3; p0 = ac.kblk + (size_t)code * 16u + (size_t)(a >> 6) * ((size_t)ncodes * 16u);
            } else if (ac.walk2 && len == 1 && m - done >= 2u * ac.J) {
                bool v0 = false, v1 = false;
                q0 = nib16 ? query_code16(qst, m - 1u - done, false, v0) : code_of(done, ac.J, ac.wbits, v0);
                q1 = nib16 ? query_code16(qst, m - 17u - done, false, v1) : code_of(done + ac.J, ac.J, ac.wbits, v1);
                if (!(v0 && v1)) break;
                kind = 1; p0 = reinterpret_cast<const uint8_t*>(ac.walk2) + (size_t)lb * kWalk2EntryBytes;
            } else if (ac.walk && len == 1 && m - done >= ac.J) {
                bool v = false;
                q0 = nib16 ? query_code16(qst, m - 1u - done, false, v) : code_of(done, ac.J, ac.wbits, v);
                if (!v) break;
                kind = 2; p0 = reinterpret_cast<const uint8_t*>(ac.walk) + (size_t)lb * kWalkEntryBytes;
            } else if (ac.kblk) {
                bool valid = false;
                const uint32_t code = code_of(done, K, 0, valid);
                if (!valid) break;
                kind = 3; p0 = ac.kblk + (size_t)code * 16u + (size_t)(a >> 6) * ((size_t)ncodes * 16u);      // same 12-byte entry shape as Format A
            } else {                                             // no context table: one symbol per iteration from the occurrence table itself
                if (done >= m) break;
                const uint32_t c = qstage_get(qst, sq, m - 1 - done);
                if (c >= sigma) break;
                idx_t ra, rb;
                occ.lf2(lb, lb + len, c, ra, rb);
                tbytes += 24u; tacc += 2u;
                if (rb == ra) break;
                lb = ra; len = rb - ra; ++steps; ++done;
                continue;
            }
            const uint4 r0 = *reinterpret_cast<const uint4*>(p0);
            uint4 r1 = r0;
            tbytes += kind == 2u ? (uint32_t)kWalkEntryBytes : (kind == 1u ? (uint32_t)kWalk2EntryBytes : 12u); ++tacc;
            // the entry's fields by row width: the row reached (all ones: a delimiter on the way) and the code(s) of the symbols met
            const bool w_none = kWide ? (r0.x == 0xffffffffu && r0.y == 0xffffffffu) : r0.x == 0xffffffffu;
            const idx_t w_row = kWide ? (idx_t)((uint64_t)r0.x | ((uint64_t)r0.y << 32)) : (idx_t)r0.x;
            const uint32_t w_c0 = kWide ? r0.z : r0.y, w_c1 = kWide ? r0.w : r0.z;
            if (kind == 3u && (a >> 6) != (b >> 6)) { r1 = *reinterpret_cast<const uint4*>(p0 + ((size_t)(b >> 6) - (size_t)(a >> 6)) * ((size_t)ncodes * 16u)); tbytes += 12u; ++tacc; }
            if (kind == 1u) {
                if (w_none) break;
                if (w_c0 != q0 || w_c1 != q1) {                  // symbols matching before the first differing one
                    const uint32_t same = w_c0 != q0 ? ((uint32_t)__ffs((int)(w_c0 ^ q0)) - 1u) / ac.wbits : ac.J + ((uint32_t)__ffs((int)(w_c1 ^ q1)) - 1u) / ac.wbits;
                    walks = false; limit = done + same;
                    continue;
                }
                lb = w_row; done += 2u * ac.J; steps += 2u * ac.J;
            } else if (kind == 2u) {
                if (w_none) break;
                if (w_c0 != q0) { walks = false; limit = done + ((uint32_t)__ffs((int)(w_c0 ^ q0)) - 1u) / ac.wbits; continue; }
                lb = w_row; done += ac.J; steps += ac.J;
            } else {
                const idx_t ra = r0.x + popc64(((uint64_t)r0.y | ((uint64_t)r0.z << 32)) & lowmask(a & 63u));
                const idx_t rb = r1.x + popc64(((uint64_t)r1.y | ((uint64_t)r1.z << 32)) & lowmask(b & 63u));
                if (rb == ra) break;
                lb = ra; len = rb - ra; steps += K; done += K;
            }
        }
        // tail phase: single steps — the step that failed in a table (until the interval is empty), the symbols after an odd one,
        // or the left-over symbols
        while (len != 0 && done < m) {
            uint32_t c = qstage_get(qst, sq, m - 1 - done);
            ++steps; ++done;
            if (c >= sigma) { lb = 0; len = 0; break; }
            idx_t ra, rb;
            occ.lf2(lb, lb + len, c, ra, rb);
            tbytes += 24u; tacc += 2u;
            lb = ra; len = rb - ra;
        }
        store_interval(out_lb, out_len, q, lb, len);
    }
    add_counters(steps_total, steps, tbytes, tacc);
}

namespace api {
#include "fmgpu_api_decl.h"

static int search_exact(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq,
                        uint64_t* out_lb, uint64_t* out_len, bool packed, fmgpu_stats* stats, void* stream_) {
    size_t dev_extra_lds = 0;                                      // dev knob: unused dynamic LDS per block, to limit the resident blocks per CU
    { const char* ev = dev_env("FMGPU_DEV_EXACT_LDS"); if (ev) dev_extra_lds = (size_t)atoi(ev); }
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (int drc = on_handle_device(x)) return drc;
    if (stats) *stats = fmgpu_stats{};
    if (nq == 0) return 0;
    if (!qbuf || !qoff || !out_lb || (!out_len && !packed)) return fail(FMGPU_ERR_INVALID, "qbuf / qoff / out_lb / out_len is null");
    if (kWide && packed) return fail(FMGPU_ERR_UNSUPPORTED, "the one-word interval form (lb << 32 | len) needs rows below 2^32; use fmgpu_search_exact");
    hipStream_t stream = (hipStream_t)stream_;
    Staged soff, sbuf, slb, slen;
    int rc;
    if ((rc = soff.in(qoff, (nq + 1) * 8, stream))) return rc;
    uint64_t total = 0;
    uint32_t shape_max = 0, shape_min = 0;
    bool have_shape = false;                                     // offsets in HBM: total and length range come back in one copy
    if (is_device_pointer(qoff)) { if ((rc = query_shape((const uint64_t*)soff.dev, nq, stream, &shape_max, &shape_min, &total))) return rc; have_shape = true; }
    else total = qoff[nq];
    if ((rc = sbuf.in(qbuf, total, stream))) return rc;
    if ((rc = slb.out(out_lb, nq * 8, stream))) return rc;
    if (out_len && (rc = slen.out(out_len, nq * 8, stream))) return rc;   // (packed form: slen.dev stays null and the kernels write one word per query)
    unsigned long long* dsteps = nullptr;
    if ((rc = step_counters(stats != nullptr, stream, &dsteps))) return rc;
    EventTimer timer(stream, stats != nullptr);
    FM_GRID(grid, nq);
    const dim3 block(256);
    const idx_t n = (idx_t)x->bwt.n;
    timer.start();
    uint32_t kq_words = 0, kq_max = 0, kq_nib = x->bwt.sigma <= 15 ? 1u : 0u;
    // the pair table with an interval table in front of it (and no other table): k_exact_p starts from the entry of the read's last symbols
    const bool pair_lut = x->bwt.sigma == 5 && x->bwt.pairs && x->bwt.slut && !x->bwt.kblk && !x->bwt.walkj && x->bwt.search_family() == FAM_A &&
                          !(kernel_flags() & (FMGPU_SEL_EXACT_ONE_SYMBOL | FMGPU_SEL_NO_EXACT_LUT));
    // ... and so does k_exact_s behind the symbol planes
    const bool flat_lut = x->bwt.flat && x->bwt.search_family() != FAM_A && x->bwt.slut && !x->bwt.kblk && !x->bwt.walkj && !(kernel_flags() & (FMGPU_SEL_EXACT_ON_TREE | FMGPU_SEL_NO_EXACT_LUT));
    const bool accel = (x->bwt.kblk || x->bwt.slut || x->bwt.walkj) && !pair_lut && !flat_lut;
    if (accel) {                                                 // LDS staging needs the longest query of the batch
        uint32_t mn = 0;
        if (have_shape) kq_max = shape_max;
        else if ((rc = query_len_range((const uint64_t*)soff.dev, nq, stream, &kq_max, &mn))) return rc;
        kq_words = kq_nib ? (kq_max + 7) / 8 : (kq_max + 3) / 4;
        if ((size_t)kq_words * 1024 > 48 * 1024) kq_words = 0;  // very long queries: read them from global memory
        timer.start();
    }
    if (accel) {
        const DevString& bs = x->bwt;
        ExactAccel ac{bs.kblk, bs.kstep, bs.kcodes, bs.slut, bs.slut_len, bs.walkj, bs.walk_J, bs.walk_bits, bs.walk2j};
        rc = dispatch_occ(bs, [&](auto occ, auto) {
            k_exact_kstep<decltype(occ)><<<grid, block, (size_t)kq_words * 1024 + dev_extra_lds, stream>>>(occ, ac, (uint32_t)bs.sigma - 1,
                                                                    (const uint8_t*)sbuf.dev, (const uint64_t*)soff.dev, nq, n, (uint64_t*)slb.dev,
                                                                    (uint64_t*)slen.dev, dsteps, kq_words, kq_nib, kq_max);
            return 0;
        });
    } else
    if (x->bwt.search_family() == FAM_A) {
        auto qb = (const uint8_t*)sbuf.dev; auto qo = (const uint64_t*)soff.dev; auto ol = (uint64_t*)slb.dev; auto on = (uint64_t*)slen.dev;
        // k_exact_a runs best with 5 resident blocks per CU, not the 8 its 28 registers allow (measured on the 3.09 Gbp index, 10 M x 101 bp: 8 / 6 / 5 / 4 / 3
        // blocks = 19.53 / 19.23 / 18.92 / 19.10 / 18.95 ms — more waves only queue up at the memory system): 28 KB of unused dynamic LDS set the residency
        const size_t lds_a = dev_env("FMGPU_DEV_EXACT_LDS") ? dev_extra_lds : (size_t)28 * 1024;
        if (x->bwt.sigma == 5 && x->bwt.pairs && !(kernel_flags() & (1 << 22)))
            k_exact_p<<<grid, block, 4 * 8 * kCoopRegion + dev_extra_lds, stream>>>(OccA<5>{x->bwt.va}, x->bwt.pairs, x->bwt.pairs_ex, x->bwt.pairs_nex, x->bwt.pairs_super,
                                                                                    pair_lut ? (const void*)x->bwt.slut : nullptr, x->bwt.slut_len, qb, qo, nq, n, ol, on, dsteps);
        else if (x->bwt.sigma == 5) k_exact_a<5><<<grid, block, lds_a, stream>>>(OccA<5>{x->bwt.va}, qb, qo, nq, n, ol, on, dsteps);
        else k_exact_a<0><<<grid, block, lds_a, stream>>>(OccA<0>{x->bwt.va}, qb, qo, nq, n, ol, on, dsteps);
    } else if (x->bwt.flat && x->bwt.search_family() != FAM_A && !(kernel_flags() & (1 << 21))) {
        const uint32_t entries = x->bwt.flat_nsb * (uint32_t)x->bwt.sigma;
        size_t lds_max = kFlatSuperLdsMax;
        if (const char* ev = dev_env("FMGPU_DEV_FLAT_SUPER_MAX")) lds_max = (size_t)atoll(ev);      // (dev knob: 0 = the super table is read through L2)
        const uint32_t super_lds = flat_super_lds_bytes(entries) <= lds_max ? entries : 0u;      // (sigma = 28: 6.7 KB at 2 x 10^9 rows, 19 KB at 4.5 x 10^9, 42 KB at 10^10)
        uint32_t lanes = flat_block_lanes(super_lds);
        if (const char* ev = dev_env("FMGPU_DEV_FLAT_LANES")) { const uint32_t v = (uint32_t)atoi(ev); if ((v == 256 || v == 512) && flat_lds_bytes(v, super_lds) <= kLdsPerCu) lanes = v; }     // (dev knob)
        const size_t lds = flat_lds_bytes(lanes, super_lds) + dev_extra_lds;
        auto launch = [&](auto kernel) -> int {
            if (lds > 64 * 1024) FM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));     // (more than a launch gets without asking)
            dim3 g; if (int grc = grid_of((nq + lanes - 1) / lanes * 256u, &g)) return grc;
            kernel<<<g, dim3(lanes), lds, stream>>>(x->bwt.flat, x->bwt.flat_super, (uint32_t)x->bwt.sigma, flat_count_bits((uint32_t)x->bwt.sigma),
                                                     flat_lut ? (const void*)x->bwt.slut : nullptr, x->bwt.slut_len, (const uint8_t*)sbuf.dev,
                                                     (const uint64_t*)soff.dev, nq, n, (uint64_t*)slb.dev, (uint64_t*)slen.dev, dsteps, super_lds);
            return 0;
        };
        if ((rc = lanes == 256 ? launch(k_exact_s<256>) : launch(k_exact_s<512>))) return rc;
    } else if (x->bwt.search_family() == FAM_WAVELET) {
        uint32_t mx = shape_max, mn = 0;
        if (!have_shape && (rc = query_len_range((const uint64_t*)soff.dev, nq, stream, &mx, &mn))) return rc;
        uint32_t qw = (mx + 3) / 4;
        if ((size_t)qw * 1024 > 48 * 1024) qw = 0;
        timer.start();
        const ViewM& vm = x->bwt.vm;
        const uint32_t m_super_rows = kWide && x->bwt.sup_bytes ? (uint32_t)std::min<uint64_t>(0xffffffffu, (x->bwt.sup_bytes - ((uint64_t)vm.nnodes * 4 + 63) / 64 * 64) / 64) : 0u;
        auto qb = (const uint8_t*)sbuf.dev; auto qo = (const uint64_t*)soff.dev; auto ol = (uint64_t*)slb.dev; auto on = (uint64_t*)slen.dev;
        const size_t lds = (size_t)qw * 1024;
        switch (vm.bitct) {                                      // the digits of digits_of() as template arguments
        case 1: k_exact_m<1, 0, 0><<<grid, block, lds + dev_extra_lds, stream>>>(vm, qb, qo, nq, n, ol, on, dsteps, qw, m_super_rows); break;
        case 2: k_exact_m<2, 0, 0><<<grid, block, lds + dev_extra_lds, stream>>>(vm, qb, qo, nq, n, ol, on, dsteps, qw, m_super_rows); break;
        case 3: k_exact_m<3, 0, 0><<<grid, block, lds + dev_extra_lds, stream>>>(vm, qb, qo, nq, n, ol, on, dsteps, qw, m_super_rows); break;
        case 4: k_exact_m<2, 2, 0><<<grid, block, lds + dev_extra_lds, stream>>>(vm, qb, qo, nq, n, ol, on, dsteps, qw, m_super_rows); break;
        case 5: k_exact_m<3, 2, 0><<<grid, block, lds + dev_extra_lds, stream>>>(vm, qb, qo, nq, n, ol, on, dsteps, qw, m_super_rows); break;
        case 6: k_exact_m<3, 3, 0><<<grid, block, lds + dev_extra_lds, stream>>>(vm, qb, qo, nq, n, ol, on, dsteps, qw, m_super_rows); break;
        case 7: k_exact_m<3, 2, 2><<<grid, block, lds + dev_extra_lds, stream>>>(vm, qb, qo, nq, n, ol, on, dsteps, qw, m_super_rows); break;
        default: k_exact_m<3, 3, 2><<<grid, block, lds + dev_extra_lds, stream>>>(vm, qb, qo, nq, n, ol, on, dsteps, qw, m_super_rows); break;
        }
    } else {
        rc = dispatch_occ(x->bwt, [&](auto occ, auto) {
            k_exact<decltype(occ)><<<grid, block, 0, stream>>>(occ, (const uint8_t*)sbuf.dev, (const uint64_t*)soff.dev, nq, n,
                                                              (uint64_t*)slb.dev, (uint64_t*)slen.dev, dsteps);
            return 0;
        });
    }
    timer.stop();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "k_exact launch");
    if (stats) {
        unsigned long long hs[kCounterKinds] = {0, 0, 0, 0};
        if ((rc = read_step_counters(dsteps, stream, hs))) return rc;
        stats->lf_steps = hs[0]; stats->hits = nq; stats->kernel_ms = timer.ms();
        stats->table_bytes = hs[1]; stats->table_accesses = hs[2]; stats->table_steps = hs[3];
    }
    if ((rc = slb.finish())) return rc;
    if (out_len && (rc = slen.finish())) return rc;
    if (stats || slb.owned || slen.owned) (void)hipStreamSynchronize(stream);
    return 0;
}

int fmgpu_search_exact(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq,
                       uint64_t* out_lb, uint64_t* out_len, fmgpu_stats* stats, void* stream) {
    return search_exact(h, qbuf, qoff, nq, out_lb, out_len, false, stats, stream);
}

int fmgpu_search_exact_packed(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq,
                              uint64_t* out_interval, fmgpu_stats* stats, void* stream) {
    return search_exact(h, qbuf, qoff, nq, out_interval, nullptr, true, stats, stream);
}

int fmgpu_search_exact_depth(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, uint32_t* out_depth, void* stream_) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (int drc = on_handle_device(x)) return drc;
    if (nq == 0) return 0;
    if (!qbuf || !qoff || !out_depth) return fail(FMGPU_ERR_INVALID, "qbuf / qoff / out_depth is null");
    hipStream_t stream = (hipStream_t)stream_;
    Staged soff, sbuf, sout;
    int rc;
    if ((rc = soff.in(qoff, (nq + 1) * 8, stream))) return rc;
    uint64_t total = 0;
    if (is_device_pointer(qoff)) { FM_HIP(hipMemcpyAsync(&total, qoff + nq, 8, hipMemcpyDeviceToHost, stream)); FM_HIP(hipStreamSynchronize(stream)); }
    else total = qoff[nq];
    if ((rc = sbuf.in(qbuf, total, stream)) || (rc = sout.out(out_depth, nq * 4, stream))) return rc;
    FM_GRID(grid, nq);
    rc = dispatch_occ(x->bwt, [&](auto occ, auto) {
        k_exact_depth<decltype(occ)><<<grid, dim3(256), 0, stream>>>(occ, (const uint8_t*)sbuf.dev, (const uint64_t*)soff.dev, nq, (idx_t)x->bwt.n, (uint32_t*)sout.dev);
        return 0;
    });
    FM_LAUNCHED("k_exact_depth");
    return sout.finish();
}

}  // namespace api
}  // namespace FMGPU_NS

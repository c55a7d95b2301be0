// fmgpu_search.hip — the hot path: one query per lane.
//
//  k_exact         search_no_errors::search            (search/SearchNoErrors.h:12-26)
//  k_scheme        search_ng26::search<Edit=false>     (search/SearchNg26.h:18-433, Hamming reduction: SURVEY.md appendix A)
//  k_backtracking  search_backtracking::search         (search/Backtracking.h:42-102)
//  k_locate        FMIndex::locate / BiFMIndex::locate (fmindex/FMIndex.h:113-124)
//
// The two DFS kernels are flat state machines: every loop iteration performs exactly one memory phase per lane
// (the occurrence-table blocks at both interval ends, Occ::all2) followed by register-only control logic, so lanes
// that sit in different branches of the reference's recursion (extend-all node, exact tail, single-row fast path,
// resumed sibling) still issue their gathers together.  Pending siblings of a branching node live in a per-lane
// stack in HBM (one frame per query position at most, lane-interleaved), and are re-derived from the parent
// cursor when popped; children with an empty interval are never pushed (the reference returns from them at once).
#include "fmgpu_common.h"

#include <algorithm>
#include <cstdlib>

namespace fmgpu {

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

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
        out_lb[q] = lb; out_len[q] = len;
    }
    uint32_t tot = wave_sum(steps);
    if ((threadIdx.x & 63u) == 0 && tot) atomicAdd(steps_total, (unsigned long long)tot);
}

// ---- exact search, tuned variants (Format A only) ---------------------------------------------------------------
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

template <int SIGMA, int VARIANT>
__global__ __launch_bounds__(256) void k_exact_a(OccA<SIGMA> occ, const uint8_t* __restrict__ qbuf, const uint64_t* __restrict__ qoff,
                                                 uint64_t nq, idx_t n, uint64_t* __restrict__ out_lb, uint64_t* __restrict__ out_len,
                                                 unsigned long long* __restrict__ steps_total) {
    uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t steps = 0;
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
                EntryA ea = load_entry_a(occ.v.blk, occ.v.bstride, a, c);
                EntryA eb = ea;
                if (VARIANT < 2 || (a >> 6) != (b >> 6)) eb = load_entry_a(occ.v.blk, occ.v.bstride, b, c);
                idx_t ra = ea.cnt + popc64(ea.bits & lowmask(a & 63u));
                idx_t rb = eb.cnt + popc64(eb.bits & lowmask(b & 63u));
                lb = ra; len = rb - ra;
                if (len == 0) break;
            }
        }
        out_lb[q] = lb; out_len[q] = len;
    }
    uint32_t tot = wave_sum(steps);
    if ((threadIdx.x & 63u) == 0 && tot) atomicAdd(steps_total, (unsigned long long)tot);
}

// two queries per lane, interleaved: twice the loads in flight per wave
template <int SIGMA>
__global__ __launch_bounds__(256) void k_exact_a2(OccA<SIGMA> occ, const uint8_t* __restrict__ qbuf, const uint64_t* __restrict__ qoff,
                                                  uint64_t nq, idx_t n, uint64_t* __restrict__ out_lb, uint64_t* __restrict__ out_len,
                                                  unsigned long long* __restrict__ steps_total) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t half = (nq + 1) / 2;
    uint32_t steps = 0;
    if (t < half) {
        const uint64_t q0 = t, q1 = t + half;
        const bool has1 = q1 < nq;
        const uint32_t sigma = occ.sigma();
        uint64_t o0 = qoff[q0], o1 = has1 ? qoff[q1] : 0;
        uint32_t m0 = (uint32_t)(qoff[q0 + 1] - o0), m1 = has1 ? (uint32_t)(qoff[q1 + 1] - o1) : 0;
        idx_t lb0 = 0, len0 = n, lb1 = 0, len1 = n;
        QueryReader r0, r1;
        if (m0) r0.init(qbuf, o0, m0);
        if (m1) r1.init(qbuf, o1, m1);
        bool live0 = m0 != 0, live1 = m1 != 0;
        uint32_t i0 = 0, i1 = 0;
        while (live0 || live1) {
            uint32_t c0 = 0, c1 = 0;
            if (live0) { c0 = r0.next(); ++steps; if (c0 >= sigma) { lb0 = 0; len0 = 0; live0 = false; } }
            if (live1) { c1 = r1.next(); ++steps; if (c1 >= sigma) { lb1 = 0; len1 = 0; live1 = false; } }
            EntryA ea0{}, eb0{}, ea1{}, eb1{};
            const idx_t a0 = lb0, b0 = lb0 + len0, a1 = lb1, b1 = lb1 + len1;
            if (live0) { ea0 = load_entry_a(occ.v.blk, occ.v.bstride, a0, c0); eb0 = ea0; if ((a0 >> 6) != (b0 >> 6)) eb0 = load_entry_a(occ.v.blk, occ.v.bstride, b0, c0); }
            if (live1) { ea1 = load_entry_a(occ.v.blk, occ.v.bstride, a1, c1); eb1 = ea1; if ((a1 >> 6) != (b1 >> 6)) eb1 = load_entry_a(occ.v.blk, occ.v.bstride, b1, c1); }
            if (live0) {
                idx_t ra = ea0.cnt + popc64(ea0.bits & lowmask(a0 & 63u)), rb = eb0.cnt + popc64(eb0.bits & lowmask(b0 & 63u));
                lb0 = ra; len0 = rb - ra; ++i0;
                if (len0 == 0 || i0 == m0) live0 = false;
            }
            if (live1) {
                idx_t ra = ea1.cnt + popc64(ea1.bits & lowmask(a1 & 63u)), rb = eb1.cnt + popc64(eb1.bits & lowmask(b1 & 63u));
                lb1 = ra; len1 = rb - ra; ++i1;
                if (len1 == 0 || i1 == m1) live1 = false;
            }
        }
        out_lb[q0] = lb0; out_len[q0] = len0;
        if (has1) { out_lb[q1] = lb1; out_len[q1] = len1; }
    }
    uint32_t tot = wave_sum(steps);
    if ((threadIdx.x & 63u) == 0 && tot) atomicAdd(steps_total, (unsigned long long)tot);
}

// ------------------------------------------------------------------ DFS machinery
constexpr int kMaxParts = 16;
constexpr int kMaxSearches = 16;

struct SchemeDev {             // flattened [search][part]; values fit a byte (errors <= 255, parts <= 16)
    int S, P;
    uint8_t pi[kMaxSearches * kMaxParts], l[kMaxSearches * kMaxParts], u[kMaxSearches * kMaxParts];
    uint32_t partition[kMaxParts];   // used when uniform == 0
    uint32_t psum;                   // sum of partition[] (queries of another length are skipped)
    int uniform;
};

struct Counters { unsigned long long hits, nodes, next; };

// lane-interleaved frame stack: frame d of lane g at word (d * nlanes + g) of three u64 planes
struct StackView { uint64_t *p0, *p1, *p2; uint64_t nlanes; uint32_t depth; };

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

constexpr uint32_t kNoResume = 0xffffffffu;

// ---- search_ng26 Hamming --------------------------------------------------------------------------------------
template <class Occ, int MAXSIG>
__global__ __launch_bounds__(256) void k_scheme(Occ fw, Occ rv, SchemeDev sch, const uint8_t* __restrict__ qbuf,
                                                const uint64_t* __restrict__ qoff, uint64_t nq, idx_t n, uint64_t max_hits,
                                                fmgpu_hit* __restrict__ out, uint64_t cap, Counters* ctr, StackView stk) {
    __shared__ uint8_t s_pi[kMaxSearches * kMaxParts], s_l[kMaxSearches * kMaxParts], s_u[kMaxSearches * kMaxParts];
    __shared__ uint32_t s_part[kMaxParts];
    for (int i = threadIdx.x; i < kMaxSearches * kMaxParts; i += blockDim.x) { s_pi[i] = sch.pi[i]; s_l[i] = sch.l[i]; s_u[i] = sch.u[i]; }
    if (threadIdx.x < kMaxParts) s_part[threadIdx.x] = sch.partition[threadIdx.x];
    __syncthreads();

    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t sigma = fw.sigma();
    const int P = sch.P, S = sch.S;
    uint32_t nodes = 0;

    for (uint64_t q = gid; q < nq; q += stk.nlanes) {
        const uint64_t qo = qoff[q];
        const uint32_t m = (uint32_t)(qoff[q + 1] - qo);
        const uint8_t* qs = qbuf + qo;
        if (m < (uint32_t)P || m > stk.depth) continue;          // expand.h:325-327 precondition (the reference asserts)
        if (!sch.uniform && m != sch.psum) continue;             // an explicit partition must cover the query exactly
        uint64_t quota = max_hits;
        uint32_t seq = 0;
        bool query_done = false;
        auto part_len = [&](uint32_t p) -> uint32_t {             // createUniformPartition, expand.h:324-335
            return sch.uniform ? (m / (uint32_t)P + (p < m % (uint32_t)P ? 1u : 0u)) : s_part[p];
        };
        for (int si = 0; si < S && !query_done; ++si) {
            const uint8_t* pi = s_pi + si * kMaxParts; const uint8_t* L = s_l + si * kMaxParts; const uint8_t* U = s_u + si * kMaxParts;
            // run(): SearchNg26.h:62-79
            Cur cur{0, 0, n};
            uint32_t e = 0, part = 0, qL = 0, qR = 0, pev, tail = 0, sp = 0, resume = kNoResume;
            for (uint32_t i = 0; i < pi[0]; ++i) { qL += part_len(i); qR += part_len(i); }
            qL -= 1;                                               // may wrap; not read until it is valid again
            pev = part_len(pi[0]);
            bool right = true;                                     // part == 0 -> Right
            bool running = cur.len != 0;
            // invariant at the loop head: a STEP state (cur.len > 0, part < P, `right` set), possibly a resumed frame
            while (running) {
                const Occ& occ = right ? rv : fw;
                const idx_t a = right ? cur.lbRev : cur.lb;
                idx_t lfa[MAXSIG], lfb[MAXSIG];
                occ.template all2<MAXSIG>(a, a + cur.len, lfa, lfb);   // the memory phase
                const uint32_t c = qs[right ? qR : qL];
                const SymSet<MAXSIG> alive = alive_set<MAXSIG>(lfa, lfb, sigma);
                bool back = false, advance = false, to_next = false;
                if (tail) {                                        // search_next_dir_no_errors, :225-250 (one extension per iteration)
                    ++nodes;
                    if (!alive.test(c)) back = true;
                    else {
                        cur = kid_of<MAXSIG>(lfa, lfb, cur, c, right, sigma);
                        if (right) ++qR; else --qL;
                        if (--tail == 0) { ++part; pev = part != (uint32_t)P ? part_len(pi[part]) : 0; to_next = true; }
                    }
                } else {
                    const bool mOK = (pev > 1 || L[part] <= e) && e <= U[part];
                    const bool sOK = (pev > 1 || L[part] <= e + 1) && e + 1 <= U[part];
                    const bool xOK = e + 1 <= U[part];
                    // node accounting mirrors the reference's work: one per extend-all / extend; the single-row path
                    // extends once before it decides (:267-277) and once more per exact-tail step
                    if (cur.len > 1) {                             // search_next_dir, :143-224
                        if (resume == kNoResume && (xOK || mOK)) ++nodes;
                        if (xOK) {
                            SymSet<MAXSIG> subs = alive;           // substitution children: FirstSymb = 1 (fmindex/BiFMIndex.h:26), != c
                            if (!sOK) subs.clear();
                            subs.remove(0); subs.remove(c);
                            bool match = mOK && alive.test(c);
                            if (resume != kNoResume) { subs.clear_below(resume); match = false; }
                            if (!match && !subs.any()) back = true;
                            else {
                                uint32_t take = c;
                                if (!match) { take = subs.first(); subs.remove(take); }
                                if (subs.any()) {                  // (re-)push the parent: its remaining siblings start at subs.first()
                                    uint64_t w0 = (uint64_t)cur.lb | ((uint64_t)cur.lbRev << 32);
                                    uint64_t w1 = (uint64_t)cur.len | ((uint64_t)(pev & 0xffffu) << 32) | ((uint64_t)(qR & 0xffffu) << 48);
                                    uint64_t w2 = (uint64_t)subs.first() | ((uint64_t)(e & 0xffu) << 32) | ((uint64_t)(part & 0x7fu) << 40) |
                                                  ((uint64_t)(right ? 1u : 0u) << 47) | ((uint64_t)((qL + 1u) & 0xffffu) << 48);
                                    uint64_t o = (uint64_t)sp * stk.nlanes + gid;
                                    stk.p0[o] = w0; stk.p1[o] = w1; stk.p2[o] = w2;
                                    ++sp;
                                }
                                cur = kid_of<MAXSIG>(lfa, lfb, cur, take, right, sigma);
                                if (!match) e += 1;
                                advance = true;
                            }
                        } else if (mOK) {                          // exact tail; this iteration's blocks serve its first extension
                            if (!alive.test(c)) back = true;
                            else {
                                cur = kid_of<MAXSIG>(lfa, lfb, cur, c, right, sigma);
                                if (right) ++qR; else --qL;
                                tail = pev - 1;
                                if (tail == 0) { ++part; pev = part != (uint32_t)P ? part_len(pi[part]) : 0; to_next = true; }
                            }
                        } else back = true;
                    } else {                                       // search_next_dir_single, :251-365: the one alive kid is the BWT symbol
                        const uint32_t b = alive.first();
                        ++nodes;
                        if (!alive.any() || b < 1) back = true;
                        else if (b == c) {
                            if (!mOK) back = true;
                            else if (!xOK) {                       // exact tail from here
                                ++nodes;
                                cur = kid_of<MAXSIG>(lfa, lfb, cur, c, right, sigma);
                                if (right) ++qR; else --qL;
                                tail = pev - 1;
                                if (tail == 0) { ++part; pev = part != (uint32_t)P ? part_len(pi[part]) : 0; to_next = true; }
                            } else { cur = kid_of<MAXSIG>(lfa, lfb, cur, b, right, sigma); advance = true; }
                        } else if (xOK && sOK) { cur = kid_of<MAXSIG>(lfa, lfb, cur, b, right, sigma); e += 1; advance = true; }
                        else back = true;
                    }
                }
                resume = kNoResume;
                if (advance) {                                     // search_next_pos, :119-141 (children are never empty here)
                    if (right) ++qR; else --qL;
                    if (--pev == 0) { ++part; if (part != (uint32_t)P) pev = part_len(pi[part]); to_next = true; }
                }
                if (to_next) {                                     // search_next, :98-117
                    if (part == (uint32_t)P) {
                        if (L[P - 1] <= e && e <= U[P - 1]) {      // delegate with search_n clipping, :412-420
                            Cur r = cur;
                            if ((uint64_t)r.len > quota) r.len = (idx_t)quota;
                            quota -= r.len;
                            emit_hit(out, cap, ctr, q, r, e, seq++);
                            if (quota == 0) { query_done = true; break; }
                        }
                        back = true;
                    } else {
                        right = pi[part - 1] < pi[part];
                    }
                }
                if (back) {
                    if (sp == 0) break;
                    --sp;
                    uint64_t o = (uint64_t)sp * stk.nlanes + gid;
                    uint64_t w0 = stk.p0[o], w1 = stk.p1[o], w2 = stk.p2[o];
                    cur.lb = (idx_t)w0; cur.lbRev = (idx_t)(w0 >> 32);
                    cur.len = (idx_t)w1; pev = (uint32_t)(w1 >> 32) & 0xffffu; qR = (uint32_t)(w1 >> 48) & 0xffffu;
                    resume = (uint32_t)w2; e = (uint32_t)(w2 >> 32) & 0xffu; part = (uint32_t)(w2 >> 40) & 0x7fu;
                    right = (w2 >> 47) & 1u;
                    qL = ((uint32_t)(w2 >> 48) & 0xffffu) - 1u;
                    tail = 0;
                }
            }
        }
    }
    uint32_t tot = wave_sum(nodes);
    if ((threadIdx.x & 63u) == 0 && tot) atomicAdd(&ctr->nodes, (unsigned long long)tot);
}

// ---- search_backtracking ----------------------------------------------------------------------------------------
template <class Occ, int MAXSIG>
__global__ __launch_bounds__(256) void k_backtracking(Occ fw, bool bidir, const uint8_t* __restrict__ qbuf, const uint64_t* __restrict__ qoff,
                                                      uint64_t nq, idx_t n, uint32_t K, fmgpu_hit* __restrict__ out, uint64_t cap,
                                                      Counters* ctr, StackView stk) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t sigma = fw.sigma();
    uint32_t nodes = 0;
    for (uint64_t q = gid; q < nq; q += stk.nlanes) {
        const uint64_t qo = qoff[q];
        const uint32_t m = (uint32_t)(qoff[q + 1] - qo);
        const uint8_t* qs = qbuf + qo;
        if (m > stk.depth) continue;
        Cur cur{0, 0, n};
        uint32_t e = 0, i = 0, sp = 0, seq = 0, resume = kNoResume;
        bool running = n != 0;                                       // Backtracking.h:43: empty cursor
        while (running) {
            if (i == m && resume == kNoResume) {                     // :63 / :76 report
                emit_hit(out, cap, ctr, q, cur, e, seq++);
            } else {
                idx_t lfa[MAXSIG], lfb[MAXSIG];
                fw.template all2<MAXSIG>(cur.lb, cur.lb + cur.len, lfa, lfb);
                if (resume == kNoResume) ++nodes;
                const uint32_t r = qs[m - i - 1];
                const SymSet<MAXSIG> alive = alive_set<MAXSIG>(lfa, lfb, sigma);
                SymSet<MAXSIG> subs = alive;                         // :52-56: s in [1, sigma), s != r, while e < K
                if (e >= K) subs.clear();
                subs.remove(0); subs.remove(r);
                if (resume != kNoResume) subs.clear_below(resume);
                resume = kNoResume;
                const bool match = alive.test(r);
                if (subs.any()) {
                    uint32_t s = subs.first();
                    subs.remove(s);
                    if (subs.any() || match) {                       // something is still pending at this node
                        uint64_t o = (uint64_t)sp * stk.nlanes + gid;
                        stk.p0[o] = (uint64_t)cur.lb | ((uint64_t)cur.lbRev << 32);
                        stk.p1[o] = (uint64_t)cur.len | ((uint64_t)i << 32);
                        stk.p2[o] = (uint64_t)(subs.any() ? subs.first() : 256u) | ((uint64_t)e << 32);
                        ++sp;
                    }
                    cur = kid_of<MAXSIG>(lfa, lfb, cur, s, false, sigma);
                    if (!bidir) cur.lbRev = 0;
                    e += 1; i += 1;
                    continue;
                }
                if (match) {                                         // :58-61 / :70-74 continue with the query symbol
                    cur = kid_of<MAXSIG>(lfa, lfb, cur, r, false, sigma);
                    if (!bidir) cur.lbRev = 0;
                    i += 1;
                    continue;
                }
            }
            // dead end or reported: resume the innermost pending node
            if (sp == 0) break;
            --sp;
            uint64_t o = (uint64_t)sp * stk.nlanes + gid;
            uint64_t w0 = stk.p0[o], w1 = stk.p1[o], w2 = stk.p2[o];
            cur.lb = (idx_t)w0; cur.lbRev = (idx_t)(w0 >> 32); cur.len = (idx_t)w1; i = (uint32_t)(w1 >> 32);
            resume = (uint32_t)w2; e = (uint32_t)(w2 >> 32);
        }
    }
    uint32_t tot = wave_sum(nodes);
    if ((threadIdx.x & 63u) == 0 && tot) atomicAdd(&ctr->nodes, (unsigned long long)tot);
}

// ------------------------------------------------------------------ locate
constexpr uint32_t kLocateStepCap = 1u << 24;   // a valid index reaches a sampled row long before; bounds a corrupt one

template <class Occ>
__global__ __launch_bounds__(256) void k_locate(Occ occ, ViewSA sa, const uint64_t* __restrict__ rows, uint64_t count, idx_t n,
                                                uint64_t* __restrict__ out_seq, uint64_t* __restrict__ out_pos, uint64_t* __restrict__ out_steps,
                                                unsigned long long* __restrict__ steps_total) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t steps = 0;
    if (t < count) {
        uint64_t r64 = rows[t];
        uint64_t seq = ~0ull, pos = ~0ull, st = ~0ull;
        if (r64 < n) {
            idx_t row = (idx_t)r64;
            while (!sa_present(sa, row) && steps < kLocateStepCap) {    // fmindex/FMIndex.h:116-121
                uint32_t c;
                row = occ.lf_symbol(row, c);
                ++steps;
            }
            if (sa_present(sa, row)) {
                uint64_t k = sa_rank(sa, row);                          // suffixarray/SparseArray.h:63-70
                seq = dense_access(sa.f0, sa.bits0, sa.div0, k);
                pos = dense_access(sa.f1, sa.bits1, sa.div1, k);
                st = steps;
            }
        }
        out_seq[t] = seq; out_pos[t] = pos; out_steps[t] = st;
    }
    uint32_t tot = wave_sum(steps);
    if ((threadIdx.x & 63u) == 0 && tot) atomicAdd(steps_total, (unsigned long long)tot);
}

// ------------------------------------------------------------------ host launchers
struct EventTimer {
    hipEvent_t a = nullptr, b = nullptr; hipStream_t s; bool on;
    EventTimer(hipStream_t s_, bool on_) : s(s_), on(on_) { if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); } }
    void start() { if (on) (void)hipEventRecord(a, s); }
    void stop() { if (on) (void)hipEventRecord(b, s); }
    float ms() { float v = 0; if (on) { (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&v, a, b); } return v; }
    ~EventTimer() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};

static int query_max_len(const uint64_t* qoff_host_or_dev, uint64_t nq, hipStream_t stream, uint32_t* out);

template <class F>
static int dispatch_occ(const DevString& s, F&& f) {
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
        if (s.sigma <= 32) return f(OccW{s.vw}, std::integral_constant<int, 32>{});
        return f(OccW{s.vw}, std::integral_constant<int, 256>{});
    }
}

// max query length: a tiny reduction kernel (queries may live in HBM)
__global__ __launch_bounds__(256) void k_max_len(const uint64_t* __restrict__ qoff, uint64_t nq, unsigned long long* __restrict__ out) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = 0;
    for (uint64_t q = t; q < nq; q += (uint64_t)gridDim.x * blockDim.x) { unsigned long long l = qoff[q + 1] - qoff[q]; v = l > v ? l : v; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { unsigned long long o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    if ((threadIdx.x & 63u) == 0) atomicMax(out, v);
}

static int query_max_len(const uint64_t* dqoff, uint64_t nq, hipStream_t stream, uint32_t* out) {
    unsigned long long* d = nullptr;
    FM_HIP(hipMalloc((void**)&d, 8));
    FM_HIP(hipMemsetAsync(d, 0, 8, stream));
    unsigned blocks = (unsigned)std::min<uint64_t>((nq + 255) / 256, 1024);
    k_max_len<<<dim3(blocks), dim3(256), 0, stream>>>(dqoff, nq, d);
    unsigned long long h = 0;
    hipError_t e = hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d);
    if (e != hipSuccess) return hip_fail(e, "k_max_len");
    *out = (uint32_t)std::min<unsigned long long>(h, 0xffffffffull);
    return 0;
}

struct DfsWorkspace {
    uint64_t* planes = nullptr; Counters* ctr = nullptr; StackView view{};
    unsigned grid = 0;
    int init(uint32_t depth, uint64_t nq, hipStream_t stream) {
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        uint64_t want = (uint64_t)cus * 8;                                  // 8 blocks of 256 lanes per CU = full occupancy
        grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(want, (nq + 255) / 256));
        view.nlanes = (uint64_t)grid * 256; view.depth = depth;
        uint64_t words = view.nlanes * ((uint64_t)depth + 1);
        FM_HIP(hipMalloc((void**)&planes, words * 8 * 3));
        view.p0 = planes; view.p1 = planes + words; view.p2 = planes + 2 * words;
        FM_HIP(hipMalloc((void**)&ctr, sizeof(Counters)));
        FM_HIP(hipMemsetAsync(ctr, 0, sizeof(Counters), stream));
        return 0;
    }
    ~DfsWorkspace() { if (planes) (void)hipFree(planes); if (ctr) (void)hipFree(ctr); }
};

}  // namespace fmgpu

using namespace fmgpu;

extern "C" {

int fmgpu_search_exact(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq,
                       uint64_t* out_lb, uint64_t* out_len, fmgpu_stats* stats, void* stream_) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (stats) *stats = fmgpu_stats{0, 0, 0.f};
    if (nq == 0) return 0;
    if (!qbuf || !qoff || !out_lb || !out_len) return fail(FMGPU_ERR_INVALID, "qbuf / qoff / out_lb / out_len is null");
    hipStream_t stream = (hipStream_t)stream_;
    Staged soff, sbuf, slb, slen;
    int rc;
    if ((rc = soff.in(qoff, (nq + 1) * 8, stream))) return rc;
    uint64_t total = 0;
    if (is_device_pointer(qoff)) { FM_HIP(hipMemcpyAsync(&total, qoff + nq, 8, hipMemcpyDeviceToHost, stream)); FM_HIP(hipStreamSynchronize(stream)); }
    else total = qoff[nq];
    if ((rc = sbuf.in(qbuf, total, stream))) return rc;
    if ((rc = slb.out(out_lb, nq * 8, stream))) return rc;
    if ((rc = slen.out(out_len, nq * 8, stream))) return rc;
    unsigned long long* dsteps = nullptr;
    FM_HIP(hipMalloc((void**)&dsteps, 8));
    FM_HIP(hipMemsetAsync(dsteps, 0, 8, stream));
    EventTimer timer(stream, stats != nullptr);
    dim3 grid((unsigned)((nq + 255) / 256)), block(256);
    const idx_t n = (idx_t)x->bwt.n;
    const int variant = [] { const char* e = getenv("FMGPU_EXACT_VARIANT"); return e ? atoi(e) : 2; }();   // dev knob
    timer.start();
    if (x->bwt.family == FAM_A && variant >= 1) {
        auto qb = (const uint8_t*)sbuf.dev; auto qo = (const uint64_t*)soff.dev; auto ol = (uint64_t*)slb.dev; auto on = (uint64_t*)slen.dev;
        auto launch = [&](auto occ) {
            using O = decltype(occ);
            constexpr int SG = std::is_same_v<O, OccA<5>> ? 5 : 0;
            if (variant == 1) k_exact_a<SG, 1><<<grid, block, 0, stream>>>(occ, qb, qo, nq, n, ol, on, dsteps);
            else if (variant == 2) k_exact_a<SG, 2><<<grid, block, 0, stream>>>(occ, qb, qo, nq, n, ol, on, dsteps);
            else k_exact_a2<SG><<<dim3((unsigned)(((nq + 1) / 2 + 255) / 256)), block, 0, stream>>>(occ, qb, qo, nq, n, ol, on, dsteps);
        };
        if (x->bwt.sigma == 5) launch(OccA<5>{x->bwt.va}); else launch(OccA<0>{x->bwt.va});
    } else {
        rc = dispatch_occ(x->bwt, [&](auto occ, auto) {
            k_exact<decltype(occ)><<<grid, block, 0, stream>>>(occ, (const uint8_t*)sbuf.dev, (const uint64_t*)soff.dev, nq, n,
                                                              (uint64_t*)slb.dev, (uint64_t*)slen.dev, dsteps);
            return 0;
        });
    }
    timer.stop();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { (void)hipFree(dsteps); return hip_fail(e, "k_exact launch"); }
    if (stats) {
        unsigned long long hs = 0;
        e = hipMemcpyAsync(&hs, dsteps, 8, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) { (void)hipFree(dsteps); return hip_fail(e, "k_exact"); }
        stats->lf_steps = hs; stats->hits = nq; stats->kernel_ms = timer.ms();
    }
    if ((rc = slb.finish())) { (void)hipFree(dsteps); return rc; }
    if ((rc = slen.finish())) { (void)hipFree(dsteps); return rc; }
    if (stats || slb.owned || slen.owned) (void)hipStreamSynchronize(stream);
    (void)hipFree(dsteps);   // hipFree synchronises the device: the counter is no longer in use afterwards
    return 0;
}

static int run_dfs(Index* x, bool scheme_mode, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, const fmgpu_scheme* scheme,
                   uint64_t max_hits, uint32_t K, fmgpu_hit* out, uint64_t capacity, uint64_t* out_count, fmgpu_stats* stats, hipStream_t stream) {
    if (stats) *stats = fmgpu_stats{0, 0, 0.f};
    if (out_count) *out_count = 0;
    if (nq == 0) return 0;
    if (!qbuf || !qoff || (!out && capacity) || !out_count) return fail(FMGPU_ERR_INVALID, "qbuf / qoff / out / out_count is null");
    SchemeDev sd{};
    if (scheme_mode) {
        if (!x->bidirectional) return fail(FMGPU_ERR_INVALID, "search_ng26 needs a BiFMIndex (bwt_rev)");
        if (!scheme || !scheme->pi || !scheme->l || !scheme->u) return fail(FMGPU_ERR_INVALID, "scheme is null");
        if (scheme->n_searches < 0 || scheme->n_searches > kMaxSearches || scheme->n_parts < 1 || scheme->n_parts > kMaxParts)
            return fail(FMGPU_ERR_UNSUPPORTED, "scheme larger than 16 searches x 16 parts");
        if (max_hits == 0 || scheme->n_searches == 0) return 0;                       // SearchNg26.h:408-409
        sd.S = scheme->n_searches; sd.P = scheme->n_parts; sd.uniform = scheme->partition ? 0 : 1;
        for (int s = 0; s < sd.S; ++s) {
            uint32_t seen = 0;
            for (int p = 0; p < sd.P; ++p) {
                uint64_t pi = scheme->pi[s * sd.P + p], l = scheme->l[s * sd.P + p], u = scheme->u[s * sd.P + p];
                if (pi >= (uint64_t)sd.P || l > 255 || u > 254) return fail(FMGPU_ERR_INVALID, "scheme entry out of range");
                seen |= 1u << pi;
                sd.pi[s * kMaxParts + p] = (uint8_t)pi; sd.l[s * kMaxParts + p] = (uint8_t)l; sd.u[s * kMaxParts + p] = (uint8_t)u;
            }
            if (seen != (1u << sd.P) - 1u) return fail(FMGPU_ERR_INVALID, "scheme pi is not a permutation of the parts");
            // connectivity (search_scheme/isValid.h:18-33): the kernel's cursor only grows at its two ends
            uint32_t lo = sd.pi[s * kMaxParts], hi = lo;
            for (int p = 1; p < sd.P; ++p) {
                uint32_t v = sd.pi[s * kMaxParts + p];
                if (v == hi + 1) hi = v; else if (v + 1 == lo) lo = v; else return fail(FMGPU_ERR_INVALID, "scheme pi is not contiguous");
            }
        }
        if (scheme->partition) for (int p = 0; p < sd.P; ++p) {
            if (scheme->partition[p] == 0 || scheme->partition[p] > 0xffffu) return fail(FMGPU_ERR_INVALID, "partition entries must be in [1, 65535]");
            sd.partition[p] = (uint32_t)scheme->partition[p];
            sd.psum += sd.partition[p];
        }
    }
    Staged soff, sbuf, sout;
    int rc;
    if ((rc = soff.in(qoff, (nq + 1) * 8, stream))) return rc;
    uint64_t total = 0;
    if (is_device_pointer(qoff)) { FM_HIP(hipMemcpyAsync(&total, qoff + nq, 8, hipMemcpyDeviceToHost, stream)); FM_HIP(hipStreamSynchronize(stream)); }
    else total = qoff[nq];
    if ((rc = sbuf.in(qbuf, total, stream))) return rc;
    if ((rc = sout.out(out, capacity * sizeof(fmgpu_hit), stream))) return rc;
    uint32_t maxlen = 0;
    if ((rc = query_max_len((const uint64_t*)soff.dev, nq, stream, &maxlen))) return rc;
    if (maxlen > 0xfffeu) return fail(FMGPU_ERR_UNSUPPORTED, "queries longer than 65534 symbols");
    DfsWorkspace ws;
    if ((rc = ws.init(maxlen, nq, stream))) return rc;
    EventTimer timer(stream, stats != nullptr);
    const idx_t n = (idx_t)x->bwt.n;
    dim3 grid(ws.grid), block(256);
    timer.start();
    if (scheme_mode) {
        const DevString& rv = x->rev;
        rc = dispatch_occ(x->bwt, [&](auto occ, auto ms) {
            using O = decltype(occ);
            O r{};
            if constexpr (std::is_same_v<O, OccA<5>> || std::is_same_v<O, OccA<0>>) r = O{rv.va};
            else if constexpr (std::is_same_v<O, OccW>) r = O{rv.vw};
            else r = O{rv.vr};
            k_scheme<O, decltype(ms)::value><<<grid, block, 0, stream>>>(occ, r, sd, (const uint8_t*)sbuf.dev, (const uint64_t*)soff.dev, nq, n,
                                                                         max_hits, (fmgpu_hit*)sout.dev, capacity, ws.ctr, ws.view);
            return 0;
        });
    } else {
        rc = dispatch_occ(x->bwt, [&](auto occ, auto ms) {
            k_backtracking<decltype(occ), decltype(ms)::value><<<grid, block, 0, stream>>>(occ, x->bidirectional, (const uint8_t*)sbuf.dev,
                                                                                           (const uint64_t*)soff.dev, nq, n, K, (fmgpu_hit*)sout.dev,
                                                                                           capacity, ws.ctr, ws.view);
            return 0;
        });
    }
    timer.stop();
    FM_HIP(hipGetLastError());
    Counters hc{};
    FM_HIP(hipMemcpyAsync(&hc, ws.ctr, sizeof hc, hipMemcpyDeviceToHost, stream));
    FM_HIP(hipStreamSynchronize(stream));
    *out_count = hc.hits;
    if (stats) { stats->lf_steps = hc.nodes; stats->hits = hc.hits; stats->kernel_ms = timer.ms(); }
    if (hc.hits > capacity) {
        if (sout.writeback) { sout.bytes = capacity * sizeof(fmgpu_hit); (void)sout.finish(); }
        return fail(FMGPU_ERR_CAPACITY, "result buffer holds " + std::to_string(capacity) + " records, " + std::to_string(hc.hits) + " produced");
    }
    if (sout.writeback) sout.bytes = hc.hits * sizeof(fmgpu_hit);
    return sout.finish();
}

int fmgpu_search_scheme(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, const fmgpu_scheme* scheme,
                        uint64_t max_hits_per_query, fmgpu_hit* out, uint64_t capacity, uint64_t* out_count, fmgpu_stats* stats, void* stream) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    return run_dfs(x, true, qbuf, qoff, nq, scheme, max_hits_per_query, 0, out, capacity, out_count, stats, (hipStream_t)stream);
}

int fmgpu_search_backtracking(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, uint64_t max_errors,
                              fmgpu_hit* out, uint64_t capacity, uint64_t* out_count, fmgpu_stats* stats, void* stream) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (max_errors > 254) return fail(FMGPU_ERR_INVALID, "max_errors > 254");
    return run_dfs(x, false, qbuf, qoff, nq, nullptr, ~0ull, (uint32_t)max_errors, out, capacity, out_count, stats, (hipStream_t)stream);
}

int fmgpu_locate(fmgpu_index_t h, const uint64_t* rows, uint64_t count, uint64_t* out_seq, uint64_t* out_pos, uint64_t* out_steps,
                 fmgpu_stats* stats, void* stream_) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (!x->has_sa) return fail(FMGPU_ERR_INVALID, "index was created without an annotated (sampled suffix) array");
    if (stats) *stats = fmgpu_stats{0, 0, 0.f};
    if (count == 0) return 0;
    if (!rows || !out_seq || !out_pos || !out_steps) return fail(FMGPU_ERR_INVALID, "rows / outputs is null");
    hipStream_t stream = (hipStream_t)stream_;
    Staged srows, sseq, spos, sst;
    int rc;
    if ((rc = srows.in(rows, count * 8, stream))) return rc;
    if ((rc = sseq.out(out_seq, count * 8, stream))) return rc;
    if ((rc = spos.out(out_pos, count * 8, stream))) return rc;
    if ((rc = sst.out(out_steps, count * 8, stream))) return rc;
    unsigned long long* dsteps = nullptr;
    FM_HIP(hipMalloc((void**)&dsteps, 8));
    FM_HIP(hipMemsetAsync(dsteps, 0, 8, stream));
    EventTimer timer(stream, stats != nullptr);
    dim3 grid((unsigned)((count + 255) / 256)), block(256);
    const idx_t n = (idx_t)x->bwt.n;
    timer.start();
    rc = dispatch_occ(x->bwt, [&](auto occ, auto) {
        k_locate<decltype(occ)><<<grid, block, 0, stream>>>(occ, x->vsa, (const uint64_t*)srows.dev, count, n, (uint64_t*)sseq.dev,
                                                           (uint64_t*)spos.dev, (uint64_t*)sst.dev, dsteps);
        return 0;
    });
    timer.stop();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { (void)hipFree(dsteps); return hip_fail(e, "k_locate launch"); }
    if (stats) {
        unsigned long long hs = 0;
        e = hipMemcpyAsync(&hs, dsteps, 8, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) { (void)hipFree(dsteps); return hip_fail(e, "k_locate"); }
        stats->lf_steps = hs; stats->hits = count; stats->kernel_ms = timer.ms();
    }
    rc = sseq.finish(); if (!rc) rc = spos.finish(); if (!rc) rc = sst.finish();
    (void)hipStreamSynchronize(stream);
    (void)hipFree(dsteps);
    return rc;
}

}  // extern "C"

// fmgpu_search.hip — the search-scheme searches (depth-first walks of the bidirectional index), one query per lane, and the hit-record helpers.
//
//  k_scheme_lean       search_ng26::search<Edit=false> on the plain sigma = 5 index (equal-length batches; top / bottom frames in LDS, hit ring per wave, 2-bit reads)
//  k_scheme_fast       the same with tables: per-step table, prefix table, LF..LF^3 and LF^16 walk tables (search/SearchNg26.h:18-433)
//  k_scheme_fast_edit  search_ng26::search<Edit=true>, the same frame with the insertion / deletion branches (:146-218, :286-362); top frame in LDS (write-back)
//  k_scheme, k_scheme_edit, k_ng21   the general forms (ragged small batches, explicit partitions, every layout, search_ng21): flat state machines
//  k_backtracking      search_backtracking::search         (search/Backtracking.h:42-102)
//  (exact search: fmgpu_exact.hip; locate: fmgpu_locate.hip; what they share: fmgpu_search_shared.h)
//
// The general DFS kernels are flat state machines: every loop iteration performs exactly one memory phase per lane (the occurrence-table
// blocks at both interval ends, Occ::all2, or one LF-table load for a one-row cursor) followed by register-only control logic.  Pending
// siblings of a branching node live in a per-lane stack in HBM and are re-derived from the parent cursor when popped; children with an empty
// interval are never pushed (the reference returns from them at once).  Queries are handed out and hit records written by whole waves.
#include "fmgpu_search_shared.h"

namespace FMGPU_NS {

// ---- search_ng26 Hamming --------------------------------------------------------------------------------------
template <class Occ, int MAXSIG>
// (5 resident blocks — 4 with 64-bit rows — is what the LDS of a 101-symbol batch allows anyway; 6 spilled 45 registers once the sharing state came in)
__global__ __launch_bounds__(256, MAXSIG <= 5 ? (kWide ? 4 : 5) : 1) void k_scheme(Occ fw, Occ rv, SchemeDev sch, const uint8_t* __restrict__ qbuf,
                                                const uint64_t* __restrict__ qoff, uint64_t nq, idx_t n, uint64_t max_hits,
                                                fmgpu_hit* __restrict__ out, uint64_t cap, Counters* ctr, StackView stk, uint32_t qwords, uint32_t qnib, LfView lfv,
                                                const uint32_t* __restrict__ order) {   // order (or null): the batch is handed out in this order (heavy reads first)
    extern __shared__ uint32_t s_query[];
    const QStage qst{s_query, qwords, qnib};
    __shared__ uint8_t s_pi[kMaxSearches * kMaxParts], s_l[kMaxSearches * kMaxParts], s_u[kMaxSearches * kMaxParts];
    __shared__ uint32_t s_part[kMaxParts];
    for (int i = threadIdx.x; i < kMaxSearches * kMaxParts; i += blockDim.x) { s_pi[i] = sch.pi[i]; s_l[i] = sch.l[i]; s_u[i] = sch.u[i]; }
    if (threadIdx.x < kMaxParts) s_part[threadIdx.x] = sch.partition[threadIdx.x];
    __shared__ idx_t s_C[257];
    if (lfv.fw) for (uint32_t i = threadIdx.x; i <= fw.sigma(); i += blockDim.x) s_C[i] = lfv.C[i];
    __syncthreads();

    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t sigma = fw.sigma();
    const uint32_t P = (uint32_t)sch.P, S = (uint32_t)sch.S;
    uint32_t nodes = 0;
    uint64_t tbytes = 0; uint32_t tacc = 0;                         // table bytes consumed / table accesses issued (fmgpu_stats; blocks, LF entries, frames)

    // One flat loop over (query, search, node): a lane that finishes a search starts its next search — or its next
    // query — in the same iteration the other lanes of the wave spend on a node, so the wave never waits for its
    // slowest search.  The node logic is written branch-light: the reference's case analysis (exact tail / extend-all
    // node / single-row fast path / resumed sibling) is folded into a few predicates that pick ONE child symbol
    // `take`; the child cursor is then computed once for all cases.
    // Queries are handed out and hits written out by whole waves (see wave_hand_out / wave_flush_hits above).
    const uint32_t lane = threadIdx.x & 63u;
    __shared__ uint32_t s_hb[kWaveHitWords];
    uint32_t nh = 0;
    wave_ring_init(s_hb);
    uint64_t q = 0;
    uint32_t si = 0;                        // current search
    bool idle = false, have_query = false, fresh = false;
    const uint8_t* qs = qbuf; uint32_t m = 0, pbase = 0, prem = 0;
    uint64_t quota = 0; uint32_t seq = 0;
    Cur cur{0, 0, 0};
    uint32_t e = 0, part = 0, qL = 0, qR = 0, pev = 0, tail = 0, sp = 0, resume = kNoResume;
    bool right = true;
    const uint8_t *pi = s_pi, *L = s_l, *U = s_u;
    auto part_len = [&](uint32_t p) -> uint32_t {                  // createUniformPartition, expand.h:324-335
        return sch.uniform ? pbase + (p < prem ? 1u : 0u) : s_part[p];
    };
    bool need_search = true;
    // work sharing at the end of the batch (see k_scheme_fast): once the queries are handed out, a lane that is out of work takes the bottom
    // frame of a busy lane of its wave; the hits of a read are ordered by path keys then (Hamming: <= 2 substitutions)
    bool is_task = false;
    uint32_t sbase = 0, mark = 0;
    uint64_t pkey = 0;
    auto frame_words = [&](uint64_t& w0, uint64_t& w1, uint64_t& w2, uint32_t nxt) {      // the node the lane stands on as a frame; nxt = its next child
        if constexpr (kWide) {                                     // rows of up to 40 bits: one cursor field per word
            w0 = (uint64_t)cur.lb | ((uint64_t)(pev & 0xffffu) << 40) | ((uint64_t)(e & 0xffu) << 56);
            w1 = (uint64_t)cur.lbRev | ((uint64_t)(qR & 0xffffu) << 40) | ((uint64_t)(nxt & 0xffu) << 56);
            w2 = (uint64_t)cur.len | ((uint64_t)((qL + 1u) & 0xffffu) << 40) | ((uint64_t)(part & 0x7fu) << 56) | ((uint64_t)(right ? 1u : 0u) << 63);
        } else {
            w0 = (uint64_t)cur.lb | ((uint64_t)cur.lbRev << 32);
            w1 = (uint64_t)cur.len | ((uint64_t)(pev & 0xffffu) << 32) | ((uint64_t)(qR & 0xffffu) << 48);
            w2 = (uint64_t)nxt | ((uint64_t)(e & 0xffu) << 32) | ((uint64_t)(part & 0x7fu) << 40) |
                 ((uint64_t)(right ? 1u : 0u) << 47) | ((uint64_t)((qL + 1u) & 0xffffu) << 48);
        }
    };
    auto frame_take = [&](uint64_t w0, uint64_t w1, uint64_t w2) {  // stand on a frame's node again
        if constexpr (kWide) {
            const uint64_t m40 = (1ull << 40) - 1ull;
            cur.lb = (idx_t)(w0 & m40); pev = (uint32_t)(w0 >> 40) & 0xffffu; e = (uint32_t)(w0 >> 56) & 0xffu;
            cur.lbRev = (idx_t)(w1 & m40); qR = (uint32_t)(w1 >> 40) & 0xffffu; resume = (uint32_t)(w1 >> 56) & 0xffu;
            cur.len = (idx_t)(w2 & m40); qL = ((uint32_t)(w2 >> 40) & 0xffffu) - 1u; part = (uint32_t)(w2 >> 56) & 0x7fu; right = (w2 >> 63) & 1u;
        } else {
            cur.lb = (idx_t)w0; cur.lbRev = (idx_t)(w0 >> 32);
            cur.len = (idx_t)w1; pev = (uint32_t)(w1 >> 32) & 0xffffu; qR = (uint32_t)(w1 >> 48) & 0xffffu;
            resume = (uint32_t)w2; e = (uint32_t)(w2 >> 32) & 0xffu; part = (uint32_t)(w2 >> 40) & 0x7fu;
            right = (w2 >> 47) & 1u;
            qL = ((uint32_t)(w2 >> 48) & 0xffffu) - 1u;
        }
        tail = 0;
    };
    for (;;) {
        // ---- wave-synchronous part: all 64 lanes pass here in every iteration
        if (sch.sharing) {
            if (is_task && need_search) { is_task = false; idle = true; }      // a task is one subtree of one search
            const bool offer = !idle && !need_search && sp > sbase && nodes - mark >= kShareNodes;
            const uint64_t idlem = __ballot(idle), offerm = __ballot(offer);
            if (idlem && offerm) {
                const WavePairs wp(idlem, offerm, lane);
                const bool give = wp.gives(offer);
                const bool take = wp.takes(idle);
                uint64_t w0 = 0, w1 = 0, w2 = 0, wk = 0;
                if (give) {
                    const uint64_t o = (uint64_t)sbase * stk.nlanes + gid;
                    w0 = stk.p0[o]; w1 = stk.p1[o]; w2 = stk.p2[o];
                    wk = key_prefix(pkey, kWide ? ((uint32_t)(w0 >> 56) & 0xffu) : ((uint32_t)(w2 >> 32) & 0xffu));
                    ++sbase; mark = nodes; tbytes += 24u; ++tacc;
                }
                const int vl = wp.partner(take);                       // my partner: the lane whose frame this lane takes
                const uint64_t t0 = __shfl(w0, vl, 64), t1 = __shfl(w1, vl, 64), t2 = __shfl(w2, vl, 64), tk = __shfl(wk, vl, 64);
                const uint64_t tq = __shfl(q, vl, 64), tqs = __shfl((uint64_t)qs, vl, 64);
                const uint32_t tsi = __shfl(si, vl, 64), tm = __shfl(m, vl, 64);
                if (take) {
                    q = tq; qs = reinterpret_cast<const uint8_t*>(tqs); si = tsi; m = tm;
                    pbase = m / P; prem = m - pbase * P;
                    pi = s_pi + si * kMaxParts; L = s_l + si * kMaxParts; U = s_u + si * kMaxParts;
                    frame_take(t0, t1, t2);
                    pkey = tk;
                    const uint32_t vt = (threadIdx.x & ~63u) | (uint32_t)vl;
                    for (uint32_t w = 0; w < qwords; ++w) s_query[w * 256u + threadIdx.x] = s_query[w * 256u + vt];      // the partner's staged query
                    idle = false; is_task = true; need_search = false; have_query = true; fresh = false;
                    quota = max_hits; seq = 0; sp = 0; sbase = 0; mark = nodes;
                }
            }
        }
        const bool want_q = !idle && need_search && !(have_query && si + 1 < S && quota != 0);   // search_impl / search_n_impl, SearchNg26.h:369-391, :407-423
        const uint64_t got = wave_hand_out(want_q, ctr, lane);
        if (want_q) {
            q = got;
            have_query = false;
            if (q >= nq) idle = true;
            else {
                if (order) q = order[q];
                const uint64_t qo = qoff[q];
                m = (uint32_t)(qoff[q + 1] - qo);
                qs = qbuf + qo;
                // expand.h:325-327 precondition (the reference asserts); an explicit partition must cover the query exactly
                if (m >= P && m <= stk.depth && (sch.uniform || m == sch.psum) && n != 0) { have_query = true; fresh = true; }
            }
        }
        const bool full = wave_ring_fill(s_hb) >= kWaveRingFlush; const uint64_t busy = __ballot(!idle);
        if (full || !busy) wave_flush_hits(s_hb, nh, lane, out, cap, ctr);
        if (!busy) break;
        if (idle) continue;
        if (need_search) {
            if (!have_query) continue;                             // the query fetched was unusable: the next iteration fetches another
            if (fresh) {
                fresh = false; si = 0; quota = max_hits; seq = 0;
                pbase = m / P; prem = m - pbase * P;
                qstage_load(qst, qbuf, qoff[q], m, sigma);
            } else ++si;
            pi = s_pi + si * kMaxParts; L = s_l + si * kMaxParts; U = s_u + si * kMaxParts;
            // run(): SearchNg26.h:62-79
            cur = Cur{0, 0, n};
            e = 0; part = 0; qL = 0; qR = 0; tail = 0; sp = 0; resume = kNoResume;
            sbase = 0; mark = nodes; pkey = (uint64_t)si << 48;
            for (uint32_t i = 0; i < pi[0]; ++i) { uint32_t pl = part_len(i); qL += pl; qR += pl; }
            qL -= 1;                                               // may wrap; not read until it is valid again
            pev = part_len(pi[0]);
            right = true;                                          // part == 0 -> Right
            need_search = false;
        }
        // invariant here: a STEP state (cur.len > 0, part < P, `right` set), possibly a resumed frame
        const Occ& occ = right ? rv : fw;
        const idx_t a = right ? cur.lbRev : cur.lb;
        idx_t lfa[MAXSIG], lfb[MAXSIG];
        const bool via_lf = lfv.fw != nullptr && cur.len == 1;      // one row: its only child comes from the LF table
        idx_t lf1 = 0;
        if (via_lf) { lf1 = (right ? lfv.rv : lfv.fw)[a]; tbytes += (uint32_t)sizeof(idx_t); ++tacc; }
        else {
            occ.template all2<MAXSIG>(a, a + cur.len, lfa, lfb);        // the memory phase
            const bool same = (a >> 6) == ((a + cur.len) >> 6);
            tbytes += (MAXSIG <= 5 ? 64u : 12u * sigma) * (same ? 1u : 2u); tacc += same ? 1u : 2u;
        }
        const uint32_t c = qstage_get(qst, qs, right ? qR : qL);
        SymSet<MAXSIG> alive;
        if (via_lf) { alive.clear(); alive.insert(symbol_of_lf_lds(s_C, sigma, lf1)); }
        else alive = alive_set<MAXSIG>(lfa, lfb, sigma);
        const bool c_alive = alive.test(c);

        // ---- case analysis -> (ok, take, is_sub, start_tail, push) ------------------------------------------------
        const bool in_tail = tail != 0;                            // search_next_dir_no_errors, :225-250
        const bool multi = cur.len > 1;                            // search_next_dir (:143-224) vs search_next_dir_single (:251-365)
        const bool resuming = resume != kNoResume;
        const uint32_t Lp = L[part], Up = U[part];
        const bool mOK = (pev > 1 || Lp <= e) && e <= Up;
        const bool sOK = (pev > 1 || Lp <= e + 1) && e + 1 <= Up;
        const bool xOK = e + 1 <= Up;
        SymSet<MAXSIG> subs = alive;                               // substitution children of an extend-all node:
        subs.remove(0); subs.remove(c);                            //   FirstSymb = 1 (fmindex/BiFMIndex.h:26), != query symbol
        if (in_tail || !multi || !xOK || !sOK) subs.clear();
        if (resuming) subs.clear_below(resume);
        const uint32_t b = alive.first();                          // single row: the one alive child is the BWT symbol
        const bool single_ok = !in_tail && !multi && alive.any() && b >= 1;
        const bool take_match = in_tail ? c_alive
                              : multi   ? (!resuming && mOK && c_alive)                   // match child first (also the exact-tail start when !xOK)
                                        : (single_ok && b == c && mOK);
        const bool take_sub = !take_match && (multi ? subs.any() : (single_ok && b != c && xOK && sOK));
        const bool ok = take_match || take_sub;
        uint32_t take = c;
        if (take_sub) { take = multi ? subs.first() : b; }
        if (take_sub && multi) subs.remove(take);
        if (take_match && multi && !xOK) subs.clear();
        const bool start_tail = !in_tail && take_match && !xOK;    // :225 from dir (:222) or from single (:310-314)
        // node accounting mirrors the reference's work (one per extend-all / extend; the single-row path extends once
        // before deciding, :267-277, and once more per exact-tail step)
        nodes += in_tail ? 1u : multi ? ((!resuming && (xOK || mOK)) ? 1u : 0u) : (1u + ((take_match && !xOK) ? 1u : 0u));
        if (ok && multi && !in_tail && subs.any()) {               // (re-)push the parent: its remaining siblings start at subs.first()
            uint64_t w0, w1, w2;
            frame_words(w0, w1, w2, subs.first());
            uint64_t o = (uint64_t)sp * stk.nlanes + gid;
            stk.p0[o] = w0; stk.p1[o] = w1; stk.p2[o] = w2;
            ++sp; tbytes += 24u; ++tacc;
        }
        resume = kNoResume;
        bool back = !ok, to_next = false;
        if (ok) {
            if (via_lf) cur = right ? Cur{cur.lb, lf1, 1} : Cur{lf1, cur.lbRev, 1};
            else cur = kid_of<MAXSIG>(lfa, lfb, cur, take, right, sigma);
            if (take_sub) { if (sch.use_key) pkey = key_with(pkey, e, m, qR - qL - 1u, take); e += 1; }   // (qR - qL - 1 = symbols consumed so far = the step)
            if (right) ++qR; else --qL;                            // one query symbol consumed (search_next_pos :122-124 / tail)
            if (in_tail) { to_next = --tail == 0; if (to_next) { ++part; pev = part != P ? part_len(pi[part]) : 0; } }
            else if (start_tail) { tail = pev - 1; to_next = tail == 0; if (to_next) { ++part; pev = part != P ? part_len(pi[part]) : 0; } }
            else { to_next = --pev == 0; if (to_next) { ++part; if (part != P) pev = part_len(pi[part]); } }
        }
        if (to_next) {                                             // search_next, :98-117
            if (part == P) {
                if (L[P - 1] <= e && e <= U[P - 1]) {              // delegate with search_n clipping, :412-420
                    Cur r = cur;
                    if ((uint64_t)r.len > quota) r.len = (idx_t)quota;
                    quota -= r.len;
                    if (sch.dev_flags & 1) ++seq;
                    else if (sch.use_key) { wave_keep_hit(s_hb, nh, out, cap, ctr, q, r, e | ((uint32_t)(pkey >> 32) << 8), (uint32_t)pkey); ++seq; }
                    else wave_keep_hit(s_hb, nh, out, cap, ctr, q, r, e, seq++);
                    if (quota == 0) { need_search = true; continue; }     // delegate returned true: skip the remaining searches
                }
                back = true;
            } else {
                right = pi[part - 1] < pi[part];
            }
        }
        if (back) {
            if (sp == sbase) { need_search = true; continue; }
            --sp;
            uint64_t o = (uint64_t)sp * stk.nlanes + gid;
            uint64_t w0 = stk.p0[o], w1 = stk.p1[o], w2 = stk.p2[o];
            tbytes += 24u; ++tacc;
            frame_take(w0, w1, w2);
            pkey = key_prefix(pkey, e);
        }
    }
    uint32_t tot = wave_sum(nodes);
    const unsigned long long tb = wave_sum64(tbytes); const uint32_t ta = wave_sum(tacc);
    if ((threadIdx.x & 63u) == 0 && tot) {
        atomicAdd(&ctr->nodes, (unsigned long long)tot);
        atomicAdd(&ctr->table_bytes, tb); atomicAdd(&ctr->table_accesses, (unsigned long long)ta);
    }
}

// ---- edit path keys (described at k_scheme_fast_edit) ----
constexpr uint64_t kEditKeyNone = 0x400040004000ull;
__device__ __forceinline__ uint64_t ekey_prefix(uint64_t key, uint32_t e) {   // the key of an ancestor that had made e errors
    const uint64_t keep = e == 0u ? 0ull : (e == 1u ? 0xffff00000000ull : (e == 2u ? 0xffffffff0000ull : 0xffffffffffffull));
    return (key & (0xfull << 48)) | (key & keep) | (kEditKeyNone & ~keep);
}
__device__ __forceinline__ uint64_t ekey_with(uint64_t key, uint32_t e_before, bool before_match, uint32_t depth, uint32_t code) {
    if (e_before >= 3u) return key;
    const uint32_t sh = 32u - 16u * e_before;
    const uint64_t comp = before_match ? (uint64_t)((depth << 6) | code) : (uint64_t)(0x8000u | ((255u - depth) << 6) | code);
    return (key & ~(0xffffull << sh)) | (comp << sh);
}

// ---- edit-distance frames (k_scheme_edit, k_ng21): four row-sized fields and four 32-bit words, the frames of a lane consecutive in memory — a DFS
// pushes and pops them in order, so four (32-bit rows: 32 bytes each; 64-bit rows: 48) share a line or two, where interleaving them by lane
// would touch one line per frame (the lanes of a wave sit at different depths)
constexpr uint32_t kEditFrameQuads = kWide ? 3u : 2u;
constexpr int kEditFramePlanes = kWide ? 6 : 4;                  // (in 8-byte planes of the DFS workspace)
__device__ __forceinline__ uint4* edit_frame(const StackView& stk, uint64_t gid, uint32_t slot) {
    return reinterpret_cast<uint4*>(stk.p0) + (size_t)kEditFrameQuads * (gid * ((uint64_t)stk.depth + 1u) + slot);
}
__device__ __forceinline__ void edit_frame_put(uint4* f, idx_t r0, idx_t r1, idx_t r2, idx_t r3, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    if constexpr (kWide) {
        f[0] = make_uint4((uint32_t)r0, (uint32_t)((uint64_t)r0 >> 32), (uint32_t)r1, (uint32_t)((uint64_t)r1 >> 32));
        f[1] = make_uint4((uint32_t)r2, (uint32_t)((uint64_t)r2 >> 32), (uint32_t)r3, (uint32_t)((uint64_t)r3 >> 32));
        f[2] = make_uint4(a, b, c, d);
    } else { f[0] = make_uint4((uint32_t)r0, (uint32_t)r1, (uint32_t)r2, a); f[1] = make_uint4(b, c, d, (uint32_t)r3); }
}
__device__ __forceinline__ void edit_frame_get(const uint4* f, idx_t& r0, idx_t& r1, idx_t& r2, idx_t& r3, uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
    if constexpr (kWide) {
        const uint4 x = f[0], y = f[1], z = f[2];
        r0 = (idx_t)((uint64_t)x.x | ((uint64_t)x.y << 32)); r1 = (idx_t)((uint64_t)x.z | ((uint64_t)x.w << 32));
        r2 = (idx_t)((uint64_t)y.x | ((uint64_t)y.y << 32)); r3 = (idx_t)((uint64_t)y.z | ((uint64_t)y.w << 32));
        a = z.x; b = z.y; c = z.z; d = z.w;
    } else { const uint4 x = f[0], y = f[1]; r0 = x.x; r1 = x.y; r2 = x.z; a = x.w; b = y.x; c = y.y; d = y.z; r3 = y.w; }
}
__device__ __forceinline__ void edit_frame_set_r2(uint4* f, idx_t v) {     // the third row field alone
    if constexpr (kWide) reinterpret_cast<uint64_t*>(f)[2] = (uint64_t)v; else reinterpret_cast<uint32_t*>(f)[2] = (uint32_t)v;
}

// ---- search_ng26, Edit = true (SearchNg26.h:143-224, :251-365 with insertions and deletions) ------------------------------------
// Same flat (query, search, node) loop and frame stack as k_scheme; a node's children are numbered in the reference's call order
//   several rows:  0 match | 2i-1 deletion of index symbol i | 2i substitution by i  (i = 1 .. sigma-1) | 2*sigma-1 insertion
//   one row:       0 insertion | 1 match or substitution (the row's symbol) | 2 deletion
// a frame keeps the parent and the number of its next child; the child cursor comes from the parent's extend-all, which is
// recomputed when the frame is resumed.  State beyond the Hamming kernel's: the last index / query symbol per side (:36-39) and
// the last operation per side (LInfo / RInfo: 0 M, 1 S, 2 I, 3 D).
constexpr int kEditWaves = 4;         // waves per SIMD the register allocation aims at (6 or 8 spill and measure the same)
// ---- the node step of search_ng26<Edit = true> (search/SearchNg26.h:143-365), shared by k_scheme_edit and k_scheme_fast_edit ------------------------------------------
// Which child of a node the walk takes next, the reference's call order written as child numbers.  A node of several rows (:143-224): 0 the match child, 2i - 1 the deletion
// of symbol i, 2i the substitution by i (i = 1 .. sigma - 1, != the query symbol), 2 sigma - 1 the insertion.  A node of ONE row (:251-365): 0 the insertion, 1 the match or
// the substitution by the row's symbol, 2 its deletion.  In an exact tail (:225-250) only the match child exists.  `start` = the first number still to try (a resumed node).
// kind: 0 match, 1 substitution, 2 deletion, 3 insertion, 4 nothing left; take = the index symbol of the child; nxt = the number of the following child (kNoResume: none — the
// parent need not be kept); code = the child's number (its component of the path key); start_tail: a match child that cannot afford an error any more starts the exact tail.
struct EditStep { uint32_t kind, take, nxt, code; bool start_tail; };
template <int MAXSIG>
__device__ __forceinline__ EditStep edit_next_child(const SymSet<MAXSIG>& alive, uint32_t c, bool in_tail, bool multi, bool resuming, uint32_t start, bool mOK, bool iOK, bool xOK,
                                                    bool Deletion, bool Insertion, uint32_t INS) {
    EditStep r{4u, c, kNoResume, 0u, false};
    const bool c_alive = alive.test(c);
    if (in_tail) { if (c_alive) r.kind = 0u; }
    else if (multi) {
        if (!xOK) { if (!resuming && mOK && c_alive) { r.kind = 0u; r.start_tail = true; } }
        else {
            SymSet<MAXSIG> dels = alive; dels.remove(0); if (!Deletion) dels.clear();
            SymSet<MAXSIG> subs = alive; subs.remove(0); subs.remove(c); if (!iOK) subs.clear();
            const bool insOK = Insertion && iOK;
            auto child_from = [&](uint32_t s0) -> uint32_t {                // number of the first existing child >= s0
                if (s0 == 0u && mOK && c_alive) return 0u;
                SymSet<MAXSIG> dd = dels, ss = subs;
                dd.clear_below(s0 <= 1u ? 1u : (s0 + 2u) >> 1);              // 2i - 1 >= s0
                ss.clear_below(s0 <= 2u ? 1u : (s0 + 1u) >> 1);              // 2i     >= s0
                uint32_t best = kNoResume;
                if (dd.any()) best = 2u * dd.first() - 1u;
                if (ss.any()) { uint32_t v = 2u * ss.first(); if (v < best) best = v; }
                if (best == kNoResume && insOK && s0 <= INS) best = INS;
                return best;
            };
            const uint32_t idx = child_from(start);
            if (idx != kNoResume) {
                if (idx == 0u) r.kind = 0u;
                else if (idx == INS) r.kind = 3u;
                else { r.kind = (idx & 1u) ? 2u : 1u; r.take = (idx + 1u) >> 1; }
                if (idx != INS) r.nxt = child_from(idx + 1u);
                r.code = idx;
            }
        }
    } else {
        const uint32_t b = alive.first();                           // the row's symbol (symbolLeft / symbolRight, :267-277)
        const bool valid = alive.any() && b >= 1u;                  // :298-301 only insertions below FirstSymb
        const bool same = valid && b == c;
        const bool en0 = Insertion && iOK;
        const bool en1 = same ? mOK : (valid && xOK && iOK);
        const bool en2 = Deletion && valid && xOK;
        uint32_t idx = kNoResume;
        if (start <= 0u && en0) idx = 0u; else if (start <= 1u && en1) idx = 1u; else if (start <= 2u && en2) idx = 2u;
        if (idx == 0u) { r.kind = 3u; r.nxt = en1 ? 1u : (en2 ? 2u : kNoResume); }
        else if (idx == 1u) {
            r.take = b;
            if (same) { r.kind = 0u; r.start_tail = !xOK; r.nxt = (en2 && xOK) ? 2u : kNoResume; }   // :310-314: the exact tail's result is returned as is
            else { r.kind = 1u; r.nxt = en2 ? 2u : kNoResume; }
        } else if (idx == 2u) { r.kind = 2u; r.take = b; }
        r.code = idx == kNoResume ? 0u : idx;
    }
    return r;
}
// nodes the reference counts for this visit (a resumed node was counted when it was first met; a one-row match child that starts the exact tail counts the tail's first node too)
__device__ __forceinline__ uint32_t edit_nodes_visited(bool in_tail, bool resuming, bool multi, bool xOK, bool mOK, bool start_tail) {
    return in_tail ? 1u : (resuming ? 0u : (multi ? ((xOK || mOK) ? 1u : 0u) : (1u + (start_tail ? 1u : 0u))));
}
// the edge taken, noted for the next node's rules (:146-163): side = lastRank | lastQRank per direction (8 bits each), info = the edge's kind per direction (2 bits: 0 match,
// 1 substitution, 2 insertion, 3 deletion); d = 0 left, 1 right
__device__ __forceinline__ void edit_note_edge(uint32_t kind, uint32_t take, uint32_t c, uint32_t d, uint32_t& side, uint32_t& info) {
    const uint32_t rmask = ~(255u << (8u * d)), qmask = ~(255u << (16u + 8u * d)), imask = ~(3u << (2u * d));
    if (kind == 0u) { side = (side & rmask & qmask) | (c << (8u * d)) | (c << (16u + 8u * d)); info = info & imask; }      // (in the exact tail the values written last survive, :236-237)
    else if (kind == 1u) { side = (side & rmask & qmask) | (take << (8u * d)) | (c << (16u + 8u * d)); info = (info & imask) | (1u << (2u * d)); }
    else if (kind == 2u) { side = (side & rmask) | (take << (8u * d)); info = (info & imask) | (3u << (2u * d)); }
    else { side = (side & qmask) | (c << (16u + 8u * d)); info = (info & imask) | (2u << (2u * d)); }
}

template <class Occ, int MAXSIG>
__global__ __launch_bounds__(256, MAXSIG <= 5 ? kEditWaves : 1) void k_scheme_edit(Occ fw, Occ rv, SchemeDev sch, const uint8_t* __restrict__ qbuf,
                                                     const uint64_t* __restrict__ qoff, uint64_t nq, idx_t n, uint64_t max_hits,
                                                     fmgpu_hit* __restrict__ out, uint64_t cap, Counters* ctr, StackView stk, uint32_t qwords, uint32_t qnib, LfView lfv, uint32_t maxm, const uint4* __restrict__ lut, uint32_t lutL,
                                                     const uint32_t* __restrict__ order) {
    extern __shared__ uint32_t s_query[];
    const QStage qst{s_query, qwords, qnib};
    __shared__ uint8_t s_pi[kMaxSearches * kMaxParts], s_l[kMaxSearches * kMaxParts], s_u[kMaxSearches * kMaxParts];
    __shared__ uint32_t s_part[kMaxParts];
    for (int i = threadIdx.x; i < kMaxSearches * kMaxParts; i += blockDim.x) { s_pi[i] = sch.pi[i]; s_l[i] = sch.l[i]; s_u[i] = sch.u[i]; }
    if (threadIdx.x < kMaxParts) s_part[threadIdx.x] = sch.partition[threadIdx.x];
    __shared__ idx_t s_C[257];
    if (lfv.fw) for (uint32_t i = threadIdx.x; i <= fw.sigma(); i += blockDim.x) s_C[i] = lfv.C[i];
    __syncthreads();

    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t sigma = fw.sigma();
    const uint32_t P = (uint32_t)sch.P, S = (uint32_t)sch.S;
    const uint32_t INS = 2u * sigma - 1u;
    uint32_t nodes = 0;
    const uint32_t lane = threadIdx.x & 63u;
    __shared__ uint32_t s_hb[kWaveHitWords];
    uint32_t nh = 0;
    wave_ring_init(s_hb);
    uint64_t q = 0;
    uint32_t si = 0;
    bool idle = false, have_query = false, fresh = false;
    const uint8_t* qs = qbuf; uint32_t m = 0, pbase = 0, prem = 0;
    uint64_t quota = 0; uint32_t seq = 0;
    Cur cur{0, 0, 0};
    uint32_t e = 0, part = 0, qL = 0, qR = 0, pev = 0, tail = 0, sp = 0, resume = kNoResume;
    uint32_t side = 0;                      // lastRank[left] | lastRank[right] << 8 | lastQRank[left] << 16 | lastQRank[right] << 24
    uint32_t info = 0;                      // LInfo | RInfo << 2
    idx_t cached_lf = 0, cached_lf2 = kNoRow;        // a resumed one-row frame: LF of its row, LF of that row (if a child reported it)
    bool lf_known = false; idx_t lf_val = 0;         // LF of the NEXT node's row is already known (same row after an insertion, or reported)
    uint32_t report_slot = kNoResume;                // frame whose deletion child will stand on this node's row: tell it this node's LF
    bool right = true;
    const uint8_t *pi = s_pi, *L = s_l, *U = s_u;
    auto part_len = [&](uint32_t p) -> uint32_t { return sch.uniform ? pbase + (p < prem ? 1u : 0u) : s_part[p]; };
    bool need_search = true;
    // path keys and work sharing at the end of the batch, as in k_scheme (keys: see k_scheme_fast_edit; ndel = deletions on the path, for the tree depth)
    bool is_task = false;
    uint32_t sbase = 0, mark = 0, ndel = 0;
    uint64_t pkey = 0;
    auto frame_take = [&](idx_t f_lb, idx_t f_rev, idx_t f_len, idx_t f_lf, uint32_t fa_, uint32_t fb_, uint32_t fc_, uint32_t f_side) {
        const uint64_t w2 = (uint64_t)fb_ | ((uint64_t)fc_ << 32);
        const bool one_row = ((uint32_t)w2 >> 20) & 1u;
        cur.lb = f_lb; cur.lbRev = f_rev; cur.len = one_row ? (idx_t)1 : f_len; cached_lf2 = one_row ? f_len : kNoRow; cached_lf = f_lf; side = f_side;
        pev = fa_ & 0xffffu; qR = fa_ >> 16;
        resume = (uint32_t)w2 & 0xffffu; info = ((uint32_t)w2 >> 16) & 15u; ndel = ((uint32_t)w2 >> 21) & 0xffu;
        e = (uint32_t)(w2 >> 32) & 0xffu; part = (uint32_t)(w2 >> 40) & 0x7fu;
        lf_known = false; report_slot = kNoResume;
        right = (w2 >> 47) & 1u;
        qL = ((uint32_t)(w2 >> 48) & 0xffffu) - 1u;
        tail = 0;
    };
    for (;;) {
        // ---- wave-synchronous part: all 64 lanes pass here in every iteration (a lane without work idles until the wave is done)
        if (sch.sharing) {
            if (is_task && need_search) { is_task = false; idle = true; }      // a task is one subtree of one search
            // (a frame whose running child still owes it the row's LF^2 — report_slot — stays with its owner for that one iteration)
            const bool offer = !idle && !need_search && sp > sbase && nodes - mark >= kShareNodes && report_slot != sbase;
            const uint64_t idlem = __ballot(idle), offerm = __ballot(offer);
            if (idlem && offerm) {
                const WavePairs wp(idlem, offerm, lane);
                const bool give = wp.gives(offer);
                const bool take = wp.takes(idle);
                idx_t g0 = 0, g1 = 0, g2 = 0, g3 = 0; uint32_t ga = 0, gb = 0, gc = 0, gd = 0; uint64_t gk = 0;
                if (give) {
                    edit_frame_get(edit_frame(stk, gid, sbase), g0, g1, g2, g3, ga, gb, gc, gd);
                    gk = ekey_prefix(pkey, gc & 0xffu);             // (e sits in the low byte of the frame's third word)
                    ++sbase; mark = nodes;
                }
                const int vl = wp.partner(take);                       // my partner: the lane whose frame this lane takes
                const uint64_t t0 = __shfl((uint64_t)g0, vl, 64), t1 = __shfl((uint64_t)g1, vl, 64), t2 = __shfl((uint64_t)g2, vl, 64), t3 = __shfl((uint64_t)g3, vl, 64);
                const uint32_t ta = __shfl(ga, vl, 64), tb = __shfl(gb, vl, 64), tc = __shfl(gc, vl, 64), td = __shfl(gd, vl, 64);
                const uint64_t tk = __shfl(gk, vl, 64), tq = __shfl(q, vl, 64), tqs = __shfl((uint64_t)qs, vl, 64);
                const uint32_t tsi = __shfl(si, vl, 64), tm = __shfl(m, vl, 64);
                if (take) {
                    q = tq; qs = reinterpret_cast<const uint8_t*>(tqs); si = tsi; m = tm;
                    pbase = m / P; prem = m - pbase * P;
                    pi = s_pi + si * kMaxParts; L = s_l + si * kMaxParts; U = s_u + si * kMaxParts;
                    frame_take((idx_t)t0, (idx_t)t1, (idx_t)t2, (idx_t)t3, ta, tb, tc, td);
                    pkey = tk;
                    const uint32_t vt = (threadIdx.x & ~63u) | (uint32_t)vl;
                    for (uint32_t w = 0; w < qwords; ++w) s_query[w * 256u + threadIdx.x] = s_query[w * 256u + vt];      // the partner's staged query
                    idle = false; is_task = true; need_search = false; have_query = true; fresh = false;
                    quota = max_hits; seq = 0; sp = 0; sbase = 0; mark = nodes;
                }
            }
        }
        const bool want_q = !idle && need_search && !(have_query && si + 1 < S && quota != 0);   // search_impl / search_n_impl, :369-391, :407-423
        const uint64_t got = wave_hand_out(want_q, ctr, lane);
        if (want_q) {
            q = got;
            have_query = false;
            if (q >= nq) idle = true;
            else {
                if (order) q = order[q];
                const uint64_t qo = qoff[q];
                m = (uint32_t)(qoff[q + 1] - qo);
                qs = qbuf + qo;
                // expand.h:325-327 precondition (the reference asserts); an explicit partition must cover the query exactly
                if (m >= P && m <= maxm && (sch.uniform || m == sch.psum) && n != 0) { have_query = true; fresh = true; }
            }
        }
        const bool full = wave_ring_fill(s_hb) >= kWaveRingFlush; const uint64_t busy = __ballot(!idle);
        if (full || !busy) wave_flush_hits(s_hb, nh, lane, out, cap, ctr);
        if (!busy) break;
        if (idle) continue;
        if (need_search) {
            if (!have_query) continue;                             // the query fetched was unusable: the next iteration fetches another
            if (fresh) {
                fresh = false; si = 0; quota = max_hits; seq = 0;
                pbase = m / P; prem = m - pbase * P;
                qstage_load(qst, qbuf, qoff[q], m, sigma);
            } else ++si;
            pi = s_pi + si * kMaxParts; L = s_l + si * kMaxParts; U = s_u + si * kMaxParts;
            cur = Cur{0, 0, n};                                    // run(): :62-79
            e = 0; part = 0; qL = 0; qR = 0; tail = 0; sp = 0; resume = kNoResume; side = 0; info = 0;
            sbase = 0; mark = nodes; ndel = 0; pkey = ((uint64_t)si << 48) | kEditKeyNone;
            for (uint32_t i = 0; i < pi[0]; ++i) { uint32_t pl = part_len(i); qL += pl; qR += pl; }
            qL -= 1;
            pev = part_len(pi[0]);
            right = true;
            need_search = false;
            if (lut && U[0] == 0 && pev > lutL && n > 1) {          // an always-exact first part starts from the prefix table (fmgpu_index_accelerate_search)
                uint32_t code = 0, mul = 1; bool valid = true;
                for (uint32_t t = 0; t < lutL; ++t) {
                    const uint32_t c = qstage_get(qst, qs, qR + t);
                    valid = valid && c >= 1 && c < sigma;
                    code += (c - 1) * mul; mul *= sigma - 1;
                }
                if (valid) {
                    const uint4 en = lut[code];
                    nodes += en.w;                                  // the extensions the reference performs before the interval is empty (:225-250)
                    if (en.z == 0) { need_search = true; continue; }
                    cur = Cur{en.x, en.y, en.z};
                    qR += lutL; tail = pev - lutL;                  // the rest of the part is an exact tail
                    const uint32_t lastc = qstage_get(qst, qs, qR - 1u);
                    side = (lastc << 8) | (lastc << 24);           // lastRank / lastQRank of the right side (:236-237 at the end of the tail anyway)
                }
            }
        }
        const Occ& occ = right ? rv : fw;
        const idx_t a = right ? cur.lbRev : cur.lb;
        idx_t lfa[MAXSIG], lfb[MAXSIG];
        const bool via_lf = lfv.fw != nullptr && cur.len == 1;      // one row: its only child comes from the LF table
        idx_t lf1 = cached_lf;                                      // a resumed one-row node brings its LF value along in the frame
        if (via_lf) {
            if (resume == kNoResume) {
                lf1 = lf_known ? lf_val : (right ? lfv.rv : lfv.fw)[a];
                if (report_slot != kNoResume) edit_frame_set_r2(edit_frame(stk, gid, report_slot), lf1);   // (the waiting deletion child of the parent starts from this same row)
            }
        } else occ.template all2<MAXSIG>(a, a + cur.len, lfa, lfb);
        lf_known = false; report_slot = kNoResume;
        const uint32_t c = qstage_get(qst, qs, right ? qR : qL);
        SymSet<MAXSIG> alive;
        if (via_lf) { alive.clear(); alive.insert(symbol_of_lf_lds(s_C, sigma, lf1)); }
        else alive = alive_set<MAXSIG>(lfa, lfb, sigma);
        const uint32_t d = right ? 1u : 0u;
        const uint32_t T = (info >> (2u * d)) & 3u;
        const uint32_t lastR = (side >> (8u * d)) & 255u, lastQ = (side >> (16u + 8u * d)) & 255u;
        const bool Deletion = T != 1u && T != 2u, Insertion = T != 1u && T != 3u;                      // :146-147
        const bool in_tail = tail != 0, multi = cur.len > 1, resuming = resume != kNoResume;
        const uint32_t Lp = L[part], Up = U[part];
        const bool mOK = (pev > 1 || Lp <= e) && e <= Up && (T != 2u || c != lastQ) && (T != 3u || c != lastR);   // :160-163
        const bool iOK = (pev > 1 || Lp <= e + 1) && e + 1 <= Up;                                        // insertion / substitution allowed
        const bool xOK = e + 1 <= Up;
        const uint32_t start = resuming ? resume : 0u;
        // kind: 0 match, 1 substitution, 2 deletion, 3 insertion, 4 nothing left; `take` = index symbol of the child; `nxt` = number of the following child
        const EditStep es = edit_next_child<MAXSIG>(alive, c, in_tail, multi, resuming, start, mOK, iOK, xOK, Deletion, Insertion, INS);
        const uint32_t kind = es.kind, take = es.take, nxt = es.nxt, code = es.code;
        const bool start_tail = es.start_tail;
        nodes += edit_nodes_visited(in_tail, resuming, multi, xOK, mOK, start_tail);
        if (kind != 4u && nxt != kNoResume) {                       // keep the parent: its remaining children start at nxt
            const uint64_t w2 = (uint64_t)(nxt | (info << 16) | ((via_lf ? 1u : 0u) << 20) | ((ndel & 0xffu) << 21)) | ((uint64_t)(e & 0xffu) << 32) | ((uint64_t)(part & 0x7fu) << 40) |
                                ((uint64_t)(right ? 1u : 0u) << 47) | ((uint64_t)((qL + 1u) & 0xffffu) << 48);
            // one-row frames (len = 1): the third row field holds LF(LF(row)) once the first child on that row has loaded it
            edit_frame_put(edit_frame(stk, gid, sp), cur.lb, cur.lbRev, via_lf ? kNoRow : cur.len, lf1, (pev & 0xffffu) | ((qR & 0xffffu) << 16), (uint32_t)w2, (uint32_t)(w2 >> 32), side);
            if (via_lf && nxt == 2u && (kind == 0u || kind == 1u)) report_slot = sp;
            ++sp;
        }
        resume = kNoResume;
        bool back = kind == 4u, to_next = false;
        if (kind != 4u) {
            if (kind != 3u) {
                if (via_lf) {
                    cur = right ? Cur{cur.lb, lf1, 1} : Cur{lf1, cur.lbRev, 1};
                    if (kind == 2u && resuming && cached_lf2 != kNoRow) { lf_known = true; lf_val = cached_lf2; }   // reported by the first child on that row
                } else cur = kid_of<MAXSIG>(lfa, lfb, cur, take, right, sigma);
            } else if (via_lf) { lf_known = true; lf_val = lf1; }    // insertion: the next node stands on the same row
            if (kind != 0u) {                                       // an error edge: one more component of the path key (depth = symbols consumed + deletions)
                if (sch.use_key) pkey = ekey_with(pkey, e, !multi && kind == 3u, (qR - qL - 1u) + ndel, code);
                e += 1;
                if (kind == 2u) ++ndel;
            }
            edit_note_edge(kind, take, c, d, side, info);
            if (kind != 2u) {                                       // NextPos: one query symbol consumed (:122-133)
                if (right) ++qR; else --qL;
                if (in_tail) { to_next = --tail == 0; if (to_next) { ++part; pev = part != P ? part_len(pi[part]) : 0; } }
                else if (start_tail) { tail = pev - 1; to_next = tail == 0; if (to_next) { ++part; pev = part != P ? part_len(pi[part]) : 0; } }
                else { to_next = --pev == 0; if (to_next) { ++part; if (part != P) pev = part_len(pi[part]); } }
            }
        }
        if (to_next) {                                             // search_next, :98-117
            if (part == P) {
                const uint32_t li = info & 3u, ri = (info >> 2) & 3u;
                if ((li == 0u || li == 2u) && (ri == 0u || ri == 2u) && L[P - 1] <= e && e <= U[P - 1]) {
                    Cur r = cur;
                    if ((uint64_t)r.len > quota) r.len = (idx_t)quota;
                    quota -= r.len;
                    if (sch.dev_flags & 1) ++seq;
                    else if (sch.use_key) { wave_keep_hit(s_hb, nh, out, cap, ctr, q, r, e | ((uint32_t)(pkey >> 32) << 8), (uint32_t)pkey); ++seq; }
                    else wave_keep_hit(s_hb, nh, out, cap, ctr, q, r, e, seq++);
                    if (quota == 0) { need_search = true; continue; }
                }
                back = true;
            } else {
                const bool nr = pi[part - 1] < pi[part];
                if (nr != right) { lf_known = false; report_slot = kNoResume; }   // the other index: another LF table
                right = nr;
            }
        }
        if (back) {
            if (sp == sbase) { need_search = true; continue; }
            --sp;
            idx_t f0 = 0, f1 = 0, f2 = 0, f3 = 0; uint32_t fa_ = 0, fb_ = 0, fc_ = 0, fd_ = 0;
            edit_frame_get(edit_frame(stk, gid, sp), f0, f1, f2, f3, fa_, fb_, fc_, fd_);
            frame_take(f0, f1, f2, f3, fa_, fb_, fc_, fd_);
            pkey = ekey_prefix(pkey, e);
        }
    }
    uint32_t tot = wave_sum(nodes);
    if ((threadIdx.x & 63u) == 0 && tot) atomicAdd(&ctr->nodes, (unsigned long long)tot);
}

// ---- search_ng21 (search/SearchNg21.h:26-156) -------------------------------------------------------------------
// Edit-distance search over an EXPANDED scheme: one table word per (search, step) = query position | l << 16 | u << 23 | right << 30
// (search_scheme/expand.h:146-165, prepare_reorder :184-200).  Differences from search_ng26<Edit = true> that decide the results: the lower
// bound applies at every step, deletion and substitution children exist only for symbols 1..sigma-1 other than the query's, one lastRank
// for both sides, the previous STEP's query symbol in place of lastQRank, no one-row path.  Children of a node are numbered in the
// reference's call order: 0 match, 2i-1 deletion of i, 2i substitution by i, 2*sigma-1 insertion.  Same wave-synchronous head, hit buffers and
// 32-byte lane-major frames as k_scheme_edit.
template <class Occ, int MAXSIG>
__global__ __launch_bounds__(256, MAXSIG <= 5 ? 5 : 1) void k_ng21(Occ fw, Occ rv, const uint32_t* __restrict__ tab, uint32_t S, uint32_t M,
                                                     const uint8_t* __restrict__ qbuf, const uint64_t* __restrict__ qoff, uint64_t nq, idx_t n, uint64_t max_hits,
                                                     fmgpu_hit* __restrict__ out, uint64_t cap, Counters* ctr, StackView stk, uint32_t qwords, uint32_t qnib,
                                                     LfView lfv, uint32_t tab_lds, const uint4* __restrict__ lut, uint32_t lutL, int use_key, int sharing,
                                                     const uint32_t* __restrict__ order) {
    extern __shared__ uint32_t s_query[];
    const QStage qst{s_query, qwords, qnib};
    __shared__ uint32_t s_hb[kWaveHitWords];
    wave_ring_init(s_hb);
    __shared__ idx_t s_C[257];
    if (lfv.fw) for (uint32_t i = threadIdx.x; i <= fw.sigma(); i += blockDim.x) s_C[i] = lfv.C[i];
    if (tab_lds) {                                                  // the step table sits behind the staged queries
        uint32_t* st = s_query + (size_t)qwords * 256u;
        for (uint32_t i = threadIdx.x; i < S * M; i += blockDim.x) st[i] = tab[i];
        tab = st;
    }
    __syncthreads();
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t sigma = fw.sigma();
    const uint32_t INS = 2u * sigma - 1u;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t nodes = 0, nh = 0;
    uint64_t q = 0, quota = 0;
    uint32_t si = 0, seq = 0;
    bool idle = false, have_query = false, fresh = false, need_search = true;
    const uint8_t* qs = qbuf;
    const uint32_t* T = tab;
    Cur cur{0, 0, 0};
    uint32_t e = 0, k = 0, sp = 0, resume = kNoResume, lastRank = 0, info = 0;     // info: LInfo | RInfo << 2 (0 M, 1 S, 2 I, 3 D)
    idx_t cached_lf = 0;                                            // a resumed one-row frame brings the LF value of its row along
    // path keys and work sharing at the end of the batch, as in k_scheme_edit.  Every error child of a search_ng21 node follows its match child
    // (2i - 1 deletion, 2i substitution, insertion last), so every component of the key is (2, 255 - depth, child index)
    bool is_task = false;
    uint32_t sbase = 0, mark = 0, ndel = 0;
    uint64_t pkey = 0;
    for (;;) {
        if (sharing) {
            if (is_task && need_search) { is_task = false; idle = true; }
            const bool offer = !idle && !need_search && sp > sbase && nodes - mark >= kShareNodes;
            const uint64_t idlem = __ballot(idle), offerm = __ballot(offer);
            if (idlem && offerm) {
                const WavePairs wp(idlem, offerm, lane);
                const bool give = wp.gives(offer);
                const bool take = wp.takes(idle);
                idx_t g0 = 0, g1 = 0, g2 = 0, g3 = 0; uint32_t ga = 0, gb = 0, gc = 0, gd = 0; uint64_t gk = 0;
                if (give) {
                    edit_frame_get(edit_frame(stk, gid, sbase), g0, g1, g2, g3, ga, gb, gc, gd);
                    gk = ekey_prefix(pkey, gc & 255u);
                    ++sbase; mark = nodes;
                }
                const int vl = wp.partner(take);                       // my partner: the lane whose frame this lane takes
                const uint64_t t0 = __shfl((uint64_t)g0, vl, 64), t1 = __shfl((uint64_t)g1, vl, 64), t2 = __shfl((uint64_t)g2, vl, 64), t3 = __shfl((uint64_t)g3, vl, 64);
                const uint32_t ta = __shfl(ga, vl, 64), tb = __shfl(gb, vl, 64), tc = __shfl(gc, vl, 64), td = __shfl(gd, vl, 64);
                const uint64_t tk = __shfl(gk, vl, 64), tq = __shfl(q, vl, 64), tqs = __shfl((uint64_t)qs, vl, 64);
                const uint32_t tsi = __shfl(si, vl, 64);
                if (take) {
                    q = tq; qs = reinterpret_cast<const uint8_t*>(tqs); si = tsi; T = tab + (size_t)si * M;
                    cur = Cur{(idx_t)t0, (idx_t)t1, (idx_t)t2}; cached_lf = (idx_t)t3; k = ta; resume = tb;
                    e = tc & 255u; info = (tc >> 8) & 15u; lastRank = (tc >> 16) & 255u; ndel = td;
                    pkey = tk;
                    const uint32_t vt = (threadIdx.x & ~63u) | (uint32_t)vl;
                    for (uint32_t w = 0; w < qwords; ++w) s_query[w * 256u + threadIdx.x] = s_query[w * 256u + vt];
                    idle = false; is_task = true; need_search = false; have_query = true; fresh = false;
                    quota = max_hits; seq = 0; sp = 0; sbase = 0; mark = nodes;
                }
            }
        }
        const bool want_q = !idle && need_search && !(have_query && si + 1 < S && quota != 0);   // search_reordered, :168-181: the next search unless the delegate said stop
        const uint64_t got = wave_hand_out(want_q, ctr, lane);
        if (want_q) {
            q = got;
            have_query = false;
            if (q >= nq) idle = true;
            else {
                if (order) q = order[q];
                const uint64_t qo = qoff[q];
                qs = qbuf + qo;
                if ((uint32_t)(qoff[q + 1] - qo) >= M && n != 0) { have_query = true; fresh = true; }   // (a shorter query is read out of bounds by the reference)
            }
        }
        const bool full = wave_ring_fill(s_hb) >= kWaveRingFlush; const uint64_t busy = __ballot(!idle);
        if (full || !busy) wave_flush_hits(s_hb, nh, lane, out, cap, ctr);
        if (!busy) break;
        if (idle) continue;
        if (need_search) {
            if (!have_query) continue;
            if (fresh) { fresh = false; si = 0; quota = max_hits; seq = 0; qstage_load(qst, qbuf, qoff[q], M, sigma); }
            else ++si;
            T = tab + (size_t)si * M;
            cur = Cur{0, 0, n};                                    // run(): :43-48
            e = 0; k = 0; sp = 0; resume = kNoResume; lastRank = 0; info = 0;
            sbase = 0; mark = nodes; ndel = 0; pkey = ((uint64_t)si << 48) | kEditKeyNone;
            need_search = false;
            if (lut && (T[0] >> 31)) {                              // the first lutL steps are exact, rightwards and adjacent: start from the prefix table
                const uint32_t p0 = T[0] & 0xffffu;
                uint32_t code = 0, mul = 1, c = 0; bool valid = true;
                for (uint32_t t = 0; t < lutL; ++t) {
                    c = qstage_get(qst, qs, p0 + t);
                    valid = valid && c >= 1 && c < sigma;
                    code += (c - 1) * mul; mul *= sigma - 1;
                }
                if (valid) {
                    const uint4 en = lut[code];
                    nodes += en.w;                                  // the extensions performed before the interval is empty (:146-148 per step)
                    if (en.z == 0) { need_search = true; continue; }
                    cur = Cur{en.x, en.y, en.z};
                    k = lutL; lastRank = c;
                }
            }
        }
        const uint32_t tw = T[k];
        const bool right = (tw >> 30) & 1u;
        const uint32_t Lk = (tw >> 16) & 127u, Uk = (tw >> 23) & 127u;
        const uint32_t c = qstage_get(qst, qs, tw & 0xffffu);
        const uint32_t d = right ? 1u : 0u;
        const uint32_t TI = (info >> (2u * d)) & 3u;
        const bool Deletion = TI == 0u || TI == 3u, Insertion = TI == 0u || TI == 2u;                       // :91-92
        const bool resuming = resume != kNoResume;
        bool mOK = Lk <= e && e <= Uk && (TI != 3u || c != lastRank);                                        // :105-107
        if (mOK && TI == 2u) mOK = c != qstage_get(qst, qs, T[k - 1u] & 0xffffu);                           // (TI = I only after a step)
        const bool xOK = Lk <= e + 1u && e + 1u <= Uk;                                                      // :108
        uint32_t kind = 4u, take = c, nxt = kNoResume, code = 0; // kind: 0 match, 1 substitution, 2 deletion, 3 insertion, 4 nothing left; code: the child's number
        idx_t lfa[MAXSIG], lfb[MAXSIG];
        const bool via_lf = lfv.fw != nullptr && cur.len == 1;      // one row: its only child comes from the LF table (one 4-byte load)
        idx_t lf1 = cached_lf;
        if (xOK || (mOK && !resuming)) {
            const Occ& occ = right ? rv : fw;
            const idx_t a = right ? cur.lbRev : cur.lb;
            SymSet<MAXSIG> alive;
            if (via_lf) {
                if (!resuming) lf1 = (right ? lfv.rv : lfv.fw)[a];
                alive.clear(); alive.insert(symbol_of_lf_lds(s_C, sigma, lf1));
            } else {
                occ.template all2<MAXSIG>(a, a + cur.len, lfa, lfb);
                alive = alive_set<MAXSIG>(lfa, lfb, sigma);
            }
            if (!resuming) ++nodes;                                 // one extend call per node (:111 or :146)
            if (!xOK) { if (alive.test(c)) kind = 0u; }
            else {
                SymSet<MAXSIG> subs = alive; subs.remove(0); subs.remove(c);
                SymSet<MAXSIG> dels = subs; if (!Deletion) dels.clear();
                const bool c_alive = alive.test(c);
                auto child_from = [&](uint32_t s0) -> uint32_t {    // number of the first existing child >= s0
                    if (s0 == 0u && mOK && c_alive) return 0u;
                    SymSet<MAXSIG> dd = dels, ss = subs;
                    dd.clear_below(s0 <= 1u ? 1u : (s0 + 2u) >> 1);  // 2i - 1 >= s0
                    ss.clear_below(s0 <= 2u ? 1u : (s0 + 1u) >> 1);  // 2i     >= s0
                    uint32_t best = kNoResume;
                    if (dd.any()) best = 2u * dd.first() - 1u;
                    if (ss.any()) { uint32_t v = 2u * ss.first(); if (v < best) best = v; }
                    if (best == kNoResume && Insertion && s0 <= INS) best = INS;
                    return best;
                };
                const uint32_t idx = child_from(resuming ? resume : 0u);
                if (idx != kNoResume) {
                    if (idx == 0u) kind = 0u;
                    else if (idx == INS) kind = 3u;
                    else { kind = (idx & 1u) ? 2u : 1u; take = (idx + 1u) >> 1; }
                    if (idx != INS) nxt = child_from(idx + 1u);
                    code = idx;
                }
            }
        }
        if (kind != 4u && nxt != kNoResume) {                       // keep the parent: its remaining children start at nxt
            edit_frame_put(edit_frame(stk, gid, sp), cur.lb, cur.lbRev, cur.len, lf1, k, nxt, e | (info << 8) | (lastRank << 16), ndel);
            ++sp;
        }
        resume = kNoResume;
        bool back = kind == 4u;
        if (kind != 4u) {
            if (kind != 3u) {
                if (via_lf) cur = right ? Cur{cur.lb, lf1, 1} : Cur{lf1, cur.lbRev, 1};
                else cur = kid_of<MAXSIG>(lfa, lfb, cur, take, right, sigma);
            }
            const uint32_t imask = ~(3u << (2u * d));
            if (kind == 0u) { info = info & imask; lastRank = c; }
            else {
                if (use_key) pkey = ekey_with(pkey, e, false, k + ndel, code);
                e += 1u;
                if (kind == 2u) ++ndel;
                info = (info & imask) | ((kind == 1u ? 1u : (kind == 2u ? 3u : 2u)) << (2u * d));
                if (kind != 3u) lastRank = take;
            }
            if (kind != 2u) ++k;
            if (k == M) {                                           // search_next, :73-78
                const uint32_t li = info & 3u, ri = (info >> 2) & 3u;
                if ((li == 0u || li == 2u) && (ri == 0u || ri == 2u)) {
                    Cur r = cur;
                    if ((uint64_t)r.len > quota) r.len = (idx_t)quota;       // search_n, :229-235
                    quota -= r.len;
                    if (use_key) { wave_keep_hit(s_hb, nh, out, cap, ctr, q, r, e | ((uint32_t)(pkey >> 32) << 8), (uint32_t)pkey); ++seq; }
                    else wave_keep_hit(s_hb, nh, out, cap, ctr, q, r, e, seq++);
                    if (quota == 0) { need_search = true; continue; }
                }
                back = true;
            }
        }
        if (back) {
            if (sp == sbase) { need_search = true; continue; }
            --sp;
            uint32_t fc_ = 0;
            edit_frame_get(edit_frame(stk, gid, sp), cur.lb, cur.lbRev, cur.len, cached_lf, k, resume, fc_, ndel);
            e = fc_ & 255u; info = (fc_ >> 8) & 15u; lastRank = (fc_ >> 16) & 255u;
            pkey = ekey_prefix(pkey, e);
        }
    }
    uint32_t tot = wave_sum(nodes);
    if ((threadIdx.x & 63u) == 0 && tot) atomicAdd(&ctr->nodes, (unsigned long long)tot);
}

// ---- search_ng26 Hamming, fast path ------------------------------------------------------------------------------
// For a batch of equal-length queries on a Format-A BiFMIndex the host expands the scheme once into a per-step table
// (query position, direction, error window [minE, maxE] of the step, "last character of its part"; SearchNg26.h:160-168:
// match allowed <=> minE <= e <= maxE, substitution allowed <=> minE <= e+1 <= maxE, where minE = l[part] on a part's
// last character and 0 before it, maxE = u[part]) — the same thing search_scheme::expand() does for the reference's
// per-character searches (search_scheme/expand.h:146-165).  The kernel then carries only (cursor, e, step) per lane.
// A single-row cursor (the common case after ~log4(n) characters) takes its one child from the explicit LF table with
// ONE 4-byte load; the symbol is recovered from C.  Multi-row cursors use the 64-byte blocks as before.
struct FastArgs {
    const idx_t* lf_fw; const idx_t* lf_rv;     // LF tables of bwt / bwtRev
    const idx_t* w3_fw; const idx_t* w3_rv;     // LF, LF^2, LF^3 per row (fmgpu_index_accelerate_search), or null
    const uint4* lut; uint32_t lutL, lut_ok;    // prefix table, its string length, bit s: search s may start from it
    const uint2* wj_fw; const uint2* wj_rv;     // per row {LF^16, the 16 symbols met} (2 bits each), or null
    const uint32_t* steps;                      // [S][m + 1]: pos:16 | right:1 | lastOfPart:1 | minE:5 | maxE:6 | run16:1 | run:2; entry m = final window;
                                                // followed by [S][m + 1] stretch words for run16 steps: tb:5 | hi1:6 | hi2:6 | lo:5 | simple:1 (see build_step_table)
    uint32_t S, m;
    idx_t C1[8];                                // C[1..] for sigma <= 8 (symbol of an LF value), unused otherwise
};

template <int SIGMA>
__device__ __forceinline__ uint32_t symbol_of_lf(const FastArgs& fa, const idx_t* C, uint32_t sigma, idx_t t) {
    uint32_t b = 0;
    if (SIGMA > 0 && SIGMA <= 8) {
#pragma unroll
        for (int k = 1; k < SIGMA; ++k) b += t >= fa.C1[k - 1] ? 1u : 0u;
    } else {
        for (uint32_t k = 1; k < sigma; ++k) b += t >= C[k] ? 1u : 0u;
    }
    return b;
}


#if !FMGPU_WIDE   // ======== 32-bit rows only: the table-driven k-mismatch kernels (LF / walk / prefix tables, their frames and transport words hold 32-bit rows)
// when a wave fetches and stages new queries (a wave-synchronous phase of ~15 loads and LDS stores per lane, paid by all 64 lanes): when the
// iterations its idle lanes have lost add up to kRefillWaste lane-iterations.  A refill costs the working lanes more than it looks: measured on
// 10 M reads, waste threshold 256 / 512 / 2048 / 4096 lane-iterations = uniform text with tables 8.1 / 6.65 / 6.64 / 6.64 ms, uniform plain index
// 77 / 71 / 61.6 / 64.6 ms, genome text with tables 134 / 131 / 129 / 129 ms — on a uniform text the lanes of a wave finish within a few
// iterations of each other and one refill serves them all; next to a heavy read the idle lanes are first served by its shared frames
constexpr uint32_t kRefillWaste = 2048;

// a counter that costs nothing where it is not wanted
template <class T, bool ON> struct Tally {
    T v = 0;
    __device__ __forceinline__ void operator+=(T x) { if (ON) v += x; }
    __device__ __forceinline__ void operator++() { if (ON) ++v; }
};

// PLAIN: the index holds no tables (no LF, walk or prefix table: the ~6 GB configuration) — every node reads blocks and the table paths are
// compiled out (fewer scalar registers spilled; genome text 168 -> 164 ms).
// What was measured on this kernel and did NOT pay (10 M x 101 bp, k = 2, genome-like text, plain index; 91 % of the lanes of a wave are busy
// in an iteration, an iteration takes ~8.6 us, VALU busy 41 %, scratch-free at 96 VGPRs = 5 waves per SIMD, which the LDS allows too):
//   * the substitution children of a node waiting on the stack as nodes of their own (no node is read twice): 183 -> 227 ms;
//   * asking for a waiting frame at the end of an iteration and taking it at the end of the next (the L2 round trip under the next block loads
//     instead of behind them): 164 -> 222 ms — four more live registers spill to scratch inside the loop;
//   * 6 / 7 / 8 waves per SIMD by launch bounds: 32 / 48 / 86 VGPRs spilled.
// The loop is bound by instruction issue and register pressure together with the random-line rate, not by lane utilisation.
template <int SIGMA, int MAXSIG, bool PLAIN>
__global__ __launch_bounds__(256, SIGMA == 5 ? 5 : 1) void k_scheme_fast(OccA<SIGMA> fw, OccA<SIGMA> rv, FastArgs fa, const uint8_t* __restrict__ qbuf,
                                                                          const uint64_t* __restrict__ qoff, uint64_t nq, idx_t n, uint64_t max_hits,
                                                                          fmgpu_hit* __restrict__ out, uint64_t cap, Counters* ctr, StackView stk,
                                                                          uint32_t qwords, uint32_t qnib, int dev_flags, const uint32_t* __restrict__ qmap, int sharing, int use_key, unsigned long long* next_ctr) {
    // One flat loop per lane over (query, search, node) with every slow path wave-synchronous.  A wave pays for every slow path any of its 64
    // lanes takes, so nothing with a dependent memory round trip is lane-private: queries are fetched and staged TOGETHER — as soon as
    // kRefillLanes lanes of the wave are out of work (one atomicAdd per refill, query words issued back to back) — and hits are kept in LDS
    // and written out together (one atomicAdd per flush).  The work per read is heavy-tailed on a repeat-rich text (a read from a young repeat
    // family or a satellite visits 10^4 - 10^5 nodes where the median read visits 200): in lock-step rounds of 64 reads nearly every round
    // holds such a read and the other 63 lanes wait for it (measured: 10 % lane utilisation on the genome-like text); refilling lanes as
    // they finish keeps them busy.  Every iteration performs ONE memory phase for whatever its lanes are doing:
    //   search start     : the prefix table entry of the first L symbols of the always-exact first part;
    //   multi-row cursor : the 64-byte blocks at both ends (extend-all);
    //   single-row cursor: ONE load of a walk entry / LF, LF^2, LF^3 / LF and up to 16 / 3 / 1 steps from it (or the row's block: plain index).
    extern __shared__ uint32_t s_dyn[];
    uint32_t* s_steps = s_dyn + (size_t)qwords * 256u;
    const uint32_t S = fa.S, m = fa.m, stride = fa.m + 1;
    const uint32_t* s_stretch = s_steps + S * stride;               // stretch words of the run16 steps
    const uint32_t* s_stretch3 = s_steps + 2u * S * stride;         // ... and of the <= 3 steps of a `run`
    uint32_t* s_hb = s_steps + 3u * S * stride;                     // kWaveHitWords: the waves' hit rings (wave_keep_hit / wave_flush_hits)
    wave_ring_init(s_hb);
    const QStage qst{s_dyn, qwords, qnib};
    for (uint32_t i = threadIdx.x; i < 3u * S * stride; i += blockDim.x) s_steps[i] = fa.steps[i];
    __syncthreads();

    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t sigma = fw.sigma(), R = sigma - 1;
    // 16-byte frames, lane-interleaved: frame d of lane g at [d * nlanes + g] — one store, one load (two 8-byte planes: genome text, plain index 196 -> 183 ms).
    // (Tried and dropped: the substitution children of a node waiting on the stack as nodes of their own, so that nobody re-reads the node's blocks
    // for the next sibling — the same number of frames, but deriving and storing up to three children in the node's iteration costs every lane of
    // the wave their instructions: genome text, plain index 183 -> 227 ms, with tables 130 -> 141 ms.)
    ulonglong2* const frames = reinterpret_cast<ulonglong2*>(stk.p0);
    uint32_t nodes = 0;
    Tally<uint64_t, !PLAIN> tbytes; Tally<uint32_t, !PLAIN> tacc;   // table bytes consumed / table accesses issued (fmgpu_stats; the plain index is priced per node)
    const uint32_t refill_waste = ((uint32_t)dev_flags >> 8) & 0xffffu ? (((uint32_t)dev_flags >> 8) & 0xffffu) : kRefillWaste;   // (dev knob: bits 8..23)
    uint32_t waste = 0;                                             // lane-iterations the wave's idle lanes have lost since its last refill (wave-uniform)
    uint32_t nh = 0, count_only = 0; [[maybe_unused]] uint32_t nodes0 = 0;
    bool is_task = false;                                           // the lane works on a subtree it took over from another lane
    uint32_t sbase = 0, mark = 0;                                   // frames below sbase were handed out; nodes at the lane's last hand-out
    const uint32_t share_nodes = ((uint32_t)dev_flags >> 25) & 31u ? 1u << (((uint32_t)dev_flags >> 25) & 31u) : kShareNodes;   // (dev knob: bits 25..29)
    const uint32_t share_heavy = (dev_flags & 129) == 128 ? 0u : kShareHeavy;   // (dev knob: bit 7 alone = every read may offer)
    uint64_t pkey = 0;                                              // path key of the node the lane stands on
    bool have = false, exhausted = n == 0, need_start = false, query_over = false;
    uint64_t q = 0, quota = 0;
    const uint8_t* qs = qbuf;
    const uint32_t* tab = s_steps;
    uint32_t seq = 0, si = 0;
    Cur cur{0, 0, 0};                                               // state of the search in progress
    uint32_t e = 0, j = 0, sp = 0, resume = kNoResume;
    bool in_tail = false;
    for (;;) {
        // ---- wave-synchronous part: every lane passes here in every iteration
        if (sharing) {
            const uint64_t idlem = __ballot(!have), offerm = __ballot(have && sp > sbase && nodes - mark >= share_nodes && nodes - nodes0 >= share_heavy);
            if (idlem && offerm) {
                // the i-th idle lane takes the bottom frame of the i-th offering lane
                const WavePairs wp(idlem, offerm, lane);
                const bool give = wp.gives(have && sp > sbase && nodes - mark >= share_nodes && nodes - nodes0 >= share_heavy);
                const bool take = wp.takes(!have);
                uint64_t w0 = 0, w1 = 0, w2 = 0;
                if (give) {
                    const ulonglong2 fr = frames[(uint64_t)sbase * stk.nlanes + gid];
                    w0 = fr.x; w1 = fr.y; w2 = key_prefix(pkey, (uint32_t)(w1 >> 48) & 0xffu);
                    ++sbase; mark = nodes; tbytes += 16u; ++tacc;
                }
                const int vl = wp.partner(take);                       // my partner: the lane whose frame this lane takes
                const uint64_t tw0 = __shfl(w0, vl, 64), tw1 = __shfl(w1, vl, 64), tw2 = __shfl(w2, vl, 64), tq_ = __shfl(q, vl, 64);
                const uint64_t tqs = __shfl((uint64_t)qs, vl, 64);
                const uint32_t tsi = __shfl(si, vl, 64);
                if (take) {
                    q = tq_; si = tsi; qs = reinterpret_cast<const uint8_t*>(tqs);
                    cur.lb = (idx_t)tw0; cur.lbRev = (idx_t)(tw0 >> 32); cur.len = (idx_t)tw1;
                    j = (uint32_t)(tw1 >> 32) & 0xffffu; e = (uint32_t)(tw1 >> 48) & 0xffu; resume = (uint32_t)(tw1 >> 56) & 0xffu;
                    pkey = tw2;
                    const uint32_t vt = (threadIdx.x & ~63u) | (uint32_t)vl;
                    for (uint32_t w = 0; w < qwords; ++w) s_dyn[w * 256u + threadIdx.x] = s_dyn[w * 256u + vt];     // the partner's staged read
                    have = true; is_task = true; need_start = false; query_over = false; quota = max_hits; seq = 0;
                    tab = s_steps + si * stride; sp = 0; sbase = 0; in_tail = false; mark = nodes; nodes0 = nodes;
                }
            }
        }
        const uint64_t needm = __ballot(!have && !exhausted), busym = __ballot(have);
        waste += (uint32_t)__popcll(needm);
        if (needm && (waste >= refill_waste || !busym)) {
            waste = 0;
            const bool want = !have && !exhausted;
            bool fresh = false; uint64_t qo = 0;
            const uint64_t got = wave_hand_out_at(want, next_ctr, lane);    // nq = queries of this launch; qmap (if any) names them within the batch
            if (want) {
                if (got >= nq) exhausted = true;
                else {
                    q = qmap ? (uint64_t)qmap[got] : got; qo = qoff[q]; qs = qbuf + qo; fresh = true;
                    have = true; is_task = false; si = 0; need_start = true; quota = max_hits; seq = 0; query_over = false; nodes0 = nodes; mark = nodes;
                }
            }
            qstage_load_sync(qst, qbuf, qo, m, sigma, fresh, m);
            __builtin_amdgcn_s_waitcnt(0x0f70);                    // vmcnt(0) in this rare path: or the compiler, unsure whether a staging load is still pending, waits for vmcnt(0) in every iteration
        }
        {
            const bool full = wave_ring_fill(s_hb) >= kWaveRingFlush; const uint64_t busy = __ballot(have);
            if (full || !busy) wave_flush_hits(s_hb, nh, lane, out, cap, ctr);
            if (!busy) { if (__ballot(!exhausted) == 0ull) break; continue; }
        }
        if (!have) continue;
        {
            bool lut_start = false; uint32_t lut_code = 0;
            if (need_start) {                                       // search_impl (SearchNg26.h:385-390) -> run(): :62-79
                need_start = false;
                tab = s_steps + si * stride;
                cur = Cur{0, 0, n}; e = 0; j = 0; sp = 0; sbase = 0; resume = kNoResume; in_tail = false;
                pkey = (uint64_t)si << 48;
                if (!PLAIN && fa.lut && ((fa.lut_ok >> si) & 1u) && n > 1) {  // the exact first part starts from the prefix table
                    uint32_t code = 0, mul = 1; bool valid = true;
                    for (uint32_t t = 0; t < fa.lutL; ++t) {
                        uint32_t c = qstage_get(qst, qs, tab[t] & 0xffffu);
                        valid = valid && c >= 1 && c < sigma;
                        code += (c - 1) * mul; mul *= R;
                    }
                    lut_start = valid; lut_code = code;
                }
            }
            {
                const uint32_t ent = tab[j];
                const bool right = (ent >> 16) & 1u;
                const bool multi = !lut_start && cur.len > 1;
                const idx_t a = right ? cur.lbRev : cur.lb;
                bool back = false, search_over = false;
                // ---- memory phase: the lanes of a wave sit in different kinds of nodes; every lane issues its loads here, before any lane
                // consumes one, so that an iteration costs one round trip and not one per kind of node present in the wave
                // (ONE unconditional 16-byte load per lane — the first quarter of a multi-row lane's block, or a one-row lane's walk / LF^1..3 /
                // LF entry, all dword-aligned, the tables carry 16 bytes of slack — and the rest of the two blocks right behind it)
                constexpr bool kSplit = SIGMA > 0 && SIGMA <= 5;
                const uint2* wj = PLAIN ? nullptr : (right ? fa.wj_rv : fa.wj_fw);
                const idx_t* w3 = PLAIN ? nullptr : (right ? fa.w3_rv : fa.w3_fw);
                const bool use_wj = !PLAIN && !lut_start && !multi && wj && ((ent >> 29) & 1u);
                const uint8_t* blk = (right ? rv : fw).v.blk;
                // the plain index (no LF tables: the ~6 GB configuration): a one-row node reads its 64-byte block and takes the row's symbol and LF from it
                const bool from_block = PLAIN ? !multi : (kSplit && !lut_start && !multi && fa.lf_fw == nullptr);
                const uint8_t* p0 = lut_start ? reinterpret_cast<const uint8_t*>(fa.lut + lut_code)
                                  : ((multi && kSplit) || from_block) ? blk + (size_t)(a >> 6) * 64u
                                  : use_wj ? reinterpret_cast<const uint8_t*>(wj + a)
                                  : w3 ? reinterpret_cast<const uint8_t*>(w3 + 3u * (size_t)a)
                                  : reinterpret_cast<const uint8_t*>((right ? fa.lf_rv : fa.lf_fw) + a);
                const uint4 r0 = *reinterpret_cast<const uint4*>(p0);
                uint4 r1, r2, r3, s0, s1, s2, s3;
                if (multi && kSplit) {
                    const uint4* pa = reinterpret_cast<const uint4*>(p0);
                    const uint4* pb = reinterpret_cast<const uint4*>(blk + (size_t)((a + cur.len) >> 6) * 64u);
                    r1 = pa[1]; r2 = pa[2]; r3 = pa[3]; s0 = pb[0]; s1 = pb[1]; s2 = pb[2]; s3 = pb[3];
                    const bool same = (a >> 6) == ((a + cur.len) >> 6);
                    tbytes += same ? 64u : 128u; tacc += same ? 1u : 2u;
                } else if (multi) { tbytes += 24u * sigma; tacc += 2u; }
                else { tbytes += lut_start ? 16u : from_block ? 64u : use_wj ? 8u : w3 ? 12u : 4u; ++tacc; }
                // r0 is first READ here, behind the loads of the rest of the blocks: without this the compiler copies two of its words out of the way right
                // after the load — behind a vmcnt(0), i.e. a multi-row node paid two dependent round trips per iteration (rocprofv3 / ISA, round 3)
                uint4 r0v = r0;
                asm volatile("" : "+v"(r0v.x), "+v"(r0v.y), "+v"(r0v.z), "+v"(r0v.w));
#define r0 r0v
                const uint2 we = make_uint2(r0.x, r0.y);
                idx_t t0 = r0.x, t1 = r0.y, t2 = r0.z;
                uint32_t row_sym = 0;                               // plain index: the symbol of a one-row node's row, read off its block
                if constexpr (kSplit) {
                    if (from_block) {
                        const uint4* pa = reinterpret_cast<const uint4*>(p0);
                        r1 = pa[1]; r2 = pa[2]; r3 = pa[3];
                        const uint32_t da[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
                        const uint32_t bit = a & 63u;
                        t0 = 0;
#pragma unroll
                        for (uint32_t cc = 1; cc < (uint32_t)SIGMA; ++cc) {      // (a delimiter row — no symbol >= 1 claims it — ends the walk; entry 0 may hold presence bits)
                            const uint64_t bits = (uint64_t)da[3 * cc + 1] | ((uint64_t)da[3 * cc + 2] << 32);
                            if ((bits >> bit) & 1ull) { t0 = da[3 * cc] + popc64(bits & lowmask(bit)); row_sym = cc; }
                        }
                    }
                }
                if (lut_start) {
                    cur = Cur{r0.x, r0.y, r0.z};
                    nodes += r0.w;                                  // the extensions the reference performs before the interval is empty (:225-250)
                    j = fa.lutL; in_tail = true;
                    if (r0.z == 0) search_over = true;
                } else if (multi) {
                    // ---- extend-all node (search_next_dir, :143-224) or exact-tail step over several rows
                    const uint32_t pos = ent & 0xffffu, minE = (ent >> 18) & 0x1fu, maxE = (ent >> 23) & 0x3fu;
                    const bool lastp = (ent >> 17) & 1u;
                    idx_t lfa[MAXSIG], lfb[MAXSIG];
                    const OccA<SIGMA>& occ = right ? rv : fw;
                    if constexpr (kSplit) {
                        const uint32_t da[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
                        const uint32_t db[16] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w, s2.x, s2.y, s2.z, s2.w, s3.x, s3.y, s3.z, s3.w};
                        occ.all2_of(da, db, a, a + cur.len, lfa, lfb);
                    } else occ.template all2<MAXSIG>(a, a + cur.len, lfa, lfb);
                    const SymSet<MAXSIG> alive = alive_set<MAXSIG>(lfa, lfb, sigma);
                    const uint32_t c = qstage_get(qst, qs, pos);
                    const bool c_alive = alive.test(c);
                    const bool resuming = resume != kNoResume;
                    const bool mOK = minE <= e && e <= maxE;
                    const bool sOK = minE <= e + 1 && e + 1 <= maxE;
                    const bool xOK = e + 1 <= maxE;
                    SymSet<MAXSIG> subs = alive;                   // substitution children: FirstSymb = 1 (fmindex/BiFMIndex.h:26), != query symbol
                    subs.remove(0); subs.remove(c);
                    if (!sOK) subs.clear();
                    if (resuming) subs.clear_below(resume);
                    const bool take_match = !resuming && mOK && c_alive;   // match child first (:171-181)
                    const bool take_sub = !take_match && subs.any();
                    uint32_t take = c;
                    if (take_sub) { take = subs.first(); subs.remove(take); }
                    nodes += (!resuming && (in_tail || xOK || mOK)) ? 1u : 0u;   // node accounting as the reference works (see k_scheme)
                    if ((take_match || take_sub) && subs.any()) {  // (re-)push the parent: remaining siblings start at subs.first()
                        frames[(uint64_t)sp * stk.nlanes + gid] = make_ulonglong2((uint64_t)cur.lb | ((uint64_t)cur.lbRev << 32),
                            (uint64_t)cur.len | ((uint64_t)(j & 0xffffu) << 32) | ((uint64_t)(e & 0xffu) << 48) | ((uint64_t)(subs.first() & 0xffu) << 56));
                        ++sp; tbytes += 16u; ++tacc;
                    }
                    resume = kNoResume;
                    if (take_match || take_sub) {
                        cur = kid_of<MAXSIG>(lfa, lfb, cur, take, right, sigma);
                        if (take_sub) { pkey = key_with(pkey, e, m, j, take); e += 1; }
                        in_tail = !lastp && (in_tail || (take_match && !xOK));
                        ++j;
                    } else back = true;
                } else {
                    // ---- single row (search_next_dir_single, :251-365): the only child is the BWT symbol of the row.  A stretch of n steps in one
                    // direction — 16 from a walk entry, or up to three from the LF / LF^2 / LF^3 values — is described by the steps at which the
                    // row's symbols differ from the query's (mm, bit 2k = step k), its stretch word (windows) and the row it leads to, and then
                    // walked run by run by ONE piece of code for both kinds.
                    bool walked = false, stretch = false, del_after = false;
                    uint32_t n = 0, mm = 0, sword = 0, sym3 = 0;   // sym3: the symbols of a <= 3-step stretch, 8 bits each (a 16-step stretch keeps them in we.y)
                    idx_t target = 0;
                    if (from_block) {
                        // plain index: ONE step from the row's own block — none of the stretch machinery below (a wave pays for the instructions
                        // of every path one of its lanes takes; with no tables no lane takes them)
                        walked = true;
                        const uint32_t minE = (ent >> 18) & 0x1fu, maxE = (ent >> 23) & 0x3fu;
                        const bool lastp = (ent >> 17) & 1u;
                        const uint32_t c = qstage_get(qst, qs, ent & 0xffffu);
                        const bool mOK = minE <= e && e <= maxE;
                        const bool sOK = minE <= e + 1 && e + 1 <= maxE;
                        const bool xOK = e + 1 <= maxE;
                        const bool is_match = row_sym >= 1u && row_sym == c && mOK;
                        nodes += 1u + ((!in_tail && is_match && !xOK) ? 1u : 0u);
                        bool dead = false;
                        if (row_sym < 1u) dead = true;              // :295-297: a delimiter row ends the walk
                        else if (row_sym == c) { if (!mOK) dead = true; }
                        else if (sOK) { pkey = key_with(pkey, e, m, j, row_sym); e += 1; }
                        else dead = true;
                        if (dead) back = true;
                        else { in_tail = !lastp && (in_tail || (is_match && !xOK)); if (right) cur.lbRev = t0; else cur.lb = t0; ++j; }
                    } else if (use_wj && we.x != 0xffffffffu) {
                        walked = true;
                        bool qvalid = false;
                        const uint32_t qc = query_code16(qst, ent & 0xffffu, right, qvalid);
                        const uint32_t sw = s_stretch[si * stride + j];
                        if (qvalid && ((sw >> 22) & 1u)) {
                            const uint32_t diff = qc ^ we.y;
                            stretch = true; n = 16u; mm = (diff | (diff >> 1)) & 0x55555555u; sword = sw; target = we.x;
                        } else {                                     // an odd query symbol or two part ends inside: step by step
                            bool dead = false;
                            for (uint32_t kk = 0; kk < 16u && !dead; ++kk) {
                                const uint32_t en = tab[j + kk];
                                const uint32_t minE = (en >> 18) & 0x1fu, maxE = (en >> 23) & 0x3fu;
                                const bool lastp = (en >> 17) & 1u;
                                const uint32_t b = ((we.y >> (2u * kk)) & 3u) + 1u;
                                const uint32_t c = qstage_get(qst, qs, en & 0xffffu);
                                const bool mOK = minE <= e && e <= maxE;
                                const bool sOK = minE <= e + 1 && e + 1 <= maxE;
                                const bool xOK = e + 1 <= maxE;
                                const bool is_match = b == c && mOK;
                                nodes += 1u + ((!in_tail && is_match && !xOK) ? 1u : 0u);
                                if (b == c) { if (!mOK) dead = true; }
                                else if (sOK) { pkey = key_with(pkey, e, m, j + kk, b); e += 1; }
                                else dead = true;
                                if (!dead) in_tail = !lastp && (in_tail || (is_match && !xOK));
                            }
                            if (dead) back = true;
                            else { if (right) cur.lbRev = we.x; else cur.lb = we.x; j += 16u; }
                        }
                    }
                    if (!walked) {
                        const uint32_t run = w3 ? (ent >> 30) : 1u;    // consecutive steps in this direction (<= 3), never past the query end
                        if (use_wj) {                                   // (a walk entry that crosses a delimiter: the steps one load later)
                            if (w3) { const idx_t* p = w3 + 3u * (size_t)a; t0 = p[0]; t1 = p[1]; t2 = p[2]; tbytes += 12u; }
                            else { t0 = (right ? fa.lf_rv : fa.lf_fw)[a]; tbytes += 4u; }
                            ++tacc;
                        }
                        const uint32_t sw3 = s_stretch3[si * stride + j];
                        if ((sw3 >> 22) & 1u) {
                            // the row's next symbols against the query's; a delimiter row ends the branch at its step (:295-297), after the steps before it
                            uint32_t nn = run;
#pragma unroll
                            for (uint32_t kk = 0; kk < 3; ++kk) {
                                if (kk < nn) {
                                    const idx_t tk = kk == 0 ? t0 : (kk == 1 ? t1 : t2);
                                    const uint32_t bsym = symbol_of_lf<SIGMA>(fa, fw.v.C, sigma, tk);
                                    const uint32_t c = qstage_get(qst, qs, (kk == 0 ? ent : tab[j + kk]) & 0xffffu);
                                    if (bsym < 1u) { nn = kk; del_after = true; }
                                    else { if (bsym != c) mm |= 1u << (2u * kk); target = tk; sym3 |= bsym << (8u * kk); }
                                }
                            }
                            stretch = true; n = nn; sword = sw3;
                        } else {
                            idx_t tk = t0, last = t0;
                            uint32_t k = 0;
                            bool dead = false;
#pragma unroll
                            for (uint32_t kk = 0; kk < 3; ++kk) {
                                if (kk < run && !dead) {
                                    const uint32_t en = kk == 0 ? ent : tab[j + kk];
                                    const uint32_t pos = en & 0xffffu, minE = (en >> 18) & 0x1fu, maxE = (en >> 23) & 0x3fu;
                                    const bool lastp = (en >> 17) & 1u;
                                    tk = kk == 0 ? t0 : (kk == 1 ? t1 : t2);
                                    const uint32_t b = symbol_of_lf<SIGMA>(fa, fw.v.C, sigma, tk);
                                    const uint32_t c = qstage_get(qst, qs, pos);
                                    const bool mOK = minE <= e && e <= maxE;
                                    const bool sOK = minE <= e + 1 && e + 1 <= maxE;
                                    const bool xOK = e + 1 <= maxE;
                                    const bool is_match = b >= 1 && b == c && mOK;
                                    nodes += 1u + ((!in_tail && is_match && !xOK) ? 1u : 0u);
                                    if (b < 1) dead = true;                 // :295-297: a delimiter row ends the walk
                                    else if (b == c) { if (!mOK) dead = true; }
                                    else if (sOK) { pkey = key_with(pkey, e, m, j + kk, b); e += 1; }
                                    else dead = true;
                                    if (!dead) { in_tail = !lastp && (in_tail || (is_match && !xOK)); last = tk; ++k; }
                                }
                            }
                            if (dead) back = true;
                            else { if (right) cur.lbRev = last; else cur.lb = last; j += k; }   // one row: the other side's prefix count is 0
                        }
                    }
                    if (stretch) {
                        // Steps up to the part end inside the stretch (offset tb; n: none) stretch the window [.., hi1], later ones [.., hi2]; the lower
                        // bound lo binds at step tb only.  Extensions are counted as the reference performs them: one per step, one more where an
                        // exact tail starts (a match with no error left and not in a tail yet, :310-314), and the step at which a branch ends is
                        // its last.  A wave spends as many rounds here as its lane with the most differing symbols, not n.
                        const uint32_t tb = sword & 31u, hi1 = (sword >> 5) & 63u, hi2 = (sword >> 11) & 63u, lo = (sword >> 17) & 31u;
                        uint32_t x = 0;
                        bool dead = false;
                        while (!dead && x < n) {
                            const uint32_t rest = mm >> (2u * x);
                            uint32_t p = rest ? x + (((uint32_t)__ffs((int)rest) - 1u) >> 1) : n;      // next differing step
                            p = p < n ? p : n;
                            uint32_t xb = x;
                            if (x < p && x <= tb) {                                              // matching steps up to the part end
                                if (e > hi1 || (x == tb && e < lo)) { nodes += 1u; dead = true; }
                                else {
                                    const uint32_t bonus = (!in_tail && e == hi1) ? 1u : 0u;
                                    in_tail = in_tail || e == hi1;
                                    if (tb < p) {
                                        if (tb > x && e < lo) { nodes += (tb - x) + bonus + 1u; dead = true; }
                                        else { nodes += (tb + 1u - x) + bonus; in_tail = false; xb = tb + 1u; }
                                    } else { nodes += (p - x) + bonus; xb = p; }
                                }
                            }
                            if (!dead && xb < p) {                                               // matching steps behind the part end
                                if (e > hi2) { nodes += 1u; dead = true; }
                                else { nodes += (p - xb) + ((!in_tail && e == hi2) ? 1u : 0u); in_tail = in_tail || e == hi2; }
                            }
                            if (!dead && p < n) {                                                // the differing step: a substitution or the end
                                nodes += 1u;
                                const uint32_t maxp = p <= tb ? hi1 : hi2, minp = p == tb ? lo : 0u;
                                if (minp <= e + 1u && e + 1u <= maxp) {
                                    const uint32_t bs = n == 16u ? ((we.y >> (2u * p)) & 3u) + 1u : (sym3 >> (8u * p)) & 255u;
                                    pkey = key_with(pkey, e, m, j + p, bs);
                                    e += 1u; in_tail = p != tb && in_tail;
                                } else dead = true;
                            }
                            x = p + 1u;
                        }
                        if (!dead && del_after) { nodes += 1u; dead = true; }
                        if (dead) back = true;
                        else if (n) { if (right) cur.lbRev = target; else cur.lb = target; j += n; }
                    }
                }
#undef r0
                if (!lut_start && !back && j == m) {                // search_next at part == P (:101-108)
                    const uint32_t fin = tab[m];
                    if (((fin >> 18) & 0x1fu) <= e && e <= ((fin >> 23) & 0x3fu)) {
                        Cur r = cur;
                        if ((uint64_t)r.len > quota) r.len = (idx_t)quota;
                        quota -= r.len;
                        if (dev_flags & 1) ++count_only;
                        else {
                            if (use_key) wave_keep_hit(s_hb, nh, out, cap, ctr, q, r, e | ((uint32_t)(pkey >> 32) << 8), (uint32_t)pkey);
                            else wave_keep_hit(s_hb, nh, out, cap, ctr, q, r, e, seq);
                            tbytes += 40u; ++tacc;
                        }
                        ++seq;
                        if (quota == 0) { search_over = true; query_over = true; }   // delegate returned true: no further searches (:372-383)
                    }
                    back = !search_over;
                }
                if (back) {
                    if (sp == sbase) search_over = true;
                    else {
                        --sp;
                        const ulonglong2 fr = frames[(uint64_t)sp * stk.nlanes + gid];
                        const uint64_t w0 = fr.x, w1 = fr.y;
                        tbytes += 16u; ++tacc;
                        cur.lb = (idx_t)w0; cur.lbRev = (idx_t)(w0 >> 32); cur.len = (idx_t)w1;
                        j = (uint32_t)(w1 >> 32) & 0xffffu; e = (uint32_t)(w1 >> 48) & 0xffu; resume = (uint32_t)(w1 >> 56) & 0xffu;
                        pkey = key_prefix(pkey, e);
                        in_tail = false;
                    }
                }
                if (search_over) {                                  // the next search of the scheme, or the lane is out of work
                    ++si;
                    if (is_task) { have = false; is_task = false; }  // a task is one subtree of one search: its owner goes on with the other searches
                    else if (si == S || query_over) {
                        have = false;
#ifdef FMGPU_DEV
                        if ((dev_flags & 129) == 129 && (q + 1) * 8 <= cap * sizeof(fmgpu_hit)) reinterpret_cast<uint64_t*>(out)[q] = nodes - nodes0;   // dev build: nodes per query instead of records (count-only mode)
#endif
                    } else need_start = true;
                }
            }
        }
    }
    uint32_t tot = wave_sum(nodes);
    const unsigned long long tb = wave_sum64(tbytes.v); const uint32_t ta = wave_sum(tacc.v), co = wave_sum(count_only);
    if ((threadIdx.x & 63u) == 0 && (tot || co)) {
        atomicAdd(&ctr->nodes, (unsigned long long)tot);
        atomicAdd(&ctr->table_bytes, tb); atomicAdd(&ctr->table_accesses, (unsigned long long)ta);
        if (co) atomicAdd(&ctr->hits, (unsigned long long)co);
    }
}

// ---- search_ng26 Edit = true, table-driven ------------------------------------------------------------------------------------------
// k_scheme_edit's node logic in k_scheme_fast's frame: equal-length queries, the scheme expanded into the per-step table (a step = one query
// symbol consumed: deletions stay on their step), ONE flat loop per lane over (query, search, node) with the slow paths wave-synchronous
// (queries fetched and staged together, hits kept in LDS and written out by the wave), and the lanes of a wave sharing the work of large reads.
// Round 1 ran rounds of 64 reads per wave with the searches in step: on a repeat-rich text nearly every round holds a read that visits
// 10^5 nodes while the median read visits a few hundred, and the other 63 lanes waited for it (measured: 2 M x 101 bp, k = 2:
// 1178 ms on the genome-like text against 41 ms on the uniform one).
//
// Edit path key.  The callback order of the reference is the depth-first order of SearchNg26.h:143-365: a node with several rows tries its
// match child first, then for every symbol s its deletion and its substitution child (2s-1, 2s), the insertion child last (:171-218); a
// node with one row tries the insertion FIRST, then the match or substitution child, then the deletion (:286-362).  For two hits of one
// read and search that is the lexicographic order of their paths written as one letter per tree depth — match, or the error edge taken —
// with the alphabet  { insertion at a one-row node } < match < { the other error edges in child order }.  Only the <= 3 error edges of a
// path are kept, 16 bits each: an edge that precedes the match child as (0, depth, 0) — the smaller depth diverges first and wins —, "no
// further error" as 0x4000, an edge that follows the match child as (2, 255 - depth, child index): a deeper one is met earlier on the way
// back up.  depth = query symbols consumed + deletions made.  Layout: search:4 | 3 x 16 bits; it travels like the Hamming key (the low
// 32 bits in fmgpu_hit::seq, the rest in the upper 24 bits of fmgpu_hit::errors) until fmgpu_hits_sort turns it into the callback index.
template <int SIGMA, int MAXSIG, bool PLAIN = false>      // PLAIN: sigma = 5 and no LF table — known when the kernel is compiled, so that the table paths (and the scalar registers their pointers hold) are not in it
__global__ __launch_bounds__(256) void k_scheme_fast_edit(OccA<SIGMA> fw, OccA<SIGMA> rv, FastArgs fa, const uint8_t* __restrict__ qbuf,
                                                          const uint64_t* __restrict__ qoff, uint64_t nq, idx_t n, uint64_t max_hits,
                                                          fmgpu_hit* __restrict__ out, uint64_t cap, Counters* ctr, StackView stk,
                                                          uint32_t qwords, uint32_t qnib, int dev_flags, const uint32_t* __restrict__ qmap, int sharing, int use_key, WorkBoard* board, unsigned long long* next_ctr) {
    extern __shared__ uint32_t s_dyn[];
    uint32_t* s_steps = s_dyn + (size_t)qwords * 256u;
    const uint32_t S = fa.S, m = fa.m, stride = fa.m + 1;
    uint32_t* s_hb = s_steps + 2u * S * stride;                     // kWaveHitWords
    // the TOP frame of every lane's stack lives in LDS (two 16-byte halves per lane) and reaches HBM only when another frame is pushed on top of it: a node with k
    // viable children is popped and pushed back k - 1 times, and with write-through frames (round 3) these were 45 % of the kernel's memory requests.  A pop reads the
    // slot and at once asks for the frame below it by LDS-DMA (global_load_lds: no register holds it), an iteration before it can be needed: the stack is off the
    // dependent chain.  Invariants: the slot is valid iff sp > sbase; every frame below the top is in HBM; `tos_dirty`: the slot's frame is not (or is newer than) its HBM copy.
    typedef uint32_t __attribute__((ext_vector_type(4))) e_u32x4;
    typedef __attribute__((address_space(3))) e_u32x4 e_lds_u32x4;
    e_lds_u32x4* const s_tos0 = (e_lds_u32x4*)(s_dyn + (((size_t)qwords * 256u + 2u * S * stride + kWaveHitWords + 3u) & ~(size_t)3));     // (16-byte aligned)
    e_lds_u32x4* const s_tos1 = s_tos0 + 256;
    e_lds_u32x4* const tos0 = s_tos0 + threadIdx.x; e_lds_u32x4* const tos1 = s_tos1 + threadIdx.x;
    wave_ring_init(s_hb);
    const QStage qst{s_dyn, qwords, qnib};
    for (uint32_t i = threadIdx.x; i < 2u * S * stride; i += blockDim.x) s_steps[i] = fa.steps[i];
    __syncthreads();

    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t sigma = fw.sigma(), R = sigma - 1;
    const uint32_t INS = 2u * sigma - 1u;
    if (board) board_enter(board, lane);                            // (work sharing between the waves of the launch: fmgpu_search_shared.h)
    uint32_t pass = 0;
    const uint32_t board_heavy = board ? board->heavy : 0u, board_period = board ? board->period - 1u : 0u;      // (the period is a power of two: its mask)
#ifdef FMGPU_DEV
    unsigned long long bt_wait = 0, bt_give = 0, bt_look = 0, bn_wait = 0, bn_give = 0, bn_look = 0; const unsigned long long bt_start = __builtin_amdgcn_s_memtime();
#define BSTAMP() __builtin_amdgcn_s_memtime()
#else
#define BSTAMP() 0ull
#endif
    uint4* const frames = reinterpret_cast<uint4*>(stk.p0) + 2u * (gid * ((uint64_t)stk.depth + 1u));   // this lane's frames, 32 bytes each
    uint32_t nodes = 0, nh = 0, count_only = 0;
    const uint32_t refill_waste = ((uint32_t)dev_flags >> 8) & 0xffffu ? (((uint32_t)dev_flags >> 8) & 0xffffu) : kRefillWaste;   // (dev knob: bits 8..23)
    const uint32_t share_nodes = ((uint32_t)dev_flags >> 25) & 31u ? 1u << (((uint32_t)dev_flags >> 25) & 31u) : kShareNodes;   // (dev knob: bits 25..29)
    const uint32_t share_heavy = (dev_flags & 129) == 128 ? 0u : kShareHeavy;   // (dev knob: bit 7 alone = every read may offer)
    uint32_t waste = 0;                                             // lane-iterations lost by idle lanes since the wave's last refill (wave-uniform)
    bool is_task = false;                                           // the lane works on a subtree it took over from another lane
    uint32_t sbase = 0, mark = 0;                                   // frames below sbase were handed out; nodes at the lane's last hand-out
    uint64_t pkey = 0;                                              // path key of the node the lane stands on
    bool have = false, exhausted = n == 0, need_start = false, query_over = false;
    uint64_t q = 0, quota = 0;
    const uint8_t* qs = qbuf;
    const uint32_t* tab = s_steps;
    uint32_t seq = 0, si = 0;
    Cur cur{0, 0, 0};
    uint32_t e = 0, j = 0, sp = 0, resume = kNoResume, side = 0, info = 0, ndel = 0, nodes0 = 0;
    bool in_tail = false, lf_known = false;
    idx_t lf_val = 0, cached_lf = 0, cached_lf2 = 0xffffffffu;
    uint32_t report_slot = kNoResume;
#ifdef FMGPU_DEV
    uint32_t dev_slot_bad = 0;
    uint32_t dev_multi = 0, dev_revisit = 0, dev_busy = 0, dev_iter = 0, dev_noload = 0;   // dev build: lane-iterations on nodes of several rows / of those, re-visits for the next child / all / wave iterations / one-row iterations that load nothing
#endif
    bool tos_dirty = false, tos_pending = false;                    // tos_pending: a refill of the slot by LDS-DMA may still be in flight (settle() before the slot is touched)
    // the refill was issued at the end of an earlier iteration; by the places that touch the slot every load of the current iteration has been consumed, so this
    // wait is for that one transfer (and for frame spills / record flushes of the previous iteration)
    auto settle = [&]() {
        if (__ballot(tos_pending)) { __builtin_amdgcn_s_waitcnt(0x0f70); asm volatile("" ::: "memory"); }
        tos_pending = false;
    };
    // a lane takes over a subtree (a frame handed over inside the wave, or through the board): the frame's node becomes the lane's, as a task of its own
    auto adopt = [&](const uint4& t0, const uint4& t1, uint64_t key) {
        const bool one_row = (t0.w >> 24) & 1u;
        cur.lb = t0.x; cur.lbRev = t0.y; cur.len = one_row ? 1u : t0.z; cached_lf2 = one_row ? t0.z : 0xffffffffu;
        j = t0.w & 0xffffu; e = (t0.w >> 16) & 0xffu; info = (t0.w >> 25) & 15u;
        resume = t1.x; side = t1.y; cached_lf = t1.z; ndel = t1.w;
        pkey = key;
        have = true; is_task = true; need_start = false; query_over = false; quota = max_hits; seq = 0;
        tab = s_steps + si * stride; sp = 0; sbase = 0; in_tail = false; lf_known = false; report_slot = kNoResume; mark = nodes; nodes0 = nodes;
        tos_dirty = false;
    };
    for (;;) {
        // ---- wave-synchronous part: every lane passes here in every iteration
        if (sharing) {
            // (a frame whose running child still owes it the row's LF^2 — report_slot — stays with its owner for that one iteration)
            const bool offer = have && sp > sbase && nodes - mark >= share_nodes && nodes - nodes0 >= share_heavy && report_slot != sbase;
            const uint64_t idlem = __ballot(!have), offerm = __ballot(offer);
            if (idlem && offerm) {
                // the i-th idle lane takes the bottom frame of the i-th offering lane
                const WavePairs wp(idlem, offerm, lane);
                const bool give = wp.gives(offer);
                const bool take = wp.takes(!have);
                uint4 g0 = make_uint4(0, 0, 0, 0), g1 = g0;
                uint64_t gk = 0;
                if (__ballot(give && sbase + 1u == sp)) settle();   // (a bottom frame that is also the top frame is in the slot)
                if (give) {
                    if (sbase + 1u == sp) { const e_u32x4 u0 = *tos0, u1 = *tos1; g0 = make_uint4(u0.x, u0.y, u0.z, u0.w); g1 = make_uint4(u1.x, u1.y, u1.z, u1.w); tos_dirty = false; }
                    else { g0 = frames[2u * sbase]; g1 = frames[2u * sbase + 1u]; }
                    gk = ekey_prefix(pkey, (g0.w >> 16) & 0xffu);
                    ++sbase; mark = nodes;
                }
                const int vl = wp.partner(take);                       // my partner: the lane whose frame this lane takes
                uint4 t0, t1;
                t0.x = __shfl(g0.x, vl, 64); t0.y = __shfl(g0.y, vl, 64); t0.z = __shfl(g0.z, vl, 64); t0.w = __shfl(g0.w, vl, 64);
                t1.x = __shfl(g1.x, vl, 64); t1.y = __shfl(g1.y, vl, 64); t1.z = __shfl(g1.z, vl, 64); t1.w = __shfl(g1.w, vl, 64);
                const uint64_t tk = __shfl(gk, vl, 64), tq_ = __shfl(q, vl, 64), tqs = __shfl((uint64_t)qs, vl, 64);
                const uint32_t tsi = __shfl(si, vl, 64);
                if (take) {
                    q = tq_; si = tsi; qs = reinterpret_cast<const uint8_t*>(tqs);
                    adopt(t0, t1, tk);
                    const uint32_t vt = (threadIdx.x & ~63u) | (uint32_t)vl;
                    for (uint32_t w = 0; w < qwords; ++w) s_dyn[w * 256u + threadIdx.x] = s_dyn[w * 256u + vt];     // the partner's staged read
                }
            }
        }
        // ---- ... and with the other waves of the launch: every kBoardPeriod-th pass a wave with subtrees to give looks whether a wave waits for work, and hands it the bottom
        // frames of ALL its offering lanes (the same frames, keys and thresholds as above; the taker stages the reads itself)
        if (board && (++pass & board_period) == 0u) {
            const bool cand = have && sp > sbase && nodes - mark >= share_nodes && nodes - nodes0 >= board_heavy && report_slot != sbase;
            const uint64_t cm = __ballot(cand);
            uint32_t bslot = 0, bidx = 0;
            [[maybe_unused]] const unsigned long long t0_ = BSTAMP();
            const bool got_ = cm && board_reserve(board, lane, &bslot, &bidx);
#ifdef FMGPU_DEV
            if (cm) { bt_look += BSTAMP() - t0_; ++bn_look; }
#endif
            if (got_) {
                if (__ballot(cand && sbase + 1u == sp)) settle();
                // a batch holds 64 subtrees: few offering lanes give SEVERAL frames each, from the bottom of their stacks up — the frames below the top are complete in HBM; the top
                // frame (in its slot, possibly waiting for a child's LF) goes only when it is the lane's one frame, as inside the wave
                const uint32_t per = min(kBoardFramesPerLane, 64u / (uint32_t)__popcll(cm));
                const uint32_t mine = !cand ? 0u : (sbase + 1u == sp ? 1u : min(per, sp - 1u - sbase));
                const uint32_t r0 = wave_excl_scan(mine, lane), total = __shfl(r0 + mine, 63, 64);
                for (uint32_t i = 0; i < mine; ++i) {
                    uint4 g0, g1;
                    if (sbase + 1u == sp) { const e_u32x4 u0 = *tos0, u1 = *tos1; g0 = make_uint4(u0.x, u0.y, u0.z, u0.w); g1 = make_uint4(u1.x, u1.y, u1.z, u1.w); tos_dirty = false; }
                    else { g0 = frames[2u * (sbase + i)]; g1 = frames[2u * (sbase + i) + 1u]; }
                    const uint64_t gk = ekey_prefix(pkey, (g0.w >> 16) & 0xffu);
                    const uint32_t w[kBoardWords] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, (uint32_t)gk, (uint32_t)(gk >> 32), (uint32_t)q, (uint32_t)(q >> 32), si, 0u};
#pragma unroll
                    for (uint32_t d_ = 0; d_ < kBoardWords; ++d_) board_put(board, bslot, d_, r0 + i, w[d_]);
                }
                if (mine) { sbase += mine; mark = nodes; }
                board_publish(board, lane, bslot, bidx, total);
#ifdef FMGPU_DEV
                bt_give += BSTAMP() - t0_; ++bn_give;
#endif
            }
        }
        const uint64_t needm = __ballot(!have && !exhausted), busym = __ballot(have);
        waste += (uint32_t)__popcll(needm);
        if (needm && (waste >= refill_waste || !busym)) {
            waste = 0;
            const bool want = !have && !exhausted;
            bool fresh = false; uint64_t qo = 0;
            const uint64_t got = wave_hand_out_at(want, next_ctr, lane);    // nq = queries of this launch; qmap (if any) names them within the batch
            if (want) {
                if (got >= nq) exhausted = true;
                else {
                    q = qmap ? (uint64_t)qmap[got] : got; qo = qoff[q]; qs = qbuf + qo; fresh = true;
                    have = true; is_task = false; si = 0; need_start = true; quota = max_hits; seq = 0; query_over = false; mark = nodes; nodes0 = nodes;
                }
            }
            qstage_load_sync(qst, qbuf, qo, m, sigma, fresh, m);
            __builtin_amdgcn_s_waitcnt(0x0f70);                    // vmcnt(0) in this rare path: or the compiler, unsure whether a staging load is still pending, waits for vmcnt(0) in every iteration
        }
        {
            const bool full = wave_ring_fill(s_hb) >= kWaveRingFlush; const uint64_t busy = __ballot(have);
            if (full || !busy) wave_flush_hits(s_hb, nh, lane, out, cap, ctr);
            if (!busy) {
                if (__ballot(!exhausted) != 0ull) continue;
                if (!board) break;
                // the wave is out of work and the batch is handed out: it waits for subtrees of the reads other waves still walk, or for the end of the launch
                uint32_t bslot = 0, bidx = 0;
                [[maybe_unused]] const unsigned long long t0_ = BSTAMP();
                const uint32_t k = board_wait(board, lane, &bslot, &bidx);
#ifdef FMGPU_DEV
                bt_wait += BSTAMP() - t0_; ++bn_wait;
#endif
                if (k == 0u) break;
                const bool fresh = lane < k; uint64_t qo = 0;
                if (fresh) {
                    uint32_t w[kBoardWords];
#pragma unroll
                    for (uint32_t d_ = 0; d_ < kBoardWords; ++d_) w[d_] = board_word(board, bslot, d_, lane);
                    q = (uint64_t)w[10] | ((uint64_t)w[11] << 32); si = w[12];
                    qo = qoff[q]; qs = qbuf + qo;
                    adopt(make_uint4(w[0], w[1], w[2], w[3]), make_uint4(w[4], w[5], w[6], w[7]), (uint64_t)w[8] | ((uint64_t)w[9] << 32));
                }
                board_release(board, lane, bslot, bidx);
                qstage_load_sync(qst, qbuf, qo, m, sigma, fresh, m);
                __builtin_amdgcn_s_waitcnt(0x0f70);
                continue;
            }
        }
        if (!have) continue;
        bool lut_start = false; uint32_t lut_code = 0;
        if (need_start) {                                           // search_impl (SearchNg26.h:385-390) -> run(): :62-79
            need_start = false;
            tab = s_steps + si * stride;
            cur = Cur{0, 0, n}; e = 0; j = 0; sp = 0; sbase = 0; resume = kNoResume; side = 0; info = 0; ndel = 0;
            in_tail = false; lf_known = false; cached_lf = 0; cached_lf2 = 0xffffffffu; report_slot = kNoResume; tos_dirty = false;
            pkey = ((uint64_t)si << 48) | kEditKeyNone;
            if (fa.lut && ((fa.lut_ok >> si) & 1u) && n > 1) {      // the always-exact first part starts from the prefix table
                uint32_t code = 0, mul = 1; bool valid = true;
                for (uint32_t t = 0; t < fa.lutL; ++t) {
                    uint32_t c = qstage_get(qst, qs, tab[t] & 0xffffu);
                    valid = valid && c >= 1 && c < sigma;
                    code += (c - 1) * mul; mul *= R;
                }
                lut_start = valid; lut_code = code;
            }
        }
        const uint32_t ent = tab[j];
        const bool right = (ent >> 16) & 1u;
        const uint32_t minE = (ent >> 18) & 0x1fu, maxE = (ent >> 23) & 0x3fu;
        const bool lastp = (ent >> 17) & 1u;
        const uint32_t c = qstage_get(qst, qs, ent & 0xffffu);
        const bool multi = !lut_start && cur.len > 1, resuming = resume != kNoResume;
        const idx_t a = right ? cur.lbRev : cur.lb;
        const uint32_t d = right ? 1u : 0u;
        const uint32_t T = (info >> (2u * d)) & 3u;
        const uint32_t lastR = (side >> (8u * d)) & 255u, lastQ = (side >> (16u + 8u * d)) & 255u;
        const bool Deletion = T != 1u && T != 2u, Insertion = T != 1u && T != 3u;                      // :146-147
        const bool mOK = minE <= e && e <= maxE && (T != 2u || c != lastQ) && (T != 3u || c != lastR);    // :160-163
        const bool iOK = minE <= e + 1 && e + 1 <= maxE;
        const bool xOK = e + 1 <= maxE;
        // ---- memory phase: ONE 16-byte load per lane, whatever kind of node it stands on (the prefix table entry of a search start, the first
        // quarter of a multi-row node's block, a one-row node's LF entry — dword-aligned, the tables carry 16 bytes of slack), and the rest
        // of the two blocks right behind it: an iteration costs the wave one round trip, not one per kind of node.  A one-row node that is
        // resumed, or that an insertion or a cached deletion led to, knows its LF value and loads nothing (about half of the iterations)
        constexpr bool kSplit = SIGMA > 0 && SIGMA <= 5;
        const uint8_t* blk = (right ? rv : fw).v.blk;
        // (the plain index — no LF table, sigma = 5: a one-row node reads its row's symbol and LF off the row's block, like a multi-row node reads its counts)
        constexpr bool plain = PLAIN && kSplit;
        const uint8_t* p0 = lut_start ? reinterpret_cast<const uint8_t*>(fa.lut + lut_code)
                          : ((multi && kSplit) || plain) ? blk + (size_t)(a >> 6) * 64u
                          : reinterpret_cast<const uint8_t*>((right ? fa.lf_rv : fa.lf_fw) + a);
        uint4 r0 = make_uint4(0, 0, 0, 0);
        const bool row_load = !lut_start && !multi && !resuming && !lf_known;
#ifdef FMGPU_DEV
        ++dev_busy; dev_multi += multi ? 1u : 0u; dev_revisit += (multi && resuming) ? 1u : 0u; dev_noload += (!lut_start && !multi && !row_load) ? 1u : 0u;
        if (lane == (uint32_t)__ffsll((unsigned long long)__ballot(true)) - 1u) ++dev_iter;
#endif
        if (lut_start || multi || row_load) r0 = *reinterpret_cast<const uint4*>(p0);
        uint4 q1 = make_uint4(0, 0, 0, 0), q2 = q1, q3 = q1;
        if (plain && row_load) { const uint4* pa = reinterpret_cast<const uint4*>(p0); q1 = pa[1]; q2 = pa[2]; q3 = pa[3]; }
        bool back = false, search_over = false;
        if (lut_start) {
            cur = Cur{r0.x, r0.y, r0.z};
            nodes += r0.w;
            j = fa.lutL; in_tail = true;
            const uint32_t lastc = qstage_get(qst, qs, tab[fa.lutL - 1u] & 0xffffu);
            side = (lastc << 8) | (lastc << 24);
            if (r0.z == 0) search_over = true;
        } else {
            idx_t lfa[MAXSIG], lfb[MAXSIG];
            SymSet<MAXSIG> alive;
            idx_t lf1 = cached_lf;
            if (multi) {
                const OccA<SIGMA>& occ = right ? rv : fw;
                if constexpr (kSplit) {
                    const uint4* pa = reinterpret_cast<const uint4*>(p0);
                    const uint4* pb = reinterpret_cast<const uint4*>(blk + (size_t)((a + cur.len) >> 6) * 64u);
                    const uint4 r1 = pa[1], r2 = pa[2], r3 = pa[3], s0 = pb[0], s1 = pb[1], s2 = pb[2], s3 = pb[3];
                    const uint32_t da[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
                    const uint32_t db[16] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w, s2.x, s2.y, s2.z, s2.w, s3.x, s3.y, s3.z, s3.w};
                    occ.all2_of(da, db, a, a + cur.len, lfa, lfb);
                } else occ.template all2<MAXSIG>(a, a + cur.len, lfa, lfb);
                alive = alive_set<MAXSIG>(lfa, lfb, sigma);
            } else {
                if (!resuming) {
                    lf1 = lf_known ? lf_val : (idx_t)r0.x;
                    if (plain && row_load) {                        // the symbol that claims the row and its LF; a delimiter row: 0 (below C[1]: symbol 0, a dead end)
                        const uint32_t dd[16] = {r0.x, r0.y, r0.z, r0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
                        const uint32_t bit = (uint32_t)a & 63u;
                        lf1 = 0;
#pragma unroll
                        for (uint32_t s_ = 1; s_ < 5u; ++s_) {
                            const uint64_t bits = (uint64_t)dd[3 * s_ + 1] | ((uint64_t)dd[3 * s_ + 2] << 32);
                            if ((bits >> bit) & 1ull) lf1 = dd[3 * s_] + popc64(bits & lowmask(bit));
                        }
                    }
                    // (the parent that waits for this row's LF was pushed in the previous iteration: it is the top frame, in its slot)
                    if (__ballot(report_slot != kNoResume)) settle();
                    if (report_slot != kNoResume) { e_u32x4 u0 = *tos0; u0.z = (uint32_t)lf1; *tos0 = u0; }
                }
                alive.clear(); alive.insert(symbol_of_lf<SIGMA>(fa, fw.v.C, sigma, lf1));
            }
            lf_known = false; report_slot = kNoResume;
            // ---- child selection (edit_next_child: one definition for this kernel and k_scheme_edit)
            const uint32_t start = resuming ? resume : 0u;
            const EditStep es = edit_next_child<MAXSIG>(alive, c, in_tail, multi, resuming, start, mOK, iOK, xOK, Deletion, Insertion, INS);
            const uint32_t kind = es.kind, take = es.take, nxt = es.nxt, code = es.code;   // code: the child's index among the error children of its node (path key)
            const bool start_tail = es.start_tail;
            nodes += edit_nodes_visited(in_tail, resuming, multi, xOK, mOK, start_tail);
            const bool keep = kind != 4u && nxt != kNoResume;
            if (__ballot(keep || kind == 4u)) settle();             // (the slot is touched below: by a push, or by the pop of a lane that goes back)
            if (keep) {                                             // keep the parent: its remaining children start at nxt
                if (sp > sbase && tos_dirty) {                      // the frame the slot holds is buried now: to HBM
                    const e_u32x4 u0 = *tos0, u1 = *tos1;
                    uint4* f = frames + 2u * (sp - 1u);
                    f[0] = make_uint4(u0.x, u0.y, u0.z, u0.w); f[1] = make_uint4(u1.x, u1.y, u1.z, u1.w);
                }
                const e_u32x4 n0 = {(uint32_t)cur.lb, (uint32_t)cur.lbRev, multi ? (uint32_t)cur.len : 0xffffffffu, (j & 0xffffu) | ((e & 0xffu) << 16) | ((multi ? 0u : 1u) << 24) | (info << 25)};
                const e_u32x4 n1 = {nxt, side, (uint32_t)lf1, ndel};
                *tos0 = n0; *tos1 = n1; tos_dirty = true;
                if (!multi && nxt == 2u && (kind == 0u || kind == 1u)) report_slot = sp;
                ++sp;
            }
            resume = kNoResume;
            back = kind == 4u;
            if (kind != 4u) {
                if (kind != 3u) {
                    if (multi) cur = kid_of<MAXSIG>(lfa, lfb, cur, take, right, sigma);
                    else {
                        cur = right ? Cur{cur.lb, lf1, 1} : Cur{lf1, cur.lbRev, 1};
                        if (kind == 2u && resuming && cached_lf2 != 0xffffffffu) { lf_known = true; lf_val = cached_lf2; }
                    }
                } else if (!multi) { lf_known = true; lf_val = lf1; }
                if (kind != 0u) {                                   // an error edge: one more component of the path key
                    if (use_key) pkey = ekey_with(pkey, e, !multi && kind == 3u, j + ndel, code);
                    e += 1;
                    if (kind == 2u) ++ndel;
                }
                edit_note_edge(kind, take, c, d, side, info);
                if (kind != 2u) {                                   // one query symbol consumed: the next step of the table
                    in_tail = !lastp && (in_tail || start_tail);
                    ++j;
                    if (j < m && (((tab[j] >> 16) & 1u) != (right ? 1u : 0u))) { lf_known = false; report_slot = kNoResume; }   // the other index from here on
                }
            }
            // (Tried and dropped, round 4: a one-row node that is popped for its next child taking that child in the SAME iteration — its LF comes with the frame, nothing is loaded; such
            // re-visits are 16 % of the lane-iterations on the genome text.  Parity held; genome text 112.9 -> 119.2 ms per 2 M reads (with tables 103.5 -> 114.1), uniform text 49.9 -> 49.0:
            // the child selection a second time in every iteration of the wave costs more than the iterations it saves)
            if (!back && j == m) {                                  // search_next at part == P (:101-108)
                const uint32_t fin = tab[m];
                const uint32_t li = info & 3u, ri = (info >> 2) & 3u;
                if ((li == 0u || li == 2u) && (ri == 0u || ri == 2u) && ((fin >> 18) & 0x1fu) <= e && e <= ((fin >> 23) & 0x3fu)) {
                    Cur r = cur;
                    if ((uint64_t)r.len > quota) r.len = (idx_t)quota;
                    quota -= r.len;
                    if (dev_flags & 1) ++count_only;
                    else if (use_key) wave_keep_hit(s_hb, nh, out, cap, ctr, q, r, e | ((uint32_t)(pkey >> 32) << 8), (uint32_t)pkey);
                    else wave_keep_hit(s_hb, nh, out, cap, ctr, q, r, e, seq);
                    ++seq;
                    if (quota == 0) { search_over = true; query_over = true; }
                }
                back = !search_over;
            }
            // A lane that pops now and popped an iteration ago may not have loaded anything in between (a one-row node that knows its LF; a child that completes the read): nothing it
            // consumed orders the refill of its slot before this read of it.  Tests passed without this wait for half a round; with the board keeping the memory system busy to the end of a
            // launch, one run of the full-size edit test counted 82 nodes too many (a stale slot = a frame walked twice).
            if (__ballot(back && sp > sbase && tos_pending)) settle();
            if (back) {
                if (sp == sbase) search_over = true;
                else {
                    --sp;
                    asm volatile("" ::: "memory");                  // (the slot is read here, not ahead of the branch)
                    const e_u32x4 a0 = *tos0, a1 = *tos1;           // the top frame, from its slot ...
#ifdef FMGPU_DEV
                    if (!tos_dirty) {                               // dev build: a clean slot holds what HBM holds
                        const uint4 h0 = frames[2u * sp], h1 = frames[2u * sp + 1u];
                        if (h0.x != a0.x || h0.y != a0.y || h0.z != a0.z || h0.w != a0.w || h1.x != a1.x || h1.y != a1.y || h1.z != a1.z || h1.w != a1.w) ++dev_slot_bad;
                    }
#endif
                    if (sp > sbase) {                               // ... and the one below it, requested an iteration before it can be needed, straight into the slot
                        const uint4* below = frames + 2u * (sp - 1u);
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)below, (__attribute__((address_space(3))) void*)(s_tos0 + (threadIdx.x & ~63u)), 16, 0, 0);
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(below + 1), (__attribute__((address_space(3))) void*)(s_tos1 + (threadIdx.x & ~63u)), 16, 0, 0);
                        tos_pending = true;
                    }
                    tos_dirty = false;
                    const bool one_row = (a0.w >> 24) & 1u;
                    cur.lb = a0.x; cur.lbRev = a0.y; cur.len = one_row ? 1u : a0.z; cached_lf2 = one_row ? a0.z : 0xffffffffu;
                    j = a0.w & 0xffffu; e = (a0.w >> 16) & 0xffu; info = (a0.w >> 25) & 15u;
                    resume = a1.x; side = a1.y; cached_lf = a1.z; ndel = a1.w;
                    pkey = ekey_prefix(pkey, e);
                    in_tail = false; lf_known = false; report_slot = kNoResume;
                }
            }
        }
        if (search_over) {                                          // the next search of the scheme, or the lane is out of work
            ++si;
            if (is_task) { have = false; is_task = false; }          // a task is one subtree of one search: its owner goes on with the other searches
            else if (si == S || query_over) have = false;
            else need_start = true;
        }
    }
    const uint32_t tot = wave_sum(nodes), co = wave_sum(count_only);
    if ((threadIdx.x & 63u) == 0 && (tot || co)) {
        atomicAdd(&ctr->nodes, (unsigned long long)tot);
        if (co) atomicAdd(&ctr->hits, (unsigned long long)co);
    }
#ifdef FMGPU_DEV
    { const uint32_t bad = wave_sum(dev_slot_bad); if ((threadIdx.x & 63u) == 0 && bad) atomicAdd(reinterpret_cast<unsigned long long*>(ctr) + 20, (unsigned long long)bad); }
    if (board && lane == 0) {
        atomicAdd(&board->dev[0], bt_wait); atomicAdd(&board->dev[1], bt_give); atomicAdd(&board->dev[2], bt_look); atomicAdd(&board->dev[3], bn_wait); atomicAdd(&board->dev[4], bn_give);
        atomicAdd(&board->dev[5], bn_look); atomicAdd(&board->dev[6], BSTAMP() - bt_start);
    }
    {   // (tools/edit_mix_probe.py: table_accesses = multi | wave iterations << 40, table_bytes = busy | load-free << 40, table_steps = re-visits)
        const uint32_t a1 = wave_sum(dev_multi), a2 = wave_sum(dev_iter), a3 = wave_sum(dev_busy), a4 = wave_sum(dev_revisit), a5 = wave_sum(dev_noload);
        if ((threadIdx.x & 63u) == 0) {
            atomicAdd(&ctr->table_accesses, (unsigned long long)a1 | ((unsigned long long)a2 << 40)); atomicAdd(&ctr->table_bytes, (unsigned long long)a3 | ((unsigned long long)a5 << 40));
            atomicAdd(reinterpret_cast<unsigned long long*>(ctr) + 21, (unsigned long long)a4);
        }
    }
#endif
}

#endif  // !FMGPU_WIDE (the table-driven kernels)

// ---- search_ng26 Hamming on the PLAIN index (sigma = 5, no table of any kind): the lean kernel -----------------------------------------
// k_scheme_fast<5, 5, PLAIN> carries the frame of the table-driven kernel; on the plain index it ran at 0.47 of the HBM roofline,
// bound by latency at 5 waves per SIMD: per iteration a wave paid (a) the pop of a frame from its stack in HBM and then, dependent on it, (b) the
// block loads of the popped node, (c) one returning atomicAdd on ONE device-wide word per ~6 hit records (8.8 M of them per 10 M reads — the word
// saturates near 88 M/s).  This kernel does the same walk (search/SearchNg26.h:143-365, Edit = false) with
//   * the top frame of a lane's stack cached in LDS: a pop is served from there, and the frame below is requested at once by an LDS-DMA load
//     (global_load_lds: no register holds it, nothing waits for it) a whole iteration before it can be needed — the stack never sits on the
//     dependent chain (pushes still write through, so the frames in HBM are complete);
//   * hit records appended to a ring per WAVE in LDS (slot by LDS atomic) and written out with one reservation per >= kRingFlush records;
//   * reads staged with 2 bits per symbol (a read with a byte outside 1..4 is read from global memory instead), 12 instead of 16 dwords per
//     block end (the entry of the delimiter is derived: a query that holds a 0 takes a slow path), no table paths, no quota:
//   ~76 registers and 27 KB of LDS per block at 101 bp.
// Path keys, sharing of the bottom frame between the lanes of a wave, heavy reads first: as in k_scheme_fast.
// Both row widths.  With 64-bit rows (n < 2^38, reads of <= 255 symbols) a 16-byte frame holds lb:38 | lbRev:38 | len:38 | step:8 | errors:3 | next sibling:3,
// the counts of a block are relative to its super-block of 2^30 rows (the super table, a few hundred bytes, is staged in LDS), a hit record takes 7 words.
constexpr uint32_t kRingCap = 160;       // hit records a wave keeps in LDS ...
constexpr uint32_t kRingFlush = 32;      // ... written out as soon as there are this many: 128 more always fit; a pass (kLeanSteps node steps) that reports more sends the excess out one by one
constexpr uint32_t kRingWords = kWide ? 7u : 6u;   // qidx, lb, lbRev, len, errors | key high, key low (, the rows' bits 32..39)
constexpr uint32_t kLeanRefillWaste = 2048;        // (kRefillWaste of k_scheme_fast)
constexpr uint32_t kLeanShareHeavy = 64;           // nodes a lane spends on a read before it offers subtrees of it
constexpr uint32_t kLeanSuperRows = kWide ? 64u : 1u;   // super-block rows per direction staged in LDS (64 x 2^30 rows: more than HBM holds)
constexpr uint32_t kLeanNoResume = 7u;
constexpr int kLeanWaves = 4;            // resident blocks per CU the register allocation allows (the grid asks for 3: the loop is bound by the L1 access rate, not by latency)
constexpr uint64_t kLeanBoardReads = ~0ull;   // batches of at most this many reads run the BOARD instantiation of the lean kernel
constexpr int kLeanSteps = 4;            // node steps per pass through the wave-synchronous part (genome text, kernel ms at 101 / 151 bp: 1 step 119 / 213, 2: 114 / 195, 4: 112 / 183; round 4, 8 steps for 101 bp on Format D: 99.3 against ~96 ms)

struct LeanArgs {
    const uint8_t* fw; const uint8_t* rv;    // Format A blocks of bwt / bwtRev (64 bytes per 64 rows)
    const uint32_t* steps;                   // [S][m + 1]: pos:16 | right:1 | lastOfPart:1 | minE:5 | maxE:6 (build_step_table)
    uint32_t S, m;
    idx_t ksum;                              // C[1] + ... + C[4] mod 2^width: LF(i, 0) = i + ksum - sum of the other symbols' LF (the kernel never reads entry 0)
    const uint64_t* sup_fw; const uint64_t* sup_rv;   // 64-bit rows: [row >> 30][5] counts at the start of each super-block
    uint32_t super_rows;
    const uint4* dfw; const uint4* drv;               // Format D blocks (DENSE instantiation), two uint4 per 64 rows
    const uint32_t* ex_fw; const uint32_t* ex_rv; uint32_t nex_fw, nex_rv;   // ... and their delimiter rows, ascending
    const uint4* lut; uint32_t lutL, lut_ok;          // prefix table (fmgpu_index_accelerate_search(h, L <= 16, 0): the bidirectional interval of every string of lutL symbols, or null); bit s: search s may start from it
};
// a frame as two 64-bit words (what travels between lanes when a subtree is handed over) <-> its fields
__device__ __forceinline__ void lean_pack(idx_t lb, idx_t lbRev, idx_t len, uint32_t j, uint32_t e, uint32_t next, uint64_t& w0, uint64_t& w1) {
    if constexpr (kWide) {
        w0 = (uint64_t)lb | ((uint64_t)lbRev << 38);
        w1 = ((uint64_t)lbRev >> 26) | ((uint64_t)len << 12) | ((uint64_t)(j & 0xffu) << 50) | ((uint64_t)(e & 7u) << 58) | ((uint64_t)(next & 7u) << 61);
    } else {
        w0 = (uint64_t)lb | ((uint64_t)lbRev << 32);
        w1 = (uint64_t)len | ((uint64_t)(j & 0xffffu) << 32) | ((uint64_t)(e & 0xffu) << 48) | ((uint64_t)(next & 0xffu) << 56);
    }
}
__device__ __forceinline__ void lean_unpack(uint64_t w0, uint64_t w1, idx_t& lb, idx_t& lbRev, idx_t& len, uint32_t& j, uint32_t& e, uint32_t& next) {
    if constexpr (kWide) {
        const uint64_t m38 = (1ull << 38) - 1ull;
        lb = (idx_t)(w0 & m38); lbRev = (idx_t)(((w0 >> 38) | (w1 << 26)) & m38); len = (idx_t)((w1 >> 12) & m38);
        j = (uint32_t)(w1 >> 50) & 0xffu; e = (uint32_t)(w1 >> 58) & 7u; next = (uint32_t)(w1 >> 61) & 7u;
    } else {
        lb = (idx_t)w0; lbRev = (idx_t)(w0 >> 32); len = (idx_t)w1;
        j = (uint32_t)(w1 >> 32) & 0xffffu; e = (uint32_t)(w1 >> 48) & 0xffu; next = (uint32_t)(w1 >> 56) & 0xffu;
    }
}
__device__ __forceinline__ uint32_t lean_frame_errors(uint64_t w1) { return kWide ? (uint32_t)(w1 >> 58) & 7u : (uint32_t)(w1 >> 48) & 0xffu; }
typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
struct __attribute__((packed, aligned(4))) Quad4 { uint32_t x, y, z, w; };   // 16 bytes at dword alignment (the entries of symbols 1..4 start 12 bytes into a block)

// eight query bytes in 1..4 -> 16 bits (2 per symbol, symbol - 1); ok = false if a byte is outside 1..4
__device__ __forceinline__ uint32_t pack2_8(uint64_t x, bool& ok) {
    const uint64_t k1 = 0x0101010101010101ull;
    const uint64_t v = x - k1;                                    // (a zero byte borrows: caught below)
    if ((((x - k1) & ~x & (0x80ull * k1)) != 0ull) || ((v & (0xfcull * k1)) != 0ull)) ok = false;
    uint64_t t = v & (0x03ull * k1);
    t = (t | (t >> 6)) & 0x000f000f000f000full;
    t = (t | (t >> 12)) & 0x000000ff000000ffull;
    t = (t | (t >> 24)) & 0xffffull;
    return (uint32_t)t;
}
// wave-synchronous staging of one equal-length read per lane, 16 symbols per LDS word; returns false for a read that holds a byte outside 1..4
__device__ __forceinline__ bool stage2_sync(uint32_t* lds, const uint8_t* __restrict__ qbuf, uint64_t off, uint32_t m, bool active) {
    const uint8_t* p = qbuf + off;
    const uint32_t mis = (uint32_t)((uint64_t)p & 7ull);
    const uint64_t* base = reinterpret_cast<const uint64_t*>(p - mis);      // (pointer arithmetic, not integer: the loads stay global_load, not flat_load)
    const uint32_t nw = active ? ((mis + m + 7u) >> 3) : 0u;
    bool ok = true;
    uint64_t carry = 0;
    uint32_t half = 0;
    for (uint32_t k0 = 0; k0 <= ((m + 14u) >> 3); k0 += 8) {
        uint64_t r[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) r[k] = (k0 + k < nw) ? base[k0 + k] : 0ull;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            const uint32_t widx = k0 + k;
            if (widx == 0) { carry = r[k]; continue; }
            uint64_t x = mis ? ((carry >> (8u * mis)) | (r[k] << (64u - 8u * mis))) : carry;
            carry = r[k];
            const uint32_t g = widx - 1u, p0 = g * 8u;             // group g = query positions 8g .. 8g + 7
            if (active && p0 < m) {
                const uint32_t nv = m - p0;                        // bytes of the group that belong to the read
                if (nv < 8u) x = (x & ((1ull << (8u * nv)) - 1ull)) | (0x0101010101010101ull << (8u * nv));
                const uint32_t code = pack2_8(x, ok);
                if (g & 1u) lds[(g >> 1) * 256u + threadIdx.x] = half | (code << 16);
                else { half = code; if (nv <= 8u) lds[(g >> 1) * 256u + threadIdx.x] = half; }     // (the read's last group)
            }
        }
    }
    return ok;
}

typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ void ring_flush(lds_u32* s_cnt_w, const uint32_t* ring, uint32_t lane, fmgpu_hit* __restrict__ out, uint64_t cap, Counters* ctr) {   // all lanes call
    const uint32_t total = min(__hip_atomic_load(s_cnt_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT), kRingCap);      // (appends beyond the ring went out one by one)
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&ctr->hits, (unsigned long long)total);
    base = ((unsigned long long)__shfl((uint32_t)(base >> 32), 0, 64) << 32) | __shfl((uint32_t)base, 0, 64);
    for (uint32_t k = lane; k < total; k += 64u) {
        const unsigned long long at = base + k;
        if (at < cap) {
            fmgpu_hit rec;
            rec.qidx = ring[k]; rec.lb = ring[kRingCap + k]; rec.lb_rev = ring[2u * kRingCap + k]; rec.len = ring[3u * kRingCap + k];
            rec.errors = ring[4u * kRingCap + k]; rec.seq = ring[5u * kRingCap + k];
            if constexpr (kWide) { const uint64_t hi = ring[6u * kRingCap + k]; rec.lb |= (hi & 0xffull) << 32; rec.lb_rev |= ((hi >> 8) & 0xffull) << 32; rec.len |= ((hi >> 16) & 0xffull) << 32; }
            out[at] = rec;
        }
    }
    if (lane == 0) __hip_atomic_store(s_cnt_w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// NSTEP: nodes a lane visits per pass through the wave-synchronous part (work sharing, refill, ring flush: ~150 of the ~550 wave instructions of an
// iteration — the loop is bound by instruction issue): a lane that runs out of work inside the inner loop waits for the next pass
// DENSE: the blocks are Format D (fmgpu_common.h; 32 bytes per 64 rows: two vector-memory instructions per interval end instead of three — the loop is bound
// by their number); the delimiter rows, written as 'A' there, are corrected from a list behind a filter, both staged in LDS.
constexpr uint32_t kDenseFilterBits = 32768;      // block number mod this: 25 delimiter rows mark 0.08 % of the blocks, ~7 % of the iterations of a wave meet one
// LUT: searches may start from the prefix table (la.lut): an instantiation of its own, so that the kernel of the plain index carries none of it (with the start
// path compiled into the one kernel the genome text went from 95.7 to 106.8 ms although no table was there to be used)
// BOARD: the waves of the launch share work too (the board, fmgpu_search_shared.h) — an instantiation of its own, taken for the batches whose end it shortens (launch_lean)
template <int WAVES, int NSTEP, bool DENSE, bool LUT = false, bool BOARD = false>      // WAVES: waves per SIMD the register allocation must allow (= resident 256-lane blocks per CU)
__global__ __launch_bounds__(256, WAVES) void k_scheme_lean(LeanArgs la, const uint8_t* __restrict__ qbuf, const uint64_t* __restrict__ qoff, uint64_t nq, idx_t n,
                                                             fmgpu_hit* __restrict__ out, uint64_t cap, Counters* ctr, ulonglong2* __restrict__ frames, uint64_t nlanes,
                                                             uint32_t qwords, const uint32_t* __restrict__ qmap, uint32_t refill_waste, uint32_t share_heavy, WorkBoard* board, unsigned long long* next_ctr) {
    extern __shared__ uint32_t s_dyn[];                             // [qwords][256] staged reads | [256] top frames (16 B) | [S][m + 1] steps | [4] ring fill | 4 x [kRingWords][kRingCap] rings
    __shared__ uint64_t s_sup[2u * kLeanSuperRows * 5u];            // 64-bit rows: the super tables of bwt and bwtRev
    __shared__ uint32_t s_filt[DENSE ? 2u * kDenseFilterBits / 32u : 1u];   // Format D: which blocks (mod kDenseFilterBits) may hold a delimiter row, per direction
    __shared__ uint32_t s_ex[DENSE ? 2u * 256u : 1u];               // ... and the rows themselves, ascending
    const uint32_t S = la.S, m = la.m, stride = la.m + 1;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
    lds_u32x4* const s_tos = (lds_u32x4*)(s_dyn + (size_t)qwords * 256u);
    lds_u32x4* const tos_slot = s_tos + tid;                        // the frame at depth sp - 1 of this lane (valid while sp > sbase)
    lds_u32x4* const s_bos = s_tos + 256;
    lds_u32x4* const bos_slot = s_bos + tid;                        // the frame at depth sbase (the next one a lane hands over), valid while sp > sbase
    uint32_t* const s_steps = s_dyn + (size_t)qwords * 256u + 2048u;
    lds_u32* const s_cnt_w = (lds_u32*)(s_steps + S * stride + wave);
    uint32_t* const ring = s_steps + S * stride + 4u + wave * (kRingWords * kRingCap);
    for (uint32_t i = tid; i < S * stride; i += 256u) s_steps[i] = la.steps[i];
    if (tid < 4u) s_steps[S * stride + tid] = 0u;
    if constexpr (DENSE) {
        for (uint32_t i = tid; i < 2u * kDenseFilterBits / 32u; i += 256u) s_filt[i] = 0u;
        for (uint32_t i = tid; i < 512u; i += 256u) s_ex[i] = i < 256u ? (i < la.nex_fw ? la.ex_fw[i] : 0xffffffffu) : (i - 256u < la.nex_rv ? la.ex_rv[i - 256u] : 0xffffffffu);
        __syncthreads();
        for (uint32_t i = tid; i < 512u; i += 256u) {
            const uint32_t r = s_ex[i];
            if (r != 0xffffffffu) { const uint32_t bk = (r >> 6) & (kDenseFilterBits - 1u); atomicOr(&s_filt[(i >> 8) * (kDenseFilterBits / 32u) + (bk >> 5)], 1u << (bk & 31u)); }
        }
    }
    const uint64_t* sup_fw = la.sup_fw; const uint64_t* sup_rv = la.sup_rv;
    if constexpr (kWide) {
        if (la.super_rows <= kLeanSuperRows) {
            for (uint32_t i = tid; i < la.super_rows * 5u; i += 256u) { s_sup[i] = la.sup_fw[i]; s_sup[kLeanSuperRows * 5u + i] = la.sup_rv[i]; }
            sup_fw = s_sup; sup_rv = s_sup + kLeanSuperRows * 5u;
        }
    }
    __syncthreads();

    const uint32_t gid = blockIdx.x * 256u + tid;                   // frame d of this lane at frames[d * nlanes + gid]
    uint32_t nodes = 0, mark = 0, waste = 0, nodes0 = 0;            // nodes0: the lane's count when it took its current read (or subtree)
    uint32_t lut_nodes = 0;                                         // nodes that prefix-table entries stood for (fmgpu_stats::table_steps)
    uint32_t blk_loads = 0, multi_nodes = 0;                        // blocks this lane fetched (a second end in another block counts) / visited nodes of several rows: fmgpu_stats::table_accesses, ::table_bytes
#ifdef FMGPU_DEV_STAMPS
    unsigned long long st_top = 0, st_share = 0, st_refill = 0, st_sync = 0, st_issue = 0, st_wait = 0, st_node = 0, st_tail = 0, st_t = __builtin_amdgcn_s_memtime(), st_steps = 0;
#define STAMP(acc) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - st_t; st_t = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(acc) do { } while (0)
#endif
#ifdef FMGPU_DEV
    uint32_t dev_multi = 0, dev_iter = 0, dev_busy = 0;             // dev build: lane-iterations on multi-row nodes / wave iterations / busy lane-iterations (reported through table_accesses, table_bytes)
    uint32_t dev_slot_bad = 0;                                      // dev build: pops / hand-overs whose LDS slot did not hold the frame that is in HBM (reported through table_steps: must be 0)
#endif
    bool have = false, exhausted = n == 0, need_start = false, is_task = false, odd = false, in_tail = false;
    uint32_t q = 0, si = 0, e = 0, j = 0, sp = 0, sbase = 0, resume = kLeanNoResume;
    idx_t lb = 0, lbRev = 0, len = 0;
    uint32_t k1 = 0, k2 = 0;                                        // path key fields of the 1st / 2nd substitution on the lane's path: (m - step) << 8 | symbol, 0 = none (key_with / key_prefix)
    [[maybe_unused]] uint32_t pass = 0, board_heavy = 0, board_period = 1;
    if constexpr (BOARD) { board_enter(board, lane); board_heavy = board->heavy; board_period = board->period - 1u; }      // (the period is a power of two: its mask)
    // Order of the top-frame slot's accesses.  The slot is refilled by an LDS-DMA load issued at the END of an iteration (after a pop); every other
    // access of the slot in the node phase (push, pop) comes after the lane has consumed its block loads of that iteration, which were issued after the
    // DMA: vector-memory operations of a wave complete in order, so the DMA has landed.  The one access outside the node phase (handing the only
    // frame to an idle lane) waits for vmcnt(0) itself.
    for (;;) {
        // ---- wave-synchronous part
        STAMP(st_top);
        {
            // a lane offers the bottom frame of its stack once its read has proven heavy (share_heavy nodes; the median read visits a few hundred, a read in a
            // satellite 10^5): with an offer from every read after 2 nodes (round 2) the hand-over ran in nearly every pass and cost a quarter of the kernel's cycles
            const bool can_give = have && sp > sbase && nodes - mark >= kShareNodes && nodes - nodes0 >= share_heavy;
            const uint64_t idlem = __ballot(!have), offerm = __ballot(can_give);
            if (idlem && offerm) {                                  // the i-th idle lane takes the bottom frame of the i-th offering lane
                const WavePairs wp(idlem, offerm, lane);
                const bool give = wp.gives(can_give);
                const bool take = wp.takes(!have);
                uint64_t w0 = 0, w1 = 0, w2 = 0;
                if (give) {
                    {                                               // the bottom frame from its LDS slot (written by the push onto an empty stack, or refilled below: an
                        asm volatile("" ::: "memory");              // LDS-DMA issued a whole pass ago — the node steps in between consumed block loads issued after it)
                        const u32x4 t = *bos_slot;
                        w0 = (uint64_t)t.x | ((uint64_t)t.y << 32); w1 = (uint64_t)t.z | ((uint64_t)t.w << 32);
#ifdef FMGPU_DEV
                        {   // the invariant the slot rests on: it holds what the write-through stack holds at depth sbase
                            const u32x4 hb = *reinterpret_cast<const u32x4*>(frames + ((uint64_t)sbase * nlanes + gid));
                            if (hb.x != t.x || hb.y != t.y || hb.z != t.z || hb.w != t.w) ++dev_slot_bad;
                        }
#endif
                    }
                    { const uint32_t fe = lean_frame_errors(w1); w2 = ((uint64_t)(fe >= 1u ? k1 : 0u) << 32) | (fe >= 2u ? k2 : 0u); }      // the key of the frame's node: the fields of later substitutions cleared
                    ++sbase; mark = nodes;
                    if (sp > sbase) {                               // the new bottom frame, straight into the slot (nothing waits for it)
                        uint32_t g = gid; asm volatile("" : "+v"(g));
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(frames + ((uint64_t)sbase * nlanes + g)),
                                                         (__attribute__((address_space(3))) void*)(s_bos + wave * 64u), 16, 0, 0);
                    }
                }
                const int vl = wp.partner(take);                       // my partner: the lane whose frame this lane takes
                const uint64_t tw0 = __shfl(w0, vl, 64), tw1 = __shfl(w1, vl, 64), tw2 = __shfl(w2, vl, 64);
                const uint32_t tq = __shfl(q, vl, 64), tsi = __shfl(si, vl, 64), todd = __shfl((uint32_t)odd, vl, 64);
                if (take) {
                    q = tq; si = tsi; odd = todd != 0u;
                    lean_unpack(tw0, tw1, lb, lbRev, len, j, e, resume);
                    k1 = (uint32_t)(tw2 >> 32); k2 = (uint32_t)tw2;
                    const uint32_t vt = (tid & ~63u) | (uint32_t)vl;
                    for (uint32_t w = 0; w < qwords; ++w) s_dyn[w * 256u + tid] = s_dyn[w * 256u + vt];     // the partner's staged read
                    have = true; is_task = true; need_start = false; sp = 0; sbase = 0; in_tail = false; mark = nodes; nodes0 = nodes;
                }
            }
        }
        if constexpr (BOARD) {
            // ... and with the other waves of the launch: every board_period-th pass a wave with subtrees to give looks whether a wave waits for work, and hands it the bottom
            // frames of ALL its offering lanes.  (A lane that gave inside the wave in this pass is no candidate — its mark is fresh —, so the slot read here was refilled a pass ago.)
            if ((++pass & board_period) == 0u) {
                const bool cand = have && sp > sbase && nodes - mark >= kShareNodes && nodes - nodes0 >= board_heavy;
                const uint64_t cm = __ballot(cand);
                uint32_t bslot = 0, bidx = 0;
                if (cm && board_reserve(board, lane, &bslot, &bidx)) {
                    // a batch holds 64 subtrees: few offering lanes give SEVERAL frames each, from the bottom of their stacks up (a heavy read walks down its match path and leaves a
                    // frame of untried substitutions per symbol: given one per look, a hundred of them were a chain of a hundred looks)
                    const uint32_t per = min(kBoardFramesPerLane, 64u / (uint32_t)__popcll(cm));
                    const uint32_t mine = cand ? min(per, sp - sbase) : 0u;
                    const uint32_t r0 = wave_excl_scan(mine, lane), total = __shfl(r0 + mine, 63, 64);
                    for (uint32_t i = 0; i < mine; ++i) {
                        u32x4 t;
                        if (i == 0u) { asm volatile("" ::: "memory"); t = *bos_slot; }
                        else t = *reinterpret_cast<const u32x4*>(frames + ((uint64_t)(sbase + i) * nlanes + gid));      // (pushes write through: the stack in HBM is complete)
                        const uint32_t fe = lean_frame_errors((uint64_t)t.z | ((uint64_t)t.w << 32));
                        const uint32_t w[8] = {t.x, t.y, t.z, t.w, fe >= 1u ? k1 : 0u, fe >= 2u ? k2 : 0u, q, si};
#pragma unroll
                        for (uint32_t d_ = 0; d_ < 8u; ++d_) board_put(board, bslot, d_, r0 + i, w[d_]);
                    }
                    if (mine) {
                        sbase += mine; mark = nodes;
                        if (sp > sbase) {                           // the new bottom frame, straight into the slot
                            uint32_t g = gid; asm volatile("" : "+v"(g));
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(frames + ((uint64_t)sbase * nlanes + g)),
                                                             (__attribute__((address_space(3))) void*)(s_bos + wave * 64u), 16, 0, 0);
                        }
                    }
                    board_publish(board, lane, bslot, bidx, total);
                }
            }
        }
        STAMP(st_share);
        {
            const uint64_t needm = __ballot(!have && !exhausted), busym = __ballot(have);
            waste += (uint32_t)__popcll(needm);
            if (needm && (waste >= refill_waste || !busym)) {
                waste = 0;
                const bool want = !have && !exhausted;
                bool fresh = false; uint64_t qo = 0;
                const uint64_t got = wave_hand_out_at(want, next_ctr, lane);
                if (want) {
                    if (got >= nq) exhausted = true;
                    else { q = qmap ? qmap[got] : (uint32_t)got; qo = qoff[q]; fresh = true; have = true; is_task = false; si = 0; need_start = true; mark = nodes; nodes0 = nodes; }
                }
                const bool ok = stage2_sync(s_dyn, qbuf, qo, m, fresh);
                if (fresh) odd = !ok;
                __builtin_amdgcn_s_waitcnt(0x0f70);                // vmcnt(0) here, in the rare path: or the compiler, unsure whether a staging load is still pending on some path,
                                                                    // waits for vmcnt(0) in EVERY iteration before it overwrites one of their registers — ahead of the block loads
            }
        }
        STAMP(st_refill);
        {
            const uint32_t filled = __hip_atomic_load(s_cnt_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            const uint64_t busy = __ballot(have);
            if (filled >= kRingFlush || (!busy && filled)) ring_flush(s_cnt_w, ring, lane, out, cap, ctr);
            if (!busy) {
                if (__ballot(!exhausted) != 0ull) continue;
                if constexpr (!BOARD) break;
                else {
                    // the wave is out of work and the batch is handed out: it waits for subtrees of the reads other waves still walk, or for the end of the launch
                    uint32_t bslot = 0, bidx = 0;
                    const uint32_t k = board_wait(board, lane, &bslot, &bidx);
                    if (k == 0u) break;
                    const bool fresh = lane < k; uint64_t qo = 0;
                    if (fresh) {
                        uint32_t w[8];
#pragma unroll
                        for (uint32_t d_ = 0; d_ < 8u; ++d_) w[d_] = board_word(board, bslot, d_, lane);
                        lean_unpack((uint64_t)w[0] | ((uint64_t)w[1] << 32), (uint64_t)w[2] | ((uint64_t)w[3] << 32), lb, lbRev, len, j, e, resume);
                        k1 = w[4]; k2 = w[5]; q = w[6]; si = w[7];
                        qo = qoff[q];
                        have = true; is_task = true; need_start = false; sp = 0; sbase = 0; in_tail = false; mark = nodes; nodes0 = nodes;
                    }
                    board_release(board, lane, bslot, bidx);
                    const bool ok = stage2_sync(s_dyn, qbuf, qo, m, fresh);
                    if (fresh) odd = !ok;
                    __builtin_amdgcn_s_waitcnt(0x0f70);
                    continue;
                }
            }
        }
        STAMP(st_sync);
#pragma unroll 1
        for (int step = 0; step < NSTEP; ++step) {
        if (!have) continue;
#ifdef FMGPU_DEV_STAMPS
        ++st_steps;
#endif
        // ---- one node per lane
        bool lut_start = false; uint32_t lut_code = 0;
        if (need_start) {                                           // search_impl (SearchNg26.h:385-390) -> run(): :62-79
            need_start = false;
            lb = 0; lbRev = 0; len = n; e = 0; j = 0; sp = 0; sbase = 0; resume = kLeanNoResume; in_tail = false;
            k1 = 0; k2 = 0;
            // the always-exact first part of the search (u[0] = 0; it extends to the right, so its first lutL symbols are consecutive in the read) starts from its entry of
            // the prefix table: this iteration's ONE load is that entry instead of a block, and lutL nodes — the ones whose interval ends lie furthest apart — are not walked
            if (LUT && ((la.lut_ok >> si) & 1u) && !odd) {
                const uint32_t p0 = s_steps[si * stride] & 0xffffu, w = p0 >> 4;
                lut_code = __funnelshift_r(s_dyn[w * 256u + tid], s_dyn[(w + 1u) * 256u + tid], (p0 & 15u) * 2u) & (la.lutL >= 16u ? 0xffffffffu : (1u << (2u * la.lutL)) - 1u);
                lut_start = true;
            }
        }
        const uint32_t ent = s_steps[si * stride + j];
        const bool right = (ent >> 16) & 1u, multi = !(LUT && lut_start) && len > 1u;
#ifdef FMGPU_DEV
        dev_multi += multi ? 1u : 0u; ++dev_busy; if (lane == (uint32_t)__ffsll((unsigned long long)__ballot(true)) - 1u) ++dev_iter;
#endif
        const idx_t a = right ? lbRev : lb, b = a + len;
        const uint8_t* blk = right ? (DENSE ? reinterpret_cast<const uint8_t*>(la.drv) : la.rv) : (DENSE ? reinterpret_cast<const uint8_t*>(la.dfw) : la.fw);
        // memory phase: the entries of symbols 1..4 of the block(s) of both interval ends (one end for a one-row node) — Format A: 48 bytes, 12 into the 64-byte
        // block; Format D: the whole 32-byte block
        constexpr uint32_t kBlk = DENSE ? 32u : 64u, kOff = DENSE ? 0u : 12u;
        const Quad4* pa = (LUT && lut_start) ? reinterpret_cast<const Quad4*>(la.lut + lut_code) : reinterpret_cast<const Quad4*>(blk + (size_t)(a >> 6) * kBlk + kOff);
        const Quad4 a0 = pa[0], a1 = pa[1];                         // (a table entry is 16 bytes: what follows it is read and ignored — the tables carry that much slack)
        Quad4 a2, b0, b1, b2;                                       // (only a multi-row node reads b: no default, or the compiler waits for A before it asks for B)
        if constexpr (!DENSE) a2 = pa[2];
        // the second end's block only when it is another one (the ends of a short interval share their block)
        const bool far = multi && (a >> 6) != (b >> 6);
        if (far) { const Quad4* pb = reinterpret_cast<const Quad4*>(blk + (size_t)(b >> 6) * kBlk + kOff); b0 = pb[0]; b1 = pb[1]; if constexpr (!DENSE) b2 = pb[2]; }
        blk_loads += (LUT && lut_start) ? 0u : (far ? 2u : 1u);
        STAMP(st_issue);
#ifdef FMGPU_DEV_STAMPS
        __builtin_amdgcn_s_waitcnt(0x0f70); STAMP(st_wait);
#endif
        const uint32_t pos = ent & 0xffffu, minE = (ent >> 18) & 0x1fu, maxE = (ent >> 23) & 0x3fu;
        const bool lastp = (ent >> 17) & 1u;
        uint32_t c;
        if (odd) { const uint32_t v = qbuf[qoff[q] + pos]; c = v < 5u ? v : 255u; }      // a read with bytes outside 1..4: from global memory (rare)
        else c = ((s_dyn[(pos >> 4) * 256u + tid] >> ((pos & 15u) * 2u)) & 3u) + 1u;
        // LF of symbols 1..4 at a (and b): cnt + popcount(bits below the row)
        const uint32_t abit = (uint32_t)a & 63u, bbit = (uint32_t)b & 63u;
        const uint32_t ma_lo = abit >= 32u ? 0xffffffffu : (1u << abit) - 1u, ma_hi = abit >= 32u ? (1u << (abit - 32u)) - 1u : 0u;
        const uint32_t mb_lo = bbit >= 32u ? 0xffffffffu : (1u << bbit) - 1u, mb_hi = bbit >= 32u ? (1u << (bbit - 32u)) - 1u : 0u;
        idx_t la1, la2, la3, la4;
        // Format D: delimiter rows before position i of its block (they sit in the planes as code 0 and were counted as symbol 1); `at` = whether row i itself is one
        [[maybe_unused]] auto delims_before = [&](idx_t i, bool& at) -> uint32_t {
            const uint32_t bk = ((uint32_t)i >> 6) & (kDenseFilterBits - 1u), dir = right ? 1u : 0u;
            at = false;
            if (!((s_filt[dir * (kDenseFilterBits / 32u) + (bk >> 5)] >> (bk & 31u)) & 1u)) return 0u;
            const uint32_t* ex = s_ex + dir * 256u;
            const uint32_t first = (uint32_t)i & ~63u;
            uint32_t cnt = 0;
            for (uint32_t t = 0; t < 256u && ex[t] <= (uint32_t)i; ++t) { if (ex[t] >= first && ex[t] < (uint32_t)i) ++cnt; if (ex[t] == (uint32_t)i) at = true; }
            return cnt;
        };
        bool a_is_delim = false;
        if constexpr (DENSE) {
            // a0 = occ[1..4], a1 = {p0 lo, p0 hi, p1 lo, p1 hi}: symbol c matches where the two planes spell c - 1
            const uint32_t n0l = ~a1.x, n0h = ~a1.y, n1l = ~a1.z, n1h = ~a1.w;
            la1 = a0.x + __popc(n0l & n1l & ma_lo) + __popc(n0h & n1h & ma_hi) - delims_before(a, a_is_delim);
            la2 = a0.y + __popc(a1.x & n1l & ma_lo) + __popc(a1.y & n1h & ma_hi);
            la3 = a0.z + __popc(n0l & a1.z & ma_lo) + __popc(n0h & a1.w & ma_hi);
            la4 = a0.w + __popc(a1.x & a1.z & ma_lo) + __popc(a1.y & a1.w & ma_hi);
        } else {
            la1 = a0.x + __popc(a0.y & ma_lo) + __popc(a0.z & ma_hi); la2 = a0.w + __popc(a1.x & ma_lo) + __popc(a1.y & ma_hi);
            la3 = a1.z + __popc(a1.w & ma_lo) + __popc(a2.x & ma_hi); la4 = a2.y + __popc(a2.z & ma_lo) + __popc(a2.w & ma_hi);
        }
        if constexpr (kWide) {                                      // block counts are relative to the super-block: the rest from the super table
            const uint64_t* sp_ = (right ? sup_rv : sup_fw) + (size_t)(a >> kSuperShift) * 5u;
            la1 += sp_[1]; la2 += sp_[2]; la3 += sp_[3]; la4 += sp_[4];
        }
        bool back = false, search_over = false;
        const bool mOK = minE <= e && e <= maxE, sOK = minE <= e + 1u && e + 1u <= maxE, xOK = e + 1u <= maxE;
        if (LUT && lut_start) {                                     // {lb, lbRev, len, nodes the walk of these symbols visits (fewer than lutL where the interval empties on the way)}
            lb = (idx_t)a0.x; lbRev = (idx_t)a0.y; len = (idx_t)a0.z; nodes += a0.w; lut_nodes += a0.w;
            j = la.lutL; in_tail = true;                            // (the first part is longer than lutL: the rest of it is exact too)
            if (len == 0u) back = true;
        } else
        if (multi) {
            // ---- extend-all node (search_next_dir, :143-224) or exact-tail step over several rows
            if (!far) { b0 = a0; b1 = a1; if constexpr (!DENSE) b2 = a2; }   // (after the loads have been waited for: selects, no copy ahead of the second end's loads)
            idx_t lb1, lb2, lb3, lb4;
            if constexpr (DENSE) {
                const uint32_t n0l = ~b1.x, n0h = ~b1.y, n1l = ~b1.z, n1h = ~b1.w;
                bool unused_at;
                lb1 = b0.x + __popc(n0l & n1l & mb_lo) + __popc(n0h & n1h & mb_hi) - delims_before(b, unused_at);
                lb2 = b0.y + __popc(b1.x & n1l & mb_lo) + __popc(b1.y & n1h & mb_hi);
                lb3 = b0.z + __popc(n0l & b1.z & mb_lo) + __popc(n0h & b1.w & mb_hi);
                lb4 = b0.w + __popc(b1.x & b1.z & mb_lo) + __popc(b1.y & b1.w & mb_hi);
            } else {
                lb1 = b0.x + __popc(b0.y & mb_lo) + __popc(b0.z & mb_hi); lb2 = b0.w + __popc(b1.x & mb_lo) + __popc(b1.y & mb_hi);
                lb3 = b1.z + __popc(b1.w & mb_lo) + __popc(b2.x & mb_hi); lb4 = b2.y + __popc(b2.z & mb_lo) + __popc(b2.w & mb_hi);
            }
            if constexpr (kWide) {
                const uint64_t* sp_ = (right ? sup_rv : sup_fw) + (size_t)(b >> kSuperShift) * 5u;
                lb1 += sp_[1]; lb2 += sp_[2]; lb3 += sp_[3]; lb4 += sp_[4];
            }
            const idx_t d1 = lb1 - la1, d2 = lb2 - la2, d3 = lb3 - la3, d4 = lb4 - la4;
            const idx_t d0 = len - (d1 + d2 + d3 + d4);             // rows of the interval that hold the delimiter
            const uint32_t alive = (uint32_t)min(d0, (idx_t)1) | ((uint32_t)min(d1, (idx_t)1) << 1) | ((uint32_t)min(d2, (idx_t)1) << 2) | ((uint32_t)min(d3, (idx_t)1) << 3) | ((uint32_t)min(d4, (idx_t)1) << 4);
            const bool resuming = resume != kLeanNoResume;
            uint32_t subs = alive & ~1u & ~(c < 5u ? 1u << c : 0u);  // substitution children: FirstSymb = 1 (fmindex/BiFMIndex.h:26), != query symbol
            if (!sOK) subs = 0u;
            if (resuming) subs &= ~((1u << resume) - 1u);
            const bool take_match = !resuming && mOK && c < 5u && ((alive >> c) & 1u);     // match child first (:171-181)
            const bool take_sub = !take_match && subs != 0u;
            uint32_t take = c;
            if (take_sub) { take = (uint32_t)__ffs((int)subs) - 1u; subs &= subs - 1u; }
            { const uint32_t cnt_ = (!resuming && (in_tail || xOK || mOK)) ? 1u : 0u; nodes += cnt_; multi_nodes += cnt_; }
            if ((take_match || take_sub) && subs) {                 // (re-)push the parent: its remaining siblings start at the lowest of subs
                uint64_t fw0, fw1;
                lean_pack(lb, lbRev, len, j, e, (uint32_t)__ffs((int)subs) - 1u, fw0, fw1);
                const u32x4 f = {(uint32_t)fw0, (uint32_t)(fw0 >> 32), (uint32_t)fw1, (uint32_t)(fw1 >> 32)};
                uint32_t g = gid; asm volatile("" : "+v"(g));     // (kept out of loop-invariant hoisting: the 64-bit address of the lane's frame column would hold two registers through the loop)
                *reinterpret_cast<u32x4*>(frames + ((uint64_t)sp * nlanes + g)) = f;     // write-through: the stack in HBM is always complete
                *tos_slot = f;
                if (sp == sbase) *bos_slot = f;                     // the first frame of the stack is also its bottom
                ++sp;
            }
            resume = kLeanNoResume;
            if (take_match || take_sub) {
                idx_t kla = la1, kd = d1, pre = d0;
                if (take == 2u) { kla = la2; kd = d2; pre = d0 + d1; }
                else if (take == 3u) { kla = la3; kd = d3; pre = d0 + d1 + d2; }
                else if (take == 4u) { kla = la4; kd = d4; pre = d0 + d1 + d2 + d3; }
                else if (take == 0u) { kla = a + la.ksum - (la1 + la2 + la3 + la4); kd = d0; pre = 0u; }   // a delimiter in the QUERY matched against delimiter rows: the ranks of all symbols at a add up to a
                len = kd;
                if (right) { lbRev = kla; lb += pre; } else { lb = kla; lbRev += pre; }     // fmindex/BiFMIndexCursor.h:58-82
                if (take_sub) { const uint32_t kf = ((m - j) << 8) | take; if (e == 0u) k1 = kf; else if (e == 1u) k2 = kf; e += 1u; }
                in_tail = !lastp && (in_tail || (take_match && !xOK));
                ++j;
            } else back = true;
        } else {
            // ---- single row (search_next_dir_single, :251-365): the only child is the BWT symbol of the row, read off its block
            const uint32_t bt_lo = abit >= 32u ? 0u : 1u << abit, bt_hi = abit >= 32u ? 1u << (abit - 32u) : 0u;
            uint32_t row_sym = 0u; idx_t t0 = 0;
            if constexpr (DENSE) {                                  // the row's symbol from the two planes; a delimiter row (listed) ends the walk
                const uint32_t code = (((a1.x & bt_lo) | (a1.y & bt_hi)) ? 1u : 0u) | (((a1.z & bt_lo) | (a1.w & bt_hi)) ? 2u : 0u);
                row_sym = a_is_delim ? 0u : code + 1u;
                t0 = code == 0u ? la1 : (code == 1u ? la2 : (code == 2u ? la3 : la4));
            } else {
                if ((a0.y & bt_lo) | (a0.z & bt_hi)) { row_sym = 1u; t0 = la1; }
                if ((a1.x & bt_lo) | (a1.y & bt_hi)) { row_sym = 2u; t0 = la2; }
                if ((a1.w & bt_lo) | (a2.x & bt_hi)) { row_sym = 3u; t0 = la3; }
                if ((a2.z & bt_lo) | (a2.w & bt_hi)) { row_sym = 4u; t0 = la4; }
            }
            const bool is_match = row_sym >= 1u && row_sym == c && mOK;
            nodes += 1u + ((!in_tail && is_match && !xOK) ? 1u : 0u);
            bool dead = false;
            if (row_sym < 1u) dead = true;                          // :295-297: a delimiter row ends the walk
            else if (row_sym == c) { if (!mOK) dead = true; }
            else if (sOK) { const uint32_t kf = ((m - j) << 8) | row_sym; if (e == 0u) k1 = kf; else if (e == 1u) k2 = kf; e += 1u; }
            else dead = true;
            if (dead) back = true;
            else { in_tail = !lastp && (in_tail || (is_match && !xOK)); if (right) lbRev = t0; else lb = t0; ++j; }
        }
        STAMP(st_node);
        if (!back && j == m) {                                      // search_next at part == P (:101-108)
            const uint32_t fin = s_steps[si * stride + m];
            if (((fin >> 18) & 0x1fu) <= e && e <= ((fin >> 23) & 0x3fu)) {
                const uint32_t slot = __hip_atomic_fetch_add(s_cnt_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                const uint32_t kw = e | ((si << 16 | k1 >> 8) << 8), ks = (k1 << 24) | k2;    // path key = si:8 | k1:24 | k2:24 (key_with): errors word carries its upper 24 bits, seq the lower 32
                if (slot < kRingCap) {                              // the wave flushes at kRingFlush and appends at most 64 NSTEP records per pass: room unless NSTEP > 2
                    ring[slot] = q; ring[kRingCap + slot] = (uint32_t)lb; ring[2u * kRingCap + slot] = (uint32_t)lbRev; ring[3u * kRingCap + slot] = (uint32_t)len;
                    ring[4u * kRingCap + slot] = kw; ring[5u * kRingCap + slot] = ks;
                    if constexpr (kWide) ring[6u * kRingCap + slot] = (uint32_t)((uint64_t)lb >> 32) | ((uint32_t)((uint64_t)lbRev >> 32) << 8) | ((uint32_t)((uint64_t)len >> 32) << 16);
                } else emit_hit(out, cap, ctr, q, Cur{lb, lbRev, len}, kw, ks);
            }
            back = true;
        }
        if (back) {
            if (sp == sbase) search_over = true;
            else {
                --sp;
                asm volatile("" ::: "memory");                      // (the slot is read here, not ahead of the branch)
                const u32x4 t = *tos_slot;                          // the cached top frame ...
#ifdef FMGPU_DEV
                {   // the ordering invariant above, checked: the LDS-DMA that refilled the slot has landed, so the slot holds the frame the stack holds at depth sp
                    const u32x4 hb = *reinterpret_cast<const u32x4*>(frames + ((uint64_t)sp * nlanes + gid));
                    if (hb.x != t.x || hb.y != t.y || hb.z != t.z || hb.w != t.w) ++dev_slot_bad;
                }
#endif
                lean_unpack((uint64_t)t.x | ((uint64_t)t.y << 32), (uint64_t)t.z | ((uint64_t)t.w << 32), lb, lbRev, len, j, e, resume);
                if (e == 0u) k1 = 0u;                                // the key of the popped node: the fields of later substitutions cleared
                if (e <= 1u) k2 = 0u;
                in_tail = false;
                uint32_t g = gid; asm volatile("" : "+v"(g));
                if (sp > sbase)                                     // ... and the one below it, requested an iteration before it can be needed, straight into the slot
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(frames + ((uint64_t)(sp - 1u) * nlanes + g)),
                                                     (__attribute__((address_space(3))) void*)(s_tos + wave * 64u), 16, 0, 0);
            }
        }
        if (search_over) {                                          // the next search of the scheme, or the lane is out of work
            ++si;
            if (is_task) { have = false; is_task = false; }         // a task is one subtree of one search: its owner goes on with the other searches
            else if (si == S) have = false;
            else need_start = true;
        }
        STAMP(st_tail);
        }   // NSTEP
    }
    const uint32_t tot = wave_sum(nodes);
    if (lane == 0 && tot) atomicAdd(&ctr->nodes, (unsigned long long)tot);
#ifdef FMGPU_DEV_STAMPS
    if (lane == 0) {                                                // (wave-uniform sums, cycles of s_memtime; a debug area behind the counters)
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(ctr) + 8;
        atomicAdd(dbg + 0, st_sync); atomicAdd(dbg + 1, st_issue); atomicAdd(dbg + 2, st_wait); atomicAdd(dbg + 3, st_node); atomicAdd(dbg + 4, st_tail); atomicAdd(dbg + 5, st_steps); atomicAdd(dbg + 6, 1ull); atomicAdd(dbg + 7, st_share); atomicAdd(dbg + 8, st_refill); atomicAdd(dbg + 9, st_top);
    }
#endif
#ifdef FMGPU_DEV
    { const uint32_t a1 = wave_sum(dev_multi), a2 = wave_sum(dev_iter), a3 = wave_sum(dev_busy);
      if (lane == 0) { atomicAdd(&ctr->table_accesses, (unsigned long long)a1 | ((unsigned long long)a2 << 40)); atomicAdd(&ctr->table_bytes, (unsigned long long)a3); }
      const uint32_t bad = wave_sum(dev_slot_bad);
      if (lane == 0 && bad) atomicAdd(reinterpret_cast<unsigned long long*>(ctr) + 20, (unsigned long long)bad); }
    (void)blk_loads; (void)multi_nodes;
#else
    // what the kernel asked of the occurrence tables: table_accesses = blocks fetched (both ends of a node whose ends lie in two blocks; a node re-visited for its next
    // sibling fetches again), table_bytes = visited nodes of several rows (the rest of `nodes` stood on one row: one interval end) — bench.py prices both
    { const uint32_t a1 = wave_sum(blk_loads), a2 = wave_sum(multi_nodes);
      if (lane == 0 && (a1 | a2)) { atomicAdd(&ctr->table_accesses, (unsigned long long)a1); atomicAdd(&ctr->table_bytes, (unsigned long long)a2); } }
#endif
    { const uint32_t a3 = wave_sum(lut_nodes); if (lane == 0 && a3) atomicAdd(reinterpret_cast<unsigned long long*>(ctr) + 21, (unsigned long long)a3); }
}


// ---- search_backtracking ----------------------------------------------------------------------------------------
template <class Occ, int MAXSIG>
__global__ __launch_bounds__(256) void k_backtracking(Occ fw, bool bidir, const uint8_t* __restrict__ qbuf, const uint64_t* __restrict__ qoff,
                                                      uint64_t nq, idx_t n, uint32_t K, fmgpu_hit* __restrict__ out, uint64_t cap,
                                                      Counters* ctr, StackView stk, uint32_t qwords, uint32_t qnib) {
    extern __shared__ uint32_t s_query[];
    const QStage qst{s_query, qwords, qnib};
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t sigma = fw.sigma();
    uint32_t nodes = 0;
    for (uint64_t q = gid; q < nq; q += stk.nlanes) {
        const uint64_t qo = qoff[q];
        const uint32_t m = (uint32_t)(qoff[q + 1] - qo);
        const uint8_t* qs = qbuf + qo;
        if (m > stk.depth) continue;
        if (m) qstage_load(qst, qbuf, qo, m, sigma);
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
                const uint32_t r = qstage_get(qst, qs, m - i - 1);
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
                        const uint32_t nxt = subs.any() ? subs.first() : 256u;
                        if constexpr (kWide) {
                            stk.p0[o] = (uint64_t)cur.lb | ((uint64_t)i << 40);
                            stk.p1[o] = (uint64_t)cur.lbRev | ((uint64_t)e << 40);
                            stk.p2[o] = (uint64_t)cur.len | ((uint64_t)nxt << 40);
                        } else {
                            stk.p0[o] = (uint64_t)cur.lb | ((uint64_t)cur.lbRev << 32);
                            stk.p1[o] = (uint64_t)cur.len | ((uint64_t)i << 32);
                            stk.p2[o] = (uint64_t)nxt | ((uint64_t)e << 32);
                        }
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
            if constexpr (kWide) {
                const uint64_t m40 = (1ull << 40) - 1ull;
                cur.lb = (idx_t)(w0 & m40); i = (uint32_t)(w0 >> 40); cur.lbRev = (idx_t)(w1 & m40); e = (uint32_t)(w1 >> 40);
                cur.len = (idx_t)(w2 & m40); resume = (uint32_t)(w2 >> 40);
            } else {
                cur.lb = (idx_t)w0; cur.lbRev = (idx_t)(w0 >> 32); cur.len = (idx_t)w1; i = (uint32_t)(w1 >> 32);
                resume = (uint32_t)w2; e = (uint32_t)(w2 >> 32);
            }
        }
    }
    uint32_t tot = wave_sum(nodes);
    if ((threadIdx.x & 63u) == 0 && tot) atomicAdd(&ctr->nodes, (unsigned long long)tot);
}

static bool build_step_table(const SchemeDev& sd, uint32_t m, uint32_t lutL, uint32_t J, std::vector<uint32_t>& tab, uint32_t& lut_ok) {
    const uint32_t S = (uint32_t)sd.S, P = (uint32_t)sd.P;
    lut_ok = 0;
    if (m < P || m > 0xfffeu || (uint64_t)S * (m + 1) > 2730) return false;     // three tables of S * (m + 1) words in LDS: 32 KB at most
    std::vector<uint32_t> plen(P);
    uint32_t sum = 0;
    for (uint32_t p = 0; p < P; ++p) { plen[p] = sd.uniform ? m / P + (p < m % P ? 1u : 0u) : sd.partition[p]; sum += plen[p]; }
    if (sum != m) return false;
    const size_t half = (size_t)S * (m + 1);
    tab.assign(3 * half, 0);
    for (uint32_t s = 0; s < S; ++s) {
        const uint8_t* pi = sd.pi + s * kMaxParts; const uint8_t* L = sd.l + s * kMaxParts; const uint8_t* U = sd.u + s * kMaxParts;
        uint32_t* T = tab.data() + (size_t)s * (m + 1);
        uint32_t* T2 = tab.data() + half + (size_t)s * (m + 1);
        uint32_t* T3 = tab.data() + 2 * half + (size_t)s * (m + 1);
        uint32_t start = 0;
        for (uint32_t i = 0; i < pi[0]; ++i) start += plen[i];
        uint32_t qR = start, qL = start - 1, j = 0;              // SearchNg26.h:62-79
        for (uint32_t p = 0; p < P; ++p) {
            if (L[p] > U[p] || L[p] > 31 || U[p] > 63) return false;    // the table form relies on l <= u (search_scheme/isValid.h:87-91)
            if (p && (U[p] < U[p - 1] || L[p] < L[p - 1])) return false; // ... and on bounds that never shrink
            const bool right = p == 0 || pi[p - 1] < pi[p];      // :111
            const uint32_t len = plen[pi[p]];
            for (uint32_t k = 0; k < len; ++k, ++j) {
                const uint32_t pos = right ? qR++ : qL--;
                const bool last = k + 1 == len;
                T[j] = (pos & 0xffffu) | ((right ? 1u : 0u) << 16) | ((last ? 1u : 0u) << 17) | ((last ? (uint32_t)L[p] : 0u) << 18) | ((uint32_t)U[p] << 23);
            }
        }
        T[m] = ((uint32_t)L[P - 1] << 18) | ((uint32_t)U[P - 1] << 23);
        for (uint32_t k = 0; k < m; ++k) {                       // run: steps from k on with the same direction, capped at 3 and at the query end
            uint32_t run = 1;
            while (run < 3 && k + run < m && ((T[k + run] >> 16) & 1u) == ((T[k] >> 16) & 1u)) ++run;
            T[k] |= run << 30;
            {   // the stretch word of these `run` steps (same fields as for 16; tb = run if no part ends inside)
                uint32_t tb = run, ends = 0, lo = 0;
                for (uint32_t t = 0; t < run; ++t)
                    if ((T[k + t] >> 17) & 1u) { if (!ends) tb = t; ++ends; lo = std::max(lo, (T[k + t] >> 18) & 0x1fu); }
                const uint32_t hi1 = (T[k] >> 23) & 0x3fu, hi2 = tb + 1 < run ? (T[k + tb + 1] >> 23) & 0x3fu : hi1;
                T3[k] = tb | (hi1 << 5) | (hi2 << 11) | (lo << 17) | ((ends <= 1 ? 1u : 0u) << 22);
            }
            // run16: J steps in this direction from k on.  Their stretch word: tb = offset of the first part end inside the stretch (J if none),
            // hi1 / hi2 = upper bound before / after it, lo = the largest lower bound that applies inside, simple = at most one part end
            if (J && J <= 31 && k + J <= m) {
                bool same = true; uint32_t tb = J, ends = 0, lo = 0;
                for (uint32_t t = 0; t < J; ++t) {
                    same = same && ((T[k + t] >> 16) & 1u) == ((T[k] >> 16) & 1u);
                    if ((T[k + t] >> 17) & 1u) { if (!ends) tb = t; ++ends; lo = std::max(lo, (T[k + t] >> 18) & 0x1fu); }
                }
                if (same) {
                    T[k] |= 1u << 29;
                    const uint32_t hi1 = (T[k] >> 23) & 0x3fu, hi2 = tb + 1 < J ? (T[k + tb + 1] >> 23) & 0x3fu : hi1;
                    T2[k] = tb | (hi1 << 5) | (hi2 << 11) | (lo << 17) | ((ends <= 1 ? 1u : 0u) << 22);
                }
            }
        }
        if (lutL && plen[pi[0]] > lutL && U[0] == 0) lut_ok |= 1u << s;
    }
    return true;
}

// (length, query number) pairs for the length buckets of a ragged batch
// Heavy reads first.  The reads of a batch are handed out in order, and the few that sit in high-copy repeats visit 10^3 times the nodes of the median
// read: those that start last keep a handful of waves busy long after the others have drained (measured on the genome-like text: 90 % of the
// waves are done at 89 ms of 105).  The prefix table tells which reads these are before the search starts — the interval of the first 16
// symbols of the first search IS the copy number — so the batch is handed out with them in front (a stable partition of the read numbers).
struct LutPositions { uint32_t pos[16]; };
__global__ __launch_bounds__(256) void k_heavy_flags(const uint4* __restrict__ lut, uint32_t lutL, uint32_t R, LutPositions lp, const uint8_t* __restrict__ qbuf,
                                                     const uint64_t* __restrict__ qoff, uint64_t nq, uint32_t sigma, uint32_t threshold, uint8_t* __restrict__ flags,
                                                     uint32_t* __restrict__ count) {
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool heavy = false;
    if (q < nq) {
        const uint8_t* s = qbuf + qoff[q];
        uint32_t code = 0, mul = 1; bool valid = true;
        for (uint32_t t = 0; t < lutL; ++t) {
            const uint32_t c = s[lp.pos[t]];
            valid = valid && c >= 1 && c < sigma;
            code += (c - 1) * mul; mul *= R;
        }
        heavy = valid && lut[code].z > threshold;
        flags[q] = heavy ? 1 : 0;
    }
    const uint64_t m = __ballot(heavy);
    if ((threadIdx.x & 63u) == 0 && m) atomicAdd(count, (uint32_t)__popcll(m));
}
// the same question without tables: the interval of the read's last 16 symbols by backward search on the blocks (16 of the ~500 nodes a read visits)
template <class Occ>
__global__ __launch_bounds__(256) void k_heavy_flags_plain(Occ occ, idx_t n, const uint8_t* __restrict__ qbuf, const uint64_t* __restrict__ qoff, uint64_t nq, uint32_t m,
                                                           uint32_t threshold, uint8_t* __restrict__ flags, uint32_t* __restrict__ count) {   // m = 0: every read has its own length
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool heavy = false;
    if (q < nq) {
        const uint64_t o = qoff[q];
        const uint32_t len_q = m ? m : (uint32_t)(qoff[q + 1] - o);
        const uint8_t* s = qbuf + o;
        idx_t lb = 0, len = len_q >= 16u ? n : (idx_t)0;
        const uint32_t sigma = occ.sigma();
        for (uint32_t t = 0; t < 16u && len != 0; ++t) {
            const uint32_t c = s[len_q - 1u - t];
            if (c < 1 || c >= sigma) { len = 0; break; }
            idx_t ra, rb;
            occ.lf2(lb, lb + len, c, ra, rb);
            lb = ra; len = rb - ra;
        }
        heavy = len > threshold;
        flags[q] = heavy ? 1 : 0;
    }
    const uint64_t mk = __ballot(heavy);
    if ((threadIdx.x & 63u) == 0 && mk) atomicAdd(count, (uint32_t)__popcll(mk));
}
// rows of the 16-symbol interval above which a read counts as one of a high-copy repeat (genome-like text, plain index / with tables:
// > 2 rows 157.8 / 115.7 ms, > 8: 155.8 / 111.5, > 64: 151.3 / 111.1, > 1000: 157.1 / 114.6)
constexpr uint32_t kHeavyInterval = 64;
static uint32_t heavy_rows() { const char* e = dev_env("FMGPU_DEV_HEAVY_ROWS"); return e && atoi(e) > 0 ? (uint32_t)atoi(e) : kHeavyInterval; }   // (dev knob)

__global__ __launch_bounds__(256) void k_len_pairs(const uint64_t* __restrict__ qoff, uint64_t nq, uint32_t* __restrict__ len, uint32_t* __restrict__ idx) {
    uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nq) { len[q] = (uint32_t)(qoff[q + 1] - qoff[q]); idx[q] = (uint32_t)q; }
}

constexpr size_t kFrameCache = (size_t)2 << 30;
constexpr uint32_t kDfsSlots = 8;            // launches of a ragged batch that run side by side at most (CallScratch holds that many streams)
struct DfsWorkspace {
    uint64_t* planes = nullptr; Counters* ctr = nullptr; StackView view{};
    unsigned grid = 0;
    bool own_planes = false;
    // blocks_per_cu: resident 256-lane blocks of the kernel that will run (the lanes walk the batch with a static stride, so
    // every block must be resident from the start or the late ones form a tail).
    // The frame stacks (~1 GB for a full-chip launch over 101-symbol reads) stay with the calling host thread between calls: allocating and
    // freeing them per call costs ~0.2 ms (hipFree synchronises the device), 2-3 % of a 10 M-read k = 2 call.
    WorkBoard* board = nullptr;                                    // (with_board) sharing between the waves of a launch
    // slots > 1 (a ragged batch: one launch per read length): that many launches run side by side, each on a stream, a share of the grid, a stretch of the frame stacks, a hand-out counter
    // and a board of its own — `grid` is then ONE launch's grid, `view` the first launch's stretch
    uint32_t slots = 1;
    CallScratch* scratch = nullptr;
    size_t slot_bytes = 0;                                         // bytes of the frame stacks per concurrent launch
    // with_board: every resident block is launched however small the batch — a block without reads of its own waits at the board for subtrees of the heavy reads
    int init(uint32_t depth, uint64_t nq, int blocks_per_cu, hipStream_t stream, int nplanes = 3, bool with_board = false, uint32_t want_slots = 1) {
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        uint64_t want = (uint64_t)cus * (uint64_t)std::max(1, std::min(8, blocks_per_cu));
        slots = std::max(1u, std::min(kDfsSlots, want_slots));
        if (slots > 1) want = std::max<uint64_t>(1, want / slots);
        grid = (unsigned)std::max<uint64_t>(1, with_board ? want : std::min<uint64_t>(want, (nq + 255) / 256));
        view.nlanes = (uint64_t)grid * 256; view.depth = depth;
        uint64_t words = view.nlanes * ((uint64_t)depth + 1) * slots;
        const size_t need = words * 8 * (size_t)nplanes;
        CallScratch* sc = nullptr;
        int rc = call_scratch(&sc); if (rc) return rc;
        if (need <= kFrameCache) {
            if (sc->frames_bytes < need) {
                if (sc->frames) { (void)hipFree(sc->frames); sc->frames = nullptr; sc->frames_bytes = 0; }
                FM_HIP(hipMalloc(&sc->frames, need));
                sc->frames_bytes = need;
            }
            planes = (uint64_t*)sc->frames;
        } else {
            FM_HIP(hipMalloc((void**)&planes, need));
            own_planes = true;
        }
        view.p0 = planes; view.p1 = planes + words; view.p2 = planes + 2 * words; view.p3 = nplanes > 3 ? planes + 3 * words : nullptr;
        slot_bytes = (need / slots) & ~(size_t)15;
        scratch = sc;
        ctr = (Counters*)sc->dfs_ctr;
        FM_HIP(hipMemsetAsync(ctr, 0, 256, stream));                         // next: the query hand-out counter of the scheme kernels (+ a debug area)
        if (with_board) {
            if (sc->board_slots < slots) {
                if (sc->board) { (void)hipFree(sc->board); sc->board = nullptr; sc->board_slots = 0; }
                FM_HIP(hipMalloc(&sc->board, sizeof(WorkBoard) * slots));
                sc->board_slots = slots;
            }
            board = (WorkBoard*)sc->board;
        }
        if (slots > 1) {
            for (uint32_t k = 0; k < slots; ++k) if (!sc->dfs_streams[k]) FM_HIP(hipStreamCreateWithFlags(&sc->dfs_streams[k], hipStreamNonBlocking));
            for (uint32_t k = 0; k <= slots; ++k) if (!sc->dfs_events[k]) FM_HIP(hipEventCreateWithFlags(&sc->dfs_events[k], hipEventDisableTiming));
        }
        return 0;
    }
    // what the k-th concurrent launch works with
    hipStream_t stream_of(uint32_t k, hipStream_t caller) const { return slots > 1 ? scratch->dfs_streams[k] : caller; }
    unsigned long long* next_of(uint32_t k) const { return slots > 1 ? reinterpret_cast<unsigned long long*>(ctr) + 24 + k : &ctr->next; }      // (the hand-out counters of side-by-side launches: words 24..31 of the counter area)
    WorkBoard* board_of(uint32_t k) const { return board ? board + (slots > 1 ? k : 0) : nullptr; }
    StackView view_of(uint32_t k) const {                          // (the fast kernels address their frames from p0 alone: a stretch of the whole area per launch)
        StackView v = view;
        v.p0 = reinterpret_cast<uint64_t*>(reinterpret_cast<uint8_t*>(planes) + (size_t)k * slot_bytes);
        return v;
    }
    // side-by-side launches start after everything the caller's stream holds, and the caller's stream goes on after all of them
    int fork(hipStream_t caller) {
        if (slots <= 1) return 0;
        FM_HIP(hipEventRecord(scratch->dfs_events[slots], caller));
        for (uint32_t k = 0; k < slots; ++k) FM_HIP(hipStreamWaitEvent(scratch->dfs_streams[k], scratch->dfs_events[slots], 0));
        forked = true;
        return 0;
    }
    int join(hipStream_t caller) {
        if (slots <= 1) return 0;
        for (uint32_t k = 0; k < slots; ++k) { FM_HIP(hipEventRecord(scratch->dfs_events[k], scratch->dfs_streams[k])); FM_HIP(hipStreamWaitEvent(caller, scratch->dfs_events[k], 0)); }
        forked = false;
        return 0;
    }
    int reset_board(hipStream_t stream, uint32_t k = 0) {          // before every launch that uses it (k: which of the side-by-side launches)
        if (!board) return 0;
        WorkBoard* const board = board_of(k);
        FM_HIP(hipMemsetAsync(board, 0, kBoardResetBytes, stream));
        uint32_t cfg[3] = {kBoardHeavy, kBoardPeriod, kBoardWaiters};
        if (const char* e = dev_env("FMGPU_DEV_BOARD_HEAVY")) cfg[0] = (uint32_t)atoi(e);
        if (const char* e = dev_env("FMGPU_DEV_BOARD_PERIOD")) { cfg[1] = 1; while (cfg[1] * 2 <= (uint32_t)std::max(1, atoi(e))) cfg[1] *= 2; }      // (a power of two: the kernels test pass & (period - 1))
        if (const char* e = dev_env("FMGPU_DEV_BOARD_WAITERS")) cfg[2] = (uint32_t)std::max(1, atoi(e));
        FM_HIP(hipMemsetD32Async((hipDeviceptr_t)&board->heavy, (int)cfg[0], 1, stream));      // (fills, not copies: no host buffer has to outlive the call)
        FM_HIP(hipMemsetD32Async((hipDeviceptr_t)&board->period, (int)cfg[1], 1, stream));
        FM_HIP(hipMemsetD32Async((hipDeviceptr_t)&board->waiters, (int)cfg[2], 1, stream));
        return 0;
    }
    // after the launch has been synchronised: a waiting wave that gave up means results may be missing
    int check_board() {
        if (!board) return 0;
        unsigned long long f = 0;
        for (uint32_t k = 0; k < slots; ++k) { unsigned long long fk = 0; FM_HIP(hipMemcpy(&fk, &board_of(k)->failed, 8, hipMemcpyDeviceToHost)); f += fk; }
        if (dev_env("FMGPU_DEV_BOARD_LOG")) {                        // (development build: what went over the board in the last launch)
            unsigned long long v[2] = {0, 0}, t = 0;
            (void)hipMemcpy(&v[0], &board->ht, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&t, &board->tasks, 8, hipMemcpyDeviceToHost);
            unsigned long long d[7] = {0}; uint32_t tries = 0;
            (void)hipMemcpy(d, &board->dev[0], sizeof d, hipMemcpyDeviceToHost); (void)hipMemcpy(&tries, &board->cas_tries, 4, hipMemcpyDeviceToHost);
            const double clk = 2.4e3;                              // (s_memtime counts the shader clock here: ~2.4 GHz — cycles per microsecond)
            fprintf(stderr, "board: %llu batches asked for, %llu published, %llu subtrees, %u compare-and-swaps; wave-time: waiting %.1f %% (%llu waits), giving %.2f %% (%llu, %.1f us each), looking %.2f %% (%llu, %.1f us each)\n",
                    v[0] >> 32, v[0] & 0xffffffffull, t, tries, d[6] ? 100.0 * d[0] / d[6] : 0.0, d[3], d[6] ? 100.0 * d[1] / d[6] : 0.0, d[4], d[4] ? d[1] / (double)d[4] / clk : 0.0,
                    d[6] ? 100.0 * d[2] / d[6] : 0.0, d[5], d[5] ? d[2] / (double)d[5] / clk : 0.0);
        }
        return f ? fail(FMGPU_ERR_HIP, "work sharing between waves: " + std::to_string(f) + " waiting wave(s) gave up") : 0;
    }
    bool forked = false;                                           // side-by-side launches are out and have not been joined
    ~DfsWorkspace() {
        if (forked) for (uint32_t k = 0; k < slots; ++k) (void)hipStreamSynchronize(scratch->dfs_streams[k]);      // (an error path left between fork and join: nothing of this call may outlive it)
        if (planes && own_planes) (void)hipFree(planes);
    }
};

// the hand-out order of a batch with the reads of high-copy repeats in front (see k_heavy_flags): flag_pass(count, flags, counter) launches the
// flag kernel over the first `count` reads.  A 64 k sample decides whether the pass over the whole batch is worth it; *out_order stays null if not.
// The buffers live in the calling thread's scratch (the order is valid until the thread's next search call).
template <class FlagPass>
static int heavy_first_order(uint64_t nq, hipStream_t stream, FlagPass&& flag_pass, uint32_t** out_order) {
    *out_order = nullptr;
    CallScratch* sc = nullptr;
    int rc = call_scratch(&sc); if (rc) return rc;
    size_t tb = 0;
    (void)hipcub::DevicePartition::Flagged(nullptr, tb, hipcub::CountingInputIterator<uint32_t>(0u), (const uint8_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)nq, stream);
    const size_t off_flags = (nq * 4 + 255) / 256 * 256, off_cnt = off_flags + (nq + 255) / 256 * 256, off_tmp = off_cnt + 256;
    const size_t need = off_tmp + std::max<size_t>(tb, 16);
    if (sc->order_bytes < need) {
        if (sc->order) { (void)hipFree(sc->order); sc->order = nullptr; sc->order_bytes = 0; }
        FM_HIP(hipMalloc(&sc->order, need));
        sc->order_bytes = need;
    }
    uint8_t* base = (uint8_t*)sc->order;
    uint32_t* order = (uint32_t*)base; uint8_t* flags = base + off_flags; uint32_t* cnt = (uint32_t*)(base + off_cnt);
    const uint64_t ns = std::min<uint64_t>(nq, 1u << 16);
    FM_HIP(hipMemsetAsync(cnt, 0, 8, stream));
    flag_pass(ns, flags, cnt);
    FM_LAUNCHED("k_heavy_flags");
    uint32_t heavy = 0;
    FM_HIP(hipMemcpyAsync(&heavy, cnt, 4, hipMemcpyDeviceToHost, stream));
    FM_HIP(hipStreamSynchronize(stream));
    if ((uint64_t)heavy * 2000u < ns) return 0;                     // < 0.05 % of the sample: nothing worth moving
    if (nq > ns) { flag_pass(nq, flags, cnt); FM_LAUNCHED("k_heavy_flags"); }
    FM_HIP(hipcub::DevicePartition::Flagged(base + off_tmp, tb, hipcub::CountingInputIterator<uint32_t>(0u), flags, order, cnt + 1, (int)nq, stream));
    *out_order = order;
    return 0;
}

namespace api {
#include "fmgpu_api_decl.h"
int fmgpu_hits_pack16(const fmgpu_hit* hits, uint64_t count, uint64_t* out, void* stream);
int fmgpu_hits_pack24(const fmgpu_hit* hits, uint64_t count, uint64_t* out, void* stream);
int fmgpu_hits_sort(fmgpu_hit* hits, uint64_t count, void* stream);

static size_t lean_lds_bytes(uint32_t m, size_t step_words) {
    return (size_t)((m + 15) / 16) * 1024 + 8192 + step_words * 4 + 16 + (size_t)4 * kRingWords * kRingCap * 4;      // staged reads | top and bottom frame slots | steps | ring fill | rings
}
static void launch_lean(const Index* x, const uint32_t* d_steps, uint32_t S, uint32_t m, size_t step_words, dim3 g, const uint8_t* dq, const uint64_t* doff, uint64_t count,
                        fmgpu_hit* dout, uint64_t capacity, const DfsWorkspace& ws, const uint32_t* qm, hipStream_t stream, uint32_t lut_ok = 0, uint32_t slot = 0) {
    const idx_t n = (idx_t)x->bwt.n;
    const bool dense = x->bwt.dense && x->rev.dense && !(kernel_flags() & (1 << 29));      // (bit 29 of FMGPU_DEV_FLAGS: read Format A although Format D exists)
    LeanArgs la{x->bwt.va.blk, x->rev.va.blk, d_steps, S, m, (idx_t)(x->hC[1] + x->hC[2] + x->hC[3] + x->hC[4]), x->bwt.va.super, x->rev.va.super,
                kWide ? (uint32_t)((x->bwt.n >> kSuperShift) + 1) : 0u,
                (const uint4*)x->bwt.dense, (const uint4*)x->rev.dense, x->bwt.dense_ex, x->rev.dense_ex, x->bwt.dense_nex, x->rev.dense_nex, nullptr, 0u, 0u};
    if (!kWide && x->lut && x->lut_len >= 1 && x->lut_len <= 16 && lut_ok && !(kernel_flags() & FMGPU_SEL_NO_PREFIX_TABLE)) { la.lut = x->lut; la.lutL = x->lut_len; la.lut_ok = lut_ok; }
    const size_t lds = lean_lds_bytes(m, step_words);
    uint32_t waste = kLeanRefillWaste, heavy = kLeanShareHeavy; [[maybe_unused]] int steps = kLeanSteps;
    if (const char* ev = dev_env("FMGPU_DEV_LEAN_WASTE")) waste = (uint32_t)std::max(1, atoi(ev));
    if (const char* ev = dev_env("FMGPU_DEV_LEAN_HEAVY")) heavy = (uint32_t)std::max(0, atoi(ev));
    if (const char* ev = dev_env("FMGPU_DEV_LEAN_STEPS")) steps = atoi(ev);
    WorkBoard* const board = ws.board_of(slot);
    auto launch = [&](auto kern) { kern<<<g, dim3(256), lds, stream>>>(la, dq, doff, count, n, dout, capacity, ws.ctr, reinterpret_cast<ulonglong2*>(ws.view_of(slot).p0), ws.view.nlanes,
                                                                    (m + 15) / 16, qm, waste, heavy, board, ws.next_of(slot)); };
#ifdef FMGPU_DEV
    if (steps == 1) launch(k_scheme_lean<kLeanWaves, 1, false>); else if (steps == 2) launch(k_scheme_lean<kLeanWaves, 2, false>); else if (steps == 8) launch(k_scheme_lean<kLeanWaves, 8, false>); else
#endif
    if constexpr (!kWide) {
        if (la.lut) { if (dense) launch(k_scheme_lean<kLeanWaves, kLeanSteps, true, true>); else launch(k_scheme_lean<kLeanWaves, kLeanSteps, false, true>); return; }
        if (dense) { if (board) launch(k_scheme_lean<kLeanWaves, kLeanSteps, true, false, true>); else launch(k_scheme_lean<kLeanWaves, kLeanSteps, true>); return; }
    }
    if (board) launch(k_scheme_lean<kLeanWaves, kLeanSteps, false, false, true>); else launch(k_scheme_lean<kLeanWaves, kLeanSteps, false>);
}

static int run_dfs(Index* x, bool scheme_mode, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, const fmgpu_scheme* scheme,
                   uint64_t max_hits, uint32_t K, fmgpu_hit* out, uint64_t capacity, uint64_t* out_count, fmgpu_stats* stats, hipStream_t stream) {
    if (stats) *stats = fmgpu_stats{};
    if (out_count) *out_count = 0;
    if (nq == 0) return 0;
    if (!qbuf || !qoff || (!out && capacity) || !out_count) return fail(FMGPU_ERR_INVALID, "qbuf / qoff / out / out_count is null");
    SchemeDev sd{};
    bool edit = false;
    uint32_t max_u = 0;
    if (scheme_mode) {
        if (!x->bidirectional) return fail(FMGPU_ERR_INVALID, "search_ng26 needs a BiFMIndex (bwt_rev)");
        if (!scheme || !scheme->pi || !scheme->l || !scheme->u) return fail(FMGPU_ERR_INVALID, "scheme is null");
        if (scheme->n_searches < 0 || scheme->n_searches > kMaxSearches || scheme->n_parts < 1 || scheme->n_parts > kMaxParts)
            return fail(FMGPU_ERR_UNSUPPORTED, "scheme larger than 16 searches x 16 parts");
        if (max_hits == 0 || scheme->n_searches == 0) return 0;                       // SearchNg26.h:408-409
        sd.S = scheme->n_searches; sd.P = scheme->n_parts; sd.uniform = scheme->partition ? 0 : 1;
        edit = scheme->edit != 0;
        sd.dev_flags = kernel_flags();
        sd.use_key = 0; sd.sharing = 0;                              // set below for the general Hamming kernel
        for (int s = 0; s < sd.S; ++s) {
            uint32_t seen = 0;
            for (int p = 0; p < sd.P; ++p) {
                uint64_t pi = scheme->pi[s * sd.P + p], l = scheme->l[s * sd.P + p], u = scheme->u[s * sd.P + p];
                if (pi >= (uint64_t)sd.P || l > 255 || u > 254) return fail(FMGPU_ERR_INVALID, "scheme entry out of range");
                seen |= 1u << pi; max_u = std::max<uint32_t>(max_u, (uint32_t)u);
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
    uint32_t maxlen = 0, minlen = 0;
    const bool have_shape = is_device_pointer(qoff);             // offsets in HBM: total and length range come back in one copy
    if (have_shape) { if ((rc = query_shape((const uint64_t*)soff.dev, nq, stream, &maxlen, &minlen, &total))) return rc; }
    else total = qoff[nq];
    if ((rc = sbuf.in(qbuf, total, stream))) return rc;
    if ((rc = sout.out(out, capacity * sizeof(fmgpu_hit), stream))) return rc;
    if (!have_shape && (rc = query_len_range((const uint64_t*)soff.dev, nq, stream, &maxlen, &minlen))) return rc;
    if (maxlen > 0xfffeu) return fail(FMGPU_ERR_UNSUPPORTED, "queries longer than 65534 symbols");
    static std::mutex occ_mu; static std::map<std::tuple<int, int, int, size_t>, int> occ_cache;
    int bpc = 8;
    // LDS budget of a 256-lane block of the DFS kernels: 64 KB in all — the staged queries (1 KB per word and block) beside the kernels' own
    // tables and hit buffers (<= 17 KB static in the general kernels; per-step tables + hit buffers in the table-driven ones, reserved below)
    const size_t stage_words = x->bwt.sigma <= 15 ? (maxlen + 7) / 8 : (maxlen + 3) / 4;
    const size_t tables_lds = scheme_mode ? (size_t)3 * (size_t)std::max(sd.S, 1) * ((size_t)maxlen + 1) * 4 + (size_t)kWaveHitWords * 4 : 0;
    const size_t stage_budget = (size_t)64 * 1024 - std::max<size_t>((kWide ? 24 : 17) * 1024, std::min<size_t>(tables_lds, 47 * 1024));
    const size_t occ_lds = stage_words * 1024 > stage_budget ? 0 : stage_words * 1024;
    const auto occ_key = std::make_tuple(x->bwt.search_family(), x->bwt.sigma, (int)scheme_mode + (edit ? 2 : 0), occ_lds);
    bool occ_known = false;
    { std::lock_guard<std::mutex> g(occ_mu); auto it = occ_cache.find(occ_key); if (it != occ_cache.end()) { bpc = it->second; occ_known = true; } }
    if (!occ_known) {   // residency of the kernel instantiation that will run (queried once: the call is slow)
        auto occ_of = [&](auto kernel) { int nb = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 256, occ_lds) == hipSuccess && nb > 0) bpc = nb; else (void)hipGetLastError(); };
        dispatch_occ(x->bwt, [&](auto occ, auto ms) {
            using O = decltype(occ);
            if (scheme_mode && edit) occ_of(k_scheme_edit<O, decltype(ms)::value>); else
            if (scheme_mode) occ_of(k_scheme<O, decltype(ms)::value>); else occ_of(k_backtracking<O, decltype(ms)::value>);
            return 0;
        });
        std::lock_guard<std::mutex> g(occ_mu); occ_cache[occ_key] = bpc;
    }
    // query staging: 8 (nibbles) or 4 (bytes) symbols per LDS word, 256 lanes per block; above 48 KB the kernels read global memory
    const uint32_t qnib = x->bwt.sigma <= 15 ? 1u : 0u;
    uint32_t qwords = qnib ? (maxlen + 7) / 8 : (maxlen + 3) / 4;
    if ((size_t)qwords * 1024 > stage_budget) qwords = 0;
    const size_t lds_bytes = (size_t)qwords * 1024;
    DfsWorkspace ws;
    EventTimer timer(stream, stats != nullptr);
    const idx_t n = (idx_t)x->bwt.n;
    const dim3 block(256);
    uint32_t* d_qmap = nullptr;                                    // hand-out order: the thread's call scratch (heavy reads first) or qmap_buf (length buckets)
    DBuf qmap_buf, steps_buf;                                      // released on every return path
    uint32_t* d_steps = nullptr;
    [[maybe_unused]] bool fast = false;
    float prepass_ms = 0.f;                                        // the hand-out order pass (flag kernel, sample read-back, partition): reported beside kernel_ms, never inside it
#if FMGPU_WIDE
    // 64-bit rows hold no tables; an equal-length Hamming batch on the plain sigma = 5 blocks still takes k_scheme_lean (16-byte frames of 38-bit rows, reads of
    // <= 255 symbols, path keys, unlimited hits per read) — a genome with its reverse complement (6.2 G rows) is not confined to the general kernel
    std::vector<uint32_t> wide_tab;
    bool lean_wide = false;
    if (scheme_mode && !edit && x->bwt.family == FAM_A && !x->bwt.shadow && x->bwt.sigma == 5 && x->bwt.va.bstride == 64u && minlen == maxlen && maxlen <= 255 && max_hits == ~0ull &&
        sd.S <= 16 && max_u <= 2 && nq <= 0xffffffffull && n >= 2 && n < ((idx_t)1 << 38) && !(sd.dev_flags & (2 | (1 << 24) | (1 << 30)))) {
        uint32_t lut_ok = 0;
        lean_wide = build_step_table(sd, maxlen, 0, 0, wide_tab, lut_ok);
        if (lean_wide) {
            wide_tab.resize(wide_tab.size() / 3);                  // (the stretch words serve the walk tables)
            if (maxlen >= 16 && nq >= (1u << 16) && nq < 0x7fffffffull && opt_on(FMGPU_OPT_HEAVY_FIRST)) {
                uint32_t* order = nullptr;
                const auto pre_t0 = std::chrono::steady_clock::now();
                if ((rc = heavy_first_order(nq, stream, [&](uint64_t count_reads, uint8_t* flags, uint32_t* cnt) {
                        k_heavy_flags_plain<OccA<5>><<<dim3((unsigned)((count_reads + 255) / 256)), 256, 0, stream>>>(OccA<5>{x->bwt.va}, n, (const uint8_t*)sbuf.dev, (const uint64_t*)soff.dev,
                                                                                                                   count_reads, maxlen, heavy_rows(), flags, cnt);
                    }, &order))) return rc;
                if (order) { d_qmap = order; prepass_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - pre_t0).count(); }
            }
            bpc = 3;
        }
    }
#endif
#if !FMGPU_WIDE
    // fast path: equal-length batch on a Format-A BiFMIndex — with LF tables, or (Hamming, sigma <= 5) on the blocks alone
    // 16-symbol walk: 2-bit symbols, queries staged in LDS as nibbles
    const bool have_lf = x->bwt.lf_table && x->rev.lf_table;
    const bool use_wj = !edit && !(sd.dev_flags & 32) && have_lf && x->bwt.walkj && x->rev.walkj && x->bwt.walk_bits == 2 && x->rev.walk_bits == 2 && qnib && qwords;
    const bool fast_ok = scheme_mode && x->bwt.search_family() == FAM_A && (have_lf || x->bwt.sigma == 5) && x->bwt.sigma <= 32 && !(sd.dev_flags & 2);   // (sigma = 5: the fast kernels also run on the plain index)
    const uint32_t lutL = (sd.dev_flags & 4) ? 0 : x->lut_len;
    // one launch of the table-driven kernel per query length: an equal-length batch is one bucket; a ragged batch is sorted by length on the
    // device (the kernel reads its queries through the sorted index) as long as the buckets stay large enough to be worth a launch each
    struct Bucket { uint32_t m; uint64_t first, count; std::vector<uint32_t> tab; uint32_t lut_ok; };
    std::vector<Bucket> buckets;
    if (fast_ok && minlen == maxlen) {
        Bucket b{maxlen, 0, nq, {}, 0};
        fast = build_step_table(sd, maxlen, lutL, use_wj ? 16u : 0u, b.tab, b.lut_ok);
        if (fast) buckets.push_back(std::move(b));
        const bool hf_on = opt_on(FMGPU_OPT_HEAVY_FIRST);
        const bool by_lut = fast && x->lut && lutL >= 15 && lutL <= 16 && (buckets[0].lut_ok & 1u);     // (no step table — m < P, a scheme too large for it: the general kernel below)
        const bool by_blocks = !have_lf && x->bwt.sigma == 5 && maxlen >= 16;     // (the plain-index instantiation)
        if (fast && (by_lut || by_blocks) && nq >= (1u << 16) && nq < 0x7fffffffull && hf_on) {
            // hand the reads of high-copy repeats out first (k_heavy_flags; decided on a sample of the batch — a text without repeats has nothing to
            // reorder, and the pass would cost 5 % of a 7 ms batch)
            LutPositions lp{};
            if (by_lut) for (uint32_t t = 0; t < lutL; ++t) lp.pos[t] = buckets[0].tab[t] & 0xffffu;
            uint32_t* order = nullptr;
            const auto pre_t0 = std::chrono::steady_clock::now();
            if ((rc = heavy_first_order(nq, stream, [&](uint64_t count_reads, uint8_t* flags, uint32_t* cnt) {
                    const dim3 g((unsigned)((count_reads + 255) / 256));
                    if (by_lut) k_heavy_flags<<<g, 256, 0, stream>>>(x->lut, lutL, (uint32_t)x->bwt.sigma - 1u, lp, (const uint8_t*)sbuf.dev, (const uint64_t*)soff.dev, count_reads,
                                                                     (uint32_t)x->bwt.sigma, heavy_rows(), flags, cnt);
                    else k_heavy_flags_plain<OccA<5>><<<g, 256, 0, stream>>>(OccA<5>{x->bwt.va}, n, (const uint8_t*)sbuf.dev, (const uint64_t*)soff.dev, count_reads, maxlen, heavy_rows(), flags, cnt);
                }, &order))) return rc;
            if (order) { d_qmap = order; prepass_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - pre_t0).count(); }
        }
    } else if (fast_ok && nq >= (1u << 16) && nq < 0x7fffffffull && !(sd.dev_flags & 64)) {
        uint32_t *klen = nullptr, *kidx = nullptr, *slen = nullptr, *runs = nullptr;
        void* tmp = nullptr; size_t tmp_bytes = 0, tmp2 = 0;
        const uint32_t max_runs = maxlen - minlen + 1;
        auto drop = [&] { for (void* p : {(void*)klen, (void*)kidx, (void*)slen, (void*)runs, tmp}) if (p) (void)hipFree(p); };
        hipError_t he = hipMalloc((void**)&klen, nq * 4);
        if (he == hipSuccess) he = hipMalloc((void**)&kidx, nq * 4);
        if (he == hipSuccess) he = hipMalloc((void**)&slen, nq * 4);
        if (he == hipSuccess) { if (qmap_buf.alloc(nq * 4) == 0) d_qmap = qmap_buf.as<uint32_t>(); else he = hipErrorOutOfMemory; }
        if (he == hipSuccess) he = hipMalloc((void**)&runs, ((size_t)max_runs * 2 + 1) * 4);
        if (he == hipSuccess) {
            (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, klen, slen, kidx, d_qmap, (int)nq, 0, 16, stream);
            (void)hipcub::DeviceRunLengthEncode::Encode(nullptr, tmp2, slen, runs, runs + max_runs, runs + 2 * max_runs, (int)nq, stream);
            he = hipMalloc(&tmp, std::max(tmp_bytes, tmp2));
        }
        if (he != hipSuccess) { drop(); return hip_fail(he, "hipMalloc(length buckets)"); }
        k_len_pairs<<<dim3((unsigned)((nq + 255) / 256)), 256, 0, stream>>>((const uint64_t*)soff.dev, nq, klen, kidx);
        size_t tb = std::max(tmp_bytes, tmp2);
        he = hipcub::DeviceRadixSort::SortPairs(tmp, tb, klen, slen, kidx, d_qmap, (int)nq, 0, 16, stream);
        tb = std::max(tmp_bytes, tmp2);
        if (he == hipSuccess) he = hipcub::DeviceRunLengthEncode::Encode(tmp, tb, slen, runs, runs + max_runs, runs + 2 * max_runs, (int)nq, stream);
        std::vector<uint32_t> hruns((size_t)max_runs * 2 + 1);
        if (he == hipSuccess) he = hipMemcpyAsync(hruns.data(), runs, hruns.size() * 4, hipMemcpyDeviceToHost, stream);
        if (he == hipSuccess) he = hipStreamSynchronize(stream);
        drop();
        if (he != hipSuccess) return hip_fail(he, "length buckets");
        const uint32_t nruns = hruns[(size_t)max_runs * 2];
        fast = nruns > 0 && nq / nruns >= 4096;                    // buckets of a few thousand queries at least (a launch and a step table each)
        uint64_t first = 0;
        for (uint32_t r = 0; r < nruns && fast; ++r) {
            const uint32_t m = hruns[r]; const uint64_t cnt = hruns[max_runs + r];
            // expand.h:325-327 precondition; an explicit partition must cover the query exactly; other lengths are skipped as in k_scheme
            if (m >= (uint32_t)sd.P && (sd.uniform || m == sd.psum) && n != 0) {
                Bucket b{m, first, cnt, {}, 0};
                fast = build_step_table(sd, m, lutL, use_wj ? 16u : 0u, b.tab, b.lut_ok);
                if (fast) buckets.push_back(std::move(b));
            }
            first += cnt;
        }
        if (!fast) { buckets.clear(); qmap_buf.release(); d_qmap = nullptr; }
    }
    // the workspace is sized for the kernel that will run: the table-driven edit-distance kernel keeps 5 blocks per CU resident where the
    // general one (launch bounds for 4) keeps 4 — a grid of 4 leaves a fifth of its wave slots empty (84.5 -> 77.3 ms per 4 M reads)
    if (fast && edit) {
        size_t max_tab = 0;
        for (const Bucket& b : buckets) max_tab = std::max(max_tab, b.tab.size());
        const size_t lds_fast = lds_bytes + max_tab * 4 + (size_t)kWaveHitWords * 4 + 2 * 256 * 16 + 16;      // (+ the top frames of the edit kernel's stacks)
        const auto key = std::make_tuple(-1, x->bwt.sigma, 3, lds_fast);
        bool known = false;
        { std::lock_guard<std::mutex> g(occ_mu); auto it = occ_cache.find(key); if (it != occ_cache.end()) { bpc = it->second; known = true; } }
        if (!known) {
            int nb = 0;
            hipError_t oe = x->bwt.sigma == 5 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_scheme_fast_edit<5, 5>, 256, lds_fast)
                                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_scheme_fast_edit<0, 32>, 256, lds_fast);
            if (oe == hipSuccess && nb > 0) bpc = nb; else (void)hipGetLastError();
            std::lock_guard<std::mutex> g(occ_mu); occ_cache[key] = bpc;
        }
    }
#endif  // !FMGPU_WIDE
#if !FMGPU_WIDE
    // path keys order the hits of a read whoever finds them (<= 2 substitutions fit the key); with them and no limit on the hits per read the
    // lanes of a wave share the work of large reads
    // (edit distance: <= 3 error edges of 16 bits each, the tree depth — query length + deletions — in 8 bits, the child index in 6)
    const int use_key = !fast || sd.S > 16 ? 0 : (!edit ? (max_u <= 2 ? 1 : 0) : (max_u <= 3 && maxlen + max_u <= 250 && x->bwt.sigma <= 32 ? 1 : 0));
    const int sharing = use_key && max_hits == ~0ull && !(sd.dev_flags & (1 << 24)) ? 1 : 0;
    // the plain index (sigma = 5, no table) with path keys and unlimited hits per read: the lean kernel (k_scheme_lean); bit 30 of FMGPU_DEV_FLAGS keeps
    // k_scheme_fast<PLAIN> (the parity tests run both)
    const bool lean = fast && !edit && !have_lf && x->bwt.sigma == 5 && sharing && nq <= 0xffffffffull && n >= 2 && !(sd.dev_flags & (1 << 30));
    [[maybe_unused]] const uint32_t lean_qwords = (maxlen + 15) / 16;
    size_t lean_lds = 0;
    if (lean) {
        size_t max_tab = 0;
        for (const Bucket& b : buckets) max_tab = std::max(max_tab, b.tab.size() / 3);
        lean_lds = lean_lds_bytes(maxlen, max_tab);
        const auto key = std::make_tuple(-2, x->bwt.sigma, 4, lean_lds);
        bool known = false;
        { std::lock_guard<std::mutex> g(occ_mu); auto it = occ_cache.find(key); if (it != occ_cache.end()) { bpc = it->second; known = true; } }
        if (!known) {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_scheme_lean<kLeanWaves, kLeanSteps, false>, 256, lean_lds) == hipSuccess && nb > 0) bpc = nb; else { (void)hipGetLastError(); bpc = 4; }
            std::lock_guard<std::mutex> g(occ_mu); occ_cache[key] = bpc;
        }
        // measured on the genome-like text, 10 M x 101 bp (kernel ms): 7 / 6 / 5 / 4 resident blocks per CU = 141 / 126 / 120 / 118 (151 bp: 212 / 211 / 210 / 204); with
        // 4 node steps per pass 4 / 3 / 2 blocks = 111 / 106-108 / 134 (151 bp: 189 / 180-182 / 221) — the loop is bound by instruction issue (rocprofv3: ~550
        // wave instructions per iteration, the waves of a SIMD active 100 % of its time at 4 per SIMD; at 7 they spend 48 % of their cycles waiting to
        // issue), so resident waves beyond the 3-4 that keep the issue port busy only add contention
        bpc = std::min(bpc, 3);
    }
    // k_scheme_fast<PLAIN> runs best with 4 resident blocks per CU (measured on the genome-like text, 10 M x 101 bp: 2 / 3 / 4 / 5 blocks =
    // 197 / 159 / 150 / 159 ms): a fifth block adds issue contention and cache pressure, not throughput
    else if (fast && !edit && !have_lf) bpc = std::min(bpc, 4);
#endif
    { const char* ev = dev_env("FMGPU_DEV_BPC"); if (ev && atoi(ev) > 0) bpc = atoi(ev); }   // dev knob: resident blocks per CU the grid is sized for
    // frames: one per node of the current path; deletions lengthen the path beyond the query by at most the largest upper bound
    // sharing between the waves of a launch (the board, fmgpu_search_shared.h): the table-driven edit-distance kernel, wherever its lanes share inside a wave
    bool with_board = false;
#if !FMGPU_WIDE
    with_board = fast && edit && sharing && !(kernel_flags() & FMGPU_SEL_NO_BOARD);
    const bool lean_any = lean;
#else
    const bool lean_any = lean_wide;
#endif
    {   // the lean kernel: its BOARD instantiation for the batches whose end it shortens (kLeanBoardReads)
        uint64_t lim = kLeanBoardReads;
        if (const char* ev = dev_env("FMGPU_DEV_LEAN_BOARD_READS")) lim = (uint64_t)atoll(ev);
        if (lean_any && nq <= lim && !(kernel_flags() & FMGPU_SEL_NO_BOARD)) with_board = true;
    }
    // a ragged batch through the equal-length kernels is one launch per read length: up to four of them run side by side (each on a quarter of the grid), so that the end of one
    // launch — the waves that hold its heaviest reads — overlaps the bulk of the next ones
    uint32_t dfs_slots = 1;
#if !FMGPU_WIDE
    if (fast && buckets.size() > 1 && !(kernel_flags() & FMGPU_SEL_NO_BOARD)) {
        dfs_slots = (uint32_t)std::min<size_t>(kDfsSlots, buckets.size());
        if (const char* ev = dev_env("FMGPU_DEV_DFS_SLOTS")) dfs_slots = (uint32_t)std::max(1, std::min((int)kDfsSlots, atoi(ev)));
    }
#endif
    if ((rc = ws.init(edit ? maxlen + max_u + 2 : maxlen, nq, bpc, stream, edit ? kEditFramePlanes : 3, with_board, dfs_slots))) return rc;
    const dim3 grid(ws.grid);
#if !FMGPU_WIDE
    size_t steps_words = 0;
    if (fast) {
        for (const Bucket& b : buckets) steps_words += b.tab.size();
        if ((rc = steps_buf.alloc(std::max<size_t>(steps_words, 1) * 4))) return rc;
        d_steps = steps_buf.as<uint32_t>();
        size_t at = 0;
        for (const Bucket& b : buckets) { FM_HIP(hipMemcpyAsync(d_steps + at, b.tab.data(), b.tab.size() * 4, hipMemcpyHostToDevice, stream)); at += b.tab.size(); }
    }
#endif
#if FMGPU_WIDE
    if (lean_wide) {
        if ((rc = steps_buf.alloc(wide_tab.size() * 4))) return rc;
        d_steps = steps_buf.as<uint32_t>();
        FM_HIP(hipMemcpyAsync(d_steps, wide_tab.data(), wide_tab.size() * 4, hipMemcpyHostToDevice, stream));
    }
#endif
    timer.start();
#if FMGPU_WIDE
    if (lean_wide) {
        FM_HIP(hipMemsetAsync(&ws.ctr->next, 0, 8, stream));
        if ((rc = ws.reset_board(stream))) return rc;
        launch_lean(x, d_steps, (uint32_t)sd.S, maxlen, wide_tab.size(), grid, (const uint8_t*)sbuf.dev, (const uint64_t*)soff.dev, nq, (fmgpu_hit*)sout.dev, capacity, ws, d_qmap, stream);
    } else
#endif
#if !FMGPU_WIDE
    if (fast) {
        size_t at = 0;
        if ((rc = ws.fork(stream))) return rc;
        uint32_t turn = 0;
        for (const Bucket& b : buckets) {
            const uint32_t slot = turn++ % ws.slots;                // (one launch after the other per slot: its stream orders them)
            hipStream_t lstream = ws.stream_of(slot, stream);
            const StackView lview = ws.view_of(slot);
            unsigned long long* const lnext = ws.next_of(slot);
            FastArgs fa{};
            if (have_lf) { fa.lf_fw = x->bwt.lf_table; fa.lf_rv = x->rev.lf_table; }
            fa.steps = d_steps + at; fa.S = (uint32_t)sd.S; fa.m = b.m;
            at += b.tab.size();
            if (have_lf && !(sd.dev_flags & 8)) { fa.w3_fw = x->bwt.walk3; fa.w3_rv = x->rev.walk3; }
            if (use_wj) { fa.wj_fw = x->bwt.walkj; fa.wj_rv = x->rev.walkj; }
            fa.lut = b.lut_ok ? x->lut : nullptr; fa.lutL = x->lut_len; fa.lut_ok = b.lut_ok;
            for (int k = 1; k < x->bwt.sigma && k <= 8; ++k) fa.C1[k - 1] = (idx_t)x->hC[k];
            const size_t lds_fast = lds_bytes + b.tab.size() * 4 + (size_t)kWaveHitWords * 4 + (edit ? 2 * 256 * 16 + 16 : 0);     // (edit distance: + the top frames of the stacks)
            const dim3 g((unsigned)std::max<uint64_t>(1, with_board ? (uint64_t)ws.grid : std::min<uint64_t>(ws.grid, (b.count + 255) / 256)));
            FM_HIP(hipMemsetAsync(lnext, 0, 8, lstream));          // reads are handed out from 0
            if ((rc = ws.reset_board(lstream, slot))) return rc;
            const uint32_t* qm = d_qmap ? d_qmap + b.first : nullptr;
            if (edit) {
                if (x->bwt.sigma == 5) {
                    if (!have_lf) k_scheme_fast_edit<5, 5, true><<<g, block, lds_fast, lstream>>>(OccA<5>{x->bwt.va}, OccA<5>{x->rev.va}, fa, (const uint8_t*)sbuf.dev, (const uint64_t*)soff.dev,
                                                                             b.count, n, max_hits, (fmgpu_hit*)sout.dev, capacity, ws.ctr, lview, qwords, qnib, sd.dev_flags, qm, sharing, use_key, ws.board_of(slot), lnext);
                    else k_scheme_fast_edit<5, 5><<<g, block, lds_fast, lstream>>>(OccA<5>{x->bwt.va}, OccA<5>{x->rev.va}, fa, (const uint8_t*)sbuf.dev, (const uint64_t*)soff.dev,
                                                                             b.count, n, max_hits, (fmgpu_hit*)sout.dev, capacity, ws.ctr, lview, qwords, qnib, sd.dev_flags, qm, sharing, use_key, ws.board_of(slot), lnext);
                } else
                    k_scheme_fast_edit<0, 32><<<g, block, lds_fast, lstream>>>(OccA<0>{x->bwt.va}, OccA<0>{x->rev.va}, fa, (const uint8_t*)sbuf.dev, (const uint64_t*)soff.dev,
                                                                              b.count, n, max_hits, (fmgpu_hit*)sout.dev, capacity, ws.ctr, lview, qwords, qnib, sd.dev_flags, qm, sharing, use_key, ws.board_of(slot), lnext);
            } else if (lean) {
                launch_lean(x, fa.steps, (uint32_t)sd.S, b.m, b.tab.size() / 3, g, (const uint8_t*)sbuf.dev, (const uint64_t*)soff.dev, b.count, (fmgpu_hit*)sout.dev, capacity, ws, qm, lstream, b.lut_ok, slot);
            } else if (x->bwt.sigma == 5 && !have_lf)
                k_scheme_fast<5, 5, true><<<g, block, lds_fast, lstream>>>(OccA<5>{x->bwt.va}, OccA<5>{x->rev.va}, fa, (const uint8_t*)sbuf.dev, (const uint64_t*)soff.dev,
                                                                          b.count, n, max_hits, (fmgpu_hit*)sout.dev, capacity, ws.ctr, lview, qwords, qnib, sd.dev_flags, qm, sharing, use_key, lnext);
            else if (x->bwt.sigma == 5)
                k_scheme_fast<5, 5, false><<<g, block, lds_fast, lstream>>>(OccA<5>{x->bwt.va}, OccA<5>{x->rev.va}, fa, (const uint8_t*)sbuf.dev, (const uint64_t*)soff.dev,
                                                                    b.count, n, max_hits, (fmgpu_hit*)sout.dev, capacity, ws.ctr, lview, qwords, qnib, sd.dev_flags, qm, sharing, use_key, lnext);
            else
                k_scheme_fast<0, 32, false><<<g, block, lds_fast, lstream>>>(OccA<0>{x->bwt.va}, OccA<0>{x->rev.va}, fa, (const uint8_t*)sbuf.dev, (const uint64_t*)soff.dev,
                                                                     b.count, n, max_hits, (fmgpu_hit*)sout.dev, capacity, ws.ctr, lview, qwords, qnib, sd.dev_flags, qm, sharing, use_key, lnext);
        }
        if ((rc = ws.join(stream))) return rc;
    } else
#endif
    if (scheme_mode) {
        // the general kernels: path keys order the hits of a read (Hamming: <= 2 substitutions; edit distance: <= 3 errors), and with them and no limit on the hits per read the
        // lanes that run out of queries at the end of the batch take subtrees from the busy lanes of their wave
        sd.use_key = sd.S > 16 ? 0 : (!edit ? (max_u <= 2 ? 1 : 0) : (max_u <= 3 && maxlen + max_u <= 250 && x->bwt.sigma <= 32 ? 1 : 0));
        sd.sharing = sd.use_key && max_hits == ~0ull && !(sd.dev_flags & (1 << 24)) ? 1 : 0;
        // ... and the reads of high-copy repeats are handed out first here too (16 LF steps per read on whatever layout the index has)
        uint32_t* gen_order = nullptr;
        {
            if (nq >= (1u << 16) && nq < 0x7fffffffull && minlen >= 1 && opt_on(FMGPU_OPT_HEAVY_FIRST)) {
                int orc = 0;
                const auto pre_t0 = std::chrono::steady_clock::now();
                rc = dispatch_occ(x->bwt, [&](auto occ, auto) {
                    orc = heavy_first_order(nq, stream, [&](uint64_t count_reads, uint8_t* flags, uint32_t* cnt) {
                        k_heavy_flags_plain<decltype(occ)><<<dim3((unsigned)((count_reads + 255) / 256)), 256, 0, stream>>>(occ, n, (const uint8_t*)sbuf.dev, (const uint64_t*)soff.dev,
                                                                                                                         count_reads, 0u, heavy_rows(), flags, cnt);
                    }, &gen_order);
                    return 0;
                });
                if (rc || orc) return rc ? rc : orc;
                if (gen_order) prepass_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - pre_t0).count();
                timer.start();                                      // kernel_ms = the search kernel alone, as in the table-driven path
            }
        }
        const DevString& rv = x->rev;
        LfView lfv{nullptr, nullptr, nullptr};
        if (x->bwt.lf_table && rv.lf_table && !(sd.dev_flags & 16)) lfv = LfView{x->bwt.lf_table, rv.lf_table, x->dC};
        rc = dispatch_occ(x->bwt, [&](auto occ, auto ms) {
            using O = decltype(occ);
            O r{};
            if constexpr (std::is_same_v<O, OccA<5>> || std::is_same_v<O, OccA<0>>) r = O{rv.va};
            else if constexpr (std::is_same_v<O, OccM>) r = O{rv.vm};
            else r = O{rv.vr};
            (void)hipMemsetAsync(&ws.ctr->next, 0, 8, stream);         // queries are handed out from 0, one reservation per wave
            if (edit) {
                k_scheme_edit<O, decltype(ms)::value><<<grid, block, lds_bytes, stream>>>(occ, r, sd, (const uint8_t*)sbuf.dev, (const uint64_t*)soff.dev, nq, n,
                                                                                  max_hits, (fmgpu_hit*)sout.dev, capacity, ws.ctr, ws.view, qwords, qnib, lfv, maxlen,
                                                                                  (sd.dev_flags & 4) ? nullptr : x->lut, x->lut_len, gen_order);
                return 0;
            }
            k_scheme<O, decltype(ms)::value><<<grid, block, lds_bytes, stream>>>(occ, r, sd, (const uint8_t*)sbuf.dev, (const uint64_t*)soff.dev, nq, n,
                                                                         max_hits, (fmgpu_hit*)sout.dev, capacity, ws.ctr, ws.view, qwords, qnib, lfv, gen_order);
            return 0;
        });
    } else {
        rc = dispatch_occ(x->bwt, [&](auto occ, auto ms) {
            k_backtracking<decltype(occ), decltype(ms)::value><<<grid, block, lds_bytes, stream>>>(occ, x->bidirectional, (const uint8_t*)sbuf.dev,
                                                                                           (const uint64_t*)soff.dev, nq, n, K, (fmgpu_hit*)sout.dev,
                                                                                           capacity, ws.ctr, ws.view, qwords, qnib);
            return 0;
        });
    }
    timer.stop();
    hipError_t le = hipGetLastError();
    Counters hc{};
    if (le == hipSuccess) le = hipMemcpyAsync(&hc, ws.ctr, sizeof hc, hipMemcpyDeviceToHost, stream);
    if (le == hipSuccess) le = hipStreamSynchronize(stream);
    if (le != hipSuccess) return hip_fail(le, "search kernel");
    if ((rc = ws.check_board())) return rc;
#ifdef FMGPU_DEV_STAMPS
    { unsigned long long dbg[12]; (void)hipMemcpy(dbg, reinterpret_cast<unsigned long long*>(ws.ctr) + 8, sizeof dbg, hipMemcpyDeviceToHost);
      if (dbg[6]) fprintf(stderr, "stamps: waves %llu, wave node-steps %llu; cycles per node-step: top %.0f share %.0f refill %.0f flush+rest %.0f issue %.0f wait %.0f node %.0f tail %.0f\n", dbg[6], dbg[5],
                          (double)dbg[9] / dbg[5], (double)dbg[7] / dbg[5], (double)dbg[8] / dbg[5], (double)dbg[0] / dbg[5], (double)dbg[1] / dbg[5], (double)dbg[2] / dbg[5], (double)dbg[3] / dbg[5], (double)dbg[4] / dbg[5]); }
#endif
    *out_count = hc.hits;
    if (stats) { stats->lf_steps = hc.nodes; stats->hits = hc.hits; stats->kernel_ms = timer.ms(); stats->prepass_ms = prepass_ms; stats->table_bytes = hc.table_bytes; stats->table_accesses = hc.table_accesses; }
    if (stats) { unsigned long long served = 0; (void)hipMemcpy(&served, reinterpret_cast<unsigned long long*>(ws.ctr) + 21, 8, hipMemcpyDeviceToHost); stats->table_steps = served; }   // (k_scheme_lean: nodes that prefix-table entries stood for)
#ifdef FMGPU_DEV
    if (stats) { unsigned long long bad = 0; (void)hipMemcpy(&bad, reinterpret_cast<unsigned long long*>(ws.ctr) + 20, 8, hipMemcpyDeviceToHost); stats->hits |= bad << 48; }     // (dev build: LDS frame slots that disagreed with the stack in HBM, in the top bits of `hits`)
#endif
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
    if (int drc = on_handle_device(x)) return drc;
    return run_dfs(x, true, qbuf, qoff, nq, scheme, max_hits_per_query, 0, out, capacity, out_count, stats, (hipStream_t)stream);
}

int fmgpu_search_backtracking(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, uint64_t max_errors,
                              fmgpu_hit* out, uint64_t capacity, uint64_t* out_count, fmgpu_stats* stats, void* stream) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (int drc = on_handle_device(x)) return drc;
    if (max_errors > 254) return fail(FMGPU_ERR_INVALID, "max_errors > 254");
    return run_dfs(x, false, qbuf, qoff, nq, nullptr, ~0ull, (uint32_t)max_errors, out, capacity, out_count, stats, (hipStream_t)stream);
}

int fmgpu_search_ng21(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, const fmgpu_expanded_scheme* scheme,
                      uint64_t max_hits_per_query, fmgpu_hit* out, uint64_t capacity, uint64_t* out_count, fmgpu_stats* stats, void* stream_) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (int drc = on_handle_device(x)) return drc;
    hipStream_t stream = (hipStream_t)stream_;
    if (stats) *stats = fmgpu_stats{};
    if (out_count) *out_count = 0;
    if (!x->bidirectional) return fail(FMGPU_ERR_INVALID, "search_ng21 needs a BiFMIndex (bwt_rev)");
    if (!scheme) return fail(FMGPU_ERR_INVALID, "scheme is null");
    if (scheme->n_searches < 0 || scheme->n_searches > 4096) return fail(FMGPU_ERR_UNSUPPORTED, "more than 4096 searches");
    if (nq == 0 || scheme->n_searches == 0) return 0;                                  // SearchNg21.h:205
    if (!qbuf || !qoff || (!out && capacity) || !out_count) return fail(FMGPU_ERR_INVALID, "qbuf / qoff / out / out_count is null");
    if (!scheme->pi || !scheme->l || !scheme->u) return fail(FMGPU_ERR_INVALID, "scheme arrays are null");
    const uint64_t M = scheme->length;
    if (M == 0 || M > 0xfffeu) return fail(FMGPU_ERR_INVALID, "expanded scheme length must be in [1, 65534]");
    const uint32_t S = (uint32_t)scheme->n_searches;
    if ((uint64_t)S * M > (1ull << 24)) return fail(FMGPU_ERR_UNSUPPORTED, "expanded scheme with more than 2^24 entries");
    std::vector<uint32_t> tab((size_t)S * M);
    uint32_t max_u = 0;
    for (uint32_t s = 0; s < S; ++s) {
        const uint64_t *pi = scheme->pi + (size_t)s * M, *l = scheme->l + (size_t)s * M, *u = scheme->u + (size_t)s * M;
        uint64_t lo = pi[0], hi = pi[0];
        for (uint64_t k = 0; k < M; ++k) {
            if (pi[k] >= M) return fail(FMGPU_ERR_INVALID, "scheme pi entry out of range");
            if (l[k] > 127 || u[k] > 127) return fail(FMGPU_ERR_UNSUPPORTED, "error bounds above 127");
            if (k) {        // the cursor only grows at its two ends (search_scheme/isValid.h:18-33)
                if (pi[k] == hi + 1) hi = pi[k]; else if (pi[k] + 1 == lo) lo = pi[k]; else return fail(FMGPU_ERR_INVALID, "scheme pi is not contiguous");
            }
            const bool right = k == 0 ? (M < 2 || pi[0] < pi[1]) : pi[k - 1] < pi[k];                    // prepare_reorder, :184-200
            tab[(size_t)s * M + k] = (uint32_t)pi[k] | ((uint32_t)l[k] << 16) | ((uint32_t)u[k] << 23) | ((right ? 1u : 0u) << 30);
            max_u = std::max<uint32_t>(max_u, (uint32_t)u[k]);
        }
        // bit 31 of a search's first word: its first lut_len steps are error-free, rightwards and adjacent, so the prefix table
        // (fmgpu_index_accelerate_search) holds the cursor they lead to
        const uint32_t LL = x->lut ? x->lut_len : 0;
        bool ok = LL > 0 && M > LL && x->bwt.n > 1;
        for (uint64_t k = 0; ok && k < LL; ++k) ok = u[k] == 0 && l[k] == 0 && pi[k] == pi[0] + k;
        if (ok) tab[(size_t)s * M] |= 1u << 31;
    }
    Staged soff, sbuf, sout;
    int rc;
    if ((rc = soff.in(qoff, (nq + 1) * 8, stream))) return rc;
    uint64_t total = 0;
    if (is_device_pointer(qoff)) { FM_HIP(hipMemcpyAsync(&total, qoff + nq, 8, hipMemcpyDeviceToHost, stream)); FM_HIP(hipStreamSynchronize(stream)); }
    else total = qoff[nq];
    if ((rc = sbuf.in(qbuf, total, stream))) return rc;
    if ((rc = sout.out(out, capacity * sizeof(fmgpu_hit), stream))) return rc;
    const uint32_t qnib = x->bwt.sigma <= 15 ? 1u : 0u;
    uint32_t qwords = qnib ? ((uint32_t)M + 7) / 8 : ((uint32_t)M + 3) / 4;
    if ((size_t)qwords * 1024 > 48 * 1024) qwords = 0;             // 64 KB of LDS per block, 14 KB of hit buffers: very long queries stay in global memory
    const uint32_t tab_lds = (size_t)qwords * 1024 + tab.size() * 4 <= 44 * 1024 ? 1u : 0u;   // step table behind the staged queries if both fit
    const size_t lds_bytes = (size_t)qwords * 1024 + (tab_lds ? tab.size() * 4 : 0);
    LfView lfv{nullptr, nullptr, nullptr};
    if (x->bwt.lf_table && x->rev.lf_table) lfv = LfView{x->bwt.lf_table, x->rev.lf_table, x->dC};
    static std::mutex occ_mu; static std::map<std::tuple<int, int, size_t>, int> occ_cache;
    int bpc = 4;
    const auto occ_key = std::make_tuple(x->bwt.search_family(), x->bwt.sigma, lds_bytes);
    bool occ_known = false;
    { std::lock_guard<std::mutex> g(occ_mu); auto it = occ_cache.find(occ_key); if (it != occ_cache.end()) { bpc = it->second; occ_known = true; } }
    if (!occ_known) {
        dispatch_occ(x->bwt, [&](auto occ, auto ms) {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_ng21<decltype(occ), decltype(ms)::value>, 256, lds_bytes) == hipSuccess && nb > 0) bpc = nb;
            else (void)hipGetLastError();
            return 0;
        });
        std::lock_guard<std::mutex> g(occ_mu); occ_cache[occ_key] = bpc;
    }
    // path keys (<= 3 errors, depth and child index within their bit fields) order the hits of a read; with them and no limit on the hits per read the
    // lanes that find the query queue empty take subtrees from the busy lanes of their wave
    const int use_key = S <= 16 && max_u <= 3 && M + max_u <= 250 && x->bwt.sigma <= 32 ? 1 : 0;
    const int sharing = use_key && max_hits_per_query == ~0ull && !(kernel_flags() & (1 << 24)) ? 1 : 0;
    DfsWorkspace ws;
    if ((rc = ws.init((uint32_t)M + max_u + 2, nq, bpc, stream, kEditFramePlanes))) return rc;       // deletions lengthen the path beyond the query by at most the largest upper bound
    DBuf tab_buf;
    if ((rc = tab_buf.alloc(tab.size() * 4))) return rc;
    uint32_t* d_tab = tab_buf.as<uint32_t>();
    hipError_t le = hipMemcpyAsync(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, stream);
    EventTimer timer(stream, stats != nullptr);
    const idx_t n = (idx_t)x->bwt.n;
    const auto pre_t0 = std::chrono::steady_clock::now();
    uint32_t* order = nullptr;                                     // heavy reads first, as in search_ng26
    if (le == hipSuccess && nq >= (1u << 16) && nq < 0x7fffffffull && opt_on(FMGPU_OPT_HEAVY_FIRST)) {
        int orc = 0;
        rc = dispatch_occ(x->bwt, [&](auto occ, auto) {
            orc = heavy_first_order(nq, stream, [&](uint64_t count_reads, uint8_t* flags, uint32_t* cnt) {
                k_heavy_flags_plain<decltype(occ)><<<dim3((unsigned)((count_reads + 255) / 256)), 256, 0, stream>>>(occ, n, (const uint8_t*)sbuf.dev, (const uint64_t*)soff.dev,
                                                                                                                 count_reads, 0u, heavy_rows(), flags, cnt);
            }, &order);
            return 0;
        });
        if (rc || orc) return rc ? rc : orc;
    }
    const float prepass_ms = order ? std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - pre_t0).count() : 0.f;
    timer.start();                                                  // kernel_ms = the search kernel alone, in every path (the hand-out order is prepass_ms)
    if (le == hipSuccess) rc = dispatch_occ(x->bwt, [&](auto occ, auto ms) {
        using O = decltype(occ);
        const DevString& rv = x->rev;
        O r{};
        if constexpr (std::is_same_v<O, OccA<5>> || std::is_same_v<O, OccA<0>>) r = O{rv.va};
        else if constexpr (std::is_same_v<O, OccM>) r = O{rv.vm};
        else r = O{rv.vr};
        k_ng21<O, decltype(ms)::value><<<dim3(ws.grid), dim3(256), lds_bytes, stream>>>(occ, r, d_tab, S, (uint32_t)M, (const uint8_t*)sbuf.dev, (const uint64_t*)soff.dev,
                                                                                   nq, n, max_hits_per_query, (fmgpu_hit*)sout.dev, capacity, ws.ctr, ws.view, qwords, qnib, lfv, tab_lds, x->lut, x->lut_len, use_key, sharing, order);
        return 0;
    });
    timer.stop();
    if (le == hipSuccess) le = hipGetLastError();
    Counters hc{};
    if (le == hipSuccess) le = hipMemcpyAsync(&hc, ws.ctr, sizeof hc, hipMemcpyDeviceToHost, stream);
    if (le == hipSuccess) le = hipStreamSynchronize(stream);
    if (le != hipSuccess) return hip_fail(le, "search_ng21 kernel");
    if (rc) return rc;
    *out_count = hc.hits;
    if (stats) { stats->lf_steps = hc.nodes; stats->hits = hc.hits; stats->kernel_ms = timer.ms(); stats->prepass_ms = prepass_ms; }
    if (hc.hits > capacity) {
        if (sout.writeback) { sout.bytes = capacity * sizeof(fmgpu_hit); (void)sout.finish(); }
        return fail(FMGPU_ERR_CAPACITY, "result buffer holds " + std::to_string(capacity) + " records, " + std::to_string(hc.hits) + " produced");
    }
    if (sout.writeback) sout.bytes = hc.hits * sizeof(fmgpu_hit);
    return sout.finish();
}

#if !FMGPU_WIDE   // (fmgpu_hit holds 64-bit fields: the record helpers are width-independent and live in the 32-bit-row build)
// ---- hit records in the reference's callback order -----------------------------------------------------------------------------------
// order key of a record inside its read: fmgpu_hit::seq, extended by the upper 24 bits of fmgpu_hit::errors (the path key of k_scheme_fast;
// zero for the kernels that number their reports themselves)
__global__ __launch_bounds__(256) void k_hit_keys(const fmgpu_hit* __restrict__ h, uint64_t count, uint64_t* __restrict__ key_q, uint64_t* __restrict__ key_s,
                                                  uint32_t* __restrict__ idx) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count) { key_q[t] = h[t].qidx; key_s[t] = ((uint64_t)(h[t].errors >> 8) << 32) | h[t].seq; idx[t] = (uint32_t)t; }
}
__global__ __launch_bounds__(256) void k_gather_u64(const uint64_t* __restrict__ src, const uint32_t* __restrict__ perm, uint64_t count, uint64_t* __restrict__ dst) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count) dst[t] = src[perm[t]];
}
// records in their final order; first[t] = t where a new read starts, else 0 (for the running maximum that finds every record's read start)
__global__ __launch_bounds__(256) void k_gather_hits(const fmgpu_hit* __restrict__ src, const uint32_t* __restrict__ perm, const uint64_t* __restrict__ sorted_q,
                                                     uint64_t count, fmgpu_hit* __restrict__ dst, uint32_t* __restrict__ first) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    dst[t] = src[perm[t]];
    first[t] = (t > 0 && sorted_q[t] != sorted_q[t - 1]) ? (uint32_t)t : 0u;
}
// seq = position of the record within its read (the dense callback index), errors = the error count alone
__global__ __launch_bounds__(256) void k_renumber_hits(fmgpu_hit* __restrict__ h, const uint32_t* __restrict__ start, uint64_t count) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    h[t].seq = (uint32_t)t - start[t];
    h[t].errors &= 0xffu;
}
struct MaxU32 { __host__ __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; } };

// 16-byte transport form of a hit: word 0 = qidx:32 | lb:32, word 1 = len:32 | errors:8 | seq:24; a record that does not fit raises *bad
__global__ __launch_bounds__(256) void k_hits_pack16(const fmgpu_hit* __restrict__ h, uint64_t count, ulonglong2* __restrict__ out, unsigned int* __restrict__ bad) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const fmgpu_hit r = h[t];
    if ((r.qidx | r.lb | r.len) >> 32 || r.errors > 0xffu || r.seq > 0xffffffu) atomicOr(bad, 1u);
    out[t] = make_ulonglong2((r.qidx & 0xffffffffull) | (r.lb << 32),
                             (r.len & 0xffffffffull) | ((uint64_t)(r.errors & 0xffu) << 32) | ((uint64_t)(r.seq & 0xffffffu) << 40));
}
// 24-byte transport form: word 0 = qidx:32 | lb:32, word 1 = len:32 | errors:32 (with the upper key bits), word 2 = lb_rev:32 | seq:32 — everything
// of a record whose rows and read number are below 2^32, order key included
__global__ __launch_bounds__(256) void k_hits_pack24(const fmgpu_hit* __restrict__ h, uint64_t count, uint64_t* __restrict__ out, unsigned int* __restrict__ bad) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const fmgpu_hit r = h[t];
    if ((r.qidx | r.lb | r.len | r.lb_rev) >> 32) atomicOr(bad, 1u);
    out[3 * t] = (r.qidx & 0xffffffffull) | (r.lb << 32);
    out[3 * t + 1] = (r.len & 0xffffffffull) | ((uint64_t)r.errors << 32);
    out[3 * t + 2] = (r.lb_rev & 0xffffffffull) | ((uint64_t)r.seq << 32);
}

static int hits_pack(const fmgpu_hit* hits, uint64_t count, uint64_t* out, int words, void* stream_) {
    if (count == 0) return 0;
    if (!hits || !out) return fail(FMGPU_ERR_INVALID, "hits / out is null");
    hipStream_t stream = (hipStream_t)stream_;
    Staged sh, so;
    int rc = sh.in(hits, count * sizeof(fmgpu_hit), stream); if (rc) return rc;
    if ((rc = so.out(out, count * 8 * (size_t)words, stream))) return rc;
    CallScratch* sc = nullptr;
    if ((rc = call_scratch(&sc))) return rc;
    unsigned int* bad = reinterpret_cast<unsigned int*>(sc->len2);      // (a word of the thread's scratch; the read-back below orders this call's use of it)
    FM_HIP(hipMemsetAsync(bad, 0, 4, stream));
    FM_GRID(grid, count);
    if (words == 2) k_hits_pack16<<<grid, dim3(256), 0, stream>>>((const fmgpu_hit*)sh.dev, count, (ulonglong2*)so.dev, bad);
    else k_hits_pack24<<<grid, dim3(256), 0, stream>>>((const fmgpu_hit*)sh.dev, count, (uint64_t*)so.dev, bad);
    FM_LAUNCHED("k_hits_pack");
    unsigned int* hbad = reinterpret_cast<unsigned int*>(sc->pinned);
    FM_HIP(hipMemcpyAsync(hbad, bad, 4, hipMemcpyDeviceToHost, stream));
    FM_HIP(hipStreamSynchronize(stream));
    if (*hbad) return fail(FMGPU_ERR_UNSUPPORTED, words == 2 ? "a hit record does not fit the 16-byte transport form (qidx, lb, len < 2^32, errors < 256, seq < 2^24: records in callback order, fmgpu_hits_sort)"
                                                             : "a hit record does not fit the 24-byte transport form (qidx, lb, lb_rev, len < 2^32)");
    return so.finish();
}
int fmgpu_hits_pack16(const fmgpu_hit* hits, uint64_t count, uint64_t* out, void* stream) { return hits_pack(hits, count, out, 2, stream); }
int fmgpu_hits_pack24(const fmgpu_hit* hits, uint64_t count, uint64_t* out, void* stream) { return hits_pack(hits, count, out, 3, stream); }

int fmgpu_hits_sort(fmgpu_hit* hits, uint64_t count, void* stream_) {
    if (count == 0) return 0;
    if (!hits) return fail(FMGPU_ERR_INVALID, "hits is null");
    if (count >= 0x7fffffffull) return fail(FMGPU_ERR_UNSUPPORTED, "more than 2^31 - 1 records");
    hipStream_t stream = (hipStream_t)stream_;
    Staged sh;
    int rc = sh.out(hits, count * sizeof(fmgpu_hit), stream); if (rc) return rc;            // in and out: host records are copied in first
    if (sh.writeback) FM_HIP(hipMemcpyAsync(sh.dev, hits, count * sizeof(fmgpu_hit), hipMemcpyHostToDevice, stream));
    // stable LSD order: by the order key first, then by qidx; then the keys are replaced by the dense callback index
    DBuf kq, kq2, ks, ks2, ix, ix2, tmp_h, first, tmp;
    if ((rc = kq.alloc(count * 8)) || (rc = kq2.alloc(count * 8)) || (rc = ks.alloc(count * 8)) || (rc = ks2.alloc(count * 8)) || (rc = ix.alloc(count * 4)) ||
        (rc = ix2.alloc(count * 4)) || (rc = tmp_h.alloc(count * sizeof(fmgpu_hit))) || (rc = first.alloc(count * 4))) return rc;
    size_t b1 = 0, b2 = 0, b3 = 0;
    FM_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, b1, ks.as<uint64_t>(), ks2.as<uint64_t>(), ix.as<uint32_t>(), ix2.as<uint32_t>(), (int)count, 0, 56, stream));
    FM_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, b2, kq.as<uint64_t>(), kq2.as<uint64_t>(), ix2.as<uint32_t>(), ix.as<uint32_t>(), (int)count, 0, 64, stream));
    FM_HIP(hipcub::DeviceScan::InclusiveScan(nullptr, b3, first.as<uint32_t>(), first.as<uint32_t>(), MaxU32{}, (int)count, stream));
    const size_t tb0 = std::max(b1, std::max(b2, b3));
    if ((rc = tmp.alloc(tb0))) return rc;
    FM_GRID(grid, count);
    const dim3 block(256);
    const fmgpu_hit* dh = (const fmgpu_hit*)sh.dev;
    k_hit_keys<<<grid, block, 0, stream>>>(dh, count, kq.as<uint64_t>(), ks.as<uint64_t>(), ix.as<uint32_t>());
    FM_LAUNCHED("k_hit_keys");
    size_t tb = tb0;
    FM_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, ks.as<uint64_t>(), ks2.as<uint64_t>(), ix.as<uint32_t>(), ix2.as<uint32_t>(), (int)count, 0, 56, stream));   // ix2: order by key
    k_gather_u64<<<grid, block, 0, stream>>>(kq.as<uint64_t>(), ix2.as<uint32_t>(), count, kq2.as<uint64_t>());                                                       // qidx in that order
    FM_LAUNCHED("k_gather_u64");
    tb = tb0;
    FM_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, kq2.as<uint64_t>(), kq.as<uint64_t>(), ix2.as<uint32_t>(), ix.as<uint32_t>(), (int)count, 0, 64, stream));   // stable by qidx: ix = final order
    k_gather_hits<<<grid, block, 0, stream>>>(dh, ix.as<uint32_t>(), kq.as<uint64_t>(), count, tmp_h.as<fmgpu_hit>(), first.as<uint32_t>());
    FM_LAUNCHED("k_gather_hits");
    tb = tb0;
    FM_HIP(hipcub::DeviceScan::InclusiveScan(tmp.p, tb, first.as<uint32_t>(), first.as<uint32_t>(), MaxU32{}, (int)count, stream));
    k_renumber_hits<<<grid, block, 0, stream>>>(tmp_h.as<fmgpu_hit>(), first.as<uint32_t>(), count);
    FM_LAUNCHED("k_renumber_hits");
    FM_HIP(hipMemcpyAsync(sh.dev, tmp_h.p, count * sizeof(fmgpu_hit), hipMemcpyDeviceToDevice, stream));
    FM_HIP(hipStreamSynchronize(stream));
    return sh.finish();
}

#endif  // !FMGPU_WIDE

}  // namespace api
}  // namespace FMGPU_NS

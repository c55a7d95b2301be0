// fmgpu_locate.hip — FMIndex::locate / BiFMIndex::locate (fmindex/FMIndex.h:113-124, fmindex/BiFMIndex.h:176-202):
//  k_locate_coop   fused presence bits, blocks fetched by quads through LDS, the rows of a workgroup handed out as lanes fall idle
//  k_locate_fused  fused presence bits, one row per lane
//  k_locate        any layout (presence-bit probe + LF step), or the explicit LF table
//  k_locate_tab    the per-row answer table (fmgpu_index_accelerate_locate)
#include "fmgpu_search_shared.h"

namespace FMGPU_NS {

// ------------------------------------------------------------------ locate
#if !FMGPU_WIDE
// with the per-row answer table (fmgpu_index_accelerate_locate): one 12-byte load per row
__global__ __launch_bounds__(256) void k_locate_tab(const uint32_t* __restrict__ tab, const uint64_t* __restrict__ rows, uint64_t count, idx_t n,
                                                    uint64_t* __restrict__ out_seq, uint64_t* __restrict__ out_pos, uint64_t* __restrict__ out_steps,
                                                    unsigned long long* __restrict__ steps_total) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t steps = 0;
    if (t < count) {
        uint64_t r = rows[t], seq = ~0ull, pos = ~0ull, st = ~0ull;
        if (r < n) {
            const uint32_t* p = tab + 3u * (size_t)r;
            const uint32_t a = p[0], b = p[1], c = p[2];
            if (c != 0xffffffffu) { seq = a; pos = b; st = c; steps = c; }
        }
        out_seq[t] = seq; out_pos[t] = pos; out_steps[t] = st;
    }
    add_counters(steps_total, steps, 0u, 0u);
}

#endif
constexpr uint32_t kLocateStepCap = 1u << 24;   // a valid index reaches a sampled row long before; bounds a corrupt one

// FMIndex::locate on a Format A table with fused presence bits (sigma <= 5; fmgpu_common.h): ONE 64-byte block per step answers "is this row
// sampled", "which symbol precedes it" and "where does that lead" (fmindex/FMIndex.h:113-124 with suffixarray/SparseArray.h:63-70's presence test
// read from the block); the Bitvector2L rank and the two DenseVector reads happen once, at the sampled row.
template <int SIGMA>
__global__ __launch_bounds__(256) void k_locate_fused(OccA<SIGMA> occ, ViewSA sa, const uint64_t* __restrict__ rows, uint64_t count, idx_t n,
                                                      uint64_t* __restrict__ out_seq, uint64_t* __restrict__ out_pos, uint64_t* __restrict__ out_steps,
                                                      unsigned long long* __restrict__ steps_total) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t steps = 0;
    if (t < count) {
        uint64_t r64 = rows[t];
        uint64_t seq = ~0ull, pos = ~0ull, st = ~0ull;
        if (r64 < n) {
            idx_t row = (idx_t)r64;
            bool found = false;
            uint64_t k = 0;
            const uint32_t s = occ.sigma();
            while (steps < kLocateStepCap) {
                const uint4* p = reinterpret_cast<const uint4*>(occ.v.blk + (size_t)(row >> 6) * 64u);
                const uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];             // the whole block: one line
                const uint32_t d[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
                const uint32_t bit = (uint32_t)row & 63u;
                const uint64_t present = (uint64_t)d[1] | ((uint64_t)d[2] << 32);
                if ((present >> bit) & 1ull) {
                    found = true;
                    if constexpr (!kWide) k = (uint64_t)d[15] + popc64(present & lowmask(bit));      // sampled rows before this one: the block's own count + its presence bits
                    break;
                }
                idx_t next = row + occ.v.ksum;                                   // the delimiter's LF unless a symbol >= 1 claims the row
                bool claimed = false;
#pragma unroll
                for (uint32_t c = 1; c < (uint32_t)(SIGMA > 0 ? SIGMA : 5); ++c) {
                    if (c < s) {
                        const uint64_t bits = (uint64_t)d[3 * c + 1] | ((uint64_t)d[3 * c + 2] << 32);
                        idx_t lfc = d[3 * c] + popc64(bits & lowmask(bit));
                        if constexpr (kWide) lfc += occ.v.super[(size_t)(row >> kSuperShift) * s + c];
                        if ((bits >> bit) & 1ull) { claimed = true; next = lfc; }
                        else if (!claimed) next -= lfc;
                    }
                }
                row = next;
                ++steps;
            }
            if (found) {
                if constexpr (kWide) k = sa_rank(sa, row);
                seq = dense_access(sa.f0, sa.bits0, sa.div0, k);
                pos = dense_access(sa.f1, sa.bits1, sa.div1, k);
                st = steps;
            }
        }
        out_seq[t] = seq; out_pos[t] = pos; out_steps[t] = st;
    }
    add_counters(steps_total, steps, 0u, 0u);
}

// The same walk with the blocks fetched by the four lanes of a QUAD together and the rows of a workgroup handed out as lanes fall idle.  k_locate_fused keeps one
// row per lane: the rows of a wave need 0..15 steps, so half of the lanes idle while the slowest walks, and every lane reads its 64-byte block with four
// 16-byte loads of its own (four address translations and four passes through the texture path per block, on a table of 3-4 GB: fmgpu_common.h / DESIGN 4.3).
// Here a workgroup owns kLocRows rows, staged in LDS; a lane that reaches its sampled row parks the result in the row's LDS slot and takes the next unassigned row
// of the pool (one LDS atomic per wave and refill, slots by ballot + prefix count: the tail in which lanes run dry is that of 2048 rows, not of a wave's 512);
// instruction k of a round has the four lanes of every quad load the four 16-byte pieces of the block of the quad's lane k straight into LDS (one 64-byte request
// and one translation per block); the owner reads its block from there.  The value words (two DenseVector reads per row) are fetched after the walk, by all
// lanes at once.  The loop is wave-uniform; the waves of a workgroup meet only at its end.
constexpr uint32_t kLocRows = 2048;                  // rows per workgroup
constexpr uint32_t kLocRegion = 1024u + 16u;         // bytes per region of a round (64 pieces + padding that spreads the owners' reads over the LDS banks)
constexpr uint32_t kLocSlotWords = kWide ? 4u : 2u;  // a row's LDS slot: the row, later {rank among the sampled rows (32-bit rows) or the sampled row itself, steps}
constexpr uint32_t kLocWaveWords = 4u * (kLocRegion / 4u);
constexpr uint32_t kLocBlockWords = 4u * kLocWaveWords + kLocRows * kLocSlotWords + 4u;
constexpr uint32_t kLocNoSteps = 0xffffffffu;
template <int SIGMA>
__global__ __launch_bounds__(256) void k_locate_coop(OccA<SIGMA> occ, ViewSA sa, const uint64_t* __restrict__ rows, uint64_t count, idx_t n,
                                                     uint64_t* __restrict__ out_seq, uint64_t* __restrict__ out_pos, uint64_t* __restrict__ out_steps,
                                                     unsigned long long* __restrict__ steps_total) {
    extern __shared__ uint32_t s_loc[];                             // 4 waves x 4 regions | kLocRows slots | the pool's hand-out counter
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    lds_word* const wave_lds = (lds_word*)(s_loc + wave * kLocWaveWords);
    uint32_t* const slots = s_loc + 4u * kLocWaveWords;
    lds_word* const s_next = (lds_word*)(slots + kLocRows * kLocSlotWords);
    const lds_word* const own = wave_lds + (lane & 3u) * (kLocRegion / 4u) + (lane >> 2) * 16u;
    const uint64_t base = (uint64_t)blockIdx.x * kLocRows;
    const uint32_t cnt = base < count ? (uint32_t)min((uint64_t)kLocRows, count - base) : 0u;
    for (uint32_t t = threadIdx.x; t < cnt; t += 256u) {
        const uint64_t r = rows[base + t];
        if constexpr (kWide) { slots[4u * t] = (uint32_t)r; slots[4u * t + 1u] = (uint32_t)(r >> 32); slots[4u * t + 2u] = r < n ? 0u : kLocNoSteps; }
        else { slots[2u * t] = (uint32_t)r; slots[2u * t + 1u] = r < n ? 0u : kLocNoSteps; }
    }
    if (threadIdx.x == 0) __hip_atomic_store(s_next, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    const uint32_t s = occ.sigma();
    uint32_t my = 0, steps = 0, total_steps = 0;
    idx_t row = 0;
    bool active = false, dry = cnt == 0u;                           // dry (wave-uniform): the pool has no unassigned row left
    const uint64_t below = (1ull << lane) - 1ull;
    for (;;) {
        const uint64_t idle = __ballot(!active);
        if (idle && !dry) {                                         // the i-th idle lane takes row first + i of the pool
            uint32_t first = 0;
            if (lane == 0) first = __hip_atomic_fetch_add(s_next, (uint32_t)__popcll(idle), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            first = __shfl(first, 0, 64);
            const uint32_t at = first + (uint32_t)__popcll(idle & below);
            if (!active && at < cnt) {
                const uint32_t* e = slots + (size_t)at * kLocSlotWords;
                if (e[kLocSlotWords - (kWide ? 2u : 1u)] != kLocNoSteps) {      // (a row beyond the index keeps its "no answer" mark)
                    if constexpr (kWide) row = (idx_t)e[0] | ((idx_t)e[1] << 32); else row = (idx_t)e[0];
                    my = at; steps = 0; active = true;
                }
            }
            dry = first + (uint32_t)__popcll(idle) >= cnt;
        }
        if (!__ballot(active)) { if (dry) break; continue; }
        const uint32_t blk = active ? (uint32_t)(row >> 6) : 0u;    // (n < 2^38: a block number fits 32 bits; an idle lane rides along with block 0)
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            const uint32_t l = __shfl(blk, (int)((lane & ~3u) | k), 64);
            const uint8_t* g = occ.v.blk + (size_t)l * 64u + (lane & 3u) * 16u;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(wave_lds + k * (kLocRegion / 4u)), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);                         // vmcnt(0): the round's pieces are in LDS
        asm volatile("" ::: "memory");
        if (active) {
            const flat_u32x4 q0 = *reinterpret_cast<const __attribute__((address_space(3))) flat_u32x4*>(own), q1 = *reinterpret_cast<const __attribute__((address_space(3))) flat_u32x4*>(own + 4);
            const flat_u32x4 q2 = *reinterpret_cast<const __attribute__((address_space(3))) flat_u32x4*>(own + 8), q3 = *reinterpret_cast<const __attribute__((address_space(3))) flat_u32x4*>(own + 12);
            const uint32_t d[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
            const uint32_t bit = (uint32_t)row & 63u;
            const uint64_t present = (uint64_t)d[1] | ((uint64_t)d[2] << 32);
            uint32_t* e = slots + (size_t)my * kLocSlotWords;
            if ((present >> bit) & 1ull) {                          // the sampled row: park the answer's coordinates in the slot
                if constexpr (kWide) { e[0] = (uint32_t)row; e[1] = (uint32_t)((uint64_t)row >> 32); e[2] = steps; }
                else { e[0] = d[15] + popc64(present & lowmask(bit)); e[1] = steps; }      // sampled rows before this one: the block's own count + its presence bits
                active = false;
            } else {
                idx_t nxt = row + occ.v.ksum;                       // the delimiter's LF unless a symbol >= 1 claims the row
                bool claimed = false;
#pragma unroll
                for (uint32_t c = 1; c < (uint32_t)(SIGMA > 0 ? SIGMA : 5); ++c) {
                    if (c < s) {
                        const uint64_t bits = (uint64_t)d[3 * c + 1] | ((uint64_t)d[3 * c + 2] << 32);
                        idx_t lfc = d[3 * c] + popc64(bits & lowmask(bit));
                        if constexpr (kWide) lfc += occ.v.super[(size_t)(row >> kSuperShift) * s + c];
                        if ((bits >> bit) & 1ull) { claimed = true; nxt = lfc; }
                        else if (!claimed) nxt -= lfc;
                    }
                }
                row = nxt;
                ++steps; ++total_steps;
                if (steps >= kLocateStepCap) { e[kLocSlotWords - (kWide ? 2u : 1u)] = kLocNoSteps; active = false; }     // (a corrupt index: no answer)
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                         // lgkmcnt(0): the next round overwrites the regions
        asm volatile("" ::: "memory");
    }
    __syncthreads();                                                // every wave has parked its rows
    for (uint32_t t = threadIdx.x; t < cnt; t += 256u) {            // the values of the sampled rows (suffixarray/SparseArray.h:63-70), all lanes at once
        const uint32_t* e = slots + (size_t)t * kLocSlotWords;
        uint64_t seq = ~0ull, pos = ~0ull, st = ~0ull;
        const uint32_t ns = e[kLocSlotWords - (kWide ? 2u : 1u)];
        if (ns != kLocNoSteps) {
            uint64_t k;
            if constexpr (kWide) k = sa_rank(sa, (idx_t)e[0] | ((idx_t)e[1] << 32)); else k = e[0];
            seq = dense_access(sa.f0, sa.bits0, sa.div0, k);
            pos = dense_access(sa.f1, sa.bits1, sa.div1, k);
            st = ns;
        }
        out_seq[base + t] = seq; out_pos[base + t] = pos; out_steps[base + t] = st;
    }
    add_counters(steps_total, total_steps, 0u, 0u);
}

template <class Occ>
__global__ __launch_bounds__(256) void k_locate(Occ occ, const idx_t* __restrict__ lf_table, ViewSA sa, const uint64_t* __restrict__ rows, uint64_t count, idx_t n,
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
                if (lf_table) row = lf_table[row];                      // one load per LF step when the explicit table exists
                else { uint32_t c; row = occ.lf_symbol(row, c); }
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
    add_counters(steps_total, steps, 0u, 0u);
}

namespace api {
#include "fmgpu_api_decl.h"

int fmgpu_locate(fmgpu_index_t h, const uint64_t* rows, uint64_t count, uint64_t* out_seq, uint64_t* out_pos, uint64_t* out_steps,
                 fmgpu_stats* stats, void* stream_) {
    // k_locate runs best with 4 resident blocks per CU (9 M rows of the 3.09 Gbp index: 8 / 5 / 4 / 3 blocks = 4.73 / 4.20 / 3.99 / 4.03 ms — its lanes
    // leave after 0 .. 15 LF steps and more waves only queue up at the memory system): 36 KB of unused dynamic LDS set the residency
    size_t locate_lds = (size_t)36 * 1024;
    { const char* ev = dev_env("FMGPU_DEV_LOCATE_LDS"); if (ev) locate_lds = (size_t)atoi(ev); }
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (int drc = on_handle_device(x)) return drc;
    if (!x->has_sa) return fail(FMGPU_ERR_INVALID, "index was created without an annotated (sampled suffix) array");
    if (stats) *stats = fmgpu_stats{};
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
    if ((rc = step_counters(stats != nullptr, stream, &dsteps))) return rc;
    EventTimer timer(stream, stats != nullptr);
    FM_GRID(grid, count);
    const dim3 block(256);
    const idx_t n = (idx_t)x->bwt.n;
    timer.start();
#if !FMGPU_WIDE
    if (x->loc_tab)
        k_locate_tab<<<grid, block, 0, stream>>>(x->loc_tab, (const uint64_t*)srows.dev, count, n, (uint64_t*)sseq.dev, (uint64_t*)spos.dev, (uint64_t*)sst.dev, dsteps);
    else
#endif
    if (x->bwt.va.fused && x->bwt.search_family() == FAM_A) {      // one line per step: presence bit, symbol and LF from the row's block (also ahead of the explicit LF table: that is two lines per step)
        const bool coop = (uint64_t)x->bwt.n < (1ull << 38) && !(kernel_flags() & (1 << 23));   // (bit 23: one row per lane, k_locate_fused)
        const dim3 cgrid((unsigned)((count + kLocRows - 1u) / kLocRows));
        const size_t coop_lds = (size_t)kLocBlockWords * 4 + (dev_env("FMGPU_DEV_LOCATE_LDS") ? locate_lds : 0);
        if (coop && count / kLocRows < kMaxGridBlocks) {
            if (x->bwt.sigma == 5) k_locate_coop<5><<<cgrid, block, coop_lds, stream>>>(OccA<5>{x->bwt.va}, x->vsa, (const uint64_t*)srows.dev, count, n, (uint64_t*)sseq.dev, (uint64_t*)spos.dev, (uint64_t*)sst.dev, dsteps);
            else k_locate_coop<0><<<cgrid, block, coop_lds, stream>>>(OccA<0>{x->bwt.va}, x->vsa, (const uint64_t*)srows.dev, count, n, (uint64_t*)sseq.dev, (uint64_t*)spos.dev, (uint64_t*)sst.dev, dsteps);
        } else
        if (x->bwt.sigma == 5) k_locate_fused<5><<<grid, block, locate_lds, stream>>>(OccA<5>{x->bwt.va}, x->vsa, (const uint64_t*)srows.dev, count, n, (uint64_t*)sseq.dev, (uint64_t*)spos.dev, (uint64_t*)sst.dev, dsteps);
        else k_locate_fused<0><<<grid, block, locate_lds, stream>>>(OccA<0>{x->bwt.va}, x->vsa, (const uint64_t*)srows.dev, count, n, (uint64_t*)sseq.dev, (uint64_t*)spos.dev, (uint64_t*)sst.dev, dsteps);
    } else
    rc = dispatch_occ(x->bwt, [&](auto occ, auto) {
        k_locate<decltype(occ)><<<grid, block, locate_lds, stream>>>(occ, x->bwt.lf_table, x->vsa, (const uint64_t*)srows.dev, count, n, (uint64_t*)sseq.dev,
                                                           (uint64_t*)spos.dev, (uint64_t*)sst.dev, dsteps);
        return 0;
    });
    timer.stop();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "k_locate launch");
    if (stats) {
        unsigned long long hs[kCounterKinds] = {0, 0, 0, 0};
        if ((rc = read_step_counters(dsteps, stream, hs))) return rc;
        stats->lf_steps = hs[0]; stats->hits = count; stats->kernel_ms = timer.ms();
    }
    rc = sseq.finish(); if (!rc) rc = spos.finish(); if (!rc) rc = sst.finish();
    if (stats || sseq.owned || spos.owned || sst.owned) (void)hipStreamSynchronize(stream);
    return rc;
}

#if FMGPU_WIDE
int fmgpu_index_accelerate_locate(fmgpu_index_t h, int32_t enable) {
    if (!h) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (!enable) return 0;
    return fail(FMGPU_ERR_UNSUPPORTED, "the locate answer table is not available for indices of 2^32 rows or more (64-bit-row build)");
}
#else
// answer table for locate: every row is located once, the triples are kept (12 bytes per row)
__global__ __launch_bounds__(256) void k_pack_locate(const uint64_t* __restrict__ seq, const uint64_t* __restrict__ pos, const uint64_t* __restrict__ st, uint64_t first,
                                                     uint64_t count, uint32_t* __restrict__ tab) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    uint32_t* p = tab + 3u * (size_t)(first + t);
    const bool ok = st[t] != ~0ull && seq[t] <= 0xfffffffeull && pos[t] <= 0xffffffffull;
    p[0] = ok ? (uint32_t)seq[t] : 0u; p[1] = ok ? (uint32_t)pos[t] : 0u; p[2] = ok ? (uint32_t)st[t] : 0xffffffffu;
}
__global__ __launch_bounds__(256) void k_iota64(uint64_t* __restrict__ out, uint64_t first, uint64_t count) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count) out[t] = first + t;
}

int fmgpu_index_accelerate_locate(fmgpu_index_t h, int32_t enable) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (int drc = on_handle_device(x)) return drc;
    const uint64_t n = x->bwt.n;
    if (x->loc_tab) { (void)hipFree(x->loc_tab); x->loc_tab = nullptr; x->device_bytes -= n * 12; }
    if (!enable || n == 0) return 0;
    if (!x->has_sa) return fail(FMGPU_ERR_INVALID, "index was created without an annotated (sampled suffix) array");
    DBuf tab, staging;
    int rc;
    const uint64_t chunk = 1ull << 26;
    if ((rc = tab.alloc(n * 12 + 16)) || (rc = staging.alloc(chunk * 8 * 4))) return rc;
    uint64_t* buf = staging.as<uint64_t>();                        // rows | seq | pos | steps of one chunk
    for (uint64_t first = 0; first < n; first += chunk) {
        const uint64_t cnt = std::min(chunk, n - first);
        k_iota64<<<dim3((unsigned)((cnt + 255) / 256)), 256>>>(buf, first, cnt);
        FM_LAUNCHED("k_iota64");
        if ((rc = api::fmgpu_locate(h, buf, cnt, buf + chunk, buf + 2 * chunk, buf + 3 * chunk, nullptr, nullptr))) return rc;
        k_pack_locate<<<dim3((unsigned)((cnt + 255) / 256)), 256>>>(buf + chunk, buf + 2 * chunk, buf + 3 * chunk, first, cnt, tab.as<uint32_t>());
        FM_LAUNCHED("k_pack_locate");
    }
    FM_HIP(hipDeviceSynchronize());
    x->loc_tab = (uint32_t*)tab.take();
    x->device_bytes += n * 12;
    return 0;
}
#endif  // FMGPU_WIDE

}  // namespace api
}  // namespace FMGPU_NS

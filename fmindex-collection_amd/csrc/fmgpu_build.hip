// fmgpu_build.hip — index construction on the GPU.  Compiled once per row width.
//
// Replaces the reference's constructors FMIndex(Sequences, samplingRate, threads) (fmindex/FMIndex.h:58-104) and
// BiFMIndex(Sequences, samplingRate, threads) (fmindex/BiFMIndex.h:107-167), whose heavy lifting is libsais
// (utils.h:97-129; libsais64 from n >= 2^31 on, utils.h:243-247).  Same outputs: text = every sequence followed by a 0 delimiter
// (utils.h:382-411), suffix order of the plain byte string (a proper prefix sorts first), bwt[i] = text[(sa[i]+n-1) % n]
// (utils.h:145-163), sampled entries (seqId, pos) where pos % samplingRate == 0 (FMIndex.h:79-101), bwtRev = BWT of the reversed
// concatenation (BiFMIndex.h:78-92).
//
// Suffix sorting, MI355X style (suffix indices are idx_t: 32 bits below 2^32 - 64 rows, 64 bits above):
//   1. key[i] = the first K symbols of suffix i packed into 64 bits (symbol+1 per field, 0 = past the end, so shorter
//      sorts first); one rocPRIM radix sort of (key, i) pairs orders all suffixes by their K-prefix
//      (K = 21 for DNA): on a random 3.1 Gbp text that already separates all but ~0.1 % of them.
//   2. prefix doubling on the ties only: rows whose group is not a singleton are compacted, keyed by
//      (group start, rank of suffix i+h) and radix-sorted (wide rows: two stable passes, second key first); groups split, h doubles,
//      until no ties are left.  Repeats cost rounds, not correctness: a run of r equal symbols needs log2(r / K) rounds over its rows.
// BWT, the LF-ready 64-byte blocks (Format A) or the multi-ary wavelet tree (Format M, fmgpu_common.h) and the reference-layout sampled
// suffix array are then produced by streaming kernels without leaving HBM.
#include "fmgpu_common.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <memory>
#include <new>

namespace FMGPU_NS {

using cnt_t = std::conditional_t<kWide, uint64_t, uint32_t>;      // running occurrence / sample counts
__device__ __forceinline__ void add_cnt(cnt_t* p, uint32_t v) {
    if constexpr (kWide) atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v);
    else atomicAdd(reinterpret_cast<unsigned int*>(p), v);
}

// ------------------------------------------------------------------ text assembly
__global__ __launch_bounds__(256) void k_assemble_text(const uint8_t* __restrict__ seqs, const uint64_t* __restrict__ seq_off, uint64_t nseq,
                                                       uint8_t* __restrict__ text, uint64_t n) {
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x) {
        // sequence s occupies text [seq_off[s]-seq_off[0]+s, seq_off[s+1]-seq_off[0]+s], the last slot being the delimiter
        uint64_t lo = 0, hi = nseq;           // largest s with start(s) <= p
        const uint64_t base = seq_off[0];
        while (hi - lo > 1) { uint64_t mid = (lo + hi) >> 1; if (seq_off[mid] - base + mid <= p) lo = mid; else hi = mid; }
        uint64_t o = p - (seq_off[lo] - base + lo), len = seq_off[lo + 1] - seq_off[lo];
        text[p] = o < len ? seqs[seq_off[lo] + o] : 0;
    }
}
__global__ __launch_bounds__(256) void k_reverse(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n) {
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x) out[p] = in[n - 1 - p];
}
__global__ __launch_bounds__(256) void k_check_symbols(const uint8_t* __restrict__ t, uint64_t n, uint32_t sigma, unsigned int* bad) {
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x) if (t[p] >= sigma) atomicOr(bad, 1u);
}

// ------------------------------------------------------------------ suffix sorting
__global__ __launch_bounds__(256) void k_pack_keys(const uint8_t* __restrict__ t, uint64_t n, uint32_t K, uint32_t b,
                                                   uint64_t* __restrict__ keys, idx_t* __restrict__ vals) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t k = 0;
        for (uint32_t j = 0; j < K; ++j) k = (k << b) | (i + j < n ? (uint64_t)t[i + j] + 1ull : 0ull);
        keys[i] = k; vals[i] = (idx_t)i;
    }
}
// head[j] = 1 if row j starts a new group; v[j] = head ? (SA index of row j) : 0 (for the running-max scan).
// keys != null: rows are equal iff their 64-bit keys are (the first sort; narrow doubling rounds); keys == null (wide doubling rounds): a row's
// key is the pair (rank[p], rank[p + h] + 1) of its suffix p = pos[j], read from the rank array of the previous round
__global__ __launch_bounds__(256) void k_heads(const uint64_t* __restrict__ keys, const idx_t* __restrict__ pos, const idx_t* __restrict__ rank, uint64_t n, uint64_t h,
                                               uint64_t m, const idx_t* __restrict__ where, idx_t* __restrict__ v, uint8_t* __restrict__ head) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x) {
        bool hd = j == 0;
        if (!hd) {
            if (keys) hd = keys[j] != keys[j - 1];
            else {
                const uint64_t p = pos[j], q = pos[j - 1];
                const uint64_t sp = p + h < n ? (uint64_t)rank[p + h] + 1ull : 0ull, sq = q + h < n ? (uint64_t)rank[q + h] + 1ull : 0ull;
                hd = rank[p] != rank[q] || sp != sq;
            }
        }
        head[j] = hd ? 1 : 0;
        v[j] = hd ? (where ? where[j] : (idx_t)j) : (idx_t)0;
    }
}
struct MaxOp { __host__ __device__ __forceinline__ idx_t operator()(idx_t a, idx_t b) const { return a > b ? a : b; } };

// after the scan: gs[j] = group start (SA index); rank[pos] = gs; flag rows of non-singleton groups
__global__ __launch_bounds__(256) void k_apply_groups(const idx_t* __restrict__ pos, const idx_t* __restrict__ gs, const uint8_t* __restrict__ head,
                                                      uint64_t m, idx_t* __restrict__ rank, uint8_t* __restrict__ active,
                                                      idx_t* __restrict__ sa, const idx_t* __restrict__ where) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x) {
        idx_t p = pos[j];
        rank[p] = gs[j];
        if (sa) sa[where[j]] = p;
        bool single = head[j] && (j + 1 == m || head[j + 1]);
        active[j] = single ? 0 : 1;
    }
}
// keys of a doubling round.  narrow: (rank[p] << 32) | (rank[p + h] + 1) in one word.  wide: `second` alone (pass 1) ...
__global__ __launch_bounds__(256) void k_round_keys(const idx_t* __restrict__ aidx, const idx_t* __restrict__ sa, const idx_t* __restrict__ rank,
                                                    uint64_t m, uint64_t n, uint64_t h, uint64_t* __restrict__ keys, idx_t* __restrict__ pos) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x) {
        idx_t p = sa[aidx[j]];
        uint64_t second = (uint64_t)p + h < n ? (uint64_t)rank[p + h] + 1ull : 0ull;
        keys[j] = kWide ? second : (((uint64_t)rank[p] << 32) | second);
        pos[j] = p;
    }
}
// ... then the first key of the rows in the order pass 1 left them in (pass 2 is stable, so the pair order results)
__global__ __launch_bounds__(256) void k_first_keys(const idx_t* __restrict__ pos, const idx_t* __restrict__ rank, uint64_t m, uint64_t* __restrict__ keys) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x) keys[j] = rank[pos[j]];
}

struct Temp {   // hipcub temporary storage, grown on demand
    DBuf buf;
    int ensure(size_t bytes) { if (bytes > buf.bytes) return buf.alloc(bytes); return 0; }
};

template <class F>
static int cub_call(Temp& tmp, F&& f) {
    size_t bytes = 0;
    FM_HIP(f(nullptr, bytes));
    int rc = tmp.ensure(bytes); if (rc) return rc;
    bytes = tmp.buf.bytes;
    FM_HIP(f(tmp.buf.p, bytes));
    return 0;
}

constexpr uint64_t kStreamGridCap = 1u << 22;      // the streaming kernels here run grid-stride loops
static inline dim3 grid_for(uint64_t n) {
    uint64_t b = (n + 255) / 256;
    return dim3((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(b, kStreamGridCap)));
}

static uint32_t bit_width64(uint64_t v) { uint32_t r = 0; while (v) { ++r; v >>= 1; } return r; }

// sa_out: n idx_t (device).  text: device, n symbols < sigma.
static int build_suffix_array(const uint8_t* text, uint64_t n, uint32_t sigma, idx_t* sa_out, hipStream_t stream) {
    if (n == 0) return 0;
    uint32_t b = 0; while ((1u << b) <= sigma) ++b;         // bits for values 0..sigma
    const uint32_t K = 64 / b;
    const int rank_bits = (int)bit_width64(n);               // a rank + 1 fits these bits
    Temp tmp;
    DBuf rank; int rc;
    if ((rc = rank.alloc(n * sizeof(idx_t)))) return rc;
    DBuf aidx;                                               // active SA indices (ascending)
    uint64_t m = 0;
    {
        DBuf k0, k1, v1, gsb, flags;
        if ((rc = k0.alloc(n * 8)) || (rc = k1.alloc(n * 8)) || (rc = v1.alloc(n * sizeof(idx_t)))) return rc;
        // sa_out doubles as the first value buffer
        k_pack_keys<<<grid_for(n), 256, 0, stream>>>(text, n, K, b, k0.as<uint64_t>(), sa_out);
        FM_LAUNCHED("k_pack_keys");
        hipcub::DoubleBuffer<uint64_t> dk(k0.as<uint64_t>(), k1.as<uint64_t>());
        hipcub::DoubleBuffer<idx_t> dv(sa_out, v1.as<idx_t>());
        rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceRadixSort::SortPairs(t, bytes, dk, dv, (size_t)n, 0, (int)(K * b), stream); });
        if (rc) return rc;
        if (dv.Current() != sa_out) FM_HIP(hipMemcpyAsync(sa_out, dv.Current(), n * sizeof(idx_t), hipMemcpyDeviceToDevice, stream));
        FM_HIP(hipStreamSynchronize(stream));
        v1.release();
        const uint64_t* sorted = dk.Current();
        idx_t* gsv; uint8_t *headv, *actv;
        if (kWide) {                                         // 8-byte group starts do not fit the spare key buffer beside the flags
            if ((rc = gsb.alloc(n * sizeof(idx_t))) || (rc = flags.alloc(2 * n))) return rc;
            gsv = gsb.as<idx_t>(); headv = flags.as<uint8_t>(); actv = headv + n;
        } else {                                             // the spare key buffer is reused: u32 gs + u8 head + u8 active need 6n bytes <= 8n
            uint64_t* spare = dk.Alternate();
            gsv = reinterpret_cast<idx_t*>(spare); headv = reinterpret_cast<uint8_t*>(spare) + n * 4; actv = headv + n;
        }
        k_heads<<<grid_for(n), 256, 0, stream>>>(sorted, nullptr, nullptr, n, 0, n, nullptr, gsv, headv);
        FM_LAUNCHED("k_heads");
        if (kWide) { FM_HIP(hipStreamSynchronize(stream)); k0.release(); k1.release(); }
        rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceScan::InclusiveScan(t, bytes, gsv, gsv, MaxOp{}, (size_t)n, stream); });
        if (rc) return rc;
        k_apply_groups<<<grid_for(n), 256, 0, stream>>>(sa_out, gsv, headv, n, rank.as<idx_t>(), actv, nullptr, nullptr);
        FM_LAUNCHED("k_apply_groups");
        // compact the active SA indices
        DBuf cnt; if ((rc = cnt.alloc(8))) return rc;
        if ((rc = aidx.alloc(n * sizeof(idx_t)))) return rc;
        hipcub::CountingInputIterator<idx_t> iota((idx_t)0);
        rc = cub_call(tmp, [&](void* t, size_t& bytes) {
            return hipcub::DeviceSelect::Flagged(t, bytes, iota, actv, aidx.as<idx_t>(), cnt.as<uint64_t>(), (size_t)n, stream); });
        if (rc) return rc;
        FM_HIP(hipMemcpyAsync(&m, cnt.p, 8, hipMemcpyDeviceToHost, stream));
        FM_HIP(hipStreamSynchronize(stream));
    }
    uint64_t h = K;
    DBuf k0, k1, p0, p1, gs, head, act, aidx2, cnt;
    if (m) {
        if ((rc = k0.alloc(m * 8)) || (rc = k1.alloc(m * 8)) || (rc = p0.alloc(m * sizeof(idx_t))) || (rc = p1.alloc(m * sizeof(idx_t))) || (rc = gs.alloc(m * sizeof(idx_t))) ||
            (rc = head.alloc(m)) || (rc = act.alloc(m)) || (rc = aidx2.alloc(m * sizeof(idx_t))) || (rc = cnt.alloc(8))) return rc;
    }
    int rounds = 0;
    while (m) {
        if (++rounds > 64) return fail(FMGPU_ERR_INVALID, "suffix sorting did not converge");
        k_round_keys<<<grid_for(m), 256, 0, stream>>>(aidx.as<idx_t>(), sa_out, rank.as<idx_t>(), m, n, h, k0.as<uint64_t>(), p0.as<idx_t>());
        FM_LAUNCHED("k_round_keys");
        hipcub::DoubleBuffer<uint64_t> dk(k0.as<uint64_t>(), k1.as<uint64_t>());
        hipcub::DoubleBuffer<idx_t> dv(p0.as<idx_t>(), p1.as<idx_t>());
        if (kWide) {
            rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceRadixSort::SortPairs(t, bytes, dk, dv, (size_t)m, 0, rank_bits + 1, stream); });
            if (rc) return rc;
            k_first_keys<<<grid_for(m), 256, 0, stream>>>(dv.Current(), rank.as<idx_t>(), m, dk.Current());
            FM_LAUNCHED("k_first_keys");
            rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceRadixSort::SortPairs(t, bytes, dk, dv, (size_t)m, 0, rank_bits, stream); });
            if (rc) return rc;
            k_heads<<<grid_for(m), 256, 0, stream>>>(nullptr, dv.Current(), rank.as<idx_t>(), n, h, m, aidx.as<idx_t>(), gs.as<idx_t>(), head.as<uint8_t>());
        } else {
            rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceRadixSort::SortPairs(t, bytes, dk, dv, (size_t)m, 0, 64, stream); });
            if (rc) return rc;
            k_heads<<<grid_for(m), 256, 0, stream>>>(dk.Current(), nullptr, nullptr, n, h, m, aidx.as<idx_t>(), gs.as<idx_t>(), head.as<uint8_t>());
        }
        FM_LAUNCHED("k_heads");
        rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceScan::InclusiveScan(t, bytes, gs.as<idx_t>(), gs.as<idx_t>(), MaxOp{}, (size_t)m, stream); });
        if (rc) return rc;
        k_apply_groups<<<grid_for(m), 256, 0, stream>>>(dv.Current(), gs.as<idx_t>(), head.as<uint8_t>(), m, rank.as<idx_t>(), act.as<uint8_t>(),
                                                         sa_out, aidx.as<idx_t>());
        FM_LAUNCHED("k_apply_groups");
        rc = cub_call(tmp, [&](void* t, size_t& bytes) {
            return hipcub::DeviceSelect::Flagged(t, bytes, aidx.as<idx_t>(), act.as<uint8_t>(), aidx2.as<idx_t>(), cnt.as<uint64_t>(), (size_t)m, stream); });
        if (rc) return rc;
        uint64_t m2 = 0;
        FM_HIP(hipMemcpyAsync(&m2, cnt.p, 8, hipMemcpyDeviceToHost, stream));
        FM_HIP(hipStreamSynchronize(stream));
        std::swap(aidx.p, aidx2.p); std::swap(aidx.bytes, aidx2.bytes);
        m = m2; h *= 2;
    }
    FM_HIP(hipStreamSynchronize(stream));
    return 0;
}

// ------------------------------------------------------------------ BWT and Format A blocks
__global__ __launch_bounds__(256) void k_bwt(const uint8_t* __restrict__ text, const idx_t* __restrict__ sa, uint64_t n, uint8_t* __restrict__ bwt) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t p = sa[i];
        bwt[i] = text[p ? p - 1 : n - 1];
    }
}
__global__ __launch_bounds__(256) void k_histogram(const uint8_t* __restrict__ s, uint64_t n, uint32_t shift, unsigned long long* __restrict__ hist) {
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    // (a block's share stays below 2^32: 2^40 rows over >= 2048 blocks)
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) atomicAdd(&h[s[i] >> shift], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}
// one wave per 64-row block: bits by ballot, per-block counts into cnt[c * nblocks + B]
__global__ __launch_bounds__(256) void k_blocks_bits(const uint8_t* __restrict__ bwt, uint64_t n, uint64_t nblocks, uint32_t sigma, uint32_t bstride,
                                                     uint8_t* __restrict__ blk, cnt_t* __restrict__ cnt) {
    uint32_t lane = threadIdx.x & 63u;
    for (uint64_t B = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; B < nblocks; B += ((uint64_t)gridDim.x * blockDim.x) >> 6) {
        uint64_t row = B * 64 + lane;
        uint32_t s = row < n ? bwt[row] : 0xffffffffu;
        for (uint32_t c0 = 0; c0 < sigma; c0 += 64) {
            uint64_t mine = 0;
            for (uint32_t c = c0; c < sigma && c < c0 + 64; ++c) {
                uint64_t bits = __ballot(s == c);
                if (lane == c - c0) mine = bits;
            }
            uint32_t c = c0 + lane;
            if (c < sigma) {
                uint32_t* o = reinterpret_cast<uint32_t*>(blk + B * bstride + 12ull * c);
                o[1] = (uint32_t)mine; o[2] = (uint32_t)(mine >> 32);
                cnt[(uint64_t)c * nblocks + B] = (cnt_t)__popcll(mine);
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_blocks_counts(const cnt_t* __restrict__ cnt, uint64_t nblocks, uint32_t sigma, uint32_t bstride,
                                                       const idx_t* __restrict__ C, uint8_t* __restrict__ blk, uint64_t* __restrict__ super) {
    const uint64_t total = nblocks * sigma;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t B = t % nblocks; uint32_t c = (uint32_t)(t / nblocks);
        const uint64_t base = kWide ? (uint64_t)cnt[(uint64_t)c * nblocks + super_first_block(B)] + C[c] : 0;
        put_entry_count_a(blk, super, B, c, sigma, bstride, (uint64_t)cnt[t] + C[c], base);
    }
}

// ------------------------------------------------------------------ sampled suffix array (reference layout)
__device__ __forceinline__ void seq_of(const uint64_t* __restrict__ sstart, uint64_t nseq, uint64_t p, uint64_t& s, uint64_t& o) {
    uint64_t lo = 0, hi = nseq;
    while (hi - lo > 1) { uint64_t mid = (lo + hi) >> 1; if (sstart[mid] <= p) lo = mid; else hi = mid; }
    s = lo; o = p - sstart[lo];
}
// one thread per row, launched in slices of at most kSliceRows rows (a wave's ballot is the presence word of its 64 rows).  A launch holds fewer than 2^32 threads: the
// dispatch packet's grid size is a 32-bit number, and a larger launch is cut to its low 32 bits WITHOUT an error (round 4: with 2^36 here, an index of 4.5 x 10^9 rows got
// presence bits for its first 2.1 x 10^8 rows only — found by comparing against the bucketed sorters, whose writers stride)
constexpr uint64_t kSliceRows = 1ull << 31;
__global__ __launch_bounds__(256) void k_sa_bits(const idx_t* __restrict__ sa, uint64_t n, const uint64_t* __restrict__ sstart, uint64_t nseq, uint64_t rate,
                                                 uint64_t* __restrict__ bits, cnt_t* __restrict__ blockcnt, uint64_t first) {
    uint64_t i = first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool has = false;
    if (i < n) { uint64_t s, o; seq_of(sstart, nseq, sa[i], s, o); has = o % rate == 0; }
    uint64_t w = __ballot(has);
    if ((threadIdx.x & 63u) == 0 && i < n) {
        bits[i >> 6] = w;
        add_cnt(&blockcnt[i >> 9], (uint32_t)__popcll(w));
    }
}
__global__ __launch_bounds__(256) void k_sa_levels(const cnt_t* __restrict__ g, uint64_t nl1, uint64_t nl0, uint64_t* __restrict__ l0, uint16_t* __restrict__ l1) {
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nl1; b += (uint64_t)gridDim.x * blockDim.x) {
        cnt_t base = g[(b >> 7) << 7]; l1[b] = (uint16_t)(g[b] - base);
        if (b < nl0) l0[b] = g[b << 7];
    }
}
__global__ __launch_bounds__(256) void k_sa_values(const idx_t* __restrict__ sa, uint64_t n, const uint64_t* __restrict__ sstart, uint64_t nseq, uint64_t rate,
                                                   const uint64_t* __restrict__ bits, const cnt_t* __restrict__ g,
                                                   unsigned long long* __restrict__ f0, unsigned long long* __restrict__ f1,
                                                   uint32_t w0, uint32_t w1, uint64_t d0, uint64_t d1) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (!((bits[i >> 6] >> (i & 63u)) & 1ull)) continue;
        uint64_t s, o; seq_of(sstart, nseq, sa[i], s, o);
        uint64_t r = g[i >> 9];
        const uint64_t* w = bits + (i >> 9) * 8;
        uint32_t bit = (uint32_t)(i & 511u);
        for (uint32_t k = 0; k < (bit >> 6); ++k) r += (uint64_t)__popcll(w[k]);
        if (bit & 63u) r += (uint64_t)__popcll(w[bit >> 6] & ((1ull << (bit & 63u)) - 1ull));
        auto put = [](unsigned long long* data, uint32_t width, uint64_t idx, uint64_t v) {   // DenseVector::push_back layout, DenseVector.h:124-144
            uint64_t begin = idx * width; uint32_t off = (uint32_t)(begin & 63u);
            atomicOr(&data[begin >> 6], (unsigned long long)(v << off));
            if (off + width > 64) atomicOr(&data[(begin >> 6) + 1], (unsigned long long)(v >> (64u - off)));
        };
        put(f0, w0, r, s / d0);
        put(f1, w1, r, o / d1);
    }
}

// ---- the same outputs from a bucket of the bucketed sorter (fmgpu_bucketsort.hip): rows first .. first + count - 1 with their text positions, the suffix array itself is never held
// pass 1: BWT symbols; flag[j] = row first + j is sampled
__global__ __launch_bounds__(256) void k_bucket_bwt(const uint8_t* __restrict__ text, const idx_t* __restrict__ pos, uint64_t count, uint64_t first, uint64_t n, uint8_t* __restrict__ bwt,
                                                    const uint64_t* __restrict__ sstart, uint64_t nseq, uint64_t rate, uint32_t* __restrict__ flag) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p = pos[j];
        bwt[first + j] = text[p ? p - 1 : n - 1];
        if (flag) { uint64_t s, o; seq_of(sstart, nseq, p, s, o); flag[j] = o % rate == 0 ? 1u : 0u; }
    }
}
// pass 2: presence bits and values; at[j] = sampled rows of the bucket before row j, *base = sampled rows of the buckets before this one
__global__ __launch_bounds__(256) void k_bucket_samples(const idx_t* __restrict__ pos, uint64_t count, uint64_t first, const uint64_t* __restrict__ sstart, uint64_t nseq,
                                                        const uint32_t* __restrict__ flag, const uint32_t* __restrict__ at, const unsigned long long* __restrict__ base,
                                                        unsigned long long* __restrict__ bits, unsigned long long* __restrict__ f0, unsigned long long* __restrict__ f1,
                                                        uint32_t w0, uint32_t w1, uint64_t d0, uint64_t d1) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (uint64_t)gridDim.x * blockDim.x) {
        if (!flag[j]) continue;
        uint64_t s, o; seq_of(sstart, nseq, pos[j], s, o);
        const uint64_t i = first + j, r = *base + at[j];
        atomicOr(&bits[i >> 6], 1ull << (i & 63u));
        auto put = [](unsigned long long* data, uint32_t width, uint64_t idx, uint64_t v) {   // DenseVector::push_back layout, DenseVector.h:124-144
            uint64_t begin = idx * width; uint32_t off = (uint32_t)(begin & 63u);
            atomicOr(&data[begin >> 6], (unsigned long long)(v << off));
            if (off + width > 64) atomicOr(&data[(begin >> 6) + 1], (unsigned long long)(v >> (64u - off)));
        };
        put(f0, w0, r, s / d0);
        put(f1, w1, r, o / d1);
    }
}
__global__ void k_bucket_advance(unsigned long long* __restrict__ base, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ at, uint64_t count) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && count) *base += (unsigned long long)at[count - 1] + flag[count - 1];
}
// ---- ... and from the inverse suffix array (sort_suffixes_isa): rank[p] = the row of suffix p
__global__ __launch_bounds__(256) void k_isa_bwt(const uint8_t* __restrict__ text, const idx_t* __restrict__ rank, uint64_t n, uint8_t* __restrict__ bwt) {
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x) bwt[rank[p]] = text[p ? p - 1 : n - 1];
}
// the i-th sampled position of the text: sequence s = the last one with sbase[s] <= i, offset (i - sbase[s]) * rate
__device__ __forceinline__ void sampled_position(const uint64_t* __restrict__ sstart, const uint64_t* __restrict__ sbase, uint64_t nseq, uint64_t rate, uint64_t i, uint64_t& s, uint64_t& o) {
    uint64_t lo = 0, hi = nseq;
    while (hi - lo > 1) { uint64_t mid = (lo + hi) >> 1; if (sbase[mid] <= i) lo = mid; else hi = mid; }
    s = lo; o = (i - sbase[lo]) * rate;
}
__global__ __launch_bounds__(256) void k_isa_presence(const idx_t* __restrict__ rank, const uint64_t* __restrict__ sstart, const uint64_t* __restrict__ sbase, uint64_t nseq, uint64_t rate,
                                                      uint64_t nsampled, unsigned long long* __restrict__ bits) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nsampled; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t s, o; sampled_position(sstart, sbase, nseq, rate, i, s, o);
        const uint64_t r = rank[sstart[s] + o];
        atomicOr(&bits[r >> 6], 1ull << (r & 63u));
    }
}
__global__ __launch_bounds__(256) void k_isa_values(const idx_t* __restrict__ rank, const uint64_t* __restrict__ sstart, const uint64_t* __restrict__ sbase, uint64_t nseq, uint64_t rate,
                                                    uint64_t nsampled, const uint64_t* __restrict__ bits, const cnt_t* __restrict__ g,
                                                    unsigned long long* __restrict__ f0, unsigned long long* __restrict__ f1, uint32_t w0, uint32_t w1, uint64_t d0, uint64_t d1) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nsampled; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t s, o; sampled_position(sstart, sbase, nseq, rate, i, s, o);
        const uint64_t row = rank[sstart[s] + o];
        uint64_t r = g[row >> 9];                                   // sampled rows before this one
        const uint64_t* w = bits + (row >> 9) * 8;
        const uint32_t bit = (uint32_t)(row & 511u);
        for (uint32_t k = 0; k < (bit >> 6); ++k) r += (uint64_t)__popcll(w[k]);
        if (bit & 63u) r += (uint64_t)__popcll(w[bit >> 6] & ((1ull << (bit & 63u)) - 1ull));
        auto put = [](unsigned long long* data, uint32_t width, uint64_t idx, uint64_t v) {   // DenseVector::push_back layout, DenseVector.h:124-144
            uint64_t begin = idx * width; uint32_t off = (uint32_t)(begin & 63u);
            atomicOr(&data[begin >> 6], (unsigned long long)(v << off));
            if (off + width > 64) atomicOr(&data[(begin >> 6) + 1], (unsigned long long)(v >> (64u - off)));
        };
        put(f0, w0, r, s / d0);
        put(f1, w1, r, o / d1);
    }
}
// sampled rows per 512 rows, from the finished presence bits (what k_sa_bits counts on the way)
__global__ __launch_bounds__(256) void k_count_presence(const uint64_t* __restrict__ bits, uint64_t nl1, cnt_t* __restrict__ g) {
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nl1; b += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t c = 0;
        for (uint32_t k = 0; k < 8; ++k) c += (uint32_t)__popcll(bits[b * 8 + k]);
        g[b] = (cnt_t)c;
    }
}

static int make_format_a(const uint8_t* bwt, uint64_t n, uint32_t sigma, const idx_t* dC, DevString& s, int layout, hipStream_t stream) {
    uint64_t nblocks = n / 64 + 1;
    uint32_t bstride = sigma <= 5 ? 64u : 12u * sigma;
    s.layout = layout; s.sigma = (int)sigma; s.n = n; s.family = FAM_A;
    s.bitct = (int)bit_width64((uint64_t)sigma - 1);
    DBuf blk, sup, cnt; int rc;
    if ((rc = blk.alloc(nblocks * bstride + 64))) return rc;
    FM_HIP(hipMemsetAsync(blk.p, 0, blk.bytes, stream));
    if (kWide) {
        if ((rc = sup.alloc(((n >> kSuperShift) + 1) * sigma * 8))) return rc;
        FM_HIP(hipMemsetAsync(sup.p, 0, sup.bytes, stream));
    }
    if ((rc = cnt.alloc(nblocks * sigma * sizeof(cnt_t)))) return rc;
    k_blocks_bits<<<grid_for(nblocks * 64), 256, 0, stream>>>(bwt, n, nblocks, sigma, bstride, blk.as<uint8_t>(), cnt.as<cnt_t>());
    FM_LAUNCHED("k_blocks_bits");
    Temp tmp;
    for (uint32_t c = 0; c < sigma; ++c) {
        cnt_t* p = cnt.as<cnt_t>() + (uint64_t)c * nblocks;
        rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceScan::ExclusiveSum(t, bytes, p, p, (size_t)nblocks, stream); });
        if (rc) return rc;
    }
    k_blocks_counts<<<grid_for(nblocks * sigma), 256, 0, stream>>>(cnt.as<cnt_t>(), nblocks, sigma, bstride, dC, blk.as<uint8_t>(), sup.as<uint64_t>());
    FM_LAUNCHED("k_blocks_counts");
    FM_HIP(hipStreamSynchronize(stream));
    s.blk_bytes = blk.bytes; s.blk = blk.take();
    s.sup_bytes = kWide ? sup.bytes : 0; s.sup = sup.take();
    s.va = ViewA{(const uint8_t*)s.blk, bstride, sigma, dC, (const uint64_t*)s.sup, 0u, 0};
    return 0;
}

// ------------------------------------------------------------------ Format M from the symbols (string/Wavelet.h:40-72 restated as bulk passes)
// Level l of the tree sees the symbols stably ordered by the digits above it (each push_back of the reference appends to the node of the
// symbol's prefix, Wavelet.h:56-65): one stable radix sort on those bits yields every node of the level as a contiguous slice.
struct LevelNodes {             // per level, per node of the level: first position in the sorted order, length, first block (within the level)
    uint64_t start[64], len[64], first_block[64];
    uint32_t nnodes;
};
// one wave per block of 64 positions of the level: digit planes by ballot, per-value popcounts into cnt[v * nblk + t]
__global__ __launch_bounds__(256) void k_m_planes(const uint8_t* __restrict__ sym, LevelNodes ln, uint64_t nblk, uint32_t bits, uint32_t shift, uint32_t stride,
                                                  uint8_t* __restrict__ data, uint64_t level_off, cnt_t* __restrict__ cnt) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t t = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; t < nblk; t += ((uint64_t)gridDim.x * blockDim.x) >> 6) {
        uint32_t lo = 0, hi = ln.nnodes;                           // last node with first_block <= t
        while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (ln.first_block[mid] <= t) lo = mid; else hi = mid; }
        const uint64_t pos = (t - ln.first_block[lo]) * 64 + lane;
        const bool valid = pos < ln.len[lo];
        const uint32_t digit = valid ? ((uint32_t)sym[ln.start[lo] + pos] >> shift) & ((1u << bits) - 1u) : 0u;
        uint8_t* blk = data + level_off + t * stride;             // (the blocks of a level's nodes are contiguous in node order)
        for (uint32_t k = 0; k < bits; ++k) {
            const uint64_t plane = __ballot(valid && ((digit >> k) & 1u));
            if (lane == k) reinterpret_cast<uint64_t*>(blk + (4u << bits))[k] = plane;
        }
        for (uint32_t v = 0; v < (1u << bits); ++v) {
            const uint64_t m = __ballot(valid && digit == v);
            if (lane == v) cnt[(uint64_t)v * nblk + t] = (cnt_t)__popcll(m);
        }
    }
}
// after the per-value exclusive scans over the level: counts before each block within its node (wide: within its super-block, the rest into `super`)
__global__ __launch_bounds__(256) void k_m_counts(const cnt_t* __restrict__ scan, LevelNodes ln, uint64_t nblk, uint32_t bits, uint32_t stride,
                                                  uint8_t* __restrict__ data, uint64_t level_off, uint64_t* __restrict__ super, const uint32_t* __restrict__ node_super, uint32_t first_node) {
    const uint64_t total = nblk << bits;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t t = w >> bits; const uint32_t v = (uint32_t)(w & ((1u << bits) - 1u));
        uint32_t lo = 0, hi = ln.nnodes;
        while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (ln.first_block[mid] <= t) lo = mid; else hi = mid; }
        const uint64_t fb = ln.first_block[lo], rel = t - fb;
        const cnt_t* sv = scan + (uint64_t)v * nblk;
        uint32_t* out = reinterpret_cast<uint32_t*>(data + level_off + t * stride) + v;
        if constexpr (kWide) {
            const uint64_t sb_first = fb + super_first_block(rel);
            *out = (uint32_t)(sv[t] - sv[sb_first]);
            if (t == sb_first) super[((uint64_t)node_super[first_node + lo] + (rel >> (kSuperShift - 6))) * 8u + v] = sv[t] - sv[fb];
        } else *out = (uint32_t)(sv[t] - sv[fb]);
    }
}

int make_format_m(const uint8_t* symbols, uint64_t n, uint32_t sigma, const idx_t* dC, DevString& s, int layout, hipStream_t stream) {
    const uint32_t bitct = std::max<uint32_t>(1, bit_width64((uint64_t)sigma - 1));    // Wavelet.h:31  bits = bit_width(Sigma-1)
    uint32_t dig[3], nlev = 0;
    digits_of(bitct, dig, &nlev);
    s.layout = layout; s.sigma = (int)sigma; s.n = n; s.family = FAM_WAVELET; s.bitct = (int)bitct;
    ViewM vm{};
    vm.C = dC; vm.sigma = sigma; vm.bitct = bitct; vm.nlevels = nlev;
    std::vector<LevelNodes> lnodes(nlev);
    std::vector<uint64_t> node_off, level_off(nlev), level_blocks(nlev);
    std::vector<uint32_t> node_super;
    uint64_t total_bytes = 0, super_rows = 0;
    uint32_t above = 0, first_node = 0;
    DBuf hist, sorted; Temp tmp; int rc;
    if ((rc = hist.alloc(256 * 8))) return rc;
    // pass 1: node lengths per level -> block offsets
    for (uint32_t l = 0; l < nlev; ++l) {
        const uint32_t d = dig[l], shift = bitct - above - d, stride = d == 3 ? 64u : (d == 2 ? 32u : 16u), nn = 1u << above;
        vm.lv[l] = LevelM{d, shift, stride, first_node};
        unsigned long long hh[256];
        std::memset(hh, 0, sizeof hh);
        if (above == 0) hh[0] = n;
        else {
            FM_HIP(hipMemsetAsync(hist.p, 0, 256 * 8, stream));
            k_histogram<<<dim3(2048), 256, 0, stream>>>(symbols, n, shift + d, hist.as<unsigned long long>());
            FM_LAUNCHED("k_histogram");
            FM_HIP(hipMemcpy(hh, hist.p, 256 * 8, hipMemcpyDeviceToHost));
        }
        LevelNodes& ln = lnodes[l];
        ln.nnodes = nn;
        uint64_t acc = 0, blocks = 0;
        level_off[l] = total_bytes;
        for (uint32_t p = 0; p < nn; ++p) {
            ln.start[p] = acc; ln.len[p] = hh[p]; ln.first_block[p] = blocks;
            node_off.push_back(total_bytes + blocks * stride);
            node_super.push_back((uint32_t)super_rows);
            acc += hh[p]; blocks += hh[p] / 64 + 1;
            super_rows += (hh[p] >> kSuperShift) + 1;
        }
        level_blocks[l] = blocks;
        total_bytes += blocks * stride;
        first_node += nn; above += d;
    }
    vm.nnodes = first_node;
    DBuf data, aux, sup;
    if ((rc = data.alloc(total_bytes + 64)) || (rc = aux.alloc(node_off.size() * 8))) return rc;
    FM_HIP(hipMemsetAsync(data.p, 0, data.bytes, stream));
    FM_HIP(hipMemcpy(aux.p, node_off.data(), node_off.size() * 8, hipMemcpyHostToDevice));
    const size_t ns_bytes = (node_super.size() * 4 + 63) / 64 * 64;            // wide: [node_super u32...][super rows of 8 u64]
    if (kWide) {
        if ((rc = sup.alloc(ns_bytes + super_rows * 64))) return rc;
        FM_HIP(hipMemsetAsync(sup.p, 0, sup.bytes, stream));
        FM_HIP(hipMemcpy(sup.p, node_super.data(), node_super.size() * 4, hipMemcpyHostToDevice));
    }
    uint64_t* d_super = kWide ? reinterpret_cast<uint64_t*>(sup.as<uint8_t>() + ns_bytes) : nullptr;
    const uint32_t* d_node_super = kWide ? sup.as<uint32_t>() : nullptr;
    // pass 2: per level, sort by the digits above, cut the slices into blocks
    DBuf cnt;
    above = 0;
    for (uint32_t l = 0; l < nlev; ++l) {
        const LevelM L = vm.lv[l];
        const uint8_t* src = symbols;
        if (above > 0 && n > 0) {
            if (!sorted.p && (rc = sorted.alloc(n + 64))) return rc;
            rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceRadixSort::SortKeys(t, bytes, symbols, sorted.as<uint8_t>(), (size_t)n, (int)(L.shift + L.bits), (int)bitct, stream); });
            if (rc) return rc;
            src = sorted.as<uint8_t>();
        }
        const uint64_t nblk = level_blocks[l];
        if ((rc = cnt.alloc((nblk << L.bits) * sizeof(cnt_t)))) return rc;
        k_m_planes<<<grid_for(nblk * 64), 256, 0, stream>>>(src, lnodes[l], nblk, L.bits, L.shift, L.stride, data.as<uint8_t>(), level_off[l], cnt.as<cnt_t>());
        FM_LAUNCHED("k_m_planes");
        for (uint32_t v = 0; v < (1u << L.bits); ++v) {
            cnt_t* p = cnt.as<cnt_t>() + (uint64_t)v * nblk;
            rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceScan::ExclusiveSum(t, bytes, p, p, (size_t)nblk, stream); });
            if (rc) return rc;
        }
        k_m_counts<<<grid_for(nblk << L.bits), 256, 0, stream>>>(cnt.as<cnt_t>(), lnodes[l], nblk, L.bits, L.stride, data.as<uint8_t>(), level_off[l], d_super, d_node_super, L.first_node);
        FM_LAUNCHED("k_m_counts");
        FM_HIP(hipStreamSynchronize(stream));
        above += L.bits;
    }
    s.blk_bytes = data.bytes; s.blk = data.take();
    s.aux_bytes = aux.bytes; s.aux = aux.take();
    s.sup_bytes = kWide ? sup.bytes : 0; s.sup = sup.take();
    vm.data = (const uint8_t*)s.blk; vm.node_off = (const uint64_t*)s.aux;
    vm.node_super = kWide ? (const uint32_t*)s.sup : nullptr;
    vm.super = kWide ? reinterpret_cast<const uint64_t*>((const uint8_t*)s.sup + ns_bytes) : nullptr;
    s.vm = vm;
    return 0;
}

template <class Occ>
__global__ __launch_bounds__(256) void k_symbols(Occ occ, uint64_t n, uint8_t* __restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = (uint8_t)occ.symbol((idx_t)i);
}

int build_format_a_shadow(DevString& s, const idx_t* dC, hipStream_t stream) {
    if (s.family == FAM_A || s.shadow || s.n == 0) return 0;
    DBuf sym; int rc;
    if ((rc = sym.alloc(s.n + 64))) return rc;
    rc = dispatch_native(s, [&](auto occ, auto) {
        k_symbols<<<grid_for(s.n), 256, 0, stream>>>(occ, s.n, sym.as<uint8_t>());
        return 0;
    });
    FM_LAUNCHED("k_symbols");
    DevString t;
    if ((rc = make_format_a(sym.as<uint8_t>(), s.n, (uint32_t)s.sigma, dC, t, s.layout, stream))) return rc;
    s.shadow = t.blk; s.shadow_bytes = t.blk_bytes + t.sup_bytes; s.shadow_sup = t.sup; s.shadow_sup_bytes = t.sup_bytes; s.va = t.va;
    return 0;
}

static int make_string(const uint8_t* bwt, uint64_t n, uint32_t sigma, const idx_t* dC, DevString& s, int layout, hipStream_t stream) {
    return layout == FMGPU_WAVELET ? make_format_m(bwt, n, sigma, dC, s, layout, stream) : make_format_a(bwt, n, sigma, dC, s, layout, stream);
}

namespace api {
#include "fmgpu_api_decl.h"

int fmgpu_build_index(const uint8_t* seqs, const uint64_t* seq_off, uint64_t nseq, int32_t sigma, int32_t layout, uint64_t sampling_rate,
                      int32_t bidirectional, int32_t keep_host, fmgpu_index_t* out, fmgpu_built_t* built_out) {
    if (!out) return fail(FMGPU_ERR_INVALID, "out is null");
    *out = nullptr;
    if (built_out) *built_out = nullptr;
    if (!seqs || !seq_off || nseq == 0) return fail(FMGPU_ERR_INVALID, "seqs / seq_off is null or nseq == 0");
    if (sigma < 2 || sigma > 256) return fail(FMGPU_ERR_INVALID, "sigma must be in [2, 256]");
    if (sampling_rate == 0) return fail(FMGPU_ERR_INVALID, "sampling_rate must be >= 1");
    // the layout names the reference type the caller replaces; on the device every blocked layout is held as the LF-ready block
    // table (Format A) and Wavelet as the multi-ary wavelet tree (Format M) — the answers of a String_c do not depend on its layout
    if (layout < FMGPU_IB8 || layout > FMGPU_FBV_2048_64K) return fail(FMGPU_ERR_INVALID, "unknown layout id");
    hipStream_t stream = nullptr;
    Staged soff, sseq;
    int rc;
    if ((rc = soff.in(seq_off, (nseq + 1) * 8, stream))) return rc;
    std::vector<uint64_t> hoff(nseq + 1);
    FM_HIP(hipMemcpy(hoff.data(), soff.dev, (nseq + 1) * 8, hipMemcpyDeviceToHost));
    for (uint64_t s = 0; s < nseq; ++s) if (hoff[s + 1] < hoff[s]) return fail(FMGPU_ERR_INVALID, "seq_off is not non-decreasing");
    const uint64_t total = hoff[nseq] - hoff[0], n = total + nseq;
    if (!kWide && n >= kNarrowLimit) return fail(FMGPU_ERR_UNSUPPORTED, "the 32-bit-row build indexes fewer than 2^32 - 64 rows");
    if ((rc = sseq.in(seqs, hoff[nseq], stream))) return rc;

    std::unique_ptr<Index> x(new (std::nothrow) Index());
    std::unique_ptr<Built> built(keep_host ? new (std::nothrow) Built() : nullptr);
    if (!x || (keep_host && !built)) return fail(FMGPU_ERR_NOMEM, "host allocation");
    if (built) built->part.resize(9);
    auto bail = [&](int code) { api::fmgpu_index_destroy(reinterpret_cast<fmgpu_index_t>(x.release())); return code; };
    (void)hipGetDevice(&x->hdr.device);

    // the suffix sorter (FMGPU_OPT_SUFFIX_SORTER; 0: by the memory each needs, in this order):
    //   1 all suffixes at once — suffix array, rank array, key buffers of all rows: 30 / 42 bytes per row beside the text (build_suffix_array);
    //   2 bucket by bucket with the inverse suffix array as the rank array of the doubling rounds: 6 / 10 bytes per row + one bucket + the tied rows (sort_suffixes_isa);
    //   3 bucket by bucket with no array of n entries: 2 bytes per row + one bucket, ties broken by further symbols of the text (sort_suffixes_bucketed)
    const uint64_t bucket_rows = (uint64_t)opt(FMGPU_OPT_BUCKET_ROWS);
    int sorter = (int)opt(FMGPU_OPT_SUFFIX_SORTER);
    if (sorter == 0) {
        size_t free_b = 0, total_b = 0;
        sorter = 1;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            if ((double)n * (kWide ? 44.0 : 32.0) > (double)free_b) sorter = 2;
            if (sorter == 2 && (double)n * (2.0 + sizeof(idx_t)) * 1.5 > (double)free_b) sorter = 3;
        }
    }
    const bool bucketed = sorter != 1;
    DBuf text, sa, bwt;
    if ((rc = text.alloc(n)) || (rc = bwt.alloc(n)) || (!bucketed && (rc = sa.alloc(n * sizeof(idx_t))))) return bail(rc);
    k_assemble_text<<<grid_for(n), 256, 0, stream>>>((const uint8_t*)sseq.dev, (const uint64_t*)soff.dev, nseq, text.as<uint8_t>(), n);
    {
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) return bail(hip_fail(le, "k_assemble_text"));
        DBuf bad; if ((rc = bad.alloc(4))) return bail(rc);
        (void)hipMemsetAsync(bad.p, 0, 4, stream);
        k_check_symbols<<<grid_for(n), 256, 0, stream>>>(text.as<uint8_t>(), n, (uint32_t)sigma, bad.as<unsigned int>());
        unsigned int hb = 0;
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(&hb, bad.p, 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return bail(hip_fail(e, "k_check_symbols"));
        if (hb) return bail(fail(FMGPU_ERR_INVALID, "a sequence holds a symbol >= sigma"));
    }

    // ---- sampled suffix array in the reference's SparseArray layout (suffixarray/SparseArray.h:31-76): sizes and arrays
    std::vector<uint64_t> sstart(nseq + 1);
    uint64_t maxlen = 0, nsampled = 0;
    for (uint64_t s = 0; s <= nseq; ++s) sstart[s] = hoff[s] - hoff[0] + s;
    for (uint64_t s = 0; s < nseq; ++s) {
        const uint64_t len = hoff[s + 1] - hoff[s];                  // positions 0..len (delimiter slot included)
        maxlen = std::max(maxlen, len);
        nsampled += len / sampling_rate + 1;                        // the sampled ones among them
    }
    // DenseMultiVector ctor (DenseMultiVector.h:65-103): largest value and gcd per field
    uint64_t largest0 = nseq - 1, div0 = nseq >= 2 ? 1 : 0;
    uint64_t largest1 = (maxlen / sampling_rate) * sampling_rate, div1 = largest1 ? sampling_rate : 0;
    if (div0 == 0) div0 = 1; if (largest0 == 0) largest0 = 1;
    if (div1 == 0) div1 = 1; if (largest1 == 0) largest1 = 1;
    const uint32_t w0 = bit_width64(largest0 / div0), w1 = bit_width64(largest1 / div1);
    const uint64_t nl0 = n / 65536 + 1, nl1 = n / 512 + 1, nwords = nl1 * 8;
    DBuf dstart, g;
    if ((rc = dstart.alloc((nseq + 1) * 8)) || (rc = g.alloc((nl1 + 1) * sizeof(cnt_t)))) return bail(rc);
    {
        hipError_t e = hipMemcpy(dstart.p, sstart.data(), (nseq + 1) * 8, hipMemcpyHostToDevice);
        if (e != hipSuccess) return bail(hip_fail(e, "copy sstart"));
        if ((e = hipMalloc(&x->sa_bits, nwords * 8)) != hipSuccess || (e = hipMalloc(&x->sa_l0, nl0 * 8)) != hipSuccess || (e = hipMalloc(&x->sa_l1, nl1 * 2)) != hipSuccess)
            return bail(hip_fail(e, "hipMalloc(sa)"));
        (void)hipMemsetAsync(x->sa_bits, 0, nwords * 8, stream);
        (void)hipMemsetAsync(g.p, 0, (nl1 + 1) * sizeof(cnt_t), stream);
    }
    uint64_t f0words = 0, f1words = 0;
    auto alloc_fields = [&](uint64_t nvalues) -> int {
        f0words = (nvalues * w0 + 63) / 64; f1words = (nvalues * w1 + 63) / 64;
        hipError_t e;
        if ((e = hipMalloc(&x->sa_f0, (f0words + 1) * 8)) != hipSuccess || (e = hipMalloc(&x->sa_f1, (f1words + 1) * 8)) != hipSuccess) return hip_fail(e, "hipMalloc(sa fields)");
        (void)hipMemsetAsync(x->sa_f0, 0, (f0words + 1) * 8, stream);
        (void)hipMemsetAsync(x->sa_f1, 0, (f1words + 1) * 8, stream);
        return 0;
    };
    // g = sampled rows per 512 rows -> running counts, the two counter levels; returns the number of sampled rows
    auto finish_levels = [&](uint64_t* nvalues_out) -> int {
        Temp tmp;
        int r = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceScan::ExclusiveSum(t, bytes, g.as<cnt_t>(), g.as<cnt_t>(), (size_t)(nl1 + 1), stream); });
        if (r) return r;
        cnt_t nvalues_c = 0;
        hipError_t e = hipMemcpy(&nvalues_c, g.as<cnt_t>() + nl1, sizeof(cnt_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return hip_fail(e, "sample count");
        *nvalues_out = nvalues_c;
        k_sa_levels<<<grid_for(nl1), 256, 0, stream>>>(g.as<cnt_t>(), nl1, nl0, (uint64_t*)x->sa_l0, (uint16_t*)x->sa_l1);
        FM_LAUNCHED("k_sa_levels");
        return 0;
    };

    // bwt (and, forward text only, the sampled entries) of `text` through the bucketed sorter
    auto bucketed_pass = [&](bool with_samples) -> int {
        DBuf base; int r;
        if ((r = base.alloc(8))) return r;
        FM_HIP(hipMemsetAsync(base.p, 0, 8, stream));
        Temp tmp;
        return sort_suffixes_bucketed(text.as<uint8_t>(), n, (uint32_t)sigma, bucket_rows, [&](uint64_t first, const idx_t* pos, uint64_t count, void* scratch, size_t scratch_bytes) -> int {
            if (with_samples && scratch_bytes < count * 8) return fail(FMGPU_ERR_HIP, "bucket scratch too small");
            uint32_t* flag = with_samples ? reinterpret_cast<uint32_t*>(scratch) : nullptr;
            uint32_t* at = with_samples ? flag + count : nullptr;
            k_bucket_bwt<<<grid_for(count), 256, 0, stream>>>(text.as<uint8_t>(), pos, count, first, n, bwt.as<uint8_t>(), dstart.as<uint64_t>(), nseq, sampling_rate, flag);
            FM_LAUNCHED("k_bucket_bwt");
            if (!with_samples) return 0;
            int r2 = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceScan::ExclusiveSum(t, bytes, flag, at, (size_t)count, stream); });
            if (r2) return r2;
            k_bucket_samples<<<grid_for(count), 256, 0, stream>>>(pos, count, first, dstart.as<uint64_t>(), nseq, flag, at, base.as<unsigned long long>(), (unsigned long long*)x->sa_bits,
                                                                   (unsigned long long*)x->sa_f0, (unsigned long long*)x->sa_f1, w0, w1, div0, div1);
            FM_LAUNCHED("k_bucket_samples");
            k_bucket_advance<<<1, 64, 0, stream>>>(base.as<unsigned long long>(), flag, at, count);
            FM_LAUNCHED("k_bucket_advance");
            return 0;
        }, stream);
    };

    // ... and through the sorter that leaves the inverse suffix array: bwt[rank[p]] = text[p - 1]; the sampled entries are the text's positions at multiples of the rate
    // within their sequence — enumerated directly (sequence s holds len_s / rate + 1 of them), their rows read off the array
    DBuf sbase;                                                     // sbase[s] = sampled positions of the sequences before s
    auto isa_pass = [&](bool with_samples) -> int {
        DBuf rank; int r;
        if ((r = rank.alloc(n * sizeof(idx_t)))) return r;
        if ((r = sort_suffixes_isa(text.as<uint8_t>(), n, (uint32_t)sigma, bucket_rows, rank.as<idx_t>(), stream))) return r;
        k_isa_bwt<<<grid_for(n), 256, 0, stream>>>(text.as<uint8_t>(), rank.as<idx_t>(), n, bwt.as<uint8_t>());
        FM_LAUNCHED("k_isa_bwt");
        if (with_samples) {
            std::vector<uint64_t> hb(nseq + 1);
            hb[0] = 0;
            for (uint64_t s = 0; s < nseq; ++s) hb[s + 1] = hb[s] + (hoff[s + 1] - hoff[s]) / sampling_rate + 1;
            if ((r = sbase.alloc((nseq + 1) * 8))) return r;
            FM_HIP(hipMemcpy(sbase.p, hb.data(), (nseq + 1) * 8, hipMemcpyHostToDevice));
            k_isa_presence<<<grid_for(nsampled), 256, 0, stream>>>(rank.as<idx_t>(), dstart.as<uint64_t>(), sbase.as<uint64_t>(), nseq, sampling_rate, nsampled, (unsigned long long*)x->sa_bits);
            FM_LAUNCHED("k_isa_presence");
            k_count_presence<<<grid_for(nl1), 256, 0, stream>>>((const uint64_t*)x->sa_bits, nl1, g.as<cnt_t>());
            FM_LAUNCHED("k_count_presence");
            uint64_t got = 0;
            if ((r = finish_levels(&got))) return r;
            if (got != nsampled) return fail(FMGPU_ERR_HIP, "the suffix sorter marked " + std::to_string(got) + " sampled rows, the text holds " + std::to_string(nsampled));
            k_isa_values<<<grid_for(nsampled), 256, 0, stream>>>(rank.as<idx_t>(), dstart.as<uint64_t>(), sbase.as<uint64_t>(), nseq, sampling_rate, nsampled, (const uint64_t*)x->sa_bits,
                                                                 g.as<cnt_t>(), (unsigned long long*)x->sa_f0, (unsigned long long*)x->sa_f1, w0, w1, div0, div1);
            FM_LAUNCHED("k_isa_values");
        }
        FM_HIP(hipStreamSynchronize(stream));
        return 0;
    };

    uint64_t nvalues = 0;
    if (sorter == 3) {
        if ((rc = alloc_fields(nsampled))) return bail(rc);
        if ((rc = bucketed_pass(true))) return bail(rc);
        k_count_presence<<<grid_for(nl1), 256, 0, stream>>>((const uint64_t*)x->sa_bits, nl1, g.as<cnt_t>());
        if ((rc = finish_levels(&nvalues))) return bail(rc);
        if (nvalues != nsampled) return bail(fail(FMGPU_ERR_HIP, "the bucketed sorter marked " + std::to_string(nvalues) + " sampled rows, the text holds " + std::to_string(nsampled)));
    } else if (sorter == 2) {
        if ((rc = alloc_fields(nsampled))) return bail(rc);
        if ((rc = isa_pass(true))) return bail(rc);
        nvalues = nsampled;
    } else {
        if ((rc = build_suffix_array(text.as<uint8_t>(), n, (uint32_t)sigma, sa.as<idx_t>(), stream))) return bail(rc);
        k_bwt<<<grid_for(n), 256, 0, stream>>>(text.as<uint8_t>(), sa.as<idx_t>(), n, bwt.as<uint8_t>());
    }

    // C[c] = #symbols < c  (utils.h:199-206)
    {
        DBuf hist; if ((rc = hist.alloc(256 * 8))) return bail(rc);
        (void)hipMemsetAsync(hist.p, 0, 256 * 8, stream);
        k_histogram<<<dim3(2048), 256, 0, stream>>>(bwt.as<uint8_t>(), n, 0, hist.as<unsigned long long>());
        unsigned long long hh[256];
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(hh, hist.p, 256 * 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return bail(hip_fail(e, "histogram"));
        uint64_t acc = 0;
        std::vector<idx_t> cdev(sigma + 1);
        for (int c = 0; c <= sigma; ++c) { x->hC[c] = acc; cdev[c] = (idx_t)acc; if (c < sigma) acc += hh[c]; }
        hipError_t e2 = hipMalloc((void**)&x->dC, (sigma + 1) * sizeof(idx_t));
        if (e2 != hipSuccess) return bail(hip_fail(e2, "hipMalloc(C)"));
        e2 = hipMemcpy(x->dC, cdev.data(), (sigma + 1) * sizeof(idx_t), hipMemcpyHostToDevice);
        if (e2 != hipSuccess) return bail(hip_fail(e2, "hipMemcpy(C)"));
    }
    if ((rc = make_string(bwt.as<uint8_t>(), n, (uint32_t)sigma, x->dC, x->bwt, layout, stream))) return bail(rc);
    if (built) {
        built->part[0].resize(n);
        hipError_t e = hipMemcpy(built->part[0].data(), bwt.p, n, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return bail(hip_fail(e, "copy bwt"));
        built->part[2].resize((sigma + 1) * 8);
        std::memcpy(built->part[2].data(), x->hC, (sigma + 1) * 8);
    }

    {
        if (sorter == 1) {
            for (uint64_t first = 0; first < n; first += kSliceRows) {
                const uint64_t rows = std::min(kSliceRows, n - first);
                k_sa_bits<<<dim3((unsigned)(((rows + 63) / 64 * 64 + 255) / 256)), 256, 0, stream>>>(sa.as<idx_t>(), n, dstart.as<uint64_t>(), nseq, sampling_rate, (uint64_t*)x->sa_bits,
                                                                                                   g.as<cnt_t>(), first);
            }
            hipError_t le = hipGetLastError();
            if (le != hipSuccess) return bail(hip_fail(le, "k_sa_bits"));
            if ((rc = finish_levels(&nvalues))) return bail(rc);
            if ((rc = alloc_fields(nvalues))) return bail(rc);
            k_sa_values<<<grid_for(n), 256, 0, stream>>>(sa.as<idx_t>(), n, dstart.as<uint64_t>(), nseq, sampling_rate, (const uint64_t*)x->sa_bits, g.as<cnt_t>(),
                                                         (unsigned long long*)x->sa_f0, (unsigned long long*)x->sa_f1, w0, w1, div0, div1);
        }
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return bail(hip_fail(e, "sampled suffix array kernels"));
        x->vsa = ViewSA{(const uint64_t*)x->sa_l0, (const uint16_t*)x->sa_l1, (const uint64_t*)x->sa_bits, (const uint64_t*)x->sa_f0, (const uint64_t*)x->sa_f1, w0, w1, div0, div1};
        x->has_sa = true;
        x->device_bytes += nwords * 8 + nl0 * 8 + nl1 * 2 + (f0words + f1words + 2) * 8;
        x->sa_bytes[0] = nl0 * 8; x->sa_bytes[1] = nl1 * 2; x->sa_bytes[2] = nwords * 8; x->sa_bytes[3] = (f0words + 1) * 8; x->sa_bytes[4] = (f1words + 1) * 8;
        if (built) {
            auto grab = [&](int part, const void* dev, size_t bytes) -> int {
                built->part[part].resize(bytes);
                if (bytes) FM_HIP(hipMemcpy(built->part[part].data(), dev, bytes, hipMemcpyDeviceToHost));
                return 0;
            };
            if ((rc = grab(3, x->sa_l0, nl0 * 8)) || (rc = grab(4, x->sa_l1, nl1 * 2)) || (rc = grab(5, x->sa_bits, nwords * 8)) ||
                (rc = grab(6, x->sa_f0, f0words * 8)) || (rc = grab(7, x->sa_f1, f1words * 8))) return bail(rc);
            uint64_t params[8] = {nvalues * w0, w0, largest0, div0, nvalues * w1, w1, largest1, div1};
            built->part[8].resize(sizeof params);
            std::memcpy(built->part[8].data(), params, sizeof params);
        }
    }
    if (bidirectional) {
        // BiFMIndex.h:78-92: reverse the whole concatenation (delimiters included), second suffix sort
        k_reverse<<<grid_for(n), 256, 0, stream>>>(text.as<uint8_t>(), bwt.as<uint8_t>(), n);
        FM_HIP(hipMemcpyAsync(text.p, bwt.p, n, hipMemcpyDeviceToDevice, stream));
        if (sorter == 3) { if ((rc = bucketed_pass(false))) return bail(rc); }
        else if (sorter == 2) { if ((rc = isa_pass(false))) return bail(rc); }
        else {
            if ((rc = build_suffix_array(text.as<uint8_t>(), n, (uint32_t)sigma, sa.as<idx_t>(), stream))) return bail(rc);
            k_bwt<<<grid_for(n), 256, 0, stream>>>(text.as<uint8_t>(), sa.as<idx_t>(), n, bwt.as<uint8_t>());
        }
        if ((rc = make_string(bwt.as<uint8_t>(), n, (uint32_t)sigma, x->dC, x->rev, layout, stream))) return bail(rc);
        x->bidirectional = true;
        if (built) {
            built->part[1].resize(n);
            hipError_t e = hipMemcpy(built->part[1].data(), bwt.p, n, hipMemcpyDeviceToHost);
            if (e != hipSuccess) return bail(hip_fail(e, "copy bwt_rev"));
        }
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return bail(hip_fail(e, "index construction"));
    text.release(); sa.release(); bwt.release();
    if (opt_on(FMGPU_OPT_LF_TABLE)) {
        if ((rc = build_lf_table(x->bwt, stream))) return bail(rc);
        if (x->bidirectional && (rc = build_lf_table(x->rev, stream))) return bail(rc);
    }
    if ((rc = auto_shadow(x.get(), stream))) return bail(rc);
    if (x->bidirectional) {
        for (DevString* t : {&x->bwt, &x->rev}) if ((rc = build_dense_dna(*t, stream))) return bail(rc);
        if (x->bwt.dense && !x->rev.dense) { (void)hipFree(x->bwt.dense); (void)hipFree(x->bwt.dense_ex); x->bwt.dense = nullptr; x->bwt.dense_ex = nullptr; x->bwt.dense_bytes = 0; x->bwt.dense_nex = 0; }
    }
    x->device_bytes += x->bwt.blk_bytes + x->bwt.aux_bytes + x->bwt.sup_bytes + x->rev.blk_bytes + x->rev.aux_bytes + x->rev.sup_bytes + x->bwt.dense_bytes + x->rev.dense_bytes +
                       x->bwt.shadow_bytes + x->rev.shadow_bytes +
                       (x->bwt.lf_table ? n * sizeof(idx_t) : 0) + (x->rev.lf_table ? n * sizeof(idx_t) : 0);
    if ((rc = fuse_presence_bits(x.get(), stream))) return bail(rc);
    if ((rc = build_pair_table(x.get(), stream))) return bail(rc);
    if ((rc = build_flat_table(x.get(), stream))) return bail(rc);
    *out = reinterpret_cast<fmgpu_index_t>(x.release());
    if (built_out) *built_out = reinterpret_cast<fmgpu_built_t>(built.release());
    return 0;
}

}  // namespace api
}  // namespace FMGPU_NS

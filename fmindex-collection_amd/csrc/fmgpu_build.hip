// fmgpu_build.hip — index construction on the GPU.
//
// Replaces the reference's constructors FMIndex(Sequences, samplingRate, threads) (fmindex/FMIndex.h:58-104) and
// BiFMIndex(Sequences, samplingRate, threads) (fmindex/BiFMIndex.h:107-167), whose heavy lifting is libsais
// (utils.h:97-129).  Same outputs: text = every sequence followed by a 0 delimiter (utils.h:382-411), suffix order of
// the plain byte string (a proper prefix sorts first), bwt[i] = text[(sa[i]+n-1) % n] (utils.h:145-163), sampled
// entries (seqId, pos) where pos % samplingRate == 0 (FMIndex.h:79-101), bwtRev = BWT of the reversed concatenation
// (BiFMIndex.h:78-92).
//
// Suffix sorting, MI355X style (n < 2^32 - 64, 32-bit suffix indices):
//   1. key[i] = the first K symbols of suffix i packed into 64 bits (symbol+1 per field, 0 = past the end, so shorter
//      sorts first); one rocPRIM radix sort of (key, i) pairs orders all suffixes by their K-prefix
//      (K = 21 for DNA): on a random 3.1 Gbp text that already separates all but ~0.1 % of them.
//   2. prefix doubling on the ties only: rows whose group is not a singleton are compacted, keyed by
//      (group start, rank of suffix i+h) and radix-sorted; groups split, h doubles, until no ties are left.
// BWT, the LF-ready 64-byte blocks (Format A, fmgpu_common.h) and the reference-layout sampled suffix array are then
// produced by streaming kernels without leaving HBM.
#include "fmgpu_common.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <memory>
#include <new>

namespace fmgpu {

struct DBuf {    // RAII device allocation
    void* p = nullptr; size_t bytes = 0;
    int alloc(size_t b) { release(); bytes = b ? b : 8; FM_HIP(hipMalloc(&p, bytes)); return 0; }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
    ~DBuf() { release(); }
    DBuf() = default;
    DBuf(const DBuf&) = delete; DBuf& operator=(const DBuf&) = delete;
};

// ------------------------------------------------------------------ text assembly
__global__ __launch_bounds__(256) void k_assemble_text(const uint8_t* __restrict__ seqs, const uint64_t* __restrict__ seq_off, uint64_t nseq,
                                                       uint8_t* __restrict__ text, uint64_t n) {
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    // sequence s occupies text [seq_off[s]-seq_off[0]+s, seq_off[s+1]-seq_off[0]+s], the last slot being the delimiter
    uint64_t lo = 0, hi = nseq;           // largest s with start(s) <= p
    const uint64_t base = seq_off[0];
    while (hi - lo > 1) { uint64_t mid = (lo + hi) >> 1; if (seq_off[mid] - base + mid <= p) lo = mid; else hi = mid; }
    uint64_t o = p - (seq_off[lo] - base + lo), len = seq_off[lo + 1] - seq_off[lo];
    text[p] = o < len ? seqs[seq_off[lo] + o] : 0;
}
__global__ __launch_bounds__(256) void k_reverse(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n) {
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) out[p] = in[n - 1 - p];
}
__global__ __launch_bounds__(256) void k_check_symbols(const uint8_t* __restrict__ t, uint64_t n, uint32_t sigma, unsigned int* bad) {
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n && t[p] >= sigma) atomicOr(bad, 1u);
}

// ------------------------------------------------------------------ suffix sorting
__global__ __launch_bounds__(256) void k_pack_keys(const uint8_t* __restrict__ t, uint64_t n, uint32_t K, uint32_t b,
                                                   uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = 0;
    for (uint32_t j = 0; j < K; ++j) k = (k << b) | (i + j < n ? (uint64_t)t[i + j] + 1ull : 0ull);
    keys[i] = k; vals[i] = (uint32_t)i;
}
// head[i] = 1 if row i starts a new group; v[i] = head ? i : 0 (for the running-max scan)
__global__ __launch_bounds__(256) void k_heads(const uint64_t* __restrict__ keys, uint64_t m, const uint32_t* __restrict__ where,
                                               uint32_t* __restrict__ v, uint8_t* __restrict__ head) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    bool h = j == 0 || keys[j] != keys[j - 1];
    head[j] = h ? 1 : 0;
    v[j] = h ? (where ? where[j] : (uint32_t)j) : 0u;
}
struct MaxOp { __host__ __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; } };

// after the scan: gs[j] = group start (SA index); rank[pos] = gs; flag rows of non-singleton groups
__global__ __launch_bounds__(256) void k_apply_groups(const uint32_t* __restrict__ pos, const uint32_t* __restrict__ gs, const uint8_t* __restrict__ head,
                                                      uint64_t m, uint32_t* __restrict__ rank, uint8_t* __restrict__ active,
                                                      uint32_t* __restrict__ sa, const uint32_t* __restrict__ where) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    uint32_t p = pos[j];
    rank[p] = gs[j];
    if (sa) sa[where[j]] = p;
    bool single = head[j] && (j + 1 == m || head[j + 1]);
    active[j] = single ? 0 : 1;
}
__global__ __launch_bounds__(256) void k_iota(uint32_t* __restrict__ a, uint64_t m) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) a[j] = (uint32_t)j;
}
__global__ __launch_bounds__(256) void k_round_keys(const uint32_t* __restrict__ aidx, const uint32_t* __restrict__ sa, const uint32_t* __restrict__ rank,
                                                    uint64_t m, uint64_t n, uint64_t h, uint64_t* __restrict__ keys, uint32_t* __restrict__ pos) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    uint32_t p = sa[aidx[j]];
    uint64_t second = (uint64_t)p + h < n ? (uint64_t)rank[p + h] + 1ull : 0ull;
    keys[j] = ((uint64_t)rank[p] << 32) | second;
    pos[j] = p;
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

static inline dim3 grid_for(uint64_t n) { return dim3((unsigned)((n + 255) / 256)); }

// sa_out: n uint32 (device).  text: device, n symbols < sigma.
static int build_suffix_array(const uint8_t* text, uint64_t n, uint32_t sigma, uint32_t* sa_out, hipStream_t stream) {
    if (n == 0) return 0;
    uint32_t b = 0; while ((1u << b) <= sigma) ++b;         // bits for values 0..sigma
    const uint32_t K = 64 / b;
    Temp tmp;
    DBuf rank; int rc;
    if ((rc = rank.alloc(n * 4))) return rc;
    DBuf aidx;                                               // active SA indices (ascending)
    uint64_t m = 0;
    {
        DBuf k0, k1, v1;
        if ((rc = k0.alloc(n * 8)) || (rc = k1.alloc(n * 8)) || (rc = v1.alloc(n * 4))) return rc;
        // sa_out doubles as the first value buffer
        k_pack_keys<<<grid_for(n), 256, 0, stream>>>(text, n, K, b, k0.as<uint64_t>(), sa_out);
        hipcub::DoubleBuffer<uint64_t> dk(k0.as<uint64_t>(), k1.as<uint64_t>());
        hipcub::DoubleBuffer<uint32_t> dv(sa_out, v1.as<uint32_t>());
        rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceRadixSort::SortPairs(t, bytes, dk, dv, (size_t)n, 0, (int)(K * b), stream); });
        if (rc) return rc;
        if (dv.Current() != sa_out) FM_HIP(hipMemcpyAsync(sa_out, dv.Current(), n * 4, hipMemcpyDeviceToDevice, stream));
        const uint64_t* sorted = dk.Current();
        uint64_t* spare = dk.Alternate();                    // reused: u32 gs + u8 head + u8 active need 6n bytes <= 8n
        uint32_t* gsv = reinterpret_cast<uint32_t*>(spare);
        uint8_t* headv = reinterpret_cast<uint8_t*>(spare) + n * 4;
        uint8_t* actv = headv + n;
        k_heads<<<grid_for(n), 256, 0, stream>>>(sorted, n, nullptr, gsv, headv);
        rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceScan::InclusiveScan(t, bytes, gsv, gsv, MaxOp{}, (size_t)n, stream); });
        if (rc) return rc;
        k_apply_groups<<<grid_for(n), 256, 0, stream>>>(sa_out, gsv, headv, n, rank.as<uint32_t>(), actv, nullptr, nullptr);
        // compact the active SA indices
        DBuf cnt; if ((rc = cnt.alloc(8))) return rc;
        v1.release();
        DBuf iota; if ((rc = iota.alloc(n * 4)) || (rc = aidx.alloc(n * 4))) return rc;
        k_iota<<<grid_for(n), 256, 0, stream>>>(iota.as<uint32_t>(), n);
        rc = cub_call(tmp, [&](void* t, size_t& bytes) {
            return hipcub::DeviceSelect::Flagged(t, bytes, iota.as<uint32_t>(), actv, aidx.as<uint32_t>(), cnt.as<uint64_t>(), (size_t)n, stream); });
        if (rc) return rc;
        FM_HIP(hipMemcpyAsync(&m, cnt.p, 8, hipMemcpyDeviceToHost, stream));
        FM_HIP(hipStreamSynchronize(stream));
    }
    uint64_t h = K;
    DBuf k0, k1, p0, p1, gs, head, act, aidx2, cnt;
    if (m) {
        if ((rc = k0.alloc(m * 8)) || (rc = k1.alloc(m * 8)) || (rc = p0.alloc(m * 4)) || (rc = p1.alloc(m * 4)) || (rc = gs.alloc(m * 4)) ||
            (rc = head.alloc(m)) || (rc = act.alloc(m)) || (rc = aidx2.alloc(m * 4)) || (rc = cnt.alloc(8))) return rc;
    }
    int rounds = 0;
    while (m) {
        if (++rounds > 40) return fail(FMGPU_ERR_INVALID, "suffix sorting did not converge");
        k_round_keys<<<grid_for(m), 256, 0, stream>>>(aidx.as<uint32_t>(), sa_out, rank.as<uint32_t>(), m, n, h, k0.as<uint64_t>(), p0.as<uint32_t>());
        hipcub::DoubleBuffer<uint64_t> dk(k0.as<uint64_t>(), k1.as<uint64_t>());
        hipcub::DoubleBuffer<uint32_t> dv(p0.as<uint32_t>(), p1.as<uint32_t>());
        rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceRadixSort::SortPairs(t, bytes, dk, dv, (size_t)m, 0, 64, stream); });
        if (rc) return rc;
        k_heads<<<grid_for(m), 256, 0, stream>>>(dk.Current(), m, aidx.as<uint32_t>(), gs.as<uint32_t>(), head.as<uint8_t>());
        rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceScan::InclusiveScan(t, bytes, gs.as<uint32_t>(), gs.as<uint32_t>(), MaxOp{}, (size_t)m, stream); });
        if (rc) return rc;
        k_apply_groups<<<grid_for(m), 256, 0, stream>>>(dv.Current(), gs.as<uint32_t>(), head.as<uint8_t>(), m, rank.as<uint32_t>(), act.as<uint8_t>(),
                                                         sa_out, aidx.as<uint32_t>());
        rc = cub_call(tmp, [&](void* t, size_t& bytes) {
            return hipcub::DeviceSelect::Flagged(t, bytes, aidx.as<uint32_t>(), act.as<uint8_t>(), aidx2.as<uint32_t>(), cnt.as<uint64_t>(), (size_t)m, stream); });
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
__global__ __launch_bounds__(256) void k_bwt(const uint8_t* __restrict__ text, const uint32_t* __restrict__ sa, uint64_t n, uint8_t* __restrict__ bwt) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t p = sa[i];
    bwt[i] = text[p ? p - 1 : n - 1];
}
__global__ __launch_bounds__(256) void k_histogram(const uint8_t* __restrict__ s, uint64_t n, unsigned long long* __restrict__ hist) {
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) atomicAdd(&h[s[i]], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}
// one wave per 64-row block: bits by ballot, per-block counts into cnt[c * nblocks + B]
__global__ __launch_bounds__(256) void k_blocks_bits(const uint8_t* __restrict__ bwt, uint64_t n, uint64_t nblocks, uint32_t sigma, uint32_t bstride,
                                                     uint8_t* __restrict__ blk, uint32_t* __restrict__ cnt) {
    uint64_t B = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t lane = threadIdx.x & 63u;
    if (B >= nblocks) return;
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
            cnt[(uint64_t)c * nblocks + B] = (uint32_t)__popcll(mine);
        }
    }
}
__global__ __launch_bounds__(256) void k_blocks_counts(const uint32_t* __restrict__ cnt, uint64_t nblocks, uint32_t sigma, uint32_t bstride,
                                                       const idx_t* __restrict__ C, uint8_t* __restrict__ blk) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblocks * sigma) return;
    uint64_t B = t % nblocks; uint32_t c = (uint32_t)(t / nblocks);
    *reinterpret_cast<uint32_t*>(blk + B * bstride + 12ull * c) = cnt[t] + C[c];
}

// ------------------------------------------------------------------ sampled suffix array (reference layout)
__device__ __forceinline__ void seq_of(const uint64_t* __restrict__ sstart, uint64_t nseq, uint64_t p, uint64_t& s, uint64_t& o) {
    uint64_t lo = 0, hi = nseq;
    while (hi - lo > 1) { uint64_t mid = (lo + hi) >> 1; if (sstart[mid] <= p) lo = mid; else hi = mid; }
    s = lo; o = p - sstart[lo];
}
__global__ __launch_bounds__(256) void k_sa_bits(const uint32_t* __restrict__ sa, uint64_t n, const uint64_t* __restrict__ sstart, uint64_t nseq, uint64_t rate,
                                                 uint64_t* __restrict__ bits, uint32_t* __restrict__ blockcnt) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool has = false;
    if (i < n) { uint64_t s, o; seq_of(sstart, nseq, sa[i], s, o); has = o % rate == 0; }
    uint64_t w = __ballot(has);
    if ((threadIdx.x & 63u) == 0 && i < n) {
        bits[i >> 6] = w;
        atomicAdd(&blockcnt[i >> 9], (uint32_t)__popcll(w));
    }
}
__global__ __launch_bounds__(256) void k_sa_levels(const uint32_t* __restrict__ g, uint64_t nl1, uint64_t nl0, uint64_t* __restrict__ l0, uint16_t* __restrict__ l1) {
    uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nl1) { uint32_t base = g[(b >> 7) << 7]; l1[b] = (uint16_t)(g[b] - base); }
    if (b < nl0) l0[b] = g[b << 7];
}
__global__ __launch_bounds__(256) void k_sa_values(const uint32_t* __restrict__ sa, uint64_t n, const uint64_t* __restrict__ sstart, uint64_t nseq, uint64_t rate,
                                                   const uint64_t* __restrict__ bits, const uint32_t* __restrict__ g,
                                                   unsigned long long* __restrict__ f0, unsigned long long* __restrict__ f1,
                                                   uint32_t w0, uint32_t w1, uint64_t d0, uint64_t d1) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!((bits[i >> 6] >> (i & 63u)) & 1ull)) return;
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

struct Built {   // host copies of construction by-products (keep_host)
    std::vector<std::vector<uint8_t>> part;
};

static uint32_t bit_width64(uint64_t v) { uint32_t r = 0; while (v) { ++r; v >>= 1; } return r; }

static int make_format_a(const uint8_t* bwt, uint64_t n, uint32_t sigma, const idx_t* dC, DevString& s, int layout, hipStream_t stream) {
    uint64_t nblocks = n / 64 + 1;
    uint32_t bstride = sigma <= 5 ? 64u : 12u * sigma;
    s.layout = layout; s.sigma = (int)sigma; s.n = n; s.family = FAM_A;
    s.bitct = (int)bit_width64((uint64_t)sigma - 1);
    s.blk_bytes = nblocks * bstride + 64;
    FM_HIP(hipMalloc(&s.blk, s.blk_bytes));
    FM_HIP(hipMemsetAsync(s.blk, 0, s.blk_bytes, stream));
    DBuf cnt; int rc;
    if ((rc = cnt.alloc(nblocks * sigma * 4))) return rc;
    k_blocks_bits<<<dim3((unsigned)((nblocks * 64 + 255) / 256)), 256, 0, stream>>>(bwt, n, nblocks, sigma, bstride, (uint8_t*)s.blk, cnt.as<uint32_t>());
    Temp tmp;
    for (uint32_t c = 0; c < sigma; ++c) {
        uint32_t* p = cnt.as<uint32_t>() + (uint64_t)c * nblocks;
        rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceScan::ExclusiveSum(t, bytes, p, p, (size_t)nblocks, stream); });
        if (rc) return rc;
    }
    k_blocks_counts<<<grid_for(nblocks * sigma), 256, 0, stream>>>(cnt.as<uint32_t>(), nblocks, sigma, bstride, dC, (uint8_t*)s.blk);
    FM_HIP(hipStreamSynchronize(stream));
    s.va = ViewA{(const uint8_t*)s.blk, bstride, sigma, dC};
    return 0;
}

// ------------------------------------------------------------------ Format W from the BWT (string/Wavelet.h:40-72 restated as bulk passes)
// Level b of the reference's tree sees the symbols stably ordered by their top b bits (each push_back appends to the node of the
// symbol's prefix, Wavelet.h:56-65): one stable radix sort on those bits yields every node of the level as a contiguous slice.
__global__ __launch_bounds__(256) void k_prefix_hist(const uint8_t* __restrict__ sym, uint64_t n, uint32_t shift, unsigned int* __restrict__ hist) {
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) atomicAdd(&h[sym[i] >> shift], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// one thread per 64-bit payload word; the six words of a line are handled by six neighbouring threads
__global__ __launch_bounds__(256) void k_wavelet_words(const uint8_t* __restrict__ sorted, uint32_t bit, const uint32_t* __restrict__ node_base,
                                                       const uint32_t* __restrict__ node_start, const uint32_t* __restrict__ node_len, uint32_t first_node, uint32_t nnodes_level,
                                                       uint64_t first_line, uint64_t nlines, uint64_t* __restrict__ lines, uint32_t* __restrict__ line_ones) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nlines * 6) return;
    uint64_t L = first_line + t / 6; uint32_t j = (uint32_t)(t % 6);
    uint32_t lo = 0, hi = nnodes_level;                 // last node of the level with node_base <= L
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (node_base[first_node + mid] <= L) lo = mid; else hi = mid; }
    uint32_t id = first_node + lo;
    uint64_t p0 = (L - node_base[id]) * 384 + (uint64_t)j * 64, len = node_len[id];
    uint64_t w = 0;
    if (p0 < len) {
        uint32_t cnt = (uint32_t)(len - p0 < 64 ? len - p0 : 64);
        const uint8_t* src = sorted + node_start[id] + p0;
        for (uint32_t k = 0; k < cnt; ++k) w |= (uint64_t)((src[k] >> bit) & 1u) << k;
    }
    lines[L * 8 + 2 + j] = w;
    atomicAdd(&line_ones[L - first_line], (uint32_t)__popcll(w));
}

__global__ __launch_bounds__(256) void k_wavelet_headers(const uint32_t* __restrict__ scan, const uint32_t* __restrict__ node_base, uint32_t first_node, uint32_t nnodes_level,
                                                         uint64_t first_line, uint64_t nlines, uint64_t* __restrict__ lines) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nlines) return;
    uint64_t L = first_line + t;
    uint32_t lo = 0, hi = nnodes_level;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (node_base[first_node + mid] <= L) lo = mid; else hi = mid; }
    uint32_t id = first_node + lo;
    uint64_t* Lp = lines + L * 8;
    Lp[0] = scan[t] - scan[node_base[id] - first_line];
    uint64_t cum = 0, h1 = 0;
    for (uint32_t j = 0; j < 6; ++j) {
        if (j) h1 |= cum << (9 * (j - 1));
        cum += (uint64_t)__popcll(Lp[2 + j]);
    }
    Lp[1] = h1;
}

static int make_format_w(const uint8_t* bwt, uint64_t n, uint32_t sigma, const idx_t* dC, DevString& s, int layout, hipStream_t stream) {
    const uint32_t bits = bit_width64((uint64_t)sigma - 1) ? bit_width64((uint64_t)sigma - 1) : 1;    // Wavelet.h:26  bitct = bit_width(Sigma-1)
    const uint32_t nnodes = 1u << bits;                                                                  // Wavelet.h:27  bvct  = bit_ceil(Sigma); the last id stays empty
    s.layout = layout; s.sigma = (int)sigma; s.n = n; s.family = FAM_WAVELET; s.bitct = (int)bits;
    std::vector<uint32_t> base(nnodes, 0), start(nnodes, 0), len(nnodes, 0);
    DBuf hist, sorted; Temp tmp; int rc;
    if ((rc = hist.alloc(256 * 4)) || (rc = sorted.alloc(n + 64))) return rc;
    // pass 1: node lengths per level -> line offsets
    uint64_t total_lines = 0;
    for (uint32_t b = 0; b < bits; ++b) {
        unsigned int hh[256];
        FM_HIP(hipMemsetAsync(hist.p, 0, 256 * 4, stream));
        k_prefix_hist<<<dim3(2048), 256, 0, stream>>>(bwt, n, bits - b, hist.as<unsigned int>());
        FM_HIP(hipMemcpy(hh, hist.p, 256 * 4, hipMemcpyDeviceToHost));
        uint64_t acc = 0;
        for (uint32_t pfx = 0; pfx < (1u << b); ++pfx) {
            uint32_t id = ((1u << b) - 1u) + pfx;
            start[id] = (uint32_t)acc; len[id] = hh[pfx]; acc += hh[pfx];
            base[id] = (uint32_t)total_lines;
            total_lines += (uint64_t)hh[pfx] / 384 + 1;
        }
    }
    base[nnodes - 1] = (uint32_t)total_lines; total_lines += 1;     // the unused last id: one empty line
    if (total_lines >= 0xffffffffull) return fail(FMGPU_ERR_UNSUPPORTED, "wavelet too large for 32-bit line offsets");
    s.blk_bytes = total_lines * 64; s.aux_bytes = nnodes * 4;
    FM_HIP(hipMalloc(&s.blk, s.blk_bytes));
    FM_HIP(hipMemsetAsync(s.blk, 0, s.blk_bytes, stream));
    FM_HIP(hipMalloc(&s.aux, s.aux_bytes));
    FM_HIP(hipMemcpy(s.aux, base.data(), nnodes * 4, hipMemcpyHostToDevice));
    DBuf dstart, dlen, ones;
    uint64_t max_level_lines = n / 384 + nnodes + 1;
    if ((rc = dstart.alloc(nnodes * 4)) || (rc = dlen.alloc(nnodes * 4)) || (rc = ones.alloc((max_level_lines + 1) * 4))) return rc;
    FM_HIP(hipMemcpy(dstart.p, start.data(), nnodes * 4, hipMemcpyHostToDevice));
    FM_HIP(hipMemcpy(dlen.p, len.data(), nnodes * 4, hipMemcpyHostToDevice));
    // pass 2: per level, sort by the top b bits, cut the slices into lines
    for (uint32_t b = 0; b < bits; ++b) {
        const uint8_t* src = bwt;
        if (b > 0) {
            rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceRadixSort::SortKeys(t, bytes, bwt, sorted.as<uint8_t>(), (size_t)n, (int)(bits - b), (int)bits, stream); });
            if (rc) return rc;
            src = sorted.as<uint8_t>();
        }
        uint32_t first_node = (1u << b) - 1u, nl = 1u << b;
        uint64_t first_line = base[first_node];
        uint64_t last = first_node + nl - 1;
        uint64_t nlines = (uint64_t)base[last] + len[last] / 384 + 1 - first_line;
        FM_HIP(hipMemsetAsync(ones.p, 0, (nlines + 1) * 4, stream));
        k_wavelet_words<<<grid_for(nlines * 6), 256, 0, stream>>>(src, bits - 1 - b, (const uint32_t*)s.aux, dstart.as<uint32_t>(), dlen.as<uint32_t>(), first_node, nl,
                                                                   first_line, nlines, (uint64_t*)s.blk, ones.as<uint32_t>());
        rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceScan::ExclusiveSum(t, bytes, ones.as<uint32_t>(), ones.as<uint32_t>(), (size_t)(nlines + 1), stream); });
        if (rc) return rc;
        k_wavelet_headers<<<grid_for(nlines), 256, 0, stream>>>(ones.as<uint32_t>(), (const uint32_t*)s.aux, first_node, nl, first_line, nlines, (uint64_t*)s.blk);
    }
    FM_HIP(hipStreamSynchronize(stream));
    s.vw = ViewW{(const uint64_t*)s.blk, (const uint32_t*)s.aux, dC, sigma, bits};
    return 0;
}

template <class Occ>
__global__ __launch_bounds__(256) void k_symbols(Occ occ, uint64_t n, uint8_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint8_t)occ.symbol((idx_t)i);
}

int build_format_a_shadow(DevString& s, const idx_t* dC, hipStream_t stream) {
    if (s.family == FAM_A || s.shadow || s.n == 0) return 0;
    DBuf sym; int rc;
    if ((rc = sym.alloc(s.n + 64))) return rc;
    switch (s.family) {
    case FAM_EPR:   k_symbols<<<grid_for(s.n), 256, 0, stream>>>(OccR<false>{s.vr}, s.n, sym.as<uint8_t>()); break;
    case FAM_EPRV2: k_symbols<<<grid_for(s.n), 256, 0, stream>>>(OccR<true>{s.vr}, s.n, sym.as<uint8_t>()); break;
    default:        k_symbols<<<grid_for(s.n), 256, 0, stream>>>(OccW{s.vw}, s.n, sym.as<uint8_t>()); break;
    }
    DevString t;
    if ((rc = make_format_a(sym.as<uint8_t>(), s.n, (uint32_t)s.sigma, dC, t, s.layout, stream))) { if (t.blk) (void)hipFree(t.blk); return rc; }
    s.shadow = t.blk; s.shadow_bytes = t.blk_bytes; s.va = t.va;
    return 0;
}

static int make_string(const uint8_t* bwt, uint64_t n, uint32_t sigma, const idx_t* dC, DevString& s, int layout, hipStream_t stream) {
    return layout == FMGPU_WAVELET ? make_format_w(bwt, n, sigma, dC, s, layout, stream) : make_format_a(bwt, n, sigma, dC, s, layout, stream);
}

}  // namespace fmgpu

using namespace fmgpu;

extern "C" {

int fmgpu_built_free(fmgpu_built_t b) { delete reinterpret_cast<Built*>(b); return 0; }

int fmgpu_built_get(fmgpu_built_t b_, int32_t part, const void** ptr, uint64_t* bytes) {
    Built* b = reinterpret_cast<Built*>(b_);
    if (!b || !ptr || !bytes) return fail(FMGPU_ERR_INVALID, "null argument");
    if (part < 0 || (size_t)part >= b->part.size()) return fail(FMGPU_ERR_INVALID, "no such part");
    *ptr = b->part[part].data(); *bytes = b->part[part].size();
    return 0;
}

int fmgpu_build_index(const uint8_t* seqs, const uint64_t* seq_off, uint64_t nseq, int32_t sigma, int32_t layout, uint64_t sampling_rate,
                      int32_t bidirectional, int32_t keep_host, fmgpu_index_t* out, fmgpu_built_t* built_out) {
    if (!out) return fail(FMGPU_ERR_INVALID, "out is null");
    *out = nullptr;
    if (built_out) *built_out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { (void)hipGetLastError(); return fail(FMGPU_ERR_NO_DEVICE, "no HIP device visible — the product path has no CPU fallback"); }
    if (!seqs || !seq_off || nseq == 0) return fail(FMGPU_ERR_INVALID, "seqs / seq_off is null or nseq == 0");
    if (sigma < 2 || sigma > 256) return fail(FMGPU_ERR_INVALID, "sigma must be in [2, 256]");
    if (sampling_rate == 0) return fail(FMGPU_ERR_INVALID, "sampling_rate must be >= 1");
    // the layout names the reference type the caller replaces; on the device every blocked layout is held as the LF-ready block
    // table (Format A) and Wavelet as wavelet lines (Format W) — the answers of a String_c do not depend on its layout
    if (layout < FMGPU_IB8 || layout > FMGPU_FBV_2048_64K) return fail(FMGPU_ERR_INVALID, "unknown layout id");
    hipStream_t stream = nullptr;
    Staged soff, sseq;
    int rc;
    if ((rc = soff.in(seq_off, (nseq + 1) * 8, stream))) return rc;
    std::vector<uint64_t> hoff(nseq + 1);
    FM_HIP(hipMemcpy(hoff.data(), soff.dev, (nseq + 1) * 8, hipMemcpyDeviceToHost));
    for (uint64_t s = 0; s < nseq; ++s) if (hoff[s + 1] < hoff[s]) return fail(FMGPU_ERR_INVALID, "seq_off is not non-decreasing");
    const uint64_t total = hoff[nseq] - hoff[0], n = total + nseq;
    if (n >= 0xffffffffull - 64) return fail(FMGPU_ERR_UNSUPPORTED, "this build indexes fewer than 2^32 - 64 rows");
    if ((rc = sseq.in(seqs, hoff[nseq], stream))) return rc;

    std::unique_ptr<Index> x(new (std::nothrow) Index());
    std::unique_ptr<Built> built(keep_host ? new (std::nothrow) Built() : nullptr);
    if (!x || (keep_host && !built)) return fail(FMGPU_ERR_NOMEM, "host allocation");
    if (built) built->part.resize(9);
    auto bail = [&](int code) { fmgpu_index_destroy(reinterpret_cast<fmgpu_index_t>(x.release())); return code; };
    (void)hipGetDevice(&x->device);

    DBuf text, sa, bwt;
    if ((rc = text.alloc(n)) || (rc = sa.alloc(n * 4)) || (rc = bwt.alloc(n))) return bail(rc);
    k_assemble_text<<<grid_for(n), 256, 0, stream>>>((const uint8_t*)sseq.dev, (const uint64_t*)soff.dev, nseq, text.as<uint8_t>(), n);
    {
        DBuf bad; if ((rc = bad.alloc(4))) return bail(rc);
        (void)hipMemsetAsync(bad.p, 0, 4, stream);
        k_check_symbols<<<grid_for(n), 256, 0, stream>>>(text.as<uint8_t>(), n, (uint32_t)sigma, bad.as<unsigned int>());
        unsigned int hb = 0;
        hipError_t e = hipMemcpy(&hb, bad.p, 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return bail(hip_fail(e, "k_check_symbols"));
        if (hb) return bail(fail(FMGPU_ERR_INVALID, "a sequence holds a symbol >= sigma"));
    }
    if ((rc = build_suffix_array(text.as<uint8_t>(), n, (uint32_t)sigma, sa.as<uint32_t>(), stream))) return bail(rc);
    k_bwt<<<grid_for(n), 256, 0, stream>>>(text.as<uint8_t>(), sa.as<uint32_t>(), n, bwt.as<uint8_t>());

    // C[c] = #symbols < c  (utils.h:199-206)
    {
        DBuf hist; if ((rc = hist.alloc(256 * 8))) return bail(rc);
        (void)hipMemsetAsync(hist.p, 0, 256 * 8, stream);
        k_histogram<<<dim3(2048), 256, 0, stream>>>(bwt.as<uint8_t>(), n, hist.as<unsigned long long>());
        unsigned long long hh[256];
        hipError_t e = hipMemcpy(hh, hist.p, 256 * 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return bail(hip_fail(e, "histogram"));
        uint64_t acc = 0;
        std::vector<idx_t> c32(sigma + 1);
        for (int c = 0; c <= sigma; ++c) { x->hC[c] = acc; c32[c] = (idx_t)acc; if (c < sigma) acc += hh[c]; }
        hipError_t e2 = hipMalloc((void**)&x->dC, (sigma + 1) * sizeof(idx_t));
        if (e2 != hipSuccess) return bail(hip_fail(e2, "hipMalloc(C)"));
        e2 = hipMemcpy(x->dC, c32.data(), (sigma + 1) * sizeof(idx_t), hipMemcpyHostToDevice);
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

    // ---- sampled suffix array in the reference's SparseArray layout (suffixarray/SparseArray.h:31-76)
    {
        std::vector<uint64_t> sstart(nseq + 1);
        uint64_t maxlen = 0;
        for (uint64_t s = 0; s <= nseq; ++s) sstart[s] = hoff[s] - hoff[0] + s;
        for (uint64_t s = 0; s < nseq; ++s) maxlen = std::max(maxlen, hoff[s + 1] - hoff[s]);     // positions 0..len (delimiter slot included)
        // DenseMultiVector ctor (DenseMultiVector.h:65-103): largest value and gcd per field
        uint64_t largest0 = nseq - 1, div0 = nseq >= 2 ? 1 : 0;
        uint64_t largest1 = (maxlen / sampling_rate) * sampling_rate, div1 = largest1 ? sampling_rate : 0;
        if (div0 == 0) div0 = 1; if (largest0 == 0) largest0 = 1;
        if (div1 == 0) div1 = 1; if (largest1 == 0) largest1 = 1;
        uint32_t w0 = bit_width64(largest0 / div0), w1 = bit_width64(largest1 / div1);
        uint64_t nl0 = n / 65536 + 1, nl1 = n / 512 + 1, nwords = nl1 * 8;
        DBuf dstart, g;
        if ((rc = dstart.alloc((nseq + 1) * 8)) || (rc = g.alloc((nl1 + 1) * 4))) return bail(rc);
        hipError_t e = hipMemcpy(dstart.p, sstart.data(), (nseq + 1) * 8, hipMemcpyHostToDevice);
        if (e != hipSuccess) return bail(hip_fail(e, "copy sstart"));
        if ((e = hipMalloc(&x->sa_bits, nwords * 8)) != hipSuccess || (e = hipMalloc(&x->sa_l0, nl0 * 8)) != hipSuccess || (e = hipMalloc(&x->sa_l1, nl1 * 2)) != hipSuccess)
            return bail(hip_fail(e, "hipMalloc(sa)"));
        (void)hipMemsetAsync(x->sa_bits, 0, nwords * 8, stream);
        (void)hipMemsetAsync(g.p, 0, (nl1 + 1) * 4, stream);
        k_sa_bits<<<grid_for((n + 63) / 64 * 64), 256, 0, stream>>>(sa.as<uint32_t>(), n, dstart.as<uint64_t>(), nseq, sampling_rate, (uint64_t*)x->sa_bits, g.as<uint32_t>());
        Temp tmp;
        rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceScan::ExclusiveSum(t, bytes, g.as<uint32_t>(), g.as<uint32_t>(), (size_t)(nl1 + 1), stream); });
        if (rc) return bail(rc);
        uint32_t nvalues = 0;
        e = hipMemcpy(&nvalues, g.as<uint32_t>() + nl1, 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return bail(hip_fail(e, "sample count"));
        k_sa_levels<<<grid_for(nl1), 256, 0, stream>>>(g.as<uint32_t>(), nl1, nl0, (uint64_t*)x->sa_l0, (uint16_t*)x->sa_l1);
        uint64_t f0words = ((uint64_t)nvalues * w0 + 63) / 64, f1words = ((uint64_t)nvalues * w1 + 63) / 64;
        if ((e = hipMalloc(&x->sa_f0, (f0words + 1) * 8)) != hipSuccess || (e = hipMalloc(&x->sa_f1, (f1words + 1) * 8)) != hipSuccess) return bail(hip_fail(e, "hipMalloc(sa fields)"));
        (void)hipMemsetAsync(x->sa_f0, 0, (f0words + 1) * 8, stream);
        (void)hipMemsetAsync(x->sa_f1, 0, (f1words + 1) * 8, stream);
        k_sa_values<<<grid_for(n), 256, 0, stream>>>(sa.as<uint32_t>(), n, dstart.as<uint64_t>(), nseq, sampling_rate, (const uint64_t*)x->sa_bits, g.as<uint32_t>(),
                                                     (unsigned long long*)x->sa_f0, (unsigned long long*)x->sa_f1, w0, w1, div0, div1);
        e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return bail(hip_fail(e, "sampled suffix array kernels"));
        x->vsa = ViewSA{(const uint64_t*)x->sa_l0, (const uint16_t*)x->sa_l1, (const uint64_t*)x->sa_bits, (const uint64_t*)x->sa_f0, (const uint64_t*)x->sa_f1, w0, w1, div0, div1};
        x->has_sa = true;
        x->device_bytes += nwords * 8 + nl0 * 8 + nl1 * 2 + (f0words + f1words + 2) * 8;
        if (built) {
            auto grab = [&](int part, const void* dev, size_t bytes) -> int {
                built->part[part].resize(bytes);
                if (bytes) FM_HIP(hipMemcpy(built->part[part].data(), dev, bytes, hipMemcpyDeviceToHost));
                return 0;
            };
            if ((rc = grab(3, x->sa_l0, nl0 * 8)) || (rc = grab(4, x->sa_l1, nl1 * 2)) || (rc = grab(5, x->sa_bits, nwords * 8)) ||
                (rc = grab(6, x->sa_f0, f0words * 8)) || (rc = grab(7, x->sa_f1, f1words * 8))) return bail(rc);
            uint64_t params[8] = {(uint64_t)nvalues * w0, w0, largest0, div0, (uint64_t)nvalues * w1, w1, largest1, div1};
            built->part[8].resize(sizeof params);
            std::memcpy(built->part[8].data(), params, sizeof params);
        }
    }
    if (bidirectional) {
        // BiFMIndex.h:78-92: reverse the whole concatenation (delimiters included), second suffix sort
        k_reverse<<<grid_for(n), 256, 0, stream>>>(text.as<uint8_t>(), bwt.as<uint8_t>(), n);
        FM_HIP(hipMemcpyAsync(text.p, bwt.p, n, hipMemcpyDeviceToDevice, stream));
        if ((rc = build_suffix_array(text.as<uint8_t>(), n, (uint32_t)sigma, sa.as<uint32_t>(), stream))) return bail(rc);
        k_bwt<<<grid_for(n), 256, 0, stream>>>(text.as<uint8_t>(), sa.as<uint32_t>(), n, bwt.as<uint8_t>());
        if ((rc = make_string(bwt.as<uint8_t>(), n, (uint32_t)sigma, x->dC, x->rev, layout, stream))) return bail(rc);
        x->bidirectional = true;
        if (built) {
            built->part[1].resize(n);
            hipError_t e = hipMemcpy(built->part[1].data(), bwt.p, n, hipMemcpyDeviceToHost);
            if (e != hipSuccess) return bail(hip_fail(e, "copy bwt_rev"));
        }
    }
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return bail(hip_fail(e, "index construction"));
    text.release(); sa.release(); bwt.release();
    if ((rc = build_lf_table(x->bwt, stream))) return bail(rc);
    if (x->bidirectional && (rc = build_lf_table(x->rev, stream))) return bail(rc);
    x->device_bytes += x->bwt.blk_bytes + x->rev.blk_bytes + (x->bwt.lf_table ? n * sizeof(idx_t) : 0) + (x->rev.lf_table ? n * sizeof(idx_t) : 0);
    *out = reinterpret_cast<fmgpu_index_t>(x.release());
    if (built_out) *built_out = reinterpret_cast<fmgpu_built_t>(built.release());
    return 0;
}

}  // extern "C"

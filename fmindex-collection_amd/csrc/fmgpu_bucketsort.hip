// fmgpu_bucketsort.hip — suffix sorting for texts whose suffix array does not fit beside them.  Compiled once per row width.
//
// The sorter of fmgpu_build.hip (radix sort of every suffix's K-symbol prefix, prefix doubling on the ties) holds the text, the suffix array, the rank array, two key
// buffers and a value buffer: 30 bytes per row with 32-bit suffix indices, 42 with 64-bit ones — ~6 x 10^9 rows in 288 GB.  The reference switches to libsais64 there
// (utils.h:243-247) and has the host's memory to do it in.  What construction needs of the suffix array is its ORDER, once, front to back: bwt[i] = text[sa[i] - 1]
// (utils.h:145-163) and the sampled entries (FMIndex.h:79-101).  So this sorter never holds the array: it cuts the suffixes into buckets by their first symbols
// (a bucket = a range of rows), sorts one bucket at a time and hands its rows to the caller, in row order:
//   1. one pass over the text counts the suffixes per bin (the top kBinBits bits of the packed K-symbol key); the host cuts the bins into buckets of <= bucket_rows rows;
//   2. per bucket: a pass over the text collects (key, position) of its suffixes; one radix sort orders them by their K-symbol prefix;
//   3. ties only: rows whose K-prefix is shared are compacted and re-sorted by (group, the next symbols of the suffix) — as many symbols as fit 64 bits beside the
//      dense group number — until every group is a single row.  Not doubling (there is no rank array to double with): a repeat of length L costs L / symbols-per-round
//      rounds over ITS rows, which is nothing for a protein database or a text without long exact repeats, and would be hours for megabase runs of one symbol —
//      the work is bounded (kMaxRefineWork passes over a bucket's rows) and such a text is refused with an error instead (the doubling sorter handles it, up to its size);
//   4. the bucket's positions, now in suffix order, go to the caller's sink (BWT symbols, sampled suffix array entries).
// Memory: the text + ~40 bytes per row of ONE bucket (+ the tie buffers), whatever n is.
#include "fmgpu_common.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

namespace FMGPU_NS {

namespace {

constexpr uint32_t kBinBits = 22;               // bins of the bucket histogram: the top bits of the K-symbol key
constexpr uint32_t kPerThread = 16;             // consecutive text positions a thread of the two text passes handles (one rolling key)
constexpr uint64_t kMaxRefineWork = 96;         // tie rounds may re-sort this many times a bucket's rows in total

struct Temp {
    DBuf buf;
    int ensure(size_t bytes) { if (bytes > buf.bytes) return buf.alloc(bytes); return 0; }
};
template <class F>
int cub_call(Temp& tmp, F&& f) {
    size_t bytes = 0;
    FM_HIP(f(nullptr, bytes));
    int rc = tmp.ensure(bytes); if (rc) return rc;
    bytes = tmp.buf.bytes;
    FM_HIP(f(tmp.buf.p, bytes));
    return 0;
}
inline dim3 grid_for(uint64_t threads) {
    uint64_t b = (threads + 255) / 256;
    return dim3((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(b, 1u << 22)));
}
uint32_t bit_width64(uint64_t v) { uint32_t r = 0; while (v) { ++r; v >>= 1; } return r; }

// field of text position i in a packed key: symbol + 1, 0 = past the end (a proper prefix sorts first)
__device__ __forceinline__ uint64_t key_field(const uint8_t* __restrict__ t, uint64_t n, uint64_t i) { return i < n ? (uint64_t)t[i] + 1ull : 0ull; }
__device__ __forceinline__ uint64_t pack_key(const uint8_t* __restrict__ t, uint64_t n, uint64_t i, uint32_t K, uint32_t b) {
    uint64_t k = 0;
    for (uint32_t j = 0; j < K; ++j) k = (k << b) | key_field(t, n, i + j);
    return k;
}
// the K-symbol keys of kPerThread consecutive positions starting at `first`, by a rolling window; f(position, key)
template <class F>
__device__ __forceinline__ void rolling_keys(const uint8_t* __restrict__ t, uint64_t n, uint64_t first, uint32_t K, uint32_t b, F&& f) {
    if (first >= n) return;
    const uint64_t mask = K * b >= 64u ? ~0ull : (1ull << (K * b)) - 1ull;
    uint64_t k = pack_key(t, n, first, K, b);
    for (uint32_t s = 0; s < kPerThread; ++s) {
        const uint64_t i = first + s;
        if (i >= n) break;
        f(i, k);
        k = ((k << b) | key_field(t, n, i + K)) & mask;
    }
}

__global__ __launch_bounds__(256) void k_bin_histogram(const uint8_t* __restrict__ t, uint64_t n, uint32_t K, uint32_t b, uint32_t shift, unsigned long long* __restrict__ hist) {
    const uint64_t nthreads = (n + kPerThread - 1) / kPerThread;
    for (uint64_t th = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; th < nthreads; th += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t run_bin = ~0ull; uint32_t run = 0;                 // consecutive positions of one bin (a run of one symbol) cost one atomic
        rolling_keys(t, n, th * kPerThread, K, b, [&](uint64_t, uint64_t k) {
            const uint64_t bin = k >> shift;
            if (bin == run_bin) ++run;
            else { if (run) atomicAdd(&hist[run_bin], (unsigned long long)run); run_bin = bin; run = 1; }
        });
        if (run) atomicAdd(&hist[run_bin], (unsigned long long)run);
    }
}

// (key, position) of every suffix whose bin lies in [bin_lo, bin_hi), in no particular order: the sort that follows orders them, and ties are broken by text alone
__global__ __launch_bounds__(256) void k_collect(const uint8_t* __restrict__ t, uint64_t n, uint32_t K, uint32_t b, uint32_t shift, uint64_t bin_lo, uint64_t bin_hi,
                                                 uint64_t* __restrict__ keys, idx_t* __restrict__ pos, unsigned long long* __restrict__ cursor, uint64_t cap) {
    const uint64_t nthreads = (n + kPerThread - 1) / kPerThread;
    const uint64_t rounds = (nthreads + (uint64_t)gridDim.x * blockDim.x - 1) / ((uint64_t)gridDim.x * blockDim.x);
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t r = 0; r < rounds; ++r) {                         // (every lane of a wave takes part in every round: shuffles below)
        const uint64_t th = r * gridDim.x * blockDim.x + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        uint32_t mine = 0;
        if (th < nthreads) rolling_keys(t, n, th * kPerThread, K, b, [&](uint64_t, uint64_t k) { const uint64_t bin = k >> shift; mine += (bin >= bin_lo && bin < bin_hi) ? 1u : 0u; });
        uint32_t x = mine;                                          // inclusive scan over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off, 64); if (lane >= (uint32_t)off) x += y; }
        const uint32_t total = __shfl(x, 63, 64);
        if (total == 0) continue;
        unsigned long long base = 0;
        if (lane == 63) base = atomicAdd(cursor, (unsigned long long)total);
        base = ((unsigned long long)__shfl((uint32_t)(base >> 32), 63, 64) << 32) | __shfl((uint32_t)base, 63, 64);
        uint64_t at = base + (x - mine);
        if (mine) rolling_keys(t, n, th * kPerThread, K, b, [&](uint64_t i, uint64_t k) {
            const uint64_t bin = k >> shift;
            if (bin >= bin_lo && bin < bin_hi) { if (at < cap) { keys[at] = k; pos[at] = (idx_t)i; } ++at; }
        });
    }
}

// after a sort by key: head[j] = row j starts a group of equal keys; act[j] = its group has more than one row
__global__ __launch_bounds__(256) void k_heads_active(const uint64_t* __restrict__ keys, uint64_t m, uint32_t* __restrict__ head, uint32_t* __restrict__ act) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k = keys[j];
        const bool hd = j == 0 || keys[j - 1] != k;
        const bool next_hd = j + 1 == m || keys[j + 1] != k;
        head[j] = hd ? 1u : 0u;
        act[j] = (hd && next_hd) ? 0u : 1u;
    }
}
// the rows with act[j] != 0, in order: their position, their row within the bucket, their head flag.  `at` = exclusive prefix sums of act.
// row_in == null: row j of the bucket itself (the first round); rowpos (if given): every row's position goes to its place in the bucket's order
__global__ __launch_bounds__(256) void k_compact(const idx_t* __restrict__ pos_in, const uint32_t* __restrict__ row_in, const uint32_t* __restrict__ head, const uint32_t* __restrict__ act,
                                                 const uint32_t* __restrict__ at, uint64_t m, idx_t* __restrict__ pos_out, uint32_t* __restrict__ row_out, uint32_t* __restrict__ head_out,
                                                 idx_t* __restrict__ rowpos) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x) {
        const idx_t p = pos_in[j];
        const uint32_t row = row_in ? row_in[j] : (uint32_t)j;
        if (rowpos) rowpos[row] = p;
        if (act[j]) { const uint32_t o = at[j]; pos_out[o] = p; row_out[o] = row; head_out[o] = head[j]; }
    }
}
// keys of a tie round: (dense group number, the next `nsym` symbols of the suffix from depth d on).  gid_incl = inclusive prefix sums of the head flags
__global__ __launch_bounds__(256) void k_round_keys(const uint8_t* __restrict__ t, uint64_t n, const idx_t* __restrict__ pos, const uint32_t* __restrict__ gid_incl, uint64_t m,
                                                    uint64_t d, uint32_t nsym, uint32_t b, uint64_t* __restrict__ keys) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x)
        keys[j] = ((uint64_t)(gid_incl[j] - 1u) << (nsym * b)) | pack_key(t, n, (uint64_t)pos[j] + d, nsym, b);
}

}  // namespace

// Sorts the suffixes of text[0, n) (plain byte order; a proper prefix sorts first) bucket by bucket and hands every bucket to `sink`, buckets in ascending row order:
// sink(first_row, pos, count, scratch, scratch_bytes) — pos[0..count) = the text positions of rows first_row .. first_row + count - 1; scratch is device memory the sink may use
// until it returns (8 bytes per row of the largest bucket).  bucket_rows = the most rows a bucket should hold (0: from the free device memory).
int sort_suffixes_bucketed(const uint8_t* text, uint64_t n, uint32_t sigma, uint64_t bucket_rows, const SuffixSink& sink, hipStream_t stream) {
    if (n == 0) return 0;
    uint32_t b = 0; while ((1u << b) <= sigma) ++b;                 // bits for the fields 0..sigma
    const uint32_t K = 64 / b, kbits = K * b;
    const uint32_t bin_bits = std::min(kBinBits, kbits), shift = kbits - bin_bits;
    const uint64_t nbins = 1ull << bin_bits;
    int rc;
    // ---- 1. suffixes per bin, bins -> buckets
    std::vector<unsigned long long> hist(nbins);
    {
        DBuf dh; if ((rc = dh.alloc(nbins * 8))) return rc;
        FM_HIP(hipMemsetAsync(dh.p, 0, nbins * 8, stream));
        k_bin_histogram<<<grid_for((n + kPerThread - 1) / kPerThread), 256, 0, stream>>>(text, n, K, b, shift, dh.as<unsigned long long>());
        FM_LAUNCHED("k_bin_histogram");
        FM_HIP(hipMemcpyAsync(hist.data(), dh.p, nbins * 8, hipMemcpyDeviceToHost, stream));
        FM_HIP(hipStreamSynchronize(stream));
    }
    const size_t row_bytes = 16 + 2 * sizeof(idx_t) + 12;           // key and position double buffers, head / active / scan words
    if (bucket_rows == 0) {
        size_t free_b = 0, total_b = 0;
        FM_HIP(hipMemGetInfo(&free_b, &total_b));
        bucket_rows = std::max<uint64_t>(1u << 20, (uint64_t)((double)free_b * 0.55 / (double)row_bytes));     // (the rest: tie buffers, the radix sort's scratch)
    }
    bucket_rows = std::min<uint64_t>(bucket_rows, 0xfff00000ull);   // rows within a bucket are 32-bit numbers
    struct Bucket { uint64_t bin_lo, bin_hi, rows; };
    std::vector<Bucket> buckets;
    uint64_t largest = 0;
    {
        Bucket cur{0, 0, 0};
        for (uint64_t bin = 0; bin < nbins; ++bin) {
            const uint64_t c = hist[bin];
            if (cur.rows && cur.rows + c > bucket_rows) { cur.bin_hi = bin; buckets.push_back(cur); cur = Bucket{bin, bin, 0}; }
            cur.rows += c;
        }
        cur.bin_hi = nbins;
        if (cur.rows) buckets.push_back(cur);
        uint64_t sum = 0;
        for (const Bucket& k : buckets) { largest = std::max(largest, k.rows); sum += k.rows; }
        if (sum != n) return fail(FMGPU_ERR_HIP, "bucket histogram does not add up to the text length");
        if (largest > 0xfff00000ull) return fail(FMGPU_ERR_UNSUPPORTED, "more than 2^32 suffixes share their first " + std::to_string(bin_bits / b) + " symbols: the bucketed suffix sorter cannot cut them apart");
    }
    // ---- buffers of one bucket
    DBuf k0, k1, p0, p1, head, act, at, cursor;
    if ((rc = k0.alloc(largest * 8)) || (rc = k1.alloc(largest * 8)) || (rc = p0.alloc(largest * sizeof(idx_t))) || (rc = p1.alloc(largest * sizeof(idx_t))) ||
        (rc = head.alloc(largest * 4)) || (rc = act.alloc(largest * 4)) || (rc = at.alloc((largest + 1) * 4)) || (rc = cursor.alloc(8))) return rc;
    Temp tmp;
    // tie buffers, grown on demand (a text without long repeats leaves ~0.1 % of its rows tied after the first sort)
    DBuf tk0, tk1, tp0, tp1, trow0, trow1, thead0, thead1, tact, tat, tgid;
    uint64_t tcap = 0;
    auto tie_buffers = [&](uint64_t m) -> int {
        if (m <= tcap) return 0;
        const uint64_t c = m + m / 8 + 1024;
        int r;
        if ((r = tk0.alloc(c * 8)) || (r = tk1.alloc(c * 8)) || (r = tp0.alloc(c * sizeof(idx_t))) || (r = tp1.alloc(c * sizeof(idx_t))) || (r = trow0.alloc(c * 4)) || (r = trow1.alloc(c * 4)) ||
            (r = thead0.alloc(c * 4)) || (r = thead1.alloc(c * 4)) || (r = tact.alloc(c * 4)) || (r = tat.alloc((c + 1) * 4)) || (r = tgid.alloc(c * 4))) return r;
        tcap = c;
        return 0;
    };
    uint64_t first_row = 0;
    for (const Bucket& bk : buckets) {
        const uint64_t m = bk.rows;
        // ---- 2. collect and sort by the K-symbol prefix
        FM_HIP(hipMemsetAsync(cursor.p, 0, 8, stream));
        k_collect<<<grid_for((n + kPerThread - 1) / kPerThread), 256, 0, stream>>>(text, n, K, b, shift, bk.bin_lo, bk.bin_hi, k0.as<uint64_t>(), p0.as<idx_t>(),
                                                                                     cursor.as<unsigned long long>(), m);
        FM_LAUNCHED("k_collect");
        unsigned long long got = 0;
        FM_HIP(hipMemcpyAsync(&got, cursor.p, 8, hipMemcpyDeviceToHost, stream));
        FM_HIP(hipStreamSynchronize(stream));
        if (got != m) return fail(FMGPU_ERR_HIP, "a bucket collected " + std::to_string(got) + " suffixes where the histogram counted " + std::to_string(m));
        hipcub::DoubleBuffer<uint64_t> dk(k0.as<uint64_t>(), k1.as<uint64_t>());
        hipcub::DoubleBuffer<idx_t> dv(p0.as<idx_t>(), p1.as<idx_t>());
        rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceRadixSort::SortPairs(t, bytes, dk, dv, (size_t)m, 0, (int)kbits, stream); });
        if (rc) return rc;
        idx_t* rowpos = dv.Current();                               // the bucket's order; tied rows are overwritten below
        // ---- 3. ties
        k_heads_active<<<grid_for(m), 256, 0, stream>>>(dk.Current(), m, head.as<uint32_t>(), act.as<uint32_t>());
        FM_LAUNCHED("k_heads_active");
        rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceScan::ExclusiveSum(t, bytes, act.as<uint32_t>(), at.as<uint32_t>(), (size_t)m, stream); });
        if (rc) return rc;
        uint32_t last_at = 0, last_act = 0;
        FM_HIP(hipMemcpyAsync(&last_at, at.as<uint32_t>() + (m - 1), 4, hipMemcpyDeviceToHost, stream));
        FM_HIP(hipMemcpyAsync(&last_act, act.as<uint32_t>() + (m - 1), 4, hipMemcpyDeviceToHost, stream));
        FM_HIP(hipStreamSynchronize(stream));
        uint64_t ma = (uint64_t)last_at + last_act;
        uint64_t work = 0, d = K;
        idx_t *cp = nullptr, *np = nullptr; uint32_t *crow = nullptr, *nrow = nullptr, *chead = nullptr, *nhead = nullptr;
        if (ma) {
            if ((rc = tie_buffers(ma))) return rc;
            cp = tp0.as<idx_t>(); np = tp1.as<idx_t>(); crow = trow0.as<uint32_t>(); nrow = trow1.as<uint32_t>(); chead = thead0.as<uint32_t>(); nhead = thead1.as<uint32_t>();
            k_compact<<<grid_for(m), 256, 0, stream>>>(rowpos, nullptr, head.as<uint32_t>(), act.as<uint32_t>(), at.as<uint32_t>(), m, cp, crow, chead, nullptr);
            FM_LAUNCHED("k_compact");
        }
        while (ma) {
            work += ma;
            if (work > kMaxRefineWork * std::max<uint64_t>(m, 1u << 20))
                return fail(FMGPU_ERR_UNSUPPORTED, "the text holds exact repeats too long for the bucketed suffix sorter (depth " + std::to_string(d) + " symbols reached with " +
                                                   std::to_string(ma) + " rows still tied): the doubling sorter handles such texts up to the size its buffers fit");
            const uint32_t gbits = std::max(1u, bit_width64(ma - 1));
            const uint32_t nsym = std::min(K, (64u - gbits) / b);
            rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceScan::InclusiveSum(t, bytes, chead, tgid.as<uint32_t>(), (size_t)ma, stream); });
            if (rc) return rc;
            k_round_keys<<<grid_for(ma), 256, 0, stream>>>(text, n, cp, tgid.as<uint32_t>(), ma, d, nsym, b, tk0.as<uint64_t>());
            FM_LAUNCHED("k_round_keys");
            hipcub::DoubleBuffer<uint64_t> tk(tk0.as<uint64_t>(), tk1.as<uint64_t>());
            hipcub::DoubleBuffer<idx_t> tv(cp, np);
            rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceRadixSort::SortPairs(t, bytes, tk, tv, (size_t)ma, 0, (int)(gbits + nsym * b), stream); });
            if (rc) return rc;
            idx_t* sorted_p = tv.Current(); idx_t* other_p = tv.Alternate();
            k_heads_active<<<grid_for(ma), 256, 0, stream>>>(tk.Current(), ma, nhead, tact.as<uint32_t>());
            FM_LAUNCHED("k_heads_active");
            rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceScan::ExclusiveSum(t, bytes, tact.as<uint32_t>(), tat.as<uint32_t>(), (size_t)ma, stream); });
            if (rc) return rc;
            // the t-th row of the sorted ties takes the t-th tied row of the bucket (the rows of a group are consecutive, and groups keep their order)
            k_compact<<<grid_for(ma), 256, 0, stream>>>(sorted_p, crow, nhead, tact.as<uint32_t>(), tat.as<uint32_t>(), ma, other_p, nrow, chead, rowpos);
            FM_LAUNCHED("k_compact");
            FM_HIP(hipMemcpyAsync(&last_at, tat.as<uint32_t>() + (ma - 1), 4, hipMemcpyDeviceToHost, stream));
            FM_HIP(hipMemcpyAsync(&last_act, tact.as<uint32_t>() + (ma - 1), 4, hipMemcpyDeviceToHost, stream));
            FM_HIP(hipStreamSynchronize(stream));
            ma = (uint64_t)last_at + last_act;
            cp = other_p; np = sorted_p;
            std::swap(crow, nrow);                                  // (chead was written in place of the old flags: k_compact reads nhead, writes chead)
            d += nsym;
        }
        // ---- 4. the bucket's rows, in suffix order
        if ((rc = sink(first_row, rowpos, m, dk.Current(), (size_t)largest * 8))) return rc;
        FM_HIP(hipStreamSynchronize(stream));
        first_row += m;
    }
    return 0;
}

}  // namespace FMGPU_NS

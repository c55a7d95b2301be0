// fmgpu_bucketsort.hip — suffix sorting for texts whose suffix array does not fit beside them.  Compiled once per row width.
//
// The sorter of fmgpu_build.hip (radix sort of every suffix's K-symbol prefix, prefix doubling on the ties) holds the text, the suffix array, the rank array, two key
// buffers and a value buffer: 30 bytes per row with 32-bit suffix indices, 42 with 64-bit ones — ~6 x 10^9 rows in 288 GB.  The reference switches to libsais64 there
// (utils.h:243-247) and has the host's memory to do it in.  What construction needs of the suffix array is its ORDER, once: bwt[i] = text[sa[i] - 1] (utils.h:145-163)
// and the sampled entries (FMIndex.h:79-101).  The two sorters here never hold the array.  Both cut the suffixes into buckets by their first symbols (a bucket = a range of
// rows of the final order) and sort one bucket at a time:
//   1. one pass over the text counts the suffixes per bin (the top kBinBits bits of the packed K-symbol key); the host cuts the bins into buckets of <= bucket_rows rows;
//   2. per bucket: a pass over the text collects (key, position) of its suffixes; one radix sort orders them by their K-symbol prefix; a flag pass finds the groups of equal prefixes.
// sort_suffixes_isa (FMGPU_OPT_SUFFIX_SORTER 2, the default beyond the all-at-once sorter's size) then doubles: every suffix gets the first row of its group as rank[position], the
// rows of groups of two and more join ONE list of tied rows, and rounds with step h = K, 2K, 4K ... sort that list by (group, rank[position + h]) until it is empty — rank is the
// inverse suffix array then, and the caller writes the BWT and the sampled entries from it.  Memory: the text, 4 / 8 bytes per row of rank array, one bucket, ~60 bytes per tied row.
// sort_suffixes_bucketed (sorter 3, where even the rank array does not fit) holds NO array of n entries: it breaks a bucket's ties by reading further symbols of the text —
//   3. rows whose K-prefix is shared are compacted and re-sorted by (group, the next symbols of the suffix) — as many symbols as fit 64 bits beside the dense group number —
//      until every group is a single row.  Not doubling: a repeat of length L costs L / symbols-per-round rounds over ITS rows, which is nothing for a protein database or a
//      text without long exact repeats, and would be hours for megabase runs of one symbol — the work is bounded (kMaxRefineWork passes over a bucket's rows) and such a text is
//      refused with an error instead;
//   4. the bucket's positions, now in suffix order, go to the caller's sink (BWT symbols, sampled suffix array entries) —
// the text + ~40 bytes per row of ONE bucket (+ the tie buffers), whatever n is.
#include "fmgpu_common.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <memory>
#include <new>
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


// ---- the two sorters share: the bucket plan and the K-symbol sort of one bucket
struct Bucket { uint64_t bin_lo, bin_hi, rows; };
struct Plan {
    uint32_t b = 0, K = 0, kbits = 0, shift = 0, bin_bits = 0;
    std::vector<Bucket> buckets;
    uint64_t largest = 0;
};
// row_bytes: device bytes a row of a bucket costs the caller (bucket_rows = 0: buckets as large as 55 % of the free memory allows)
int plan_buckets(const uint8_t* text, uint64_t n, uint32_t sigma, uint64_t bucket_rows, size_t row_bytes, hipStream_t stream, Plan& pl) {
    uint32_t b = 0; while ((1u << b) <= sigma) ++b;                 // bits for the fields 0..sigma
    pl.b = b; pl.K = 64 / b; pl.kbits = pl.K * b;
    pl.bin_bits = std::min(kBinBits, pl.kbits); pl.shift = pl.kbits - pl.bin_bits;
    const uint64_t nbins = 1ull << pl.bin_bits;
    int rc;
    std::vector<unsigned long long> hist(nbins);
    {
        DBuf dh; if ((rc = dh.alloc(nbins * 8))) return rc;
        FM_HIP(hipMemsetAsync(dh.p, 0, nbins * 8, stream));
        k_bin_histogram<<<grid_for((n + kPerThread - 1) / kPerThread), 256, 0, stream>>>(text, n, pl.K, b, pl.shift, dh.as<unsigned long long>());
        FM_LAUNCHED("k_bin_histogram");
        FM_HIP(hipMemcpyAsync(hist.data(), dh.p, nbins * 8, hipMemcpyDeviceToHost, stream));
        FM_HIP(hipStreamSynchronize(stream));
    }
    if (bucket_rows == 0) {
        size_t free_b = 0, total_b = 0;
        FM_HIP(hipMemGetInfo(&free_b, &total_b));
        bucket_rows = std::max<uint64_t>(1u << 20, (uint64_t)((double)free_b * 0.55 / (double)row_bytes));     // (the rest: tie buffers, the radix sort's scratch)
    }
    bucket_rows = std::min<uint64_t>(bucket_rows, 0xfff00000ull);   // rows within a bucket are 32-bit numbers
    Bucket cur{0, 0, 0};
    for (uint64_t bin = 0; bin < nbins; ++bin) {
        const uint64_t c = hist[bin];
        if (cur.rows && cur.rows + c > bucket_rows) { cur.bin_hi = bin; pl.buckets.push_back(cur); cur = Bucket{bin, bin, 0}; }
        cur.rows += c;
    }
    cur.bin_hi = nbins;
    if (cur.rows) pl.buckets.push_back(cur);
    uint64_t sum = 0;
    for (const Bucket& k : pl.buckets) { pl.largest = std::max(pl.largest, k.rows); sum += k.rows; }
    if (sum != n) return fail(FMGPU_ERR_HIP, "bucket histogram does not add up to the text length");
    if (pl.largest > 0xfff00000ull)
        return fail(FMGPU_ERR_UNSUPPORTED, "more than 2^32 suffixes share their first " + std::to_string(pl.bin_bits / b) + " symbols: the bucketed suffix sorter cannot cut them apart");
    return 0;
}
struct BucketBuffers {
    DBuf k0, k1, p0, p1, head, act, at, cursor;
    Temp tmp;
    int alloc(uint64_t largest) {
        int rc;
        if ((rc = k0.alloc(largest * 8)) || (rc = k1.alloc(largest * 8)) || (rc = p0.alloc(largest * sizeof(idx_t))) || (rc = p1.alloc(largest * sizeof(idx_t))) ||
            (rc = head.alloc(largest * 4)) || (rc = act.alloc(largest * 4)) || (rc = at.alloc((largest + 1) * 4)) || (rc = cursor.alloc(8))) return rc;
        return 0;
    }
};
// collects the bucket's suffixes and sorts them by their K-symbol prefix; head / act / at (exclusive sums of act) describe the groups of equal prefixes; *ties = rows in such groups of two and more
int sort_bucket(const uint8_t* text, uint64_t n, const Plan& pl, const Bucket& bk, BucketBuffers& bb, hipStream_t stream, uint64_t** keys_sorted, uint64_t** keys_spare,
                idx_t** pos_sorted, uint64_t* ties) {
    const uint64_t m = bk.rows;
    int rc;
    FM_HIP(hipMemsetAsync(bb.cursor.p, 0, 8, stream));
    k_collect<<<grid_for((n + kPerThread - 1) / kPerThread), 256, 0, stream>>>(text, n, pl.K, pl.b, pl.shift, bk.bin_lo, bk.bin_hi, bb.k0.as<uint64_t>(), bb.p0.as<idx_t>(),
                                                                                 bb.cursor.as<unsigned long long>(), m);
    FM_LAUNCHED("k_collect");
    unsigned long long got = 0;
    FM_HIP(hipMemcpyAsync(&got, bb.cursor.p, 8, hipMemcpyDeviceToHost, stream));
    FM_HIP(hipStreamSynchronize(stream));
    if (got != m) return fail(FMGPU_ERR_HIP, "a bucket collected " + std::to_string(got) + " suffixes where the histogram counted " + std::to_string(m));
    hipcub::DoubleBuffer<uint64_t> dk(bb.k0.as<uint64_t>(), bb.k1.as<uint64_t>());
    hipcub::DoubleBuffer<idx_t> dv(bb.p0.as<idx_t>(), bb.p1.as<idx_t>());
    rc = cub_call(bb.tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceRadixSort::SortPairs(t, bytes, dk, dv, (size_t)m, 0, (int)pl.kbits, stream); });
    if (rc) return rc;
    *keys_sorted = dk.Current(); *keys_spare = dk.Alternate(); *pos_sorted = dv.Current();
    k_heads_active<<<grid_for(m), 256, 0, stream>>>(dk.Current(), m, bb.head.as<uint32_t>(), bb.act.as<uint32_t>());
    FM_LAUNCHED("k_heads_active");
    rc = cub_call(bb.tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceScan::ExclusiveSum(t, bytes, bb.act.as<uint32_t>(), bb.at.as<uint32_t>(), (size_t)m, stream); });
    if (rc) return rc;
    uint32_t last_at = 0, last_act = 0;
    FM_HIP(hipMemcpyAsync(&last_at, bb.at.as<uint32_t>() + (m - 1), 4, hipMemcpyDeviceToHost, stream));
    FM_HIP(hipMemcpyAsync(&last_act, bb.act.as<uint32_t>() + (m - 1), 4, hipMemcpyDeviceToHost, stream));
    FM_HIP(hipStreamSynchronize(stream));
    *ties = (uint64_t)last_at + last_act;
    return 0;
}

// ---- kernels of the sorter that keeps the inverse suffix array
struct MaxIdx { __host__ __device__ __forceinline__ idx_t operator()(idx_t a, idx_t b) const { return a > b ? a : b; } };
struct ToIdx { __host__ __device__ __forceinline__ idx_t operator()(uint32_t v) const { return (idx_t)v; } };
// v[j] = row of j if it starts a group, else 0 (rows ascend: a running maximum turns this into every row's group start)
__global__ __launch_bounds__(256) void k_group_rows(const uint32_t* __restrict__ head, const idx_t* __restrict__ row, uint64_t first_row, uint64_t m, idx_t* __restrict__ v) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x) v[j] = head[j] ? (row ? row[j] : (idx_t)(first_row + j)) : (idx_t)0;
}
__global__ __launch_bounds__(256) void k_write_ranks(const idx_t* __restrict__ pos, const idx_t* __restrict__ gs, uint64_t m, idx_t* __restrict__ rank) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x) rank[pos[j]] = gs[j];
}
// the tied rows of a bucket, appended to the list of all tied rows: position, (global) row, head flag
__global__ __launch_bounds__(256) void k_take_ties(const idx_t* __restrict__ pos, uint64_t first_row, const uint32_t* __restrict__ head, const uint32_t* __restrict__ act,
                                                   const uint32_t* __restrict__ at, uint64_t m, idx_t* __restrict__ pos_out, idx_t* __restrict__ row_out, uint32_t* __restrict__ head_out) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x)
        if (act[j]) { const uint32_t o = at[j]; pos_out[o] = pos[j]; row_out[o] = (idx_t)(first_row + j); head_out[o] = head[j]; }
}
// keys of a doubling round over the tied rows [0, m) of one segment: (group number within the segment, rank of the suffix h symbols on + 1; 0 = past the end)
__global__ __launch_bounds__(256) void k_isa_keys(const idx_t* __restrict__ pos, const idx_t* __restrict__ gid_incl, idx_t g0, uint64_t m, const idx_t* __restrict__ rank, uint64_t n, uint64_t h,
                                                  uint32_t rbits, uint64_t* __restrict__ keys) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p = (uint64_t)pos[j] + h;
        const uint64_t second = p < n ? (uint64_t)rank[p] + 1ull : 0ull;
        keys[j] = ((uint64_t)(gid_incl[j] - 1u - g0) << rbits) | second;
    }
}
// first tied row whose (inclusive) group count reaches want[i]: the segment boundaries of a round
__global__ void k_lower_bounds(const idx_t* __restrict__ gid_incl, uint64_t m, const uint64_t* __restrict__ want, uint32_t count, uint64_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint64_t lo = 0, hi = m;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if ((uint64_t)gid_incl[mid] < want[i]) lo = mid + 1; else hi = mid; }
    out[i] = lo;
}
__global__ __launch_bounds__(256) void k_active_idx(const uint32_t* __restrict__ act, uint64_t m, idx_t* __restrict__ out) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x) out[j] = (idx_t)act[j];
}
__global__ __launch_bounds__(256) void k_compact_ties(const idx_t* __restrict__ pos_in, const idx_t* __restrict__ row_in, const uint32_t* __restrict__ head, const uint32_t* __restrict__ act,
                                                      const idx_t* __restrict__ at, uint64_t m, idx_t* __restrict__ pos_out, idx_t* __restrict__ row_out, uint32_t* __restrict__ head_out) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x)
        if (act[j]) { const uint64_t o = at[j]; pos_out[o] = pos_in[j]; row_out[o] = row_in[j]; head_out[o] = head[j]; }
}

}  // namespace

// Sorts the suffixes of text[0, n) (plain byte order; a proper prefix sorts first) bucket by bucket and hands every bucket to `sink`, buckets in ascending row order:
// sink(first_row, pos, count, scratch, scratch_bytes) — pos[0..count) = the text positions of rows first_row .. first_row + count - 1; scratch is device memory the sink may use
// until it returns (8 bytes per row of the largest bucket).  bucket_rows = the most rows a bucket should hold (0: from the free device memory).
// Ties are broken by reading further symbols of the text (no rank array: the text + one bucket is all the memory there is); texts with very long exact repeats are refused.
int sort_suffixes_bucketed(const uint8_t* text, uint64_t n, uint32_t sigma, uint64_t bucket_rows, const SuffixSink& sink, hipStream_t stream) {
    if (n == 0) return 0;
    Plan pl; int rc;
    if ((rc = plan_buckets(text, n, sigma, bucket_rows, 16 + 2 * sizeof(idx_t) + 12, stream, pl))) return rc;
    const uint32_t K = pl.K, b = pl.b;
    BucketBuffers bb;
    if ((rc = bb.alloc(pl.largest))) return rc;
    Temp& tmp = bb.tmp;
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
    for (const Bucket& bk : pl.buckets) {
        const uint64_t m = bk.rows;
        // ---- collect, sort by the K-symbol prefix
        uint64_t *keys_sorted = nullptr, *keys_spare = nullptr; idx_t* rowpos = nullptr;     // rowpos: the bucket's order; tied rows are overwritten below
        uint64_t ma = 0;
        if ((rc = sort_bucket(text, n, pl, bk, bb, stream, &keys_sorted, &keys_spare, &rowpos, &ma))) return rc;
        // ---- ties
        uint64_t work = 0, d = K;
        uint32_t last_at = 0, last_act = 0;
        idx_t *cp = nullptr, *np = nullptr; uint32_t *crow = nullptr, *nrow = nullptr, *chead = nullptr, *nhead = nullptr;
        if (ma) {
            if ((rc = tie_buffers(ma))) return rc;
            cp = tp0.as<idx_t>(); np = tp1.as<idx_t>(); crow = trow0.as<uint32_t>(); nrow = trow1.as<uint32_t>(); chead = thead0.as<uint32_t>(); nhead = thead1.as<uint32_t>();
            k_compact<<<grid_for(m), 256, 0, stream>>>(rowpos, nullptr, bb.head.as<uint32_t>(), bb.act.as<uint32_t>(), bb.at.as<uint32_t>(), m, cp, crow, chead, nullptr);
            FM_LAUNCHED("k_compact");
        }
        while (ma) {
            work += ma;
            if (work > kMaxRefineWork * std::max<uint64_t>(m, 1u << 20))
                return fail(FMGPU_ERR_UNSUPPORTED, "the text holds exact repeats too long for the suffix sorter that keeps no rank array (depth " + std::to_string(d) + " symbols reached with " +
                                                   std::to_string(ma) + " rows still tied): the doubling sorters handle such texts up to the size their arrays fit");
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
        // ---- the bucket's rows, in suffix order
        if ((rc = sink(first_row, rowpos, m, keys_sorted, (size_t)pl.largest * 8))) return rc;
        FM_HIP(hipStreamSynchronize(stream));
        first_row += m;
    }
    return 0;
}

// The same bucketed first sort, then prefix doubling on the ties with the inverse suffix array as the rank array: rank[p] = the row of suffix p when it returns (n entries, device).
// After the K-symbol sort of a bucket every suffix of it gets the first row of its group as its rank, and the rows of groups of two and more join the list of tied rows
// (position, row, head-of-group flag).  A round with step h sorts the tied rows by (group, rank[p + h]) — both known to h symbols — which orders them by 2h symbols; groups split,
// single rows leave the list, h doubles: a repeat of length L costs log2(L / K) rounds over its rows, as in the all-at-once sorter of fmgpu_build.hip, but neither the suffix array
// nor keys of all n suffixes are ever held: the text, the rank array (4 / 8 bytes per row), one bucket, and ~50-70 bytes per TIED row.
// A round's sort key is group : rank in 64 bits; where the tied rows have more groups than fit beside a rank (n = 10^10: 2^30), the list is sorted in segments of that many groups
// (groups are contiguous and keep their order, so the segments are independent).  bucket_rows also caps a segment's groups (so that small tests run through several segments).
int sort_suffixes_isa(const uint8_t* text, uint64_t n, uint32_t sigma, uint64_t bucket_rows, idx_t* rank, hipStream_t stream) {
    if (n == 0) return 0;
    Plan pl; int rc;
    if ((rc = plan_buckets(text, n, sigma, bucket_rows, 16 + 2 * sizeof(idx_t) + 12, stream, pl))) return rc;
    struct Seg { DBuf pos, row, head; uint64_t count = 0; };
    std::vector<std::unique_ptr<Seg>> segs;
    uint64_t M = 0;
    {
        BucketBuffers bb;
        if ((rc = bb.alloc(pl.largest))) return rc;
        uint64_t first_row = 0;
        for (const Bucket& bk : pl.buckets) {
            const uint64_t m = bk.rows;
            uint64_t *keys_sorted = nullptr, *keys_spare = nullptr; idx_t* pos = nullptr; uint64_t ties = 0;
            if ((rc = sort_bucket(text, n, pl, bk, bb, stream, &keys_sorted, &keys_spare, &pos, &ties))) return rc;
            idx_t* gs = reinterpret_cast<idx_t*>(keys_spare);          // (8 bytes per row: room for a row number)
            k_group_rows<<<grid_for(m), 256, 0, stream>>>(bb.head.as<uint32_t>(), nullptr, first_row, m, gs);
            FM_LAUNCHED("k_group_rows");
            rc = cub_call(bb.tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceScan::InclusiveScan(t, bytes, gs, gs, MaxIdx{}, (size_t)m, stream); });
            if (rc) return rc;
            k_write_ranks<<<grid_for(m), 256, 0, stream>>>(pos, gs, m, rank);
            FM_LAUNCHED("k_write_ranks");
            if (ties) {
                std::unique_ptr<Seg> sg(new (std::nothrow) Seg());
                if (!sg) return fail(FMGPU_ERR_NOMEM, "host allocation");
                if ((rc = sg->pos.alloc(ties * sizeof(idx_t))) || (rc = sg->row.alloc(ties * sizeof(idx_t))) || (rc = sg->head.alloc(ties * 4))) return rc;
                sg->count = ties;
                k_take_ties<<<grid_for(m), 256, 0, stream>>>(pos, first_row, bb.head.as<uint32_t>(), bb.act.as<uint32_t>(), bb.at.as<uint32_t>(), m, sg->pos.as<idx_t>(), sg->row.as<idx_t>(),
                                                              sg->head.as<uint32_t>());
                FM_LAUNCHED("k_take_ties");
                segs.push_back(std::move(sg));
                M += ties;
            }
            FM_HIP(hipStreamSynchronize(stream));
            first_row += m;
        }
    }                                                               // (the bucket's buffers are gone before the rounds' are made)
    if (M == 0) return 0;
    // ---- the list of tied rows, and the buffers of a round
    DBuf pa, pb, ra, rb, ha, hb, ka, kb, act, scan, bounds_d, want_d;
    if ((rc = pa.alloc(M * sizeof(idx_t))) || (rc = pb.alloc(M * sizeof(idx_t))) || (rc = ra.alloc(M * sizeof(idx_t))) || (rc = rb.alloc(M * sizeof(idx_t))) || (rc = ha.alloc(M * 4)) ||
        (rc = hb.alloc(M * 4)) || (rc = ka.alloc(M * 8)) || (rc = kb.alloc(M * 8)) || (rc = act.alloc(M * 4)) || (rc = scan.alloc(M * sizeof(idx_t)))) return rc;
    {
        uint64_t at = 0;
        for (auto& sg : segs) {
            FM_HIP(hipMemcpyAsync(pa.as<idx_t>() + at, sg->pos.p, sg->count * sizeof(idx_t), hipMemcpyDeviceToDevice, stream));
            FM_HIP(hipMemcpyAsync(ra.as<idx_t>() + at, sg->row.p, sg->count * sizeof(idx_t), hipMemcpyDeviceToDevice, stream));
            FM_HIP(hipMemcpyAsync(ha.as<uint32_t>() + at, sg->head.p, sg->count * 4, hipMemcpyDeviceToDevice, stream));
            at += sg->count;
        }
        FM_HIP(hipStreamSynchronize(stream));
        segs.clear();
    }
    idx_t *pos = pa.as<idx_t>(), *pos2 = pb.as<idx_t>(), *row = ra.as<idx_t>(), *row2 = rb.as<idx_t>();
    uint32_t *head = ha.as<uint32_t>(), *head2 = hb.as<uint32_t>();
    uint64_t *keys = ka.as<uint64_t>(), *keys2 = kb.as<uint64_t>();
    Temp tmp;
    const uint32_t rbits = bit_width64(n);                          // a rank + 1 fits these bits
    uint64_t seg_groups = rbits >= 63 ? 2 : (1ull << (64 - rbits));
    if (bucket_rows) seg_groups = std::max<uint64_t>(2, std::min(seg_groups, bucket_rows));
    uint64_t h = pl.K;
    for (int round = 0; M; ++round) {
        if (round > 80) return fail(FMGPU_ERR_HIP, "suffix sorting did not converge");
        idx_t* gid = scan.as<idx_t>();
        hipcub::TransformInputIterator<idx_t, ToIdx, const uint32_t*> head_it(head, ToIdx{});
        rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceScan::InclusiveSum(t, bytes, head_it, gid, (size_t)M, stream); });
        if (rc) return rc;
        idx_t groups_i = 0;
        FM_HIP(hipMemcpyAsync(&groups_i, gid + (M - 1), sizeof(idx_t), hipMemcpyDeviceToHost, stream));
        FM_HIP(hipStreamSynchronize(stream));
        const uint64_t groups = groups_i;
        const uint64_t nseg = (groups + seg_groups - 1) / seg_groups;
        std::vector<uint64_t> bounds(nseg + 1, 0);
        bounds[nseg] = M;
        if (nseg > 1) {                                             // first tied row of every segment: the first row whose group number reaches i * seg_groups
            std::vector<uint64_t> want(nseg);
            for (uint64_t i = 0; i < nseg; ++i) want[i] = i * seg_groups + 1;
            if ((rc = want_d.alloc(nseg * 8)) || (rc = bounds_d.alloc(nseg * 8))) return rc;
            FM_HIP(hipMemcpyAsync(want_d.p, want.data(), nseg * 8, hipMemcpyHostToDevice, stream));
            k_lower_bounds<<<dim3((unsigned)((nseg + 255) / 256)), 256, 0, stream>>>(gid, M, want_d.as<uint64_t>(), (uint32_t)nseg, bounds_d.as<uint64_t>());
            FM_LAUNCHED("k_lower_bounds");
            FM_HIP(hipMemcpyAsync(bounds.data(), bounds_d.p, nseg * 8, hipMemcpyDeviceToHost, stream));
            FM_HIP(hipStreamSynchronize(stream));
        }
        const uint32_t lbits = std::max(1u, bit_width64(std::min(groups, seg_groups) - 1));
        for (uint64_t i = 0; i < nseg; ++i) {
            const uint64_t e0 = bounds[i], cnt = bounds[i + 1] - e0;
            if (!cnt) continue;
            k_isa_keys<<<grid_for(cnt), 256, 0, stream>>>(pos + e0, gid + e0, (idx_t)(i * seg_groups), cnt, rank, n, h, rbits, keys + e0);
            FM_LAUNCHED("k_isa_keys");
            rc = cub_call(tmp, [&](void* t, size_t& bytes) {
                return hipcub::DeviceRadixSort::SortPairs(t, bytes, keys + e0, keys2 + e0, pos + e0, pos2 + e0, (size_t)cnt, 0, (int)std::min(64u, lbits + rbits), stream); });
            if (rc) return rc;
            k_heads_active<<<grid_for(cnt), 256, 0, stream>>>(keys2 + e0, cnt, head2 + e0, act.as<uint32_t>() + e0);
            FM_LAUNCHED("k_heads_active");
        }
        // ranks of the new groups: the first row of each (the rows of the list ascend, the t-th sorted tie sits in the t-th tied row)
        idx_t* gs = scan.as<idx_t>();
        k_group_rows<<<grid_for(M), 256, 0, stream>>>(head2, row, 0, M, gs);
        FM_LAUNCHED("k_group_rows");
        rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceScan::InclusiveScan(t, bytes, gs, gs, MaxIdx{}, (size_t)M, stream); });
        if (rc) return rc;
        k_write_ranks<<<grid_for(M), 256, 0, stream>>>(pos2, gs, M, rank);
        FM_LAUNCHED("k_write_ranks");
        // rows that are alone in their group leave the list
        idx_t* at = scan.as<idx_t>();
        hipcub::TransformInputIterator<idx_t, ToIdx, const uint32_t*> act_it(act.as<uint32_t>(), ToIdx{});
        rc = cub_call(tmp, [&](void* t, size_t& bytes) { return hipcub::DeviceScan::ExclusiveSum(t, bytes, act_it, at, (size_t)M, stream); });
        if (rc) return rc;
        idx_t last_at = 0; uint32_t last_act = 0;
        FM_HIP(hipMemcpyAsync(&last_at, at + (M - 1), sizeof(idx_t), hipMemcpyDeviceToHost, stream));
        FM_HIP(hipMemcpyAsync(&last_act, act.as<uint32_t>() + (M - 1), 4, hipMemcpyDeviceToHost, stream));
        k_compact_ties<<<grid_for(M), 256, 0, stream>>>(pos2, row, head2, act.as<uint32_t>(), at, M, pos, row2, head);
        FM_LAUNCHED("k_compact_ties");
        FM_HIP(hipStreamSynchronize(stream));
        M = (uint64_t)last_at + last_act;
        std::swap(row, row2);
        h *= 2;
    }
    return 0;
}

}  // namespace FMGPU_NS

// fmgpu_index.hip — index upload / re-layout, optional accelerator tables, String_c batch queries, cursor steps.  Compiled once per row width.
#include "fmgpu_common.h"

#include <algorithm>
#include <cstdlib>
#include <memory>
#include <new>

#include <hipcub/hipcub.hpp>

namespace FMGPU_NS {

// ------------------------------------------------------------------ layout parameters (mirror of the reference's struct layouts)
struct RefLayout {
    int family;      // reference family: 0 IB, 1 IBP, 2 EPR, 3 EPRV2, 4 wavelet
    uint32_t bt, K, bits_off, stride, rows, bitct;
    uint64_t period;
};

static int bit_width_u(uint64_t v) { int r = 0; while (v) { ++r; v >>= 1; } return r; }

static int ref_layout(int layout, int sigma, RefLayout& L) {
    uint32_t align = 8;
    L.bitct = (uint32_t)bit_width_u((uint64_t)sigma - 1);
    switch (layout) {
    case FMGPU_IB8:      L.family = 0; L.bt = 1; break;
    case FMGPU_IB16:     L.family = 0; L.bt = 2; break;
    case FMGPU_IB32:     L.family = 0; L.bt = 4; break;
    case FMGPU_IB16A:    L.family = 0; L.bt = 2; align = 64; break;
    case FMGPU_IBP16:    L.family = 1; L.bt = 2; break;
    case FMGPU_EPR8:     L.family = 2; L.bt = 1; break;
    case FMGPU_EPR16:    L.family = 2; L.bt = 2; break;
    case FMGPU_EPR32:    L.family = 2; L.bt = 4; break;
    case FMGPU_EPRV2_8:  L.family = 3; L.bt = 1; break;
    case FMGPU_EPRV2_16: L.family = 3; L.bt = 2; break;
    case FMGPU_EPRV2_32: L.family = 3; L.bt = 4; break;
    case FMGPU_WAVELET:  L.family = 4; L.bt = 0; return 0;
    default: return -1;
    }
    uint64_t full = 1ull << (8 * L.bt);
    L.rows = 64; L.period = full;
    if (L.family <= 1) L.K = (uint32_t)sigma;
    else if (L.family == 3) L.K = L.bitct;
    else { L.K = 1; L.rows = 64 / L.bitct; L.period = (full / L.rows) * L.rows; }
    L.bits_off = (uint32_t)(((uint64_t)sigma * L.bt + 7) / 8 * 8);
    L.stride = (uint32_t)(((uint64_t)L.bits_off + 8ull * L.K + align - 1) / align * align);
    return 0;
}

// ------------------------------------------------------------------ Format A conversion kernel
// thread = (device block B, symbol c).  Reference row p lives at bit (p+1)&63 of block (p+1)>>6
// (string/InterleavedBitvector.h:64-94); prefix layout stores s[j] <= c (InterleavedBitvectorPrefix.h:86-100).
__global__ __launch_bounds__(256) void k_convert_ib(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ super,
                                                    const idx_t* __restrict__ C, uint8_t* __restrict__ out, uint64_t* __restrict__ out_super,
                                                    uint64_t nblocks, uint32_t sigma, uint32_t bt, uint32_t bits_off,
                                                    uint32_t stride, uint64_t period, uint32_t bstride, int prefix) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nblocks * sigma; t += (uint64_t)gridDim.x * blockDim.x) {      // (strided: a launch holds fewer than 2^32 threads)
    uint64_t B = t / sigma;
    uint32_t c = (uint32_t)(t % sigma);
    auto count_of = [&](uint64_t b, uint32_t s) -> uint64_t {
        const uint8_t* p = raw + b * stride + (uint64_t)s * bt;
        if (bt == 2) return *reinterpret_cast<const uint16_t*>(p);
        if (bt == 1) return *p;
        return *reinterpret_cast<const uint32_t*>(p);
    };
    auto word_of = [&](uint64_t b, uint32_t s) -> uint64_t {
        return *reinterpret_cast<const uint64_t*>(raw + b * stride + bits_off + 8ull * s);
    };
    auto eval = [&](uint64_t b, uint64_t* bits_out) -> uint64_t {       // C[c] + occurrences of c in rows [0, 64 b)
        uint64_t sb = (64ull * b) / period;
        uint64_t w = word_of(b, c), cnt = count_of(b, c) + super[sb * sigma + c];
        uint64_t wn = b + 1 < nblocks ? word_of(b + 1, c) : 0;
        if (prefix && c > 0) {   // cumulative planes -> exclusive planes
            uint64_t wl = word_of(b, c - 1);
            uint64_t wln = b + 1 < nblocks ? word_of(b + 1, c - 1) : 0;
            uint64_t bmask = bt == 2 ? 0xffffull : (bt == 1 ? 0xffull : 0xffffffffull);
            cnt = ((count_of(b, c) - count_of(b, c - 1)) & bmask) + super[sb * sigma + c] - super[sb * sigma + c - 1];
            w &= ~wl; wn &= ~wln;
        }
        if (bits_out) *bits_out = (w >> 1) | ((wn & 1ull) << 63);
        return cnt + (w & 1ull) + C[c];
    };
    uint64_t bits = 0;
    const uint64_t total = eval(B, &bits);
    const uint64_t base = kWide ? eval(super_first_block(B), nullptr) : 0;
    put_entry_a(out, out_super, B, c, sigma, bstride, total, base, bits);
    }
}

static int upload(const void* host, size_t bytes, void** dev) {
    *dev = nullptr;
    if (bytes == 0) bytes = 8;
    FM_HIP(hipMalloc(dev, bytes));
    if (host) {
        hipError_t e = hipMemcpy(*dev, host, bytes, hipMemcpyDefault);
        if (e != hipSuccess) { (void)hipFree(*dev); *dev = nullptr; return hip_fail(e, "hipMemcpy(upload)"); }
    }
    return 0;
}

// allocates the Format A block table (+ the wide super table) of a string of n rows
static int alloc_format_a(DevString& s, uint64_t n, uint32_t sigma, const idx_t* dC) {
    const uint64_t nblocks = n / 64 + 1;
    const uint32_t bstride = sigma <= 5 ? 64u : 12u * sigma;
    DBuf blk, sup;
    int rc;
    if ((rc = blk.alloc(nblocks * bstride + 64))) return rc;
    FM_HIP(hipMemset(blk.p, 0, blk.bytes));
    if (kWide) {
        if ((rc = sup.alloc(((n >> kSuperShift) + 1) * sigma * 8))) return rc;
        FM_HIP(hipMemset(sup.p, 0, sup.bytes));
    }
    s.blk_bytes = blk.bytes; s.blk = blk.take();
    s.sup_bytes = kWide ? sup.bytes : 0; s.sup = sup.take();
    s.family = FAM_A;
    s.va = ViewA{(const uint8_t*)s.blk, bstride, sigma, dC, (const uint64_t*)s.sup, 0u, 0};
    return 0;
}

// ---- fused presence bits (fmgpu_common.h): word B of the SparseArray's presence bitvector (bitvector/Bitvector2L.h:30-33, bit r of word B = row 64 B + r)
// replaces the bitmap of entry 0 in block B of the bwt's Format A table
__global__ __launch_bounds__(256) void k_fuse_presence(uint8_t* __restrict__ blk, ViewSA sa, uint64_t nblocks) {
    const uint64_t B = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (B >= nblocks) return;
    const uint64_t w = sa.bits[B];
    uint32_t* o = reinterpret_cast<uint32_t*>(blk + B * 64u);
    o[1] = (uint32_t)w; o[2] = (uint32_t)(w >> 32);
    // 32-bit rows: the block's 4 spare bytes hold the number of sampled rows before it (bitvector/Bitvector2L.h:123-142 evaluated at the block's first
    // row), so that the rank of a sampled row — the index of its SparseArray value — comes from the line that said it is sampled
    if constexpr (!kWide) o[15] = (uint32_t)sa_rank(sa, (idx_t)(B * 64u));
}
int fuse_presence_bits(Index* x, hipStream_t stream) {
    DevString& s = x->bwt;
    if (!x->has_sa || s.search_family() != FAM_A || s.va.bstride != 64u || s.sigma < 2 || s.va.fused || !opt_on(FMGPU_OPT_FUSED_LOCATE)) return 0;   // (the string's own Format A blocks, or its expansion)
    const uint64_t nblocks = s.n / 64 + 1;
    FM_GRID(grid, nblocks);
    k_fuse_presence<<<grid, dim3(256), 0, stream>>>(const_cast<uint8_t*>(s.va.blk), x->vsa, nblocks);
    FM_LAUNCHED("k_fuse_presence");
    FM_HIP(hipStreamSynchronize(stream));
    idx_t ksum = 0;
    for (int c = 1; c < s.sigma; ++c) ksum += (idx_t)x->hC[c];
    s.va.fused = 1u; s.va.ksum = ksum;
    return 0;
}

int auto_shadow(Index* x, hipStream_t stream) {
    if (x->bwt.sigma != 5 || !opt_on(FMGPU_OPT_EXPAND_DNA)) return 0;
    for (DevString* t : {&x->bwt, &x->rev}) {
        if (t->n == 0 || t->family == FAM_A || t->shadow) continue;
        int rc = build_format_a_shadow(*t, x->dC, stream);
        if (rc) return rc;
    }
    return 0;
}

// ---- Format D (fmgpu_common.h): from the Format A blocks of a sigma = 5 string.  One thread per block; a delimiter row (no symbol >= 1 claims it) is appended to the list
constexpr uint32_t kDenseMaxDelims = 256;
__global__ __launch_bounds__(256) void k_dense_dna(const uint8_t* __restrict__ blk, uint64_t nblocks, uint64_t n, uint4* __restrict__ out, uint32_t* __restrict__ ex, uint32_t* __restrict__ nex) {
    const uint64_t B = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (B >= nblocks) return;
    const uint32_t* d = reinterpret_cast<const uint32_t*>(blk + B * 64u);
    uint64_t bits[5];
    for (int c = 1; c < 5; ++c) bits[c] = (uint64_t)d[3 * c + 1] | ((uint64_t)d[3 * c + 2] << 32);
    const uint64_t p0 = bits[2] | bits[4], p1 = bits[3] | bits[4];
    out[2 * B] = make_uint4(d[3], d[6], d[9], d[12]);
    out[2 * B + 1] = make_uint4((uint32_t)p0, (uint32_t)(p0 >> 32), (uint32_t)p1, (uint32_t)(p1 >> 32));
    const uint64_t rows = n - B * 64u < 64u ? n - B * 64u : 64u;
    uint64_t none = ~(bits[1] | bits[2] | bits[3] | bits[4]);
    if (rows < 64u) none &= (1ull << rows) - 1ull;
    while (none) {
        const uint32_t r = (uint32_t)__ffsll((unsigned long long)none) - 1u;
        none &= none - 1ull;
        const uint32_t k = atomicAdd(nex, 1u);
        if (k < kDenseMaxDelims) ex[k] = (uint32_t)(B * 64u + r);
    }
}
int build_dense_dna(DevString& s, hipStream_t stream) {
    if (kWide || s.sigma != 5 || s.search_family() != FAM_A || s.va.bstride != 64u || s.n < 2 || s.dense || !opt_on(FMGPU_OPT_DENSE_DNA)) return 0;   // (Format A blocks: the string's own or its expansion)
    // (reads the entries of symbols 1..4 only: entry 0's bitmap may already hold the presence bits)
    const uint64_t nblocks = s.n / 64 + 1;
    DBuf out, ex, cnt; int rc;
    if ((rc = out.alloc(nblocks * 32 + 64)) || (rc = ex.alloc(kDenseMaxDelims * 4)) || (rc = cnt.alloc(8))) return rc;
    FM_HIP(hipMemsetAsync(cnt.p, 0, 8, stream));
    FM_HIP(hipMemsetAsync(ex.p, 0xff, kDenseMaxDelims * 4, stream));
    FM_GRID(grid, nblocks);
    k_dense_dna<<<grid, dim3(256), 0, stream>>>(s.va.blk, nblocks, s.n, out.as<uint4>(), ex.as<uint32_t>(), cnt.as<uint32_t>());
    FM_LAUNCHED("k_dense_dna");
    uint32_t nex = 0;
    FM_HIP(hipMemcpyAsync(&nex, cnt.p, 4, hipMemcpyDeviceToHost, stream));
    FM_HIP(hipStreamSynchronize(stream));
    if (nex > kDenseMaxDelims) return 0;                         // many sequences: the k-mismatch kernel reads Format A
    std::vector<uint32_t> rows(nex);
    if (nex) FM_HIP(hipMemcpy(rows.data(), ex.p, nex * 4, hipMemcpyDeviceToHost));
    std::sort(rows.begin(), rows.end());
    if (nex) FM_HIP(hipMemcpy(ex.p, rows.data(), nex * 4, hipMemcpyHostToDevice));
    s.dense_bytes = out.bytes; s.dense = out.take(); s.dense_ex = (uint32_t*)ex.take(); s.dense_nex = nex;
    return 0;
}

// ---- Format P (fmgpu_common.h): the occurrence table of the two-symbol-step BWT of a sigma = 5 string, from its Format A blocks.
// Row i's pair is (x, y) = (s[LF(i)], s[i]): prepending "xy" to the suffixes of an interval is one rank of the pair.  A row whose pair holds a delimiter
// is written as code 0, left out of the counts and listed.
constexpr uint32_t kPairMaxRows = 512;              // listed rows (two per sequence): more than these and the table is not built
__global__ __launch_bounds__(256) void k_pair_codes(OccA<5> occ, uint64_t n, uint8_t* __restrict__ lines, uint32_t* __restrict__ part, uint64_t nlines,
                                                    idx_t* __restrict__ ex, uint32_t* __restrict__ nex) {
    __shared__ uint32_t s_cnt[4][16];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint64_t chunk = blockIdx.x; chunk * 2u < nlines; chunk += gridDim.x) {     // 256 rows = two lines per pass (a grid-stride loop: the thread count of a launch stays below 2^32)
        const uint64_t i = chunk * 256u + threadIdx.x;
        uint32_t code = 0; bool listed = false;
        if (i < n) {
            const uint32_t y = occ.symbol((idx_t)i);
            if (y == 0) listed = true;
            else {
                const uint32_t x = occ.symbol(occ.lf((idx_t)i, y));
                if (x == 0) listed = true; else code = (x - 1u) * 4u + (y - 1u);
            }
        }
        const bool counted = i < n && !listed;
        uint64_t plane[4];
        for (int k = 0; k < 4; ++k) plane[k] = __ballot(counted && ((code >> k) & 1u));
        const uint64_t valid = __ballot(counted);
        uint64_t lm = __ballot(listed);
        if (lane == 0) while (lm) {
            const uint32_t r = (uint32_t)__ffsll((unsigned long long)lm) - 1u; lm &= lm - 1ull;
            const uint32_t k = atomicAdd(nex, 1u);
            if (k < kPairMaxRows) ex[k] = (idx_t)(chunk * 256u + wave * 64u + r);
        }
        const uint64_t L = chunk * 2u + (wave >> 1);                // this wave's 128-row line
        if (lane < 4 && L < nlines) reinterpret_cast<uint64_t*>(lines + L * 128u + 64u + (wave & 1u) * 32u)[lane] = plane[lane];
        if (lane < 16) {
            uint64_t mm = valid;
            for (int k = 0; k < 4; ++k) mm &= ((lane >> k) & 1u) ? plane[k] : ~plane[k];
            s_cnt[wave][lane] = (uint32_t)__popcll((unsigned long long)mm);
        }
        __syncthreads();
        if (threadIdx.x < 32) {
            const uint32_t b = threadIdx.x >> 4, pc = threadIdx.x & 15u;
            const uint64_t L2 = chunk * 2u + b;
            if (L2 < nlines) part[(size_t)pc * nlines + L2] = s_cnt[2 * b][pc] + s_cnt[2 * b + 1][pc];
        }
        __syncthreads();
    }
}
// counts of line L = where the interval of the pair starts + the pair's occurrences before the line (part: scanned per pair)
__global__ __launch_bounds__(256) void k_pair_counts(OccA<5> occ, const uint32_t* __restrict__ part, uint64_t nlines, uint8_t* __restrict__ lines) {
    __shared__ uint32_t s_c2[16];
    if (threadIdx.x < 16) {                                             // "xy" from the whole table: y first, then x
        const uint32_t x = threadIdx.x >> 2, y = threadIdx.x & 3u;
        s_c2[threadIdx.x] = (uint32_t)occ.lf(occ.lf(0, y + 1u), x + 1u);
    }
    __syncthreads();
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t L = t >> 4; const uint32_t pc = (uint32_t)t & 15u;
    if (L < nlines) reinterpret_cast<uint32_t*>(lines + L * 128u)[pc] = s_c2[pc] + part[(size_t)pc * nlines + L];
}
// 64-bit rows: workgroup (sb, pc) runs the count of pair pc through the lines of super-block sb (2^30 rows) and writes it into the lines; the super-block's total
__global__ __launch_bounds__(256) void k_pair_counts_seg(const uint32_t* __restrict__ part, uint64_t nlines, uint8_t* __restrict__ lines, uint64_t* __restrict__ totals) {
    typedef hipcub::BlockScan<uint32_t, 256> Scan;
    __shared__ typename Scan::TempStorage tmp;
    const uint32_t sb = blockIdx.x, pc = blockIdx.y;
    const uint64_t first = (uint64_t)sb << (kSuperShift - 7), last = min(nlines, first + (1ull << (kSuperShift - 7)));
    uint32_t run = 0;                                                // < 2^30: fits
    for (uint64_t base = first; base < last; base += 256u) {
        const uint64_t L = base + threadIdx.x;
        const uint32_t v = L < last ? part[(size_t)pc * nlines + L] : 0u;
        uint32_t before, sum;
        Scan(tmp).ExclusiveSum(v, before, sum);
        __syncthreads();
        if (L < last) reinterpret_cast<uint32_t*>(lines + L * 128u)[pc] = run + before;
        run += sum;
    }
    if (threadIdx.x == 0) totals[(size_t)sb * 16u + pc] = run;
}
__global__ void k_pair_super(OccA<5> occ, const uint64_t* __restrict__ totals, uint32_t nsb, idx_t* __restrict__ super) {
    const uint32_t pc = threadIdx.x;
    if (pc >= 16u) return;
    idx_t run = occ.lf(occ.lf(0, (pc & 3u) + 1u), (pc >> 2) + 1u);  // "xy" from the whole table: y first, then x
    for (uint32_t sb = 0; sb < nsb; ++sb) { super[(size_t)sb * 16u + pc] = run; run += (idx_t)totals[(size_t)sb * 16u + pc]; }
}
int build_pair_table(Index* x, hipStream_t stream) {
    DevString& s = x->bwt;
    if (s.sigma != 5 || s.search_family() != FAM_A || s.va.bstride != 64u || s.n < 2 || s.n >= (1ull << 38) || s.pairs || !opt_on(FMGPU_OPT_PAIR_TABLE)) return 0;
    const uint64_t nlines = s.n / 128 + 1;
    DBuf out, part, ex, cnt, tmp, totals, super; int rc;
    if ((rc = out.alloc(nlines * 128)) || (rc = part.alloc(nlines * 16 * 4)) || (rc = ex.alloc(kPairMaxRows * sizeof(idx_t))) || (rc = cnt.alloc(8))) return rc;
    FM_HIP(hipMemsetAsync(cnt.p, 0, 8, stream));
    FM_HIP(hipMemsetAsync(ex.p, 0xff, kPairMaxRows * sizeof(idx_t), stream));
    dim3 grid; if ((rc = grid_of(nlines * 128, &grid, 1u << 22))) return rc;
    k_pair_codes<<<grid, dim3(256), 0, stream>>>(OccA<5>{s.va}, s.n, out.as<uint8_t>(), part.as<uint32_t>(), nlines, ex.as<idx_t>(), cnt.as<uint32_t>());
    FM_LAUNCHED("k_pair_codes");
    uint32_t nex = 0;
    FM_HIP(hipMemcpyAsync(&nex, cnt.p, 4, hipMemcpyDeviceToHost, stream));
    FM_HIP(hipStreamSynchronize(stream));
    if (nex > kPairMaxRows) return 0;                             // many sequences: exact search keeps its one-symbol steps
    uint32_t nsb = 0;
    if constexpr (kWide) {
        nsb = (uint32_t)(s.n >> kSuperShift) + 1u;
        if ((rc = totals.alloc((size_t)nsb * 16 * 8)) || (rc = super.alloc((size_t)nsb * 16 * sizeof(idx_t)))) return rc;
        k_pair_counts_seg<<<dim3(nsb, 16), dim3(256), 0, stream>>>(part.as<uint32_t>(), nlines, out.as<uint8_t>(), totals.as<uint64_t>());
        FM_LAUNCHED("k_pair_counts_seg");
        k_pair_super<<<1, 16, 0, stream>>>(OccA<5>{s.va}, totals.as<uint64_t>(), nsb, super.as<idx_t>());
        FM_LAUNCHED("k_pair_super");
    } else {
        size_t tb = 0;
        FM_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, part.as<uint32_t>(), part.as<uint32_t>(), (size_t)nlines));
        if ((rc = tmp.alloc(tb))) return rc;
        for (uint32_t pc = 0; pc < 16; ++pc) {
            uint32_t* p = part.as<uint32_t>() + (size_t)pc * nlines; size_t b2 = tb;
            FM_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, b2, p, p, (size_t)nlines, stream));
        }
        FM_GRID(grid2, nlines * 16);
        k_pair_counts<<<grid2, dim3(256), 0, stream>>>(OccA<5>{s.va}, part.as<uint32_t>(), nlines, out.as<uint8_t>());
        FM_LAUNCHED("k_pair_counts");
    }
    std::vector<idx_t> rows(nex);
    if (nex) FM_HIP(hipMemcpyAsync(rows.data(), ex.p, nex * sizeof(idx_t), hipMemcpyDeviceToHost, stream));
    FM_HIP(hipStreamSynchronize(stream));
    std::sort(rows.begin(), rows.end());
    if (nex) FM_HIP(hipMemcpy(ex.p, rows.data(), nex * sizeof(idx_t), hipMemcpyHostToDevice));
    s.pairs_bytes = out.bytes + super.bytes; s.pairs = (uint8_t*)out.take(); s.pairs_ex = (idx_t*)ex.take(); s.pairs_nex = nex;
    s.pairs_super = kWide ? (idx_t*)super.take() : nullptr; s.pairs_nsb = nsb;
    x->device_bytes += s.pairs_bytes;
    return 0;
}

// ---- Format S (fmgpu_common.h): a flat one-line-per-64-rows occurrence table from the symbols of a Wavelet string.
// The counts inside a line are flat_count_bits(sigma) wide (as many as the 88 bytes behind the planes allow: sigma = 28: 25, sigma = 29: 24) and relative to the line's
// super-block of 2^bits rows: the wider the counts, the smaller the super table k_exact_s keeps in LDS.
template <class Occ>
__global__ __launch_bounds__(256) void k_flat_planes(Occ occ, uint64_t n, uint32_t sigma, uint8_t* __restrict__ lines, uint8_t* __restrict__ cnt8, uint64_t nlines) {
    const uint32_t lane = threadIdx.x & 63u;
    // (a grid-stride loop: a launch of more than 2^32 threads is not a thing HIP does)
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; (i >> 6) < nlines; i += (uint64_t)gridDim.x * 256u) {
        const uint64_t L = i >> 6;
        const bool in = i < n;
        const uint32_t sym = in ? occ.symbol((idx_t)i) : 0u;
        uint64_t plane[5];
        for (int k = 0; k < 5; ++k) plane[k] = __ballot(in && ((sym >> k) & 1u));
        const uint64_t valid = __ballot(in);
        if (lane < 5) reinterpret_cast<uint64_t*>(lines + L * 128u)[lane] = plane[lane];
        if (lane < sigma) {
            uint64_t mm = valid;
            for (int k = 0; k < 5; ++k) mm &= ((lane >> k) & 1u) ? plane[k] : ~plane[k];
            cnt8[(size_t)lane * nlines + L] = (uint8_t)__popcll((unsigned long long)mm);
        }
    }
}
// workgroup (sb, c): the running count of symbol c through the lines of super-block sb, written into the lines (bits [c * cbits, (c + 1) * cbits) of the 88 bytes
// behind the planes; the lines start zeroed, workgroups of different symbols share dwords: atomicOr); the super-block's total
__global__ __launch_bounds__(256) void k_flat_counts(const uint8_t* __restrict__ cnt8, uint64_t nlines, uint32_t sigma, uint32_t cbits, uint8_t* __restrict__ lines, unsigned long long* __restrict__ totals) {
    typedef hipcub::BlockScan<uint32_t, 256> Scan;
    __shared__ typename Scan::TempStorage tmp;
    const uint32_t sb = blockIdx.x, c = blockIdx.y;
    const uint64_t first = (uint64_t)sb << (cbits - 6), last = min(nlines, first + (1ull << (cbits - 6)));
    const uint32_t bitpos = c * cbits, w = 10u + (bitpos >> 5), sh = bitpos & 31u;
    unsigned long long run = 0;
    for (uint64_t base = first; base < last; base += 256u) {
        const uint64_t L = base + threadIdx.x;
        const uint32_t v = L < last ? cnt8[(size_t)c * nlines + L] : 0u;
        uint32_t before, sum;
        Scan(tmp).ExclusiveSum(v, before, sum);
        __syncthreads();
        if (L < last) {
            const uint32_t t = (uint32_t)run + before;              // (< 2^cbits: the rows of the super-block before the line)
            uint32_t* ln = reinterpret_cast<uint32_t*>(lines + L * 128u);
            if (t << sh) atomicOr(&ln[w], t << sh);
            if (sh + cbits > 32u && (t >> (32u - sh))) atomicOr(&ln[w + 1u], t >> (32u - sh));
        }
        run += sum;
    }
    if (threadIdx.x == 0) totals[(size_t)sb * sigma + c] = run;
}
__global__ void k_flat_super(const idx_t* __restrict__ C, const unsigned long long* __restrict__ totals, uint32_t nsb, uint32_t sigma, idx_t* __restrict__ super) {
    const uint32_t c = threadIdx.x;
    if (c >= sigma) return;
    idx_t run = C[c];
    for (uint32_t sb = 0; sb < nsb; ++sb) { super[(size_t)sb * sigma + c] = run; run += (idx_t)totals[(size_t)sb * sigma + c]; }
}
int build_flat_table(Index* x, hipStream_t stream) {
    DevString& s = x->bwt;
    // (beside a Wavelet, or EPR / EPRV2 blocks read in place: Format A reads one line per step and end already; n < 2^38: a line number fits 32 bits in k_exact_s)
    if (s.family == FAM_A || s.sigma < 6 || s.sigma > 29 || s.n < 2 || s.n >= (1ull << 38) || s.flat || !opt_on(FMGPU_OPT_SYMBOL_PLANES)) return 0;
    const uint64_t nlines = s.n / 64 + 1;
    const uint32_t sigma = (uint32_t)s.sigma, cbits = flat_count_bits(sigma);
    const uint32_t nsb = (uint32_t)(s.n >> cbits) + 1u;
    DBuf out, cnt8, totals, super; int rc;
    if ((rc = out.alloc(nlines * 128)) || (rc = cnt8.alloc(nlines * sigma)) || (rc = totals.alloc((size_t)nsb * sigma * 8)) || (rc = super.alloc((size_t)nsb * sigma * sizeof(idx_t)))) return rc;
    FM_HIP(hipMemsetAsync(out.p, 0, out.bytes, stream));
    dim3 grid; if ((rc = grid_of(nlines * 64, &grid, 1u << 22))) return rc;
    rc = dispatch_native(s, [&](auto occ, auto) {
        k_flat_planes<decltype(occ)><<<grid, dim3(256), 0, stream>>>(occ, s.n, sigma, out.as<uint8_t>(), cnt8.as<uint8_t>(), nlines);
        return 0;
    });
    if (rc) return rc;
    FM_LAUNCHED("k_flat_planes");
    k_flat_counts<<<dim3(nsb, sigma), dim3(256), 0, stream>>>(cnt8.as<uint8_t>(), nlines, sigma, cbits, out.as<uint8_t>(), totals.as<unsigned long long>());
    FM_LAUNCHED("k_flat_counts");
    k_flat_super<<<1, 32, 0, stream>>>(x->dC, totals.as<unsigned long long>(), nsb, sigma, super.as<idx_t>());
    FM_LAUNCHED("k_flat_super");
    FM_HIP(hipStreamSynchronize(stream));
    s.flat_bytes = out.bytes + super.bytes; s.flat = (uint8_t*)out.take(); s.flat_super = (idx_t*)super.take(); s.flat_nsb = nsb;
    x->device_bytes += s.flat_bytes;
    return 0;
}

// EPRV3 / EPRV4 / EPRV5 / InterleavedEPRV7 -> Format A.  thread = (block B, symbol c): the symbol-match mask of the bit planes
// (EPRV3.h:55-68) becomes the entry's bitmap (position p <-> bit p & 63, as in Format A), the counters of every level that
// cover row 64B (EPRV3.h:205-213, EPRV4.h:128-142, EPRV5.h:126-139, InterleavedEPRV7.h:190-199) are summed into cnt.
struct HierView {
    const uint8_t* bits; uint32_t bits_stride;
    const uint8_t* lev[3]; uint32_t lev_w[3], lev_shift[3], lev_stride[3], lev_off[3]; int nlev;
    const uint64_t* super; uint32_t sshift;
};
__global__ __launch_bounds__(256) void k_convert_hier(HierView v, const idx_t* __restrict__ C, uint8_t* __restrict__ out, uint64_t* __restrict__ out_super,
                                                      uint64_t nblocks, uint64_t n, uint32_t sigma, uint32_t bitct, uint32_t bstride) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nblocks * sigma; t += (uint64_t)gridDim.x * blockDim.x) {      // (strided: a launch holds fewer than 2^32 threads)
    const uint64_t B = t / sigma;
    uint32_t c = (uint32_t)(t % sigma);
    auto eval = [&](uint64_t b) -> uint64_t {
        const uint64_t row = b * 64;
        uint64_t cnt = v.super[(row >> v.sshift) * sigma + c];
        for (int L = 0; L < v.nlev; ++L) {
            const uint8_t* p = v.lev[L] + (row >> v.lev_shift[L]) * v.lev_stride[L] + v.lev_off[L] + (uint64_t)c * v.lev_w[L];
            if (v.lev_w[L] == 1) cnt += *p;
            else if (v.lev_w[L] == 2) { uint16_t x; memcpy(&x, p, 2); cnt += x; }
            else { uint32_t x; memcpy(&x, p, 4); cnt += x; }
        }
        return cnt + C[c];
    };
    const uint64_t row = B * 64;
    uint64_t m = ~0ull;
    for (uint32_t i = 0; i < bitct; ++i) {
        uint64_t w; memcpy(&w, v.bits + B * v.bits_stride + 8ull * i, 8);            // V7's packed structs are not 8-byte aligned
        m &= w ^ (0ull - (uint64_t)((~c >> i) & 1u));
    }
    if (n - row < 64) m &= (1ull << (n - row)) - 1ull;                                 // rows past the end read as symbol 0 in the planes
    put_entry_a(out, out_super, B, c, sigma, bstride, eval(B), kWide ? eval(super_first_block(B)) : 0, m);
    }
}

static int create_hier(const fmgpu_string_desc& d, const idx_t* dC, DevString& s) {
    const uint32_t sigma = (uint32_t)d.sigma, bitct = (uint32_t)bit_width_u((uint64_t)sigma - 1);
    HierView v{};
    v.bits_stride = 8 * bitct;
    switch (d.layout) {
    case FMGPU_EPRV3_8: case FMGPU_EPRV3_16: case FMGPU_EPRV3_32: {
        uint32_t bt = d.layout == FMGPU_EPRV3_8 ? 1 : (d.layout == FMGPU_EPRV3_16 ? 2 : 4);
        v.nlev = 1; v.lev_w[0] = bt; v.lev_shift[0] = 6; v.sshift = 8 * bt; break;
    }
    case FMGPU_EPRV4: v.nlev = 3; v.lev_w[0] = 1; v.lev_w[1] = 2; v.lev_w[2] = 4; v.lev_shift[0] = 6; v.lev_shift[1] = 8; v.lev_shift[2] = 16; v.sshift = 32; break;
    default:          v.nlev = 2; v.lev_w[0] = 1; v.lev_w[1] = 2; v.lev_shift[0] = 6; v.lev_shift[1] = 8; v.sshift = 16; break;
    }
    const bool v7 = d.layout == FMGPU_IEPRV7;
    if (v7) v.bits_stride += sigma;
    const uint64_t nblocks = d.n / 64 + 1, nsuper = (d.n >> v.sshift) + 1;
    if (!d.blocks || !d.super_blocks) return fail(FMGPU_ERR_INVALID, "bits / super_blocks missing");
    if (d.blocks_bytes < nblocks * v.bits_stride) return fail(FMGPU_ERR_INVALID, "bits array too short for n rows (expected >= " + std::to_string(nblocks * v.bits_stride) + " bytes)");
    if (d.n_super_blocks < nsuper) return fail(FMGPU_ERR_INVALID, "too few super blocks for n rows");
    void* held[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    auto drop = [&] { for (void* p : held) if (p) (void)hipFree(p); };
    int rc = upload(d.blocks, nblocks * v.bits_stride, &held[0]); if (rc) { drop(); return rc; }
    rc = upload(d.super_blocks, nsuper * sigma * 8, &held[1]); if (rc) { drop(); return rc; }
    v.bits = (const uint8_t*)held[0]; v.super = (const uint64_t*)held[1];
    for (int L = 0; L < v.nlev; ++L) {
        v.lev_stride[L] = v.lev_w[L] * sigma; v.lev_off[L] = 0;
        if (L == 0 && v7) { v.lev[0] = v.bits; v.lev_stride[0] = v.bits_stride; v.lev_off[0] = 8 * bitct; continue; }   // level0 inside the packed struct
        const uint64_t need = ((d.n >> v.lev_shift[L]) + 1) * v.lev_stride[L];
        if (!d.levels[L] || d.level_bytes[L] < need) { drop(); return fail(FMGPU_ERR_INVALID, "counter level " + std::to_string(L) + " missing or too short for n rows"); }
        rc = upload(d.levels[L], need, &held[2 + L]); if (rc) { drop(); return rc; }
        v.lev[L] = (const uint8_t*)held[2 + L];
    }
    if ((rc = alloc_format_a(s, d.n, sigma, dC))) { drop(); return rc; }
    dim3 grid;
    if ((rc = grid_of(nblocks * sigma, &grid, 1u << 22))) { drop(); return rc; }
    k_convert_hier<<<grid, dim3(256)>>>(v, dC, (uint8_t*)s.blk, (uint64_t*)s.sup, nblocks, d.n, sigma, bitct, s.va.bstride);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    drop();
    if (e != hipSuccess) return hip_fail(e, "k_convert_hier");
    s.bitct = (int)bitct;
    return 0;
}

// FlattenedBitvectors2L<sigma, l1_bits, 65536> -> Format A (FlattenedBitvectors2L.h:209-224): thread = (64-row block B, symbol c)
__global__ __launch_bounds__(256) void k_convert_fbv(const uint8_t* __restrict__ bits, const uint64_t* __restrict__ l0, const uint16_t* __restrict__ l1,
                                                     const idx_t* __restrict__ C, uint8_t* __restrict__ out, uint64_t* __restrict__ out_super, uint64_t nblocks, uint64_t n,
                                                     uint32_t sigma, uint32_t bitct, uint32_t l1_bits, uint32_t bstride) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nblocks * sigma; t += (uint64_t)gridDim.x * blockDim.x) {      // (strided: a launch holds fewer than 2^32 threads)
    const uint64_t B = t / sigma;
    const uint32_t c = (uint32_t)(t % sigma), sig1 = sigma + 1;
    auto eval = [&](uint64_t b, uint64_t* bits_out) -> uint64_t {
        const uint64_t row = b * 64, blk = row / l1_bits, sb = row >> 16;
        const uint32_t w = (uint32_t)((row % l1_bits) / 64);
        const uint8_t* base = bits + blk * ((uint64_t)bitct * l1_bits / 8);
        auto have = [&](uint32_t word) -> uint64_t {
            uint64_t m = ~0ull;
            for (uint32_t i = 0; i < bitct; ++i) {
                uint64_t v = *reinterpret_cast<const uint64_t*>(base + (uint64_t)i * (l1_bits / 8) + 8ull * word);
                m &= v ^ (0ull - (uint64_t)((~c >> i) & 1u));
            }
            return m;
        };
        uint64_t cnt = l0[sb * sig1 + c + 1] - l0[sb * sig1 + c] + (uint64_t)l1[blk * sig1 + c + 1] - (uint64_t)l1[blk * sig1 + c];
        for (uint32_t j = 0; j < w; ++j) cnt += (uint64_t)__popcll(have(j));
        if (bits_out) {
            uint64_t m = have(w);
            if (n - row < 64) m &= (1ull << (n - row)) - 1ull;
            *bits_out = m;
        }
        return cnt + C[c];
    };
    uint64_t m = 0;
    const uint64_t total = eval(B, &m);
    put_entry_a(out, out_super, B, c, sigma, bstride, total, kWide ? eval(super_first_block(B), nullptr) : 0, m);
    }
}

static int create_fbv(const fmgpu_string_desc& d, const idx_t* dC, DevString& s) {
    const uint32_t sigma = (uint32_t)d.sigma, bitct = (uint32_t)bit_width_u((uint64_t)sigma - 1);
    const uint32_t l1_bits = d.layout == FMGPU_FBV_64_64K ? 64u : (d.layout == FMGPU_FBV_512_64K ? 512u : 2048u);
    const uint64_t nsuper = d.n / 65536 + 1, nl1 = nsuper * (65536 / l1_bits), stride = (uint64_t)bitct * l1_bits / 8, nblocks = d.n / 64 + 1;
    if (!d.blocks || !d.super_blocks || !d.levels[0]) return fail(FMGPU_ERR_INVALID, "bits / l0 (super_blocks) / l1 (levels[0]) missing");
    if (d.blocks_bytes < nl1 * stride) return fail(FMGPU_ERR_INVALID, "bits array too short for n rows (expected " + std::to_string(nl1 * stride) + " bytes)");
    if (d.n_super_blocks < nsuper) return fail(FMGPU_ERR_INVALID, "too few l0 blocks for n rows");
    if (d.level_bytes[0] < nl1 * (sigma + 1) * 2) return fail(FMGPU_ERR_INVALID, "l1 array too short for n rows");
    void *db = nullptr, *d0 = nullptr, *d1 = nullptr;
    auto drop = [&] { if (db) (void)hipFree(db); if (d0) (void)hipFree(d0); if (d1) (void)hipFree(d1); };
    int rc = upload(d.blocks, nl1 * stride, &db); if (rc) { drop(); return rc; }
    rc = upload(d.super_blocks, nsuper * (sigma + 1) * 8, &d0); if (rc) { drop(); return rc; }
    rc = upload(d.levels[0], nl1 * (sigma + 1) * 2, &d1); if (rc) { drop(); return rc; }
    if ((rc = alloc_format_a(s, d.n, sigma, dC))) { drop(); return rc; }
    dim3 grid;
    if ((rc = grid_of(nblocks * sigma, &grid, 1u << 22))) { drop(); return rc; }
    k_convert_fbv<<<grid, dim3(256)>>>((const uint8_t*)db, (const uint64_t*)d0, (const uint16_t*)d1, dC, (uint8_t*)s.blk, (uint64_t*)s.sup,
                                       nblocks, d.n, sigma, bitct, l1_bits, s.va.bstride);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    drop();
    if (e != hipSuccess) return hip_fail(e, "k_convert_fbv");
    s.bitct = (int)bitct;
    return 0;
}

int on_handle_device(const Index* x) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return fail(FMGPU_ERR_NO_DEVICE, "no HIP device visible — the product path has no CPU fallback"); }
    if (dev != x->hdr.device) return fail(FMGPU_ERR_INVALID, "the handle lives on device " + std::to_string(x->hdr.device) + ", the calling thread's current device is " + std::to_string(dev));
    return 0;
}

void free_string(DevString& s) {
    for (void* p : {s.blk, s.aux, s.sup, (void*)s.lf_table, (void*)s.kblk, (void*)s.walk3, s.shadow, s.shadow_sup, (void*)s.slut, (void*)s.walkj, (void*)s.walk2j, s.dense, (void*)s.dense_ex, (void*)s.pairs, (void*)s.pairs_ex, (void*)s.pairs_super, (void*)s.flat, (void*)s.flat_super})
        if (p) (void)hipFree(p);
    s = DevString{};
}

template <class Occ>
__global__ __launch_bounds__(256) void k_lf_table(Occ occ, uint64_t n, idx_t* __restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t c;
        out[i] = occ.lf_symbol((idx_t)i, c);
    }
}

int build_lf_table(DevString& s, hipStream_t stream) {
    if (s.n == 0 || s.lf_table) return 0;
    DBuf t; int rc;
    if ((rc = t.alloc(s.n * sizeof(idx_t) + 16))) return rc;   // (+16: k_scheme_fast reads 16 bytes at a row)
    dim3 grid;
    if ((rc = grid_of(s.n, &grid, 1u << 22))) return rc;
    rc = dispatch_native(s, [&](auto occ, auto) {
        k_lf_table<decltype(occ)><<<grid, dim3(256), 0, stream>>>(occ, s.n, t.as<idx_t>());
        return 0;
    });
    FM_LAUNCHED("k_lf_table");
    FM_HIP(hipStreamSynchronize(stream));
    s.lf_table = (idx_t*)t.take();
    return 0;
}

template <class Occ>
__global__ __launch_bounds__(256) void k_symbols_w(OccW occ, uint64_t n, uint8_t* __restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = (uint8_t)occ.symbol((idx_t)i);
}

static int create_string(const fmgpu_string_desc& d, const idx_t* dC, DevString& s) {
    if (d.sigma < 2 || d.sigma > 256) return fail(FMGPU_ERR_INVALID, "sigma must be in [2, 256]");
    if (!kWide && d.n >= kNarrowLimit) return fail(FMGPU_ERR_UNSUPPORTED, "the 32-bit-row build indexes fewer than 2^32 - 64 rows per string");
    if (d.layout >= FMGPU_EPRV3_8 && d.layout <= FMGPU_IEPRV7) {
        s.layout = d.layout; s.sigma = d.sigma; s.n = d.n;
        return create_hier(d, dC, s);
    }
    if (d.layout >= FMGPU_FBV_64_64K && d.layout <= FMGPU_FBV_2048_64K) {
        s.layout = d.layout; s.sigma = d.sigma; s.n = d.n;
        return create_fbv(d, dC, s);
    }
    RefLayout L{};
    if (ref_layout(d.layout, d.sigma, L) != 0) return fail(FMGPU_ERR_INVALID, "unknown layout id");
    s.layout = d.layout; s.sigma = d.sigma; s.n = d.n; s.bitct = (int)L.bitct;
    const uint32_t sigma = (uint32_t)d.sigma;

    if (L.family <= 1) {   // InterleavedBitvector* / InterleavedBitvectorPrefix* -> Format A
        uint64_t nblocks = d.n / 64 + 1, nsuper = d.n / L.period + 1;
        if (!d.blocks || !d.super_blocks) return fail(FMGPU_ERR_INVALID, "blocks / super_blocks missing");
        if (d.blocks_bytes != nblocks * L.stride) return fail(FMGPU_ERR_INVALID, "blocks_bytes does not match n / layout (expected " + std::to_string(nblocks * L.stride) + ")");
        if (d.n_super_blocks != nsuper) return fail(FMGPU_ERR_INVALID, "n_super_blocks does not match n / layout");
        void *raw = nullptr, *sup = nullptr;
        int rc = upload(d.blocks, d.blocks_bytes, &raw); if (rc) return rc;
        rc = upload(d.super_blocks, nsuper * sigma * 8, &sup); if (rc) { (void)hipFree(raw); return rc; }
        auto drop = [&] { (void)hipFree(raw); (void)hipFree(sup); };
        if ((rc = alloc_format_a(s, d.n, sigma, dC))) { drop(); return rc; }
        dim3 grid;
        if ((rc = grid_of(nblocks * sigma, &grid, 1u << 22))) { drop(); return rc; }
        k_convert_ib<<<grid, dim3(256)>>>((const uint8_t*)raw, (const uint64_t*)sup, dC, (uint8_t*)s.blk, (uint64_t*)s.sup, nblocks, sigma, L.bt, L.bits_off, L.stride,
                                          L.period, s.va.bstride, L.family == 1 ? 1 : 0);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
        drop();
        if (e != hipSuccess) return hip_fail(e, "k_convert_ib");
        return 0;
    }
    if (L.family == 2 || L.family == 3) {   // EPR / EPRV2: reference layout verbatim
        if (!d.blocks || !d.super_blocks) return fail(FMGPU_ERR_INVALID, "blocks / super_blocks missing");
        if (d.blocks_bytes % L.stride != 0) return fail(FMGPU_ERR_INVALID, "blocks_bytes is not a multiple of sizeof(Block)");
        uint64_t need_blocks = d.n / L.rows + 1;
        if (d.blocks_bytes / L.stride < need_blocks) return fail(FMGPU_ERR_INVALID, "too few blocks for n rows");
        if (d.n_super_blocks < d.n / L.period + 1) return fail(FMGPU_ERR_INVALID, "too few super blocks for n rows");
        int rc = upload(d.blocks, d.blocks_bytes, &s.blk); if (rc) return rc;
        rc = upload(d.super_blocks, d.n_super_blocks * sigma * 8, &s.aux); if (rc) return rc;
        s.blk_bytes = d.blocks_bytes; s.aux_bytes = d.n_super_blocks * sigma * 8;
        s.family = L.family == 2 ? FAM_EPR : FAM_EPRV2;
        ViewR v{};
        v.blk = (const uint8_t*)s.blk; v.super = (const uint64_t*)s.aux; v.C = dC;
        v.stride = L.stride; v.bits_off = L.bits_off; v.bt = L.bt; v.sigma = sigma; v.bitct = L.bitct;
        v.rows = L.rows; v.period_shift = 8 * L.bt; v.period = (uint32_t)std::min<uint64_t>(L.period, 0xffffffffull);
        // InterleavedEPR.h:28-47
        uint64_t entries = 64 / L.bitct, cm = (1ull << L.bitct) - 1, mk = 1ull << L.bitct;
        for (uint64_t i = 0; i < entries; i += 2) { v.maskEven = (v.maskEven << (2 * L.bitct)) | cm; v.bitMask = (v.bitMask << (2 * L.bitct)) | mk; }
        s.vr = v;
        return 0;
    }
    // Wavelet: the node bitvectors (bitvector/Bitvector.h:31-34) are laid out as Format W lines on the host, the symbols are read back from them on
    // the device (string/Wavelet.h:77-102) and the multi-ary tree (Format M) is built from those; the lines are dropped again
    uint64_t nnodes = 1; while (nnodes < sigma) nnodes <<= 1;
    if (!d.nodes || d.n_nodes != nnodes) return fail(FMGPU_ERR_INVALID, "wavelet needs bit_ceil(sigma) node descriptors");
    std::vector<uint32_t> base(nnodes, 0);
    uint64_t total_lines = 0;
    for (uint64_t k = 0; k < nnodes; ++k) {
        base[k] = (uint32_t)total_lines;
        total_lines += d.nodes[k].total_length / 384 + 1;
        if (d.nodes[k].n_bits < d.nodes[k].total_length / 64 + 1) return fail(FMGPU_ERR_INVALID, "wavelet node bits array too short");
        if (total_lines >= 0xffffffffull) return fail(FMGPU_ERR_UNSUPPORTED, "wavelet too large for 32-bit line offsets");
    }
    if (d.nodes[0].total_length != d.n) return fail(FMGPU_ERR_INVALID, "wavelet root node length differs from n");
    std::unique_ptr<uint64_t[]> lines(new (std::nothrow) uint64_t[total_lines * 8]());
    if (!lines) return fail(FMGPU_ERR_NOMEM, "host staging for wavelet lines");
    for (uint64_t k = 0; k < nnodes; ++k) {
        const fmgpu_wavelet_node& nd = d.nodes[k];
        uint64_t nl = nd.total_length / 384 + 1, nwords = nd.total_length / 64 + 1, ones = 0;
        for (uint64_t li = 0; li < nl; ++li) {
            uint64_t* Lp = lines.get() + (base[k] + li) * 8;
            Lp[0] = ones;
            uint64_t cum = 0, h1 = 0;
            for (uint64_t j = 0; j < 6; ++j) {
                uint64_t wi = li * 6 + j;
                uint64_t w = wi < nwords ? nd.bits[wi] : 0;
                if (j) h1 |= cum << (9 * (j - 1));
                Lp[2 + j] = w;
                cum += (uint64_t)__builtin_popcountll(w);
            }
            Lp[1] = h1;
            ones += cum;
        }
    }
    DBuf dl, db, sym;
    int rc;
    if ((rc = dl.alloc(total_lines * 64)) || (rc = db.alloc(nnodes * 4)) || (rc = sym.alloc(d.n + 64))) return rc;
    FM_HIP(hipMemcpy(dl.p, lines.get(), total_lines * 64, hipMemcpyHostToDevice));
    FM_HIP(hipMemcpy(db.p, base.data(), nnodes * 4, hipMemcpyHostToDevice));
    lines.reset();
    if (d.n) {
        dim3 grid;
        if ((rc = grid_of(d.n, &grid, 1u << 22))) return rc;
        k_symbols_w<OccW><<<grid, dim3(256)>>>(OccW{ViewW{dl.as<uint64_t>(), db.as<uint32_t>(), dC, sigma, L.bitct}}, d.n, sym.as<uint8_t>());
        FM_LAUNCHED("k_symbols_w");
        FM_HIP(hipDeviceSynchronize());
    }
    dl.release(); db.release();
    return make_format_m(sym.as<uint8_t>(), d.n, sigma, dC, s, d.layout, nullptr);
}

constexpr uint64_t kTableGridCap = 1u << 22;      // blocks of the grid-stride table builders

// ------------------------------------------------------------------ exact-search tables, both row widths (entry shapes: fmgpu_common.h)
// suffix table: the interval of the L symbols c_0 (consumed first = the query's last symbol), c_1, ...
template <class Occ>
__global__ __launch_bounds__(256) void k_suffix_lut(Occ occ, uint64_t entries, uint32_t L, uint32_t R, idx_t n, void* __restrict__ lut) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < entries; w += (uint64_t)gridDim.x * blockDim.x) {   // (2^32 entries: more than one launch's threads)
        idx_t lb = 0, len = n;
        uint64_t rest = w;
        for (uint32_t t = 0; t < L && len != 0; ++t) {
            uint32_t c = (uint32_t)(rest % R) + 1; rest /= R;
            idx_t ra, rb;
            occ.lf2(lb, lb + len, c, ra, rb);
            lb = ra; len = rb - ra;
        }
        if constexpr (kWide) reinterpret_cast<ulonglong2*>(lut)[w] = make_ulonglong2(lb, len);
        else reinterpret_cast<uint2*>(lut)[w] = make_uint2(lb, len);
    }
}
// J LF steps from every row, remembering the symbols met
__global__ __launch_bounds__(256) void k_walkj(const idx_t* __restrict__ lf, const idx_t* __restrict__ C, uint32_t sigma, uint64_t n, uint32_t J, uint32_t bits,
                                               void* __restrict__ out) {
    __shared__ idx_t sC[257];
    for (uint32_t i = threadIdx.x; i <= sigma; i += blockDim.x) sC[i] = C[i];
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        idx_t r = (idx_t)i;
        uint32_t code = 0; bool ok = true;
        for (uint32_t t = 0; t < J; ++t) {
            idx_t nr = lf[r];
            uint32_t lo = 0, hi = sigma;                   // symbol of the step: C[s] <= LF < C[s+1]
            while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (sC[mid] <= nr) lo = mid; else hi = mid; }
            if (lo == 0) { ok = false; break; }
            code |= (lo - 1u) << (bits * t);
            r = nr;
        }
        if constexpr (kWide) reinterpret_cast<uint4*>(out)[i] = ok ? make_uint4((uint32_t)r, (uint32_t)((uint64_t)r >> 32), code, 0u) : make_uint4(0xffffffffu, 0xffffffffu, 0u, 0u);
        else reinterpret_cast<uint2*>(out)[i] = ok ? make_uint2((uint32_t)r, code) : make_uint2(0xffffffffu, 0u);
    }
}
// 2J steps = two J-step entries chained
__global__ __launch_bounds__(256) void k_walk2j(const void* __restrict__ wj_, uint64_t n, void* __restrict__ out_) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if constexpr (kWide) {
            const uint4* wj = reinterpret_cast<const uint4*>(wj_);
            const uint4 a = wj[i];
            uint4 r = make_uint4(0xffffffffu, 0xffffffffu, 0u, 0u);
            if (!(a.x == 0xffffffffu && a.y == 0xffffffffu)) {
                const uint4 b = wj[(uint64_t)a.x | ((uint64_t)a.y << 32)];
                if (!(b.x == 0xffffffffu && b.y == 0xffffffffu)) r = make_uint4(b.x, b.y, a.z, b.z);
            }
            reinterpret_cast<uint4*>(out_)[i] = r;
        } else {
            const uint2* wj = reinterpret_cast<const uint2*>(wj_);
            uint32_t* out = reinterpret_cast<uint32_t*>(out_);
            const uint2 a = wj[i];
            uint32_t r = 0xffffffffu, c0 = 0, c1 = 0;
            if (a.x != 0xffffffffu) { const uint2 b = wj[a.x]; if (b.x != 0xffffffffu) { r = b.x; c0 = a.y; c1 = b.y; } }
            out[3 * i] = r; out[3 * i + 1] = c0; out[3 * i + 2] = c1;
        }
    }
}

#if !FMGPU_WIDE
// ------------------------------------------------------------------ multi-symbol-step table
// context code of row j: walk K LF steps from j collecting the BWT symbols s_1 (immediately before the suffix), s_2, ...;
// w = s_K ... s_1 in text order, code = sum (s_t - 1) * R^(t-1) with R = sigma - 1; 255 if a delimiter is met.
template <class Occ>
__global__ __launch_bounds__(256) void k_kstep_codes(Occ occ, uint64_t n, uint32_t K, uint32_t R, uint8_t* __restrict__ code) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
        idx_t row = (idx_t)j;
        uint32_t c = 0, mul = 1; bool ok = true;
        for (uint32_t t = 0; t < K; ++t) {
            uint32_t s;
            row = occ.lf_symbol(row, s);
            if (s == 0) { ok = false; break; }
            c += (s - 1) * mul; mul *= R;
        }
        code[j] = ok ? (uint8_t)c : (uint8_t)255;
    }
}
// one wave per 64-row block: plane bits by ballot; per-block counts into cnt[w * nblocks + B]
__global__ __launch_bounds__(256) void k_kstep_bits(const uint8_t* __restrict__ code, uint64_t n, uint64_t nblocks, uint32_t ncodes,
                                                    uint8_t* __restrict__ kblk, uint32_t* __restrict__ cnt) {
    uint32_t lane = threadIdx.x & 63u;
    for (uint64_t B = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; B < nblocks; B += ((uint64_t)gridDim.x * blockDim.x) >> 6) {
        uint64_t row = B * 64 + lane;
        uint32_t s = row < n ? code[row] : 0xffffffffu;
        for (uint32_t c0 = 0; c0 < ncodes; c0 += 64) {
            uint64_t mine = 0;
            for (uint32_t c = c0; c < ncodes && c < c0 + 64; ++c) {
                uint64_t bits = __ballot(s == c);
                if (lane == c - c0) mine = bits;
            }
            uint32_t c = c0 + lane;
            if (c < ncodes) {
                uint32_t* o = reinterpret_cast<uint32_t*>(kblk + (B * ncodes + c) * 16ull);
                o[1] = (uint32_t)mine; o[2] = (uint32_t)(mine >> 32); o[3] = 0;
                cnt[(uint64_t)c * nblocks + B] = (uint32_t)__popcll(mine);
            }
        }
    }
}
// C_k[w] = LF_k(0, w): the k single steps of the context from row 0, last symbol first
template <class Occ>
__global__ void k_kstep_base(Occ occ, uint32_t K, uint32_t R, uint32_t ncodes, idx_t* __restrict__ base) {
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= ncodes) return;
    idx_t i = 0; uint32_t rest = w;
    for (uint32_t t = 0; t < K; ++t) { uint32_t s = rest % R + 1; rest /= R; i = occ.lf(i, s); }
    base[w] = i;
}
__global__ __launch_bounds__(256) void k_kstep_counts(const uint32_t* __restrict__ cnt, const idx_t* __restrict__ base, uint64_t nblocks, uint32_t ncodes,
                                                      uint8_t* __restrict__ kblk) {
    const uint64_t total = nblocks * ncodes;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t B = t % nblocks; uint32_t c = (uint32_t)(t / nblocks);
        *reinterpret_cast<uint32_t*>(kblk + (B * ncodes + c) * 16ull) = cnt[t] + base[c];
    }
}

// ------------------------------------------------------------------ search accelerators (prefix table, walk table)
__global__ __launch_bounds__(256) void k_walk3(const idx_t* __restrict__ lf, uint64_t n, idx_t* __restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        idx_t a = lf[i], b = lf[a], c = lf[b];
        out[3 * i] = a; out[3 * i + 1] = b; out[3 * i + 2] = c;
    }
}
// bidirectional interval of every string w of L symbols in [1, sigma): extendRight symbol by symbol (fmindex/BiFMIndexCursor.h:121-128)
template <class Occ>
__global__ __launch_bounds__(256) void k_prefix_lut(Occ rv, uint64_t entries, uint32_t L, uint32_t R, idx_t n, uint4* __restrict__ lut) {
    const uint32_t sigma = rv.sigma();
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < entries; w += (uint64_t)gridDim.x * blockDim.x) {   // (2^32 entries: more than one launch's threads)
        idx_t lb = 0, lbRev = 0, len = n;
        uint32_t used = 0;
        uint64_t rest = w;
        for (uint32_t t = 0; t < L && len != 0; ++t) {
            uint32_t c = (uint32_t)(rest % R) + 1; rest /= R;
            idx_t pre = 0, ra = 0, rb = 0;
            for (uint32_t d = 0; d <= c && d < sigma; ++d) {
                idx_t x, y;
                rv.lf2(lbRev, lbRev + len, d, x, y);
                if (d < c) pre += y - x; else { ra = x; rb = y; }
            }
            lb += pre; lbRev = ra; len = rb - ra;
            ++used;
        }
        lut[w] = make_uint4(lb, lbRev, len, used);
    }
}


// the table is built into local allocations and installed in the handle only after every launch and the final synchronisation have succeeded
template <class Occ>
static int accelerate_with(DevString& s, Occ occ, uint32_t K) {
    const uint32_t R = (uint32_t)s.sigma - 1;
    uint64_t nc = 1;
    for (uint32_t t = 0; t < K; ++t) { nc *= R; if (nc > 255) return fail(FMGPU_ERR_UNSUPPORTED, "(sigma-1)^kstep must be <= 255"); }
    const uint32_t ncodes = (uint32_t)nc;
    const uint64_t n = s.n, nblocks = n / 64 + 1;
    DBuf code, cnt, base, kblk, tmp;
    int rc;
    if ((rc = code.alloc(n + 64)) || (rc = cnt.alloc((size_t)ncodes * nblocks * 4)) || (rc = base.alloc(ncodes * sizeof(idx_t))) ||
        (rc = kblk.alloc((size_t)nblocks * ncodes * 16 + 128))) return rc;
    dim3 g_rows, g_waves, g_all;
    if ((rc = grid_of(n, &g_rows, kTableGridCap)) || (rc = grid_of(nblocks * 64, &g_waves, kTableGridCap)) || (rc = grid_of(nblocks * ncodes, &g_all, kTableGridCap))) return rc;
    k_kstep_codes<Occ><<<g_rows, 256>>>(occ, n, K, R, code.as<uint8_t>());
    FM_LAUNCHED("k_kstep_codes");
    k_kstep_bits<<<g_waves, 256>>>(code.as<uint8_t>(), n, nblocks, ncodes, kblk.as<uint8_t>(), cnt.as<uint32_t>());
    FM_LAUNCHED("k_kstep_bits");
    k_kstep_base<Occ><<<dim3((ncodes + 63) / 64), 64>>>(occ, K, R, ncodes, base.as<idx_t>());
    FM_LAUNCHED("k_kstep_base");
    size_t tb = 0;
    FM_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, cnt.as<uint32_t>(), cnt.as<uint32_t>(), (size_t)nblocks));
    if ((rc = tmp.alloc(tb ? tb : 8))) return rc;
    for (uint32_t c = 0; c < ncodes; ++c) {
        uint32_t* p = cnt.as<uint32_t>() + (uint64_t)c * nblocks;
        size_t b2 = tb;
        FM_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, b2, p, p, (size_t)nblocks));
    }
    k_kstep_counts<<<g_all, 256>>>(cnt.as<uint32_t>(), base.as<idx_t>(), nblocks, ncodes, kblk.as<uint8_t>());
    FM_LAUNCHED("k_kstep_counts");
    FM_HIP(hipDeviceSynchronize());
    if (s.kblk) (void)hipFree(s.kblk);
    s.kblk_bytes = kblk.bytes; s.kblk = (uint8_t*)kblk.take(); s.kstep = K; s.kcodes = ncodes;
    return 0;
}
#endif  // !FMGPU_WIDE

// ------------------------------------------------------------------ String_c batch kernel
template <class Occ>
__global__ __launch_bounds__(256) void k_string_query(Occ occ, const uint64_t* __restrict__ idx, const uint8_t* __restrict__ symb,
                                                      const uint8_t* __restrict__ what, uint64_t count, uint64_t n, uint64_t* __restrict__ out) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    uint32_t c = symb ? symb[t] : 0;
    uint32_t w = what ? what[t] : 0;
    const uint64_t i64 = idx[t];
    uint64_t r = ~0ull;                                   // (a position the reference would read out of bounds)
    if (i64 <= n && (w != 2 || i64 < n) && (w == 2 || c < occ.sigma() || (w == 1 && c == occ.sigma()))) {
        const idx_t i = (idx_t)i64;
        if (w == 0) r = occ.rank(i, c);
        else if (w == 1) r = c == occ.sigma() ? (uint64_t)i : (uint64_t)occ.prefix_rank(i, c);
        else r = occ.symbol(i);
    }
    out[t] = r;
}

// ------------------------------------------------------------------ cursor steps (fmindex/BiFMIndexCursor.h:58-128, fmindex/FMIndexCursor.h:33-53)
// one thread per cursor: extendLeft(c) / extendRight(c) (symb != null) or extendLeft() / extendRight() over all symbols (sigma outputs per cursor)
template <class Occ, int MAXSIG>
__global__ __launch_bounds__(256) void k_cursor_extend(Occ fw, Occ rv, bool bidir, int right, uint64_t count, const uint64_t* __restrict__ lb, const uint64_t* __restrict__ lb_rev,
                                                       const uint64_t* __restrict__ len, const uint8_t* __restrict__ symb, uint64_t n,
                                                       uint64_t* __restrict__ out_lb, uint64_t* __restrict__ out_lb_rev, uint64_t* __restrict__ out_len) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint32_t sigma = fw.sigma();
    const uint64_t a0 = lb[t], r0 = lb_rev ? lb_rev[t] : 0, l0 = len[t];
    const bool valid = a0 <= n && l0 <= n - a0 && r0 <= n && l0 <= n - r0;
    const Occ& occ = right ? rv : fw;
    const idx_t a = (idx_t)(right ? r0 : a0), b = (idx_t)(a + l0);
    if (symb) {
        const uint32_t c = symb[t];
        uint64_t nlb = 0, nrev = 0, nlen = 0;
        if (valid && c < sigma) {
            idx_t ra, rb, pre = 0;
            occ.lf2(a, b, c, ra, rb);
            if (bidir) pre = occ.prefix_rank(b, c) - occ.prefix_rank(a, c);
            nlen = rb - ra;
            if (right) { nrev = ra; nlb = a0 + pre; } else { nlb = ra; nrev = bidir ? r0 + pre : 0; }
        }
        out_lb[t] = nlb; if (out_lb_rev) out_lb_rev[t] = nrev; out_len[t] = nlen;
        return;
    }
    idx_t pre = 0;
    for (uint32_t c = 0; c < sigma; ++c) {
        uint64_t nlb = 0, nrev = 0, nlen = 0;
        if (valid) {
            idx_t ra, rb;
            occ.lf2(a, b, c, ra, rb);
            nlen = rb - ra;
            if (right) { nrev = ra; nlb = a0 + pre; } else { nlb = ra; nrev = bidir ? r0 + pre : 0; }
            pre += rb - ra;
        }
        out_lb[t * sigma + c] = nlb; if (out_lb_rev) out_lb_rev[t * sigma + c] = nrev; out_len[t * sigma + c] = nlen;
    }
}

namespace api {
#include "fmgpu_api_decl.h"

int fmgpu_index_destroy(fmgpu_index_t h) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return 0;
    free_string(x->bwt); free_string(x->rev);
    for (void* p : {(void*)x->dC, x->sa_l0, x->sa_l1, x->sa_bits, x->sa_f0, x->sa_f1, (void*)x->lut, (void*)x->loc_tab}) if (p) (void)hipFree(p);
    x->hdr.magic = 0;
    delete x;
    return 0;
}

static bool lf_table_wanted() { return opt_on(FMGPU_OPT_LF_TABLE); }

int fmgpu_index_create(const fmgpu_index_desc* desc, fmgpu_index_t* out) {
    if (!desc || !out) return fail(FMGPU_ERR_INVALID, "desc / out is null");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { (void)hipGetLastError(); return fail(FMGPU_ERR_NO_DEVICE, "no HIP device visible — the product path has no CPU fallback"); }
    if (!desc->C) return fail(FMGPU_ERR_INVALID, "C is null");
    if (desc->bwt_rev && (desc->bwt_rev->n != desc->bwt.n || desc->bwt_rev->sigma != desc->bwt.sigma))
        return fail(FMGPU_ERR_INVALID, "bwt don't have the same size: " + std::to_string(desc->bwt.n) + " " + std::to_string(desc->bwt_rev->n));   // fmindex/BiFMIndex.h:48-50
    std::unique_ptr<Index> x(new (std::nothrow) Index());
    if (!x) return fail(FMGPU_ERR_NOMEM, "host allocation");
    (void)hipGetDevice(&x->hdr.device);
    const int sigma = desc->bwt.sigma;
    if (sigma < 2 || sigma > 256) return fail(FMGPU_ERR_INVALID, "sigma must be in [2, 256]");
    std::vector<idx_t> cdev(sigma + 1);
    for (int i = 0; i <= sigma; ++i) {
        if (desc->C[i] > desc->bwt.n) return fail(FMGPU_ERR_INVALID, "C[] entry exceeds n");
        x->hC[i] = desc->C[i]; cdev[i] = (idx_t)desc->C[i];
    }
    int rc = upload(cdev.data(), (sigma + 1) * sizeof(idx_t), (void**)&x->dC);
    auto bail = [&](int code) { api::fmgpu_index_destroy(reinterpret_cast<fmgpu_index_t>(x.release())); return code; };
    if (rc) return bail(rc);
    rc = create_string(desc->bwt, x->dC, x->bwt); if (rc) return bail(rc);
    if (desc->bwt_rev) { rc = create_string(*desc->bwt_rev, x->dC, x->rev); if (rc) return bail(rc); x->bidirectional = true; }
    rc = auto_shadow(x.get(), nullptr); if (rc) return bail(rc);
    if (x->bidirectional) for (DevString* t : {&x->bwt, &x->rev}) { rc = build_dense_dna(*t, nullptr); if (rc) return bail(rc); }
    if (x->bwt.dense && !x->rev.dense) { (void)hipFree(x->bwt.dense); (void)hipFree(x->bwt.dense_ex); x->bwt.dense = nullptr; x->bwt.dense_ex = nullptr; x->bwt.dense_bytes = 0; x->bwt.dense_nex = 0; }
    if (lf_table_wanted()) {
        rc = build_lf_table(x->bwt, nullptr); if (rc) return bail(rc);
        if (x->bidirectional) { rc = build_lf_table(x->rev, nullptr); if (rc) return bail(rc); }
    }
    x->device_bytes = x->bwt.blk_bytes + x->bwt.aux_bytes + x->bwt.sup_bytes + x->rev.blk_bytes + x->rev.aux_bytes + x->rev.sup_bytes + x->bwt.dense_bytes + x->rev.dense_bytes +
                      x->bwt.shadow_bytes + x->rev.shadow_bytes +
                      (x->bwt.lf_table ? x->bwt.n * sizeof(idx_t) : 0) + (x->rev.lf_table ? x->rev.n * sizeof(idx_t) : 0);
    rc = build_pair_table(x.get(), nullptr); if (rc) return bail(rc);
    rc = build_flat_table(x.get(), nullptr); if (rc) return bail(rc);
    if (const fmgpu_sparse_array_desc* sa = desc->annotated_array) {
        if (sa->n != desc->bwt.n) return bail(fail(FMGPU_ERR_INVALID, "annotated_array.n != bwt.n"));
        if (sa->n_l0 < sa->n / 65536 + 1 || sa->n_l1 < sa->n / 512 + 1 || sa->n_bit_words < (sa->n / 512 + 1) * 8)
            return bail(fail(FMGPU_ERR_INVALID, "annotated_array presence bitvector arrays too short"));
        for (int f = 0; f < 2; ++f)
            if (sa->field[f].bits == 0 || sa->field[f].bits > 64 || sa->field[f].common_divisor == 0)
                return bail(fail(FMGPU_ERR_INVALID, "annotated_array dense vector has bits == 0 or > 64"));
        rc = upload(sa->l0, sa->n_l0 * 8, &x->sa_l0); if (rc) return bail(rc);
        rc = upload(sa->l1, sa->n_l1 * 2, &x->sa_l1); if (rc) return bail(rc);
        rc = upload(sa->bits, sa->n_bit_words * 8, &x->sa_bits); if (rc) return bail(rc);
        // one spare word so that the two-word read of dense_access never leaves the buffer
        for (int f = 0; f < 2; ++f) {
            void** dst = f == 0 ? &x->sa_f0 : &x->sa_f1;
            size_t bytes = (sa->field[f].n_words + 1) * 8;
            hipError_t e = hipMalloc(dst, bytes);
            if (e != hipSuccess) return bail(hip_fail(e, "hipMalloc(sa field)"));
            (void)hipMemset(*dst, 0, bytes);
            if (sa->field[f].n_words) {
                e = hipMemcpy(*dst, sa->field[f].data, sa->field[f].n_words * 8, hipMemcpyDefault);
                if (e != hipSuccess) return bail(hip_fail(e, "hipMemcpy(sa field)"));
            }
            x->device_bytes += bytes;
        }
        x->device_bytes += sa->n_l0 * 8 + sa->n_l1 * 2 + sa->n_bit_words * 8;
        x->sa_bytes[0] = sa->n_l0 * 8; x->sa_bytes[1] = sa->n_l1 * 2; x->sa_bytes[2] = sa->n_bit_words * 8;
        x->sa_bytes[3] = (sa->field[0].n_words + 1) * 8; x->sa_bytes[4] = (sa->field[1].n_words + 1) * 8;
        x->vsa = ViewSA{(const uint64_t*)x->sa_l0, (const uint16_t*)x->sa_l1, (const uint64_t*)x->sa_bits,
                        (const uint64_t*)x->sa_f0, (const uint64_t*)x->sa_f1,
                        sa->field[0].bits, sa->field[1].bits, sa->field[0].common_divisor, sa->field[1].common_divisor};
        x->has_sa = true;
        rc = fuse_presence_bits(x.get(), nullptr); if (rc) return bail(rc);
    }
    *out = reinterpret_cast<fmgpu_index_t>(x.release());
    return 0;
}

int fmgpu_index_accelerate_lf(fmgpu_index_t h, int32_t enable) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (int drc = on_handle_device(x)) return drc;
    for (DevString* s : {&x->bwt, &x->rev}) {
        if (s->n == 0) continue;
        if (!enable) {
            if (s->walk3 || s->walkj || s->walk2j) return fail(FMGPU_ERR_INVALID, "the walk tables are built on the LF table: drop them first (fmgpu_index_accelerate_exact / _search with walk = 0)");
            if (s->lf_table) { (void)hipFree(s->lf_table); s->lf_table = nullptr; x->device_bytes -= s->n * sizeof(idx_t); }
        } else if (!s->lf_table) {
            int rc = build_lf_table(*s, nullptr); if (rc) return rc;
            x->device_bytes += s->n * sizeof(idx_t);
        }
    }
    return 0;
}

// dropping the Format A expansion of a string takes the tables derived from it along (pair lines, dense DNA blocks: no kernel could reach them any more)
static void drop_shadow(Index* x, DevString& t) {
    if (!t.shadow) return;
    (void)hipFree(t.shadow); if (t.shadow_sup) (void)hipFree(t.shadow_sup);
    x->device_bytes -= t.shadow_bytes;
    t.shadow = nullptr; t.shadow_sup = nullptr; t.shadow_bytes = 0; t.shadow_sup_bytes = 0; t.va = ViewA{};
    if (t.pairs) { (void)hipFree(t.pairs); if (t.pairs_ex) (void)hipFree(t.pairs_ex); if (t.pairs_super) (void)hipFree(t.pairs_super);
                   x->device_bytes -= t.pairs_bytes; t.pairs = nullptr; t.pairs_ex = nullptr; t.pairs_super = nullptr; t.pairs_bytes = 0; t.pairs_nex = 0; t.pairs_nsb = 0; }
    if (t.dense) { (void)hipFree(t.dense); if (t.dense_ex) (void)hipFree(t.dense_ex);
                   x->device_bytes -= t.dense_bytes; t.dense = nullptr; t.dense_ex = nullptr; t.dense_bytes = 0; t.dense_nex = 0; }
}
// ... and (re)building it brings them back, as at creation: presence bits fused into the blocks, dense DNA blocks, pair lines
static int rebuild_derived(Index* x) {
    int rc;
    if ((rc = fuse_presence_bits(x, nullptr))) return rc;
    if (x->bidirectional) {
        for (DevString* t : {&x->bwt, &x->rev}) { const size_t had = t->dense_bytes; if ((rc = build_dense_dna(*t, nullptr))) return rc; x->device_bytes += t->dense_bytes - had; }
        if (x->bwt.dense && !x->rev.dense) { (void)hipFree(x->bwt.dense); (void)hipFree(x->bwt.dense_ex); x->device_bytes -= x->bwt.dense_bytes; x->bwt.dense = nullptr; x->bwt.dense_ex = nullptr; x->bwt.dense_bytes = 0; x->bwt.dense_nex = 0; }
    }
    return build_pair_table(x, nullptr);
}

#if FMGPU_WIDE
static int no_wide(const char* what) { return fail(FMGPU_ERR_UNSUPPORTED, std::string(what) + " is not available for indices of 2^32 rows or more (64-bit-row build): searches run on the plain occurrence tables"); }
int fmgpu_index_accelerate(fmgpu_index_t h, int32_t kstep) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (kstep < 0 || kstep > 8) return fail(FMGPU_ERR_INVALID, "kstep must be in [0, 8]");
    if (kstep <= 1) {                                      // Format A expansion of EPR / Wavelet strings (kstep = 1), or dropping it (0): available
        if (int drc = on_handle_device(x)) return drc;
        if (kstep == 0) {
            for (DevString* t : {&x->bwt, &x->rev}) drop_shadow(x, *t);
            return 0;
        }
        bool built = false;
        for (DevString* t : {&x->bwt, &x->rev}) {
            if (t->n == 0 || t->family == FAM_A || t->shadow) continue;
            int rc = build_format_a_shadow(*t, x->dC, nullptr); if (rc) return rc;
            x->device_bytes += t->shadow_bytes; built = true;
        }
        return built ? rebuild_derived(x) : 0;
    }
    return no_wide("the multi-symbol-step table");
}
int fmgpu_index_accelerate_search(fmgpu_index_t h, int32_t prefix_len, int32_t walk) {
    if (!h) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (prefix_len < 0 || prefix_len > 32) return fail(FMGPU_ERR_INVALID, "prefix_len must be in [0, 32]");
    if (prefix_len > 0 || walk) return no_wide("the prefix / walk tables");
    return 0;
}
#else
int fmgpu_index_accelerate(fmgpu_index_t h, int32_t kstep) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (int drc = on_handle_device(x)) return drc;
    DevString& s = x->bwt;
    if (kstep < 0 || kstep > 8) return fail(FMGPU_ERR_INVALID, "kstep must be in [0, 8]");
    x->device_bytes -= s.kblk_bytes;
    if (s.kblk) { (void)hipFree(s.kblk); s.kblk = nullptr; s.kstep = s.kcodes = 0; s.kblk_bytes = 0; }
    if (kstep == 0) {                                  // drop the Format A shadows as well
        for (DevString* t : {&x->bwt, &x->rev}) drop_shadow(x, *t);
        return 0;
    }
    if (s.n == 0) return 0;
    int rc = 0;
    bool built = false;
    for (DevString* t : {&x->bwt, &x->rev}) {          // EPR / Wavelet strings: searches read a Format A expansion from here on
        if (t->n == 0 || t->family == FAM_A || t->shadow) continue;
        if ((rc = build_format_a_shadow(*t, x->dC, nullptr))) return rc;
        x->device_bytes += t->shadow_bytes; built = true;
    }
    if (built && (rc = rebuild_derived(x))) return rc;
    if (kstep <= 1) return 0;
    rc = dispatch_occ(s, [&](auto occ, auto) { return accelerate_with(s, occ, (uint32_t)kstep); });
    if (rc == 0) x->device_bytes += s.kblk_bytes;
    return rc;
}

int fmgpu_index_accelerate_search(fmgpu_index_t h, int32_t prefix_len, int32_t walk) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (int drc = on_handle_device(x)) return drc;
    if (!x->bidirectional) return fail(FMGPU_ERR_INVALID, "search accelerators need a BiFMIndex");
    if (prefix_len < 0 || prefix_len > 32) return fail(FMGPU_ERR_INVALID, "prefix_len must be in [0, 32]");
    const uint64_t n = x->bwt.n;
    // drop what exists
    if (x->lut) { (void)hipFree(x->lut); x->device_bytes -= x->lut_entries * 16; x->lut = nullptr; x->lut_len = 0; x->lut_entries = 0; }
    for (DevString* s : {&x->bwt, &x->rev}) if (s->walk3) { (void)hipFree(s->walk3); s->walk3 = nullptr; x->device_bytes -= n * 12; }
    if (x->rev.walkj) { (void)hipFree(x->rev.walkj); x->rev.walkj = nullptr; x->rev.walk_J = 0; x->device_bytes -= n * 8; }
    if (n == 0) return 0;
    int rc;
    if (walk) {
        const uint32_t sigma = (uint32_t)x->bwt.sigma;
        uint32_t bits = 1; while ((1u << bits) < sigma - 1) ++bits;
        dim3 grid;
        if ((rc = grid_of(n, &grid, kTableGridCap))) return rc;
        for (DevString* s : {&x->bwt, &x->rev}) {
            if (!s->lf_table) { if ((rc = build_lf_table(*s, nullptr))) return rc; x->device_bytes += n * sizeof(idx_t); }
            if (walk & 1) {
                DBuf w3;
                if ((rc = w3.alloc(n * 12 + 16))) return rc;
                k_walk3<<<grid, 256>>>(s->lf_table, n, w3.as<idx_t>());
                FM_LAUNCHED("k_walk3");
                FM_HIP(hipDeviceSynchronize());
                s->walk3 = (idx_t*)w3.take();
                x->device_bytes += n * 12;
            }
            if ((walk & 2) && !s->walkj) {                          // (the forward one may exist already: fmgpu_index_accelerate_exact)
                DBuf wj;
                if ((rc = wj.alloc(n * 8 + 16))) return rc;
                k_walkj<<<grid, 256>>>(s->lf_table, x->dC, sigma, n, 32u / bits, bits, wj.p);
                FM_LAUNCHED("k_walkj");
                FM_HIP(hipDeviceSynchronize());
                s->walkj = (uint2*)wj.take(); s->walk_J = 32u / bits; s->walk_bits = bits;
                x->device_bytes += n * 8;
            }
        }
    }
    if (prefix_len > 0) {
        const uint32_t R = (uint32_t)x->bwt.sigma - 1;
        uint64_t entries = 1;
        for (int t = 0; t < prefix_len; ++t) { entries *= R; if (entries > (1ull << 32)) return fail(FMGPU_ERR_UNSUPPORTED, "prefix table would exceed 2^32 entries"); }
        DBuf lut;
        if ((rc = lut.alloc(entries * 16 + 64))) return rc;            // (+ slack: the lean kernel reads a block's worth behind an entry)
        dim3 grid;
        if ((rc = grid_of(entries, &grid, kTableGridCap))) return rc;
        rc = dispatch_native(x->rev, [&](auto occ, auto) {
            k_prefix_lut<decltype(occ)><<<grid, dim3(256)>>>(occ, entries, (uint32_t)prefix_len, R, (idx_t)n, lut.as<uint4>());
            return 0;
        });
        FM_LAUNCHED("k_prefix_lut");
        FM_HIP(hipDeviceSynchronize());
        x->lut = (uint4*)lut.take(); x->lut_len = (uint32_t)prefix_len; x->lut_entries = entries;
        x->device_bytes += entries * 16;
    }
    return 0;
}
#endif  // FMGPU_WIDE

// the interval and walk tables of the exact search, both row widths (entry shapes: fmgpu_common.h)
int fmgpu_index_accelerate_exact(fmgpu_index_t h, int32_t kstep, int32_t lut_len, int32_t walk) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (int drc = on_handle_device(x)) return drc;
    if (lut_len < 0 || lut_len > 32) return fail(FMGPU_ERR_INVALID, "lut_len must be in [0, 32]");
    int rc = api::fmgpu_index_accelerate(h, kstep);    // (64-bit rows: kstep <= 1 — the Format A expansion; the multi-symbol-step table holds 32-bit counts)
    if (rc) return rc;
    DevString& s = x->bwt;
    const uint64_t n = s.n;
    if (s.slut) { (void)hipFree(s.slut); x->device_bytes -= s.slut_entries * kSlutEntryBytes; s.slut = nullptr; s.slut_len = 0; s.slut_entries = 0; }
    if (s.walk2j) { (void)hipFree(s.walk2j); x->device_bytes -= n * kWalk2EntryBytes; s.walk2j = nullptr; }
    if (s.walkj && !walk) { (void)hipFree(s.walkj); x->device_bytes -= n * kWalkEntryBytes; s.walkj = nullptr; s.walk_J = 0; }
    if (n == 0) return 0;
    const uint32_t sigma = (uint32_t)s.sigma, R = sigma - 1;
    if (lut_len > 0) {
        uint64_t entries = 1;
        for (int t = 0; t < lut_len; ++t) { entries *= R; if (entries > (1ull << 32)) return fail(FMGPU_ERR_UNSUPPORTED, "suffix table would exceed 2^32 entries"); }
        DBuf lut;
        if ((rc = lut.alloc(entries * kSlutEntryBytes))) return rc;
        dim3 grid;
        if ((rc = grid_of(entries, &grid, kTableGridCap))) return rc;
        rc = dispatch_occ(s, [&](auto occ, auto) {
            k_suffix_lut<decltype(occ)><<<grid, dim3(256)>>>(occ, entries, (uint32_t)lut_len, R, (idx_t)n, lut.p);
            return 0;
        });
        FM_LAUNCHED("k_suffix_lut");
        FM_HIP(hipDeviceSynchronize());
        s.slut = (uint2*)lut.take(); s.slut_len = (uint32_t)lut_len; s.slut_entries = entries;
        x->device_bytes += entries * kSlutEntryBytes;
    }
    if (walk) {
        if (!s.lf_table) { if ((rc = build_lf_table(s, nullptr))) return rc; x->device_bytes += n * sizeof(idx_t); }
        uint32_t bits = 1; while ((1u << bits) < R) ++bits;               // symbols 1 .. sigma-1 stored as 0 .. sigma-2
        const uint32_t J = 32u / bits;
        dim3 grid;
        if ((rc = grid_of(n, &grid, kTableGridCap))) return rc;
        if (!s.walkj) {
            DBuf wj;
            if ((rc = wj.alloc(n * kWalkEntryBytes + 16))) return rc;
            k_walkj<<<grid, 256>>>(s.lf_table, x->dC, sigma, n, J, bits, wj.p);
            FM_LAUNCHED("k_walkj");
            FM_HIP(hipDeviceSynchronize());
            s.walkj = (uint2*)wj.take(); s.walk_J = J; s.walk_bits = bits;
            x->device_bytes += n * kWalkEntryBytes;
        }
        if (walk >= 2) {
            DBuf w2;
            if ((rc = w2.alloc(n * kWalk2EntryBytes + 16))) return rc;
            k_walk2j<<<grid, 256>>>(s.walkj, n, w2.p);
            FM_LAUNCHED("k_walk2j");
            FM_HIP(hipDeviceSynchronize());
            s.walk2j = (uint32_t*)w2.take();
            x->device_bytes += n * kWalk2EntryBytes;
        }
    }
    return 0;
}


int fmgpu_index_info(fmgpu_index_t h, uint64_t* n, int32_t* sigma, int32_t* layout, int32_t* bidirectional, uint64_t* device_bytes) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (n) *n = x->bwt.n;
    if (sigma) *sigma = x->bwt.sigma;
    if (layout) *layout = x->bwt.layout;
    if (bidirectional) *bidirectional = x->bidirectional ? 1 : 0;
    if (device_bytes) *device_bytes = x->device_bytes;
    return 0;
}

int fmgpu_index_formats(fmgpu_index_t h, uint32_t* mask) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x || !mask) return fail(FMGPU_ERR_INVALID, "index handle / mask is null");
    const DevString& s = x->bwt;
    uint32_t m = 0;
    if (s.search_family() == FAM_A) m |= FMGPU_FMT_BLOCKS;
    if (s.pairs) m |= FMGPU_FMT_PAIRS;
    if (s.dense) m |= FMGPU_FMT_DENSE;
    if (s.flat) m |= FMGPU_FMT_PLANES;
    if (s.family == FAM_WAVELET) m |= FMGPU_FMT_TREE;
    if (s.family == FAM_EPR || s.family == FAM_EPRV2) m |= FMGPU_FMT_REFERENCE;
    if (s.lf_table) m |= FMGPU_FMT_LF;
    if (s.kblk) m |= FMGPU_FMT_KSTEP;
    if (s.slut) m |= FMGPU_FMT_INTERVALS;
    if (s.walkj || s.walk3) m |= FMGPU_FMT_WALK;
    if (x->lut) m |= FMGPU_FMT_PREFIX;
    if (x->loc_tab) m |= FMGPU_FMT_LOCATE;
    if (s.va.fused) m |= FMGPU_FMT_FUSED;
    *mask = m;
    return 0;
}

int fmgpu_string_query(fmgpu_index_t h, int which, const uint64_t* idx, const uint8_t* symb, const uint8_t* what,
                       uint64_t count, uint64_t* out, void* stream_) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (int drc = on_handle_device(x)) return drc;
    if (which != 0 && !(which == 1 && x->bidirectional)) return fail(FMGPU_ERR_INVALID, "which must be 0 (bwt) or 1 (bwtRev of a BiFMIndex)");
    if (count == 0) return 0;
    if (!idx || !out) return fail(FMGPU_ERR_INVALID, "idx / out is null");
    hipStream_t stream = (hipStream_t)stream_;
    const DevString& s = which ? x->rev : x->bwt;
    Staged sidx, ssym, swhat, sout;
    int rc;
    if ((rc = sidx.in(idx, count * 8, stream))) return rc;
    if ((rc = ssym.in(symb, symb ? count : 0, stream))) return rc;
    if ((rc = swhat.in(what, what ? count : 0, stream))) return rc;
    if ((rc = sout.out(out, count * 8, stream))) return rc;
    FM_GRID(grid, count);
    auto a = (const uint64_t*)sidx.dev; auto b = (const uint8_t*)ssym.dev; auto c = (const uint8_t*)swhat.dev; auto o = (uint64_t*)sout.dev;
    rc = dispatch_native(s, [&](auto occ, auto) {
        k_string_query<decltype(occ)><<<grid, dim3(256), 0, stream>>>(occ, a, b, c, count, s.n, o);
        return 0;
    });
    FM_LAUNCHED("k_string_query");
    return sout.finish();
}

int fmgpu_cursor_extend(fmgpu_index_t h, int32_t direction, uint64_t count, const uint64_t* lb, const uint64_t* lb_rev, const uint64_t* len, const uint8_t* symb,
                        uint64_t* out_lb, uint64_t* out_lb_rev, uint64_t* out_len, void* stream_) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (int drc = on_handle_device(x)) return drc;
    if (direction != 0 && direction != 1) return fail(FMGPU_ERR_INVALID, "direction must be 0 (extendLeft) or 1 (extendRight)");
    if (direction == 1 && !x->bidirectional) return fail(FMGPU_ERR_INVALID, "extendRight needs a BiFMIndex");
    if (count == 0) return 0;
    if (!lb || !len || !out_lb || !out_len || (x->bidirectional && (!lb_rev || !out_lb_rev))) return fail(FMGPU_ERR_INVALID, "cursor arrays are null");
    hipStream_t stream = (hipStream_t)stream_;
    const uint64_t fan = symb ? 1 : (uint64_t)x->bwt.sigma;
    Staged slb, srev, slen, ssym, olb, orev, olen;
    int rc;
    if ((rc = slb.in(lb, count * 8, stream)) || (rc = srev.in(lb_rev, lb_rev ? count * 8 : 0, stream)) || (rc = slen.in(len, count * 8, stream)) ||
        (rc = ssym.in(symb, symb ? count : 0, stream)) || (rc = olb.out(out_lb, count * fan * 8, stream)) ||
        (rc = orev.out(out_lb_rev, out_lb_rev ? count * fan * 8 : 0, stream)) || (rc = olen.out(out_len, count * fan * 8, stream))) return rc;
    FM_GRID(grid, count);
    const DevString& fw = x->bwt; const DevString& rv = x->bidirectional ? x->rev : x->bwt;
    rc = dispatch_native(fw, [&](auto occ, auto ms) {
        using O = decltype(occ);
        O r = occ;
        if constexpr (std::is_same_v<O, OccA<5>> || std::is_same_v<O, OccA<0>>) r = O{rv.va};
        else if constexpr (std::is_same_v<O, OccM>) r = O{rv.vm};
        else r = O{rv.vr};
        k_cursor_extend<O, decltype(ms)::value><<<grid, dim3(256), 0, stream>>>(occ, r, x->bidirectional, direction, count, (const uint64_t*)slb.dev, (const uint64_t*)srev.dev,
                                                                               (const uint64_t*)slen.dev, (const uint8_t*)ssym.dev, fw.n, (uint64_t*)olb.dev, (uint64_t*)orev.dev, (uint64_t*)olen.dev);
        return 0;
    });
    FM_LAUNCHED("k_cursor_extend");
    if ((rc = olb.finish()) || (rc = orev.finish()) || (rc = olen.finish())) return rc;
    return 0;
}

}  // namespace api
}  // namespace FMGPU_NS

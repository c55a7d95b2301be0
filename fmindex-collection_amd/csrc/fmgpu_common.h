// fmgpu_common.h — internal to libfmgpu.so: device formats, occurrence-table accessors, host helpers.
//
// Device formats (HBM layout).  Row indices are 32-bit on the device (n < 2^32 - 64);
// the ABI speaks uint64_t like the reference's size_t.
//
//  Format A  ("LF-ready interleaved bitvectors", built from the reference's InterleavedBitvector* /
//             InterleavedBitvectorPrefix* arrays by k_convert_ib):
//     one block per 64 rows, entry c of block B at byte  B*bstride + 12*c :
//         u32 cnt   = C[c] + #{ j < 64B : s[j] == c }          (C folded in: LF needs no second table)
//         u64 bits  = bit k set  <=>  s[64B + k] == c           (row p <-> bit p&63 of block p>>6)
//     sigma <= 5: bstride = 64 (one HBM line per block, 4 spare bytes); otherwise bstride = 12*sigma.
//     LF(i, c) = cnt + popc(bits & lowmask(i & 63)),  rank(i, c) = LF(i, c) - C[c].
//     The reference stores u16 counts relative to a 65 536-row super-block plus a u64 super-block table
//     (string/InterleavedBitvector.h:13-60) and shifts rows by one bit (row p <-> bit (p+1)&63 of block
//     (p+1)>>6); both are normalised away at upload, results are identical.
//
//  Format R  (reference layout as is — InterleavedEPR*, InterleavedEPRV2*): blocks + superBlocks copied verbatim.
//
//  Format W  (wavelet; built from Wavelet::bitvector[*] at upload, or on the device by the builder): every node is an array
//     of 64-byte lines { u64 hdr0 = ones before the line (within the node); u64 hdr1 = cum[1..5], 9 bits each, cum[k] = ones
//     in bits[0..k); u64 bits[6] } (384 payload bits per line); nodes are concatenated, line offset of node k in node_base[k].
//     One node-rank = one line, fetched with two loads (header + the word that holds the position); the reference touches
//     three arrays (bitvector/Bitvector.h:147-166).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/fmgpu.h"

namespace fmgpu {

using idx_t = uint32_t;

enum Family : int { FAM_A = 0, FAM_EPR = 1, FAM_EPRV2 = 2, FAM_WAVELET = 3 };

// ------------------------------------------------------------------ device-side views (kernel arguments)
struct ViewA {            // Format A
    const uint8_t* blk;
    uint32_t bstride;
    uint32_t sigma;
    const idx_t* C;       // sigma+1 entries (device)
};

struct ViewR {            // Format R (EPR / EPRV2 reference layout)
    const uint8_t* blk;
    const uint64_t* super;
    const idx_t* C;
    uint32_t stride, bits_off, bt, sigma, bitct;
    uint32_t rows;        // rows per block (EPR: 64 / bitct, EPRV2: 64)
    uint32_t period_shift;// EPRV2: log2(rows per super-block), EPR: 0 (uses period)
    uint32_t period;      // EPR: rows per super-block (not a power of two)
    uint64_t maskEven, bitMask;
};

struct ViewW {            // Format W
    const uint64_t* lines;      // 8 u64 per line
    const uint32_t* node_base;  // line offset per node
    const idx_t* C;
    uint32_t sigma, bitct;
};

// ------------------------------------------------------------------ small device helpers
__device__ __forceinline__ uint64_t lowmask(uint32_t k) { return (1ull << k) - 1ull; }   // k in [0, 63]
__device__ __forceinline__ uint32_t popc64(uint64_t v) { return (uint32_t)__popcll(v); }

struct EntryA { uint32_t cnt; uint64_t bits; };

__device__ __forceinline__ EntryA load_entry_a(const uint8_t* blk, uint32_t bstride, idx_t i, uint32_t c) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(blk + (size_t)(i >> 6) * bstride + c * 12u);
    EntryA e;
    uint32_t a = p[0], b = p[1], d = p[2];
    e.cnt = a;
    e.bits = (uint64_t)b | ((uint64_t)d << 32);
    return e;
}

// ------------------------------------------------------------------ occurrence-table accessors
// Every Occ exposes:
//   lf2(a, b, c, &ra, &rb)      LF(a,c), LF(b,c)   (= rank + C[c]) with both ends' loads issued together
//   rank(i, c), prefix_rank(i, c), symbol(i)
//   all2(a, b, lfa[], lfb[])    LF for every symbol at both ends (extend-all)
//   sigma()
template <int SIGMA>   // SIGMA > 0: compile-time alphabet, 0: runtime
struct OccA {
    ViewA v;
    static constexpr int kMaxSigma = SIGMA > 0 ? SIGMA : 256;
    __device__ __forceinline__ uint32_t sigma() const { return SIGMA > 0 ? (uint32_t)SIGMA : v.sigma; }

    __device__ __forceinline__ void lf2(idx_t a, idx_t b, uint32_t c, idx_t& ra, idx_t& rb) const {
        EntryA ea = load_entry_a(v.blk, v.bstride, a, c);
        EntryA eb = load_entry_a(v.blk, v.bstride, b, c);
        ra = ea.cnt + popc64(ea.bits & lowmask(a & 63u));
        rb = eb.cnt + popc64(eb.bits & lowmask(b & 63u));
    }
    __device__ __forceinline__ idx_t lf(idx_t i, uint32_t c) const {
        EntryA e = load_entry_a(v.blk, v.bstride, i, c);
        return e.cnt + popc64(e.bits & lowmask(i & 63u));
    }
    __device__ __forceinline__ idx_t rank(idx_t i, uint32_t c) const { return lf(i, c) - v.C[c]; }
    __device__ __forceinline__ idx_t prefix_rank(idx_t i, uint32_t c) const {
        idx_t r = 0;
        for (uint32_t d = 0; d < c; ++d) r += rank(i, d);
        return r;
    }
    __device__ __forceinline__ uint32_t symbol(idx_t i) const {
        const uint32_t s = sigma();
        uint32_t bit = i & 63u;
        for (uint32_t c = 0; c + 1 < s; ++c) {
            EntryA e = load_entry_a(v.blk, v.bstride, i, c);
            if ((e.bits >> bit) & 1ull) return c;
        }
        return s - 1;
    }
    // symbol and LF of that symbol from one pass over the block (locate)
    __device__ __forceinline__ idx_t lf_symbol(idx_t i, uint32_t& symb) const {
        const uint32_t s = sigma();
        uint32_t bit = i & 63u;
        uint32_t c = 0;
        EntryA e = load_entry_a(v.blk, v.bstride, i, 0);
        while (c + 1 < s && !((e.bits >> bit) & 1ull)) { ++c; e = load_entry_a(v.blk, v.bstride, i, c); }
        symb = c;
        return e.cnt + popc64(e.bits & lowmask(bit));
    }
    // all2 in two halves for SIGMA <= 5 (one 64-byte block per end): the loads alone, so that a caller can issue them next to other lanes'
    // loads of a divergent wave before anything is consumed, and the arithmetic
    __device__ __forceinline__ void load2(idx_t a, idx_t b, uint32_t* da, uint32_t* db) const {
        const uint4* pa = reinterpret_cast<const uint4*>(v.blk + (size_t)(a >> 6) * 64u);
        const uint4* pb = reinterpret_cast<const uint4*>(v.blk + (size_t)(b >> 6) * 64u);
#pragma unroll
        for (int k = 0; k < 4; ++k) { uint4 t = pa[k]; da[4 * k] = t.x; da[4 * k + 1] = t.y; da[4 * k + 2] = t.z; da[4 * k + 3] = t.w; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { uint4 t = pb[k]; db[4 * k] = t.x; db[4 * k + 1] = t.y; db[4 * k + 2] = t.z; db[4 * k + 3] = t.w; }
    }
    __device__ __forceinline__ void all2_of(const uint32_t* da, const uint32_t* db, idx_t a, idx_t b, idx_t* lfa, idx_t* lfb) const {
        const uint64_t ma = lowmask(a & 63u), mb = lowmask(b & 63u);
#pragma unroll
        for (uint32_t c = 0; c < (uint32_t)(SIGMA > 0 ? SIGMA : 1); ++c) {
            lfa[c] = da[3 * c] + popc64(((uint64_t)da[3 * c + 1] | ((uint64_t)da[3 * c + 2] << 32)) & ma);
            lfb[c] = db[3 * c] + popc64(((uint64_t)db[3 * c + 1] | ((uint64_t)db[3 * c + 2] << 32)) & mb);
        }
    }
    template <int MS>
    __device__ __forceinline__ void all2(idx_t a, idx_t b, idx_t* lfa, idx_t* lfb) const {
        const uint32_t s = sigma();
        if (SIGMA > 0 && SIGMA <= 5) {
            // 64-byte block = one line: fetch it whole (4 x dwordx4); both ends usually share the block once the
            // interval is short, then the second fetch is skipped
            // both ends' loads are issued back to back (one round trip); a wave holds 64 out-of-phase lanes, so "rare"
            // second fetches would otherwise be paid by the whole wave in nearly every iteration
            uint32_t da[16], db[16];
            const uint4* pa = reinterpret_cast<const uint4*>(v.blk + (size_t)(a >> 6) * 64u);
            const uint4* pb = reinterpret_cast<const uint4*>(v.blk + (size_t)(b >> 6) * 64u);
#pragma unroll
            for (int k = 0; k < 4; ++k) { uint4 t = pa[k]; da[4 * k] = t.x; da[4 * k + 1] = t.y; da[4 * k + 2] = t.z; da[4 * k + 3] = t.w; }
#pragma unroll
            for (int k = 0; k < 4; ++k) { uint4 t = pb[k]; db[4 * k] = t.x; db[4 * k + 1] = t.y; db[4 * k + 2] = t.z; db[4 * k + 3] = t.w; }
            const uint64_t ma = lowmask(a & 63u), mb = lowmask(b & 63u);
#pragma unroll
            for (uint32_t c = 0; c < (uint32_t)SIGMA; ++c) {
                lfa[c] = da[3 * c] + popc64(((uint64_t)da[3 * c + 1] | ((uint64_t)da[3 * c + 2] << 32)) & ma);
                lfb[c] = db[3 * c] + popc64(((uint64_t)db[3 * c + 1] | ((uint64_t)db[3 * c + 2] << 32)) & mb);
            }
            return;
        }
        if (MS <= 32) {
#pragma unroll
            for (uint32_t c = 0; c < (uint32_t)MS; ++c) {
                if (c < s) {
                    EntryA ea = load_entry_a(v.blk, v.bstride, a, c);
                    EntryA eb = load_entry_a(v.blk, v.bstride, b, c);
                    lfa[c] = ea.cnt + popc64(ea.bits & lowmask(a & 63u));
                    lfb[c] = eb.cnt + popc64(eb.bits & lowmask(b & 63u));
                }
            }
        } else {
            for (uint32_t c = 0; c < s; ++c) {
                EntryA ea = load_entry_a(v.blk, v.bstride, a, c);
                EntryA eb = load_entry_a(v.blk, v.bstride, b, c);
                lfa[c] = ea.cnt + popc64(ea.bits & lowmask(a & 63u));
                lfb[c] = eb.cnt + popc64(eb.bits & lowmask(b & 63u));
            }
        }
    }
};

// Format R: reference layout read in place.  EPR: string/InterleavedEPR.h:63-103, :154-178; EPRV2: string/InterleavedEPRV2.h:28-105, :191-213
template <bool V2>
struct OccR {
    ViewR v;
    static constexpr int kMaxSigma = 256;
    __device__ __forceinline__ uint32_t sigma() const { return v.sigma; }

    __device__ __forceinline__ uint32_t cnt(size_t b, uint32_t c) const {
        const uint8_t* p = v.blk + b * v.stride + c * v.bt;
        if (v.bt == 2) return *reinterpret_cast<const uint16_t*>(p);
        if (v.bt == 1) return *p;
        return *reinterpret_cast<const uint32_t*>(p);
    }
    __device__ __forceinline__ uint64_t word(size_t b, uint32_t k) const {
        return *reinterpret_cast<const uint64_t*>(v.blk + b * v.stride + v.bits_off + 8u * k);
    }
    __device__ __forceinline__ uint64_t have(size_t b, uint32_t symb) const {   // EPRV2 symbol-match mask
        uint64_t r = ~0ull;
        for (uint32_t i = 0; i < v.bitct; ++i) {
            uint64_t inv = (~symb >> i) & 1u;
            r &= word(b, i) ^ (0ull - inv);
        }
        return r;
    }
    // EPR: one bit per slot (at bit slot*bitct) set where slot value <= symb
    __device__ __forceinline__ uint64_t le_mask(uint64_t in, uint32_t symb) const {
        uint64_t rb = 0;     // rb[symb], InterleavedEPR.h:49-61
        uint64_t mk = (uint64_t)symb | (1ull << v.bitct);
        for (uint32_t i = 0; i < 64u / v.bitct; i += 2) rb = (rb << (2 * v.bitct)) | mk;
        uint64_t te = ((rb - (in & v.maskEven)) & v.bitMask) >> v.bitct;
        uint64_t to = (rb - ((in >> v.bitct) & v.maskEven)) & v.bitMask;
        return te | to;
    }
    __device__ __forceinline__ void split(idx_t i, size_t& b, size_t& sb, uint32_t& bit) const {
        if (V2) { b = i >> 6; sb = v.period_shift >= 32 ? 0 : (i >> v.period_shift); bit = i & 63u; }
        else    { b = i / v.rows; sb = i / v.period; bit = i % v.rows; }
    }
    __device__ __forceinline__ idx_t rank(idx_t i, uint32_t c) const {
        size_t b, sb; uint32_t bit; split(i, b, sb, bit);
        uint64_t sup = v.super[sb * v.sigma + c];
        if (V2) {
            uint64_t m = have(b, c);
            uint32_t in = bit == 0 ? 0u : popc64(m << (64u - bit));
            return (idx_t)(sup + cnt(b, c) + in);
        } else {
            uint64_t in = word(b, 0);
            uint64_t lim = (1ull << (bit * v.bitct)) - 1ull;
            uint32_t hi = popc64(le_mask(in, c) & lim);
            uint32_t lo = c == 0 ? 0u : popc64(le_mask(in, c - 1) & lim);
            return (idx_t)(sup + cnt(b, c) + hi - lo);
        }
    }
    __device__ __forceinline__ idx_t lf(idx_t i, uint32_t c) const { return rank(i, c) + v.C[c]; }
    __device__ __forceinline__ void lf2(idx_t a, idx_t b, uint32_t c, idx_t& ra, idx_t& rb) const {
        ra = lf(a, c); rb = lf(b, c);
    }
    __device__ __forceinline__ idx_t prefix_rank(idx_t i, uint32_t c) const {
        idx_t r = 0;
        for (uint32_t d = 0; d < c; ++d) r += rank(i, d);
        return r;
    }
    __device__ __forceinline__ uint32_t symbol(idx_t i) const {
        size_t b, sb; uint32_t bit; split(i, b, sb, bit);
        if (V2) {
            uint32_t s = 0;
            for (uint32_t k = v.bitct; k > 0; --k) s = (s << 1) | (uint32_t)((word(b, k - 1) >> bit) & 1ull);
            return s;
        }
        return (uint32_t)((word(b, 0) >> (bit * v.bitct)) & ((1ull << v.bitct) - 1ull));
    }
    __device__ __forceinline__ idx_t lf_symbol(idx_t i, uint32_t& symb) const { symb = symbol(i); return lf(i, symb); }
    template <int MS>
    __device__ __forceinline__ void all2(idx_t a, idx_t b, idx_t* lfa, idx_t* lfb) const {
        for (uint32_t c = 0; c < v.sigma && c < (uint32_t)MS; ++c) { lfa[c] = lf(a, c); lfb[c] = lf(b, c); }
    }
};

// Format W: string/Wavelet.h:77-141 over one-line node ranks
struct OccW {
    ViewW v;
    static constexpr int kMaxSigma = 256;
    __device__ __forceinline__ uint32_t sigma() const { return v.sigma; }

    __device__ __forceinline__ idx_t node_rank(uint32_t id, idx_t i, uint32_t* bit_out) const {
        const uint32_t line = i / 384u, r = i - line * 384u;
        const uint32_t k = r >> 6, part = r & 63u;
        const uint64_t* L = v.lines + ((size_t)v.node_base[id] + line) * 8u;
        const uint64_t h0 = L[0], h1 = L[1];
        const uint64_t w = L[2 + k];
        const uint32_t cum = k ? (uint32_t)(h1 >> (9u * (k - 1u))) & 0x1ffu : 0u;
        if (bit_out) *bit_out = (uint32_t)((w >> part) & 1ull);
        return (idx_t)h0 + cum + popc64(w & lowmask(part));
    }
    __device__ __forceinline__ idx_t rank(idx_t i, uint32_t c) const {
        for (uint32_t b = 0; b < v.bitct; ++b) {
            uint32_t bitId = v.bitct - b - 1;
            uint32_t bit = (c >> bitId) & 1u;
            uint32_t id = ((1u << b) - 1u) + (c >> (bitId + 1));
            idx_t r = node_rank(id, i, nullptr);
            i = bit ? r : i - r;
        }
        return i;
    }
    __device__ __forceinline__ idx_t lf(idx_t i, uint32_t c) const { return rank(i, c) + v.C[c]; }
    __device__ __forceinline__ void lf2(idx_t a, idx_t b, uint32_t c, idx_t& ra, idx_t& rb) const {
        // both ends descend the same node path: interleave the two dependent chains
        for (uint32_t lv = 0; lv < v.bitct; ++lv) {
            uint32_t bitId = v.bitct - lv - 1;
            uint32_t bit = (c >> bitId) & 1u;
            uint32_t id = ((1u << lv) - 1u) + (c >> (bitId + 1));
            idx_t x = node_rank(id, a, nullptr), y = node_rank(id, b, nullptr);
            a = bit ? x : a - x;
            b = bit ? y : b - y;
        }
        ra = a + v.C[c]; rb = b + v.C[c];
    }
    __device__ __forceinline__ idx_t prefix_rank(idx_t i, uint32_t c) const {
        if (c == 0) return 0;
        c -= 1;
        idx_t a = 0;
        for (uint32_t b = 0; b < v.bitct; ++b) {
            uint32_t bitId = v.bitct - b - 1;
            uint32_t bit = (c >> bitId) & 1u;
            uint32_t id = ((1u << b) - 1u) + (c >> (bitId + 1));
            idx_t r = node_rank(id, i, nullptr);
            if (!bit) i = i - r; else { a += i - r; i = r; }
        }
        return a + i;
    }
    __device__ __forceinline__ idx_t lf_symbol(idx_t i, uint32_t& symb) const {
        uint32_t s = 0;
        for (uint32_t b = 0; b < v.bitct; ++b) {
            uint32_t id = ((1u << b) - 1u) + s;
            uint32_t bit;
            idx_t r = node_rank(id, i, &bit);
            s = (s << 1) | bit;
            i = bit ? r : i - r;
        }
        symb = s;
        return i + v.C[s];
    }
    __device__ __forceinline__ uint32_t symbol(idx_t i) const { uint32_t s; (void)lf_symbol(i, s); return s; }
    template <int MS>
    __device__ __forceinline__ void all2(idx_t a, idx_t b, idx_t* lfa, idx_t* lfb) const {
        for (uint32_t c = 0; c < v.sigma && c < (uint32_t)MS; ++c) lf2(a, b, c, lfa[c], lfb[c]);
    }
};

// sampled suffix array on the device: the reference's arrays verbatim (suffixarray/SparseArray.h:31-76)
struct ViewSA {
    const uint64_t* l0; const uint16_t* l1; const uint64_t* bits;
    const uint64_t* f0; const uint64_t* f1;
    uint32_t bits0, bits1; uint64_t div0, div1;
};

__device__ __forceinline__ bool sa_present(const ViewSA& s, idx_t i) { return (s.bits[i >> 6] >> (i & 63u)) & 1ull; }
__device__ __forceinline__ uint64_t sa_rank(const ViewSA& s, idx_t i) {   // bitvector/Bitvector2L.h:123-142
    uint32_t bitId = i & 511u;
    const uint64_t* w = s.bits + (size_t)(i >> 9) * 8u;
    uint32_t cnt = 0;
    for (uint32_t k = 0; k < (bitId >> 6); ++k) cnt += popc64(w[k]);
    if (bitId & 63u) cnt += popc64(w[bitId >> 6] & lowmask(bitId & 63u));
    return s.l0[i >> 16] + s.l1[i >> 9] + cnt;
}
__device__ __forceinline__ uint64_t dense_access(const uint64_t* data, uint32_t bits, uint64_t div, uint64_t i) {   // DenseVector.h:154-182
    uint64_t begin = i * bits, end = begin + bits - 1;
    uint64_t s = begin >> 6, e = end >> 6;
    uint32_t off = (uint32_t)(begin & 63u);
    uint64_t mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
    uint64_t val = data[s] >> off;
    if (s != e) val |= data[e] << (64u - off);
    return (val & mask) * div;
}

// ------------------------------------------------------------------ host side
struct DevString {
    int layout = 0, family = 0, sigma = 0, bitct = 0;
    uint64_t n = 0;
    void* blk = nullptr;       // Format A blocks / Format R blocks / Format W lines
    void* aux = nullptr;       // Format R superBlocks / Format W node_base
    size_t blk_bytes = 0, aux_bytes = 0;
    ViewA va{}; ViewR vr{}; ViewW vw{};
    // explicit LF mapping: lf_table[i] = C[s[i]] + rank(i, s[i])  (n entries; the symbol is recovered from C).
    // 4 bytes per row buy one-load single-row DFS nodes and one-load locate steps; skipped with FMGPU_LF_TABLE=0.
    idx_t* lf_table = nullptr;
    // multi-symbol-step table (fmgpu_index_accelerate): block B, context w at  kblk + (B * kcodes + w) * 16 :
    //   { u32 cnt = LF_k(64B, w); u32 bits_lo; u32 bits_hi; u32 0 }   (bit r: the kstep symbols preceding suffix 64B+r spell w;
    //   the first 12 bytes are fetched with one dwordx3 load, like a Format A entry)
    uint8_t* kblk = nullptr; uint32_t kstep = 0, kcodes = 0; size_t kblk_bytes = 0;
    // walk table (fmgpu_index_accelerate_search): walk3[3*i .. 3*i+2] = LF(i), LF^2(i), LF^3(i)
    idx_t* walk3 = nullptr;
    // exact-search accelerators (fmgpu_index_accelerate_exact):
    //   slut[code] = {lb, len} of the backward search of the slut_len symbols with code = sum (c_t - 1) * (sigma-1)^t, c_0 = the LAST symbol;
    //   walkj[row] = {LF^J(row), sum (s_t - 1) << (walk_bits * t)} with s_1.. the symbols met (s_0 = BWT symbol of row), or {~0, 0} if a delimiter is met
    uint2* slut = nullptr; uint32_t slut_len = 0; uint64_t slut_entries = 0;
    uint2* walkj = nullptr; uint32_t walk_J = 0, walk_bits = 0;
    //   walk2j[3*row ..] = {LF^(2J)(row), code of symbols 0 .. J-1, code of symbols J .. 2J-1} (walk = 2 in fmgpu_index_accelerate_exact), or {~0, 0, 0}
    uint32_t* walk2j = nullptr;
    // Format A shadow of a Format R / W string (fmgpu_index_accelerate, kstep >= 1): the searches then read `va` (one line per
    // LF step instead of bitct lines); fmgpu_string_query keeps answering from the native format.
    void* shadow = nullptr; size_t shadow_bytes = 0;
    int search_family() const { return shadow ? (int)FAM_A : family; }
};

// builds s.shadow / s.va from the string's own symbols; defined in fmgpu_build.hip
int build_format_a_shadow(DevString& s, const idx_t* dC, hipStream_t stream);

// 0 if the calling thread's current device is the one the handle lives on; defined in fmgpu_index.hip
int on_handle_device(const struct Index* x);

// fills s.lf_table from the device string (all layouts); defined in fmgpu_index.hip
int build_lf_table(DevString& s, hipStream_t stream);
void free_string(DevString& s);

struct Index {
    DevString bwt, rev;
    bool bidirectional = false, has_sa = false;
    int device = 0;
    idx_t* dC = nullptr;
    uint64_t hC[258] = {0};
    // sampled SA
    void *sa_l0 = nullptr, *sa_l1 = nullptr, *sa_bits = nullptr, *sa_f0 = nullptr, *sa_f1 = nullptr;
    ViewSA vsa{};
    // fmgpu_index_accelerate_locate: the (seqId, pos, steps) answer of every row, 3 x u32 per row (or null)
    uint32_t* loc_tab = nullptr;
    size_t device_bytes = 0;
    // prefix table (fmgpu_index_accelerate_search): lut[code(w)] = { lb, lbRev, len, symbols consumed before the interval emptied (or L) }
    uint4* lut = nullptr; uint32_t lut_len = 0; uint64_t lut_entries = 0;
};

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
int hip_fail(hipError_t e, const char* what);

#define FM_HIP(call)                                                   \
    do {                                                               \
        hipError_t e_ = (call);                                        \
        if (e_ != hipSuccess) return ::fmgpu::hip_fail(e_, #call);     \
    } while (0)

bool is_device_pointer(const void* p);

// A temporary device mirror of a caller buffer: device pointers are used in place, host pointers are staged.
struct Staged {
    void* dev = nullptr;
    void* host = nullptr;
    size_t bytes = 0;
    bool owned = false, writeback = false;
    hipStream_t stream = nullptr;
    int in(const void* p, size_t nbytes, hipStream_t s);      // read-only input
    int out(void* p, size_t nbytes, hipStream_t s);           // output (copied back by finish())
    int finish();                                             // D2H of outputs (synchronises when anything was staged)
    ~Staged();
};

}  // namespace fmgpu

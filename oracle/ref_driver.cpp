// oracle/ref_driver.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A thin extern "C" driver around the REAL reference headers, compiled in place from
// /root/reference/src (nothing is copied into this repository).  Only the parts of the
// reference that build with this image's toolchain and *no stand-in headers* are used:
//   * the rank-support strings  (src/fmindex-collection/string/*.h, bitvector/Bitvector.h)
//   * the search-scheme tables  (src/fmindex-collection/search_scheme/*.h)
// FMIndex/BiFMIndex/cursors/search_* include utils.h, which needs <libsais.h> and
// <mmser/mmser.h> (fetched from the network by the reference's CMake) — unbuildable here,
// see DESIGN.md "Oracle".
//
// Output: oracle/_ref/libfmref.so (git-ignored).  Used by tests/ and the golden generator
// to pin oracle/fmoracle.c (layouts byte for byte, rank/prefix_rank/symbol, scheme tables).

#include <string/InterleavedBitvector.h>
#include <string/InterleavedBitvectorPrefix.h>
#include <string/InterleavedEPR.h>
#include <string/InterleavedEPRV2.h>
#include <string/Wavelet.h>
#include <string/EPRV3.h>
#include <string/EPRV4.h>
#include <string/EPRV5.h>
#include <string/InterleavedEPRV7.h>
#include <bitvector/Bitvector.h>
#include <search_scheme/generator/h2.h>
#include <search_scheme/generator/pigeon.h>
#include <search_scheme/generator/backtracking.h>
#include <search_scheme/expand.h>
#include <search_scheme/isValid.h>
#include <search_scheme/isComplete.h>
#include <search_scheme/nodeCount.h>
#include <search_scheme/weightedNodeCount.h>

#include <cstdint>
#include <cstring>
#include <memory>
#include <span>
#include <vector>

namespace {

// layout ids shared with oracle/fmoracle.h and include/fmgpu.h
enum Layout : int {
    L_IB8 = 0, L_IB16 = 1, L_IB32 = 2, L_IB16A = 3,
    L_IBP16 = 4,
    L_EPR8 = 5, L_EPR16 = 6, L_EPR32 = 7,
    L_EPRV2_8 = 8, L_EPRV2_16 = 9, L_EPRV2_32 = 10,
    L_WAVELET = 11,
    L_EPRV3_8 = 12, L_EPRV3_16 = 13, L_EPRV3_32 = 14, L_EPRV4 = 15, L_EPRV5 = 16, L_IEPRV7 = 17,
};

struct AnyString {
    virtual ~AnyString() = default;
    virtual uint64_t size() const = 0;
    virtual uint64_t rank(uint64_t i, uint64_t c) const = 0;
    virtual uint64_t prefix_rank(uint64_t i, uint64_t c) const = 0;
    virtual uint64_t symbol(uint64_t i) const = 0;
    virtual void all_ranks_and_prefix_ranks(uint64_t i, uint64_t* rs, uint64_t* prs) const = 0;
    // raw arrays (part 0 = blocks, part 1 = superBlocks; wavelet: node*3 + {0,1,2})
    virtual int raw(int part, void const** ptr, uint64_t* bytes) const = 0;
    virtual uint64_t block_stride() const = 0;
};

template <typename S>
struct Blocked : AnyString {
    S s;
    explicit Blocked(std::span<uint8_t const> t) : s{t} {}
    uint64_t size() const override { return s.size(); }
    uint64_t rank(uint64_t i, uint64_t c) const override { return s.rank(i, c); }
    uint64_t prefix_rank(uint64_t i, uint64_t c) const override { return s.prefix_rank(i, c); }
    uint64_t symbol(uint64_t i) const override { return s.symbol(i); }
    void all_ranks_and_prefix_ranks(uint64_t i, uint64_t* rs, uint64_t* prs) const override {
        auto [a, b] = s.all_ranks_and_prefix_ranks(i);
        for (size_t k = 0; k < S::Sigma; ++k) { rs[k] = a[k]; prs[k] = b[k]; }
    }
    int raw(int part, void const** ptr, uint64_t* bytes) const override {
        if (part == 0) { *ptr = s.blocks.data(); *bytes = s.blocks.size() * sizeof(s.blocks[0]); return 0; }
        if (part == 1) { *ptr = s.superBlocks.data(); *bytes = s.superBlocks.size() * sizeof(s.superBlocks[0]); return 0; }
        return -1;
    }
    uint64_t block_stride() const override { return sizeof(s.blocks[0]); }
};

template <typename S>
struct Wave : AnyString {
    S s;
    explicit Wave(std::span<uint8_t const> t) : s{t} {}
    uint64_t size() const override { return s.size(); }
    uint64_t rank(uint64_t i, uint64_t c) const override { return s.rank(i, c); }
    uint64_t prefix_rank(uint64_t i, uint64_t c) const override { return s.prefix_rank(i, c); }
    uint64_t symbol(uint64_t i) const override { return s.symbol(i); }
    void all_ranks_and_prefix_ranks(uint64_t i, uint64_t* rs, uint64_t* prs) const override {
        auto [a, b] = s.all_ranks_and_prefix_ranks(i);
        for (size_t k = 0; k < S::Sigma; ++k) { rs[k] = a[k]; prs[k] = b[k]; }
    }
    int raw(int part, void const** ptr, uint64_t* bytes) const override {
        size_t node = part / 4, what = part % 4;
        if (node >= s.bitvector.size()) return -1;
        auto const& bv = s.bitvector[node];
        switch (what) {
        case 0: *ptr = bv.superblocks.data(); *bytes = bv.superblocks.size() * 8; return 0;
        case 1: *ptr = bv.blocks.data();      *bytes = bv.blocks.size(); return 0;
        case 2: *ptr = bv.bits.data();        *bytes = bv.bits.size() * 8; return 0;
        case 3: *ptr = &bv.totalLength;       *bytes = 8; return 0;
        }
        return -1;
    }
    uint64_t block_stride() const override { return 0; }
};

// EPRV3 / EPRV4 / EPRV5 / InterleavedEPRV7: part 0 = bits, 1 = superBlocks, 2.. = counter levels bottom-up
template <typename S>
struct Hier : AnyString {
    S s;
    explicit Hier(std::span<uint8_t const> t) : s{t} {}
    uint64_t size() const override { return s.size(); }
    uint64_t rank(uint64_t i, uint64_t c) const override { return s.rank(i, c); }
    uint64_t prefix_rank(uint64_t i, uint64_t c) const override { return s.prefix_rank(i, c); }
    uint64_t symbol(uint64_t i) const override { return s.symbol(i); }
    void all_ranks_and_prefix_ranks(uint64_t i, uint64_t* rs, uint64_t* prs) const override {
        auto [a, b] = s.all_ranks_and_prefix_ranks(i);
        for (size_t k = 0; k < S::Sigma; ++k) { rs[k] = a[k]; prs[k] = b[k]; }
    }
    template <typename V>
    static int give(V const& v, void const** ptr, uint64_t* bytes) { *ptr = v.data(); *bytes = v.size() * sizeof(v[0]); return 0; }
    int raw(int part, void const** ptr, uint64_t* bytes) const override {
        if (part == 0) return give(s.bits, ptr, bytes);
        if (part == 1) return give(s.superBlocks, ptr, bytes);
        if constexpr (requires { s.blocks_; }) { if (part == 2) return give(s.blocks_, ptr, bytes); }
        if constexpr (requires { s.level0; }) { if (part == 2) return give(s.level0, ptr, bytes); }
        else if (part == 2) { *ptr = nullptr; *bytes = 0; return 0; }
        if constexpr (requires { s.level1; }) { if (part == 3) return give(s.level1, ptr, bytes); }
        if constexpr (requires { s.level2; }) { if (part == 4) return give(s.level2, ptr, bytes); }
        return -1;
    }
    uint64_t block_stride() const override { return sizeof(s.bits[0]); }
};

template <size_t Sigma>
AnyString* make(int layout, std::span<uint8_t const> t) {
    using namespace fmc::string;
    switch (layout) {
    case L_IB8:      return new Blocked<InterleavedBitvector8<Sigma>>{t};
    case L_IB16:     return new Blocked<InterleavedBitvector16<Sigma>>{t};
    case L_IB32:     return new Blocked<InterleavedBitvector32<Sigma>>{t};
    case L_IB16A:    return new Blocked<InterleavedBitvector16Aligned<Sigma>>{t};
    case L_IBP16:    return new Blocked<InterleavedBitvectorPrefix16<Sigma>>{t};
    case L_EPR8:     return new Blocked<InterleavedEPR8<Sigma>>{t};
    case L_EPR16:    return new Blocked<InterleavedEPR16<Sigma>>{t};
    case L_EPR32:    return new Blocked<InterleavedEPR32<Sigma>>{t};
    case L_EPRV2_8:  return new Blocked<InterleavedEPRV2_8<Sigma>>{t};
    case L_EPRV2_16: return new Blocked<InterleavedEPRV2_16<Sigma>>{t};
    case L_EPRV2_32: return new Blocked<InterleavedEPRV2_32<Sigma>>{t};
    case L_WAVELET:  return new Wave<Wavelet<Sigma>>{t};
    case L_EPRV3_8:  return new Hier<EPRV3_8<Sigma>>{t};
    case L_EPRV3_16: return new Hier<EPRV3_16<Sigma>>{t};
    case L_EPRV3_32: return new Hier<EPRV3_32<Sigma>>{t};
    case L_EPRV4:    return new Hier<EPRV4<Sigma>>{t};
    case L_EPRV5:    return new Hier<EPRV5<Sigma>>{t};
    case L_IEPRV7:   return new Hier<InterleavedEPRV7<Sigma>>{t};
    }
    return nullptr;
}

int flatten(fmc::search_scheme::Scheme const& ss, uint64_t* pi, uint64_t* l, uint64_t* u, uint64_t cap, uint64_t* parts) {
    uint64_t k = 0;
    *parts = ss.empty() ? 0 : ss[0].pi.size();
    for (auto const& s : ss) {
        for (size_t i = 0; i < s.pi.size(); ++i, ++k) {
            if (k >= cap) return -1;
            pi[k] = s.pi[i]; l[k] = s.l[i]; u[k] = s.u[i];
        }
    }
    return static_cast<int>(ss.size());
}

fmc::search_scheme::Scheme unflatten(int nsearch, uint64_t parts, uint64_t const* pi, uint64_t const* l, uint64_t const* u) {
    auto ss = fmc::search_scheme::Scheme{};
    for (int s = 0; s < nsearch; ++s) {
        auto x = fmc::search_scheme::Search{};
        for (uint64_t i = 0; i < parts; ++i) {
            x.pi.push_back(pi[s * parts + i]);
            x.l.push_back(l[s * parts + i]);
            x.u.push_back(u[s * parts + i]);
        }
        ss.push_back(x);
    }
    return ss;
}

} // namespace

extern "C" {

void* fmref_string_create(int layout, int sigma, uint8_t const* symbols, uint64_t n) {
    auto t = std::span<uint8_t const>{symbols, n};
    switch (sigma) {
    case 4:   return make<4>(layout, t);
    case 5:   return make<5>(layout, t);
    case 6:   return make<6>(layout, t);
    case 21:  return make<21>(layout, t);
    case 28:  return make<28>(layout, t);
    case 255: return make<255>(layout, t);
    case 256: return make<256>(layout, t);
    }
    return nullptr;
}
void fmref_string_destroy(void* h) { delete static_cast<AnyString*>(h); }
uint64_t fmref_string_size(void* h) { return static_cast<AnyString*>(h)->size(); }
uint64_t fmref_string_rank(void* h, uint64_t i, uint64_t c) { return static_cast<AnyString*>(h)->rank(i, c); }
uint64_t fmref_string_prefix_rank(void* h, uint64_t i, uint64_t c) { return static_cast<AnyString*>(h)->prefix_rank(i, c); }
uint64_t fmref_string_symbol(void* h, uint64_t i) { return static_cast<AnyString*>(h)->symbol(i); }
void fmref_string_all_ranks_and_prefix_ranks(void* h, uint64_t i, uint64_t* rs, uint64_t* prs) {
    static_cast<AnyString*>(h)->all_ranks_and_prefix_ranks(i, rs, prs);
}
// bulk tables: out[(i * sigma) + c] for i in [0, n], c in [0, sigma)
void fmref_string_rank_table(void* h, int sigma, uint64_t* out_rank, uint64_t* out_prefix) {
    auto* s = static_cast<AnyString*>(h);
    for (uint64_t i = 0; i <= s->size(); ++i)
        for (int c = 0; c < sigma; ++c) {
            if (out_rank)   out_rank[i * sigma + c]   = s->rank(i, c);
            if (out_prefix) out_prefix[i * sigma + c] = s->prefix_rank(i, c);
        }
}
int fmref_string_raw(void* h, int part, void const** ptr, uint64_t* bytes) { return static_cast<AnyString*>(h)->raw(part, ptr, bytes); }
uint64_t fmref_string_block_stride(void* h) { return static_cast<AnyString*>(h)->block_stride(); }

// ---- search schemes: flattened [search][part] tables; return number of searches, parts via *parts
int fmref_scheme_h2(uint64_t N, uint64_t minK, uint64_t K, uint64_t* pi, uint64_t* l, uint64_t* u, uint64_t cap, uint64_t* parts) {
    return flatten(fmc::search_scheme::generator::h2(N, minK, K), pi, l, u, cap, parts);
}
int fmref_scheme_pigeon_opt(uint64_t minK, uint64_t K, uint64_t* pi, uint64_t* l, uint64_t* u, uint64_t cap, uint64_t* parts) {
    return flatten(fmc::search_scheme::generator::pigeon_opt(minK, K), pi, l, u, cap, parts);
}
int fmref_scheme_pigeon_trivial(uint64_t minK, uint64_t K, uint64_t* pi, uint64_t* l, uint64_t* u, uint64_t cap, uint64_t* parts) {
    return flatten(fmc::search_scheme::generator::pigeon_trivial(minK, K), pi, l, u, cap, parts);
}
int fmref_scheme_backtracking(uint64_t N, uint64_t minK, uint64_t K, uint64_t* pi, uint64_t* l, uint64_t* u, uint64_t cap, uint64_t* parts) {
    return flatten(fmc::search_scheme::generator::backtracking(N, minK, K), pi, l, u, cap, parts);
}
int fmref_scheme_expand(int nsearch, uint64_t parts_in, uint64_t const* pi_in, uint64_t const* l_in, uint64_t const* u_in, uint64_t newLen,
                        uint64_t* pi, uint64_t* l, uint64_t* u, uint64_t cap, uint64_t* parts) {
    auto ss = unflatten(nsearch, parts_in, pi_in, l_in, u_in);
    return flatten(fmc::search_scheme::expand(ss, newLen), pi, l, u, cap, parts);
}
int fmref_scheme_limit_to_hamming(int nsearch, uint64_t parts_in, uint64_t const* pi_in, uint64_t const* l_in, uint64_t const* u_in,
                                  uint64_t* pi, uint64_t* l, uint64_t* u, uint64_t cap, uint64_t* parts) {
    auto ss = unflatten(nsearch, parts_in, pi_in, l_in, u_in);
    return flatten(fmc::search_scheme::limitToHamming(ss), pi, l, u, cap, parts);
}
int fmref_scheme_is_valid(int nsearch, uint64_t parts, uint64_t const* pi, uint64_t const* l, uint64_t const* u) {
    return fmc::search_scheme::isValid(unflatten(nsearch, parts, pi, l, u)) ? 1 : 0;
}
int fmref_scheme_is_complete(int nsearch, uint64_t parts, uint64_t const* pi, uint64_t const* l, uint64_t const* u, uint64_t minK, uint64_t maxK) {
    return fmc::search_scheme::isComplete(unflatten(nsearch, parts, pi, l, u), minK, maxK) ? 1 : 0;
}
double fmref_scheme_node_count_hamming(int nsearch, uint64_t parts, uint64_t const* pi, uint64_t const* l, uint64_t const* u, uint64_t sigma) {
    return static_cast<double>(fmc::search_scheme::nodeCount<false>(unflatten(nsearch, parts, pi, l, u), sigma));
}
// expandByWNC (expand.h:218-247) as the example calls it (src/example/main.cpp:116, :135): Edit = true for the optimisation
int fmref_scheme_expand_by_wnc(int nsearch, uint64_t parts_in, uint64_t const* pi_in, uint64_t const* l_in, uint64_t const* u_in, uint64_t newLen, uint64_t sigma, uint64_t N,
                               int edit, uint64_t* pi, uint64_t* l, uint64_t* u, uint64_t cap, uint64_t* parts) {
    auto ss = unflatten(nsearch, parts_in, pi_in, l_in, u_in);
    return flatten(edit ? fmc::search_scheme::expandByWNC<true>(ss, newLen, sigma, N) : fmc::search_scheme::expandByWNC<false>(ss, newLen, sigma, N), pi, l, u, cap, parts);
}
double fmref_scheme_weighted_node_count(int nsearch, uint64_t parts, uint64_t const* pi, uint64_t const* l, uint64_t const* u, uint64_t sigma, uint64_t N, int edit) {
    auto ss = unflatten(nsearch, parts, pi, l, u);
    return static_cast<double>(edit ? fmc::search_scheme::weightedNodeCount<true>(ss, sigma, N) : fmc::search_scheme::weightedNodeCount<false>(ss, sigma, N));
}
void fmref_uniform_partition(uint64_t parts, uint64_t total, uint64_t* out) {
    auto p = fmc::search_scheme::createUniformPartition(parts, total);
    for (size_t i = 0; i < p.size(); ++i) out[i] = p[i];
}

} // extern "C"

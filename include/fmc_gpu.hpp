// fmc_gpu.hpp — C++ host mirror of the reference's template API for the backward-search path, on top of the C-ABI
// (include/fmgpu.h, libfmgpu.so).  Header-only, C++17.  Same names, argument meaning and error behaviour as
// SGSSGene/fmindex-collection for: FMIndex / BiFMIndex (fmindex/FMIndex.h:14-134, fmindex/BiFMIndex.h:17-216), their cursors
// (fmindex/FMIndexCursor.h, fmindex/BiFMIndexCursor.h: lb, lbRev, len, steps, count(), empty(), begin/end, extendLeft / extendRight with and
// without a symbol, symbolLeft / symbolRight), single_locate_step, fmc::Search{...}(),
// search_no_errors::search (search/SearchNoErrors.h), search_backtracking::search (search/Backtracking.h),
// search_ng26::search (search/SearchNg26.h:426-444), fmc::search<Edit> (search/search.h:26-35), LocateLinear (locate.h:14-57),
// search_scheme::{Search, Scheme, generator::{h2, pigeon_opt, pigeon_trivial, backtracking}, createUniformPartition, expand,
// limitToHamming, isValid, isComplete} (search_scheme/).
//
// Differences that follow from batching on a GPU: searches run over the whole `queries` range in one call and the delegates
// are invoked on the host afterwards — in ascending qidx and, inside a query, in the reference's callback order.  Both Hamming
// (Edit = false) and edit distance (Edit = true, the reference's default) run on the GPU.
#pragma once

#include "fmgpu.h"

#include <algorithm>
#include <cmath>
#include <array>
#include <cstddef>
#include <cstdint>
#include <limits>
#include <numeric>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

namespace fmc {

namespace detail {
inline void check(int rc) {
    if (rc != 0) throw std::runtime_error(std::string("fmindex-collection (gpu): ") + fmgpu_last_error());
}
template <typename Seqs>
inline void flatten(Seqs const& seqs, std::vector<uint8_t>& buf, std::vector<uint64_t>& off) {
    off.assign(1, 0);
    for (auto const& s : seqs) {
        for (auto c : s) buf.push_back(static_cast<uint8_t>(c));
        off.push_back(buf.size());
    }
    if (buf.empty()) buf.push_back(0);
}
}  // namespace detail

// ------------------------------------------------------------------------------------------------ occurrence-table tags
namespace string {
#define FMC_GPU_STRING_TAG(NAME, ID) \
    template <size_t TSigma> struct NAME { static constexpr size_t Sigma = TSigma; static constexpr int layout = ID; };
FMC_GPU_STRING_TAG(InterleavedBitvector8, FMGPU_IB8)
FMC_GPU_STRING_TAG(InterleavedBitvector16, FMGPU_IB16)
FMC_GPU_STRING_TAG(InterleavedBitvector32, FMGPU_IB32)
FMC_GPU_STRING_TAG(InterleavedBitvector16Aligned, FMGPU_IB16A)
FMC_GPU_STRING_TAG(InterleavedBitvectorPrefix16, FMGPU_IBP16)
FMC_GPU_STRING_TAG(InterleavedEPR8, FMGPU_EPR8)
FMC_GPU_STRING_TAG(InterleavedEPR16, FMGPU_EPR16)
FMC_GPU_STRING_TAG(InterleavedEPR32, FMGPU_EPR32)
FMC_GPU_STRING_TAG(InterleavedEPRV2_8, FMGPU_EPRV2_8)
FMC_GPU_STRING_TAG(InterleavedEPRV2_16, FMGPU_EPRV2_16)
FMC_GPU_STRING_TAG(InterleavedEPRV2_32, FMGPU_EPRV2_32)
FMC_GPU_STRING_TAG(Wavelet, FMGPU_WAVELET)
FMC_GPU_STRING_TAG(EPRV3_8, FMGPU_EPRV3_8)
FMC_GPU_STRING_TAG(EPRV3_16, FMGPU_EPRV3_16)
FMC_GPU_STRING_TAG(EPRV3_32, FMGPU_EPRV3_32)
FMC_GPU_STRING_TAG(EPRV4, FMGPU_EPRV4)
FMC_GPU_STRING_TAG(EPRV5, FMGPU_EPRV5)
FMC_GPU_STRING_TAG(InterleavedEPRV7, FMGPU_IEPRV7)
FMC_GPU_STRING_TAG(FlattenedBitvectors_64_64k, FMGPU_FBV_64_64K)
FMC_GPU_STRING_TAG(FlattenedBitvectors_512_64k, FMGPU_FBV_512_64K)
FMC_GPU_STRING_TAG(FlattenedBitvectors_2048_64k, FMGPU_FBV_2048_64K)
#undef FMC_GPU_STRING_TAG
}  // namespace string

// ------------------------------------------------------------------------------------------------ indices
template <size_t TSigma, template <size_t> class String, bool Bidirectional>
struct GpuIndexBase {
    static constexpr size_t Sigma = TSigma;
    static constexpr size_t FirstSymb = 1;
    static constexpr bool IsBidirectional = Bidirectional;
    using LEntry = std::tuple<uint32_t, uint32_t, size_t>;   // (seqId, pos, steps): decltype(tuple_cat(ADEntry{}, tuple<size_t>{}))

    fmgpu_index_t handle{};
    uint64_t n{};

    GpuIndexBase() = default;
    GpuIndexBase(GpuIndexBase const&) = delete;
    GpuIndexBase(GpuIndexBase&& o) noexcept : handle{o.handle}, n{o.n} { o.handle = nullptr; }
    auto operator=(GpuIndexBase&& o) noexcept -> GpuIndexBase& { std::swap(handle, o.handle); std::swap(n, o.n); return *this; }
    ~GpuIndexBase() { if (handle) fmgpu_index_destroy(handle); }

    // FMIndex(Sequences, samplingRate, threadNbr) / BiFMIndex(Sequences, samplingRate, threadNbr): built on the GPU
    template <typename Seqs>
    GpuIndexBase(Seqs const& input, size_t samplingRate, size_t /*threadNbr*/) {
        std::vector<uint8_t> buf; std::vector<uint64_t> off;
        detail::flatten(input, buf, off);
        detail::check(fmgpu_build_index(buf.data(), off.data(), off.size() - 1, static_cast<int32_t>(Sigma), String<Sigma>::layout, samplingRate,
                                        Bidirectional ? 1 : 0, 0, &handle, nullptr));
        detail::check(fmgpu_index_info(handle, &n, nullptr, nullptr, nullptr, nullptr));
    }

    size_t size() const { return n; }

    auto locate(size_t idx) const -> LEntry {
        uint64_t row = idx, seq{}, pos{}, steps{};
        detail::check(fmgpu_locate(handle, &row, 1, &seq, &pos, &steps, nullptr, nullptr));
        return {static_cast<uint32_t>(seq), static_cast<uint32_t>(pos), static_cast<size_t>(steps)};
    }
    // single_locate_step(idx) (fmindex/FMIndex.h:126-128): the sampled entry of row idx, if that row is sampled
    auto single_locate_step(size_t idx) const -> std::optional<std::tuple<uint32_t, uint32_t>> {
        auto [seq, pos, steps] = locate(idx);
        if (steps != 0) return std::nullopt;
        return std::tuple<uint32_t, uint32_t>{seq, pos};
    }
    auto locate(std::vector<uint64_t> const& rows) const -> std::vector<LEntry> {
        std::vector<uint64_t> seq(rows.size()), pos(rows.size()), steps(rows.size());
        detail::check(fmgpu_locate(handle, rows.data(), rows.size(), seq.data(), pos.data(), steps.data(), nullptr, nullptr));
        std::vector<LEntry> out(rows.size());
        for (size_t i = 0; i < rows.size(); ++i) out[i] = {static_cast<uint32_t>(seq[i]), static_cast<uint32_t>(pos[i]), static_cast<size_t>(steps[i])};
        return out;
    }
};

template <size_t TSigma, template <size_t> class String = string::FlattenedBitvectors_512_64k>   // fmindex/FMIndex.h:14
struct FMIndex : GpuIndexBase<TSigma, String, false> {
    using GpuIndexBase<TSigma, String, false>::GpuIndexBase;
};
template <size_t TSigma, template <size_t> class String = string::FlattenedBitvectors_512_64k>   // fmindex/BiFMIndex.h:17
struct BiFMIndex : GpuIndexBase<TSigma, String, true> {
    using GpuIndexBase<TSigma, String, true>::GpuIndexBase;
};

// saveIndex / loadIndex (fmindex/diskStorage.h:12-27) on this library's own index file (include/fmgpu.h: fmgpu_index_save / fmgpu_index_load — a
// flat file of the device arrays, NOT the reference's cereal archive).  withTables: also the optional tables the handle holds right now.
template <typename Index>
void saveIndex(Index const& index, std::string const& fileName, bool withTables = true) {
    detail::check(fmgpu_index_save(index.handle, fileName.c_str(), withTables ? 1 : 0));
}
template <typename Index>
auto loadIndex(std::string const& fileName) -> Index {
    Index index{};
    detail::check(fmgpu_index_load(fileName.c_str(), &index.handle));
    int32_t sigma{}, bidir{};
    detail::check(fmgpu_index_info(index.handle, &index.n, &sigma, nullptr, &bidir, nullptr));
    if (static_cast<size_t>(sigma) != Index::Sigma || (bidir != 0) != Index::IsBidirectional)
        throw std::runtime_error("loadIndex: " + fileName + " holds an index of Sigma " + std::to_string(sigma) + (bidir ? " (BiFMIndex)" : " (FMIndex)"));
    return index;
}

// ------------------------------------------------------------------------------------------------ cursors
struct IntIterator {   // utils.h:656-669
    size_t i;
    auto operator*() const -> size_t { return i; }
    auto operator++() -> IntIterator& { ++i; return *this; }
    bool operator!=(IntIterator const& o) const { return i != o.i; }
};
namespace detail {
// one cursor step on the device (fmgpu_cursor_extend): symb < 0 = every symbol
inline void extend(fmgpu_index_t h, int direction, uint64_t lb, uint64_t lbRev, uint64_t len, int symb, bool bidir, uint64_t* olb, uint64_t* orev, uint64_t* olen) {
    uint8_t c = static_cast<uint8_t>(symb < 0 ? 0 : symb);
    check(fmgpu_cursor_extend(h, direction, 1, &lb, bidir ? &lbRev : nullptr, &len, symb < 0 ? nullptr : &c, olb, bidir ? orev : nullptr, olen, nullptr));
}
inline size_t symbol_at(fmgpu_index_t h, int which, uint64_t row) {
    uint8_t what = 2; uint64_t out{};
    check(fmgpu_string_query(h, which, &row, nullptr, &what, 1, &out, nullptr));
    return static_cast<size_t>(out);
}
}  // namespace detail

// Single cursor steps go through the batched device call one cursor at a time: they are here so that code written against the reference's cursor
// API runs unchanged (tests, small tools); a search loop belongs into the batched search calls below.
template <typename Index>
struct FMIndexCursor {   // fmindex/FMIndexCursor.h:12-63
    static constexpr size_t Sigma = Index::Sigma;
    Index const* index{};
    size_t lb{}, len{};
    FMIndexCursor() = default;
    explicit FMIndexCursor(Index const& idx) : index{&idx}, lb{0}, len{idx.size()} {}
    FMIndexCursor(Index const& idx, size_t lb_, size_t len_) : index{&idx}, lb{lb_}, len{len_} {}
    bool empty() const { return len == 0; }
    size_t count() const { return len; }
    auto extendLeft(size_t symb) const -> FMIndexCursor {                            // :33-37
        uint64_t nlb{}, nlen{};
        detail::extend(index->handle, 0, lb, 0, len, static_cast<int>(symb), false, &nlb, nullptr, &nlen);
        return FMIndexCursor{*index, static_cast<size_t>(nlb), static_cast<size_t>(nlen)};
    }
    auto extendLeft() const -> std::array<FMIndexCursor, Sigma> {                    // :38-53
        std::array<uint64_t, Sigma> nlb{}, nlen{};
        detail::extend(index->handle, 0, lb, 0, len, -1, false, nlb.data(), nullptr, nlen.data());
        std::array<FMIndexCursor, Sigma> out;
        for (size_t c = 0; c < Sigma; ++c) out[c] = FMIndexCursor{*index, static_cast<size_t>(nlb[c]), static_cast<size_t>(nlen[c])};
        return out;
    }
};
template <typename Index>
struct BiFMIndexCursor {   // fmindex/BiFMIndexCursor.h:13-191
    static constexpr size_t Sigma = Index::Sigma;
    Index const* index{};
    size_t lb{}, lbRev{}, len{}, steps{};
    BiFMIndexCursor() = default;
    explicit BiFMIndexCursor(Index const& idx) : index{&idx}, lb{0}, lbRev{0}, len{idx.size()}, steps{0} {}
    BiFMIndexCursor(Index const& idx, size_t lb_, size_t lbRev_, size_t len_, size_t steps_) : index{&idx}, lb{lb_}, lbRev{lbRev_}, len{len_}, steps{steps_} {}
    bool empty() const { return len == 0; }
    size_t count() const { return len; }
    bool operator==(BiFMIndexCursor const& o) const noexcept { return lb == o.lb && len == o.len; }
    auto extendLeft(size_t symb) const -> BiFMIndexCursor { return step(0, symb); }   // :113-120
    auto extendRight(size_t symb) const -> BiFMIndexCursor { return step(1, symb); }  // :121-128
    auto extendLeft() const -> std::array<BiFMIndexCursor, Sigma> { return fan(0); }  // :58-69
    auto extendRight() const -> std::array<BiFMIndexCursor, Sigma> { return fan(1); } // :71-82
    auto symbolLeft() const -> size_t { return detail::symbol_at(index->handle, 0, lb); }      // :180-184
    auto symbolRight() const -> size_t { return detail::symbol_at(index->handle, 1, lbRev); }  // :186-190
private:
    auto step(int direction, size_t symb) const -> BiFMIndexCursor {
        uint64_t nlb{}, nrev{}, nlen{};
        detail::extend(index->handle, direction, lb, lbRev, len, static_cast<int>(symb), true, &nlb, &nrev, &nlen);
        return BiFMIndexCursor{*index, static_cast<size_t>(nlb), static_cast<size_t>(nrev), static_cast<size_t>(nlen), steps + 1};
    }
    auto fan(int direction) const -> std::array<BiFMIndexCursor, Sigma> {
        std::array<uint64_t, Sigma> nlb{}, nrev{}, nlen{};
        detail::extend(index->handle, direction, lb, lbRev, len, -1, true, nlb.data(), nrev.data(), nlen.data());
        std::array<BiFMIndexCursor, Sigma> out;
        for (size_t c = 0; c < Sigma; ++c) out[c] = BiFMIndexCursor{*index, static_cast<size_t>(nlb[c]), static_cast<size_t>(nrev[c]), static_cast<size_t>(nlen[c]), steps + 1};
        return out;
    }
};
template <typename C> auto begin(C const& c) -> decltype(IntIterator{c.lb}) { return IntIterator{c.lb}; }
template <typename C> auto end(C const& c) -> decltype(IntIterator{c.lb + c.len}) { return IntIterator{c.lb + c.len}; }

template <typename Index> struct select_cursor { using type = FMIndexCursor<Index>; };
template <size_t S, template <size_t> class Str> struct select_cursor<BiFMIndex<S, Str>> { using type = BiFMIndexCursor<BiFMIndex<S, Str>>; };
template <typename Index> using select_cursor_t = typename select_cursor<Index>::type;

// ------------------------------------------------------------------------------------------------ search schemes
namespace search_scheme {
struct Search {
    std::vector<size_t> pi, l, u;
    bool operator==(Search const& o) const { return std::tie(pi, l, u) == std::tie(o.pi, o.l, o.u); }
};
using Scheme = std::vector<Search>;

inline auto isValid(Search const& s) -> bool {   // isValid.h:55-93
    if (s.pi.empty() || s.pi.size() != s.l.size() || s.pi.size() != s.u.size()) return false;
    size_t lo = s.pi[0], hi = s.pi[0];
    for (size_t i = 1; i < s.pi.size(); ++i) {
        if (s.pi[i] == hi + 1) hi = s.pi[i];
        else if (s.pi[i] + 1 == lo) lo = s.pi[i];
        else return false;
    }
    if (lo != 0) return false;
    for (size_t i = 0; i < s.pi.size(); ++i) {
        if (i && (s.l[i - 1] > s.l[i] || s.u[i - 1] > s.u[i])) return false;
        if (s.l[i] > s.u[i]) return false;
    }
    return true;
}
inline auto isValid(Scheme const& ss) -> bool {
    return std::all_of(ss.begin(), ss.end(), [&](Search const& s) { return isValid(s) && s.pi.size() == ss.front().pi.size(); });
}

inline auto createUniformPartition(size_t parts, size_t totalSum) -> std::vector<size_t> {   // expand.h:324-335
    if (parts == 0 || totalSum < parts) throw std::invalid_argument("createUniformPartition: need 0 < parts <= totalSum");
    auto counts = std::vector<size_t>(parts, totalSum / parts);
    for (size_t i = 0; i < totalSum % parts; ++i) counts[i] += 1;
    return counts;
}
inline auto createUniformPartition(Scheme const& ss, size_t totalSum) -> std::vector<size_t> {
    if (ss.empty()) throw std::invalid_argument("createUniformPartition: empty scheme");
    return createUniformPartition(ss[0].pi.size(), totalSum);
}

// part p of the search becomes counts[p] parts (expand.h:167-177)
inline auto expand(Search const& s, std::vector<size_t> const& counts) -> std::optional<Search> {
    size_t P = s.pi.size(), newLen = 0;
    if (counts.size() != P) throw std::invalid_argument("expand: one count per part");
    std::vector<size_t> starts(P, 0);
    for (size_t i = 1; i < P; ++i) starts[i] = starts[i - 1] + counts[i - 1];
    for (size_t c : counts) newLen += c;
    Search r;
    for (size_t i = 0; i < P; ++i) {
        bool forward = i == 0 ? (P == 1 || s.pi[1] > s.pi[0]) : s.pi[i] > s.pi[i - 1];
        size_t lo = starts[s.pi[i]], cnt = counts[s.pi[i]];
        for (size_t j = 0; j < cnt; ++j) r.pi.push_back(forward ? lo + j : lo + cnt - 1 - j);
        if (cnt >= 1) { for (size_t j = 1; j < cnt; ++j) r.l.push_back(i ? s.l[i - 1] : 0); r.l.push_back(s.l[i]); }
        else if (!r.l.empty()) r.l.back() = s.l[i];
        for (size_t j = 0; j < cnt; ++j) r.u.push_back(s.u[i]);
    }
    if (r.pi.size() != newLen || !isValid(r)) return std::nullopt;
    return r;
}
inline auto expand(Search const& s, size_t newLen) -> std::optional<Search> {   // expand.h:146-155: uniformly
    size_t P = s.pi.size();
    std::vector<size_t> counts(P, newLen / P);
    for (size_t i = 0; i < newLen % P; ++i) counts[i] += 1;
    return expand(s, counts);
}
inline auto expand(Scheme const& ss, size_t newLen) -> Scheme {
    Scheme r;
    for (auto const& s : ss) if (auto o = expand(s, newLen)) r.push_back(*o);
    return r;
}
inline auto expand(Scheme const& ss, std::vector<size_t> const& counts) -> Scheme {   // expand.h:179-189
    Scheme r;
    for (auto const& s : ss) if (auto o = expand(s, counts)) r.push_back(*o);
    return r;
}
// nodes a search visits when a node of depth n survives with probability min(1, N / sigma^n) (weightedNodeCount.h:21-69; the weight in double, the sums in
// long double as there, so that expandByWNC takes the reference's decisions)
template <bool Edit>
inline long double weightedNodeCount(Search const& s, size_t sigma, size_t N) {
    size_t e = *std::max_element(s.u.begin(), s.u.end());
    std::vector<long double> last(e + 1, 0), cur(e + 1, 0);
    last[0] = 1;
    long double acc = 0;
    for (size_t n = 1; n <= s.pi.size(); ++n) {
        double f = static_cast<double>(N) / std::pow(static_cast<double>(sigma), static_cast<double>(n));
        if (f > 1) f = 1.;
        for (size_t i = 0; i <= e; ++i) {
            if (s.l[n - 1] <= i && i <= s.u[n - 1]) {
                cur[i] = last[i];
                if (i > 0) cur[i] += Edit ? (sigma - 1) * last[i - 1] + sigma * last[i - 1] + last[i - 1] : (sigma - 1) * last[i - 1];
                cur[i] *= f;
                acc += cur[i];
            } else cur[i] = 0;
        }
        std::swap(cur, last);
    }
    return acc;
}
template <bool Edit>
inline long double weightedNodeCount(Scheme const& ss, size_t sigma, size_t N) {
    long double v = 0;
    for (auto const& s : ss) v = v + weightedNodeCount<Edit>(s, sigma, N);
    return v;
}
// expand.h:218-247: the parts grow one position at a time, each time where the weighted node count of the expanded scheme is smallest
template <bool Edit = false>
inline auto optimizeByWNC(Scheme const& ss, size_t newLen, size_t sigma, size_t N) -> std::vector<size_t> {
    if (ss.empty()) return {};
    size_t P = ss[0].pi.size();
    if (newLen < P) throw std::invalid_argument("optimizeByWNC: the new length is shorter than the scheme");
    std::vector<size_t> counts(P, 1);
    for (size_t i = 0; i < newLen - P; ++i) {
        double best = std::numeric_limits<double>::max();      // (a double, like the reference's running best)
        size_t bestPos = 0;
        for (size_t j = 0; j < P; ++j) {
            counts[j] += 1;
            long double f = weightedNodeCount<Edit>(expand(ss, counts), sigma, N);
            counts[j] -= 1;
            if (f < best) { best = static_cast<double>(f); bestPos = j; }
        }
        counts[bestPos] += 1;
    }
    return counts;
}
template <bool Edit = false>
inline auto expandByWNC(Scheme const& ss, size_t newLen, size_t sigma, size_t N) -> Scheme {
    return expand(ss, optimizeByWNC<Edit>(ss, newLen, sigma, N));
}
inline auto limitToHamming(Scheme ss) -> Scheme {   // expand.h:301-319
    for (auto& s : ss) {
        for (size_t i = s.pi.size() - 1; i > 0; --i) { if (s.l[i] == 0) break; s.l[i - 1] = std::max(s.l[i - 1], s.l[i] - 1); }
        for (size_t i = 1; i < s.pi.size(); ++i) s.u[i] = std::min(s.u[i], s.u[i - 1] + 1);
    }
    return ss;
}
inline auto isComplete(Scheme const& ss, size_t minK, size_t maxK) -> bool {   // isComplete.h:69-84
    if (ss.empty()) return false;
    size_t P = ss[0].pi.size();
    std::vector<size_t> cfg(P, 0);
    auto covered = [&]() {
        for (auto const& s : ss) {
            size_t a = 0; bool ok = true;
            for (size_t i = 0; i < P && ok; ++i) { a += cfg[s.pi[i]]; ok = s.l[i] <= a && a <= s.u[i]; }
            if (ok) return true;
        }
        return false;
    };
    bool complete = minK > 0 || covered();
    auto rec = [&](auto&& self, size_t k, size_t start) -> void {
        if (k >= maxK || !complete) return;
        for (size_t i = start; i < P && complete; ++i) {
            cfg[i] += 1;
            if (k + 1 >= minK && !covered()) complete = false;
            self(self, k + 1, i);
            cfg[i] -= 1;
        }
    };
    rec(rec, 0, 0);
    return complete;
}

namespace generator {
inline auto backtracking(size_t N, size_t minK, size_t K) -> Scheme {   // generator/backtracking.h:14-21
    Search s{std::vector<size_t>(N), std::vector<size_t>(N, 0), std::vector<size_t>(N, K)};
    std::iota(s.pi.begin(), s.pi.end(), size_t{0});
    s.l.back() = minK;
    return {s};
}
inline auto pigeon(size_t minK, size_t K, bool opt) -> Scheme {   // generator/pigeon.h:14-102
    size_t N = K + 1;
    Scheme res;
    for (size_t i = 0; i < N; ++i) {
        Search s;
        s.pi.push_back(i); s.l.push_back(0); s.u.push_back(0);
        for (size_t j = i; j > 0; --j) { s.pi.push_back(j - 1); s.l.push_back(opt ? i - j + 1 : 0); s.u.push_back(opt ? K - j + 1 : K); }
        for (size_t j = i + 1; j < N; ++j) { s.pi.push_back(j); s.l.push_back(opt ? i : 0); s.u.push_back(K); }
        s.l.back() = std::max(s.l.back(), minK);
        res.push_back(s);
    }
    return res;
}
inline auto pigeon_opt(size_t minK, size_t K) -> Scheme { return pigeon(minK, K, true); }
inline auto pigeon_trivial(size_t minK, size_t K) -> Scheme { return pigeon(minK, K, false); }

inline auto h2(size_t N, size_t minK, size_t K) -> Scheme {   // generator/h2.h:128-153
    if (N <= K || minK > K) throw std::invalid_argument("h2(N, minK, K) needs N > K >= minK");
    size_t R = K + 1;
    auto at = [&](std::vector<size_t>& m, size_t r, size_t c) -> size_t& { return m[r * N + c]; };
    std::vector<size_t> pi(R * N), l(R * N, 0), u(R * N, 0), d(R * N, 0);
    for (size_t r = 0; r < R; ++r) for (size_t c = 0; c < N; ++c) {
        size_t skip = K - r;
        at(pi, r, c) = c < N - skip ? c + skip : N - c - 1;
        if (c >= N - (K - r + 1)) at(l, r, c) = r;
        at(d, r, c) = c >= K ? K - r : (r < K ? (r + K - c) % K : K);
    }
    auto fits = [&](size_t row, size_t col, size_t v) {
        if (row == col) return false;
        if (row > col) { for (size_t i = 0; i < col; ++i) if (at(d, row, i) < v) return false; }
        else for (size_t i = row + 1; i < col; ++i) if (at(d, row, i) > v) return false;
        return true;
    };
    for (size_t c = 0; c < N; ++c) for (size_t r = 0; r < R; ++r) {
        if (c == r || at(d, r, c) == 0 || fits(r, c, at(d, r, c))) continue;
        for (size_t o = r + 1; o < R; ++o)
            if (fits(r, c, at(d, o, c)) && fits(o, c, at(d, r, c))) { std::swap(at(d, r, c), at(d, o, c)); break; }
    }
    for (size_t c = 1; c < N; ++c) for (size_t r = R; r-- > 0;)
        at(u, r, c) = std::max(at(u, r, c - 1), at(l, r, c - 1) + at(d, K - r, at(pi, r, c)));
    Scheme ss;
    for (size_t r = 0; r < R; ++r) {
        Search s;
        s.pi.assign(pi.begin() + r * N, pi.begin() + (r + 1) * N);
        s.l.assign(l.begin() + r * N, l.begin() + (r + 1) * N);
        s.u.assign(u.begin() + r * N, u.begin() + (r + 1) * N);
        s.l.back() = std::max(s.l.back(), minK);
        ss.push_back(s);
    }
    return ss;
}
}  // namespace generator
}  // namespace search_scheme

// ------------------------------------------------------------------------------------------------ searches
namespace detail {
template <typename Index, typename Delegate>
void report(Index const& index, std::vector<fmgpu_hit>& hits, Delegate&& delegate) {
    check(fmgpu_hits_sort(hits.data(), hits.size(), nullptr));    // (qidx, seq): the reference's callback order, sorted on the device
    using cursor_t = select_cursor_t<Index>;
    for (auto const& h : hits) {
        cursor_t cur{};
        cur.index = &index; cur.lb = h.lb; cur.len = h.len;
        if constexpr (std::is_same_v<cursor_t, BiFMIndexCursor<Index>>) cur.lbRev = h.lb_rev;
        delegate(static_cast<size_t>(h.qidx), cur, static_cast<size_t>(h.errors));
    }
}
template <typename Call>
std::vector<fmgpu_hit> run_hits(size_t nq, Call&& call) {
    std::vector<fmgpu_hit> hits(std::max<size_t>(1024, 4 * nq));
    for (;;) {
        uint64_t count = 0;
        int rc = call(hits.data(), hits.size(), &count);
        if (rc == FMGPU_ERR_CAPACITY) { hits.resize(count); continue; }
        check(rc);
        hits.resize(count);
        return hits;
    }
}
}  // namespace detail

namespace search_no_errors {
// search(index, queries, delegate(qidx, cursor)) — search/SearchNoErrors.h:28-86; only non-empty cursors are reported
template <typename Index, typename Queries, typename Delegate>
void search(Index const& index, Queries const& queries, Delegate&& delegate, size_t /*BatchSize*/ = 32) {
    std::vector<uint8_t> buf; std::vector<uint64_t> off;
    detail::flatten(queries, buf, off);
    size_t nq = off.size() - 1;
    std::vector<uint64_t> lb(nq), len(nq);
    detail::check(fmgpu_search_exact(index.handle, buf.data(), off.data(), nq, lb.data(), len.data(), nullptr, nullptr));
    using cursor_t = select_cursor_t<Index>;
    for (size_t q = 0; q < nq; ++q) {
        if (len[q] == 0) continue;
        cursor_t cur{};
        cur.index = &index; cur.lb = lb[q]; cur.len = len[q];
        delegate(q, cur);
    }
}
}  // namespace search_no_errors

namespace search_backtracking {
// search(index, queries, maxError, delegate(qidx, cursor, errors)) — search/Backtracking.h:85-89
template <typename Index, typename Queries, typename Delegate>
void search(Index const& index, Queries const& queries, size_t maxError, Delegate&& delegate) {
    std::vector<uint8_t> buf; std::vector<uint64_t> off;
    detail::flatten(queries, buf, off);
    size_t nq = off.size() - 1;
    auto hits = detail::run_hits(nq, [&](fmgpu_hit* out, uint64_t cap, uint64_t* count) {
        return fmgpu_search_backtracking(index.handle, buf.data(), off.data(), nq, maxError, out, cap, count, nullptr, nullptr);
    });
    detail::report(index, hits, delegate);
}
}  // namespace search_backtracking

namespace search_ng26 {
namespace detail2 {
// one scheme over one batch; qmap (optional) renames the batch's query numbers
template <bool Edit, typename Index>
std::vector<fmgpu_hit> run(Index const& index, std::vector<uint8_t> const& buf, std::vector<uint64_t> const& off, search_scheme::Scheme const& scheme,
                           std::vector<size_t> const& partition, size_t n, std::vector<uint64_t> const* qmap) {
    size_t nq = off.size() - 1;
    if (scheme.empty() || nq == 0 || n == 0) return {};
    size_t P = scheme[0].pi.size();
    std::vector<uint64_t> pi, l, u, part(partition.begin(), partition.end());
    for (auto const& s : scheme) {
        if (s.pi.size() != P) throw std::runtime_error("fmindex-collection (gpu): searches of a scheme must have the same number of parts");
        pi.insert(pi.end(), s.pi.begin(), s.pi.end()); l.insert(l.end(), s.l.begin(), s.l.end()); u.insert(u.end(), s.u.begin(), s.u.end());
    }
    fmgpu_scheme sc{static_cast<int32_t>(scheme.size()), static_cast<int32_t>(P), pi.data(), l.data(), u.data(), part.empty() ? nullptr : part.data(),
                    Edit ? 1 : 0, 0};
    auto hits = detail::run_hits(nq, [&](fmgpu_hit* out, uint64_t cap, uint64_t* count) {
        return fmgpu_search_scheme(index.handle, buf.data(), off.data(), nq, &sc, n, out, cap, count, nullptr, nullptr);
    });
    if (qmap) for (auto& h : hits) h.qidx = (*qmap)[h.qidx];
    return hits;
}
}  // namespace detail2

// search<Edit>(index, queries, scheme, partition, delegate(qidx, cursor, errors), n) — search/SearchNg26.h:426-433.
// Edit = true (the reference's default) adds insertions and deletions (:146-218, :286-362); Edit = false is Hamming distance.
template <bool Edit = true, typename Index, typename Queries, typename Delegate>
void search(Index const& index, Queries const& queries, search_scheme::Scheme const& scheme, std::vector<size_t> const& partition,
            Delegate&& delegate, size_t n = std::numeric_limits<size_t>::max()) {
    std::vector<uint8_t> buf; std::vector<uint64_t> off;
    detail::flatten(queries, buf, off);
    auto hits = detail2::run<Edit>(index, buf, off, scheme, partition, n, nullptr);
    detail::report(index, hits, delegate);
}
// search<Edit>(index, queries, maxErrors, delegate, n) — search/SearchNg26.h:436-444: per query length the cached scheme
// h2(maxErrors + (length == 2 ? 1 : 2), 0, maxErrors) (CachedSearchScheme.h:16-36) with a uniform partition.  For Edit = false the
// reference additionally applies limitToHamming to the un-expanded scheme, which loses hits — not reproduced here.
template <bool Edit = true, typename Index, typename Queries, typename Delegate>
void search(Index const& index, Queries const& queries, size_t maxErrors, Delegate&& delegate, size_t n = std::numeric_limits<size_t>::max()) {
    std::vector<fmgpu_hit> all;
    for (int shortLen = 0; shortLen < 2; ++shortLen) {
        std::vector<uint8_t> buf; std::vector<uint64_t> off{0}, qmap;
        size_t qidx = 0;
        for (auto const& q : queries) {
            if ((q.size() == 2) == (shortLen == 1)) { buf.insert(buf.end(), q.begin(), q.end()); off.push_back(buf.size()); qmap.push_back(qidx); }
            ++qidx;
        }
        if (qmap.empty()) continue;
        auto hits = detail2::run<Edit>(index, buf, off, search_scheme::generator::h2(maxErrors + (shortLen ? 1 : 2), 0, maxErrors), {}, n, &qmap);
        all.insert(all.end(), hits.begin(), hits.end());
    }
    detail::report(index, all, delegate);
}
// search_best<Edit>(index, queries, {(scheme, partition), ...}, delegate, n) — search/SearchNg26.h:447-473: per query the first scheme
// that reports anything wins
template <bool Edit = true, typename Index, typename Queries, typename Delegate>
void search_best(Index const& index, Queries const& queries, std::vector<std::tuple<search_scheme::Scheme, std::vector<size_t>>> const& schemes,
                 Delegate&& delegate, size_t n = std::numeric_limits<size_t>::max()) {
    std::vector<uint64_t> todo;
    for (size_t i = 0; i < queries.size(); ++i) todo.push_back(i);
    std::vector<fmgpu_hit> all;
    for (auto const& [scheme, partition] : schemes) {
        if (todo.empty()) break;
        std::vector<uint8_t> buf; std::vector<uint64_t> off{0};
        for (auto qi : todo) { auto const& q = queries[qi]; buf.insert(buf.end(), q.begin(), q.end()); off.push_back(buf.size()); }
        auto hits = detail2::run<Edit>(index, buf, off, scheme, partition, n, &todo);
        std::vector<uint8_t> found(queries.size(), 0);
        for (auto const& h : hits) found[h.qidx] = 1;
        all.insert(all.end(), hits.begin(), hits.end());
        std::vector<uint64_t> rest;
        for (auto qi : todo) if (!found[qi]) rest.push_back(qi);
        todo.swap(rest);
    }
    detail::report(index, all, delegate);
}
// search_best<Edit>(index, queries, maxErrors, delegate, n) — search/SearchNg26.h:476-487: the whole batch with 0, 1, ... maxErrors - 1
// errors (the reference's loop ends before maxErrors), stopping at the first error count for which any query reports a hit
template <bool Edit = true, typename Index, typename Queries, typename Delegate>
void search_best(Index const& index, Queries const& queries, size_t maxErrors, Delegate&& delegate, size_t n = std::numeric_limits<size_t>::max()) {
    bool found = false;
    for (size_t i = 0; i < maxErrors && !found; ++i)
        search<Edit>(index, queries, i, [&](size_t qidx, auto cursor, size_t e) { if (cursor.count() == 0) return; found = true; delegate(qidx, cursor, e); }, n);
}
}  // namespace search_ng26

// search/SearchNg21.h: edit-distance search over an EXPANDED scheme (search_scheme::expand(scheme, query length): one {pi, l, u} entry per
// query symbol), so the queries of a call have that length; shorter ones — which the reference would read out of bounds — report nothing.
namespace search_ng21 {
namespace detail2 {
template <typename Index>
std::vector<fmgpu_hit> run(Index const& index, std::vector<uint8_t> const& buf, std::vector<uint64_t> const& off, search_scheme::Scheme const& scheme,
                           size_t n, std::vector<uint64_t> const* qmap) {
    size_t nq = off.size() - 1;
    if (scheme.empty() || nq == 0) return {};                                         // :205
    size_t M = scheme[0].pi.size();
    std::vector<uint64_t> pi, l, u;
    for (auto const& s : scheme) {
        if (s.pi.size() != M || s.l.size() != M || s.u.size() != M) throw std::runtime_error("fmindex-collection (gpu): searches of an expanded scheme must have the same length");
        pi.insert(pi.end(), s.pi.begin(), s.pi.end()); l.insert(l.end(), s.l.begin(), s.l.end()); u.insert(u.end(), s.u.begin(), s.u.end());
    }
    fmgpu_expanded_scheme sc{static_cast<int32_t>(scheme.size()), 0, M, pi.data(), l.data(), u.data()};
    auto hits = detail::run_hits(nq, [&](fmgpu_hit* out, uint64_t cap, uint64_t* count) {
        return fmgpu_search_ng21(index.handle, buf.data(), off.data(), nq, &sc, n, out, cap, count, nullptr, nullptr);
    });
    if (qmap) for (auto& h : hits) h.qidx = (*qmap)[h.qidx];
    return hits;
}
template <typename Index, typename Queries, typename Delegate>
void best(Index const& index, Queries const& queries, std::vector<search_scheme::Scheme> const& schemes, size_t n, Delegate&& delegate) {
    std::vector<uint64_t> todo;
    for (size_t i = 0; i < queries.size(); ++i) todo.push_back(i);
    std::vector<fmgpu_hit> all;
    for (auto const& scheme : schemes) {
        if (todo.empty()) break;
        std::vector<uint8_t> buf; std::vector<uint64_t> off{0};
        for (auto qi : todo) { auto const& q = queries[qi]; buf.insert(buf.end(), q.begin(), q.end()); off.push_back(buf.size()); }
        auto hits = run(index, buf, off, scheme, n, &todo);
        std::vector<uint8_t> found(queries.size(), 0);
        for (auto const& h : hits) if (h.len) found[h.qidx] = 1;                      // `if (ct > 0) break;` (:261, :290)
        all.insert(all.end(), hits.begin(), hits.end());
        std::vector<uint64_t> rest;
        for (auto qi : todo) if (!found[qi]) rest.push_back(qi);
        todo.swap(rest);
    }
    detail::report(index, all, delegate);
}
}  // namespace detail2

// search(index, queries, search_scheme, delegate(qidx, cursor, errors)) — :205-217
template <typename Index, typename Queries, typename Delegate>
void search(Index const& index, Queries const& queries, search_scheme::Scheme const& scheme, Delegate&& delegate) {
    std::vector<uint8_t> buf; std::vector<uint64_t> off;
    detail::flatten(queries, buf, off);
    auto hits = detail2::run(index, buf, off, scheme, std::numeric_limits<size_t>::max(), nullptr);
    detail::report(index, hits, delegate);
}
// search_n(index, queries, search_scheme, n, delegate) — :220-240: at most n rows per query, the last cursor clipped
template <typename Index, typename Queries, typename Delegate>
void search_n(Index const& index, Queries const& queries, search_scheme::Scheme const& scheme, size_t n, Delegate&& delegate) {
    std::vector<uint8_t> buf; std::vector<uint64_t> off;
    detail::flatten(queries, buf, off);
    auto hits = detail2::run(index, buf, off, scheme, n, nullptr);
    detail::report(index, hits, delegate);
}
// search_best(index, queries, search_schemes, delegate) — :242-264: per query the first scheme of the list that reports any row
template <typename Index, typename Queries, typename Delegate>
void search_best(Index const& index, Queries const& queries, std::vector<search_scheme::Scheme> const& schemes, Delegate&& delegate) {
    detail2::best(index, queries, schemes, std::numeric_limits<size_t>::max(), std::forward<Delegate>(delegate));
}
// search_best_n(index, queries, search_schemes, n, delegate) — :267-293
template <typename Index, typename Queries, typename Delegate>
void search_best_n(Index const& index, Queries const& queries, std::vector<search_scheme::Scheme> const& schemes, size_t n, Delegate&& delegate) {
    detail2::best(index, queries, schemes, n, std::forward<Delegate>(delegate));
}
}  // namespace search_ng21

// fmc::search<EditDistance>(index, queries, errors, delegate(qidx, cursor, errors)) — search/search.h:26-35
template <bool EditDistance, typename Index, typename Queries, typename Delegate>
void search(Index const& index, Queries const& queries, size_t errors, Delegate&& delegate) {
    if (errors == 0) search_no_errors::search(index, queries, [&](size_t qidx, auto const& cursor) { delegate(qidx, cursor, size_t{0}); });
    else search_ng26::search<EditDistance>(index, queries, errors, std::forward<Delegate>(delegate));
}
// fmc::search_n<EditDistance>(index, queries, errors, n, delegate) — search/search.h:43-46
template <bool EditDistance, typename Index, typename Queries, typename Delegate>
void search_n(Index const& index, Queries const& queries, size_t errors, size_t n, Delegate&& delegate) {
    search_ng26::search<EditDistance>(index, queries, errors, std::forward<Delegate>(delegate), n);
}

// LocateLinear{index, cursor}: for (auto [seqId, pos, offset] : LocateLinear{index, cursor}) — locate.h:14-57 (one batched call)
template <typename Index, typename Cursor>
struct LocateLinear {
    std::vector<typename Index::LEntry> entries;
    LocateLinear(Index const& index, Cursor const& cursor) {
        std::vector<uint64_t> rows(cursor.len);
        std::iota(rows.begin(), rows.end(), uint64_t{cursor.lb});
        entries = index.locate(rows);
    }
    auto begin() const { return entries.begin(); }
    auto end() const { return entries.end(); }
};
template <typename Index, typename Cursor> LocateLinear(Index const&, Cursor const&) -> LocateLinear<Index, Cursor>;

// fmc::Search{index, queries, editDistance, errors, maxResults, reportFunc}() — search/search.h:48-75: searches, locates every row of every
// reported cursor and calls reportFunc(qidx, seqId, pos + offset, errors)
template <typename index_t, typename queries_t, typename delegate_t>
struct Search {
    index_t const&        index;
    queries_t const&      queries;
    bool                  editDistance{true};
    size_t                errors{0};
    std::optional<size_t> maxResults{};
    delegate_t const&     reportFunc;
    void operator()() {
        auto report = [&](size_t qidx, auto const& cursor, size_t e) {
            for (auto [sid, spos, offset] : LocateLinear{index, cursor}) reportFunc(qidx, sid, spos + offset, e);
        };
        if (maxResults) {
            if (editDistance) search_n<true>(index, queries, errors, *maxResults, report);
            else search_n<false>(index, queries, errors, *maxResults, report);
        } else {
            if (editDistance) search<true>(index, queries, errors, report);
            else search<false>(index, queries, errors, report);
        }
    }
};
template <typename I, typename Q, typename D> Search(I const&, Q const&, bool, size_t, std::optional<size_t>, D const&) -> Search<I, Q, D>;

// Library options (include/fmgpu.h: fmgpu_option) — what a new handle is given, which of several result-identical kernels serves a call; the library reads no
// environment variable.  setOption(FMGPU_OPT_LF_TABLE, 0) before an index is made keeps it the plain configuration (occurrence tables + sampled suffix array).
inline void setOption(fmgpu_option option, int64_t value) { detail::check(fmgpu_set_option(static_cast<int32_t>(option), value)); }
inline auto getOption(fmgpu_option option) -> int64_t { int64_t v{}; detail::check(fmgpu_get_option(static_cast<int32_t>(option), &v)); return v; }

// One index file on several GPUs of this process (include/fmgpu.h: fmgpu_replicas_*): loadReplicas<Index>(file, {0, 1, 2, 3}) reads the file ONCE, onto the first
// listed device, and copies the handle to the others device to device (peerCopies() tells how many replicas were made that way; empty list: every visible device);
// searches cut the batch into contiguous ranges, one per replica, run them concurrently and report in batch order.
// `front()` is the first replica as an Index (locate and cursor steps go through it with its device current).
template <typename Index>
struct Replicas {
    fmgpu_replicas_t handle{nullptr};
    Index first{};
    Replicas() = default;
    Replicas(Replicas const&) = delete;
    Replicas& operator=(Replicas const&) = delete;
    Replicas(Replicas&& o) noexcept : handle{o.handle}, first{std::move(o.first)} { o.handle = nullptr; o.first.handle = nullptr; }
    ~Replicas() { first.handle = nullptr; if (handle) fmgpu_replicas_destroy(handle); }       // (the first replica's handle is borrowed from the set)
    auto size() const -> size_t { int32_t n{}; detail::check(fmgpu_replicas_info(handle, &n, nullptr, 0, nullptr)); return static_cast<size_t>(n); }
    auto front() const -> Index const& { return first; }
    auto peerCopies() const -> size_t { int32_t n{}; detail::check(fmgpu_replicas_peer_copies(handle, &n)); return static_cast<size_t>(n); }
    // search_no_errors::search over the replicas: delegate(qidx, cursor) for every non-empty cursor, in batch order
    template <typename Queries, typename Delegate>
    void searchNoErrors(Queries const& queries, Delegate&& delegate) const {
        std::vector<uint8_t> buf; std::vector<uint64_t> off;
        detail::flatten(queries, buf, off);
        size_t nq = off.size() - 1;
        std::vector<uint64_t> lb(nq), len(nq);
        detail::check(fmgpu_replicas_search_exact(handle, buf.data(), off.data(), nq, lb.data(), len.data(), nullptr));
        using cursor_t = select_cursor_t<Index>;
        for (size_t q = 0; q < nq; ++q) {
            if (len[q] == 0) continue;
            cursor_t cur{};
            cur.index = &first; cur.lb = lb[q]; cur.len = len[q];
            delegate(q, cur);
        }
    }
};
template <typename Index>
auto loadReplicas(std::string const& fileName, std::vector<int32_t> const& devices = {}) -> Replicas<Index> {
    Replicas<Index> r;
    detail::check(fmgpu_replicas_load(fileName.c_str(), devices.empty() ? nullptr : devices.data(), static_cast<int32_t>(devices.size()), &r.handle));
    detail::check(fmgpu_replicas_info(r.handle, nullptr, nullptr, 0, &r.first.handle));
    int32_t sigma{}, bidir{};
    detail::check(fmgpu_index_info(r.first.handle, &r.first.n, &sigma, nullptr, &bidir, nullptr));
    if (static_cast<size_t>(sigma) != Index::Sigma || (bidir != 0) != Index::IsBidirectional)
        throw std::runtime_error("loadReplicas: " + fileName + " holds an index of Sigma " + std::to_string(sigma) + (bidir ? " (BiFMIndex)" : " (FMIndex)"));
    return r;
}

}  // namespace fmc

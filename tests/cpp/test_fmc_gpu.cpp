// The reference's own tests for the path, re-stated against include/fmc_gpu.hpp (same calls, same expected values):
// search/checkSearches.cpp:14-72, :104-117, :1173-1199; search/checkSearchBacktracking.cpp:42-102; fmindex/checkBiFMIndexCursor.cpp:12-30;
// fmindex/checkBiFMIndexCursor.cpp:12-103; fmindex/checkFMIndexCursor.cpp:13-66; search_scheme/expand.cpp:11-60; search_scheme/checkGeneratorsIsComplete.cpp:48-60.  Needs a GPU; exit code 0 = all checks passed.
#include "../../include/fmc_gpu.hpp"

#include <cstdio>
#include <optional>
#include <tuple>
#include <vector>

static int failures = 0;
#define CHECK(x) do { if (!(x)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #x); ++failures; } } while (0)

using Results = std::vector<std::tuple<size_t, size_t, size_t>>;

// `test_fmc_gpu wnc`: for every line "gen len edit sigma N" on stdin print expandByWNC's scheme as "searches parts" and the three flattened tables
// (tests/test_cpp_mirror.py compares them with the real reference's, tests/golden/ref_schemes.json)
static int wncMode() {
    namespace ss = fmc::search_scheme;
    char gen[64]; unsigned long long len, sigma, N; int edit;
    while (std::scanf("%63s %llu %d %llu %llu", gen, &len, &edit, &sigma, &N) == 5) {
        std::string const g = gen;
        ss::Scheme s = g == "h2-k1" ? ss::generator::h2(3, 0, 1) : g == "h2-k2" ? ss::generator::h2(4, 0, 2) : g == "h2-k3" ? ss::generator::h2(5, 0, 3)
                     : g == "pigeon_opt-k2" ? ss::generator::pigeon_opt(0, 2) : g == "pigeon-k1" ? ss::generator::pigeon_trivial(0, 1) : ss::generator::backtracking(1, 0, 2);
        auto const e = edit ? ss::expandByWNC<true>(s, len, sigma, N) : ss::expandByWNC<false>(s, len, sigma, N);
        std::printf("%zu %zu\n", e.size(), e.empty() ? size_t{0} : e[0].pi.size());
        for (int t = 0; t < 3; ++t) {
            for (auto const& x : e) for (size_t v : (t == 0 ? x.pi : t == 1 ? x.l : x.u)) std::printf("%zu ", v);
            std::printf("\n");
        }
        std::printf("%.17g\n", static_cast<double>(edit ? ss::weightedNodeCount<true>(e, sigma, N) : ss::weightedNodeCount<false>(e, sigma, N)));
    }
    return 0;
}

int main(int argc, char** argv) {
    namespace ss = fmc::search_scheme;
    if (argc > 1 && std::string(argv[1]) == "wnc") return wncMode();
    {   // search_scheme/expand.cpp
        auto real = ss::expand(ss::Scheme{ss::Search{{0, 1}, {0, 0}, {0, 1}}}, 4);
        CHECK(ss::isValid(real));
        CHECK((real == ss::Scheme{ss::Search{{0, 1, 2, 3}, {0, 0, 0, 0}, {0, 0, 1, 1}}}));
        real = ss::expand(ss::Scheme{ss::Search{{0, 1}, {0, 0}, {0, 0}}}, 10);
        CHECK(real.size() == 1 && real[0].pi.size() == 10 && real[0].u.back() == 0);
        for (size_t N = 1; N < 10; ++N)
            for (size_t minK = 0; minK < std::min(N, size_t{5}); ++minK)
                for (size_t maxK = minK; maxK < std::min(N, size_t{5}); ++maxK)
                    CHECK(ss::isComplete(ss::generator::h2(N, minK, maxK), minK, maxK));
        auto h = ss::generator::h2(4, 0, 2);
        CHECK((h[0] == ss::Search{{2, 3, 1, 0}, {0, 0, 0, 0}, {0, 0, 2, 2}}));
        CHECK((h[1] == ss::Search{{1, 2, 3, 0}, {0, 0, 1, 1}, {0, 1, 1, 2}}));
        CHECK((h[2] == ss::Search{{0, 1, 2, 3}, {0, 0, 0, 2}, {0, 1, 2, 2}}));
        CHECK((ss::createUniformPartition(h, 101) == std::vector<size_t>{26, 25, 25, 25}));
    }
    int ndev = 0;
    if (fmgpu_device_count(&ndev) != 0 || ndev == 0) { std::printf("no GPU: host-only checks %s\n", failures ? "FAILED" : "passed"); return failures ? 1 : 77; }

    auto input = std::vector<std::vector<uint8_t>>{{'A', 'A', 'A', 'C', 'A', 'A', 'A', 'B', 'A', 'A', 'A'}, {'A', 'A', 'A', 'B', 'A', 'A', 'A', 'C', 'A', 'A', 'A'}};
    auto queries = std::vector<std::vector<uint8_t>>{{'C', 'C'}, {'B', 'B'}};
    auto expected = Results{{0, 0, 2}, {0, 0, 3}, {0, 1, 6}, {0, 1, 7}, {1, 0, 6}, {1, 0, 7}, {1, 1, 2}, {1, 1, 3}};
    {   // "check searches with errors": BiFMIndex<256>, samplingRate 1
        using Index = fmc::BiFMIndex<256>;
        auto index = Index{input, /*samplingRate*/ 1, /*threadNbr*/ 1};
        auto results = Results{};
        fmc::search_backtracking::search(index, queries, 1, [&](auto qidx, auto cursor, auto errors) {
            (void)errors;
            for (auto [sid, spos, offset] : fmc::LocateLinear{index, cursor}) results.emplace_back(qidx, sid, spos + offset);
        });
        std::sort(results.begin(), results.end());
        CHECK(results == expected);

        results.clear();
        fmc::search_no_errors::search(index, queries, [&](auto qidx, auto cursor) {
            for (auto [sid, spos, offset] : fmc::LocateLinear{index, cursor}) results.emplace_back(qidx, sid, spos + offset);
        });
        CHECK(results.empty());

        results.clear();
        auto scheme = ss::generator::pigeon_opt(0, 1);
        auto partition = ss::createUniformPartition(scheme, queries[0].size());
        fmc::search_ng26::search</*EditDistance=*/false>(index, queries, scheme, partition, [&](auto qidx, auto cursor, auto errors) {
            (void)errors;
            for (auto [sid, spos, offset] : fmc::LocateLinear{index, cursor}) results.emplace_back(qidx, sid, spos + offset);
        });
        std::sort(results.begin(), results.end());
        CHECK(results == expected);

        // edit distance (Edit = true is the default template argument): checkSearches.cpp "search ng26, all search" / "all search_n",
        // "search, all search, no search scheme", "search, all search_n, no search scheme"
        auto locate_all = [&](auto qidx, auto cursor, auto errors) {
            (void)errors;
            for (auto [sid, spos, offset] : fmc::LocateLinear{index, cursor}) results.emplace_back(qidx, sid, spos + offset);
        };
        auto q2 = std::vector<std::vector<uint8_t>>{{'C', 'D'}, {'D', 'B'}};
        results.clear();
        fmc::search_ng26::search(index, q2, scheme, partition, locate_all);
        std::sort(results.begin(), results.end());
        CHECK((results == Results{{0, 0, 3}, {0, 1, 7}, {1, 0, 7}, {1, 1, 3}}));
        results.clear();
        fmc::search_ng26::search(index, queries, scheme, partition, locate_all, 3);
        std::sort(results.begin(), results.end());
        CHECK((results == Results{{0, 0, 3}, {0, 1, 7}, {0, 1, 7}, {1, 0, 7}, {1, 0, 7}, {1, 1, 3}}));
        results.clear();
        fmc::search</*EditDistance=*/true>(index, queries, /*maxErrors*/ 1, locate_all);
        std::sort(results.begin(), results.end());
        CHECK((results == Results{{0, 0, 3}, {0, 0, 3}, {0, 1, 7}, {0, 1, 7}, {1, 0, 7}, {1, 0, 7}, {1, 1, 3}, {1, 1, 3}}));
        results.clear();                                   // search_best: nothing at 0 errors, and the loop ends before maxErrors = 1 ...
        fmc::search_ng26::search_best(index, queries, /*maxErrors*/ 1, locate_all);
        CHECK(results.empty());
        fmc::search_ng26::search_best(index, queries, /*maxErrors*/ 2, locate_all);     // ... so 2 is needed to see the 1-error hits
        CHECK(results.size() == 8);
        results.clear();
        fmc::search_n</*EditDistance=*/true>(index, queries, /*maxErrors*/ 1, /*n*/ 3, locate_all);
        std::sort(results.begin(), results.end());
        CHECK((results == Results{{0, 0, 3}, {0, 1, 7}, {0, 1, 7}, {1, 0, 7}, {1, 0, 7}, {1, 1, 3}}));
        // search_ng21 over expanded schemes: checkSearches.cpp "search ng21, all search" / "all search_n" / "all search_best" / "all search_best_n"
        auto ex = [&](size_t minK, size_t maxK) { return ss::expand(ss::generator::pigeon_opt(minK, maxK), queries[0].size()); };
        auto all8 = Results{{0, 0, 3}, {0, 0, 3}, {0, 1, 7}, {0, 1, 7}, {1, 0, 7}, {1, 0, 7}, {1, 1, 3}, {1, 1, 3}};
        auto top3 = Results{{0, 0, 3}, {0, 1, 7}, {0, 1, 7}, {1, 0, 7}, {1, 0, 7}, {1, 1, 3}};
        results.clear();
        fmc::search_ng21::search(index, queries, ex(0, 1), locate_all);
        std::sort(results.begin(), results.end());
        CHECK(results == all8);
        results.clear();
        fmc::search_ng21::search_n(index, queries, ex(0, 1), 3, locate_all);
        std::sort(results.begin(), results.end());
        CHECK(results == top3);
        results.clear();
        fmc::search_ng21::search_best(index, queries, std::vector{ex(0, 0), ex(1, 1), ex(2, 2)}, locate_all);
        std::sort(results.begin(), results.end());
        CHECK(results == all8);
        results.clear();
        fmc::search_ng21::search_best_n(index, queries, std::vector{ex(0, 0), ex(1, 1)}, 3, locate_all);
        std::sort(results.begin(), results.end());
        CHECK(results == top3);
    }
    {   // "backtracking with errors": FMIndex<256>
        auto index = fmc::FMIndex<256>{input, 1, 1};
        auto results = Results{};
        fmc::search_backtracking::search(index, queries, 1, [&](auto qidx, auto cursor, auto errors) {
            (void)errors;
            for (auto [sid, spos, offset] : fmc::LocateLinear{index, cursor}) results.emplace_back(qidx, sid, spos + offset);
        });
        std::sort(results.begin(), results.end());
        CHECK(results == expected);
    }
    {   // "searching with collection and backtracking": interval of 'A' and the locate table
        auto in2 = std::vector<std::vector<uint8_t>>{{'A', 'A', 'A', 'C', 'A', 'A', 'A', 'C', 'A', 'A', 'A'}, {'A', 'A', 'A', 'B', 'A', 'A', 'A', 'B', 'A', 'A', 'A'}};
        auto index = fmc::BiFMIndex<255>{in2, 1, 1};
        CHECK(index.size() == 24);
        auto query = std::vector<std::vector<uint8_t>>{{'A'}};
        size_t calls = 0;
        fmc::search_backtracking::search(index, query, 0, [&](auto qidx, auto result, auto errors) {
            ++calls;
            CHECK(qidx == 0); CHECK(errors == 0); CHECK(result.lb == 2); CHECK(result.count() == 18);
        });
        CHECK(calls == 1);
        auto exp = std::vector<std::tuple<uint32_t, uint32_t>>{{1, 11}, {0, 11}, {1, 10}, {0, 10}, {1, 9}, {0, 9}, {1, 8}, {0, 8}, {1, 4}, {1, 0}, {0, 4}, {0, 0},
                                                               {1, 5}, {1, 1}, {0, 5}, {0, 1}, {1, 6}, {1, 2}, {0, 6}, {0, 2}, {1, 7}, {1, 3}, {0, 7}, {0, 3}};
        for (size_t i = 0; i < exp.size(); ++i) {
            auto [il, pl, offset] = index.locate(i);
            CHECK(il == std::get<0>(exp[i])); CHECK(pl + offset == std::get<1>(exp[i]));
        }
    }
    {   // fmc::search<false> facade, k = 1 on 20-symbol reads
        auto text = std::vector<std::vector<uint8_t>>{std::vector<uint8_t>(400)};
        uint64_t s = 42;
        for (auto& c : text[0]) { s = s * 6364136223846793005ull + 1442695040888963407ull; c = 1 + (s >> 33) % 4; }
        auto index = fmc::BiFMIndex<5>{text, 4, 1};
        auto reads = std::vector<std::vector<uint8_t>>{};
        for (size_t p = 0; p + 20 <= 400; p += 37) { reads.emplace_back(text[0].begin() + p, text[0].begin() + p + 20); }
        reads[1][7] = reads[1][7] % 4 + 1;
        size_t found = 0;
        fmc::search<false>(index, reads, 1, [&](size_t qidx, auto cursor, size_t errors) {
            for (auto [sid, spos, offset] : fmc::LocateLinear{index, cursor})
                if (sid == 0 && spos + offset == qidx * 37) { ++found; CHECK(errors == (qidx == 1 ? 1u : 0u)); }
        });
        CHECK(found == reads.size());
    }
    {   // the same reads through FMIndex<28, Wavelet> (exact) and BiFMIndex<5, EPRV5> (k = 1): other String types, same answers
        auto text = std::vector<std::vector<uint8_t>>{std::vector<uint8_t>(600)};
        uint64_t s = 7;
        for (auto& c : text[0]) { s = s * 6364136223846793005ull + 1442695040888963407ull; c = 1 + (s >> 33) % 27; }
        auto index = fmc::FMIndex<28, fmc::string::Wavelet>{text, 4, 1};
        auto reads = std::vector<std::vector<uint8_t>>{};
        for (size_t p = 0; p + 12 <= 600; p += 49) reads.emplace_back(text[0].begin() + p, text[0].begin() + p + 12);
        size_t found = 0;
        fmc::search_no_errors::search(index, reads, [&](size_t qidx, auto cursor) {
            for (auto [sid, spos, offset] : fmc::LocateLinear{index, cursor}) if (sid == 0 && spos + offset == qidx * 49) ++found;
        });
        CHECK(found == reads.size());
        for (auto& c : text[0]) c = 1 + c % 4;
        auto bi = fmc::BiFMIndex<5, fmc::string::EPRV5>{text, 4, 1};
        reads.clear();
        for (size_t p = 0; p + 24 <= 600; p += 49) reads.emplace_back(text[0].begin() + p, text[0].begin() + p + 24);
        reads[2][5] = reads[2][5] % 4 + 1;
        found = 0;
        fmc::search<false>(bi, reads, 1, [&](size_t qidx, auto cursor, size_t errors) {
            for (auto [sid, spos, offset] : fmc::LocateLinear{bi, cursor})
                if (sid == 0 && spos + offset == qidx * 49) { ++found; CHECK(errors == (qidx == 2 ? 1u : 0u)); }
        });
        CHECK(found == reads.size());
    }
    {   // fmindex/checkBiFMIndexCursor.cpp:12-103 and checkFMIndexCursor.cpp: single cursor steps through the mirror
        auto data = std::vector<std::vector<uint8_t>>{std::vector<uint8_t>{1, 1, 1, 1, 2, 2, 2}};
        using Index = fmc::BiFMIndex<256>;
        auto index = Index{data, 1, 1};
        auto cursor = fmc::BiFMIndexCursor{index};
        CHECK(cursor.count() == index.size()); CHECK(!cursor.empty()); CHECK(cursor.lb == 0); CHECK(cursor.len == index.size());
        size_t const want_count[4] = {1, 4, 3, 0}, want_lb[4] = {0, 1, 5, 8};
        for (size_t c = 0; c < 4; ++c) {
            auto l = cursor.extendLeft(c), r = cursor.extendRight(c);
            CHECK(l.count() == want_count[c]); CHECK(l.lb == want_lb[c]);
            CHECK(r.count() == want_count[c]); CHECK(r.lb == want_lb[c]);
            CHECK(l.steps == 1); CHECK(r.steps == 1);
        }
        auto allL = cursor.extendLeft(), allR = cursor.extendRight();
        for (size_t i = 0; i < 256; ++i) {
            auto l = cursor.extendLeft(i), r = cursor.extendRight(i);
            CHECK(l.index == allL[i].index); CHECK(l.lb == allL[i].lb); CHECK(l.len == allL[i].len); CHECK(l.lbRev == allL[i].lbRev);
            CHECK(r.index == allR[i].index); CHECK(r.lb == allR[i].lb); CHECK(r.len == allR[i].len); CHECK(r.lbRev == allR[i].lbRev);
        }
        size_t seen = 0, sum = 0;
        for (auto pos : cursor) { ++seen; sum += pos; }
        CHECK(seen == 8); CHECK(sum == 28);
        // a two-step walk: "21" occurs once (text 1111222$): left then right extension meet the same interval
        auto a = cursor.extendLeft(2).extendLeft(1), b = cursor.extendRight(1).extendRight(2);
        CHECK(a.count() == 1); CHECK(b.count() == 1); CHECK(a.lb == b.lb); CHECK(a.lbRev == b.lbRev); CHECK(a.steps == 2);
        CHECK(a.symbolLeft() == 1); CHECK(a.symbolRight() == 2);                         // the symbols around the one occurrence of "12": 1 [12] 2
        // unidirectional twin (fmindex/checkFMIndexCursor.cpp:13-66)
        auto fidx = fmc::FMIndex<256>{data, 1, 1};
        auto fc = fmc::FMIndexCursor{fidx};
        for (size_t c = 0; c < 4; ++c) { auto l = fc.extendLeft(c); CHECK(l.count() == want_count[c]); CHECK(l.lb == want_lb[c]); }
        auto fall = fc.extendLeft();
        for (size_t i = 0; i < 256; ++i) { auto l = fc.extendLeft(i); CHECK(l.lb == fall[i].lb); CHECK(l.len == fall[i].len); }
        // single_locate_step (fmindex/FMIndex.h:126-128): at sampling rate 1 every row is sampled and equals locate()
        for (size_t i = 0; i < index.size(); ++i) {
            auto one = index.single_locate_step(i);
            auto [sid, spos, off] = index.locate(i);
            CHECK(one.has_value()); CHECK(off == 0);
            if (one) { CHECK(std::get<0>(*one) == sid); CHECK(std::get<1>(*one) == spos); }
        }
        auto sparse = Index{data, 4, 1};
        size_t sampled = 0;
        for (size_t i = 0; i < sparse.size(); ++i) if (sparse.single_locate_step(i)) ++sampled;
        CHECK(sampled == 2);                                                          // positions 0 and 4 of the one 8-symbol sequence (delimiter included)
    }
    {   // fmc::Search{...}() (search/search.h:48-75) against the loop it stands for
        auto index = fmc::BiFMIndex<256>{input, 1, 1};
        auto viaStruct = Results{}, viaCalls = Results{};
        auto rep = [&](size_t qidx, size_t sid, size_t pos, size_t errors) { (void)errors; viaStruct.emplace_back(qidx, sid, pos); };
        fmc::Search{index, queries, /*editDistance*/ true, /*errors*/ size_t{1}, std::optional<size_t>{}, rep}();
        fmc::search<true>(index, queries, 1, [&](auto qidx, auto cursor, auto) {
            for (auto [sid, spos, offset] : fmc::LocateLinear{index, cursor}) viaCalls.emplace_back(qidx, sid, spos + offset);
        });
        CHECK(viaStruct == viaCalls); CHECK(viaStruct.size() == 8);
        viaStruct.clear();
        fmc::Search{index, queries, /*editDistance*/ false, size_t{1}, std::optional<size_t>{3}, rep}();
        CHECK(viaStruct.size() == 6);
    }
    {   // saveIndex / loadIndex (fmindex/diskStorage.h:12-27) and the same file on several replicas of this process (here: one device listed twice)
        auto text = std::vector<std::vector<uint8_t>>{std::vector<uint8_t>(900)};
        for (size_t i = 0; i < text[0].size(); ++i) text[0][i] = static_cast<uint8_t>(1 + (i * 2654435761u >> 7) % 4);
        auto index = fmc::BiFMIndex<5>{text, 4, 1};
        std::string const file = "/tmp/fmc_gpu_test_index.fmgpu";
        fmc::saveIndex(index, file);
        auto again = fmc::loadIndex<fmc::BiFMIndex<5>>(file);
        auto reads = std::vector<std::vector<uint8_t>>{};
        for (size_t p = 0; p + 20 <= 900; p += 37) reads.emplace_back(text[0].begin() + p, text[0].begin() + p + 20);
        reads[3][7] = reads[3][7] % 4 + 1;
        auto run = [&](auto&& searchFn) { Results r; searchFn([&](size_t qidx, auto cursor) { r.emplace_back(qidx, cursor.lb, cursor.len); }); return r; };
        auto direct = run([&](auto&& d) { fmc::search_no_errors::search(index, reads, d); });
        auto loaded = run([&](auto&& d) { fmc::search_no_errors::search(again, reads, d); });
        auto replicas = fmc::loadReplicas<fmc::BiFMIndex<5>>(file, {0, 0});
        auto sharded = run([&](auto&& d) { replicas.searchNoErrors(reads, d); });
        CHECK(!direct.empty() && direct == loaded && direct == sharded && replicas.size() == 2);
        CHECK(replicas.peerCopies() == 1);                          // the file was read once: the second replica is a device-to-device copy of the first
        fmc::setOption(FMGPU_OPT_KERNEL_SELECT, FMGPU_SEL_EXACT_ONE_SYMBOL);      // library options go through the ABI: the same search on the one-symbol kernel
        auto single = run([&](auto&& d) { fmc::search_no_errors::search(index, reads, d); });
        fmc::setOption(FMGPU_OPT_KERNEL_SELECT, 0);
        CHECK(single == direct && fmc::getOption(FMGPU_OPT_KERNEL_SELECT) == 0 && fmc::getOption(FMGPU_OPT_PAIR_TABLE) == 1);
        size_t located = 0;
        replicas.searchNoErrors(reads, [&](size_t qidx, auto cursor) {
            for (auto [sid, spos, offset] : fmc::LocateLinear{replicas.front(), cursor}) if (sid == 0 && spos + offset == qidx * 37) ++located;
        });
        CHECK(located + 1 >= reads.size());
        std::remove(file.c_str());
    }
    std::printf("%s (%d failures)\n", failures ? "FAILED" : "all checks passed", failures);
    return failures ? 1 : 0;
}

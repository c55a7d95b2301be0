// The reference's `example` harness (src/example/main.cpp:21-275, argp.h:11-107, utils.h:27-105) on the GPU path:
// FASTA -> ranks, reverse complements, BiFMIndex<5, InterleavedBitvector16> at sampling rate 16, one search per k in
// [min_k, max_k] with the chosen algorithm and search-scheme generator, LocateLinear of every reported cursor, the same
// statistics line and the same `--save_output` file ("queryId seqId pos" per located row, in callback order).
//
// Same flags as the reference.  What differs:
//   * the index is built on the GPU at every start (seconds for a human genome) instead of being cached in `<fasta>.tab.dense.index`
//     (a cereal archive, not read or written here); --partialBuildUp, --threads and --ext are accepted and have nothing to switch;
//   * --algo: `ng21` (all four modes, main.cpp:176-185), `noerror` (:213-215), and `ng26` (search_ng26::search, Edit = true, over the
//     un-expanded scheme with a uniform partition) are available; the other research variants are not part of this build;
//   * --gen: backtracking, pigeon, pigeon_opt, h2-k1, h2-k2, h2-k3 (generator/all.h:35-96); `_dyn` (expandByWNC) is not available;
//   * locating is one batched call over all rows of all cursors (the rows and their order are the reference's).
#include "../../include/fmc_gpu.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <filesystem>
#include <fstream>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_set>
#include <vector>

namespace {

struct Config {                                                   // argp.h:11-36
    std::string generator = "h2-k2";
    bool generator_dyn = false;
    size_t maxQueries{};
    size_t readLength{};
    std::string saveOutput;
    size_t minK{0}, maxK{6}, k_stepSize{1};
    bool reverse{true};
    bool help{false};
    bool convertUnknownChar{false};
    std::vector<std::string> algorithms;
    std::string queryPath{};
    std::string indexPath{};
    enum class Mode { All, BestHits };
    Mode mode{Mode::All};
    size_t maxHitsPerQuery{0};
};

Config loadConfig(int argc, char const* const* argv) {            // argp.h:38-107
    Config config;
    for (int i{1}; i < argc; ++i) {
        auto a = std::string{argv[i]};
        bool more = i + 1 < argc;
        if (a == "--query" && more) config.queryPath = argv[++i];
        else if (a == "--index" && more) config.indexPath = argv[++i];
        else if (a == "--algo" && more) config.algorithms.emplace_back(argv[++i]);
        else if (a == "--ext" && more) ++i;
        else if (a == "--gen" && more) {
            config.generator = argv[++i];
            if (config.generator.size() > 4 && config.generator.substr(config.generator.size() - 4) == "_dyn") {
                config.generator = config.generator.substr(0, config.generator.size() - 4);
                config.generator_dyn = true;
            }
        }
        else if (a == "--queries" && more) config.maxQueries = static_cast<size_t>(std::stod(argv[++i]));
        else if (a == "--threads" && more) ++i;
        else if (a == "--read_length" && more) config.readLength = static_cast<size_t>(std::stod(argv[++i]));
        else if (a == "--save_output" && more) config.saveOutput = argv[++i];
        else if (a == "--min_k" && more) config.minK = static_cast<size_t>(std::stod(argv[++i]));
        else if (a == "--max_k" && more) config.maxK = static_cast<size_t>(std::stod(argv[++i]));
        else if (a == "--stepSize_k" && more) config.k_stepSize = static_cast<size_t>(std::stod(argv[++i]));
        else if (a == "--no-reverse") config.reverse = false;
        else if (a == "--help") config.help = true;
        else if (a == "--partialBuildUp") {}
        else if (a == "--convertUnknownChar") config.convertUnknownChar = true;
        else if (a == "--mode" && more) {
            auto s = std::string{argv[++i]};
            if (s == "all") config.mode = Config::Mode::All;
            else if (s == "besthits") config.mode = Config::Mode::BestHits;
            else throw std::runtime_error("invalid mode \"" + s + "\", must be any of \"all\", \"besthits\"");
        }
        else if (a == "--maxhitperquery" && more) config.maxHitsPerQuery = static_cast<size_t>(std::stod(argv[++i]));
        else throw std::runtime_error("unknown commandline " + a);
    }
    return config;
}

std::vector<uint8_t> readFile(std::string const& file) {
    auto ifs = std::ifstream{file, std::ios::binary};
    ifs.seekg(0, std::ios::end);
    auto buffer = std::vector<uint8_t>(static_cast<size_t>(ifs.tellg()));
    ifs.seekg(0, std::ios::beg);
    ifs.read(reinterpret_cast<char*>(buffer.data()), static_cast<std::streamsize>(buffer.size()));
    return buffer;
}

// utils.h:27-105: '>' lines are names, every other byte is a symbol ($ACGT -> 0..4, N -> 5 only for Sigma 6, newlines skipped, anything else
// rank 1 under --convertUnknownChar or an error); a record ends at the next '>' or at the LAST byte of the file, which is never read as
// a symbol (a file that does not end in a newline loses its last base, as in the reference).
template <size_t Sigma>
auto loadQueries(std::string const& path, bool reverse, bool convertUnknownChar) {
    std::vector<std::vector<uint8_t>> queries;
    std::vector<std::pair<std::string, bool>> queryInfos;
    if (path.empty() || !std::filesystem::exists(path)) return std::make_tuple(queries, queryInfos);
    auto b = readFile(path);
    if (b.empty() || b[0] != '>') throw std::runtime_error("can't read fasta file");
    auto ptr = b.data();
    auto const end = b.data() + b.size();
    std::vector<uint8_t> query;
    bool inName = true;
    while (ptr != end) {
        if (inName) {
            std::string name;
            if (*ptr != '>') throw std::runtime_error("expected '>'");
            ++ptr;
            if (ptr != end && *ptr == ' ') ++ptr;
            while (ptr != end && *ptr != '\n') { name += static_cast<char>(*ptr); ++ptr; }
            if (ptr != end) ++ptr;
            inName = false;
            queryInfos.emplace_back(name, false);
            if (reverse) queryInfos.emplace_back(name, true);
        } else if (*ptr == '>' || (ptr + 1) == end) {
            queries.push_back(query);
            if (reverse) {
                std::reverse(query.begin(), query.end());
                for (auto& c : query) {
                    if (c == 1) c = 4; else if (c == 2) c = 3; else if (c == 3) c = 2; else if (c == 4) c = 1;
                }
                queries.push_back(query);
            }
            query.clear();
            inName = true;
            if ((ptr + 1) == end) ++ptr;
        } else {
            auto ch = *ptr;
            if (ch == '$') query.push_back(0);
            else if (ch == 'A' || ch == 'a') query.push_back(1);
            else if (ch == 'C' || ch == 'c') query.push_back(2);
            else if (ch == 'G' || ch == 'g') query.push_back(3);
            else if (ch == 'T' || ch == 't') query.push_back(4);
            else if ((ch == 'N' || ch == 'n') && Sigma == 6) query.push_back(5);
            else if (ch == '\n') {}
            else if (convertUnknownChar) query.push_back(Sigma == 6 ? 5 : 1);
            else throw std::runtime_error("unknown alphabet");
            ++ptr;
        }
    }
    return std::make_tuple(queries, queryInfos);
}

fmc::search_scheme::Scheme generate(std::string const& name, size_t minK, size_t maxK) {   // generator/all.h:35-96, the entries this build has
    namespace g = fmc::search_scheme::generator;
    if (name == "backtracking") return g::backtracking(1, minK, maxK);
    if (name == "pigeon") return g::pigeon_trivial(minK, maxK);
    if (name == "pigeon_opt") return g::pigeon_opt(minK, maxK);
    if (name == "h2-k1") return g::h2(maxK + 1, minK, maxK);
    if (name == "h2-k2") return g::h2(maxK + 2, minK, maxK);
    if (name == "h2-k3") return g::h2(maxK + 3, minK, maxK);
    throw std::runtime_error("unknown search scheme generetaror \"" + name + "\"");
}

struct StopWatch {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    double reset() {
        auto n = std::chrono::steady_clock::now();
        double d = std::chrono::duration<double>(n - t).count();
        t = n;
        return d;
    }
};

}  // namespace

int main(int argc, char const* const* argv) try {
    constexpr size_t Sigma = 5;
    auto config = loadConfig(argc, argv);
    if (config.help) {
        std::printf("Usage:\n"
                    "./example --index somefile.fasta\n"
                    "   this will only build the index for somefile.fasta (on the GPU; nothing is written)\n"
                    "\n"
                    "./example --index somefile.fasta\\\n"
                    "          --query queryfile.fasta\\\n"
                    "          --algo [ng21, ng26, noerror]\\\n"
                    "          --gen <backtracking|pigeon|pigeon_opt|h2-k1|h2-k2|h2-k3>\\\n"
                    "          --queries <int> (maximal of number of queries)\\\n"
                    "          --read_length <int> (shorten all queries to this length)\\\n"
                    "          --save_output <file> (saves output at the end)\\\n"
                    "          --min_k <int> (minimal number of errors)\\\n"
                    "          --max_k <int> (maximal number of errors)\\\n"
                    "          --stepSize_k <int> (steps of errors)\\\n"
                    "          --no-reverse (don't use reverse compliment)\\\n"
                    "          --mode [all, besthits] (all: all hits with k errors (default), besthits: all hits with the lowest hit)\\\n"
                    "          --maxhitperquery <int> (some int, 0 = infinite hits)\n");
        return 0;
    }
    if (config.generator_dyn) throw std::runtime_error("the _dyn generators (expandByWNC) are not part of this build");
    auto const [queries, queryInfos] = loadQueries<Sigma>(config.queryPath, config.reverse, config.convertUnknownChar);
    if (!queries.empty()) {
        std::printf("loaded %zu queries (incl reverse complements)\n", queries.size());
        std::printf("%-15s: %10s  (%10s +%10s ) %10s    - results: %10s/%10s/%10s/%10s - mem: %13s\n", "name", "time_search + time_locate", "time_search",
                    "time_locate", "(time_search+time_locate)/queries.size()", "resultCt", "results.size()", "uniqueResults.size()", "readIds.size()", "memory");
    }

    std::string name = "str";                                     // visitAllStrings: the one String of the example, InterleavedBitvector16 (utils.h:262-265)
    std::printf("start loading %s ...", name.c_str());
    std::fflush(stdout);
    size_t samplingRate = 16;
    using Index = fmc::BiFMIndex<Sigma, fmc::string::InterleavedBitvector16>;
    auto index = [&]() {                                          // loadDenseIndex, utils.h:150-259 (always the build branch)
        auto [ref, refInfo] = loadQueries<Sigma>(config.indexPath, false, config.convertUnknownChar);
        if (ref.empty()) throw std::runtime_error("no sequences in --index " + config.indexPath);
        return Index{ref, samplingRate, 1};
    }();
    std::printf("done\n");

    for (auto const& algorithm : config.algorithms) {
        std::printf("using algorithm %s\n", algorithm.c_str());
        if (algorithm != "ng21" && algorithm != "ng26" && algorithm != "noerror")
            throw std::runtime_error("algorithm \"" + algorithm + "\" is not part of this build (available: ng21, ng26, noerror)");
        auto mut_queries = queries;
        if (config.maxQueries != 0) mut_queries.resize(std::min(mut_queries.size(), config.maxQueries));
        if (config.readLength != 0) for (auto& q : mut_queries) q.resize(std::min(config.readLength, q.size()));
        if (mut_queries.empty()) continue;

        for (size_t k{config.minK}; k <= config.maxK; k = k + config.k_stepSize) {
            auto len = mut_queries[0].size();
            auto oss = generate(config.generator, 0, k);
            auto search_scheme = fmc::search_scheme::expand(oss, len);
            auto search_schemes = std::vector<fmc::search_scheme::Scheme>{};
            for (size_t j{0}; j <= k; ++j) search_schemes.emplace_back(fmc::search_scheme::expand(generate(config.generator, j, j), len));

            size_t resultCt{};
            StopWatch sw;
            auto results = std::vector<std::tuple<size_t, size_t, size_t, size_t>>{};
            auto resultCursors = std::vector<std::tuple<size_t, fmc::BiFMIndexCursor<Index>, size_t>>{};
            auto res_cb = [&](size_t queryId, auto cursor, size_t errors) { resultCursors.emplace_back(queryId, cursor, errors); };

            if (algorithm == "ng21") {                            // main.cpp:176-185
                if (config.mode == Config::Mode::All) {
                    if (config.maxHitsPerQuery == 0) fmc::search_ng21::search(index, mut_queries, search_scheme, res_cb);
                    else fmc::search_ng21::search_n(index, mut_queries, search_scheme, config.maxHitsPerQuery, res_cb);
                } else {
                    if (config.maxHitsPerQuery == 0) fmc::search_ng21::search_best(index, mut_queries, search_schemes, res_cb);
                    else fmc::search_ng21::search_best_n(index, mut_queries, search_schemes, config.maxHitsPerQuery, res_cb);
                }
            } else if (algorithm == "ng26") {
                auto n = config.maxHitsPerQuery == 0 ? std::numeric_limits<size_t>::max() : config.maxHitsPerQuery;
                if (config.mode == Config::Mode::All) fmc::search_ng26::search<true>(index, mut_queries, oss, {}, res_cb, n);
                else {
                    auto list = std::vector<std::tuple<fmc::search_scheme::Scheme, std::vector<size_t>>>{};
                    for (size_t j{0}; j <= k; ++j) list.emplace_back(generate(config.generator, j, j), std::vector<size_t>{});
                    fmc::search_ng26::search_best<true>(index, mut_queries, list, res_cb, n);
                }
            } else {                                              // noerror, main.cpp:213-215
                fmc::search_no_errors::search(index, mut_queries, [&](size_t queryId, auto cursor) { res_cb(queryId, cursor, 0); });
            }
            auto time_search = sw.reset();

            {   // main.cpp:236-243: LocateLinear over every cursor, here as one batched locate of all their rows
                std::vector<uint64_t> rows;
                for (auto const& [queryId, cursor, e] : resultCursors) {
                    for (size_t r = 0; r < cursor.len; ++r) rows.push_back(cursor.lb + r);
                    resultCt += cursor.len;
                }
                auto located = index.locate(rows);
                results.reserve(rows.size());
                size_t at = 0;
                for (auto const& [queryId, cursor, e] : resultCursors)
                    for (size_t r = 0; r < cursor.len; ++r, ++at) {
                        auto [seqId, pos, offset] = located[at];
                        results.emplace_back(queryId, seqId, pos + offset, e);
                    }
            }
            auto time_locate = sw.reset();

            auto uniqueResults = results;
            std::sort(uniqueResults.begin(), uniqueResults.end());
            uniqueResults.erase(std::unique(uniqueResults.begin(), uniqueResults.end()), uniqueResults.end());
            std::unordered_set<size_t> readIds;
            for (auto const& [queryId, cursor, e] : resultCursors) {   // main.cpp:251-258
                if (queryId > mut_queries.size() / 2) readIds.insert(queryId - mut_queries.size() / 2);
                else readIds.insert(queryId);
            }
            std::printf("%-15s %3zu: %10.3gs (%10.3gs+%10.3gs) %10.3gq/s - results: %10zu/%10zu/%10zu/%10zu - mem: %13zu\n", name.c_str(), k,
                        time_search + time_locate, time_search, time_locate, mut_queries.size() / (time_search + time_locate), resultCt, results.size(),
                        uniqueResults.size(), readIds.size(), size_t{0});
            if (!config.saveOutput.empty()) {
                auto ofs = std::fopen(config.saveOutput.c_str(), "w");
                if (!ofs) throw std::runtime_error("cannot write " + config.saveOutput);
                for (auto const& [queryId, seqId, pos, e] : results) std::fprintf(ofs, "%zu %zu %zu\n", queryId, seqId, pos);
                std::fclose(ofs);
            }
        }
    }
    return 0;
} catch (std::exception const& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
}

// The reference's `example` harness (src/example/main.cpp, with its command line and FASTA reader) on the GPU path — same behaviour, own code:
// FASTA -> ranks, reverse complements, BiFMIndex<5, InterleavedBitvector16> at sampling rate 16, one search per k in
// [min_k, max_k] with the chosen algorithm and search-scheme generator, LocateLinear of every reported cursor, the same
// statistics line and the same `--save_output` file ("queryId seqId pos" per located row, in callback order).
//
// Same flags as the reference.  What differs:
//   * the index is built on the GPU at every start (seconds for a human genome) instead of being cached in `<fasta>.tab.dense.index`
//     (a cereal archive, not read or written here); --partialBuildUp, --threads and --ext are accepted and have nothing to switch;
//   * --algo: `ng21` (all four modes, main.cpp:176-185), `noerror` (:213-215), and `ng26` (search_ng26::search, Edit = true, over the
//     un-expanded scheme with a uniform partition) are available; the other research variants are not part of this build;
//   * --gen: backtracking, pigeon, pigeon_opt, h2-k1, h2-k2, h2-k3 (generator/all.h:35-96); `<name>_dyn` stretches the scheme to the read length by expandByWNC (Edit = true, sigma 4, 3e9 rows: main.cpp:116) instead of uniformly;
//   * locating is one batched call over all rows of all cursors (the rows and their order are the reference's).
#include "../../include/fmc_gpu.hpp"

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <filesystem>
#include <fstream>
#include <iterator>
#include <cstdlib>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_set>
#include <vector>

namespace {

// ---- command line --------------------------------------------------------------------------------------------------------------
// What the reference's example accepts (its argp.h), described as a table: flag, whether a value follows, what the value does.  Numbers are
// read as floating point and truncated, so "--queries 1e5" works as it does there; a flag that needs a value but is the last token, and any
// token that is no flag, is reported as "unknown commandline <token>".
enum class HitMode { all, besthits };
struct Options {
    std::string referenceFasta, readsFasta, outputFile;
    std::vector<std::string> algorithms;
    std::string schemeName = "h2-k2";
    bool schemeDyn = false;
    size_t firstK = 0, lastK = 6, stepK = 1;
    size_t readLimit = 0, trimTo = 0, hitsPerRead = 0;          // 0 = no limit
    bool withReverseComplement = true, unknownToA = false, wantHelp = false;
    HitMode hitMode = HitMode::all;
};

size_t asCount(char const* text) { return static_cast<size_t>(std::strtod(text, nullptr)); }

struct FlagRule {
    char const* name;
    bool takesValue;
    void (*apply)(Options&, char const*);
};
FlagRule const kFlags[] = {
    {"--index", true, [](Options& o, char const* v) { o.referenceFasta = v; }},
    {"--query", true, [](Options& o, char const* v) { o.readsFasta = v; }},
    {"--save_output", true, [](Options& o, char const* v) { o.outputFile = v; }},
    {"--algo", true, [](Options& o, char const* v) { o.algorithms.emplace_back(v); }},
    {"--gen", true, [](Options& o, char const* v) {
         o.schemeName = v;
         static std::string const suffix = "_dyn";
         if (o.schemeName.size() > suffix.size() && o.schemeName.compare(o.schemeName.size() - suffix.size(), suffix.size(), suffix) == 0) {
             o.schemeName.erase(o.schemeName.size() - suffix.size());
             o.schemeDyn = true;
         }
     }},
    {"--min_k", true, [](Options& o, char const* v) { o.firstK = asCount(v); }},
    {"--max_k", true, [](Options& o, char const* v) { o.lastK = asCount(v); }},
    {"--stepSize_k", true, [](Options& o, char const* v) { o.stepK = asCount(v); }},
    {"--queries", true, [](Options& o, char const* v) { o.readLimit = asCount(v); }},
    {"--read_length", true, [](Options& o, char const* v) { o.trimTo = asCount(v); }},
    {"--maxhitperquery", true, [](Options& o, char const* v) { o.hitsPerRead = asCount(v); }},
    {"--mode", true, [](Options& o, char const* v) {
         std::string const m = v;
         if (m == "all") o.hitMode = HitMode::all;
         else if (m == "besthits") o.hitMode = HitMode::besthits;
         else throw std::runtime_error("invalid mode \"" + m + "\", must be any of \"all\", \"besthits\"");
     }},
    {"--no-reverse", false, [](Options& o, char const*) { o.withReverseComplement = false; }},
    {"--convertUnknownChar", false, [](Options& o, char const*) { o.unknownToA = true; }},
    {"--help", false, [](Options& o, char const*) { o.wantHelp = true; }},
    // accepted for compatibility; nothing to switch in this build (no index cache file, no host threads, one String type)
    {"--ext", true, [](Options&, char const*) {}},
    {"--threads", true, [](Options&, char const*) {}},
    {"--partialBuildUp", false, [](Options&, char const*) {}},
};

Options parseCommandLine(int argc, char const* const* argv) {
    Options o;
    for (int at = 1; at < argc; ++at) {
        std::string const token = argv[at];
        FlagRule const* rule = nullptr;
        for (auto const& f : kFlags) if (token == f.name) { rule = &f; break; }
        if (!rule || (rule->takesValue && at + 1 >= argc)) throw std::runtime_error("unknown commandline " + token);
        rule->apply(o, rule->takesValue ? argv[++at] : nullptr);
    }
    return o;
}

// ---- FASTA ------------------------------------------------------------------------------------------------------------------------
// The reader the example uses, by its observable rules: the file must begin with '>'; a name line runs to the next newline; every other
// byte of a record is a symbol — $ A C G T (either case) are ranks 0..4, N is rank 5 when the alphabet has six symbols, newlines are
// dropped, any other byte is rank 1 (5 with six symbols) under --convertUnknownChar and an "unknown alphabet" error without it; a '>'
// anywhere in the sequence part starts the next record; the very last byte of the file ends the last record and is never a symbol itself
// (a file whose last line has no newline loses its last base); a name line that runs to the end of the file yields no record.
enum ByteClass : uint8_t { kSymbol0 = 0, /* 1..5: that rank */ kSkip = 6, kOther = 7 };
std::array<uint8_t, 256> byteClasses(size_t sigma) {
    std::array<uint8_t, 256> t;
    t.fill(kOther);
    t[static_cast<unsigned char>('\n')] = kSkip;
    t[static_cast<unsigned char>('$')] = 0;
    char const* letters = "ACGT";
    for (uint8_t r = 0; r < 4; ++r) {
        t[static_cast<unsigned char>(letters[r])] = static_cast<uint8_t>(r + 1);
        t[static_cast<unsigned char>(letters[r] - 'A' + 'a')] = static_cast<uint8_t>(r + 1);
    }
    if (sigma == 6) t[static_cast<unsigned char>('N')] = t[static_cast<unsigned char>('n')] = 5;
    return t;
}

std::vector<std::vector<uint8_t>> readFasta(std::string const& path, size_t sigma, bool unknownToA) {
    std::vector<std::vector<uint8_t>> records;
    if (path.empty() || !std::filesystem::exists(path)) return records;
    std::ifstream in{path, std::ios::binary};
    std::vector<char> bytes{std::istreambuf_iterator<char>{in}, std::istreambuf_iterator<char>{}};
    if (bytes.empty() || bytes[0] != '>') throw std::runtime_error("can't read fasta file");
    auto const classes = byteClasses(sigma);
    uint8_t const fallback = sigma == 6 ? 5 : 1;
    size_t const size = bytes.size();
    size_t at = 0;
    while (at < size) {                                            // `at` stands on the '>' of a name line
        while (at < size && bytes[at] != '\n') ++at;
        ++at;                                                      // first byte of the sequence part
        if (at >= size) break;
        std::vector<uint8_t> ranks;
        for (; at + 1 < size && bytes[at] != '>'; ++at) {
            uint8_t const cls = classes[static_cast<unsigned char>(bytes[at])];
            if (cls < kSkip) ranks.push_back(cls);
            else if (cls == kOther) {
                if (!unknownToA) throw std::runtime_error("unknown alphabet");
                ranks.push_back(fallback);
            }
        }
        records.push_back(std::move(ranks));
        if (at + 1 >= size) break;                                 // stopped on the last byte of the file
    }
    return records;
}

// every read followed by its reverse complement (A <-> T, C <-> G; other ranks stay)
std::vector<std::vector<uint8_t>> withReverseComplements(std::vector<std::vector<uint8_t>> const& reads) {
    static uint8_t const complement[6] = {0, 4, 3, 2, 1, 5};
    std::vector<std::vector<uint8_t>> both;
    both.reserve(2 * reads.size());
    for (auto const& r : reads) {
        both.push_back(r);
        std::vector<uint8_t> rc(r.rbegin(), r.rend());
        for (auto& c : rc) if (c < 6) c = complement[c];
        both.push_back(std::move(rc));
    }
    return both;
}

// ---- search-scheme generators by name (the ones this build has; the reference's list is search_scheme/generator/all.h:35-96) ----------------
fmc::search_scheme::Scheme schemeByName(std::string const& name, size_t minErrors, size_t maxErrors) {
    namespace g = fmc::search_scheme::generator;
    struct Entry { char const* name; fmc::search_scheme::Scheme (*make)(size_t, size_t); };
    static Entry const table[] = {
        {"backtracking", [](size_t lo, size_t hi) { return g::backtracking(1, lo, hi); }},
        {"pigeon", [](size_t lo, size_t hi) { return g::pigeon_trivial(lo, hi); }},
        {"pigeon_opt", [](size_t lo, size_t hi) { return g::pigeon_opt(lo, hi); }},
        {"h2-k1", [](size_t lo, size_t hi) { return g::h2(hi + 1, lo, hi); }},
        {"h2-k2", [](size_t lo, size_t hi) { return g::h2(hi + 2, lo, hi); }},
        {"h2-k3", [](size_t lo, size_t hi) { return g::h2(hi + 3, lo, hi); }},
    };
    for (auto const& e : table) if (name == e.name) return e.make(minErrors, maxErrors);
    std::string known;
    for (auto const& e : table) known += std::string(known.empty() ? "" : ", ") + e.name;
    throw std::runtime_error("unknown search scheme generator \"" + name + "\" (this build has: " + known + ")");
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

// ---- one run of the harness ---------------------------------------------------------------------------------------------------------
// What the reference's main loop does for one (algorithm, k) pair, in this order: search (every reported cursor is a Hit), locate every row of
// every cursor (one batched device call here; the rows and their order are those of the per-cursor LocateLinear loops), count, print the
// statistics line, write the optional output file.
constexpr size_t Sigma = 5;                                      // $ A C G T (src/example/utils.h:81-85)
using Index = fmc::BiFMIndex<Sigma, fmc::string::InterleavedBitvector16>;
enum class Algorithm { ng21, ng26, noerror };

Algorithm algorithmByName(std::string const& name) {
    if (name == "ng21") return Algorithm::ng21;
    if (name == "ng26") return Algorithm::ng26;
    if (name == "noerror") return Algorithm::noerror;
    throw std::runtime_error("algorithm \"" + name + "\" is not part of this build (available: ng21, ng26, noerror)");
}

Index openIndex(Options const& config) {
    std::printf("start loading %s ...", "str");                  // (the one String of the example: InterleavedBitvector16)
    std::fflush(stdout);
    auto reference = readFasta(config.referenceFasta, Sigma, config.unknownToA);
    if (reference.empty()) throw std::runtime_error("no sequences in --index " + config.referenceFasta);
    Index index{reference, /*samplingRate*/ 16, /*threads*/ 1};   // built on the GPU at every start (seconds for a human genome)
    std::printf("done\n");
    return index;
}

struct Hit { size_t read; uint64_t firstRow, rows; size_t errors; };
struct Placement {
    size_t read, sequence, position, errors;
    bool operator<(Placement const& o) const { return std::tie(read, sequence, position, errors) < std::tie(o.read, o.sequence, o.position, o.errors); }
    bool operator==(Placement const& o) const { return std::tie(read, sequence, position, errors) == std::tie(o.read, o.sequence, o.position, o.errors); }
};

struct Run {
    Options const& config;
    Index index;

    // a scheme stretched to the read length: uniformly, or (--gen <name>_dyn) part by part where the weighted node count grows least — with the constants
    // the reference's example passes (src/example/main.cpp:116, :135: Edit = true, sigma 4, 3e9 rows)
    fmc::search_scheme::Scheme stretch(fmc::search_scheme::Scheme const& scheme, size_t len) const {
        return config.schemeDyn ? fmc::search_scheme::expandByWNC<true>(scheme, len, 4, 3'000'000'000) : fmc::search_scheme::expand(scheme, len);
    }

    std::vector<Hit> search(Algorithm kind, size_t k, std::vector<std::vector<uint8_t>> const& reads) const {
        std::vector<Hit> hits;
        auto collect = [&](size_t read, auto const& cursor, size_t errors) { hits.push_back({read, cursor.lb, cursor.len, errors}); };
        size_t const perRead = config.hitsPerRead == 0 ? std::numeric_limits<size_t>::max() : config.hitsPerRead;
        auto schemesUpTo = [&](auto&& make) {                      // best-hit modes try 0, 1, .. k errors in turn
            using T = decltype(make(size_t{}));
            std::vector<T> list;
            for (size_t j = 0; j <= k; ++j) list.push_back(make(j));
            return list;
        };
        switch (kind) {
        case Algorithm::noerror:
            fmc::search_no_errors::search(index, reads, [&](size_t read, auto const& cursor) { collect(read, cursor, 0); });
            break;
        case Algorithm::ng26: {
            if (config.hitMode == HitMode::all) fmc::search_ng26::search<true>(index, reads, schemeByName(config.schemeName, 0, k), {}, collect, perRead);
            else fmc::search_ng26::search_best<true>(index, reads, schemesUpTo([&](size_t j) {
                     return std::tuple<fmc::search_scheme::Scheme, std::vector<size_t>>{schemeByName(config.schemeName, j, j), {}}; }), collect, perRead);
            break;
        }
        case Algorithm::ng21: {
            size_t const len = reads[0].size();
            if (config.hitMode == HitMode::all) {
                auto const expanded = stretch(schemeByName(config.schemeName, 0, k), len);
                if (config.hitsPerRead == 0) fmc::search_ng21::search(index, reads, expanded, collect);
                else fmc::search_ng21::search_n(index, reads, expanded, config.hitsPerRead, collect);
            } else {
                auto const ladder = schemesUpTo([&](size_t j) { return stretch(schemeByName(config.schemeName, j, j), len); });
                if (config.hitsPerRead == 0) fmc::search_ng21::search_best(index, reads, ladder, collect);
                else fmc::search_ng21::search_best_n(index, reads, ladder, config.hitsPerRead, collect);
            }
            break;
        }
        }
        return hits;
    }

    std::vector<Placement> locate(std::vector<Hit> const& hits) const {
        std::vector<uint64_t> rows;
        for (auto const& h : hits) for (uint64_t r = 0; r < h.rows; ++r) rows.push_back(h.firstRow + r);
        auto const where = index.locate(rows);
        std::vector<Placement> placed;
        placed.reserve(rows.size());
        size_t at = 0;
        for (auto const& h : hits)
            for (uint64_t r = 0; r < h.rows; ++r, ++at) {
                auto const& [sequence, sampled, walked] = where[at];
                placed.push_back({h.read, sequence, sampled + walked, h.errors});
            }
        return placed;
    }

    void oneErrorBudget(Algorithm kind, size_t k, std::vector<std::vector<uint8_t>> const& reads) const {
        StopWatch clock;
        auto const hits = search(kind, k, reads);
        double const tSearch = clock.reset();
        auto const placed = locate(hits);
        double const tLocate = clock.reset();

        auto distinct = placed;
        std::sort(distinct.begin(), distinct.end());
        distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
        std::unordered_set<size_t> readsWithHits;                  // a read and its reverse complement (second half of the batch) count once
        for (auto const& h : hits) readsWithHits.insert(h.read > reads.size() / 2 ? h.read - reads.size() / 2 : h.read);
        std::printf("%-15s %3zu: %10.3gs (%10.3gs+%10.3gs) %10.3gq/s - results: %10zu/%10zu/%10zu/%10zu - mem: %13zu\n", "str", k, tSearch + tLocate, tSearch, tLocate,
                    reads.size() / (tSearch + tLocate), placed.size(), placed.size(), distinct.size(), readsWithHits.size(), size_t{0});
        if (config.outputFile.empty()) return;
        auto* out = std::fopen(config.outputFile.c_str(), "w");
        if (!out) throw std::runtime_error("cannot write " + config.outputFile);
        for (auto const& p : placed) std::fprintf(out, "%zu %zu %zu\n", p.read, p.sequence, p.position);
        std::fclose(out);
    }
};

}  // namespace

int main(int argc, char const* const* argv) try {
    auto config = parseCommandLine(argc, argv);
    if (config.wantHelp) {
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
    auto reads = readFasta(config.readsFasta, Sigma, config.unknownToA);
    if (config.withReverseComplement) reads = withReverseComplements(reads);
    if (!reads.empty()) {
        std::printf("loaded %zu queries (incl reverse complements)\n", reads.size());
        std::printf("%-15s: %10s  (%10s +%10s ) %10s    - results: %10s/%10s/%10s/%10s - mem: %13s\n", "name", "time_search + time_locate", "time_search",
                    "time_locate", "(time_search+time_locate)/queries.size()", "resultCt", "results.size()", "uniqueResults.size()", "readIds.size()", "memory");
    }
    Run run{config, openIndex(config)};
    if (config.readLimit != 0 && reads.size() > config.readLimit) reads.resize(config.readLimit);
    if (config.trimTo != 0) for (auto& r : reads) if (r.size() > config.trimTo) r.resize(config.trimTo);
    for (auto const& algorithm : config.algorithms) {
        std::printf("using algorithm %s\n", algorithm.c_str());
        auto const kind = algorithmByName(algorithm);
        if (reads.empty()) continue;
        for (size_t k = config.firstK; k <= config.lastK; k += config.stepK) run.oneErrorBudget(kind, k, reads);
    }
    return 0;
} catch (std::exception const& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
}

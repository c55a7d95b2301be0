#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run in the build container, where /root/reference exists).

  reference_tests.json   DATA of the reference's own tests for this path — literal input vectors and expected values
                         picked out of src/test_fmindex-collection/*.cpp (file:line recorded per entry).  No source
                         text is kept, only numbers.
  ref_strings.json       vectors produced by the REAL reference headers (oracle/_ref/libfmref.so): rank / prefix_rank /
                         symbol tables and layout checksums of the occurrence-table types on seeded texts.
  ref_search_intervals.npz  (lb, len) of exact searches — BASELINE configs[0] (1 MB of DNA, 10k x 31 bp, InterleavedBitvector16), a small
                         FMIndex<28, Wavelet> case and EPR / EPRV2 / EPRV5 / Prefix16 cases — computed with the REAL reference's rank.
  ref_schemes.json       search-scheme tables produced by the real reference (h2, pigeon_opt, backtracking, expand,
                         limitToHamming, isValid, isComplete, nodeCount, createUniformPartition).

    python tests/golden/make_golden.py [--tests-only]
"""
import json
import os
import re
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
TESTS = "/root/reference/src/test_fmindex-collection"


# ----------------------------------------------------------------------------------------------- reference test data
def _tok(t):
    t = t.strip()
    if not t:
        return None
    if t.startswith("'"):
        body = t[1:-1]
        if body == "\\0":
            return 0
        if body.startswith("\\"):
            return {"n": 10, "t": 9, "\\": 92, "'": 39}[body[1]]
        return ord(body)
    t = re.sub(r"(ull|ul|u|l)$", "", t)
    return int(t, 0)


def _list_after(src, marker, start=0):
    """numbers of the first brace-enclosed initialiser list after `marker`"""
    i = src.index(marker, start)
    i = src.index("{", i + len(marker) - 1) if not marker.endswith("{") else i + len(marker) - 1
    depth, j = 0, i
    while True:
        if src[j] == "{":
            depth += 1
        elif src[j] == "}":
            depth -= 1
            if depth == 0:
                break
        j += 1
    body = src[i + 1: j]
    # split on commas that are not inside a char literal
    toks = re.findall(r"'(?:\\.|[^'])'|[^,\s{}]+", body)
    return [v for v in (_tok(t) for t in toks) if v is not None], j


def _line_of(src, pos):
    return src.count("\n", 0, pos) + 1


def reference_tests():
    out = {"_provenance": "values transcribed by tests/golden/make_golden.py from the reference's test sources (data only)"}
    # --- string/unittest.cpp: hand counted "Hallo Welt" (sigma 255) and the 310-symbol text
    src = open(f"{TESTS}/string/unittest.cpp").read()
    text, _ = _list_after(src, "auto text = std::vector<uint8_t>{'H', 'a', 'l', 'l', 'o', ' ', 'W', 'e', 'l', 't'};".split("{")[0] + "{")
    rank, prank = [], []
    for m in re.finditer(r"CHECK\(vec\.(rank|prefix_rank)\(\s*(\d+),\s*('(?:\\.|[^'])')\)\s*==\s*(\d+)\);", src):
        (rank if m.group(1) == "rank" else prank).append([int(m.group(2)), _tok(m.group(3)), int(m.group(4))])
    out["hallo_welt"] = {"source": "string/unittest.cpp:52-312", "sigma": 255, "text": text[:10], "rank": rank, "prefix_rank": prank}
    i = src.index("check symbol vectors construction on text longer than 255 characters")
    long_text, _ = _list_after(src, "auto text = std::vector<uint8_t>{", i)
    out["long_text"] = {"source": "string/unittest.cpp:314-398", "sigma": 255, "text": long_text,
                        "expect": "rank/prefix_rank/symbol/all_ranks_and_prefix_ranks equal naive counts for every idx and symbol"}
    # --- fmindex/checkFMIndex.cpp, checkBiFMIndex.cpp: literal BWT / SA
    src = open(f"{TESTS}/fmindex/checkFMIndex.cpp").read()
    bwt, _ = _list_after(src, "auto bwt    = std::vector<uint8_t>{")
    sa, _ = _list_after(src, "auto sa     = std::vector<uint64_t>{")
    out["fmindex_hallo"] = {"source": "fmindex/checkFMIndex.cpp:15-110", "sigma": 255, "bwt": bwt, "sa": sa,
                            "expect": "locate(i): seqId == 0 and pos + offset == sa[i] for full, every-2nd-row(+sa==0), odd-row and every-2nd-text sampling"}
    src = open(f"{TESTS}/fmindex/checkBiFMIndex.cpp").read()
    bwt, p = _list_after(src, "auto bwt    = std::vector<uint8_t>{")
    bwtr, p = _list_after(src, "auto bwtRev = std::vector<uint8_t>{", p)
    sa, p = _list_after(src, "auto sa     = std::vector<uint64_t>{", p)
    out["bifmindex_hallo"] = {"source": "fmindex/checkBiFMIndex.cpp:13-105", "sigma": 255, "bwt": bwt, "bwtRev": bwtr, "sa": sa}
    j = src.index("checking bidirectional fm index on longer text")
    bwt, p = _list_after(src, "auto bwt    = std::vector<uint8_t>{", j)
    bwtr, p = _list_after(src, "auto bwtRev = std::vector<uint8_t>{", p)
    sa, p = _list_after(src, "auto sa     = std::vector<uint64_t>{", p)
    out["bifmindex_long"] = {"source": "fmindex/checkBiFMIndex.cpp:136-222", "sigma": 255, "bwt": bwt, "bwtRev": bwtr, "sa": sa}
    # --- cursors
    out["cursor"] = {"source": "fmindex/checkFMIndexCursor.cpp:13-66, fmindex/checkBiFMIndexCursor.cpp:12-103", "sigma": 256,
                     "data": [[1, 1, 1, 1, 2, 2, 2]], "sampling_rate": 1,
                     "extend": [[0, 1, 0], [1, 4, 1], [2, 3, 5], [3, 0, 8]],     # symbol, count, lb (left and right)
                     "rows": [0, 1, 2, 3, 4, 5, 6, 7]}
    # --- searches
    src = open(f"{TESTS}/search/checkSearches.cpp").read()
    A, B, Cc, D = ord("A"), ord("B"), ord("C"), ord("D")
    inp = [[A, A, A, Cc, A, A, A, B, A, A, A], [A, A, A, B, A, A, A, Cc, A, A, A]]
    i = src.index('SECTION("backtracking, all search")')
    exp, _ = _list_after(src, "auto expected = std::vector<std::tuple<size_t, size_t, size_t>> {", i)
    exp8 = [exp[k: k + 3] for k in range(0, len(exp), 3)]
    i = src.index('SECTION("search, hamming distance, all search, no search scheme")')
    exp, _ = _list_after(src, "auto expected = std::vector<std::tuple<size_t, size_t, size_t>> {", i)
    exph = [exp[k: k + 3] for k in range(0, len(exp), 3)]
    i = src.index('SECTION("search no errors, all search")')
    exp, _ = _list_after(src, "auto expected = std::vector<std::tuple<size_t, size_t, size_t>> {", i)
    def expected_of(section, start=0):
        k = src.index('SECTION("%s")' % section, start)
        e, _ = _list_after(src, "auto expected = std::vector<std::tuple<size_t, size_t, size_t>> {", k)
        return [e[j: j + 3] for j in range(0, len(e), 3)]
    live = src.index("#endif", src.index('SECTION("search ng26, all search")'))          # the second "no search scheme" sections are the compiled ones
    out["searches"] = {"source": "search/checkSearches.cpp:14-72, :104-117, :1173-1199, :1482-1505", "sigma": 256, "sampling_rate": 1,
                       "input": inp, "queries": [[Cc, Cc], [B, B]],
                       "backtracking_k1": exp8, "hamming_k1_facade": exph, "no_errors": [exp[k: k + 3] for k in range(0, len(exp), 3)]}
    # --- edit distance (Edit = true is search_ng26::search's default template argument)
    out["searches_edit"] = {"source": "search/checkSearches.cpp:1093-1121, :1148-1171, :1422-1444, :1446-1466", "sigma": 256, "sampling_rate": 1, "input": inp,
                            "ng26_pigeon_opt_CD_DB": {"queries": [[Cc, D], [D, B]], "expected": expected_of("search ng26, all search")},
                            "ng26_pigeon_opt_n3": {"queries": [[Cc, Cc], [B, B]], "n": 3, "expected": expected_of("search ng26, all search_n")},
                            "facade_k1": {"queries": [[Cc, Cc], [B, B]], "expected": expected_of("search, all search, no search scheme", live)},
                            "facade_k1_n3": {"queries": [[Cc, Cc], [B, B]], "n": 3, "expected": expected_of("search, all search_n, no search scheme", live)}}
    # --- search_ng21 (always edit distance) over expanded pigeon_opt schemes
    out["searches_ng21"] = {"source": "search/checkSearches.cpp:422-525", "sigma": 256, "sampling_rate": 1, "input": inp, "queries": [[Cc, Cc], [B, B]],
                            "search": {"scheme": [0, 1], "expected": expected_of("search ng21, all search")},
                            "search_n": {"scheme": [0, 1], "n": 3, "expected": expected_of("search ng21, all search_n")},
                            "search_best": {"schemes": [[0, 0], [1, 1], [2, 2]], "expected": expected_of("search ng21, all search_best")},
                            "search_best_n": {"schemes": [[0, 0], [1, 1]], "n": 3, "expected": expected_of("search ng21, all search_best_n")}}
    src = open(f"{TESTS}/search/checkSearchBacktracking.cpp").read()
    i = src.index("searching with collection and backtracking")
    bexp, p = _list_after(src, "auto expected = std::vector<uint8_t>{", i)
    j = src.index("auto expected = std::vector<std::tuple<uint32_t, uint32_t>> {", p)
    nums = [int(x) for x in re.findall(r"make_tuple\((\d+)ull,\s*(\d+)ull\)", src[j: src.index("};", j)]) for x in x]
    out["collection"] = {"source": "search/checkSearchBacktracking.cpp:42-102", "sigma": 255, "sampling_rate": 1,
                         "input": [[A, A, A, Cc, A, A, A, Cc, A, A, A], [A, A, A, B, A, A, A, B, A, A, A]],
                         "bwt": bexp, "query_A": {"lb": 2, "count": 18},
                         "locate": [nums[k: k + 2] for k in range(0, len(nums), 2)]}
    out["single"] = {"source": "search/checkSearchBacktracking.cpp:12-40", "sigma": 255, "sampling_rate": 1,
                     "input": [[A, A, A, Cc, A, A, A, Cc, A, A, A]],
                     "bwt": [A, A, A, Cc, Cc, 0, A, A, A, A, A, A], "query_A": {"lb": 1, "count": 9}}
    out["fmindex_backtracking"] = {"source": "search/checkSearchBacktracking.cpp:295-327", "sigma": 256, "sampling_rate": 1,
                                   "input": inp, "queries": [[Cc, Cc], [B, B]], "k": 1, "expected": exp8}
    # --- search_scheme/expand.cpp
    out["expand"] = {"source": "search_scheme/expand.cpp:11-60", "cases": [
        {"in": [[0, 1], [0, 0], [0, 0]], "len": 10, "out": [list(range(10)), [0] * 10, [0] * 10]},
        {"in": [[0, 1], [0, 0], [0, 1]], "len": 4, "out": [[0, 1, 2, 3], [0, 0, 0, 0], [0, 0, 1, 1]]},
        {"in": [[0, 1], [0, 0], [0, 1]], "len": 3, "out": [[0, 1, 2], [0, 0, 0], [0, 0, 1]]}]}
    out["h2_complete"] = {"source": "search_scheme/checkGeneratorsIsComplete.cpp:48-60",
                          "expect": "isComplete(h2(N, minK, maxK), minK, maxK) for N in 1..9, minK <= maxK < min(N, 5)"}
    # --- DenseVector / DenseMultiVector (the value arrays of the sampled suffix array): inputs, how each vector is made, the divisor the tests assert
    def dense_cases(path, lines):
        src = open(path).read()
        cases = []
        for m in re.finditer(r'TEST_CASE\("([^"]+)"', src):
            body = src[m.end(): (src.index("TEST_CASE(", m.end()) if "TEST_CASE(" in src[m.end():] else len(src))]
            vecs = [[_tok(t) for t in re.findall(r"[^,\s{}]+", v)] for v in re.findall(r"std::vector<uint64_t>\{([^}]*)\}", body)]
            case = {"name": m.group(1), "line": _line_of(src, m.start()), "inputs": vecs, "concat": "::concat(" in body}
            ld = re.search(r"DenseVector\(/\*\.largestValue =\*/\s*(\d+),\s*/\*\.commonDivisor =\s*\*/\s*(\d+)\)", body)
            if ld:
                case["largest_divisor"] = [int(ld.group(1)), int(ld.group(2))]
            cd = re.search(r"commonDivisor == (\d+)", body)
            if cd:
                case["common_divisor"] = int(cd.group(1))
            cases.append(case)
        return {"source": lines, "cases": cases, "expect": "vec[i] == input[i] for every i (concat: the left inputs followed by the right ones)"}
    out["dense_vector"] = dense_cases(f"{TESTS}/checkDenseVector.cpp", "checkDenseVector.cpp:8-82")
    out["dense_multi_vector"] = dense_cases(f"{TESTS}/checkDenseMultiVector.cpp", "checkDenseMultiVector.cpp:8-89")
    # --- bitvector/unittest.cpp (runs over Bitvector2L_512_64k, bitvector/allBitVectors.h:45 — the presence bitvector of the sampled suffix array)
    src = open(f"{TESTS}/bitvector/unittest.cpp").read()
    short, p = _list_after(src, "auto text = std::vector<uint8_t>{")
    sym = [[int(a), int(b)] for a, b in re.findall(r"CHECK\(vec\.symbol\(\s*(\d+)\) == (\d+)\);", src[p: src.index('SECTION("longer text")')])]
    rk = [[int(a), int(b)] for a, b in re.findall(r"CHECK\(vec\.rank\(\s*(\d+)\) == (\d+)\);", src[p: src.index('SECTION("longer text")')])]
    j = src.index('SECTION("longer text")')
    long_bits, p2 = _list_after(src, "auto text = std::vector<uint8_t>{", j)
    loop = re.search(r"for \(size_t i\{0\}; i < (\d+); i \+= (\d+)\)", src[p2:])
    per = [[int(a), int(b)] for a, b in re.findall(r"CHECK\(vec\.rank\(i \+\s*(\d+)\) == (\d+) \+ i/16\*8\);", src[p2:])][:17]
    long_rank = [[i + a, b + i // 16 * 8] for i in range(0, int(loop.group(1)), int(loop.group(2))) for a, b in per]
    out["bitvector"] = {"source": "bitvector/unittest.cpp:14-140", "short": {"bits": short, "symbol": sym, "rank": rk},
                        "long": {"bits": long_bits, "rank": long_rank, "expect": "symbol(i) == bits[i]; rank(i + j) == r_j + i / 16 * 8 for i = 0, 16, .. 496 (expanded here)"}}
    # --- suffixarray/checkCSA.cpp: suffix array of "Hello$World$" and the (sequence, position) the sampled rows answer at sampling rates 3, 4, 5, 8
    src = open(f"{TESTS}/suffixarray/checkCSA.cpp").read()
    sa_col = [int(m.group(1)) for m in re.finditer(r"//\s+\S+\s+\d\s+\d\s+(\d+)\s*$", src, flags=re.M)]
    csa = {"source": "suffixarray/checkCSA.cpp:9-81", "sequences": [[ord(c) for c in "Hello"], [ord(c) for c in "World"]], "sigma": 256, "sa": sa_col, "sampling": {}}
    for m in re.finditer(r'SECTION\("sampling (\d+)"\)', src):
        body = src[m.end(): src.index("check(csa, expected);", m.end())]
        csa["sampling"][m.group(1)] = [[int(a), int(b), int(c)] for a, b, c in re.findall(r"expected\[(\d+)\] = \{(\d+), (\d+)\};", body)]     # row, sequence, position
    out["csa"] = csa
    # --- search_scheme: node counts, validity, completeness, the generators' validity ranges
    src = open(f"{TESTS}/search_scheme/nodeCount.cpp").read()
    known = [[int(a), int(b), int(c), int(d), int(e)] for a, b, c, d, e in
             re.findall(r"CHECK\(\s*(\d+) == ss::nodeCount</\*Edit=\*/false>\(gen::backtracking\((\d+), (\d+), (\d+)\), (\d+)\)\);", src)]
    out["node_count"] = {"source": "search_scheme/nodeCount.cpp:13-34", "sigma": 4, "zero_errors": {"n_from": 1, "n_to": 999, "expect": "nodeCount<false>(backtracking(n, 0, 0)) == n == nodeCount<false>(expand(backtracking(1, 0, 0), n))"},
                         "known": known}                            # [count, N, minK, K, sigma]
    src = open(f"{TESTS}/search_scheme/weightedNodeCount.cpp").read()
    known = [[int(a), int(b), int(c), int(d), int(e), int(f.replace("'", ""))] for a, b, c, d, e, f in
             re.findall(r"CHECK\(\s*(\d+) == ss::weightedNodeCount</\*Edit=\*/false>\(gen::backtracking\((\d+), (\d+), (\d+)\), (\d+), ([\d']+)\)\);", src)]
    out["weighted_node_count"] = {"source": "search_scheme/weightedNodeCount.cpp:13-45", "sigma": 4, "N": 1_000_000_000,
                                  "exact_below": 14, "bounded": {"n_from": 15, "n_to": 999, "below": 16}, "known": known}
    def searches_of(path, call):
        src = open(path).read()
        res = []
        for m in re.finditer(r"CHECK\((not )?ss::%s\(ss::S" % call, src):
            nums, end = _list_after(src, "{", src.index("ss::Search", m.end() - 5) + len("ss::Search"))
            tail = src[end: src.index(";", end)]
            args = [int(x) for x in re.findall(r"\b(\d+)\b", tail)]
            k = len(nums) // 3
            res.append({"line": _line_of(src, m.start()), "expected": m.group(1) is None, "pi": nums[:k], "l": nums[k:2 * k], "u": nums[2 * k:], "args": args})
        return res
    src = open(f"{TESTS}/search_scheme/isValid.cpp").read()
    first, _ = _list_after(src, "auto search = ss::Search{")
    out["is_valid"] = {"source": "search_scheme/isValid.cpp:10-65", "cases": [{"line": 12, "expected": True, "pi": first[:1], "l": first[1:2], "u": first[2:], "args": []}] +
                       searches_of(f"{TESTS}/search_scheme/isValid.cpp", "isValid")}
    out["is_complete"] = {"source": "search_scheme/isComplete.cpp:10-35", "cases": searches_of(f"{TESTS}/search_scheme/isComplete.cpp", "isComplete")}     # args = [minK, maxK]
    out["generators_valid"] = {"source": "search_scheme/checkGenerators.cpp:21-132",
                               "backtracking": {"N": [1, 19], "minK": [0, 9], "maxK_below": 10}, "h2": {"N": [1, 19], "K_below": "min(N, 10)"},
                               "pigeon_trivial": {"minK": [0, 19], "maxK_below": 20}, "pigeon_opt": {"minK": [0, 19], "maxK_below": 20},
                               "expect": "isValid(generator(...)) over the ranges the reference's loops run (the generators this build restates)"}
    return out


# ----------------------------------------------------------------------------------------------- vectors from the real reference
def ref_vectors():
    import fmoracle as fo
    from tests.util import make_text
    if not fo.ref_available():
        raise SystemExit("oracle/_ref/libfmref.so missing: run `make -C oracle ref` first")
    strings = []
    cases = [("IB16", 5, 300), ("IB16", 5, 4100), ("IB8", 5, 600), ("IB32", 5, 200), ("IB16A", 5, 300), ("IBP16", 5, 700),
             ("EPR16", 5, 500), ("EPR8", 5, 504), ("EPR8", 5, 700), ("EPRV2_16", 5, 600), ("EPRV2_8", 5, 512), ("EPRV2_8", 5, 700),
             ("WAVELET", 5, 900), ("WAVELET", 28, 1200), ("IB16", 28, 400), ("EPR16", 28, 300), ("EPRV2_16", 28, 300),
             ("IB16", 255, 310), ("WAVELET", 255, 310), ("IB16", 4, 257), ("EPRV2_16", 6, 129), ("IB16", 21, 200),
             ("EPRV3_8", 5, 512), ("EPRV3_8", 5, 700), ("EPRV3_16", 5, 600), ("EPRV3_32", 5, 300), ("EPRV4", 5, 900), ("EPRV5", 5, 1024),
             ("EPRV5", 28, 300), ("IEPRV7", 5, 777), ("IEPRV7", 28, 256), ("EPRV4", 255, 310), ("EPRV3_16", 6, 129)]
    for layout, sigma, n in cases:
        text = make_text(n, sigma, seed=n + sigma, lo=0)
        r = fo.RefString(layout, sigma, text)
        rk, pr = r.rank_table()
        sym = [r.symbol(i) for i in range(n)]
        entry = {"layout": layout, "sigma": sigma, "n": n, "seed": n + sigma, "block_stride": r.block_stride(),
                 "rank_crc": zlib.crc32(rk.tobytes()), "prefix_rank_crc": zlib.crc32(pr.tobytes()),
                 "symbol_crc": zlib.crc32(np.array(sym, dtype=np.uint8).tobytes()),
                 "rank_last_row": rk[-1].tolist(), "prefix_last_row": pr[-1].tolist()}
        if layout in fo.HIER_LAYOUTS:
            entry["level_crc"] = [zlib.crc32(a.tobytes()) for a in r.level_fields()]
            entry["level_bytes"] = [int(a.size) for a in r.level_fields()]
        elif layout != "WAVELET":
            cnt, words, sup = r.block_fields()
            entry.update({"n_blocks": int(cnt.shape[0]), "n_super": int(sup.shape[0]),
                          "counts_crc": zlib.crc32(cnt.tobytes()), "words_crc": zlib.crc32(words.tobytes()), "super_crc": zlib.crc32(sup.tobytes())})
        else:
            nn = 1 << max(1, (sigma - 1).bit_length())
            entry["node_crc"] = [zlib.crc32(b"".join(r.raw(4 * k + j).tobytes() for j in range(4))) for k in range(nn)]
        if n <= 310:
            entry["rank_table"] = rk.tolist() if sigma <= 6 else None
        strings.append(entry)
    schemes = {"h2": [], "pigeon_opt": [], "pigeon_trivial": [], "backtracking": [], "expand": [], "limitToHamming": [], "partition": []}
    for K in range(0, 5):
        for minK in range(0, K + 1):
            for N in range(K + 1, K + 4):
                s = fo.ref_scheme_h2(N, minK, K)
                schemes["h2"].append({"N": N, "minK": minK, "K": K, "pi": s[0].tolist(), "l": s[1].tolist(), "u": s[2].tolist(),
                                      "valid": fo.ref_scheme_is_valid(s), "complete": fo.ref_scheme_is_complete(s, minK, K),
                                      "nodeCount_sigma5": fo.ref_scheme_node_count_hamming(s, 5)})
                if minK == 0:
                    for L in (N, N + 3, 31, 101):
                        e = fo.ref_scheme_expand(s, L)
                        h = fo.ref_scheme_limit_to_hamming(e)
                        schemes["expand"].append({"gen": "h2", "N": N, "K": K, "len": L, "searches": int(e[0].shape[0]),
                                                  "crc": zlib.crc32(b"".join(x.tobytes() for x in e)),
                                                  "hamming_crc": zlib.crc32(b"".join(x.tobytes() for x in h))})
                    h = fo.ref_scheme_limit_to_hamming(s)
                    schemes["limitToHamming"].append({"N": N, "K": K, "l": h[1].tolist(), "u": h[2].tolist()})
            for name, fn in (("pigeon_opt", fo.ref_scheme_pigeon_opt), ("pigeon_trivial", fo.ref_scheme_pigeon_trivial)):
                s = fn(minK, K)
                schemes[name].append({"minK": minK, "K": K, "pi": s[0].tolist(), "l": s[1].tolist(), "u": s[2].tolist(),
                                      "complete": fo.ref_scheme_is_complete(s, minK, K)})
            s = fo.ref_scheme_backtracking(4, minK, K)
            schemes["backtracking"].append({"N": 4, "minK": minK, "K": K, "pi": s[0].tolist(), "l": s[1].tolist(), "u": s[2].tolist()})
    # expandByWNC as the example's `--gen <name>_dyn` calls it (src/example/main.cpp:116, :135: Edit = true, sigma = 4, N = 3e9) and with Edit = false
    schemes["expandByWNC"] = []
    gens = {"h2-k1": lambda: fo.ref_scheme_h2(3, 0, 1), "h2-k2": lambda: fo.ref_scheme_h2(4, 0, 2), "h2-k3": lambda: fo.ref_scheme_h2(5, 0, 3),
            "pigeon_opt-k2": lambda: fo.ref_scheme_pigeon_opt(0, 2), "pigeon-k1": lambda: fo.ref_scheme_pigeon_trivial(0, 1), "backtracking-k2": lambda: fo.ref_scheme_backtracking(1, 0, 2)}
    for gname, make in gens.items():
        s = make()
        for L in (s[0].shape[1], 20, 31, 50, 101, 151, 250):
            for edit, sigma, N in ((True, 4, 3_000_000_000), (False, 4, 3_000_000_000), (True, 27, 2_000_000_000), (True, 4, 1000)):
                if L < s[0].shape[1]:
                    continue
                e = fo.ref_scheme_expand_by_wnc(s, L, sigma, N, edit)
                schemes["expandByWNC"].append({"gen": gname, "len": L, "edit": edit, "sigma": sigma, "N": N, "searches": int(e[0].shape[0]),
                                               "crc": zlib.crc32(b"".join(x.tobytes() for x in e)),
                                               "wnc": fo.ref_scheme_weighted_node_count(e, sigma, N, edit),
                                               "first_search_pi_runs": [int(c) for c in np.bincount(np.cumsum(np.abs(np.diff(e[0][0].astype(np.int64))) != 1))] if e[0].shape[0] else []})
    for parts, total in ((4, 101), (4, 151), (3, 31), (2, 2), (5, 7), (1, 9)):
        schemes["partition"].append({"parts": parts, "total": total, "out": fo.ref_uniform_partition(parts, total).tolist()})
    return strings, schemes


REF_SEARCH_CASES = {    # name: (layout, sigma, text symbols, text seed, reads, read length)
    "config0_ib16": ("IB16", 5, 1_000_000, 42, 10_000, 31),          # BASELINE.json configs[0]
    "protein_wavelet": ("WAVELET", 28, 300_000, 7, 3_000, 40),        # configs[4] in small: FMIndex<28, Wavelet>, 40 aa
    "epr16": ("EPR16", 5, 200_000, 8, 2_000, 31),
    "eprv2_16": ("EPRV2_16", 5, 200_000, 9, 2_000, 31),
    "eprv5": ("EPRV5", 5, 200_000, 10, 2_000, 31),
    "ibp16": ("IBP16", 5, 200_000, 11, 2_000, 31),
}


def ref_search_intervals():
    """exact-search intervals computed with the REAL reference's rank functions (oracle/_ref/libfmref.so): backward search as
    FMIndexCursor::extendLeft does it (fmindex/FMIndexCursor.h:33-37) over the BWT of the text, early exit on an empty interval like
    search/SearchNoErrors.h:12-26.  The suffix order (hence the BWT) is a property of the text, so any correct suffix sorter gives the
    reference's; here the restatement's.  (lb, len) per read; texts / reads = tests.util.make_text / sample_reads(seed=1, mutate=1)."""
    import fmoracle as fo
    from tests.util import make_text, sample_reads
    out = {}
    for name, (layout, sigma, tn, seed, nreads, rl) in REF_SEARCH_CASES.items():
        text = make_text(tn, sigma, seed=seed)
        ox = fo.OraIndex.build("IB16", sigma, [text], 16, False)
        bs = ox.bwt_string()
        n = bs.size()
        bwt = np.fromiter((bs.symbol(i) for i in range(n)), dtype=np.uint8, count=n)
        ref = fo.RefString(layout, sigma, bwt)
        Cc = np.concatenate([[0], np.cumsum(np.bincount(bwt, minlength=sigma))]).astype(np.int64)
        reads = sample_reads(text, nreads, rl, seed=1, mutate=1, sigma=sigma)
        res = np.zeros((len(reads), 2), dtype=np.uint32)
        for q, r in enumerate(reads):
            lb, ln = 0, n
            for c in r[::-1]:
                a, b = ref.rank(lb, int(c)), ref.rank(lb + ln, int(c))
                lb, ln = int(Cc[c]) + a, b - a
                if ln == 0:
                    break
            res[q] = (lb, ln)
        out[name] = res
    np.savez_compressed(os.path.join(HERE, "ref_search_intervals.npz"), **out)
    return out


def main():
    if "--tests-only" in sys.argv:                               # only the data of the reference's own tests (needs /root/reference, not oracle/_ref)
        with open(os.path.join(HERE, "reference_tests.json"), "w") as f:
            json.dump(reference_tests(), f, separators=(",", ":"))
        return
    ref_search_intervals()
    with open(os.path.join(HERE, "reference_tests.json"), "w") as f:
        json.dump(reference_tests(), f, separators=(",", ":"))
    strings, schemes = ref_vectors()
    with open(os.path.join(HERE, "ref_strings.json"), "w") as f:
        json.dump({"_provenance": "produced by the real reference headers via oracle/_ref/libfmref.so (tests/golden/make_golden.py); "
                                  "texts = tests.util.make_text(n, sigma, seed, lo=0)", "cases": strings}, f, separators=(",", ":"))
    with open(os.path.join(HERE, "ref_schemes.json"), "w") as f:
        json.dump({"_provenance": "produced by the real reference's search_scheme headers via oracle/_ref/libfmref.so", **schemes}, f, separators=(",", ":"))
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()

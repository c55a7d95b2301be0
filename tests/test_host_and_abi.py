"""Host-side logic of the product (search-scheme tables, query flattening, locate expansion, sharding) and the C-ABI surface.
CPU only: no compute call is made — without a GPU the library must refuse loudly."""
import ctypes as C
import json
import os
import re
import subprocess
import sys
import zlib

import numpy as np
import pytest

import fmindex_collection_amd as fm
from fmindex_collection_amd import capi, search_scheme as ss

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REFSCH = json.load(open(os.path.join(GOLD, "ref_schemes.json")))
REF = json.load(open(os.path.join(GOLD, "reference_tests.json")))


def _eq(a, b):
    return all(np.asarray(x).shape == np.asarray(y).shape and np.array_equal(x, y) for x, y in zip(a, b))


@pytest.fixture(scope="module", autouse=True)
def _built_library():
    subprocess.run(["make", "-C", os.path.join(ROOT, "fmindex-collection_amd", "csrc"), "-j4", "-s"], check=True)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "fmgpu.h")).read()
    declared = set(re.findall(r"\b(fmgpu_[a-z0-9_]+)\s*\(", header))
    assert declared and declared == set(capi.EXPORTS)
    L = capi.lib()
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert L.fmgpu_abi_version() == 6


def test_no_gpu_means_loud_failure_not_fallback():
    n = C.c_int(-1)
    rc = capi.lib().fmgpu_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(fm.FmgpuError) as e:
        fm.FMIndex.from_sequences([[1, 2, 3]], 5)
    assert e.value.code == capi.FMGPU_ERR_NO_DEVICE
    d = capi.IndexDesc()
    h = C.c_void_p()
    assert capi.lib().fmgpu_index_create(C.byref(d), C.byref(h)) == capi.FMGPU_ERR_NO_DEVICE
    assert b"no CPU fallback" in capi.lib().fmgpu_last_error()


def test_index_file_errors_without_a_gpu(tmp_path):
    """fmgpu_index_load refuses what is not an index file of this library before it touches a device: missing file, foreign bytes (e.g. a cereal archive
    of the reference), a truncated header, another format version, another ABI version, the other byte order"""
    L = capi.lib()
    h = C.c_void_p()
    assert L.fmgpu_index_load(None, C.byref(h)) == capi.FMGPU_ERR_INVALID
    assert L.fmgpu_index_load(os.fsencode(tmp_path / "missing.idx"), C.byref(h)) == capi.FMGPU_ERR_INVALID and b"cannot open" in L.fmgpu_last_error()
    assert L.fmgpu_index_save(None, b"/tmp/x", 1) == capi.FMGPU_ERR_INVALID
    import struct
    def header(magic=b"FMGPUIDX", version=1, abi=L.fmgpu_abi_version(), wide=0, probe=0x01020304):
        return magic + struct.pack("<IIIIQQQQQ", version, abi, wide, probe, 0, 0, 0, 0, 0)
    cases = [(b"\x01\x00\x00\x00cereal-like bytes" * 8, capi.FMGPU_ERR_INVALID, b"not an index file"), (header()[:40], capi.FMGPU_ERR_INVALID, b"truncated"),
             (header(version=2), capi.FMGPU_ERR_UNSUPPORTED, b"format version"), (header(abi=4), capi.FMGPU_ERR_UNSUPPORTED, b"ABI version"),
             (header(probe=0x04030201), capi.FMGPU_ERR_UNSUPPORTED, b"byte order"), (header(wide=7), capi.FMGPU_ERR_INVALID, b"row width")]
    for k, (blob, code, msg) in enumerate(cases):
        f = tmp_path / ("bad%d.idx" % k)
        f.write_bytes(blob)
        assert L.fmgpu_index_load(os.fsencode(f), C.byref(h)) == code and msg in L.fmgpu_last_error(), (k, L.fmgpu_last_error())
        assert not h.value
    # the replica set (one process, several devices) refuses the same way: null arguments; no device in this container
    r = C.c_void_p()
    assert L.fmgpu_replicas_load(None, None, 0, C.byref(r)) == capi.FMGPU_ERR_INVALID
    assert L.fmgpu_replicas_load(b"/tmp/x.idx", None, 0, C.byref(r)) in (capi.FMGPU_ERR_NO_DEVICE, capi.FMGPU_ERR_INVALID) and not r.value
    assert L.fmgpu_replicas_search_exact(None, None, None, 0, None, None, None) == capi.FMGPU_ERR_INVALID and L.fmgpu_replicas_destroy(None) == 0


def test_argument_errors():
    L = capi.lib()
    assert L.fmgpu_index_create(None, None) == capi.FMGPU_ERR_INVALID
    assert L.fmgpu_search_exact(None, None, None, 1, None, None, None, None) == capi.FMGPU_ERR_INVALID
    assert L.fmgpu_index_destroy(None) == 0


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "fmindex-collection_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "fmoracle" not in src and "oracle/" not in src, f
    for f in os.listdir(os.path.join(ROOT, "include")):
        assert "fmoracle" not in open(os.path.join(ROOT, "include", f)).read()


# ------------------------------------------------------------------------------------------------ search schemes (host tables)
def test_scheme_generators_match_the_real_reference():
    for g in REFSCH["h2"]:
        s = ss.h2(g["N"], g["minK"], g["K"])
        assert _eq(s, (g["pi"], g["l"], g["u"])), g
        assert ss.isValid(s) == g["valid"] and ss.isComplete(s, g["minK"], g["K"]) == g["complete"]
        assert ss.nodeCount(s, 5) == pytest.approx(g["nodeCount_sigma5"], rel=1e-9)
    for name, fn in (("pigeon_opt", ss.pigeon_opt), ("pigeon_trivial", ss.pigeon_trivial)):
        for g in REFSCH[name]:
            s = fn(g["minK"], g["K"])
            assert _eq(s, (g["pi"], g["l"], g["u"])) and ss.isComplete(s, g["minK"], g["K"]) == g["complete"]
    for g in REFSCH["backtracking"]:
        assert _eq(ss.backtracking(g["N"], g["minK"], g["K"]), (g["pi"], g["l"], g["u"]))
    for g in REFSCH["expand"]:
        e = ss.expand(ss.h2(g["N"], 0, g["K"]), g["len"])
        assert e[0].shape[0] == g["searches"] and zlib.crc32(b"".join(np.ascontiguousarray(x).tobytes() for x in e)) == g["crc"], g
        assert zlib.crc32(b"".join(np.ascontiguousarray(x).tobytes() for x in ss.limitToHamming(e))) == g["hamming_crc"]
    for g in REFSCH["limitToHamming"]:
        h = ss.limitToHamming(ss.h2(g["N"], 0, g["K"]))
        assert h[1].tolist() == g["l"] and h[2].tolist() == g["u"]
    for g in REFSCH["partition"]:
        assert ss.createUniformPartition(g["parts"], g["total"]).tolist() == g["out"]


def test_expand_by_wnc_equals_the_real_reference():
    """expand.h:218-247 + weightedNodeCount.h:21-69 (long double sums, a double running best): the Python mirror reproduces every scheme the real
    reference's expandByWNC produced (tests/golden/ref_schemes.json), incl. the example's call — Edit = true, sigma 4, N = 3e9"""
    for g in REFSCH["expandByWNC"]:
        if g["len"] > 101 and g["gen"] not in ("h2-k2", "pigeon-k1"):
            continue                                              # (the longest cases of every generator run in the C++ twin of this test)
        base = {"h2-k1": lambda: ss.h2(3, 0, 1), "h2-k2": lambda: ss.h2(4, 0, 2), "h2-k3": lambda: ss.h2(5, 0, 3), "pigeon_opt-k2": lambda: ss.pigeon_opt(0, 2),
                "pigeon-k1": lambda: ss.pigeon_trivial(0, 1), "backtracking-k2": lambda: ss.backtracking(1, 0, 2)}[g["gen"]]()
        e = ss.expandByWNC(base, g["len"], g["sigma"], g["N"], g["edit"])
        assert e[0].shape[0] == g["searches"] and zlib.crc32(b"".join(np.ascontiguousarray(x, dtype=np.uint64).tobytes() for x in e)) == g["crc"], g
        assert float(ss.weightedNodeCount(e, g["sigma"], g["N"], g["edit"])) == pytest.approx(g["wnc"], rel=1e-12)


def test_scheme_reference_test_cases():
    """search_scheme/expand.cpp:11-60, checkGeneratorsIsComplete.cpp:48-60"""
    for c in REF["expand"]["cases"]:
        e = ss.expand(tuple(np.array([x], dtype=np.uint64) for x in c["in"]), c["len"])
        assert ss.isValid(e) and _eq(e, tuple(np.array([x]) for x in c["out"]))
    for N in range(1, 10):
        for minK in range(0, min(N, 5)):
            for maxK in range(minK, min(N, 5)):
                assert ss.isComplete(ss.h2(N, minK, maxK), minK, maxK)
    assert not ss.isValid((np.array([[0, 2, 1]]), np.zeros((1, 3), dtype=int), np.ones((1, 3), dtype=int)))
    assert not ss.isComplete((np.array([[0, 1, 2]]), np.zeros((1, 3), dtype=int), np.array([[0, 1, 1]])), 0, 2)
    with pytest.raises(ValueError):
        ss.h2(2, 0, 2)
    with pytest.raises(ValueError):
        ss.createUniformPartition(4, 3)
    assert ss.createUniformPartition(ss.h2(4, 0, 2), 101).tolist() == [26, 25, 25, 25]
    assert ss.createUniformPartition(4, 151).tolist() == [38, 38, 38, 37]


def test_flatten_and_locate_expansion():
    qbuf, qoff = fm.flatten([[1, 2, 3], [], [4]])
    assert qoff.tolist() == [0, 3, 3, 4] and qbuf[:4].tolist() == [1, 2, 3, 4]
    qbuf, qoff = fm.flatten([])
    assert qoff.tolist() == [0]
    ll = fm.LocateLinear(None, [10, 3, 7], [2, 0, 3])
    assert ll.rows.tolist() == [10, 11, 7, 8, 9] and ll.owner.tolist() == [0, 0, 2, 2, 2]


def test_shard_ranges_cover_the_batch():
    from fmindex_collection_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 100, 10_000_001):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


_RUNTIME_PROBE = """
import sys
sys.path.insert(0, %r)
from fmindex_collection_amd import capi
capi.lib()
import torch
libs = sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l))
print('HIPLIBS', len(libs))
if %d:
    import ctypes as C
    p = C.c_void_p()
    assert capi.lib().fmgpu_malloc(C.byref(p), 1 << 20) == 0
    assert float(torch.ones(8, device='cuda').sum()) == 8.0
    print('GPU ok')
"""


def test_one_hip_runtime_when_the_library_is_loaded_before_torch():
    """the torch wheel bundles its own libamdhip64.so; loading libfmgpu.so first used to leave two HIP runtimes in the process (and the second one
    to initialise finds no GPU): capi.lib() binds to torch's copy when torch is installed"""
    pytest.importorskip("torch")
    r = subprocess.run([sys.executable, "-c", _RUNTIME_PROBE % (ROOT, 0)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "HIPLIBS 1" in r.stdout, r.stdout + r.stderr[-800:]


@pytest.mark.gpu
def test_library_first_then_torch_on_the_gpu():
    pytest.importorskip("torch")
    r = subprocess.run([sys.executable, "-c", _RUNTIME_PROBE % (ROOT, 1)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "HIPLIBS 1" in r.stdout and "GPU ok" in r.stdout, r.stdout + r.stderr[-800:]


def test_bench_line_is_compact():
    """the driver keeps 8 KB of stdout: bench.py's last line (headline + roofline + cpu_baseline + a summary of every record) must stay well below
    that however many records a run holds; the full records go to a file"""
    sys.path.insert(0, ROOT)
    import bench
    long_text = {"text": "genome", "generator": "x" * 300, "symbols": 3088286401, "sequences": 25, "repeat_fraction": 0.45}
    def rec(rid, kernel, frac):
        return {"id": rid, "metric": "queries/sec (GRCh38-sized index, 10M x 101bp, exact)", "value": 5.3e8, "unit": "queries/s", "n_gpus": 1, "steps": 20, "warmup": 5,
                "ms_per_step": 18.912345678, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                "config": {"workload": "grch38_exact", "text": long_text, "sigma": 5, "layout": "InterleavedBitvector16", "index": "FMIndex", "index_kind": rid.split("/")[-1],
                           "queries_per_gpu": 10_000_000, "read_len": 101, "index_device_bytes": 4_200_000_000, "tables": {"a": "y" * 200}},
                "gbp_per_s": 53.5, "hits": 9_000_000,
                "roofline": {"bound": "hbm", "achieved": 5600.123456, "peak": 8000.0, "unit": "GB/s", "frac": frac, "traffic": 1.349e11, "kernel": kernel, "kernel_ms": 18.9,
                             "units_per_launch": 961847926.0, "bytes_per_unit": 112, "accounting": "z" * 400, "traffic_source": "w" * 300, "rule": "u" * 600,
                             "frac_kernel_format": frac, "frac_sec8d": None, "frac_loaded": 0.4123456, "frac_traffic": 0.9, "sec8d": {"frac_uncapped": 1.2345678, "what": "v" * 300}},
                "clocks": {"sclk_mhz_mean": 2100, "sclk_mhz_min": 2050, "sclk_mhz_max": 2400, "mclk_mhz": 2000, "socket_power_w_mean": 900, "power_cap_w": 1400.0, "samples": 40, "what": "q" * 100},
                "exchange": {"collective": "gather", "bytes_per_rank_and_step": 80_000_000, "verified_on_rank0": True, "world_size_seen": 8, "record": "r" * 200}}
    ids = ["%s/%s/%s" % (t_, w, k) for t_ in ("genome", "uniform") for w in ("exact", "locate", "k2", "k2_151", "k2_edit") for k in ("plain", "tables")]
    ids += ["protein/exact/wavelet", "protein/exact/tables", "protein_wide/exact/wavelet", "protein_wide/exact/tables"]
    records = [rec(i, "k_scheme_fast_edit", 0.4712345) for i in ids]
    head = next(r for r in records if r["id"] == "genome/exact/plain")
    head["cpu_baseline"] = {"value": 2.6e5, "unit": "queries/s", "cores": 16, "kind": "port", "sample_reads": 3_104_452, "seconds": 11.9, "sample": "s" * 400,
                            "single_thread": {"value": 1.3e5, "unit": "queries/s", "sample": "t" * 100}, "parallel_efficiency": 0.98, "gpu_results_match_on_sample": True,
                            "compared": "c" * 100}
    next(r for r in records if r["id"] == "genome/k2/plain")["cpu_baseline"] = dict(head["cpu_baseline"])
    for multi in (False, True):
        line = bench.compact_line(records, multi, "bench_records.json")
        assert len(line) < bench.MAX_LINE <= 4000 and "\n" not in line
        d = json.loads(line)
        assert d["config"]["record"] == "genome/exact/plain" and d["config"]["index_kind"] == "plain"       # the headline is the plain index
        assert d["roofline"]["frac"] == pytest.approx(0.4712, rel=1e-3) and d["roofline"]["bytes_per_unit"] == 112 and d["roofline"]["kernel_ms"] == 18.9
        assert d["cpu_baseline"]["cores"] == 16 and d["cpu_baseline"]["gpu_results_match_on_sample"] is True
        # every record of the headline's text and of the protein configurations is in the line; the comparison text's go to the records file when the line would grow past its room
        assert d["with_tables"]["record"] == "genome/exact/tables" and set(d["summary"]) | {i for i in ids if i.split("/")[0] in d.get("summary_also_in_records_file", [])} == set(ids)
        assert {i for i in ids if not i.startswith("uniform/")} <= set(d["summary"])
        # one rule, three figures: kernel format (= frac), SURVEY 8d (null where it exceeds the peak, the uncapped figure beside it), counted in the kernel; the clocks of the run
        assert d["roofline"]["frac_kernel_format"] == d["roofline"]["frac"] and d["roofline"]["frac_sec8d"] is None and d["roofline"]["sec8d_uncapped"] == pytest.approx(1.235, rel=1e-3)
        assert d["roofline"]["frac_loaded"] == pytest.approx(0.4123, rel=1e-3) and d["clocks"]["sclk_mhz_mean"] == 2100 and len(d["summary"]["genome/k2/plain"]) == 4
        assert ("secondary" in d) == multi
    # a run with absurdly many records still prints a parseable headline (the optional parts go first)
    many = records + [rec("x%d/exact/plain" % i, "k", 0.1) for i in range(400)]
    line = bench.compact_line(many, False, "bench_records.json")
    assert len(line) < bench.MAX_LINE and json.loads(line)["roofline"]["kernel"] == "k_scheme_fast_edit"


def test_options_are_set_through_the_abi_not_the_environment(monkeypatch):
    """fmgpu_set_option / fmgpu_get_option (include/fmgpu.h): defaults, round trip, unknown options and selection bits outside FMGPU_SEL_ALL are refused;
    the shipped library reads no environment variable (an FMGPU_* variable in the environment changes nothing)"""
    import fmindex_collection_amd as fm
    L = capi.lib()
    for name, value in capi.OPTION_DEFAULTS.items():
        assert fm.options[name] == value, name
    monkeypatch.setenv("FMGPU_PAIRS", "0")
    monkeypatch.setenv("FMGPU_DEV_FLAGS", "2")
    assert fm.options["pair_table"] == 1 and fm.options["kernel_select"] == 0
    with fm.options(pair_table=0, kernel_select=capi.SEL_GENERAL_DFS | capi.SEL_NO_LEAN):
        assert fm.options["pair_table"] == 0 and fm.options["kernel_select"] == capi.SEL_GENERAL_DFS | capi.SEL_NO_LEAN
    assert fm.options["pair_table"] == 1 and fm.options["kernel_select"] == 0
    v = C.c_int64()
    assert L.fmgpu_set_option(99, 1) == capi.FMGPU_ERR_INVALID and L.fmgpu_get_option(-1, C.byref(v)) == capi.FMGPU_ERR_INVALID and L.fmgpu_get_option(0, None) == capi.FMGPU_ERR_INVALID
    assert L.fmgpu_set_option(capi.OPTIONS["kernel_select"], 1) == capi.FMGPU_ERR_INVALID and b"FMGPU_SEL_ALL" in L.fmgpu_last_error()      # (bit 0 — count only — exists in development builds alone)
    assert L.fmgpu_set_option(capi.OPTIONS["suffix_sorter"], 4) == capi.FMGPU_ERR_INVALID and L.fmgpu_set_option(capi.OPTIONS["suffix_sorter"], -1) == capi.FMGPU_ERR_INVALID
    assert L.fmgpu_set_option(capi.OPTIONS["bucket_rows"], -5) == capi.FMGPU_ERR_INVALID
    assert L.fmgpu_set_option(capi.OPTIONS["kernel_select"], capi.SEL_NO_BOARD) == 0 and L.fmgpu_set_option(capi.OPTIONS["kernel_select"], 0) == 0
    src = "".join(open(os.path.join(ROOT, "fmindex-collection_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "fmindex-collection_amd", "csrc")) if f.endswith((".hip", ".h")))
    import re
    outside_dev = re.sub(r"#ifdef FMGPU_DEV\b.*?#e(?:lse|ndif)", "", src, flags=re.S)
    assert "getenv" not in outside_dev


def test_scheme_mirror_against_the_reference_tests_own_values():
    """search_scheme/nodeCount.cpp:13-34, weightedNodeCount.cpp:13-45, isValid.cpp:10-65, isComplete.cpp:10-35, checkGenerators.cpp:21-132 through the Python
    mirror of the reference's search_scheme functions (the values are the reference tests' own: tests/golden/reference_tests.json)"""
    import numpy as np
    from fmindex_collection_amd import search_scheme as ss
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_tests.json")))
    g = ref["node_count"]
    for n in list(range(1, 40)) + list(range(40, g["zero_errors"]["n_to"] + 1, 37)):
        assert ss.nodeCount(ss.backtracking(n, 0, 0), g["sigma"]) == n == ss.nodeCount(ss.expand(ss.backtracking(1, 0, 0), n), g["sigma"])
    for count, N, minK, K, sigma in g["known"]:
        assert ss.nodeCount(ss.backtracking(N, minK, K), sigma) == count
    w = ref["weighted_node_count"]
    for n in range(1, w["exact_below"]):
        assert ss.weightedNodeCount(ss.backtracking(n, 0, 0), w["sigma"], w["N"]) == n == ss.weightedNodeCount(ss.expand(ss.backtracking(1, 0, 0), n), w["sigma"], w["N"])
    for n in list(range(w["bounded"]["n_from"], 60)) + list(range(60, w["bounded"]["n_to"] + 1, 53)):
        assert ss.weightedNodeCount(ss.backtracking(n, 0, 0), w["sigma"], w["N"]) < w["bounded"]["below"]
        assert ss.weightedNodeCount(ss.expand(ss.backtracking(1, 0, 0), n), w["sigma"], w["N"]) < w["bounded"]["below"]
    for count, N, minK, K, sigma, size in w["known"]:
        assert ss.weightedNodeCount(ss.backtracking(N, minK, K), sigma, size) == count
    as_scheme = lambda c: tuple(np.array([c[k]], dtype=np.uint64) for k in ("pi", "l", "u"))
    for c in ref["is_valid"]["cases"]:
        assert bool(ss.isValid(as_scheme(c))) == c["expected"], c
    for c in ref["is_complete"]["cases"]:
        assert bool(ss.isComplete(as_scheme(c), *c["args"])) == c["expected"], c
    for N in range(1, 20):
        for minK in range(0, 10):
            for maxK in range(minK, 10):
                assert ss.isValid(ss.backtracking(N, minK, maxK))
        for minK in range(0, min(N, 10)):
            for maxK in range(minK, min(N, 10)):
                assert ss.isValid(ss.h2(N, minK, maxK))
    for minK in range(0, 20):
        for maxK in range(minK, 20):
            assert ss.isValid(ss.pigeon_trivial(minK, maxK)) and ss.isValid(ss.pigeon_opt(minK, maxK))

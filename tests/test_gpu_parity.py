"""Parity of the HIP path against the oracle, through the C-ABI (include/fmgpu.h).  Run with -m gpu on an MI355X.

Bit-exact bar: SA intervals [lb, lb+len), lbRev, error counts, callback order, locate triples, String_c values."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import fmoracle as fo
import fmindex_collection_amd as fm
from fmindex_collection_amd import capi
from tests.util import make_text, sample_reads, oracle_arrays, string_arrays

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF = json.load(open(os.path.join(GOLD, "reference_tests.json")))
LAYOUTS = ["IB8", "IB16", "IB32", "IB16A", "IBP16", "EPR8", "EPR16", "EPR32", "EPRV2_8", "EPRV2_16", "EPRV2_32", "WAVELET",
           "EPRV3_8", "EPRV3_16", "EPRV3_32", "EPRV4", "EPRV5", "IEPRV7", "FBV_64_64K", "FBV_512_64K", "FBV_2048_64K"]
HIT_KEYS = ("qidx", "lb", "lb_rev", "len", "errors")


def gpu_index(ox):
    cls = fm.BiFMIndex if ox.bidirectional else fm.FMIndex
    return cls.from_reference_arrays(**oracle_arrays(ox))


def same_hits(g, o, unidirectional=False):
    if len(g) != len(o):
        return False
    keys = [k for k in HIT_KEYS if not (unidirectional and k == "lb_rev")]
    return all(np.array_equal(g[k].astype(np.uint64), o[k].astype(np.uint64)) for k in keys)


def repeat_text(seed, n=3000):
    rng = np.random.default_rng(seed)
    base = rng.integers(1, 5, size=n // 3, dtype=np.uint8)
    return [np.concatenate([base, base[n // 12: n // 4], rng.integers(1, 5, size=n // 3, dtype=np.uint8)]), base[::-1].copy(),
            np.tile(np.array([1, 2, 1, 3], dtype=np.uint8), 40)]


def mutated_queries(seqs, count, lo, hi, maxsub, seed, sigma=5):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        s = seqs[int(rng.integers(0, len(seqs)))]
        m = int(rng.integers(lo, min(hi, len(s))))
        p = int(rng.integers(0, len(s) - m + 1))
        q = s[p: p + m].copy()
        for _ in range(int(rng.integers(0, maxsub + 1))):
            q[int(rng.integers(0, m))] = rng.integers(1, sigma)
        out.append(q)
    return out


# ------------------------------------------------------------------------------------------------ String_c
@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("sigma", [4, 5, 28, 255])
def test_string_concept_all_layouts(layout, sigma):
    """rank / prefix_rank / symbol for every idx and symbol (string/concepts.h:50-64), sizes around block / super-block edges"""
    for n in (1, 64, 65, 256, 300, 1300):
        if sigma == 255 and n > 300:
            continue
        text = make_text(n, sigma, seed=n + sigma, lo=0)
        s = fo.OraString(layout, sigma, text)
        C_arr = np.array([int(np.count_nonzero(text < c)) for c in range(sigma + 1)], dtype=np.uint64)
        gx = fm.FMIndex.from_reference_arrays(bwt=string_arrays(s), C_array=C_arr)
        ork, opr = s.rank_table()
        idx = np.repeat(np.arange(n + 1, dtype=np.uint64), sigma)
        sym = np.tile(np.arange(sigma, dtype=np.uint8), n + 1)
        assert np.array_equal(gx.rank(idx, sym).reshape(n + 1, sigma), ork), (layout, sigma, n)
        assert np.array_equal(gx.prefix_rank(idx, sym).reshape(n + 1, sigma), opr), (layout, sigma, n)
        assert np.array_equal(gx.symbol(np.arange(n, dtype=np.uint64)), text.astype(np.uint64))


def test_string_hallo_welt_golden():
    """string/unittest.cpp:52-312 through the device"""
    g = REF["hallo_welt"]
    text = np.array(g["text"], dtype=np.uint8)
    for layout in ("IB16", "EPRV2_16", "WAVELET"):
        s = fo.OraString(layout, 255, text)
        gx = fm.FMIndex.from_reference_arrays(bwt=string_arrays(s), C_array=np.array([int(np.count_nonzero(text < c)) for c in range(256)], dtype=np.uint64))
        r = np.array(g["rank"]); p = np.array(g["prefix_rank"])
        assert np.array_equal(gx.rank(r[:, 0], r[:, 1]), r[:, 2].astype(np.uint64))
        assert np.array_equal(gx.prefix_rank(p[:, 0], p[:, 1]), p[:, 2].astype(np.uint64))


# ------------------------------------------------------------------------------------------------ exact search
@pytest.mark.parametrize("layout", LAYOUTS)
def test_exact_search_all_layouts(layout):
    seqs = repeat_text(1)
    ox = fo.OraIndex.build(layout, 5, seqs, 8, False)
    gx = gpu_index(ox)
    queries = mutated_queries(seqs, 700, 1, 120, 1, seed=2)
    qbuf, qoff = fm.flatten(queries)
    lb, ln, st = fm.search_no_errors.search(gx, (qbuf, qoff), want_stats=True)
    olb, oln, ost = ox.search_exact(qbuf, qoff, want_steps=True)
    assert np.array_equal(ln, oln) and np.array_equal(lb, olb)
    assert st.lf_steps == int(ost.sum())


@pytest.mark.parametrize("shape", ["repeats", "many_sequences", "poly_a", "too_many_sequences", "tiny"])
def test_exact_search_in_pair_steps(shape, monkeypatch):
    """Format P / k_exact_p (two symbols per step on a sigma = 5 index): intervals, rows of misses and step counts equal the one-symbol search's
    (the oracle's: SearchNoErrors.h:12-26) — reads of odd and even length, misses found in the first and in the second symbol of a pair, delimiters
    and bytes outside the alphabet inside reads, 'AA'-rich texts (listed rows sit in the planes as the pair 'AA')."""
    rng = np.random.default_rng(77)
    if shape == "repeats":
        seqs = repeat_text(5, n=6000)
    elif shape == "many_sequences":
        seqs = [rng.integers(1, 5, size=int(rng.integers(1, 90)), dtype=np.uint8) for _ in range(250)]
    elif shape == "poly_a":
        seqs = [np.where(rng.random(int(rng.integers(2, 700))) < 0.85, 1, rng.integers(1, 5, size=1)[0]).astype(np.uint8) for _ in range(60)]
    elif shape == "too_many_sequences":
        seqs = [rng.integers(1, 5, size=int(rng.integers(1, 30)), dtype=np.uint8) for _ in range(300)]
    else:
        seqs = [np.array([1], dtype=np.uint8), np.array([2, 1], dtype=np.uint8)]
    ox = fo.OraIndex.build("IB16", 5, seqs, 4, False)
    gx = gpu_index(ox)
    fm.options["pair_table"] = "0"
    gx_single = gpu_index(ox)
    del fm.options["pair_table"]
    has_table = gx.device_bytes > gx_single.device_bytes
    assert has_table == (shape != "too_many_sequences")
    long_enough = [q for q in seqs if len(q) > 2]
    queries = mutated_queries(long_enough, 1500, 1, 140, 2, seed=6) if long_enough else []
    queries += [[], [1], [1, 1], [1, 1, 1], [4, 4, 4, 4], [0], [1, 0], [0, 1], [1, 0, 1, 1], [2, 1, 0],
                [1] * 64, [1] * 65, [1] * 129, [2, 1], [1, 2]]
    for s_ in seqs[:40]:                                     # whole sequences, their ends and what follows a delimiter
        queries += [s_, s_[-3:], s_[:3], np.concatenate([s_[-2:], [0]]), np.concatenate([[0], s_[:2]])]
    qbuf, qoff = fm.flatten(queries)
    olb, oln, ost = ox.search_exact(qbuf, qoff, want_steps=True)
    for flags in ("0", str(1 << 22)):
        fm.options["kernel_select"] = flags
        for g in (gx, gx_single):
            lb, ln, st = fm.search_no_errors.search(g, (qbuf, qoff), want_stats=True)
            assert np.array_equal(ln, oln) and np.array_equal(lb, olb), (shape, flags)
            assert st.lf_steps == int(ost.sum())
    del fm.options["kernel_select"]
    # the packed form (one word per read) goes through the same kernel
    packed = fm.search_no_errors.search_packed(gx, (qbuf, qoff))
    assert np.array_equal(packed, (olb << np.uint64(32)) | oln)
    # bytes outside the alphabet (undefined in the reference, an empty interval here): both kernels end at the same step
    odd = fm.flatten([[1, 2, 9, 1], [9, 1], [1, 9], [7], [1, 1, 9], [9, 1, 1], [1, 1, 1, 9, 1, 1]])
    a = fm.search_no_errors.search(gx, odd, want_stats=True)
    b = fm.search_no_errors.search(gx_single, odd, want_stats=True)
    assert not a[1].any() and np.array_equal(a[0], b[0]) and a[2].lf_steps == b[2].lf_steps
    # an interval table in front of the pair table (fmgpu_index_accelerate_exact(h, 1, L, 0)): reads whose last L symbols are ordinary start from its entry — the
    # intervals, the rows of misses and the step counts stay the one-symbol search's, also for reads shorter than L, strings the text does not hold (the
    # entry is empty: walked from the start), delimiters and foreign bytes among the last L symbols
    for lut_len in (1, 2, 5, 7):
        gx.accelerate(1, lut_len=lut_len, walk=0)
        assert bool(gx.formats & capi.FMT_INTERVALS)
        for sel in (0, capi.SEL_NO_EXACT_LUT):               # k_exact_p behind the table, then the table-driven kernel on the same handle
            with fm.options(kernel_select=sel):
                lb, ln, st = fm.search_no_errors.search(gx, (qbuf, qoff), want_stats=True)
                c = fm.search_no_errors.search(gx, odd, want_stats=True)
            assert np.array_equal(ln, oln) and np.array_equal(lb, olb) and st.lf_steps == int(ost.sum()), (shape, lut_len, sel)
            assert not c[1].any() and np.array_equal(c[0], b[0]) and c[2].lf_steps == b[2].lf_steps
            if sel == 0 and has_table:
                served = int(sum(lut_len for q_, l_ in zip(queries, oln) if len(q_) >= lut_len and all(1 <= int(v) <= 4 for v in list(q_)[-lut_len:]) and
                                 ox.search_exact(*fm.flatten([list(q_)[-lut_len:]]))[1][0] > 0))
                assert st.table_steps == served, (shape, lut_len)
    gx.accelerate(1, lut_len=0, walk=0)
    assert not (gx.formats & capi.FMT_INTERVALS)


@pytest.mark.parametrize("sigma", [6, 21, 28, 29, 30])
@pytest.mark.parametrize("built_on_gpu", [False, True])
def test_exact_search_on_symbol_planes(sigma, built_on_gpu, monkeypatch):
    """Format S / k_exact_s (one line per LF step and end beside a Wavelet bwt, 6 <= sigma <= 29): intervals, miss rows and step counts equal the
    search on the wavelet levels (k_exact_m) and the oracle's; sigma = 30 has no room for its counts in a line and keeps the tree."""
    rng = np.random.default_rng(sigma)
    base = rng.integers(1, sigma, size=5000, dtype=np.uint8)
    seqs = [np.concatenate([base, base[700:1900]]), rng.integers(1, min(sigma, 4), size=900, dtype=np.uint8), rng.integers(1, sigma, size=64, dtype=np.uint8),
            np.full(200, sigma - 1, dtype=np.uint8), np.array([1], dtype=np.uint8)]
    ox = fo.OraIndex.build("WAVELET", sigma, seqs, 8, False)
    make = (lambda: fm.FMIndex.from_sequences(seqs, sigma, "WAVELET", 8)) if built_on_gpu else (lambda: gpu_index(ox))
    gx = make()
    fm.options["symbol_planes"] = "0"
    gx_tree = make()
    del fm.options["symbol_planes"]
    assert (gx.device_bytes > gx_tree.device_bytes) == (sigma <= 29)
    queries = mutated_queries([q for q in seqs if len(q) > 2], 1500, 1, 90, 1, seed=3, sigma=sigma)
    queries += [[], [1], [sigma - 1], [0], [1, 0], [0, 1], [sigma - 1] * 64, [sigma - 1] * 201, seqs[2], seqs[2][1:], np.concatenate([seqs[2][-5:], [0]])]
    qbuf, qoff = fm.flatten(queries)
    olb, oln, ost = ox.search_exact(qbuf, qoff, want_steps=True)
    for flags in ("0", str(1 << 21)):
        fm.options["kernel_select"] = flags
        for g in (gx, gx_tree):
            lb, ln, st = fm.search_no_errors.search(g, (qbuf, qoff), want_stats=True)
            assert np.array_equal(ln, oln) and np.array_equal(lb, olb), (sigma, flags)
            assert st.lf_steps == int(ost.sum())
    del fm.options["kernel_select"]
    odd = fm.flatten([[1, 2, sigma, 1], [sigma + 3, 1], [1, 255]])       # bytes outside the alphabet: an empty interval, at the same step in both kernels
    a = fm.search_no_errors.search(gx, odd, want_stats=True)
    b = fm.search_no_errors.search(gx_tree, odd, want_stats=True)
    assert not a[1].any() and np.array_equal(a[0], b[0]) and a[2].lf_steps == b[2].lf_steps
    if sigma <= 29:
        # an interval table in front of the symbol planes (fmgpu_index_accelerate_exact(h, 0, L, 0)): k_exact_s starts from the entry of a read's last L symbols —
        # intervals, miss rows and step counts unchanged; switched off by selection, the table-driven kernel serves the same handle
        for lut_len in (1, 2, 3):
            gx.accelerate(0, lut_len=lut_len, walk=0)
            for sel in (0, capi.SEL_NO_EXACT_LUT):
                with fm.options(kernel_select=sel):
                    lb, ln, st = fm.search_no_errors.search(gx, (qbuf, qoff), want_stats=True)
                    c = fm.search_no_errors.search(gx, odd, want_stats=True)
                assert np.array_equal(ln, oln) and np.array_equal(lb, olb) and st.lf_steps == int(ost.sum()), (sigma, lut_len, sel)
                assert not c[1].any() and np.array_equal(c[0], b[0]) and c[2].lf_steps == b[2].lf_steps
                assert (st.table_steps > 0) == (sel == 0) and st.table_steps % lut_len == 0
        gx.accelerate(0, lut_len=0, walk=0)


@pytest.mark.parametrize("layout", ["EPR16", "EPRV2_16", "EPR32"])
@pytest.mark.parametrize("sigma", [6, 21, 28])
def test_symbol_planes_beside_epr_blocks(layout, sigma):
    """Format S is also derived beside InterleavedEPR* / InterleavedEPRV2* blocks that are read in place (6 <= sigma <= 29): exact search then takes
    k_exact_s; intervals, miss rows and step counts equal the oracle's and the layout's own kernel's (FMGPU_SEL_EXACT_ON_TREE)"""
    rng = np.random.default_rng(sigma * 7 + len(layout))
    base = rng.integers(1, sigma, size=4000, dtype=np.uint8)
    seqs = [np.concatenate([base, base[500:1500]]), rng.integers(1, min(sigma, 4), size=700, dtype=np.uint8), np.full(150, sigma - 1, dtype=np.uint8)]
    ox = fo.OraIndex.build(layout, sigma, seqs, 8, False)
    gx = gpu_index(ox)
    fm.options["symbol_planes"] = "0"
    try:
        gx_own = gpu_index(ox)
    finally:
        del fm.options["symbol_planes"]
    assert gx.device_bytes > gx_own.device_bytes
    queries = mutated_queries([q for q in seqs if len(q) > 2], 1200, 1, 80, 1, seed=4, sigma=sigma) + [[], [1], [0], [sigma - 1] * 70, [1, 0, 1]]
    qbuf, qoff = fm.flatten(queries)
    olb, oln, ost = ox.search_exact(qbuf, qoff, want_steps=True)
    for flags in ("0", str(1 << 21)):
        fm.options["kernel_select"] = flags
        try:
            for g in (gx, gx_own):
                lb, ln, st = fm.search_no_errors.search(g, (qbuf, qoff), want_stats=True)
                assert np.array_equal(ln, oln) and np.array_equal(lb, olb) and st.lf_steps == int(ost.sum()), (layout, sigma, flags)
        finally:
            del fm.options["kernel_select"]


def test_exact_search_edge_cases():
    text = make_text(5000, 5, seed=9)
    ox = fo.OraIndex.build("IB16", 5, [text], 16, False)
    gx = gpu_index(ox)
    # empty batch
    lb, ln = fm.search_no_errors.search(gx, [])
    assert lb.size == 0 and ln.size == 0
    # ragged: empty query (full interval), 1 symbol, longer than the text, maximum symbol, delimiter symbol
    queries = [[], [1], text[10:11], text[100:1200], np.concatenate([text, text]), [4, 4, 4, 4], [0], text[-30:], text[:30]]
    qbuf, qoff = fm.flatten(queries)
    lb, ln = fm.search_no_errors.search(gx, (qbuf, qoff))
    olb, oln = ox.search_exact(qbuf, qoff)
    assert np.array_equal(lb, olb) and np.array_equal(ln, oln)
    assert (lb[0], ln[0]) == (0, ox.n)
    # a symbol outside the alphabet can not occur: empty interval (the reference's behaviour is undefined there)
    lb, ln = fm.search_no_errors.search(gx, [[1, 2, 9, 1]])
    assert ln[0] == 0
    # device-resident queries and results give the same answer as host buffers
    dq, do = fm.DeviceBuffer.from_array(qbuf), fm.DeviceBuffer.from_array(qoff)
    dlb, dln = fm.DeviceBuffer(8 * len(queries)), fm.DeviceBuffer(8 * len(queries))
    fm.search_no_errors.search(gx, (dq, do), out=(dlb, dln))
    capi.check(capi.lib().fmgpu_synchronize(None))
    assert np.array_equal(dlb.to_array(np.uint64, len(queries)), olb) and np.array_equal(dln.to_array(np.uint64, len(queries)), oln)
    # unaligned query buffer start
    off = np.zeros(len(qbuf) + 3, dtype=np.uint8); off[3:] = qbuf
    dbig = fm.DeviceBuffer.from_array(off)
    class View:  # (ptr, nbytes) device view starting 3 bytes in
        ptr, nbytes = dbig.ptr + 3, len(qbuf)
    fm.search_no_errors.search(gx, (View, do), out=(dlb, dln))
    capi.check(capi.lib().fmgpu_synchronize(None))
    assert np.array_equal(dlb.to_array(np.uint64, len(queries)), olb)


@pytest.mark.parametrize("layout,sigma,kstep", [("IB16", 5, 2), ("IB16", 5, 3), ("IB16A", 4, 2), ("EPRV2_16", 5, 3), ("WAVELET", 5, 2), ("IB16", 6, 3), ("WAVELET", 28, 1), ("IB16", 28, 1)])
def test_exact_search_with_kstep_accelerator(layout, sigma, kstep):
    """fmgpu_index_accelerate: same cursors (also for misses: lb/len of the step that emptied the interval) and same step counts"""
    rng = np.random.default_rng(kstep + sigma)
    base = rng.integers(1, sigma, size=1500, dtype=np.uint8)
    seqs = [np.concatenate([base, base[200:700]]), rng.integers(1, sigma, size=900, dtype=np.uint8), np.array([1, 1, 2], dtype=np.uint8)]
    ox = fo.OraIndex.build(layout, sigma, seqs, 8, False)
    gx = gpu_index(ox).accelerate(kstep)
    queries = []
    for i in range(900):
        s = seqs[i % 2]; m = int(rng.integers(1, 80)); p = int(rng.integers(0, len(s) - m)); q = s[p: p + m].copy()
        if i % 3 == 0:
            q[int(rng.integers(0, m))] = rng.integers(1, sigma)
        queries.append(q)
    queries += [[], [1], [0], [1, 0, 1, 1], [sigma - 1] * 7, [1, 9, 1, 1, 1, 1]]
    qbuf, qoff = fm.flatten(queries)
    lb, ln, st = fm.search_no_errors.search(gx, (qbuf, qoff), want_stats=True)
    valid = [i for i, q in enumerate(queries) if all(c < sigma for c in q)]
    vb, vo = fm.flatten([queries[i] for i in valid])
    olb, oln, ost = ox.search_exact(vb, vo, want_steps=True)
    assert np.array_equal(lb[valid], olb) and np.array_equal(ln[valid], oln)
    assert ln[-1] == 0
    for lut_len, walk, ks in ((4, False, kstep), (0, True, kstep), (5, True, kstep), (3, True, 1), (2, False, 1), (4, 2, kstep), (0, 2, 1)):   # suffix table / walk table, with and without the k-step table
        gx.accelerate(ks, lut_len=lut_len, walk=walk)
        lb3, ln3, st3 = fm.search_no_errors.search(gx, (qbuf, qoff), want_stats=True)
        assert np.array_equal(lb3, lb) and np.array_equal(ln3, ln) and st3.lf_steps == st.lf_steps, (lut_len, walk, ks)
    gx.accelerate(0)
    lb2, ln2 = fm.search_no_errors.search(gx, (qbuf, qoff))
    assert np.array_equal(lb2, lb) and np.array_equal(ln2, ln)


@pytest.mark.parametrize("sigma,walk", [(5, 2), (5, True), (4, 2), (6, True)])
def test_exact_search_in_repeats_and_runs(sigma, walk):
    """reads that never reach one row — inside exact and slightly diverged repeat copies, tandem arrays and runs of one symbol — through every table
    combination of the exact search: cursors and step counts equal the CPU walk, also where the interval shrinks in the middle of a table
    stretch, at delimiters and at the text ends"""
    rng = np.random.default_rng(40 + sigma)
    unit = rng.integers(1, sigma, size=400, dtype=np.uint8)
    copies = []
    for k in range(7):
        cp = unit.copy()
        for _ in range(k):                                      # copy k differs from the consensus in k places
            cp[int(rng.integers(0, cp.size))] = rng.integers(1, sigma)
        copies.append(cp); copies.append(rng.integers(1, sigma, size=int(rng.integers(5, 60)), dtype=np.uint8))
    tandem = np.tile(rng.integers(1, sigma, size=7, dtype=np.uint8), 90)
    seqs = [np.concatenate(copies), np.concatenate([np.full(700, 1, dtype=np.uint8), tandem, unit[:250]]), np.full(130, 2, dtype=np.uint8)]
    ox = fo.OraIndex.build("IB16", sigma, seqs, 8, False)
    gx = gpu_index(ox)
    queries = []
    for i in range(1500):
        sq = seqs[i % 3]; m = int(rng.integers(20, 121)); m = min(m, len(sq)); p = int(rng.integers(0, len(sq) - m + 1)); q = sq[p: p + m].copy()
        if i % 5 == 0:
            q[int(rng.integers(0, m))] = rng.integers(1, sigma)
        queries.append(q)
    queries += [np.full(101, 1, dtype=np.uint8), np.full(131, 2, dtype=np.uint8), np.tile(tandem[:7], 15)[:101]]
    qbuf, qoff = fm.flatten(queries)
    olb, oln, ost = ox.search_exact(qbuf, qoff, want_steps=True)
    assert (oln > 1).sum() > 500                              # most of these reads end on several rows
    for kstep, lut_len in ((3, 4), (2, 0), (1, 3)):
        gx.accelerate(kstep, lut_len=lut_len, walk=walk)
        lb, ln, st = fm.search_no_errors.search(gx, (qbuf, qoff), want_stats=True)
        assert np.array_equal(lb, olb) and np.array_equal(ln, oln) and st.lf_steps == int(ost.sum()), (kstep, lut_len, walk)


@pytest.mark.parametrize("layout,sigma", [("IB16", 256), ("IB16", 21), ("EPRV5", 6), ("FBV_512_64K", 255)])
def test_exact_search_tables_other_alphabets(layout, sigma):
    """interval table and walk table with symbols wider than 2 bits (walk length 32 / bit_width(sigma-2): 4 symbols at sigma = 256)"""
    rng = np.random.default_rng(sigma)
    hi = min(sigma, 9)
    base = rng.integers(1, hi, size=2500, dtype=np.uint8)
    seqs = [np.concatenate([base, base[500:900]]), rng.integers(1, sigma, size=600, dtype=np.uint8)]
    ox = fo.OraIndex.build(layout, sigma, seqs, 8, False)
    gx = gpu_index(ox)
    queries = []
    for i in range(600):
        s = seqs[i % 2]; m = int(rng.integers(1, 70)); p = int(rng.integers(0, len(s) - m)); q = s[p: p + m].copy()
        if i % 4 == 0:
            q[int(rng.integers(0, m))] = rng.integers(1, sigma)
        queries.append(q)
    queries += [[], [0], [sigma - 1] * 5]
    qbuf, qoff = fm.flatten(queries)
    olb, oln, ost = ox.search_exact(qbuf, qoff, want_steps=True)
    for kstep, lut_len, walk in ((1, 2, True), (1, 0, True), (1, 1, False), (2 if sigma <= 6 else 1, 2, True), (1, 1, 2)):
        gx.accelerate(kstep, lut_len=lut_len, walk=walk)
        lb, ln, st = fm.search_no_errors.search(gx, (qbuf, qoff), want_stats=True)
        assert np.array_equal(lb, olb) and np.array_equal(ln, oln) and st.lf_steps == int(ost.sum()), (kstep, lut_len, walk)
    with pytest.raises(fm.FmgpuError):
        gx.accelerate(1, lut_len=32 if sigma > 6 else 33)         # table too large / out of range


@pytest.mark.parametrize("seed", list(range(24)) + [462, 772])
def test_exact_search_randomised_layouts_and_tables(seed):
    """random layout, alphabet, text, table combination and ragged queries (with symbols outside the alphabet and delimiters): cursors and
    step counts of the exact search equal the CPU walk"""
    rng = np.random.default_rng(900 + seed)
    layout = LAYOUTS[int(rng.integers(0, len(LAYOUTS)))]
    sigma = int(rng.choice([3, 4, 5, 5, 6, 8, 21, 28]))
    hi = min(sigma, 9)
    base = rng.integers(1, hi, size=int(rng.integers(300, 3000)), dtype=np.uint8)
    seqs = [np.concatenate([base, base[50:250]]), rng.integers(1, sigma, size=int(rng.integers(1, 500)), dtype=np.uint8)]
    ox = fo.OraIndex.build(layout, sigma, seqs, 4, False)
    if int(ox.C[-1]) != ox.n:
        # the reference's own defect: rank(n, c) double-counts when n is a multiple of the super-block period of InterleavedEPR* / EPRV3_*
        # (DESIGN.md §2), and computeC (utils.h:199-206) reads exactly that — such a reference index has a corrupt C; it is refused
        with pytest.raises(fm.FmgpuError):
            gpu_index(ox)
        return
    gx = gpu_index(ox)
    R = sigma - 1
    kstep = int(rng.integers(1, 5))
    while kstep > 1 and R ** kstep > 255:
        kstep -= 1
    lut_len = int(rng.integers(0, 6))
    while lut_len > 0 and R ** lut_len > (1 << 22):
        lut_len -= 1
    gx.accelerate(kstep, lut_len=lut_len, walk=int(rng.integers(0, 3)))
    queries = []
    for i in range(500):
        s = seqs[i % 2]; m = int(rng.integers(1, max(2, min(90, len(s))))); p = int(rng.integers(0, len(s) - m + 1)); q = s[p: p + m].copy()
        r = int(rng.integers(0, 10))
        if r == 0: q[int(rng.integers(0, m))] = rng.integers(1, sigma)
        elif r == 1: q[int(rng.integers(0, m))] = 0
        elif r == 2: q[int(rng.integers(0, m))] = min(255, sigma + int(rng.integers(0, 3)))
        queries.append(q)
    queries.append(np.array([], dtype=np.uint8))
    qbuf, qoff = fm.flatten(queries)
    lb, ln, st = fm.search_no_errors.search(gx, (qbuf, qoff), want_stats=True)
    valid = [i for i, q in enumerate(queries) if all(c < sigma for c in q)]
    vb, vo = fm.flatten([queries[i] for i in valid])
    olb, oln, ost = ox.search_exact(vb, vo, want_steps=True)
    assert np.array_equal(lb[valid], olb) and np.array_equal(ln[valid], oln), (seed, layout, sigma, kstep, lut_len)
    invalid = [i for i in range(len(queries)) if i not in set(valid)]
    assert not ln[invalid].any()                               # a symbol outside the alphabet cannot occur


def test_queries_too_long_for_lds_staging():
    """queries beyond the LDS staging budget (48 KB per block: > 384 DNA symbols, > 192 symbols of a larger alphabet) are read from global
    memory by the same kernels: exact search with every table, the table-driven k-mismatch and edit-distance kernels, the general ones"""
    rng = np.random.default_rng(3)
    for sigma, L in ((5, 700), (28, 300), (5, 300), (5, 336), (5, 380), (28, 170)):       # around the budget as well
        hi = min(sigma, 9)
        base = rng.integers(1, hi, size=6000, dtype=np.uint8)
        seqs = [np.concatenate([base, base[1000:3000]]), rng.integers(1, hi, size=900, dtype=np.uint8)]
        ox = fo.OraIndex.build("IB16", sigma, seqs, 8, True)
        gx = gpu_index(ox)
        gx.accelerate(3 if sigma == 5 else 1, lut_len=4, walk=2).accelerate_search(4, 3)
        queries = []
        for i in range(120):
            p = int(rng.integers(0, len(seqs[0]) - L)); q = seqs[0][p: p + L].copy()
            for _ in range(i % 3):
                q[int(rng.integers(0, L))] = rng.integers(1, hi)
            queries.append(q)
        qbuf, qoff = fm.flatten(queries)
        lb, ln, st = fm.search_no_errors.search(gx, (qbuf, qoff), want_stats=True)
        olb, oln, ost = ox.search_exact(qbuf, qoff, want_steps=True)
        assert np.array_equal(lb, olb) and np.array_equal(ln, oln) and st.lf_steps == int(ost.sum())
        sch = fm.search_scheme.h2(4, 0, 2)
        for edit in (False, True):
            hits, st = fm.search_ng26.search(gx, (qbuf[: qoff[30]], qoff[:31]) if edit else (qbuf, qoff), sch, want_stats=True, edit=edit, capacity=1 << 20)
            ohits, _, nodes = ox.search_ng26(qbuf[: qoff[30]], qoff[:31], sch, edit=True, cap=1 << 20) if edit else ox.search_ng26(qbuf, qoff, sch, cap=1 << 20)
            assert same_hits(hits, ohits) and st.lf_steps == nodes, (sigma, edit)
        for other in ("WAVELET", "EPR16", "EPRV2_16"):                 # the kernels of the other device formats, without tables
            oy = fo.OraIndex.build(other, sigma, seqs, 8, False)
            lb2, ln2 = fm.search_no_errors.search(gpu_index(oy), (qbuf, qoff))
            assert np.array_equal(lb2, olb) and np.array_equal(ln2, oln), other
        ragged = [q[: L - (i % 5)] for i, q in enumerate(queries[:40])]
        rb, ro = fm.flatten(ragged)
        assert same_hits(fm.search_ng26.search(gx, (rb, ro), sch, capacity=1 << 20), ox.search_ng26(rb, ro, sch, cap=1 << 20)[0])
        assert same_hits(fm.search_backtracking.search(gx, (rb[: ro[10]], ro[:11]), 1), ox.search_backtracking(rb[: ro[10]], ro[:11], 1)[0])


def test_exact_search_tiny_indices():
    for seqs in ([[1]], [[]], [[1], [1], [2, 1]], [[3] * 70]):
        ox = fo.OraIndex.build("IB16", 5, seqs, 1, True)
        gx = gpu_index(ox)
        queries = [[1], [1, 1], [2, 1], [3] * 64, [3] * 70, [3] * 71, []]
        qbuf, qoff = fm.flatten(queries)
        lb, ln = fm.search_no_errors.search(gx, (qbuf, qoff))
        olb, oln = ox.search_exact(qbuf, qoff)
        assert np.array_equal(lb, olb) and np.array_equal(ln, oln), seqs


def test_plumbing_config0():
    """BASELINE.json configs[0]: 1 MB random DNA (sigma 5), 10k x 31 bp exact, FMIndex<InterleavedBitvector16>"""
    text = make_text(1_000_000, 5, seed=42)
    ox = fo.OraIndex.build("IB16", 5, [text], 16, False)
    gx = gpu_index(ox)
    reads = sample_reads(text, 10_000, 31, seed=1, mutate=1)
    qbuf, qoff = fm.flatten(reads)
    lb, ln = fm.search_no_errors.search(gx, (qbuf, qoff))
    olb, oln = ox.search_exact(qbuf, qoff, nthreads=4)
    assert np.array_equal(lb, olb) and np.array_equal(ln, oln)
    assert int((ln > 0).sum()) >= 5000


@pytest.mark.parametrize("name", ["config0_ib16", "protein_wavelet", "epr16", "eprv2_16", "eprv5", "ibp16"])
def test_search_intervals_from_the_real_reference_rank(name):
    """tests/golden/ref_search_intervals.npz — (lb, len) computed by backward search over the REAL reference's rank functions
    (make_golden.py::ref_search_intervals; config0_ib16 = BASELINE.json configs[0]) — reproduced by the HIP path, plain and with the tables"""
    from tests.golden.make_golden import REF_SEARCH_CASES
    layout, sigma, tn, seed, nreads, rl = REF_SEARCH_CASES[name]
    want = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_search_intervals.npz"))[name]
    text = make_text(tn, sigma, seed=seed)
    gx = gpu_index(fo.OraIndex.build(layout, sigma, [text], 16, False))
    qbuf, qoff = fm.flatten(sample_reads(text, nreads, rl, seed=1, mutate=1, sigma=sigma))
    for tables in (False, True):
        if tables:
            gx.accelerate(3 if sigma == 5 else 1, lut_len=8 if sigma == 5 else 3, walk=2)
        lb, ln = fm.search_no_errors.search(gx, (qbuf, qoff))
        assert np.array_equal(lb, want[:, 0]) and np.array_equal(ln, want[:, 1]), (name, tables)


# ------------------------------------------------------------------------------------------------ k-mismatch
@pytest.mark.parametrize("k", [1, 2, 3])
def test_scheme_search_matches_reference_order(k):
    """search_ng26<Edit=false> with h2(k+2, 0, k): same hits in the same callback order, same node count"""
    seqs = repeat_text(10 + k)
    ox = fo.OraIndex.build("IB16", 5, seqs, 4, True)
    gx = gpu_index(ox)
    queries = mutated_queries(seqs, 500, k + 2, 60, k + 1, seed=3 + k)
    qbuf, qoff = fm.flatten(queries)
    sch = fm.search_scheme.h2(k + 2, 0, k)
    hits, st = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True)
    ohits, qc, nodes = ox.search_ng26(qbuf, qoff, sch)
    assert same_hits(hits, ohits)
    assert st.lf_steps == nodes and st.hits == len(ohits)
    # scheme order -> DFS order: seq numbers are 0..count-1 per query
    for q in range(len(queries)):
        assert hits[hits["qidx"] == q]["seq"].tolist() == list(range(int(qc[q])))


@pytest.mark.parametrize("k,length", [(1, 20), (2, 31), (2, 101), (3, 64), (2, 151), (1, 40)])
def test_scheme_search_equal_length_fast_path(k, length):
    """equal-length batches take the table-driven kernel (k_scheme_fast): same hits, order and node count as the reference walk"""
    seqs = repeat_text(40 + k, n=6000)
    ox = fo.OraIndex.build("IB16", 5, seqs, 4, True)
    gx = gpu_index(ox)
    queries = mutated_queries(seqs, 1500, length, length + 1, k + 1, seed=11 + k)
    assert len({len(q) for q in queries}) == 1
    qbuf, qoff = fm.flatten(queries)
    for accel in (None, (0, True), (3, False), (4, True), (2, True), (0, 2), (4, 3), (3, 2)):     # plain, walk tables only, prefix table only, both; 16-symbol walk tables
        if accel is not None:
            gx.accelerate_search(*accel)
        for sch in (fm.search_scheme.h2(k + 2, 0, k), fm.search_scheme.pigeon_opt(0, k), fm.search_scheme.backtracking(2, 0, k)):
            hits, st = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True)
            ohits, qc, nodes = ox.search_ng26(qbuf, qoff, sch)
            assert same_hits(hits, ohits) and st.lf_steps == nodes, (accel, k, length)
        hits = fm.search_ng26.search(gx, (qbuf, qoff), fm.search_scheme.h2(k + 2, 0, k), n=2)
        assert same_hits(hits, ox.search_ng26(qbuf, qoff, fm.search_scheme.h2(k + 2, 0, k), max_hits=2)[0])
    fm.options["kernel_select"] = "2"                       # force the generic kernel: both kernels agree
    try:
        hits2 = fm.search_ng26.search(gx, (qbuf, qoff), fm.search_scheme.h2(k + 2, 0, k))
    finally:
        del fm.options["kernel_select"]
    assert same_hits(hits2, ox.search_ng26(qbuf, qoff, fm.search_scheme.h2(k + 2, 0, k))[0])


@pytest.mark.parametrize("mix", ["all_heavy", "none_heavy", "some_heavy"])
def test_heavy_reads_first_hand_out_order(mix):
    """batches of 2^16 reads and more are handed out with the reads of high-copy repeats in front (k_heavy_flags / k_heavy_flags_plain, decided on a
    sample): records, order and node counts are those of the CPU walk whatever the mix — with the prefix table, on the plain index, with the
    order switched off, for Hamming and edit distance"""
    rng = np.random.default_rng(5)
    unit = rng.integers(1, 5, size=300, dtype=np.uint8)
    rep = np.concatenate([np.concatenate([unit, rng.integers(1, 5, size=7, dtype=np.uint8)]) for _ in range(80)])      # 80 copies: intervals of 80 rows (> the 64 that count as heavy)
    uniq = rng.integers(1, 5, size=30000, dtype=np.uint8)
    seqs = [rep, uniq]
    ox = fo.OraIndex.build("IB16", 5, seqs, 8, True)
    L, nq = 40, 70_000
    src = {"all_heavy": [0], "none_heavy": [1], "some_heavy": [1] * 19 + [0]}[mix]
    reads = np.empty((nq, L), dtype=np.uint8)
    for i in range(nq):
        sq = seqs[src[i % len(src)]]
        p = int(rng.integers(0, len(sq) - L))
        reads[i] = sq[p: p + L]
    flip = rng.integers(0, nq, size=nq // 3)
    reads[flip, rng.integers(0, L, size=flip.size)] = rng.integers(1, 5, size=flip.size)
    qbuf, qoff = reads.reshape(-1), np.arange(nq + 1, dtype=np.uint64) * L
    sch = fm.search_scheme.h2(3, 0, 1)
    want, _, wnodes = ox.search_ng26(qbuf, qoff, sch, cap=1 << 23)
    fm.options["lf_table"] = "0"
    try:
        gx = gpu_index(ox)                                    # plain index: the sample and the flags come from 16 LF steps on the blocks
    finally:
        del fm.options["lf_table"]
    hits, st = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, capacity=1 << 23)
    assert same_hits(hits, want) and st.lf_steps == wnodes
    gx.accelerate_lf(True); gx.accelerate_search(8, 1)        # ... from the 8-symbol prefix table
    hits, st = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, capacity=1 << 23)
    assert same_hits(hits, want) and st.lf_steps == wnodes
    ewant, _, enodes = ox.search_ng26(qbuf[: 66_000 * L], qoff[: 66_001], sch, edit=True, cap=1 << 23)
    ehits, est = fm.search_ng26.search(gx, (qbuf[: 66_000 * L], qoff[: 66_001]), sch, want_stats=True, edit=True, capacity=1 << 23)
    assert same_hits(ehits, ewant) and est.lf_steps == enodes
    fm.options["heavy_first"] = "0"
    try:
        hits, st = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, capacity=1 << 23)
    finally:
        del fm.options["heavy_first"]
    assert same_hits(hits, want) and st.lf_steps == wnodes
    fm.options["kernel_select"] = "2"                       # the general kernels order their hand-out too (16 LF steps per read on any layout)
    try:
        hits, st = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, capacity=1 << 23)
        ehits, est = fm.search_ng26.search(gx, (qbuf[: 66_000 * L], qoff[: 66_001]), sch, want_stats=True, edit=True, capacity=1 << 23)
    finally:
        del fm.options["kernel_select"]
    assert same_hits(hits, want) and st.lf_steps == wnodes
    assert same_hits(ehits, ewant) and est.lf_steps == enodes
    ex = fm.search_scheme.expand(sch, L)                     # search_ng21 too
    h21, st21 = fm.search_ng21.search(gx, (qbuf[: 66_000 * L], qoff[: 66_001]), ex, want_stats=True, capacity=1 << 23)
    o21, _, n21 = ox.search_ng21(qbuf[: 66_000 * L], qoff[: 66_001], ex, cap=1 << 23)
    assert same_hits(h21, o21) and st21.lf_steps == n21
    if mix == "some_heavy":                                   # ragged lengths, 64-bit rows
        rl = rng.integers(14, L + 1, size=nq)
        rq = np.concatenate([reads[i, : rl[i]] for i in range(nq)])
        ro = np.concatenate([[0], np.cumsum(rl)]).astype(np.uint64)
        rwant, _, rnodes = ox.search_ng26(rq, ro, sch, cap=1 << 23)
        fm.options["force_wide"] = "1"
        try:
            wx = gpu_index(ox)
        finally:
            del fm.options["force_wide"]
        assert wx.row_bits == 64
        rhits, rst = fm.search_ng26.search(wx, (rq, ro), sch, want_stats=True, capacity=1 << 23)
        assert same_hits(rhits, rwant) and rst.lf_steps == rnodes



def test_ragged_batch_in_launches_side_by_side():
    """A ragged batch through the equal-length kernels is one launch per read length, and up to eight of them run side by side — each on a stream, a share of the grid, a stretch of
    the frame stacks, a hand-out counter and a board of its own (DfsWorkspace::fork / join).  45 000 reads of nine lengths in the caller's (mixed) order, on a repeat-rich text: records in
    callback order and node counts equal the CPU walk — Hamming on the plain index (k_scheme_lean) and with LF tables (k_scheme_fast), edit distance (k_scheme_fast_edit) —, and the
    same one launch after the other (FMGPU_SEL_NO_BOARD)"""
    seqs = repeat_text(91, n=12000) + [np.tile(np.array([1, 1, 2, 3], dtype=np.uint8), 400)]
    ox = fo.OraIndex.build("IB16", 5, seqs, 8, True)
    rng = np.random.default_rng(5)
    lengths = [24, 25, 31, 40, 41, 42, 57, 64, 80]
    queries = []
    for L in lengths:
        queries += mutated_queries([q for q in seqs if len(q) > L], 5000, L, L + 1, 3, seed=100 + L)
    order = rng.permutation(len(queries))
    queries = [queries[int(t)] for t in order]
    qbuf, qoff = fm.flatten(queries)
    sch = fm.search_scheme.h2(4, 0, 2)
    want = {False: ox.search_ng26(qbuf, qoff, sch, cap=1 << 25, nthreads=8, records=True), True: None}
    eq, eo = fm.flatten([q for q in queries if len(q) in (31, 40)])       # (edit distance: two lengths, two launches side by side; the CPU walk of it runs on one thread)
    want[True] = ox.search_ng26(eq, eo, sch, edit=True, cap=1 << 25)
    for lf in (0, 1):
        with fm.options(lf_table=lf):
            gx = gpu_index(ox)
        for sel in (0, capi.SEL_NO_BOARD):
            with fm.options(kernel_select=sel):
                hits, st = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, capacity=1 << 25)
                ehits, est = fm.search_ng26.search(gx, (eq, eo), sch, want_stats=True, edit=True, capacity=1 << 25)
            assert same_hits(hits, want[False][0]) and st.lf_steps == want[False][2], (lf, sel)
            assert same_hits(ehits, want[True][0]) and est.lf_steps == want[True][2], (lf, sel)


@pytest.mark.parametrize("k,length", [(1, 20), (2, 31), (2, 101), (2, 151), (2, 255), (3, 64), (0, 40)])
def test_lean_kernel_on_the_plain_index(k, length):
    """equal-length batches on a BiFMIndex<5> WITHOUT any table take k_scheme_lean (top frame of the stack cached in LDS and refilled by LDS-DMA, hit ring
    per wave, 2-bit staged reads): records in callback order and node counts equal the CPU walk and k_scheme_fast<PLAIN> (FMGPU_SEL_NO_LEAN) —
    on a repeat-rich text (deep stacks, thousands of hits per read: ring flushes, work sharing), with reads that hold delimiters and bytes outside
    the alphabet (read from global memory), and with more hits than the caller's buffer holds"""
    seqs = repeat_text(70 + k, n=9000) + [np.tile(np.array([1, 1, 1, 2], dtype=np.uint8), 300), np.full(700, 3, dtype=np.uint8)]
    ox = fo.OraIndex.build("IB16", 5, seqs, 4, True)
    fm.options["lf_table"] = "0"
    try:
        gx = gpu_index(ox)
    finally:
        del fm.options["lf_table"]
    queries = mutated_queries([q for q in seqs if len(q) > length], 3000, length, length + 1, k + 1, seed=21 + k)
    rng = np.random.default_rng(k)
    for i in range(0, len(queries), 97):                      # delimiters inside reads, at the ends and inside: the lane reads such a read from global memory
        queries[i][int(rng.integers(0, length))] = 0
    queries[5][0] = 0; queries[6][length - 1] = 0
    assert len({len(q) for q in queries}) == 1
    qbuf, qoff = fm.flatten(queries)
    schemes = [fm.search_scheme.h2(k + 2, 0, k), fm.search_scheme.pigeon_opt(0, k)] if k else [fm.search_scheme.backtracking(1, 0, 0)]
    for sch in schemes:
        ohits, qc, nodes = ox.search_ng26(qbuf, qoff, sch, cap=1 << 24)
        hits, st = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, capacity=1 << 24)
        assert same_hits(hits, ohits) and st.lf_steps == nodes, (k, length)
        for flags in (1 << 30, 1 << 29):                           # k_scheme_fast<PLAIN>; k_scheme_lean on the Format A blocks (the default reads the dense Format D)
            fm.options["kernel_select"] = str(flags)
            try:
                hits2, st2 = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, capacity=1 << 24)
            finally:
                del fm.options["kernel_select"]
            assert same_hits(hits2, ohits) and st2.lf_steps == nodes, flags
        if len(ohits) > 10:                                       # a buffer that is too small: FMGPU_ERR_CAPACITY with the exact count (the wrapper then asks again with that capacity)
            assert same_hits(fm.search_ng26.search(gx, (qbuf, qoff), sch, capacity=len(ohits) // 2), ohits)
    # bytes outside the alphabet are outside the reference's domain (it indexes an array of sigma cursors with them); the kernels treat them as "matches nothing":
    # the three Hamming kernels agree on it
    odd = [q.copy() for q in queries[:600]]
    for i, q in enumerate(odd):
        q[int(rng.integers(0, length))] = [5, 9, 255, 15][i % 4]
    qb2, qo2 = fm.flatten(odd)
    got = []
    for flags in (None, 1 << 30, 2):
        if flags is not None:
            fm.options["kernel_select"] = str(flags)
        try:
            got.append(fm.search_ng26.search(gx, (qb2, qo2), schemes[0], want_stats=True, capacity=1 << 22))
        finally:
            fm.options.pop("kernel_select")
    assert same_hits(got[0][0], got[1][0]) and same_hits(got[0][0], got[2][0]) and got[0][1].lf_steps == got[1][1].lf_steps == got[2][1].lf_steps
    # a prefix table in front of the blocks (fmgpu_index_accelerate_search(h, L, 0): no LF table, so the lean kernel still serves the batch): a search whose first part is
    # exact and longer than L symbols starts from the entry of its first L symbols — records, callback order and node counts unchanged (reads with a delimiter
    # or a foreign byte walk from the start); the table switched off by selection gives the same
    for lut_len in ((2, 5, 9) if length >= 50 else (2, 5)):
        gx.accelerate_search(lut_len, 0)
        assert bool(gx.formats & capi.FMT_PREFIX) and not (gx.formats & capi.FMT_LF)
        for sch in schemes:
            ohits, qc, nodes = ox.search_ng26(qbuf, qoff, sch, cap=1 << 24)
            hits, st = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, capacity=1 << 24)
            assert same_hits(hits, ohits) and st.lf_steps == nodes, (k, length, lut_len)
            if k <= 2:                                              # (the lean kernel serves these; it reports the nodes its table entries stood for)
                assert 0 < st.table_steps <= nodes and st.table_steps <= lut_len * len(queries) * int(np.asarray(sch[0]).shape[0]), (k, length, lut_len, st.table_steps)
            with fm.options(kernel_select=capi.SEL_NO_PREFIX_TABLE):
                hits2, st2 = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, capacity=1 << 24)
            assert same_hits(hits2, ohits) and st2.lf_steps == nodes and st2.table_steps == 0
        g0 = fm.search_ng26.search(gx, (qb2, qo2), schemes[0], want_stats=True, capacity=1 << 22)
        assert same_hits(g0[0], got[0][0]) and g0[1].lf_steps == got[0][1].lf_steps
    gx.accelerate_search(0, 0)
    # the poly-A / satellite reads alone: 64 lanes of a wave all deep in one repeat
    sat = [seqs[3][i: i + length] for i in range(0, 400)] + [seqs[4][:length]] * 200
    if all(len(q) == length for q in sat):
        qbuf, qoff = fm.flatten(sat)
        sch = schemes[0]
        ohits, _, nodes = ox.search_ng26(qbuf, qoff, sch, cap=1 << 25)
        hits, st = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, capacity=1 << 25)
        assert same_hits(hits, ohits) and st.lf_steps == nodes


def test_equal_length_batch_without_a_step_table():
    """an equal-length batch of 2^16 reads whose scheme has NO per-step table — reads shorter than the scheme has parts (skipped, expand.h:325-327), and
    reads so long that the three tables would not fit the LDS (3 searches x 911 steps) — on an index WITH an 8-symbol prefix table: the host must not
    look at a bucket that was never made (it did, round 2) and the general kernel serves the batch"""
    rng = np.random.default_rng(17)
    text = rng.integers(1, 5, size=40_000, dtype=np.uint8)
    ox = fo.OraIndex.build("IB16", 5, [text], 8, True)
    gx = gpu_index(ox)
    gx.accelerate_lf(True); gx.accelerate_search(8, 1)
    sch = fm.search_scheme.h2(4, 0, 2)                            # 3 searches, 4 parts
    nq = 1 << 16
    short = rng.integers(1, 5, size=(nq, 3), dtype=np.uint8)      # m = 3 < P = 4
    qbuf, qoff = short.reshape(-1), np.arange(nq + 1, dtype=np.uint64) * 3
    hits, st = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True)
    assert len(hits) == 0 and len(ox.search_ng26(qbuf, qoff, sch)[0]) == 0
    L = 912                                                       # 3 x 913 > 2730 words
    starts = rng.integers(0, len(text) - L, size=nq)
    reads = text[starts[:, None] + np.arange(L)[None, :]].copy()
    flip = rng.integers(0, nq, size=nq // 2)
    reads[flip, rng.integers(0, L, size=flip.size)] = rng.integers(1, 5, size=flip.size)
    qbuf, qoff = reads.reshape(-1), np.arange(nq + 1, dtype=np.uint64) * L
    want, _, wnodes = ox.search_ng26(qbuf, qoff, sch, cap=1 << 22)
    hits, st = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, capacity=1 << 22)
    assert same_hits(hits, want) and st.lf_steps == wnodes


def test_scheme_search_ragged_batch_in_length_buckets():
    """a large ragged batch is sorted by length on the device and runs the table-driven kernel once per length: same records, order and
    node count as the CPU walk and as the general kernel; queries shorter than the number of parts are skipped in both"""
    seqs = repeat_text(77, n=8000)
    ox = fo.OraIndex.build("IB16", 5, seqs, 4, True)
    gx = gpu_index(ox)
    gx.accelerate_search(4, 3)
    rng = np.random.default_rng(5)
    queries = []
    s0 = seqs[0]
    for i in range(70_000):
        m = int(rng.integers(24, 32)) if i % 1000 else 2          # a few too-short ones
        p = int(rng.integers(0, len(s0) - m)); q = s0[p: p + m].copy()
        if i % 3: q[int(rng.integers(0, m))] = rng.integers(1, 5)
        queries.append(q)
    qbuf, qoff = fm.flatten(queries)
    sch = fm.search_scheme.h2(3, 0, 1)
    hits, st = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, capacity=1 << 22)
    keep = [i for i, q in enumerate(queries) if len(q) >= 3]
    kb, ko = fm.flatten([queries[i] for i in keep])
    ohits, _, nodes = ox.search_ng26(kb, ko, sch, cap=1 << 22)
    ohits = ohits.copy(); ohits["qidx"] = np.array(keep, dtype=np.uint64)[ohits["qidx"].astype(np.int64)]
    assert same_hits(hits, ohits) and st.lf_steps == nodes
    fm.options["kernel_select"] = "64"                      # no length buckets: the general kernel
    try:
        hits2, st2 = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, capacity=1 << 22)
    finally:
        del fm.options["kernel_select"]
    assert hits2.tobytes() == hits.tobytes() and st2.lf_steps == nodes
    assert same_hits(fm.search_ng26.search(gx, (qbuf, qoff), sch, n=1, capacity=1 << 22), ox_n1(ox, kb, ko, sch, keep))


def ox_n1(ox, kb, ko, sch, keep):
    h = ox.search_ng26(kb, ko, sch, max_hits=1, cap=1 << 22)[0].copy()
    h["qidx"] = np.array(keep, dtype=np.uint64)[h["qidx"].astype(np.int64)]
    return h


def test_scheme_search_variants():
    seqs = repeat_text(20)
    ox = fo.OraIndex.build("IB16", 5, seqs, 4, True)
    gx = gpu_index(ox)
    same_len = [q for q in mutated_queries(seqs, 2000, 31, 32, 2, seed=5)]
    qbuf, qoff = fm.flatten(same_len)
    for sch in (fm.search_scheme.pigeon_opt(0, 2), fm.search_scheme.pigeon_trivial(0, 1), fm.search_scheme.backtracking(3, 0, 2),
                fm.search_scheme.h2(5, 0, 2), fm.search_scheme.h2(4, 1, 2), fm.search_scheme.limitToHamming(fm.search_scheme.h2(4, 0, 2))):
        hits = fm.search_ng26.search(gx, (qbuf, qoff), sch)
        ohits, _, _ = ox.search_ng26(qbuf, qoff, sch)
        assert same_hits(hits, ohits)
    # explicit (non-uniform) partition
    sch = fm.search_scheme.h2(4, 0, 2)
    part = np.array([5, 10, 9, 7], dtype=np.uint64)
    hits = fm.search_ng26.search(gx, (qbuf, qoff), sch, partition=part)
    ohits, _, _ = ox.search_ng26(qbuf, qoff, sch, partition=part)
    assert same_hits(hits, ohits) and len(hits) > 0
    # search_n: stop after n rows (SearchNg26.h:407-423)
    for n in (1, 2, 5):
        hits = fm.search_ng26.search(gx, (qbuf, qoff), sch, n=n)
        ohits, _, _ = ox.search_ng26(qbuf, qoff, sch, max_hits=n)
        assert same_hits(hits, ohits)
    # capacity protocol: too small a buffer reports the needed size
    out = np.zeros(3, dtype=capi.HIT_DTYPE); cnt = C.c_uint64()
    pi, l, u = (np.ascontiguousarray(x, dtype=np.uint64) for x in sch)
    sc = capi.Scheme(); sc.n_searches, sc.n_parts = pi.shape
    sc.pi, sc.l, sc.u = (x.ctypes.data_as(capi.u64p) for x in (pi, l, u))
    rc = capi.lib().fmgpu_search_scheme(gx._h, capi.ptr(qbuf), capi.ptr(qoff), len(same_len), C.byref(sc), capi.UINT64_MAX,
                                        capi.ptr(out), 3, C.byref(cnt), None, None)
    full, _, _ = ox.search_ng26(qbuf, qoff, sch)
    assert rc == capi.FMGPU_ERR_CAPACITY and cnt.value == len(full)
    # ragged lengths with the uniform partition, including queries shorter than the number of parts (skipped)
    ragged = mutated_queries(seqs, 300, 2, 90, 2, seed=6) + [[1], [1, 2, 3]]
    qb, qo = fm.flatten(ragged)
    hits = fm.search_ng26.search(gx, (qb, qo), sch)
    keep = [i for i, q in enumerate(ragged) if len(q) >= 4]
    kb, ko = fm.flatten([ragged[i] for i in keep])
    ohits, _, _ = ox.search_ng26(kb, ko, sch)
    ohits = ohits.copy(); ohits["qidx"] = np.array(keep, dtype=np.uint64)[ohits["qidx"].astype(np.int64)]
    assert same_hits(hits, ohits)
    # unidirectional index is rejected like the reference (no extendRight on FMIndexCursor)
    fx = gpu_index(fo.OraIndex.build("IB16", 5, seqs, 4, False))
    with pytest.raises(fm.FmgpuError):
        fm.search_ng26.search(fx, (qbuf, qoff), sch)


@pytest.mark.parametrize("layout,sigma", [("IB16", 5), ("IBP16", 5), ("EPR16", 5), ("EPRV2_16", 5), ("WAVELET", 5), ("WAVELET", 28), ("IB16", 28), ("IB8", 6),
                                          ("EPRV5", 5), ("IEPRV7", 5), ("EPRV3_16", 28), ("EPRV4", 6), ("FBV_512_64K", 5), ("FBV_2048_64K", 28)])
def test_k_mismatch_other_layouts(layout, sigma):
    rng = np.random.default_rng(sigma)
    base = rng.integers(1, sigma, size=900, dtype=np.uint8)
    seqs = [np.concatenate([base, base[100:400]]), rng.integers(1, sigma, size=500, dtype=np.uint8)]
    ox = fo.OraIndex.build(layout, sigma, seqs, 4, True)
    gx = gpu_index(ox)
    queries = []
    for i in range(200):
        m = int(rng.integers(4, 30)); p = int(rng.integers(0, len(seqs[0]) - m)); q = seqs[0][p: p + m].copy()
        if i % 2:
            q[int(rng.integers(0, m))] = rng.integers(1, sigma)
        queries.append(q)
    qbuf, qoff = fm.flatten(queries)
    sch = fm.search_scheme.h2(3, 0, 1)
    assert same_hits(fm.search_ng26.search(gx, (qbuf, qoff), sch), ox.search_ng26(qbuf, qoff, sch)[0])
    assert same_hits(fm.search_backtracking.search(gx, (qbuf, qoff), 1), ox.search_backtracking(qbuf, qoff, 1)[0])


@pytest.mark.parametrize("layout,sigma", [("EPR16", 5), ("EPRV2_16", 5), ("EPRV2_8", 6), ("WAVELET", 5), ("WAVELET", 28), ("WAVELET", 256), ("EPR32", 21)])
def test_occurrence_table_expansion(layout, sigma):
    """fmgpu_index_accelerate(h, 1) on EPR / Wavelet indices: the searches read the expanded block table, results stay those of the
    reference layout (exact, search scheme on the table-driven and the generic kernel, backtracking, locate); dropping it again too"""
    rng = np.random.default_rng(sigma + len(layout))
    base = rng.integers(1, sigma, size=1300, dtype=np.uint8)
    seqs = [np.concatenate([base, base[100:500]]), rng.integers(1, sigma, size=700, dtype=np.uint8)]
    ox = fo.OraIndex.build(layout, sigma, seqs, 4, True)
    fm.options["expand_dna"] = "0"                          # (sigma = 5 strings get the expansion at creation unless told otherwise: below)
    try:
        gx = gpu_index(ox)
    finally:
        del fm.options["expand_dna"]
    before = gx.device_bytes
    gx.accelerate(1)
    assert gx.device_bytes > before
    if sigma == 5:
        auto = gpu_index(ox)                                  # the same index as it is created by default: expanded (and with the formats derived from the expansion)
        assert auto.device_bytes >= gx.device_bytes
        qb_, qo_ = fm.flatten(mutated_queries(seqs, 400, 24, 25, 2, seed=3, sigma=sigma))
        a_hits, a_st = fm.search_ng26.search(auto, (qb_, qo_), fm.search_scheme.h2(4, 0, 2), want_stats=True)
        o_hits, _, o_nodes = ox.search_ng26(qb_, qo_, fm.search_scheme.h2(4, 0, 2))
        assert same_hits(a_hits, o_hits) and a_st.lf_steps == o_nodes
        a_lb, a_ln, a_st = fm.search_no_errors.search(auto, (qb_, qo_), want_stats=True)
        o_lb, o_ln, o_steps = ox.search_exact(qb_, qo_, want_steps=True)
        assert np.array_equal(a_lb, o_lb) and np.array_equal(a_ln, o_ln) and a_st.lf_steps == int(o_steps.sum())
        rows_ = np.arange(0, ox.n, 11, dtype=np.uint64)
        assert [tuple(int(v) for v in t) for t in zip(*auto.locate(rows_))] == [ox.locate(int(r)) for r in rows_]
        auto.accelerate(0)                                     # dropping the expansion: the layout's own kernels again, same answers
        assert np.array_equal(fm.search_no_errors.search(auto, (qb_, qo_))[1], o_ln)
        assert same_hits(fm.search_ng26.search(auto, (qb_, qo_), fm.search_scheme.h2(4, 0, 2)), o_hits)
    same = [q for q in mutated_queries(seqs, 600, 24, 25, 2, seed=sigma, sigma=sigma)]
    ragged = mutated_queries(seqs, 300, 1, 60, 1, seed=sigma + 1, sigma=sigma)
    sch = fm.search_scheme.h2(3, 0, 1)
    for queries in (same, ragged):
        qbuf, qoff = fm.flatten(queries)
        lb, ln = fm.search_no_errors.search(gx, (qbuf, qoff))
        olb, oln = ox.search_exact(qbuf, qoff)
        assert np.array_equal(lb, olb) and np.array_equal(ln, oln)
        keep = [i for i, q in enumerate(queries) if len(q) >= 3]
        kb, ko = fm.flatten([queries[i] for i in keep])
        hits, st = fm.search_ng26.search(gx, (kb, ko), sch, want_stats=True)
        ohits, _, nodes = ox.search_ng26(kb, ko, sch)
        assert same_hits(hits, ohits) and st.lf_steps == nodes
        assert same_hits(fm.search_backtracking.search(gx, (qbuf, qoff), 1), ox.search_backtracking(qbuf, qoff, 1)[0])
    rows = np.arange(0, ox.n, 7, dtype=np.uint64)
    seq, pos, steps = gx.locate(rows)
    assert [(int(a), int(b), int(c)) for a, b, c in zip(seq, pos, steps)] == [ox.locate(int(r)) for r in rows]
    idx = np.repeat(np.arange(0, ox.n + 1, 5, dtype=np.uint64), sigma); sym = np.tile(np.arange(sigma, dtype=np.uint8), len(idx) // sigma)
    r1 = gx.rank(idx, sym)
    if sigma <= 6:
        gx.accelerate(2)                                           # k-step table over the expansion
        qbuf, qoff = fm.flatten(ragged)
        lb, ln = fm.search_no_errors.search(gx, (qbuf, qoff))
        olb, oln = ox.search_exact(qbuf, qoff)
        assert np.array_equal(lb, olb) and np.array_equal(ln, oln)
    gx.accelerate(0)
    assert gx.device_bytes == before
    qbuf, qoff = fm.flatten(same)
    assert same_hits(fm.search_ng26.search(gx, (qbuf, qoff), sch), ox.search_ng26(qbuf, qoff, sch)[0])
    assert np.array_equal(gx.rank(idx, sym), r1)


# ------------------------------------------------------------------------------------------------ edit distance
def test_edit_distance_reference_vectors():
    """search/checkSearches.cpp:1093-1121, :1148-1171, :1422-1466 — the located multisets the reference's own tests expect from
    search_ng26::search (Edit = true by default) and fmc::search<true> / search_n<true>"""
    g = REF["searches_edit"]
    ox = fo.OraIndex.build("IB16", g["sigma"], g["input"], g["sampling_rate"], True)
    gx = gpu_index(ox)
    for key, sch in (("ng26_pigeon_opt_CD_DB", fm.search_scheme.pigeon_opt(0, 1)), ("ng26_pigeon_opt_n3", fm.search_scheme.pigeon_opt(0, 1))):
        c = g[key]
        hits = fm.search_ng26.search(gx, c["queries"], sch, n=c.get("n", fm.UINT64_MAX), edit=True)
        owner, seq, pos, steps = fm.LocateLinear(gx, hits["lb"], hits["len"])()
        got = sorted([int(hits["qidx"][o]), int(a), int(b + s)] for o, a, b, s in zip(owner, seq, pos, steps))
        assert got == c["expected"], key
    for key in ("facade_k1", "facade_k1_n3"):
        c = g[key]
        hits = fm.search(gx, c["queries"], 1, n=c.get("n", fm.UINT64_MAX), edit=True)
        owner, seq, pos, steps = fm.LocateLinear(gx, hits["lb"], hits["len"])()
        got = sorted([int(hits["qidx"][o]), int(a), int(b + s)] for o, a, b, s in zip(owner, seq, pos, steps))
        assert got == c["expected"], key


@pytest.mark.parametrize("layout,sigma,k", [("IB16", 5, 1), ("IB16", 5, 2), ("IB16", 5, 3), ("WAVELET", 28, 1), ("EPRV2_16", 5, 2), ("IB16", 256, 1), ("EPR16", 6, 2)])
def test_edit_distance_matches_the_cpu_walk(layout, sigma, k):
    """search_ng26<Edit = true>: same cursors, error counts, callback order and number of extensions as the CPU restatement, for
    equal and ragged query lengths, several schemes, search_n clipping and an explicit partition"""
    rng = np.random.default_rng(sigma + k)
    base = rng.integers(1, min(sigma, 8), size=700, dtype=np.uint8)
    seqs = [np.concatenate([base, base[100:400]]), rng.integers(1, min(sigma, 8), size=300, dtype=np.uint8)]
    ox = fo.OraIndex.build(layout, sigma, seqs, 4, True)
    gx = gpu_index(ox)
    queries = []
    for i in range(250 if k < 3 else 80):
        m = int(rng.integers(k + 2, 36 if k < 3 else 22)); p = int(rng.integers(0, len(seqs[0]) - m - 1)); q = list(seqs[0][p: p + m])
        for _ in range(int(rng.integers(0, k + 2))):
            op = int(rng.integers(0, 3)); j = int(rng.integers(0, len(q)))
            if op == 0: q[j] = int(rng.integers(1, min(sigma, 8)))
            elif op == 1: q.insert(j, int(rng.integers(1, min(sigma, 8))))
            elif len(q) > k + 3: del q[j]
        queries.append(np.array(q, dtype=np.uint8))
    qbuf, qoff = fm.flatten(queries)
    for sch in (fm.search_scheme.h2(k + 2, 0, k), fm.search_scheme.pigeon_opt(0, k), fm.search_scheme.backtracking(2, 0, k)):
        hits, st = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, edit=True)
        ohits, qc, nodes = ox.search_ng26(qbuf, qoff, sch, edit=True)
        assert same_hits(hits, ohits) and st.lf_steps == nodes and len(ohits) > 0, (layout, k)
    sch = fm.search_scheme.h2(k + 2, 0, k)
    for n in (1, 3):
        assert same_hits(fm.search_ng26.search(gx, (qbuf, qoff), sch, n=n, edit=True), ox.search_ng26(qbuf, qoff, sch, max_hits=n, edit=True)[0])
    same = [q for q in queries if len(q) == 20][:1] * 3 + [np.array(list(seqs[0][5:25]), dtype=np.uint8)]
    sb, so = fm.flatten(same)
    part = np.array([20 - 3 * (k + 1)] + [3] * (k + 1), dtype=np.uint64)
    assert same_hits(fm.search_ng26.search(gx, (sb, so), sch, partition=part, edit=True), ox.search_ng26(sb, so, sch, partition=part, edit=True)[0])
    # Edit = false through the same entry point still takes the Hamming kernels
    assert same_hits(fm.search_ng26.search(gx, (qbuf, qoff), sch), ox.search_ng26(qbuf, qoff, sch)[0])
    if sigma <= 8:                                            # with the prefix table the always-exact first part starts from its entry
        gx.accelerate_search(3, 1)
        for sch2 in (sch, fm.search_scheme.pigeon_opt(0, k)):
            hits, st = fm.search_ng26.search(gx, (qbuf, qoff), sch2, want_stats=True, edit=True)
            ohits, _, nodes = ox.search_ng26(qbuf, qoff, sch2, edit=True)
            assert same_hits(hits, ohits) and st.lf_steps == nodes


@pytest.mark.parametrize("k,length,sigma", [(1, 20, 5), (2, 31, 5), (2, 101, 5), (3, 40, 5), (1, 30, 28), (2, 24, 6)])
def test_edit_distance_equal_length_fast_path(k, length, sigma):
    """equal-length batches take the table-driven edit-distance kernel (k_scheme_fast_edit): same cursors, errors, callback order and
    extension counts as the CPU walk, with and without the prefix table, with search_n clipping, and equal to the general kernel"""
    rng = np.random.default_rng(100 + k + length)
    hi = min(sigma, 8)
    base = rng.integers(1, hi, size=2000, dtype=np.uint8)
    seqs = [np.concatenate([base, base[300:900]]), rng.integers(1, hi, size=500, dtype=np.uint8)]
    ox = fo.OraIndex.build("IB16", sigma, seqs, 4, True)
    gx = gpu_index(ox)
    queries = []
    for i in range(700 if k < 3 else 150):
        p = int(rng.integers(0, len(seqs[0]) - length - 4)); q = list(seqs[0][p: p + length + 3])
        for _ in range(int(rng.integers(0, k + 2))):
            op = int(rng.integers(0, 3)); jj = int(rng.integers(0, len(q)))
            if op == 0: q[jj] = int(rng.integers(1, hi))
            elif op == 1: q.insert(jj, int(rng.integers(1, hi)))
            else: del q[jj]
        queries.append(np.array(q[:length], dtype=np.uint8))
    assert len({len(q) for q in queries}) == 1
    qbuf, qoff = fm.flatten(queries)
    for accel in (None, (3, 1)):
        if accel is not None:
            gx.accelerate_search(*accel)
        for sch in (fm.search_scheme.h2(k + 2, 0, k), fm.search_scheme.pigeon_opt(0, k), fm.search_scheme.backtracking(2, 0, k)):
            hits, st = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, edit=True, capacity=1 << 21)
            ohits, qc, nodes = ox.search_ng26(qbuf, qoff, sch, edit=True, cap=1 << 21)
            assert same_hits(hits, ohits) and st.lf_steps == nodes and len(ohits) > 0, (accel, k, length)
        sch = fm.search_scheme.h2(k + 2, 0, k)
        for n in (1, 4):
            assert same_hits(fm.search_ng26.search(gx, (qbuf, qoff), sch, n=n, edit=True), ox.search_ng26(qbuf, qoff, sch, max_hits=n, edit=True)[0])
    fm.options["kernel_select"] = "2"                       # the general kernel
    try:
        hits2 = fm.search_ng26.search(gx, (qbuf, qoff), sch, edit=True, capacity=1 << 21)
    finally:
        del fm.options["kernel_select"]
    assert same_hits(hits2, ox.search_ng26(qbuf, qoff, sch, edit=True, cap=1 << 21)[0])


@pytest.mark.parametrize("k", [0, 1, 2, 3])
@pytest.mark.parametrize("length", [30, 101])
def test_edit_distance_fast_kernel_on_the_plain_index(k, length):
    """edit distance (search_ng26<true>, the reference's default) on a BiFMIndex<5> WITHOUT any table: equal-length batches take k_scheme_fast_edit, whose
    one-row nodes read the row's symbol and LF off the row's block — cursors, errors, callback order and extension counts equal the CPU walk and
    the general kernel (FMGPU_SEL_GENERAL_DFS); reads with delimiters inside; search_n clipping"""
    rng = np.random.default_rng(300 + k + length)
    seqs = repeat_text(40 + k, n=6000) + [np.tile(np.array([1, 1, 2], dtype=np.uint8), 200)]
    ox = fo.OraIndex.build("IB16", 5, seqs, 4, True)
    fm.options["lf_table"] = "0"
    try:
        gx = gpu_index(ox)
    finally:
        del fm.options["lf_table"]
    queries = []
    src = [q for q in seqs if len(q) > length + 8]
    for i in range(600 if k < 3 else 120):
        s_ = src[i % len(src)]
        p = int(rng.integers(0, len(s_) - length - 8)); q = list(s_[p: p + length + 6])
        for _ in range(int(rng.integers(0, k + 2))):
            op = int(rng.integers(0, 3)); jj = int(rng.integers(0, len(q)))
            if op == 0: q[jj] = int(rng.integers(1, 5))
            elif op == 1: q.insert(jj, int(rng.integers(1, 5)))
            else: del q[jj]
        queries.append(np.array(q[:length], dtype=np.uint8))
    queries[3][length // 2] = 0; queries[4][0] = 0
    assert len({len(q) for q in queries}) == 1
    qbuf, qoff = fm.flatten(queries)
    for sch in (fm.search_scheme.h2(k + 2, 0, k), fm.search_scheme.pigeon_opt(0, k)):
        ohits, qc, nodes = ox.search_ng26(qbuf, qoff, sch, edit=True, cap=1 << 22)
        assert len(ohits) > 0
        for flags in ("0", "2"):
            fm.options["kernel_select"] = flags
            try:
                hits, st = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, edit=True, capacity=1 << 22)
            finally:
                del fm.options["kernel_select"]
            assert same_hits(hits, ohits) and st.lf_steps == nodes, (k, length, flags)
    sch = fm.search_scheme.h2(k + 2, 0, k)
    for n in (1, 4):
        assert same_hits(fm.search_ng26.search(gx, (qbuf, qoff), sch, n=n, edit=True), ox.search_ng26(qbuf, qoff, sch, max_hits=n, edit=True)[0])


def test_edit_distance_on_a_repeat_structured_text():
    """k = 2 edit distance on 4 Mbp of the genome-like text (repeat families, satellites, single-symbol runs): reads out of repeats visit 10^4 - 10^6
    nodes where the median read visits hundreds, so the lanes of a wave hand subtrees to each other all the time (k_scheme_fast_edit) and the
    callback order has to come out of the path keys: records, order and extension counts equal the CPU walk; with work sharing switched off too"""
    torch = pytest.importorskip("torch")
    from fmindex_collection_amd import datasets
    lengths = [2_500_000, 1_500_000]
    text, stats = datasets.genome_like_text(lengths, seed=11, device=torch.device("cuda", 0))
    host = text.cpu().numpy()
    seqs = np.split(host, np.cumsum(lengths)[:-1])
    ox = fo.OraIndex.build("IB16", 5, seqs, 16, True)
    gx = gpu_index(ox)
    gx.accelerate_search(8, 1)
    rng = np.random.default_rng(12)
    L, queries = 50, []
    for i in range(6000):
        sq = seqs[i & 1]
        p = int(rng.integers(0, len(sq) - L - 4)); q = list(sq[p: p + L + 3])
        for _ in range(int(rng.integers(0, 3))):
            op = int(rng.integers(0, 3)); jj = int(rng.integers(0, len(q)))
            if op == 0: q[jj] = int(rng.integers(1, 5))
            elif op == 1: q.insert(jj, int(rng.integers(1, 5)))
            else: del q[jj]
        queries.append(np.array(q[:L], dtype=np.uint8))
    qbuf, qoff = fm.flatten(queries)
    sch = fm.search_scheme.h2(4, 0, 2)
    ohits, qc, nodes = ox.search_ng26(qbuf, qoff, sch, edit=True, cap=1 << 24)
    per_read = np.bincount(ohits["qidx"].astype(np.int64), minlength=len(queries))
    assert per_read.max() > 50 * max(1, int(np.median(per_read)))           # the heavy tail this test is about
    hits, st = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, edit=True, capacity=1 << 24)
    assert same_hits(hits, ohits) and st.lf_steps == nodes
    fm.options["kernel_select"] = str(1 << 24)               # no work sharing: the callback index is counted, not derived from keys
    try:
        hits2, st2 = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, edit=True, capacity=1 << 24)
    finally:
        del fm.options["kernel_select"]
    assert same_hits(hits2, ohits) and st2.lf_steps == nodes
    # ... and between the WAVES of the launch (the board, csrc/fmgpu_search_shared.h): 6 000 reads leave most of the chip's waves without reads of their own — they wait at the
    # board and take subtrees of the heavy reads; switched off, and on the plain index (no LF table: the PLAIN instantiation), the records are the same
    with fm.options(kernel_select=capi.SEL_NO_BOARD):
        hits5, st5 = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, edit=True, capacity=1 << 24)
    assert same_hits(hits5, ohits) and st5.lf_steps == nodes
    with fm.options(lf_table=0):
        px = gpu_index(ox)
    for sel in (0, capi.SEL_NO_BOARD):
        with fm.options(kernel_select=sel):
            hits6, st6 = fm.search_ng26.search(px, (qbuf, qoff), sch, want_stats=True, edit=True, capacity=1 << 24)
        assert same_hits(hits6, ohits) and st6.lf_steps == nodes
    assert same_hits(fm.search_ng26.search(gx, (qbuf, qoff), sch, n=3, edit=True), ox.search_ng26(qbuf, qoff, sch, max_hits=3, edit=True)[0])
    # the general kernels (ragged batches, other layouts, 64-bit rows) share work at the end of the batch, with the same keys; search_ng21 too
    hh, _, hnodes = ox.search_ng26(qbuf, qoff, sch, cap=1 << 24)
    fm.options["kernel_select"] = "2"
    try:
        hits3, st3 = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, edit=True, capacity=1 << 24)
        hits4, st4 = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, capacity=1 << 24)
    finally:
        del fm.options["kernel_select"]
    assert same_hits(hits3, ohits) and st3.lf_steps == nodes
    assert same_hits(hits4, hh) and st4.lf_steps == hnodes
    ex = fm.search_scheme.expand(sch, L)
    h21, st21 = fm.search_ng21.search(gx, (qbuf, qoff), ex, want_stats=True, capacity=1 << 24)
    o21, _, n21 = ox.search_ng21(qbuf, qoff, ex, cap=1 << 24)
    assert same_hits(h21, o21) and st21.lf_steps == n21


_SLOT_PROBE = r"""
import sys
sys.path.insert(0, %r)
import numpy as np, torch
import fmindex_collection_amd as fm
from fmindex_collection_amd import datasets
dev = torch.device("cuda", 0)
lengths = [2_500_000, 1_200_000, 300_000]
text, _ = datasets.genome_like_text(lengths, seed=11, device=dev)
n = int(text.numel())
seq_off = torch.from_numpy(np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)).to(dev)
g = torch.Generator(device=dev); g.manual_seed(5)
class V:
    def __init__(self, t): self.t, self.ptr, self.nbytes = t, t.data_ptr(), t.numel() * t.element_size()
out = []
for wide in (0, 1):
    with fm.options(lf_table=0, force_wide=wide):
        gx = fm.BiFMIndex.from_sequences((V(text), V(seq_off)), 5, "IB16", 16)
    for L in (101, 151):
        nq = 150_000
        starts = torch.randint(0, n - L, (nq,), generator=g, device=dev, dtype=torch.int64)
        reads = text[starts[:, None] + torch.arange(L, device=dev)[None, :]].contiguous()
        rows = torch.arange(nq, device=dev)
        for k in range(2):
            sel = rows[rows %% 3 > k]
            p = torch.randint(0, L, (sel.numel(),), generator=g, device=dev)
            reads[sel, p] = reads[sel, p] %% 4 + 1
        hits, st = fm.search_ng26.search(gx, (reads.reshape(-1).cpu().numpy(), np.arange(nq + 1, dtype=np.uint64) * L), fm.search_scheme.h2(4, 0, 2), want_stats=True, capacity=1 << 25)
        out.append((wide, L, len(hits), int(st.lf_steps), int(st.hits) >> 48))      # (the development build reports slots that disagreed with the stack in the top bits of stats.hits)
        if not wide and L == 101:                                 # the edit-distance kernel keeps its top frames in LDS slots too (write-back: a clean slot holds what HBM holds)
            hq = reads[:60_000].reshape(-1).cpu().numpy()
            ehits, est = fm.search_ng26.search(gx, (hq, np.arange(60_001, dtype=np.uint64) * L), fm.search_scheme.h2(4, 0, 2), want_stats=True, edit=True, capacity=1 << 25)
            out.append((2, L, len(ehits), int(est.lf_steps), int(est.hits) >> 48))
    gx.close()
print("SLOTS", out)
"""


def test_lean_kernel_lds_slots_hold_the_frames_of_the_stack():
    """k_scheme_lean reads the top and the bottom frame of a lane's stack from LDS slots that an LDS-DMA load refills; that the load has landed when a slot is read
    rests on an ordering argument the compiler does not know (fmgpu_search.hip, "Order of the top-frame slot's accesses").  The development build (make DEV=1:
    libfmgpu_dev.so) compares every slot it reads with the frame the write-through stack holds in HBM and counts the differences: zero over 600 k reads of a
    repeat-structured text (deep stacks, hand-overs between lanes), 101 and 151 bp, 32- and 64-bit rows; k_scheme_fast_edit (edit distance, write-back slots) likewise."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dev_lib = os.path.join(root, "fmindex-collection_amd", "libfmgpu_dev.so")
    if not os.path.exists(dev_lib):
        pytest.skip("libfmgpu_dev.so is not built (make -C fmindex-collection_amd/csrc DEV=1)")
    env = {k: v for k, v in os.environ.items() if not k.startswith("FMGPU_")}
    env["FMGPU_LIBRARY"] = dev_lib
    r = subprocess.run([sys.executable, "-c", _SLOT_PROBE % root], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "SLOTS" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    got = eval(r.stdout.split("SLOTS", 1)[1].strip())
    assert len(got) == 5 and all(hits > 100_000 and nodes > 10_000_000 and bad == 0 for _, _, hits, nodes, bad in got), got


@pytest.mark.parametrize("layout,sigma,tables", [("IB16", 5, False), ("IB16", 5, True), ("WAVELET", 28, False), ("EPR16", 5, False), ("IB16", 256, False)])
def test_packed_exact_intervals(layout, sigma, tables):
    """fmgpu_search_exact_packed: one word lb << 32 | len per query, equal to the two-array form on every kernel variant"""
    rng = np.random.default_rng(sigma)
    text = rng.integers(1, min(sigma, 8), size=40000, dtype=np.uint8)
    ox = fo.OraIndex.build(layout, sigma, [text], 8, False)
    gx = gpu_index(ox)
    if tables:
        gx.accelerate(3, lut_len=5, walk=2)
    reads = sample_reads(text, 5000, 70, seed=4, mutate=1) + [np.zeros(0, dtype=np.uint8)]
    qbuf, qoff = fm.flatten(reads)
    lb, ln = ox.search_exact(qbuf, qoff)
    word, st = fm.search_no_errors.search_packed(gx, (qbuf, qoff), want_stats=True)
    assert np.array_equal(word, (lb << np.uint64(32)) | ln) and st.lf_steps > 0


def test_hit_records_pack16():
    """fmgpu_hits_pack16: the 16-byte transport form of the 40-byte hit record, host and device buffers"""
    rng = np.random.default_rng(3)
    n = 5000
    hits = np.zeros(n, dtype=fm.HIT_DTYPE)
    hits["qidx"] = rng.integers(0, 2**32, size=n); hits["lb"] = rng.integers(0, 2**32, size=n); hits["lb_rev"] = rng.integers(0, 2**32, size=n)
    hits["len"] = rng.integers(0, 2**32, size=n); hits["errors"] = rng.integers(0, 256, size=n); hits["seq"] = rng.integers(0, 2**24, size=n)
    want = np.empty((n, 2), dtype=np.uint64)
    want[:, 0] = hits["qidx"] | (hits["lb"] << np.uint64(32))
    want[:, 1] = hits["len"] | (hits["errors"].astype(np.uint64) << np.uint64(32)) | (hits["seq"].astype(np.uint64) << np.uint64(40))
    out = np.zeros((n, 2), dtype=np.uint64)
    capi.check(capi.lib().fmgpu_hits_pack16(capi.ptr(hits), n, capi.ptr(out), None))
    assert np.array_equal(out, want)
    dh, do = fm.DeviceBuffer.from_array(hits), fm.DeviceBuffer(n * 16)
    capi.check(capi.lib().fmgpu_hits_pack16(C.c_void_p(dh.ptr), n, C.c_void_p(do.ptr), None))
    capi.check(capi.lib().fmgpu_synchronize(None))
    assert np.array_equal(do.to_array(np.uint64, 2 * n).reshape(n, 2), want)


def test_hit_records_pack24_and_range_checks():
    """fmgpu_hits_pack24 carries the whole record of a 32-bit-row index, order key included; both transport forms refuse what does not fit"""
    rng = np.random.default_rng(4)
    n = 3000
    hits = np.zeros(n, dtype=fm.HIT_DTYPE)
    for f in ("qidx", "lb", "lb_rev", "len", "errors", "seq"):
        hits[f] = rng.integers(0, 2**32, size=n)
    out = np.zeros((n, 3), dtype=np.uint64)
    capi.check(capi.lib().fmgpu_hits_pack24(capi.ptr(hits), n, capi.ptr(out), None))
    assert np.array_equal(out[:, 0], hits["qidx"] | (hits["lb"] << np.uint64(32)))
    assert np.array_equal(out[:, 1], hits["len"] | (hits["errors"].astype(np.uint64) << np.uint64(32)))
    assert np.array_equal(out[:, 2], hits["lb_rev"] | (hits["seq"].astype(np.uint64) << np.uint64(32)))
    bad = hits[:10].copy(); bad["lb"][3] = 2**32
    assert capi.lib().fmgpu_hits_pack24(capi.ptr(bad), 10, capi.ptr(out), None) == capi.FMGPU_ERR_UNSUPPORTED
    for field, value in (("errors", 256), ("seq", 2**24), ("qidx", 2**32), ("len", 2**33)):
        ok = np.zeros(4, dtype=fm.HIT_DTYPE)
        assert capi.lib().fmgpu_hits_pack16(capi.ptr(ok), 4, capi.ptr(out), None) == 0
        ok[field][2] = value
        assert capi.lib().fmgpu_hits_pack16(capi.ptr(ok), 4, capi.ptr(out), None) == capi.FMGPU_ERR_UNSUPPORTED, field


def test_hits_sort_orders_by_path_key_and_normalises():
    """fmgpu_hits_sort: ascending qidx, inside a read ascending (errors >> 8, seq); afterwards seq is the position inside the read and errors the
    error count alone — whatever mixture of dense indices and path keys came in"""
    rng = np.random.default_rng(9)
    n = 20000
    hits = np.zeros(n, dtype=fm.HIT_DTYPE)
    hits["qidx"] = rng.integers(0, 300, size=n)
    key = rng.permutation(n).astype(np.uint64) * np.uint64(2**40 // n)           # distinct 56-bit keys
    hits["seq"] = (key & np.uint64(0xffffffff)).astype(np.uint32)
    e = rng.integers(0, 3, size=n).astype(np.uint32)
    hits["errors"] = e | ((key >> np.uint64(32)).astype(np.uint32) << np.uint32(8))
    hits["lb"] = np.arange(n)
    order = np.lexsort((key, hits["qidx"]))
    got = hits.copy()
    capi.check(capi.lib().fmgpu_hits_sort(capi.ptr(got), n, None))
    assert np.array_equal(got["lb"], hits["lb"][order]) and np.array_equal(got["errors"], e[order])
    starts = np.r_[0, np.nonzero(np.diff(got["qidx"].astype(np.int64)))[0] + 1]
    want_seq = np.arange(n) - np.repeat(starts, np.diff(np.r_[starts, n]))
    assert np.array_equal(got["seq"], want_seq.astype(np.uint32))


# ------------------------------------------------------------------------------------------------ concurrency (threads, streams)
def test_concurrent_host_threads_on_one_handle():
    """the reference is re-entrant on a const index; so is the C-ABI: four host threads run exact, k-mismatch, edit-distance, search_ng21
    and locate calls on the same handle at the same time (ctypes releases the GIL during a call) and every result equals the oracle's"""
    import threading
    rng = np.random.default_rng(21)
    base = rng.integers(1, 5, size=6000, dtype=np.uint8)
    seqs = [np.concatenate([base, base[1000:3000]]), rng.integers(1, 5, size=1500, dtype=np.uint8)]
    ox = fo.OraIndex.build("IB16", 5, seqs, 4, True)
    gx = gpu_index(ox)
    gx.accelerate(3, lut_len=5, walk=True).accelerate_search(4, 3)
    sch = fm.search_scheme.h2(3, 0, 1)
    work = []
    for t in range(4):
        r = np.random.default_rng(100 + t)
        reads = []
        for i in range(600):
            p = int(r.integers(0, len(seqs[0]) - 40)); q = seqs[0][p: p + 36].copy()
            if i % 2: q[int(r.integers(0, 36))] = r.integers(1, 5)
            reads.append(q)
        qbuf, qoff = fm.flatten(reads)
        rows = r.integers(0, ox.n, size=500).astype(np.uint64)
        ex = fm.search_scheme.expand(fm.search_scheme.pigeon_opt(0, 1), 36)
        want = (ox.search_exact(qbuf, qoff), ox.search_ng26(qbuf, qoff, sch)[0], ox.search_ng26(qbuf, qoff, sch, edit=True)[0],
                ox.search_ng21(qbuf, qoff, ex)[0], [ox.locate(int(x)) for x in rows])
        work.append((qbuf, qoff, rows, ex, want))
    errors = []

    def run(t):
        try:
            qbuf, qoff, rows, ex, want = work[t]
            for rep in range(6):
                lb, ln = fm.search_no_errors.search(gx, (qbuf, qoff))
                assert np.array_equal(lb, want[0][0]) and np.array_equal(ln, want[0][1])
                assert same_hits(fm.search_ng26.search(gx, (qbuf, qoff), sch), want[1])
                assert same_hits(fm.search_ng26.search(gx, (qbuf, qoff), sch, edit=True), want[2])
                assert same_hits(fm.search_ng21.search(gx, (qbuf, qoff), ex), want[3])
                seq, pos, steps = gx.locate(rows)
                assert [(int(a), int(b), int(c)) for a, b, c in zip(seq, pos, steps)] == [tuple(int(v) for v in w) for w in want[4]]
        except BaseException as e:                            # noqa: BLE001 — reported by the main thread
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=run, args=(t,)) for t in range(4)]
    for th in threads: th.start()
    for th in threads: th.join()
    assert not errors, errors


def test_calls_on_caller_streams_with_device_buffers():
    """queries and results in HBM, two caller-owned HIP streams in flight at once: each call's result is complete once ITS stream is synchronised
    (streams from the HIP runtime libfmgpu.so itself is linked against — a second runtime in the process would not see its device)"""
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    rng = np.random.default_rng(22)
    text = rng.integers(1, 5, size=50000, dtype=np.uint8)
    ox = fo.OraIndex.build("IB16", 5, [text], 8, False)
    gx = gpu_index(ox)
    L = capi.lib()
    jobs = []
    for k in range(2):
        st = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(st)) == 0
        reads = sample_reads(text, 20000, 60, seed=30 + k, mutate=1)
        qbuf, qoff = fm.flatten(reads)
        nq = len(reads)
        dq, do = fm.DeviceBuffer.from_array(qbuf), fm.DeviceBuffer.from_array(qoff.astype(np.uint64))
        dl, dn = fm.DeviceBuffer.from_array(np.full(nq, 2**64 - 1, dtype=np.uint64)), fm.DeviceBuffer.from_array(np.full(nq, 2**64 - 1, dtype=np.uint64))
        jobs.append((st, dq, do, dl, dn, ox.search_exact(qbuf, qoff), nq))
    for rep in range(3):
        for st, dq, do, dl, dn, want, nq in jobs:
            capi.check(L.fmgpu_search_exact(gx._h, C.c_void_p(dq.ptr), C.c_void_p(do.ptr), nq, C.c_void_p(dl.ptr), C.c_void_p(dn.ptr), None, st))
        for st, dq, do, dl, dn, want, nq in jobs:
            assert hip.hipStreamSynchronize(st) == 0
            assert np.array_equal(dl.to_array(np.uint64, nq), want[0]) and np.array_equal(dn.to_array(np.uint64, nq), want[1])
    for st, *_ in jobs:
        hip.hipStreamDestroy(st)


# ------------------------------------------------------------------------------------------------ search_ng21 (expanded schemes)
def test_ng21_reference_vectors():
    """search/checkSearches.cpp:422-525: the located multisets the reference's tests expect from search_ng21::search / search_n /
    search_best / search_best_n over expand(pigeon_opt(..), 2)"""
    g = REF["searches_ng21"]
    ox = fo.OraIndex.build("IB16", g["sigma"], g["input"], g["sampling_rate"], True)
    gx = gpu_index(ox)
    m = len(g["queries"][0])
    ex = lambda a: fm.search_scheme.expand(fm.search_scheme.pigeon_opt(*a), m)

    def located(hits):
        owner, seq, pos, steps = fm.LocateLinear(gx, hits["lb"], hits["len"])()
        return sorted([int(hits["qidx"][o]), int(a), int(b + s)] for o, a, b, s in zip(owner, seq, pos, steps))

    assert located(fm.search_ng21.search(gx, g["queries"], ex(g["search"]["scheme"]))) == g["search"]["expected"]
    assert located(fm.search_ng21.search_n(gx, g["queries"], ex(g["search_n"]["scheme"]), g["search_n"]["n"])) == g["search_n"]["expected"]
    assert located(fm.search_ng21.search_best(gx, g["queries"], [ex(a) for a in g["search_best"]["schemes"]])) == g["search_best"]["expected"]
    assert located(fm.search_ng21.search_best_n(gx, g["queries"], [ex(a) for a in g["search_best_n"]["schemes"]], g["search_best_n"]["n"])) == g["search_best_n"]["expected"]


@pytest.mark.parametrize("layout,sigma,k,length", [("IB16", 5, 1, 20), ("IB16", 5, 2, 31), ("IB16", 5, 3, 18), ("WAVELET", 28, 1, 24), ("EPRV2_16", 5, 2, 27),
                                                   ("IB16", 256, 1, 16), ("EPR16", 6, 2, 22), ("FBV_512_64K", 5, 2, 40), ("IB16", 5, 2, 101)])
def test_ng21_matches_the_cpu_walk(layout, sigma, k, length):
    """search_ng21: same cursors, error counts, callback order and number of extensions as the CPU restatement — several expanded schemes,
    search_n clipping, search_best over 0..k errors, queries longer than the scheme (prefix searched) and shorter (skipped)"""
    rng = np.random.default_rng(300 + sigma + k + length)
    hi = min(sigma, 8)
    base = rng.integers(1, hi, size=1500, dtype=np.uint8)
    seqs = [np.concatenate([base, base[300:800]]), rng.integers(1, hi, size=400, dtype=np.uint8)]
    ox = fo.OraIndex.build(layout, sigma, seqs, 4, True)
    gx = gpu_index(ox)
    queries = []
    for i in range(400 if k < 3 else 100):
        p = int(rng.integers(0, len(seqs[0]) - length - 4)); q = list(seqs[0][p: p + length + 3])
        for _ in range(int(rng.integers(0, k + 2))):
            op = int(rng.integers(0, 3)); jj = int(rng.integers(0, len(q)))
            if op == 0: q[jj] = int(rng.integers(1, hi))
            elif op == 1: q.insert(jj, int(rng.integers(1, hi)))
            else: del q[jj]
        queries.append(np.array(q[:length], dtype=np.uint8))
    qbuf, qoff = fm.flatten(queries)
    schemes = [fm.search_scheme.expand(sc, length) for sc in (fm.search_scheme.h2(k + 2, 0, k), fm.search_scheme.pigeon_opt(0, k), fm.search_scheme.backtracking(1, 0, k))]
    total = 0
    for ex in schemes:
        hits, st = fm.search_ng21.search(gx, (qbuf, qoff), ex, want_stats=True, capacity=1 << 21)
        ohits, qc, nodes = ox.search_ng21(qbuf, qoff, ex, cap=1 << 21)
        assert same_hits(hits, ohits) and st.lf_steps == nodes, (layout, k)
        total += len(ohits)
    assert total > 0
    for n in (1, 3):
        assert same_hits(fm.search_ng21.search_n(gx, (qbuf, qoff), schemes[0], n), ox.search_ng21(qbuf, qoff, schemes[0], max_hits=n)[0])
    best = [fm.search_scheme.expand(fm.search_scheme.pigeon_opt(e, e), length) for e in range(k + 1)]
    for n in (fm.UINT64_MAX, 2):
        got = fm.search_ng21.search_best(gx, (qbuf, qoff), best, n)
        want, _ = ox.search_ng21_best(qbuf, qoff, best, max_hits=n)
        assert all(np.array_equal(got[f], want[f]) for f in ("qidx", "lb", "lb_rev", "len", "errors"))
    ragged = [np.concatenate([queries[0], queries[1][:5]]), queries[2][: length - 1], queries[3], np.zeros(0, dtype=np.uint8)]
    rb, ro = fm.flatten(ragged)
    assert same_hits(fm.search_ng21.search(gx, (rb, ro), schemes[0]), ox.search_ng21(rb, ro, schemes[0])[0])
    if sigma <= 8:                                            # with LF tables a one-row cursor takes its child from one 4-byte load
        gx.accelerate_search(3, 1)
        for ex in schemes:
            hits, st = fm.search_ng21.search(gx, (qbuf, qoff), ex, want_stats=True, capacity=1 << 21)
            ohits, qc, nodes = ox.search_ng21(qbuf, qoff, ex, cap=1 << 21)
            assert same_hits(hits, ohits) and st.lf_steps == nodes, (layout, k)


@pytest.mark.parametrize("seed", list(range(12)))
def test_ng21_randomised_expanded_schemes(seed):
    """hand-made expanded schemes — a random walk of the cursor's two ends, arbitrary (also decreasing) per-symbol bounds, several searches —
    equal the CPU walk record by record"""
    rng = np.random.default_rng(5000 + seed)
    sigma = int(rng.choice([4, 5, 6]))
    length = int(rng.integers(6, 30))
    base = rng.integers(1, sigma, size=int(rng.integers(600, 1500)), dtype=np.uint8)
    seqs = [np.concatenate([base, base[100:400]]), base[50:350][::-1].copy()]
    ox = fo.OraIndex.build(str(rng.choice(["IB16", "EPRV2_16", "IBP16"])), sigma, seqs, 4, True)
    gx = gpu_index(ox)
    K = int(rng.integers(1, 4))
    rows = []
    for _ in range(int(rng.integers(1, 5))):
        lo = hi = int(rng.integers(0, length)); pi = [lo]
        while len(pi) < length:
            if lo > 0 and (hi == length - 1 or rng.random() < 0.5): lo -= 1; pi.append(lo)
            else: hi += 1; pi.append(hi)
        u = np.minimum(K, np.sort(rng.integers(0, K + 2, size=length)))
        if rng.random() < 0.3: u[int(rng.integers(0, length))] = int(rng.integers(0, K + 1))      # a bound that drops again
        l = np.minimum(u, np.sort(rng.integers(0, K + 1, size=length)) * (rng.random(length) < 0.5))
        rows.append((pi, l, u))
    ex = tuple(np.array([r[j] for r in rows], dtype=np.uint64) for j in range(3))
    queries = []
    for i in range(200):
        p = int(rng.integers(0, len(seqs[0]) - length - 4)); q = list(seqs[0][p: p + length + 3])
        for _ in range(int(rng.integers(0, K + 1))):
            op = int(rng.integers(0, 3)); jj = int(rng.integers(0, len(q)))
            if op == 0: q[jj] = int(rng.integers(1, sigma))
            elif op == 1: q.insert(jj, int(rng.integers(1, sigma)))
            else: del q[jj]
        queries.append(np.array(q[:length], dtype=np.uint8))
    qbuf, qoff = fm.flatten(queries)
    for n in (fm.UINT64_MAX, 2, fm.UINT64_MAX):
        hits, st = fm.search_ng21.search(gx, (qbuf, qoff), ex, want_stats=True, capacity=1 << 21, n=n)
        ohits, _, nodes = ox.search_ng21(qbuf, qoff, ex, max_hits=n, cap=1 << 21)
        assert same_hits(hits, ohits) and st.lf_steps == nodes, seed
        if n == 2:
            gx.accelerate_search(int(rng.integers(2, 5)), 1)  # the third pass runs with LF tables


def test_ng21_argument_errors():
    rng = np.random.default_rng(9)
    text = rng.integers(1, 5, size=500, dtype=np.uint8)
    bx = gpu_index(fo.OraIndex.build("IB16", 5, [text], 4, True))
    ux = gpu_index(fo.OraIndex.build("IB16", 5, [text], 4, False))
    ex = fm.search_scheme.expand(fm.search_scheme.pigeon_opt(0, 1), 10)
    q = [text[5:15]]
    with pytest.raises(fm.FmgpuError):
        fm.search_ng21.search(ux, q, ex)                               # needs a BiFMIndex
    pi, l, u = (np.array(a, dtype=np.uint64).copy() for a in ex)
    bad = pi.copy(); bad[0, 0] = 10
    with pytest.raises(fm.FmgpuError):
        fm.search_ng21.search(bx, q, (bad, l, u))
    gap = pi.copy(); gap[0, [1, 2]] = gap[0, [2, 1]]
    with pytest.raises(fm.FmgpuError):
        fm.search_ng21.search(bx, q, (gap, l, u))                      # the cursor cannot jump over a symbol
    big = u.copy(); big[0, -1] = 200
    with pytest.raises(fm.FmgpuError):
        fm.search_ng21.search(bx, q, (pi, l, big))
    empty = tuple(np.zeros((0, 10), dtype=np.uint64) for _ in range(3))
    assert len(fm.search_ng21.search(bx, q, empty)) == 0                # SearchNg21.h:205: an empty scheme reports nothing


@pytest.mark.parametrize("seed", list(range(40)))
def test_randomised_schemes_partitions_and_tables(seed):
    """random texts, read lengths, schemes (h2 / pigeon / backtracking / expanded), explicit partitions with tiny parts (several part ends
    inside one 16-symbol stretch) and accelerator combinations: Hamming and edit distance equal the CPU walk record by record"""
    rng = np.random.default_rng(1000 + seed)
    sigma = int(rng.choice([4, 5, 5, 6]))
    base = rng.integers(1, sigma, size=int(rng.integers(1500, 4000)), dtype=np.uint8)
    seqs = [np.concatenate([base, base[200:900], rng.integers(1, sigma, size=400, dtype=np.uint8)]), base[100:700][::-1].copy()]
    ox = fo.OraIndex.build("IB16", sigma, seqs, 4, True)
    gx = gpu_index(ox)
    gx.accelerate_search(int(rng.integers(0, 6)), int(rng.integers(0, 4)))
    L = int(rng.integers(18, 90))
    k = int(rng.integers(1, 4))
    queries = []
    for i in range(400):
        s = seqs[0]; p = int(rng.integers(0, len(s) - L - 9)); q = list(s[p: p + L + 8])
        for _ in range(int(rng.integers(0, k + 2))):
            op = int(rng.integers(0, 4)); jj = int(rng.integers(0, len(q)))
            if op <= 1: q[jj] = int(rng.integers(1, sigma))
            elif op == 2: q.insert(jj, int(rng.integers(1, sigma)))
            else: del q[jj]
        queries.append(np.array(q[:L] if seed % 4 else q[: L - (i % 3)], dtype=np.uint8))      # every fourth configuration is a ragged batch
    qbuf, qoff = fm.flatten(queries)
    gens = [fm.search_scheme.h2(k + 2, 0, k), fm.search_scheme.pigeon_opt(0, k), fm.search_scheme.backtracking(int(rng.integers(1, 5)), 0, k),
            fm.search_scheme.h2(k + 3, 0, k), fm.search_scheme.h2(k + 2, 1, k)]
    sch = gens[int(rng.integers(0, len(gens)))]
    P = sch[0].shape[1]
    partition = None
    if rng.integers(0, 2):                                     # explicit partition, often with very short parts
        cuts = np.sort(rng.choice(np.arange(1, L), size=P - 1, replace=False)) if P > 1 else np.array([], dtype=np.int64)
        partition = np.diff(np.concatenate([[0], cuts, [L]])).astype(np.uint64)
    n = fm.UINT64_MAX if rng.integers(0, 3) else int(rng.integers(1, 4))
    for edit in (False, True):
        hits, st = fm.search_ng26.search(gx, (qbuf, qoff), sch, partition=partition, n=n, want_stats=True, edit=edit, capacity=1 << 21)
        ohits, _, nodes = ox.search_ng26(qbuf, qoff, sch, partition=partition, max_hits=n, edit=edit, cap=1 << 21)
        assert same_hits(hits, ohits) and st.lf_steps == nodes, (seed, sigma, L, k, edit, None if partition is None else partition.tolist())


@pytest.mark.parametrize("edit", [False, True])
def test_search_n_and_search_best(edit):
    """fmc::search_n (search/search.h:38-46) and search_ng26::search_best (SearchNg26.h:447-487): the host-side drivers around the search
    kernels — per-length cached schemes, the convenience overload's loop over 0 .. maxErrors-1 that stops once any query has a hit, the
    explicit overload's per-query first-scheme-wins"""
    rng = np.random.default_rng(17)
    base = rng.integers(1, 5, size=900, dtype=np.uint8)
    seqs = [np.concatenate([base, base[300:600]]), rng.integers(1, 5, size=300, dtype=np.uint8)]
    ox = fo.OraIndex.build("IB16", 5, seqs, 2, True)
    gx = gpu_index(ox)
    queries = [q for q in mutated_queries(seqs, 120, 6, 30, 2, seed=23)] + [np.array([1, 2], dtype=np.uint8), np.array([4, 4], dtype=np.uint8)]
    qbuf, qoff = fm.flatten(queries)

    def oracle_auto(k, n):                                        # SearchNg26.h:436-444 with the oracle
        out = []
        for short in (False, True):
            sel = [i for i, q in enumerate(queries) if (len(q) == 2) == short]
            if not sel:
                continue
            qb, qo = fm.flatten([queries[i] for i in sel])
            h = ox.search_ng26(qb, qo, fo.scheme_h2(k + (1 if short else 2), 0, k), max_hits=n, edit=edit)[0].copy()
            h["qidx"] = np.array(sel, dtype=np.uint64)[h["qidx"].astype(np.int64)]
            out.append(h)
        h = np.concatenate(out)
        return h[np.argsort(h["qidx"], kind="stable")]             # the oracle emits in callback order: a stable sort keeps it inside a query

    for k, n in ((0, 2), (1, 3), (2, fm.UINT64_MAX)):
        assert same_hits(fm.search_n(gx, (qbuf, qoff), k, n, edit=edit), oracle_auto(k, n)), (k, n)
    want = None
    for k in range(2):
        want = oracle_auto(k, 5)
        if len(want):
            break
    assert same_hits(fm.search_best(gx, (qbuf, qoff), 2, n=5, edit=edit), want)
    # explicit scheme list: exact first, then one error for the queries that found nothing
    lists = [(fm.search_scheme.h2(2, 0, 0), None), (fm.search_scheme.h2(3, 0, 1), None)]
    keep = [i for i, q in enumerate(queries) if len(q) >= 3]
    kb, ko = fm.flatten([queries[i] for i in keep])
    got = fm.search_best(gx, (kb, ko), 1, edit=edit, schemes=lists)
    h0 = ox.search_ng26(kb, ko, fo.scheme_h2(2, 0, 0), edit=edit)[0]
    rest = [i for i in range(len(keep)) if i not in set(h0["qidx"].tolist())]
    rb, ro = fm.flatten([queries[keep[i]] for i in rest])
    h1 = ox.search_ng26(rb, ro, fo.scheme_h2(3, 0, 1), edit=edit)[0].copy()
    h1["qidx"] = np.array(rest, dtype=np.uint64)[h1["qidx"].astype(np.int64)]
    want = np.concatenate([h0, h1]); want = want[np.argsort(want["qidx"], kind="stable")]
    assert same_hits(got, want) and len(h1) > 0 and len(h0) > 0


@pytest.mark.parametrize("bidir", [False, True])
@pytest.mark.parametrize("k", [0, 1, 2])
def test_backtracking(bidir, k):
    """search_backtracking::search on FMIndex and BiFMIndex: hits in callback order, error counts as reported (:63, :76)"""
    seqs = repeat_text(30 + k)
    ox = fo.OraIndex.build("IB16", 5, seqs, 4, bidir)
    gx = gpu_index(ox)
    queries = mutated_queries(seqs, 300, 1, 40, k + 1, seed=8) + [[]]
    qbuf, qoff = fm.flatten(queries)
    hits, st = fm.search_backtracking.search(gx, (qbuf, qoff), k, want_stats=True)
    ohits, nodes = ox.search_backtracking(qbuf, qoff, k)
    assert same_hits(hits, ohits, unidirectional=not bidir) and st.lf_steps == nodes


def test_reference_search_fixtures_on_gpu():
    """search/checkSearches.cpp:23-72, :1482-1505; checkSearchBacktracking.cpp:42-102, :295-327"""
    g = REF["searches"]
    for bidir in (False, True):
        ox = fo.OraIndex.build("IB16", g["sigma"], g["input"], 1, bidir)
        gx = gpu_index(ox)
        hits = fm.search_backtracking.search(gx, g["queries"], 1)
        owner, seq, pos, steps = fm.LocateLinear(gx, hits["lb"], hits["len"])()
        got = sorted([int(hits["qidx"][o]), int(s), int(p + t)] for o, s, p, t in zip(owner, seq, pos, steps))
        assert got == g["backtracking_k1"]
        lb, ln = fm.search_no_errors.search(gx, g["queries"])
        assert ln.tolist() == [0, 0]
    gx = gpu_index(fo.OraIndex.build("IB16", g["sigma"], g["input"], 1, True))
    hits = fm.search(gx, g["queries"], 1, compat_auto_scheme=True)       # fmc::search<false>(index, queries, 1, cb), search/search.h:26-35
    owner, seq, pos, steps = fm.LocateLinear(gx, hits["lb"], hits["len"])()
    assert sorted([int(hits["qidx"][o]), int(s), int(p + t)] for o, s, p, t in zip(owner, seq, pos, steps)) == g["hamming_k1_facade"]
    c = REF["collection"]
    gx = gpu_index(fo.OraIndex.build("IB16", c["sigma"], c["input"], 1, True))
    assert gx.symbol(np.arange(gx.n, dtype=np.uint64)).tolist() == c["bwt"]
    seq, pos, steps = gx.locate(np.arange(gx.n, dtype=np.uint64))
    assert [[int(s), int(p + t)] for s, p, t in zip(seq, pos, steps)] == c["locate"]
    hits = fm.search_backtracking.search(gx, [[ord("A")]], 0)
    assert (int(hits[0]["lb"]), int(hits[0]["len"]), int(hits[0]["errors"])) == (c["query_A"]["lb"], c["query_A"]["count"], 0)


# ------------------------------------------------------------------------------------------------ locate
SAMPLINGS = {"full": lambda i, s: True, "every2nd_row": lambda i, s: i % 2 == 0 or s == 0, "odd_rows": lambda i, s: i % 2 == 1,
             "every2nd_text": lambda i, s: s % 2 == 0}


@pytest.mark.parametrize("fixture,bidir", [("fmindex_hallo", False), ("bifmindex_hallo", True), ("bifmindex_long", True)])
def test_locate_reference_fixtures(fixture, bidir):
    """fmindex/checkFMIndex.cpp:15-110, fmindex/checkBiFMIndex.cpp:13-105, :136-222 — literal BWT / SA, four sampling rules"""
    g = REF[fixture]
    bwt, sa = np.array(g["bwt"], dtype=np.uint8), np.array(g["sa"], dtype=np.uint64)
    rev = np.array(g["bwtRev"], dtype=np.uint8) if bidir else None
    for name, rule in SAMPLINGS.items():
        if fixture == "bifmindex_long" and name == "odd_rows":
            continue
        has = np.array([rule(i, int(sa[i])) for i in range(len(sa))], dtype=np.uint8)
        for layout in ("IB16", "EPRV2_16", "WAVELET"):
            ox = fo.OraIndex.from_bwt(layout, g["sigma"], bwt, rev, has, np.zeros(len(sa), dtype=np.uint64), sa)
            gx = gpu_index(ox)
            seq, pos, steps = gx.locate(np.arange(len(sa), dtype=np.uint64))
            assert np.all(seq == 0) and np.array_equal(pos + steps, sa), (name, layout)
            for i in range(len(sa)):
                assert (int(seq[i]), int(pos[i]), int(steps[i])) == ox.locate(i)


@pytest.mark.parametrize("layout", ["IB16", "EPRV2_16", "WAVELET"])
def test_locate_csa_and_dense_vector_reference_fixtures(layout):
    """suffixarray/checkCSA.cpp:9-81 — Hello$World$ at sampling rates 3, 4, 5, 8: the rows the reference samples answer its (sequence, position) in zero steps and
    every other row walks to the position its suffix-array entry names; checkDenseVector.cpp:8-82 / checkDenseMultiVector.cpp:8-89 — the value arrays of those tests
    (greatest common divisors 1, 2 and 5: DenseVector.h:154-182's scaling) as the two fields of a SparseArray, read back by fmgpu_locate"""
    g = REF["csa"]
    seqs = [np.array(s_, dtype=np.uint8) for s_ in g["sequences"]]
    text = np.concatenate([np.concatenate([s_, [0]]) for s_ in seqs]).astype(np.uint8)
    sa = np.array(g["sa"], dtype=np.int64)
    starts = np.concatenate([[0], np.cumsum([len(s_) + 1 for s_ in seqs])])
    seq_of = np.searchsorted(starts, sa, side="right") - 1
    pos_of = sa - starts[seq_of]
    bwt = text[(sa - 1) % len(text)]
    rows = np.arange(len(sa), dtype=np.uint64)
    for rate, expected in g["sampling"].items():
        has = (pos_of % int(rate) == 0).astype(np.uint8)
        ox = fo.OraIndex.from_bwt(layout, g["sigma"], bwt, None, has, seq_of.astype(np.uint64), pos_of.astype(np.uint64))
        gx = gpu_index(ox)
        for sel in (0, capi.SEL_LOCATE_PER_LANE):
            with fm.options(kernel_select=sel):
                seq, pos, steps = gx.locate(rows)
            for r, s_, p_ in expected:
                assert (int(seq[r]), int(pos[r]), int(steps[r])) == (s_, p_, 0), (rate, r)
            assert np.array_equal(seq, seq_of.astype(np.uint64)) and np.array_equal(pos + steps, pos_of.astype(np.uint64)), rate
            assert [(int(a), int(b), int(c)) for a, b, c in zip(seq, pos, steps)] == [ox.locate(int(r)) for r in rows]
    for which in ("dense_vector", "dense_multi_vector"):
        cases = REF[which]["cases"]
        for a, b in ((cases[0]["inputs"][0], cases[1]["inputs"][0]), (cases[2]["inputs"][0], cases[1]["inputs"][0]), (cases[5]["inputs"][0], cases[5]["inputs"][1])):
            assert len(a) == len(b) == len(sa)                    # 12 values each: one per row of the 12-row index above, every row sampled
            ox = fo.OraIndex.from_bwt(layout, g["sigma"], bwt, None, np.ones(len(sa), dtype=np.uint8), np.array(a, dtype=np.uint64), np.array(b, dtype=np.uint64))
            gx = gpu_index(ox)
            seq, pos, steps = gx.locate(rows)
            assert seq.tolist() == a and pos.tolist() == b and not steps.any(), which


@pytest.mark.parametrize("layout", ["IB16", "IBP16", "EPR16", "EPRV2_16", "WAVELET"])
@pytest.mark.parametrize("rate", [1, 3, 16, 64])
def test_locate_random(layout, rate):
    seqs = repeat_text(rate)
    ox = fo.OraIndex.build(layout, 5, seqs, rate, False)
    gx = gpu_index(ox)
    rows = np.arange(ox.n, dtype=np.uint64)
    seq, pos, steps, st = gx.locate(rows, want_stats=True)
    want = np.array([ox.locate(int(r)) for r in rows], dtype=np.uint64)
    assert np.array_equal(np.stack([seq, pos, steps], axis=1), want)
    assert st.lf_steps == int(want[:, 2].sum())
    if bool(gx.formats & capi.FMT_FUSED):
        # the quad-cooperative kernel (rows of a wave handed out as lanes fall idle) against the one-row-per-lane kernel: the same triples and step count, also for a batch
        # that ends inside a wave's chunk, repeated rows, and rows beyond the index mixed in
        mixed = np.concatenate([rows[::-1], rows[:777], np.array([ox.n, 2**40, ox.n - 1], dtype=np.uint64), rows[: 3 * 512 + 5]])
        got = gx.locate(mixed, want_stats=True)
        with fm.options(kernel_select=capi.SEL_LOCATE_PER_LANE):
            ref = gx.locate(mixed, want_stats=True)
            one, _, _, st1 = gx.locate(rows, want_stats=True)
        assert all(np.array_equal(a, b) for a, b in zip(got[:3], ref[:3])) and got[3].lf_steps == ref[3].lf_steps and st1.lf_steps == st.lf_steps and np.array_equal(one, seq)
    # out-of-range rows are flagged, not walked
    seq, pos, steps = gx.locate(np.array([ox.n, ox.n + 5], dtype=np.uint64))
    assert np.all(steps == np.uint64(2**64 - 1))
    # an index with a sampled suffix array keeps the presence bits in entry 0 of its sigma <= 5 blocks (one line per locate step): every String_c answer,
    # the delimiter's included, is unchanged, and so are searches whose queries hold delimiters
    ork, opr = ox.bwt_string().rank_table()
    idx = np.repeat(np.arange(ox.n + 1, dtype=np.uint64), 5)
    sym = np.tile(np.arange(5, dtype=np.uint8), ox.n + 1)
    assert np.array_equal(gx.rank(idx, sym).reshape(ox.n + 1, 5), ork) and np.array_equal(gx.prefix_rank(idx, sym).reshape(ox.n + 1, 5), opr)
    bw = ox.bwt_string()
    assert np.array_equal(gx.symbol(rows), np.array([bw.symbol(int(r)) for r in rows], dtype=np.uint64))
    queries = [[0], [1, 0], [0, 2], [2, 0, 1], [0, 0], list(seqs[0][-3:]) + [0], [0] + list(seqs[1][:4])] + [list(q) for q in mutated_queries(seqs, 50, 2, 30, 0, seed=rate)]
    qbuf, qoff = fm.flatten(queries)
    lb, ln = fm.search_no_errors.search(gx, (qbuf, qoff))
    olb, oln = ox.search_exact(qbuf, qoff)
    assert np.array_equal(lb, olb) and np.array_equal(ln, oln)
    fm.options["fused_locate"] = "0"                    # the same index without the fusion: the two-line locate
    try:
        gx0 = gpu_index(ox)
    finally:
        del fm.options["fused_locate"]
    assert all(np.array_equal(a, b) for a, b in zip(gx0.locate(rows), gx.locate(rows)))


# ------------------------------------------------------------------------------------------------ GPU index construction
BUILD_CASES = {
    "random": lambda: [make_text(3000, 5, 1)],
    "three_sequences": lambda: [make_text(3000, 5, 2), make_text(10, 5, 3), make_text(700, 5, 4)],
    "all_A": lambda: [np.ones(5000, dtype=np.uint8)],
    "periodic": lambda: [np.tile(np.array([1, 2, 3, 1, 2], dtype=np.uint8), 800)],
    "tiny": lambda: [np.array([1], dtype=np.uint8)],
    "empty_sequence_between": lambda: [np.array([2, 1], dtype=np.uint8), np.array([], dtype=np.uint8), np.array([2, 1, 2], dtype=np.uint8)],
    "binary_70k": lambda: [make_text(70000, 3, 7)],
    "many_short": lambda: [make_text(int(5 + i % 37), 5, 100 + i) for i in range(300)],
}


@pytest.mark.parametrize("name", list(BUILD_CASES))
@pytest.mark.parametrize("bidir", [False, True])
def test_gpu_builder_equals_reference_construction(name, bidir):
    """fmgpu_build_index against FMIndex(Sequences, samplingRate) / BiFMIndex(...) as restated by the oracle: BWT, bwtRev, C,
    and every array of the SparseArray (presence bits, l0, l1, bit-packed seqId / pos, widths and divisors)"""
    seqs = BUILD_CASES[name]()
    for rate in (1, 4, 16):
        gx = (fm.BiFMIndex if bidir else fm.FMIndex).from_sequences(seqs, 5, "IB16", rate, keep_host=True)
        ox = fo.OraIndex.build("IB16", 5, seqs, rate, bidir)
        n = ox.n
        assert gx.n == n
        assert np.array_equal(gx.built_array(0), np.array([ox.bwt_string().symbol(i) for i in range(n)], dtype=np.uint8)) or n > 20000
        if bidir and n <= 20000:
            assert np.array_equal(gx.built_array(1), np.array([ox.bwt_string(rev=True).symbol(i) for i in range(n)], dtype=np.uint8))
        assert np.array_equal(gx.built_array(2, np.uint64), ox.C)
        sp = ox.sparse()
        assert np.array_equal(gx.built_array(3, np.uint64), sp["l0"]) and np.array_equal(gx.built_array(4, np.uint16), sp["l1"])
        assert np.array_equal(gx.built_array(5, np.uint64), sp["bits"])
        assert np.array_equal(gx.built_array(6, np.uint64), sp["fields"][0]["data"]) and np.array_equal(gx.built_array(7, np.uint64), sp["fields"][1]["data"])
        want = [int(sp["fields"][f][k]) for f in (0, 1) for k in ("bitCount", "bits", "largestValue", "commonDivisor")]
        assert gx.built_array(8, np.uint64).tolist() == want
        idx = np.repeat(np.arange(n + 1, dtype=np.uint64), 5); sym = np.tile(np.arange(5, dtype=np.uint8), n + 1)
        if n <= 20000:
            assert np.array_equal(gx.rank(idx, sym).reshape(n + 1, 5), ox.bwt_string().rank_table()[0])
        rows = np.arange(0, n, max(1, n // 400), dtype=np.uint64)
        seq, pos, steps = gx.locate(rows)
        assert [(int(a), int(b), int(c)) for a, b, c in zip(seq, pos, steps)] == [ox.locate(int(r)) for r in rows]


@pytest.mark.parametrize("seed", list(range(16)))
def test_gpu_builder_randomised(seed):
    """random collections — 1 to 40 sequences, empty ones, long runs, tandem repeats, duplicated sequences (ties that only the delimiter
    order resolves), alphabets 2 .. 7, several sampling rates — against the oracle's construction: BWT, bwtRev, C, SparseArray arrays, locate"""
    rng = np.random.default_rng(500 + seed)
    sigma = int(rng.integers(2, 8))
    seqs = []
    for _ in range(int(rng.integers(1, 41))):
        kind = int(rng.integers(0, 6))
        m = int(rng.integers(0, 400))
        if kind == 0 or sigma == 2: q = np.ones(m, dtype=np.uint8)
        elif kind == 1: q = np.tile(rng.integers(1, sigma, size=int(rng.integers(1, 6)), dtype=np.uint8), m // 3 + 1)[:m]
        elif kind == 2 and seqs: q = seqs[int(rng.integers(0, len(seqs)))].copy()
        elif kind == 3 and seqs: base = seqs[int(rng.integers(0, len(seqs)))]; q = base[: len(base) // 2].copy()
        else: q = rng.integers(1, sigma, size=m, dtype=np.uint8)
        seqs.append(q.astype(np.uint8))
    rate = int(rng.choice([1, 2, 5, 16, 64]))
    bidir = bool(rng.integers(0, 2))
    gx = (fm.BiFMIndex if bidir else fm.FMIndex).from_sequences(seqs, sigma, "IB16", rate, keep_host=True)
    ox = fo.OraIndex.build("IB16", sigma, seqs, rate, bidir)
    n = ox.n
    assert gx.n == n
    assert np.array_equal(gx.built_array(0), np.array([ox.bwt_string().symbol(i) for i in range(n)], dtype=np.uint8)), (seed, sigma, len(seqs))
    if bidir:
        assert np.array_equal(gx.built_array(1), np.array([ox.bwt_string(rev=True).symbol(i) for i in range(n)], dtype=np.uint8))
    assert np.array_equal(gx.built_array(2, np.uint64), ox.C)
    sp = ox.sparse()
    assert np.array_equal(gx.built_array(3, np.uint64), sp["l0"]) and np.array_equal(gx.built_array(4, np.uint16), sp["l1"])
    assert np.array_equal(gx.built_array(5, np.uint64), sp["bits"])
    assert np.array_equal(gx.built_array(6, np.uint64), sp["fields"][0]["data"]) and np.array_equal(gx.built_array(7, np.uint64), sp["fields"][1]["data"])
    want = [int(sp["fields"][f][k]) for f in (0, 1) for k in ("bitCount", "bits", "largestValue", "commonDivisor")]
    assert gx.built_array(8, np.uint64).tolist() == want
    rows = np.arange(0, n, max(1, n // 300), dtype=np.uint64)
    seq, pos, steps = gx.locate(rows)
    assert [(int(a), int(b), int(c)) for a, b, c in zip(seq, pos, steps)] == [ox.locate(int(r)) for r in rows]


def test_gpu_builder_other_alphabets_and_errors():
    for sigma, n in ((28, 2000), (4, 500), (256, 700)):
        seqs = [make_text(n, sigma, seed=sigma), make_text(n // 3, sigma, seed=sigma + 1)]
        gx = fm.FMIndex.from_sequences(seqs, sigma, "IB16", 8, keep_host=True)
        ox = fo.OraIndex.build("IB16", sigma, seqs, 8, False)
        assert np.array_equal(gx.built_array(0), np.array([ox.bwt_string().symbol(i) for i in range(ox.n)], dtype=np.uint8))
        qbuf, qoff = fm.flatten(sample_reads(seqs[0], 200, 12, seed=4, mutate=0, sigma=sigma))
        lb, ln = fm.search_no_errors.search(gx, (qbuf, qoff))
        olb, oln = ox.search_exact(qbuf, qoff)
        assert np.array_equal(lb, olb) and np.array_equal(ln, oln)
    with pytest.raises(fm.FmgpuError) as e:
        fm.FMIndex.from_sequences([[1, 7, 2]], 5)                 # symbol >= sigma
    assert e.value.code == capi.FMGPU_ERR_INVALID
    gx = fm.BiFMIndex.from_sequences(seqs, sigma, "EPRV5", 8)    # any blocked layout name: held as the block table on the device
    assert gx.layout == "EPRV5"
    sch = fm.search_scheme.h2(3, 0, 1)
    ox = fo.OraIndex.build("EPRV5", sigma, seqs, 8, True)
    assert same_hits(fm.search_ng26.search(gx, (qbuf, qoff), sch), ox.search_ng26(qbuf, qoff, sch)[0])
    with pytest.raises(fm.FmgpuError):
        fm.FMIndex.from_sequences([[1, 2]], 5, "IB16", 0)


@pytest.mark.parametrize("sigma,n", [(28, 6000), (5, 3000), (4, 900), (256, 2500), (21, 383), (6, 384 * 3)])
def test_gpu_builder_wavelet(sigma, n):
    """Wavelet indices built on the device (level-wise stable sort of the BWT) answer exactly like the reference's
    string::Wavelet (string/Wavelet.h:40-72 push_back construction) over the same BWT: rank / prefix_rank / symbol for every
    (row, symbol), exact and 1-mismatch searches, locate."""
    seqs = [make_text(n, sigma, seed=sigma + 3), make_text(n // 4 + 1, sigma, seed=sigma + 9)]
    gx = fm.BiFMIndex.from_sequences(seqs, sigma, "WAVELET", 4, keep_host=True)
    ox = fo.OraIndex.build("WAVELET", sigma, seqs, 4, True)
    N = ox.n
    assert gx.n == N
    for rev in (False, True):
        st = ox.bwt_string(rev=rev)
        ranks, pranks = st.rank_table()
        rows = np.arange(0, N + 1, max(1, (N + 1) * sigma // 400_000), dtype=np.uint64)
        idx = np.repeat(rows, sigma); sym = np.tile(np.arange(sigma, dtype=np.uint8), len(rows))
        assert np.array_equal(gx.rank(idx, sym, rev=rev).reshape(len(rows), sigma), ranks[rows.astype(np.int64)])
        assert np.array_equal(gx.prefix_rank(idx, sym, rev=rev).reshape(len(rows), sigma), pranks[rows.astype(np.int64)])
        r2 = np.arange(N, dtype=np.uint64)
        assert np.array_equal(gx.symbol(r2, rev=rev).astype(np.uint8), np.array([st.symbol(int(i)) for i in r2], dtype=np.uint8))
    reads = sample_reads(seqs[0], 300, 14, seed=8, mutate=1, sigma=sigma)
    qbuf, qoff = fm.flatten(reads)
    lb, ln = fm.search_no_errors.search(gx, (qbuf, qoff))
    olb, oln = ox.search_exact(qbuf, qoff)
    assert np.array_equal(lb, olb) and np.array_equal(ln, oln)
    sch = fm.search_scheme.h2(3, 0, 1)
    hits = fm.search_ng26.search(gx, reads, sch)
    assert same_hits(hits, ox.search_ng26(qbuf, qoff, sch)[0])
    rows = np.arange(0, N, max(1, N // 300), dtype=np.uint64)
    seq, pos, steps = gx.locate(rows)
    assert [(int(a), int(b), int(c)) for a, b, c in zip(seq, pos, steps)] == [ox.locate(int(r)) for r in rows]


@pytest.mark.parametrize("layout,sigma,rate", [("IB16", 5, 16), ("IB16", 5, 1), ("WAVELET", 28, 4), ("EPRV2_16", 5, 7)])
def test_locate_answer_table(layout, sigma, rate):
    """fmgpu_index_accelerate_locate: the per-row answer table returns the triples of the LF walk (fmindex/FMIndex.h:113-124), also for
    rows outside the index and after the table is dropped again"""
    seqs = [make_text(3000, sigma, seed=rate), make_text(41, sigma, seed=rate + 1), make_text(700, sigma, seed=rate + 2)]
    ox = fo.OraIndex.build(layout, sigma, seqs, rate, True)
    gx = gpu_index(ox)
    rows = np.concatenate([np.arange(ox.n, dtype=np.uint64), np.array([ox.n, ox.n + 5, 2 ** 40], dtype=np.uint64)])
    want = gx.locate(rows)
    before = gx.device_bytes
    gx.accelerate_locate()
    assert gx.device_bytes == before + 12 * ox.n
    *got, st = gx.locate(rows, want_stats=True)
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    assert st.lf_steps == int(want[2][: ox.n].sum())
    assert [(int(a), int(b), int(c)) for a, b, c in zip(*[g[: ox.n: 37] for g in got])] == [ox.locate(int(r)) for r in range(0, ox.n, 37)]
    gx.accelerate_locate(False)
    assert gx.device_bytes == before and all(np.array_equal(a, b) for a, b in zip(gx.locate(rows), want))


@pytest.mark.parametrize("layout,sigma,bidir", [("IB16", 5, True), ("IB16", 5, False), ("WAVELET", 28, False), ("EPRV2_16", 5, True), ("EPR16", 6, False), ("IB16", 28, True)])
def test_index_file_round_trip(layout, sigma, bidir, tmp_path):
    """fmgpu_index_save -> fmgpu_index_load (the library's own flat file): the loaded handle answers exact search, k-mismatch search, locate and String_c
    queries bit for bit like the one that was saved — plain, and with every optional table in the file; a damaged, truncated or foreign file is an
    error code and no handle"""
    seqs = repeat_text(5 + sigma, n=4000) if sigma == 5 else [make_text(3000, sigma, seed=sigma), make_text(500, sigma, seed=sigma + 1)]
    ox = fo.OraIndex.build(layout, sigma, seqs, 8, bidir)
    gx = gpu_index(ox)
    queries = mutated_queries(seqs, 400, 8, 60, 1, seed=3, sigma=sigma)
    qbuf, qoff = fm.flatten(queries)
    rows = np.arange(0, ox.n, 3, dtype=np.uint64)
    sch = fm.search_scheme.h2(3, 0, 1)

    def answers(x):
        out = [fm.search_no_errors.search(x, (qbuf, qoff)), x.locate(rows), (x.rank(rows, 1), x.prefix_rank(rows, sigma - 1), x.symbol(rows[:-1]))]
        if bidir:
            h = fm.search_ng26.search(x, (qbuf, qoff), sch)
            out.append(tuple(h[k] for k in HIT_KEYS))
        return out

    def same(a, b):
        return all(all(np.array_equal(u, v) for u, v in zip(p, q)) for p, q in zip(a, b))

    want = answers(gx)
    f1 = str(tmp_path / "plain.fmgpu")
    gx.save(f1, tables=False)
    lx = fm.FMIndex.load(f1)
    assert type(lx) is (fm.BiFMIndex if bidir else fm.FMIndex) and (lx.n, lx.Sigma, lx.row_bits) == (gx.n, gx.Sigma, gx.row_bits)
    assert same(answers(lx), want)
    # with tables: whatever the handle holds when it is saved
    gx.accelerate(2 if sigma <= 6 else 1, lut_len=3, walk=2)
    if bidir:
        gx.accelerate_search(4, 3)
    gx.accelerate_locate()
    f2 = str(tmp_path / "tables.fmgpu")
    gx.save(f2, tables=True)
    assert os.path.getsize(f2) > os.path.getsize(f1)
    tx = fm.FMIndex.load(f2)
    assert abs(tx.device_bytes - gx.device_bytes) < 4096 and same(answers(tx), want) and same(answers(gx), want)      # (the file also counts the tables' slack bytes)
    tx.save(str(tmp_path / "again.fmgpu"), tables=True)           # a loaded handle saves to the same bytes
    assert open(str(tmp_path / "again.fmgpu"), "rb").read() == open(f2, "rb").read()
    # damage: one flipped payload byte (checksum), a cut file (trailer / short read), a foreign header
    blob = bytearray(open(f1, "rb").read())
    for name, data in (("flip", bytes(blob[:len(blob) // 2]) + bytes([blob[len(blob) // 2] ^ 0x40]) + bytes(blob[len(blob) // 2 + 1:])),
                       ("cut", bytes(blob[:len(blob) - 24])), ("cut_more", bytes(blob[:len(blob) // 3])), ("foreign", b"CEREAL\0\0" + bytes(blob[8:]))):
        bad = str(tmp_path / (name + ".fmgpu"))
        open(bad, "wb").write(data)
        with pytest.raises(fm.FmgpuError) as e:
            fm.FMIndex.load(bad)
        assert e.value.code == capi.FMGPU_ERR_INVALID, name
    # a device-to-device copy of the handle with its tables (fmgpu_index_clone: what the replica set is made with) answers the same
    cx = gx.clone()
    assert abs(cx.device_bytes - gx.device_bytes) < 4096 and cx.formats == gx.formats and same(answers(cx), want)
    cx.close()
    # a description block that does not fit its own n / sigma / layouts — a stale or hand-edited file whose (unkeyed) checksum was recomputed — is refused before
    # anything is allocated: one size field each of the block table, the sampled suffix array, C, the interval table, and n itself
    import struct
    M64 = (1 << 64) - 1

    def mix(h, data):
        for i in range(0, len(data), 8):
            w = int.from_bytes(data[i:i + 8].ljust(8, b"\0"), "little")
            h = ((h ^ w) * 0x9e3779b97f4a7c15) & M64
            h ^= h >> 29
        return h
    tb = bytearray(open(f2, "rb").read())
    meta_bytes, meta_sum = struct.unpack_from("<QQ", tb, 24)
    assert mix(0x13198a2e03707344, bytes(tb[64:64 + meta_bytes])) == meta_sum      # (the test's restatement of the checksum is the library's)
    str0 = 64 + 8 + 16 + 258 * 8 + 5 * 8 + 64 + 4 * 8 + 8                            # offset of SavedIndex::str[0] (n, 4 ints, hC, sa_bytes, ViewSA, 4 sizes, lut_len + reserved)
    head0 = struct.unpack_from("<iiiiQ", tb, str0)                                   # layout, family, sigma, bitct, n of the bwt
    assert (head0[2], head0[4]) == (sigma, gx.n)
    edits = {"n": (64, lambda v: v + 64), "sa_bits_bytes": (64 + 8 + 16 + 258 * 8 + 2 * 8, lambda v: v // 2), "C_bytes": (64 + 8 + 16 + 258 * 8 + 5 * 8 + 64, lambda v: v + 8),
             "blk_bytes": (str0 + 24, lambda v: v // 2), "slut_bytes": (str0 + 24 + 6 * 8, lambda v: v * 2)}
    for name, (at, change) in edits.items():
        t2 = bytearray(tb)
        (v,) = struct.unpack_from("<Q", t2, at)
        struct.pack_into("<Q", t2, at, change(v))
        struct.pack_into("<Q", t2, 32, mix(0x13198a2e03707344, bytes(t2[64:64 + meta_bytes])))
        bad = str(tmp_path / ("edited_" + name + ".fmgpu"))
        open(bad, "wb").write(bytes(t2))
        with pytest.raises(fm.FmgpuError) as e:
            fm.FMIndex.load(bad)
        assert e.value.code == capi.FMGPU_ERR_INVALID and "description block" in str(e.value), (name, str(e.value))
    if layout == "IB16" and sigma == 5 and bidir:                 # 64-bit rows: its own file, refused by nothing but read by the wide build
        fm.options["force_wide"] = "1"
        try:
            wx = gpu_index(ox)
        finally:
            del fm.options["force_wide"]
        f3 = str(tmp_path / "wide.fmgpu")
        wx.save(f3)
        lw = fm.FMIndex.load(f3)
        assert lw.row_bits == 64 and same(answers(lw), want)


def test_replicas_shard_a_batch_over_devices(tmp_path):
    """fmgpu_replicas_* (one process, several devices): an index file loaded once per listed device, the batch cut into contiguous
    ranges — intervals, step counts and hit records equal the single-handle calls' on the whole batch.  One GPU here: the same device listed
    twice / three times exercises the sharding, the threads and the re-numbering of the queries (N > 1 devices: the same code, other ids)."""
    seqs = repeat_text(11, n=5000)
    ox = fo.OraIndex.build("IB16", 5, seqs, 4, True)
    gx = gpu_index(ox)
    path = str(tmp_path / "bi.fmgpu")
    gx.save(path)
    queries = mutated_queries(seqs, 1001, 20, 90, 2, seed=8)
    qbuf, qoff = fm.flatten(queries)
    olb, oln, ost = ox.search_exact(qbuf, qoff, want_steps=True)
    sch = fm.search_scheme.h2(4, 0, 2)
    want = fm.search_ng26.search(gx, (qbuf, qoff), sch)
    capi.check(capi.lib().fmgpu_hits_sort(capi.ptr(want), len(want), None))
    for devices in ([0], [0, 0], [0, 0, 0], None):
        r = fm.Replicas.load(path, devices)
        assert r.devices == (devices if devices else list(range(fm.device_count())))
        assert r.peer_copies == len(r.devices) - 1               # the file was read once: every other replica is a device-to-device copy of the first
        lb, ln, st = r.search_exact((qbuf, qoff), want_stats=True)
        assert np.array_equal(lb, olb) and np.array_equal(ln, oln) and st.lf_steps == int(ost.sum())
        hits, st = r.search_scheme((qbuf, qoff), sch, want_stats=True)
        capi.check(capi.lib().fmgpu_hits_sort(capi.ptr(hits), len(hits), None))
        assert same_hits(hits, want) and st.hits == len(want)
        small = r.search_scheme((qbuf, qoff), sch, capacity=len(want))          # exactly enough room for the whole call, not for an uneven share: grown inside
        assert len(small) == len(want)
        with pytest.raises(fm.FmgpuError):                                        # too small for the call: the caller's to grow (the wrapper's retry is off with n given)
            capi.check(capi.lib().fmgpu_replicas_search_scheme(r._r, capi.ptr(qbuf), capi.ptr(qoff), len(queries), C.byref(_scheme_struct(sch)), fm.UINT64_MAX,
                                                              capi.ptr(np.zeros(8, dtype=fm.HIT_DTYPE)), 8, C.byref(C.c_uint64()), None))
        eq = [q for q in queries if len(q) >= 40]
        eq = [q[:40] for q in eq]
        ex = fm.search_scheme.expand(fm.search_scheme.pigeon_opt(0, 1), 40)
        w21 = fm.search_ng21.search(gx, eq, ex)
        h21 = r.search_ng21(eq, ex)
        for h_ in (w21, h21):
            capi.check(capi.lib().fmgpu_hits_sort(capi.ptr(h_), len(h_), None))
        assert same_hits(h21, w21) and len(w21) > 0
        rows = np.concatenate([want["lb"][:300], olb[oln > 0][:300]]).astype(np.uint64)
        one = gx.locate(rows)
        many = r.locate(rows)
        assert all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(one, many))
        assert r.search_exact([])[0].size == 0 and len(r.search_scheme([], sch)) == 0 and r.locate([])[0].size == 0
        two = r.search_exact([queries[0], queries[1]])                            # fewer queries than replicas
        assert np.array_equal(two[0], olb[:2])
        dq = fm.DeviceBuffer.from_array(qbuf)
        with pytest.raises(fm.FmgpuError):                                        # device buffers belong to one device
            r.search_exact((dq, qoff))
        r.close()
    with pytest.raises(fm.FmgpuError):
        fm.Replicas.load(path, [0, 99])
    with pytest.raises(fm.FmgpuError):
        fm.Replicas.load(str(tmp_path / "missing.fmgpu"), [0])


def _scheme_struct(scheme):
    pi, l, u = (np.ascontiguousarray(x, dtype=np.uint64) for x in scheme)
    sc = capi.Scheme()
    sc.n_searches, sc.n_parts = pi.shape
    sc.pi, sc.l, sc.u = (x.ctypes.data_as(capi.u64p) for x in (pi, l, u))
    sc._keep = (pi, l, u)
    return sc


def test_index_create_argument_checks():
    text = make_text(500, 5, 3)
    ox = fo.OraIndex.build("IB16", 5, [text], 4, True)
    arr = oracle_arrays(ox)
    bad = dict(arr); bad["bwt"] = dict(arr["bwt"]); bad["bwt"]["blocks"] = arr["bwt"]["blocks"][:-8]
    with pytest.raises(fm.FmgpuError) as e:
        fm.BiFMIndex.from_reference_arrays(**bad)
    assert e.value.code == capi.FMGPU_ERR_INVALID
    other = fo.OraIndex.build("IB16", 5, [text[:100]], 4, True)
    bad = dict(arr); bad["bwt_rev"] = oracle_arrays(other)["bwt_rev"]
    with pytest.raises(fm.FmgpuError) as e:
        fm.BiFMIndex.from_reference_arrays(**bad)                  # fmindex/BiFMIndex.h:48-50
    assert "same size" in str(e.value)
    gx = fm.FMIndex.from_reference_arrays(bwt=arr["bwt"], C_array=arr["C_array"])
    with pytest.raises(fm.FmgpuError):
        gx.locate(np.array([0], dtype=np.uint64))                  # no annotated array given


# ------------------------------------------------------------------------------------------------ full-size properties
def test_full_size_properties():
    """BASELINE.json configs[1] sizes, size-independent properties: every unmutated read is found, a mutated read is found by
    k = 1, locate(row) walks back to the read's origin, the checksum of a second run is identical (idempotence)."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    n, nq, L = 3_088_286_401, 10_000_000, 101
    g = torch.Generator(device=dev); g.manual_seed(7)
    text = torch.empty(n, dtype=torch.uint8, device=dev)
    for lo in range(0, n, 1 << 28):
        hi = min(n, lo + (1 << 28))
        text[lo:hi] = torch.randint(1, 5, (hi - lo,), generator=g, device=dev, dtype=torch.uint8)
    starts = torch.randint(0, n - L, (nq,), generator=g, device=dev, dtype=torch.int64)
    reads = torch.empty((nq, L), dtype=torch.uint8, device=dev)
    ar = torch.arange(L, device=dev)
    for lo in range(0, nq, 1 << 20):
        hi = min(nq, lo + (1 << 20))
        reads[lo:hi] = text[starts[lo:hi, None] + ar[None, :]]
    seq_off = torch.tensor([0, n], dtype=torch.int64, device=dev)

    class V:
        def __init__(self, t):
            self.t, self.ptr, self.nbytes = t, t.data_ptr(), t.numel() * t.element_size()
    gx = fm.FMIndex.from_sequences((V(text), V(seq_off)), 5, "IB16", 16)
    del text
    qoff = torch.arange(nq + 1, device=dev, dtype=torch.int64) * L
    out = torch.zeros(2 * nq, dtype=torch.int64, device=dev)
    lbv, lnv = out[:nq], out[nq:]
    fm.search_no_errors.search(gx, (V(reads), V(qoff)), out=(V(lbv), V(lnv)))
    torch.cuda.synchronize()
    assert int((lnv >= 1).sum()) == nq                         # every read occurs where it was copied from
    chk1 = int((lbv * 31 + lnv).sum().item())
    uniq = torch.nonzero(lnv == 1).flatten()[:2_000_000]
    rows = lbv[uniq].contiguous()
    seq = torch.empty_like(rows); pos = torch.empty_like(rows); steps = torch.empty_like(rows)
    capi.check(capi.lib().fmgpu_locate(gx._h, C.c_void_p(rows.data_ptr()), rows.numel(), C.c_void_p(seq.data_ptr()),
                                       C.c_void_p(pos.data_ptr()), C.c_void_p(steps.data_ptr()), None, None))
    torch.cuda.synchronize()
    assert bool(torch.all(pos + steps == starts[uniq])) and bool(torch.all(seq == 0)) and int(steps.max()) < 16
    fm.search_no_errors.search(gx, (V(reads), V(qoff)), out=(V(lbv), V(lnv)))
    torch.cuda.synchronize()
    assert int((lbv * 31 + lnv).sum().item()) == chk1          # idempotent
    # one substitution in the middle: exact search misses (a 101-mer is unique in a random 3 Gbp text), the interval is empty
    reads[:, 50] = reads[:, 50] % 4 + 1
    fm.search_no_errors.search(gx, (V(reads), V(qoff)), out=(V(lbv), V(lnv)))
    torch.cuda.synchronize()
    assert int((lnv == 0).sum()) >= nq - 10
    miss_lb = lbv.clone()                                      # cursor of the step that emptied the interval: part of the result
    # ---- the optional tables do not change a single cursor at this size (rows close to 2^32 exercise the 32-bit row arithmetic)
    reads[:, 50] = (reads[:, 50] + 2) % 4 + 1                  # undo the substitution ((x % 4 + 1) is a 4-cycle: three more steps)
    for tables in ((3, 0, False), (3, 12, True), (1, 10, True), (3, 15, 2)):
        gx.accelerate(tables[0], lut_len=tables[1], walk=tables[2])
        fm.search_no_errors.search(gx, (V(reads), V(qoff)), out=(V(lbv), V(lnv)))
        torch.cuda.synchronize()
        assert int((lbv * 31 + lnv).sum().item()) == chk1, tables
    reads[:, 50] = reads[:, 50] % 4 + 1                        # mutated again: misses report the same cursor with every table
    fm.search_no_errors.search(gx, (V(reads), V(qoff)), out=(V(lbv), V(lnv)))
    torch.cuda.synchronize()
    assert bool(torch.equal(lbv, miss_lb)) and int((lnv == 0).sum()) >= nq - 10
    gx.accelerate_locate()
    seq2 = torch.empty_like(rows); pos2 = torch.empty_like(rows); steps2 = torch.empty_like(rows)
    capi.check(capi.lib().fmgpu_locate(gx._h, C.c_void_p(rows.data_ptr()), rows.numel(), C.c_void_p(seq2.data_ptr()),
                                       C.c_void_p(pos2.data_ptr()), C.c_void_p(steps2.data_ptr()), None, None))
    torch.cuda.synchronize()
    assert bool(torch.equal(seq, seq2) and torch.equal(pos, pos2) and torch.equal(steps, steps2))


class _V:
    def __init__(self, t):
        self.t, self.ptr, self.nbytes = t, t.data_ptr(), t.numel() * t.element_size()


def _oracle_from_built(gx, bidir, sigma=5, layout="IB16", threads=None):
    """the CPU restatement over the GPU-built BWT(s): what bench.py's cpu_baseline does"""
    ox = fo.OraIndex.from_bwt(layout, sigma, gx.built_array(0), gx.built_array(1) if bidir else None, None, None, None)
    ox.spread(threads or len(os.sched_getaffinity(0)))
    return ox


def _sorted_records(hits):
    return hits[np.lexsort((hits["seq"], hits["qidx"]))]


def test_full_size_exact_records_equal_the_cpu_walk():
    """BASELINE.json configs[1] at full index size on the repeat-structured text (what bench.py's headline runs): 200 k reads of 101 bp cut from the text, every
    tenth with one substitution — the interval, the row of a miss and the step count of every read from the two-symbol-step kernel (k_exact_p), from the same kernel
    behind a 12-symbol interval table, from the one-symbol kernel (k_exact_a) and from the CPU walk over the GPU-built BWT are equal; and, an anchor that does not
    pass through that BWT: every unchanged read that lies inside one sequence is found at the text offset it was cut from (locate of its interval)."""
    torch = pytest.importorskip("torch")
    from fmindex_collection_amd import datasets
    import bench
    dev = torch.device("cuda", 0)
    lengths = list(bench.GRCH38_LENGTHS)
    text, stats = datasets.genome_like_text(lengths, seed=42, device=dev)
    n = int(text.numel())
    seq_off = torch.from_numpy(np.concatenate([[0], np.cumsum(np.asarray(lengths, dtype=np.int64))])).to(dev)
    g = torch.Generator(device=dev); g.manual_seed(23)
    L, nq = 101, 200_000
    starts = torch.randint(0, n - L, (nq,), generator=g, device=dev, dtype=torch.int64)
    reads = text[starts[:, None] + torch.arange(L, device=dev)[None, :]].contiguous()
    sel = torch.arange(0, nq, 10, device=dev)
    p = torch.randint(0, L, (sel.numel(),), generator=g, device=dev)
    reads[sel, p] = reads[sel, p] % 4 + 1
    hq, ho = reads.reshape(-1).cpu().numpy(), np.arange(nq + 1, dtype=np.uint64) * L
    org = starts.cpu().numpy()
    with fm.options(lf_table=0):
        gx = fm.FMIndex.from_sequences((_V(text), _V(seq_off)), 5, "IB16", 16, keep_host=True)
    del text, reads
    assert bool(gx.formats & capi.FMT_PAIRS) and bool(gx.formats & capi.FMT_FUSED)
    ox = _oracle_from_built(gx, False)
    olb, oln, ost = ox.search_exact(hq, ho, want_steps=True, nthreads=len(os.sched_getaffinity(0)))
    assert int((oln == 0).sum()) > nq // 20                      # (most of the changed reads miss: the rows and step counts of misses are compared too)

    def check(tag, sel_bits=0):
        with fm.options(kernel_select=sel_bits):
            lb, ln, st = fm.search_no_errors.search(gx, (hq, ho), want_stats=True)
        assert np.array_equal(ln, oln) and np.array_equal(lb, olb) and st.lf_steps == int(ost.sum()), tag
        return st
    check("pair steps")
    check("one-symbol steps", capi.SEL_EXACT_ONE_SYMBOL)
    # the text-side anchor: unchanged reads inside one sequence, intervals of at most 4096 rows
    seq_off_h = seq_off.cpu().numpy()
    unchanged = np.ones(nq, dtype=bool); unchanged[::10] = False
    inside = seq_off_h[np.searchsorted(seq_off_h, org, side="right")] >= org + L
    use = np.nonzero(unchanged & inside & (oln > 0) & (oln <= 4096))[0][:30_000]
    cnt = oln[use].astype(np.int64)
    owner = np.repeat(np.arange(use.size), cnt)
    rows = np.repeat(olb[use].astype(np.int64), cnt) + (np.arange(owner.size) - np.repeat(np.cumsum(cnt) - cnt, cnt))
    seq, pos, steps = gx.locate(rows.astype(np.uint64))
    tpos = seq_off_h[seq.astype(np.int64)] + pos.astype(np.int64) + steps.astype(np.int64)
    found = np.zeros(use.size, dtype=bool)
    np.logical_or.at(found, owner, tpos == org[use][owner])
    assert use.size > 20_000 and bool(found.all()), (int(use.size), int((~found).sum()))
    with fm.options(kernel_select=capi.SEL_LOCATE_PER_LANE):     # (the one-row-per-lane locate kernel agrees)
        seq1, pos1, steps1 = gx.locate(rows[:200_000].astype(np.uint64))
    assert np.array_equal(seq1, seq[:200_000]) and np.array_equal(pos1, pos[:200_000]) and np.array_equal(steps1, steps[:200_000])
    # the same search behind the 12-symbol interval table (134 MB): results and step counts unchanged, 12 steps per read served by one entry
    before = gx.device_bytes
    gx.accelerate(1, lut_len=12, walk=0)
    assert gx.device_bytes - before == 4 ** 12 * 8
    st = check("interval table + pair steps")
    assert st.table_steps > 0.95 * 12 * nq and st.table_steps % 12 == 0
    check("interval table, table-driven kernel", capi.SEL_NO_EXACT_LUT)
    gx.close()


def test_full_size_k2_records_equal_the_cpu_walk():
    """BASELINE.json configs[2] and configs[3] at full index size (3.09 Gbp BiFMIndex) on the repeat-structured text: the oracle is built from the
    GPU-built BWTs and every hit record (qidx, lb, lb_rev, len, errors, callback order) of 100 k reads at 101 bp and at 151 bp is compared with
    the CPU walk — with the plain index (blocks only), with LF tables, and with every accelerator (16-symbol prefix table = 2^32 entries, rows
    close to 2^32); node counts too.  The general kernel must agree as well."""
    torch = pytest.importorskip("torch")
    from fmindex_collection_amd import datasets
    import bench
    dev = torch.device("cuda", 0)
    lengths = list(bench.GRCH38_LENGTHS)
    text, stats = datasets.genome_like_text(lengths, seed=42, device=dev)
    n = int(text.numel())
    seq_off = torch.from_numpy(np.concatenate([[0], np.cumsum(np.asarray(lengths, dtype=np.int64))])).to(dev)
    g = torch.Generator(device=dev); g.manual_seed(11)
    batches, origins = {}, {}
    for L, nq in ((101, 100_000), (151, 100_000)):
        starts = torch.randint(0, n - L, (nq,), generator=g, device=dev, dtype=torch.int64)
        origins[L] = starts.cpu().numpy()
        reads = text[starts[:, None] + torch.arange(L, device=dev)[None, :]].contiguous()
        rows = torch.arange(nq, device=dev)
        for k in range(2):                                     # 0 / 1 / 2 substitutions
            sel = rows[rows % 3 > k]
            p = torch.randint(0, L, (sel.numel(),), generator=g, device=dev)
            reads[sel, p] = reads[sel, p] % 4 + 1
        batches[L] = (reads.reshape(-1).cpu().numpy(), (np.arange(nq + 1, dtype=np.uint64) * L))
    fm.options["lf_table"] = "0"
    try:
        gx = fm.BiFMIndex.from_sequences((_V(text), _V(seq_off)), 5, "IB16", 16, keep_host=True)
    finally:
        del fm.options["lf_table"]
    del text
    ox = _oracle_from_built(gx, True)
    sch = fm.search_scheme.h2(4, 0, 2)
    want = {}
    for L, (hq, ho) in batches.items():
        oh, _, onodes = ox.search_ng26(hq, ho, sch, nthreads=len(os.sched_getaffinity(0)), records=True, cap=1 << 22)
        want[L] = (oh, onodes)
        assert int(oh["errors"].max()) == 2 and len(oh) >= len(ho) - 1 - 100

    def check(tag):
        for L, (hq, ho) in batches.items():
            hits, st = fm.search_ng26.search(gx, (hq, ho), sch, want_stats=True)
            oh, onodes = want[L]
            assert st.lf_steps == onodes, (tag, L, st.lf_steps, onodes)
            assert same_hits(hits, oh), (tag, L)
    check("plain index")
    # The same batches again and again, in slices small enough that most waves of the chip wait at the board (work sharing between the waves of a launch) in every
    # launch, Hamming and edit distance, with the board and without: always the same records (a lost or doubled subtree would change their number; a waiting wave that gave up is an error)
    for L, (hq, ho) in batches.items():
        for edit in (False, True):
            nq_s = 30_000 if edit else 100_000
            sl = (hq[: nq_s * L], ho[: nq_s + 1])
            with fm.options(kernel_select=capi.SEL_NO_BOARD):
                ref_hits, ref_st = fm.search_ng26.search(gx, sl, sch, want_stats=True, edit=edit, capacity=1 << 23)
            for _ in range(12):
                h, st = fm.search_ng26.search(gx, sl, sch, want_stats=True, edit=edit, capacity=1 << 23)
                assert st.lf_steps == ref_st.lf_steps and same_hits(h, ref_hits), (L, edit)
    # An anchor that does not pass through the GPU-built BWT the oracle was fed: every read was cut from the TEXT at a known offset with <= 2
    # substitutions, so that offset must be among the located positions of its e <= 2 hits (reads that straddle two sequences, and the few
    # reads of high-copy repeats whose hits cover more than 4096 rows, are left out)
    seq_off_h = seq_off.cpu().numpy()
    for L, (hq, ho) in batches.items():
        nq_a = 3000
        hits = fm.search_ng26.search(gx, (hq[: nq_a * L], ho[: nq_a + 1]), sch)
        org = origins[L][:nq_a]
        inside = seq_off_h[np.searchsorted(seq_off_h, org, side="right")] >= org + L        # the read lies within one sequence
        per_read = np.bincount(hits["qidx"].astype(np.int64), weights=hits["len"].astype(np.float64), minlength=nq_a)
        use = inside & (per_read <= 4096)
        hsel = hits[use[hits["qidx"].astype(np.int64)]]
        owner = np.repeat(hsel["qidx"].astype(np.int64), hsel["len"].astype(np.int64))
        first = np.repeat(hsel["lb"].astype(np.int64), hsel["len"].astype(np.int64))
        within = np.arange(owner.size) - np.repeat(np.cumsum(hsel["len"].astype(np.int64)) - hsel["len"].astype(np.int64), hsel["len"].astype(np.int64))
        seq, pos, steps = gx.locate((first + within).astype(np.uint64))
        tpos = seq_off_h[seq.astype(np.int64)] + pos.astype(np.int64) + steps.astype(np.int64)
        found = np.zeros(nq_a, dtype=bool)
        np.logical_or.at(found, owner, tpos == org[owner])
        assert use.sum() > nq_a * 0.9 and bool(found[use].all()), (L, int(use.sum()), int((~found[use]).sum()))
    gx.accelerate_lf(True)
    check("LF tables")
    for accel in ((11, 1), (0, 2), (16, 3)):
        gx.accelerate_search(*accel)
        check(accel)
    fm.options["kernel_select"] = "2"                        # the general kernel
    try:
        check("general kernel")
    finally:
        del fm.options["kernel_select"]
    # edit distance at full size (20 k reads, single-threaded CPU walk): the table-driven kernel with its path keys, work sharing and the
    # heavy-reads-first hand-out order (the batch is below its 64 k threshold: the kernel selection keeps the order on for a repeated batch)
    hq, ho = batches[101]
    eq, eo = hq[: 20_000 * 101], ho[: 20_001]
    oe, _, enodes = ox.search_ng26(eq, eo, sch, edit=True, cap=1 << 22)
    ehits, est = fm.search_ng26.search(gx, (eq, eo), sch, want_stats=True, edit=True, capacity=1 << 22)
    assert est.lf_steps == enodes and same_hits(ehits, oe)
    rep = 4                                                    # 80 k reads: the sampled, reordered hand-out
    rq = np.tile(eq, rep); ro = (np.arange(20_000 * rep + 1, dtype=np.uint64) * 101)
    rhits, rst = fm.search_ng26.search(gx, (rq, ro), sch, want_stats=True, edit=True, capacity=1 << 24)
    assert rst.lf_steps == rep * enodes and len(rhits) == rep * len(oe)
    first = rhits[rhits["qidx"] < 20_000]
    assert same_hits(first, oe)


def test_gpu_builder_on_a_repeat_structured_text():
    """fmgpu_build_index on 51 Mbp of the genome-like text (interspersed repeats, satellites, 2.5 % of it in one run of one symbol per sequence:
    prefix doubling over long ties): the BWT and the located positions equal the oracle's suffix array, which is itself checked for sortedness
    with the inverse-permutation criterion (suffix(a) < suffix(b) <=> (text[a], isa[a+1]) < (text[b], isa[b+1]))"""
    torch = pytest.importorskip("torch")
    from fmindex_collection_amd import datasets
    import bench
    lengths = [max(1, int(l * 0.0165)) for l in bench.GRCH38_LENGTHS]
    text, stats = datasets.genome_like_text(lengths, seed=42, device=torch.device("cuda", 0))
    assert 0.35 < stats["repeat_fraction_written"] < 0.5 and stats["run_fraction_written"] > 0.04
    host = text.cpu().numpy()
    seqs = np.split(host, np.cumsum(lengths)[:-1])
    gx = fm.FMIndex.from_sequences(seqs, 5, "IB16", 16, keep_host=True)
    # the text as the index sees it: every sequence followed by a delimiter (utils.h:382-411)
    cat = np.concatenate([np.concatenate([s_, np.zeros(1, dtype=np.uint8)]) for s_ in seqs])
    n = cat.size
    assert gx.n == n
    sa = np.empty(n, dtype=np.uint64)
    assert fo.lib().ora_suffix_array(cat.ctypes.data_as(C.POINTER(C.c_uint8)), n, sa.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
    sai = sa.astype(np.int64)
    isa = np.empty(n, dtype=np.int64); isa[sai] = np.arange(n)
    a, b = sai[:-1], sai[1:]
    nxa = np.where(a + 1 < n, isa[np.minimum(a + 1, n - 1)], -1); nxb = np.where(b + 1 < n, isa[np.minimum(b + 1, n - 1)], -1)
    assert bool(((cat[a] < cat[b]) | ((cat[a] == cat[b]) & (nxa < nxb))).all())          # the oracle's suffix array is sorted
    assert np.array_equal(gx.built_array(0), cat[(sai + n - 1) % n])                       # bwt[i] = text[(sa[i] + n - 1) % n], utils.h:145-163
    rows = np.random.default_rng(3).integers(0, n, size=200_000).astype(np.uint64)
    seq, pos, steps = gx.locate(rows)
    starts = np.concatenate([[0], np.cumsum(np.asarray(lengths) + 1)])
    assert np.array_equal(starts[seq.astype(np.int64)] + pos.astype(np.int64) + steps.astype(np.int64), sai[rows.astype(np.int64)])


def test_rows_beyond_2_32():
    """An index of 4.4e9 rows (64-bit-row build, real size): built on the GPU, the oracle is built from its BWTs; exact intervals, 1-mismatch hit
    records and located positions of 10 k reads equal the CPU walk, and every sampled read locates back to where it was copied from."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    lengths = [3_000_000_000, 1_400_000_123]
    n_sym = sum(lengths)
    g = torch.Generator(device=dev); g.manual_seed(5)
    text = torch.empty(n_sym, dtype=torch.uint8, device=dev)
    for lo in range(0, n_sym, 1 << 28):
        hi = min(n_sym, lo + (1 << 28))
        text[lo:hi] = torch.randint(1, 5, (hi - lo,), generator=g, device=dev, dtype=torch.uint8)
    L, nq = 60, 10_000
    starts = torch.cat([torch.randint(0, lengths[0] - L, (nq // 2,), generator=g, device=dev, dtype=torch.int64),
                        lengths[0] + torch.randint(0, lengths[1] - L, (nq // 2,), generator=g, device=dev, dtype=torch.int64)])
    reads = text[starts[:, None] + torch.arange(L, device=dev)[None, :]].contiguous()
    mut = torch.arange(0, nq, 4, device=dev)
    reads[mut, 17] = reads[mut, 17] % 4 + 1                    # every fourth read carries one substitution
    seq_off = torch.tensor([0, lengths[0], n_sym], dtype=torch.int64, device=dev)
    gx = fm.BiFMIndex.from_sequences((_V(text), _V(seq_off)), 5, "IB16", 16, keep_host=True)
    del text
    torch.cuda.empty_cache()
    assert gx.row_bits == 64 and gx.n == n_sym + 2 and gx.n > 2**32
    ox = _oracle_from_built(gx, True)
    hq, ho = reads.reshape(-1).cpu().numpy(), np.arange(nq + 1, dtype=np.uint64) * L
    lb, ln = fm.search_no_errors.search(gx, (hq, ho))
    olb, oln = ox.search_exact(hq, ho, nthreads=8)
    assert np.array_equal(lb, olb) and np.array_equal(ln, oln)
    assert int(lb.max()) > 2**32 and int((ln > 0).sum()) == nq - len(mut)
    hit = np.nonzero(ln == 1)[0]
    seq, pos, steps = gx.locate(lb[hit])
    st_host = starts.cpu().numpy()
    want_seq = (st_host[hit] >= lengths[0]).astype(np.uint64)
    assert np.array_equal(seq, want_seq) and np.array_equal(pos + steps, (st_host[hit] - want_seq.astype(np.int64) * lengths[0]).astype(np.uint64))
    assert int(steps.max()) < 16                               # every 16th position is sampled — in ALL rows (a launch of 2^32 threads and more is cut short without an error)
    sch = fm.search_scheme.h2(3, 0, 1)
    hits, st = fm.search_ng26.search(gx, (hq, ho), sch, want_stats=True)
    oh, _, onodes = ox.search_ng26(hq, ho, sch, nthreads=8, records=True)
    assert same_hits(hits, oh) and st.lf_steps == onodes and len(hits) >= nq
    idx = np.array([0, 1, 2**32 - 1, 2**32, 2**32 + 12345, gx.n - 1, gx.n], dtype=np.uint64)
    for c in range(5):
        want = np.array([ox.bwt_string().rank(int(i), c) for i in idx], dtype=np.uint64)
        assert np.array_equal(gx.rank(idx, c), want)


def test_full_size_protein_wavelet_equals_the_cpu_walk():
    """BASELINE.json configs[4] at full index size (2e9 residues, sigma = 28, FMIndex<28, Wavelet>): the oracle's Wavelet is built from the GPU-built
    BWT and the (lb, len) of 100 k reads x 40 aa — every tenth with one substitution — are compared, on the multi-ary tree itself and on its
    block-table expansion with the exact-search tables; located origins too."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    nseq, slen, sigma, L, nq = 4_000_000, 500, 28, 40, 100_000
    n = nseq * slen
    g = torch.Generator(device=dev); g.manual_seed(42)
    text = torch.empty(n, dtype=torch.uint8, device=dev)
    for lo in range(0, n, 1 << 28):
        hi = min(n, lo + (1 << 28))
        text[lo:hi] = torch.randint(1, sigma, (hi - lo,), generator=g, device=dev, dtype=torch.uint8)
    seq_id = torch.randint(0, nseq, (nq,), generator=g, device=dev, dtype=torch.int64)
    off = torch.randint(0, slen - L + 1, (nq,), generator=g, device=dev, dtype=torch.int64)
    reads = text[(seq_id * slen + off)[:, None] + torch.arange(L, device=dev)[None, :]].contiguous()
    mut = torch.arange(0, nq, 10, device=dev)
    reads[mut, 7] = reads[mut, 7] % (sigma - 1) + 1
    seq_off = torch.arange(nseq + 1, device=dev, dtype=torch.int64) * slen
    gx = fm.FMIndex.from_sequences((_V(text), _V(seq_off)), sigma, "WAVELET", 16, keep_host=True)
    del text
    torch.cuda.empty_cache()
    ox = _oracle_from_built(gx, False, sigma=sigma, layout="WAVELET")
    hq, ho = reads.reshape(-1).cpu().numpy(), np.arange(nq + 1, dtype=np.uint64) * L
    olb, oln = ox.search_exact(hq, ho, nthreads=len(os.sched_getaffinity(0)))
    lb, ln, st = fm.search_no_errors.search(gx, (hq, ho), want_stats=True)
    assert np.array_equal(lb, olb) and np.array_equal(ln, oln) and st.table_accesses > 0
    assert int((ln > 0).sum()) >= nq - len(mut)
    hit = np.nonzero(ln == 1)[0][:20000]
    sq, pos, steps = gx.locate(lb[hit])
    assert np.array_equal(sq, seq_id.cpu().numpy()[hit].astype(np.uint64)) and np.array_equal(pos + steps, off.cpu().numpy()[hit].astype(np.uint64))
    gx.accelerate(1, lut_len=6, walk=2)
    lb2, ln2 = fm.search_no_errors.search(gx, (hq, ho))
    assert np.array_equal(lb2, olb) and np.array_equal(ln2, oln)


def test_failed_table_requests_leave_the_handle_as_it_was():
    """a table that cannot be built is refused with an error code — FMGPU_ERR_UNSUPPORTED for more than 2^32 entries, FMGPU_ERR_NOMEM / _HIP for an
    allocation that does not fit — and neither device_bytes nor the results change (tables are built into local buffers and installed after success)"""
    text = make_text(50_000, 5, seed=2)
    ox = fo.OraIndex.build("IB16", 5, [text], 16, True)
    gx = gpu_index(ox)
    qbuf, qoff = fm.flatten(sample_reads(text, 500, 30, seed=1, mutate=1))
    want = fm.search_no_errors.search(gx, (qbuf, qoff))
    before = gx.device_bytes
    for call in (lambda: gx.accelerate_search(17, 0), lambda: gx.accelerate(3, lut_len=17), lambda: gx.accelerate(9)):
        with pytest.raises(fm.FmgpuError) as ei:
            call()
        assert ei.value.code in (capi.FMGPU_ERR_UNSUPPORTED, capi.FMGPU_ERR_INVALID)
    gx.accelerate_search(0, 0); gx.accelerate(1)
    dbytes = C.c_uint64()
    capi.check(capi.lib().fmgpu_index_info(gx._h, None, None, None, None, C.byref(dbytes)))
    assert dbytes.value == before
    got = fm.search_no_errors.search(gx, (qbuf, qoff))
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


def test_call_scratch_allocation_failure_is_an_error_not_a_fault():
    """the option fail_scratch = k (FMGPU_OPT_FAIL_SCRATCH) fails the k-th allocation of the per-thread call scratch: the call returns an error code, nothing half-initialised
    stays behind, and the next call (on a fresh host thread, whose scratch is created anew) works"""
    import threading
    text = make_text(20_000, 5, seed=5)
    ox = fo.OraIndex.build("IB16", 5, [text], 16, False)
    gx = gpu_index(ox)
    qbuf, qoff = fm.flatten(sample_reads(text, 200, 25, seed=2))
    want = ox.search_exact(qbuf, qoff)
    out = {}

    def worker(tag, env):
        if env:
            fm.options["fail_scratch"] = env
        try:
            capi.check(capi.lib().fmgpu_set_device(0))
            try:
                out[tag] = fm.search_no_errors.search(gx, (qbuf, qoff), want_stats=True)[:2]
            except fm.FmgpuError as ex:
                out[tag] = ex
                out[tag + "_retry"] = None
                fm.options.pop("fail_scratch")
                out[tag + "_retry"] = fm.search_no_errors.search(gx, (qbuf, qoff), want_stats=True)[:2]    # same thread: the scratch is built now
        finally:
            fm.options.pop("fail_scratch")
    for k in ("1", "4", "6"):
        t = threading.Thread(target=worker, args=("fail" + k, k)); t.start(); t.join()
        assert isinstance(out["fail" + k], fm.FmgpuError) and out["fail" + k].code in (capi.FMGPU_ERR_NOMEM, capi.FMGPU_ERR_HIP)
        r = out["fail" + k + "_retry"]
        assert r is not None and np.array_equal(r[0], want[0]) and np.array_equal(r[1], want[1])


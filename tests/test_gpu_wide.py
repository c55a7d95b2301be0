"""The 64-bit-row build of the kernels (indices of 2^32 rows or more; the reference switches to libsais64 at 2^31 rows, utils.h:243-247),
exercised on small inputs: the option force_wide = 1 (FMGPU_OPT_FORCE_WIDE) routes a new index to it whatever its size, so every kernel of that build is compared with
the oracle exactly like its 32-bit twin.  The test at real size (n > 2^32) is test_gpu_parity.py::test_rows_beyond_2_32."""
import contextlib
import os

import numpy as np
import pytest

import fmoracle as fo
import fmindex_collection_amd as fm
from fmindex_collection_amd import capi
from tests.util import make_text, sample_reads, oracle_arrays, string_arrays
from tests.test_gpu_parity import LAYOUTS, same_hits, repeat_text, mutated_queries

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def force_wide():
    with fm.options(force_wide=1):
        yield


def wide_index(ox):
    cls = fm.BiFMIndex if ox.bidirectional else fm.FMIndex
    with force_wide():
        gx = cls.from_reference_arrays(**oracle_arrays(ox))
    assert gx.row_bits == 64
    return gx


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("sigma", [5, 28, 255])
def test_wide_string_concept(layout, sigma):
    for n in (1, 64, 65, 300, 1300) if sigma != 255 else (65, 300):
        text = make_text(n, sigma, seed=n + sigma, lo=0)
        s = fo.OraString(layout, sigma, text)
        C_arr = np.array([int(np.count_nonzero(text < c)) for c in range(sigma + 1)], dtype=np.uint64)
        with force_wide():
            gx = fm.FMIndex.from_reference_arrays(bwt=string_arrays(s), C_array=C_arr)
        assert gx.row_bits == 64
        ork, opr = s.rank_table()
        idx = np.repeat(np.arange(n + 1, dtype=np.uint64), sigma)
        sym = np.tile(np.arange(sigma, dtype=np.uint8), n + 1)
        assert np.array_equal(gx.rank(idx, sym).reshape(n + 1, sigma), ork), (layout, sigma, n)
        assert np.array_equal(gx.prefix_rank(idx, sym).reshape(n + 1, sigma), opr), (layout, sigma, n)
        assert np.array_equal(gx.symbol(np.arange(n, dtype=np.uint64)), text.astype(np.uint64))


@pytest.mark.parametrize("layout,sigma", [("IB16", 5), ("IBP16", 5), ("EPR16", 5), ("EPRV2_16", 5), ("WAVELET", 5), ("WAVELET", 28), ("WAVELET", 256), ("IB16", 28), ("EPRV5", 6), ("FBV_512_64K", 21)])
def test_wide_exact_search_and_locate(layout, sigma):
    text = make_text(40_000, sigma, seed=3)
    ox = fo.OraIndex.build(layout, sigma, [text[:25_000], text[25_000:]], 8, False)
    gx = wide_index(ox)
    reads = sample_reads(text, 3000, 24, seed=9, mutate=1, sigma=sigma)
    qbuf, qoff = fm.flatten(reads)
    lb, ln, st = fm.search_no_errors.search(gx, (qbuf, qoff), want_stats=True)
    olb, oln = ox.search_exact(qbuf, qoff)
    assert np.array_equal(lb, olb) and np.array_equal(ln, oln)
    rows = np.concatenate([lb[ln > 0][:500], np.arange(0, 400, dtype=np.uint64)])
    seq, pos, steps = gx.locate(rows)
    for k in range(0, rows.size, 7):
        assert (seq[k], pos[k], steps[k]) == ox.locate(int(rows[k]))
    # what only the 32-bit-row build offers says so instead of misbehaving
    for call in (lambda: gx.accelerate(3), lambda: gx.accelerate_locate(), lambda: fm.search_no_errors.search_packed(gx, (qbuf, qoff))):
        with pytest.raises(fm.FmgpuError) as ei:
            call()
        assert ei.value.code == capi.FMGPU_ERR_UNSUPPORTED
    # the exact-search tables of 64-bit rows (16-byte entries): interval table, LF^J walk, LF^2J walk — on the native layout or its Format A expansion;
    # cursor and executed steps are those of the CPU walk whatever the tables
    osteps = ox.search_exact(qbuf, qoff, want_steps=True)[2]
    long_reads = sample_reads(text, 1500, 70 if sigma <= 6 else 40, seed=19, mutate=1, sigma=sigma)
    lq, lo_ = fm.flatten(long_reads)
    llb, lln, lsteps = ox.search_exact(lq, lo_, want_steps=True)
    for kstep, lut_len, walk in ((1, 4, 0), (1, 0, 1), (1, 3, 2), (0, 2, 1)):
        if sigma > 32 and lut_len > 2:
            lut_len = 2
        gx.accelerate(kstep, lut_len=lut_len, walk=walk)
        lb2, ln2, st2 = fm.search_no_errors.search(gx, (qbuf, qoff), want_stats=True)
        assert np.array_equal(lb2, olb) and np.array_equal(ln2, oln) and st2.lf_steps == int(osteps.sum()), (kstep, lut_len, walk)
        lb3, ln3, st3 = fm.search_no_errors.search(gx, (lq, lo_), want_stats=True)
        assert np.array_equal(lb3, llb) and np.array_equal(ln3, lln) and st3.lf_steps == int(lsteps.sum()), (kstep, lut_len, walk)
    gx.accelerate(0)


@pytest.mark.parametrize("layout,sigma,k", [("IB16", 5, 1), ("IB16", 5, 2), ("EPRV2_16", 5, 2), ("WAVELET", 28, 1), ("IB16", 256, 1), ("IB8", 6, 2)])
@pytest.mark.parametrize("lf", [True, False])
def test_wide_k_mismatch_and_backtracking(layout, sigma, k, lf):
    seqs = repeat_text(4) if sigma == 5 else [make_text(2500, sigma, seed=8), make_text(700, sigma, seed=9)]
    ox = fo.OraIndex.build(layout, sigma, seqs, 4, True)
    gx = wide_index(ox)
    if not lf:
        gx.accelerate_lf(False)
    queries = mutated_queries(seqs, 300, 12, 40, k, seed=21, sigma=sigma)
    qbuf, qoff = fm.flatten(queries)
    sch = fm.search_scheme.h2(k + 2, 0, k)
    hits, st = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True)
    ohits, _, onodes = ox.search_ng26(qbuf, qoff, sch)
    assert same_hits(hits, ohits) and st.lf_steps == onodes
    few = fm.flatten(queries[:60])
    bh = fm.search_backtracking.search(gx, few, k)
    obh = ox.search_backtracking(few[0], few[1], k)[0]
    assert same_hits(bh, obh)
    # cursor steps
    lb0, rev0, len0 = hits["lb"][:50], hits["lb_rev"][:50], hits["len"][:50]
    for right in (False, True):
        olb, orev, olen = gx.extend(lb0, rev0, len0, None, right=right)
        for i in range(0, len(lb0), 5):
            cur = fo.Cursor(int(lb0[i]), int(rev0[i]), int(len0[i]))
            exp = ox.extend_right_all(cur) if right else ox.extend_left_all(cur)
            assert [tuple(map(int, t)) for t in zip(olb[i], orev[i], olen[i])] == [(c.lb, c.lb_rev, c.len) for c in exp]


@pytest.mark.parametrize("k,length", [(1, 20), (2, 31), (2, 101), (2, 151), (2, 255), (0, 40)])
def test_wide_lean_kernel(k, length):
    """equal-length Hamming batches on a 64-bit-row BiFMIndex<5> take k_scheme_lean too (16-byte frames of 38-bit rows, super-block counts from LDS, 7-word
    hit records): records in callback order and node counts equal the CPU walk and the general kernel, delimiters in the reads included"""
    seqs = repeat_text(90 + k, n=9000) + [np.tile(np.array([1, 1, 1, 2], dtype=np.uint8), 300), np.full(700, 3, dtype=np.uint8)]
    ox = fo.OraIndex.build("IB16", 5, seqs, 4, True)
    gx = wide_index(ox)
    queries = mutated_queries([q for q in seqs if len(q) > length], 2500, length, length + 1, k + 1, seed=31 + k)
    rng = np.random.default_rng(k)
    for i in range(0, len(queries), 89):
        queries[i][int(rng.integers(0, length))] = 0
    qbuf, qoff = fm.flatten(queries)
    schemes = [fm.search_scheme.h2(k + 2, 0, k), fm.search_scheme.pigeon_opt(0, k)] if k else [fm.search_scheme.backtracking(1, 0, 0)]
    for sch in schemes:
        ohits, _, nodes = ox.search_ng26(qbuf, qoff, sch, cap=1 << 24)
        hits, st = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, capacity=1 << 24)
        assert same_hits(hits, ohits) and st.lf_steps == nodes, (k, length)
        fm.options["kernel_select"] = "2"                        # the general kernel
        try:
            hits2, st2 = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, capacity=1 << 24)
        finally:
            del fm.options["kernel_select"]
        assert same_hits(hits2, ohits) and st2.lf_steps == nodes


@pytest.mark.parametrize("layout,sigma,k", [("IB16", 5, 1), ("IB16", 5, 2), ("EPRV2_16", 5, 2), ("WAVELET", 28, 1), ("IB16", 256, 1), ("IB16", 5, 3)])
@pytest.mark.parametrize("lf", [True, False])
def test_wide_edit_distance_and_ng21(layout, sigma, k, lf):
    """search_ng26<Edit = true> and search_ng21 on 64-bit rows (48-byte frames): records, callback order and extension counts of the CPU walk,
    search_n clipping, ragged batches"""
    rng = np.random.default_rng(500 + sigma + k)
    hi = min(sigma, 8)
    base = rng.integers(1, hi, size=1500, dtype=np.uint8)
    seqs = [np.concatenate([base, base[300:800]]), rng.integers(1, hi, size=400, dtype=np.uint8)]
    ox = fo.OraIndex.build(layout, sigma, seqs, 4, True)
    gx = wide_index(ox)
    if not lf:
        gx.accelerate_lf(False)
    length, queries = 26, []
    for i in range(300 if k < 3 else 80):
        p = int(rng.integers(0, len(seqs[0]) - length - 4)); q = list(seqs[0][p: p + length + 3])
        for _ in range(int(rng.integers(0, k + 2))):
            op = int(rng.integers(0, 3)); jj = int(rng.integers(0, len(q)))
            if op == 0: q[jj] = int(rng.integers(1, hi))
            elif op == 1: q.insert(jj, int(rng.integers(1, hi)))
            else: del q[jj]
        queries.append(np.array(q[:length], dtype=np.uint8))
    qbuf, qoff = fm.flatten(queries)
    total = 0
    for sch in (fm.search_scheme.h2(k + 2, 0, k), fm.search_scheme.pigeon_opt(0, k)):
        hits, st = fm.search_ng26.search(gx, (qbuf, qoff), sch, want_stats=True, edit=True, capacity=1 << 21)
        ohits, _, nodes = ox.search_ng26(qbuf, qoff, sch, edit=True, cap=1 << 21)
        assert same_hits(hits, ohits) and st.lf_steps == nodes, (layout, k, lf)
        ex = fm.search_scheme.expand(sch, length)
        hits, st = fm.search_ng21.search(gx, (qbuf, qoff), ex, want_stats=True, capacity=1 << 21)
        ohits, _, nodes = ox.search_ng21(qbuf, qoff, ex, cap=1 << 21)
        assert same_hits(hits, ohits) and st.lf_steps == nodes, (layout, k, lf)
        total += len(ohits)
    assert total > 0
    sch = fm.search_scheme.h2(k + 2, 0, k)
    for n in (1, 3):
        assert same_hits(fm.search_ng26.search(gx, (qbuf, qoff), sch, n=n, edit=True), ox.search_ng26(qbuf, qoff, sch, max_hits=n, edit=True)[0])
        ex = fm.search_scheme.expand(sch, length)
        assert same_hits(fm.search_ng21.search_n(gx, (qbuf, qoff), ex, n), ox.search_ng21(qbuf, qoff, ex, max_hits=n)[0])
    ragged = mutated_queries(seqs, 200, 12, 40, k, seed=23, sigma=sigma)
    rb, ro = fm.flatten(ragged)
    assert same_hits(fm.search_ng26.search(gx, (rb, ro), sch, edit=True, capacity=1 << 21), ox.search_ng26(rb, ro, sch, edit=True, cap=1 << 21)[0])


@pytest.mark.parametrize("bidir", [False, True])
@pytest.mark.parametrize("layout,sigma", [("IB16", 5), ("WAVELET", 28), ("IB16", 256)])
def test_wide_gpu_builder(layout, sigma, bidir):
    seqs = repeat_text(7, 2400) if sigma == 5 else [make_text(1800, sigma, seed=2), make_text(333, sigma, seed=3), np.zeros(0, dtype=np.uint8)]
    ox = fo.OraIndex.build(layout, sigma, seqs, 4, bidir)
    cls = fm.BiFMIndex if bidir else fm.FMIndex
    with force_wide():
        gx = cls.from_sequences(seqs, sigma, layout, 4, keep_host=True)
    assert gx.row_bits == 64
    n = ox.n
    assert gx.n == n
    assert np.array_equal(gx.built_array(0), np.array([ox.bwt_string().symbol(i) for i in range(n)], dtype=np.uint8))
    if bidir:
        assert np.array_equal(gx.built_array(1), np.array([ox.bwt_string(rev=True).symbol(i) for i in range(n)], dtype=np.uint8))
    sp = ox.sparse()
    assert np.array_equal(gx.built_array(3, np.uint64), sp["l0"]) and np.array_equal(gx.built_array(4, np.uint16), sp["l1"])
    assert np.array_equal(gx.built_array(5, np.uint64), sp["bits"])
    assert np.array_equal(gx.built_array(6, np.uint64), sp["fields"][0]["data"]) and np.array_equal(gx.built_array(7, np.uint64), sp["fields"][1]["data"])
    qbuf, qoff = fm.flatten(mutated_queries([s for s in seqs if len(s) > 50], 400, 8, 30, 1, seed=5, sigma=sigma))
    lb, ln = fm.search_no_errors.search(gx, (qbuf, qoff))
    olb, oln = ox.search_exact(qbuf, qoff)
    assert np.array_equal(lb, olb) and np.array_equal(ln, oln)
    rows = np.arange(0, min(gx.n, 1500), dtype=np.uint64)
    seq, pos, steps = gx.locate(rows)
    for r in range(0, rows.size, 11):
        assert (seq[r], pos[r], steps[r]) == ox.locate(int(r))


@pytest.mark.parametrize("shape", ["repeats", "many_sequences", "poly_a"])
def test_wide_exact_search_in_pair_steps(shape, monkeypatch):
    """Format P / k_exact_p on 64-bit rows (line counts relative to super-blocks of 2^30 rows + a super table, listed rows as 64-bit numbers):
    intervals, miss rows and step counts equal the oracle's and the one-symbol kernel's"""
    rng = np.random.default_rng(78)
    if shape == "repeats":
        seqs = repeat_text(6, n=6000)
    elif shape == "many_sequences":
        seqs = [rng.integers(1, 5, size=int(rng.integers(1, 90)), dtype=np.uint8) for _ in range(250)]
    else:
        seqs = [np.where(rng.random(int(rng.integers(2, 700))) < 0.85, 1, rng.integers(1, 5, size=1)[0]).astype(np.uint8) for _ in range(60)]
    ox = fo.OraIndex.build("IB16", 5, seqs, 4, False)
    gx = wide_index(ox)
    fm.options["pair_table"] = "0"
    gx_single = wide_index(ox)
    del fm.options["pair_table"]
    assert gx.device_bytes > gx_single.device_bytes
    queries = mutated_queries([q for q in seqs if len(q) > 2], 1500, 1, 140, 2, seed=6)
    queries += [[], [1], [1, 1], [1, 1, 1], [4, 4, 4, 4], [0], [1, 0], [0, 1], [1, 0, 1, 1], [2, 1, 0], [1] * 64, [1] * 65, [1] * 129, [2, 1], [1, 2]]
    for s_ in seqs[:40]:
        queries += [s_, s_[-3:], s_[:3], np.concatenate([s_[-2:], [0]]), np.concatenate([[0], s_[:2]])]
    qbuf, qoff = fm.flatten(queries)
    olb, oln, ost = ox.search_exact(qbuf, qoff, want_steps=True)
    for flags in ("0", str(1 << 22)):
        fm.options["kernel_select"] = flags
        for g in (gx, gx_single):
            lb, ln, st = fm.search_no_errors.search(g, (qbuf, qoff), want_stats=True)
            assert np.array_equal(ln, oln) and np.array_equal(lb, olb), (shape, flags)
            assert st.lf_steps == int(ost.sum())

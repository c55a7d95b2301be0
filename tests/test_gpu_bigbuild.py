"""Construction bucket by bucket (csrc/fmgpu_bucketsort.hip) — the two suffix sorters that never hold the suffix array (suffix_sorter = 2: prefix doubling on the ties with the
inverse suffix array as rank array; 3: no array of n entries at all, ties broken by further symbols of the text), for texts beyond the ~6e9 rows the all-at-once sorter's
buffers fit (UniRef50-sized protein databases: BASELINE.json configs[4]).  FMIndex(Sequences, samplingRate) / BiFMIndex(...) (fmindex/FMIndex.h:58-104, BiFMIndex.h:107-167,
utils.h:97-163) as restated by the oracle, and the library's own all-at-once sorter (pinned against the oracle in test_gpu_parity.py), are what it is compared with:
every built array byte for byte — BWT, bwtRev, C, the SparseArray's presence bits, both counter levels, both bit-packed fields and their parameters."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import fmindex_collection_amd as fm  # noqa: E402
from fmindex_collection_amd import capi  # noqa: E402
from tests.util import make_text, sample_reads  # noqa: E402

pytestmark = pytest.mark.gpu


def _built(gx, bidir):
    parts = [gx.built_array(0).copy(), gx.built_array(2, np.uint64).copy()] + [gx.built_array(k, np.uint64 if k != 4 else np.uint16).copy() for k in range(3, 9)]
    if bidir:
        parts.append(gx.built_array(1).copy())
    return parts


def _same(a, b):
    return len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))


def _collection(seed):
    """1 to 30 sequences over alphabets 2 .. 28: random ones, runs of one symbol, tandem repeats, exact and truncated copies of earlier sequences (ties that only a delimiter resolves)"""
    rng = np.random.default_rng(9000 + seed)
    sigma = int(rng.choice([2, 3, 5, 5, 6, 21, 28]))
    seqs = []
    for _ in range(int(rng.integers(1, 31))):
        kind = int(rng.integers(0, 6))
        m = int(rng.integers(0, 600))
        if kind == 0 or sigma == 2: q = np.ones(m, dtype=np.uint8)
        elif kind == 1: q = np.tile(rng.integers(1, sigma, size=int(rng.integers(1, 9)), dtype=np.uint8), m // 2 + 1)[:m]
        elif kind == 2 and seqs: q = seqs[int(rng.integers(0, len(seqs)))].copy()
        elif kind == 3 and seqs: base = seqs[int(rng.integers(0, len(seqs)))]; q = base[len(base) // 3:].copy()
        else: q = rng.integers(1, sigma, size=m, dtype=np.uint8)
        seqs.append(q.astype(np.uint8))
    return sigma, seqs, int(rng.choice([1, 3, 16, 64])), bool(rng.integers(0, 2))


@pytest.mark.parametrize("seed", list(range(12)))
@pytest.mark.parametrize("wide", [False, True])
def test_bucketed_construction_equals_the_all_at_once_sorter(seed, wide):
    """the same collection built with the all-at-once sorter and bucket by bucket, by both bucketed sorters — buckets of 1, 61 and 4000 rows and one bucket for everything
    (the bucket size also caps the groups a doubling round sorts together: 61 makes every round run through several segments) — in both row widths"""
    sigma, seqs, rate, bidir = _collection(seed)
    cls = fm.BiFMIndex if bidir else fm.FMIndex
    layout = "WAVELET" if sigma > 5 else "IB16"
    with fm.options(force_wide=1 if wide else 0):
        with fm.options(suffix_sorter=1):
            want = _built(cls.from_sequences(seqs, sigma, layout, rate, keep_host=True), bidir)
        for sorter in (2, 3):
            for rows in ((1,) if seed % 4 == 0 else ()) + (61, 4000, 1 << 40):
                with fm.options(bucket_rows=rows, suffix_sorter=sorter):
                    gx = cls.from_sequences(seqs, sigma, layout, rate, keep_host=True)
                assert gx.row_bits == (64 if wide else 32)
                assert _same(_built(gx, bidir), want), (seed, wide, sorter, rows, sigma, len(seqs), rate, bidir)


@pytest.mark.parametrize("sorter", [2, 3])
def test_bucketed_construction_against_the_oracle(sorter):
    """... and directly against the CPU restatement of the reference's construction (BWT, C, SparseArray arrays, locate), 200 kbp in 23 buckets"""
    import fmoracle as fo
    seqs = [make_text(120_000, 5, 31), make_text(7, 5, 32), make_text(80_000, 5, 33)]
    ox = fo.OraIndex.build("IB16", 5, seqs, 16, True)
    with fm.options(bucket_rows=9000, suffix_sorter=sorter):
        gx = fm.BiFMIndex.from_sequences(seqs, 5, "IB16", 16, keep_host=True)
    n = ox.n
    assert gx.n == n
    assert np.array_equal(gx.built_array(2, np.uint64), ox.C)
    sp = ox.sparse()
    assert np.array_equal(gx.built_array(3, np.uint64), sp["l0"]) and np.array_equal(gx.built_array(4, np.uint16), sp["l1"]) and np.array_equal(gx.built_array(5, np.uint64), sp["bits"])
    assert np.array_equal(gx.built_array(6, np.uint64), sp["fields"][0]["data"]) and np.array_equal(gx.built_array(7, np.uint64), sp["fields"][1]["data"])
    bw = gx.built_array(0); bwr = gx.built_array(1)
    st, sr = ox.bwt_string(), ox.bwt_string(rev=True)
    for i in range(0, n, 37):
        assert bw[i] == st.symbol(i) and bwr[i] == sr.symbol(i)
    reads = sample_reads(np.concatenate(seqs[::2]), 3000, 40, seed=5, mutate=1)
    qbuf, qoff = fm.flatten(reads)
    lb, ln = fm.search_no_errors.search(gx, (qbuf, qoff))
    olb, oln = ox.search_exact(qbuf, qoff)
    assert np.array_equal(lb, olb) and np.array_equal(ln, oln)
    rows = np.arange(0, n, 211, dtype=np.uint64)
    seq, pos, steps = gx.locate(rows)
    assert [(int(a), int(b), int(c)) for a, b, c in zip(seq, pos, steps)] == [ox.locate(int(r)) for r in rows]


@pytest.mark.parametrize("sorter", [2, 3])
def test_bucketed_construction_of_a_protein_text_with_shared_domains(sorter):
    """sigma = 28, 3 M residues in 6 000 sequences, a fifth of them carrying one of 40 'domains' of 30 .. 300 residues verbatim (ties of hundreds of symbols between distant
    sequences: several tie rounds), 40 buckets: every array equals the all-at-once sorter's; exact search of 40-residue reads finds every read where it was cut"""
    rng = np.random.default_rng(77)
    domains = [rng.integers(1, 28, size=int(rng.integers(30, 301)), dtype=np.uint8) for _ in range(40)]
    seqs = []
    for i in range(6000):
        q = rng.integers(1, 28, size=500, dtype=np.uint8)
        if i % 5 == 0:
            d = domains[int(rng.integers(0, 40))]; at = int(rng.integers(0, 500 - len(d) + 1)); q[at:at + len(d)] = d
        seqs.append(q)
    with fm.options(suffix_sorter=1):
        want = _built(fm.FMIndex.from_sequences(seqs, 28, "WAVELET", 16, keep_host=True), False)
    with fm.options(bucket_rows=80_000, suffix_sorter=sorter):
        gx = fm.FMIndex.from_sequences(seqs, 28, "WAVELET", 16, keep_host=True)
    assert _same(_built(gx, False), want)
    pick = rng.integers(0, 6000, size=2000); off = rng.integers(0, 460, size=2000)
    reads = [seqs[int(s)][int(o):int(o) + 40] for s, o in zip(pick, off)]
    lb, ln = fm.search_no_errors.search(gx, fm.flatten(reads))
    assert (ln >= 1).all()
    seq, pos, steps = gx.locate(lb)                                 # (the first row of each interval — SOME occurrence of the read: the sampled entry + the LF steps that led to it, FMIndex.h:113-124)
    for s, p, k, r in zip(seq, pos, steps, reads):
        assert np.array_equal(seqs[int(s)][int(p) + int(k):int(p) + int(k) + 40], r)


def test_a_megabase_run():
    """a run of 3 M equal symbols.  The sorter without a rank array breaks ties by reading further symbols of the text — ~2e5 rounds over the run's rows: its work is bounded and
    the text is refused with an error that says so.  The sorter that keeps the inverse suffix array doubles (18 rounds) and builds what the all-at-once sorter builds"""
    seqs = [np.concatenate([make_text(2000, 5, 3), np.ones(3_000_000, dtype=np.uint8), make_text(2000, 5, 4)])]
    with fm.options(bucket_rows=1 << 40, suffix_sorter=3):
        with pytest.raises(fm.FmgpuError) as e:
            fm.FMIndex.from_sequences(seqs, 5, "IB16", 16)
    assert e.value.code == capi.FMGPU_ERR_UNSUPPORTED and "doubling" in str(e.value)
    with fm.options(suffix_sorter=1):
        want = _built(fm.BiFMIndex.from_sequences(seqs, 5, "IB16", 16, keep_host=True), True)
    with fm.options(bucket_rows=700_000, suffix_sorter=2):
        gx = fm.BiFMIndex.from_sequences(seqs, 5, "IB16", 16, keep_host=True)
    assert gx.n == 3_004_001 and _same(_built(gx, True), want)


def test_repeat_structured_text_through_the_rank_array_sorter():
    """20 Mbp of the repeat-structured genome stand-in (interspersed repeat families, satellite arrays, runs of one symbol: ties thousands of symbols deep over a large share of the
    rows), BiFMIndex, 7 buckets: every built array equals the all-at-once sorter's"""
    from fmindex_collection_amd import datasets
    lengths = [6_000_000, 9_000_000, 5_000_000]
    text, _ = datasets.genome_like_text(lengths, seed=11, device="cuda:0")
    flat = text.cpu().numpy(); off = np.concatenate([[0], np.cumsum(lengths)])
    seqs = [flat[off[i]:off[i + 1]] for i in range(len(lengths))]
    with fm.options(suffix_sorter=1):
        want = _built(fm.BiFMIndex.from_sequences(seqs, 5, "IB16", 16, keep_host=True), True)
    with fm.options(bucket_rows=3_000_000, suffix_sorter=2):
        gx = fm.BiFMIndex.from_sequences(seqs, 5, "IB16", 16, keep_host=True)
    assert _same(_built(gx, True), want)


def test_beyond_2_32_rows_both_ways():
    """FMIndex<28, Wavelet> over 4.5e9 residues (9 M sequences; 64-bit rows at their real size), built by the all-at-once sorter and bucket by bucket with the inverse suffix array as
    rank array: the same device bytes, the same intervals for 1 M reads cut from the text (every tenth with a substitution: not found), and — the check that found the launch of 2^32
    threads cut short — every located row spells its read within 15 LF steps of a sampled entry"""
    torch = pytest.importorskip("torch")
    import ctypes as C
    dev = torch.device("cuda", 0)
    nseq, slen, sigma, L, nq = 9_000_000, 500, 28, 40, 1_000_000
    g = torch.Generator(device=dev); g.manual_seed(42)
    text = torch.empty(nseq * slen, dtype=torch.uint8, device=dev)
    for lo in range(0, text.numel(), 1 << 28):
        hi = min(text.numel(), lo + (1 << 28))
        text[lo:hi] = torch.randint(1, sigma, (hi - lo,), generator=g, device=dev, dtype=torch.uint8)
    seq_off = torch.arange(nseq + 1, device=dev, dtype=torch.int64) * slen
    starts = torch.randint(0, nseq, (nq,), generator=g, device=dev, dtype=torch.int64) * slen + torch.randint(0, slen - L + 1, (nq,), generator=g, device=dev, dtype=torch.int64)
    reads = text[starts[:, None] + torch.arange(L, device=dev)[None, :]].contiguous()
    bad = torch.arange(0, nq, 10, device=dev)
    reads[bad, 7] = reads[bad, 7] % (sigma - 1) + 1
    qoff = torch.arange(nq + 1, device=dev, dtype=torch.int64) * L

    class V:
        def __init__(self, t): self.t, self.ptr, self.nbytes = t, t.data_ptr(), t.numel() * t.element_size()
    seen = []
    for sorter in (1, 2):
        with fm.options(suffix_sorter=sorter, lf_table=0):
            gx = fm.FMIndex.from_sequences((V(text), V(seq_off)), sigma, "WAVELET", 16)
        assert gx.row_bits == 64 and gx.n == nseq * (slen + 1)
        out = torch.empty(2 * nq, dtype=torch.int64, device=dev)
        capi.check(capi.lib().fmgpu_search_exact(gx._h, C.c_void_p(reads.data_ptr()), C.c_void_p(qoff.data_ptr()), nq, C.c_void_p(out[:nq].data_ptr()), C.c_void_p(out[nq:].data_ptr()), None, None))
        found = out[nq:] > 0
        assert int(found.sum().item()) == nq - bad.numel() and not bool(found[bad].any().item())
        rows = out[:nq][found][:100_000].contiguous(); k = rows.numel()
        loc = torch.empty(3 * k, dtype=torch.int64, device=dev)
        capi.check(capi.lib().fmgpu_locate(gx._h, C.c_void_p(rows.data_ptr()), k, C.c_void_p(loc[:k].data_ptr()), C.c_void_p(loc[k:2 * k].data_ptr()), C.c_void_p(loc[2 * k:].data_ptr()), None, None))
        at = loc[:k] * slen + loc[k:2 * k] + loc[2 * k:]
        assert bool(torch.equal(text[at[:, None] + torch.arange(L, device=dev)[None, :]], reads[found][:100_000]))
        assert int(loc[2 * k:].max().item()) < 16
        seen.append((gx.device_bytes, out.clone(), loc.clone()))
        gx.close(); del gx, out, loc
        torch.cuda.empty_cache()
    assert seen[0][0] == seen[1][0] and torch.equal(seen[0][1], seen[1][1]) and torch.equal(seen[0][2], seen[1][2])

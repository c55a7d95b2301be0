"""fmindex-collection_amd/datasets.py on the CPU: the FASTA reader against a byte-by-byte restatement of the reference example's loop
(src/example/utils.h:26-104 with --convertUnknownChar, sigma = 5, no reverse complements), and the genome-like generator's determinism and make-up."""
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fmindex_collection_amd import datasets  # noqa: E402


def reference_loop(b):
    """utils.h:38-103, state machine as written there: Name / Sequence modes over the file's bytes"""
    seqs, q, i, n, name_mode = [], [], 0, len(b), True
    assert b[0] == ord(">")                                   # :39-41
    while i != n:
        if name_mode:
            assert b[i] == ord(">")                           # :45-47
            i += 1
            if i < n and b[i] == ord(" "):
                i += 1
            while i != n and b[i] != ord("\n"):
                i += 1
            i += 1
            name_mode = False
            if i > n:
                break
        elif b[i] == ord(">") or i + 1 == n:                  # :64: a record ends at '>' or at the last byte of the file
            seqs.append(q); q = []; name_mode = True
            if i + 1 == n:
                i += 1
        else:
            c = chr(b[i])
            if c == "$": q.append(0)
            elif c in "Aa": q.append(1)
            elif c in "Cc": q.append(2)
            elif c in "Gg": q.append(3)
            elif c in "Tt": q.append(4)
            elif c != "\n": q.append(1)                       # :86-98 convertUnknownChar at sigma = 5
            i += 1
    return seqs


def test_fasta_reader_follows_the_reference_loop(tmp_path):
    random.seed(1)
    checked = 0
    for t in range(400):
        parts = []
        for r in range(random.randint(1, 5)):
            parts.append(">" + "".join(random.choice("ab >x") for _ in range(random.randint(0, 6))) + "\n")
            for l in range(random.randint(0, 4)):
                parts.append("".join(random.choice("ACGTacgtNn$x") for _ in range(random.randint(0, 12))) + random.choice(["\n", "\n", ""]))
        txt = "".join(parts)
        if random.random() < 0.5 and not txt.endswith("\n"):
            txt += "\n"
        b = np.frombuffer(txt.encode(), dtype=np.uint8)
        try:
            want = reference_loop(b)
        except AssertionError:
            continue                                          # a file the reference refuses ("expected '>'")
        path = tmp_path / "x.fa"
        path.write_bytes(b.tobytes())
        sym, off = datasets.load_fasta(str(path))
        got = [sym[off[k]: off[k + 1]].tolist() for k in range(len(off) - 1)]
        assert got == want, txt
        checked += 1
    assert checked > 200
    (tmp_path / "bad.fa").write_bytes(b"ACGT\n")
    with pytest.raises(ValueError):
        datasets.load_fasta(str(tmp_path / "bad.fa"))


def test_genome_like_text_is_deterministic_and_repeat_structured():
    pytest.importorskip("torch")
    lengths = [int(x * 0.003) for x in (248956422, 242193529, 198295559, 190214555, 16569)]
    a, st = datasets.genome_like_text(lengths, seed=42, device="cpu")
    b, _ = datasets.genome_like_text(lengths, seed=42, device="cpu")
    c, _ = datasets.genome_like_text(lengths, seed=43, device="cpu")
    assert a.numel() == sum(lengths) and bool((a == b).all()) and not bool((a == c).all())
    t = a.numpy()
    assert t.min() >= 1 and t.max() <= 4
    assert 0.35 < st["repeat_fraction_written"] < 0.5 and 0.005 < st["satellite_fraction_written"] < 0.03 and 0.04 < st["run_fraction_written"] < 0.06
    share = np.bincount(t, minlength=5) / t.size                 # the runs of one symbol are written as rank 1
    assert share[1] > 0.27 and abs(share[2] - share[3]) < 0.01
    # repeats: 24-mers that occur more than once are common here and essentially absent from a uniform text of this size
    k = 24
    w = np.lib.stride_tricks.sliding_window_view(t[: 400_000], k)
    keys = (w.astype(np.uint64) * (np.uint64(5) ** np.arange(k, dtype=np.uint64))).sum(axis=1)
    dup = 1.0 - np.unique(keys).size / keys.size
    assert dup > 0.05

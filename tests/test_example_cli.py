"""The reference's `example` harness on the GPU path (fmindex-collection_amd/example/main.cpp; reference: src/example/main.cpp, argp.h,
utils.h): flags, FASTA reading incl. the reference parser's quirks, reverse complements, and the `--save_output` file ("queryId seqId pos"
per located row in callback order) against the oracle driven by an independent restatement of the same flow in Python."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
PKG = os.path.join(ROOT, "fmindex-collection_amd")
EXE = os.path.join(PKG, "example", "example")

import fmoracle as fo  # noqa: E402
from fmindex_collection_amd import search_scheme as ss_host  # noqa: E402   (expandByWNC, checked against the real reference in test_host_and_abi.py)


def _build():
    subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "-j4", "-s"], check=True)


def test_example_builds_and_parses_flags():
    _build()
    r = subprocess.run([EXE, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--maxhitperquery" in r.stdout and "--save_output" in r.stdout
    r = subprocess.run([EXE, "--bogus"], capture_output=True, text=True)
    assert r.returncode == 1 and "unknown commandline --bogus" in r.stderr            # argp.h:103
    r = subprocess.run([EXE, "--mode", "some"], capture_output=True, text=True)
    assert r.returncode == 1 and "invalid mode" in r.stderr                            # argp.h:96


# ---- the flow of src/example (utils.h:27-105, main.cpp:74-270), restated for the oracle
RANK = {"$": 0, "A": 1, "a": 1, "C": 2, "c": 2, "G": 3, "g": 3, "T": 4, "t": 4}


def load_fasta(data, reverse, convert=False):
    out, i, n = [], 0, len(data)
    assert data[0] == ">"
    query, in_name = [], True
    while i != n:
        if in_name:
            i += 1
            while i != n and data[i] != "\n":
                i += 1
            i += 1
            in_name = False
        elif data[i] == ">" or i + 1 == n:                   # the last byte of the file is never a symbol
            out.append(list(query))
            if reverse:
                out.append([{1: 4, 2: 3, 3: 2, 4: 1}.get(c, c) for c in reversed(query)])
            query, in_name = [], True
            if i + 1 == n:
                i += 1
        else:
            ch = data[i]
            if ch in RANK:
                query.append(RANK[ch])
            elif ch != "\n":
                assert convert
                query.append(1)
            i += 1
    return out


def generate(name, min_k, max_k):
    return {"backtracking": lambda: fo.scheme_backtracking(1, min_k, max_k), "pigeon": lambda: fo.scheme_pigeon_trivial(min_k, max_k),
            "pigeon_opt": lambda: fo.scheme_pigeon_opt(min_k, max_k), "h2-k1": lambda: fo.scheme_h2(max_k + 1, min_k, max_k),
            "h2-k2": lambda: fo.scheme_h2(max_k + 2, min_k, max_k), "h2-k3": lambda: fo.scheme_h2(max_k + 3, min_k, max_k)}[name]()


def stretch(scheme, m, dyn):
    """main.cpp:114-123: the scheme at read length — uniformly, or (`<name>_dyn`) by expandByWNC<Edit = true>(oss, len, 4, 3e9)"""
    if not dyn:
        return fo.scheme_expand(scheme, m)
    e = ss_host.expandByWNC(scheme, m, 4, 3_000_000_000, True)
    return tuple(np.ascontiguousarray(x, dtype=np.uint64) for x in e)


def expected_output(ref_fa, query_fa, algo, gen, k, mode="all", maxhits=0, reverse=True, max_queries=0, read_length=0, convert=False, dyn=False):
    ref = load_fasta(ref_fa, False, convert)
    x = fo.OraIndex.build("IB16", 5, [np.array(s, dtype=np.uint8) for s in ref], 16, True)
    queries = load_fasta(query_fa, reverse, convert)
    if max_queries:
        queries = queries[:max_queries]
    if read_length:
        queries = [q[:read_length] for q in queries]
    qbuf, qoff = fo.flatten_queries([np.array(q, dtype=np.uint8) for q in queries])
    n = maxhits if maxhits else fo.UINT64_MAX
    m = len(queries[0])
    if algo == "ng21":
        if mode == "all":
            hits, _, _ = x.search_ng21(qbuf, qoff, stretch(generate(gen, 0, k), m, dyn), max_hits=n)
        else:
            hits, _ = x.search_ng21_best(qbuf, qoff, [stretch(generate(gen, j, j), m, dyn) for j in range(k + 1)], max_hits=n)
    elif algo == "ng26":
        hits, _, _ = x.search_ng26(qbuf, qoff, generate(gen, 0, k), max_hits=n, edit=True)
    else:
        lb, ln = x.search_exact(qbuf, qoff)
        hits = [{"qidx": q, "lb": lb[q], "len": ln[q]} for q in range(len(queries)) if ln[q]]
    lines = []
    for h in hits:
        for r in range(int(h["lb"]), int(h["lb"]) + int(h["len"])):
            s, p, o = x.locate(r)
            lines.append("%d %d %d" % (int(h["qidx"]), s, p + o))
    return lines


def _fasta(rng, tmp_path):
    a = "".join("ACGT"[i] for i in rng.integers(0, 4, size=2400))
    b = a[500:1100] + "".join("ACGT"[i] for i in rng.integers(0, 4, size=900))
    c = "".join("acgt"[i] for i in rng.integers(0, 4, size=300))
    ref = ">chrA first\n" + "\n".join(a[i: i + 60] for i in range(0, len(a), 60)) + "\n>chrB\n" + "\n".join(b[i: i + 70] for i in range(0, len(b), 70)) + \
          "\n> chrC lower case\n" + c + "\n"
    reads = []
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    for i in range(120):
        src = a if i % 3 else b
        p = int(rng.integers(0, len(src) - 40))
        r = list(src[p: p + 36])
        for _ in range(int(rng.integers(0, 3))):
            op = int(rng.integers(0, 3)); at = int(rng.integers(1, len(r) - 1))
            if op == 0: r[at] = "ACGT"[int(rng.integers(0, 4))]
            elif op == 1: del r[at]
            else: r.insert(at, "ACGT"[int(rng.integers(0, 4))])
        r = "".join(r)[:32]
        if i % 4 == 0:
            r = "".join(comp[ch] for ch in reversed(r))       # found through its reverse complement only
        reads.append(r)
    qry = "".join(">read%d\n%s\n" % (i, r) for i, r in enumerate(reads))
    rp, qp = tmp_path / "ref.fasta", tmp_path / "reads.fasta"
    rp.write_text(ref); qp.write_text(qry)
    return ref, qry, str(rp), str(qp)


@pytest.mark.gpu
def test_example_output_matches_the_reference_flow(tmp_path):
    _build()
    rng = np.random.default_rng(11)
    ref, qry, rp, qp = _fasta(rng, tmp_path)
    out = str(tmp_path / "out.txt")
    cases = [
        (["--algo", "ng21", "--gen", "h2-k2", "--min_k", "2", "--max_k", "2"], dict(algo="ng21", gen="h2-k2", k=2)),
        (["--algo", "ng21", "--gen", "pigeon_opt", "--min_k", "0", "--max_k", "1", "--mode", "besthits"], dict(algo="ng21", gen="pigeon_opt", k=1, mode="besthits")),
        (["--algo", "ng21", "--gen", "h2-k1", "--min_k", "1", "--max_k", "1", "--maxhitperquery", "2", "--no-reverse"],
         dict(algo="ng21", gen="h2-k1", k=1, maxhits=2, reverse=False)),
        (["--algo", "ng21", "--gen", "backtracking", "--min_k", "1", "--max_k", "1", "--mode", "besthits", "--maxhitperquery", "1", "--queries", "50", "--read_length", "24"],
         dict(algo="ng21", gen="backtracking", k=1, mode="besthits", maxhits=1, max_queries=50, read_length=24)),
        (["--algo", "ng21", "--gen", "h2-k2_dyn", "--min_k", "2", "--max_k", "2"], dict(algo="ng21", gen="h2-k2", k=2, dyn=True)),
        (["--algo", "ng21", "--gen", "pigeon_opt_dyn", "--min_k", "1", "--max_k", "1", "--mode", "besthits"], dict(algo="ng21", gen="pigeon_opt", k=1, mode="besthits", dyn=True)),
        (["--algo", "noerror", "--min_k", "0", "--max_k", "0"], dict(algo="noerror", gen="h2-k2", k=0)),
        (["--algo", "ng26", "--gen", "h2-k2", "--min_k", "2", "--max_k", "2"], dict(algo="ng26", gen="h2-k2", k=2)),
    ]
    for flags, kw in cases:
        if os.path.exists(out):
            os.remove(out)
        r = subprocess.run([EXE, "--index", rp, "--query", qp, "--save_output", out] + flags, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        want = expected_output(ref, qry, **kw)
        got = open(out).read().split("\n")[:-1]
        assert len(want) > 0 and got == want, (flags, len(got), len(want))
        nq = 2 * 120 if kw.get("reverse", True) else 120
        assert ("loaded %d queries (incl reverse complements)" % nq) in r.stdout
        stats = [ln for ln in r.stdout.split("\n") if ln.startswith("str ")][-1]
        assert ("%10d/%10d" % (len(want), len(want))) in stats                                   # resultCt / results.size()


@pytest.mark.gpu
def test_example_fasta_quirks_and_errors(tmp_path):
    """no trailing newline: the reference's reader drops the last base (utils.h:63, :78-80); unknown letters: error unless --convertUnknownChar
    (-> rank 1, utils.h:91-99); unknown generator / algorithm: error"""
    _build()
    rng = np.random.default_rng(12)
    ref, qry, rp, qp = _fasta(rng, tmp_path)
    out = str(tmp_path / "out.txt")
    q2 = tmp_path / "q2.fasta"
    body = qry.rstrip("\n")                                   # the last read loses its last base
    q2.write_text(body)
    r = subprocess.run([EXE, "--index", rp, "--query", str(q2), "--save_output", out, "--algo", "ng21", "--min_k", "0", "--max_k", "0", "--no-reverse", "--queries", "119"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(out).read().split("\n")[:-1] == expected_output(ref, body, algo="ng21", gen="h2-k2", k=0, reverse=False, max_queries=119)
    q3 = tmp_path / "q3.fasta"
    q3.write_text(qry.replace("A", "N", 3))
    base = [EXE, "--index", rp, "--query", str(q3), "--save_output", out, "--algo", "ng21", "--min_k", "1", "--max_k", "1"]
    r = subprocess.run(base, capture_output=True, text=True)
    assert r.returncode == 1 and "unknown alphabet" in r.stderr
    r = subprocess.run(base + ["--convertUnknownChar"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(out).read().split("\n")[:-1] == expected_output(ref, qry.replace("A", "N", 3), algo="ng21", gen="h2-k2", k=1, convert=True)
    for extra, msg in ((["--gen", "kianfar"], "unknown search scheme"), (["--gen", "kianfar_dyn"], "unknown search scheme"), (["--algo", "ng12"], "not part of this build")):
        r = subprocess.run([EXE, "--index", rp, "--query", qp, "--min_k", "1", "--max_k", "1"] + (["--algo", "ng21"] if extra[0] != "--algo" else []) + extra,
                           capture_output=True, text=True)
        assert r.returncode == 1 and msg in r.stderr, (extra, r.stderr)

#!/usr/bin/env python3
"""one-off: the example harness at scale — a synthetic 200 Mbp FASTA (8 records), 1 M reads x 101 bp with 0/1/2 substitutions, reverse
complements on (2 M queries); prints the harness' own statistics lines.  usage (through gpurun): python tools/example_probe.py [Mbp] [reads]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "fmindex-collection_amd", "example", "example")
mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
rng = np.random.default_rng(1)
tmp = os.environ.get("TMPDIR", "/tmp")
rp, qp = os.path.join(tmp, "probe_ref.fasta"), os.path.join(tmp, "probe_reads.fasta")
letters = np.frombuffer(b"ACGT", dtype=np.uint8)
recs = []
with open(rp, "wb") as f:
    for i in range(8):
        s = letters[rng.integers(0, 4, size=mbp * 1_000_000 // 8)]
        recs.append(s)
        f.write(b">chr%d\n" % i)
        f.write(s.tobytes())
        f.write(b"\n")
L = 101
with open(qp, "wb") as f:
    src = recs[0]
    starts = rng.integers(0, len(src) - L, size=nreads)
    reads = src[starts[:, None] + np.arange(L)[None, :]].copy()
    for k in range(2):
        sel = np.nonzero(np.arange(nreads) % 3 > k)[0]
        pos = rng.integers(0, L, size=sel.size)
        reads[sel, pos] = letters[rng.integers(0, 4, size=sel.size)]
    lines = np.empty((nreads, L + 1), dtype=np.uint8)
    lines[:, :L] = reads
    lines[:, L] = 10
    for i in range(nreads):
        f.write(b">r%d\n" % i)
        f.write(lines[i].tobytes())
for flags in (["--algo", "ng26", "--gen", "h2-k2", "--min_k", "0", "--max_k", "2"], ["--algo", "ng21", "--gen", "h2-k2", "--min_k", "0", "--max_k", "2"],
              ["--algo", "noerror", "--min_k", "0", "--max_k", "0"]):
    t = time.time()
    r = subprocess.run([EXE, "--index", rp, "--query", qp] + flags, capture_output=True, text=True)
    print(" ".join(flags), "-> rc", r.returncode, "wall %.1f s" % (time.time() - t))
    print("\n".join(ln for ln in (r.stdout + r.stderr).split("\n") if ln.startswith(("str ", "loaded", "error"))), flush=True)
os.remove(rp); os.remove(qp)

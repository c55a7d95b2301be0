"""dev probe: search_ng21 (k_ng21) against search_ng26<edit> (k_scheme_fast_edit / general) on the same plain index and reads"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fmindex_collection_amd as fm
from fmindex_collection_amd import datasets
n, nq, L, k = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000, 200_000, 101, 2
rng = np.random.default_rng(1)
text = datasets.genome_like(n, seed=3) if hasattr(datasets, "genome_like") else rng.integers(1, 5, size=n, dtype=np.uint8)
os.environ["FMGPU_LF_TABLE"] = "0"
t0 = time.time()
ix = fm.BiFMIndex.from_sequences([np.asarray(text, dtype=np.uint8)], 5, "IB16", 16)
print("built %.1f s" % (time.time() - t0), flush=True)
pos = rng.integers(0, n - L, size=nq)
reads = np.stack([np.asarray(text[p: p + L]) for p in pos]).astype(np.uint8)
for r in reads[::3]:
    r[rng.integers(0, L)] = rng.integers(1, 5)
qbuf, qoff = reads.reshape(-1).copy(), (np.arange(nq + 1, dtype=np.uint64) * L)
sch = fm.search_scheme.h2(k + 2, 0, k)
ex = fm.search_scheme.expand(sch, L)
for name, fn in (("ng26 edit (fast)", lambda: fm.search_ng26.search(ix, (qbuf, qoff), sch, edit=True, want_stats=True, capacity=1 << 26)),
                 ("ng21", lambda: fm.search_ng21.search(ix, (qbuf, qoff), ex, want_stats=True, capacity=1 << 26))):
    for _ in range(2):
        t0 = time.time(); hits, st = fn(); dt = time.time() - t0
    print("%-18s kernel %.1f ms, wall %.1f ms, %d hits, %d nodes" % (name, st.kernel_ms, dt * 1e3, len(hits), st.lf_steps), flush=True)
os.environ["FMGPU_DEV_FLAGS"] = "2"
hits, st = fm.search_ng26.search(ix, (qbuf, qoff), sch, edit=True, want_stats=True, capacity=1 << 26)
print("ng26 edit (general) kernel %.1f ms, %d hits, %d nodes" % (st.kernel_ms, len(hits), st.lf_steps))

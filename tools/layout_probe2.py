"""dev probe: the same through fmgpu_index_create from the arrays a reference index holds (Format R for EPR / EPRV2, Format A for the rest)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import fmoracle as fo
import fmindex_collection_amd as fm
from tests.util import oracle_arrays
n, nq, L = 10_000_000, 500_000, 101
rng = np.random.default_rng(1)
text = rng.integers(1, 5, size=n, dtype=np.uint8)
pos = rng.integers(0, n - L, size=nq)
reads = text[(pos[:, None] + np.arange(L)[None, :])].astype(np.uint8)
for r in reads[::3]:
    r[rng.integers(0, L)] = rng.integers(1, 5)
qbuf, qoff = reads.reshape(-1).copy(), (np.arange(nq + 1, dtype=np.uint64) * L)
sch = fm.search_scheme.h2(4, 0, 2)
os.environ["FMGPU_LF_TABLE"] = "0"
for layout in ("IB16", "EPR16", "EPRV2_16"):
    t0 = time.time()
    ox = fo.OraIndex.build(layout, 5, [text], 16, True)
    ix = fm.BiFMIndex.from_reference_arrays(**oracle_arrays(ox))
    for _ in range(2):
        lb, ln, st = fm.search_no_errors.search(ix, (qbuf, qoff), want_stats=True)
    for _ in range(2):
        hits, st2 = fm.search_ng26.search(ix, (qbuf[: 100_000 * L], qoff[: 100_001]), sch, want_stats=True, capacity=1 << 24)
    print("%-10s (oracle build %.0f s) %5.2f GB  exact %7.3f ms  k2 (100k reads) %8.3f ms" % (layout, time.time() - t0, ix.device_bytes / 1e9, st.kernel_ms, st2.kernel_ms), flush=True)

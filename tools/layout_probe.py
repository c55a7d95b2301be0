"""dev probe: exact and k = 2 search speed by occurrence-table layout on the same text and reads (plain index, no tables)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fmindex_collection_amd as fm
n, nq, L = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000, 1_000_000, 101
rng = np.random.default_rng(1)
text = rng.integers(1, 5, size=n, dtype=np.uint8)
pos = rng.integers(0, n - L, size=nq)
reads = text[(pos[:, None] + np.arange(L)[None, :])].astype(np.uint8)
for r in reads[::3]:
    r[rng.integers(0, L)] = rng.integers(1, 5)
qbuf, qoff = reads.reshape(-1).copy(), (np.arange(nq + 1, dtype=np.uint64) * L)
sch = fm.search_scheme.h2(4, 0, 2)
os.environ["FMGPU_LF_TABLE"] = "0"
for layout in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["IB16", "EPR16", "EPRV2_16", "EPRV5", "FBV_512_64K", "WAVELET"]):
    ix = fm.BiFMIndex.from_sequences([text], 5, layout, 16)
    for _ in range(2):
        lb, ln, st = fm.search_no_errors.search(ix, (qbuf, qoff), want_stats=True)
    for _ in range(2):
        hits, st2 = fm.search_ng26.search(ix, (qbuf[: 200_000 * L], qoff[: 200_001]), sch, want_stats=True, capacity=1 << 24)
    print("%-12s %6.2f GB  exact %7.3f ms  k2 (200k reads) %8.3f ms  hits %d" % (layout, ix.device_bytes / 1e9, st.kernel_ms, st2.kernel_ms, len(hits)), flush=True)
    ix.close()

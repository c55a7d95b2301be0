"""dev probe: search_backtracking (k = 2, Hamming) at GRCh38 scale, 1 M x 101 bp"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fmindex_collection_amd as fm
from fmindex_collection_amd import capi
import bench
dev = torch.device("cuda", 0)
total = sum(bench.GRCH38_LENGTHS)
g = torch.Generator(device=dev); g.manual_seed(42)
text = torch.randint(1, 5, (total,), generator=g, device=dev, dtype=torch.uint8)
seq_off = torch.from_numpy(np.concatenate([[0], np.cumsum(np.asarray(bench.GRCH38_LENGTHS, dtype=np.int64))])).to(dev)
index = fm.FMIndex.from_sequences((bench._Dev(text), bench._Dev(seq_off)), 5, "IB16", 16)
nq, L = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 101
starts = torch.randint(0, total - L, (nq,), generator=g, device=dev, dtype=torch.int64)
reads = text[starts[:, None] + torch.arange(L, device=dev)[None, :]].contiguous()
rows = torch.arange(nq, device=dev)
for k in range(2):
    sel = rows[rows % 3 > k]; p = torch.randint(0, L, (sel.numel(),), generator=g, device=dev)
    reads[sel, p] = reads[sel, p] % 4 + 1
qoff = torch.arange(nq + 1, device=dev, dtype=torch.int64) * L
hits = torch.empty(4 * nq * 40, dtype=torch.uint8, device=dev)
cnt = C.c_uint64(); st = capi.Stats()
for it in range(2):
    capi.check(capi.lib().fmgpu_search_backtracking(index._h, C.c_void_p(reads.data_ptr()), C.c_void_p(qoff.data_ptr()), nq, 2,
                                                    C.c_void_p(hits.data_ptr()), 4 * nq, C.byref(cnt), C.byref(st), None))
    print("backtracking k=2: %.1f ms, %.3g reads/s, %d hits, %d nodes" % (st.kernel_ms, nq / st.kernel_ms * 1e3, cnt.value, st.lf_steps), flush=True)

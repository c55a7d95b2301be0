"""dev probe: exact search through the C-ABI with HOST buffers (what include/fmc_gpu.hpp does): PCIe-inclusive time per batch"""
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
index.accelerate(3, lut_len=12, walk=True)
nq, L = 10_000_000, 101
starts = torch.randint(0, total - L, (nq,), generator=g, device=dev, dtype=torch.int64)
reads = text[starts[:, None] + torch.arange(L, device=dev)[None, :]].contiguous()
hq = reads.reshape(-1).cpu().numpy(); ho = (np.arange(nq + 1, dtype=np.uint64) * L)
lb = np.empty(nq, dtype=np.uint64); ln = np.empty(nq, dtype=np.uint64)
for it in range(3):
    t0 = time.perf_counter()
    capi.check(capi.lib().fmgpu_search_exact(index._h, capi.ptr(hq), capi.ptr(ho), nq, capi.ptr(lb), capi.ptr(ln), None, None))
    dt = time.perf_counter() - t0
    print("host-buffer call: %.1f ms -> %.3g reads/s (found %d)" % (dt * 1e3, nq / dt, int((ln > 0).sum())), flush=True)

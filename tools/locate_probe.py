"""dev probe: locate throughput at GRCh38 scale (rows = the intervals of 10 M exact hits)"""
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
t0 = time.time()
index = fm.FMIndex.from_sequences((bench._Dev(text), bench._Dev(seq_off)), 5, "IB16", 16)
print("build", time.time() - t0, flush=True)
nr = 10_000_000
rows = torch.randint(0, total, (nr,), generator=g, device=dev, dtype=torch.int64)
seq = torch.empty_like(rows); pos = torch.empty_like(rows); steps = torch.empty_like(rows)
st = capi.Stats()
for it in range(3):
    capi.check(capi.lib().fmgpu_locate(index._h, C.c_void_p(rows.data_ptr()), nr, C.c_void_p(seq.data_ptr()), C.c_void_p(pos.data_ptr()),
                                       C.c_void_p(steps.data_ptr()), C.byref(st), None))
    torch.cuda.synchronize()
    print("locate ms", st.kernel_ms, "rows/s", nr / st.kernel_ms * 1e3, "lf steps", st.lf_steps, flush=True)

t0 = time.time(); index.accelerate_locate(); print("locate table build", time.time() - t0, "device bytes", index.device_bytes, flush=True)
seq2 = torch.empty_like(rows); pos2 = torch.empty_like(rows); steps2 = torch.empty_like(rows)
for it in range(3):
    capi.check(capi.lib().fmgpu_locate(index._h, C.c_void_p(rows.data_ptr()), nr, C.c_void_p(seq2.data_ptr()), C.c_void_p(pos2.data_ptr()),
                                       C.c_void_p(steps2.data_ptr()), C.byref(st), None))
    torch.cuda.synchronize()
    print("locate (table) ms", st.kernel_ms, "rows/s", nr / st.kernel_ms * 1e3, "lf steps", st.lf_steps, flush=True)
print("equal", bool(torch.equal(seq, seq2) and torch.equal(pos, pos2) and torch.equal(steps, steps2)))

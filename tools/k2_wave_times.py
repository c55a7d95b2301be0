"""dev tool: when the waves of k_scheme_fast start and finish (FMGPU_DEV_FLAGS=65: count-only mode, wave times into the hit buffer):
python tools/k2_wave_times.py genome [plain]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np, torch
import fmindex_collection_amd as fm
from fmindex_collection_amd import capi, datasets
import bench
dev = torch.device("cuda", 0)
name = sys.argv[1]; plain = len(sys.argv) > 2 and sys.argv[2] == "plain"
nq = 10_000_000
class Ctx: pass
c = Ctx(); c.torch, c.dev, c.rank, c.np, c.datasets = torch, dev, 0, np, datasets
c.args = type("A", (), {"scale": 1.0})()
text, seq_off, lengths, info = bench.make_text(c, name)
qb, qo = bench.sample_reads(c, text, lengths, 101, nq, 2000 + 17 * 101, "k2")
os.environ["FMGPU_LF_TABLE"] = "0"
gx = fm.BiFMIndex.from_sequences((bench._Dev(text), bench._Dev(seq_off)), 5, "IB16", 16)
os.environ.pop("FMGPU_LF_TABLE")
del text
if not plain:
    gx.accelerate_lf(True); gx.accelerate_search(16, 3)
sc = bench._scheme_struct(capi, fm.search_scheme.h2(4, 0, 2))
out = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
cnt = C.c_uint64(); stats = capi.Stats()
os.environ["FMGPU_DEV_FLAGS"] = "65"
for rep in range(2):
    out.zero_()
    rc = capi.lib().fmgpu_search_scheme(gx._h, C.c_void_p(qb.data_ptr()), C.c_void_p(qo.data_ptr()), nq, C.byref(sc[0]), capi.UINT64_MAX,
                                        C.c_void_p(out.data_ptr()), 1 << 17, C.byref(cnt), C.byref(stats), None)
torch.cuda.synchronize()
v = out.cpu().numpy().astype(np.int64).reshape(-1, 2)
v = v[v[:, 1] > 0]
t0 = v[:, 0].min(); end = (v[:, 1] - t0) / 100e3; start = (v[:, 0] - t0) / 100e3          # ms
total = end.max()
busy = (end - start).sum() / (len(v) * total)
print(json.dumps({"rc": rc, "kernel_ms": stats.kernel_ms, "waves": int(len(v)), "span_ms": float(total), "wave_busy_fraction": float(busy),
                  "finish_ms_percentiles": {str(p): float(np.percentile(end, p)) for p in (1, 10, 25, 50, 75, 90, 99, 100)},
                  "start_ms_max": float(start.max())}))

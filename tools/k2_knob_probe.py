"""dev tool: k = 2 kernel time for several FMGPU_DEV_FLAGS settings on one index (text = genome | uniform; tables on):
python tools/k2_knob_probe.py genome 0,1024,2048,8192 [plain|tables [edit]]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np, torch
import fmindex_collection_amd as fm
from fmindex_collection_amd import capi, datasets
import bench
dev = torch.device("cuda", 0)
name = sys.argv[1]
flags = [int(x) for x in sys.argv[2].split(",")]
plain = len(sys.argv) > 3 and sys.argv[3] == "plain"
edit = len(sys.argv) > 4 and sys.argv[4] == "edit"
nq = 2_000_000 if edit else 10_000_000
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
sc[0].edit = 1 if edit else 0
probe = C.c_uint64()
capi.lib().fmgpu_search_scheme(gx._h, C.c_void_p(qb.data_ptr()), C.c_void_p(qo.data_ptr()), nq, C.byref(sc[0]), capi.UINT64_MAX, None, 0, C.byref(probe), None, None)
cap = int(probe.value) + 1024
out = torch.empty(cap * 40, dtype=torch.uint8, device=dev)
cnt = C.c_uint64(); stats = capi.Stats()
for f in flags:
    os.environ["FMGPU_DEV_FLAGS"] = str(f)
    ms = []
    for rep in range(3):
        capi.check(capi.lib().fmgpu_search_scheme(gx._h, C.c_void_p(qb.data_ptr()), C.c_void_p(qo.data_ptr()), nq, C.byref(sc[0]), capi.UINT64_MAX,
                                                  C.c_void_p(out.data_ptr()), cap, C.byref(cnt), C.byref(stats), None))
        ms.append(round(stats.kernel_ms, 3))
    print(json.dumps({"text": name, "plain": plain, "edit": edit, "flags": f, "kernel_ms": ms, "hits": int(cnt.value), "nodes": int(stats.lf_steps)}), flush=True)

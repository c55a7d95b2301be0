import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fmindex_collection_amd import capi
n = C.c_int()
print("device_count rc", capi.lib().fmgpu_device_count(C.byref(n)), n.value, flush=True)
p = C.c_void_p()
print("malloc rc", capi.lib().fmgpu_malloc(C.byref(p), 1 << 20), flush=True)
import torch
t = torch.ones(10, device="cuda")
print("torch ok", float(t.sum()), flush=True)
maps = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l]
print(sorted(set(maps)))

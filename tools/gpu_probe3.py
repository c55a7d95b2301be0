"""exact-search kernel variants at scale (dev tool)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import fmindex_collection_amd as fm
from tests.util import make_text, splitmix64
n = int(os.environ.get("PROBE_N", 1_000_000_000)); nq = int(os.environ.get("PROBE_NQ", 10_000_000))
t0 = time.time(); text = make_text(n, 5, seed=42); print(f"text gen {time.time()-t0:.1f}s", flush=True)
t0 = time.time(); gx = fm.FMIndex.from_sequences((text, np.array([0, n], dtype=np.uint64)), 5, "IB16", 16)
print(f"n={n}: GPU build {time.time()-t0:.1f}s device_bytes={gx.device_bytes/1e9:.2f} GB", flush=True)
with np.errstate(over="ignore"):
    r = splitmix64(np.arange(nq, dtype=np.uint64) + (np.uint64(5) << np.uint64(32)))
starts = (r % np.uint64(n - 101)).astype(np.int64)
qbuf = np.empty(nq * 101, dtype=np.uint8)
for lo in range(0, nq, 1_000_000):
    hi = min(nq, lo + 1_000_000)
    qbuf[lo * 101: hi * 101] = text[starts[lo:hi, None] + np.arange(101)[None, :]].reshape(-1)
# 10 % of the reads get one substitution (early exits)
mut = np.arange(0, nq, 10); pos = (r[mut] >> np.uint64(20)) % np.uint64(101)
ix = mut * 101 + pos.astype(np.int64); qbuf[ix] = qbuf[ix] % 4 + 1
qoff = np.arange(nq + 1, dtype=np.uint64) * np.uint64(101)
dq, do = fm.DeviceBuffer.from_array(qbuf), fm.DeviceBuffer.from_array(qoff)
dlb, dln = fm.DeviceBuffer(nq * 8), fm.DeviceBuffer(nq * 8)
ref = None
for v in (0, 1, 2, 3, 2, 0):
    os.environ["FMGPU_EXACT_VARIANT"] = str(v)
    for it in range(3):
        lb, ln, st = fm.search_no_errors.search(gx, (dq, do), out=(dlb, dln), want_stats=True)
    res = (dlb.to_array(np.uint64, nq), dln.to_array(np.uint64, nq))
    if ref is None: ref = res
    same = np.array_equal(res[0], ref[0]) and np.array_equal(res[1], ref[1])
    print(f"variant {v}: {st.kernel_ms:.2f} ms, steps {st.lf_steps}, {nq / st.kernel_ms / 1e3:.1f} Mq/s, {st.lf_steps * 112 / st.kernel_ms / 1e6:.0f} GB/s alg; same as v0: {same}", flush=True)
print("hits", int((ref[1] > 0).sum()))

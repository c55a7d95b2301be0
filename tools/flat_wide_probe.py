"""dev probe: Format S (k_exact_s) against the wavelet levels (k_exact_m) on a protein index of > 2^32 rows"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import fmindex_collection_amd as fm
nseq, slen, sigma, nq, L = int(sys.argv[1]) if len(sys.argv) > 1 else 9_000_000, 500, 28, 10_000_000, 40
dev = torch.device("cuda:0")
total = nseq * slen
g = torch.Generator(device=dev); g.manual_seed(42)
text = torch.empty(total, dtype=torch.uint8, device=dev)
for lo in range(0, total, 1 << 28):
    hi = min(total, lo + (1 << 28))
    text[lo:hi] = torch.randint(1, sigma, (hi - lo,), generator=g, device=dev, dtype=torch.uint8)
seq_off = torch.arange(nseq + 1, device=dev, dtype=torch.int64) * slen
s = torch.randint(0, nseq, (nq,), generator=g, device=dev) * slen + torch.randint(0, slen - L, (nq,), generator=g, device=dev)
qbuf = text[(s[:, None] + torch.arange(L, device=dev)[None, :]).reshape(-1)].contiguous()
qoff = torch.arange(nq + 1, device=dev, dtype=torch.int64) * L
class V:
    def __init__(s_, t): s_.ptr, s_.nbytes = t.data_ptr(), t.numel() * t.element_size()
os.environ["FMGPU_LF_TABLE"] = "0"
t0 = time.time()
ix = fm.FMIndex.from_sequences((V(text), V(seq_off)), sigma, "WAVELET", 16)
print("built in %.1f s, row_bits %d, device_bytes %.2f GB" % (time.time() - t0, ix.row_bits, ix.device_bytes / 1e9), flush=True)
res = {}
for flags in ("0", str(1 << 21)):
    os.environ["FMGPU_DEV_FLAGS"] = flags
    lb, ln = fm.DeviceBuffer(8 * nq), fm.DeviceBuffer(8 * nq)
    for _ in range(3):
        st = fm.search_no_errors.search(ix, (V(qbuf), V(qoff)), out=(lb, ln), want_stats=True)[-1]
    fm.capi.check(fm.capi.lib().fmgpu_synchronize(None))
    print("kernel_ms %.3f steps %d accesses %d" % (st.kernel_ms, st.lf_steps, st.table_accesses))
    res[flags] = (lb.to_array(np.uint64, nq), ln.to_array(np.uint64, nq))
    print("flags", flags, "hits", int((res[flags][1] > 0).sum()), "first", res[flags][0][:4], res[flags][1][:4], flush=True)
a, b = res["0"], res[str(1 << 21)]
print("equal:", bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])))

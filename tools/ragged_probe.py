"""tools/ragged_probe.py — a RAGGED batch (trimmed reads: lengths uniform in 50 .. 151) through k = 2 Hamming search on the plain genome-text index: the library cuts the batch into one
launch per read length, so every launch is small and ends with the waves that hold its heaviest reads — with and without work sharing between waves (the board)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fmindex_collection_amd as fm
from fmindex_collection_amd import capi, datasets
import bench
dev = torch.device("cuda", 0)
class _V:
    def __init__(self, t): self.t, self.ptr, self.nbytes = t, t.data_ptr(), t.numel() * t.element_size()
lengths = list(bench.GRCH38_LENGTHS)
text, _ = datasets.genome_like_text(lengths, seed=42, device=dev)
seq_off = torch.from_numpy(np.concatenate([[0], np.cumsum(np.asarray(lengths, dtype=np.int64))])).to(dev)
fm.options["lf_table"] = 0
gx = fm.BiFMIndex.from_sequences((_V(text), _V(seq_off)), 5, "IB16", 16)
nq = int(os.environ.get("RAGGED_READS", "4000000"))
g = torch.Generator(device=dev); g.manual_seed(9)
rl = torch.randint(50, 152, (nq,), generator=g, device=dev, dtype=torch.int64)
qoff = torch.zeros(nq + 1, dtype=torch.int64, device=dev); qoff[1:] = torch.cumsum(rl, 0)
starts = torch.randint(0, text.numel() - 200, (nq,), generator=g, device=dev, dtype=torch.int64)
total = int(qoff[-1].item())
owner = torch.repeat_interleave(torch.arange(nq, device=dev), rl)
within = torch.arange(total, device=dev) - qoff[:-1][owner]
qbuf = text[starts[owner] + within].contiguous()
sub = torch.nonzero(torch.arange(nq, device=dev) % 3 > 0)[:, 0]                      # two reads in three carry a substitution
at = qoff[:-1][sub] + (torch.randint(0, 50, (sub.numel(),), generator=g, device=dev))
qbuf[at] = qbuf[at] % 4 + 1
del text, owner, within
out = torch.empty(200_000_000 * 6, dtype=torch.int64, device=dev)
sc, keep = bench._scheme_struct(capi, fm.search_scheme.h2(4, 0, 2))
for sel, tag in ((capi.SEL_NO_BOARD, "without the board"), (0, "with the board")):
    fm.options["kernel_select"] = sel
    best = None
    for _ in range(3):
        st = capi.Stats(); cnt = C.c_uint64()
        capi.check(capi.lib().fmgpu_search_scheme(gx._h, C.c_void_p(qbuf.data_ptr()), C.c_void_p(qoff.data_ptr()), nq, C.byref(sc), capi.UINT64_MAX, C.c_void_p(out.data_ptr()), 200_000_000,
                                                  C.byref(cnt), C.byref(st), None))
        best = st.kernel_ms if best is None else min(best, st.kernel_ms)
    print("%-18s %d reads of 50 .. 151 bp (102 launches): kernels %8.1f ms, nodes %d, records %d" % (tag, nq, best, st.lf_steps, cnt.value), flush=True)

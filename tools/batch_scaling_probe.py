"""tools/batch_scaling_probe.py — kernel time of the k = 2 searches (Hamming: k_scheme_lean, edit distance: k_scheme_fast_edit<PLAIN>) on the plain genome-text index against the
batch size: a time that does not fall with the batch is the tail of the heaviest reads, not throughput"""
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
if os.environ.get("PROBE_NO_BOARD"):
    fm.options["kernel_select"] = capi.SEL_NO_BOARD
gx = fm.BiFMIndex.from_sequences((_V(text), _V(seq_off)), 5, "IB16", 16)
c = bench.Ctx(); c.torch, c.dev = torch, dev
qb, qo = bench.sample_reads(c, text, lengths, 101, 10_000_000, 2017 + 17 * 101, "k2")
del text
out = torch.empty(400_000_000 * 6, dtype=torch.int64, device=dev)       # room for the hit records (48 bytes each)
modes = ((1, (125_000, 250_000, 500_000, 1_000_000, 2_000_000)), (0, (500_000, 1_000_000, 2_000_000, 5_000_000, 10_000_000)))
if os.environ.get("PROBE_EDIT_ONLY"):
    modes = modes[:1]
if os.environ.get("PROBE_HAMMING_ONLY"):
    modes = modes[1:]
for edit, sizes in modes:
    sc, keep = bench._scheme_struct(capi, fm.search_scheme.h2(4, 0, 2))
    sc.edit = edit
    for nq in sizes:
        st = capi.Stats(); cnt = C.c_uint64()
        best = None
        for _ in range(int(os.environ.get("PROBE_REPEATS", "3"))):
            capi.check(capi.lib().fmgpu_search_scheme(gx._h, C.c_void_p(qb.data_ptr()), C.c_void_p(qo.data_ptr()), nq, C.byref(sc), capi.UINT64_MAX, C.c_void_p(out.data_ptr()), 400_000_000,
                                                      C.byref(cnt), C.byref(st), None))
            best = st.kernel_ms if best is None else min(best, st.kernel_ms)
        print("%s  %9d reads  kernel %8.2f ms  nodes %12d  %6.1f G nodes/s  hits %d" % ("edit   " if edit else "hamming", nq, best, st.lf_steps, st.lf_steps / best / 1e6, cnt.value), flush=True)

"""dev tool (library built with `make DEV=1`): how the lane-iterations of k_scheme_fast_edit<PLAIN> split on the genome-like and the uniform text — nodes of several rows, re-visits of
such nodes for their next child, one-row iterations that load nothing (k = 2 edit distance, 101 bp, 500 k reads): FMGPU_LIBRARY=fmindex-collection_amd/libfmgpu_dev.so python tools/edit_mix_probe.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fmindex_collection_amd as fm
from fmindex_collection_amd import capi, datasets
import bench
dev = torch.device("cuda", 0)
class _V:
    def __init__(self, t): self.t, self.ptr, self.nbytes = t, t.data_ptr(), t.numel() * t.element_size()
nq = 500_000
for name in ("genome", "uniform"):
    lengths = list(bench.GRCH38_LENGTHS)
    if name == "genome":
        text, _ = datasets.genome_like_text(lengths, seed=42, device=dev)
    else:
        g = torch.Generator(device=dev); g.manual_seed(42)
        text = torch.randint(1, 5, (sum(lengths),), generator=g, device=dev, dtype=torch.uint8)
    seq_off = torch.from_numpy(np.concatenate([[0], np.cumsum(np.asarray(lengths, dtype=np.int64))])).to(dev)
    fm.options["lf_table"] = 0
    gx = fm.BiFMIndex.from_sequences((_V(text), _V(seq_off)), 5, "IB16", 16)
    c = bench.Ctx(); c.torch, c.dev = torch, dev
    qb, qo = bench.sample_reads(c, text, lengths, 101, nq, 2017 + 17 * 101, "k2")
    sc, keep = bench._scheme_struct(capi, fm.search_scheme.h2(4, 0, 2))
    sc.edit = 1
    st = capi.Stats(); cnt = C.c_uint64()
    capi.lib().fmgpu_search_scheme(gx._h, C.c_void_p(qb.data_ptr()), C.c_void_p(qo.data_ptr()), nq, C.byref(sc), capi.UINT64_MAX, None, 0, C.byref(cnt), C.byref(st), None)   # (no room for records: the counters are what is read)
    m40 = (1 << 40) - 1
    multi = st.table_accesses & m40; iters = st.table_accesses >> 40; busy = st.table_bytes & m40; noload = st.table_bytes >> 40; revisit = st.table_steps
    print("%s: nodes %d, busy lane-iterations %d, on nodes of several rows %d (%.1f %%; re-visits for the next child %d = %.1f %% of all), one-row iterations without a load %d (%.1f %%), "
          "wave iterations %d, lanes busy per iteration %.1f of 64, kernel %.2f ms" % (name, st.lf_steps, busy, multi, 100.0 * multi / max(busy, 1), revisit, 100.0 * revisit / max(busy, 1),
                                                                                      noload, 100.0 * noload / max(busy, 1), iters, busy / max(iters, 1), st.kernel_ms))
    gx.close(); del text

"""tools/board_stress.py — the board under repetition: 400 launches each of the k = 2 edit-distance and Hamming searches on small batches of the genome text (most waves of the chip wait at
the board in every one of them), batch sizes varied; every launch must return the record count and node count of the first launch of its batch size, and none may report a wave that gave up"""
import os, sys, ctypes as C, time
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
fm.options["lf_table"] = int(os.environ.get("STRESS_LF_TABLE", "0"))      # 1: the table-driven instantiations (k_scheme_fast, k_scheme_fast_edit<5, 5>)
gx = fm.BiFMIndex.from_sequences((_V(text), _V(seq_off)), 5, "IB16", 16)
c = bench.Ctx(); c.torch, c.dev = torch, dev
qb, qo = bench.sample_reads(c, text, lengths, 101, 600_000, 2017 + 17 * 101, "k2")
del text
out = torch.empty(60_000_000 * 6, dtype=torch.int64, device=dev)
rounds = int(os.environ.get("STRESS_ROUNDS", "400"))
t0 = time.time()
for edit, sizes in ((1, (20_000, 61_000, 125_000)), (0, (50_000, 200_000, 600_000))):
    sc, keep = bench._scheme_struct(capi, fm.search_scheme.h2(4, 0, 2)); sc.edit = edit
    want = {}
    for r in range(rounds):
        nq = sizes[r % len(sizes)]
        st = capi.Stats(); cnt = C.c_uint64()
        capi.check(capi.lib().fmgpu_search_scheme(gx._h, C.c_void_p(qb.data_ptr()), C.c_void_p(qo.data_ptr()), nq, C.byref(sc), capi.UINT64_MAX, C.c_void_p(out.data_ptr()), 60_000_000,
                                                  C.byref(cnt), C.byref(st), None))
        got = (cnt.value, st.lf_steps)
        assert (st.hits >> 48) == 0, ("development build: LDS frame slots that disagreed with the stack in HBM", st.hits >> 48)
        if nq not in want: want[nq] = got
        assert got == want[nq], (edit, nq, r, got, want[nq])
    print("%s: %d launches, records / nodes per batch size %s, %.1f s so far" % ("edit distance" if edit else "hamming", rounds, want, time.time() - t0), flush=True)
print("board stress ok")

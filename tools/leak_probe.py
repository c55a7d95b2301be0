#!/usr/bin/env python3
"""dev probe: device memory before / after thousands of small calls of every search entry point (host and device buffers, several threads)"""
import ctypes as C, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import fmindex_collection_amd as fm
import fmoracle as fo
from tests.util import make_text, sample_reads, oracle_arrays
hip = C.CDLL("libamdhip64.so.7")
def free_bytes():
    a, b = C.c_size_t(), C.c_size_t(); assert hip.hipMemGetInfo(C.byref(a), C.byref(b)) == 0; return a.value
text = make_text(100_000, 5, seed=3)
ox = fo.OraIndex.build("IB16", 5, [text], 8, True)
gx = fm.BiFMIndex.from_reference_arrays(**oracle_arrays(ox))
gx.accelerate(3, lut_len=5, walk=2).accelerate_search(4, 3).accelerate_locate()
reads = sample_reads(text, 300, 40, seed=5, mutate=1)
qbuf, qoff = fm.flatten(reads)
sch = fm.search_scheme.h2(3, 0, 1)
ex = fm.search_scheme.expand(fm.search_scheme.pigeon_opt(0, 1), 40)
rows = np.arange(0, 2000, dtype=np.uint64)
def work(n):
    for i in range(n):
        fm.search_no_errors.search(gx, (qbuf, qoff)); fm.search_no_errors.search_packed(gx, (qbuf, qoff))
        fm.search_ng26.search(gx, (qbuf, qoff), sch); fm.search_ng26.search(gx, (qbuf, qoff), sch, edit=True)
        fm.search_ng21.search(gx, (qbuf, qoff), ex); fm.search_backtracking.search(gx, (qbuf, qoff), 1); gx.locate(rows)
work(50)
before = free_bytes()
ths = [threading.Thread(target=work, args=(400,)) for _ in range(3)]
for t in ths: t.start()
for t in ths: t.join()
work(400)
after = free_bytes()
print("free before %d MB, after %d MB, difference %d KB (threads that ended returned their frame stacks)" % (before >> 20, after >> 20, (before - after) >> 10))
assert before - after < (64 << 20), "device memory grows with the number of calls"

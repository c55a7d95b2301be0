"""first GPU contact: smoke + a small timing probe (dev tool, not part of the product)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import __graft_entry__ as g
t0 = time.time(); g.smoke(); print("smoke took", time.time() - t0, flush=True)
import fmindex_collection_amd as fm
import fmoracle as fo
from tests.util import make_text, sample_reads, oracle_arrays
n = int(os.environ.get("PROBE_N", 4_000_000))
text = make_text(n, 5, seed=42)
t0 = time.time(); ox = fo.OraIndex.build("IB16", 5, [text], 16, True); print("oracle build", time.time() - t0, flush=True)
gx = fm.BiFMIndex.from_reference_arrays(**oracle_arrays(ox))
nq = 1_000_000
reads = sample_reads(text, nq, 101, seed=5, mutate=0)
qbuf = np.ascontiguousarray(reads).reshape(-1); qoff = (np.arange(nq + 1, dtype=np.uint64) * np.uint64(101))
dq, do = fm.DeviceBuffer.from_array(qbuf), fm.DeviceBuffer.from_array(qoff)
dlb, dln = fm.DeviceBuffer(nq * 8), fm.DeviceBuffer(nq * 8)
for it in range(4):
    lb, ln, st = fm.search_no_errors.search(gx, (dq, do), out=(dlb, dln), want_stats=True)
    print(f"exact: {st.kernel_ms:.3f} ms, steps {st.lf_steps}, {nq / st.kernel_ms / 1e3:.2f} Mq/s, {st.lf_steps * 112 / st.kernel_ms / 1e6:.1f} GB/s algorithmic", flush=True)
ln = dln.to_array(np.uint64, nq); lb = dlb.to_array(np.uint64, nq)
t0 = time.time(); olb, oln = ox.search_exact(qbuf, qoff, nthreads=8); dt = time.time() - t0
print("oracle exact 8 threads:", nq / dt / 1e6, "Mq/s; parity:", np.array_equal(lb, olb) and np.array_equal(ln, oln), flush=True)
# k = 2
nq2 = 200_000
reads2 = sample_reads(text, nq2, 101, seed=9, mutate=2)
qb2 = np.ascontiguousarray(np.stack(reads2) if isinstance(reads2, list) else reads2).reshape(-1); qo2 = (np.arange(nq2 + 1, dtype=np.uint64) * np.uint64(101))
sch = fm.search_scheme.h2(4, 0, 2)
for it in range(3):
    hits, st = fm.search_ng26.search(gx, (qb2, qo2), sch, want_stats=True)
    print(f"k=2: {st.kernel_ms:.3f} ms, nodes {st.lf_steps}, hits {st.hits}, {nq2 / st.kernel_ms / 1e3:.3f} Mq/s", flush=True)
t0 = time.time(); ohits, qc, nodes = ox.search_ng26(qb2, qo2, sch, cap=1 << 22); dt = time.time() - t0
print("oracle k=2 1 thread:", nq2 / dt / 1e3, "kq/s nodes", nodes, "hits", len(ohits))
ok = len(hits) == len(ohits) and all(np.array_equal(hits[k], ohits[k]) for k in ("qidx", "lb", "lb_rev", "len")) and np.array_equal(hits["errors"], ohits["errors"])
print("k=2 parity:", ok, "nodes equal:", nodes == st.lf_steps)
hb, st = fm.search_backtracking.search(gx, (qb2[:101 * 20000], qo2[:20001]), 2, want_stats=True)
ob, onodes = ox.search_backtracking(qb2[:101 * 20000], qo2[:20001], 2, cap=1 << 22)
okb = len(hb) == len(ob) and all(np.array_equal(hb[k], ob[k]) for k in ("qidx", "lb", "lb_rev", "len")) and np.array_equal(hb["errors"], ob["errors"])
print(f"backtracking k=2 20k reads: {st.kernel_ms:.2f} ms parity {okb} nodes {st.lf_steps} vs {onodes}")

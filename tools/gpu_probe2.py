"""GPU builder validation + first at-scale timing (dev tool)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import fmindex_collection_amd as fm
import fmoracle as fo
from tests.util import make_text, sample_reads, oracle_arrays, splitmix64

def check_build(seqs, sigma, rate, bidir, name):
    gx = (fm.BiFMIndex if bidir else fm.FMIndex).from_sequences(seqs, sigma, "IB16", rate, keep_host=True)
    ox = fo.OraIndex.build("IB16", sigma, seqs, rate, bidir)
    n = ox.n
    obwt = np.array([ox.bwt_string().symbol(i) for i in range(n)], dtype=np.uint8) if n <= 20000 else None
    gb = gx.built_array(0)
    ok = True
    if obwt is not None: ok &= np.array_equal(gb, obwt)
    ok &= np.array_equal(gx.built_array(2, np.uint64), ox.C)
    sp = ox.sparse()
    ok &= np.array_equal(gx.built_array(3, np.uint64), sp["l0"])
    ok &= np.array_equal(gx.built_array(4, np.uint16), sp["l1"])
    ok &= np.array_equal(gx.built_array(5, np.uint64), sp["bits"])
    ok &= np.array_equal(gx.built_array(6, np.uint64), sp["fields"][0]["data"])
    ok &= np.array_equal(gx.built_array(7, np.uint64), sp["fields"][1]["data"])
    prm = gx.built_array(8, np.uint64)
    exp = [sp["fields"][0][k] for k in ("bitCount", "bits", "largestValue", "commonDivisor")] + [sp["fields"][1][k] for k in ("bitCount", "bits", "largestValue", "commonDivisor")]
    ok &= list(map(int, prm)) == list(map(int, exp))
    # rank tables of the device index == oracle
    idx = np.repeat(np.arange(n + 1, dtype=np.uint64), sigma); sym = np.tile(np.arange(sigma, dtype=np.uint8), n + 1)
    if n <= 20000:
        r = gx.rank(idx, sym); orr, _ = ox.bwt_string().rank_table()
        ok &= np.array_equal(r.reshape(n + 1, sigma), orr)
        if bidir:
            r = gx.rank(idx, sym, rev=True); orr, _ = ox.bwt_string(rev=True).rank_table()
            ok &= np.array_equal(r.reshape(n + 1, sigma), orr)
    rows = np.arange(n, dtype=np.uint64)
    s, p, st = gx.locate(rows)
    for r_ in range(0, n, max(1, n // 500)):
        ok &= (int(s[r_]), int(p[r_]), int(st[r_])) == ox.locate(r_)
    print(f"build check {name}: n={n} {'OK' if ok else 'MISMATCH'}", flush=True)
    return ok

allok = True
allok &= check_build([make_text(1000, 5, 1)], 5, 4, True, "random1k")
allok &= check_build([make_text(3000, 5, 2), make_text(10, 5, 3), make_text(700, 5, 4)], 5, 16, True, "3seq")
allok &= check_build([np.ones(5000, dtype=np.uint8)], 5, 7, True, "allA")
allok &= check_build([np.tile(np.array([1, 2, 3, 1, 2], dtype=np.uint8), 800)], 5, 3, True, "periodic")
allok &= check_build([np.array([1], dtype=np.uint8)], 5, 1, False, "tiny")
allok &= check_build([make_text(200, 28, 5), make_text(331, 28, 6)], 28, 5, False, "protein-sigma28")
allok &= check_build([make_text(70000, 4, 7, lo=1) % 2 + 1], 5, 16, True, "binary70k")
allok &= check_build([make_text(300000, 5, 8)], 5, 16, True, "random300k")
print("ALL BUILD CHECKS:", allok, flush=True)

# ---------- scale
for n in [int(x) for x in os.environ.get("PROBE_SIZES", "100000000,1000000000").split(",")]:
    t0 = time.time()
    text = make_text(n, 5, seed=42)
    print(f"n={n}: text gen {time.time()-t0:.1f}s", flush=True)
    t0 = time.time()
    soff = np.array([0, n], dtype=np.uint64)
    gx = fm.FMIndex.from_sequences((text, soff), 5, "IB16", 16)
    print(f"n={n}: GPU build {time.time()-t0:.1f}s device_bytes={gx.device_bytes/1e9:.2f} GB", flush=True)
    nq = 10_000_000
    with np.errstate(over="ignore"):
        r = splitmix64(np.arange(nq, dtype=np.uint64) + (np.uint64(5) << np.uint64(32)))
    starts = (r % np.uint64(n - 101)).astype(np.int64)
    t0 = time.time()
    qbuf = np.empty(nq * 101, dtype=np.uint8)
    for lo in range(0, nq, 1_000_000):
        hi = min(nq, lo + 1_000_000)
        qbuf[lo * 101: hi * 101] = text[starts[lo:hi, None] + np.arange(101)[None, :]].reshape(-1)
    qoff = np.arange(nq + 1, dtype=np.uint64) * np.uint64(101)
    print(f"reads gen {time.time()-t0:.1f}s", flush=True)
    dq, do = fm.DeviceBuffer.from_array(qbuf), fm.DeviceBuffer.from_array(qoff)
    dlb, dln = fm.DeviceBuffer(nq * 8), fm.DeviceBuffer(nq * 8)
    for it in range(4):
        lb, ln, st = fm.search_no_errors.search(gx, (dq, do), out=(dlb, dln), want_stats=True)
        print(f"n={n} exact: {st.kernel_ms:.2f} ms, steps {st.lf_steps}, {nq / st.kernel_ms / 1e3:.1f} Mq/s, {st.lf_steps * 112 / st.kernel_ms / 1e6:.0f} GB/s algorithmic", flush=True)
    ln = dln.to_array(np.uint64, nq)
    print("hits:", int((ln > 0).sum()), "of", nq, "multi:", int((ln > 1).sum()), flush=True)
    # locate a sample of results and verify against the known read origins
    lb = dlb.to_array(np.uint64, nq)
    sel = np.nonzero(ln == 1)[0][:200000]
    s, p, stp = gx.locate(lb[sel])
    print("locate parity with read origins:", bool(np.all(p + stp == starts[sel].astype(np.uint64))), flush=True)
    del gx, dq, do, dlb, dln

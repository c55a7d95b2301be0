"""shared helpers of the test-suite / smoke / bench: seeded synthetic inputs and the oracle -> C-ABI adapter"""
import numpy as np

MASK64 = (1 << 64) - 1


def splitmix64(x):
    """counter-based generator (vectorised): value i of stream `seed` = splitmix64(seed * 2^32 + i)"""
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(MASK64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def make_text(n, sigma, seed=42, lo=1):
    """n symbols uniform in [lo, sigma)"""
    with np.errstate(over="ignore"):
        r = splitmix64(np.arange(n, dtype=np.uint64) + (np.uint64(seed) << np.uint64(32)))
    return (r % np.uint64(sigma - lo) + np.uint64(lo)).astype(np.uint8)


def sample_reads(text, nreads, length, seed=1, mutate=0, sigma=5):
    """reads copied from uniform offsets of `text`; read i gets (i % (mutate+1)) substitutions at uniform positions"""
    with np.errstate(over="ignore"):
        r = splitmix64(np.arange(nreads, dtype=np.uint64) + (np.uint64(seed) << np.uint64(32)))
    starts = (r % np.uint64(len(text) - length + 1)).astype(np.int64)
    reads = text[starts[:, None] + np.arange(length)[None, :]].copy()
    if mutate:
        for k in range(mutate):
            with np.errstate(over="ignore"):
                rr = splitmix64(np.arange(nreads, dtype=np.uint64) + (np.uint64(seed + 101 + k) << np.uint64(32)))
            rows = np.nonzero((np.arange(nreads) % (mutate + 1)) > k)[0]
            pos = (rr[rows] % np.uint64(length)).astype(np.int64)
            shift = ((rr[rows] >> np.uint64(40)) % np.uint64(sigma - 2)).astype(np.uint8) + 1
            old = reads[rows, pos]
            reads[rows, pos] = (old - 1 + shift) % (sigma - 1) + 1
    return [reads[i] for i in range(nreads)] if nreads <= 200000 else reads


def string_arrays(s):
    """an OraString (standing in for a reference String object) -> keyword dict for the product's descriptors"""
    d = {"layout": s.layout, "sigma": s.sigma, "n": s.size()}
    if s.layout == "WAVELET":
        nn = 1 << max(1, (s.sigma - 1).bit_length())
        nodes = []
        for k in range(nn):
            nodes.append((s.raw(4 * k).view("<u8"), s.raw(4 * k + 1), s.raw(4 * k + 2).view("<u8"), int(s.raw(4 * k + 3).view("<u8")[0])))
        d["nodes"] = nodes
    elif s.layout.startswith("FBV_"):
        d["blocks"] = s.raw(0)                                   # String::bits
        d["super_blocks"] = s.raw(1).view("<u8")                 # String::l0, [k][sigma + 1]
        d["levels"] = [s.raw(2), None, None]                     # String::l1
        d["super_row"] = s.sigma + 1
    elif s.layout in ("EPRV3_8", "EPRV3_16", "EPRV3_32", "EPRV4", "EPRV5", "IEPRV7"):
        d["blocks"] = s.raw(0)                                   # String::bits
        d["super_blocks"] = s.raw(1).view("<u8")
        d["levels"] = [s.raw(2), s.raw(3), s.raw(4)]             # blocks_ / level0, level1, level2 (None or empty where absent)
    else:
        d["blocks"] = s.raw(0)
        d["super_blocks"] = s.raw(1).view("<u8")
    return d


def oracle_arrays(ox):
    """an OraIndex -> kwargs of FMIndex.from_reference_arrays (what a reference index object would hand over)"""
    out = {"bwt": string_arrays(ox.bwt_string()), "C_array": ox.C}
    if ox.bidirectional:
        out["bwt_rev"] = string_arrays(ox.bwt_string(rev=True))
    if ox.p.contents.sa:
        out["sparse"] = ox.sparse()
    return out


def occurrences(text_with_delims, pattern, max_mismatch=0):
    """brute force: start positions where pattern matches with <= max_mismatch substitutions"""
    t = np.asarray(text_with_delims)
    m = len(pattern)
    if m == 0 or m > len(t):
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    mism = np.zeros(len(t) - m + 1, dtype=np.int64)
    for j in range(m):
        mism += t[j: len(t) - m + 1 + j] != pattern[j]
    pos = np.nonzero(mism <= max_mismatch)[0]
    return pos, mism[pos]

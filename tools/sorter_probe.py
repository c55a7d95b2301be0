"""tools/sorter_probe.py TEXT [SORTERS] — build time of the suffix sorters (FMGPU_OPT_SUFFIX_SORTER 1 / 2 / 3) on a bench-sized text and equality of what they build, seen through
2 M exact searches + 200 k located rows.  TEXT: genome (3.09 Gbp repeat-structured, BiFMIndex) | uniform (3.09 Gbp uniform DNA, BiFMIndex) | protein (2e9) | protein_wide (4.5e9)"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fmindex_collection_amd as fm  # noqa: E402
from fmindex_collection_amd import capi, datasets  # noqa: E402


class Dev:
    def __init__(self, t):
        self.t, self.ptr, self.nbytes = t, t.data_ptr(), t.numel() * t.element_size()


def main():
    kind = sys.argv[1]
    sorters = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2").split(",")]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(5)
    if kind in ("genome", "uniform"):
        sys.path.insert(0, ROOT)
        import bench
        lengths = bench.GRCH38_LENGTHS
        sigma, L, layout, cls = 5, 101, "IB16", fm.BiFMIndex
        if kind == "genome":
            text, _ = datasets.genome_like_text(lengths, seed=42, device="cuda:0")
        else:
            text = torch.randint(1, 5, (int(sum(lengths)),), generator=g, device=dev, dtype=torch.uint8)
        seq_off = torch.tensor([0] + list(torch.tensor(lengths).cumsum(0).tolist()), device=dev, dtype=torch.int64)
    else:
        nseq = 4_000_000 if kind == "protein" else 9_000_000
        sigma, L, layout, cls = 28, 40, "WAVELET", fm.FMIndex
        text = torch.empty(nseq * 500, dtype=torch.uint8, device=dev)
        for lo in range(0, text.numel(), 1 << 28):
            hi = min(text.numel(), lo + (1 << 28))
            text[lo:hi] = torch.randint(1, sigma, (hi - lo,), generator=g, device=dev, dtype=torch.uint8)
        seq_off = torch.arange(nseq + 1, device=dev, dtype=torch.int64) * 500
    nq = 2_000_000
    starts = torch.randint(0, text.numel() - L, (nq,), generator=g, device=dev, dtype=torch.int64)
    reads = text[starts[:, None] + torch.arange(L, device=dev)[None, :]].contiguous()
    qoff = torch.arange(nq + 1, device=dev, dtype=torch.int64) * L
    fm.options["lf_table"] = 0
    out = {}
    ref = None
    for srt in sorters:
        fm.options["suffix_sorter"] = srt
        torch.cuda.synchronize(); t0 = time.time()
        try:
            index = cls.from_sequences((Dev(text), Dev(seq_off)), sigma, layout, 16)
        except fm.FmgpuError as e:
            out[srt] = {"error": str(e)[:200], "seconds": round(time.time() - t0, 2)}
            continue
        build_s = time.time() - t0
        res = torch.empty(2 * nq, dtype=torch.int64, device=dev)
        capi.check(capi.lib().fmgpu_search_exact(index._h, C.c_void_p(reads.data_ptr()), C.c_void_p(qoff.data_ptr()), nq, C.c_void_p(res[:nq].data_ptr()), C.c_void_p(res[nq:].data_ptr()), None, None))
        rows = res[:nq][res[nq:] > 0][:200_000].to(torch.uint64).contiguous()
        loc = torch.empty(3 * rows.numel(), dtype=torch.int64, device=dev)
        k = rows.numel()
        capi.check(capi.lib().fmgpu_locate(index._h, C.c_void_p(rows.data_ptr()), k, C.c_void_p(loc[:k].data_ptr()), C.c_void_p(loc[k:2 * k].data_ptr()), C.c_void_p(loc[2 * k:].data_ptr()), None, None))
        torch.cuda.synchronize()
        got = (res.clone(), loc.clone())
        if ref is None:
            ref = got
        out[srt] = {"build_s": round(build_s, 2), "device_bytes": index.device_bytes, "equal_to_first": bool(torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])),
                    "found": int((res[nq:] > 0).sum().item())}
        index.close(); del index
        torch.cuda.empty_cache()
    print(json.dumps({"text": kind, "symbols": text.numel(), "sorters": out}))


if __name__ == "__main__":
    main()

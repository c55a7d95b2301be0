"""dev tool: distribution of visited nodes per read for the k = 2 workload on the genome-like text (FMGPU_DEV_FLAGS=129 makes k_scheme_fast write the
node count of every read into the hit buffer instead of records)"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np, torch
import fmindex_collection_amd as fm
from fmindex_collection_amd import capi, datasets
import bench
dev = torch.device("cuda", 0)
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
lengths = [max(1, int(l * scale)) for l in bench.GRCH38_LENGTHS]
text, st = datasets.genome_like_text(lengths, seed=42, device=dev)
class Ctx: pass
c = Ctx(); c.torch, c.dev, c.rank = torch, dev, 0
qb, qo = bench.sample_reads(c, text, lengths, 101, nq, 2000 + 17 * 101, "k2")
seq_off = torch.from_numpy(np.concatenate([[0], np.cumsum(np.asarray(lengths, dtype=np.int64))])).to(dev)
os.environ["FMGPU_LF_TABLE"] = "0"
gx = fm.BiFMIndex.from_sequences((bench._Dev(text), bench._Dev(seq_off)), 5, "IB16", 16)
os.environ.pop("FMGPU_LF_TABLE")
# where the reads come from
total = int(text.numel())
del text
sc = bench._scheme_struct(capi, fm.search_scheme.h2(4, 0, 2))
out = torch.zeros(nq, dtype=torch.int64, device=dev)
cnt = C.c_uint64(); stats = capi.Stats()
os.environ["FMGPU_DEV_FLAGS"] = "129"
rc = capi.lib().fmgpu_search_scheme(gx._h, C.c_void_p(qb.data_ptr()), C.c_void_p(qo.data_ptr()), nq, C.byref(sc[0]), capi.UINT64_MAX,
                                    C.c_void_p(out.data_ptr()), nq, C.byref(cnt), C.byref(stats), None)
os.environ.pop("FMGPU_DEV_FLAGS")
torch.cuda.synchronize()
v = out.cpu().numpy().astype(np.int64)
srt = np.sort(v)[::-1]
res = {"rc": rc, "kernel_ms": stats.kernel_ms, "nodes_total": int(v.sum()), "stats_nodes": int(stats.lf_steps), "hits": int(cnt.value), "max": int(srt[0]), "top10": srt[:10].tolist(),
       "p50": int(np.percentile(v, 50)), "p90": int(np.percentile(v, 90)), "p99": int(np.percentile(v, 99)), "p999": int(np.percentile(v, 99.9)), "p9999": int(np.percentile(v, 99.99)),
       "share_top_0.01pct": float(srt[: max(1, nq // 10000)].sum() / v.sum()), "share_top_0.1pct": float(srt[: nq // 1000].sum() / v.sum()), "share_top_1pct": float(srt[: nq // 100].sum() / v.sum()),
       "reads_over_1e4": int((v > 1e4).sum()), "reads_over_1e5": int((v > 1e5).sum()), "reads_over_1e6": int((v > 1e6).sum())}
print(json.dumps(res))

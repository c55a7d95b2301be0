"""The N > 1 path over gloo with world_size 2 on CPU: contiguous query shards, per-rank search (the oracle stands in for
the GPU kernel here — this test covers the sharding and the gather, not the kernel), gather to rank 0, identical to one rank."""
import os
import sys
import tempfile

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fmoracle as fo
    from fmindex_collection_amd.parallel import shard_range, gather_ragged, gather_fixed
    from tests.util import make_text, sample_reads
    text = make_text(20000, 5, seed=3)
    x = fo.OraIndex.build("IB16", 5, [text], 16, True)                 # replicated index
    reads = sample_reads(text, 1001, 40, seed=8, mutate=1)
    lo, hi = shard_range(len(reads), world, rank)
    qbuf, qoff = fo.flatten_queries(reads[lo:hi])
    lb, ln = x.search_exact(qbuf, qoff)
    local = torch.from_numpy(np.stack([lb, ln], axis=1).astype(np.int64))
    parts = gather_ragged(local, dst=0)
    sch = fo.scheme_h2(3, 0, 1)
    hits, _, _ = x.search_ng26(qbuf, qoff, sch)
    rec = np.stack([hits["qidx"] + lo, hits["lb"], hits["lb_rev"], hits["len"], hits["errors"]], axis=1).astype(np.int64)
    hparts = gather_ragged(torch.from_numpy(rec), dst=0)
    fixed = gather_fixed(torch.tensor([rank, hi - lo], dtype=torch.int64), dst=0)
    if rank == 0:
        allq, allo = fo.flatten_queries(reads)
        flb, fln = x.search_exact(allq, allo)
        got = torch.cat(parts).numpy()
        assert np.array_equal(got[:, 0], flb.astype(np.int64)) and np.array_equal(got[:, 1], fln.astype(np.int64))
        fh, _, _ = x.search_ng26(allq, allo, sch)
        want = np.stack([fh["qidx"], fh["lb"], fh["lb_rev"], fh["len"], fh["errors"]], axis=1).astype(np.int64)
        assert np.array_equal(torch.cat(hparts).numpy(), want)
        assert [int(t[0]) for t in fixed] == list(range(world)) and sum(int(t[1]) for t in fixed) == len(reads)
        open(os.path.join(outdir, "ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_search_and_gather():
    port = 29500 + os.getpid() % 2000
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, port, d), nprocs=2, join=True)
        assert os.path.exists(os.path.join(d, "ok"))


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    """bench.py's N > 1 control flow (per-rank batches, double-buffered asynchronous exchange, barrier + max-over-ranks timing, one JSON line
    from rank 0) with two ranks sharing the one GPU of the test box; the collective travels over gloo here, over RCCL in the driver's run"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--scale", "0.02",
                        "--nq", "200000", "--total-k2-reads", "100000", "--lut-len", "10", "--prefix-len", "11", "--dist-backend", "gloo", "--all-ranks-device0"], capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1 and r.stdout.rstrip().endswith(lines[0]), r.stdout[-500:] + r.stderr[-1500:]
    assert len(lines[0]) < 4000                                   # the driver keeps 8 KB of stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and "cpu_baseline" not in out
    assert out["config"]["workload"] == "grch38_exact" and out["config"]["queries_per_gpu"] == 200000
    assert out["config"]["index_kind"] == "plain" and out["roofline"]["kernel"] == "k_exact_p"      # the headline: the plain index, at every N
    assert out["exchange"]["verified_on_rank0"] is True and out["exchange"]["bytes_per_rank_and_step"] > 0   # rank 0 received what both ranks sent
    assert out["exchange"]["world_size_seen"] == 2
    assert 0 < out["roofline"]["frac"] <= 1.0
    sec = out["secondary"]                                       # configs[3]: k = 2, 151 bp, the batch sharded over the ranks (strong scaling)
    assert sec["scaling"] == "strong" and sec["value"] > 0 and sec["exchange"]["verified_on_rank0"] is True and sec["exchange"]["world_size_seen"] == 2
    full = json.load(open(os.path.join(root, out["records_file"])))   # every full record, written beside bench.py
    rec = next(x for x in full["records"] if x["id"] == sec["record"])
    assert rec["config"]["read_len"] == 151 and rec["config"]["queries_per_gpu"] == 50000 and rec["config"]["partition"] == [38, 38, 38, 37]
    assert rec["config"]["index_kind"] == "plain"

#!/usr/bin/env python3
"""bench.py — throughput of the backward-search hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic reads that are already resident in HBM.
Default workload = BASELINE.json configs[1]: GRCh38-sized FMIndex (25 sequences with the GRCh38 chromosome
lengths, 3.09 Gbp, sigma = 5, uniform random bases — the real assembly is not available offline), 10 M x 101 bp
exact search.  `--workload k2` runs configs[2] (BiFMIndex, h2(4,0,2) search scheme, Hamming distance);
`--workload protein` runs configs[4] (UniRef50 stand-in: 4 M sequences x 500 residues = 2.0e9 residues uniform in
{1..27}, sigma = 28, FMIndex<28, Wavelet>, 10 M x 40 aa exact search).

    python bench.py --gpus N --steps K --warmup W

N > 1: launched by torch.distributed.run, one rank per GPU; the index is replicated, every rank searches its own
batch (weak scaling: per-GPU work fixed), and the resulting SA intervals are gathered to rank 0 over RCCL inside
the timed region — the path's only exchange step.  Result buffers are double-buffered so that the gather of step i
crosses xGMI while the kernel of step i+1 runs; the last gather is drained before the closing barrier.

At N = 1 the default run appends `secondary`: the k = 2 workload measured in a child process before this one touches the GPU
(BASELINE.json's metric names exact AND k = 2; `value` stays the exact figure).

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes (executed LF steps x
2 x sizeof(InterleavedBitvector16<5>::Block) = 112 B) / the search kernel's launch duration, measured with HIP
events on the launch stream inside the C-ABI.  `cpu_baseline` = the CPU restatement (oracle/, parity-pinned) on
the host cores over a bounded sample of the same reads.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GRCH38_LENGTHS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717,
                  133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285,
                  58617616, 64444167, 46709983, 50818468, 156040895, 57227415, 16569]
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BLOCK_BYTES_IB16_S5 = 56       # sizeof(InterleavedBitvector16<5>::Block), SURVEY.md appendix B
NOTE_DNA = ("unit = executed LF step (exact) / visited node (k=2), identical to the CPU walk; bytes_per_unit = 2 x sizeof(InterleavedBitvector16<5>::Block) "
            "of the reference layout (SURVEY 8d). The tables serve many LF steps per touched line (12 from the interval table, 3 per k-step entry, 16 per "
            "walk entry), so algorithmic bytes exceed the traffic; see line_rate for the hardware bound")
NOTE_PROTEIN = ("unit = executed LF step, identical to the CPU walk; bytes_per_unit = 2 ends x 5 levels x (8 + 1 + 8) B the reference's Wavelet rank reads "
                "(SURVEY 8d). The expanded block table answers a step from one 12-byte entry per end and the walk table 6 steps per entry, so algorithmic bytes exceed the traffic; see "
                "line_rate for the hardware bound")
PROTEIN_SEQS, PROTEIN_SEQ_LEN = 4_000_000, 500      # UniRef50 stand-in (the release itself is not available offline): 2.0e9 residues


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="exact", choices=["exact", "k2", "protein"])
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the text (dev runs only; the judged run uses 1.0)")
    ap.add_argument("--nq", type=int, default=10_000_000)
    ap.add_argument("--read-len", type=int, default=0, help="0 = the workload's own (101 bp, 40 aa)")
    ap.add_argument("--kstep", type=int, default=-1, help="exact search: symbols per table step (fmgpu_index_accelerate); 1 = plain occurrence table; "
                    "protein: 1 = block-table expansion of the wavelet (default), 0 = search the wavelet lines themselves")
    ap.add_argument("--no-search-accel", action="store_true", help="k2: no prefix / walk tables (fmgpu_index_accelerate_search)")
    ap.add_argument("--no-exact-tables", action="store_true", help="exact / protein: only the k-step table (no suffix-interval table, no walk table)")
    ap.add_argument("--lut-len", type=int, default=0, help="exact / protein: symbols of the interval table (0 = 15 bp / 6 aa)")
    ap.add_argument("--walk", type=int, default=2, help="exact / protein: 1 = LF^J walk table, 2 = LF^J and LF^2J")
    ap.add_argument("--prefix-len", type=int, default=16, help="k2: symbols of the prefix table (fmgpu_index_accelerate_search)")
    ap.add_argument("--trim", type=int, default=0, help="dev: every read loses 0..TRIM symbols at its end (a ragged batch)")
    ap.add_argument("--single-rank-collectives", action="store_true", help="rehearsal only: run the N > 1 code path (process group, asynchronous gather, barrier) with one rank")
    ap.add_argument("--sub-every", type=int, default=10, help="exact / protein: every n-th read carries one substitution (default 10 = SURVEY 8d; 0 = none; a dev knob)")
    ap.add_argument("--ng21", action="store_true", help="k2: search_ng21 over expand(h2(4,0,2), read length) (edit distance; the reference's older algorithm) — a side measurement")
    ap.add_argument("--edit", action="store_true", help="k2: edit distance (search_ng26<Edit = true>) instead of Hamming distance")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="default run at N = 1: do not append the k = 2 measurement (BASELINE's metric names exact AND k = 2)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="rehearsal only: gloo runs the N > 1 control flow where RCCL cannot (all ranks on one card); results travel through host memory")
    ap.add_argument("--all-ranks-device0", action="store_true", help="rehearsal only: every rank uses GPU 0")
    ap.add_argument("--cpu-sample", type=int, default=0, help="reads in the CPU baseline sample (0 = auto)")
    return ap.parse_args()


def main():
    args = parse()
    if args.ng21:                                             # a side measurement: no CPU leg (the restatement of search_ng21 is single-threaded), no second workload
        args.no_cpu_baseline = True
        args.no_secondary = True
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    secondary = None
    if world == 1 and args.workload == "exact" and not args.no_secondary and args.scale == 1.0 and args.nq == 10_000_000 and args.trim == 0:
        # the metric's second half — k = 2 Hamming on the same text — in a child process of its own, BEFORE this process touches the GPU (the two
        # indices do not fit the HBM together, and a GPU-initialised process must not start other programs); never part of `value`
        import subprocess
        try:
            child = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "k2", "--steps", str(args.steps), "--warmup", str(args.warmup),
                                    "--no-cpu-baseline"] + (["--read-len", str(args.read_len)] if args.read_len else []),
                                   capture_output=True, text=True, timeout=600)
            line = [l for l in child.stdout.splitlines() if l.startswith("{")]
            if child.returncode == 0 and line:
                k2 = json.loads(line[-1])
                secondary = {k: k2[k] for k in ("metric", "value", "unit", "ms_per_step", "gbp_per_s", "hits", "config", "roofline") if k in k2}
            else:
                secondary = {"error": (child.stderr or child.stdout)[-400:]}
        except Exception as ex:                                  # the primary measurement stands on its own
            secondary = {"error": repr(ex)}
    import numpy as np
    import torch
    import fmindex_collection_amd as fm
    from fmindex_collection_amd import capi

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.all_ranks_device0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    capi.check(capi.lib().fmgpu_set_device(local_rank))
    dev = torch.device("cuda", local_rank)
    dist = None
    multi = world > 1 or args.single_rank_collectives         # (dev: the N > 1 control flow and its RCCL calls with a group of one rank)
    if multi:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    # ---------------------------------------------------------------- synthetic inputs, generated in HBM
    protein = args.workload == "protein"
    sigma, layout = (28, "WAVELET") if protein else (5, "IB16")
    if not args.read_len:
        args.read_len = 40 if protein else 101
    if protein:
        lengths = [PROTEIN_SEQ_LEN] * max(1, int(PROTEIN_SEQS * args.scale))
    else:
        lengths = [max(1, int(l * args.scale)) for l in GRCH38_LENGTHS]
    total = sum(lengths)
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    text = torch.empty(total, dtype=torch.uint8, device=dev)
    chunk = 1 << 28
    for lo in range(0, total, chunk):                         # bases uniform in {1..4}; 0 is the delimiter
        hi = min(total, lo + chunk)
        text[lo:hi] = torch.randint(1, sigma, (hi - lo,), generator=g, device=dev, dtype=torch.uint8)
    seq_off = torch.from_numpy(np.concatenate([[0], np.cumsum(np.asarray(lengths, dtype=np.int64))])).to(dev)
    L, nq = args.read_len, args.nq
    gq = torch.Generator(device=dev)
    gq.manual_seed(1000 + rank)                               # every rank searches its own batch
    # reads = substrings of a random chromosome-internal position (so every unmutated read has >= 1 hit)
    if protein:                                               # inside one sequence: 500-residue entries are short next to the read
        starts = (torch.randint(0, len(lengths), (nq,), generator=gq, device=dev, dtype=torch.int64) * PROTEIN_SEQ_LEN +
                  torch.randint(0, PROTEIN_SEQ_LEN - L + 1, (nq,), generator=gq, device=dev, dtype=torch.int64))
    else:
        starts = torch.randint(0, total - L, (nq,), generator=gq, device=dev, dtype=torch.int64)
    reads = torch.empty((nq, L), dtype=torch.uint8, device=dev)
    ar = torch.arange(L, device=dev, dtype=torch.int64)
    for lo in range(0, nq, 1 << 20):
        hi = min(nq, lo + (1 << 20))
        reads[lo:hi] = text[starts[lo:hi, None] + ar[None, :]]
    if args.workload != "k2":                                 # 10 % of the reads carry one substitution (early exits)
        rows = torch.arange(0, nq, args.sub_every, device=dev) if args.sub_every > 0 else torch.arange(0, 0, device=dev)
        nsub = torch.ones_like(rows)
    else:                                                     # 0 / 1 / 2 substitutions in ratio 1:1:1
        rows = torch.arange(0, nq, device=dev)
        nsub = rows % 3
    for k in range(2):
        sel = rows[nsub > k]
        pos = torch.randint(0, L, (sel.numel(),), generator=gq, device=dev)
        shift = torch.randint(1, sigma - 1, (sel.numel(),), generator=gq, device=dev, dtype=torch.uint8)
        reads[sel, pos] = (reads[sel, pos] - 1 + shift) % (sigma - 1) + 1
    if args.trim > 0:
        lens = L - torch.randint(0, args.trim + 1, (nq,), generator=gq, device=dev, dtype=torch.int64)
        qbuf = reads[ar[None, :] < lens[:, None]].contiguous()
        qoff = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(lens, 0)])
    else:
        qbuf = reads.reshape(-1)
        qoff = (torch.arange(nq + 1, device=dev, dtype=torch.int64) * L)
    torch.cuda.synchronize()

    # ---------------------------------------------------------------- index construction on the GPU (not timed as a step)
    bidir = args.workload == "k2"
    want_cpu = (rank == 0 and world == 1 and not args.no_cpu_baseline)
    t0 = time.time()
    cls = fm.BiFMIndex if bidir else fm.FMIndex
    index = cls.from_sequences((_Dev(text), _Dev(seq_off)), sigma, layout, 16, keep_host=want_cpu)
    if args.kstep < 0:
        args.kstep = 1 if protein else 3
    if not bidir and args.kstep >= 1 and (not args.no_exact_tables or args.kstep > 1 or protein):
        if args.no_exact_tables:
            index.accelerate(args.kstep)
        else:                                                   # + interval table of the last 15 bp / 6 aa, + per-row LF^J walk table (J = 16 bp / 6 aa)
            index.accelerate(args.kstep, lut_len=args.lut_len or (6 if protein else 15), walk=args.walk)
    if bidir and not args.no_search_accel:
        try:
            index.accelerate_search(args.prefix_len, 3)
        except fm.FmgpuError as ex:                            # (the 16-symbol table is 69 GB: should the card be short of memory, one symbol less)
            if args.prefix_len < 16:
                raise
            print("bench.py: %s; retrying with a 15-symbol prefix table" % ex, file=sys.stderr, flush=True)
            args.prefix_len = 15
            index.accelerate_search(args.prefix_len, 3)
    build_s = time.time() - t0
    if not want_cpu:
        del text
    torch.cuda.empty_cache()

    # results are double-buffered: the gather of step i travels over xGMI while the kernel of step i+1 runs
    outs = [torch.empty(2 * nq, dtype=torch.int64, device=dev) for _ in range(2 if multi else 1)]   # [lb | len], one buffer so that the gather sends it as is
    scheme = fm.search_scheme.h2(4, 0, 2)
    hit_cap = (16 if (args.edit or args.ng21) else 4) * nq
    ex21 = None
    if args.ng21:
        import numpy as np
        if args.trim > 0 or not bidir:
            raise SystemExit("--ng21 needs --workload k2 and equal-length reads")
        args.edit = True
        arrs = tuple(np.ascontiguousarray(a, dtype=np.uint64) for a in fm.search_scheme.expand(scheme, L))
        ex = capi.ExpandedScheme()
        ex.n_searches, ex.length = arrs[0].shape
        ex.pi, ex.l, ex.u = (a.ctypes.data_as(capi.u64p) for a in arrs)
        ex21 = (ex, arrs)
    hits_bufs = [torch.empty(hit_cap * 40, dtype=torch.uint8, device=dev) for _ in range(2 if multi else 1)] if bidir else None
    via_host = multi and args.dist_backend == "gloo"
    count_dev = torch.zeros(1, dtype=torch.int64, device="cpu" if via_host else dev)
    packed = [torch.empty(nq, dtype=torch.int64, device=dev) for _ in range(2)] if (multi and not bidir) else None
    # k = 2: a hit travels as 16 bytes — qidx:32 | lb:32, len:32 | errors:8 | seq:24 (rows and batch sizes are < 2^32; lbRev stays on the rank that
    # found the hit: it only serves further extension) — instead of the 40-byte record of the C-ABI
    packed_hits = [torch.empty((2 * nq, 2), dtype=torch.int64, device=dev) for _ in range(2)] if (multi and bidir) else None
    gathered = None
    if multi and rank == 0:                               # k=2 messages are sized per step (largest hit count over the ranks), at most 2*nq records
        full = 2 * nq * (16 if bidir else 8)
        gathered = [[torch.empty(full, dtype=torch.uint8, device="cpu" if via_host else dev) for _ in range(world)] for _ in range(2)]

    import ctypes as C
    stats = capi.Stats()
    kernel_ms, units = [], []
    pending = [None, None]
    state = {"i": 0, "out": outs[0]}

    def step():
        b = state["i"] % len(outs)
        state["i"] += 1
        if pending[b] is not None:                             # the buffer's previous gather must have left before it is overwritten
            pending[b].wait(); pending[b] = None
        out = outs[b]
        state["out"] = out
        if not bidir and multi:                               # N > 1: the kernel writes the 8-byte transport word (lb << 32 | len) itself
            capi.check(capi.lib().fmgpu_search_exact_packed(index._h, C.c_void_p(qbuf.data_ptr()), C.c_void_p(qoff.data_ptr()), nq,
                                                            C.c_void_p(packed[b].data_ptr()), C.byref(stats), None))
        elif not bidir:
            capi.check(capi.lib().fmgpu_search_exact(index._h, C.c_void_p(qbuf.data_ptr()), C.c_void_p(qoff.data_ptr()), nq,
                                                     C.c_void_p(out[:nq].data_ptr()), C.c_void_p(out[nq:].data_ptr()),
                                                     C.byref(stats), None))
        elif args.ng21:
            cnt = C.c_uint64()
            capi.check(capi.lib().fmgpu_search_ng21(index._h, C.c_void_p(qbuf.data_ptr()), C.c_void_p(qoff.data_ptr()), nq,
                                                    C.byref(ex21[0]), capi.UINT64_MAX, C.c_void_p(hits_bufs[b].data_ptr()), hit_cap,
                                                    C.byref(cnt), C.byref(stats), None))
        else:
            sc = _scheme_struct(capi, scheme)
            sc[0].edit = 1 if args.edit else 0
            cnt = C.c_uint64()
            capi.check(capi.lib().fmgpu_search_scheme(index._h, C.c_void_p(qbuf.data_ptr()), C.c_void_p(qoff.data_ptr()), nq,
                                                      C.byref(sc[0]), capi.UINT64_MAX, C.c_void_p(hits_bufs[b].data_ptr()), hit_cap,
                                                      C.byref(cnt), C.byref(stats), None))
        kernel_ms.append(stats.kernel_ms)
        units.append(stats.lf_steps)
        if multi:                                          # the path's one exchange: SA intervals to rank 0 over RCCL/xGMI
            if bidir:
                count_dev.fill_(int(cnt.value))
                dist.all_reduce(count_dev, op=dist.ReduceOp.MAX)
                m = (int(count_dev.item()) + 65535) // 65536 * 65536
                if m > 2 * nq:
                    raise SystemExit("more than 2 hits per read on average: raise the gather buffers")
                capi.check(capi.lib().fmgpu_hits_pack16(C.c_void_p(hits_bufs[b].data_ptr()), int(cnt.value), C.c_void_p(packed_hits[b].data_ptr()), None))
                payload = packed_hits[b][:m].view(torch.uint8).view(-1)
            else:                                              # (lb, len) as one 64-bit word per read: rows are < 2^32, 80 MB per rank instead of 160
                payload = packed[b].view(torch.uint8)
            if via_host:
                payload = payload.cpu()
            state["payload"], state["b"] = payload, b
            pending[b] = exchange(payload, b)

    xch = {"mode": "gather", "bufs": [None, None]}

    def exchange(payload, b):
        """SA intervals / hit records to rank 0.  torch.distributed.gather (grouped send/recv in RCCL) is the exchange the path needs; should this
        build refuse it, every rank switches to all_gather_into_tensor — more bytes over xGMI, same information on rank 0"""
        if xch["mode"] == "gather":
            try:
                return dist.gather(payload, [g[: payload.numel()] for g in gathered[b]] if rank == 0 else None, dst=0, async_op=True)
            except (RuntimeError, NotImplementedError, ValueError) as ex:
                xch["mode"] = "all_gather"
                if rank == 0:
                    print("bench.py: dist.gather unavailable (%s); using all_gather_into_tensor" % ex, file=sys.stderr, flush=True)
        need = world * payload.numel()
        if xch["bufs"][b] is None or xch["bufs"][b].numel() < need:
            xch["bufs"][b] = torch.empty(need, dtype=torch.uint8, device=payload.device)
        return dist.all_gather_into_tensor(xch["bufs"][b][:need], payload, async_op=True)

    def drain():
        for b in range(2):
            if pending[b] is not None:
                pending[b].wait(); pending[b] = None

    for _ in range(args.warmup):
        step()
    drain()
    kernel_ms.clear(); units.clear()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    gather_ok = None
    if multi:                                             # outside the timed region: rank 0 holds what every rank sent in the last step
        pl = state["payload"]
        chk = pl.view(torch.int64).sum().reshape(1).to(count_dev.device)
        sums = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(sums, chk)
        if rank == 0:
            b = state["b"]
            if xch["mode"] == "gather":
                got = [gathered[b][r][: pl.numel()].view(torch.int64).sum().item() for r in range(world)]
            else:
                got = [xch["bufs"][b][r * pl.numel(): (r + 1) * pl.numel()].view(torch.int64).sum().item() for r in range(world)]
            gather_ok = got == [int(x.item()) for x in sums]
            if not gather_ok:
                raise SystemExit("bench.py: the gathered intervals on rank 0 differ from what the ranks sent")
    if rank != 0:
        if multi:
            dist.destroy_process_group()
        return
    ms_per_step = elapsed / args.steps * 1e3
    qps = world * nq * args.steps / elapsed
    k_ms = sum(kernel_ms) / len(kernel_ms)
    steps_per_launch = sum(units) / len(units)
    unit_bytes = 2 * 5 * 17 if protein else 2 * BLOCK_BYTES_IB16_S5      # Wavelet 2 x levels x (8 + 1 + 8) B per LF step
    alg_bytes = steps_per_launch * unit_bytes
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    out_lb, out_len = state["out"][:nq], state["out"][nq:]
    if multi and not bidir:                                   # the last step's transport words, taken apart again
        w = packed[state["b"]]
        out_lb, out_len = (w >> 32) & 0xffffffff, w & 0xffffffff
    hits = int((out_len > 0).sum().item()) if not bidir else int(stats.hits)
    traffic, lines = None, None                               # HBM bytes / line requests per launch from the committed PMC passes
    try:
        tall = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if protein:
            key = "protein_exact_wavelet_lines" if not args.kstep else ("protein_exact_block_table_only" if args.no_exact_tables else "protein_exact")
        elif bidir:
            key = "grch38_k2_ng21" if args.ng21 else "grch38_k2_edit" if args.edit else "grch38_k2"
        elif args.no_exact_tables:
            key = "grch38_exact_kstep3_only" if args.kstep == 3 else "grch38_exact_kstep%d" % args.kstep
        else:
            key = "grch38_exact" if args.kstep == 3 else "grch38_exact_kstep%d_tables" % args.kstep
        tj = tall[key]
        if args.scale == 1.0 and nq == 10_000_000 and L == (40 if protein else 101):
            traffic, lines = tj["bytes_per_launch"], tj["line_requests_per_launch"]
            ceiling = tall["_random_line_ceiling_G_per_s"]["value"]
    except Exception:
        traffic, lines = None, None
    result = {
        "metric": ("queries/sec (sigma=28 protein index, 10M x 40aa, exact, Wavelet)" if protein else
                   "queries/sec (GRCh38-sized index, 10M x 101bp, %s)" % ("exact" if not bidir else ("k=2 edit distance, search_ng21 over expand(h2(4,0,2))" if args.ng21 else "k=2 edit distance, h2(4,0,2)" if args.edit else "k=2 Hamming, h2(4,0,2)"))),
        "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": "protein_exact" if protein else "grch38_%s" % ("exact" if not bidir else ("k2_ng21" if args.ng21 else "k2_edit" if args.edit else "k2")), "text_symbols": total, "sequences": len(lengths),
                   "sigma": sigma, "layout": "Wavelet" if protein else "InterleavedBitvector16", "queries_per_gpu": nq, "read_len": L,
                   "index": "BiFMIndex" if bidir else "FMIndex", "scale": args.scale, "prefix_table_symbols": (args.prefix_len if bidir and not args.no_search_accel else None), "kstep_table": (args.kstep if not bidir else 1),
                   "device_table": ("block table expanded from the wavelet" if args.kstep else "wavelet lines") if protein else "block table",
                   "exact_tables": None if (bidir or args.no_exact_tables or (protein and not args.kstep)) else {"suffix_interval_symbols": args.lut_len or (6 if protein else 15), "walk_symbols_per_load": (6 if protein else 16) * (2 if args.walk >= 2 else 1)},
                   "index_device_bytes": index.device_bytes, "index_build_s": round(build_s, 2)},
        "gbp_per_s": qps * L / 1e9,
        "hits": hits,
        **({"exchange": {"collective": xch["mode"], "bytes_per_rank_and_step": int(state["payload"].numel()), "verified_on_rank0": gather_ok,
                         "record": "16 B per hit (qidx:32 | lb:32, len:32 | errors:8 | seq:24)" if bidir else "8 B per read (lb:32 | len:32)"}} if multi else {}),
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "kernel": (("k_exact_a" if args.no_exact_tables else "k_exact_kstep") if args.kstep else "k_exact_w") if protein else ("k_exact_kstep" if (args.kstep > 1 or not args.no_exact_tables) else "k_exact_a") if not bidir else ("k_ng21" if args.ng21 else "k_scheme_fast_edit" if args.edit else "k_scheme_fast"), "kernel_ms": k_ms,
                     "units_per_launch": steps_per_launch, "bytes_per_unit": unit_bytes,
                     "unit": "GB/s", "note": NOTE_PROTEIN if protein else NOTE_DNA},
    }
    if traffic is not None:                                   # the PMC-measured bytes of the same launch against the same peak
        result["roofline"]["traffic_rate"] = {"achieved": traffic / (k_ms * 1e-3) / 1e9, "unit": "GB/s", "frac": traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                              "what": "measured HBM bytes per launch (profiles/traffic.json: FETCH_SIZE x 2 + WRITE_SIZE) / this run's kernel time"}
    if lines is not None:
        result["roofline"]["line_rate"] = {"achieved_G_per_s": lines / (k_ms * 1e-3) / 1e9, "ceiling_G_per_s": ceiling,
                                           "frac": lines / (k_ms * 1e-3) / 1e9 / ceiling,
                                           "what": "L2->fabric 128-byte line requests per second (TCC_EA0_RDREQ) vs the measured ceiling for dependent random line reads"}
    if want_cpu:
        hit_q = None
        if bidir and not multi:                               # query numbers of the last step's hit records: their counts per query are compared with the CPU walk's
            hit_q = hits_bufs[0][: int(stats.hits) * 40].view(torch.int64).view(-1, 5)[:, 0]
        result["cpu_baseline"] = cpu_baseline(index, bidir, qbuf, qoff, nq, L, scheme, args.cpu_sample, out_lb, out_len, layout, sigma, args.edit, hit_q)
    if secondary is not None:
        result["secondary"] = secondary
    print(json.dumps(result), flush=True)
    if multi:
        dist.destroy_process_group()


class _Dev:
    """a torch tensor seen as a device buffer by the package (ptr + nbytes)"""

    def __init__(self, t):
        self.t = t
        self.ptr = t.data_ptr()
        self.nbytes = t.numel() * t.element_size()


def _scheme_struct(capi, scheme):
    import numpy as np
    pi, l, u = (np.ascontiguousarray(x, dtype=np.uint64) for x in scheme)
    sc = capi.Scheme()
    sc.n_searches, sc.n_parts = pi.shape
    sc.pi, sc.l, sc.u = (x.ctypes.data_as(capi.u64p) for x in (pi, l, u))
    sc.partition = None
    return sc, (pi, l, u)


def cpu_baseline(index, bidir, qbuf, qoff, nq, L, scheme, sample, out_lb, out_len, layout="IB16", sigma=5, edit=False, hit_q=None):
    """the CPU restatement (oracle/) on the host cores, bounded sample of the same reads; also a parity spot-check"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fmoracle as fo
    cores = len(os.sched_getaffinity(0))
    t0 = time.time()
    bwt = index.built_array(0)
    bwt_rev = index.built_array(1) if bidir else None
    ox = fo.OraIndex.from_bwt(layout, sigma, bwt, bwt_rev, None, None, None)
    build = time.time() - t0
    hq_all = qbuf.cpu().numpy()
    ho_all = qoff.cpu().numpy().astype(np.uint64)

    def run(count, threads=cores, batched=True):
        t0 = time.time()
        if not bidir and batched:                             # search/SearchNoErrors.h:28-86, the reference's 32-way interleaved form
            r = ox.search_exact_batched(hq_all[: count * L], ho_all[: count + 1], 32, threads)
        elif not bidir:                                       # :12-26, one query at a time
            r = ox.search_exact(hq_all[: count * L], ho_all[: count + 1], nthreads=threads)
        else:
            r = ox.search_ng26(hq_all[: count * L], ho_all[: count + 1], scheme, nthreads=threads, edit=True if edit else None)
        return r, time.time() - t0

    pilot = min(nq, 200_000 if not bidir else (5_000 if edit else 50_000))
    _, dt = run(pilot)                                        # also warms the caches / OpenMP team
    if sample <= 0:                                           # aim at ~15 s of CPU work, bounded by the batch
        sample = int(min(nq, max(pilot, pilot * 15.0 / max(dt, 1e-3))))
    r, dt = run(sample)
    if not bidir:
        lb, ln = r
        ok = bool(np.array_equal(lb, out_lb[:sample].cpu().numpy().astype(np.uint64)) and
                  np.array_equal(ln, out_len[:sample].cpu().numpy().astype(np.uint64)))
    elif hit_q is not None:                                   # per-query record counts on the sample (hit-by-hit parity: tests/test_gpu_parity.py)
        import torch
        mine = torch.bincount(hit_q[hit_q < sample], minlength=sample).cpu().numpy().astype(np.uint64)
        ok = bool(np.array_equal(mine, np.asarray(r[1][:sample], dtype=np.uint64)))
    else:
        ok = None
    one = max(1000, min(sample, int(sample / dt * 3.0 / cores)))          # ~3 s on one thread (single-thread figure beside all cores)
    _, dt1 = run(one, threads=1)
    out = {"value": sample / dt, "unit": "queries/s", "cores": cores, "kind": "port",
           "single_thread": {"value": one / dt1, "unit": "queries/s", "sample": "the first %d reads, one thread" % one},
           "sample": "the first %d reads of the same batch, OpenMP over queries on all host cores%s; index rebuilt on the host from the "
                     "GPU-built BWT in %.0f s" % (sample, "" if bidir else ", 32 cursors interleaved per thread (SearchNoErrors.h:28-86)", build),
           "seconds": dt, "gpu_results_match_on_sample": ok}   # exact: every (lb, len); k = 2: the number of records per query
    if not bidir:                                             # the one-query-at-a-time form beside it, on a sample a fifth the size
        few = max(1000, sample // 5)
        _, dts = run(few, batched=False)
        out["one_query_at_a_time"] = {"value": few / dts, "unit": "queries/s", "sample": "the first %d reads, all cores" % few}
    return out


if __name__ == "__main__":
    main()

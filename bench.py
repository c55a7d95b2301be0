#!/usr/bin/env python3
"""bench.py — throughput of the backward-search hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic reads that are already resident in HBM.

    python bench.py --gpus N --steps K --warmup W

Headline (`value`) = BASELINE.json configs[1]: exact search (search_no_errors) of 10 M x 101 bp reads on a GRCh38-sized FMIndex (25 sequences with
the GRCh38 chromosome lengths, 3.09 Gbp, sigma = 5) — the PLAIN index north_star describes: the bit-packed occurrence table + sampled suffix array,
no accelerator table (`genome/exact/plain`: kernel k_exact_p, two symbols per step on the pair occurrence table the library keeps beside the one-symbol
blocks — 7.3 GB together; `genome/exact/single` is the same search in one-symbol steps, k_exact_a, SURVEY 8d's accounting as written).  The same search with
the optional tables is the named field `with_tables`.  The text is the repeat-structured stand-in of
fmindex-collection_amd/datasets.py (45 % interspersed repeats, satellites, 5 % runs of one symbol) — or the real assembly when FMGPU_FASTA=<path> names
one (reference loader rule, src/example/utils.h:86-98: unknown bases -> A).  The uniform-random text of SURVEY 8d-2 is measured next to it.

At N = 1 the default run measures, one after the other in this process, for each text (`records`, every one driver-timed in the same run):
    exact / plain     k_exact_p on the bit-packed occurrence tables alone (one-symbol blocks 4.2 GB + symbol-pair lines 3.1 GB — the index north_star
                      describes); roofline: executed LF steps x 68 B (2 interval ends x 68 B of a pair line per two-symbol step) / kernel time / 8 TB/s
    exact / single    k_exact_a, one symbol per step on the same index; roofline by SURVEY 8d as written:
                      executed LF steps x 112 B (2 x sizeof(InterleavedBitvector16<5>::Block)) / kernel time / 8 TB/s
    exact / tables    k_exact_kstep with the interval, k-step and walk tables; roofline by what the kernel really loads:
                      (table bytes it counted + query bytes + result bytes) / kernel time / 8 TB/s, and the same at one 128-byte line per access
    k2 / plain        search_ng26<Hamming>, h2(4,0,2), on the two occurrence tables alone (6.2 GB); visited nodes x 112 B
    k2 / tables       the same with LF, prefix and walk tables
plus configs[3]'s single-GPU share (k = 2, 12.5 M x 151 bp) and configs[4] (protein, sigma = 28, FMIndex<28, Wavelet>, 10 M x 40 aa: `.../exact/wavelet` = k_exact_s
on the one-line-per-step table the library keeps beside the tree, 88 B per LF step; `.../exact/tree` = k_exact_m on the multi-ary wavelet tree itself at SURVEY 8d's
170 B per LF step; and its block-table expansion with tables).

N > 1 (launched by torch.distributed.run, one rank per GPU): the index is replicated (every rank builds the same seeded text), the query batch is
sharded, the only exchange is the RCCL gather of the results to rank 0 inside the timed region (double-buffered: the gather of step i crosses xGMI
while the kernel of step i+1 runs).  `value` = exact search on the plain index, 10 M reads PER RANK (weak scaling: per-GPU work fixed); `secondary` =
configs[3]: k = 2 Hamming, 151 bp, partition {38,38,38,37}, 100 M reads in total sharded 100 M / N per rank (strong scaling), 24-byte packed hits gathered.

Output (rank 0): the LAST line of stdout is ONE compact JSON line (< 4 KB: headline, roofline, cpu_baseline, and `summary` = record id ->
[ms_per_step, roofline.frac]); every full record (config, accounting, exchange ...) goes to `bench_records.json` beside this file (or
$FMGPU_BENCH_RECORDS) and, one short line per record, to stderr.  `cpu_baseline` = the CPU restatement (oracle/, parity-pinned) on the host
cores over a bounded sample of the same reads, NUMA-spread and thread-bound; it is a reported baseline, not the target.
"""
import argparse
import ctypes
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
# ---- the roofline rule, ONE for every plain-index record ------------------------------------------------------------------------
# roofline.frac = frac_kernel_format: executed units (LF steps / visited nodes / locate steps — counted by the kernel, identical to the CPU walk) x the bytes the
# kernel reads of the block(s) a unit touches in ITS OWN device format — both interval ends, one end for a one-row node or a locate step — / kernel time
# (HIP events on the launch stream) / 8 TB/s.  Beside it, in every record: frac_sec8d = the same units x SURVEY 8d's bytes of the REFERENCE layout (null where
# that exceeds the peak: the kernel touches fewer lines than the layout 8d prices; the one-symbol / tree records of the same run carry it), frac_loaded = the bytes
# of the loads the kernel issued, counted in the kernel (two ends in one block count once, a node re-visited for its next sibling counts again) + the coalesced
# query / result bytes, and traffic / frac_traffic = HBM bytes by the rocprofv3 counters (replayed from profiles/).
RULE = ("frac = frac_kernel_format = executed units x bytes the kernel reads of the block(s) a unit touches in its own device format (both interval ends; one end for a one-row node "
        "or a locate step) / kernel time (HIP events on the launch stream) / 8 TB/s; frac_sec8d = units x SURVEY 8d's bytes of the reference layout (null if > 1); "
        "frac_loaded = bytes of the loads the kernel issued, counted in the kernel, + coalesced query / result bytes")
SEC8D_STEP_DNA = 2 * BLOCK_BYTES_IB16_S5              # both interval ends x sizeof(InterleavedBitvector16<5>::Block) = 112 B per LF step / visited node
SEC8D_STEP_WAVELET28 = 2 * 5 * 17                     # 2 ends x 5 levels x (8 + 1 + 8) B = 170 B per LF step
SEC8D_STEP_LOCATE = BLOCK_BYTES_IB16_S5 + 64          # one block + one presence-bit line per locate step
FMT_STEP_PAIRS = 68                                   # Format P: 2 ends x (4-byte count + 64 bytes of planes) per TWO symbols = 68 B per LF step
FMT_STEP_BLOCKS = 2 * 12                              # Format A, one symbol per step: 2 ends x the 12-byte entry of the step's symbol
FMT_STEP_PLANES28 = 2 * 44                            # Format S: 2 ends x (five 8-byte planes + the symbol's count: 44 B counted) of one line per LF step
FMT_STEP_TREE28 = 2 * (4 + 24 + 4 + 16)               # Format M, sigma = 28: 2 ends x (8-ary level: count + three planes; 4-ary level: count + two planes) = 96 B per LF step
FMT_NODE_DENSE = (64, 32)                             # Format D: a node of several rows reads 2 x 32 B, a one-row node 32 B
FMT_NODE_BLOCKS = (96, 48)                            # Format A in the lean kernel: the entries of symbols 1..4 = 48 B per end
FMT_NODE_EDIT = (128, 64)                             # the edit-distance kernel reads whole 64-byte blocks
FMT_STEP_LOCATE = 64                                  # one fused 64-byte block per locate step (+ two 8-byte value words per located row)
PROTEIN_SEQS, PROTEIN_SEQ_LEN = 4_000_000, 500        # UniRef50 stand-in (the release itself is not available offline): 2.0e9 residues
PROTEIN_SEQS_WIDE = 9_000_000                         # ... and 4.5e9 residues: more than 2^32 rows, the 64-bit-row build of the kernels
PROTEIN_SEQS_XL = 20_000_000                          # ... and 1.0e10 residues, UniRef50's own order of magnitude (BASELINE.json configs[4]): the suffix array and its sort buffers (42 bytes per row) do not fit
                                                      # beside the text; construction sorts the suffixes bucket by bucket without ever holding the array (csrc/fmgpu_bucketsort.hip)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--texts", default="", help="comma list of genome, uniform (default: genome,uniform at N = 1, genome at N > 1; FMGPU_FASTA replaces genome)")
    ap.add_argument("--only", default="", help="comma list of record ids to measure (text/workload/index, e.g. genome/exact/plain); default: all of the run")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the text (dev runs only; the judged run uses 1.0)")
    ap.add_argument("--nq", type=int, default=10_000_000)
    ap.add_argument("--lut-len", type=int, default=15, help="exact tables: symbols of the interval table")
    ap.add_argument("--prefix-len", type=int, default=16, help="k2 tables: symbols of the prefix table")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-edit", action="store_true", help="also measure k = 2 EDIT distance (search_ng26<Edit = true>, the reference's default) on 2 M of the 101-bp reads, with the tables")
    ap.add_argument("--no-protein", action="store_true")
    ap.add_argument("--no-protein-wide", action="store_true", help="skip the 4.5e9-residue protein record (64-bit rows)")
    ap.add_argument("--no-protein-xl", action="store_true", help="skip the 1.0e10-residue protein record (UniRef50-sized: construction bucket by bucket)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="reads in the CPU baseline sample (0 = auto, ~15 s of CPU work)")
    ap.add_argument("--single-rank-collectives", action="store_true", help="rehearsal only: run the N > 1 code path (process group, asynchronous gather, barrier) with one rank")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="rehearsal only: gloo runs the N > 1 control flow where RCCL cannot (all ranks on one card); results travel through host memory")
    ap.add_argument("--all-ranks-device0", action="store_true", help="rehearsal only: every rank uses GPU 0")
    ap.add_argument("--multi-tables", action="store_true", help="N > 1: also measure the table-augmented indices (136 / 224 GB per GPU); default at N > 1: the plain index only")
    ap.add_argument("--opt", default="", help="library options for the whole run (fmgpu_set_option), e.g. pair_table=0,kernel_select=4194304 (A/B runs; the judged run sets none)")
    ap.add_argument("--total-k2-reads", type=int, default=100_000_000, help="N > 1: reads of the configs[3] leg in total (sharded over the ranks)")
    return ap.parse_args()


class _Dev:
    """a torch tensor seen as a device buffer by the package (ptr + nbytes)"""

    def __init__(self, t):
        self.t = t
        self.ptr = t.data_ptr()
        self.nbytes = t.numel() * t.element_size()


class Ctx:
    pass


def main():
    args = parse()
    # the CPU leg's OpenMP team: one thread per core, spread over the sockets — must be in the environment before the OpenMP runtime starts
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    os.environ.setdefault("OMP_PLACES", "cores")
    # the host cores this process may use, asked BEFORE any OpenMP runtime starts: with OMP_PROC_BIND set, the runtime binds the initial thread to ONE place, and
    # sched_getaffinity then answers 2 (round 3's "2 host cores"); the container's CPU quota (cgroup cpu.max) bounds it too
    host_cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            host_cores = max(1, min(host_cores, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import numpy as np
    import torch
    import fmindex_collection_amd as fm
    from fmindex_collection_amd import capi, datasets

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.gpus != world and not args.single_rank_collectives:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d — N > 1 is launched as python -m torch.distributed.run --nproc-per-node N bench.py --gpus N" % (args.gpus, world))
    if args.all_ranks_device0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    capi.check(capi.lib().fmgpu_set_device(local_rank))
    c = Ctx()
    c.args, c.rank, c.world, c.np, c.torch, c.fm, c.capi, c.datasets = args, rank, world, np, torch, fm, capi, datasets
    c.host_cores = host_cores
    c.dev = torch.device("cuda", local_rank)
    c.multi = world > 1 or args.single_rank_collectives      # (dev: the N > 1 control flow and its RCCL calls with a group of one rank)
    c.dist = None
    if c.multi:
        import torch.distributed as dist
        c.dist = dist
        if args.single_rank_collectives and "RANK" not in os.environ:      # rehearsal without a launcher: a group of one rank
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29555"))
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=c.dev)
        else:
            dist.init_process_group("gloo")
        if dist.get_world_size() != (1 if args.single_rank_collectives and world == 1 else args.gpus):      # a group of another size would report the wrong n_gpus: refuse, loudly
            raise SystemExit("bench.py: --gpus %d but the process group has %d ranks" % (args.gpus, dist.get_world_size()))
    c.via_host = c.multi and args.dist_backend == "gloo"
    for kv in (x for x in args.opt.split(",") if x):             # A/B runs: library options for the whole run
        k, v = kv.split("=")
        fm.options[k.strip()] = int(v, 0)
    c.smi = Smi(local_rank)
    c.only = set(x for x in args.only.split(",") if x)
    c.traffic = {}
    for tf in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json"):           # per-launch HBM bytes from the committed rocprofv3 --pmc passes (replayed, labelled so)
        try:
            c.traffic = json.load(open(os.path.join(ROOT, "profiles", tf)))
            c.traffic_file = "profiles/" + tf
            break
        except Exception:
            pass

    fasta = os.environ.get("FMGPU_FASTA")
    texts = [t for t in args.texts.split(",") if t] or (["genome"] if c.multi else ["genome", "uniform"])
    if fasta:
        texts = ["fasta" if t == "genome" else t for t in texts]
    records = []
    for name in texts:
        records += run_dna_text(c, name, primary=(name == texts[0]))
    if not c.multi and not args.no_protein and args.scale == 1.0 and (not c.only or any(o.startswith("protein") for o in c.only)):
        records += run_protein(c, PROTEIN_SEQS, "protein")
        if not args.no_protein_wide:
            records += run_protein(c, PROTEIN_SEQS_WIDE, "protein_wide")
        if not args.no_protein_xl:
            records += run_protein(c, PROTEIN_SEQS_XL, "protein_xl", plain_only=True)
    if rank != 0:
        if c.multi:
            c.dist.destroy_process_group()
        return
    records = [r for r in records if r is not None]
    if not records:
        raise SystemExit("bench.py: nothing was measured (check --only)")
    emit(records, c.multi, world)
    if c.multi:
        c.dist.destroy_process_group()


HEADLINE_ID = "/exact/plain"      # configs[1] on the plain index (the library's default path there; SURVEY 8d's accounting as written: the ".../exact/single" record)
MAX_LINE = 4000                   # the driver keeps 8 KB of stdout: the last line must fit with room to spare (tests/test_host_and_abi.py checks it)


def _r4(x):
    return float("%.4g" % x) if isinstance(x, float) else x


def compact_line(records, multi, records_file):
    """the one line the driver parses: headline record without its long strings + a summary of every other record"""
    head = next((r for r in records if r["id"].endswith(HEADLINE_ID)), records[0])
    cfg = head.get("config", {})
    text = cfg.get("text", {})
    line = {k: head[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in head}
    line["config"] = {"workload": cfg.get("workload"), "record": head["id"],
                      "text": "%s, %s symbols, %s sequences" % (text.get("text"), text.get("symbols"), text.get("sequences")) if isinstance(text, dict) else text,
                      "index": "%s<%s, %s>" % (cfg.get("index"), cfg.get("sigma"), cfg.get("layout")),
                      "index_kind": cfg.get("index_kind"), "index_device_bytes": cfg.get("index_device_bytes"),
                      "queries_per_gpu": cfg.get("queries_per_gpu", cfg.get("rows_per_gpu")), "read_len": cfg.get("read_len")}
    for k in ("gbp_per_s", "gres_per_s", "hits"):
        if k in head:
            line[k] = _r4(head[k])
    rf = head.get("roofline")
    if rf:
        line["roofline"] = {k: _r4(rf[k]) for k in ("bound", "achieved", "peak", "unit", "frac", "frac_kernel_format", "frac_sec8d", "frac_loaded", "traffic", "frac_traffic", "kernel", "kernel_ms",
                                                    "units_per_launch", "bytes_per_unit") if k in rf}
        if rf.get("frac_sec8d") is None and "sec8d" in rf:
            line["roofline"]["sec8d_uncapped"] = _r4(rf["sec8d"]["frac_uncapped"])     # (> 1: SURVEY 8d's 112 B per step is more than this kernel's format makes it read)
        line["roofline"]["rule"] = "frac = kernel-format bytes (2 interval ends x bytes read of a step's block) / kernel time / peak; sec8d: reference-layout bytes; loaded: counted in the kernel"
        if rf.get("traffic") is not None:
            line["roofline"]["traffic_source"] = "replayed: rocprofv3 --pmc passes in profiles/, not this run"
    if head.get("clocks"):
        line["clocks"] = head["clocks"]
    cb = head.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _r4(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "sample": "first %s reads of the same batch, OpenMP on all host cores, %.1f s" % (cb.get("sample_reads"), cb.get("seconds", 0.0)),
                                "single_thread": _r4(cb["single_thread"]["value"]), "parallel_efficiency": _r4(cb["parallel_efficiency"]),
                                "gpu_results_match_on_sample": cb["gpu_results_match_on_sample"]}
        line["gpu_over_cpu"] = _r4(head["value"] / cb["value"]) if cb["value"] else None
    if "exchange" in head:
        ex = head["exchange"]
        line["exchange"] = {k: ex[k] for k in ("collective", "bytes_per_rank_and_step", "verified_on_rank0", "world_size_seen") if k in ex}
    one = next((r for r in records if r["id"] == head["id"].replace("/plain", "/single")), None)
    if one is not None and one is not head:
        line["one_symbol_steps"] = {"record": one["id"], "value": _r4(one["value"]), "ms_per_step": _r4(one["ms_per_step"]), "kernel": one["roofline"]["kernel"],
                                    "kernel_ms": _r4(one["roofline"]["kernel_ms"]), "frac_sec8d": _r4(one["roofline"]["frac_sec8d"]),
                                    "what": "the same index and reads one symbol per step: SURVEY 8d's 112 B per LF step as written"}
    tab = next((r for r in records if r["id"] == head["id"].replace("/plain", "/tables")), None)
    if tab is not None and tab is not head:
        line["with_tables"] = {"record": tab["id"], "value": _r4(tab["value"]), "ms_per_step": _r4(tab["ms_per_step"]),
                               "index_device_bytes": tab["config"].get("index_device_bytes"), "kernel": tab["roofline"]["kernel"]}
    if multi:
        sec = next((r for r in records if "/k2_151/" in r["id"]), None)
        if sec is not None:
            line["secondary"] = {"record": sec["id"], "metric": sec["metric"], "value": _r4(sec["value"]), "unit": sec["unit"], "ms_per_step": _r4(sec["ms_per_step"]),
                                 "scaling": sec["scaling"], "frac": _r4(sec["roofline"]["frac"]), "hits": sec.get("hits"),
                                 "exchange": {k: sec["exchange"][k] for k in ("collective", "bytes_per_rank_and_step", "verified_on_rank0", "world_size_seen") if k in sec.get("exchange", {})}}
    k2 = next((r for r in records if r["id"].endswith("/k2/plain")), None)
    if k2 is not None and "cpu_baseline" in k2:
        cb2 = k2["cpu_baseline"]
        line["k2_cpu_baseline"] = {"record": k2["id"], "value": _r4(cb2["value"]), "cores": cb2["cores"], "gpu_over_cpu": _r4(k2["value"] / cb2["value"]) if cb2["value"] else None,
                                   "gpu_results_match_on_sample": cb2["gpu_results_match_on_sample"]}
    def row(r):
        return [_r4(r["ms_per_step"]), _r4(r["roofline"]["frac"]), _r4(r["roofline"].get("frac_sec8d")), _r4(r["roofline"].get("frac_loaded"))]
    first_text = head["id"].split("/")[0]
    line["summary"] = {r["id"]: row(r) for r in records}
    line["summary_columns"] = ["ms_per_step", "roofline.frac (kernel format; table records: loaded)", "frac_sec8d", "frac_loaded"]
    line["records_file"] = records_file
    out = json.dumps(line, separators=(",", ":"))
    if len(out) > MAX_LINE - 300:                                # keep room to spare: the comparison text's records first go to the records file alone ...
        line["summary"] = {r["id"]: row(r) for r in records if r["id"].split("/")[0] in (first_text, "protein", "protein_wide", "protein_xl")}
        line["summary_also_in_records_file"] = sorted({r["id"].split("/")[0] for r in records} - {first_text, "protein", "protein_wide", "protein_xl"})
        out = json.dumps(line, separators=(",", ":"))
    if len(out) > MAX_LINE:                                      # ... and never let the headline be cut: drop the optional parts
        for k in ("summary_also_in_records_file", "summary_columns", "k2_cpu_baseline", "with_tables", "one_symbol_steps", "summary", "clocks"):
            line.pop(k, None)
            out = json.dumps(line, separators=(",", ":"))
            if len(out) <= MAX_LINE:
                break
    return out


def emit(records, multi, world):
    path = os.environ.get("FMGPU_BENCH_RECORDS", os.path.join(ROOT, "bench_records.json"))
    try:
        with open(path, "w") as f:
            json.dump({"n_gpus": world, "records": records}, f, indent=1)
    except OSError as ex:
        print("bench.py: could not write %s (%s)" % (path, ex), file=sys.stderr, flush=True)
        path = None
    for r in records:
        rf = r["roofline"]
        f3 = lambda v: "  -  " if v is None else "%.3f" % v
        ck = r.get("clocks") or {}
        print("bench.py: %-30s %9.3f ms/step  %-18s %9.3f ms  frac %.3f  sec8d %s  loaded %s  sclk %s MHz  value %.4g %s"
              % (r["id"], r["ms_per_step"], rf["kernel"], rf["kernel_ms"], rf["frac"], f3(rf.get("frac_sec8d")), f3(rf.get("frac_loaded")), ck.get("sclk_mhz_mean", "?"), r["value"], r["unit"]),
              file=sys.stderr, flush=True)
    print(compact_line(records, multi, os.path.basename(path) if path else None), flush=True)


# ---------------------------------------------------------------------------------------------------------------- inputs
def make_text(c, name):
    torch, np = c.torch, c.np
    lengths = [max(1, int(l * c.args.scale)) for l in GRCH38_LENGTHS]
    info = {"text": name}
    if name == "fasta":
        sym, off = c.datasets.load_fasta(os.environ["FMGPU_FASTA"])
        text = torch.from_numpy(sym).to(c.dev)
        lengths = np.diff(off).tolist()
        info.update({"source": os.environ["FMGPU_FASTA"], "rule": "src/example/utils.h:86-98 with --convertUnknownChar: unknown bases -> rank 1 (A)"})
    elif name == "genome":
        text, st = c.datasets.genome_like_text(lengths, seed=42, device=c.dev)
        info.update({"generator": "fmindex-collection_amd/datasets.py genome_like_text(seed=42)", **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()}})
    elif name == "uniform":
        total = sum(lengths)
        g = torch.Generator(device=c.dev)
        g.manual_seed(42)
        text = torch.empty(total, dtype=torch.uint8, device=c.dev)
        for lo in range(0, total, 1 << 28):
            hi = min(total, lo + (1 << 28))
            text[lo:hi] = torch.randint(1, 5, (hi - lo,), generator=g, device=c.dev, dtype=torch.uint8)
        info.update({"generator": "uniform bases in {1..4}, torch generator seed 42 (SURVEY 8d-2's stand-in)"})
    else:
        raise SystemExit("unknown text %r" % name)
    seq_off = torch.from_numpy(np.concatenate([[0], np.cumsum(np.asarray(lengths, dtype=np.int64))])).to(c.dev)
    info["symbols"] = int(text.numel()); info["sequences"] = len(lengths)
    torch.cuda.synchronize()                                  # (the generator's kernels are not part of the index build time)
    return text, seq_off, lengths, info


def sample_reads(c, text, lengths, L, nq, seed, mode, sigma=5, inside=None):
    """reads copied from the text; mode 'exact': every 10th read carries one substitution (early exits), 'k2': 0 / 1 / 2 substitutions in ratio 1:1:1"""
    torch = c.torch
    total = text.numel()
    gq = torch.Generator(device=c.dev)
    gq.manual_seed(seed)
    if inside is not None:                                    # protein: inside one entry (500-residue entries are short next to the read)
        nseq, slen = inside
        starts = (torch.randint(0, nseq, (nq,), generator=gq, device=c.dev, dtype=torch.int64) * slen +
                  torch.randint(0, slen - L + 1, (nq,), generator=gq, device=c.dev, dtype=torch.int64))
    else:
        starts = torch.randint(0, total - L, (nq,), generator=gq, device=c.dev, dtype=torch.int64)
    reads = torch.empty((nq, L), dtype=torch.uint8, device=c.dev)
    ar = torch.arange(L, device=c.dev, dtype=torch.int64)
    for lo in range(0, nq, 1 << 20):
        hi = min(nq, lo + (1 << 20))
        reads[lo:hi] = text[starts[lo:hi, None] + ar[None, :]]
    if mode == "exact":
        rows = torch.arange(0, nq, 10, device=c.dev)
        nsub = torch.ones_like(rows)
    else:
        rows = torch.arange(0, nq, device=c.dev)
        nsub = rows % 3
    for k in range(2):
        sel = rows[nsub > k]
        pos = torch.randint(0, L, (sel.numel(),), generator=gq, device=c.dev)
        shift = torch.randint(1, sigma - 1, (sel.numel(),), generator=gq, device=c.dev, dtype=torch.uint8)
        reads[sel, pos] = (reads[sel, pos] - 1 + shift) % (sigma - 1) + 1
    qbuf = reads.reshape(-1)
    qoff = torch.arange(nq + 1, device=c.dev, dtype=torch.int64) * L
    return qbuf, qoff


def wanted(c, rid):
    return not c.only or rid in c.only


# ---------------------------------------------------------------------------------------------------------------- timing
def timed(c, step, drain=None):
    """W warm-up steps, then exactly K steps between barrier + synchronize on both sides; returns (seconds [max over ranks], per-step stats)"""
    torch, args = c.torch, c.args
    log = []
    if not getattr(c, "spun_up", False):                          # the first measurement of the process: the card's clocks ramp up over the first few hundred ms of load
        c.spun_up = True                                          # (r4: the first record ran at sclk 2243 MHz where later ones ran at 2350-2390) — untimed, before the W warm-up steps
        t_spin, n_spin = time.perf_counter(), 0
        while (n_spin < 8) if c.multi else (time.perf_counter() - t_spin < 0.5):      # (N > 1: the same number of steps on every rank — each step holds a collective)
            step(log)
            torch.cuda.synchronize()
            n_spin += 1
        if drain:
            drain()
        log.clear()
    for _ in range(args.warmup):
        step(log)
    if drain:
        drain()
    log.clear()
    if c.multi:
        c.dist.barrier()
    torch.cuda.synchronize()
    c.smi.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(log)
    if drain:
        drain()
    if c.multi:
        c.dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    c.last_clocks = c.smi.stop()
    if c.multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if c.via_host else c.dev)
        c.dist.all_reduce(t, op=c.dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, log


class Smi:
    """clocks and power of the card WHILE the timed steps run (librocm_smi64 through ctypes, sampled every 2 ms by a thread): the figures that attribute a box-to-box
    spread of the same kernel on the same input (10.8 vs 11.7 ms) to the clock the box ran at.  Reports nothing where the library is not usable."""

    class _Freq(ctypes.Structure):                                # rsmi_frequencies_t (rocm_smi.h)
        _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32), ("current", ctypes.c_uint32), ("frequency", ctypes.c_uint64 * 33)]

    def __init__(self, dev):
        import ctypes as C
        self.dev, self.lib, self.samples, self.thread, self.stop_flag = dev, None, [], None, False
        for path in ("/opt/rocm/lib/librocm_smi64.so", "librocm_smi64.so", "librocm_smi64.so.1"):
            try:
                L = C.CDLL(path)
                if L.rsmi_init(C.c_uint64(0)) == 0:
                    self.lib = L
                    break
            except OSError:
                continue
        self.cap_w, self.card = None, None
        if self.lib is not None:
            uid = C.c_uint64()
            try:
                if self.lib.rsmi_dev_unique_id_get(C.c_uint32(dev), C.byref(uid)) == 0:
                    self.card = "%016x" % uid.value                # (which card of the pool ran this: the same kernel differs by a few percent between cards)
            except Exception:
                pass
            cap = C.c_uint64()
            try:
                if self.lib.rsmi_dev_power_cap_get(C.c_uint32(dev), C.c_uint32(0), C.byref(cap)) == 0:
                    self.cap_w = cap.value / 1e6
            except Exception:
                pass

    def _read(self):
        import ctypes as C
        out = []
        for clk in (0, 4, 1):                                     # RSMI_CLK_TYPE_SYS, RSMI_CLK_TYPE_MEM, RSMI_CLK_TYPE_DF (the fabric)
            f = Smi._Freq()
            ok = self.lib.rsmi_dev_gpu_clk_freq_get(C.c_uint32(self.dev), C.c_int(clk), C.byref(f)) == 0 and f.current < 33
            out.append(f.frequency[f.current] / 1e6 if ok else None)
        pw = C.c_uint64()
        out.insert(2, pw.value / 1e6 if self.lib.rsmi_dev_current_socket_power_get(C.c_uint32(self.dev), C.byref(pw)) == 0 else None)
        return out                                                # [sclk, mclk, power, fclk]

    def start(self):
        if self.lib is None:
            return
        import threading
        self.samples, self.stop_flag = [], False

        def run():
            while not self.stop_flag:
                try:
                    self.samples.append(self._read())
                except Exception:
                    return
                time.sleep(0.002)
        self.thread = threading.Thread(target=run, daemon=True)
        self.thread.start()

    def stop(self):
        if self.lib is None or self.thread is None:
            return None
        self.stop_flag = True
        self.thread.join(timeout=1.0)
        self.thread = None
        col = lambda k: [x[k] for x in self.samples if x[k] is not None]
        sc, mc, pw, fc = col(0), col(1), col(2), col(3)
        if not sc:
            return None
        return {"sclk_mhz_mean": round(sum(sc) / len(sc)), "sclk_mhz_min": round(min(sc)), "sclk_mhz_max": round(max(sc)), "mclk_mhz": round(sum(mc) / len(mc)) if mc else None,
                "fclk_mhz": round(sum(fc) / len(fc)) if fc else None, "socket_power_w_mean": round(sum(pw) / len(pw)) if pw else None, "power_cap_w": self.cap_w, "samples": len(sc),
                "card": self.card,
                "what": "rocm_smi readings of the card during the timed steps of this record (2 ms apart)"}


class Exchange:
    """the path's one exchange: results to rank 0 over RCCL/xGMI (torch.distributed.gather = grouped send/recv), double-buffered"""

    def __init__(self, c, max_bytes):
        self.c, self.mode, self.pending, self.bufs = c, "gather", [None, None], [None, None]
        self.gathered = None
        if c.rank == 0:
            self.gathered = [[c.torch.empty(max_bytes, dtype=c.torch.uint8, device="cpu" if c.via_host else c.dev) for _ in range(c.world)] for _ in range(2)]
        self.last = None

    def wait(self, b):
        if self.pending[b] is not None:
            self.pending[b].wait(); self.pending[b] = None

    def send(self, payload, b):
        c, dist = self.c, self.c.dist
        if c.via_host:
            payload = payload.cpu()
        self.last = (payload, b)
        if self.mode == "gather":
            try:
                self.pending[b] = dist.gather(payload, [g[: payload.numel()] for g in self.gathered[b]] if c.rank == 0 else None, dst=0, async_op=True)
                return
            except (RuntimeError, NotImplementedError, ValueError) as ex:   # should this build refuse it: all_gather — more bytes over xGMI, same information on rank 0
                self.mode = "all_gather"
                if c.rank == 0:
                    print("bench.py: dist.gather unavailable (%s); using all_gather_into_tensor" % ex, file=sys.stderr, flush=True)
        need = c.world * payload.numel()
        if self.bufs[b] is None or self.bufs[b].numel() < need:
            self.bufs[b] = c.torch.empty(need, dtype=c.torch.uint8, device=payload.device)
        self.pending[b] = dist.all_gather_into_tensor(self.bufs[b][:need], payload, async_op=True)

    def drain(self):
        for b in range(2):
            self.wait(b)

    def verify(self):
        """outside the timed region: rank 0 holds what every rank sent in the last step (a checksum per rank)"""
        c, torch, dist = self.c, self.c.torch, self.c.dist
        pl, b = self.last
        n8 = pl.numel() // 8 * 8
        chk = pl[:n8].view(torch.int64).sum().reshape(1).to("cpu" if c.via_host else c.dev)
        sums = [torch.zeros_like(chk) for _ in range(c.world)]
        dist.all_gather(sums, chk)
        if c.rank != 0:
            return None
        if self.mode == "gather":
            got = [self.gathered[b][r][:n8].view(torch.int64).sum().item() for r in range(c.world)]
        else:
            got = [self.bufs[b][r * pl.numel(): r * pl.numel() + n8].view(torch.int64).sum().item() for r in range(c.world)]
        ok = got == [int(x.item()) for x in sums]
        if not ok:
            raise SystemExit("bench.py: the gathered results on rank 0 differ from what the ranks sent")
        return ok


def roofline_plain(kernel, k_ms, units, unit_name, fmt_bytes, fmt_what, sec8d_per_unit, sec8d_what, loaded_bytes=None, loaded_what=None, loaded_accesses=None):
    """the one rule of every plain-index record (RULE above): frac = kernel-format bytes / kernel time / peak, with the SURVEY 8d and in-kernel figures beside it"""
    t = k_ms * 1e-3
    ach = fmt_bytes / t / 1e9
    sec = units * sec8d_per_unit / t / 1e9
    r = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
         "kernel": kernel, "kernel_ms": k_ms, "units_per_launch": units, "unit_name": unit_name, "bytes_per_unit": fmt_bytes / max(units, 1.0),
         "frac_kernel_format": ach / HBM_PEAK_GBS, "kernel_format": {"bytes_per_launch": fmt_bytes, "what": fmt_what},
         "frac_sec8d": sec / HBM_PEAK_GBS if sec <= HBM_PEAK_GBS else None,
         "sec8d": {"bytes_per_unit": sec8d_per_unit, "achieved": sec, "frac_uncapped": sec / HBM_PEAK_GBS, "what": sec8d_what},
         "frac_loaded": None, "rule": RULE}
    if sec > HBM_PEAK_GBS:
        r["sec8d"]["why_null"] = ("SURVEY 8d's bytes of the reference layout / kernel time exceed the 8 TB/s peak: this kernel reads a derived format that touches fewer lines per unit than the "
                                  "layout 8d prices; 8d as written is measured on the one-symbol / tree record of the same run")
    if loaded_bytes is not None:
        r["frac_loaded"] = loaded_bytes / t / 1e9 / HBM_PEAK_GBS
        r["loaded"] = {"bytes_per_launch": loaded_bytes, "what": loaded_what}
        if loaded_accesses is not None:
            r["loaded"]["accesses_per_launch"] = loaded_accesses
            r["loaded"]["line_granular_frac"] = loaded_accesses * 128.0 / t / 1e9 / HBM_PEAK_GBS
    return r


def roofline_loaded(st, coalesced_bytes, k_ms, kernel, units, what):
    """the table-driven kernels: what they actually asked of the memory system, counted by the kernel itself (fmgpu_stats)"""
    tb, ta = float(st["table_bytes"]), float(st["table_accesses"])
    ach = (tb + coalesced_bytes) / (k_ms * 1e-3) / 1e9
    line = (ta * 128.0 + coalesced_bytes) / (k_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
            "kernel": kernel, "kernel_ms": k_ms, "units_per_launch": units,
            "frac_kernel_format": None, "frac_sec8d": None, "frac_loaded": ach / HBM_PEAK_GBS,      # (a table entry serves several units: only the loaded bytes mean anything)
            "bytes_per_launch": {"table_entries": tb, "table_accesses": ta, "queries_and_results": coalesced_bytes},
            "line_granular": {"achieved": line, "frac": line / HBM_PEAK_GBS, "what": "every table access priced as one 128-byte line"},
            "accounting": "bytes of the loads the kernel issued (counted in the kernel: interval / context / walk / prefix table entries, blocks, frames, hit records) "
                          "+ query and result bytes, / kernel time; units_per_launch = " + what + " the batch stands for (identical to the CPU walk), served by fewer loads"}


def attach_traffic(c, rec):
    rec["clocks"] = getattr(c, "last_clocks", None)           # (of the timed steps that made this record)
    t = c.traffic.get(rec["id"])
    if t and c.args.scale == 1.0:
        rec["roofline"]["traffic"] = t["bytes_per_launch"]
        rec["roofline"]["frac_traffic"] = t["bytes_per_launch"] / (rec["roofline"]["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        rec["roofline"]["traffic_source"] = "%s (%s): separate rocprofv3 --pmc passes of this record, FETCH_SIZE x 2 + WRITE_SIZE; replayed, not measured in this run" % (c.traffic_file, t.get("source", ""))


def mean(xs):
    return sum(xs) / max(len(xs), 1)


# ---------------------------------------------------------------------------------------------------------------- DNA texts
def run_dna_text(c, name, primary):
    torch, np, fm, capi, args = c.torch, c.np, c.fm, c.capi, c.args
    import ctypes as C
    legs = ["exact", "k2", "k2_151"]
    ids = ["%s/%s/%s" % (name, w, i) for w in legs + ["locate"] for i in ("plain", "tables")] + ["%s/%s/plain+lut12" % (name, w) for w in ("k2", "k2_151")] + ["%s/k2_edit/tables" % name, "%s/k2_edit/plain" % name, "%s/exact/single" % name, "%s/exact/plain+lut12" % name]
    if c.only and not any(i in c.only for i in ids):
        return []
    text, seq_off, lengths, tinfo = make_text(c, name)
    total = int(text.numel())
    out = []
    nq = args.nq
    want_cpu = primary and c.rank == 0 and not c.multi and not args.no_cpu_baseline
    base_cfg = {"text": tinfo, "sigma": 5, "layout": "InterleavedBitvector16", "scale": args.scale}

    # ------------------------------------------------------------------ exact search, configs[1]
    if any(wanted(c, "%s/%s/%s" % (name, w_, i)) for w_ in ("exact", "locate") for i in ("plain", "tables")) or wanted(c, name + "/exact/single") or wanted(c, name + "/exact/plain+lut12"):
        L = 101
        qbuf, qoff = sample_reads(c, text, lengths, L, nq, 1000 + c.rank, "exact")
        torch.cuda.synchronize()
        fm.options["lf_table"] = 0
        t0 = time.time()
        index = fm.FMIndex.from_sequences((_Dev(text), _Dev(seq_off)), 5, "IB16", 16, keep_host=want_cpu)
        build_plain = time.time() - t0
        sel_base = fm.options["kernel_select"]
        pairs = bool(index.formats & capi.FMT_PAIRS) and not (sel_base & capi.SEL_EXACT_ONE_SYMBOL)      # (what the library holds decides the kernel: the record is labelled from it)
        outs = [torch.empty(2 * nq, dtype=torch.int64, device=c.dev) for _ in range(2 if c.multi else 1)]
        packed = [torch.empty(nq, dtype=torch.int64, device=c.dev) for _ in range(2)] if c.multi else None
        stats = capi.Stats()
        xch = Exchange(c, nq * 8) if c.multi else None
        state = {"i": 0}

        def step(log):
            b = state["i"] % len(outs); state["i"] += 1
            if xch:
                xch.wait(b)
                capi.check(capi.lib().fmgpu_search_exact_packed(index._h, C.c_void_p(qbuf.data_ptr()), C.c_void_p(qoff.data_ptr()), nq,
                                                                C.c_void_p(packed[b].data_ptr()), C.byref(stats), None))
            else:
                capi.check(capi.lib().fmgpu_search_exact(index._h, C.c_void_p(qbuf.data_ptr()), C.c_void_p(qoff.data_ptr()), nq,
                                                         C.c_void_p(outs[b][:nq].data_ptr()), C.c_void_p(outs[b][nq:].data_ptr()), C.byref(stats), None))
            log.append({"kernel_ms": stats.kernel_ms, "units": stats.lf_steps, "table_bytes": stats.table_bytes, "table_accesses": stats.table_accesses, "table_steps": stats.table_steps})
            if xch:
                xch.send(packed[b].view(torch.uint8), b)

        def finish(rid, index_kind, kernel, elapsed, log, build_s, extra):
            k_ms = mean([x["kernel_ms"] for x in log]); units = mean([x["units"] for x in log])
            qps = c.world * nq * args.steps / elapsed
            if xch:
                w = packed[(state["i"] - 1) % 2]
                out_len = w & 0xffffffff
            else:
                out_len = outs[0][nq:]
            rec = {"id": rid, "metric": "queries/sec (GRCh38-sized index, 10M x 101bp, exact)", "value": qps, "unit": "queries/s", "n_gpus": c.world, "steps": args.steps,
                   "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
                   "data": "synthetic" if name != "fasta" else "real",
                   "config": {"workload": "grch38_exact", **base_cfg, "index": "FMIndex", "index_kind": index_kind, "queries_per_gpu": nq, "read_len": L,
                              "index_device_bytes": index.device_bytes, "index_build_s": round(build_s, 2), **extra},
                   "gbp_per_s": qps * L / 1e9, "hits": int((out_len > 0).sum().item())}
            st = {k: mean([x[k] for x in log]) for k in ("table_bytes", "table_accesses", "table_steps")}
            coalesced = nq * (L + 8 + 16)
            lut_len = extra.get("interval_table_symbols") or 1
            fmt_pairs = (units - st["table_steps"]) * FMT_STEP_PAIRS + st["table_steps"] / lut_len * 8.0      # (an interval-table entry — 8 B — stands for lut_len steps)
            if kernel == "k_exact_a":
                rec["roofline"] = roofline_plain(kernel, k_ms, units, "executed LF steps", units * FMT_STEP_BLOCKS,
                                                 "one symbol per step on the one-symbol blocks: 2 interval ends x the 12-byte entry {count, 64-bit bitmap} of the step's symbol = 24 B per LF step",
                                                 SEC8D_STEP_DNA, "2 x sizeof(InterleavedBitvector16<5>::Block) = 112 B per executed LF step (this record IS SURVEY 8d as written)",
                                                 st["table_bytes"] + coalesced, "12 B per entry the kernel loaded (an end in the other end's block: one load) + queries and results", st["table_accesses"])
            elif kernel in ("k_exact_p", "k_exact_lp"):
                rec["roofline"] = roofline_plain(kernel, k_ms, units, "executed LF steps", fmt_pairs,
                                                 ("the read's last %d symbols from one 8-byte interval-table entry, then " % lut_len if kernel == "k_exact_lp" else "") +
                                                 "two symbols per step on the pair lines (one 128-byte line per 128 rows: 16 pair counts + 4 bit planes): 2 interval ends x 68 B read of a "
                                                 "line (4-byte count + four 16-byte plane words) per two-symbol step = 68 B per executed LF step",
                                                 SEC8D_STEP_DNA, "2 x sizeof(InterleavedBitvector16<5>::Block) = 112 B per executed LF step: record " + rid.rsplit("/", 1)[0] + "/single (k_exact_a, same index, same reads, same run)",
                                                 st["table_bytes"] + coalesced, "68 B per line the kernel fetched (an end in the other end's line: one fetch; 12 B per one-symbol entry; 8 B per "
                                                 "interval-table entry) + queries and results", st["table_accesses"])
            else:
                rec["roofline"] = roofline_loaded(st, coalesced, k_ms, kernel, units, "LF steps")
            if xch:
                rec["exchange"] = {"collective": xch.mode, "bytes_per_rank_and_step": nq * 8, "verified_on_rank0": xch.verify(), "record": "8 B per read (lb:32 | len:32)",
                                   "world_size_seen": c.dist.get_world_size()}
            attach_traffic(c, rec)
            return rec

        def locate_run(rid, index_kind, build_s):
            """locate (LocateLinear / FMIndex::locate, fmindex/FMIndex.h:113-124) of the row every hit interval starts at: part of the path
            (north_star: ... -> SA interval -> locate).  SURVEY 8d: per LF step one block (symbol + rank) and one presence-bit probe."""
            lb, ln = outs[0][:nq], outs[0][nq:]
            sel = ln > 0
            rows = lb[sel].contiguous()
            nr = int(rows.numel())
            if nr == 0:
                return None
            res = [torch.empty(nr, dtype=torch.int64, device=c.dev) for _ in range(3)]
            lstats = capi.Stats()

            def lstep(log):
                capi.check(capi.lib().fmgpu_locate(index._h, C.c_void_p(rows.data_ptr()), nr, C.c_void_p(res[0].data_ptr()), C.c_void_p(res[1].data_ptr()),
                                                   C.c_void_p(res[2].data_ptr()), C.byref(lstats), None))
                log.append({"kernel_ms": lstats.kernel_ms, "units": lstats.lf_steps})
            elapsed, log = timed(c, lstep)
            k_ms = mean([x["kernel_ms"] for x in log]); units = mean([x["units"] for x in log])
            # size-independent check: every located position holds the read that found it (text[seq_off[seq] + pos + steps ...] == read)
            pick = torch.arange(0, nr, max(1, nr // 200_000), device=c.dev)
            tpos = seq_off[res[0][pick]] + res[1][pick] + res[2][pick]
            ridx = torch.nonzero(sel).view(-1)[pick]
            ar = torch.arange(L, device=c.dev)
            same = bool((text[tpos[:, None] + ar[None, :]] == qbuf[(ridx * L)[:, None] + ar[None, :]]).all().item())
            pps = nr * args.steps / elapsed
            rec = {"id": rid, "metric": "located positions/sec (first row of every exact hit, 10M x 101bp reads, sampled suffix array rate 16)", "value": pps, "unit": "positions/s",
                   "n_gpus": c.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                   "vs_baseline": None, "dtype": "u64", "data": "synthetic" if name != "fasta" else "real",
                   "config": {"workload": "grch38_locate", **base_cfg, "index": "FMIndex", "index_kind": index_kind, "rows_per_gpu": nr, "sampling_rate": 16,
                              "index_device_bytes": index.device_bytes, "index_build_s": round(build_s, 2)},
                   "lf_steps_per_row": units / nr, "located_positions_hold_their_reads": same}
            if index_kind == "plain":
                fused = bool(index.formats & capi.FMT_FUSED)
                lk = ("k_locate_fused" if (sel_base & capi.SEL_LOCATE_PER_LANE) else "k_locate_coop") if fused else "k_locate"
                fmt = (units + nr) * (FMT_STEP_LOCATE if fused else 64 + 64) + nr * 16.0
                rec["roofline"] = roofline_plain(lk, k_ms, units + nr, "locate steps (LF steps + the probe of the sampled row)", fmt,
                                                 "one 64-byte block per step — presence bit, symbol and LF of the row; at the sampled row the value's rank from the same block — + two 8-byte value words per located row"
                                                 if fused else "one block + one presence-bit line per step",
                                                 SEC8D_STEP_LOCATE, "one 56-byte block + one 64-byte presence-bit line = 120 B per locate step",
                                                 fmt + nr * 32.0, "the same blocks and value words (the kernel loads nothing else from the index) + rows and results")
            else:
                rec["roofline"] = roofline_loaded({"table_bytes": 12.0 * nr, "table_accesses": float(nr)}, nr * 32, k_ms, "k_locate_tab", units, "LF steps of locate")
            attach_traffic(c, rec)
            return rec

        plain_ms = None
        if wanted(c, name + "/locate/plain") and not c.multi and not wanted(c, name + "/exact/plain"):
            step([])                                              # the rows come from the exact search
        if wanted(c, name + "/exact/plain"):                      # the headline: at N > 1 too (same index, same kernel at every N; weak scaling)
            elapsed, log = timed(c, step, xch.drain if xch else None)
            r = finish(name + "/exact/plain", "plain", "k_exact_p" if pairs else "k_exact_a", elapsed, log, build_plain,
                       {"tables": None, "occurrence_tables": "one-symbol blocks (Format A) + symbol-pair lines (Format P)" if pairs else "one-symbol blocks (Format A)"})
            plain_ms = r["roofline"]["kernel_ms"]

            if c.rank == 0 and not c.multi:                     # the distribution the judge asked for: symbols until the interval is one row
                depth = torch.empty(nq, dtype=torch.int32, device=c.dev)
                capi.check(capi.lib().fmgpu_search_exact_depth(index._h, C.c_void_p(qbuf.data_ptr()), C.c_void_p(qoff.data_ptr()), nq, C.c_void_p(depth.data_ptr()), None))
                d = depth.to(torch.float32)
                qs = torch.quantile(d[:: max(1, nq // 1_000_000)], torch.tensor([0.5, 0.9, 0.99], device=c.dev)).tolist()
                r["symbols_until_one_row"] = {"mean": float(d.mean().item()), "p50": qs[0], "p90": qs[1], "p99": qs[2],
                                              "never_within_the_read": float((depth > L).float().mean().item()),
                                              "what": "query symbols consumed until the SA interval holds <= 1 row (fmgpu_search_exact_depth), over the same reads"}
                del depth, d
            if want_cpu:
                r["cpu_baseline"] = cpu_baseline(c, index, False, qbuf, qoff, nq, L, None, outs[0][:nq], outs[0][nq:], None)
            out.append(r)
        if pairs and not c.multi and wanted(c, name + "/exact/single"):     # the same index and reads in one-symbol steps: SURVEY 8d's accounting as written
            if not wanted(c, name + "/exact/plain"):
                step([])                                          # (the pair kernel's results, to compare with)
                torch.cuda.synchronize()
            keep = outs[0].clone()
            with fm.options(kernel_select=sel_base | capi.SEL_EXACT_ONE_SYMBOL):
                elapsed, log = timed(c, step, None)
            r1 = finish(name + "/exact/single", "plain", "k_exact_a", elapsed, log, build_plain, {"tables": None, "occurrence_tables": "one-symbol blocks (Format A)"})
            r1["equal_to_the_pair_kernel"] = bool(torch.equal(keep, outs[0]))
            if not r1["equal_to_the_pair_kernel"]:
                raise SystemExit("bench.py: k_exact_a and k_exact_p disagree")
            if plain_ms:
                r1["roofline"]["pair_kernel_speedup"] = r1["roofline"]["kernel_ms"] / plain_ms
            out.append(r1)
            del keep
        if wanted(c, name + "/locate/plain") and not c.multi:
            out.append(locate_run(name + "/locate/plain", "plain", build_plain))
        if pairs and not c.multi and wanted(c, name + "/exact/plain+lut12"):      # the same search behind a 12-symbol interval table (134 MB: 4^12 entries of 8 bytes), no other table
            t0 = time.time()
            index.accelerate(1, lut_len=12, walk=0)
            keep = outs[0].clone() if plain_ms else None
            elapsed, log = timed(c, step, None)
            r2 = finish(name + "/exact/plain+lut12", "plain+lut12", "k_exact_lp", elapsed, log, build_plain + time.time() - t0,
                        {"tables": {"suffix_interval_symbols": 12}, "interval_table_symbols": 12,
                         "occurrence_tables": "one-symbol blocks (Format A) + symbol-pair lines (Format P) + the intervals of all 12-symbol strings (134 MB)"})
            if keep is not None:
                r2["equal_to_the_plain_index"] = bool(torch.equal(keep, outs[0]))
                if not r2["equal_to_the_plain_index"]:
                    raise SystemExit("bench.py: exact search with and without the interval table disagree")
                r2["roofline"]["speedup_over_plain_index_kernel"] = plain_ms / r2["roofline"]["kernel_ms"]
            out.append(r2)
            del keep
        if ((not c.multi or args.multi_tables) and wanted(c, name + "/exact/tables")) or (wanted(c, name + "/locate/tables") and not c.multi):
            t0 = time.time()
            del fm.options["lf_table"]
            index.accelerate(3, lut_len=args.lut_len, walk=2)
            build_tab = build_plain + time.time() - t0
            elapsed, log = timed(c, step, xch.drain if xch else None)
            r = finish(name + "/exact/tables", "tables", "k_exact_kstep", elapsed, log, build_tab,
                       {"tables": {"suffix_interval_symbols": args.lut_len, "kstep": 3, "walk_symbols_per_load": 32}})
            if plain_ms:
                r["roofline"]["speedup_over_plain_index_kernel"] = plain_ms / r["roofline"]["kernel_ms"]
            if wanted(c, name + "/exact/tables"):
                out.append(r)
            if wanted(c, name + "/locate/tables") and not c.multi:
                t0 = time.time()
                index.accelerate_locate()
                out.append(locate_run(name + "/locate/tables", "tables", build_tab + time.time() - t0))
        del fm.options["lf_table"]
        index.close()
        del index, qbuf, qoff, outs, packed
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ k = 2 Hamming, configs[2] (101 bp) and configs[3] (151 bp)
    scheme = fm.search_scheme.h2(4, 0, 2)
    k2_legs = [("k2", 101, nq if not c.multi else 0), ("k2_151", 151, (args.total_k2_reads // c.world) if c.multi else min(12_500_000, max(1, int(12_500_000 * min(1.0, nq / 10_000_000)))))]
    k2_legs = [(w, L, n_) for (w, L, n_) in k2_legs if n_ > 0 and (any(wanted(c, "%s/%s/%s" % (name, w, i)) for i in ("plain", "tables", "plain+lut12"))
                                                                    or (w == "k2" and args.with_edit and not c.multi and (wanted(c, "%s/k2_edit/tables" % name) or wanted(c, "%s/k2_edit/plain" % name))))]
    if k2_legs:
        fm.options["lf_table"] = 0
        t0 = time.time()
        keep = want_cpu and any(w == "k2" for w, _, _ in k2_legs)
        index = fm.BiFMIndex.from_sequences((_Dev(text), _Dev(seq_off)), 5, "IB16", 16, keep_host=keep)
        build_plain = time.time() - t0
        del fm.options["lf_table"]
        sel_base = fm.options["kernel_select"]
        reads = {w: sample_reads(c, text, lengths, L, n_, 2000 + c.rank + 17 * L, "k2") for (w, L, n_) in k2_legs}
        torch.cuda.synchronize()
        if not keep:
            del text
            text = None
            torch.cuda.empty_cache()
        stats = capi.Stats()
        sc = _scheme_struct(capi, scheme)

        def k2_run(w, L, n_, index_kind, build_s, edit=False):
            rid = "%s/%s%s/%s" % (name, w, "_edit" if edit else "", index_kind)
            sc[0].edit = 1 if edit else 0
            if not wanted(c, rid):
                return None
            qb, qo = reads[w]
            # capacity of the hit buffers: an untimed pass tells how many records this batch produces (repeat-rich texts report many cursors per read)
            probe = C.c_uint64()
            rc = capi.lib().fmgpu_search_scheme(index._h, C.c_void_p(qb.data_ptr()), C.c_void_p(qo.data_ptr()), n_, C.byref(sc[0]), capi.UINT64_MAX,
                                                None, 0, C.byref(probe), None, None)
            if rc not in (0, capi.FMGPU_ERR_CAPACITY):
                capi.check(rc)
            hit_cap = int(probe.value) + 1024
            if c.multi:                                           # every rank sizes its buffers for the largest count
                t_ = torch.tensor([hit_cap], dtype=torch.int64, device="cpu" if c.via_host else c.dev)
                c.dist.all_reduce(t_, op=c.dist.ReduceOp.MAX)
                hit_cap = int(t_.item())
            hits_bufs = [torch.empty(hit_cap * 40, dtype=torch.uint8, device=c.dev) for _ in range(2 if c.multi else 1)]
            pk_cap = (hit_cap + 65535) // 65536 * 65536          # the message size of every step and rank, agreed above: nothing is negotiated inside the timed region
            packed_hits = [torch.zeros((pk_cap + 1, 3), dtype=torch.int64, device=c.dev) for _ in range(2)] if c.multi else None
            xch = Exchange(c, (pk_cap + 1) * 24) if c.multi else None
            state = {"i": 0, "cnt": 0}

            def step(log):
                b = state["i"] % len(hits_bufs); state["i"] += 1
                if xch:
                    xch.wait(b)
                cnt = C.c_uint64()
                capi.check(capi.lib().fmgpu_search_scheme(index._h, C.c_void_p(qb.data_ptr()), C.c_void_p(qo.data_ptr()), n_, C.byref(sc[0]), capi.UINT64_MAX,
                                                          C.c_void_p(hits_bufs[b].data_ptr()), hit_cap, C.byref(cnt), C.byref(stats), None))
                state["cnt"] = int(cnt.value)
                log.append({"kernel_ms": stats.kernel_ms, "units": stats.lf_steps, "table_bytes": stats.table_bytes, "table_accesses": stats.table_accesses, "hits": stats.hits, "table_steps": stats.table_steps})
                if xch:                                           # 24 bytes per hit (qidx:32 | lb:32, len:32 | errors + key:32, lb_rev:32 | key:32); the rank's count rides in the last row
                    capi.check(capi.lib().fmgpu_hits_pack24(C.c_void_p(hits_bufs[b].data_ptr()), int(cnt.value), C.c_void_p(packed_hits[b].data_ptr()), None))
                    packed_hits[b][pk_cap, 0] = int(cnt.value)
                    xch.send(packed_hits[b].view(torch.uint8).view(-1), b)

            elapsed, log = timed(c, step, xch.drain if xch else None)
            k_ms = mean([x["kernel_ms"] for x in log]); units = mean([x["units"] for x in log]); nh = mean([x["hits"] for x in log])
            qps = c.world * n_ * args.steps / elapsed
            lean_off = sel_base & capi.SEL_NO_LEAN
            dense = bool(index.formats & capi.FMT_DENSE) and not (sel_base & capi.SEL_LEAN_FORMAT_A)
            kernel = "k_scheme_fast_edit" if edit else ("k_scheme_lean" if index_kind.startswith("plain") and not lean_off else "k_scheme_fast")
            part = [L // 4 + (1 if p < L % 4 else 0) for p in range(4)]
            rec = {"id": rid, "metric": "queries/sec (GRCh38-sized index, %s x %dbp, k=2 %s, h2(4,0,2))" % ("10M" if w == "k2" and not edit else ("%.1fM per GPU" % (n_ / 1e6)), L, "edit distance" if edit else "Hamming"),
                   "value": qps, "unit": "queries/s", "n_gpus": c.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                   "higher_is_better": True, "scaling": "strong" if (c.multi and w == "k2_151") else "weak", "vs_baseline": None, "dtype": "u64",
                   "data": "synthetic" if name != "fasta" else "real",
                   "config": {"workload": "grch38_k2" if w == "k2" else "grch38_k2_151bp (configs[3]%s)" % ("" if c.multi else ": one GPU's share of 8"), **base_cfg,
                              "index": "BiFMIndex", "index_kind": index_kind, "queries_per_gpu": n_, "read_len": L, "scheme": "h2(4,0,2)", "partition": part,
                              "index_device_bytes": index.device_bytes, "index_build_s": round(build_s, 2),
                              "tables": None if index_kind == "plain" else ({"prefix_symbols": 12} if index_kind == "plain+lut12" else {"lf": True, "prefix_symbols": args.prefix_len, "walk": "LF,LF^2,LF^3 + LF^16"})},
                   "gbp_per_s": qps * L / 1e9, "hits": int(nh)}
            st = {k: mean([x[k] for x in log]) for k in ("table_bytes", "table_accesses", "table_steps")}
            if kernel == "k_scheme_lean":
                # fmgpu_stats of the lean kernel: table_accesses = blocks it fetched, table_bytes = visited nodes of several rows (the rest of the walked nodes stood on one row),
                # table_steps = nodes that prefix-table entries stood for (one 16-byte entry per 12 of them)
                multi, single, blocks = st["table_bytes"], units - st["table_bytes"] - st["table_steps"], st["table_accesses"]
                per = FMT_NODE_DENSE if dense else FMT_NODE_BLOCKS
                rec["roofline"] = roofline_plain(kernel, k_ms, units, "visited nodes (cursor extensions)", multi * per[0] + single * per[1] + st["table_steps"] / 12.0 * 16.0,
                                                 ("dense DNA blocks (32 B per 64 rows: four counts + two bit planes)" if dense else "the entries of symbols 1..4 of the one-symbol blocks (48 B)") +
                                                 ": a node of several rows reads both interval ends' blocks (%d B), a one-row node its row's block (%d B); %.1f %% of the nodes hold several rows"
                                                 % (per[0], per[1], 100.0 * multi / max(units, 1.0)),
                                                 SEC8D_STEP_DNA, "2 x sizeof(InterleavedBitvector16<5>::Block) = 112 B per visited node",
                                                 blocks * per[1] + nh * 40.0 + n_ * (L + 8.0), "%d B per block the kernel fetched (both ends in one block: one fetch; a node re-visited for its next sibling "
                                                 "fetches again) + 40 B per hit record + the reads; its 16-byte stack frames (write-through, read back by LDS-DMA) are not counted" % per[1], blocks)
                rec["roofline"]["nodes_of_several_rows"] = multi
            elif index_kind == "plain" or edit:                # (the edit-distance kernel: whole 64-byte blocks; what it loads is counted in the kernel where it counts)
                rec["roofline"] = roofline_plain(kernel, k_ms, units, "visited nodes (cursor extensions)", units * float(FMT_NODE_EDIT[0] if index_kind == "plain" else SEC8D_STEP_DNA),
                                                 "both interval ends' 64-byte blocks per visited node (upper bound: a one-row node reads one block, or nothing where its LF is known)"
                                                 if index_kind == "plain" else "SURVEY 8d's 112 B per visited node (the table-driven kernel serves one-row nodes from LF entries)",
                                                 SEC8D_STEP_DNA, "2 x sizeof(InterleavedBitvector16<5>::Block) = 112 B per visited node",
                                                 (st["table_bytes"] + n_ * (L + 8.0)) if st["table_bytes"] else None, "bytes of the loads the kernel issued (blocks, LF entries, frames, hit records), counted in the kernel, + the reads",
                                                 st["table_accesses"] if st["table_bytes"] else None)
            else:
                rec["roofline"] = roofline_loaded(st, n_ * (L + 8), k_ms, kernel, units, "visited nodes")
            if xch:
                rec["exchange"] = {"collective": xch.mode, "bytes_per_rank_and_step": int(xch.last[0].numel()), "verified_on_rank0": xch.verify(),
                                   "record": "24 B per hit (qidx:32 | lb:32, len:32 | errors + order key:32, lb_rev:32 | order key:32); message = the largest count over the ranks "
                                             "(agreed once, before the timed region) + one trailing row that carries the rank's own count",
                                   "world_size_seen": c.dist.get_world_size()}
            attach_traffic(c, rec)
            if keep and w == "k2" and index_kind == "plain" and not edit:
                rec["cpu_baseline"] = cpu_baseline(c, index, True, qb, qo, n_, L, scheme, None, None, (hits_bufs[0], state["cnt"]))
            return rec

        for (w, L, n_) in k2_legs:
            out.append(k2_run(w, L, n_, "plain", build_plain))
        if not c.multi and any(wanted(c, "%s/%s/plain+lut12" % (name, w)) for w, _, _ in k2_legs):      # the same searches behind a 12-symbol prefix table (268 MB), no other table
            t0 = time.time()
            index.accelerate_search(12, 0)
            for (w, L, n_) in k2_legs:
                out.append(k2_run(w, L, n_, "plain+lut12", build_plain + time.time() - t0))
            index.accelerate_search(0, 0)
        if args.with_edit and not c.multi and any(w == "k2" for w, _, _ in k2_legs) and wanted(c, "%s/k2_edit/plain" % name):
            qb, qo = reads["k2"]                                    # edit distance (the reference's default, search_ng26<true>) on the plain index: one-row nodes read their row's block
            n_e = min(2_000_000, nq)
            full = reads["k2"]
            reads["k2"] = (qb[: n_e * 101], qo[: n_e + 1])
            out.append(k2_run("k2", 101, n_e, "plain", build_plain, edit=True))
            reads["k2"] = full
        # the optional tables (LF, prefix, walk: 224 GB at this size).  N > 1 runs on the plain index north_star replicates, unless --multi-tables
        want_tab = (not c.multi or args.multi_tables) and (any(wanted(c, "%s/%s/tables" % (name, w)) for w, _, _ in k2_legs) or
                                                            (args.with_edit and not c.multi and wanted(c, "%s/k2_edit/tables" % name)))
        if want_tab:
            t0 = time.time()
            index.accelerate_lf(True)
            if c.multi and args.prefix_len >= 16 and max(n_ for _, _, n_ in k2_legs) > 12_500_000:
                # fewer than 8 ranks: a rank's share of the 100 M reads grows (2 ranks: 50 M reads, ~2 x 10^8 hit records in double-buffered raw, packed and
                # gathered form: ~50 GB beside 232 GB of index and reads) — the 15-symbol prefix table (17 GB instead of 69) keeps it well inside 288 GB
                print("bench.py: %d reads per rank: using a 15-symbol prefix table" % max(n_ for _, _, n_ in k2_legs), file=sys.stderr, flush=True)
                args.prefix_len = 15
            if args.prefix_len >= 16:                              # the 16-symbol prefix table is 69 GB: keep room for the walk tables (99 GB) and this run's hit buffers
                free_b, _ = torch.cuda.mem_get_info()
                if free_b < (69 + 99 + 24) * (1 << 30):
                    print("bench.py: %.0f GB of HBM free: using a 15-symbol prefix table" % (free_b / 2**30), file=sys.stderr, flush=True)
                    args.prefix_len = 15
            try:
                index.accelerate_search(args.prefix_len, 3)
            except fm.FmgpuError as ex:                            # (the 16-symbol table is 69 GB: should the card be short of memory, one symbol less)
                if args.prefix_len < 16:
                    raise
                print("bench.py: %s; retrying with a 15-symbol prefix table" % ex, file=sys.stderr, flush=True)
                args.prefix_len = 15
                index.accelerate_search(args.prefix_len, 3)
            build_tab = build_plain + time.time() - t0
            for (w, L, n_) in k2_legs:
                out.append(k2_run(w, L, n_, "tables", build_tab))
            if args.with_edit and not c.multi and any(w == "k2" for w, _, _ in k2_legs):
                qb, qo = reads["k2"]
                n_e = min(2_000_000, nq)
                reads["k2"] = (qb[: n_e * 101], qo[: n_e + 1])
                out.append(k2_run("k2", 101, n_e, "tables", build_tab, edit=True))
        index.close()
        del index, reads
    del text
    torch.cuda.empty_cache()
    return out


def _scheme_struct(capi, scheme):
    import numpy as np
    pi, l, u = (np.ascontiguousarray(x, dtype=np.uint64) for x in scheme)
    sc = capi.Scheme()
    sc.n_searches, sc.n_parts = pi.shape
    sc.pi, sc.l, sc.u = (x.ctypes.data_as(capi.u64p) for x in (pi, l, u))
    sc.partition = None
    sc.edit = 0
    return sc, (pi, l, u)


# ---------------------------------------------------------------------------------------------------------------- protein, configs[4]
def run_protein(c, nseq, tag, plain_only=False):
    torch, np, fm, capi, args = c.torch, c.np, c.fm, c.capi, c.args
    import ctypes as C
    ids = [tag + "/exact/wavelet", tag + "/exact/tables", tag + "/exact/tree", tag + "/exact/wavelet+lut5"]
    if plain_only:                                            # (the optional tables of a 1e10-row index — 16 bytes per row and table — are not this record's subject)
        ids[1] = ids[3] = None
    if c.only and not any(i in c.only for i in ids if i):
        return []
    sigma, L, nq = 28, 40, args.nq
    total = nseq * PROTEIN_SEQ_LEN
    g = torch.Generator(device=c.dev)
    g.manual_seed(42)
    text = torch.empty(total, dtype=torch.uint8, device=c.dev)
    for lo in range(0, total, 1 << 28):
        hi = min(total, lo + (1 << 28))
        text[lo:hi] = torch.randint(1, sigma, (hi - lo,), generator=g, device=c.dev, dtype=torch.uint8)
    seq_off = torch.arange(nseq + 1, device=c.dev, dtype=torch.int64) * PROTEIN_SEQ_LEN
    qbuf, qoff = sample_reads(c, text, None, L, nq, 1000, "exact", sigma=sigma, inside=(nseq, PROTEIN_SEQ_LEN))
    torch.cuda.synchronize()
    fm.options["lf_table"] = 0
    # (fmgpu_build_index's own rule, csrc/fmgpu_build.hip: all suffixes at once where 44 bytes per row with 64-bit rows / 32 with 32-bit rows fit the free memory, else bucket by bucket
    # with the inverse suffix array as rank array, 10 / 6 bytes per row)
    rows_ = total + nseq
    srt = fm.options["suffix_sorter"] or (1 if rows_ * (44 if rows_ >= (1 << 32) - 64 else 32) <= torch.cuda.mem_get_info()[0] else 2)
    sorter = {1: "all suffixes at once (suffix array + rank array + keys of all rows)", 2: "bucket by bucket, prefix doubling on the ties with the inverse suffix array as rank array (no suffix array)",
              3: "bucket by bucket, no array of n entries"}[srt]
    t0 = time.time()
    index = fm.FMIndex.from_sequences((_Dev(text), _Dev(seq_off)), sigma, "WAVELET", 16, keep_host=False)
    build_s = time.time() - t0
    del fm.options["lf_table"]
    sel_base = fm.options["kernel_select"]
    out_t = torch.empty(2 * nq, dtype=torch.int64, device=c.dev)
    stats = capi.Stats()
    # text-side check of the built index, before the text goes: the first row of 200 k found reads, located, must spell the read where the sampled entry + LF steps point
    capi.check(capi.lib().fmgpu_search_exact(index._h, C.c_void_p(qbuf.data_ptr()), C.c_void_p(qoff.data_ptr()), nq, C.c_void_p(out_t[:nq].data_ptr()), C.c_void_p(out_t[nq:].data_ptr()), None, None))
    pick = torch.nonzero(out_t[nq:] > 0)[:200_000, 0]
    rows = out_t[:nq][pick].contiguous()
    loc = torch.empty(3 * rows.numel(), dtype=torch.int64, device=c.dev)
    k = rows.numel()
    capi.check(capi.lib().fmgpu_locate(index._h, C.c_void_p(rows.data_ptr()), k, C.c_void_p(loc[:k].data_ptr()), C.c_void_p(loc[k:2 * k].data_ptr()), C.c_void_p(loc[2 * k:].data_ptr()), None, None))
    at = loc[:k] * PROTEIN_SEQ_LEN + loc[k:2 * k] + loc[2 * k:]
    located_ok = bool(torch.equal(text[at[:, None] + torch.arange(L, device=c.dev)[None, :]], qbuf.reshape(nq, L)[pick])) and int(loc[2 * k:].max().item()) < 16
    if not located_ok:
        raise SystemExit("bench.py: %s — located rows do not spell their reads (or a row needs 16 LF steps and more at sampling rate 16)" % tag)
    del text, loc, at, rows, pick
    torch.cuda.empty_cache()

    def step(log):
        capi.check(capi.lib().fmgpu_search_exact(index._h, C.c_void_p(qbuf.data_ptr()), C.c_void_p(qoff.data_ptr()), nq,
                                                 C.c_void_p(out_t[:nq].data_ptr()), C.c_void_p(out_t[nq:].data_ptr()), C.byref(stats), None))
        log.append({"kernel_ms": stats.kernel_ms, "units": stats.lf_steps, "table_bytes": stats.table_bytes, "table_accesses": stats.table_accesses, "table_steps": stats.table_steps})

    def rec_of(rid, kind, kernel, elapsed, log, b_s, tables):
        k_ms = mean([x["kernel_ms"] for x in log]); units = mean([x["units"] for x in log])
        qps = nq * args.steps / elapsed
        r = {"id": rid, "metric": "queries/sec (sigma=28 protein index, 10M x 40aa, exact, Wavelet)", "value": qps, "unit": "queries/s", "n_gpus": 1, "steps": args.steps,
             "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
             "config": {"workload": "protein_exact", "text": {"text": "uniform residues in {1..27}", "symbols": total, "sequences": nseq}, "sigma": sigma, "layout": "Wavelet",
                        "index": "FMIndex", "index_kind": kind, "row_bits": index.row_bits, "queries_per_gpu": nq, "read_len": L, "index_device_bytes": index.device_bytes,
                        "index_build_s": round(b_s, 2), "suffix_sorter": sorter, "tables": tables},
             "gres_per_s": qps * L / 1e9, "hits": int((out_t[nq:] > 0).sum().item())}
        # every read was cut from the text; every 10th then got one substitution (27^40 strings of its length: such a read is not in the text): exactly the others are found
        r["located_origins_checked"] = 200_000                    # (against the text, right after construction)
        r["hits_expected"] = nq - (nq + 9) // 10
        if r["hits"] != r["hits_expected"]:
            raise SystemExit("bench.py: %s found %d of the %d reads that were cut from the text" % (rid, r["hits"], r["hits_expected"]))
        st = {k: mean([x[k] for x in log]) for k in ("table_bytes", "table_accesses", "table_steps")}
        if kind.startswith("wavelet") and kernel in ("k_exact_s", "k_exact_ls"):
            r["roofline"] = roofline_plain(kernel, k_ms, units, "executed LF steps", (units - st["table_steps"]) * FMT_STEP_PLANES28 + st["table_steps"] / 5.0 * 8.0,
                                           ("the read's last 5 symbols from one interval-table entry, then " if kernel == "k_exact_ls" else "") + "one line per step on the symbol planes beside the tree (one 128-byte line per 64 rows: 5 symbol planes + sigma bit-packed counts): 2 interval ends x "
                                           "44 B read of a line (40 B of planes + the symbol's count) = 88 B per executed LF step",
                                           SEC8D_STEP_WAVELET28, "2 ends x 5 levels x (8 + 1 + 8) useful bytes of the reference's binary wavelet tree = 170 B per LF step: record " + tag + "/exact/tree (k_exact_m, same index, same reads, same run)",
                                           st["table_bytes"] + nq * (L + 8 + 16), "44 B per line the kernel fetched (an end in the other end's line: one fetch) + queries and results", st["table_accesses"])
        elif kind == "wavelet":
            r["roofline"] = roofline_plain(kernel, k_ms, units, "executed LF steps", units * FMT_STEP_TREE28,
                                           "the two-level multi-ary tree: per interval end an 8-ary block (4-byte count + three 8-byte planes) and a 4-ary block (4-byte count + two planes) = 96 B per executed LF step",
                                           SEC8D_STEP_WAVELET28, "2 ends x 5 levels x (8 + 1 + 8) useful bytes of the reference's binary wavelet tree = 170 B per LF step (this record IS SURVEY 8d as written)",
                                           st["table_bytes"] + nq * (L + 8 + 16), "bytes the kernel read of the blocks (an end in the other end's block: one read) + queries and results", st["table_accesses"])
        else:
            r["roofline"] = roofline_loaded(st, nq * (L + 8 + 16), k_ms, kernel, units, "LF steps")
        attach_traffic(c, r)
        return r

    out = []
    flat = bool(index.formats & capi.FMT_PLANES) and not (sel_base & capi.SEL_EXACT_ON_TREE)
    flat_ms = None
    if wanted(c, ids[0]):
        elapsed, log = timed(c, step)
        out.append(rec_of(ids[0], "wavelet", "k_exact_s" if flat else "k_exact_m", elapsed, log, build_s, None))
        flat_ms = out[-1]["roofline"]["kernel_ms"]
    if flat and wanted(c, ids[2]):                                # the same index and reads on the tree itself: SURVEY 8d's accounting as written
        if not wanted(c, ids[0]):
            step([]); torch.cuda.synchronize()
        keep = out_t.clone()
        with fm.options(kernel_select=sel_base | capi.SEL_EXACT_ON_TREE):
            elapsed, log = timed(c, step)
        r1 = rec_of(ids[2], "wavelet", "k_exact_m", elapsed, log, build_s, None)
        r1["equal_to_the_line_kernel"] = bool(torch.equal(keep, out_t))
        if not r1["equal_to_the_line_kernel"]:
            raise SystemExit("bench.py: k_exact_m and k_exact_s disagree")
        if flat_ms:
            r1["roofline"]["line_kernel_speedup"] = r1["roofline"]["kernel_ms"] / flat_ms
        out.append(r1)
        del keep
    if flat and ids[3] and wanted(c, ids[3]):                     # the same search behind a 5-symbol interval table (27^5 entries: 115 MB with 32-bit rows), no other table
        if not wanted(c, ids[0]):
            step([]); torch.cuda.synchronize()
        keep = out_t.clone()
        t0 = time.time()
        index.accelerate(0, lut_len=5, walk=0)
        elapsed, log = timed(c, step)
        r2 = rec_of(tag + "/exact/wavelet+lut5", "wavelet+lut5", "k_exact_ls", elapsed, log, build_s + time.time() - t0, {"suffix_interval_symbols": 5})
        r2["equal_to_the_plain_index"] = bool(torch.equal(keep, out_t))
        if not r2["equal_to_the_plain_index"]:
            raise SystemExit("bench.py: protein exact search with and without the interval table disagree")
        if flat_ms:
            r2["roofline"]["speedup_over_plain_index_kernel"] = flat_ms / r2["roofline"]["kernel_ms"]
        out.append(r2)
        del keep
    if ids[1] and wanted(c, ids[1]):
        t0 = time.time()
        wide = index.row_bits == 64                            # 64-bit rows: 16-byte entries — the 6-symbol walk alone (72 GB at 4.5e9 rows), no 12-symbol one
        index.accelerate(1, lut_len=6, walk=1 if wide else 2)
        b2 = build_s + time.time() - t0
        elapsed, log = timed(c, step)
        out.append(rec_of(ids[1], "tables", "k_exact_kstep", elapsed, log, b2, {"block_table_expansion": True, "suffix_interval_symbols": 6, "walk_symbols_per_load": 6 if wide else 12}))
    index.close()
    torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------------------------------------------- CPU leg
def cpu_baseline(c, index, bidir, qbuf, qoff, nq, L, scheme, out_lb, out_len, hits):
    """the CPU restatement (oracle/) on the host cores over a bounded sample of the same reads; also a parity check of the GPU results on that sample:
    exact — every (lb, len); k = 2 — every hit record (qidx, lb, lb_rev, len, errors) in callback order"""
    np, torch, capi = c.np, c.torch, c.capi
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fmoracle as fo
    cores = c.host_cores
    t0 = time.time()
    bwt = index.built_array(0)
    bwt_rev = index.built_array(1) if bidir else None
    ox = fo.OraIndex.from_bwt("IB16", 5, bwt, bwt_rev, None, None, None)
    ox.spread(cores)                                          # pages of the occurrence tables over all NUMA nodes (built by one thread = on one node)
    build = time.time() - t0
    sample_cap = min(nq, 12_000_000 if not bidir else 1_000_000)
    hq_all = qbuf[: sample_cap * L].cpu().numpy()
    ho_all = qoff[: sample_cap + 1].cpu().numpy().astype(np.uint64)

    def run(count, threads=cores, batched=True, records=False):
        t0 = time.time()
        if not bidir and batched:                             # search/SearchNoErrors.h:28-86, the reference's 32-way interleaved form
            r = ox.search_exact_batched(hq_all[: count * L], ho_all[: count + 1], 32, threads)
        elif not bidir:                                       # :12-26, one query at a time
            r = ox.search_exact(hq_all[: count * L], ho_all[: count + 1], nthreads=threads)
        else:                                                 # timed: the walk itself (counts per query); records=True adds a second pass that writes the hit records
            r = ox.search_ng26(hq_all[: count * L], ho_all[: count + 1], scheme, nthreads=threads, cap=16 * count + 1024, records=records)
        return r, time.time() - t0

    pilot = min(sample_cap, 400_000 if not bidir else 50_000)
    _, dt = run(pilot)                                        # also warms the caches / OpenMP team
    sample = c.args.cpu_sample or int(min(sample_cap, max(pilot, pilot * 12.0 / max(dt, 1e-3))))
    r, dt = run(sample)
    if not bidir:
        lb, ln = r
        ok = bool(np.array_equal(lb, out_lb[:sample].cpu().numpy().astype(np.uint64)) and np.array_equal(ln, out_len[:sample].cpu().numpy().astype(np.uint64)))
    else:
        checked = min(sample, 200_000)                        # hit-by-hit parity on the first reads of the sample (a second, untimed oracle pass writes the records)
        r, _ = run(checked, records=True)
        buf, cnt = hits
        rec = buf[: cnt * 40].view(torch.int64).view(-1, 5)
        mine = rec[rec[:, 0] < checked].cpu().numpy()
        e_key = mine[:, 4].astype(np.uint64)                   # errors:8 | key high:24 | seq (key low):32 -> (qidx, key) is the reference's callback order
        order = np.lexsort(((((e_key >> np.uint64(8)) & np.uint64(0xffffff)) << np.uint64(32)) | (e_key >> np.uint64(32)), mine[:, 0]))
        mine = mine[order]
        oh = r[0]
        ok = bool(len(oh) == len(mine) and all(np.array_equal(mine[:, k].astype(np.uint64), oh[f].astype(np.uint64)) for k, f in ((0, "qidx"), (1, "lb"), (2, "lb_rev"), (3, "len"))) and
                  np.array_equal((mine[:, 4] & 0xff).astype(np.uint64), oh["errors"].astype(np.uint64)))
    one = max(1000, min(sample, int(sample / dt * 3.0 / cores)))          # ~3 s on one thread (single-thread figure beside all cores)
    _, dt1 = run(one, threads=1)
    out = {"value": sample / dt, "unit": "queries/s", "cores": cores, "kind": "port", "sample_reads": sample,
           "single_thread": {"value": one / dt1, "unit": "queries/s", "sample": "the first %d reads, one thread" % one},
           "parallel_efficiency": (sample / dt) / (one / dt1) / cores,
           "sample": "the first %d reads of the same batch, OpenMP over queries on all host cores (OMP_PROC_BIND=%s, OMP_PLACES=%s; occurrence tables re-homed over the NUMA nodes "
                     "by a parallel first touch)%s; index rebuilt on the host from the GPU-built BWT in %.0f s"
                     % (sample, os.environ.get("OMP_PROC_BIND"), os.environ.get("OMP_PLACES"), "" if bidir else ", 32 cursors interleaved per thread (SearchNoErrors.h:28-86)", build),
           "seconds": dt, "gpu_results_match_on_sample": ok,
           "compared": "every (lb, len) of the sample" if not bidir else "every hit record (qidx, lb, lb_rev, len, errors) of the first %d reads, in callback order" % min(sample, 200_000)}
    if not bidir:                                             # the one-query-at-a-time form beside it, on a sample a fifth the size
        few = max(1000, sample // 5)
        _, dts = run(few, batched=False)
        out["one_query_at_a_time"] = {"value": few / dts, "unit": "queries/s", "sample": "the first %d reads, all cores" % few}
    return out


if __name__ == "__main__":
    main()

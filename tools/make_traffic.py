"""rebuild profiles/r02_traffic.json (what bench.py replays into roofline.traffic, labelled as replayed) from the committed rocprofv3 summaries:
python tools/make_traffic.py"""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDS = {"genome/exact/plain": ("exact_plain", "k_exact_a"), "genome/exact/tables": ("exact_tables", "k_exact_kstep"),
           "genome/k2/plain": ("k2_plain", "k_scheme_fast"), "genome/k2/tables": ("k2_tables", "k_scheme_fast"),
           "protein/exact/wavelet": ("protein_wavelet", "k_exact_m"), "genome/k2_edit/tables": ("edit_genome", "k_scheme_fast_edit"),
           "uniform/k2_edit/tables": ("edit_uniform", "k_scheme_fast_edit")}
out = {}
for rid, (tag, kernel) in RECORDS.items():
    path = os.path.join(ROOT, "profiles", "r02_%s_rocprof_summary.json" % tag)
    if not os.path.exists(path):
        continue
    d = json.load(open(path))
    t = d.get("hbm_traffic_bytes_per_launch")
    if not t:
        continue
    c = d["counters_per_launch"]
    ks = [k for k in d["kernel_stats"] if kernel in k["name"] and kernel + "_" not in k["name"].split("(")[0]]
    line = d.get("bench_line", {})
    recs = [line] + line.get("records", []) if line else []
    same = next((r for r in recs if r.get("id", rid if r is line else None) == rid or (r is line and "id" not in r)), None)
    hit, miss = c.get("TCC_HIT_sum", {}).get("mean"), c.get("TCC_MISS_sum", {}).get("mean")
    out[rid] = {"kernel": kernel, "bytes_per_launch": t["corrected_bytes"], "fetch_size_kb_raw": t["fetch_raw_kb"], "write_size_kb": t["write_kb"],
                "line_requests_per_launch": c.get("TCC_EA0_RDREQ_sum", {}).get("mean"),
                "kernel_avg_ms_rocprof": ks[0]["avg_ms"] if ks else None,
                "kernel_ms_hip_events_same_run": (same or {}).get("roofline", {}).get("kernel_ms"),
                "l2_hit_rate": hit / (hit + miss) if hit is not None and miss else None,
                "source": "profiles/r02_%s_rocprof_summary.json" % tag,
                "collected": "tools/profile_r02.sh: rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_* / --pmc SQ_* in separate passes over "
                             "`bench.py --steps 5 --warmup 1 --no-cpu-baseline --only %s`; bytes = FETCH_SIZE x 2 (MI355X_MICROARCH.md: 128-B fills tallied at 64 B on gfx950) "
                             "+ WRITE_SIZE, mean over the kernel launches" % rid}
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_traffic.json"), "w"), indent=1)
for k, v in out.items():
    print(k, "%.1f GB" % (v["bytes_per_launch"] / 1e9), v["kernel_avg_ms_rocprof"], v["kernel_ms_hip_events_same_run"])

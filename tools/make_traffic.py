"""rebuild profiles/r03_traffic.json (what bench.py replays into roofline.traffic, labelled as replayed) from the committed rocprofv3 summaries:
python tools/make_traffic.py [round tag, default r03]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r03"
RECORDS = {"genome/exact/plain": ("exact_plain", "k_exact_p"), "genome/exact/single": ("exact_single", "k_exact_a"), "genome/exact/tables": ("exact_tables", "k_exact_kstep"),
           "genome/k2/plain": ("k2_plain", "k_scheme_lean"), "genome/k2_151/plain": ("k2_151_plain", "k_scheme_lean"), "genome/k2/tables": ("k2_tables", "k_scheme_fast"),
           "genome/locate/plain": ("locate_plain", "k_locate_coop" if RND >= "r04" else "k_locate_fused"), "genome/exact/plain+lut12": ("exact_lut12", "k_exact_p"),
           "protein/exact/wavelet": ("protein_wavelet", "k_exact_s"), "protein/exact/tree": ("protein_tree", "k_exact_m"), "protein_wide/exact/wavelet": ("protein_wide", "k_exact_s"), "protein_xl/exact/wavelet": ("protein_xl", "k_exact_s"),
           "genome/k2_edit/tables": ("edit_genome", "k_scheme_fast_edit"), "genome/k2_edit/plain": ("edit_plain", "k_scheme_fast_edit"), "uniform/k2_edit/tables": ("edit_uniform", "k_scheme_fast_edit")}
out = {}
for rid, (tag, kernel) in RECORDS.items():
    path = os.path.join(ROOT, "profiles", "%s_%s_rocprof_summary.json" % (RND, tag))
    if not os.path.exists(path):
        continue
    d = json.load(open(path))
    t = d.get("hbm_traffic_bytes_per_launch")
    if not t:
        continue
    c = d["counters_per_launch"]
    ks = [k for k in d["kernel_stats"] if kernel in k["name"]]
    same = next((r for r in d.get("bench_records", []) if r.get("id") == rid), None)
    hit, miss = c.get("TCC_HIT_sum", {}).get("mean"), c.get("TCC_MISS_sum", {}).get("mean")
    out[rid] = {"kernel": kernel, "bytes_per_launch": t["corrected_bytes"], "fetch_size_kb_raw": t["fetch_raw_kb"], "write_size_kb": t["write_kb"],
                "line_requests_per_launch": c.get("TCC_EA0_RDREQ_sum", {}).get("mean"),
                "kernel_avg_ms_rocprof": ks[0]["avg_ms"] if ks else None,
                "kernel_ms_hip_events_same_run": (same or {}).get("roofline", {}).get("kernel_ms"),
                "l2_hit_rate": hit / (hit + miss) if hit is not None and miss else None,
                "source": "profiles/%s_%s_rocprof_summary.json" % (RND, tag),
                "collected": "tools/profile_%s.sh: rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_* / --pmc SQ_* in separate passes over "
                             "`bench.py --steps 5 --warmup 1 --no-cpu-baseline --only %s`; bytes = FETCH_SIZE x 2 (MI355X_MICROARCH.md: 128-B fills tallied at 64 B on gfx950) "
                             "+ WRITE_SIZE, mean over the kernel launches" % (RND, rid)}
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_traffic.json" % RND), "w"), indent=1)
for k, v in out.items():
    print(k, "%.1f GB" % (v["bytes_per_launch"] / 1e9), v["kernel_avg_ms_rocprof"], v["kernel_ms_hip_events_same_run"])

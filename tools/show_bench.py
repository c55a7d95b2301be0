"""prints the records of a bench.py run as a table (dev tool): python tools/show_bench.py [bench_records.json | profiles/r03_bench_records.json]"""
import json, sys
path = sys.argv[1] if len(sys.argv) > 1 else "bench_records.json"
d = json.load(open(path))
records = d if isinstance(d, list) else d.get("records", [d])
for r in records:
    rf = r["roofline"]
    print("%-28s value=%.3e %-11s ms/step=%8.2f %-18s %8.2fms frac=%.3f line=%s build=%ss idx=%.1fGB hits=%s" % (
        r["id"], r["value"], r["unit"], r["ms_per_step"], rf["kernel"], rf["kernel_ms"], rf["frac"],
        ("%.3f" % rf["line_granular"]["frac"]) if "line_granular" in rf else ("%.3f" % rf["loaded"]["line_granular_frac"] if "line_granular_frac" in rf.get("loaded", {}) else "  -  "),
        r["config"].get("index_build_s"), (r["config"].get("index_device_bytes") or 0) / 1e9, r.get("hits")))
    if "symbols_until_one_row" in r:
        print("    depth:", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in r["symbols_until_one_row"].items() if k != "what"})
    if "cpu_baseline" in r:
        c = r["cpu_baseline"]
        print("    cpu: %.3e q/s on %d cores, 1 thread %.3e, eff %.2f, match=%s" % (c["value"], c["cores"], c["single_thread"]["value"], c["parallel_efficiency"], c["gpu_results_match_on_sample"]))
    if "exchange" in r:
        print("    exchange:", r["exchange"])

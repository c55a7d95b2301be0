"""print per-kernel averages of the rocprofv3 csv outputs under a profile directory (dev tool)"""
import csv, glob, collections, sys
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(d + "/trace/*/*_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(f"{r['Name'][:80]:80s} calls={r['Calls']:>4s} avg={float(r['AverageNs'])/1e6:9.3f} ms  {r['Percentage']}%")
for f in sorted(glob.glob(d + "/pmc_*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f.split("/")[-3], {k: f"{sum(v)/len(v):.4g} (x{len(v)})" for k, v in agg.items()})

#!/usr/bin/env python3
"""Summarise rocprofv3 outputs under gpurun_out/: kernel stats (prof_<tag>) and PMC passes (pmc_<tag>_*)."""
import collections, csv, glob, sys
tag = sys.argv[1]
for f in glob.glob(f"gpurun_out/prof_{tag}/*/*_kernel_stats.csv"):
    print("== kernel stats", f)
    for r in list(csv.reader(open(f)))[:14]:
        print("  %-70s calls=%-5s avg_ns=%-12s pct=%s" % (r[0][:70], r[1], r[3][:10], r[4]))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/pmc_{tag}_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "art::" not in k:
            continue
        agg[k.split("(")[0][-44:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("==", k)
    for c, vals in sorted(v.items()):
        print("   %-24s n=%d mean=%.4g" % (c, len(vals), sum(vals) / len(vals)))

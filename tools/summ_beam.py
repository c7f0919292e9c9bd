#!/usr/bin/env python3
"""Summarise tools/prof_beam.sh output: kernel stats + PMC counters of k_beam_chi2 (averages per launch)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
for f in glob.glob("%s/beam_trace/*/*_kernel_stats.csv" % d):
    for r in list(csv.DictReader(open(f)))[:4]:
        print("%-60s calls=%s avg_ms=%.4f pct=%s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6,
                                                   r["Percentage"]))
for sub in ("pmc", "pmc2", "fetch"):
    fs = glob.glob("%s/beam_%s/*/*_counter_collection.csv" % (d, sub))
    if not fs:
        print(sub, "no counter file")
        continue
    agg, meta = collections.defaultdict(list), None
    for r in csv.DictReader(open(fs[0])):
        if "k_beam_chi2" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = r
    print(sub, {k: "%.5g" % (sum(v) / len(v)) for k, v in sorted(agg.items())})
    if meta:
        print("   vgpr", meta["VGPR_Count"], "agpr", meta["Accum_VGPR_Count"], "sgpr", meta["SGPR_Count"], "lds",
              meta["LDS_Block_Size"])

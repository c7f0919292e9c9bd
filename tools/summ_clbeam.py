#!/usr/bin/env python3
"""Summarise tools/prof_clbeam.sh output: kernel stats + PMC counters of k_cl_form / k_beam_chi2."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
for what, kern in (("cl", "k_cl_form"), ("beam", "k_beam_chi2")):
    print("==", what)
    for f in glob.glob("%s/%s_trace/vm/*_kernel_stats.csv" % (d, what)):
        for r in list(csv.DictReader(open(f)))[:6]:
            print("%-70s calls=%s avg_ms=%.4f pct=%s" % (r["Name"][:70], r["Calls"],
                                                       float(r["AverageNs"]) / 1e6, r["Percentage"]))
    for sub in ("pmc", "fetch"):
        fs = glob.glob("%s/%s_%s/vm/*_counter_collection.csv" % (d, what, sub))
        if not fs:
            continue
        agg, meta = collections.defaultdict(list), None
        for r in csv.DictReader(open(fs[0])):
            if kern in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = r
        print(sub, {k: "%.4g" % (sum(v) / len(v)) for k, v in agg.items()})
        if meta:
            print("   vgpr", meta["VGPR_Count"], "agpr", meta["Accum_VGPR_Count"], "sgpr",
                  meta["SGPR_Count"], "lds", meta["LDS_Block_Size"])

"""Summarise a tools/prof_ext.sh run: kernel stats and the PMC counters of the last k_mh_run launch."""
import csv
import glob
import sys

root = sys.argv[1]


def one(pat):
    f = glob.glob(pat)
    return f[0] if f else None


st = one(root + "/trace/*/*kernel_stats.csv")
if st:
    for r in list(csv.DictReader(open(st)))[:5]:
        print("%-70s calls=%s avg_ms=%.3f pct=%s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
for tag in ("pmc_sq", "pmc_sq2"):
    f = one(root + "/%s/*/*counter_collection.csv" % tag)
    if not f:
        continue
    rows = [r for r in csv.DictReader(open(f)) if "k_mh_run<" in r["Kernel_Name"]]
    if not rows:
        continue
    last = max(int(r["Dispatch_Id"]) for r in rows)
    acc, meta = {}, None
    for r in rows:
        if int(r["Dispatch_Id"]) != last:
            continue
        acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        meta = (r["Kernel_Name"][:60], r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"))
    print(tag, {k: "%.4g" % v for k, v in sorted(acc.items())})
    print("   ", meta)

"""Summarise a tools_prof.sh run (gpurun_out/prof_<tag>) for profiles/: kernel stats, per-launch
durations of k_mh_run from the kernel trace, PMC counters of the dominant kernel."""
import csv
import glob
import sys

root = sys.argv[1]
kern = "k_mh_run"


def one(pat):
    f = glob.glob(pat)
    return f[0] if f else None


st = one(root + "/trace/*/*kernel_stats.csv")
for r in list(csv.DictReader(open(st)))[:6]:
    print("%-60s calls=%s avg_ms=%.3f pct=%s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6,
                                                   r["Percentage"]))
tr = one(root + "/trace/*/*kernel_trace.csv")
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(tr))
       if kern in r["Kernel_Name"]]
print("# %s per-launch durations (ms): %s" % (kern, " ".join("%.2f" % d for d in dur)))
for kname in ("k_mh_run<", "k_mh_run_inc<"):
    for tag in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
        f = one(root + "/%s/*/*counter_collection.csv" % tag)
        if not f:
            continue
        rows = [r for r in csv.DictReader(open(f)) if kname in r["Kernel_Name"]]
        if not rows:
            continue
        last = max(int(r["Dispatch_Id"]) for r in rows)        # a 200-step production launch
        acc, meta = {}, None
        for r in rows:
            if int(r["Dispatch_Id"]) != last:
                continue
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            meta = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"))
        print(kname[:-1], tag, {k: "%.4g" % v for k, v in sorted(acc.items())}, "(last launch)")
        if meta:
            print("   vgpr %s agpr %s sgpr %s lds %s" % meta)

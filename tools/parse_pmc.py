import csv,glob,sys
root=sys.argv[1]
for tag in ("pmc_sq","pmc_sq2","pmc_sq3"):
    fs=glob.glob(root+"/%s/*/*counter_collection.csv"%tag)
    if not fs: continue
    rows=[r for r in csv.DictReader(open(fs[0])) if "k_mh_run" in r["Kernel_Name"]]
    for kn in ("k_mh_run<","k_mh_run_inc<"):
        rr=[r for r in rows if kn in r["Kernel_Name"]]
        if not rr: continue
        last=max(int(r["Dispatch_Id"]) for r in rr)
        acc={}
        for r in rr:
            if int(r["Dispatch_Id"])==last: acc[r["Counter_Name"]]=acc.get(r["Counter_Name"],0)+float(r["Counter_Value"])
        print(tag,kn,{k:"%.4g"%v for k,v in sorted(acc.items())})
